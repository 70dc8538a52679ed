"""Whole-`describe` hipGraph (bonus): since round 6 nothing inside
milan_describe synchronises the stream, so encoder + beam search + rerank can be captured as
ONE graph.  Eager launch vs graph replay of the same call at small neuron counts (where the
~650 launches of a pass could be latency-bound), results compared bit for bit.

    python tools/graph_describe.py [neurons ...]      (run on the GPU box)
"""
import pathlib
import sys
import time

REPO = pathlib.Path(__file__).resolve().parent.parent
for p in (REPO, REPO / 'neuron-descriptions_amd'):
    sys.path.insert(0, str(p))

import torch  # noqa: E402

from milan_amd import hip, synthetic  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1, 8, 64]
    nv = 5000
    sd = synthetic.milan_state_dict(nv + 4, config='resnet101', seed=0)
    dev = torch.device('cuda:0')
    ctx = hip.Context(hip.make_dims(sd, nv), sd, dev)
    ctx.set_precision('split_f16')
    for n in sizes:
        images, masks = synthetic.exemplars(n, k=15, size=224, seed=3, device='cuda:0')
        call = lambda: ctx.describe(images, masks, hip.RERANK, 15, 50, False, 0.2,
                                    group_size=16, check=False)
        want = call()
        torch.cuda.synchronize()
        st = [ctx.status()]
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / reps
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            call()
            torch.cuda.synchronize()
            st.append(ctx.status())
            with torch.cuda.graph(graph, stream=side):
                out = call()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        st.append(ctx.status())
        graph.replay()
        torch.cuda.synchronize()
        st.append(ctx.status())
        same = all(torch.equal(out[k], want[k]) for k in ('tokens', 'scores'))
        t0 = time.perf_counter()
        for _ in range(reps):
            graph.replay()
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / reps
        print(f'{n:4d} neurons: eager {1e3 * eager:8.2f} ms   graph replay {1e3 * replay:8.2f} ms   '
              f'({eager / replay:.3f} x)   status after eager / side-stream eager / capture / first replay / timed replays '
              f'{st + [ctx.status()]}   identical {same}', flush=True)
    ctx.close()


if __name__ == '__main__':
    main()
