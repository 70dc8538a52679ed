"""hipGraph capture/replay of the decode stage gives identical results."""
import time

import pytest
import torch

from milan_amd import hip, synthetic

pytestmark = pytest.mark.gpu


def test_decode_graph_replay_is_bit_identical_and_used():
    nv = 5000
    sd = synthetic.decoder_state_dict(nv + 4, seed=0)
    ctx = hip.Context(hip.make_dims(sd, nv), sd, 'cuda')
    g = torch.Generator().manual_seed(1)
    feats = [torch.rand(8, 15, 3904, generator=g).cuda() for _ in range(3)]
    plain = [ctx.decode(f, hip.RERANK, 15, 16, False, 0.2) for f in feats]
    ctx.enable_graphs(True)
    # call 1: direct, call 2: capture + launch, call 3+: replay
    outs = [ctx.decode(f, hip.RERANK, 15, 16, False, 0.2) for f in feats]
    outs += [ctx.decode(feats[0], hip.RERANK, 15, 16, False, 0.2)]
    torch.cuda.synchronize()
    captures, replays = ctx.graph_stats()
    assert captures == 1 and replays == 3
    for got, want in zip(outs, plain + [plain[0]]):
        for key in ('tokens', 'scores', 'beam_tokens', 'beam_scores', 'out_len'):
            assert torch.equal(got[key], want[key]), key
    # a different shape gets its own graph
    small = torch.rand(2, 15, 3904, generator=g).cuda()
    for _ in range(3):
        ctx.decode(small, hip.GREEDY, 15, 1, False, 0.2)
    assert ctx.graph_stats()[0] == 2

    def bench(fn, reps=10):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps

    t_graph = bench(lambda: ctx.decode(feats[0], hip.RERANK, 15, 16, False, 0.2))
    ctx.enable_graphs(False)
    t_plain = bench(lambda: ctx.decode(feats[0], hip.RERANK, 15, 16, False, 0.2))
    print(f'decode n=8 beam=16 rerank: graph {t_graph*1e3:.2f} ms, '
          f'plain {t_plain*1e3:.2f} ms')
    ctx.close()
