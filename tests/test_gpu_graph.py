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


def test_whole_describe_is_capturable_and_replays_identically():
    """Round 6: nothing inside milan_describe synchronises the stream or reads anything back
    (the live-image count stays on the device), so encoder + beam search + LM rerank can be
    captured as ONE hipGraph by the caller (here: torch.cuda.graph) and replayed.  Several
    replays, each compared bit for bit with the eager call -- the first builds of this path
    zero-filled through hipMemsetAsync, whose graph node replayed a recycled 16-byte pattern
    from the SECOND replay on (the feature rows of empty-mask exemplars, the LM's initial
    state): launch_zero_fill is a kernel for that reason."""
    nv, k = 1000, 15
    blocks = synthetic.RESNET_BLOCKS['resnet50']
    sd = synthetic.milan_state_dict(nv + 4, config='resnet50', seed=3)
    ctx = hip.Context(hip.make_dims(sd, nv, blocks=blocks), sd, 'cuda')
    ctx.set_precision('split_f16')
    images, masks = synthetic.exemplars(8, k=k, size=96, seed=7, device='cuda')
    masks = masks.clone()
    masks[1, 3] = 0
    masks[6, 14] = 0                      # two exemplars without work, 1.9 MB of feature rows
    call = lambda: ctx.describe(images, masks, hip.RERANK, 15, 16, False, 0.2,
                                group_size=16, check=False, want_features=True)
    want = call()
    torch.cuda.synchronize()
    assert ctx.status() == 0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        call()
        with torch.cuda.graph(graph, stream=side):
            out = call()
    torch.cuda.current_stream().wait_stream(side)
    for replay in range(4):
        if replay == 2:                   # stale outputs must not survive a replay
            for key in ('tokens', 'scores', 'beam_tokens', 'beam_scores', 'features'):
                out[key].fill_(7)
        graph.replay()
        torch.cuda.synchronize()
        for key in ('tokens', 'scores', 'beam_tokens', 'beam_scores', 'out_len', 'features'):
            assert torch.equal(out[key], want[key]), (replay, key)
        assert (out['features'][1, 3] == 0).all() and (out['features'][6, 14] == 0).all()
        assert ctx.status() == 0, replay
    ctx.close()
