"""ChunkPrefetcher (host -> HBM ingest overlapped with compute).

Round 5 moved the request for chunk i+1's H2D copy in FRONT of handing out chunk i (a
consumer that synchronises inside its call -- the encoder's 4-byte read-back -- had put
its result D2H ahead of that copy, profiles/r5_experiments.txt J).  What must hold
whatever the timing: chunks arrive in order with the bytes of their fetch, a slow fetch is
never waited for before the current chunk is handed out, a consumer may synchronise,
and an exception in the worker surfaces in the consumer.
"""
import time

import pytest
import torch

from milan_amd import hip, ingest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    hip.load_library()
    return hip.require_device('cuda')


def _chunk(i, n=4096):
    g = torch.Generator().manual_seed(100 + i)
    im = torch.randint(0, 256, (n, 3, 8, 8), dtype=torch.uint8, generator=g)
    mk = torch.randint(0, 2, (n, 1, 8, 8), dtype=torch.uint8, generator=g)
    return im, mk


@pytest.mark.parametrize('pinned', [True, False])
@pytest.mark.parametrize('consumer', ['async', 'sync', 'd2h'])
def test_chunks_arrive_in_order_with_their_bytes(dev, pinned, consumer):
    n_chunks = 7
    host = [_chunk(i) for i in range(n_chunks)]
    if pinned:
        host = [tuple(t.pin_memory() for t in c) for c in host]
    outs, keep = [], []
    for i, (im, mk) in enumerate(ingest.ChunkPrefetcher(lambda j: host[j], n_chunks, dev)):
        assert im.device.type == 'cuda' and mk.device.type == 'cuda'
        s = im.to(torch.int64).sum() * 3 + mk.to(torch.int64).sum()   # "compute" on the main stream
        if consumer == 'sync':
            torch.cuda.synchronize()          # a consumer that synchronises inside its call
        elif consumer == 'd2h':
            h = torch.empty((), dtype=torch.int64, pin_memory=True)
            h.copy_(s, non_blocking=True)     # ... or queues a result copy behind its compute
            keep.append(h)
        outs.append(s)
    torch.cuda.synchronize()
    want = [int(im.to(torch.int64).sum() * 3 + mk.to(torch.int64).sum()) for im, mk in host]
    assert [int(o) for o in outs] == want
    if consumer == 'd2h':
        assert [int(h) for h in keep] == want


def test_a_slow_fetch_is_not_waited_for_before_the_current_chunk(dev):
    """fetch(i >= 1) takes 0.3 s: chunk 0 must be handed out long before chunk 1 is staged."""
    host = [_chunk(i, 256) for i in range(3)]
    t0 = time.perf_counter()

    def fetch(i):
        if i >= 1:
            time.sleep(0.3)
        return host[i]

    seen = []
    for i, (im, mk) in enumerate(ingest.ChunkPrefetcher(fetch, 3, dev)):
        seen.append((time.perf_counter() - t0, int(im.to(torch.int64).sum())))
    assert seen[0][0] < 0.25, seen          # not held back by the lookahead
    assert [s for _, s in seen] == [int(c[0].to(torch.int64).sum()) for c in host]


def test_masks_may_be_absent_and_fetch_errors_surface(dev):
    host = [_chunk(i, 128) for i in range(3)]
    got = [(im.shape, mk) for im, mk in ingest.ChunkPrefetcher(lambda i: (host[i][0], None), 3, dev)]
    assert all(mk is None for _, mk in got)

    def bad(i):
        if i == 1:
            raise RuntimeError('disk on fire')
        return host[i]

    with pytest.raises(RuntimeError, match='disk on fire'):
        for _ in ingest.ChunkPrefetcher(bad, 3, dev):
            pass
