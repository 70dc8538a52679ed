"""Where does the pinned-host pipeline lose time?  GPU ms inside each describe() (events on the main
stream) against wall time per chunk, with / without MILAN_CHAIN's skip-empty bit (run twice)."""
import pathlib, sys, time, os
import torch
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'neuron-descriptions_amd'))
from milan_amd import hip, synthetic, ingest  # noqa: E402

hip.load_library()
dev = hip.require_device('cuda')
nv = 5000
sd = {k: v.to('cuda') for k, v in synthetic.milan_state_dict(nv + 4, 'resnet101', seed=0).items()}
ctx = hip.Context(hip.make_dims(sd, nv, blocks=synthetic.RESNET_BLOCKS['resnet101']), sd, dev)
ctx.set_precision('split_f16')
N, C = 6, 640
images, masks = synthetic.exemplars(2 * C, k=15, size=224, seed=1, device='cuda')
host = [tuple(t[i * C:(i + 1) * C].cpu().pin_memory() for t in (images, masks)) for i in range(2)]
del images, masks
for mode in ('resident', 'host'):
    dev_data = [tuple(t.cuda() for t in h) for h in host]
    ctx.describe(*dev_data[0], hip.RERANK, 15, 50, False, 0.2, group_size=16, check=False)
    torch.cuda.synchronize()
    evs, walls, keep = [], [], []
    t0 = time.perf_counter()
    if mode == 'resident':
        it = ((dev_data[i % 2]) for i in range(N))
    else:
        it = ingest.ChunkPrefetcher(lambda i: host[i % 2], N, torch.device('cuda'))
    for im, mk in it:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ta = time.perf_counter()
        a.record()
        o = ctx.describe(im, mk, hip.RERANK, 15, 50, False, 0.2, group_size=16, check=False)
        b.record()
        if os.environ.get('D2H'):
            th = torch.empty(o['tokens'].shape, dtype=o['tokens'].dtype, pin_memory=True)
            sh = torch.empty(o['scores'].shape, dtype=o['scores'].dtype, pin_memory=True)
            th.copy_(o['tokens'], non_blocking=True); sh.copy_(o['scores'], non_blocking=True)
            keep.append((th, sh))
        walls.append((ta - t0, time.perf_counter() - t0))
        evs.append((a, b))
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(mode, 'MILAN_CHAIN', os.environ.get('MILAN_CHAIN'), 'total %.1f ms' % (total * 1e3),
          'gpu per chunk', ['%.1f' % a.elapsed_time(b) for a, b in evs],
          'gaps', ['%.1f' % evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(N - 1)],
          'host in/out', ['%.0f-%.0f' % (x * 1e3, y * 1e3) for x, y in walls])
