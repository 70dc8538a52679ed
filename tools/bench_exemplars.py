#!/usr/bin/env python
"""Throughput of the exemplar-computation kernels against
the HBM roofline, with the CPU oracle timed beside them on a bounded sample.

Synthetic activations of the shapes the reference's hub dissects
(src/exemplars/models.py): resnet152 layer4 (2048 x 7 x 7) and conv1
(64 x 112 x 112), batch 128 / 32.  Stages, each bracketed by HIP events:
  tally.topk     spatial max + running top-k merge   reads the batch once
  tally.sketch   KLL append + compactions            reads the batch once,
                                                     writes it once (level 0)
  render         mask + image + masked per (unit, rank) cell at 224 x 224
Algorithmic bytes: 4 B per activation read per stage (+ 4 B written by the
sketch append); render: 7 output bytes per pixel per cell + the activation map.
    python tools/bench_exemplars.py            # one JSON line per shape
"""
import json
import pathlib
import sys
import time

REPO = pathlib.Path(__file__).resolve().parent.parent
for p in (REPO, REPO / 'neuron-descriptions_amd'):
    sys.path.insert(0, str(p))

import torch  # noqa: E402

from milan_amd import exemplars, hip  # noqa: E402

PEAK_HBM_GBS = 8000.0


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
        enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = hip.require_device('cuda')
    lib = hip.load_library()
    for name, (b, c, h, w), reps in (('resnet152.layer4', (128, 2048, 7, 7),
                                      20),
                                     ('resnet152.layer1', (32, 256, 56, 56),
                                      5),
                                     ('resnet152.conv1', (32, 64, 112, 112),
                                      3)):
        g = torch.Generator(device='cuda').manual_seed(0)
        hid = torch.randn(b, c, h, w, device=dev, generator=g).relu_()
        bytes_in = hid.numel() * 4
        topk = exemplars.RunningTopK(k=15)
        rq = exemplars.RunningQuantile(r=4096)
        t_topk = timed(lambda: topk.add_hiddens(hid), reps)
        torch.manual_seed(0)
        for _ in range(12):  # into the many-level regime (small level-0 buffer)
            rq.add_hiddens(hid)
        t_sketch = timed(lambda: rq.add_hiddens(hid), reps)
        rq.bulk = False  # one launch per append / compaction, for comparison
        t_sketch_per_op = timed(lambda: rq.add_hiddens(hid), min(reps, 3))
        rq.bulk = True
        levels = rq.quantiles(0.99)
        # render: every unit's 15 cells from this batch's first 15 images
        k, size = 15, 224
        n_units = min(c, 256)
        cells = exemplars._Cells(n_units, k, size, dev)
        images = torch.rand(b, 3, 224, 224, device=dev, generator=g)
        todo = []
        for u in range(n_units):
            for r in range(k):
                todo += [r % b, u, u, r]
        mul, add = exemplars.byte_renormalization(None)
        t_render = timed(lambda: cells.render(lib, hid, images, todo,
                                              levels[:n_units], mul, add), 3)
        render_bytes = n_units * k * (7 * size * size + h * w * 4)
        # CPU oracle beside it, bounded sample (one batch, <= 64 units)
        from oracle import exemplars_oracle as E
        cu = min(c, 64)
        hc = hid[:, :cu].cpu()
        torch.set_num_threads(min(32, torch.get_num_threads()))
        t0 = time.perf_counter()
        ot, osk = E.TopK(15), E.QuantileSketch()
        ot.add(hc.view(b, cu, -1).max(dim=2)[0])
        osk.add(hc.permute(0, 2, 3, 1).reshape(-1, cu))
        cpu_s = time.perf_counter() - t0
        line = {
            'workload': f'{name}: hiddens ({b},{c},{h},{w}) fp32, k=15, r=4096',
            'images_per_s_tally': b / ((t_topk + t_sketch) * 1e-3),
            'stages': [
                {'stage': 'tally.topk', 'ms': t_topk, 'bound': 'hbm',
                 'achieved_GBs': bytes_in / t_topk / 1e6,
                 'frac': bytes_in / t_topk / 1e6 / PEAK_HBM_GBS},
                {'stage': 'tally.sketch', 'ms': t_sketch,
                 'bound': 'hbm (batch read once per level-0 range + once for '
                          'the extremes) + LDS sorts; one launch per sketch '
                          'level (milan_exemplar_sketch_add)',
                 'ms_per_operation_path': t_sketch_per_op,
                 'achieved_GBs': 2 * bytes_in / t_sketch / 1e6,
                 'frac': 2 * bytes_in / t_sketch / 1e6 / PEAK_HBM_GBS,
                 'sketch_levels': rq.firstfree},
                {'stage': 'render', 'ms': t_render, 'bound': 'hbm',
                 'cells': n_units * k,
                 'achieved_GBs': render_bytes / t_render / 1e6,
                 'frac': render_bytes / t_render / 1e6 / PEAK_HBM_GBS},
            ],
            'cpu_baseline': {
                'kind': 'port', 'cores': torch.get_num_threads(),
                'sample': f'one batch, {cu} of {c} units, topk + sketch',
                'images_per_s_tally': b / cpu_s * (cu / c),
            },
        }
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
