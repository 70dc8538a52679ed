"""Goldens G16 / G17: netdissect's running statistics, run directly.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_sketch.py

G16 -- the SUBSAMPLING regime of `RunningQuantile`
(src/deps/netdissect/runningstats.py:343-385,409-413,1221-1224): reached once the
sketch cannot add another level (`_next_capacity` < 2), i.e. after ~1e10 samples at the
production r = 4096.  A tiny r gets there within a few hundred samples: the sample rate
halves, every later item still updates the extremes, and a Bernoulli(samplerate) portion
enters the sketch (torch's global CPU generator, seeded here).
G17 -- `state_dict()` of `RunningQuantile` and `RunningTopK` (:118-149, 428-471), the
payload of compute()'s `tally_cache_file`: what the reference writes must load into
milan_amd.exemplars and give the same answers.

Inputs are regenerated from seeds by the tests (a private torch.Generator, so the global
one is consumed only by the code under test).
"""
import json
import pathlib
import sys

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))

import torch  # noqa: E402

import make_golden  # noqa: E402

CASES = {
    # name: r, units, batches, rows per batch, data seed, global RNG seed
    'tiny_r': dict(r=8, units=3, batches=160, rows=50, data_seed=7, rng=11),
    'tiny_r_wide': dict(r=12, units=5, batches=40, rows=333, data_seed=8, rng=12),
}
QS = (0.05, 0.5, 0.95)


def batches(case):
    g = torch.Generator().manual_seed(case['data_seed'])
    for _ in range(case['batches']):
        yield torch.randn(case['rows'], case['units'], generator=g) * 3 + 1


def main():
    torch.set_num_threads(4)
    for n in ('statsmodels', 'statsmodels.stats',
              'statsmodels.stats.correlation_tools'):
        make_golden._stub(n)
    make_golden.import_reference()
    from src.deps.netdissect import runningstats
    out, meta = {}, {}
    for name, case in CASES.items():
        rq = runningstats.RunningQuantile(r=case['r'])
        torch.manual_seed(case['rng'])
        trajectory = []
        for i, batch in enumerate(batches(case)):
            rq.add(batch)
            trajectory.append([float(rq.samplerate), len(rq.data),
                               int(sum(rq.firstfree))])
        out[f'{name}_quantiles'] = torch.stack(
            [rq.quantiles([q])[:, 0].float() for q in QS])
        out[f'{name}_extremes'] = rq.minmax().clone()
        state = rq.state_dict()
        for li, level in enumerate(state['data'][:-1]):
            out[f'{name}_state_level{li}'] = torch.from_numpy(level.copy())
        meta[name] = dict(case, trajectory=trajectory, count=int(rq.count),
                          samplerate=float(rq.samplerate),
                          firstfree=[int(f) for f in rq.firstfree],
                          sizes=[int(d.shape[1]) for d in rq.data],
                          state=dict(resolution=int(state['resolution']),
                                     depth=int(state['depth']),
                                     buffersize=int(state['buffersize']),
                                     samplerate=float(state['samplerate']),
                                     sizes=[int(x) for x in state['sizes']],
                                     size=int(state['size']),
                                     batchcount=int(state['batchcount'])))
        print(name, 'samplerate', rq.samplerate, 'levels', len(rq.data), 'count',
              rq.count)
        assert rq.samplerate < 1.0, 'the case must reach the subsampling regime'
    # G17: RunningTopK state
    g = torch.Generator().manual_seed(21)
    rtk = runningstats.RunningTopK(k=4)
    for _ in range(9):
        rtk.add(torch.randn(37, 6, generator=g))
    values, index = rtk.result()
    state = rtk.state_dict()
    out['topk_values'], out['topk_index'] = values.clone(), index.clone()
    out['topk_state_top_data'] = torch.from_numpy(state['top_data'].copy())
    out['topk_state_top_index'] = torch.from_numpy(state['top_index'].copy())
    out['topk_state_linear_index'] = torch.from_numpy(
        state['linear_index'].copy())
    meta['topk'] = dict(k=int(state['k']), count=int(state['count']),
                        next=int(state['next']),
                        data_shape=[int(x) for x in state['data_shape']])
    torch.save(out, HERE / 'reference_goldens_sketch.pt')
    with open(HERE / 'reference_goldens_sketch.json', 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print('wrote', len(out), 'tensors')


if __name__ == '__main__':
    main()
