"""Goldens G15: the reference's exemplar computation.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_exemplars.py

`src.exemplars.compute.discriminative` / `generative`
(src/exemplars/compute.py:263-437 -> `compute` :27-246 -> netdissect
`tally_topk_and_quantile` (tally.py:199), `RunningTopK` / `RunningQuantile`
(runningstats.py:31,274), `ImageVisualizer` mask / image rendering
(imgviz.py:185-210, ext/netdissect/imgviz.py:56-81)) run here UNMODIFIED on the
tiny `nn.Sequential` models of the reference's own test
(tests/exemplars/compute_test.py:47-63), rebuilt from seeds through
`milan_amd.synthetic.exemplar_model`.  Absent third-party packages
(torchvision, statsmodels, ...) are empty stubs; none of them computes on this
path.

Outputs (tests/golden/reference_goldens_exemplars.{pt,json}): per case the
`images.npy` / `masks.npy` arrays the reference saved (uint8), the ids and
activations of `topk.result()`, the per-unit quantile levels, and the masked
visualisations.  Cases cover: final outputs vs a hooked layer, a unit subset,
output_size != image size, many small batches (RunningTopK buffer
compression), post-ReLU ties, the generative path, and -- with the global torch
RNG seeded, since the KLL sketch draws its random bits from it -- the
randomised regime of RunningQuantile (more than 2*r samples per unit).
"""
import json
import pathlib
import sys
import tempfile

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))

import numpy  # noqa: E402
import torch  # noqa: E402
from torch import nn  # noqa: E402
from torch.utils import data  # noqa: E402

import make_golden  # noqa: E402
from milan_amd import synthetic  # noqa: E402

CASES = {
    # name: model (units, layers, seed, relu), dataset (n, size, seed), call
    'outputs': dict(model=(3, 2, 0, False), images=(10, 16, 1),
                    call=dict(layer=None, k=5, quantile=0.99, output_size=16,
                              batch_size=128), rng=11),
    'layer_q90': dict(model=(3, 2, 0, False), images=(10, 16, 1),
                      call=dict(layer='conv_2', k=5, quantile=0.9,
                                output_size=16, batch_size=128), rng=12),
    'units': dict(model=(4, 2, 2, False), images=(12, 16, 3),
                  call=dict(layer='conv_2', k=5, quantile=0.95, output_size=16,
                            batch_size=5, units=(2, 0)), rng=13),
    'upscale': dict(model=(3, 2, 4, True), images=(9, 16, 5),
                    call=dict(layer='conv_1', k=4, quantile=0.8,
                              output_size=24, batch_size=4), rng=14),
    'compress': dict(model=(6, 2, 6, True), images=(100, 16, 7),
                     call=dict(layer='conv_2', k=7, quantile=0.97,
                               output_size=20, batch_size=8), rng=15),
    # 100 images x 18 x 18 = 32400 samples per unit > 8192: the sketch shifts
    # and decimates with random bits from the global RNG
    'kll': dict(model=(5, 2, 8, True), images=(100, 16, 9),
                call=dict(layer='conv_2', k=6, quantile=0.99, output_size=16,
                          batch_size=16), rng=16),
    'kll_big': dict(model=(4, 1, 10, True), images=(160, 32, 11),
                    call=dict(layer='conv_1', k=3, quantile=0.995,
                              output_size=32, batch_size=32), rng=17),
    'generative': dict(model=(4, 2, 12, False), images=(14, 16, 13),
                       call=dict(layer='conv_2', k=4, quantile=0.9,
                                 output_size=16, batch_size=6), rng=18,
                       generative=True),
}


class FeaturesToImage(nn.Module):
    """tests/exemplars/compute_test.py:246-253."""

    def forward(self, features):
        return torch.sigmoid(features[:, :3])


def main():
    torch.set_num_threads(8)
    for n in ('statsmodels', 'statsmodels.stats',
              'statsmodels.stats.correlation_tools'):
        make_golden._stub(n)
    make_golden.import_reference()
    from src.exemplars import compute
    out, meta = {}, {}
    for name, case in CASES.items():
        units, layers, mseed, relu = case['model']
        model = synthetic.exemplar_model(units, layers, mseed, relu=relu)
        n, size, dseed = case['images']
        dataset = data.TensorDataset(synthetic.exemplar_images(n, size, dseed))
        call = dict(case['call'])
        layer = call.pop('layer')
        fn = compute.discriminative
        if case.get('generative'):
            children = list(model.named_children()) + [('output',
                                                        FeaturesToImage())]
            import collections
            model = nn.Sequential(collections.OrderedDict(children))
            fn = compute.generative
        torch.manual_seed(case['rng'])
        with tempfile.TemporaryDirectory() as tmp:
            tmp = pathlib.Path(tmp)
            kwargs = dict(device='cpu', results_dir=tmp / 'res',
                          viz_dir=tmp / 'viz', display_progress=False,
                          num_workers=0, image_size=size, save_results=True,
                          save_viz=False, masks_cache_file=tmp / 'masks.npz',
                          **call)
            if fn is compute.discriminative:
                topk, rq = fn(model, dataset, layer=layer, **kwargs)
            else:
                topk, rq = fn(model, dataset, layer, **kwargs)
            rdir = tmp / 'res' / (str(layer) if layer is not None else 'outputs')
            out[f'{name}_images'] = torch.from_numpy(
                numpy.load(rdir / 'images.npy'))
            out[f'{name}_masks'] = torch.from_numpy(
                numpy.load(rdir / 'masks.npy'))
            ids_csv = numpy.loadtxt(rdir / 'ids.csv', delimiter=',',
                                    dtype=numpy.int64, ndmin=2)
            act_csv = (rdir / 'activations.csv').read_text()
            # the cached gather grid also holds the masked visualisations
            grid = numpy.load(tmp / 'masks.npz', allow_pickle=True)['grid']
            out[f'{name}_masked'] = torch.from_numpy(grid[:, :, :3].copy())
        k = call['k']
        acts, ids = topk.result()
        sel = sorted(call['units']) if 'units' in call else slice(None)
        acts, ids = acts[:, :k], ids[:, :k]
        assert (ids.numpy() == ids_csv).all()
        out[f'{name}_ids'] = ids.clone()
        out[f'{name}_activations'] = acts.clone()
        out[f'{name}_levels'] = rq.quantiles(call['quantile']).reshape(
            -1).clone()
        meta[name] = dict(case, activations_csv=act_csv,
                          sketch=dict(count=int(rq.count),
                                      firstfree=[int(f) for f in rq.firstfree],
                                      sizes=[int(d.shape[1]) for d in rq.data],
                                      samplerate=float(rq.samplerate)))
        del sel
        print(name, 'images', tuple(out[f'{name}_images'].shape), 'mask cover',
              float(out[f'{name}_masks'].float().mean()), 'sketch levels',
              meta[name]['sketch']['firstfree'])
    torch.save(out, HERE / 'reference_goldens_exemplars.pt')
    with open(HERE / 'reference_goldens_exemplars.json', 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    total = sum(t.numel() * t.element_size() for t in out.values())
    print(f'wrote {len(out)} tensors, {total / 1e6:.2f} MB')


if __name__ == '__main__':
    main()
