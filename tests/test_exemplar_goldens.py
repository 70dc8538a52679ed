"""CPU: the exemplar oracle against goldens G15 (the reference's own
`exemplars.compute.discriminative` / `generative`, run unmodified in
tests/golden/make_golden_exemplars.py).  Bar: bit-exact uint8 images / masks /
masked visualisations, identical top-k ids and activations, identical quantile
levels -- including the randomised KLL regime (same global-RNG stream)."""
import collections
import json

import pytest
import torch
from torch import nn
from torch.utils import data

from conftest import GOLDEN_DIR
from milan_amd import synthetic
from oracle import exemplars_oracle as E


@pytest.fixture(scope='module')
def G():
    return torch.load(GOLDEN_DIR / 'reference_goldens_exemplars.pt')


@pytest.fixture(scope='module')
def M():
    with open(GOLDEN_DIR / 'reference_goldens_exemplars.json') as f:
        return json.load(f)


CASES = ['outputs', 'layer_q90', 'units', 'upscale', 'compress', 'kll',
         'kll_big', 'generative']


class FeaturesToImage(nn.Module):

    def forward(self, features):
        return torch.sigmoid(features[:, :3])


def build(case):
    units, layers, mseed, relu = case['model']
    model = synthetic.exemplar_model(units, layers, mseed, relu=relu)
    if case.get('generative'):
        children = list(model.named_children()) + [('output',
                                                    FeaturesToImage())]
        model = nn.Sequential(collections.OrderedDict(children))
    n, size, dseed = case['images']
    return model, data.TensorDataset(synthetic.exemplar_images(n, size, dseed))


def check(got, G, name):
    assert torch.equal(got['ids'], G[f'{name}_ids']), 'top-k image ids'
    assert torch.equal(got['activations'], G[f'{name}_activations'])
    assert torch.equal(got['levels'], G[f'{name}_levels']), 'quantile levels'
    for key in ('images', 'masks', 'masked'):
        want = G[f'{name}_{key}']
        assert got[key].dtype == torch.uint8 and got[key].shape == want.shape
        assert torch.equal(got[key].cpu(), want), f'{key} differ in ' \
            f'{int((got[key].cpu() != want).sum())} bytes'


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_exemplars(G, M, name):
    case = M[name]
    model, dataset = build(case)
    call = dict(case['call'])
    layer = call.pop('layer')
    torch.manual_seed(case['rng'])  # DataLoader base seeds + KLL random bits
    fn = E.generative if case.get('generative') else E.discriminative
    got = fn(model, dataset, layer, **call)
    check(got, G, name)
    if 'kll' in name or name == 'compress':
        assert len(case['sketch']['firstfree']) > 1  # really randomised


def test_sketch_state_matches_reference(M):
    """The sketch's level occupancy after the tally equals the reference's."""
    case = M['kll_big']
    model, dataset = build(case)
    torch.manual_seed(case['rng'])
    sketch = E.QuantileSketch()
    for (images,) in data.DataLoader(dataset,
                                     batch_size=case['call']['batch_size']):
        with torch.no_grad():
            h = model.conv_1(images)
        sketch.add(h.permute(0, 2, 3, 1).reshape(-1, h.shape[1]))
    assert sketch.firstfree == case['sketch']['firstfree']
    assert [d.shape[1] for d in sketch.data] == case['sketch']['sizes']
    assert sketch.count == case['sketch']['count']
