"""GPU: the exemplar-computation kernels (csrc/exemplars.hip, through the C ABI
and the `milan_amd.exemplars` mirror of src/exemplars/compute.py) against
goldens G15 -- the reference's own `discriminative` / `generative` -- and
against the CPU oracle on larger random cases.

Bar: `images.npy` / `masks.npy` / masked visualisations bit-exact uint8, top-k
ids and activations identical, quantile levels identical (float32), including
the randomised KLL regime under the same torch seed.  To compare the KERNELS
and not two convolution libraries, the dissected model runs on the CPU in these
tests (as it did when the goldens were made) and its activations are handed to
the GPU; one end-to-end test runs the model on the GPU too."""
import collections
import json

import numpy
import pytest
import torch
from torch import nn
from torch.utils import data

from conftest import GOLDEN_DIR
from milan_amd import exemplars, hip, synthetic
from oracle import exemplars_oracle as E
from test_exemplar_goldens import CASES, FeaturesToImage, build, check

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def G():
    return torch.load(GOLDEN_DIR / 'reference_goldens_exemplars.pt')


@pytest.fixture(scope='module')
def M():
    with open(GOLDEN_DIR / 'reference_goldens_exemplars.json') as f:
        return json.load(f)


def cpu_model_callbacks(model, layer, generative=False):
    """The model on the CPU (bit-identical activations to the reference's
    run), activations shipped to the GPU."""

    def run(images):
        x, hid = images.cpu(), None
        with torch.no_grad():
            for name, child in model.named_children():
                x = child(x)
                if name == layer:
                    hid = x
                    if not generative:
                        break
        hid = x if hid is None else hid
        return hid.cuda(), x.cuda()

    if generative:
        return (lambda images: run(images)[0]), run
    return (lambda images: run(images)[0]), (lambda images: run(images)[0])


def run_case(case, tmp_path, reference_contract=False):
    model, dataset = build(case)
    call = dict(case['call'])
    layer = call.pop('layer')
    tally, acts = cpu_model_callbacks(model, layer,
                                      bool(case.get('generative')))
    if reference_contract:
        # the reference's callback contract: (pooled, activations (N, C))
        def tally(images, _inner=tally):  # noqa: E306
            h = _inner(images)
            b, c = h.shape[:2]
            return (h.view(b, c, -1).max(dim=2)[0],
                    h.permute(0, 2, 3, 1).reshape(-1, c))
    torch.manual_seed(case['rng'])
    topk, rq = exemplars.compute(tally, acts, dataset, results_dir=tmp_path,
                                 image_size=case['images'][1],
                                 num_workers=0, save_viz=False,
                                 display_progress=False, **call)
    values, ids = topk.result()
    got = dict(images=torch.from_numpy(numpy.load(tmp_path / 'images.npy')),
               masks=torch.from_numpy(numpy.load(tmp_path / 'masks.npy')),
               masked=topk.cells.masked.cpu(), ids=ids.cpu(),
               activations=values.cpu(),
               levels=rq.quantiles(call['quantile']).cpu())
    return got, topk, rq


@pytest.mark.parametrize('name', CASES)
def test_kernels_match_reference_exemplars(G, M, name, tmp_path):
    hip.require_device('cuda')
    got, topk, rq = run_case(M[name], tmp_path)
    check(got, G, name)
    assert rq.firstfree == M[name]['sketch']['firstfree']
    assert [d.shape[1] for d in rq.data] == M[name]['sketch']['sizes']
    assert rq.count == M[name]['sketch']['count']
    # the files the reference writes
    ids_csv = numpy.loadtxt(tmp_path / 'ids.csv', delimiter=',',
                            dtype=numpy.int64, ndmin=2)
    assert (ids_csv == G[f'{name}_ids'].numpy()).all()
    assert (tmp_path / 'activations.csv').read_text() == \
        M[name]['activations_csv']
    if 'units' in M[name]['call']:
        assert numpy.load(tmp_path / 'units.npy').tolist() == sorted(
            M[name]['call']['units'])


@pytest.mark.parametrize('name', ['layer_q90', 'kll'])
def test_reference_callback_contract(G, M, name, tmp_path):
    """`compute` also takes the reference's (pooled, activations) pair."""
    got, _, _ = run_case(M[name], tmp_path, reference_contract=True)
    check(got, G, name)


@pytest.mark.parametrize('seed,units,n,size,k,batch,out', [
    (1, 64, 300, 24, 15, 32, 56),   # 300 x 26 x 26 samples: deep KLL regime
    (2, 33, 70, 16, 9, 128, 224),   # one batch, production output size
    (3, 8, 40, 40, 40, 7, 33),      # k == dataset size, ragged batches
])
def test_kernels_match_oracle_on_random_cases(seed, units, n, size, k, batch,
                                              out, tmp_path):
    hip.require_device('cuda')
    model = synthetic.exemplar_model(units, 2, seed, relu=True)
    dataset = data.TensorDataset(synthetic.exemplar_images(n, size, seed + 50))
    kwargs = dict(k=k, quantile=0.99, output_size=out, batch_size=batch)
    torch.manual_seed(seed)
    want = E.discriminative(model, dataset, 'conv_2', **kwargs)
    tally, acts = cpu_model_callbacks(model, 'conv_2')
    torch.manual_seed(seed)
    topk, rq = exemplars.compute(tally, acts, dataset, results_dir=tmp_path,
                                 image_size=size, num_workers=0,
                                 save_viz=False, **kwargs)
    values, ids = topk.result()
    assert torch.equal(ids.cpu(), want['ids'])
    assert torch.equal(values.cpu(), want['activations'])
    assert torch.equal(rq.quantiles(0.99).cpu(), want['levels'])
    for key in ('images', 'masks', 'masked'):
        assert torch.equal(getattr(topk.cells, key).cpu(), want[key]), key


def test_discriminative_end_to_end_on_gpu(M, G, tmp_path):
    """The drop-in call with the model on the GPU: same files, same top
    images; masks may differ only where MIOpen's convolution rounds
    differently from the CPU's right at the threshold."""
    case = M['layer_q90']
    model, dataset = build(case)
    call = dict(case['call'])
    layer = call.pop('layer')
    torch.manual_seed(case['rng'])
    exemplars.discriminative(model, dataset, layer=layer, device='cuda',
                             results_dir=tmp_path, viz_dir=tmp_path / 'viz',
                             image_size=16, num_workers=0, save_viz=True,
                             **call)
    out = tmp_path / layer
    images = torch.from_numpy(numpy.load(out / 'images.npy'))
    masks = torch.from_numpy(numpy.load(out / 'masks.npy'))
    assert torch.equal(images, G['layer_q90_images'])
    assert (masks != G['layer_q90_masks']).float().mean() < 1e-3
    assert (tmp_path / 'viz' / layer / 'unit_0' / 'image_0.png').is_file()
    with pytest.raises(ValueError, match='k >= 1'):
        exemplars.compute(None, None, dataset, k=0, image_size=16)
    with pytest.raises(ValueError, match='quantile in range'):
        exemplars.compute(None, None, dataset, quantile=2, image_size=16)
    with pytest.raises(ValueError, match='image_size= must be set'):
        exemplars.compute(None, None, dataset)


@pytest.mark.parametrize('r,batch,channels,side,adds,subset', [
    (64, 8, 16, 28, 6, False),     # 128-bit random pool: refills mid-batch
    (3 * 1024, 4, 8, 56, 5, True),  # the class default, unit subset
    (4096, 16, 4, 112, 3, False),  # compute()'s sketch on a conv1-sized map
])
def test_bulk_sketch_add_equals_per_operation_path(r, batch, channels, side,
                                                   adds, subset):
    """milan_exemplar_sketch_add (state machine played forward, one launch per
    level) leaves the sketch in the state the append / compact sequence does:
    same buffers in the same order, extremes, random-bit cursor, quantiles."""
    hip.require_device('cuda')
    units = (torch.tensor([5, 0, 3], dtype=torch.int32, device='cuda')
             if subset else None)
    sketches = []
    for bulk in (True, False):
        torch.manual_seed(11)
        rq = exemplars.RunningQuantile(r=r)
        rq.bulk = bulk
        g = torch.Generator().manual_seed(4)
        for i in range(adds):
            b = batch if i != 1 else max(1, batch // 3)  # a ragged one
            h = torch.randn(b, channels, side, side, generator=g).cuda()
            h[0, :, 0, 0] = 7.0 + i          # ties and moving extremes
            rq.add_hiddens(h, units)
        sketches.append(rq)
    fast, slow = sketches
    assert fast.firstfree == slow.firstfree
    assert fast.currentbit == slow.currentbit and fast.count == slow.count
    assert torch.equal(fast.randbits, slow.randbits)
    assert [d.shape for d in fast.data] == [d.shape for d in slow.data]
    assert len(fast.data) > 1
    for a, b, n in zip(fast.data, slow.data, fast.firstfree):
        assert torch.equal(a[:, :n], b[:, :n])
    assert torch.equal(fast.extremes, slow.extremes)
    for q in (0.5, 0.99):
        assert torch.equal(fast.quantiles(q), slow.quantiles(q))


N_FUZZ = max(6, int(__import__('os').environ.get('MILAN_FUZZ_SEEDS', '6')) // 4)


@pytest.mark.parametrize('seed', range(N_FUZZ))
def test_fuzz_exemplars_against_oracle(seed, tmp_path):
    """Random geometries through compute(): unit counts, odd image sizes, k
    above / below the batch size, ragged last batches, output sizes, unit
    subsets, quantiles -- top-k ids / values, quantile levels and all three
    uint8 outputs equal the oracle's.  The oracle sorts the quantile summary
    stably here: the reference's `torch.sort` leaves the order of equal samples
    that sit on different sketch levels (post-ReLU zeros) unspecified, the
    kernel keeps level order (test_quantile_ties_keep_level_order)."""
    import random
    hip.require_device('cuda')
    r = random.Random(7000 + seed)
    units = r.choice([3, 8, 17, 40, 96])
    size = r.choice([9, 16, 23, 32, 47])
    n = r.randint(5, 90)
    k = r.randint(1, min(n, 20))
    batch = r.choice([1, 4, 7, 32, 128])
    out = r.choice([8, 31, 64, 224])
    quantile = r.choice([0.5, 0.9, 0.99, 0.999])
    layers = r.choice([1, 2, 3])
    subset = sorted(r.sample(range(units), r.randint(1, units))) \
        if r.random() < 0.3 else None
    model = synthetic.exemplar_model(units, layers, seed, relu=r.random() < 0.7)
    dataset = data.TensorDataset(synthetic.exemplar_images(n, size, seed + 90))
    layer = f'conv_{layers}'
    kwargs = dict(k=k, quantile=quantile, output_size=out, batch_size=batch)
    if subset is not None:
        kwargs['units'] = subset
    torch.manual_seed(seed)
    want = E.discriminative(model, dataset, layer, stable_ties=True, **kwargs)
    tally, acts = cpu_model_callbacks(model, layer)
    torch.manual_seed(seed)
    topk, rq = exemplars.compute(tally, acts, dataset, results_dir=tmp_path,
                                 image_size=size, num_workers=0,
                                 save_viz=False, **kwargs)
    values, ids = topk.result()
    assert torch.equal(ids.cpu(), want['ids'])
    assert torch.equal(values.cpu(), want['activations'])
    assert torch.equal(rq.quantiles(quantile).cpu(), want['levels'])
    for key in ('images', 'masks', 'masked'):
        assert torch.equal(getattr(topk.cells, key).cpu(), want[key]), key


def test_quantile_ties_keep_level_order(tmp_path):
    """Found by the 3000-seed campaign (2 of 750 exemplar cases, one unit each):
    with post-ReLU zeros on several sketch levels, the reference's quantile at
    the edge of the run of zeros depends on how its UNSTABLE `torch.sort`
    happens to order equal keys (CPU introsort; CUDA differs again).  The kernel
    sorts stably -- equal samples stay in level order -- and equals the oracle
    run the same way on every unit; the oracle with the reference's unstable
    call differs from that on a handful of units at most."""
    hip.require_device('cuda')
    seed, units, size, n, k, batch, out, quantile = 564, 96, 16, 37, 10, 128, 224, 0.999
    model = synthetic.exemplar_model(units, 1, seed, relu=True)
    dataset = data.TensorDataset(synthetic.exemplar_images(n, size, seed + 90))
    kwargs = dict(k=k, quantile=quantile, output_size=out, batch_size=batch)
    levels = {}
    for stable in (True, False):
        torch.manual_seed(seed)
        levels[stable] = E.discriminative(model, dataset, 'conv_1',
                                          stable_ties=stable, **kwargs)['levels']
    tally, acts = cpu_model_callbacks(model, 'conv_1')
    torch.manual_seed(seed)
    _, rq = exemplars.compute(tally, acts, dataset, results_dir=tmp_path,
                              image_size=size, num_workers=0, save_viz=False,
                              **kwargs)
    got = rq.quantiles(quantile).cpu()
    assert torch.equal(got, levels[True])
    assert int((levels[False] != levels[True]).sum()) <= 4


def test_units_are_bounds_checked_like_the_reference(tmp_path):
    """The reference selects `pooled[:, units]` (src/exemplars/compute.py:331-333):
    a negative unit counts from the end, anything else out of range is an IndexError.
    The kernels index `units[u]` unchecked, so the host side must decide this."""
    hip.require_device('cuda')
    model = synthetic.exemplar_model(8, 2, 5, relu=True)
    dataset = data.TensorDataset(synthetic.exemplar_images(24, 16, 55))
    tally, acts = cpu_model_callbacks(model, 'conv_2')
    kwargs = dict(k=4, quantile=0.99, output_size=16, batch_size=8, image_size=16,
                  num_workers=0, save_viz=False)
    with pytest.raises(IndexError):
        exemplars.compute(tally, acts, dataset, results_dir=tmp_path / 'a',
                          units=[1, 8], **kwargs)
    torch.manual_seed(0)
    neg, _ = exemplars.compute(tally, acts, dataset, results_dir=tmp_path / 'b',
                               units=[-1, 2], **kwargs)
    torch.manual_seed(0)
    pos, _ = exemplars.compute(tally, acts, dataset, results_dir=tmp_path / 'c',
                               units=[2, 7], **kwargs)
    # sorted([-1, 2]) = [-1, 2] -> channels (7, 2); sorted([2, 7]) -> (2, 7)
    v_neg, i_neg = neg.result()
    v_pos, i_pos = pos.result()
    assert torch.equal(v_neg[0], v_pos[1]) and torch.equal(v_neg[1], v_pos[0])
    assert torch.equal(i_neg[0], i_pos[1]) and torch.equal(i_neg[1], i_pos[0])
    # the running statistics validate a caller-supplied unit tensor as well
    topk = exemplars.RunningTopK(k=3)
    hiddens = torch.rand(4, 8, 5, 5, device='cuda')
    with pytest.raises(IndexError):
        topk.add_hiddens(hiddens, torch.tensor([0, 9], dtype=torch.int32,
                                               device='cuda'))


# ---------------------------------------------------------------------------
# G16 / G17: the sketch's subsampling regime, state dicts, cache files
# ---------------------------------------------------------------------------
@pytest.fixture(scope='module')
def S():
    return (torch.load(GOLDEN_DIR / 'reference_goldens_sketch.pt'),
            json.loads((GOLDEN_DIR / 'reference_goldens_sketch.json').read_text()))


def _sketch_batches(case):
    g = torch.Generator().manual_seed(case['data_seed'])
    for _ in range(case['batches']):
        yield torch.randn(case['rows'], case['units'], generator=g) * 3 + 1


@pytest.mark.parametrize('name', ['tiny_r', 'tiny_r_wide'])
def test_sketch_subsampling_regime_matches_reference(S, name):
    """netdissect's RunningQuantile with a tiny r reaches `samplerate < 1` after a few
    hundred samples (runningstats.py:343-385): the HIP sketch must follow it state for
    state -- same level fills after every batch, same quantiles, bit for bit (seeded:
    both draw the compaction bits and the Bernoulli masks from torch's global RNG)."""
    hip.require_device('cuda')
    tensors, meta = S
    case = meta[name]
    sketch = exemplars.RunningQuantile(r=case['r'])
    torch.manual_seed(case['rng'])
    for i, batch in enumerate(_sketch_batches(case)):
        sketch.add(batch.cuda())
        rate, levels, held = case['trajectory'][i]
        assert (sketch.samplerate, len(sketch.data), sum(sketch.firstfree)) == \
            (rate, levels, held), i
    assert sketch.samplerate == case['samplerate'] < 1.0
    assert sketch.count == case['count']
    got = torch.stack([sketch.quantiles(q).cpu() for q in (0.05, 0.5, 0.95)])
    assert torch.equal(got, tensors[f'{name}_quantiles'])


def test_reference_state_dicts_load(S):
    """What the reference's tally cache holds (`state_dict()` of both running
    statistics, runningstats.py:118-149,428-471) loads here and answers alike; and our
    own state dicts round-trip."""
    hip.require_device('cuda')
    tensors, meta = S
    case = meta['tiny_r']
    levels = [tensors[f'tiny_r_state_level{i}'].numpy()
              for i in range(len(case['state']['sizes']))]
    state = dict(case['state'], data=levels + [None],
                 extremes=tensors['tiny_r_extremes'].numpy())
    rq = exemplars.RunningQuantile()
    rq.set_state_dict(state)
    got = torch.stack([rq.quantiles(q).cpu() for q in (0.05, 0.5, 0.95)])
    assert torch.equal(got, tensors['tiny_r_quantiles'])
    again = exemplars.RunningQuantile()
    again.set_state_dict(rq.state_dict())
    assert torch.equal(again.quantiles(0.5).cpu(), tensors['tiny_r_quantiles'][1])
    assert again.count == case['count'] and again.samplerate == case['samplerate']
    t = meta['topk']
    topk = exemplars.RunningTopK(k=t['k'])
    topk.set_state_dict(dict(k=t['k'], count=t['count'], next=t['next'],
                             top_data=tensors['topk_state_top_data'].numpy(),
                             top_index=tensors['topk_state_top_index'].numpy(),
                             linear_index=tensors['topk_state_linear_index'].numpy()))
    values, index = topk.result()
    assert torch.equal(values.cpu(), tensors['topk_values'])
    assert torch.equal(index.cpu(), tensors['topk_index'])
    twin = exemplars.RunningTopK(k=t['k'])
    twin.set_state_dict(topk.state_dict())
    assert torch.equal(twin.result()[1].cpu(), tensors['topk_index'])


def test_cache_files_replace_the_passes_over_the_dataset(tmp_path):
    """compute.py:140-146 / tally.py:199-222: with `tally_cache_file` and
    `masks_cache_file` in place a second call must not read the dataset at all, and
    `clear_cache_files` makes it read again."""
    hip.require_device('cuda')

    class Counting(data.Dataset):
        def __init__(self, images):
            self.images, self.reads = images, 0

        def __len__(self):
            return len(self.images)

        def __getitem__(self, i):
            self.reads += 1
            return (self.images[i],)

    model = synthetic.exemplar_model(6, 2, 4, relu=True)
    dataset = Counting(synthetic.exemplar_images(40, 16, 77))
    tally, acts = cpu_model_callbacks(model, 'conv_2')
    kwargs = dict(k=5, quantile=0.9, output_size=16, batch_size=16, image_size=16,
                  num_workers=0, save_viz=False,
                  tally_cache_file=tmp_path / 'cache' / 'tally.npz',
                  masks_cache_file=tmp_path / 'cache' / 'masks.npz')
    first, rq1 = exemplars.compute(tally, acts, dataset, results_dir=tmp_path / 'a',
                                   **kwargs)
    reads = dataset.reads
    assert reads >= 40 and (tmp_path / 'cache' / 'tally.npz').exists()
    second, rq2 = exemplars.compute(tally, acts, dataset, results_dir=tmp_path / 'b',
                                    **kwargs)
    assert dataset.reads == reads, 'the cached call went back to the dataset'
    assert torch.equal(first.result()[1], second.result()[1])
    assert torch.equal(rq1.quantiles(0.9), rq2.quantiles(0.9))
    for name in ('images', 'masks'):
        a = numpy.load(tmp_path / 'a' / f'{name}.npy')
        b = numpy.load(tmp_path / 'b' / f'{name}.npy')
        assert (a == b).all()
    # the tally file follows netdissect's key layout
    keys = set(numpy.load(tmp_path / 'cache' / 'tally.npz', allow_pickle=True).keys())
    assert {'rtk.top_data', 'rq.data', 'rq.sizes', 'k', 'r', 'sample_size'} <= keys
    # a different k invalidates the cache; clear_cache_files removes it
    exemplars.compute(tally, acts, dataset, results_dir=tmp_path / 'c',
                      **dict(kwargs, k=4))
    assert dataset.reads > reads
    reads = dataset.reads
    exemplars.compute(tally, acts, dataset, results_dir=tmp_path / 'd',
                      clear_cache_files=True, **kwargs)
    assert dataset.reads > reads


def test_tally_cache_is_keyed_on_the_unit_list(tmp_path):
    """a cached tally of ANOTHER unit list of the same length, or a subset
    tally offered to an all-units run, must be recomputed, not adopted."""
    hip.require_device('cuda')
    model = synthetic.exemplar_model(6, 2, 4, relu=True)
    dataset = synthetic.exemplar_images(24, 16, 78)
    tally, acts = cpu_model_callbacks(model, 'conv_2')
    kwargs = dict(k=4, quantile=0.9, output_size=16, batch_size=8, image_size=16,
                  num_workers=0, save_viz=False, save_results=False,
                  tally_cache_file=tmp_path / 'tally.npz')
    ds = data.TensorDataset(dataset)
    a, _ = exemplars.compute(tally, acts, ds, units=[0, 3], **kwargs)
    b, _ = exemplars.compute(tally, acts, ds, units=[1, 2], **kwargs)   # same length
    fresh, _ = exemplars.compute(tally, acts, ds, units=[1, 2],
                                 **dict(kwargs, tally_cache_file=None))
    assert torch.equal(b.result()[0], fresh.result()[0])
    assert not torch.equal(a.result()[0], b.result()[0])
    every, _ = exemplars.compute(tally, acts, ds, **kwargs)              # all units
    assert every.result()[0].shape[0] == 6
    again, _ = exemplars.compute(tally, acts, ds, **kwargs)              # adopted now
    assert torch.equal(every.result()[0], again.result()[0])
