"""GPU: the HIP beam search / rerank (through the C ABI) against goldens G14 =
the reference's own `Decoder.forward(strategy='beam'|'rerank')` / `predict`
(tests/golden/make_golden_beam.py).  Near-tie excuses are counted and bounded
(tests/beamcheck.py)."""
import pytest
import torch
from torch.utils import data

from beamcheck import check_search
from milan_amd import decoders, encoders, hip, lang, lms
from milan_amd import synthetic
from test_beam_goldens import G, M, SEARCHES, model  # noqa: F401 (fixtures)

pytestmark = pytest.mark.gpu

STRATEGY = {'beam': hip.BEAM, 'rerank': hip.RERANK}


@pytest.fixture(scope='module')
def dev():
    hip.load_library()
    return hip.require_device('cuda')


def hip_forward(ctx, feats, m, group_size=0):
    mi = m['mi']
    if mi is None:  # reference default: MI unless reranking (decoders.py:385)
        mi = m['strategy'] != 'rerank'
    out = ctx.decode(feats, STRATEGY[m['strategy']], m['length'], m['beam'], mi,
                     m['temperature'], group_size=group_size)
    out['tprime'] = int(out['out_len'].max())
    return out


@pytest.mark.parametrize('precision', ['f32', 'split_f16'])
@pytest.mark.parametrize('model_key,tag', SEARCHES)
def test_hip_matches_reference_beam_and_rerank(dev, G, M, model_key, tag,
                                               precision):
    sd, feats, nv = model(M[model_key])
    ctx = hip.Context(hip.make_dims(sd, nv), sd, dev)
    ctx.set_precision(precision)
    m = M[tag]
    got = hip_forward(ctx, feats, m)
    excuses = check_search(got, G, tag, m['strategy'], nv + 1)
    # bound: at most one neuron of the batch may sit on a near-tie
    assert excuses <= 1, f'{excuses} near-tie excuses in {tag}'
    ctx.close()


@pytest.mark.parametrize('ci', [0, 1])
def test_hip_groups_match_reference_predict(dev, G, M, ci):
    """One launch over 7 neurons with group_size 3 == the reference's three
    `forward` calls of `predict(batch_size=3)` whose searches stop after
    different numbers of steps; then the same through `Decoder.predict`."""
    meta = M[f'dec_small_stop{ci}']
    sd, feats, nv = model(meta)
    want = M[f'g14_stop{ci}_predict']
    ctx = hip.Context(hip.make_dims(sd, nv), sd, dev)
    m = M[f'g14_stop{ci}_group0']
    out = hip_forward(ctx, feats, m, group_size=3)
    lens = out['out_len'].cpu().tolist()
    assert lens == [M[f'g14_stop{ci}_group{g}']['tprime'] for g in range(3)]
    for gi, lo in enumerate(range(0, 7, 3)):
        part = {k: (v[lo:lo + 3] if isinstance(v, torch.Tensor) and
                    k != 'out_len' else v) for k, v in out.items()}
        part['tprime'] = lens[gi]
        # tokens past this group's own T' are <stop> padding
        assert check_search(part, G, f'g14_stop{ci}_group{gi}', 'rerank',
                            nv + 1) == 0
    ctx.close()

    # the Python mirror: Decoder.predict(features=..., batch_size=3)
    idx = lang.Indexer(lang.Vocab(synthetic.vocab_tokens(nv)), None, True,
                       True, True, True, 15)

    class Features(encoders.Encoder):
        feature_shape = (meta['feature_size'],)

        def forward(self, images, masks=None, **_):
            raise AssertionError('features are precomputed')

    lm = lms.LanguageModel(idx, meta['emb'], meta['hidden'])
    dec = decoders.Decoder(idx, Features(), lm, embedding_size=meta['emb'],
                           hidden_size=meta['hidden'])
    res = dec.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    dec.to('cuda')
    ds = data.TensorDataset(feats)
    caps = dec.predict(ds, features=ds, batch_size=3, strategy='rerank',
                       beam_size=4, temperature=0.2, display_progress_as=None)
    assert list(caps) == want['captions']
