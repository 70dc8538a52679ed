"""GPU tests of the Python mirror of `src.milan.Decoder` (drop-in surface)."""
import numpy
import pytest
import torch

import milan_amd
from milan_amd import datasets, decoders, encoders, hip, lang, lms, synthetic
from featclass import assert_feature_class
from oracle import milan_oracle as O

pytestmark = pytest.mark.gpu

NV, WIDTH, K, SIZE = 60, 16, 5, 96
BLOCKS = synthetic.RESNET_BLOCKS['resnet50']


@pytest.fixture(scope='module')
def model():
    hip.require_device('cuda')
    idx = lang.Indexer(lang.Vocab(synthetic.vocab_tokens(NV)), None, True, True,
                       True, True, 15)
    enc = encoders.PyramidConvEncoder('resnet50', width=WIDTH, pretrained=False)
    lm = lms.LanguageModel(idx, 16, 64)
    dec = decoders.Decoder(idx, enc, lm, embedding_size=16, hidden_size=64,
                           length=10, beam_size=4)
    sd = synthetic.milan_state_dict(NV + 4, 'resnet50', seed=11, width=WIDTH,
                                    hidden_size=64, embedding_size=16,
                                    lm_hidden_size=64, lm_embedding_size=16)
    res = dec.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return dec.to('cuda'), sd


def oracle_captions(images, masks, sd, batch, strategy='rerank', **kw):
    caps, toks = [], []
    for lo in range(0, len(images), batch):
        feats = O.encode(O.byte_to_float(images[lo:lo + batch]),
                         masks[lo:lo + batch].float(), sd, blocks=BLOCKS)
        out = O.forward(feats, sd, NV, strategy, length=10, beam_size=4, **kw)
        toks.append(out['tokens'])
        caps += [O.reconstruct(t.tolist(), synthetic.vocab_tokens(NV))
                 for t in out['tokens']]
    return caps, toks


def test_forward_rerank_output_contract(model):
    dec, sd = model
    images, masks = synthetic.exemplars(4, k=K, size=SIZE, seed=21)
    out = dec(images, masks)  # defaults: strategy='rerank', beam 4, length 10
    assert isinstance(out, milan_amd.DecoderOutput)
    want_caps, want_toks = oracle_captions(images, masks, sd, 4)
    tp = want_toks[0].shape[1]
    assert out.tokens.shape == (4, tp) and out.tokens.dtype == torch.long
    assert out.beam_tokens.shape == (4, 4, tp) and out.beam_scores.shape == (4, 4)
    assert out.predictions is None and out.attentions is None
    assert out.scores.shape == (4,) and out.scores.device.type == 'cuda'
    assert list(out.captions) == want_caps
    assert len(out.beam_captions) == 4 and len(out.beam_captions[0]) == 4
    assert out.beam_captions[0][0] == dec.indexer.reconstruct(
        out.beam_tokens[0, 0].tolist())
    # float inputs (what the reference's dataset hands out) give the same
    out_f = dec(O.byte_to_float(images), masks.float())
    assert list(out_f.captions) == want_caps
    # positional field order as DecoderWithCLIP relies on (outputs[3:])
    assert out[0] is out.captions and out[2] is out.tokens


def test_forward_greedy_and_features_input(model):
    dec, sd = model
    images, masks = synthetic.exemplars(3, k=K, size=SIZE, seed=22)
    feats = dec.encode(images, masks)
    assert feats.shape == (3, K, dec.feature_size)
    want = O.encode(O.byte_to_float(images), masks.float(), sd, blocks=BLOCKS)
    assert_feature_class(feats, want)
    out = dec(feats, strategy='greedy', mi=False)
    ref = O.forward(want, sd, NV, 'greedy', length=10, mi=False)
    assert torch.equal(out.tokens.cpu(), ref['tokens'])
    assert out.predictions.shape == (3, 10, NV + 4)
    assert out.attentions.shape == (3, 10, K) and out.beam_tokens is None
    torch.testing.assert_close(out.scores.cpu(), ref['scores'], rtol=1e-4,
                               atol=2e-3)
    # default mi for greedy = True when an LM is present (decoders.py:385-387)
    out_mi = dec(feats, strategy='greedy')
    ref_mi = O.forward(want, sd, NV, 'greedy', length=10)
    assert torch.equal(out_mi.tokens.cpu(), ref_mi['tokens'])


def test_init_state_and_step_api(model):
    dec, sd = model
    g = torch.Generator().manual_seed(3)
    feats = torch.rand(3, K, dec.feature_size, generator=g)
    st = dec.init_state(feats)
    assert st.h_lm.shape == (2, 3, 64) and st.h_lm.eq(0).all()
    ost = O.init_state(feats, sd, lm=True)
    torch.testing.assert_close(st.h.cpu(), ost.h, rtol=1e-4, atol=1e-5)
    toks = torch.tensor([NV, 3, 7])
    step = dec.step(feats, toks, st)
    p, a, ost2 = O.step(feats, O.project_keys(feats, sd), toks, ost, sd, 0.2)
    torch.testing.assert_close(step.predictions.cpu(), p, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(step.attentions.cpu(), a, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(step.state.h_lm.cpu(), ost2.h_lm, rtol=1e-4,
                               atol=1e-5)
    no_lm = dec.init_state(feats, lm=False)
    assert no_lm.h_lm is None and no_lm.c_lm is None


def test_predict_on_disk_dataset_equals_reference_batching(model, tmp_path):
    """predict(batch_size=b) == the reference feeding b neurons per forward,
    although the GPU sees all neurons in one launch."""
    dec, sd = model
    images, masks = synthetic.exemplars(7, k=K, size=SIZE, seed=23,
                                        zero_every=11)
    (tmp_path / 'conv5').mkdir()
    numpy.save(tmp_path / 'conv5' / 'images.npy', images.numpy())
    numpy.save(tmp_path / 'conv5' / 'masks.npy', masks.numpy())
    ds = datasets.TopImagesDataset(tmp_path)
    want, _ = oracle_captions(images, masks, sd, 3)
    got = dec.predict(ds, batch_size=3, display_progress_as=None,
                      device='cuda')
    assert isinstance(got, tuple) and list(got) == want
    # generic (non-mmap) dataset path: a list of TopImages-like tuples
    samples = [ds[i] for i in range(len(ds))]
    got2 = dec.predict(samples, batch_size=3, display_progress_as=None)
    assert list(got2) == want
    # precomputed-features variant (decoders.py:850,858-864)
    feats = dec.encoder.map(samples, image_index=2, mask_index=3,
                            display_progress_as=False, device='cuda')
    got3 = dec.predict(ds, features=feats, batch_size=3,
                       display_progress_as=None)
    assert list(got3) == want


def test_foreign_encoder_plugs_in(model):
    """The reference's FakeEncoder pattern (tests/milan/conftest.py:9-24)."""
    dec, sd = model

    class FakeEncoder(encoders.Encoder):
        feature_shape = (61 * WIDTH,)

        def forward(self, images, masks, **kwargs):
            assert images.shape[0] == masks.shape[0] and images.shape[1] == 3
            return torch.zeros(len(images), *self.feature_shape)

        def properties(self):
            return {}

    fake = decoders.Decoder(dec.indexer, FakeEncoder(), None,
                            embedding_size=16, hidden_size=64, length=6,
                            beam_size=3)
    dsd = {k: v for k, v in sd.items()
           if not k.startswith(('encoder.', 'lm.'))}
    fake.load_state_dict(dsd, strict=True)
    fake.to('cuda')
    images, masks = synthetic.exemplars(2, k=K, size=32, seed=1)
    out = fake(images.float(), masks.float())  # strategy defaults to 'beam'
    ref = O.forward(torch.zeros(2, K, 61 * WIDTH), dsd, NV, 'beam', length=6,
                    beam_size=3, mi=False)
    tp = ref['beam_tokens'].shape[2]
    assert torch.equal(out.beam_tokens.cpu(), ref['beam_tokens'][:, :, :tp])


def test_save_load_roundtrip_gives_same_descriptions(model, tmp_path):
    dec, sd = model
    images, masks = synthetic.exemplars(2, k=K, size=SIZE, seed=24)
    before = dec(images, masks).captions
    dec.save(tmp_path / 'm.pth')
    again = decoders.Decoder.load(tmp_path / 'm.pth').to('cuda')
    assert again(images, masks).captions == before


def test_precision_switch_gives_same_captions(model):
    """decoder.precision: 'auto' (the default since round 6: split_f16 with the loud
    per-call fallback to f32) / 'split_f16' / 'f32' -- fp32-class error in all of them:
    same captions, features within the encoder tolerance."""
    dec, sd = model
    images, masks = synthetic.exemplars(4, k=K, size=SIZE, seed=25)
    import os
    assert dec.precision == os.environ.get('MILAN_PRECISION', 'auto')
    auto = dec(images, masks)
    assert dec._context().precision == ('f32' if dec.precision == 'f32' else 'split_f16')
    dec.precision = 'f32'
    f32 = dec(images, masks)
    assert auto.captions == f32.captions
    feats32 = dec.encode(images, masks)
    dec.precision = 'split_f16'
    try:
        sp = dec(images, masks)
        feats_sp = dec.encode(images, masks)
        assert dec._context().precision == 'split_f16'
    finally:
        dec.precision = 'f32'
    assert sp.captions == f32.captions
    assert_feature_class(feats_sp, feats32, what='split_f16 vs f32')
    with pytest.raises(ValueError, match='unknown precision'):
        dec.precision = 'bf16'
        dec(images, masks)
    dec.precision = os.environ.get('MILAN_PRECISION', 'auto')


def test_sample_strategy(model):
    """strategy='sample' (reference decoders.py:448-453): seeded runs repeat,
    the draw order is row by row per step (what a seeded reference run on the
    same device consumes), scores are the sampled tokens' log-probs."""
    dec, sd = model
    g = torch.Generator().manual_seed(5)
    feats = torch.rand(3, K, dec.feature_size, generator=g).to('cuda')
    torch.manual_seed(123)
    a = dec(feats, strategy='sample', mi=False, length=6)
    torch.manual_seed(123)
    b = dec(feats, strategy='sample', mi=False, length=6)
    assert torch.equal(a.tokens, b.tokens) and torch.equal(a.scores, b.scores)
    assert a.tokens.shape == (3, 6) and a.predictions.shape == (3, 6, NV + 4)
    assert a.attentions.shape == (3, 6, K) and a.beam_tokens is None
    assert int(a.tokens.min()) >= 0 and int(a.tokens.max()) < NV + 4
    picked = a.predictions.gather(2, a.tokens.unsqueeze(-1)).squeeze(-1)
    torch.testing.assert_close(a.scores, picked.sum(1), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(a.predictions.exp().sum(-1),
                               torch.ones(3, 6, device='cuda'), rtol=1e-4,
                               atol=1e-4)
    # the same loop written out against the step API draws the same tokens
    torch.manual_seed(123)
    state = dec.init_state(feats, lm=False)
    cur = torch.full((3,), NV, dtype=torch.long, device='cuda')
    for t in range(6):
        step = dec.step(feats, cur, state)
        cur = torch.stack([
            torch.distributions.Categorical(probs=p.exp()).sample()
            for p in step.predictions
        ])
        assert torch.equal(cur, a.tokens[:, t])
        state = step.state
    torch.manual_seed(321)
    c = dec(feats, strategy='sample', mi=False, length=6)
    assert not torch.equal(a.tokens, c.tokens)
