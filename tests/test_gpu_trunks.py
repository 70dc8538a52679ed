"""GPU parity of the reference's other pyramid configs ('resnet18' = basic
blocks, 'alexnet') against encoder goldens produced by the reference
(tests/golden/make_golden_trunks.py), through the C ABI and the Python
mirror."""
import pytest
import torch

from milan_amd import encoders, hip, synthetic
from featclass import assert_feature_class
from oracle import milan_oracle as O
from tests.test_trunk_goldens import TAGS, trunk_sd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    return hip.require_device('cuda')


@pytest.mark.parametrize('precision', ['f32', 'split_f16'])
@pytest.mark.parametrize('as_u8', [True, False])
@pytest.mark.parametrize('tag', TAGS)
def test_encoder_matches_reference_golden(dev, trunk_goldens, trunk_meta, tag,
                                          as_u8, precision):
    m = trunk_meta[f'g11_{tag}']
    sd = trunk_sd(m)
    blocks = synthetic.RESNET_BLOCKS.get(m['config'], (0, 0, 0, 0))
    dims = hip.make_dims(sd, 10, blocks=blocks)
    assert dims.trunk_kind == synthetic.trunk_kind(m['config'])
    assert dims.feature_size == synthetic.pyramid_feature_size(m['config'],
                                                               m['width'])
    ctx = hip.Context(dims, sd, dev)
    ctx.set_precision(precision)
    images_u8, _ = synthetic.exemplars(1, k=m['m'], size=m['size'],
                                       seed=m['image_seed'], zero_every=0)
    masks_u8 = trunk_goldens[f'g11_{tag}_masks_u8']
    if as_u8:
        got = ctx.encode(images_u8[0], masks_u8[0])
    else:
        got = ctx.encode(O.byte_to_float(images_u8[0]), masks_u8[0].float())
    want = trunk_goldens[f'g11_{tag}_features']
    assert_feature_class(got, want)
    assert got[1].eq(0).all(), 'all-zero mask must give an exactly-zero row'
    assert not torch.isnan(got).any()
    ctx.close()


@pytest.mark.parametrize('config,tag', [('resnet18', 'r18_96'),
                                        ('alexnet', 'alex_100')])
def test_python_encoder_configs(dev, trunk_goldens, trunk_meta, config, tag):
    """`PyramidConvEncoder(config=...)` (encoders.py:326-351) as a module:
    torchvision key names, strict load, forward == golden."""
    m = trunk_meta[f'g11_{tag}']
    enc = encoders.PyramidConvEncoder(config, pretrained=False,
                                      width=m['width'])
    assert enc.feature_shape == (synthetic.pyramid_feature_size(
        config, m['width']),)
    # (the synthetic AlexNet classifier is a small stand-in for the 4096-wide
    # one the reference computes and discards: leave it out)
    sd = {k: v for k, v in trunk_sd(m, prefix='encoder.model.').items()
          if '.classifier.' not in k}
    res = enc.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    # only what the reference computes and discards may be absent
    assert all(k.startswith('encoder.model.classifier.') or k in ('mean', 'std')
               for k in res.missing_keys), res.missing_keys
    enc.to('cuda')
    images_u8, _ = synthetic.exemplars(1, k=m['m'], size=m['size'],
                                       seed=m['image_seed'], zero_every=0)
    masks_u8 = trunk_goldens[f'g11_{tag}_masks_u8']
    got = enc(O.byte_to_float(images_u8[0]), masks_u8[0].float())
    assert_feature_class(got, trunk_goldens[f'g11_{tag}_features'])
    assert enc.properties()['config'] == config
    with pytest.raises(ValueError, match='encoder not supported'):
        encoders.PyramidConvEncoder('vgg16')


def test_dims_reject_inconsistent_trunk(dev):
    sd = synthetic.resnet_state_dict('resnet18', seed=1, width=16,
                                     prefix='encoder.encoder.model.')
    dims = hip.make_dims(sd, 10, blocks=synthetic.RESNET_BLOCKS['resnet18'])
    dims.trunk_kind = hip.TRUNK_BOTTLENECK  # F = 16*w is not 61*w
    with pytest.raises(ValueError):
        hip.Context(dims, sd, dev)
    dims.trunk_kind = 7
    with pytest.raises(ValueError):
        hip.Context(dims, sd, dev)


from tests.test_trunk_goldens import SPATIAL_TAGS, spatial_inputs  # noqa: E402


@pytest.mark.parametrize('precision', ['f32', 'split_f16'])
@pytest.mark.parametrize('tag', SPATIAL_TAGS)
def test_spatial_encoder_matches_reference_golden(dev, trunk_goldens,
                                                  trunk_meta, tag, precision):
    """SpatialConvEncoder (encoders.py:158-234) through milan_encode_spatial:
    u8 and float inputs, with and without masks."""
    m = trunk_meta[f'g13_{tag}']
    sd = synthetic.resnet_state_dict('resnet18', seed=m['weight_seed'],
                                     width=m['width'],
                                     prefix='encoder.encoder.model.')
    ctx = hip.Context(
        hip.make_dims(sd, 10, blocks=synthetic.RESNET_BLOCKS['resnet18']), sd,
        dev)
    ctx.set_precision(precision)
    images_u8, masks_u8 = spatial_inputs(m)
    want = trunk_goldens[f'g13_{tag}_features']
    got = ctx.encode_spatial(images_u8, masks_u8)
    assert got.shape == want.shape
    assert_feature_class(got, want)
    got_f = ctx.encode_spatial(
        O.byte_to_float(images_u8),
        None if masks_u8 is None else masks_u8.float())
    assert_feature_class(got_f, want)
    ctx.close()


def test_spatial_encoder_module(dev, trunk_goldens, trunk_meta):
    m = trunk_meta['g13_sp_96']
    enc = encoders.encoder('spatial', config='resnet18', pretrained=False,
                           width=m['width'])
    assert isinstance(enc, encoders.SpatialConvEncoder)
    assert encoders.parse('SpatialConvEncoder') is encoders.SpatialConvEncoder
    sd = synthetic.resnet_state_dict('resnet18', seed=m['weight_seed'],
                                     width=m['width'], prefix='encoder.model.')
    res = enc.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and set(res.missing_keys) <= {'mean', 'std'}
    enc.to('cuda')
    images_u8, masks_u8 = spatial_inputs(m)
    got = enc(O.byte_to_float(images_u8), masks_u8.float())
    assert_feature_class(got, trunk_goldens['g13_sp_96_features'])
    with pytest.raises(ValueError, match='encoder not supported'):
        encoders.SpatialConvEncoder('resnet50')
    # the 224x224 geometry the reference hard-codes: (49, 512)
    assert encoders.SpatialConvEncoder('resnet18',
                                       pretrained=False).feature_shape == (49,
                                                                           512)


def test_decoder_over_spatial_encoder(dev):
    """A Decoder whose encoder is a SpatialConvEncoder: (B,3,H,W) images ->
    (B, positions, C) features -> captions, against the oracle."""
    from milan_amd import decoders, lang
    nv, width, size, n = 30, 16, 96, 3
    idx = lang.Indexer(lang.Vocab(synthetic.vocab_tokens(nv)), None, True,
                       True, True, True, 15)
    enc = encoders.SpatialConvEncoder('resnet18', pretrained=False,
                                      width=width)
    dec = decoders.Decoder(idx, enc, None, embedding_size=16, hidden_size=32,
                           length=6, beam_size=3)
    assert dec.feature_size == 8 * width
    sd = synthetic.decoder_state_dict(nv + 4, feature_size=8 * width,
                                      hidden_size=32, embedding_size=16,
                                      lm=False, seed=3)
    trunk = synthetic.resnet_state_dict('resnet18', seed=4, width=width,
                                        prefix='encoder.encoder.model.')
    sd.update(trunk)
    res = dec.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    dec.to('cuda')
    g = torch.Generator().manual_seed(9)
    images = torch.rand(n, 3, size, size, generator=g)
    masks = (torch.rand(n, 1, size, size, generator=g) > 0.4).float()
    feats = O.encode_spatial(images, masks, trunk)
    want = O.forward(feats, sd, nv, 'greedy', length=6, mi=False)
    out = dec(images, masks, strategy='greedy')
    assert_feature_class(dec.encode(images, masks), feats)
    top2 = want['predictions'].topk(2, dim=-1).values
    from tests.test_gpu_parity import assert_tokens_match
    assert_tokens_match(out.tokens, want['tokens'], top2[..., 0] - top2[..., 1])
