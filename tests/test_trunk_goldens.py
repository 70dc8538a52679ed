"""CPU: the oracle's BasicBlock-ResNet and AlexNet trunks against encoder
goldens produced by the reference's `PyramidConvEncoder` for its 'resnet18'
and 'alexnet' configs (src/milan/encoders.py:326-351)."""
import pytest
import torch

from milan_amd import synthetic
from oracle import milan_oracle as O

TAGS = ['r18_96', 'r18_full', 'alex_100', 'alex_full']


def trunk_sd(m, prefix='encoder.encoder.model.'):
    if m['config'] == 'alexnet':
        return synthetic.alexnet_state_dict(seed=m['weight_seed'],
                                            width=m['width'], prefix=prefix)
    return synthetic.resnet_state_dict(m['config'], seed=m['weight_seed'],
                                       width=m['width'], prefix=prefix)


@pytest.mark.parametrize('tag', TAGS)
def test_g11_oracle_encoder(trunk_goldens, trunk_meta, tag):
    m = trunk_meta[f'g11_{tag}']
    sd = trunk_sd(m)
    images_u8, _ = synthetic.exemplars(1, k=m['m'], size=m['size'],
                                       seed=m['image_seed'], zero_every=0)
    masks_u8 = trunk_goldens[f'g11_{tag}_masks_u8']
    blocks = synthetic.RESNET_BLOCKS.get(m['config'], ())
    feats = O.encode(O.byte_to_float(images_u8), masks_u8.float(), sd,
                     blocks=blocks)[0]
    want = trunk_goldens[f'g11_{tag}_features']
    assert feats.shape == want.shape
    assert want.shape[1] == synthetic.pyramid_feature_size(m['config'],
                                                           m['width'])
    torch.testing.assert_close(feats, want, rtol=2e-4, atol=2e-5)
    assert feats[1].eq(0).all() and want[1].eq(0).all()
    if m['config'] == 'alexnet':
        # nethook's detach() shares storage with the conv output that
        # torchvision's in-place ReLU then rectifies: taps are post-ReLU
        assert (want >= 0).all()


def test_feature_sizes_match_reference_config_table():
    # encoders.py:330-350
    assert synthetic.pyramid_feature_size('alexnet') == 1152
    assert synthetic.pyramid_feature_size('resnet18') == 1024
    assert synthetic.pyramid_feature_size('resnet50') == 3904
    assert synthetic.pyramid_feature_size('resnet101') == 3904


SPATIAL_TAGS = ['sp_96', 'sp_96_nomask', 'sp_full']


def spatial_inputs(m):
    images_u8, masks_u8 = synthetic.exemplars(1, k=m['m'], size=m['size'],
                                              seed=m['image_seed'],
                                              zero_every=0)
    return images_u8[0], (masks_u8[0] if m['with_masks'] else None)


@pytest.mark.parametrize('tag', SPATIAL_TAGS)
def test_g13_oracle_spatial_encoder(trunk_goldens, trunk_meta, tag):
    """SpatialConvEncoder.forward: normalise, THEN mask, layer4 NHWC."""
    m = trunk_meta[f'g13_{tag}']
    sd = synthetic.resnet_state_dict('resnet18', seed=m['weight_seed'],
                                     width=m['width'],
                                     prefix='encoder.encoder.model.')
    images_u8, masks_u8 = spatial_inputs(m)
    got = O.encode_spatial(O.byte_to_float(images_u8),
                           None if masks_u8 is None else masks_u8.float(), sd)
    want = trunk_goldens[f'g13_{tag}_features']
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-5)
