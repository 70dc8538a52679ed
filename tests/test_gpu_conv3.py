"""Layer1's 3x3 convolutions (64 -> 64 channels, torchvision Bottleneck.conv2 as called
from src/milan/encoders.py:298) through csrc/conv3.hip: a persistent kernel that keeps
the weights in registers and the input tile in LDS.  It keeps the accumulation order of
the implicit-GEMM kernel (the accumulator travels from the wave that owns the first half
of K to the wave that owns the second), so the contract is BITWISE equality with the
implicit-GEMM schedule for every image size, on top of oracle parity.
"""
import pytest
import torch

from milan_amd import hip, synthetic
from featclass import assert_feature_class
from oracle import milan_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    hip.load_library()
    return hip.require_device('cuda')


@pytest.fixture(scope='module')
def ctx(dev):
    sd = synthetic.resnet_state_dict('resnet50', seed=9, width=64,
                                     prefix='encoder.encoder.model.')
    c = hip.Context(hip.make_dims(sd, 10, blocks=synthetic.RESNET_BLOCKS['resnet50']),
                    sd, dev)
    c.set_precision('split_f16')
    yield c, sd
    c.close()


@pytest.mark.parametrize('n,h,w', [
    (3, 224, 224),    # the real geometry: 56 x 56 at layer1 = 7 x 4 tiles of 8 x 14 pixels
    (9, 224, 224),    # more images than XCDs
    (2, 64, 64),      # 16 x 16: ragged tiles in both directions
    (5, 97, 131),     # 25 x 33
    (3, 40, 72),      # 10 x 18
    (2, 7, 9),        # 2 x 3: smaller than one tile
    (1, 1, 1),
    (40, 32, 32),     # many images, one partial tile each
    (2, 320, 256),    # 80 x 64 at layer1: 10 x 5 tiles
])
def test_conv3_is_bitwise_the_implicit_gemm(ctx, n, h, w):
    c, _ = ctx
    g = torch.Generator().manual_seed(n * 100 + h + w)
    images = torch.randint(0, 256, (n, 3, h, w), dtype=torch.uint8, generator=g)
    masks = (torch.rand(n, 1, h, w, generator=g) > 0.5).to(torch.uint8)
    c.set_fusion(chain=True, stem=True, conv3=True)
    fused = c.encode(images, masks)
    c.set_fusion(chain=True, stem=True, conv3=False)
    plain = c.encode(images, masks)
    c.set_fusion(chain=True, stem=True, conv3=True)
    assert torch.isfinite(fused).all()
    assert torch.equal(fused, plain)


def test_conv3_spatial_output_is_bitwise(ctx):
    c, _ = ctx
    g = torch.Generator().manual_seed(3)
    images = torch.rand(2, 3, 96, 128, generator=g)
    masks = (torch.rand(2, 1, 96, 128, generator=g) > 0.3).float()
    out = {}
    for on in (True, False):
        c.set_fusion(chain=True, stem=True, conv3=on)
        out[on] = c.encode_spatial(images, masks)
    c.set_fusion(chain=True, stem=True, conv3=True)
    assert torch.equal(out[True], out[False])


def test_conv3_encoder_matches_oracle(ctx):
    c, sd = ctx
    images_u8, masks = synthetic.exemplars(1, k=3, size=64, seed=29, zero_every=0)
    c.set_fusion(chain=True, stem=True, conv3=True)
    got = c.encode(images_u8[0], masks[0])
    want = O.encode(O.byte_to_float(images_u8), masks.float(), sd,
                    blocks=synthetic.RESNET_BLOCKS['resnet50'])[0]
    assert_feature_class(got, want)
