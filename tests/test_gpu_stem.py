"""Fused stem (csrc/stem.hip) through the C ABI: conv1 7x7/2 + bn1 + ReLU + maxpool
3x3/2 of torchvision's ResNet as driven by src/milan/encoders.py:286-320, in one
persistent launch over LDS-resident input tiles, the raw conv1 tensor (pyramid
level 0) written only inside the bounding box of the mask weights.

The kernel repeats the arithmetic of the three launches it replaces, so the contract
is BITWISE equality with the unfused schedule for every image size / mask shape --
on top of the parity with the oracle that the rest of the GPU suite checks with the
fused stem switched on (it is the default).
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
    # a narrow trunk behind the full-width (64-channel) stem keeps the test fast
    sd = synthetic.resnet_state_dict('resnet50', seed=5, width=64,
                                     prefix='encoder.encoder.model.')
    # negative and zero bn1 scales: the kernel pools before bn on sign-flipped values
    w = sd['encoder.encoder.model.bn1.weight']
    w[1::3] = -w[1::3]
    w[5] = 0.
    c = hip.Context(hip.make_dims(sd, 10, blocks=synthetic.RESNET_BLOCKS['resnet50']),
                    sd, dev)
    c.set_precision('split_f16')
    yield c, sd
    c.close()


def _masks(kind, n, h, w, g):
    if kind == 'none':
        return None
    if kind == 'random':
        return (torch.rand(n, 1, h, w, generator=g) > 0.7).to(torch.uint8)
    m = torch.zeros(n, 1, h, w, dtype=torch.uint8)
    if kind == 'rect':
        for i in range(n):
            y0 = int(torch.randint(0, h - 1, (1,), generator=g))
            x0 = int(torch.randint(0, w - 1, (1,), generator=g))
            y1 = int(torch.randint(y0 + 1, h + 1, (1,), generator=g))
            x1 = int(torch.randint(x0 + 1, w + 1, (1,), generator=g))
            m[i, 0, y0:y1, x0:x1] = 1
        m[n - 1] = 0  # an all-zero mask: empty bounding box, zero-mask rule
    elif kind == 'corners':
        m[:, 0, 0, 0] = 1
        m[:, 0, h - 1, w - 1] = 1
    elif kind == 'pixel':
        for i in range(n):
            m[i, 0, (i * 37) % h, (i * 91) % w] = 1
    return m


@pytest.mark.parametrize('n,h,w,kind', [
    (3, 224, 224, 'rect'),      # the real geometry: 8 x 7 tiles of 7 x 8 pooled pixels
    (2, 224, 224, 'none'),      # no masks: every raw pixel is pooled
    (9, 224, 224, 'pixel'),     # more images than XCDs, one mask pixel each
    (4, 64, 64, 'random'),      # 16 x 16 pooled: ragged tiles in both directions
    (5, 97, 131, 'rect'),       # odd sizes: h1 = 49, w1 = 66, hp = 25, wp = 33
    (2, 33, 47, 'corners'),
    (3, 7, 9, 'random'),        # smaller than one tile, taps hang over every edge
    (2, 1, 1, 'none'),
    (17, 40, 40, 'rect'),
    (2, 320, 256, 'rect'),      # larger than the benchmark geometry: 80 x 64 pooled
])
def test_fused_stem_is_bitwise_the_three_launches(ctx, n, h, w, kind):
    c, _ = ctx
    g = torch.Generator().manual_seed(n * 1000 + h * 7 + w)
    images = torch.randint(0, 256, (n, 3, h, w), dtype=torch.uint8, generator=g)
    masks = _masks(kind, n, h, w, g)
    c.set_fusion(chain=True, stem=True)
    fused = c.encode(images, masks)
    c.set_fusion(chain=True, stem=False)
    plain = c.encode(images, masks)
    c.set_fusion(chain=True, stem=True)
    assert torch.isfinite(fused).all()
    assert torch.equal(fused, plain)


def test_fused_stem_float_images_and_spatial_mode(ctx):
    c, _ = ctx
    g = torch.Generator().manual_seed(77)
    images = torch.rand(3, 3, 80, 112, generator=g)
    masks = (torch.rand(3, 1, 80, 112, generator=g) > 0.5).float()
    out = {}
    for stem in (True, False):
        c.set_fusion(chain=True, stem=stem)
        out[stem] = (c.encode(images, masks), c.encode_spatial(images, masks))
    c.set_fusion(chain=True, stem=True)
    assert torch.equal(out[True][0], out[False][0])
    assert torch.equal(out[True][1], out[False][1])


def test_fused_stem_repeated_calls_reuse_the_raw_buffer(ctx):
    """The raw tensor is only written inside each image's bounding box: stale values
    from an earlier call with larger masks must never reach the features."""
    c, _ = ctx
    g = torch.Generator().manual_seed(5)
    images = torch.randint(0, 256, (4, 3, 96, 96), dtype=torch.uint8, generator=g)
    big = torch.ones(4, 1, 96, 96, dtype=torch.uint8)
    small = torch.zeros(4, 1, 96, 96, dtype=torch.uint8)
    small[:, 0, 40:50, 30:44] = 1
    c.set_fusion(chain=True, stem=True)
    c.encode(images, big)
    fused = c.encode(images, small)
    c.set_fusion(chain=True, stem=False)
    plain = c.encode(images, small)
    c.set_fusion(chain=True, stem=True)
    assert torch.equal(fused, plain)


def test_fused_stem_matches_oracle(ctx):
    """Independent of the unfused kernels: features against the CPU oracle (fp32)."""
    c, sd = ctx
    images_u8, masks = synthetic.exemplars(1, k=3, size=64, seed=23, zero_every=0)
    c.set_fusion(chain=True, stem=True)
    got = c.encode(images_u8[0], masks[0])
    want = O.encode(O.byte_to_float(images_u8), masks.float(), sd,
                    blocks=synthetic.RESNET_BLOCKS['resnet50'])[0]
    assert_feature_class(got, want)
