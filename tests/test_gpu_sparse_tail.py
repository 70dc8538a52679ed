"""Mask-aware tail of the last stage (round 6, MILAN_FUSE_SPARSE_TAIL; csrc/encoder.hip).

The last stage's output is read by nothing but the level-4 pooling (src/milan/encoders.py:
303-320), and that reads only the pixels under the shrunk mask.  The HIP path therefore runs the
last two bottlenecks only at those pixels and the 3x3 neighbourhoods they depend on (row sets
built on the device from the pooling's own pixel lists, 1x1 convs over gathered rows, the 3x3
over an explicit im2col in the tap-inner kernel's k order).  Contract: the pooled features are
BITWISE those of the dense pass, whatever the masks look like.
"""
import pytest
import torch

from milan_amd import hip, synthetic
from featclass import assert_feature_class
from oracle import milan_oracle as O

pytestmark = pytest.mark.gpu
PREFIX = 'encoder.encoder.model.'


@pytest.fixture(scope='module')
def dev():
    hip.load_library()
    return hip.require_device('cuda')


def _ctx(arch, dev, width=64, seed=3):
    blocks = synthetic.RESNET_BLOCKS[arch]
    sd = synthetic.resnet_state_dict(arch, seed=seed, width=width, prefix=PREFIX)
    ctx = hip.Context(hip.make_dims(sd, 10, blocks=blocks), sd, dev)
    ctx.set_precision('split_f16')
    return ctx, sd


def _masks(kind, n, size, g):
    m = torch.zeros(n, 1, size, size, dtype=torch.uint8)
    for i in range(n):
        if kind == 'rect':
            a = int(torch.randint(max(1, size // 14), max(2, size * 4 // 7), (1,), generator=g))
            b = int(torch.randint(max(1, size // 14), max(2, size * 4 // 7), (1,), generator=g))
            y0 = int(torch.randint(0, size - a + 1, (1,), generator=g))
            x0 = int(torch.randint(0, size - b + 1, (1,), generator=g))
            m[i, 0, y0:y0 + a, x0:x0 + b] = 1
        elif kind == 'random':
            m[i] = (torch.rand(1, size, size, generator=g) > 0.995).to(torch.uint8)
        elif kind == 'full':
            m[i] = 1
        elif kind == 'pixel':
            m[i, 0, int(torch.randint(0, size, (1,), generator=g)),
              int(torch.randint(0, size, (1,), generator=g))] = 1
        elif kind == 'corner':
            m[i, 0, :max(1, size // 5), -max(1, size // 5):] = 1
        elif kind == 'mixed':
            if i % 3 == 0:
                pass                                   # empty: the image skips the trunk
            elif i % 3 == 1:
                m[i, 0, size // 3:size // 2, size // 4:] = 1
            else:
                m[i] = 1
    return m


@pytest.mark.parametrize('arch,n,size,kind', [
    ('resnet50', 6, 224, 'rect'),      # the benchmark's kind of mask, 7 x 7 pixels
    ('resnet101', 3, 224, 'rect'),
    ('resnet50', 5, 224, 'random'),    # scattered pixels: ragged sets
    ('resnet50', 3, 224, 'full'),      # every pixel needed: the sets are everything
    ('resnet50', 7, 224, 'pixel'),     # one pixel: sets of 1 / 4-9 / 9-25 rows
    ('resnet50', 4, 224, 'corner'),
    ('resnet50', 9, 224, 'mixed'),     # with images that skip the trunk altogether
    ('resnet50', 5, 96, 'rect'),       # 3 x 3 pixels in the last stage
    ('resnet50', 4, 64, 'mixed'),      # 2 x 2
    ('resnet50', 3, 20, 'full'),       # 1 x 1
    ('resnet50', 2, 200, 'random'),    # 7 x 7 from a size that is no multiple of 32
])
def test_sparse_tail_is_bitwise_the_dense_pass(dev, arch, n, size, kind):
    ctx, _ = _ctx(arch, dev)
    g = torch.Generator().manual_seed(size * 13 + n)
    images = torch.randint(0, 256, (n, 3, size, size), dtype=torch.uint8, generator=g)
    masks = _masks(kind, n, size, g)
    ctx.set_fusion(sparse_tail=True)
    sparse = ctx.encode(images, masks)
    ctx.set_fusion(sparse_tail=False)
    dense = ctx.encode(images, masks)
    assert torch.isfinite(sparse).all()
    assert torch.equal(sparse, dense)
    assert ctx.status() == 0
    ctx.close()


def test_sparse_tail_matches_the_oracle(dev):
    """... and independent of the dense kernels: full-width ResNet-50 features against the CPU
    oracle (fp32), at the encoder tolerance."""
    ctx, sd = _ctx('resnet50', dev, seed=11)
    images, masks = synthetic.exemplars(1, k=4, size=224, seed=21, zero_every=0)
    got = ctx.encode(images[0], masks[0])
    want = O.encode(O.byte_to_float(images), masks.float(), sd,
                    blocks=synthetic.RESNET_BLOCKS['resnet50'])[0]
    assert_feature_class(got, want)
    ctx.close()


def test_sparse_tail_skips_most_of_the_last_two_blocks(dev):
    """The launches are sized for every row and read the set sizes on the device; what says the
    pruning is real is time: the last stage of a pass with small masks is much shorter."""
    ctx, _ = _ctx('resnet101', dev)
    images, masks = synthetic.exemplars(64, k=15, size=224, seed=5, zero_every=0, device='cuda')
    images, masks = images.flatten(0, 1), masks.flatten(0, 1)
    hip.profile_enable(True)
    times = {}
    try:
        for flag in (True, False, True, False):
            ctx.set_fusion(sparse_tail=flag)
            ctx.encode(images, masks, check=False)       # warm
            hip.profile_enable(True)
            ctx.encode(images, masks, check=False)
            torch.cuda.synchronize()
            st = hip.profile_read_stages()
            times.setdefault(flag, []).append(st['enc_layer4']['region_ms'])
    finally:
        hip.profile_enable(False)
    print('layer4 ms per 960 images: sparse', times[True], 'dense', times[False])
    assert min(times[True]) < 0.9 * min(times[False])
    ctx.close()
