"""Fused expand -> reduce launches of the trunk (csrc/chain.hip) through the C ABI.

One launch computes a bottleneck's 1x1 expand conv (+ residual + ReLU) and the
next bottleneck's 1x1 reduce conv (torchvision Bottleneck.forward as called from
src/milan/encoders.py:298).  The kernel keeps the accumulation order of the two
separate GEMM launches, so the contract is BITWISE equality with the unfused
schedule -- on top of the usual parity with the reference goldens / the oracle.
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


def _ctx(arch, width, dev, seed=3):
    blocks = synthetic.RESNET_BLOCKS[arch]
    sd = synthetic.resnet_state_dict(arch, seed=seed, width=width,
                                     prefix='encoder.encoder.model.')
    ctx = hip.Context(hip.make_dims(sd, 10, blocks=blocks), sd, dev)
    ctx.set_precision('split_f16')
    return ctx, sd


# width 64: planes 64 / 128 / 256 (all three chain configurations, incl. the
# two-source layer1.0 expand) and 512 (layer4: no chain, must still work)
@pytest.mark.parametrize('arch,width,n,size', [
    ('resnet50', 64, 5, 64),     # M per stage: 1280 / 320 / 80 / 20 rows (ragged tiles)
    ('resnet50', 64, 3, 224),    # the real geometry
    ('resnet101', 64, 2, 96),    # 22 chained blocks in layer3
    ('resnet50', 16, 4, 64),     # planes 16..128: layer3 (P = 64) and layer4 (P = 128) chain
])
def test_chain_is_bitwise_the_unfused_schedule(dev, arch, width, n, size):
    ctx, _ = _ctx(arch, width, dev)
    g = torch.Generator().manual_seed(size * 7 + n)
    images = torch.randint(0, 256, (n, 3, size, size), dtype=torch.uint8,
                           generator=g)
    masks = (torch.rand(n, 1, size, size, generator=g) > 0.5).to(torch.uint8)
    ctx.set_fusion(chain=True)
    fused = ctx.encode(images, masks)
    ctx.set_fusion(chain=False)
    plain = ctx.encode(images, masks)
    assert torch.isfinite(fused).all()
    assert torch.equal(fused, plain)
    ctx.close()


# Round 6 (MILAN_FUSE_BNECK): layer1's 3x3 conv in FRONT of the chain launch
# (chain_kernel<.., CONV>): t2 never exists in memory.  Same k order, same (hl, lh, hh)
# order, same epilogue roundings as conv3_p64 / the implicit GEMM -> bitwise.
@pytest.mark.parametrize('arch,n,size', [
    ('resnet50', 3, 224),    # the real geometry: 56 x 56, 3 x 3136 = 36.75 workgroups
    ('resnet101', 1, 224),   # one image: the region reaches before / beyond the batch
    ('resnet50', 5, 64),     # 16 x 16 images: a workgroup spans several images
    ('resnet50', 2, 100),    # 25 x 25: odd width, ragged last workgroup
    ('resnet50', 7, 36),     # 9 x 9: a whole image is smaller than a wave's 32 pixels
    ('resnet50', 1, 4),      # 1 x 1 images: every tap but the centre is outside
    ('resnet50', 1, 260),    # 65 x 65 > 56 columns: falls back to the separate launches
])
def test_conv_front_is_bitwise_the_separate_launches(dev, arch, n, size):
    ctx, _ = _ctx(arch, 64, dev)
    g = torch.Generator().manual_seed(size * 11 + n)
    images = torch.randint(0, 256, (n, 3, size, size), dtype=torch.uint8, generator=g)
    masks = (torch.rand(n, 1, size, size, generator=g) > 0.5).to(torch.uint8)
    ctx.set_fusion(bneck=True)
    fused = ctx.encode(images, masks)
    ctx.set_fusion(bneck=False)
    plain = ctx.encode(images, masks)           # conv3_p64 + chain
    ctx.set_fusion(bneck=False, conv3=False)
    gemm = ctx.encode(images, masks)            # implicit GEMM + chain
    assert torch.isfinite(fused).all()
    assert torch.equal(fused, plain) and torch.equal(fused, gemm)
    ctx.close()


def test_conv_front_launches_are_counted(dev):
    """... and it is really the fused kernel that ran: the launch records of the library
    show the 'bneck' family instead of 'conv3' + 'chain' for layer1."""
    ctx, _ = _ctx('resnet50', 64, dev)
    images, masks = synthetic.exemplars(1, k=3, size=224, seed=9, zero_every=0)
    hip.profile_enable(True)
    try:
        ctx.set_fusion(bneck=True)
        ctx.encode(images[0], masks[0])
        on = hip.profile_read_kernels()
        hip.profile_enable(True)   # (resets the records)
        ctx.set_fusion(bneck=False)
        ctx.encode(images[0], masks[0])
        off = hip.profile_read_kernels()
    finally:
        hip.profile_enable(False)
    assert on['bneck']['launches'] == 3 and on['conv3']['launches'] == 0
    assert off['bneck']['launches'] == 0 and off['conv3']['launches'] == 3
    assert on['chain']['launches'] + 3 == off['chain']['launches']
    ctx.close()


def test_chain_encoder_matches_oracle_at_full_width(dev):
    """Independent of the unfused kernels: full-width ResNet-50 features against
    the CPU oracle (fp32), with the encoder tolerance of test_gpu_parity.py."""
    ctx, sd = _ctx('resnet50', 64, dev, seed=11)
    images_u8, masks = synthetic.exemplars(1, k=3, size=64, seed=21, zero_every=0)
    got = ctx.encode(images_u8[0], masks[0])
    want = O.encode(O.byte_to_float(images_u8), masks.float(), sd,
                    blocks=synthetic.RESNET_BLOCKS['resnet50'])[0]
    assert_feature_class(got, want)
    ctx.close()


def test_fusion_flags_are_validated(dev):
    ctx, _ = _ctx('resnet50', 16, dev)
    with pytest.raises((ValueError, RuntimeError)):
        hip._check(ctx.lib.milan_set_fusion(ctx._h, 0x80))
    ctx.close()
