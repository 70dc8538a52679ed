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
        hip._check(ctx.lib.milan_set_fusion(ctx._h, 0x40))
    ctx.close()
