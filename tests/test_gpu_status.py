"""Split-f16 fails LOUDLY.

The reference computes in plain fp32 (src/milan/encoders.py:295-320): it never saturates
and a NaN / Inf pixel turns every pyramid level of its image into NaN (each level pools
NaN x mask).  The split-f16 storage clamps |x * 2^act_scale| at 65504; every kernel that
writes it reports a hit through the context's status word (include/milan_hip.h,
milan_status) and the Python mirror raises FloatingPointError or -- precision 'auto' --
reruns the call in the exact-fp32 mode.  A non-finite input pixel poisons its image
exactly as in the oracle.
"""
import math
import warnings

import pytest
import torch

from milan_amd import hip, synthetic
from featclass import FEATURE_CLASS, feature_error
from oracle import milan_oracle as O
from test_gpu_dtype_error import PREFIX, _scaled_network

pytestmark = pytest.mark.gpu
BLOCKS = synthetic.RESNET_BLOCKS['resnet50']


@pytest.fixture(scope='module')
def dev():
    hip.load_library()
    return hip.require_device('cuda')


def _ctx(sd, dev):
    return hip.Context(hip.make_dims(sd, 10, blocks=BLOCKS), sd, dev)


def test_benchmark_like_run_reports_no_flags(dev):
    """The calibrated synthetic network (activations O(0.01 - 100)) never touches the
    clamp: status 0 after encode in both modes, and the status read does not change
    the result."""
    sd = synthetic.resnet_state_dict('resnet50', seed=7, width=64, prefix=PREFIX)
    images, masks = synthetic.exemplars(1, k=4, size=96, seed=3, zero_every=0)
    ctx = _ctx(sd, dev)
    ctx.set_precision('split_f16')
    a = ctx.encode(images[0], masks[0], check=False)
    assert ctx.status(clear=True) == 0
    b = ctx.encode(images[0], masks[0])          # checked path
    assert torch.equal(a, b)
    assert ctx.status() == 0
    ctx.close()


def test_large_activations_raise_instead_of_clamping(dev):
    """A homogeneously scaled network whose layer4 activations reach ~5000: beyond
    65504 / 2^5 = 2047.  split_f16 must not hand back clamped features."""
    sd = _scaled_network(150.0)
    images, masks = synthetic.exemplars(1, k=3, size=96, seed=41, zero_every=0)
    with torch.no_grad():
        ref = O.encode(O.byte_to_float(images), masks.float(), sd, blocks=BLOCKS)[0]
    top = float(ref[..., 29 * 64:].abs().max())
    assert top > 2047, top   # the pooled layer4 features alone exceed the range
    ctx = _ctx(sd, dev)
    ctx.set_precision('split_f16')
    with pytest.raises(FloatingPointError, match='clamped'):
        ctx.encode(images[0], masks[0])
    assert ctx.status() == 0                     # the failed call cleared the word
    # describe() goes through the same guard
    full = synthetic.milan_state_dict(14, config='resnet50', seed=7, width=64,
                                      hidden_size=64, embedding_size=16,
                                      lm_hidden_size=64, lm_embedding_size=16)
    for k, v in sd.items():
        full[k] = v
    ctx2 = hip.Context(hip.make_dims(full, 10, blocks=BLOCKS), full, dev)
    ctx2.set_precision('split_f16')
    with pytest.raises(FloatingPointError):
        ctx2.describe(images, masks, hip.GREEDY, 4, 1, False, 0.2)
    ctx2.close()

    # precision 'auto': the call is rerun in f32 and equals the f32 mode bit for bit
    ctx.on_saturation = 'f32'
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        got = ctx.encode(images[0], masks[0]).cpu()
    assert any('rerunning' in str(w.message) for w in caught)
    assert ctx.saturation_fallbacks == 1 and ctx.precision == 'split_f16'
    ctx.set_precision('f32')
    want = ctx.encode(images[0], masks[0]).cpu()
    assert torch.equal(got, want)
    e, where = feature_error(got, ref)
    assert e <= FEATURE_CLASS, (e, where)

    # a calibrated scale keeps the same network inside the split format: no flag, fp32 class
    ctx.on_saturation = 'raise'
    ctx.set_precision('split_f16')
    k = ctx.calibrate(images[0])
    assert 0 <= k < 5, k
    got = ctx.encode(images[0], masks[0]).cpu()
    e, where = feature_error(got, ref)
    print(f'calibrated scale 2^{k}: max|err| / level scale {e:.3g}')
    assert e <= FEATURE_CLASS, (e, where)
    ctx.close()


def test_calibrate_moves_the_scale_up_for_small_activations(dev):
    sd = _scaled_network(1e-4)
    images, masks = synthetic.exemplars(1, k=3, size=96, seed=41, zero_every=0)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        ref64 = O.encode(O.byte_to_float(images).double(), masks.double(), sd64,
                         blocks=BLOCKS)[0]
    ctx = _ctx(sd, dev)
    ctx.set_precision('split_f16')
    assert ctx.act_scale_log2 == 5
    amax = ctx.encoder_absmax(images[0])
    assert math.isfinite(amax) and amax > 0
    k = ctx.calibrate(images[0])
    # the stem (un-scaled part of this network) bounds the scale; it must still have grown
    assert k > 5 and ctx.act_scale_log2 == k
    got = ctx.encode(images[0], masks[0]).cpu()
    e, where = feature_error(got, ref64)
    print(f'factor 1e-4: absmax {amax:.3g}, calibrated 2^{k}: err {e:.3g}')
    assert e <= FEATURE_CLASS, (e, where)
    with pytest.raises(ValueError):
        ctx.set_act_scale_log2(11)
    ctx.close()


@pytest.mark.parametrize('bad', [float('nan'), float('inf'), -float('inf')])
@pytest.mark.parametrize('precision', ['split_f16', 'f32'])
def test_non_finite_pixel_poisons_its_image_like_the_oracle(dev, bad, precision):
    """One NaN / Inf pixel OUTSIDE the mask of one exemplar: the oracle's features of that
    image are NaN at every level (NaN x 0 in the pooling), every other image is untouched."""
    sd = synthetic.resnet_state_dict('resnet50', seed=11, width=16, prefix=PREFIX)
    images, masks = synthetic.exemplars(2, k=3, size=96, seed=5, zero_every=0)
    x = O.byte_to_float(images).clone()
    clean = x.clone()
    masks = masks.clone()
    masks[0, 1, 0, :20, :20] = 0
    x[0, 1, 0, 5, 7] = bad
    with torch.no_grad():
        ref = O.encode(x, masks.float(), sd, blocks=BLOCKS).reshape(6, -1)
    assert torch.isnan(ref[1]).all() and torch.isfinite(ref[[0, 2, 3, 4, 5]]).all()
    ctx = _ctx(sd, dev)
    ctx.set_precision(precision)
    flat = x.reshape(6, 3, 96, 96)
    mflat = masks.reshape(6, 1, 96, 96)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        got = ctx.encode(flat, mflat).cpu()
    assert any('NaN / Inf' in str(w.message) for w in caught)
    assert torch.equal(torch.isnan(got), torch.isnan(ref))
    want = ctx.encode(clean.reshape(6, 3, 96, 96), mflat).cpu()
    keep = [0, 2, 3, 4, 5]
    assert torch.equal(got[keep], want[keep])    # the other images: bit for bit
    assert ctx.status() == 0
    ctx.close()


def test_non_finite_pixel_spatial_encoder(dev):
    sd = synthetic.resnet_state_dict('resnet18', seed=3, width=16, prefix=PREFIX)
    blocks = synthetic.RESNET_BLOCKS['resnet18']
    ctx = hip.Context(hip.make_dims(sd, 10, blocks=blocks), sd, dev)
    images, masks = synthetic.exemplars(1, k=3, size=64, seed=9, zero_every=0)
    x = O.byte_to_float(images)[0].clone()
    x[2, 1, 10, 10] = float('nan')
    for precision in ('f32', 'split_f16'):
        ctx.set_precision(precision)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            out = ctx.encode_spatial(x, masks[0]).cpu()
        assert torch.isnan(out[2]).all() and torch.isfinite(out[:2]).all()
    ctx.close()
