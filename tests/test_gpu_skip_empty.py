"""Exemplars with an all-zero mask stay out of the trunk pass (MILAN_FUSE_SKIP_EMPTY).

The reference pools `features * mask` (src/milan/encoders.py:310-317): an all-zero mask
fails its isclose test, stays un-normalised and yields exact zeros at every pyramid
level whatever the trunk computed.  The HIP path therefore runs the trunk over the
images that have a non-empty weight list at some level only (compact_images_kernel,
csrc/encoder.hip) -- every other image's features must stay bit for bit what the full
pass produces, in place.
"""
import pytest
import torch

from milan_amd import hip, synthetic
from oracle import milan_oracle as O
from featclass import FEATURE_CLASS, feature_error
from test_gpu_dtype_error import PREFIX

pytestmark = pytest.mark.gpu
BLOCKS = synthetic.RESNET_BLOCKS['resnet50']


@pytest.fixture(scope='module')
def dev():
    hip.load_library()
    return hip.require_device('cuda')


@pytest.fixture(scope='module')
def ctx(dev):
    sd = synthetic.resnet_state_dict('resnet50', seed=13, width=16, prefix=PREFIX)
    c = hip.Context(hip.make_dims(sd, 10, blocks=BLOCKS), sd, dev)
    c.sd = sd
    yield c
    c.close()


PATTERNS = {
    'scattered': [2, 5, 6, 11],
    'first_and_last': [0, 11],
    'all_but_one': [i for i in range(12) if i != 7],
    'all': list(range(12)),
    'none': [],
}


@pytest.mark.parametrize('precision', ['split_f16', 'f32', 'f16'])
@pytest.mark.parametrize('pattern', sorted(PATTERNS))
def test_skipping_empty_masks_changes_no_bit(ctx, precision, pattern):
    images, masks = synthetic.exemplars(4, k=3, size=64, seed=17, zero_every=0)
    images = images.reshape(12, 3, 64, 64)
    masks = masks.reshape(12, 1, 64, 64).clone()
    empty = PATTERNS[pattern]
    masks[empty] = 0
    ctx.set_precision(precision)
    ctx.set_fusion(skip_empty=False)
    want = ctx.encode(images, masks).cpu()
    ctx.set_fusion(skip_empty=True)
    got = ctx.encode(images, masks).cpu()
    assert torch.equal(got, want)
    if empty:
        assert (got[empty] == 0).all()
    keep = [i for i in range(12) if i not in empty]
    if keep:
        assert (got[keep].abs().amax(dim=1) > 0).all()


def test_skipped_batch_matches_the_oracle(ctx):
    """... and the oracle agrees: zero rows where the mask is empty, fp32-class elsewhere.
    A mask with a single lit pixel is NOT empty (its coarse levels may be)."""
    images, masks = synthetic.exemplars(2, k=4, size=64, seed=23, zero_every=0)
    masks = masks.clone()
    masks[0, 1] = 0
    masks[1, 3] = 0
    masks[1, 0] = 0
    masks[1, 0, 0, 40, 9] = 1
    with torch.no_grad():
        ref = O.encode(O.byte_to_float(images), masks.float(), ctx.sd, blocks=BLOCKS)
    ref = ref.reshape(8, -1)
    ctx.set_precision('split_f16')
    ctx.set_fusion(skip_empty=True)
    got = ctx.encode(images.reshape(8, 3, 64, 64), masks.reshape(8, 1, 64, 64)).cpu()
    assert (ref[[1, 7]] == 0).all() and (got[[1, 7]] == 0).all()
    assert got[4].abs().max() > 0
    e, where = feature_error(got, ref)
    assert e <= FEATURE_CLASS, (e, where)


def test_float_images_are_never_skipped(ctx):
    """A float image may carry a NaN pixel, which the reference turns into a NaN row even
    under a zero mask (tests/test_gpu_status.py): float inputs take the full pass."""
    images, masks = synthetic.exemplars(1, k=3, size=64, seed=29, zero_every=0)
    x = O.byte_to_float(images)[0].clone()
    masks = masks[0].clone()
    masks[1] = 0
    x[1, 0, 3, 3] = float('nan')
    ctx.set_precision('split_f16')
    ctx.set_fusion(skip_empty=True)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        got = ctx.encode(x, masks).cpu()
    assert torch.isnan(got[1]).all() and torch.isfinite(got[[0, 2]]).all()


def test_full_width_trunk_kernels_read_the_count_on_the_device(dev):
    """Round 6: how many images hold work is never read back -- every trunk launch is sized
    for the whole batch and reads the count itself (GemmArgs::m_live).  Full-width ResNet-101
    at 224 x 224 is where the hand-scheduled kernels run (uint8 stem, conv3_p64, chain /
    chain3, the ping-pong tiles, persistent 128-column tiles): skipping on == off, bit for
    bit, incl. an image count that ends inside a 256-row tile of every stage, one live
    image, and none."""
    sd = synthetic.resnet_state_dict('resnet101', seed=5, width=64, prefix=PREFIX)
    c = hip.Context(hip.make_dims(sd, 10, blocks=synthetic.RESNET_BLOCKS['resnet101']), sd, dev)
    try:
        images, masks = synthetic.exemplars(1, k=14, size=224, seed=31, zero_every=0)
        images, masks = images[0], masks[0].clone()
        for precision in ('split_f16', 'f16', 'f32'):
            c.set_precision(precision)
            for empty in ([3, 9], [i for i in range(14) if i != 6], list(range(14))):
                m = masks.clone()
                m[empty] = 0
                c.set_fusion(skip_empty=False)
                want = c.encode(images, m).cpu()
                c.set_fusion(skip_empty=True)
                got = c.encode(images, m).cpu()
                assert torch.equal(got, want), (precision, empty)
                assert (got[empty] == 0).all()
    finally:
        c.close()


def test_encode_enqueues_without_synchronising(ctx):
    """include/milan_hip.h: "all work is enqueued on `stream`; no implicit sync".  Round 5's
    skip-empty path read a count back (hipStreamSynchronize inside milan_encode); now the
    whole encoder pass can be captured into a hipGraph -- a capture fails on any
    synchronisation or read-back -- and the replay equals the eager call bit for bit."""
    images, masks = synthetic.exemplars(4, k=3, size=64, seed=17, zero_every=0)
    images = images.reshape(12, 3, 64, 64).cuda()
    masks = masks.reshape(12, 1, 64, 64).clone()
    masks[[2, 5]] = 0
    masks = masks.cuda()
    ctx.set_precision('split_f16')
    ctx.set_fusion(skip_empty=True)
    want = ctx.encode(images, masks, check=False).clone()   # (also sizes the workspace)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        ctx.encode(images, masks, check=False)
        with torch.cuda.graph(graph, stream=side):
            out = ctx.encode(images, masks, check=False)
    torch.cuda.current_stream().wait_stream(side)
    for replay in range(3):   # (a memset NODE went wrong from the second replay on: launch_zero_fill)
        out.fill_(float(replay + 5))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want), replay
    assert (out[[2, 5]] == 0).all() and ctx.status() == 0
    # a different set of empty masks through the SAME graph: the count is data, not a launch
    # parameter
    masks[[2, 5]] = 1
    masks[[0, 7, 11]] = 0
    graph.replay()
    torch.cuda.synchronize()
    ctx.set_fusion(skip_empty=False)
    again = ctx.encode(images, masks, check=False)
    ctx.set_fusion(skip_empty=True)
    assert torch.equal(out, again)
