"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and
the reference-generated goldens.  Run with `pytest -m gpu` on an MI355X.

Tolerances (fp32 both sides; differences are summation order + BN folding):
  conv kernel            rtol 1e-4 of the output scale
  encoder features       rtol 2e-3 / atol 2e-4 after ~100 fp32 conv layers
  decoder step log-probs atol 1e-4
  beam / rerank scores   atol 2e-3 (sums of <=15 log-probs of magnitude ~10)
  tokens                 identical, unless the oracle's own top-2 gap at the
                         first divergence is below 1e-4 (a genuine near-tie)
"""
import pytest
import torch
import torch.nn.functional as F

from milan_amd import hip, synthetic
from featclass import assert_feature_class
from oracle import milan_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    hip.load_library()  # loud if the .so is missing
    return hip.require_device('cuda')


def close(a, b, rtol, atol):
    torch.testing.assert_close(a.cpu(), b.cpu(), rtol=rtol, atol=atol)


# --------------------------------------------------------------------------
# implicit-GEMM conv kernel
# --------------------------------------------------------------------------
CONV_CASES = [
    # n, h, w, cin, cout, k, stride, pad
    (2, 56, 56, 64, 256, 1, 1, 0),  # 1x1 expand
    (2, 56, 56, 256, 64, 1, 1, 0),  # 1x1 reduce (N<=64 tile config)
    (3, 28, 28, 128, 128, 3, 1, 1),  # 3x3
    (2, 56, 56, 128, 128, 3, 2, 1),  # 3x3 stride 2
    (2, 56, 56, 256, 512, 1, 2, 0),  # downsample
    (2, 64, 64, 4, 64, 7, 2, 3),  # stem (Cin padded 3->4), K=196 not %32
    (1, 9, 11, 8, 20, 3, 1, 1),  # ragged everything, Cin%32 != 0
    (5, 7, 7, 512, 2048, 1, 1, 0),  # layer4 shape
    (1, 1, 1, 3904, 512, 1, 1, 0),  # a Linear as a 1x1 conv, M = 1
    # 3x3 / stride 1, N % 128 == 0: also run on the LDS-strip kernel below
    (3, 14, 14, 256, 256, 3, 1, 1),  # layer3 c2 shape: tiles span image borders
    (2, 28, 28, 128, 128, 3, 1, 1),  # layer2 c2 shape (256x128 tile)
    (5, 7, 7, 64, 512, 3, 1, 1),     # layer4-like: 4 images per tile
    (1, 9, 37, 32, 128, 3, 1, 1),    # oblong, ragged last tile
    (2, 61, 63, 32, 128, 3, 1, 1),   # the widest image the strip holds (W + 1 <= 64)
    (2, 27, 27, 64, 192, 5, 1, 2),   # AlexNet features.3: 25 taps
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_kernel_matches_torch_cpu(dev, case):
    n, h, w, cin, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k))**.5
    b = torch.randn(cout, generator=g)
    want = F.conv2d(x, wt, b, stride=stride, padding=pad)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    got = hip.conv2d_nhwc(x_nhwc, wt.to(dev), b.to(dev), stride, pad)
    got = got.permute(0, 3, 1, 2)
    close(got, want, rtol=1e-4, atol=1e-4)
    # relu + residual epilogue
    res = torch.randn_like(want)
    got = hip.conv2d_nhwc(x_nhwc,
                          wt.to(dev),
                          b.to(dev),
                          stride,
                          pad,
                          residual=res.permute(0, 2, 3, 1).contiguous().to(dev))
    close(got.permute(0, 3, 1, 2), F.relu(want + res), rtol=1e-4, atol=1e-4)


def test_conv_kernel_detects_transposes(dev):
    """A = identity-like check with asymmetric operands (MI355X guide rule 16)."""
    cin = cout = 64
    x = torch.zeros(1, cin, 4, 4)
    for c in range(cin):
        x[0, c, c % 4, (c // 4) % 4] = 1.0 + c
    wt = torch.zeros(cout, cin, 1, 1)
    for o in range(cout):
        wt[o, (o * 7 + 3) % cin, 0, 0] = 1.0 + 0.01 * o
    want = F.conv2d(x, wt)
    got = hip.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev),
                          wt.to(dev))
    assert torch.equal(got.permute(0, 3, 1, 2).cpu(), want)


# --------------------------------------------------------------------------
# encoder vs reference-generated goldens
# --------------------------------------------------------------------------
def _encoder_ctx(meta, dev):
    sd = synthetic.resnet_state_dict(meta['config'],
                                     seed=meta['weight_seed'],
                                     width=meta['width'],
                                     prefix='encoder.encoder.model.')
    dims = hip.make_dims(sd, 10, blocks=synthetic.RESNET_BLOCKS[meta['config']])
    return hip.Context(dims, sd, dev), sd


@pytest.mark.parametrize('tag', ['slim224', 'slim100', 'r50_64', 'full224'])
@pytest.mark.parametrize('as_u8', [True, False])
def test_encoder_matches_reference_golden(dev, goldens, golden_meta, tag,
                                          as_u8):
    m = golden_meta[f'g1_{tag}']
    ctx, _ = _encoder_ctx(m, dev)
    images_u8, _ = synthetic.exemplars(1,
                                       k=m['m'],
                                       size=m['size'],
                                       seed=m['image_seed'],
                                       zero_every=0)
    masks_u8 = goldens[f'g1_{tag}_masks_u8']
    if as_u8:
        got = ctx.encode(images_u8[0], masks_u8[0])
    else:
        got = ctx.encode(O.byte_to_float(images_u8[0]), masks_u8[0].float())
    want = goldens[f'g1_{tag}_features']
    assert_feature_class(got, want)
    assert got[1].eq(0).all(), 'all-zero mask must give an exactly-zero row'
    assert not torch.isnan(got).any()
    ctx.close()


def test_encoder_no_mask_equals_all_ones(dev, golden_meta):
    m = golden_meta['g1_slim224']
    ctx, sd = _encoder_ctx(m, dev)
    images_u8, _ = synthetic.exemplars(1, k=2, size=96, seed=3, zero_every=0)
    got = ctx.encode(images_u8[0], None)
    ones = torch.ones(2, 1, 96, 96)
    want = O.encode(O.byte_to_float(images_u8), ones.unsqueeze(0), sd)[0]
    assert_feature_class(got, want)
    ctx.close()


@pytest.mark.parametrize('hw', [(300, 260), (256, 256), (37, 450), (16, 16),
                                (9, 13), (1, 1)])
def test_encoder_large_and_oblong_images(dev, golden_meta, hw):
    """The reference takes any image size (the pyramid follows the trunk's
    strides); nothing here may assume 224x224."""
    h, w = hw
    m = golden_meta['g1_slim224']
    ctx, sd = _encoder_ctx(m, dev)
    g = torch.Generator().manual_seed(h + w)
    images = torch.randint(0, 256, (2, 3, h, w), dtype=torch.uint8, generator=g)
    masks = (torch.rand(2, 1, h, w, generator=g) > 0.5).to(torch.uint8)
    want = O.encode(O.byte_to_float(images)[None], masks[None].float(), sd)[0]
    for precision in ('f32', 'split_f16'):
        ctx.set_precision(precision)
        assert_feature_class(ctx.encode(images, masks), want, what=precision)
    ctx.close()


# --------------------------------------------------------------------------
# decoder vs goldens / oracle
# --------------------------------------------------------------------------
def _dec(meta, dev, lm=True):
    v = meta['nvocab'] + 4
    sd = synthetic.decoder_state_dict(v,
                                      feature_size=meta['feature_size'],
                                      hidden_size=meta['hidden'],
                                      embedding_size=meta['emb'],
                                      lm=lm,
                                      lm_hidden_size=meta['hidden'],
                                      lm_embedding_size=meta['emb'],
                                      seed=meta['weight_seed'])
    g = torch.Generator().manual_seed(meta['feat_seed'])
    feats = torch.rand(meta['b'], meta['k'], meta['feature_size'], generator=g)
    dims = hip.make_dims(sd, meta['nvocab'])
    return hip.Context(dims, sd, dev), sd, feats, meta['nvocab']


MAX_TIE_ROWS = 1  # near-tie excuses allowed per test (rows / neurons)


def assert_tokens_match(got, want, gap=None, what='tokens',
                        max_ties=MAX_TIE_ROWS):
    """Identical, or first divergence sits on an oracle near-tie.  Returns a
    bool mask of the rows that are identical; at most MAX_TIE_ROWS rows may
    take the near-tie excuse (so a pass never means "nothing was compared")."""
    got, want = got.cpu(), want.cpu()
    same = (got == want).all(dim=1)
    if bool(same.all()):
        return same
    assert gap is not None, f'{what} differ and no tie information available'
    for b in range(want.shape[0]):
        diff = (got[b] != want[b]).nonzero()
        if len(diff):
            t = int(diff[0])
            assert gap[b, t] < 1e-4, (
                f'{what} diverge at row {b} step {t} where the oracle top-2 gap '
                f'is {float(gap[b, t]):.3g} (not a near-tie)')
    assert int((~same).sum()) <= max_ties, (
        f'{int((~same).sum())} rows of {what} took the near-tie excuse')
    return same


@pytest.mark.parametrize('size', ['small', 'full'])
def test_init_state_and_step_match_reference_golden(dev, goldens, golden_meta,
                                                    size):
    ctx, sd, feats, nv = _dec(golden_meta[f'dec_{size}'], dev)
    h, c = ctx.init_state(feats)
    close(h, goldens[f'g2_{size}_h'], 1e-4, 1e-5)
    close(c, goldens[f'g2_{size}_c'], 1e-4, 1e-5)
    pred, att, h2, c2, _, _ = ctx.step(feats, goldens[f'g3_{size}_tokens'],
                                       goldens[f'g2_{size}_h'],
                                       goldens[f'g2_{size}_c'], None, None, 0.2)
    close(pred, goldens[f'g3_{size}_pred'], 1e-4, 1e-4)
    close(att, goldens[f'g3_{size}_att'], 1e-4, 1e-5)
    close(h2, goldens[f'g3_{size}_h'], 1e-4, 1e-5)
    ctx.close()


def test_step_mi_branch_matches_reference_golden(dev, goldens, golden_meta):
    ctx, sd, feats, nv = _dec(golden_meta['dec_small'], dev)
    n = len(feats)
    hid = golden_meta['dec_small']['hidden']
    pred, _, _, _, hlm, clm = ctx.step(feats, goldens['g3_small_tokens'],
                                       goldens['g2_small_h'],
                                       goldens['g2_small_c'],
                                       torch.zeros(2, n, hid),
                                       torch.zeros(2, n, hid), 0.3)
    close(pred, goldens['g3_small_mi_pred'], 1e-4, 1e-4)
    close(hlm, goldens['g3_small_mi_hlm'], 1e-4, 1e-5)
    close(clm, goldens['g3_small_mi_clm'], 1e-4, 1e-5)
    ctx.close()


@pytest.mark.parametrize('mi', [False, True])
def test_greedy_small_matches_reference_golden(dev, goldens, golden_meta, mi):
    ctx, sd, feats, nv = _dec(golden_meta['dec_small'], dev)
    tag = 'g4_small_mi' if mi else 'g4_small'
    out = ctx.decode(feats, hip.GREEDY, 15, 1, mi, 0.2)
    top2 = goldens[tag + '_pred'].topk(2, dim=-1).values
    same = assert_tokens_match(out['tokens'], goldens[tag + '_tokens'],
                               top2[..., 0] - top2[..., 1])
    assert same.any()
    close(out['scores'][same], goldens[tag + '_scores'][same], 1e-4, 1e-3)
    close(out['predictions'][same], goldens[tag + '_pred'][same], 1e-4, 2e-4)
    close(out['attentions'][same], goldens[tag + '_att'][same], 1e-4, 1e-5)
    ctx.close()


def test_greedy_full_matches_reference_golden(dev, goldens, golden_meta):
    ctx, sd, feats, nv = _dec(golden_meta['dec_full'], dev)
    out = ctx.decode(feats, hip.GREEDY, 15, 1, False, 0.2)
    same = assert_tokens_match(out['tokens'], goldens['g4_full_tokens'],
                               goldens['g4_full_top2gap'])
    assert same.any()
    close(out['scores'][same], goldens['g4_full_scores'][same], 1e-4, 1e-3)
    close(out['attentions'][same], goldens['g4_full_att'][same], 1e-4, 1e-5)
    seqs = torch.cat(
        [torch.full((3, 1), nv, dtype=torch.long), goldens['g4_full_tokens']],
        1)
    close(ctx.lm_score(seqs), goldens['g5_full_lm_scores'], 1e-4, 1e-3)
    ctx.close()


def test_lm_score_matches_reference_golden_incl_stop_quirk(
        dev, goldens, golden_meta):
    ctx, sd, feats, nv = _dec(golden_meta['dec_small'], dev)
    got = ctx.lm_score(goldens['g5_small_seqs'])
    close(got, goldens['g5_small_lm_scores'], 1e-4, 1e-4)
    ctx.close()


def _check_beams(out, want_tokens, want_scores, length_used,
                 max_ties=MAX_TIE_ROWS):
    """Set-equal beams with matching scores.  A beam may be missing only if it
    sat on the score boundary of the oracle's beam (within 2e-3 of the last
    kept score), and at most MAX_TIE_ROWS neurons may use that excuse.
    Returns a bool mask of the neurons whose beams are identical, in order.
    (The goldens of the reference's own search are checked more strictly, with
    recorded selection margins, in tests/test_gpu_beam_goldens.py.)"""
    bt = out['beam_tokens'].cpu()[:, :, :length_used]
    bs = out['beam_scores'].cpu()
    close(bs, want_scores, 1e-4, 2e-3)
    n, beam, _ = bt.shape
    excused = 0
    for i in range(n):
        got = {tuple(bt[i, j].tolist()) for j in range(beam)}
        want = {tuple(want_tokens[i, j].tolist()) for j in range(beam)}
        missing = want - got
        # a beam may legitimately differ only if it was on the score boundary
        if missing:
            excused += 1
            edge = want_scores[i, -1]
            for seq in missing:
                j = [tuple(t.tolist()) for t in want_tokens[i]].index(seq)
                assert abs(want_scores[i, j] - edge) < 2e-3, (
                    f'neuron {i}: beam {j} missing from the HIP beam set')
    assert excused <= max_ties, (
        f'{excused} neurons had boundary beams swapped')
    return (bt == want_tokens).all(dim=2).all(dim=1)


def _check_rerank(out, want_t, want_s, sd, nv, same, tprime):
    """Top-1 after the rerank: identical tokens for every neuron whose beams
    are identical, unless the oracle's own top-2 PMI gap is a near-tie."""
    assert same.any(), 'no neuron left to compare the rerank choice on'
    t, s, choice = O.rerank(want_t, want_s, sd, nv, nv + 1, 0.2)
    b, beam, _ = want_t.shape
    starts = want_t.new_full((b, beam, 1), nv)
    seqs = torch.cat([starts, want_t], dim=-1).view(b * beam, -1)
    pmi = want_s - 0.2 * O.lm_score(seqs, sd, nv + 1).view(b, beam)
    top2 = pmi.topk(2, dim=-1).values if beam > 1 else None
    excused = 0
    for i in range(b):
        if not same[i]:
            continue
        if torch.equal(out['tokens'].cpu()[i, :tprime], t[i]):
            close(out['scores'][i], s[i], 1e-4, 3e-3)
        else:
            assert top2 is not None and float(top2[i, 0] - top2[i, 1]) < 1e-3, (
                f'neuron {i}: rerank picked another beam without a near-tie')
            excused += 1
    assert excused <= MAX_TIE_ROWS


@pytest.mark.parametrize('size,beam,length', [('small', 5, 8), ('small', 3, 15),
                                              ('full', 16, 15),
                                              ('full', 50, 15)])
def test_beam_search_and_rerank_match_oracle(dev, golden_meta, size, beam,
                                             length):
    ctx, sd, feats, nv = _dec(golden_meta[f'dec_{size}'], dev)
    want_t, want_s = O.beam_search(feats, sd, nv, nv + 1, length, beam)
    out = ctx.decode(feats, hip.RERANK, length, beam, False, 0.2)
    tprime = want_t.shape[2]
    assert int(out['out_len'][0]) == tprime
    # beyond T' the HIP search pads with <stop>
    assert (out['beam_tokens'][:, :, tprime:] == nv + 1).all()
    same = _check_beams(out, want_t, want_s, tprime)
    _check_rerank(out, want_t, want_s, sd, nv, same, tprime)
    # plain beam strategy = top beam
    out2 = ctx.decode(feats, hip.BEAM, length, beam, False, 0.2)
    assert torch.equal(out2['tokens'], out2['beam_tokens'][:, 0])
    close(out2['scores'], out2['beam_scores'][:, 0], 0, 0)
    ctx.close()


def test_beam_one_equals_greedy_prefix(dev, golden_meta):
    ctx, sd, feats, nv = _dec(golden_meta['dec_small'], dev)
    g = ctx.decode(feats, hip.GREEDY, 15, 1, False, 0.2)['tokens'].cpu()
    b = ctx.decode(feats, hip.BEAM, 15, 1, False, 0.2)['tokens'].cpu()
    for i in range(len(feats)):
        for t in range(15):
            assert b[i, t] == g[i, t]
            if b[i, t] == nv + 1:
                break
    ctx.close()


def test_beam_search_with_mi_matches_oracle(dev, golden_meta):
    ctx, sd, feats, nv = _dec(golden_meta['dec_small'], dev)
    want_t, want_s = O.beam_search(feats, sd, nv, nv + 1, 10, 4, mi=True,
                                   temperature=0.2)
    out = ctx.decode(feats, hip.BEAM, 10, 4, True, 0.2)
    _check_beams(out, want_t, want_s, want_t.shape[2])
    ctx.close()


def test_error_conventions(dev, golden_meta):
    """Same exception types as the reference (decoders.py:395-409)."""
    ctx, sd, feats, nv = _dec(golden_meta['dec_small'], dev, lm=False)
    with pytest.raises(ValueError, match='without an LM'):
        ctx.decode(feats, hip.RERANK, 5, 3, False, 0.2)
    with pytest.raises(ValueError, match='unknown strategy'):
        ctx.decode(feats, 7, 5, 3, False, 0.2)
    ctx.close()
    ctx, sd, feats, nv = _dec(golden_meta['dec_small'], dev, lm=True)
    with pytest.raises(ValueError, match='cannot set `mi=` decoding'):
        ctx.decode(feats, hip.RERANK, 5, 3, True, 0.2)
    ctx.close()


# --------------------------------------------------------------------------
# the whole path: images+masks -> captions, slim trunk so the oracle is fast
# --------------------------------------------------------------------------
@pytest.mark.parametrize('strategy', ['greedy', 'rerank'])
def test_describe_end_to_end_matches_oracle(dev, strategy):
    nv, width, k, n, size = 60, 16, 5, 6, 96
    sd = synthetic.milan_state_dict(nv + 4,
                                    config='resnet50',
                                    seed=11,
                                    width=width,
                                    hidden_size=64,
                                    embedding_size=16,
                                    lm_hidden_size=64,
                                    lm_embedding_size=16)
    dims = hip.make_dims(sd, nv, blocks=synthetic.RESNET_BLOCKS['resnet50'])
    ctx = hip.Context(dims, sd, dev)
    images, masks = synthetic.exemplars(n, k=k, size=size, seed=5, zero_every=7)
    feats = O.encode(O.byte_to_float(images), masks.float(), sd,
                     blocks=synthetic.RESNET_BLOCKS['resnet50'])
    want = O.forward(feats, sd, nv, strategy, length=10, beam_size=4, mi=False)
    out = ctx.describe(images,
                       masks,
                       hip.GREEDY if strategy == 'greedy' else hip.RERANK,
                       10,
                       4,
                       False,
                       0.2,
                       want_full=True,
                       want_features=True)
    close(out['features'], feats, 2e-3, 2e-4)
    if strategy == 'greedy':
        top2 = want['predictions'].topk(2, dim=-1).values
        assert_tokens_match(out['tokens'], want['tokens'],
                            top2[..., 0] - top2[..., 1])
    else:
        tp = want['beam_tokens'].shape[2]
        same = _check_beams(out, want['beam_tokens'], want['beam_scores'], tp)
        _check_rerank(out, want['beam_tokens'], want['beam_scores'], sd, nv,
                      same, tp)
    ctx.close()


# --------------------------------------------------------------------------
# split-f16 precision mode (3 x f16 MFMA on (hi,lo) pairs)
# --------------------------------------------------------------------------
SPLIT_CASES = [c for c in CONV_CASES if c[3] % 32 == 0]


@pytest.mark.parametrize('case', SPLIT_CASES)
def test_split_f16_conv_error_is_fp32_class(dev, case):
    """Against an fp64 reference the split kernel's error must be of the same
    class as the exact-fp32 MFMA kernel's own rounding error."""
    n, h, w, cin, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(7 + sum(case))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k))**.5
    b = torch.randn(cout, generator=g)
    want = F.conv2d(x.double(), wt.double(), b.double(), stride=stride,
                    padding=pad)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    errs = {}
    for prec in ('f32', 'split_f16'):
        got = hip.conv2d_nhwc(x_nhwc, wt.to(dev), b.to(dev), stride, pad,
                              precision=prec).permute(0, 3, 1, 2).cpu().double()
        errs[prec] = float((got - want).abs().max())
    scale = float(want.abs().max())
    assert errs['split_f16'] <= max(4 * errs['f32'], 2e-6 * scale), errs
    assert errs['split_f16'] <= 1e-5 * scale, errs


KXK_CASES = [c for c in SPLIT_CASES if c[5] > 1]


@pytest.mark.parametrize('case', KXK_CASES)
def test_tap_inner_k_order_is_the_same_class_as_tap_major(dev, case):
    """The trunk's k x k convs run with k in (32-channel slice, tap, channel) order (every
    shifted copy of a slice is requested in consecutive k-tile pairs and hits L2: layer3's
    3x3 convs pulled their input 8.3 x over the fabric in tap-major order).  Same products,
    another order of the fp32 additions: both orders are fp32-class against fp64, over
    strides, pads, image borders inside a tile and ragged last tiles."""
    assert len(KXK_CASES) >= 4
    n, h, w, cin, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(23 + sum(case))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k))**.5
    b = torch.randn(cout, generator=g)
    want = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=pad)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    scale = float(want.abs().max())
    err = {}
    for prec in ('split_f16', 'split_f16_tap_major', 'f32'):
        got = hip.conv2d_nhwc(x_nhwc, wt.to(dev), b.to(dev), stride, pad,
                              precision=prec).permute(0, 3, 1, 2).cpu().double()
        err[prec] = float((got - want).abs().max())
    print(case, {p: f'{e / scale:.2e}' for p, e in err.items()})
    assert err['split_f16'] <= max(2 * err['split_f16_tap_major'], 2e-6 * scale), err
    assert err['split_f16'] <= max(4 * err['f32'], 2e-6 * scale), err


STRIP_CASES = [c for c in CONV_CASES
               if c[5] == 3 and c[6] == 1 and c[3] % 32 == 0 and c[4] % 128 == 0]


@pytest.mark.parametrize('case', STRIP_CASES)
def test_lds_strip_conv3x3_kernel(dev, case):
    """The 3x3 kernel with the input strip resident in LDS (a measured-and-rejected
    experiment; compiled with `make EXPERIMENTS=1`) stays correct: fp32-class error
    against an fp64 reference, tiles spanning image borders, ragged tiles."""
    assert len(STRIP_CASES) >= 5
    n, h, w, cin, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(11 + sum(case))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k))**.5
    b = torch.randn(cout, generator=g)
    want = F.conv2d(x.double(), wt.double(), b.double(), stride=1, padding=1)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    got = {}
    for prec in ('split_f16', 'split_f16_strip'):
        try:
            y = hip.conv2d_nhwc(x_nhwc, wt.to(dev), b.to(dev), 1, 1, precision=prec)
        except ValueError as error:
            if 'experiments build' in str(error):
                pytest.skip('the LDS-strip kernel is a rejected experiment: only in '
                            'a `make EXPERIMENTS=1` build of the library')
            raise
        got[prec] = y.permute(0, 3, 1, 2).cpu().double()
    scale = float(want.abs().max())
    err = {p: float((v - want).abs().max()) for p, v in got.items()}
    assert err['split_f16_strip'] <= max(2 * err['split_f16'], 2e-6 * scale), err
    assert not torch.equal(got['split_f16'], got['split_f16_strip']), \
        'the strip kernel was not selected (chunk-major K order differs in bits)'


def test_split_f16_handles_tiny_and_large_values(dev):
    """Operand magnitudes far from 1: weights are rescaled by a power of two,
    activations saturate instead of overflowing to inf."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 8, 8, generator=g)
    x[0] *= 1e-3
    x[1] *= 300.0
    for wmag in (1e-4, 1.0, 50.0):
        wt = torch.randn(64, 64, 1, 1, generator=g) * wmag
        want = F.conv2d(x.double(), wt.double())
        got = hip.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev),
                              wt.to(dev), precision='split_f16')
        got = got.permute(0, 3, 1, 2).cpu().double()
        # image 0 (|x| ~ 1e-3): the lo halves fall into the f16 subnormal range,
        # so the error floor is absolute (~3e-8 per element) and the relative
        # error degrades gracefully; image 1 (|x| ~ 300) keeps full accuracy
        # (DESIGN.md section 4).
        for i, bound in ((0, 1e-4), (1, 5e-6)):
            rel = float((got[i] - want[i]).abs().max() / want[i].abs().max())
            assert rel < bound, (wmag, i, rel)


@pytest.mark.parametrize('tag', ['slim224', 'r50_64', 'full224'])
def test_split_f16_encoder_matches_reference_golden(dev, goldens, golden_meta,
                                                    tag):
    m = golden_meta[f'g1_{tag}']
    if m['width'] % 8:
        pytest.skip('split mode needs width % 8 == 0')
    ctx, _ = _encoder_ctx(m, dev)
    ctx.set_precision('split_f16')
    assert ctx.precision == 'split_f16'
    images_u8, _ = synthetic.exemplars(1, k=m['m'], size=m['size'],
                                       seed=m['image_seed'], zero_every=0)
    got = ctx.encode(images_u8[0], goldens[f'g1_{tag}_masks_u8'][0])
    want = goldens[f'g1_{tag}_features']
    assert_feature_class(got, want, what='split_f16')
    assert got[1].eq(0).all()
    ctx.set_precision('f32')
    got32 = ctx.encode(images_u8[0], goldens[f'g1_{tag}_masks_u8'][0])
    # report how the two modes compare against the reference
    e32 = float((got32.cpu() - want).abs().max())
    esp = float((got.cpu() - want).abs().max())
    print(f'{tag}: max|err| f32={e32:.3g} split_f16={esp:.3g} '
          f'(feature max {float(want.abs().max()):.3g})')
    ctx.close()


@pytest.mark.parametrize('hw', [(96, 96), (75, 101), (64, 33)])
def test_split_f16_pixel_pair_stem(dev, hw):
    """Split mode runs the 7x7/2 stem as a KH=7 x KW=4 conv over pixel-pair
    groups (encoder.hip).  Odd widths exercise the trailing half-empty group
    and the right-hand padding; the raw conv1 tap (first `width` feature
    columns, nethook's pre-BN 'conv1') must agree with the fp32 path to
    fp32-class error, and u8 / float inputs must give identical bits."""
    h, w = hw
    width, k = 16, 3
    blocks = synthetic.RESNET_BLOCKS['resnet50']
    sd = synthetic.resnet_state_dict('resnet50', seed=5, width=width,
                                     prefix='encoder.encoder.model.')
    ctx = hip.Context(hip.make_dims(sd, 10, blocks=blocks), sd, dev)
    g = torch.Generator().manual_seed(h * 1000 + w)
    images = torch.randint(0, 256, (k, 3, h, w), dtype=torch.uint8, generator=g)
    masks = (torch.rand(k, 1, h, w, generator=g) > 0.6).to(torch.uint8)
    want = O.encode(O.byte_to_float(images)[None], masks[None].float(), sd,
                    blocks=blocks)[0]
    f32 = ctx.encode(images, masks).cpu()
    ctx.set_precision('split_f16')
    sp_u8 = ctx.encode(images, masks).cpu()
    sp_f = ctx.encode(O.byte_to_float(images), masks.float()).cpu()
    assert torch.equal(sp_u8, sp_f)
    assert_feature_class(sp_u8, want, what='split_f16')
    tap0 = slice(0, width)
    scale = float(want[:, tap0].abs().max())
    assert float((sp_u8[:, tap0] - f32[:, tap0]).abs().max()) < 2e-6 * max(scale, 1.0)
    close(sp_u8[:, tap0], want[:, tap0], rtol=1e-4, atol=1e-5)
    ctx.close()


def test_split_f16_describe_end_to_end(dev):
    nv, width, k, n, size = 60, 32, 5, 6, 96
    blocks = synthetic.RESNET_BLOCKS['resnet50']
    sd = synthetic.milan_state_dict(nv + 4, config='resnet50', seed=12,
                                    width=width, hidden_size=64,
                                    embedding_size=32, lm_hidden_size=64,
                                    lm_embedding_size=32)
    ctx = hip.Context(hip.make_dims(sd, nv, blocks=blocks), sd, dev)
    ctx.set_precision('split_f16')
    images, masks = synthetic.exemplars(n, k=k, size=size, seed=6, zero_every=7)
    feats = O.encode(O.byte_to_float(images), masks.float(), sd, blocks=blocks)
    want = O.forward(feats, sd, nv, 'greedy', length=10, mi=False)
    out = ctx.describe(images, masks, hip.GREEDY, 10, 1, False, 0.2,
                       want_full=True, want_features=True)
    close(out['features'], feats, 2e-3, 2e-4)
    top2 = want['predictions'].topk(2, dim=-1).values
    assert_tokens_match(out['tokens'], want['tokens'],
                        top2[..., 0] - top2[..., 1])
    ctx.close()


def test_split_f16_decoder_matches_reference_goldens(dev, goldens, golden_meta):
    """Decoder / LM GEMMs through the 3xf16 path: same goldens, same bounds."""
    ctx, sd, feats, nv = _dec(golden_meta['dec_full'], dev)
    ctx.set_precision('split_f16')
    h, c = ctx.init_state(feats)
    close(h, goldens['g2_full_h'], 1e-4, 1e-5)
    pred, att, h2, _, _, _ = ctx.step(feats, goldens['g3_full_tokens'],
                                      goldens['g2_full_h'],
                                      goldens['g2_full_c'], None, None, 0.2)
    close(pred, goldens['g3_full_pred'], 1e-4, 1e-4)
    close(att, goldens['g3_full_att'], 1e-4, 1e-5)
    close(h2, goldens['g3_full_h'], 1e-4, 1e-5)
    out = ctx.decode(feats, hip.GREEDY, 15, 1, False, 0.2)
    assert_tokens_match(out['tokens'], goldens['g4_full_tokens'],
                        goldens['g4_full_top2gap'])
    close(out['scores'], goldens['g4_full_scores'], 1e-4, 1e-3)
    seqs = torch.cat(
        [torch.full((3, 1), nv, dtype=torch.long), goldens['g4_full_tokens']],
        1)
    close(ctx.lm_score(seqs), goldens['g5_full_lm_scores'], 1e-4, 1e-3)
    want_t, want_s = O.beam_search(feats, sd, nv, nv + 1, 15, 16)
    outb = ctx.decode(feats, hip.RERANK, 15, 16, False, 0.2)
    _check_beams(outb, want_t, want_s, want_t.shape[2])
    ctx.close()


@pytest.mark.parametrize('n,hw,cin,cout,res', [
    (8, 96, 256, 512, False),    # 288 x 2 tiles of 256 x 256 on 256 CUs: persistent workgroups
    (8, 95, 256, 1024, True),    # ragged last row tile, residual epilogue (an expand conv)
    (40, 28, 1024, 256, False),  # K = 1024, one column of tiles (a reduce conv)
])
def test_persistent_1x1_kernel_equals_the_one_tile_per_workgroup_kernel(dev, n, hw, cin,
                                                                         cout, res):
    """1x1 convolutions with more 256 x 256 tiles than CUs run as persistent workgroups
    that prefetch the next tile's A rows (igemm_split16_linp_kernel); the same rows
    computed image by image stay below that threshold and take the plain kernel.  Same
    k order, same epilogue: the results must agree bit for bit -- and with fp64."""
    g = torch.Generator().manual_seed(n * hw + cin)
    x = torch.randn(n, hw, hw, cin, generator=g).to(dev)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    r = torch.randn(n, hw, hw, cout, generator=g).to(dev) if res else None
    whole = hip.conv2d_nhwc(x, wt, b, 1, 0, relu=True, residual=r, precision='split_f16')
    parts = torch.cat([
        hip.conv2d_nhwc(x[i:i + 1], wt, b, 1, 0, relu=True,
                        residual=None if r is None else r[i:i + 1], precision='split_f16')
        for i in range(n)])
    assert torch.equal(whole, parts)
    want = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), b.double())
    if r is not None:
        want = want + r.permute(0, 3, 1, 2).double()
    want = want.relu().permute(0, 2, 3, 1)
    assert (whole.double() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())
