"""Top-k selection edge cases of the decode kernels (ties, flat rows)."""
import pytest
import torch

from milan_amd import hip, synthetic
from oracle import milan_oracle as O

pytestmark = pytest.mark.gpu


def _ctx(nv, hidden=64, emb=16, fsize=244, zero_output=False, seed=3):
    sd = synthetic.decoder_state_dict(nv + 4, feature_size=fsize,
                                      hidden_size=hidden, embedding_size=emb,
                                      lm_hidden_size=hidden,
                                      lm_embedding_size=emb, seed=seed)
    if zero_output:  # every logit equal: the whole row ties
        sd['output.1.weight'].zero_()
        sd['output.1.bias'].zero_()
    return hip.Context(hip.make_dims(sd, nv), sd, 'cuda'), sd


def test_all_equal_logits_pick_lowest_indices():
    """Flat distribution: top-k must be token ids 0..k-1 in order (ties ->
    lowest index), greedy must pick id 0 every step."""
    nv = 300
    ctx, sd = _ctx(nv, zero_output=True)
    feats = torch.rand(3, 5, 244)
    g = ctx.decode(feats, hip.GREEDY, 4, 1, False, 0.2)
    assert g['tokens'].eq(0).all()
    want = torch.log(torch.tensor(1.0 / (nv + 4)))
    torch.testing.assert_close(g['predictions'].cpu(),
                               want.expand(3, 4, nv + 4), rtol=1e-5, atol=1e-5)
    out = ctx.decode(feats, hip.BEAM, 3, 20, False, 0.2)
    # every candidate ties at every step, so the lowest flat indices win:
    # step 0 keeps classes 0..19, then each merge keeps the 20 candidates of
    # beam 0 (classes 0..19, backpointer 0) => beam j reads [0, 0, j]
    bt = out['beam_tokens'].cpu()
    assert bt[:, :, 0].eq(0).all() and bt[:, :, 1].eq(0).all()
    assert bt[:, :, 2].eq(torch.arange(20)).all()
    ctx.close()


@pytest.mark.parametrize('nv,k', [(40, 5), (1000, 50), (6000, 64), (7000, 16)])
def test_topk_matches_torch_over_vocab_sizes(nv, k):
    """Register kernel (V <= 6144) and LDS kernel (V > 6144) vs the oracle."""
    ctx, sd = _ctx(nv, seed=nv)
    g = torch.Generator().manual_seed(nv)
    feats = torch.rand(4, 5, 244, generator=g)
    want_t, want_s = O.beam_search(feats, sd, nv, nv + 1, 2, k)
    out = ctx.decode(feats, hip.BEAM, 2, k, False, 0.2)
    torch.testing.assert_close(out['beam_scores'].cpu(), want_s, rtol=1e-4,
                               atol=1e-3)
    assert torch.equal(out['beam_tokens'].cpu()[:, :, :want_t.shape[2]], want_t)
    ctx.close()


@pytest.mark.parametrize('group,k', [(30, 50), (7, 50), (100, 16), (2, 120)])
def test_runs_of_equal_logits_straddling_the_kth_value(group, k):
    """Row-independent logits in runs of `group` equal values (index // group
    decides the value): the top-k is exactly token ids 0..k-1 in order -- the
    k-th value's run is cut at its lowest indices -- at the real vocabulary
    size, through the register kernel's rank-count path (and, for long runs,
    its threshold fallback)."""
    nv = 5000
    sd = synthetic.decoder_state_dict(nv + 4, feature_size=244, hidden_size=64,
                                      embedding_size=16, lm_hidden_size=64,
                                      lm_embedding_size=16, seed=9)
    sd['output.1.weight'].zero_()
    sd['output.1.bias'].copy_(-(torch.arange(nv + 4) // group).float() * 0.25)
    ctx = hip.Context(hip.make_dims(sd, nv), sd, 'cuda')
    feats = torch.rand(3, 5, 244)
    for precision in ('f32', 'split_f16'):
        ctx.set_precision(precision)
        out = ctx.decode(feats, hip.BEAM, 1, k, False, 0.2)
        bt = out['beam_tokens'].cpu()[:, :, 0]
        assert bt.eq(torch.arange(k)).all(), (precision, bt[0])
        want = torch.log_softmax(sd['output.1.bias'], 0)[:k]
        torch.testing.assert_close(out['beam_scores'].cpu(),
                                   want.expand(3, k), rtol=1e-5, atol=1e-5)
    ctx.close()
