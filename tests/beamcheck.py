"""Shared checker for beam-search / rerank results against goldens G14
(tests/golden/make_golden_beam.py: the reference's own `Decoder.forward`
running on the class-shaped allennlp stand-in).

Bar: token ids identical, scores within ATOL.  The only excused differences
are genuine near-ties, decided from margins recorded WITH the goldens (never
from the result under test):
  * a beam SET may differ only if the search's smallest selection margin for
    that neuron (gap between the last kept and first dropped candidate, over
    all steps) is below TIE;
  * two beams may swap positions only if their golden scores differ by < TIE;
  * a rerank choice may differ only if the golden top-2 PMI gap is < TIE.
Every excuse taken is counted and returned; callers assert an explicit bound
on the count, so "passed" can never mean "silently skipped".
"""
import torch

TIE = 1e-3   # a score gap below this is a near-tie in fp32 sums of ~15 terms
ATOL = 2e-3  # beam / rerank score tolerance (sums of <= 15 log-probs ~ -3)


def check_search(got, goldens, tag, strategy, stop_index):
    """`got`: dict with tokens (B,>=T'), scores (B,), beam_tokens (B,beam,>=T'),
    beam_scores (B,beam), tprime (int).  Returns the number of near-tie
    excuses taken (0 = everything identical)."""
    want_bt = goldens[tag + '_beam_tokens']
    want_bs = goldens[tag + '_beam_scores']
    want_t = goldens[tag + '_tokens']
    want_s = goldens[tag + '_scores']
    margin = goldens[tag + '_select_margin']
    b, beam, tp = want_bt.shape
    assert int(got['tprime']) == tp, (
        f'{tag}: early-exit length {got["tprime"]} != reference {tp}')
    bt = got['beam_tokens'].cpu()
    assert (bt[:, :, tp:] == stop_index).all(), \
        f'{tag}: tokens beyond T\' must be <stop> padding'
    bt = bt[:, :, :tp]
    bs = got['beam_scores'].cpu()
    tokens = got['tokens'].cpu()[:, :tp]
    scores = got['scores'].cpu()
    assert (bs[:, :-1] >= bs[:, 1:]).all(), f'{tag}: beam scores not sorted'
    excuses = 0
    for i in range(b):
        beams_equal = torch.equal(bt[i], want_bt[i])
        set_equal = beams_equal
        if not beams_equal:
            got_set = {tuple(r.tolist()) for r in bt[i]}
            want_set = {tuple(r.tolist()) for r in want_bt[i]}
            set_equal = got_set == want_set
            if set_equal:
                # a permutation: every displaced beam must sit in a near-tie
                for j in range(beam):
                    if torch.equal(bt[i, j], want_bt[i, j]):
                        continue
                    k = [tuple(r.tolist()) for r in want_bt[i]].index(
                        tuple(bt[i, j].tolist()))
                    gap = abs(float(want_bs[i, j] - want_bs[i, k]))
                    assert gap < TIE, (
                        f'{tag}: neuron {i} beams {j}/{k} swapped although '
                        f'their reference scores differ by {gap:.3g}')
            else:
                assert float(margin[i]) < TIE, (
                    f'{tag}: neuron {i} beam set differs from the reference '
                    f'(smallest selection margin {float(margin[i]):.3g}: not a '
                    f'near-tie); missing {sorted(want_set - got_set)[:2]}')
            excuses += 1
        if set_equal:
            torch.testing.assert_close(bs[i], want_bs[i], rtol=1e-4, atol=ATOL,
                                       msg=lambda m: f'{tag} neuron {i}: {m}')
        # top-1
        if torch.equal(tokens[i], want_t[i]):
            torch.testing.assert_close(scores[i], want_s[i], rtol=1e-4,
                                       atol=ATOL + 1e-3)
            continue
        if strategy == 'rerank':
            tie = float(goldens[tag + '_rerank_margin'][i]) < TIE
        else:  # plain beam: top-1 = best beam; a tie between beams 0 and 1
            tie = abs(float(want_bs[i, 0] - want_bs[i, 1])) < TIE
        assert tie or not set_equal, (
            f'{tag}: neuron {i} top-1 tokens differ from the reference '
            'without a near-tie')
        if beams_equal:
            excuses += 1  # (already counted otherwise)
    return excuses
