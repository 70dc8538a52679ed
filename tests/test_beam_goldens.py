"""CPU: the oracle's beam search / rerank against goldens G14 (the reference's
own `Decoder.forward(strategy='beam'|'rerank')` + `predict`, run unmodified in
tests/golden/make_golden_beam.py on the allennlp stand-in), and both search
restatements against the hand-worked known-answer case."""
import importlib.util
import json

import pytest
import torch

from beamcheck import check_search
from conftest import GOLDEN_DIR
from milan_amd import synthetic
from oracle import milan_oracle as O


@pytest.fixture(scope='module')
def G():
    return torch.load(GOLDEN_DIR / 'reference_goldens_beam.pt')


@pytest.fixture(scope='module')
def M():
    with open(GOLDEN_DIR / 'reference_goldens_beam.json') as f:
        return json.load(f)


def standin():
    spec = importlib.util.spec_from_file_location(
        'allennlp_standin', GOLDEN_DIR / 'allennlp_standin.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def model(meta):
    v = meta['nvocab'] + 4
    sd = synthetic.decoder_state_dict(v, feature_size=meta['feature_size'],
                                      hidden_size=meta['hidden'],
                                      embedding_size=meta['emb'], lm=True,
                                      lm_hidden_size=meta['hidden'],
                                      lm_embedding_size=meta['emb'],
                                      seed=meta['weight_seed'])
    if meta.get('stop_bias'):
        sd['output.1.bias'] = sd['output.1.bias'].clone()
        sd['output.1.bias'][meta['nvocab'] + 1] += meta['stop_bias']
    g = torch.Generator().manual_seed(meta['feat_seed'])
    feats = torch.rand(meta['b'], meta['k'], meta['feature_size'], generator=g)
    return sd, feats, meta['nvocab']


SEARCHES = [
    ('dec_small', 'g14_small_beam3'), ('dec_small', 'g14_small_beam4_mi'),
    ('dec_small', 'g14_small_rerank5'), ('dec_small', 'g14_small_beam1'),
    ('dec_small', 'g14_small_beam_default_mi'),
    ('dec_small_stop0', 'g14_stop0_rerank4'),
    ('dec_small_stop0', 'g14_stop0_beam3_mi'),
    ('dec_small_stop1', 'g14_stop1_rerank4'),
    ('dec_small_stop1', 'g14_stop1_beam3_mi'),
    ('dec_full', 'g14_full_rerank16'), ('dec_full', 'g14_full_rerank50'),
    ('dec_full', 'g14_full_beam50_mi'),
]


def oracle_forward(sd, feats, nv, m):
    out = O.forward(feats, sd, nv, m['strategy'], length=m['length'],
                    beam_size=m['beam'], temperature=m['temperature'],
                    mi=m['mi'])
    out['tprime'] = out['beam_tokens'].shape[2]
    return out


@pytest.mark.parametrize('model_key,tag', SEARCHES)
def test_oracle_matches_reference_beam_and_rerank(G, M, model_key, tag):
    sd, feats, nv = model(M[model_key])
    m = M[tag]
    got = oracle_forward(sd, feats, nv, m)
    excuses = check_search(got, G, tag, m['strategy'], nv + 1)
    # oracle and reference are both torch-CPU fp32; they differ only by the
    # hoisted key projection's summation order -> at most one near-tie flip
    assert excuses <= 1, f'{excuses} near-tie excuses'
    if excuses == 0:
        caps = [O.reconstruct(t.tolist(), synthetic.vocab_tokens(nv))
                for t in got['tokens']]
        assert caps == m['captions']


@pytest.mark.parametrize('ci', [0, 1])
def test_oracle_predict_groups_match_reference_predict(G, M, ci):
    """`Decoder.predict(batch_size=3)` on 7 neurons: three forwards whose
    early-exit lengths differ (3/10/2 and 4/15/3 steps)."""
    sd, feats, nv = model(M[f'dec_small_stop{ci}'])
    want = M[f'g14_stop{ci}_predict']
    caps, lengths = [], []
    for gi, lo in enumerate(range(0, 7, want['batch_size'])):
        m = M[f'g14_stop{ci}_group{gi}']
        got = oracle_forward(sd, feats[lo:lo + 3], nv, m)
        assert check_search(got, G, f'g14_stop{ci}_group{gi}', 'rerank',
                            nv + 1) == 0
        lengths.append(got['tprime'])
        caps += [O.reconstruct(t.tolist(), synthetic.vocab_tokens(nv))
                 for t in got['tokens']]
    assert caps == want['captions']
    assert len(set(lengths)) == 3  # the groups really stop at different steps


@pytest.mark.parametrize('case', ['A', 'B'])
def test_hand_worked_case_both_restatements(case):
    """SURVEY.md 8(c) "G8": V = 8, beam 3, 4 steps, worked on paper in
    tests/golden/allennlp_standin.py.  The class-shaped stand-in (what the
    goldens were generated on) and the oracle's search core must both give it."""
    A = standin()
    want_t, want_s = A.HAND_EXPECTED[case]
    assert A.run_hand_case(case) == (want_t, want_s)
    table = A.hand_table(case)
    toks, scores = O.beam_search_core(
        lambda tokens, st: (table[tokens].clone(), st),
        lambda st, beam: st, lambda st, rows: st, None, 1, A.HAND_START,
        A.HAND_STOP, A.HAND_STEPS, A.HAND_BEAM)
    assert toks[0].tolist() == want_t and scores[0].tolist() == want_s


def test_beam_one_all_end_early_return():
    """allennlp: beam_size == 1 and every first pick == <end> returns the
    (B,1,1) picks immediately ("empty sequences")."""
    A = standin()
    table = torch.full((8, 8), -9.0)
    table[A.HAND_START, A.HAND_STOP] = -0.5

    def step(tokens, state):
        return table[tokens].clone(), state

    runner = A.BeamSearch(A.HAND_STOP, max_steps=4, beam_size=1)
    with pytest.warns(RuntimeWarning):
        t, s = runner.search(torch.full((2,), A.HAND_START), {}, step)
    ot, os_ = O.beam_search_core(step, lambda st, b: st, lambda st, r: st,
                                 {}, 2, A.HAND_START, A.HAND_STOP, 4, 1)
    assert t.tolist() == ot.tolist() == [[[A.HAND_STOP]], [[A.HAND_STOP]]]
    assert s.tolist() == os_.tolist() == [[-0.5], [-0.5]]
