"""The rerank pass without its redundant rows (round 6, csrc/decoder.hip lm_score_dedup).

The reference scores all B x beam sequences with the LM row by row (src/milan/decoders.py:495-512,
src/milan/lms.py:58-101).  The rows of a neuron are leaves of one beam-search tree: the LSTM state
at step t depends on the first t tokens only, so beams that share a prefix share the work.  The
HIP path multiplies one row per prefix class (class counts on the device) -- and must return
bit for bit what the row-by-row pass returns, because a row's arithmetic does not depend on where
in a launch it sits.
"""
import hashlib
import os
import subprocess
import sys
import pathlib

import pytest
import torch

from milan_amd import hip, synthetic

pytestmark = pytest.mark.gpu
REPO = pathlib.Path(__file__).resolve().parent.parent

SNIPPET = r'''
import sys, hashlib, torch
sys.path.insert(0, %(repo)r); sys.path.insert(0, %(pkg)r)
from milan_amd import hip, synthetic
nv = %(nv)d
sd = synthetic.decoder_state_dict(nv + 4, seed=%(seed)d)
ctx = hip.Context(hip.make_dims(sd, nv), sd, 'cuda')
ctx.set_precision('split_f16')
g = torch.Generator().manual_seed(%(seed)d)
h = hashlib.sha256()
for n, beam, group in ((5, 50, 16), (37, 16, 16), (3, 2, 1), (64, 50, 16), (2, 120, 16)):
    feats = torch.rand(n, 15, 3904, generator=g).cuda()
    out = ctx.decode(feats, hip.RERANK, 15, beam, False, 0.2, group_size=group, want_full=False)
    for key in ('tokens', 'scores', 'beam_tokens', 'beam_scores', 'out_len'):
        h.update(out[key].cpu().numpy().tobytes())
print('SHA', h.hexdigest())
'''


def _run(dedup, nv=5000, seed=4):
    env = dict(os.environ, MILAN_LM_DEDUP=str(dedup), HSA_ENABLE_IPC_MODE_LEGACY='0')
    hip.release_workspaces()
    code = SNIPPET % dict(repo=str(REPO), pkg=str(REPO / 'neuron-descriptions_amd'), nv=nv, seed=seed)
    done = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True,
                          check=True)
    return [ln for ln in done.stdout.splitlines() if ln.startswith('SHA')][0]


def test_prefix_classes_give_the_bits_of_the_row_by_row_pass():
    """MILAN_LM_DEDUP=0 restores the every-row pass; tokens, scores, beams: same sha256
    (beam 50 / 16 / 2 / 120 -- the merge kernel's limit is ~125 --, partial tiles, groups of 1)."""
    assert _run(1) == _run(0)


def test_rerank_choice_equals_the_public_row_by_row_lm_score():
    """Independent of the switch: milan_lm_score (no groups -> every row through the LM) on
    the final beams reproduces the rerank choice of milan_decode."""
    nv, beam, lam = 5000, 50, 0.2
    sd = synthetic.decoder_state_dict(nv + 4, seed=9)
    ctx = hip.Context(hip.make_dims(sd, nv), sd, 'cuda')
    ctx.set_precision('split_f16')
    g = torch.Generator().manual_seed(9)
    feats = torch.rand(24, 15, 3904, generator=g).cuda()
    out = ctx.decode(feats, hip.RERANK, 15, beam, False, lam, group_size=16, want_full=False)
    bt, bs = out['beam_tokens'], out['beam_scores']
    n, _, t = bt.shape
    start = torch.full((n * beam, 1), ctx.dims.start_index, dtype=torch.long, device='cuda')
    seqs = torch.cat([start, bt.reshape(n * beam, t)], dim=1)
    # (the decode pass masks by the group's early-exit length; full length here: all groups ran
    # the whole 15 steps on this model)
    assert int(out['out_len'].min()) == t
    lm = ctx.lm_score(seqs).reshape(n, beam)
    pmi = bs - lam * lm
    best = pmi.argmax(dim=1)
    assert torch.equal(out['tokens'], bt[torch.arange(n), best])
    torch.testing.assert_close(out['scores'], pmi[torch.arange(n), best], rtol=0, atol=2e-5)
    # and the sharing is real on this workload: fewer than 60 % of the rows are distinct work
    sys.path.insert(0, str(REPO))
    import bench
    lstm, vocab = bench.lm_row_fractions(bt)
    print(f'distinct rows: LSTM {lstm:.2f}, vocabulary {vocab:.2f} of {beam} x {t}')
    assert 0.05 < lstm < vocab < 1.0
    ctx.close()
