"""LanguageModel.forward(reduce=True) from log-softmax STATISTICS (EPI_LSE, round 5).

In the split-f16 mode the LM's vocabulary GEMM no longer writes its logits: its
epilogue leaves {max, sum exp(x - max)} per row and 64-column block plus the target
column's logit, and a wave per row combines the blocks (csrc/gemm.hip epilogue_lse,
csrc/decoder.hip lm_accumulate_lse_kernel).  The f32 mode still writes the logits and
reads them back (lm_accumulate_kernel), so the two paths check each other inside one
process; both are checked against the oracle's restatement of src/milan/lms.py:58-101
(including its stop-token off-by-one).  Vocabulary sizes: not a multiple of 64 (the last
block is ragged), one block short of a tile, the benchmark's 5004; row counts that leave
ragged row tiles; targets in the first / last column.
"""
import pytest
import torch

from milan_amd import hip, synthetic
from oracle import milan_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    hip.load_library()
    return hip.require_device('cuda')


def _ctx(dev, vocab, hidden=64, emb=32, seed=3):
    sd = synthetic.decoder_state_dict(vocab, feature_size=64, hidden_size=hidden,
                                      embedding_size=emb, lm=True, lm_hidden_size=hidden,
                                      lm_embedding_size=emb, seed=seed)
    return hip.Context(hip.make_dims(sd, vocab - 4), sd, dev), sd


@pytest.mark.parametrize('vocab,rows,length', [
    (260, 37, 9),      # 5 column blocks, the last one 4 columns wide; one ragged row tile
    (1000, 300, 6),    # 16 blocks (the last 40 wide), two row tiles
    (5004, 513, 5),    # the benchmark's vocabulary: 79 blocks over 20 column tiles
    (4096, 64, 4),     # whole tiles only
])
def test_lm_score_from_statistics_matches_logits_path_and_oracle(dev, vocab, rows, length):
    ctx, sd = _ctx(dev, vocab)
    nv = vocab - 4
    stop = ctx.dims.stop_index
    g = torch.Generator().manual_seed(vocab + rows)
    seqs = torch.randint(0, nv, (rows, length), generator=g)
    seqs[:, 0] = ctx.dims.start_index
    seqs[::3, -1] = 0                     # target in the first column
    seqs[1::3, -1] = vocab - 1            # ... in the last (ragged) block
    seqs[2::5, length // 2] = stop        # a stop in the middle (lms.py's lagging mask)
    with torch.no_grad():
        want = O.lm_score(seqs, sd, stop)
    ctx.set_precision('split_f16')
    via_stats = ctx.lm_score(seqs).cpu()
    ctx.set_precision('f32')
    via_logits = ctx.lm_score(seqs).cpu()
    scale = float(want.abs().max())
    err_stats = float((via_stats - want).abs().max())
    err_logits = float((via_logits - want).abs().max())
    print(f'V={vocab} rows={rows}: |stats - oracle| {err_stats:.2e}, |logits path - oracle| '
          f'{err_logits:.2e}, scale {scale:.1f}')
    assert err_stats <= 2e-5 * scale + 2e-5, (err_stats, scale)
    assert err_logits <= 2e-5 * scale + 2e-5, (err_logits, scale)
    assert float((via_stats - via_logits).abs().max()) <= 2e-5 * scale + 2e-5
    ctx.close()


def test_small_vocabulary_keeps_the_logits_path(dev):
    """V < 256 (fewer than one column tile) is not worth the statistics: same result either way."""
    ctx, sd = _ctx(dev, 68)
    seqs = torch.randint(0, 64, (11, 7), generator=torch.Generator().manual_seed(1))
    seqs[:, 0] = ctx.dims.start_index
    with torch.no_grad():
        want = O.lm_score(seqs, sd, ctx.dims.stop_index)
    ctx.set_precision('split_f16')
    got = ctx.lm_score(seqs).cpu()
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 2e-5
    ctx.close()
