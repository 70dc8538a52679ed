"""Generate goldens for the teacher-forced / scoring half of the path.

Run in the build container (the reference is not available on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_forced.py

Imports the reference from /root/reference (with the same stub modules as
make_golden.py) and records, for seeded synthetic weights:

  G8  `Decoder.forward(features, strategy=<tensor>, length=L)`
      (src/milan/decoders.py:444-445, teacher forcing) with and without MI:
      tokens, scores, predictions, attentions.
  G9  `Indexer.index(tokenized, ...)` (src/utils/lang.py:456-514) for a grid
      of option combinations: expected id tuples.
  G10 `Decoder.score(captions, features)` (decoders.py:636-711) with a
      whitespace tokenizer standing in for spaCy on both sides: totals.
  G12 `LanguageModel.forward` (lms.py:58-101) with reduce=False, reduce=True
      and reduce=True + caller masks.

Outputs: reference_goldens_forced.pt / .json (data only).
"""
import itertools
import json
import pathlib
import sys

import torch

HERE = pathlib.Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'neuron-descriptions_amd'))
sys.path.insert(0, str(HERE))

import make_golden  # noqa: E402  (stubs + import_reference)
from milan_amd import synthetic  # noqa: E402


class WhitespaceTokenizer:
    """Lower-case whitespace split; stands in for the spaCy tokenizer."""

    def __call__(self, texts):
        if isinstance(texts, str):
            return tuple(texts.lower().split())
        return tuple(tuple(t.lower().split()) for t in texts)


def main():
    torch.set_num_threads(8)
    decoders, encoders, lms, lang, _, _ = make_golden.import_reference()
    out, meta = {}, {}

    class FakeEncoder(encoders.Encoder):

        def __init__(self, feature_size):
            super().__init__()
            self.feature_shape = (feature_size,)

        def forward(self, images, masks=None, **_):
            raise AssertionError('not used')

        def properties(self):
            return {'feature_size': self.feature_shape[0]}

    nv, fs, hid, emb, k, b, length = 40, 244, 64, 16, 5, 4, 9
    vocab = lang.Vocab(synthetic.vocab_tokens(nv))
    indexer = lang.Indexer(vocab, WhitespaceTokenizer(), start=True,
                           stop=True, pad=True, unk=True, length=15)
    lm = lms.LanguageModel(indexer, embedding_size=emb, hidden_size=hid)
    dec = decoders.Decoder(indexer, FakeEncoder(fs), lm=lm,
                           embedding_size=emb, hidden_size=hid)
    sd = synthetic.decoder_state_dict(len(indexer), feature_size=fs,
                                      hidden_size=hid, embedding_size=emb,
                                      lm=True, lm_hidden_size=hid,
                                      lm_embedding_size=emb, seed=7)
    res = dec.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    dec.eval()
    g = torch.Generator().manual_seed(70)
    feats = torch.rand(b, k, fs, generator=g)
    meta['forced'] = dict(nvocab=nv, feature_size=fs, hidden=hid, emb=emb,
                          k=k, b=b, length=length, weight_seed=7,
                          feat_seed=70, target_seed=71)

    # ---- G8: teacher forcing ------------------------------------------------
    tg = torch.Generator().manual_seed(71)
    targets = torch.randint(0, len(indexer), (b, length), generator=tg)
    targets[1, 4] = indexer.stop_index  # scores keep accumulating after <stop>
    targets[2, 0] = indexer.unk_index
    out['g8_targets'] = targets
    with torch.no_grad():
        for mi in (False, True):
            o = dec(feats, strategy=targets, length=length, mi=mi,
                    temperature=0.2)
            tag = 'g8_mi' if mi else 'g8'
            out[tag + '_tokens'] = o.tokens.clone()
            out[tag + '_scores'] = o.scores.clone()
            out[tag + '_pred'] = o.predictions.clone()
            out[tag + '_att'] = o.attentions.clone()
            assert o.beam_tokens is None

    # ---- G9: Indexer.index ----------------------------------------------------
    toks = synthetic.vocab_tokens(nv)
    seqs = [
        (toks[3], toks[5], 'notaword', toks[1]),
        (toks[0],),
        tuple(toks[i % nv] for i in range(20)),  # longer than length
        ('zzz', 'yyy'),
    ]
    cases = []
    for start, stop, pad, unk, ln in itertools.product(
            (False, True), (False, True), (False, True), (False, True),
            (None, 3, 6)):
        ix = lang.Indexer(vocab, WhitespaceTokenizer(), start=start,
                          stop=stop, pad=pad, unk=unk, length=ln)
        batch = ix.index(seqs)
        single = ix.index(seqs[0])
        override = ix.index(seqs, start=True, stop=True, pad=False, unk=True,
                            length=4)
        cases.append(dict(start=start, stop=stop, pad=pad, unk=unk, length=ln,
                          batch=[list(x) for x in batch],
                          single=list(single),
                          override=[list(x) for x in override]))
    meta['g9_sequences'] = [list(s) for s in seqs]
    meta['g9_cases'] = cases
    # __call__ = tokenize + index (lang.py:331-393)
    texts = [' '.join(s) for s in seqs]
    meta['g9_call'] = [list(x) for x in indexer(texts)]
    meta['g9_call_single'] = list(indexer(texts[0].upper()))

    # ---- G10: Decoder.score ---------------------------------------------------
    captions = [
        f'{toks[2]} {toks[9]} {toks[4]}',
        f'{toks[7]} unknownword {toks[7]} {toks[30]}',
        f'{toks[11]}',
        ' '.join(toks[i] for i in range(12, 30)),  # truncated to length 15
    ]
    meta['g10_captions'] = captions
    with torch.no_grad():
        out['g10_scores'] = dec.score(captions, feats, mi=False).clone()
        out['g10_scores_mi'] = dec.score(captions, feats, mi=True,
                                         temperature=0.3).clone()
        out['g10_scores_broadcast'] = dec.score(captions, feats[:1],
                                                mi=False).clone()

    # ---- G12: LanguageModel.forward beyond the rerank call ---------------------
    sg = torch.Generator().manual_seed(72)
    seqs = torch.randint(0, len(indexer), (5, 7), generator=sg)
    seqs[:, 0] = indexer.start_index
    seqs[1, 3] = indexer.stop_index
    seqs[3, 1] = indexer.stop_index
    masks = torch.randint(0, 2, (5, 6), generator=sg)
    out['g12_seqs'] = seqs
    out['g12_masks'] = masks
    with torch.no_grad():
        out['g12_lps'] = lm(seqs).clone()                       # reduce=False
        out['g12_reduced'] = lm(seqs, reduce=True).clone()      # default mask
        out['g12_masked'] = lm(seqs, reduce=True, masks=masks).clone()

    torch.save(out, HERE / 'reference_goldens_forced.pt')
    with open(HERE / 'reference_goldens_forced.json', 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    size = (HERE / 'reference_goldens_forced.pt').stat().st_size
    print(f'wrote {len(out)} tensors, {size / 1e3:.0f} kB')


if __name__ == '__main__':
    main()
