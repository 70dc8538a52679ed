"""Goldens G14: the reference's OWN beam-search / rerank / predict code.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_beam.py

`src.milan.decoders.Decoder.forward(strategy='beam' | 'rerank')`
(decoders.py:465-523: BeamSearch construction, the `step` closure over
`AllenNLPDecoderState` :153-198 with its h_lm / c_lm permutes, the rerank
assembly :495-512) and `Decoder.predict` (:809-871, batches of `batch_size`)
run here UNMODIFIED.  The one thing that is not the reference's is the
`allennlp.nn.beam_search` module they call into: allennlp==2.10 is not
installable offline, so `tests/golden/allennlp_standin.py` (a class-shaped
restatement of that module, checked below against a hand-worked case) is
installed under its name before the reference is imported.

Outputs: tests/golden/reference_goldens_beam.{pt,json} -- tensors + captions +
the seeds that regenerate weights / features through `milan_amd.synthetic`.
For every search the smallest selection margin (score gap between the last
kept and the first dropped candidate, over all steps) is recorded per batch
element, so that checkers can tell a real mismatch from a near-tie.
"""
import json
import pathlib
import sys

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))

import torch  # noqa: E402
from torch.utils import data  # noqa: E402

import allennlp_standin  # noqa: E402
import make_golden  # noqa: E402
from milan_amd import synthetic  # noqa: E402

# (bias added to the <stop> logit, feature seed) of the early-exit cases
STOP_CASES = [(1.25, 74), (1.1, 72)]


class MarginLog:
    """Records, per `sample_beams` call, the gap between the last kept and the
    first dropped candidate (diagnostic only: results are untouched)."""

    def __init__(self):
        self.margins = None
        self._orig = allennlp_standin.Sampler.sample_beams

    def __enter__(self):
        log = self
        orig = self._orig

        def sample_beams(sampler, log_probs, beam_size, state):
            if log_probs.shape[-1] > beam_size:
                top = torch.topk(log_probs, beam_size + 1, dim=-1).values
                gap = (top[:, beam_size - 1] - top[:, beam_size]).clone()
                log.margins = gap if log.margins is None else torch.minimum(
                    log.margins, gap)
            return orig(sampler, log_probs, beam_size, state)

        allennlp_standin.Sampler.sample_beams = sample_beams
        self.margins = None
        return self

    def __exit__(self, *exc):
        allennlp_standin.Sampler.sample_beams = self._orig


def main():
    torch.set_num_threads(8)
    for case in 'AB':  # the stand-in must reproduce the hand-worked answers
        toks, scores = allennlp_standin.run_hand_case(case)
        want_t, want_s = allennlp_standin.HAND_EXPECTED[case]
        assert toks == want_t and scores == want_s, (case, toks, scores)

    decoders, encoders, lms, lang, _, _ = make_golden.import_reference(
        allennlp_beam_search=allennlp_standin)
    assert decoders.beam_search is allennlp_standin
    out, meta = {}, {}

    class FakeEncoder(encoders.Encoder):

        def __init__(self, feature_size):
            super().__init__()
            self.feature_shape = (feature_size,)

        def forward(self, images, masks=None, **_):
            raise AssertionError('not used')

        def properties(self):
            return {'feature_size': self.feature_shape[0]}

    def make_decoder(nvocab, fsize, hidden, emb, seed, stop_bias=0.0):
        vocab = lang.Vocab(synthetic.vocab_tokens(nvocab))
        indexer = lang.Indexer(vocab, lang.Tokenizer(nlp=object()), start=True,
                               stop=True, pad=True, unk=True, length=15)
        lm = lms.LanguageModel(indexer, embedding_size=emb, hidden_size=hidden)
        dec = decoders.Decoder(indexer, FakeEncoder(fsize), lm=lm,
                               embedding_size=emb, hidden_size=hidden)
        sd = synthetic.decoder_state_dict(len(indexer), feature_size=fsize,
                                          hidden_size=hidden,
                                          embedding_size=emb, lm=True,
                                          lm_hidden_size=hidden,
                                          lm_embedding_size=emb, seed=seed)
        if stop_bias:
            sd['output.1.bias'] = sd['output.1.bias'].clone()
            sd['output.1.bias'][indexer.stop_index] += stop_bias
        res = dec.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys, res
        return dec.eval(), indexer

    def feats_for(b, k, fsize, seed):
        g = torch.Generator().manual_seed(seed)
        return torch.rand(b, k, fsize, generator=g)

    def run(dec, feats, tag, strategy, beam, length, mi, temperature=0.2):
        with MarginLog() as log, torch.no_grad():
            o = dec(feats, strategy=strategy, beam_size=beam, length=length,
                    mi=mi, temperature=temperature)
        out[tag + '_tokens'] = o.tokens.clone()
        out[tag + '_scores'] = o.scores.clone()
        out[tag + '_beam_tokens'] = o.beam_tokens.clone()
        out[tag + '_beam_scores'] = o.beam_scores.clone()
        out[tag + '_select_margin'] = log.margins.clone()
        assert o.predictions is None and o.attentions is None
        entry = dict(strategy=strategy, beam=beam, length=length, mi=mi,
                     temperature=temperature, captions=list(o.captions),
                     tprime=int(o.beam_tokens.shape[2]),
                     beam_captions_first=[list(c[:3])
                                          for c in o.beam_captions])
        if strategy == 'rerank':
            # diagnostic: top-2 gap of the reranked scores (near-tie detector)
            b = len(feats)
            starts = o.beam_tokens.new_full((b, beam, 1),
                                            dec.lm.indexer.start_index)
            seqs = torch.cat([starts, o.beam_tokens], -1).view(b * beam, -1)
            with torch.no_grad():
                pmi = o.beam_scores - temperature * dec.lm(
                    seqs, reduce=True).view(b, beam)
            top2 = pmi.topk(2, dim=-1).values
            out[tag + '_rerank_margin'] = (top2[:, 0] - top2[:, 1]).clone()
            out[tag + '_rerank_choice'] = pmi.argmax(dim=-1)
            assert torch.equal(pmi.max(dim=-1).values, o.scores)
        meta[tag] = entry
        print(tag, 'T\'=', entry['tprime'], 'min select margin',
              float(log.margins.min()))

    # ---- small dims (same model / features as G2-G5) ------------------------
    nv, fs, hid, emb, k = 40, 244, 64, 16, 5
    meta['dec_small'] = dict(nvocab=nv, feature_size=fs, hidden=hid, emb=emb,
                             k=k, weight_seed=7, feat_seed=70, b=4)
    dec, _ = make_decoder(nv, fs, hid, emb, seed=7)
    feats = feats_for(4, k, fs, 70)
    run(dec, feats, 'g14_small_beam3', 'beam', 3, 15, False)
    run(dec, feats, 'g14_small_beam4_mi', 'beam', 4, 10, True)
    run(dec, feats, 'g14_small_rerank5', 'rerank', 5, 8, False)
    run(dec, feats, 'g14_small_beam1', 'beam', 1, 15, False)
    run(dec, feats, 'g14_small_beam_default_mi', 'beam', 6, 15, None)

    # ---- early exit: <stop> made likely, so T' < length; predict() batches --
    # (bias, feature seed) chosen so the groups of 3 end at DIFFERENT lengths
    for ci, (stop_bias, fseed) in enumerate(STOP_CASES):
        tag = f'g14_stop{ci}'
        meta[f'dec_small_stop{ci}'] = dict(meta['dec_small'],
                                           stop_bias=stop_bias,
                                           feat_seed=fseed, b=7)
        dec, _ = make_decoder(nv, fs, hid, emb, seed=7, stop_bias=stop_bias)
        feats = feats_for(7, k, fs, fseed)
        run(dec, feats, f'{tag}_rerank4', 'rerank', 4, 15, False)
        run(dec, feats, f'{tag}_beam3_mi', 'beam', 3, 15, True)
        # per-batch groups of predict(batch_size=3): 3 + 3 + 1 neurons
        for gi, lo in enumerate(range(0, 7, 3)):
            run(dec, feats[lo:lo + 3], f'{tag}_group{gi}', 'rerank', 4, 15,
                False)
        with torch.no_grad():
            captions = dec.predict(data.TensorDataset(feats), batch_size=3,
                                   features=data.TensorDataset(feats),
                                   strategy='rerank', beam_size=4,
                                   temperature=0.2, display_progress_as=None)
        meta[f'{tag}_predict'] = dict(batch_size=3, beam=4,
                                      captions=list(captions))
        group_caps = sum((meta[f'{tag}_group{g}']['captions']
                          for g in range(3)), [])
        assert list(captions) == group_caps

    # ---- full dims (F=3904, H=512, E=128, V=5004, k=15; as G2-G5 "full") ----
    nv, fs, hid, emb, k = 5000, 3904, 512, 128, 15
    meta['dec_full'] = dict(nvocab=nv, feature_size=fs, hidden=hid, emb=emb,
                            k=k, weight_seed=0, feat_seed=71, b=3)
    dec, _ = make_decoder(nv, fs, hid, emb, seed=0)
    feats = feats_for(3, k, fs, 71)
    run(dec, feats, 'g14_full_rerank16', 'rerank', 16, 15, False)
    run(dec, feats, 'g14_full_rerank50', 'rerank', 50, 15, False)
    run(dec, feats, 'g14_full_beam50_mi', 'beam', 50, 15, True)

    torch.save(out, HERE / 'reference_goldens_beam.pt')
    with open(HERE / 'reference_goldens_beam.json', 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    total = sum(t.numel() * t.element_size() for t in out.values())
    print(f'wrote {len(out)} tensors, {total / 1e6:.2f} MB')


if __name__ == '__main__':
    main()
