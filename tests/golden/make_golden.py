"""Generate golden vectors by running the REFERENCE implementation on CPU.

Run in the build container only (needs /root/reference, which does not exist
on the GPU box):

    python tests/golden/make_golden.py

What runs here is the reference's own code (`src.milan.decoders.Decoder`,
`src.milan.lms.LanguageModel`, `src.milan.encoders.PyramidConvEncoder`,
`src.utils.lang.Indexer`, `src.milannotations.datasets` byte conversion) on
seeded synthetic weights/inputs.  Third-party packages the reference imports
but that are not installed offline (spacy, allennlp, torchvision, sacrebleu,
...) are replaced by empty stub modules; the two whose ARITHMETIC the path
needs are handled as follows:
  * torchvision.models.resnet101 -> `_TVResNet` below, an nn.Module written
    from the torchvision 0.12 architecture (module names conv1/bn1/relu/
    maxpool/layer1..4/avgpool/fc so nethook taps resolve).  The reference's
    encoder code (normalisation, nethook taps, mask pooling) then runs on it
    unmodified.  The trunk itself is therefore NOT pinned by the reference.
  * allennlp BeamSearch -> not emulated; no beam-search goldens come from the
    reference (oracle restatement only, "parity unpinned").
Outputs are small tensors (+ the seeds that regenerate weights/inputs through
`milan_amd.synthetic`); no reference source text is stored.
"""
import json
import os
import pathlib
import sys
import types

sys.dont_write_bytecode = True
os.environ['PYTHONDONTWRITEBYTECODE'] = '1'

HERE = pathlib.Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO / 'neuron-descriptions_amd'))

import torch  # noqa: E402
from torch import nn  # noqa: E402

from milan_amd import synthetic  # noqa: E402


class _StubModule(types.ModuleType):

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        t = type(name, (), {'__init__': lambda self, *a, **k: None})
        setattr(self, name, t)
        return t


def _stub(name, **attrs):
    m = _StubModule(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    if '.' in name:
        parent, child = name.rsplit('.', 1)
        setattr(sys.modules[parent], child, m)
    return m


class _EasyDict(dict):

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class _Bottleneck(nn.Module):

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class _TVResNet(nn.Module):
    """torchvision-0.12-shaped bottleneck ResNet (stand-in for the absent
    package; module tree and forward order as in its resnet.py)."""

    def __init__(self, blocks, width=64, pretrained=False, **_):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = width
        for li, n in enumerate(blocks):
            planes = width * 2**li
            layers = []
            for bi in range(n):
                stride = 2 if (bi == 0 and li > 0) else 1
                ds = None
                if bi == 0:
                    ds = nn.Sequential(
                        nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False),
                        nn.BatchNorm2d(planes * 4))
                layers.append(_Bottleneck(inplanes, planes, stride, ds))
                inplanes = planes * 4
            setattr(self, f'layer{li + 1}', nn.Sequential(*layers))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(inplanes, 1000)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def import_reference(allennlp_beam_search=None):
    """Import the reference with stubs for the absent third-party packages.
    `allennlp_beam_search`: module to install as `allennlp.nn.beam_search`
    (make_golden_beam.py passes its class-shaped stand-in)."""
    for n in [
            'spacy', 'spacy.lang', 'spacy.lang.en', 'sacrebleu', 'allennlp',
            'allennlp.nn', 'allennlp.nn.beam_search', 'rouge', 'bert_score',
            'clip', 'wandb', 'h5py', 'transformers', 'boto3', 'torchvision',
            'torchvision.models', 'torchvision.transforms',
            'torchvision.transforms.functional', 'torchvision.utils',
            'torchvision.datasets'
    ]:
        _stub(n)
    _stub('easydict', EasyDict=_EasyDict)
    if allennlp_beam_search is not None:
        sys.modules['allennlp.nn.beam_search'] = allennlp_beam_search
        sys.modules['allennlp.nn'].beam_search = allennlp_beam_search

    class Normalize:

        def __init__(self, mean, std):
            self.mean, self.std = mean, std

    sys.modules['torchvision.transforms'].Normalize = Normalize
    tvm = sys.modules['torchvision.models']
    tvm.resnet101 = lambda **kw: _TVResNet(synthetic.RESNET_BLOCKS['resnet101'
                                                                   ], **kw)
    tvm.resnet50 = lambda **kw: _TVResNet(synthetic.RESNET_BLOCKS['resnet50'],
                                          **kw)
    sys.path.insert(0, '/root/reference')
    from src.milan import decoders, encoders, lms
    from src.milannotations import datasets
    from src.deps.netdissect import renormalize
    from src.utils import lang
    return decoders, encoders, lms, lang, datasets, renormalize


def main():
    torch.set_num_threads(8)
    decoders, encoders, lms, lang, datasets, renormalize = import_reference()
    out = {}
    meta = {}

    # ---- G0: byte -> float conversion (datasets.py:157,191-197) ------------
    ren = renormalize.renormalizer(source='byte', target='pt')
    allbytes = torch.arange(256, dtype=torch.uint8).view(1, 1, 16, 16).repeat(
        1, 3, 1, 1)
    out['g0_bytes_float'] = ren(allbytes.float())[0, 0].reshape(-1).clone()

    # ---- G1: pyramid encoder on a slim trunk (reference encoder code) ------
    def run_encoder(width, size, m, seed, tag, config='resnet101'):
        sd = synthetic.resnet_state_dict(config, seed=seed, width=width)
        enc = encoders.PyramidConvEncoder(config=config,
                                          pretrained=False,
                                          width=width)
        missing = enc.encoder.model.load_state_dict(sd, strict=True)
        enc.eval()
        images_u8, masks_u8 = synthetic.exemplars(1,
                                                  k=m,
                                                  size=size,
                                                  seed=seed + 10,
                                                  zero_every=0)
        masks_u8 = masks_u8.clone()
        masks_u8[0, 1] = 0  # all-zero mask -> exact zero row
        masks_u8[0, 2] = 0
        masks_u8[0, 2, 0, size // 3, size // 2] = 1  # single pixel
        images = ren(images_u8.float().view(-1, 3, size, size))
        masks = masks_u8.float().view(-1, 1, size, size)
        with torch.no_grad():
            feats = enc(images, masks)
        out[f'g1_{tag}_features'] = feats.clone()
        out[f'g1_{tag}_masks_u8'] = masks_u8.clone()
        meta[f'g1_{tag}'] = dict(config=config,
                                 width=width,
                                 size=size,
                                 m=m,
                                 weight_seed=seed,
                                 image_seed=seed + 10)
        del missing

    run_encoder(8, 224, 6, 3, 'slim224')
    run_encoder(8, 100, 4, 4, 'slim100')  # non-2^n ratios: general bilinear
    run_encoder(16, 64, 4, 5, 'r50_64', config='resnet50')
    run_encoder(64, 224, 3, 0, 'full224')  # real ResNet-101 dims

    # ---- decoder goldens ----------------------------------------------------
    class FakeEncoder(encoders.Encoder):

        def __init__(self, feature_size):
            super().__init__()
            self.feature_shape = (feature_size,)

        def forward(self, images, masks=None, **_):
            raise AssertionError('not used')

        def properties(self):
            return {'feature_size': self.feature_shape[0]}

    def make_decoder(nvocab, fsize, hidden, emb, seed, with_lm=True):
        vocab = lang.Vocab(synthetic.vocab_tokens(nvocab))
        tok = lang.Tokenizer(nlp=object())
        indexer = lang.Indexer(vocab,
                               tok,
                               start=True,
                               stop=True,
                               pad=True,
                               unk=True,
                               length=15)
        lm = None
        if with_lm:
            lm = lms.LanguageModel(indexer,
                                   embedding_size=emb,
                                   hidden_size=hidden)
        dec = decoders.Decoder(indexer,
                               FakeEncoder(fsize),
                               lm=lm,
                               embedding_size=emb,
                               hidden_size=hidden)
        v = len(indexer)
        sd = synthetic.decoder_state_dict(v,
                                          feature_size=fsize,
                                          hidden_size=hidden,
                                          embedding_size=emb,
                                          lm=with_lm,
                                          lm_hidden_size=hidden,
                                          lm_embedding_size=emb,
                                          seed=seed)
        res = dec.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys, res
        dec.eval()
        return dec, indexer, v

    def feats_for(b, k, fsize, seed):
        g = torch.Generator().manual_seed(seed)
        # post-ReLU-pooled features are non-negative, O(0.1-1)
        return torch.rand(b, k, fsize, generator=g)

    # small dims
    nv, fs, hid, emb, k = 40, 244, 64, 16, 5
    dec, indexer, v = make_decoder(nv, fs, hid, emb, seed=7)
    feats = feats_for(4, k, fs, 70)
    meta['dec_small'] = dict(nvocab=nv,
                             feature_size=fs,
                             hidden=hid,
                             emb=emb,
                             k=k,
                             weight_seed=7,
                             feat_seed=70,
                             b=4)
    with torch.no_grad():
        st = dec.init_state(feats, lm=True)
        out['g2_small_h'], out['g2_small_c'] = st.h.clone(), st.c.clone()
        toks = torch.tensor([indexer.start_index, 3, 7, indexer.stop_index])
        s1 = dec.step(feats, toks,
                      decoders.DecoderState(st.h, st.c, None, None))
        out['g3_small_tokens'] = toks
        out['g3_small_pred'] = s1.predictions.clone()
        out['g3_small_att'] = s1.attentions.clone()
        out['g3_small_h'] = s1.state.h.clone()
        out['g3_small_c'] = s1.state.c.clone()
        s1m = dec.step(feats, toks, st, temperature=0.3)  # MI branch
        out['g3_small_mi_pred'] = s1m.predictions.clone()
        out['g3_small_mi_hlm'] = s1m.state.h_lm.clone()
        out['g3_small_mi_clm'] = s1m.state.c_lm.clone()
        for mi in (False, True):
            o = dec(feats, strategy='greedy', mi=mi, temperature=0.2)
            tag = 'g4_small_mi' if mi else 'g4_small'
            out[tag + '_tokens'] = o.tokens.clone()
            out[tag + '_scores'] = o.scores.clone()
            out[tag + '_pred'] = o.predictions.clone()
            out[tag + '_att'] = o.attentions.clone()
            meta[tag + '_captions'] = list(o.captions)
        # teacher forcing (strategy=tensor)
        targets = torch.randint(0,
                                v, (4, 6),
                                generator=torch.Generator().manual_seed(5))
        o = dec(feats, strategy=targets, length=6, mi=False)
        out['g4_small_tf_targets'] = targets
        out['g4_small_tf_scores'] = o.scores.clone()
        out['g4_small_tf_pred'] = o.predictions.clone()

        # G5: LM sequence scores with stops first/middle/last/absent/double.
        S, E, P = indexer.start_index, indexer.stop_index, indexer.pad_index
        seqs = torch.tensor([
            [S, 1, 2, 3, 4, 5, 6, 7],
            [S, E, 2, 3, 4, 5, 6, 7],
            [S, 1, 2, E, 4, 5, 6, 7],
            [S, 1, 2, 3, 4, 5, 6, E],
            [S, 1, 2, E, E, E, E, E],
            [S, 1, E, 3, E, 5, 6, 7],
            [S, 9, 9, 9, 9, 9, E, P],
        ])
        out['g5_small_seqs'] = seqs
        out['g5_small_lm_scores'] = dec.lm(seqs, reduce=True).clone()
        out['g5_small_lm_full'] = dec.lm(seqs).clone()

        # rerank epilogue on hand-made beams (decoders.py:495-512)
        beam_tokens = seqs[:6, 1:].reshape(2, 3, 7)
        beam_scores = torch.tensor([[-3.0, -3.5, -4.0], [-2.0, -2.1, -2.2]])
        starts = beam_tokens.new_full((2, 3, 1), dec.lm.indexer.start_index)
        inputs_lm = torch.cat([starts, beam_tokens], dim=-1).view(6, -1)
        s_lm = dec.lm(inputs_lm, reduce=True).view(2, 3)
        sc = beam_scores - 0.2 * s_lm
        choice = sc.argmax(dim=-1)
        out['g5_small_rerank_beam_tokens'] = beam_tokens.clone()
        out['g5_small_rerank_beam_scores'] = beam_scores
        out['g5_small_rerank_scores'] = sc[torch.arange(2), choice].clone()
        out['g5_small_rerank_choice'] = choice

    # full dims, one step + greedy tokens (F=3904, H=512, E=128, V=5004, k=15)
    nv, fs, hid, emb, k = 5000, 3904, 512, 128, 15
    dec, indexer, v = make_decoder(nv, fs, hid, emb, seed=0)
    feats = feats_for(3, k, fs, 71)
    meta['dec_full'] = dict(nvocab=nv,
                            feature_size=fs,
                            hidden=hid,
                            emb=emb,
                            k=k,
                            weight_seed=0,
                            feat_seed=71,
                            b=3)
    with torch.no_grad():
        st = dec.init_state(feats, lm=False)
        out['g2_full_h'], out['g2_full_c'] = st.h.clone(), st.c.clone()
        toks = torch.tensor([indexer.start_index, 17, 4999])
        s1 = dec.step(feats, toks, st)
        out['g3_full_tokens'] = toks
        out['g3_full_pred'] = s1.predictions.clone()
        out['g3_full_att'] = s1.attentions.clone()
        out['g3_full_h'] = s1.state.h.clone()
        o = dec(feats, strategy='greedy', mi=False)
        out['g4_full_tokens'] = o.tokens.clone()
        out['g4_full_scores'] = o.scores.clone()
        out['g4_full_att'] = o.attentions.clone()
        # top-2 gap per step: lets the checker tell a real mismatch from a
        # near-tie flip.
        top2 = o.predictions.topk(2, dim=-1).values
        out['g4_full_top2gap'] = (top2[..., 0] - top2[..., 1]).clone()
        meta['g4_full_captions'] = list(o.captions)
        seqs = torch.cat([
            torch.full((3, 1), indexer.start_index, dtype=torch.long), o.tokens
        ], 1)
        out['g5_full_lm_scores'] = dec.lm(seqs, reduce=True).clone()

    # ---- G6: Indexer.reconstruct / unindex table ---------------------------
    vocab = lang.Vocab(synthetic.vocab_tokens(12))
    indexer = lang.Indexer(vocab, lang.Tokenizer(nlp=object()))
    S, E, P, U = (indexer.start_index, indexer.stop_index, indexer.pad_index,
                  indexer.unk_index)
    cases = [
        [5, 8, 0, 5, 9, E, 5, 5],
        [S, 5, 8, 1, 9, 2, 8, E, P, P],
        [8, 3, 9, 4, 5, 0, 0, 10, 11],
        [E, 5, 8],
        [U, 5, U, 8, 0],
        [5, 2, 2, 8, 0, 9, 6, 8],
        [10, 11, 10, 11, 10, 11, 10],
        [S, S, P, U],
    ]
    meta['g6_vocab'] = list(vocab.tokens)
    meta['g6_cases'] = cases
    meta['g6_expected'] = [indexer.reconstruct(c) for c in cases]
    meta['g6_expected_batch'] = list(indexer.reconstruct(cases))
    meta['g6_unindex'] = [list(indexer.unindex(c)) for c in cases]
    try:
        indexer.reconstruct([5, 99])
        meta['g6_bad_raises'] = None
    except ValueError as e:
        meta['g6_bad_raises'] = str(e)

    # ---- G7: checkpoint skeleton (serialize.py:80-118, decoders.py:1072-) --
    dec_small, _, _ = make_decoder(12, 61, 8, 4, seed=1)
    ser = dec_small.serialize()

    def skeleton(x):
        if isinstance(x, dict):
            return {str(k): skeleton(v) for k, v in x.items()}
        if isinstance(x, torch.Tensor):
            return f'tensor{tuple(x.shape)}:{str(x.dtype)}'
        if isinstance(x, (list, tuple)):
            if len(x) > 8:
                return f'{type(x).__name__}[{len(x)}]'
            return [skeleton(v) for v in x]
        if isinstance(x, (int, float, str, bool)) or x is None:
            return x
        return f'<{type(x).__name__}>'

    meta['g7_skeleton'] = skeleton(ser)

    torch.save(out, HERE / 'reference_goldens.pt')
    with open(HERE / 'reference_goldens.json', 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    total = sum(t.numel() * t.element_size() for t in out.values())
    print(f'wrote {len(out)} tensors, {total / 1e6:.2f} MB')


if __name__ == '__main__':
    main()
