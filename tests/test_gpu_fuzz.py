"""Seeded differential fuzzing of the HIP path against the CPU oracle: random
(small) model geometries, image sizes, exemplar counts, beam sizes, lengths,
vocabulary sizes, trunk kinds, MI on/off and temperatures -- the shapes nobody
thought to write a dedicated test for (odd image sizes, V not a multiple of 4,
beam == V, k == 1, a single neuron, widths that defeat the split-f16 path)."""
import os
import random

import pytest
import torch

from milan_amd import hip, synthetic
from oracle import milan_oracle as O
from tests.test_gpu_parity import _check_beams, assert_tokens_match, close

pytestmark = pytest.mark.gpu

CONFIGS = ['resnet50', 'resnet18', 'alexnet']


def draw(seed):
    r = random.Random(seed)
    config = r.choice(CONFIGS)
    width = r.choice([8, 16, 32])
    if config == 'alexnet':
        size_h, size_w = r.randint(64, 110), r.randint(64, 110)
    else:
        # down to a single pixel: the spatial size bottoms out at 1x1 early
        size_h, size_w = r.choice([r.randint(1, 12), r.randint(13, 97)]), \
            r.choice([r.randint(1, 12), r.randint(13, 97)])
    nv = r.choice([9, 21, 37, 64, 130, 700])
    hidden = r.choice([16, 32, 64])
    emb = r.choice([8, 16, 32])
    return dict(
        config=config, width=width, h=size_h, w=size_w, nv=nv, hidden=hidden,
        emb=emb, k=r.randint(1, 6), n=r.randint(1, 5),
        length=r.randint(1, 9), beam=r.randint(1, min(nv + 4, 12)),
        mi=r.random() < 0.4, temperature=r.choice([0.1, 0.2, 0.5]),
        precision=r.choice(['f32', 'split_f16']),
        zero_every=r.choice([0, 3]),
        mask_kind=r.choice(['u8', 'float01', 'soft']))


# MILAN_FUZZ_SEEDS=<n> widens the campaign (a 400-seed run is in DESIGN.md)
N_SEEDS = int(os.environ.get('MILAN_FUZZ_SEEDS', '36'))


@pytest.mark.parametrize('seed', range(N_SEEDS))
def test_fuzz_encode_and_decode(seed):
    dev = hip.require_device('cuda')
    p = draw(1000 + seed)
    blocks = synthetic.RESNET_BLOCKS.get(p['config'], (0, 0, 0, 0))
    sd = synthetic.milan_state_dict(p['nv'] + 4, config=p['config'],
                                    seed=seed, width=p['width'],
                                    hidden_size=p['hidden'],
                                    embedding_size=p['emb'],
                                    lm_hidden_size=p['hidden'],
                                    lm_embedding_size=p['emb'])
    ctx = hip.Context(hip.make_dims(sd, p['nv'], blocks=blocks), sd, dev)
    ctx.set_precision(p['precision'])
    g = torch.Generator().manual_seed(seed)
    images = torch.randint(0, 256, (p['n'], p['k'], 3, p['h'], p['w']),
                           dtype=torch.uint8, generator=g)
    masks = (torch.rand(p['n'], p['k'], 1, p['h'], p['w'], generator=g) >
             0.7).to(torch.uint8)
    if p['mask_kind'] == 'float01':
        masks = masks.float()
    elif p['mask_kind'] == 'soft':  # the reference takes any float mask
        masks = masks.float() * torch.rand(masks.shape, generator=g)
    if p['zero_every']:
        masks.view(-1, 1, p['h'], p['w'])[::p['zero_every']] = 0
    feats = O.encode(O.byte_to_float(images), masks.float(), sd, blocks=blocks)
    nv, length, beam = p['nv'], p['length'], p['beam']

    # encoder (the fused call also returns the features it decoded from)
    out = ctx.describe(images, masks, hip.GREEDY, length, 1, p['mi'],
                       p['temperature'], want_full=True, want_features=True)
    close(out['features'], feats, 2e-3, 2e-4)

    # decode from the ORACLE's features so both sides see identical inputs
    want = O.forward(feats, sd, nv, 'greedy', length=length, mi=p['mi'],
                     temperature=p['temperature'])
    got = ctx.decode(feats, hip.GREEDY, length, 1, p['mi'], p['temperature'])
    top2 = want['predictions'].topk(2, dim=-1).values
    # random tiny vocabularies tie more often than a trained model: two rows
    # per case may sit on a near-tie, everything else is compared
    same = assert_tokens_match(got['tokens'], want['tokens'],
                               top2[..., 0] - top2[..., 1], what=str(p),
                               max_ties=2)
    if same.any():
        close(got['predictions'][same], want['predictions'][same], 1e-4, 5e-4)
        close(got['attentions'][same], want['attentions'][same], 1e-4, 1e-5)
        close(got['scores'][same], want['scores'][same], 1e-4, 3e-3)

    want_t, want_s = O.beam_search(feats, sd, nv, nv + 1, length, beam,
                                   mi=p['mi'], temperature=p['temperature'])
    strategy = hip.BEAM if p['mi'] else hip.RERANK
    got_b = ctx.decode(feats, strategy, length, beam, p['mi'],
                       p['temperature'])
    tp = want_t.shape[2]
    assert int(got_b['out_len'][0]) == tp, p
    same = _check_beams(got_b, want_t, want_s, tp, max_ties=2)
    if not p['mi'] and same.any():
        t, s, _ = O.rerank(want_t, want_s, sd, nv, nv + 1, p['temperature'])
        picked = (got_b['tokens'].cpu()[:, :tp] == t).all(dim=1) & same
        close(got_b['scores'][picked], s[picked], 1e-4, 3e-3)
    ctx.close()


def draw_decoder(seed):
    r = random.Random(seed)
    nv = r.choice([5, 9, 33, 100, 257, 1023])
    return dict(
        nv=nv, feat=r.choice([64, 72, 244, 488]),
        hidden=r.choice([4, 20, 36, 64, 100]), emb=r.choice([4, 12, 28, 32]),
        k=r.randint(1, 17), n=r.randint(1, 9), length=r.randint(1, 21),
        beam=r.randint(1, min(nv + 4, 24)), mi=r.random() < 0.5,
        temperature=r.choice([0.05, 0.2, 1.0]),
        group=r.choice([0, 1, 2, 3]),
        precision=r.choice(['f32', 'split_f16']))


@pytest.mark.parametrize('seed', range(24))
def test_fuzz_decoder_only(seed):
    """Decoder / LM geometries far from the pretrained one (hidden sizes that
    are not multiples of 32, k > 15, long sequences, beam == |V|, ragged
    allennlp groups), features handed in directly."""
    dev = hip.require_device('cuda')
    p = draw_decoder(5000 + seed)
    nv = p['nv']
    sd = synthetic.decoder_state_dict(nv + 4, feature_size=p['feat'],
                                      hidden_size=p['hidden'],
                                      embedding_size=p['emb'], lm=True,
                                      lm_hidden_size=p['hidden'],
                                      lm_embedding_size=p['emb'], seed=seed)
    ctx = hip.Context(hip.make_dims(sd, nv), sd, dev)
    ctx.set_precision(p['precision'])
    g = torch.Generator().manual_seed(seed)
    feats = torch.rand(p['n'], p['k'], p['feat'], generator=g)
    length, beam = p['length'], p['beam']

    # teacher forcing on random targets: every log-prob the decoder can emit
    targets = torch.randint(0, nv + 4, (p['n'], length), generator=g)
    want_f = O.teacher_forced(feats, sd, nv, targets, mi=p['mi'],
                              temperature=p['temperature'])
    got_f = ctx.decode(feats, hip.FORCED, length, 1, p['mi'],
                       p['temperature'], forced=targets)
    close(got_f['predictions'], want_f.predictions, 1e-4, 5e-4)
    close(got_f['scores'], want_f.scores, 1e-4, 5e-3)

    # beam search with allennlp's early exit evaluated per group
    gs = p['group'] or p['n']
    strategy = hip.BEAM if p['mi'] else hip.RERANK
    got = ctx.decode(feats, strategy, length, beam, p['mi'], p['temperature'],
                     group_size=p['group'])
    for gi, lo in enumerate(range(0, p['n'], gs)):
        sl = slice(lo, min(p['n'], lo + gs))
        want_t, want_s = O.beam_search(feats[sl], sd, nv, nv + 1, length, beam,
                                       mi=p['mi'],
                                       temperature=p['temperature'])
        tp = want_t.shape[2]
        assert int(got['out_len'][gi]) == tp, (p, gi)
        part = {'beam_tokens': got['beam_tokens'][sl],
                'beam_scores': got['beam_scores'][sl]}
        _check_beams(part, want_t, want_s, tp, max_ties=2)
    ctx.close()
