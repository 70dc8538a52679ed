"""How often do the two arithmetic modes produce the same description?

    python tools/precision_agreement.py [neurons=1024]

Runs the full hot path (ResNet-101 pyramid encoder, beam 50 + rerank, V=5004)
on the same synthetic neurons in `f32` and `split_f16` mode and reports token
agreement, score differences and feature differences.
"""
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'neuron-descriptions_amd'))
from milan_amd import hip, synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = torch.device('cuda', 0)
    nv = 5000
    sd = synthetic.milan_state_dict(nv + 4, 'resnet101', seed=0)
    ctx = hip.Context(hip.make_dims(sd, nv), sd, dev)
    res = {}
    for mode in ('f32', 'split_f16'):
        ctx.set_precision(mode)
        toks, scores, feats, beams = [], [], [], []
        for lo in range(0, n, 256):
            images, masks = synthetic.exemplars(min(256, n - lo), k=15,
                                                size=224, seed=1 + lo,
                                                device='cuda:0')
            out = ctx.describe(images, masks, hip.RERANK, 15, 50, False, 0.2,
                               group_size=16, want_features=True)
            toks.append(out['tokens']); scores.append(out['scores'])
            feats.append(out['features']); beams.append(out['beam_scores'])
        res[mode] = [torch.cat(x) for x in (toks, scores, feats, beams)]
    (t32, s32, f32, b32), (tsp, ssp, fsp, bsp) = res['f32'], res['split_f16']
    same = (t32 == tsp).all(dim=1)
    print(f'neurons: {n}')
    print(f'identical top-1 descriptions: {int(same.sum())} / {n} '
          f'({100 * same.float().mean():.2f} %)')
    print(f'feature max |diff|: {float((f32 - fsp).abs().max()):.3g} '
          f'(feature max {float(f32.abs().max()):.3g}); relative '
          f'{float((f32 - fsp).abs().max() / f32.abs().max()):.2g}')
    print(f'rerank score |diff| on identical captions: max '
          f'{float((s32 - ssp)[same].abs().max()):.3g}, mean '
          f'{float((s32 - ssp)[same].abs().mean()):.3g} '
          f'(scores ~ {float(s32.mean()):.1f})')
    print(f'best-beam score |diff|: max {float((b32[:, 0] - bsp[:, 0]).abs().max()):.3g}')
    if (~same).any():
        gap = (s32 - ssp)[~same].abs()
        print(f'differing captions: rerank score |diff| max {float(gap.max()):.3g} '
              f'(near-ties between beams that swap on last-bit differences)')


if __name__ == '__main__':
    main()
