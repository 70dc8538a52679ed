"""How often do the arithmetic modes produce the same description?

    python tools/precision_agreement.py [neurons=1024] [--json]

Runs the full hot path (ResNet-101 pyramid encoder, beam 50 + rerank, V=5004)
on the same synthetic neurons in `f32` (exact fp32 MFMA: the reference's
arithmetic), `split_f16` (the bench default, fp32-class) and `f16` (the FAST mode:
layer3 / layer4 on plain f16 operands, narrower than the reference -- a reported
extra) and reports, against `f32`: caption agreement (the flip rate), rerank-score
and feature differences.  `agreement()` is what bench.py's `fast_mode` block calls.
"""
import json
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'neuron-descriptions_amd'))
from milan_amd import hip, synthetic  # noqa: E402


def agreement(ctx, n=1024, modes=('split_f16', 'f16'), chunk=256, device='cuda:0'):
    """{mode: {...}} against the exact-fp32 mode on `n` synthetic neurons."""
    saved = ctx.precision
    res = {}
    for mode in ('f32',) + tuple(modes):
        ctx.set_precision(mode)
        toks, scores, feats, beams = [], [], [], []
        for lo in range(0, n, chunk):
            images, masks = synthetic.exemplars(min(chunk, n - lo), k=15,
                                                size=224, seed=1 + lo,
                                                device=device)
            out = ctx.describe(images, masks, hip.RERANK, 15, 50, False, 0.2,
                               group_size=16, want_features=True)
            toks.append(out['tokens']); scores.append(out['scores'])
            feats.append(out['features']); beams.append(out['beam_scores'])
        res[mode] = [torch.cat(x) for x in (toks, scores, feats, beams)]
    ctx.set_precision(saved)
    t32, s32, f32, b32 = res['f32']
    report = {}
    for mode in modes:
        t, s, f, b = res[mode]
        same = (t32 == t).all(dim=1)
        flips = int((~same).sum())
        report[mode] = {
            'neurons': n,
            'identical_descriptions': int(same.sum()),
            'caption_flips': flips,
            'caption_flip_rate': flips / n,
            'max_abs_rerank_score_diff': float((s32 - s).abs().max()),
            'max_abs_rerank_score_diff_on_identical':
                float((s32 - s)[same].abs().max()) if same.any() else None,
            'mean_abs_rerank_score_diff': float((s32 - s).abs().mean()),
            'feature_max_abs_diff': float((f32 - f).abs().max()),
            'feature_scale': float(f32.abs().max()),
            'feature_max_rel_diff': float((f32 - f).abs().max() / f32.abs().max()),
            'best_beam_score_max_abs_diff': float((b32[:, 0] - b[:, 0]).abs().max()),
        }
    return report


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    n = int(args[0]) if args else 1024
    nv = 5000
    sd = synthetic.milan_state_dict(nv + 4, 'resnet101', seed=0)
    ctx = hip.Context(hip.make_dims(sd, nv), sd, torch.device('cuda', 0))
    report = agreement(ctx, n)
    if '--json' in sys.argv:
        print(json.dumps(report))
        return
    for mode, r in report.items():
        print(f'{mode} vs f32 on {n} neurons:')
        print(f'  identical top-1 descriptions: {r["identical_descriptions"]} / {n} '
              f'({100 * (1 - r["caption_flip_rate"]):.2f} %), flips {r["caption_flips"]}')
        print(f'  feature max |diff|: {r["feature_max_abs_diff"]:.3g} (feature max '
              f'{r["feature_scale"]:.3g}); relative {r["feature_max_rel_diff"]:.2g}')
        print(f'  rerank score |diff|: max {r["max_abs_rerank_score_diff"]:.3g}, mean '
              f'{r["mean_abs_rerank_score_diff"]:.3g}; best-beam score max '
              f'{r["best_beam_score_max_abs_diff"]:.3g}')


if __name__ == '__main__':
    main()
