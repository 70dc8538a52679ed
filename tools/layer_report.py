"""Per-layer GEMM throughput from a rocprofv3 rocpd database of bench.py.

usage: python tools/layer_report.py <bench_results.db> [n_images=3840] [passes=3] [fused_ds=0|1] [chain=0|1|2|3]
(chain=2: + the layer3 chains of csrc/chain3.hip; 3: + layer1's 3x3 / c1 in front of its chain launches, round 6)
chain=1 (round 3): a bottleneck's expand conv c3 and the next bottleneck's reduce conv
c1 are ONE launch (csrc/chain.hip) wherever the next block has no downsample branch and
the stage is layer1..3; such pairs are reported as `lX.x.c3>c1`.  The fused stem
(csrc/stem.hip, one launch where the implicit-GEMM conv1 was) and layer1's conv3.hip
launches take the places of the launches they replace.
Maps the igemm dispatches of one hot-path pass onto the ResNet-101 layer list
(launch order is deterministic) and prints time / algorithmic TFLOP/s per
layer group, then the decoder+LM GEMM total.
"""
import re
import sqlite3
import sys


GEMM_LIKE = ("name like '%igemm%' or name like '%conv3x3%' or name like '%chain_kernel%' "
             "or name like '%chain3_kernel%' "
             "or name like '%stem_fused%' or name like '%conv3_p64%'")


def build_layers(n, fused, chain, wide, bneck=False):
    """(name, M, N, K, flops) of the encoder's GEMM-class launches of one pass of n
    images, in launch order (ResNet-101, split-f16 mode)."""
    layers = []

    def conv(name, h, cin, cout, k, s, extra_k=0):
        ho = (h + 2 * (k // 2) - k) // s + 1
        m = n * ho * ho
        kk = k * k * cin + extra_k
        layers.append((name, m, cout, kk, 2 * m * cout * kk))
        return ho

    conv('stem', 224, 3, 64, 7, 2)
    h, inp = 56, 64
    chained_c1 = False  # this block's c1 ran inside the previous block's launch
    for li, nb in enumerate((3, 4, 23, 3)):
        pl = 64 * 2**li
        for bi in range(nb):
            s = 2 if (bi == 0 and li > 0) else 1
            # chain.hip: planes <= 256, a next block in the stage; block 0 (two-source
            # expand) only where its downsample has stride 1 (layer1)
            # ... or, for planes 64, the first block of the next stage (its c1 has 2 x
            # the planes and runs at this stage's resolution)
            boundary = chain and bi + 1 == nb and pl == 64
            will_chain = boundary or (
                chain and pl <= (256 if wide else 128) and bi + 1 < nb and
                (bi > 0 or (fused and li == 0)))
            # round 6 (MILAN_FUSE_BNECK): layer1's 3x3 conv -- and layer1.0's own c1 -- run
            # inside the block's chain launch
            front = bneck and will_chain and pl == 64
            front_c1 = front and bi == 0 and not chained_c1
            extra = 0
            if not chained_c1:
                if front_c1:
                    extra += 2 * n * h * h * pl * inp
                else:
                    conv(f'l{li+1}.{bi}.c1', h, inp, pl, 1, 1)
            if front:
                h2 = (h + 2 - 3) // s + 1
                extra += 2 * n * h2 * h2 * pl * 9 * pl
            else:
                h2 = conv(f'l{li+1}.{bi}.c2', h, pl, pl, 3, s)
            chained_c1 = will_chain
            tag = f'l{li+1}.{bi}.'
            if chained_c1:
                m = n * h2 * h2
                k3 = pl + (inp if bi == 0 else 0)
                nr = 2 * pl if boundary else pl
                name = tag + ('c3+ds>c1' if bi == 0 else
                              f'c3>l{li+2}.0.c1' if boundary else 'c3>c1')
                if front:
                    name = tag + ('c1>c2>' if front_c1 else 'c2>') + name[len(tag):]
                layers.append((name, m, pl * 4, k3 + nr,
                               2 * m * pl * 4 * k3 + 2 * m * nr * pl * 4 + extra))
            elif bi == 0 and not fused:
                conv(tag + 'ds', h, inp, pl * 4, 1, s)
                conv(tag + 'c3', h2, pl, pl * 4, 1, 1)
            elif bi == 0:
                conv(tag + 'c3+ds', h2, pl, pl * 4, 1, 1, extra_k=inp)
            else:
                conv(tag + 'c3', h2, pl, pl * 4, 1, 1)
            h, inp = h2, pl * 4
    return layers


def group_key(name):
    return name if '.0.' in name.split('>')[0] or name == 'stem' else re.sub(
        r'\.\d+\.', '.x.', name, count=1)


def main():
    db = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    fused = len(sys.argv) > 4 and sys.argv[4] == '1'  # split mode: ds folded into c3
    chain = len(sys.argv) > 5 and sys.argv[5] in ("1", "2", "3")
    wide = len(sys.argv) > 5 and sys.argv[5] in ("2", "3")  # layer3 chains on too
    bneck = len(sys.argv) > 5 and sys.argv[5] == "3"        # round 6: layer1's conv front
    rows = db.execute(
        "select name, start, end-start, grid_x from kernels where " + GEMM_LIKE +
        " order by start").fetchall()
    per = len(rows) // passes
    rows = rows[per * (passes - 1):]
    layers = build_layers(n, fused, chain, wide, bneck)
    agg, tot_t, tot_f = {}, 0, 0
    for (name, m, nn, k, fl), (_, _, du, _) in zip(layers, rows):
        key = group_key(name)
        a = agg.setdefault(key, [0, 0, 0, m, nn, k])
        a[0] += fl
        a[1] += du
        a[2] += 1
        tot_t += du
        tot_f += fl
    print(f'encoder GEMMs: {tot_f/1e12:.2f} TFLOP in {tot_t/1e6:.1f} ms = '
          f'{tot_f/tot_t/1e3:.1f} TF/s')
    for k, (fl, du, c, m, nn, kk) in agg.items():
        # round 6 (MILAN_FUSE_SPARSE_TAIL): the non-first blocks of the last stage run at the rows
        # the level-4 pooling and their neighbourhoods need -- their TF/s here is DENSE-EQUIVALENT
        note = '  (mask-aware rows: dense-equivalent TF/s)' if bneck and k.startswith('l4.x.') else ''
        print(f'{k:9s} x{c:2d} M={m:8d} N={nn:5d} K={kk:5d} {du/1e6:8.2f} ms '
              f'{fl/du/1e3:6.1f} TF/s {100*du/tot_t:5.1f}%{note}')
    dec = rows[len(layers):]
    print('decoder+lm GEMM launches', len(dec), 'time ms',
          sum(r[2] for r in dec) / 1e6)
    others = db.execute(
        "select name, count(*), sum(end-start) from kernels where not (" + GEMM_LIKE +
        ") group by name order by 3 desc limit 12").fetchall()
    for nm, c, t in others:
        print(f'  {nm[:60]:60s} x{c:5d} {t/1e6/passes:8.2f} ms/pass')


if __name__ == '__main__':
    main()
