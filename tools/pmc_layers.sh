#!/bin/bash
# Per-LAYER HBM traffic of the encoder GEMMs: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), one
# bench step, dispatches mapped onto the ResNet-101 layer list by launch order.  Results land in gpurun_out/.
cd /root/repo
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmcl_$c
  timeout 500 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcl_$c -o p -- python /root/repo/bench.py --chunk 256 --steps 1 --warmup 0 --also-f32-steps 0 --cpu-sample 0 --no-profile --from-host-steps 0 --other-configs 0 2>&1 | tail -1 | cut -c1-80
  cd /root/repo
  f=$(find /tmp/pmcl_$c -name "*.db" | head -1)
  python - "$f" $c > gpurun_out/pmc_dispatch_$c.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
# counters_collection view: one row per (dispatch, counter)
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
print('#', cols)
q = "select dispatch_id, kernel_name, sum(value) from counters_collection where kernel_name like '%igemm%' group by dispatch_id, kernel_name order by dispatch_id"
for r in db.execute(q):
    print(r[0], r[1].split('(')[0].replace(' ', ''), r[2])
PY
done
python - <<'PY'
import re
def load(c):
    rows = []
    for line in open(f'gpurun_out/pmc_dispatch_{c}.txt'):
        if line.startswith('#'): continue
        p = line.split()
        rows.append((int(p[0]), p[1], float(p[2])))
    return rows
F, W = load('FETCH_SIZE'), load('WRITE_SIZE')
n = 3840
layers = []
def conv(name, h, cin, cout, k, s, extra=0):
    ho = (h + 2*(k//2) - k)//s + 1
    m = n*ho*ho
    layers.append((name, m, cout, k*k*cin + extra, cin, ho))
    return ho
conv('stem', 224, 3, 64, 7, 2)
h, inp = 56, 64
for li, nb in enumerate((3,4,23,3)):
    pl = 64*2**li
    for bi in range(nb):
        s = 2 if (bi == 0 and li > 0) else 1
        conv(f'l{li+1}.{bi}.c1', h, inp, pl, 1, 1)
        h2 = conv(f'l{li+1}.{bi}.c2', h, pl, pl, 3, s)
        if bi == 0: conv(f'l{li+1}.{bi}.c3+ds', h2, pl, pl*4, 1, 1, extra=inp)
        else: conv(f'l{li+1}.{bi}.c3', h2, pl, pl*4, 1, 1)
        h, inp = h2, pl*4
agg = {}
for (name, m, nn, k, cin, ho), f, w in zip(layers, F, W):
    key = name if '.0.' in name or name == 'stem' else re.sub(r'\.\d+\.', '.x.', name)
    a = agg.setdefault(key, [0, 0.0, 0.0, m, nn, k])
    a[0] += 1; a[1] += f[2]; a[2] += w[2]
print('layer        x   fetch GB/launch (x2 corrected)   write GB/launch   algorithmic out GB')
for key, (c, f, w, m, nn, k) in agg.items():
    print(f'{key:10s} x{c:2d}  fetch {2*f*1024/c/1e9:7.3f}  write {w*1024/c/1e9:7.3f}   out {m*nn*4/1e9:6.3f}  in {m*k*4/1e9 if "c2" not in key else m*k*4/9/1e9:6.3f}')
PY
