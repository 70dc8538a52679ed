#!/bin/bash
# Per-LAYER HBM traffic of the encoder GEMMs: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), one
# bench step, dispatches mapped onto the ResNet-101 layer list by launch order.  Results land in gpurun_out/.
cd /root/repo
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmcl_$c
  timeout 500 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcl_$c -o p -- python /root/repo/bench.py --chunk 256 --steps 1 --warmup 0 --also-f32-steps 0 --cpu-sample 0 --no-profile --from-host-steps 0 --other-configs 0 --fast-steps 0 --live-traffic 0 2>&1 | tail -1 | cut -c1-80
  cd /root/repo
  f=$(find /tmp/pmcl_$c -name "*.db" | head -1)
  python - "$f" $c > gpurun_out/pmc_dispatch_$c.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
# counters_collection view: one row per (dispatch, counter)
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
print('#', cols)
q = ("select dispatch_id, kernel_name, sum(value) from counters_collection where kernel_name like '%igemm%' "
     "or kernel_name like '%chain_kernel%' or kernel_name like '%chain3_kernel%' or kernel_name like '%stem_fused%' or kernel_name like '%conv3_p64%' "
     "group by dispatch_id, kernel_name order by dispatch_id")
for r in db.execute(q):
    print(r[0], r[1].split('(')[0].replace(' ', ''), r[2])
PY
done
python - <<'PY'
import sys
sys.path.insert(0, 'tools')
from layer_report import build_layers, group_key
def load(c):
    rows = []
    for line in open(f'gpurun_out/pmc_dispatch_{c}.txt'):
        if line.startswith('#'): continue
        p = line.split()
        rows.append((int(p[0]), p[1], float(p[2])))
    return rows
F, W = load('FETCH_SIZE'), load('WRITE_SIZE')
n = 3840
layers = build_layers(n, True, True, True, True)  # split mode: folded downsamples, layer1/2/3 chains, layer1 conv front
agg = {}
for (name, m, nn, k, fl), f, w in zip(layers, F, W):
    a = agg.setdefault(group_key(name), [0, 0.0, 0.0, m, nn, k, f[1][:40]])
    a[0] += 1; a[1] += f[2]; a[2] += w[2]
print('layer              x   fetch GB/launch (x2 corrected)   write GB/launch   M x N x 4 B (GB)   kernel')
for key, (c, f, w, m, nn, k, kern) in agg.items():
    print(f'{key:18s} x{c:2d}  fetch {2*f*1024/c/1e9:7.3f}  write {w*1024/c/1e9:7.3f}   out {m*nn*4/1e9:6.3f}   {kern}')
PY
