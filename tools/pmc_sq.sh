#!/bin/bash
# Run on the GPU box (e.g. gpurun -- tools/pmc_sq.sh ...); results land in gpurun_out/.
# usage: scratch/pmc.sh <out_prefix> "<counters>"
out=$1; ctrs=$2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc
timeout 500 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc -o p -- python /root/repo/bench.py --chunk 256 --steps 1 --warmup 0 --also-f32-steps 0 --cpu-sample 0 --no-profile --from-host-steps 0 --other-configs 0 --fast-steps 0 --live-traffic 0 2>&1 | tail -1 | cut -c1-120
cd /root/repo
f=$(find /tmp/pmc -name "*.db" | head -1)
python - > gpurun_out/${out}_pmc.txt <<PY
import sqlite3
db=sqlite3.connect("$f")
tabs=[r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'pmc' in t.lower() or 'counter' in t.lower()])
try:
    q="select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name order by 1,2"
    for r in db.execute(q):
        if any(k in r[0] for k in ('igemm', 'chain_kernel', 'chain3_kernel', 'stem_fused', 'conv3_p64')): print(r[0][:70], r[1], r[2], r[3])
except Exception as e:
    print('ERR', e)
    for t in tabs: 
        if 'pmc' in t.lower() or 'counter' in t.lower():
            print(t, [c[1] for c in db.execute(f"pragma table_info({t})")])
PY
