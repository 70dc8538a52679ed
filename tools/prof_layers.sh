#!/bin/bash
# Run on the GPU box (e.g. gpurun -- tools/prof_layers.sh ...); results land in gpurun_out/.
# usage: scratch/prof_layers.sh <out_prefix> [extra bench args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 500 rocprofv3 --kernel-trace -d /tmp/prof -o bench -- python /root/repo/bench.py --chunk 256 --steps 3 --warmup 0 --also-f32-steps 0 --cpu-sample 0 --no-profile --from-host-steps 0 --other-configs 0 --fast-steps 0 "$@" 2>&1 | tail -1 | cut -c1-150
cd /root/repo
f=$(find /tmp/prof -name "*.db" | head -1)
python tools/layer_report.py $f 3840 3 1 1 > gpurun_out/${out}_layers.txt
python - > gpurun_out/${out}_kernels.txt <<PY
import sqlite3
db=sqlite3.connect("$f")
for r in db.execute("select name,count(*),avg(end-start)/1e3,sum(end-start)/1e6 from kernels group by name order by 4 desc limit 16"): print(r[0][:100],r[1],round(r[2],1),round(r[3],1))
PY
