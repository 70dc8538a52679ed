#!/bin/bash
# Run on the GPU box (e.g. gpurun -- tools/pmc_traffic.sh ...); results land in gpurun_out/.
cd /root/repo
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$c
  timeout 500 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python /root/repo/bench.py --steps 1 --warmup 0 --also-f32-steps 0 --cpu-sample 0 --no-profile --from-host-steps 0 --other-configs 0 --fast-steps 0 --live-traffic 0 2>&1 | tail -1 | cut -c1-100
  cd /root/repo
  f=$(find /tmp/pmc_$c -name "*.db" | head -1)
  python - > gpurun_out/traffic_$c.txt <<PY
import sqlite3
db=sqlite3.connect("$f")
q="select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name order by 4 desc"
for r in db.execute(q): print(r[0][:90].replace(' ','_'), r[1], r[2], r[3])
PY
done
