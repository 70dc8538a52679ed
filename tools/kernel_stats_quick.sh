#!/bin/bash
# Quick per-kernel times of two bench steps (run on the GPU box): rocprofv3 --kernel-trace, top kernels by total time.
# (--live-traffic 0: bench.py's own rocprofv3 leg must not run inside this one)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/ks -o p -- python /root/repo/bench.py --chunk ${CHUNK:-640} --steps 1 --warmup 1 --also-f32-steps 0 --cpu-sample 0 --no-profile --from-host-steps 0 --other-configs 0 --fast-steps 0 --live-traffic 0 > /tmp/ks.out 2> /tmp/ks.err
tail -3 /tmp/ks.err | cut -c1-300
f=$(find /tmp/ks -name "*.db" | head -1)
python - <<PY
import sqlite3
db=sqlite3.connect("$f")
tot=db.execute("select sum(end-start) from kernels").fetchone()[0]
print("kernel,calls,total_ms,avg_us,pct")
for r in db.execute("select name,count(*),sum(end-start)/1e6,avg(end-start)/1e3,100.0*sum(end-start)/%d from kernels group by name order by 3 desc limit 40" % tot):
    print('"%s",%d,%.2f,%.1f,%.2f' % (r[0][:90],r[1],r[2],r[3],r[4]))
PY
