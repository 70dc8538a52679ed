#!/bin/bash
# Run on the GPU box (e.g. gpurun -- tools/bench_and_kernel_stats.sh ...); results land in gpurun_out/.
# default bench (the number the driver will see) + rocprofv3 kernel stats of the same command
cd /root/repo
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 2500 gpurun_out/bench_default.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python /root/repo/bench.py --cpu-sample 0 --also-f32-steps 0 --from-host-steps 0 --other-configs 0 --fast-steps 0 --live-traffic 0 > /tmp/prof_bench.json 2>/tmp/prof.err
cd /root/repo
f=$(find /tmp/prof -name "*.db" | head -1)
python - > gpurun_out/kernel_stats.csv <<PY
import sqlite3
db=sqlite3.connect("$f")
tot=db.execute("select sum(end-start) from kernels").fetchone()[0]
print("kernel,calls,total_ms,avg_us,min_us,max_us,pct")
for r in db.execute("select name,count(*),sum(end-start)/1e6,avg(end-start)/1e3,min(end-start)/1e3,max(end-start)/1e3,100.0*sum(end-start)/%d from kernels group by name order by 3 desc" % tot):
    print('"%s",%d,%.2f,%.1f,%.1f,%.1f,%.2f' % (r[0][:110],r[1],r[2],r[3],r[4],r[5],r[6]))
PY
# (last argument 3: layer3 chains of csrc/chain3.hip + layer1 conv front, round 6)
python tools/layer_report.py $f 9600 17 1 3 > gpurun_out/layer_report.txt
tail -1 /tmp/prof_bench.json | cut -c1-300
