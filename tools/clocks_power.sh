#!/bin/bash
# Run on the GPU box (e.g. gpurun -- tools/clocks_power.sh ...); results land in gpurun_out/.
cd /root/repo
python bench.py --chunk 256 --steps 48 --warmup 1 --also-f32-steps 0 --cpu-sample 0 --no-profile --from-host-steps 0 --other-configs 0 --fast-steps 0 > /tmp/b.log 2>&1 &
BP=$!
for i in $(seq 1 150); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.3; kill -0 $BP 2>/dev/null || break; done > gpurun_out/clk.txt
wait $BP
tail -1 /tmp/b.log | cut -c1-200
