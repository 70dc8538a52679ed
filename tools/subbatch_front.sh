#!/bin/bash
# VERDICT r5 item 1, "the cheap bound": how much of the HBM-bound front (stem, layer1, layer2) is HBM
# *bandwidth*?  Run the encoder over sub-batches small enough for the 256 MB Infinity Cache
# (MILAN_ENC_SUB images per pass; 16 MB of activations per image -> 24 images = 384 MB of workspace
# of which ~half is live at a time) and compare the summed kernel time of the front kernels per
# 3840 images with the whole-chunk pass.  Run on the GPU box; prints one table per sub-batch size.
cd /tmp && export TMPDIR=/tmp
for sub in ${SUBS:-3840 16 24 48 96 240}; do
  rm -rf /tmp/ks
  MILAN_ENC_SUB=$sub timeout -k 10 600 rocprofv3 --kernel-trace -d /tmp/ks -o p -- python /root/repo/bench.py --chunk 256 --steps 1 --warmup 1 --also-f32-steps 0 --cpu-sample 0 --no-profile --from-host-steps 0 --other-configs 0 --fast-steps 0 --live-traffic 0 > /tmp/ks.out 2> /tmp/ks.err
  f=$(find /tmp/ks -name "*.db" | head -1)
  python - <<PY
import sqlite3, re
db = sqlite3.connect("$f")
rows = db.execute("select name, count(*), sum(end-start)/1e6 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("== MILAN_ENC_SUB=$sub: all kernels %.1f ms over 2 passes (warm-up + 1 step)" % tot)
front = 0.0
for name, calls, ms in rows:
    short = re.sub(r"\(.*", "", name)[:80]
    is_front = any(k in name for k in ("stem_fused", "conv3_p64", "chain_kernel"))
    if is_front: front += ms
    if ms / tot > 0.004: print("  %-80s x%6d %9.2f ms%s" % (short, calls, ms, "  [front]" if is_front else ""))
print("  front (stem + conv3_p64 + chain_kernel) = %.2f ms per 2 x 3840 images" % front)
PY
done
