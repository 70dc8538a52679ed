#!/bin/bash
# Same-box A/B of an environment knob (run on the GPU box, e.g.
#   gpurun -- 'bash tools/ab_env.sh MILAN_LSTM_FUSE 1 0 1 0'):
# one short default-workload bench per value, in the order given; prints value,
# token hash (results must not move) and the per-stage times the knob could touch.
# Boxes differ by a few percent in sustained clock, so only runs of ONE call compare.
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --chunk 256 --cpu-sample 0 --also-f32-steps 0 --other-configs 0 --fast-steps 0 --from-host-steps 0 --live-traffic 0 ${AB_EXTRA:-} > /tmp/ab.json 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python - <<PY
import json
d = json.load(open("/tmp/ab.json"))
st = {s["stage"]: s for s in d["roofline"]["stages"]}
keys = ("encoder.input", "encoder.stem", "encoder.layer1", "encoder.layer2", "encoder.layer3", "encoder.layer4", "decoder.search", "decoder.lm_rerank")
print("$var=$v", round(d["value"], 1), d["config"].get("gathered_tokens_sha256", "")[:12],
      " ".join("%s %.2f" % (k.split(".")[1], st[k]["ms_per_step"]) for k in keys))
PY
done
