#!/bin/bash
# Same-box sweep of the neurons-per-launch knob (run on the GPU box: gpurun -- 'bash tools/chunk_sweep.sh 640 624 ...').
for c in "$@"; do
  python bench.py --chunk $c --cpu-sample 0 --also-f32-steps 0 --other-configs 0 --fast-steps 0 --from-host-steps 0 --live-traffic 0 > /tmp/cs.json 2>/dev/null
  python - $c <<'PY'
import json, sys
d = json.loads(open('/tmp/cs.json').read().strip().splitlines()[-1])
st = {s['stage']: s for s in d['roofline']['stages']}
print('chunk', sys.argv[1], round(d['value'], 1), ' per 256 neurons: layer3', round(st['encoder.layer3']['ms_per_256_neurons'], 2),
      'layer4', round(st['encoder.layer4']['ms_per_256_neurons'], 2), 'search', round(st['decoder.search']['ms_per_256_neurons'], 2),
      'lm', round(st['decoder.lm_rerank']['ms_per_256_neurons'], 2))
PY
done
