cd /root/repo
for v in 15 31 15 31; do
  MILAN_CHAIN=$v python bench.py --steps 6 --cpu-sample 0 --also-f32-steps 0 --other-configs 0 --fast-steps 0 --live-traffic 0 > /tmp/p.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/p.json')); print('MILAN_CHAIN=$v', round(d['value'],1), round(d['pcie_inclusive_value'],1), round(d['pcie_inclusive_value']/d['value'],4))"
done
