"""Fold the two PMC passes of tools/pmc_traffic.sh into profiles/r3_hbm_traffic.json.

Usage (in the repo, after the gpurun call merged gpurun_out/traffic_*.txt):
    python tools/summarise_traffic.py [gpurun_out] [profiles/r3_hbm_traffic.json]

Counter unit is KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for
gfx950 (128-B requests of 16-B/lane streams are tallied at 64 B); WRITE_SIZE is
taken as reported.  bench.py reads `traffic_gb_per_launch` of the dominant kernel.
"""
import json
import re
import sys

src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out'
dst = sys.argv[2] if len(sys.argv) > 2 else 'profiles/r5_hbm_traffic.json'


def read(counter):
    table = {}
    for line in open(f'{src}/traffic_{counter}.txt'):
        name, ctr, launches, total = line.rsplit(None, 3)
        if ctr != counter or not any(k in name for k in ('igemm', 'chain_kernel', 'chain3_kernel', 'stem_fused', 'conv3_p64')):
            continue
        key = re.sub(r'^void_milan::|\(.*$|_', '', name)
        table[key] = (int(launches), float(total))
    return table


fetch, write = read('FETCH_SIZE'), read('WRITE_SIZE')
# the ping-pong tile and its tap-inner form (round 5) are one kernel family (bench.py: pp32_256)
for table in (fetch, write):
    fam = [k for k in table if k in ('milan::igemmsplit16pp32kernel', 'igemmsplit16pp32tkernel<256>')]
    if len(fam) == 2:
        a, b = table.pop(fam[0]), table.pop(fam[1])
        table['igemmsplit16pp32kernel+pp32tkernel<256>'] = (a[0] + b[0], a[1] + b[1])
kernels = []
for key, (launches, fetch_kb) in sorted(fetch.items(), key=lambda kv: -kv[1][1]):
    write_kb = write.get(key, (launches, 0.0))[1]
    kernels.append({
        'kernel': key,
        'launches': launches,
        'fetch_kb_raw': fetch_kb,
        'write_kb': write_kb,
        'fetch_gb_per_launch_raw': fetch_kb * 1024 / launches / 1e9,
        'fetch_gb_per_launch_corrected': 2 * fetch_kb * 1024 / launches / 1e9,
        'write_gb_per_launch': write_kb * 1024 / launches / 1e9,
    })
top = kernels[0]
out = {
    'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (two separate passes, tools/pmc_traffic.sh) -- '
               'python bench.py --steps 1 --warmup 0 --also-f32-steps 0 --cpu-sample 0 --no-profile '
               '--from-host-steps 0 --other-configs 0',
    'unit_note': 'counter unit KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests of '
                 '16-B/lane streams at 64 B); WRITE_SIZE uncorrected',
    'precision': 'split_f16',
    'dominant_kernel': top['kernel'],
    'traffic_gb_per_launch': top['fetch_gb_per_launch_corrected'] + top['write_gb_per_launch'],
    'kernels': kernels,
}
json.dump(out, open(dst, 'w'), indent=1)
print(dst, out['dominant_kernel'], round(out['traffic_gb_per_launch'], 3), 'GB/launch')
