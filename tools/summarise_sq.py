"""Merge the per-pass outputs of tools/pmc_sq.sh into one SQ-counter summary.

    python tools/summarise_sq.py gpurun_out/r5sq1_pmc.txt gpurun_out/r5sq2_pmc.txt ... > profiles/rN_sq_counters.txt

Passes (rocprofv3 --pmc takes one hardware-compatible group at a time):
    tools/pmc_sq.sh r5sq1 "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16"
    tools/pmc_sq.sh r5sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
    tools/pmc_sq.sh r5sq3 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES); LDS conflict share =
SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; wave-cycle shares over SQ_WAVE_CYCLES.
"""
import collections
import re
import sys

rows = collections.defaultdict(dict)
launches = {}
raw = []
for path in sys.argv[1:]:
    for line in open(path):
        m = re.match(r'(.*) (SQ_\w+) (\d+) ([\d.e+]+)$', line.rstrip())
        if not m:
            continue
        kernel, counter, n, value = m.group(1), m.group(2), int(m.group(3)), float(m.group(4))
        rows[kernel][counter] = value
        launches[kernel] = n
        raw.append(line.rstrip())


def pct(a, b):
    return 100.0 * a / b if b else float('nan')


tot_busy = sum(r.get('SQ_BUSY_CYCLES', 0) for r in rows.values())
tot_mfma = sum(r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for r in rows.values())
print('# SQ counters of the GEMM-class kernels, one 256-neuron step: rocprofv3 --kernel-trace --pmc <one group per pass> '
      '-- python bench.py --chunk 256 --steps 1 --warmup 0 ... (tools/pmc_sq.sh; sums over all launches).')
print('# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES): %.1f %% over all GEMM-class kernels.' % pct(tot_mfma, 32 * tot_busy))
print('# kernel | launches | MFMA busy % | LDS bank-conflict cycles / LDS-active cycles % | wave cycles: waiting (s_waitcnt / barrier) % | issue-stalled % | issuing %')
for kernel, r in sorted(rows.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0)):
    print('#   %-64s x%3d  mfma %5.1f  lds-conflict %5.1f  wait %5.1f  issue-stall %5.1f  issuing %5.1f' % (
        kernel[:64], launches[kernel],
        pct(r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), 32 * r.get('SQ_BUSY_CYCLES', 0)),
        pct(r.get('SQ_LDS_BANK_CONFLICT', 0), r.get('SQ_LDS_IDX_ACTIVE', 0)),
        pct(r.get('SQ_WAIT_ANY', 0), r.get('SQ_WAVE_CYCLES', 0)),
        pct(r.get('SQ_WAIT_INST_ANY', 0), r.get('SQ_WAVE_CYCLES', 0)),
        pct(r.get('SQ_ACTIVE_INST_ANY', 0), r.get('SQ_WAVE_CYCLES', 0))))
print('# raw: kernel, counter, launches, sum')
for line in sorted(raw):
    print(line)
