#!/usr/bin/env python3
"""Build-time ISA check of the hand-scheduled kernels (ADVICE r3 / VERDICT r4 item 8).

The ping-pong GEMM, the chain kernels, the stem and the layer1 3x3 kernel order their
LDS-DMA rings with COUNTED waits -- `s_waitcnt vmcnt(N)` where N is the number of
load-class VMEM operations the source issues behind the pieces that must have landed.
Those counts are only right while the compiler emits exactly the VMEM operations the
source spells: a toolchain that merged two 16-byte loads, split one, or spilled a
register (a scratch reload is a VMEM load AND forces vmcnt(0)) would silently change
what a count means.  This script disassembles the built code objects and asserts, per
hot kernel,

  * no scratch (spill) traffic beyond the pinned number,
  * the pinned number of MFMAs, LDS-DMA pieces, global 16-byte loads / stores and
    barriers -- the operations the hand-counted waits were written against,
  * the set of vmcnt immediates in use.

A mismatch is not necessarily a bug; it means the waits of that kernel must be
re-validated against the new instruction stream (and the table below updated).

    python tools/check_isa.py [--build-dir neuron-descriptions_amd/csrc/build] [--print]

Runs in `__graft_entry__.build()` and as `make -C neuron-descriptions_amd/csrc check-isa`.
No GPU needed.
"""
import argparse
import pathlib
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'

# kernel (substring of the mangled name) -> pinned figures.  `None` = not pinned.
PINNED = {
    'chain3_kernelILb0': {'mfma': 192, 'lds_dma': 64, 'global_load_x4': 104, 'global_store_x4': 72, 'barriers': 14, 'scratch': 4, 'vmcnt': [0, 1, 2, 4, 5, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30]},
    'igemm_split16_pp32_kernel': {'mfma': 576, 'lds_dma': 112, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4]},
    'igemm_split16_pp32n_kernelILi128': {'mfma': 288, 'lds_dma': 94, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 53, 'scratch': 0, 'vmcnt': [0, 4]},
    'igemm_f16_pp32_kernelILi256ELb0': {'mfma': 384, 'lds_dma': 112, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4]},
    'igemm_f16_pp32_kernelILi128ELb0': {'mfma': 192, 'lds_dma': 84, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4]},
    'igemm_split16_pp32t_kernelILi256': {'mfma': 576, 'lds_dma': 112, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4]},
    'igemm_split16_pp32t_kernelILi128': {'mfma': 288, 'lds_dma': 84, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4]},
    'igemm_f16_pp32_kernelILi256ELb1': {'mfma': 384, 'lds_dma': 112, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4]},
    'igemm_f16_pp32_kernelILi128ELb1': {'mfma': 192, 'lds_dma': 84, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4]},
    'chain_kernelILi128ELi8': {'mfma': 96, 'lds_dma': 14, 'global_load_x4': 40, 'global_store_x4': 24, 'barriers': 6, 'scratch': 6, 'vmcnt': [0, 1, 2, 12]},
    'chain_kernelILi64ELi8ELb0ELi0ELb1': {'mfma': 48, 'lds_dma': 10, 'global_load_x4': 30, 'global_store_x4': 16, 'barriers': 4, 'scratch': 0, 'vmcnt': [0, 2, 12]},
    'chain_kernelILi64ELi8ELb0ELi64ELb1': {'mfma': 72, 'lds_dma': 12, 'global_load_x4': 22, 'global_store_x4': 16, 'barriers': 5, 'scratch': 0, 'vmcnt': [0, 2, 4]},
    'chain_kernelILi64ELi8ELb0ELi0ELb0ELb0ELi128': {'mfma': 72, 'lds_dma': 12, 'global_load_x4': 32, 'global_store_x4': 24, 'barriers': 5, 'scratch': 0, 'vmcnt': [0, 1, 2, 12]},
    'conv3_p64_kernel': {'mfma': 162, 'lds_dma': 20, 'global_load_x4': 36, 'global_store_x4': 4, 'barriers': 5, 'scratch': 0, 'vmcnt': [0, 5]},
    'stem_fused_kernelILi7ELi8ELi8ELb0': {'mfma': 84, 'lds_dma': 6, 'global_load_x4': 33, 'global_store_x4': 10, 'barriers': 3, 'scratch': 6, 'vmcnt': [0]},
    'stem_fused_kernelILi7ELi8ELi8ELb1': {'mfma': 84, 'lds_dma': 0, 'global_load_x4': 33, 'global_store_x4': 10, 'barriers': 5, 'scratch': 8, 'vmcnt': [0]},
}


def disassemble(obj: pathlib.Path) -> str:
    with tempfile.TemporaryDirectory() as tmp:
        tmp = pathlib.Path(tmp)
        shutil.copy(obj, tmp / 'x.o')
        subprocess.run([OBJDUMP, '--offloading', 'x.o'], cwd=tmp, check=True,
                       capture_output=True)
        cos = sorted(tmp.glob('x.o.*gfx950*'))
        if not cos:
            raise RuntimeError(f'{obj}: no gfx950 code object inside')
        return subprocess.run([OBJDUMP, '-d', str(cos[0])], check=True,
                              capture_output=True, text=True).stdout


def kernels(text: str):
    """name -> list of instruction lines"""
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur is not None and line.startswith('\t'):
            out[cur].append(line.strip())
    return out


def figures(lines):
    def count(pattern):
        return sum(1 for ln in lines if re.match(pattern, ln))
    vm = sorted({int(m.group(1)) for ln in lines
                 for m in [re.search(r's_waitcnt.*vmcnt\((\d+)\)', ln)] if m})
    return {
        'mfma': count(r'v_mfma_'),
        'lds_dma': count(r'(buffer_load_\w+ .*\blds\b|global_load_lds_)'),
        'global_load_x4': count(r'global_load_dwordx4'),
        'global_store_x4': count(r'global_store_dwordx4'),
        'barriers': count(r's_barrier'),
        'scratch': count(r'scratch_'),
        'vmcnt': vm,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--build-dir', default=str(ROOT / 'neuron-descriptions_amd' / 'csrc' / 'build'))
    ap.add_argument('--print', action='store_true', help='print the figures (to update PINNED)')
    args = ap.parse_args()
    build = pathlib.Path(args.build_dir)
    seen, bad = {}, []
    for obj in ('chain3.o', 'gemm.o', 'chain.o', 'conv3.o', 'stem.o'):
        path = build / obj
        if not path.exists():
            bad.append(f'{path} is missing: build the library first')
            continue
        for name, lines in kernels(disassemble(path)).items():
            for key in PINNED:
                if key in name:
                    seen[key] = figures(lines)
    for key, want in PINNED.items():
        got = seen.get(key)
        if got is None:
            bad.append(f'{key}: kernel not found in the build')
            continue
        if args.print:
            print(f"    {key!r}: {got},")
        for field, value in want.items():
            if value is not None and got[field] != value:
                bad.append(f'{key}: {field} = {got[field]}, pinned {value}')
    if bad:
        print('check_isa: the instruction streams the counted waits were written against '
              'have changed:\n  ' + '\n  '.join(bad), file=sys.stderr)
        return 1
    if not args.print:
        print(f'check_isa: {len(PINNED)} kernels match their pinned VMEM / MFMA / barrier counts')
    return 0


if __name__ == '__main__':
    sys.exit(main())
