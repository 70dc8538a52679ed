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

Round 6: two MECHANICAL checks on top of the counts.

  * Pending-destination hazards.  chain3 / chain / conv3 issue `ds_read_b128` through inline
    asm with "=v" outputs and hand-counted `s_waitcnt lgkmcnt(N)`: between the read and the
    wait that covers it the compiler believes the destination is defined and may copy,
    spill or overwrite it.  `pending_hazards` walks the control-flow graph of every hot
    kernel with the in-order model of the LGKM (LDS) and VM (load) queues -- an LDS read
    is complete once a `lgkmcnt(N)` was passed with at least N LDS operations issued behind
    it, a VMEM load likewise for `vmcnt(N)` and later loads; stores and scalar loads only
    make the hardware stricter -- and fails if ANY instruction mentions a register that a
    still-pending read will write (except a later read of the same queue re-targeting it:
    they return in order).  It covers compiler-generated loads too, so the hand-counted
    vmcnt immediates are validated against what is really in flight.
  * No scratch traffic inside a loop (a scratch reload is a VMEM load and the compiler
    drains the queue for it: a full stall of the weight stream) beyond the pinned number.

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
    'chain3_kernelILb0': {'mfma': 192, 'lds_dma': 64, 'global_load_x4': 104, 'global_store_x4': 72, 'barriers': 14, 'scratch': 4, 'vmcnt': [0, 1, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30], 'scratch_in_loops': 1, 'hazards': 0},
    'igemm_split16_pp32_kernel': {'mfma': 576, 'lds_dma': 112, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4], 'scratch_in_loops': 0, 'hazards': 0},
    'igemm_split16_pp32n_kernelILi128': {'mfma': 288, 'lds_dma': 94, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 53, 'scratch': 0, 'vmcnt': [0, 4], 'scratch_in_loops': 0, 'hazards': 0},
    'igemm_f16_pp32_kernelILi256ELb0': {'mfma': 384, 'lds_dma': 112, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4], 'scratch_in_loops': 0, 'hazards': 0},
    'igemm_f16_pp32_kernelILi128ELb0': {'mfma': 192, 'lds_dma': 84, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4], 'scratch_in_loops': 0, 'hazards': 0},
    'igemm_split16_pp32t_kernelILi256': {'mfma': 576, 'lds_dma': 112, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4], 'scratch_in_loops': 0, 'hazards': 0},
    'igemm_split16_pp32t_kernelILi128': {'mfma': 288, 'lds_dma': 84, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4], 'scratch_in_loops': 0, 'hazards': 0},
    'igemm_f16_pp32_kernelILi256ELb1': {'mfma': 384, 'lds_dma': 112, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4], 'scratch_in_loops': 0, 'hazards': 0},
    'igemm_f16_pp32_kernelILi128ELb1': {'mfma': 192, 'lds_dma': 84, 'global_load_x4': 0, 'global_store_x4': 0, 'barriers': 52, 'scratch': 0, 'vmcnt': [0, 4], 'scratch_in_loops': 0, 'hazards': 0},
    'chain_kernelILi128ELi8ELb0ELi0ELb0ELb0ELi128ELb0ELb0': {'mfma': 96, 'lds_dma': 14, 'global_load_x4': 40, 'global_store_x4': 24, 'barriers': 6, 'scratch': 6, 'vmcnt': [0, 1, 2, 12], 'scratch_in_loops': 0, 'hazards': 0},
    'chain_kernelILi64ELi8ELb0ELi0ELb1ELb0ELi64ELb0ELb0': {'mfma': 48, 'lds_dma': 10, 'global_load_x4': 30, 'global_store_x4': 16, 'barriers': 4, 'scratch': 0, 'vmcnt': [0, 2, 12], 'scratch_in_loops': 0, 'hazards': 0},
    'chain_kernelILi64ELi8ELb0ELi64ELb1ELb0ELi64ELb0ELb0': {'mfma': 72, 'lds_dma': 12, 'global_load_x4': 22, 'global_store_x4': 16, 'barriers': 5, 'scratch': 0, 'vmcnt': [0, 2, 4], 'scratch_in_loops': 0, 'hazards': 0},
    'chain_kernelILi64ELi8ELb0ELi0ELb0ELb0ELi128ELb0ELb0': {'mfma': 72, 'lds_dma': 12, 'global_load_x4': 32, 'global_store_x4': 24, 'barriers': 5, 'scratch': 0, 'vmcnt': [0, 1, 2, 12], 'scratch_in_loops': 0, 'hazards': 0},
    # round 6: the same chains with layer1's 3x3 conv in front (chain_kernel<.., CONV>)
    'chain_kernelILi64ELi8ELb0ELi0ELb1ELb0ELi64ELb1ELb0': {'mfma': 264, 'lds_dma': 29, 'global_load_x4': 30, 'global_store_x4': 16, 'barriers': 13, 'scratch': 0, 'vmcnt': [0, 2, 12], 'scratch_in_loops': 0, 'hazards': 0},
    'chain_kernelILi64ELi8ELb0ELi64ELb1ELb0ELi64ELb1ELb0': {'mfma': 288, 'lds_dma': 31, 'global_load_x4': 22, 'global_store_x4': 16, 'barriers': 14, 'scratch': 0, 'vmcnt': [0, 2, 4], 'scratch_in_loops': 0, 'hazards': 0},
    'chain_kernelILi64ELi8ELb0ELi0ELb0ELb0ELi128ELb1ELb0': {'mfma': 288, 'lds_dma': 31, 'global_load_x4': 32, 'global_store_x4': 24, 'barriers': 14, 'scratch': 0, 'vmcnt': [0, 1, 2, 12], 'scratch_in_loops': 0, 'hazards': 0},
    # ... and with the block's own c1 in front as well (layer1.0: the whole bottleneck)
    'chain_kernelILi64ELi8ELb0ELi64ELb1ELb0ELi64ELb1ELb1': {'mfma': 312, 'lds_dma': 33, 'global_load_x4': 30, 'global_store_x4': 16, 'barriers': 15, 'scratch': 0, 'vmcnt': [0, 2, 4], 'scratch_in_loops': 0, 'hazards': 0},
    'conv3_p64_kernel': {'mfma': 162, 'lds_dma': 20, 'global_load_x4': 36, 'global_store_x4': 4, 'barriers': 5, 'scratch': 0, 'vmcnt': [0, 5], 'scratch_in_loops': 0, 'hazards': 0},
    'stem_fused_kernelILi7ELi8ELi8ELb0': {'mfma': 84, 'lds_dma': 6, 'global_load_x4': 33, 'global_store_x4': 10, 'barriers': 3, 'scratch': 6, 'vmcnt': [0], 'scratch_in_loops': 3, 'hazards': 0},
    'stem_fused_kernelILi7ELi8ELi8ELb1': {'mfma': 84, 'lds_dma': 0, 'global_load_x4': 33, 'global_store_x4': 10, 'barriers': 5, 'scratch': 8, 'vmcnt': [0], 'scratch_in_loops': 4, 'hazards': 0},
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


# ---- control-flow graph + in-order queue model -------------------------------------------
_REG = re.compile(r'\b([va])(?:(\d+)|\[(\d+):(\d+)\])(?![\w\[])')
_ADDR = re.compile(r'//\s*([0-9A-Fa-f]+):')
_TARGET = re.compile(r'<[^>]*\+0x([0-9a-fA-F]+)>\s*$')


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(2) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(1), r) for r in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


class Insn:
    __slots__ = ('text', 'op', 'addr', 'regs', 'dest', 'kind', 'succ', 'wait')

    def __init__(self, line, base):
        body = line.split('//')[0].strip()
        self.text = body
        self.op = body.split()[0] if body else ''
        m = _ADDR.search(line)
        self.addr = int(m.group(1), 16) - base if m else None
        operands = body[len(self.op):]
        self.regs = _regs(operands)
        first = operands.split(',')[0]
        self.dest = set()
        self.kind = None       # 'lds' / 'vm': an operation of that in-order queue
        self.wait = None       # (vmcnt or None, lgkmcnt or None)
        op = self.op
        if op.startswith('ds_'):
            self.kind = 'lds'
            if ('read' in op or 'permute' in op or 'swizzle' in op or '_rtn' in op or
                    'ds_load' in op):
                self.dest = _regs(first)
        elif re.match(r'(global|buffer|scratch|flat)_load', op) or (
                re.match(r'(global|buffer|flat)_atomic', op) and (' glc' in body or ' sc0' in body)):
            self.kind = 'vm'
            if not (re.search(r'\blds\b', operands) or '_lds_' in op):
                self.dest = _regs(first)
        elif op == 's_waitcnt':
            vm = re.search(r'vmcnt\((\d+)\)', body)
            lg = re.search(r'lgkmcnt\((\d+)\)', body)
            self.wait = (int(vm.group(1)) if vm else None, int(lg.group(1)) if lg else None)


def parse_kernel(lines):
    """disassembly lines of ONE kernel -> list of Insn with successor indices"""
    base = None
    for ln in lines:
        m = _ADDR.search(ln)
        if m:
            base = int(m.group(1), 16)
            break
    insns = [Insn(ln, base or 0) for ln in lines if ln.split('//')[0].strip()]
    by_addr = {i.addr: k for k, i in enumerate(insns) if i.addr is not None}
    for k, i in enumerate(insns):
        nxt = [k + 1] if k + 1 < len(insns) else []
        if i.op in ('s_endpgm', 's_endpgm_saved'):
            i.succ = []
        elif i.op == 's_branch' or i.op.startswith('s_cbranch'):
            line = lines_for(i, lines)
            m = _TARGET.search(line)
            tgt = by_addr.get(int(m.group(1), 16)) if m else None
            if tgt is None:
                raise RuntimeError(f'cannot resolve the target of: {i.text}')
            i.succ = [tgt] if i.op == 's_branch' else nxt + [tgt]
        elif i.op == 's_setpc_b64':
            # the compiler's long-branch expansion: s_getpc_b64 s[a:b]; s_add_u32 sa, sa, IMM;
            # s_addc_u32 sb, sb, 0 | -1; s_setpc_b64 s[a:b]  ->  (address after s_getpc) + IMM
            tgt = None
            if k >= 3 and insns[k - 3].op == 's_getpc_b64' and insns[k - 2].op == 's_add_u32':
                imm = int(insns[k - 2].text.split(',')[-1].strip(), 0)
                imm = imm - (1 << 32) if imm >= (1 << 31) else imm
                tgt = by_addr.get(insns[k - 3].addr + 4 + imm)
            if tgt is None:
                raise RuntimeError('indirect control flow in a hot kernel: ' + i.text)
            i.succ = [tgt]
        elif i.op == 's_swappc_b64':
            raise RuntimeError('call in a hot kernel: ' + i.text)
        else:
            i.succ = nxt
    return insns


_LINE_OF = {}


def lines_for(insn, lines):
    # (the branch target sits in the trailing comment, which Insn.text has dropped)
    key = id(lines)
    table = _LINE_OF.get(key)
    if table is None:
        table = {}
        for ln in lines:
            body = ln.split('//')[0].strip()
            m = _ADDR.search(ln)
            if body and m:
                table[(body, int(m.group(1), 16))] = ln
        _LINE_OF.clear()
        _LINE_OF[key] = table
        _LINE_OF['base', key] = min(a for (_, a) in table) if table else 0
    return table.get((insn.text, insn.addr + _LINE_OF['base', key]), '')


def pending_hazards(lines):
    """Every (reader, pending read) pair where an instruction mentions a VGPR / AGPR that
    an LDS read or VMEM load issued earlier on some path may still be about to write.

    Forward dataflow over the control-flow graph.  Per program point and queue: a map
    {pending read -> the SMALLEST number of same-queue operations issued behind it on any
    path that reaches this point}.  `s_waitcnt <q>cnt(N)` retires the entries with at least
    N operations behind them (in-order return); joins take the union with the minimum."""
    insns = parse_kernel(lines)
    if not insns:
        return []
    caps = {'lds': 15, 'vm': 63}   # a full counter stalls issue until the oldest returns
    n = len(insns)
    state = [None] * n             # entry state: ({pc: behind}, {pc: behind})
    state[0] = ({}, {})
    work = [0]
    found = {}
    while work:
        pc = work.pop()
        i = insns[pc]
        cur = {'lds': dict(state[pc][0]), 'vm': dict(state[pc][1])}
        for name, queue in cur.items():
            for q in queue:
                hit = insns[q].dest & i.regs
                if not hit:
                    continue
                # a later read of the SAME in-order queue may re-target the register
                if i.kind == name and hit <= i.dest and _only_dest_mentions(i, hit):
                    continue
                found.setdefault((pc, q), sorted(hit))
        if i.wait is not None:
            for name, cnt in (('vm', i.wait[0]), ('lds', i.wait[1])):
                if cnt is not None:
                    cur[name] = {q: b for q, b in cur[name].items() if b < cnt}
        elif i.kind is not None:
            queue = {q: b + 1 for q, b in cur[i.kind].items() if b + 1 < caps[i.kind]}
            if i.dest:
                queue[pc] = 0
            cur[i.kind] = queue
        for nx in i.succ:
            old = state[nx]
            if old is None:
                state[nx] = (dict(cur['lds']), dict(cur['vm']))
                work.append(nx)
                continue
            changed = False
            for k, name in enumerate(('lds', 'vm')):
                for q, b in cur[name].items():
                    if q not in old[k] or old[k][q] > b:
                        old[k][q] = b
                        changed = True
            if changed:
                work.append(nx)
    return [(insns[a].text, insns[b].text, regs) for (a, b), regs in sorted(found.items())]


def _only_dest_mentions(insn, regs):
    """True if `regs` appear in insn only as (part of) its first operand"""
    body = insn.text[len(insn.op):]
    rest = ','.join(body.split(',')[1:])
    return not (_regs(rest) & regs)


def scratch_in_loops(lines):
    """scratch_* instructions that sit on a cycle of the control-flow graph"""
    insns = parse_kernel(lines)
    n = len(insns)
    # Tarjan, iterative
    index = [None] * n
    low = [0] * n
    on = [False] * n
    stack, comp_of, counter, ncomp = [], [None] * n, [0], [0]
    for root in range(n):
        if index[root] is not None:
            continue
        work = [(root, 0)]
        while work:
            v, k = work.pop()
            if k == 0:
                index[v] = low[v] = counter[0]
                counter[0] += 1
                stack.append(v)
                on[v] = True
            recurse = False
            succ = insns[v].succ
            while k < len(succ):
                w = succ[k]
                k += 1
                if index[w] is None:
                    work.append((v, k))
                    work.append((w, 0))
                    recurse = True
                    break
                if on[w]:
                    low[v] = min(low[v], index[w])
            if recurse:
                continue
            if low[v] == index[v]:
                while True:
                    w = stack.pop()
                    on[w] = False
                    comp_of[w] = ncomp[0]
                    if w == v:
                        break
                ncomp[0] += 1
            if work:
                u = work[-1][0]
                low[u] = min(low[u], low[v])
    size = {}
    for c in comp_of:
        size[c] = size.get(c, 0) + 1
    looped = lambda k: size[comp_of[k]] > 1 or k in insns[k].succ
    return [insns[k].text for k in range(n) if insns[k].op.startswith('scratch_') and looped(k)]


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
        'scratch_in_loops': len(scratch_in_loops(lines)),
        'hazards': len(pending_hazards(lines)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--build-dir', default=str(ROOT / 'neuron-descriptions_amd' / 'csrc' / 'build'))
    ap.add_argument('--print', action='store_true', help='print the figures (to update PINNED)')
    args = ap.parse_args()
    build = pathlib.Path(args.build_dir)
    seen, bad, hazard_detail = {}, [], {}
    for obj in ('chain3.o', 'gemm.o', 'chain.o', 'conv3.o', 'stem.o'):
        path = build / obj
        if not path.exists():
            bad.append(f'{path} is missing: build the library first')
            continue
        for name, lines in kernels(disassemble(path)).items():
            for key in PINNED:
                if key in name:
                    seen[key] = figures(lines)
                    if seen[key]['hazards']:
                        hazard_detail[key] = pending_hazards(lines)
    for key, detail in sorted(hazard_detail.items()):
        for reader, read, regs in detail[:8]:
            bad.append(f'{key}: `{reader}` touches {regs[:4]} while `{read}` may still be in flight')
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
        print(f'check_isa: {len(PINNED)} kernels match their pinned VMEM / MFMA / barrier counts; '
              'no instruction touches the destination of a read that may still be in flight')
    return 0


if __name__ == '__main__':
    sys.exit(main())
