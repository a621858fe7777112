#!/usr/bin/env python3
"""Static check of one kernel in a `hipcc -S` dump, per basic block, for two things the back end must get right and that a
scheduling-dependent miscompile could break: (1) a VGPR that is the destination of an LDS / global / scratch load is not read
(or overwritten) before an s_waitcnt has retired that load; (2) a DPP instruction does not read a VGPR written by a VALU
instruction less than two instructions earlier (GFX9 DPP hazard, s_nop counted).  Pending loads at a block's entry are unknown
(ignored), so this can miss cross-block cases; it cannot produce false alarms inside a block except through its own parsing.
Usage: tools/isa_wait_check.py file.s kernel-name-pattern"""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + pat + r'\S*:', l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))

def regs(tok):
    """v5 / v[4:7] / a3 -> list of register names"""
    out = []
    for m in re.finditer(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]', tok):
        if m.group(1):
            out.append(m.group(1) + m.group(2))
        else:
            out += [m.group(3) + str(k) for k in range(int(m.group(4)), int(m.group(5)) + 1)]
    return out

problems = 0
lgkm, vm = [], []          # FIFOs of (dest regs, text) for in-order counters; scalar loads make lgkm out-of-order
smem_pending = False
valu_hist = []             # (instruction index in block, written vgprs)
n = 0
for i in range(start + 1, end):
    l = lines[i]
    if re.match(r'^\.LBB\d+_\d+:', l):   # a branch target: pending state unknown; fall-through-only blocks (; %bb.N) keep it
        lgkm, vm, smem_pending, valu_hist, n = [], [], False, [], 0
        continue
    if not l.startswith('\t') or l.startswith('\t.') or l.strip().startswith(';'):
        continue
    ins = re.sub(r';.*', '', l).strip()
    if not ins:
        continue
    op, _, rest = ins.partition(' ')
    ops = [o.strip() for o in rest.split(',')] if rest else []
    n += 1
    if op == 's_waitcnt':
        m = re.search(r'lgkmcnt\((\d+)\)', rest)
        if m:
            k = int(m.group(1))
            if k == 0:
                lgkm, smem_pending = [], False
            elif not smem_pending:
                lgkm = lgkm[len(lgkm) - k:] if k < len(lgkm) else lgkm
        m = re.search(r'vmcnt\((\d+)\)', rest)
        if m:
            k = int(m.group(1))
            vm = vm[len(vm) - k:] if k < len(vm) else ([] if k == 0 else vm)
        if re.fullmatch(r'\d+', rest.strip() or 'x'):   # raw immediate: treat as wait-all
            lgkm, vm, smem_pending = [], [], False
        continue
    if op.startswith('s_nop'):
        n += int(rest.strip() or 0)
        continue
    if op == 's_barrier' or op.startswith('s_cbranch') or op == 's_branch' or op.startswith('s_endpgm'):
        continue
    # uses / defs of vector registers
    is_load = op.startswith(('ds_read', 'ds_bpermute', 'ds_permute', 'global_load', 'scratch_load', 'buffer_load', 'flat_load')) or ('_rtn' in op)
    dst = regs(ops[0]) if ops and (is_load or op.startswith('v_')) and not op.startswith(('v_cmp', 'v_writelane')) else []
    if op.startswith('v_writelane') and ops:
        dst = regs(ops[0])
    srcs = []
    for o in (ops[1:] if dst else ops):
        srcs += regs(o)
    pending = {r: t for rs, t in lgkm + vm for r in rs}
    for r in srcs + ([] if is_load else dst):
        if r in pending:
            problems += 1
            print(f'line {i + 1}: `{ins}` touches {r} while `{pending[r]}` may still be in flight')
            break
    if 'dpp' in ins or 'row_' in ins or 'quad_perm' in ins:
        for k, w in valu_hist[-3:]:
            if n - k < 3 and any(r in w for r in srcs):
                problems += 1
                print(f'line {i + 1}: DPP `{ins}` reads a VGPR written {n - k} instruction(s) earlier')
    if is_load:
        (lgkm if op.startswith('ds_') else vm).append((dst, ins))
    elif op.startswith('ds_'):
        lgkm.append(([], ins))          # LDS stores / atomics count in lgkmcnt too (in order with the loads)
    elif op.startswith(('global_store', 'global_atomic', 'scratch_store', 'buffer_store', 'flat_store')):
        vm.append(([], ins))            # gfx9: stores count in vmcnt
    elif op.startswith('s_load') or op.startswith('s_buffer_load'):
        smem_pending = True
    elif op.startswith('v_') and dst:
        valu_hist.append((n, dst))
print(f'{pat}: {problems} problem(s)')
