#!/usr/bin/env python3
"""Live vector registers along a gfx950 kernel's assembly listing (hipcc -S --cuda-device-only).

    python tools/vgpr_live.py wf.s _Z15wf_frame_kernelILb0 [--top 12] [--marks] [--at LINE]

A backward liveness pass over the basic blocks of one kernel: for every instruction the number of VGPRs
(and AGPRs) that hold a value still to be read.  Prints the maximum, the lines where it is reached, and with
--marks the live count at every barrier / memory instruction, so that the phase of the kernel that binds
the register budget can be named (the compiler's own "VGPRs: 256, spill: n" says that it does not fit, not
where).  Approximations: a partial write (SDWA / op_sel destination halves) counts as a full definition; the
first operand of every instruction that is not a store / compare / branch is its destination."""
import re
import sys

REG = re.compile(r'\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]')
NO_DEST = ('global_store', 'buffer_store', 'scratch_store', 'ds_write', 'flat_store', 's_', 'v_cmp', 'v_cmpx',
           'global_atomic_add ', 'buffer_wbl2', 'buffer_inv', 'ds_gws', 'exp')
TWO_DEST = ('v_swap_b32',)
SGPR_DEST = ('v_readfirstlane_b32', 'v_readlane_b32')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
    return out


def split_operands(rest):
    ops, depth, cur = [], 0, ''
    for ch in rest:
        if ch == '[':
            depth += 1
        if ch == ']':
            depth -= 1
        if ch == ',' and depth == 0:
            ops.append(cur.strip()); cur = ''
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def parse(lines):
    insts = []            # (lineno, mnemonic, defs, uses, text)
    labels = {}
    for no, raw in lines:
        line = raw.split(';')[0].split('//')[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r'^([.\w$]+):', line)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        s = line.strip()
        if s.startswith('.'):
            continue
        parts = s.split(None, 1)
        mn = parts[0]
        rest = parts[1] if len(parts) > 1 else ''
        ops = split_operands(rest)
        defs, uses = set(), set()
        if mn.startswith(NO_DEST) or not ops:
            for o in ops:
                uses |= regs_of(o)
        elif mn.startswith(SGPR_DEST):
            for o in ops[1:]:
                uses |= regs_of(o)
        else:
            nd = 2 if mn.startswith(TWO_DEST) else 1
            for o in ops[:nd]:
                defs |= regs_of(o)
            for o in ops[nd:]:
                uses |= regs_of(o)
            if mn.startswith(TWO_DEST):
                uses |= defs
            # accumulating forms read their destination
            if mn.startswith(('v_fmac', 'v_mac', 'v_pk_fmac', 'v_dot2c', 'v_mfma')) or 'dpp' in mn or 'sdwa' in mn.lower() or 'row_' in rest or 'quad_perm' in rest or 'dst_sel' in rest:
                uses |= defs
        insts.append((no, mn, defs, uses, s))
    return insts, labels


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 8
    marks = '--marks' in sys.argv
    src = open(path).read().split('\n')
    start = next(i for i, l in enumerate(src) if l.startswith(kernel) and re.match(r'^[\w$.]+:', l))
    end = next(i for i in range(start, len(src)) if 's_endpgm' in src[i])
    lines = [(i + 1 - start, src[i]) for i in range(start + 1, end + 1)]
    insts, labels = parse(lines)
    n = len(insts)
    succ = [[] for _ in range(n)]
    for i, (no, mn, d, u, s) in enumerate(insts):
        if mn.startswith('s_branch'):
            tgt = s.split()[-1]
            if tgt in labels: succ[i].append(labels[tgt])
            continue
        if mn.startswith('s_cbranch'):
            tgt = s.split()[-1]
            if tgt in labels: succ[i].append(labels[tgt])
        if mn.startswith('s_endpgm'):
            continue
        if i + 1 < n:
            succ[i].append(i + 1)
    live_in = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            out = set()
            for j in succ[i]:
                if j < n:
                    out |= live_in[j]
            new = (out - insts[i][2]) | insts[i][3]
            if new != live_in[i]:
                live_in[i] = new; changed = True
    counts = [len(x) for x in live_in]
    mx = max(counts)
    print(f'{kernel}: {n} instructions, maximum live vector registers {mx}')
    order = sorted(range(n), key=lambda i: -counts[i])[:top]
    for i in sorted(order):
        print(f'  line {insts[i][0]:5d}  live {counts[i]:3d}  {insts[i][4][:90]}')
    if '--at' in sys.argv:                # the registers live at a line, each with the last instruction above that wrote it
        at = int(sys.argv[sys.argv.index('--at') + 1])
        i0 = next(i for i in range(n) if insts[i][0] >= at)
        print(f'live at line {insts[i0][0]} ({len(live_in[i0])}):')
        by_def = {}
        for r in sorted(live_in[i0]):
            d = next((j for j in range(i0 - 1, -1, -1) if r in insts[j][2]), None)
            by_def.setdefault(d, []).append(r)
        for d in sorted(by_def, key=lambda x: -1 if x is None else x):
            regs = ' '.join(f'{a}{b}' for a, b in by_def[d])
            print(f'  {"(entry)" if d is None else "line %5d %s" % (insts[d][0], insts[d][4][:70])}  <- {regs}')
    if marks:
        last = None
        for i, (no, mn, d, u, s) in enumerate(insts):
            if mn.startswith(('s_barrier', 'buffer_load', 'global_load', 'scratch_', 'global_atomic', 'global_store')) :
                key = mn
                if key != last:
                    print(f'  line {no:5d}  live {counts[i]:3d}  {mn}')
                last = key
            elif mn.startswith(('ds_read', 'ds_write')):
                key = mn[:7]
                if key != last:
                    print(f'  line {no:5d}  live {counts[i]:3d}  {mn}')
                last = key


if __name__ == '__main__':
    main()
