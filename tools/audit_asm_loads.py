#!/usr/bin/env python3
"""Audit of the hand-issued ring reads in the compiled fused kernels (guide 5.7).

The kernels issue their weight-ring reads by inline asm (`ds_read_b128 vDST, vA offset:N*1024`)
and retire them with COUNTED `s_waitcnt lgkmcnt(N)`; hipcc sees neither.  For every such read
the compiled code must satisfy, between its issue and the MFMA that consumes vDST:
  (a) a wait that really covers it: LDS operations return in issue order, so `lgkmcnt(N)`
      retires every LGKM operation except the N youngest -- the audit keeps the in-order queue
      of ALL LDS operations of the wave (hand-issued and compiler-issued, reads and writes)
      and fails when the consuming MFMA finds its read still among the unretired ones;
  (b) no other instruction touches vDST (a compiler copy or spill of a register whose load has
      not landed would read garbage);
  (c) no scalar memory load is issued while a hand-issued read is in flight: SMEM shares
      lgkmcnt with LDS and returns out of order, so a counted wait would prove nothing.
Round 5 (pg_evalc2.hip): the same for hand-issued `global_load_dwordx4` (weights straight from L2 into registers) and
their counted `s_waitcnt vmcnt(N)`: vector-memory operations (loads, stores, LDS-DMA) retire in issue order too.  Any
`ds_read_b128` / `global_load_dwordx4` between `;;#ASMSTART` and `;;#ASMEND` counts as hand-issued.
usage: audit_asm_loads.py kernel.s [kernel-name-regex]"""
import re
import sys

KERNEL_RE = sys.argv[2] if len(sys.argv) > 2 else r'_ZN3pgd\w*eval\w*_kernel'
txt = open(sys.argv[1]).read()
kernels = re.split(r'\n(?=' + KERNEL_RE + ')', txt)
# ring reads carry `offset:N*1024`, the reads beside the ring pipe (lds_async128: bias tiles, (a, b) rows) `offset:0+N`
HAND = re.compile(r'ds_read_b128 v\[(\d+):(\d+)\], v\d+ offset:(?:\d+\*1024|0\+\d+(?:\*64)?)')
bad = 0


def vregs(t):
    used = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', t):
        used |= set(range(int(a), int(b) + 1))
    used |= {int(x) for x in re.findall(r'(?<![\w\[])v(\d+)\b', t)}
    return used


for k in kernels:
    if not re.match(KERNEL_RE, k):
        continue
    name = k.split(':', 1)[0]
    queue = []            # in-order LGKM operations in flight: None (compiler's) or dict (hand-issued)
    vqueue = []           # in-order vector-memory operations in flight, likewise
    n_loads = 0
    in_asm = False
    for i, ln in enumerate(k.split('\n')):
        if '#ASMSTART' in ln:
            in_asm = True
        elif '#ASMEND' in ln:
            in_asm = False
        t = ln.split(';')[0].strip()
        if not t or t.endswith(':') or t.startswith('.'):
            continue
        pending = [q for q in queue if q is not None]
        vpending = [q for q in vqueue if q is not None]
        m = HAND.match(t) or (in_asm and re.match(r'ds_read_b128 v\[(\d+):(\d+)\]', t))
        if m:
            regs = set(range(int(m.group(1)), int(m.group(2)) + 1))
            for q in pending + vpending:
                if q['regs'] & regs:
                    print(f"{name}: line {i}: load overwrites still-pending {sorted(q['regs'])} from line {q['line']}")
                    bad += 1
            queue.append({'line': i, 'regs': regs})
            n_loads += 1
            continue
        m = in_asm and re.match(r'global_load_dwordx4 v\[(\d+):(\d+)\]', t)
        if m:
            regs = set(range(int(m.group(1)), int(m.group(2)) + 1))
            for q in pending + vpending:
                if q['regs'] & regs:
                    print(f"{name}: line {i}: load overwrites still-pending {sorted(q['regs'])} from line {q['line']}")
                    bad += 1
            vqueue.append({'line': i, 'regs': regs})
            n_loads += 1
            continue
        if re.match(r's_(buffer_)?load_', t):
            if pending:
                print(f"{name}: line {i}: [{t}] scalar load while {len(pending)} hand-issued LDS reads are in flight")
                bad += 1
            queue.append(None)
            continue
        if t.startswith('ds_'):                       # compiler-issued LDS operation
            queue.append(None)
            # fall through: it may also touch a pending register (checked below)
        if re.match(r'(global_|buffer_|flat_|scratch_)', t):      # compiler-issued vector-memory operation (or LDS-DMA)
            vqueue.append(None)
        if t.startswith('s_waitcnt'):
            m = re.search(r'lgkmcnt\((\d+)\)', t)
            if m:
                keep = int(m.group(1))
                queue = queue[len(queue) - keep:] if keep else []
            m = re.search(r'vmcnt\((\d+)\)', t)
            if m:
                keep = int(m.group(1))
                vqueue = vqueue[len(vqueue) - keep:] if keep else []
            continue
        if t.startswith('s_barrier') or t.startswith('s_endpgm'):
            continue
        used = vregs(t)
        for q in pending:
            if q['regs'] & used:
                print(f"{name}: line {i}: [{t}] touches {sorted(q['regs'])} of the load at line {q['line']}, "
                      f"which no s_waitcnt has retired ({len(queue) - queue.index(q) - 1} younger LGKM operations)")
                bad += 1
                queue.remove(q)
        for q in vpending:
            if q['regs'] & used:
                print(f"{name}: line {i}: [{t}] touches {sorted(q['regs'])} of the global load at line {q['line']}, "
                      f"which no s_waitcnt has retired ({len(vqueue) - vqueue.index(q) - 1} younger vector-memory operations)")
                bad += 1
                vqueue.remove(q)
    print(f"{name[:70]}: {n_loads} asm loads audited")
print("AUDIT", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
