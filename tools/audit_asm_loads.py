#!/usr/bin/env python3
"""Audit of the hand-issued ring reads in the compiled eval16 kernels (guide 5.7): between
an inline-asm `ds_read_b128 vDST, vA offset:N*1024` and the MFMA that consumes vDST there must
be (a) an inline-asm s_waitcnt and (b) no other instruction touching vDST (a compiler copy or
spill of a register whose load has not landed would read garbage); and (c) no scalar memory load
may be issued while a hand-issued read is in flight: SMEM shares lgkmcnt with LDS and returns out
of order, so a counted lgkmcnt wait would no longer prove that the LDS read has landed.
usage: audit_asm_loads.py kernel.s"""
import re, sys
txt = open(sys.argv[1]).read()
kernels = re.split(r'\n(?=_ZN3pgd\d+eval16[sw]?_kernel)', txt)
bad = 0
for k in kernels:
    if not re.match(r'_ZN3pgd\d+eval16[sw]?_kernel', k):
        continue
    name = k.split(':', 1)[0]
    lines = k.split('\n')
    pending = {}          # first reg -> (line no, set(regs), waited?)
    n_loads = 0
    for i, ln in enumerate(lines):
        t = ln.strip()
        if not t or t.startswith(';') and 'ASM' not in t:
            continue
        m = re.match(r'ds_read_b128 v\[(\d+):(\d+)\], v\d+ offset:\d+\*1024', t)
        if m:
            regs = set(range(int(m.group(1)), int(m.group(2)) + 1))
            for key, (l0, r0, w0) in list(pending.items()):
                if r0 & regs:
                    print(f"{name}: line {i}: load overwrites still-pending {sorted(r0)} from line {l0}"); bad += 1
            pending[int(m.group(1))] = (i, regs, False)
            n_loads += 1
            continue
        if re.match(r's_(buffer_)?load_', t) and pending:
            print(f"{name}: line {i}: [{t}] scalar load while {len(pending)} hand-issued LDS reads are in flight"); bad += 1
        if t.startswith('s_waitcnt lgkmcnt'):
            for key in pending:
                l0, r0, _ = pending[key]
                pending[key] = (l0, r0, True)
            continue
        used = set()
        for a, b in re.findall(r'v\[(\d+):(\d+)\]', t):
            used |= set(range(int(a), int(b) + 1))
        used |= {int(x) for x in re.findall(r'\bv(\d+)\b', t)}
        for key, (l0, r0, waited) in list(pending.items()):
            if r0 & used:
                if t.startswith('v_mfma') and waited:
                    del pending[key]
                else:
                    print(f"{name}: line {i}: [{t}] touches {sorted(r0)} of the load at line {l0} (waited={waited})"); bad += 1
                    del pending[key]
    print(f"{name[:60]}: {n_loads} asm loads audited")
print("AUDIT", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
