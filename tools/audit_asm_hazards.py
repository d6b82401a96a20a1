#!/usr/bin/env python3
"""Fails when a compiled MFMA kernel contains a packed-f32 VALU operation.

gfx950 hazard found in round 2 (tools/hwtests/pk_mfma_hazard.hip, profiles/r2_pk_mfma_hazard.txt;
DESIGN.md section 6): `v_pk_{mul,add}_f32 ... op_sel:[0,1]` directly followed by a `v_mfma` that has
to wait for the matrix pipe computes lanes 48..63 of its LOW result with a zero second operand.
hipcc (ROCm 7.2) neither avoids nor pads the sequence, and wait states only narrow the window,
so the fused kernels must not contain packed-f32 arithmetic at all: they are built with
-fno-slp-vectorize (the only source of v_pk_*_f32 in this code) and this audit keeps it that way.

Second check (round 4): the gfx940-family "trans forwarding" hazard -- a VALU instruction that reads the result of a
transcendental operation (v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos) needs one wait state behind it.
hipcc pads its own instructions but does not look inside inline asm: `v_cos_f32 v141, ..` directly followed by an
inline-asm `v_cvt_pk_f16_f32 .., v141` read stale values in the first lanes of the wave (the on-chip form of the
compensated kernel, first build: Y of lanes 0..15 off by 1 %).  The audit fails on any such adjacent pair.
usage: audit_asm_hazards.py kernel.s"""
import re
import sys

TRANS = re.compile(r"\s*v_(?:exp|log|rcp|rsq|sqrt|sin|cos)_(?:f32|f16|legacy_f32)\w*\s+v(\d+)")


def vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def trans_hazards(kernel_text):
    hits, prev = [], None
    for ln in kernel_text.split("\n"):
        code = ln.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        m = TRANS.match(code)
        if prev is not None and code.startswith("v_") and not m:
            srcs = set()
            for t in re.split(r"[,\s]+", code)[2:]:
                srcs |= vregs(t)
            if prev in srcs:
                hits.append(code)
        prev = int(m.group(1)) if m else None
    return hits

txt = open(sys.argv[1]).read()
bad = 0
for k in re.split(r'\n(?=_Z\w+:)', txt):
    m = re.match(r'(_Z\w+):', k)
    if not m or 'v_mfma' not in k:
        continue
    hits = [(i, ln.strip()) for i, ln in enumerate(k.split('\n')) if re.match(r'\s+v_pk_(mul|add|fma)_f32', ln)]
    n_mfma = len(re.findall(r'\n\s+v_mfma', k))
    print(f"{m.group(1)[:70]}: {n_mfma} MFMAs, {len(hits)} packed-f32 VALU operations")
    for i, t in hits[:5]:
        print(f"   line {i}: {t}")
    bad += len(hits)
    th = trans_hazards(k)
    for t in th[:5]:
        print(f"   transcendental result read by the next instruction: {t}")
    bad += len(th)
print("HAZARD AUDIT", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
