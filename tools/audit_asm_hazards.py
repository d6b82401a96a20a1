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

Third check (round 5): "VALU writes an SGPR -> a vector-memory instruction reads it" needs five wait states on gfx9-family
parts.  hipcc pads its own memory instructions, not those inside inline asm: a `v_readlane_b32 s9, ..` (the reload of a spilled
SGPR) directly in front of an inline-asm `global_load_dwordx4 .., s[8:9]` made the first build of pg_evalc2.hip fault on
an address with a stale upper half.  The audit fails when an inline-asm vector-memory instruction reads an SGPR that a
VALU instruction (v_readlane / v_readfirstlane / a compare or carry writing an SGPR) wrote fewer than five wait states before.

Fourth check (round 5): "XDL (MFMA) writes a VGPR -> a VALU instruction reads it" needs (passes + 2 + 1) wait states on gfx950:
7 behind a 4-pass MFMA (16x16x32), 11 behind an 8-pass one (32x32x16).  hipcc pads its own VALU instructions; an inline-asm
conversion block scheduled 5 instructions behind the last MFMA of its accumulator read it stale (pg_evalc2.hip, the first
build with conversions between the MFMA blocks: one column tile off by a few percent).  The audit fails when an inline-asm
VALU instruction reads a register an MFMA wrote fewer than 8 / 12 wait states before.
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


# opcodes whose destination is a source as well (accumulate forms): the destination token counts as read
ACCUM = re.compile(r"v_(?:pk_)?(?:fmac|mac|fmaak|madak)_|v_dot\d|v_mfma_|v_smfmac_|v_(?:cndmask|mov)\w*_(?:sdwa|dpp)|v_\w+_(?:sdwa|dpp)\b")


def trans_hazards(kernel_text):
    """non-transcendental VALU instructions that read the result of the transcendental right in front of them (a transcendental
    reader runs in the same quarter-rate unit, in order: LLVM's hasTransForwardingHazard applies to non-TRANS readers only).
    Accumulate opcodes (v_fmac / v_mac / v_pk_fmac / v_dot*, MFMA forms whose vdst is tied to the accumulator) and the
    SDWA / DPP forms (a partial write preserves, i.e. reads, the destination) read their destination register too."""
    hits, prev = [], None
    for ln in kernel_text.split("\n"):
        code = ln.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        m = TRANS.match(code)
        if prev is not None and code.startswith("v_") and not m:
            toks = re.split(r"[,\s]+", code)
            srcs = set()
            for t in toks[2:]:
                srcs |= vregs(t)
            if ACCUM.match(toks[0]) and len(toks) > 1:
                srcs |= vregs(toks[1])
            if prev in srcs:
                hits.append(code)
        prev = int(m.group(1)) if m else None
    return hits


def sregs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


VMEM = re.compile(r"\s*(global_|buffer_|flat_|scratch_)")


def sgpr_vmem_hazards(kernel_text):
    """inline-asm vector-memory instructions that read an SGPR written by a VALU instruction < 5 wait states earlier"""
    hits = []
    recent = []                 # (wait states ago, set of SGPRs written by a VALU instruction)
    in_asm = False
    for ln in kernel_text.split("\n"):
        if "#ASMSTART" in ln:
            in_asm = True
            continue
        if "#ASMEND" in ln:
            in_asm = False
            continue
        code = ln.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        toks = [t for t in re.split(r"[,\s]+", code) if t]
        if in_asm and VMEM.match(code):
            srcs = set()
            for t in toks[1:]:
                srcs |= sregs(t)
            for age, regs in recent:
                if age < 5 and regs & srcs:
                    hits.append(code)
                    break
        states = 1
        m = re.match(r"s_nop\s+(\d+)", code)
        if m:
            states = int(m.group(1)) + 1
        recent = [(a + states, r) for a, r in recent if a + states < 5]
        if code.startswith("v_"):
            written = set()
            if toks[0].startswith(("v_readlane", "v_readfirstlane")):
                written = sregs(toks[1])
            elif len(toks) > 1:
                written = sregs(toks[1]) | (sregs(toks[2]) if toks[0].startswith(("v_add_co", "v_sub_co", "v_addc", "v_subb", "v_div_scale", "v_mad_u64", "v_mad_i64")) and len(toks) > 2 else set())
            if written:
                recent.append((0, written))
    return hits



def mfma_valu_hazards(kernel_text):
    """inline-asm VALU instructions that read (or overwrite) a VGPR an MFMA wrote too few wait states earlier.
    Timing model: every instruction takes one issue slot (s_nop N: N + 1); an MFMA that follows another within its
    passes waits for the matrix pipe (4 slots behind a 16x16x32, 8 behind a 32x32x16), a VALU instruction does not."""
    hits = []
    ready = {}                  # vgpr -> first slot at which a VALU instruction may touch it
    t, pipe_free = 0, 0
    in_asm = False
    for ln in kernel_text.split("\n"):
        if "#ASMSTART" in ln:
            in_asm = True
            continue
        if "#ASMEND" in ln:
            in_asm = False
            continue
        code = ln.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        toks = [t_ for t_ in re.split(r"[,\s]+", code) if t_]
        m = re.match(r"s_nop\s+(\d+)", code)
        if toks[0].startswith("v_mfma"):
            t = max(t + 1, pipe_free)
            passes = 8 if "32x32" in toks[0] else 4
            pipe_free = t + passes
            for r in vregs(toks[1]):
                ready[r] = t + passes + 3 + 1         # passes + 2 (+ 1 on gfx950) wait states, and one of margin
            continue
        t += int(m.group(1)) + 1 if m else 1
        if code.startswith("v_"):
            touched = set()
            for t_ in toks[1:]:
                touched |= vregs(t_)
            if in_asm and any(ready.get(r, 0) > t for r in touched):
                hits.append(code)
            for r in vregs(toks[1]):
                ready.pop(r, None)
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
    sh = sgpr_vmem_hazards(k)
    for t in sh[:5]:
        print(f"   inline-asm memory instruction reads an SGPR a VALU instruction has just written: {t}")
    bad += len(sh)
    mh = mfma_valu_hazards(k)
    for t in mh[:5]:
        print(f"   inline-asm VALU instruction reads an MFMA result too early: {t}")
    bad += len(mh)
print("HAZARD AUDIT", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
