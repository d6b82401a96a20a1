"""CPU test of the host-side weight packer against a Python restatement of the kernels'
stream program (pg_program.h / pg_eval16.hip / pg_eval32.hip): the packed stream is
consumed chunk by chunk exactly as a wave does (enter a new chunk whenever the running
unit index of a segment hits a multiple of units-per-chunk), every unit is multiplied as
an MFMA A fragment with the lane-value sequences of pg_layout.h, and the result must be
the oracle's MLP output.  No GPU needed: `pg_debug_pack` is host-only.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import anerf_oracle as orc
from posegen_amd import PREC_FP16C
from posegen_amd import _ffi, synthetic as syn
from posegen_amd.config import PREC_BF16, PREC_FP16, PREC_FP32, RenderConfig, h36m_config, surreal_config
from posegen_amd.raycaster import NET_TENSOR_ORDER
from tests.helpers import oracle_cfg

J, JH, LV, LD, W, NT, VW, NTV = 24, 12, 7, 4, 256, 8, 128, 4
XSEQ, HSEQ, DSEQ_MAIN, DSEQ = 216, 128, 288, 328
BT_FEAT, BT_ALPHA, BT_VIEW, BT_RGB, BT_VIEWF = 64, 72, 73, 77, 78


def rho(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def xseq_channel(i, h):
    sb, w = divmod(i, 72)
    if w < 64:
        jj, q = 4 * sb + w // 16, w % 16
    else:
        jj, q = 4 * sb + (w - 64) // 2, 16 + (w - 64) % 2
    j = JH * h + jj
    return q * J + j if q < 15 else 360 + 3 * j + (q - 15)


def dseq_channel(i, h):
    if i < DSEQ_MAIN:
        blk, row = divmod(i, 8)
        return row * 72 + 3 * (JH * h + blk // 3) + blk % 3
    k = i - DSEQ_MAIN
    return -1 if k >= 36 else 8 * 72 + 3 * (JH * h + k // 3) + k % 3


def hseq_channel(i, h):
    return 32 * (i // 16) + rho(i % 16, h)


def pack(weights, cfg, prec, fact=False):
    lib = _ffi.load_library()
    arrs = [np.ascontiguousarray(weights[k], dtype=np.float32) for k in NET_TENSOR_ORDER]
    ptrs = (C.c_void_p * 24)(*[a.ctypes.data for a in arrs])
    shp = (C.c_int64 * 48)()
    for i, a in enumerate(arrs):
        shp[2 * i], shp[2 * i + 1] = a.shape[0], (a.shape[1] if a.ndim == 2 else 1)
    size, chunk = C.c_int64(), C.c_int32()
    rc = lib.pg_debug_pack(ptrs, shp, 24, cfg.framecode_ch, prec, int(fact), None, 0, C.byref(size), None, C.byref(chunk))
    assert rc == 0, lib.pg_last_error(None)
    buf = np.zeros(size.value, dtype=np.uint8)
    bias = np.zeros(82 * 32, dtype=np.float32)
    rc = lib.pg_debug_pack(ptrs, shp, 24, cfg.framecode_ch, prec, int(fact), buf.ctypes.data, size.value, C.byref(size),
                           bias.ctypes.data, C.byref(chunk))
    assert rc == 0, lib.pg_last_error(None)
    if (fact and fact not in (3, 4) and prec != PREC_FP16C) or fact == 2:      # Y-stage weights (the compensated kernel: record variant only)
        n = C.c_int64()
        rc = lib.pg_debug_pack_vy(ptrs, shp, 24, cfg.framecode_ch, prec, None, 0, C.byref(n))
        assert rc == 0, lib.pg_last_error(None)
        vy = np.zeros(n.value, dtype=np.uint8)
        rc = lib.pg_debug_pack_vy(ptrs, shp, 24, cfg.framecode_ch, prec, vy.ctypes.data, n.value, C.byref(n))
        assert rc == 0, lib.pg_last_error(None)
        return buf, bias, chunk.value, vy
    return buf, bias, chunk.value, None


class Wave:
    """A wave's view of the stream: chunks are entered strictly in order."""

    def __init__(self, stream, chunk_bytes, prec):
        self.s, self.cb, self.prec = stream, chunk_bytes, prec
        self.chunk = -1
        self.f32 = prec == PREC_FP32
        self.ue = 4 if self.f32 else 8
        self.upc = chunk_bytes // 1024

    def unit(self, L):
        """A fragment of unit L of the current segment -> [32 rows, 2 halves, ue] fp32."""
        if L % self.upc == 0:
            self.chunk += 1
        off = self.chunk * self.cb + (L % self.upc) * 1024
        raw = self.s[off:off + 1024].reshape(64, 16)
        if self.f32:
            v = raw.view(np.float32).reshape(64, 4)
        elif self.prec == PREC_BF16:
            v = (raw.view(np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(64, 8)
        else:
            v = raw.view(np.float16).astype(np.float32).reshape(64, 8)
        return v.reshape(2, 32, self.ue).transpose(1, 0, 2)      # lane = 32*h + row

    def mma(self, acc, L, o, bvals, u):
        """acc[o] += A(unit L) x B, B = values [2, n, 32pts] of sequence positions u*ue.."""
        a = self.unit(L)                                           # [row, h, e]
        b = bvals[:, u * self.ue:(u + 1) * self.ue, :]             # [h, e, pt]
        acc[o] += np.einsum("rhe,hep->rp", a, b)


def q16(x, prec):
    if prec == PREC_BF16:
        return torch.tensor(x).to(torch.bfloat16).to(torch.float32).numpy()
    if prec == PREC_FP16:
        return torch.tensor(x).to(torch.float16).to(torch.float32).numpy()
    return x


def vd_channel(j, k):
    if j < J:
        return (k % 9) * 72 + 3 * j + k // 9 if k < 27 else -1
    return 648 + k if k < 16 else -1


PERM16 = [1, 2, 16, 17, 0, 12, 4, 5, 18, 19, 3, 13, 7, 8, 20, 21, 6, 14, 10, 11, 22, 23, 9, 15]     # pg_layout.h: slot -> joint
PERMC = [1, 7, 2, 8, 16, 20, 17, 21, 0, 6, 12, 13, 4, 10, 5, 11, 18, 22, 19, 23, 3, 9, 15, 14]       # ... of the compensated record variant
XVC, XUC = 24, 30


def xseqc_channel(i, h):
    """pg_layout.h xseqc_channel: units 2 jj, 2 jj + 1 = the 15 cutoff-weighted values of joint slot jj of lane half h
    (+ a pad), units 24 + p = the directions of slots 2 p, 2 p + 1 (+ two pads)"""
    u, e = divmod(i, 8)
    if u < XVC:
        q = 8 * (u % 2) + e
        return q * J + PERMC[JH * h + u // 2] if q < 15 else -1
    if e >= 6:
        return -1
    return 360 + 3 * PERMC[JH * h + 2 * (u - XVC) + e // 3] + e % 3


def vy_joint(w, e, fc):
    """joint the Y-stage wave w handles as its e-th: the joint of SLOT 12 (w >> 2) + e (pg_layout.h vy_slot / slot16_joint)"""
    return PERM16[JH * (w >> 2) + e] if e < JH else (J if fc and (w >> 2) == 0 and e == JH else -1)


def vy_slot_joint(u, h, e, fc):
    if u == 0:
        return JH * h + e
    if e < 4:
        return JH * h + 8 + e
    return J if fc and h == 0 and e == 4 else -1


def y_stage(vy, tray, fc, prec):
    """Y[25, 128] of one ray (pg_pack.cpp pack_vy, pg_eval16.hip y_stage): wave w multiplies its
    units (unit n = joint n//2, k-unit n%2) as B operands (lane (hl, o): k = 16ku + 8hl + e)
    with the ray's 16-bit view values T16[j][k]."""
    ne = JH + (1 if fc else 0)
    assert vy.size == 8 * 2 * ne * 1024
    t16 = q16(tray.astype(np.float32), prec)
    y = np.zeros((J + 1, VW), dtype=np.float32)
    for w in range(8):
        for n in range(2 * ne):
            j, ku = vy_joint(w, n // 2, fc), n % 2
            off = (w * 2 * ne + n) * 1024
            unit = vy[off:off + 1024].reshape(64, 16)
            vals = ((unit.view(np.uint16).astype(np.uint32) << 16).view(np.float32) if prec == PREC_BF16
                    else unit.view(np.float16).astype(np.float32)).reshape(2, 32, 8)      # [hl, o, e]
            if j < 0:
                assert not vals.any()
                continue
            kk = t16[j, 16 * ku:16 * ku + 16].reshape(2, 8)
            y[j, 32 * (w & 3):32 * (w & 3) + 32] += np.einsum("he,hoe->o", kk, vals)
    return q16(y, prec)


def emulate(stream, bias, chunk_bytes, prec, x, cfg, fact=None):
    """x: [32 pts, 1080(+code16)] oracle input rows -> raw [32, 4] through the packed stream.
    fact = (tray [25,32], wpt [32,24], vy): the factorised view layer with x[:,432:1080] = wpt (x) tray."""
    wv = Wave(stream, chunk_bytes, prec)
    shape_a = prec in (PREC_BF16, PREC_FP16)
    ue = wv.ue

    def seq_vals(fn, n, src):
        v = np.zeros((2, n, 32), dtype=np.float32)
        for h in range(2):
            for i in range(n):
                ch = fn(i, h)
                if ch >= 0:
                    v[h, i] = src[:, ch]
        return q16(v, prec)

    def bias_tile(t):
        b = bias[t * 32:(t + 1) * 32].reshape(2, 16)
        out = np.zeros((32, 1), dtype=np.float32)
        for h in range(2):
            for r in range(16):
                out[rho(r, h), 0] = b[h, r]
        return np.repeat(out, 32, axis=1)

    def segment(no, kmajor, inputs, bias0, acc=None):
        """inputs: list of (values [2,n,32], n).  Returns acc tiles [no][32,32]."""
        if acc is None:
            acc = [bias_tile(bias0 + o) for o in range(no)]
        units = [(vals, u) for vals, n in inputs for u in range(n // ue)]
        nu = len(units)
        for L in range(nu * no):
            ui, o = (L // no, L % no) if kmajor else (L % nu, L // nu)
            wv.mma(acc, L, o, units[ui][0], units[ui][1])
        return acc

    def hidden_vals(tiles, relu=True):
        act = np.concatenate(tiles, 0)                              # [channels, pts]
        if relu:
            act = np.maximum(act, 0)
        return seq_vals(hseq_channel, len(tiles) * 16, act.T), act

    km = not shape_a
    xs = seq_vals(xseq_channel, XSEQ, x[:, :432])
    tiles = segment(NT, True, [(xs, XSEQ)], 0)
    hv, _ = hidden_vals(tiles)
    for l in range(1, 5):
        tiles = segment(NT, km, [(hv, HSEQ)], l * NT)
        hv, _ = hidden_vals(tiles)
    tiles = segment(NT, km, [(hv, HSEQ)], 5 * NT)
    tiles = segment(NT, True, [(xs, XSEQ)], 0, acc=tiles)
    hv, _ = hidden_vals(tiles)
    for l in (6, 7):
        tiles = segment(NT, km, [(hv, HSEQ)], l * NT)
        hv, _ = hidden_vals(tiles)
    if shape_a:
        # feature layer folded into the view weights: tiles [alpha | view x 4] on the trunk output
        av = segment(NTV + 1, False, [(hv, HSEQ)], BT_ALPHA,
                     acc=[bias_tile(BT_ALPHA)] + [bias_tile(BT_VIEWF + o) for o in range(NTV)])
        sigma = av[0][0]
        vt = av[1:]
    else:
        feat = segment(NT, True, [(hv, HSEQ)], BT_FEAT)
        sigma = segment(1, True, [(hv, HSEQ)], BT_ALPHA)[0][0]
        fv, _ = hidden_vals(feat, relu=False)
        vt = segment(NTV, km, [(fv, HSEQ)], BT_VIEW)
    if fact is not None:
        yq = y_stage(fact[2], fact[0], bool(cfg.framecode_ch), prec)
        wq = q16(fact[1].astype(np.float32), prec)                  # [pt, 24]
        for u in range(2):
            for h in range(2):
                for e in range(8):
                    j = vy_slot_joint(u, h, e, bool(cfg.framecode_ch))
                    if j < 0:
                        continue
                    wj = wq[:, j] if j < J else np.ones(32, dtype=np.float32)
                    for o in range(NTV):
                        vt[o] += np.outer(yq[j, 32 * o:32 * o + 32], wj)
    else:
        ins = [(seq_vals(dseq_channel, DSEQ, x[:, 432:1080]), DSEQ)]
        if cfg.framecode_ch:
            ins.append((seq_vals(lambda i, h: 8 * h + i, 8, x[:, 1080:1096]), 8))
        vt = segment(NTV, True, ins, 0, acc=vt)
    gv, _ = hidden_vals(vt)
    rgb = segment(1, km, [(gv, VW // 2)], BT_RGB)[0]
    return np.stack([rgb[0], rgb[1], rgb[2], sigma], -1), wv.chunk + 1


# ---- the 16x16x32 program of pg_eval16r.hip (pg_program.h R, pg_layout.h "small tile") ----
JG, NT16, NTV16, HU16, XU16, XV16 = 6, 16, 8, 8, 15, 12
BS_ALPHA, BS_VIEWF, BS_RGB = 128, 129, 137


def hseq16_channel(i, g):
    return 16 * (2 * (i // 8) + ((i % 8) >> 2)) + 4 * g + (i & 3)


def xseq16_channel(i, g):
    """pg_layout.h xseq16_channel: units 2 jj, 2 jj + 1 = the 15 cutoff-weighted values of joint slot jj (+ a pad),
    units 12 + p = the directions of slots 2 p, 2 p + 1 (+ two pads); slot 6 g + jj holds joint PERM16[6 g + jj]"""
    u, e = divmod(i, 8)
    if u < XV16:
        q = 8 * (u % 2) + e
        return q * J + PERM16[JG * g + u // 2] if q < 15 else -1
    if e >= 6:
        return -1
    return 360 + 3 * PERM16[JG * g + 2 * (u - XV16) + e // 3] + e % 3


def vy16_slot_joint(g, e, fc):
    return PERM16[JG * g + e] if e < JG else (J if fc and g == 0 and e == JG else -1)


def emulate_r(stream, bias16, chunk_bytes, prec, x, cfg, fact):
    """x: [32 pts, 1080(+code16)] -> raw [32, 4] through the R stream exactly as pg_eval16r.hip consumes it:
    units of 16 out rows x 32 k (lane (g, row): k = 32 u + 8 g + e), layer 0 and the skip layer's x part k-major
    over 16 out tiles, everything else out-tile-major, the rgb head's 4 units in the chunk the view tiles end in,
    the view-direction part from the per-ray Y record (fact = (tray [25,32], wpt [32,24], vy))."""
    upc = chunk_bytes // 1024
    state = {"chunk": -1}

    def unit(pos_in_segment, cont_base=None):
        if cont_base is None:
            if pos_in_segment % upc == 0:
                state["chunk"] += 1
            q = pos_in_segment % upc
        else:
            q = cont_base + pos_in_segment                  # continuation of the current chunk (rgb head)
            assert q < upc
        off = state["chunk"] * chunk_bytes + q * 1024
        raw = stream[off:off + 1024].reshape(64, 16)
        v = ((raw.view(np.uint16).astype(np.uint32) << 16).view(np.float32) if prec == PREC_BF16
             else raw.view(np.float16).astype(np.float32)).reshape(4, 16, 8)
        return v.transpose(1, 0, 2)                         # [row, g, e]

    def seq_vals(fn, n, src):
        v = np.zeros((4, n, 32), dtype=np.float32)
        for g in range(4):
            for i in range(n):
                ch = fn(i, g)
                if ch >= 0:
                    v[g, i] = src[:, ch]
        return q16(v, prec)

    def bias_tile(t):
        return np.repeat(bias16[t * 16:(t + 1) * 16].reshape(16, 1), 32, axis=1)       # row = 4 g + r

    def segment(no, kmajor, vals, nu, acc, cont_base=None):
        for L in range(nu * no):
            ui, o = (L // no, L % no) if kmajor else (L % nu, L // nu)
            acc[o] += np.einsum("rge,gep->rp", unit(L, cont_base), vals[:, ui * 8:(ui + 1) * 8, :])
        return acc

    def hidden_vals(tiles):
        act = np.maximum(np.concatenate(tiles, 0), 0)      # [channels, pts]
        return seq_vals(hseq16_channel, len(tiles) * 4, act.T)

    xs = seq_vals(xseq16_channel, XU16 * 8, x[:, :432])
    tiles = segment(NT16, True, xs, XU16, [bias_tile(o) for o in range(NT16)])
    y_onchip = None
    if fact[2] is None:     # on-chip variant: six limb chunks of direction weights behind layer 0, unit [g' (slot 6 g' + jj)][out tile t]
        t16 = q16(fact[0].astype(np.float32), prec)
        y_onchip = np.zeros((J + 1, VW), dtype=np.float32)
        for jj in range(JG):
            for gp in range(4):
                j = PERM16[JG * gp + jj]
                for t in range(NTV16):
                    a = unit(gp * NTV16 + t)                                # [row, g, e]: k = 8 g + e
                    y_onchip[j, 16 * t:16 * t + 16] = np.einsum("rge,ge->r", a, t16[j].reshape(4, 8))
        y_onchip = q16(y_onchip, prec)
    hv = hidden_vals(tiles)
    for l in range(1, 5):
        hv = hidden_vals(segment(NT16, False, hv, HU16, [bias_tile(l * NT16 + o) for o in range(NT16)]))
    tiles = segment(NT16, False, hv, HU16, [bias_tile(5 * NT16 + o) for o in range(NT16)])
    tiles = segment(NT16, True, xs, XU16, tiles)
    hv = hidden_vals(tiles)
    for l in (6, 7):
        hv = hidden_vals(segment(NT16, False, hv, HU16, [bias_tile(l * NT16 + o) for o in range(NT16)]))
    av = segment(NTV16 + 1, False, hv, HU16, [bias_tile(BS_ALPHA + o) for o in range(NTV16 + 1)])
    sigma, vt = av[0][0], av[1:]
    fc = bool(cfg.framecode_ch)
    yq = y_onchip if y_onchip is not None else y_stage(fact[2], fact[0], fc, prec)      # the ray's Y record [25, 128], 16-bit
    wq = q16(fact[1].astype(np.float32), prec)              # [pt, 24]
    for g in range(4):
        for e in range(8):
            j = vy16_slot_joint(g, e, fc)
            if j < 0:
                continue
            wj = wq[:, j] if j < J else np.ones(32, dtype=np.float32)
            for t in range(NTV16):
                vt[t] += np.outer(yq[j, 16 * t:16 * t + 16], wj)
    gv = hidden_vals(vt)
    rgb = segment(1, False, gv, VW // 32, [bias_tile(BS_RGB)], cont_base=(HU16 * (NTV16 + 1)) % upc)[0]
    return np.stack([rgb[0], rgb[1], rgb[2], sigma], -1), state["chunk"] + 1


def emulate_c(stream, bias, chunk_bytes, x, cfg, rec=None):
    """The compensated-fp16 program (pg_program.h C, pg_evalc.hip): every segment k-major, every
    (input unit, out tile) a PAIR of 1-KiB units -- plane 0 = (S-1) f16(W/S) against x1 = f16(x),
    plane 1 = f16(w1 + S (W/S - w1)) against x2 = f16(x1 + S (x - x1)) -- into one accumulator."""
    S = 129.0
    wv = Wave(stream, chunk_bytes, PREC_FP16)          # both planes are fp16 fragments

    def h16(v):
        return v.astype(np.float16).astype(np.float32)

    def pair_vals(fn, n, src):
        v = np.zeros((2, n, 32), dtype=np.float32)
        for h in range(2):
            for i in range(n):
                ch = fn(i, h)
                if ch >= 0:
                    v[h, i] = src[:, ch]
        x1 = h16(v)
        x2 = h16(x1 + np.float32(S) * (v - x1))
        return x1, x2

    def bias_tile(t):
        b = bias[t * 32:(t + 1) * 32].reshape(2, 16)
        out = np.zeros((32, 1), dtype=np.float32)
        for h in range(2):
            for r in range(16):
                out[rho(r, h), 0] = b[h, r]
        return np.repeat(out, 32, axis=1)

    def segment(no, inputs, acc):
        wv_units = [(p1, p2, u) for (p1, p2), n in inputs for u in range(n // 8)]
        for P in range(len(wv_units) * no):
            ui, o = P // no, P % no
            p1, p2, u = wv_units[ui]
            wv.mma(acc, 2 * P, o, p1, u)
            wv.mma(acc, 2 * P + 1, o, p2, u)
        return acc

    def hidden(tiles, relu=True):
        act = np.concatenate(tiles, 0)
        if relu:
            act = np.maximum(act, 0)
        return pair_vals(hseq_channel, len(tiles) * 16, act.T)

    tiles_of = lambda t0, n: [bias_tile(t0 + o) for o in range(n)]
    # record variant: the XC sequence (one chunk per joint pair, then the directions; joint slots PERMC) -- pg_layout.h
    xfn, xn = (xseqc_channel, XUC * 8) if rec is not None else (xseq_channel, XSEQ)
    xs = pair_vals(xfn, xn, x[:, :432])
    tiles = segment(NT, [(xs, xn)], tiles_of(0, NT))
    y_onchip = None
    if rec is not None and rec[2] is None:
        # on-chip form (pg_evalc.hip y_segment_c): twelve joint-pair chunks of direction weights behind layer 0, unit pairs
        # [k-unit u of 8 view values][out tile o], lane (h, col) = out channel 32 o + col, values 8 u + e of joint slot
        # 12 h + p; the ray's view values split like an activation
        t32 = rec[0].astype(np.float32)
        t1 = h16(t32)
        t2 = h16(t1 + np.float32(S) * (t32 - t1))
        y_onchip = np.zeros((J + 1, VW), dtype=np.float32)
        for pj in range(JH):
            for u in range(4):
                for o in range(NTV):
                    P = (pj * 4 + u) * NTV + o
                    a0, a1 = wv.unit(2 * P), wv.unit(2 * P + 1)     # [col, h, e]
                    for h in range(2):
                        j = PERMC[JH * h + pj]
                        y_onchip[j, 32 * o:32 * o + 32] += a0[:, h, :] @ t1[j, 8 * u:8 * u + 8] + a1[:, h, :] @ t2[j, 8 * u:8 * u + 8]
    for l in range(1, 5):
        tiles = segment(NT, [(hidden(tiles), HSEQ)], tiles_of(l * NT, NT))
    tiles = segment(NT, [(hidden(tiles), HSEQ)], tiles_of(5 * NT, NT))
    tiles = segment(NT, [(xs, xn)], tiles)
    for l in (6, 7):
        tiles = segment(NT, [(hidden(tiles), HSEQ)], tiles_of(l * NT, NT))
    av = segment(NTV + 1, [(hidden(tiles), HSEQ)], [bias_tile(BT_ALPHA)] + tiles_of(BT_VIEWF, NTV))
    sigma = av[0][0]
    if rec is None:
        ins = [(pair_vals(dseq_channel, DSEQ, x[:, 432:1080]), DSEQ)]
        if cfg.framecode_ch:
            ins.append((pair_vals(lambda i, h: 8 * h + i, 8, x[:, 1080:1096]), 8))
        vt = segment(NTV, ins, av[1:])
    else:
        # record variant (pg_rayrec.hip ray_records_c_kernel + the second stage of pg_evalc.hip): Y in fp32 from the
        # [joint][28][128] weights, split like a weight; the point's 24 cutoff weights (+ 1 for the frame code) split
        # like an activation; slots per vyc_slot_joint
        tray, wpt, vyc = rec
        fc = bool(cfg.framecode_ch)
        y = np.zeros((J + 1, VW), dtype=np.float32)
        if y_onchip is not None:
            y = y_onchip
        else:
            wy = vyc.view(np.float32).reshape(J + 1, 28, VW)
        for sl in range(J + (1 if fc else 0) if y_onchip is None else 0):       # weight blocks are in SLOT order (pack_vyc), y by joint
            j = PERMC[sl] if sl < J else J
            acc = np.zeros(VW, dtype=np.float32)
            for k in range(28):
                acc = np.float32(wy[sl, k] * np.float32(tray[j, k]) + acc) if k < 27 or j == J else acc
            y[j] = acc
        ys = (y * np.float32(1.0 / S)).astype(np.float32)
        y1 = h16(ys)
        p0 = h16(np.float32(S - 1) * y1)
        p1 = h16(y1 + np.float32(S) * (ys - y1))
        vt = av[1:]
        for u in range(2):
            a0 = np.zeros((VW, 2, 8), dtype=np.float32)            # [out, h, e]
            a1 = np.zeros((VW, 2, 8), dtype=np.float32)
            wv_ = np.zeros((2, 8, 32), dtype=np.float32)           # [h, e, pt]
            for h in range(2):
                for e in range(8):
                    sl = vy_slot_joint(u, h, e, fc)               # a SLOT of the record variant
                    if sl < 0:
                        continue
                    j = PERMC[sl] if sl < J else J
                    a0[:, h, e], a1[:, h, e] = p0[j], p1[j]
                    wv_[h, e] = wpt[:, j] if j < J else 1.0
            x1 = h16(wv_)
            x2 = h16(x1 + np.float32(S) * (wv_ - x1))
            for o in range(NTV):
                vt[o] += np.einsum("rhe,hep->rp", a0[32 * o:32 * o + 32], x1) + np.einsum("rhe,hep->rp", a1[32 * o:32 * o + 32], x2)
    rgb = segment(1, [(hidden(vt), VW // 2)], [bias_tile(BT_RGB)])[0]
    return np.stack([rgb[0], rgb[1], rgb[2], sigma], -1), wv.chunk + 1


@pytest.mark.parametrize("fc", [False, True])
@pytest.mark.parametrize("rec", [False, True, 3])
def test_packed_compensated_stream_reproduces_mlp(fc, rec):
    """The fp16c stream as the kernel consumes it (pairs of planes, k-major, every segment on a chunk
    boundary) against the fp32 oracle: the compensation itself is what is tested -- plain fp16 is at
    4e-3 on this input (test below), the pair must be 40x closer."""
    try:
        _ffi.load_library()
    except _ffi.HipLibraryError as e:
        pytest.skip(str(e))
    if rec == 3 and fc:
        pytest.skip("the on-chip form has no frame-code pseudo joint")
    cfg = h36m_config() if fc else surreal_config()
    w = syn.make_weights(cfg, 3)
    stream, bias, chunk_bytes, vyc = pack(w, cfg, PREC_FP16C, 3 if rec == 3 else 2 if rec else True)
    rng = np.random.RandomState(0)
    x = rng.uniform(-1, 1, size=(32, 1080)).astype(np.float32)
    x[:, :360] *= rng.uniform(0, 1, size=(32, 1)).astype(np.float32)
    rec_in = None
    if rec:       # one ray: view inputs = per-point joint weight x per-ray value (the record variant, S >= 64)
        tray = np.zeros((J + 1, 32), dtype=np.float32)
        tray[:J, :27] = rng.uniform(-1, 1, size=(J, 27))
        wpt = rng.uniform(0, 1, size=(32, J)).astype(np.float32)
        for j in range(J):
            for k in range(27):
                x[:, 432 + vd_channel(j, k)] = wpt[:, j] * tray[j, k]
        rec_in = (tray, wpt, vyc)
    ocfg = oracle_cfg(cfg, 79.6, 79.6)
    tw = {k: torch.tensor(v) for k, v in w.items()}
    if fc:
        idx = rng.randint(0, cfg.n_framecodes, size=(32, 1)).astype(np.float32)
        if rec:
            idx[:] = idx[0]                                            # one ray, one frame code
            rec_in[0][J, :16] = w["framecodes.codes.weight"][int(idx[0, 0])]
        ref = orc.mlp_forward(torch.tensor(np.concatenate([x, idx], 1)), tw, ocfg).numpy()
        x_em = np.concatenate([x, w["framecodes.codes.weight"][idx[:, 0].astype(int)]], 1)
    else:
        ref = orc.mlp_forward(torch.tensor(x), tw, ocfg).numpy()
        x_em = x
    raw, n_chunks = emulate_c(stream, bias, chunk_bytes, x_em, cfg, rec_in)
    assert n_chunks * chunk_bytes == stream.size, "kernel program and packer disagree on the chunk count"
    err = float(np.abs(raw - ref).max())
    print(f"fp16c stream emulation vs fp32 oracle: {err:.2e} (|ref| max {np.abs(ref).max():.2f})")
    assert err <= 1e-4 * max(1.0, float(np.abs(ref).max()) / 10)
    ocfg.quant = "fp16c"
    emu = orc.mlp_forward(torch.tensor(np.concatenate([x, idx], 1) if fc else x), tw, ocfg).numpy()
    assert float(np.abs(raw - emu).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()) / 10)


@pytest.mark.parametrize("prec,quant,tol", [(PREC_FP32, None, 2e-4), (PREC_BF16, "bf16", 2e-2), (PREC_FP16, "fp16", 4e-3)])
@pytest.mark.parametrize("fc", [False, True])
@pytest.mark.parametrize("fact", [False, True, 3])
def test_packed_stream_reproduces_mlp(prec, quant, tol, fc, fact):
    try:
        _ffi.load_library()
    except _ffi.HipLibraryError as e:
        pytest.skip(str(e))
    if fact and prec == PREC_FP32:
        pytest.skip("the fp32 kernel keeps the direct view layer")
    if fact == 3 and fc:
        pytest.skip("the on-chip variant of the 16x16x32 kernel has no frame codes")
    cfg = h36m_config() if fc else surreal_config()
    w = syn.make_weights(cfg, 3)
    stream, bias, chunk_bytes, vy = pack(w, cfg, prec, fact)
    rng = np.random.RandomState(0)
    x = rng.uniform(-1, 1, size=(32, 1080)).astype(np.float32)
    x[:, :360] *= rng.uniform(0, 1, size=(32, 1)).astype(np.float32)     # cutoff-weighted magnitudes
    fact_in = None
    if fact:      # one ray: view inputs = per-point joint weight x per-ray value
        tray = np.zeros((J + 1, 32), dtype=np.float32)
        tray[:J, :27] = rng.uniform(-1, 1, size=(J, 27))
        wpt = rng.uniform(0, 1, size=(32, J)).astype(np.float32)
        for j in range(J):
            for k in range(27):
                x[:, 432 + vd_channel(j, k)] = wpt[:, j] * tray[j, k]
        fact_in = (tray, wpt, vy)
    ocfg = oracle_cfg(cfg, 79.6, 79.6)
    ocfg.quant = quant
    tw = {k: torch.tensor(v) for k, v in w.items()}
    if fc:
        idx = rng.randint(0, cfg.n_framecodes, size=(32, 1)).astype(np.float32)
        if fact:
            idx[:] = idx[0]                                            # one ray, one frame code
            fact_in[0][J, :16] = w["framecodes.codes.weight"][int(idx[0, 0])]
        ref = orc.mlp_forward(torch.tensor(np.concatenate([x, idx], 1)), tw, ocfg).numpy()
        x_em = np.concatenate([x, w["framecodes.codes.weight"][idx[:, 0].astype(int)]], 1)
    else:
        ref = orc.mlp_forward(torch.tensor(x), tw, ocfg).numpy()
        x_em = x
    if fact:        # 16-bit precisions, rays with >= 64 samples: the 16x16x32 kernel with per-ray records
        raw, n_chunks = emulate_r(stream, bias, chunk_bytes, prec, x_em, cfg, fact_in)
    else:
        raw, n_chunks = emulate(stream, bias, chunk_bytes, prec, x_em, cfg, None)
    assert n_chunks * chunk_bytes == stream.size, "kernel program and packer disagree on the chunk count"
    np.testing.assert_allclose(raw, ref, rtol=0, atol=tol * max(1.0, float(np.abs(ref).max()) / 10))


# ---- the weight image of pg_evalc2.hip (pg_program.h T, pg_pack.cpp pack_c2): no stream; one 4-KiB block
# [tile t of the wave][plane] per (k-unit, wave), a B-fragment section for the view layer's direction part, two compact
# tables for the alpha row and the rgb rows ----
T_NW, T_FRAG, T_KBLK = 8, 1024, 4096
T_SEC_X, T_SEC_H, T_SEC_AV = XU16 * T_NW * T_KBLK, HU16 * T_NW * T_KBLK, HU16 * 4 * T_KBLK
T_NSLOT_Y = J + 1
T_SEC_Y = T_NSLOT_Y * NTV16 * 2 * T_FRAG
T_ALPHA_STRIDE, T_RGB_STRIDE = 5 * 16, 13 * 16
T_SMALL_ALPHA, T_SMALL_RGB = HU16 * 2 * T_ALPHA_STRIDE, (VW // 32) * 2 * T_RGB_STRIDE


def t_off_hid(hs):
    return T_SEC_X + hs * T_SEC_H + (T_SEC_X if hs >= 5 else 0)


T_OFF_X5 = T_SEC_X + 5 * T_SEC_H
T_OFF_AV = t_off_hid(7)
T_OFF_Y = T_OFF_AV + T_SEC_AV
T_OFF_SMALL = T_OFF_Y + T_SEC_Y
T_TOTAL = T_OFF_SMALL + T_SMALL_ALPHA + T_SMALL_RGB


def emulate_t(img, bias16, x, cfg, tray, wpt):
    """x: [pts, 432] density input, view input = wpt (x) tray (+ the frame code in tray[J]) -> raw [pts, 4] through the
    image as pg_evalc2.hip reads it: wave w's block of k-unit u = its two out tiles 2 w, 2 w + 1, two planes each
    (lane (g, row): k = 8 g + e of the unit); every product W x = plane0 . x1 + plane1 . x2 with the activation pair
    (x1, x2) = (f16(x), f16(x1 + 129 (x - x1))) the kernel forms (pg_comp.h) -- i.e. (S - 1) w1 x1 + w2 x2."""
    S = 129.0
    npt = x.shape[0]
    f16 = lambda a: a.astype(np.float16).astype(np.float32)

    def split(a):
        a = a.astype(np.float32)
        a1 = f16(a)
        return a1, f16(a1 + np.float32(S) * (a - a1))

    def frag(off):           # one A / B fragment -> [row 16, g 4, e 8]
        return img[off:off + T_FRAG].view(np.float16).astype(np.float64).reshape(4, 16, 8).transpose(1, 0, 2)

    def seq_pair(fn, nu, src):       # the B planes of a sequence: [plane][g, 8 nu, pt]
        v = np.zeros((4, nu * 8, npt), dtype=np.float32)
        for g in range(4):
            for i in range(nu * 8):
                ch = fn(i, g)
                if ch >= 0:
                    v[g, i] = src[:, ch]
        return split(v)

    def bias_tile(t):
        return np.repeat(bias16[t * 16:(t + 1) * 16].reshape(16, 1).astype(np.float64), npt, axis=1)

    def trunk(off, nu, pair, acc):           # acc: 16 out tiles of [16, pts]
        for u in range(nu):
            for w in range(T_NW):
                for tt in range(2):
                    f0 = off + ((u * T_NW + w) * 4 + tt * 2) * T_FRAG
                    for pl in range(2):
                        acc[2 * w + tt] += np.einsum("rge,gep->rp", frag(f0 + pl * T_FRAG), pair[pl][:, 8 * u:8 * u + 8, :].astype(np.float64))
        return acc

    def hidden_pair(tiles):
        act = np.maximum(np.concatenate(tiles, 0), 0).astype(np.float32)       # [channels, pts]
        return seq_pair(hseq16_channel, act.shape[0] // 32, act.T)

    xp = seq_pair(xseq16_channel, XU16, x)
    tiles = trunk(0, XU16, xp, [bias_tile(o) for o in range(NT16)])
    for hs in range(7):              # layers 1..7; layer 5 = the skip layer: hidden part + the density input again
        tiles = trunk(t_off_hid(hs), HU16, hidden_pair(tiles), [bias_tile((hs + 1) * NT16 + o) for o in range(NT16)])
        if hs == 4:
            tiles = trunk(T_OFF_X5, XU16, xp, tiles)
    hp = hidden_pair(tiles)
    # alpha from the compact table: entry g of [u][plane] = row 0, k = 8 g + e
    sigma = np.full(npt, float(bias16[BS_ALPHA * 16]))
    for u in range(HU16):
        for pl in range(2):
            tab = img[T_OFF_SMALL + (u * 2 + pl) * T_ALPHA_STRIDE:][:64].view(np.float16).astype(np.float64).reshape(4, 8)
            sigma += np.einsum("ge,gep->p", tab, hp[pl][:, 8 * u:8 * u + 8, :].astype(np.float64))
    # folded view layer, trunk part: tile pair v = tiles 2 v, 2 v + 1 of k-unit u
    vt = [bias_tile(BS_VIEWF + o) for o in range(NTV16)]
    for u in range(HU16):
        for v in range(4):
            for tt in range(2):
                f0 = T_OFF_AV + ((u * 4 + v) * 4 + tt * 2) * T_FRAG
                for pl in range(2):
                    vt[2 * v + tt] += np.einsum("rge,gep->rp", frag(f0 + pl * T_FRAG), hp[pl][:, 8 * u:8 * u + 8, :].astype(np.float64))
    # direction part: Y[slot][out] = W_vd[:, slot block] . T[slot] (the ray's 27 view values / the 16 code values, split like an
    # activation), then sum_slots w_slot(point) Y[slot] (w = 1 for the code)
    fc = bool(cfg.framecode_ch)
    for s in range(J + (1 if fc else 0)):
        j = PERM16[s] if s < J else J
        t1, t2 = split(tray[j].reshape(4, 8))
        for tt in range(NTV16):
            f0 = T_OFF_Y + ((s * NTV16 + tt) * 2) * T_FRAG
            y = np.einsum("rge,ge->r", frag(f0), t1.astype(np.float64)) + np.einsum("rge,ge->r", frag(f0 + T_FRAG), t2.astype(np.float64))
            wj = wpt[:, j].astype(np.float64) if j < J else np.ones(npt)
            vt[tt] += np.outer(y, wj)
    gp = hidden_pair(vt)
    rgb = np.repeat(bias16[BS_RGB * 16:BS_RGB * 16 + 3].reshape(3, 1).astype(np.float64), npt, axis=1)
    for u in range(VW // 32):
        for pl in range(2):
            tab = img[T_OFF_SMALL + T_SMALL_ALPHA + (u * 2 + pl) * T_RGB_STRIDE:][:12 * 16].view(np.float16).astype(np.float64).reshape(4, 3, 8)
            rgb += np.einsum("gre,gep->rp", tab, gp[pl][:, 8 * u:8 * u + 8, :].astype(np.float64))
    return np.stack([rgb[0], rgb[1], rgb[2], sigma], -1)


@pytest.mark.parametrize("fc", [False, True])
def test_tile_split_weight_image_reproduces_mlp(fc):
    """pack_c2 (the weights of pg_evalc2.hip: out tiles split over the waves, no stream) read back as the kernel reads
    them, with the kernel's compensated products, against the fp32 oracle on the same points: <= 1e-4 (plain fp16
    is at 4e-3 on this input) -- layout, plane pairing, the skip layer's two sections, the compact alpha / rgb tables,
    the direction section in slot order and the frame code's pseudo slot."""
    try:
        _ffi.load_library()
    except _ffi.HipLibraryError as e:
        pytest.skip(str(e))
    cfg = h36m_config() if fc else surreal_config()
    w = syn.make_weights(cfg, 3)
    img, bias16, _, _ = pack(w, cfg, PREC_FP16C, 4)
    assert img.size == T_TOTAL
    rng = np.random.RandomState(0)
    npt = 16
    x = rng.uniform(-1, 1, size=(npt, 1080)).astype(np.float32)
    x[:, :360] *= rng.uniform(0, 1, size=(npt, 1)).astype(np.float32)
    tray = np.zeros((J + 1, 32), dtype=np.float32)
    tray[:J, :27] = rng.uniform(-1, 1, size=(J, 27))
    wpt = rng.uniform(0, 1, size=(npt, J)).astype(np.float32)
    for j in range(J):
        for k in range(27):
            x[:, 432 + vd_channel(j, k)] = wpt[:, j] * tray[j, k]
    ocfg = oracle_cfg(cfg, 79.6, 79.6)
    tw = {k: torch.tensor(v) for k, v in w.items()}
    if fc:
        idx = np.full((npt, 1), 5, dtype=np.float32)                   # one ray, one frame code
        tray[J, :16] = w["framecodes.codes.weight"][5]
        ref = orc.mlp_forward(torch.tensor(np.concatenate([x, idx], 1)), tw, ocfg).numpy()
    else:
        ref = orc.mlp_forward(torch.tensor(x), tw, ocfg).numpy()
    raw = emulate_t(img, bias16, x[:, :432], cfg, tray, wpt)
    err = float(np.abs(raw - ref).max())
    print(f"pg_evalc2 weight image emulation vs fp32 oracle: {err:.2e} (|ref| max {np.abs(ref).max():.2f})")
    assert err <= 1e-4 * max(1.0, float(np.abs(ref).max()) / 10)


# ---- source maps (pg_load_weights_device re-forms a packed image by a gather from the flat parameter vector) ----
@pytest.mark.parametrize("fc", [False, True])
@pytest.mark.parametrize("form", [0, 1, 2])
def test_source_maps_reproduce_the_packed_images(form, fc):
    """pg_debug_pack_map: gathering the flat source vector through the recorded map, with the conversion each entry names
    (plain bf16 / fp16, plane 0 / 1 of the compensated pair, fp32 for the bias table), reproduces pg_debug_pack's image byte
    for byte -- the host statement of what pg_repack.hip does on the device (the GPU test compares renders bitwise)."""
    try:
        lib = _ffi.load_library()
    except _ffi.HipLibraryError as e:
        pytest.skip(str(e))
    cfg = h36m_config() if fc else surreal_config()
    w = syn.make_weights(cfg, 5)
    arrs = [np.ascontiguousarray(w[k], dtype=np.float32) for k in NET_TENSOR_ORDER]
    ptrs = (C.c_void_p * 24)(*[a.ctypes.data for a in arrs])
    shp = (C.c_int64 * 48)()
    for i, a in enumerate(arrs):
        shp[2 * i], shp[2 * i + 1] = a.shape[0], (a.shape[1] if a.ndim == 2 else 1)
    nm, nsrc = C.c_int64(), C.c_int64()
    assert lib.pg_debug_pack_map(ptrs, shp, 24, cfg.framecode_ch, form, None, 0, C.byref(nm), None, 0, C.byref(nsrc)) == 0, lib.pg_last_error(None)
    mp, src = np.zeros(nm.value, dtype=np.int32), np.zeros(nsrc.value, dtype=np.float32)
    assert lib.pg_debug_pack_map(ptrs, shp, 24, cfg.framecode_ch, form, mp.ctypes.data, nm.value, C.byref(nm), src.ctypes.data, nsrc.value, C.byref(nsrc)) == 0
    valid = mp >= 0
    vals = np.where(valid, src[np.where(valid, mp >> 2, 0)], 0.0).astype(np.float32)
    kind = np.where(valid, mp & 3, 0)

    def bf16_bits(x):
        u = x.view(np.uint32).astype(np.uint64)
        return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)

    if form == 2:
        _, bias16, _, _ = pack(w, cfg, PREC_BF16, 3)
        assert np.array_equal(vals.view(np.uint32), bias16[:vals.size].view(np.uint32))
        assert int(valid.sum()) == 8 * 256 + 1 + 128 + 3
        return
    if form == 0:
        for prec in (PREC_BF16, PREC_FP16):
            img, _, _, _ = pack(w, cfg, prec, 3)
            got = bf16_bits(vals) if prec == PREC_BF16 else vals.astype(np.float16).view(np.uint16)
            assert (kind == 0).all() and np.array_equal(got, img.view(np.uint16)), prec
        return
    img, _, _, _ = pack(w, cfg, PREC_FP16C, 4)
    wd = vals.astype(np.float64) / 129.0
    w1 = wd.astype(np.float32).astype(np.float16).astype(np.float64)
    p0 = (128.0 * w1).astype(np.float32).astype(np.float16).view(np.uint16)
    p1 = (w1 + 129.0 * (wd - w1)).astype(np.float32).astype(np.float16).view(np.uint16)
    got = np.where(kind == 1, p0, np.where(kind == 2, p1, 0)).astype(np.uint16)
    got[~valid] = 0
    assert set(np.unique(kind[valid])) == {1, 2} and np.array_equal(got, img.view(np.uint16))
