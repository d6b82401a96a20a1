// pg_eval16_common.h -- building blocks shared by the 16-bit fused embed+MLP kernels
// (pg_eval16.hip: v_mfma_f32_32x32x16, direct view layer; pg_eval16r.hip: v_mfma_f32_16x16x32,
// per-ray records): MFMA operand types, the hand-pipelined weight-ring reads, LDS reads beside
// that pipe, the ReLU / 16-bit packing and the embedding math.
#pragma once
#include "pg_device.h"

namespace pgd {
using namespace pgp::A;

#ifndef PG_A_WAVES
#define PG_A_WAVES 8
#endif
constexpr int NWAVE = PG_A_WAVES;          // 8: one workgroup per CU; 4: two independent workgroups per CU
constexpr int NTHR = NWAVE * 64;
constexpr int PTS = NWAVE * 32;     // points per workgroup pass

template <typename V> struct Op;
template <> struct Op<bf16x8> {
    using E = __bf16;
    static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
#if defined(PG_MFMA_TURNS)
        // experiment: the two waves of a SIMD take turns at the matrix pipe -- a wave raises its priority
        // for the issue of an MFMA and drops it right behind, so the other wave's ready MFMA wins the next slot
        asm volatile("s_setprio 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_setprio 0" : "+v"(c) : "v"(a), "v"(b));
        return c;
#endif
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ bf16x8 cvt(const float* x) {
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)x[j];
        return v;
    }
};
template <> struct Op<f16x8> {
    using E = _Float16;
    static __device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f16x8 cvt(const float* x) {
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)x[j];
        return v;
    }
};

#ifndef PG_PROGRESS_PRIO
#define PG_PROGRESS_PRIO 0    // 0: off; 4 / 2: priority steps per chunk (next_a)
#endif
#ifndef PG_SPREAD_DMA
#define PG_SPREAD_DMA 1       // weight-ring refill spread over the chunk (Stream::enter_split)
#endif
#ifndef PG_DMA_PHASE
#define PG_DMA_PHASE 2
#endif
#ifndef PG_DMA_WAVES
#define PG_DMA_WAVES 8
#endif
using StreamA = Stream<NWAVE, NCHUNK, PG_DMA_WAVES>;
static_assert(PG_DMA_WAVES == NWAVE || PG_SPREAD_DMA, "the bulk enter() waits vmcnt on every wave");

__device__ __forceinline__ const float* opaque_ptr(const float* p) {
    // LDS pointers are 32-bit offsets; keep the value in one VGPR the optimizer cannot split
    unsigned v = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
    asm volatile("" : "+v"(v));
    return (const float*)(const __attribute__((address_space(3))) float*)(uintptr_t)v;
}

#ifndef PG_ASYNC_X
#define PG_ASYNC_X true       // hand-pipelined ring reads in the x segments
#endif
#ifndef PG_ASYNC_VD
#define PG_ASYNC_VD true      // ... in the view-direction segment
#endif

template <typename V, typename ST>
__device__ __forceinline__ V unit_of(ST& st, int c, int pos) {
    return __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(st.at(c, pos * UNIT_BYTES)));
}

// Software pipeline of the A operand.  Left to hipcc, every ring read is sunk next to its MFMA
// (ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma): the whole LDS latency is exposed 1700 times per
// pass and the MFMA pipe idles ~45 %.  The reads are therefore issued by inline asm, two units
// ahead into a 3-register-set rotation, and retired by COUNTED lgkmcnt waits (the asm is
// invisible to hipcc's own waitcnt bookkeeping, whose waits for its own LDS reads only become
// more conservative).  The lookahead never crosses a chunk boundary (the next chunk is only
// readable after its enter(), which also drains lgkmcnt).
typedef __attribute__((ext_vector_type(4))) unsigned a128;
#ifndef PG_EARLY_RETIRE
#define PG_EARLY_RETIRE 0
#endif
#ifndef PG_PIPE_H
#define PG_PIPE_H 3           // register sets of the A pipe in the register-light segments (hidden layers, heads)
#endif
#ifndef PG_PIPE_X
#define PG_PIPE_X 3           // ... in the segments that hold 8 (x) or 4 (view) accumulator tiles
#endif
template <typename V, int NS = PG_PIPE_H> struct APipe { a128 r[NS]; };
template <typename V> using APipeX = APipe<V, PG_PIPE_X>;

__device__ __forceinline__ void lds_retire(a128& r, int younger) {
    switch (younger) {   // constant after unrolling
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r)); break;
        case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r)); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r)); break;
        case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r)); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(r)); break;
        case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(r)); break;
        case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(r)); break;
        case 7: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(r)); break;
        case 8: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(r)); break;
        default: __builtin_unreachable();
    }
}

// A fragment of unit L (compile-time after unrolling) of a segment with T units: NS register
// sets, reads issued NS-1 units ahead.  ASYNC = false: a plain (compiler-scheduled) read.
// c0: the segment's first chunk in the stream when the caller knows it (Stream::plain_ok: static ring bookkeeping), else -1
template <typename V, int T, bool ASYNC, int NS, typename ST>
__device__ __forceinline__ V next_a(APipe<V, NS>& p, ST& st, int L, int c0 = -1) {
    constexpr int PER = ST::PER, PSTRIDE = UPC / PER, LA = NS - 1;
    const int q = L % UPC;
    if (!ASYNC) {
        if (q == 0) st.enter(L / UPC);
        return unit_of<V>(st, 0, q);
    }
    const int rem = min(T - 1 - L, UPC - 1 - q);      // later units of this segment in this chunk
#if defined(PG_ABL_NOREAD)      // timing ablation only (wrong results): no ring reads, no waits
    if (q == 0) { if (PG_SPREAD_DMA) st.enter_split(); else st.enter(L / UPC); }
    if (PG_SPREAD_DMA) {
        if (q % PSTRIDE == PG_DMA_PHASE) st.piece(q / PSTRIDE);
        if (L == T - 1)
            for (int i = (q < PG_DMA_PHASE ? 0 : (q - PG_DMA_PHASE) / PSTRIDE + 1); i < PER; ++i) st.piece(i);
    }
    asm volatile("" : "+v"(p.r[L % NS]));
    return __builtin_bit_cast(V, p.r[L % NS]);
#endif
    if (q == 0) {
        if (PG_SPREAD_DMA) st.enter_split(c0 >= 0 && ST::plain_ok(c0 + L / UPC)); else st.enter(L / UPC);
        for (int k = 0; k < LA; ++k)
            if (k <= rem) st.issue(p.r[(L + k) % NS], q + k);
    }
#if PG_PROGRESS_PRIO
    // The two waves of a SIMD share its issue port, oldest first: the older one runs ahead through a chunk and then waits
    // at the next chunk barrier while the younger one finishes alone (stamps: the older waves spend a fifth of a pass in
    // s_barrier).  Issue priority by PROGRESS instead: a wave lowers its own priority as it advances through the chunk, so
    // whichever wave is behind wins the arbitration and the two move through the chunk together, each filling the
    // other's stalls.
    if (q % (UPC / PG_PROGRESS_PRIO) == 0) {
        switch (3 - (q / (UPC / PG_PROGRESS_PRIO)) * (4 / PG_PROGRESS_PRIO)) {       // (constant after unrolling; the builtin wants a literal)
            case 3: __builtin_amdgcn_s_setprio(3); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
        }
    }
#endif
    if (LA <= rem) st.issue(p.r[(L + LA) % NS], q + LA);
    if (PG_SPREAD_DMA) {
        // refill piece i of the freed slot goes out at unit i*PSTRIDE+PG_DMA_PHASE of this
        // chunk; a segment ending inside the chunk flushes the rest with its last unit
        if (q % PSTRIDE == PG_DMA_PHASE) st.piece(q / PSTRIDE);
        if (L == T - 1)
            for (int i = (q < PG_DMA_PHASE ? 0 : (q - PG_DMA_PHASE) / PSTRIDE + 1); i < PER; ++i) st.piece(i);
    }
    if (PG_EARLY_RETIRE) {
        // The retire "writes" its register as far as hipcc knows, and a VALU write directly
        // before an MFMA read costs an s_nop: retire unit L+1 here, one MFMA early, so only
        // the first unit of a chunk pays it.
        if (q == 0) lds_retire(p.r[L % NS], min(LA, rem));
        if (rem >= 1) lds_retire(p.r[(L + 1) % NS], min(LA, rem) - 1);
    } else {
        lds_retire(p.r[L % NS], min(LA, rem));
    }
    return __builtin_bit_cast(V, p.r[L % NS]);
}

// An LDS read that runs BESIDE the weight-ring pipe (a bias tile, the (a, b) rows of the next joint): issued by inline
// asm like the ring reads, because a read hipcc can see makes it wait `lgkmcnt(0)` in front of the consumer -- it does
// not count the hand-issued ring reads, so that wait drains the A pipe (every tile boundary and every joint of the
// embedding exposed a full LDS latency that way).  LDS operations return in order: issued in front of next_a(L)'s own
// issue, the read has landed once next_a(L + 1) has retired its unit (that wait leaves only the LA youngest ring reads
// outstanding), so it needs no wait of its own -- only lds_landed() behind such a retire as the point hipcc may
// read the register from (asm volatile statements keep their order).
__device__ __forceinline__ void lds_async128(a128& dst, unsigned lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:0+0" : "=v"(dst) : "v"(lds_addr));
}
__device__ __forceinline__ void lds_landed(a128& r) { asm volatile("" : "+v"(r)); }
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// Units of a segment that CONTINUES a chunk already entered and whose refill pieces are all out (the rgb head behind
// the alpha / view segment): positions Q0 .. Q0+T-1 of the current chunk, no chunk entry, no refill pieces.
template <typename V, int T, int NS, int Q0, typename ST>
__device__ __forceinline__ V next_a_cont(APipe<V, NS>& p, ST& st, int L) {
    static_assert(Q0 + T <= UPC, "a continuation stays inside its chunk");
    constexpr int LA = NS - 1;
    const int q = Q0 + L, rem = T - 1 - L;
    if (L == 0)
        for (int k = 0; k < LA; ++k)
            if (k <= rem) st.issue(p.r[(L + k) % NS], q + k);
    if (LA <= rem) st.issue(p.r[(L + LA) % NS], q + LA);
    lds_retire(p.r[L % NS], min(LA, rem));
    return __builtin_bit_cast(V, p.r[L % NS]);
}

#ifndef PG_PIN_TILE
#define PG_PIN_TILE 1
#endif
#ifndef PG_RELU_AT
#define PG_RELU_AT 3
#endif
#ifndef PG_BIAS_AT
#define PG_BIAS_AT 9
#endif
#ifndef PG_SETPRIO
#define PG_SETPRIO 0
#endif
// PG_ALT_PRIO (experiment): the two waves of a SIMD (w and w + NWAVE/2) take turns at raised issue priority,
// switching every out tile / unit row, so that both reach a chunk barrier together instead of the older one
// winning every arbitration and then waiting there (stamps: waves 0-3 spend 27 % of a pass in s_barrier).
__device__ __forceinline__ void alt_prio(int wave, int phase) {
#if defined(PG_ALT_PRIO)
    const bool hi = (((phase / PG_ALT_PRIO) & 1) != 0) == (wave >= 4);      // wave-uniform
    if (hi) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
}
// one B fragment against NO out tiles; unit row uu of a k-major segment with T units
template <typename V, int NO, int T, bool ASYNC, int NS, typename ST>
__device__ __forceinline__ void mma_row(f32x16* acc, APipe<V, NS>& p, ST& st, int uu, V b) {
    if (PG_SETPRIO) __builtin_amdgcn_s_setprio(PG_SETPRIO);
    alt_prio(st.wave, uu);
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] = Op<V>::mfma(next_a<V, T, ASYNC, NS>(p, st, uu * NO + o), b, acc[o]);
    if (PG_SETPRIO) __builtin_amdgcn_s_setprio(0);
}

// acc (+)= W[:, x-columns] * x : the 432-wide density input, generated on the fly
// `skb` = this lane half's 12 bone rows, `cutb` its 12 cutoff distances: both are opaque
// bases (see opaque_ptr) so that each joint is an immediate offset, not a live register.
// bone-local position of the wave's point for joint jj of its lane half
struct QFromRows {      // q = R p + t from the bone rows (classic table)
    const float* skb; float px, py, pz;
    __device__ __forceinline__ void operator()(int jj, float& qx, float& qy, float& qz) const {
        bone_local(skb + jj * 12, px, py, pz, qx, qy, qz);
    }
};
struct QFromAB {        // q = a + z b from the per-ray (a, b) table (pg_layout.h SLOTF_AB)
    const float* ab; float z;
    __device__ __forceinline__ void operator()(int jj, float& qx, float& qy, float& qz) const {
        const float4 lo = *reinterpret_cast<const float4*>(ab + jj * 8);
        const float4 hi = *reinterpret_cast<const float4*>(ab + jj * 8 + 4);
        qx = fmaf(z, hi.x, lo.x); qy = fmaf(z, hi.y, lo.y); qz = fmaf(z, hi.z, lo.z);
    }
};

template <typename V, typename ST, typename Q>
__device__ __forceinline__ void x_segment(f32x16* acc, ST& st, int cbase, const Q& qof,
                                          const float* cutb, float tau) {
    APipeX<V> p;
    constexpr int T = XU * NT;
#pragma clang loop unroll(full)
    for (int sb = 0; sb < 3; ++sb) {
        float lo[8];
#pragma clang loop unroll(full)
        for (int k = 0; k < 4; ++k) {
            const int jj = 4 * sb + k;
            float x[18];
            float qx, qy, qz;
            qof(jj, qx, qy, qz);
            joint_values_q<true>(qx, qy, qz, tau, cutb[jj], x);
            lo[2 * k] = x[16];
            lo[2 * k + 1] = x[17];
            mma_row<V, NT, T, PG_ASYNC_X>(acc, p, st, sb * 9 + 2 * k, Op<V>::cvt(x));
            mma_row<V, NT, T, PG_ASYNC_X>(acc, p, st, sb * 9 + 2 * k + 1, Op<V>::cvt(x + 8));
        }
        mma_row<V, NT, T, PG_ASYNC_X>(acc, p, st, sb * 9 + 8, Op<V>::cvt(lo));
    }
}

// ReLU on packed 16-bit floats: a negative bf16/fp16 is a negative int16, so max_i16(x, 0)
// clears exactly the negative lanes (v_pk_max_i16: one VALU op per two channels; fmaxf on the
// fp32 accumulators costs two each because hipcc canonicalises MFMA results first).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// One asm statement per fragment; the trailing s_nop 1 provides the two wait states a VALU
// write needs before an MFMA may read the register (hipcc pads nothing for inline asm).
template <typename V>
__device__ __forceinline__ V relu16(V v) {
    const u32x4 w = __builtin_bit_cast(u32x4, v);
    unsigned r0, r1, r2, r3;
    asm("v_pk_max_i16 %0, %4, 0\n\tv_pk_max_i16 %1, %5, 0\n\tv_pk_max_i16 %2, %6, 0\n\tv_pk_max_i16 %3, %7, 0\n\ts_nop 1"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)      // early-clobber: outputs are written
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));      // before the later inputs are read
    const u32x4 o = {r0, r1, r2, r3};
    return __builtin_bit_cast(V, o);
}

template <typename V>
__device__ __forceinline__ void relu_pack(const f32x16& acc, V& f0, V& f1, bool relu) {
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = acc[r];
    f0 = Op<V>::cvt(t);
    f1 = Op<V>::cvt(t + 8);
    if (relu) { f0 = relu16<V>(f0); f1 = relu16<V>(f1); }
}

// acc += W[tile o] * fin over the HU hidden units; out-tile-major segment starting at cbase
template <typename V, int T, bool ASYNC, int NS, typename ST>
__device__ __forceinline__ void row_tile(f32x16& acc, APipe<V, NS>& p, ST& st, int o, const V* fin) {
#pragma unroll
    for (int u = 0; u < HU; ++u) acc = Op<V>::mfma(next_a<V, T, ASYNC, NS>(p, st, o * HU + u), fin[u], acc);
}

// fout = relu(W fin + b), out-tile-major segment starting at chunk cbase.
// The ReLU + 16-bit packing of tile o-1 (VALU, needs that tile's last MFMA to retire) is placed
// after the first MFMAs of tile o, so it runs under them instead of draining the MFMA pipe.
template <typename V, typename ST>
__device__ __forceinline__ void hidden_layer(const V* fin, V* fout, ST& st, int cbase,
                                             const float* bias, int tile0, int h) {
    APipe<V> p;
    f32x16 prev, nextb = load_bias(bias, tile0, h);
#pragma unroll
    for (int o = 0; o < NT; ++o) {
        f32x16 acc = nextb;
        constexpr int T = HU * NT;
        alt_prio(st.wave, o);
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            acc = Op<V>::mfma(next_a<V, T, true, PG_PIPE_H>(p, st, o * HU + u), fin[u], acc);
            if (u == PG_RELU_AT && o > 0) {
                // pinned: hoisted to the tile boundary the conversions wait out the last MFMA
                if (PG_PIN_TILE) __builtin_amdgcn_sched_barrier(0);
                relu_pack<V>(prev, fout[2 * (o - 1)], fout[2 * (o - 1) + 1], true);
                if (PG_PIN_TILE) __builtin_amdgcn_sched_barrier(0);
            }
            // the next tile's bias is read mid-tile (the volatile ring reads pin it here), not at
            // the boundary where the first MFMA would wait out the LDS latency
            if (PG_PIN_TILE && u == PG_BIAS_AT && o + 1 < NT) nextb = load_bias(bias, tile0 + o + 1, h);
        }
        if (!PG_PIN_TILE && o + 1 < NT) nextb = load_bias(bias, tile0 + o + 1, h);
        prev = acc;
    }
    relu_pack<V>(prev, fout[2 * (NT - 1)], fout[2 * (NT - 1) + 1], true);
}

// debug: write a fragment array (H-sequence order) as floats to dbg[pt][256]
template <typename V, int NF>
__device__ __forceinline__ void dump_frags(const EvalArgs& a, int stage, long long gp, bool valid, const V* f, int h) {
    if (a.dbg && a.dbg_stage == stage && valid) {
#pragma unroll
        for (int u = 0; u < NF; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) a.dbg[gp * W + hseq_channel(8 * u + j, h)] = (float)f[u][j];
    }
}


#if defined(PG_STAMPS)
#define PG_STAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[k] = t_; } while (0)
#else
#define PG_STAMP(k) do {} while (0)
#endif

// ---- on-chip variants of the record kernels: the per-ray rows are formed by the workgroup itself ----
// 64 dwords by LDS-DMA with a per-lane source offset (bytes from a wave-uniform base) to LDS dst + 4 lane
// (s_nop 4: the base may sit in SGPRs a v_readlane / v_readfirstlane has just written -- a spill reload, a wave-uniform index --
// and VMEM reading such an SGPR needs 5 wait states, which hipcc does not insert in front of inline asm)
__device__ __forceinline__ void dma_dwords(const void* base, uint32_t lane_off, uint32_t lds_dst) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" :: "s"(lds_dst), "v"(lane_off), "s"(base) : "memory");
}

// One (ray, joint slot) record row of the on-chip variants (pg_eval16r.hip OC, pg_evalc.hip OC) -- what pg_rayrec.hip writes to HBM for the record variant:
// a = R_j o + t_j, b = R_j d (encoders.py:8-37) and, in a.w, the squared distance of the ray's sampled segment
// [z0, z1] from the joint (pass_far_mask).  sk = the joint's three bone rows (R | t), ray = (o, d).
__device__ __forceinline__ void ab_row(const float* sk, const float* ray, float z0, float z1, float4* dst) {
    const float ox = ray[0], oy = ray[1], oz = ray[2], dx = ray[3], dy = ray[4], dz = ray[5];
    const float ax = fmaf(sk[2], oz, fmaf(sk[1], oy, fmaf(sk[0], ox, sk[3])));
    const float ay = fmaf(sk[6], oz, fmaf(sk[5], oy, fmaf(sk[4], ox, sk[7])));
    const float az = fmaf(sk[10], oz, fmaf(sk[9], oy, fmaf(sk[8], ox, sk[11])));
    const float bx = fmaf(sk[2], dz, fmaf(sk[1], dy, sk[0] * dx));
    const float by = fmaf(sk[6], dz, fmaf(sk[5], dy, sk[4] * dx));
    const float bz = fmaf(sk[10], dz, fmaf(sk[9], dy, sk[8] * dx));
    const float bb = bx * bx + by * by + bz * bz, ab = ax * bx + ay * by + az * bz;
    float zs = bb > 0.0f ? -ab / bb : z0;
    zs = fminf(fmaxf(zs, fminf(z0, z1)), fmaxf(z0, z1));
    const float qx = fmaf(zs, bx, ax), qy = fmaf(zs, by, ay), qz = fmaf(zs, bz, az);
    const float d2 = qx * qx + qy * qy + qz * qz;
    dst[0] = make_float4(ax, ay, az, d2 == d2 ? d2 : 0.0f);
    dst[1] = make_float4(bx, by, bz, 0.0f);
}

}  // namespace pgd
