// pg_evalc.hip -- fused bone-relative embedding + NeRF MLP in COMPENSATED fp16 (PG_PREC_FP16C):
// the precision mode that meets the reference to <= 1e-4 at MFMA speed.  Shape C of pg_program.h.
//
// Replaces RayCaster.encode_inputs + run_network + NeRF.forward for one net (reference
// core/raycasters.py:476-577, core/networks/nerf.py:90-148, core/encoders.py,
// core/cutoff_embedder.py) on n*S points p = o + d*z, like pg_eval16.hip / pg_eval32.hip.
//
// Arithmetic.  Every product W x of the network is formed as
//       (S-1) w1 x1 + w2 x2,      w = W / S,  t1 = f16(t),  t2 = f16(t1 + S (t - t1)),  S = 129
// i.e. TWO v_mfma_f32_32x32x16_f16 per 16 k into ONE fp32 accumulator: the second product carries the
// first-order rounding terms of both operands (w_lo x + w x_lo) scaled by S, and S-1 = 128 is a
// power of two, so (S-1) w1 is exact and no second accumulator or epilogue is needed.  What is
// left is (S-1) w_lo x_lo + (e_w x + w e_x)/S ~ 2^-17 |w x| per product (plain fp16: 2^-11;
// tools/error_budget.py: rgb/acc within 4e-6 / 8e-6 of the fp32 oracle where fp16 has 2e-4).
// The weight planes (S-1) w1 and w2 are formed on the host in double (pg_pack.cpp); the
// activation pair is formed by conv_a / conv_b below from ONE v_cvt_pk_f16_f32 result.
//
// Shape.  Workgroup = 4 waves, one per SIMD, up to 512 registers; a wave owns 32 consecutive points
// of the flattened [ray][sample] list.  Every segment is k-major: an input unit (8 values per lane
// = one B-fragment pair) is multiplied into all out tiles of the layer.  The pre-activations of a
// layer stay in the accumulators (the compiler keeps the two ping-pong sets of 8 tiles in AGPRs);
// the NEXT unit's values are read back, ReLU'd and split into their fp16 pair in small steps that
// are pinned between the MFMA pairs of the CURRENT unit (__builtin_amdgcn_sched_barrier), because
// with one wave per SIMD nothing else would overlap that VALU work with the matrix pipe.
// The weight stream is staged L2 -> LDS by LDS-DMA into the 3 x 32 KiB ring of pg_device.h and read
// by hand-issued ds_read_b128 four units ahead with counted lgkmcnt waits (pg_eval16_common.h).
// Two forms of the view layer's direction part (template parameter REC of evalc_kernel): the direct one (per-ray
// sin/cos table in LDS x per-point cutoff weight, 32 <= samples per ray < 64) and, for rays with >= 64 samples, the
// record variant: per-ray (a, b) and split Y records of pg_rayrec.hip fetched by LDS-DMA, a second stage of 16-32 MFMAs
// on the point's cutoff weights instead of 336, no table build, no divisions.  In the record variant the joint PAIRS (the
// two lane halves' joints of a k-unit) a wave's points -- or a whole pass -- are out of cutoff range of are left out
// (x_segment_cr, Stream MASK_NX; pg_eval16r.hip explains the test), and for one pose per call without frame codes it has
// an on-chip form (OC) that needs no records at all: the rows are formed by the workgroup a pass ahead and the direction
// part Y by an MFMA segment of its own (y_segment_c).
#include <type_traits>

#include "pg_comp.h"

// cache policy of the per-ray record fetches (streamed once; must not evict the weight stream from L2)
#ifndef PG_REC_POLICY
#define PG_REC_POLICY " nt"      // (-0.4 % on the same box: the 2.9-MB weight stream shares a 4-MB L2 with the streamed records)
#endif

namespace pgd {

constexpr int NWAVE_C = 4;
constexpr int NTHR_C = NWAVE_C * 64;
constexpr int PTS_C = NWAVE_C * 32;
using StreamC = Stream<NWAVE_C, pgp::C::NCHUNK, NWAVE_C>;
// record variant: no view-direction segment; the joint-pair chunks of both x segments can be left out of a pass (Stream MASK_NX)
using StreamCR = Stream<NWAVE_C, pgp::C::NCHUNK_R, NWAVE_C, pgp::C::NPAIRJ, 0, pgp::C::C_L5XR>;
// on-chip variant (no per-ray records in HBM): the joint-pair chunks of the view layer's direction weights sit behind layer 0
using StreamCRO = Stream<NWAVE_C, pgp::C::NCHUNK_OC, NWAVE_C, pgp::C::NPAIRJ, 0, pgp::C::C_L5XR_OC, pgp::C::C_YC>;
// where the ring bookkeeping of a chunk entry is static (Stream::plain_ok): not where a maskable chunk or the wrap is within reach
static_assert(StreamCRO::plain_ok(27) && StreamCRO::plain_ok(63) && !StreamCRO::plain_ok(64) && !StreamCRO::plain_ok(70) && StreamCRO::plain_ok(82) &&
              StreamCRO::plain_ok(100) && !StreamCRO::plain_ok(101) && !StreamCRO::plain_ok(12) && !StreamCRO::plain_ok(5) && StreamCRO::plain_ok(26), "on-chip stream: L1 .. L5h, L6 .. rgb");
static_assert(StreamCR::plain_ok(15) && StreamCR::plain_ok(51) && !StreamCR::plain_ok(52) && StreamCR::plain_ok(70) && StreamCR::plain_ok(88) &&
              !StreamCR::plain_ok(89), "record stream");
#ifndef PG_NSC
#define PG_NSC 5
#endif
constexpr int NSC = PG_NSC;                   // register sets of the A pipe: reads issued NSC-1 PAIRS of units ahead (two pairs retire per wait)
constexpr int LDS_TOTAL_C = LDS_RTAB + MAXR_C * SLOTC_FLOATS * 4;
static_assert(LDS_TOTAL_C <= 160 * 1024, "LDS budget of one CU");
// LDS carve-up of the record variant (bytes): the ring, a compacted bias table (pg_layout.h BTC_*), the cutoff
// table, two (a, b) buffers (this pass / the next) and the Y records of the pass's <= MAXR_CR rays
constexpr int LDSC_BIAS = PG_RING_SLOTS * CHUNK_BYTES;
constexpr int LDSC_CUT = LDSC_BIAS + BTC_COUNT * 32 * 4;
constexpr int LDSC_AB = LDSC_CUT + 72 * 4;          // (cut: 72 floats by joint slot -- both embedders' constants, far^2)
constexpr int LDSC_Y = LDSC_AB + 2 * LDS_ABC_BYTES;
constexpr int LDS_TOTAL_CR = LDSC_Y + MAXR_CR * RECC_Y_BYTES;
static_assert(LDSC_BIAS % 16 == 0 && LDSC_CUT % 16 == 0 && LDSC_AB % 16 == 0 && LDSC_Y % 16 == 0, "LDS alignment");
static_assert(LDS_TOTAL_CR <= 160 * 1024, "LDS budget of one CU");
// on-chip variant (OC): the (a, b) buffers hold exactly MAXR_CR rays (no DMA-piece granularity), behind the Y image the
// pose's bone rows by joint slot (24 x 12 floats) and a staging area for the next pass's rays (64 floats of ray_batch
// rows, 64 floats of first / last depths)
constexpr int LDSO_ABSZ = MAXR_CR * REC_AB_BYTES;
constexpr int LDSO_Y = LDSC_AB + 2 * LDSO_ABSZ;
constexpr int LDSO_SK = LDSO_Y + MAXR_CR * RECC_Y_BYTES;
constexpr int LDSO_STAGE = LDSO_SK + J * 12 * 4;
constexpr int LDS_TOTAL_CO = LDSO_STAGE + 512;
static_assert(LDSO_Y % 16 == 0 && LDSO_SK % 16 == 0 && LDSO_STAGE % 16 == 0 && LDS_TOTAL_CO <= 160 * 1024, "LDS budget of one CU (on-chip variant)");

// one LDS-DMA piece (1 KiB, lane-linear) from a wave-uniform source to a wave-uniform LDS address; counted by the
// chunk entries' vmcnt like the ring's own pieces
__device__ __forceinline__ void dma_piece_c(const uint8_t* src, uint32_t lds_dst, uint32_t lane16) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" PG_REC_POLICY :: "s"(lds_dst), "v"(lane16), "s"(src) : "memory");
}

// ---- the A operand pair pipe -------------------------------------------------------------------
// A wave issues one instruction per ~4 cycles, so with one wave per SIMD an MFMA slot (32 cycles) has
// room for ~7 other instructions: every instruction per MFMA counts.  Versus the per-unit pipe of
// pg_eval16_common.h (next_a) this one works on PAIRS of 1-KiB units (the two planes of one weight
// tile), retires a pair with ONE counted wait, and does so between the two MFMAs of the PREVIOUS
// pair, so that the registers the wait "writes" (as far as hipcc knows) are not read by the very
// next instruction (which costs an s_nop per MFMA).  Refill pieces share their SGPR bases in
// groups of four through the instruction's offset field (it applies to the global and the LDS
// address alike): 2.75 instead of 6 instructions per piece.
constexpr int PPC = CHUNK_BYTES / 2048;       // pairs per chunk
#if defined(PG_ABL_SINGLE)
#define PG_PL1 0
#else
#define PG_PL1 1
#endif
template <int NS> struct PairPipe {
    a128 r[NS][2];
    const uint8_t* g;      // global base of this wave's current group of four refill pieces
    uint32_t m;            // its LDS base (m0)
};

__device__ __forceinline__ void retire_pair(a128& r0, a128& r1, int younger) {
    switch (younger) {   // constant after unrolling: LGKM operations younger than the pair
#define PG_RP(N) case N: asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(r0), "+v"(r1)); break;
        PG_RP(0) PG_RP(1) PG_RP(2) PG_RP(3) PG_RP(4) PG_RP(5) PG_RP(6) PG_RP(7) PG_RP(8) PG_RP(9) PG_RP(10) PG_RP(11) PG_RP(12) PG_RP(13) PG_RP(14)
#undef PG_RP
        default: __builtin_unreachable();
    }
}

template <int NS, typename ST>
__device__ __forceinline__ void piece_c(PairPipe<NS>& p, const ST& st, int i) {
#if defined(PG_ABL_NODMA)       // timing ablation only (wrong results): no refill of the ring
    return;
#endif
    if ((i & 3) == 0) {
        p.g = st.wstream + (st.cur_src + i * 1024);
        p.m = st.ring_lds + st.cur_dst + i * 1024;
        asm volatile("" : "+s"(p.g), "+s"(p.m));
    }
    switch (i & 3) {
#define PG_PC(K) case K: asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2 offset:" #K "*1024" \
                                      :: "s"(p.m), "v"(st.lane16), "s"(p.g) : "memory"); break;
        PG_PC(0) PG_PC(1) PG_PC(2) PG_PC(3)
#undef PG_PC
    }

}

#if PG_RING_SLOTS >= 4
// Continuous pair pipe (ring of >= 4 slots, Stream::enter_ahead): the reads of the first pairs of a chunk are
// issued during the last pairs of the chunk before, so nothing restarts behind a chunk barrier; the pipe is
// empty only at the start of a segment.
template <int NS, typename ST>
__device__ __forceinline__ void issue_pair(PairPipe<NS>& p, const ST& st, int P, int cur_chunk) {
    const int q = P % PPC;
    if (P / PPC == cur_chunk) { st.issue(p.r[P % NS][0], 2 * q); st.issue(p.r[P % NS][1], 2 * q + 1); }
    else { st.issue_ahead(p.r[P % NS][0], 2 * q); st.issue_ahead(p.r[P % NS][1], 2 * q + 1); }
}
template <int TP, int NS, typename ST>
__device__ __forceinline__ void pair_begin(PairPipe<NS>& p, ST& st, int P) {
    constexpr int LA = NS - 1;
    static_assert(LA <= 6, "issue_ahead covers the first 12 units of the next chunk");
    if (P % PPC != 0) return;
    st.enter_ahead();
    if (P != 0) return;                                  // the pipe is already running
#pragma clang loop unroll(full)
    for (int k = 0; k < LA; ++k)
        if (k < TP) issue_pair(p, st, k, 0);
    retire_pair(p.r[0][0], p.r[0][1], 2 * (min(LA, TP) - 1));
}
template <int TP, int NS, typename ST>
__device__ __forceinline__ void pair_mid(PairPipe<NS>& p, ST& st, int P) {
    constexpr int LA = NS - 1;
    const int q = P % PPC;
    if (P + LA < TP) issue_pair(p, st, P + LA, P / PPC);
    if (q & 1) piece_c(p, st, q >> 1);
    if (P == TP - 1)            // a segment ending inside the chunk flushes the rest of the refill
        for (int i = (q + 1) >> 1; i < ST::PER; ++i) piece_c(p, st, i);
    if (P + 1 < TP) retire_pair(p.r[(P + 1) % NS][0], p.r[(P + 1) % NS][1], 2 * (min(P + LA, TP - 1) - (P + 1)));
}
#else
// before the first MFMA of pair P (of a segment with TP pairs): chunk entry, and the pair must be there
#if defined(PG_ABL_SINGLE)      // timing ablation only (wrong results): one weight plane -- half the ring reads and half the refill
constexpr int ABL_PL = 1;
#else
constexpr int ABL_PL = 2;
#endif
// c0: the segment's first chunk in the stream (compile-time; -1 = unknown / inside a maskable range)
template <int TP, int NS, typename ST>
__device__ __forceinline__ void pair_begin(PairPipe<NS>& p, ST& st, int P, int c0 = -1) {
    constexpr int LA = NS - 1;
    if (P % PPC != 0) return;
    st.enter_split(c0 >= 0 && ST::plain_ok(c0 + P / PPC));
    const int rem = min(TP - 1 - P, PPC - 1);            // later pairs of this segment in this chunk
#if defined(PG_ABL_NOREAD)      // timing ablation only (wrong results): no ring reads, no waits for them
    asm volatile("" : "+v"(p.r[P % NS][0]), "+v"(p.r[P % NS][1]));
    if (true) return;
#endif
#pragma clang loop unroll(full)
    for (int k = 0; k < LA; ++k)
        if (k <= rem) { st.issue(p.r[(P + k) % NS][0], 2 * k); if (ABL_PL == 2) st.issue(p.r[(P + k) % NS][1], 2 * k + 1); }
    retire_pair(p.r[P % NS][0], p.r[P % NS][1], ABL_PL * min(LA - 1, rem));
}
// between the two MFMAs of pair P: read pair P+LA, one refill piece every other pair, retire pair P+1
template <int TP, int NS, typename ST>
__device__ __forceinline__ void pair_mid(PairPipe<NS>& p, ST& st, int P) {
    constexpr int LA = NS - 1;
    const int q = P % PPC;
    const int rem = min(TP - 1 - P, PPC - 1 - q);
#if !defined(PG_ABL_NOREAD)
    if (LA <= rem) { st.issue(p.r[(P + LA) % NS][0], 2 * (q + LA)); if (ABL_PL == 2) st.issue(p.r[(P + LA) % NS][1], 2 * (q + LA) + 1); }
#endif
#if defined(PG_ABL_SINGLE)
    if ((q & 3) == 3) piece_c(p, st, q >> 2);
    if (P == TP - 1)
        for (int i = (q + 1) >> 2; i < ST::PER / 2; ++i) piece_c(p, st, i);
#else
    if (q & 1) piece_c(p, st, q >> 1);
    if (P == TP - 1)            // a segment ending inside the chunk flushes the rest of the refill
        for (int i = (q + 1) >> 1; i < ST::PER; ++i) piece_c(p, st, i);
#endif
    // one counted wait per TWO pairs, behind the even pairs of a chunk (pairs P + 1 and P + 2 retire together; the pipe is one
    // pair deeper for it): four instructions less per input unit of the SIMD's only wave (-0.4 ... -0.7 %)
#if defined(PG_ABL_NOREAD)
    asm volatile("" : "+v"(p.r[(P + 1) % NS][0]), "+v"(p.r[(P + 1) % NS][1]));
    if (true) return;
#endif
    if ((q & 1) == 0) {
        if (rem >= 2) {
            const int outstanding = ABL_PL * max(min(LA, rem) - 2, 0);
            retire_pair(p.r[(P + 1) % NS][0], p.r[(P + 1) % NS][1], outstanding);
            asm volatile("" : "+v"(p.r[(P + 2) % NS][0]), "+v"(p.r[(P + 2) % NS][1]));
        } else if (rem >= 1) retire_pair(p.r[(P + 1) % NS][0], p.r[(P + 1) % NS][1], ABL_PL * min(LA - 1, rem - 1));
    }
}

#endif

// One k-major segment: NU input units against NO out tiles, `src(u, e)` = value e of unit u (indices
// are compile-time constants after unrolling).  While the 2 NO MFMAs of unit u issue, unit u+1 is
// read, (ReLU'd,) and split: 8 half steps (A and B of 4 value pairs) spread over the NO out tiles.
// T = units (of 1 KiB) of the segment in the weight stream: 2 per (input unit, out tile).
template <int NO, int NU, bool RELU, typename ST, typename SRC>
__device__ __forceinline__ void segment_c(f32x16* acc, ST& st, const SRC& src, float s129, int c0 = -1) {
    constexpr int TP = NU * NO;
    constexpr int PER = (8 + NO - 1) / NO;            // half steps per out tile
    PairPipe<NSC> p;
    FragC cur, nxt;
    float ra = 0.f, rb = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                     // unit 0 up front
        conv_a<RELU>(src(0, 2 * j), src(0, 2 * j + 1), ra, rb, cur.x1[j]);
        cur.x2[j] = conv_b(ra, rb, cur.x1[j], s129);
    }
#pragma clang loop unroll(full)
    for (int u = 0; u < NU; ++u) {
#pragma clang loop unroll(full)
        for (int o = 0; o < NO; ++o) {
            const int P = u * NO + o;
            // two out tiles' pairs interleaved, so that no MFMA accumulates onto the one right before it (-0.5 %; hipcc still
            // moves the second plane's MFMA of tile o above the first of tile o + 1 in places -- pinning the order: no change)
            if (NO % 2 == 0) {
                if ((o & 1) == 0) {
                    pair_begin<TP>(p, st, P, c0);
                    acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, p.r[P % NSC][0]), frag_v(cur.x1), acc[o], 0, 0, 0);
                    pair_mid<TP>(p, st, P);
                    pair_begin<TP>(p, st, P + 1, c0);
                    acc[o + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, p.r[(P + 1) % NSC][0]), frag_v(cur.x1), acc[o + 1], 0, 0, 0);
                    acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, p.r[P % NSC][PG_PL1]), frag_v(cur.x2), acc[o], 0, 0, 0);
                } else {
                    pair_mid<TP>(p, st, P);
                    acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, p.r[P % NSC][PG_PL1]), frag_v(cur.x2), acc[o], 0, 0, 0);
                }
            } else {
                pair_begin<TP>(p, st, P, c0);
                acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, p.r[P % NSC][0]), frag_v(cur.x1), acc[o], 0, 0, 0);
                pair_mid<TP>(p, st, P);
                acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, p.r[P % NSC][PG_PL1]), frag_v(cur.x2), acc[o], 0, 0, 0);
            }
            // second gap of the pair: one half step of the next unit's split
#if defined(PG_ABL_NOCONV)      // timing ablation only (wrong results): the next unit's values are kept live but not split
            if (u + 1 < NU) {
#pragma clang loop unroll(full)
                for (int hs = o * PER; hs < (o + 1) * PER && hs < 8; ++hs)
                    if ((hs & 1) == 0) asm volatile("" :: "v"(src(u + 1, hs)), "v"(src(u + 1, hs + 1)));
            }
            if (false) {
#else
            if (u + 1 < NU) {
#endif
#pragma clang loop unroll(full)
                for (int hs = o * PER; hs < (o + 1) * PER && hs < 8; ++hs) {
                    const int j = hs >> 1;
                    if ((hs & 1) == 0) conv_a<RELU>(src(u + 1, 2 * j), src(u + 1, 2 * j + 1), ra, rb, nxt.x1[j]);
                    else nxt.x2[j] = conv_b<(NO < 2)>(ra, rb, nxt.x1[j], s129);
                }
            }
#if !defined(PG_ABL_NOSCHEDBAR)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
#if !defined(PG_ABL_NOCONV)
        cur = nxt;
#endif
    }
}

// one input unit already in registers against NO out tiles (x segments: no look-ahead conversion here)
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `hook` runs once, right behind the segment's first chunk entry (before that chunk's first refill piece)
template <int NO, int TP, typename ST, typename HK = NoHook>
__device__ __forceinline__ void mma_row_c(f32x16* acc, PairPipe<NSC>& p, ST& st, int uu, const FragC& b, const HK& hook = HK()) {
#pragma unroll
    for (int o = 0; o < NO; ++o) {              // (interleaving two out tiles' pairs as segment_c does is +0.2 % here)
        const int P = uu * NO + o;
        pair_begin<TP>(p, st, P);
        if (P == 0) hook();
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, p.r[P % NSC][0]), frag_v(b.x1), acc[o], 0, 0, 0);
        pair_mid<TP>(p, st, P);
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, p.r[P % NSC][PG_PL1]), frag_v(b.x2), acc[o], 0, 0, 0);
    }
}

// acc += W[:, x-columns] x: the 432-wide density input generated on the fly (X sequence of pg_layout.h).
// The values of joint jj+1 are computed and split while the units of joint jj go through the matrix pipe.
template <typename ST, typename HK = NoHook>
__device__ __forceinline__ void x_segment_c(f32x16* acc, ST& st, const float* ab, float z, const float* cutb,
                                            float tl, float s129, const HK& hook = HK()) {
    PairPipe<NSC> p;
    constexpr int T = pgp::C::XU * NT;
    auto values = [&](int jj, float* x) {
        const float4 lo = *reinterpret_cast<const float4*>(ab + jj * 8);
        const float4 hi = *reinterpret_cast<const float4*>(ab + jj * 8 + 4);
        joint_values_c(fmaf(z, hi.x, lo.x), fmaf(z, hi.y, lo.y), fmaf(z, hi.z, lo.z), tl, cutb[jj], x);
    };
    float xn[18];
    values(0, xn);
    FragC f0 = frag_of(xn, s129), f1 = frag_of(xn + 8, s129);
    float lo[8];
    lo[0] = xn[16]; lo[1] = xn[17];
#pragma clang loop unroll(full)
    for (int sb = 0; sb < 3; ++sb) {
#pragma clang loop unroll(full)
        for (int k = 0; k < 4; ++k) {
            const int jj = 4 * sb + k;
            const FragC c0 = f0, c1 = f1;
            if (jj + 1 < JH) values(jj + 1, xn);                               // next joint: embedding math ...
            mma_row_c<NT, T>(acc, p, st, sb * 9 + 2 * k, c0, hook);
            if (jj + 1 < JH) f0 = frag_of(xn, s129);                           // ... and its split, between this joint's rows
            mma_row_c<NT, T>(acc, p, st, sb * 9 + 2 * k + 1, c1);
            if (jj + 1 < JH) f1 = frag_of(xn + 8, s129);
            if (k == 3) {
                const FragC fl = frag_of(lo, s129);
                mma_row_c<NT, T>(acc, p, st, sb * 9 + 8, fl);
            }
            if (jj + 1 < JH) { lo[2 * ((k + 1) & 3)] = xn[16]; lo[2 * ((k + 1) & 3) + 1] = xn[17]; }
        }
    }
}

// Record variant: the XC sequence of pg_layout.h (per joint slot jj of the lane half two units of cutoff-weighted values
// = one chunk per joint PAIR, then six units of directions).  `wmask` bit jj (wave-uniform): the pair is out of cutoff
// range of all 32 points of the wave -- every value of its two units is below 6e-8 (pg_eval16r.hip x_segment16) --
// so the wave only keeps the ring going for that chunk.  `gmask` bit jj (workgroup-uniform): out of range of the
// whole pass; the chunk is not in the pass's chunk sequence (Stream MASK_NX).
template <int NO, int TP, typename ST, typename HK>
__device__ __forceinline__ void mma_row_cr(f32x16* acc, PairPipe<NSC>& p, ST& st, int uu, const FragC& b, bool& hooked, const HK& hook) {
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        const int P = uu * NO + o;
        pair_begin<TP>(p, st, P);
        if (o == 0 && !hooked) { hook(); hooked = true; }       // (behind the segment's first chunk entry)
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, p.r[P % NSC][0]), frag_v(b.x1), acc[o], 0, 0, 0);
        pair_mid<TP>(p, st, P);
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, p.r[P % NSC][PG_PL1]), frag_v(b.x2), acc[o], 0, 0, 0);
    }
}

template <typename ST, typename HK>
__device__ __forceinline__ void x_segment_cr(f32x16* acc, ST& st, const float* ab, float z, const float* cutb,
                                             float tl, float s129, int wmask, int gmask, const HK& hook) {
    PairPipe<NSC> p;
    constexpr int T = XUC * NT;
    bool hooked = false;                    // wave-uniform
    auto local = [&](int jj, float& qx, float& qy, float& qz) {
        const float4 lo = *reinterpret_cast<const float4*>(ab + jj * 8);
        const float4 hi = *reinterpret_cast<const float4*>(ab + jj * 8 + 4);
        qx = fmaf(z, hi.x, lo.x); qy = fmaf(z, hi.y, lo.y); qz = fmaf(z, hi.z, lo.z);
    };
#pragma clang loop unroll(full)
    for (int jj = 0; jj < JH; ++jj) {
        if ((gmask >> jj) & 1) continue;
        if ((wmask >> jj) & 1) {
            st.enter_split();
            if (!hooked) { hook(); hooked = true; }
#pragma unroll
            for (int i = 0; i < ST::PER; ++i) piece_c(p, st, i);
            continue;
        }
        float x[18], qx, qy, qz;
        local(jj, qx, qy, qz);
        joint_values_c(qx, qy, qz, tl, cutb[jj], x);
        x[15] = 0.0f;                       // (the directions x[15..17] have units of their own)
        const FragC f0 = frag_of(x, s129), f1 = frag_of(x + 8, s129);
        mma_row_cr<NT, T>(acc, p, st, 2 * jj, f0, hooked, hook);
        mma_row_cr<NT, T>(acc, p, st, 2 * jj + 1, f1, hooked, hook);
    }
    // r = q / max(|q|, 1e-12) of every joint slot (not cutoff-weighted): six units of two slots each
#pragma clang loop unroll(full)
    for (int pr = 0; pr < JH / 2; ++pr) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float qx, qy, qz;
            local(2 * pr + k, qx, qy, qz);
            const float rinv = __builtin_amdgcn_rsqf(fmaxf(qx * qx + qy * qy + qz * qz, 1e-24f));
            v[3 * k] = qx * rinv; v[3 * k + 1] = qy * rinv; v[3 * k + 2] = qz * rinv;
        }
        v[6] = v[7] = 0.0f;
        const FragC f = frag_of(v, s129);
        mma_row_cr<NT, T>(acc, p, st, XVC + pr, f, hooked, hook);
    }
}

// On-chip variant: the view layer's direction part Y[ray][joint][out] = sum_k W_vd[out, (joint, k)] T[ray][joint][k]
// (pg_layout.h "factorised view layer") of the pass's rays for the joint pairs in range of the pass, as the split A
// operands the second stage reads (RECC layout) -- no per-ray record in HBM.  One stream chunk per joint pair p: unit
// pairs [k-unit u of 8 view values][out tile o]; wave w takes out tile w.  The MFMA runs transposed: A = the rays' view
// values (row r = ray r of slot p from lane half 0, row 4 + r = ray r of slot 12 + p from lane half 1, every other lane
// zero, so that the two halves' joints do not mix), B = the weights (column = out channel); lane (h, col) then holds
// Y[ray r][slot 12 h + p][32 w + col] in accumulator register r.  T from the record's b = R_j d: e = b / |b|, rows
// (e, sin e, cos e, .., sin 8 e, cos 8 e) per component (encoders.py:172-193, cutoff_embedder.py:45-46), hardware
// sin / cos, split like an activation; Y / S split like a weight (pg_rayrec.hip recc_unit).  Pairs out of range of the
// whole pass keep whatever an earlier pass left (zeros at first): the second stage multiplies them by exactly zero.
template <typename ST>
__device__ __forceinline__ void y_segment_c(ST& st, int gmask, const uint8_t* ab, uint8_t* ylds, int nrm1, int wave, int lane, float s129) {
    const int h = lane >> 5, ray = (lane & 31) - 4 * h;
    const bool live = ray >= 0 && ray <= nrm1;
    const uint8_t* brow = ab + (live ? ray : 0) * REC_AB_BYTES + JH * h * 32 + 16;        // b rows of this lane half's joint slots
    uint8_t* ydst = ylds + wave * 4096 + lane * 16;
    PairPipe<NSC> pp;
#pragma unroll 1
    for (int pj = 0; pj < JH; ++pj) {
        if ((gmask >> pj) & 1) continue;
        st.enter_split();
        uint4 w0[4], w1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint8_t* src = st.at(0, (u * NTV + wave) * 2048);
            w0[u] = *reinterpret_cast<const uint4*>(src);
            w1[u] = *reinterpret_cast<const uint4*>(src + 1024);
        }
#pragma unroll
        for (int i = 0; i < ST::PER; ++i) piece_c(pp, st, i);
        const float4 b = *reinterpret_cast<const float4*>(brow + pj * 32);
        const float inv = __builtin_amdgcn_rsqf(fmaxf(b.x * b.x + b.y * b.y + b.z * b.z, 1e-24f));
        const float ev[3] = {b.x * inv, b.y * inv, b.z * inv};
        const float rv[3] = {ev[0] * 0.15915494309189535f, ev[1] * 0.15915494309189535f, ev[2] * 0.15915494309189535f};
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float tv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 8 * u + e, c = k / ROWS_D, r9 = k % ROWS_D;        // (constants after unrolling)
                if (k >= 3 * ROWS_D) tv[e] = 0.0f;
                else if (r9 == 0) tv[e] = ev[c];
                else {
                    const float ang = rv[c] * (float)(1 << ((r9 - 1) >> 1));
                    tv[e] = ((r9 - 1) & 1) ? __builtin_amdgcn_cosf(ang) : __builtin_amdgcn_sinf(ang);
                }
            }
            // gfx940-family trans forwarding: a VALU instruction may not read the result of v_sin / v_cos in the very next
            // issue slot; hipcc pads its own instructions, not the inline asm of conv_a that consumes these values
            asm volatile("s_nop 0" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]), "+v"(tv[4]), "+v"(tv[5]), "+v"(tv[6]), "+v"(tv[7]));
            FragC f = frag_of(tv, s129);
#pragma unroll
            for (int q = 0; q < 4; ++q) { f.x1[q] = live ? f.x1[q] : 0u; f.x2[q] = live ? f.x2[q] : 0u; }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag_v(f.x1), __builtin_bit_cast(VC, w0[u]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag_v(f.x2), __builtin_bit_cast(VC, w1[u]), acc, 0, 0, 0);
        }
        uint8_t* d0 = ydst + (pj >> 3) * 2048 + (pj & 7) * 2;
#pragma unroll
        for (int r = 0; r < MAXR_CR; ++r) {
            if (r > nrm1) break;                                // (wave-uniform)
            float v = acc[r] * (1.0f / (float)COMP_S);
            asm volatile("" : "+v"(v));                         // one rounded value for both halves (no fused convert of the product)
            const _Float16 y1 = (_Float16)v;
            const float y1f = (float)y1;
            *reinterpret_cast<_Float16*>(d0 + r * RECC_Y_BYTES) = (_Float16)((float)(COMP_S - 1) * y1f);
            *reinterpret_cast<_Float16*>(d0 + r * RECC_Y_BYTES + 1024) = (_Float16)fmaf((float)COMP_S, v - y1f, y1f);
        }
    }
}

// joint pairs out of cutoff range of EVERY point of a pass, from the per-ray records (AB[ray][slot].w = squared distance
// of the ray's sampled segment from the joint, pg_rayrec.hip): one ballot per joint slot over the two lane halves'
// joints x the rays of the pass (lane = ray; lanes past the last ray repeat it).  The same in every wave.
__device__ __forceinline__ int pass_far_mask_c(const uint8_t* ab, int nrm1, const float* far2h, int h, int pt) {
    const float* row = opaque_ptr(reinterpret_cast<const float*>(ab + min(pt, nrm1) * REC_AB_BYTES) + JH * h * 8 + 3);
    int m = 0;
#pragma unroll
    for (int jj = 0; jj < JH; ++jj)
        if (__builtin_amdgcn_ballot_w64(row[jj * 8] < far2h[jj]) == 0ull) m |= 1 << jj;
    return __builtin_amdgcn_readfirstlane(m);
}

// Per-ray LDS slots of this kernel (pg_layout.h SLOTC_*), one thread per (ray, joint):
//   AB[j] = (a = R_j o + t_j, b = R_j d)                       (core/encoders.py:8-37)
//   DTAB: e = normalize(b) per joint, rows (e, sin e, cos e, ..., sin 8e | cos 8e) in D-sequence order
//   (encoders.py:172-193), accurate sincosf: the view table is per ray, its cost is nothing
template <bool FC>
__device__ __forceinline__ void ray_table_c(const EvalArgs& a, float* rt, int r0, int nr) {
    for (int idx = threadIdx.x; idx < nr * J; idx += NTHR_C) {
        const int rr = idx / J, j = idx - rr * J;
        float* slot = rt + rr * SLOTC_FLOATS;
        const float4* sk = reinterpret_cast<const float4*>(a.skts + (long long)(r0 + rr) * a.pose_stride + j * 16);
        const float4 ra = sk[0], rb = sk[1], rc = sk[2];
        const float* ry = a.rays + (long long)(r0 + rr) * 11;
        const float ox = ry[0], oy = ry[1], oz = ry[2], dx = ry[3], dy = ry[4], dz = ry[5];
        float e[3];
        e[0] = fmaf(ra.z, dz, fmaf(ra.y, dy, ra.x * dx));
        e[1] = fmaf(rb.z, dz, fmaf(rb.y, dy, rb.x * dx));
        e[2] = fmaf(rc.z, dz, fmaf(rc.y, dy, rc.x * dx));
        float4* ab = reinterpret_cast<float4*>(slot + SLOTC_AB + j * 8);
        ab[0] = make_float4(fmaf(ra.z, oz, fmaf(ra.y, oy, fmaf(ra.x, ox, ra.w))),
                            fmaf(rb.z, oz, fmaf(rb.y, oy, fmaf(rb.x, ox, rb.w))),
                            fmaf(rc.z, oz, fmaf(rc.y, oy, fmaf(rc.x, ox, rc.w))), 0.0f);
        ab[1] = make_float4(e[0], e[1], e[2], 0.0f);
        const float den = fmaxf(sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), 1e-12f);
        const int h = j / JH, jj = j - h * JH;
        float* tab = slot + SLOTC_DTAB + h * DSEQ;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ev = e[c] / den;
            float s, co;
            sincosf(ev, &s, &co);
            float* m = tab + (jj * 3 + c) * 8;
            m[0] = ev;
#pragma unroll
            for (int f = 0; f < LD; ++f) {
                m[1 + 2 * f] = s;
                if (f < LD - 1) m[2 + 2 * f] = co;
                else tab[DSEQ_MAIN + jj * 3 + c] = co;
                const float s2 = 2.0f * s * co;
                co = (co - s) * (co + s);
                s = s2;
            }
        }
        if (jj == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) tab[DSEQ_MAIN + JH * 3 + k] = 0.0f;
        }
    }
    if (FC) {
        for (int idx = threadIdx.x; idx < nr * FC_CH; idx += NTHR_C) {
            const int rr = idx / FC_CH, k = idx - rr * FC_CH;
            const float cf = a.cams ? a.cams[r0 + rr] : -1.0f;
            const int ci = cf < 0.0f ? a.n_codes : min((int)cf, a.n_codes - 1);
            rt[rr * SLOTC_FLOATS + SLOTC_CODE + k] = a.codes[ci * FC_CH + k];
        }
    }
}

// TAPS = the debug taps of pg_stage_eval (dbg_stage 0, 7, 9, 10) compiled in: a separate instantiation,
// launched only when a dump is asked for -- the cold dump blocks (128 live values each) otherwise cost the
// production kernel 34 spilled registers, and every scratch reload drains the weight DMA (vmcnt(0))
// REC = the record variant (rays with >= FACT_MIN_S samples): (a, b) and the view layer's direction part arrive as
// per-ray records of pg_rayrec.hip (ray_records_c_kernel) and are fetched by LDS-DMA -- no table build, no barriers
// at the pass boundary, no 64-bit divisions, and the 648-wide view input (336 MFMAs, 11 chunks of weights and its
// per-point products and splits) becomes <= 32 MFMAs on the point's 24 cutoff weights.
// OC = the on-chip form of the record variant (one pose shared by the launch's rays, no frame codes -- BASELINE config 2):
// no per-ray records in HBM and no record kernel in front.  The (a, b) rows of a pass's rays are formed by the workgroup
// a pass ahead from the rays themselves (LDS-DMA of their ray_batch rows and first / last depths, the pose's bone rows
// kept in LDS), and the view layer's direction part Y by y_segment_c from joint-pair chunks of the weight stream.
template <bool FC, bool TAPS, bool REC, bool OC = false>
__global__ __launch_bounds__(NTHR_C, 1) void evalc_kernel(const EvalArgs a) {
    static_assert(!OC || (REC && !FC && !TAPS), "the on-chip variant is a form of the record variant without frame codes");
    constexpr int ABSZ = OC ? LDSO_ABSZ : LDS_ABC_BYTES;          // one (a, b) buffer
    constexpr int YOFF = OC ? LDSO_Y : LDSC_Y;                    // the Y image of the pass's rays
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* bias = reinterpret_cast<float*>(smem + (REC ? LDSC_BIAS : LDS_BIAS));
    float* cut = reinterpret_cast<float*>(smem + (REC ? LDSC_CUT : LDS_CUT));
    float* rtab = reinterpret_cast<float*>(smem + LDS_RTAB);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, pt = lane & 31;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    using ST = typename std::conditional<OC, StreamCRO, typename std::conditional<REC, StreamCR, StreamC>::type>::type;
    ST st{a.wstream, smem + LDS_RING, wave, lane, 0u, 0u, 0u, lds0 + LDS_RING, lane16};
    const float* sk_lds = reinterpret_cast<const float*>(smem + LDSO_SK);
    const float* stage = reinterpret_cast<const float*>(smem + LDSO_STAGE);
    // bias tile indices of this variant's table
    constexpr int TB_ALPHA = REC ? BTC_ALPHA : BT_ALPHA, TB_VIEWF = REC ? BTC_VIEWF : BT_VIEWF, TB_RGB = REC ? BTC_RGB : BT_RGB;

    if (REC) {
        for (int i = tid; i < BTC_COUNT * 32; i += NTHR_C) {
            const int tile = i >> 5;
            const int src = tile < BTC_ALPHA ? tile : tile == BTC_ALPHA ? BT_ALPHA : tile < BTC_RGB ? BT_VIEWF + (tile - BTC_VIEWF) : BT_RGB;
            bias[i] = a.bias[src * 32 + (i & 31)];
        }
    } else {
        for (int i = tid; i < BIAS_FLOATS; i += NTHR_C) bias[i] = a.bias[i];
    }
    // cutoff table with the sigmoid constants folded in (cutoff_weight_fast)
    const float tlv = a.tau_v * 1.4426950408889634f, tld = a.tau_d * 1.4426950408889634f;
    if (REC) {      // by joint SLOT (pg_layout.h slotc_joint), + the squared distance beyond which a cutoff weight is below 2^-24
        if (tid < 48) cut[tid] = -a.cutoff[(tid < J ? 0 : J) + slotc_joint_dev(tid < J ? tid : tid - J)] * (tid < J ? tlv : tld);
        else if (tid < 72) {        // (of both embedders: the on-chip variant's mask drops a pair's view-direction part too)
            const int jt = slotc_joint_dev(tid - 48);
            const float far = fmaxf(a.cutoff[jt] + 24.0f / tlv, a.cutoff[J + jt] + 24.0f / tld);
            cut[tid] = far * far;
        }
    } else if (tid < 48) cut[tid] = -a.cutoff[tid] * (tid < J ? tlv : tld);
    if (OC) {       // the pose's bone rows by joint slot; an all-zero Y image (pairs no pass has computed yet)
        for (int i = tid; i < J * 12; i += NTHR_C) reinterpret_cast<float*>(smem + LDSO_SK)[i] = a.skts[slotc_joint_dev(i / 12) * 16 + i % 12];
        for (int i = tid; i < MAXR_CR * RECC_Y_BYTES / 16; i += NTHR_C) reinterpret_cast<uint4*>(smem + LDSO_Y)[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    float s129 = (float)COMP_S;
    asm volatile("" : "+s"(s129));              // one SGPR for the whole kernel, not a literal per use
    if (!REC) st.start();

    // record variant: ray bookkeeping without a division per pass (a 64-bit divide is ~150 VALU instructions and
    // nothing overlaps them with one wave per SIMD): the pass's first point is sample `off0` of ray `r0`, and both
    // advance by the constant step of the persistent grid
    const long long step = (long long)PTS_C * gridDim.x;
    int dq = 0, dr = 0, r0 = 0, off0 = 0, abuf = 0;
    long long p0 = (long long)blockIdx.x * PTS_C;
    const uint8_t* rec_ab = reinterpret_cast<const uint8_t*>(a.rec_ab);
    if (REC) {
        dq = __builtin_amdgcn_readfirstlane((int)(step / a.S)); dr = __builtin_amdgcn_readfirstlane((int)(step % a.S));
        r0 = __builtin_amdgcn_readfirstlane((int)(p0 / a.S));
        off0 = __builtin_amdgcn_readfirstlane((int)(p0 - (long long)r0 * a.S));
        // (a, b) of the first pass's rays into buffer 0; every later pass finds its own fetched (OC: formed) a pass ahead
        if (OC) {
            lds_barrier();                      // the bone rows are in LDS
            if (tid < MAXR_CR * J) {
                const int k = tid / J, sl = tid - k * J;
                const long long ray = min((long long)r0 + k, (long long)a.n_rays - 1);
                ab_row(sk_lds + sl * 12, a.rays + ray * 11, a.z[ray * a.S], a.z[ray * a.S + a.S - 1],
                       reinterpret_cast<float4*>(smem + LDSC_AB + k * REC_AB_BYTES + sl * 32));
            }
        } else if ((int)blockIdx.x < a.n_iters && wave < LDS_ABC_BYTES / 1024)
            dma_piece_c(rec_ab + (long long)r0 * REC_AB_BYTES + wave * 1024, lds0 + LDSC_AB + wave * 1024, lane16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    }
    // record variant: joint pairs no point of a pass is in range of.  The first pass's mask here (the weight ring starts
    // with it), every later pass learns its own a pass ahead, from the (a, b) records of its rays.
    auto rays_of_pass = [&](long long p0_, int off0_) {         // index of the pass's last ray among its (<= 3) rays
        const int last_ = (int)max(0ll, min((long long)PTS_C - 1, a.n_points - 1 - p0_));
        const int t_ = off0_ + last_;
        return (t_ >= a.S) + (t_ >= 2 * a.S);
    };
    int gmask = 0;
    if (REC) {
#if !defined(PG_NO_FAR_SKIP)
        gmask = pass_far_mask_c(smem + LDSC_AB, rays_of_pass(p0, off0), cut + 2 * J + JH * h, h, pt);
        if (!a.far_skip) gmask = 0;
#endif
        st.start((uint32_t)gmask);
    }
    // record variant: the depth of the next pass's point, fetched a pass ahead (unconditional, clamped index)
    float nx_z = 0.0f;
    if (REC && (int)blockIdx.x < a.n_iters) nx_z = a.z[min(p0 + wave * 32 + pt, a.n_points - 1)];

#if defined(PG_STAMPS)
    unsigned long long stamps[14];
#endif
    for (int it = blockIdx.x; it < a.n_iters; it += gridDim.x) {
        PG_STAMP(0);
#if defined(PG_STAMPS)
        { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[10] = t_; }
#endif
        const long long gp = p0 + wave * 32 + pt;
        const bool valid = gp < a.n_points;
        const long long gpc = valid ? gp : a.n_points - 1;
        int myr, r0n = 0, off0n = 0, nrm1 = 0;
        const float* slot = nullptr;
        const float* ab;
        const float* tab = nullptr;
        if (REC) {
            // point i of the pass is sample off0 + i of ray r0, i.e. (S >= 64, 128 points) at most 2 rays on; points
            // past the end of the launch (last pass) take the last valid ray
            const int last = (int)min((long long)PTS_C - 1, a.n_points - 1 - p0);       // wave-uniform
            const int S1 = a.S, S2 = 2 * a.S;
            const int tl_ = off0 + last, ti = off0 + wave * 32 + pt;
            nrm1 = (tl_ >= S1) + (tl_ >= S2);
            myr = min((ti >= S1) + (ti >= S2), nrm1);
            ab = opaque_ptr(reinterpret_cast<const float*>(smem + LDSC_AB + abuf * ABSZ + myr * REC_AB_BYTES) + JH * h * 8);
            off0n = off0 + dr; r0n = r0 + dq;
            if (off0n >= a.S) { off0n -= a.S; ++r0n; }
        } else {
            const long long plast = min(p0 + PTS_C - 1, a.n_points - 1);
            r0 = (int)(p0 / a.S);
            const int nr = (int)(plast / a.S) - r0 + 1;
            lds_barrier();                          // previous pass is done with the table
            ray_table_c<FC>(a, rtab, r0, nr);
            lds_barrier();
            myr = (int)(gpc / a.S) - r0;
            slot = rtab + myr * SLOTC_FLOATS;
            ab = opaque_ptr(slot + SLOTC_AB + JH * h * 8);
            tab = opaque_ptr(slot + SLOTC_DTAB + h * DSEQ);
        }
        const float* cutv = opaque_ptr(cut + JH * h);
        const float* cutd = opaque_ptr(cut + J + JH * h);
        const float zz = REC ? nx_z : a.z[gpc];
        int wmask = 0;
        if (REC) {      // joint pairs out of cutoff range of the wave's 32 points (x_segment_cr)
#if !defined(PG_NO_FAR_SKIP)
            const float* far2 = opaque_ptr(cut + 2 * J + JH * h);
#pragma unroll
            for (int jj = 0; jj < JH; ++jj) {
                const float4 lo = *reinterpret_cast<const float4*>(ab + jj * 8);
                const float4 hi = *reinterpret_cast<const float4*>(ab + jj * 8 + 4);
                const float qx = fmaf(zz, hi.x, lo.x), qy = fmaf(zz, hi.y, lo.y), qz = fmaf(zz, hi.z, lo.z);
                if (__builtin_amdgcn_ballot_w64(qx * qx + qy * qy + qz * qz < far2[jj]) == 0ull) wmask |= 1 << jj;
            }
            wmask = __builtin_amdgcn_readfirstlane(a.far_skip ? wmask : 0);
#endif
        }
        // Behind layer 0's first chunk entry every wave is done with the previous pass: its Y records and the (a, b)
        // buffer of the pass before may be overwritten.  The 4 waves share the 16 pieces of each of this pass's
        // MAXR_CR Y records, waves 0..2 fetch a piece of the NEXT pass's (a, b); issued before the chunk's refill
        // pieces, all of it has landed -- and is visible to every wave -- one chunk entry on (in-order vmcnt).
        auto fetch_records = [&]() {
            if (!REC) return;
            if (OC) {
                // the NEXT pass's rays: 64 floats of their ray_batch rows from the first one on (wave 0) and their first
                // and last depths (wave 1), lane offsets clamped to the arrays; in LDS one chunk entry on
                const long long rn = min((long long)r0n, (long long)a.n_rays - 1);
                if (wave == 0) dma_dwords(a.rays, (uint32_t)(min(rn * 11 + lane, (long long)a.n_rays * 11 - 1) * 4), lds0 + LDSO_STAGE);
                else if (wave == 1) {
                    const long long ray = min(rn + min(lane >> 1, MAXR_CR - 1), (long long)a.n_rays - 1);
                    dma_dwords(a.z, (uint32_t)((ray * a.S + ((lane & 1) ? a.S - 1 : 0)) * 4), lds0 + LDSO_STAGE + 256);
                }
                return;
            }
            constexpr int PW = MAXR_CR * (RECC_Y_BYTES / 1024) / NWAVE_C;      // 12 pieces per wave
            constexpr int PR = RECC_Y_BYTES / 1024;                             // 16 pieces per record
            const uint8_t* ysrc = a.rec_y + (size_t)r0 * RECC_Y_BYTES;
#pragma unroll
            for (int k = 0; k < PW; ++k) {           // slots past the pass's last ray re-fetch that ray (an L2 hit, not HBM; no branch)
                const int q = wave * PW + k;
                dma_piece_c(ysrc + (size_t)min(q / PR, nrm1) * RECC_Y_BYTES + (q % PR) * 1024, lds0 + LDSC_Y + q * 1024, lane16);
            }
            if (wave < LDS_ABC_BYTES / 1024)
                dma_piece_c(rec_ab + (long long)min(r0n, a.n_rays - 1) * REC_AB_BYTES + wave * 1024,
                            lds0 + LDSC_AB + (abuf ^ 1) * LDS_ABC_BYTES + wave * 1024, lane16);
        };
        static_assert(MAXR_CR * 11 <= 64 && 2 * MAXR_CR <= 64, "the staged rays fit one dword DMA each");

        PG_STAMP(1);
        // first stream chunk of the segments behind layer 0 (pg_program.h C): where the ring bookkeeping is static (Stream::plain_ok)
        constexpr int CH = pgp::C::CH_HID;
        constexpr int C_L1 = REC ? pgp::C::CH_L0XR + (OC ? pgp::C::NPAIRJ : 0) : pgp::C::CH_L0X;
        constexpr int C_L5H = C_L1 + 4 * CH, C_L6 = C_L5H + CH + (REC ? pgp::C::CH_L0XR : pgp::C::CH_L0X), C_AV = C_L6 + 2 * CH;
        static_assert(!OC || C_L5H + CH == pgp::C::C_L5XR_OC, "chunk bases follow pg_program.h");
        static_assert(!REC || OC || C_L5H + CH == pgp::C::C_L5XR, "chunk bases follow pg_program.h");
        f32x16 accA[NT], accB[NT];
        // ---- layer 0: K = 432 generated on the fly ----
#pragma unroll
        for (int o = 0; o < NT; ++o) accA[o] = load_bias(bias, BT_LAYER0 + o, h);
        if constexpr (REC) x_segment_cr(accA, st, ab, zz, cutv, tlv, s129, wmask, gmask, fetch_records);
        else x_segment_c(accA, st, ab, zz, cutv, tlv, s129, fetch_records);
        if (TAPS && a.dbg && a.dbg_stage == 0 && valid) {
#pragma unroll
            for (int o = 0; o < NT; ++o)
#pragma unroll
                for (int r = 0; r < 16; ++r) a.dbg[gp * W + 32 * o + rho(r, h)] = accA[o][r];
        }
        if constexpr (OC)   // the view layer's direction part of this pass's rays, for the joint pairs in range
            y_segment_c(st, gmask, smem + LDSC_AB + abuf * ABSZ, smem + LDSO_Y, nrm1, wave, lane, s129);
        PG_STAMP(2);
        // ---- layers 1..4 (ping-pong between the two accumulator sets) ----
        auto srcA = [&](int u, int e) { return accA[u >> 1][8 * (u & 1) + e]; };
        auto srcB = [&](int u, int e) { return accB[u >> 1][8 * (u & 1) + e]; };
#pragma unroll
        for (int o = 0; o < NT; ++o) accB[o] = load_bias(bias, BT_LAYER0 + 1 * NT + o, h);
        segment_c<NT, HU, true>(accB, st, srcA, s129, C_L1);
        if constexpr (OC) {
            // The NEXT pass's (a, b) rows from the staged rays.  The fetch was issued in layer 0's first chunk, possibly
            // between that chunk's refill pieces: the counted wait of the SECOND entry behind it covers it, and that
            // entry's barrier makes the other wave's share visible -- layer 1's eight entries lie in between.  18 of
            // each wave's lanes take one (ray, joint slot) each.
            int lane_p = lane;
            asm volatile("" : "+v"(lane_p));        // (addresses derived from it are formed here, not ahead of the pass loop)
            if (lane_p < MAXR_CR * J / NWAVE_C) {
                const int item = wave * (MAXR_CR * J / NWAVE_C) + lane_p;
                const int k = item / J, sl = item - k * J;
                ab_row(sk_lds + sl * 12, stage + 11 * k, stage[64 + 2 * k], stage[64 + 2 * k + 1],
                       reinterpret_cast<float4*>(smem + LDSC_AB + (abuf ^ 1) * ABSZ + k * REC_AB_BYTES + sl * 32));
            }
#if defined(PG_DEBUG_Y)     // diagnosis build: the Y image and the (a, b) rows of workgroup 0's first pass
            if (a.dbg && a.dbg_stage == 98 && blockIdx.x == 0 && it == 0) {
                for (int i = tid; i < MAXR_CR * RECC_Y_BYTES / 4; i += NTHR_C) a.dbg[i] = reinterpret_cast<const float*>(smem + LDSO_Y)[i];
                for (int i = tid; i < LDSO_ABSZ / 4; i += NTHR_C) a.dbg[MAXR_CR * RECC_Y_BYTES / 4 + i] = reinterpret_cast<const float*>(smem + LDSC_AB + abuf * ABSZ)[i];
                if (tid == 0) { a.dbg[13000] = (float)nrm1; a.dbg[13001] = (float)gmask; }
            }
#endif
        }
#pragma unroll
        for (int o = 0; o < NT; ++o) accA[o] = load_bias(bias, BT_LAYER0 + 2 * NT + o, h);
        segment_c<NT, HU, true>(accA, st, srcB, s129, C_L1 + CH);
#pragma unroll
        for (int o = 0; o < NT; ++o) accB[o] = load_bias(bias, BT_LAYER0 + 3 * NT + o, h);
        segment_c<NT, HU, true>(accB, st, srcA, s129, C_L1 + 2 * CH);
#pragma unroll
        for (int o = 0; o < NT; ++o) accA[o] = load_bias(bias, BT_LAYER0 + 4 * NT + o, h);
        segment_c<NT, HU, true>(accA, st, srcB, s129, C_L1 + 3 * CH);
        PG_STAMP(3);
        // ---- layer 5: [x(432), h4(256)] -> 256 (skip connection, nerf.py:99-101) ----
#pragma unroll
        for (int o = 0; o < NT; ++o) accB[o] = load_bias(bias, BT_LAYER0 + 5 * NT + o, h);
        segment_c<NT, HU, true>(accB, st, srcA, s129, C_L5H);
        PG_STAMP(4);
        if constexpr (REC) x_segment_cr(accB, st, ab, zz, cutv, tlv, s129, wmask, gmask, NoHook());
        else x_segment_c(accB, st, ab, zz, cutv, tlv, s129);
        PG_STAMP(5);
        // ---- layers 6, 7 ----
#pragma unroll
        for (int o = 0; o < NT; ++o) accA[o] = load_bias(bias, BT_LAYER0 + 6 * NT + o, h);
        segment_c<NT, HU, true>(accA, st, srcB, s129, C_L6);
#pragma unroll
        for (int o = 0; o < NT; ++o) accB[o] = load_bias(bias, BT_LAYER0 + 7 * NT + o, h);
        segment_c<NT, HU, true>(accB, st, srcA, s129, C_L6 + CH);
        if (TAPS && a.dbg && a.dbg_stage == 7 && valid) {
#pragma unroll
            for (int i = 0; i < HSEQ; ++i) a.dbg[gp * W + hseq_channel(i, h)] = fmaxf(accB[i >> 4][i & 15], 0.0f);
        }
        PG_STAMP(6);
        int gmask_n = 0;
        if (REC) {      // the NEXT pass's mask, from its (a, b) records (in LDS since this pass's second chunk entry): the ring's
                        // prefetch pointer wraps to the head of the stream within the next segments and must know it by then
#if !defined(PG_NO_FAR_SKIP)
            gmask_n = pass_far_mask_c(smem + LDSC_AB + (abuf ^ 1) * ABSZ, rays_of_pass(p0 + step, off0n),
                                      opaque_ptr(cut + 2 * J + JH * h), h, pt);
            if (!a.far_skip) gmask_n = 0;
#endif
            st.nx_mask = (uint32_t)gmask_n;
        }
        // ---- sigma head and the view layer's trunk part in one segment of 1 + 4 out tiles: feature_linear
        // has no activation and is folded into the view weights on the host (NetTensors::fold) ----
        f32x16 av[NTV + 1];
        av[0] = load_bias(bias, TB_ALPHA, h);
#pragma unroll
        for (int o = 0; o < NTV; ++o) av[1 + o] = load_bias(bias, TB_VIEWF + o, h);
        segment_c<NTV + 1, HU, true>(av, st, srcB, s129, C_AV);
        const float sigma = av[0][0];
        PG_STAMP(7);
        // ---- view directions: per-ray sin/cos table in LDS times the per-point cutoff weight ----
        {
            float wd[JH];
#pragma unroll
            for (int jj = 0; jj < JH; ++jj) {
                const float4 lo = *reinterpret_cast<const float4*>(ab + jj * 8);
                const float4 hi = *reinterpret_cast<const float4*>(ab + jj * 8 + 4);
                const float qx = fmaf(zz, hi.x, lo.x), qy = fmaf(zz, hi.y, lo.y), qz = fmaf(zz, hi.z, lo.z);
                wd[jj] = cutoff_weight_fast(__builtin_amdgcn_sqrtf(qx * qx + qy * qy + qz * qz), tld, cutd[jj]);
                // a pair left out of the pass has weights below 2^-24 in every point: exactly zero instead, so that what the
                // on-chip form's Y image still holds for the pair from an earlier pass can never reach a result
                if (OC && ((gmask >> jj) & 1)) wd[jj] = 0.0f;
            }
            if (TAPS && a.dbg && a.dbg_stage == 10 && valid) {
#pragma unroll
                for (int jj = 0; jj < JH; ++jj) a.dbg[gp * W + JH * h + jj] = wd[jj];
            }
            if constexpr (REC) {
                // second stage of the factorised view layer: av += Yc[ray] w, the cutoff weights split like an
                // activation; k-unit 0 = joints 12 h + 0..7, k-unit 1 = joints 12 h + 8..11 and the frame code
                // (weight 1, lane half 0) -- vyc_slot_joint.  A wave's 32 points lie on <= 2 rays: one round per
                // ray with the other ray's points zeroed.
                float w1[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) w1[e] = e < JH - 8 ? wd[8 + e] : 0.0f;
                if (FC && h == 0) w1[JH - 8] = 1.0f;
                const FragC f0 = frag_of(wd, s129), f1 = frag_of(w1, s129);
                const int ra = __builtin_amdgcn_readfirstlane(myr);
                const int rb = __builtin_amdgcn_readlane(myr, 63);
                for (int ray = ra; ray <= rb; ++ray) {
                    FragC g0, g1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        g0.x1[q] = myr == ray ? f0.x1[q] : 0u; g0.x2[q] = myr == ray ? f0.x2[q] : 0u;
                        g1.x1[q] = myr == ray ? f1.x1[q] : 0u; g1.x2[q] = myr == ray ? f1.x2[q] : 0u;
                    }
                    const uint8_t* yb = smem + YOFF + ray * RECC_Y_BYTES + lane * 16;
#pragma unroll
                    for (int o = 0; o < NTV; ++o) {
                        const uint4 a00 = *reinterpret_cast<const uint4*>(yb + (o * 4 + 0) * 1024);
                        const uint4 a01 = *reinterpret_cast<const uint4*>(yb + (o * 4 + 1) * 1024);
                        const uint4 a10 = *reinterpret_cast<const uint4*>(yb + (o * 4 + 2) * 1024);
                        const uint4 a11 = *reinterpret_cast<const uint4*>(yb + (o * 4 + 3) * 1024);
                        av[1 + o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, a00), frag_v(g0.x1), av[1 + o], 0, 0, 0);
                        av[1 + o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, a01), frag_v(g0.x2), av[1 + o], 0, 0, 0);
                        av[1 + o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, a10), frag_v(g1.x1), av[1 + o], 0, 0, 0);
                        av[1 + o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(VC, a11), frag_v(g1.x2), av[1 + o], 0, 0, 0);
                    }
                }
            } else {
                const float* code = slot + SLOTC_CODE + 8 * h;
                auto srcD = [&](int u, int e) {
                    if (u == pgp::C::DU) return code[e];                   // frame code rides as one more unit
                    const int k = u < JH * 3 ? u / 3 : (8 * (u - JH * 3) + e) / 3;
                    return k < JH ? tab[u * 8 + e] * wd[k] : 0.0f;
                };
                segment_c<NTV, pgp::C::DU + (FC ? 1 : 0), false>(av + 1, st, srcD, s129);
            }
        }
        if (TAPS && a.dbg && a.dbg_stage == 9 && valid) {
#pragma unroll
            for (int i = 0; i < VW / 2; ++i) a.dbg[gp * W + hseq_channel(i, h)] = fmaxf(av[1 + (i >> 4)][i & 15], 0.0f);
        }
        PG_STAMP(8);
        // ---- rgb head ----
        if (REC) nx_z = a.z[min(p0 + step + wave * 32 + pt, a.n_points - 1)];       // in flight through the rgb head and the pass boundary
        f32x16 accr = load_bias(bias, TB_RGB, h);
        auto srcV = [&](int u, int e) { return av[1 + (u >> 1)][8 * (u & 1) + e]; };
#if defined(PG_ABL_NORGB)       // timing ablation only (wrong results): the rgb head's chunk is entered and refilled, its work skipped
        { PairPipe<NSC> pq; st.enter_split(); for (int i = 0; i < ST::PER; ++i) piece_c(pq, st, i); asm volatile("" :: "v"(av[1][0]), "v"(av[2][0]), "v"(av[3][0]), "v"(av[4][0])); }
#else
        segment_c<1, HU / 2, true>(&accr, st, srcV, s129);
#endif
#if defined(PG_STAMPS_RGB)
        PG_STAMP(9);
#endif
#if defined(PG_ABL_NOSTORE)     // timing ablation only: results kept live, not stored
        asm volatile("" :: "v"(accr[0]), "v"(accr[1]), "v"(accr[2]), "v"(sigma));
#else
        if (valid && h == 0)
            *reinterpret_cast<float4*>(a.raw + gp * 4) = make_float4(accr[0], accr[1], accr[2], sigma);
#endif
        if (REC) { abuf ^= 1; r0 = r0n; off0 = off0n; gmask = gmask_n; }
        p0 += step;
#if !defined(PG_STAMPS_RGB)
        PG_STAMP(9);
#endif
#if defined(PG_STAMPS)
        { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[11] = t_; }
        if (a.dbg && a.dbg_stage == 99 && lane == 0 && it < 64) {
            reinterpret_cast<unsigned long long*>(a.dbg)[((long long)it * 8 + wave) * 16 + 12] = stamps[10];
            reinterpret_cast<unsigned long long*>(a.dbg)[((long long)it * 8 + wave) * 16 + 13] = stamps[11];
            for (int k = 0; k < 10; ++k) reinterpret_cast<unsigned long long*>(a.dbg)[((long long)it * 8 + wave) * 16 + k] = stamps[k];
            reinterpret_cast<unsigned long long*>(a.dbg)[((long long)it * 8 + wave) * 16 + 10] = st.t_vm;
            reinterpret_cast<unsigned long long*>(a.dbg)[((long long)it * 8 + wave) * 16 + 11] = st.t_bar;
            st.t_vm = 0; st.t_bar = 0;
        }
#endif
    }
    st.drain();
}

template <bool FC, bool TAPS, bool REC, bool OC = false>
static hipError_t launch_evalc(const EvalArgs& a, int grid, hipStream_t stream) {
    auto k = evalc_kernel<FC, TAPS, REC, OC>;
    constexpr int lds = OC ? LDS_TOTAL_CO : REC ? LDS_TOTAL_CR : LDS_TOTAL_C;
    static std::atomic<unsigned long long> attr_done{0};       // per device (pg_device.h)
    const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(k), lds, attr_done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHR_C), lds, stream, a);
    return hipGetLastError();
}

template <bool REC>
static hipError_t dispatch_evalc(const EvalArgs& a, int framecode, int grid, hipStream_t s) {
    if (a.dbg && a.dbg_stage != 99) return framecode ? launch_evalc<true, true, REC>(a, grid, s) : launch_evalc<false, true, REC>(a, grid, s);
    return framecode ? launch_evalc<true, false, REC>(a, grid, s) : launch_evalc<false, false, REC>(a, grid, s);
}

}  // namespace pgd

// needs S >= pgl::COMP_MIN_S, rays (no explicit points) and the shape-C stream (pg_pack.cpp); rec: S >= pgl::FACT_MIN_S,
// the record variant of the stream (pack_stream(..., rec = true)) and the records of ray_records_c_kernel in
// a.rec_ab / a.rec_y; rec == 2: its on-chip form (no records; the on-chip stream, a.pose_stride == 0, no frame codes, no
// debug taps)
extern "C" int pg_launch_evalc(const pgd::EvalArgs* a, int framecode, int rec, int grid, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (rec == 2) {
        if (framecode || a->pose_stride != 0 || (a->dbg && a->dbg_stage != 99 && a->dbg_stage != 98)) return (int)hipErrorInvalidValue;
        return (int)pgd::launch_evalc<false, false, true, true>(*a, grid, s);
    }
    return (int)(rec ? pgd::dispatch_evalc<true>(*a, framecode, grid, s) : pgd::dispatch_evalc<false>(*a, framecode, grid, s));
}

extern "C" int pg_evalc_points_per_pass(void) { return pgd::PTS_C; }
