// pg_eval16r.hip -- fused bone-relative embedding + NeRF MLP on v_mfma_f32_16x16x32 (bf16 / fp16 operands, fp32
// accumulate) for rays with >= 64 samples: the kernel behind PG_PREC_BF16 / PG_PREC_FP16 renders.
//
// Replaces RayCaster.encode_inputs + run_network + NeRF.forward for one net (reference core/raycasters.py:476-577,
// core/networks/nerf.py:90-148, core/encoders.py, core/cutoff_embedder.py) on n*S points p = o + d*z.
//
// Structure (pg_layout.h "small tile"): 8 waves x 32 points per workgroup pass, a wave keeps the activations of its
// points in registers through all layers; A = 16 out channels x 32 k (weights, streamed L2 -> LDS ring by LDS-DMA and
// shared by all waves), B = 32 k x 16 points; a wave's 32 points are two column tiles, so every A fragment read
// from the ring feeds two MFMAs; lane group g = lane>>4 generates the density embedding of joints 6g..6g+5 for the
// wave's points col and col+16 on the fly (never stored; regenerated for the skip layer).
//
// Two things set it apart from its 32x32x16 predecessor (pg_eval16.hip, now the direct-view kernel for short rays,
// explicit points and position noise):
//   * the MFMA shape: MI355X holds a ~15 % higher clock on 16x16x32 at equal cycles per FLOP (measured on this
//     pool: 2.15 vs 1.86 PFLOP/s sustained with LDS-fed operands; MI355X_MICROARCH.md, DVFS give-back 7) -- the
//     kernel is power-limited (all-zero weights run the same instruction stream 19 % faster), so this is the lever;
//   * everything that depends on the RAY only -- the bone-local ray (a_j, b_j) and the view layer's direction part
//     Y[ray][j] -- arrives as per-ray records computed by pg_rayrec.hip and is fetched with a few LDS-DMA pieces per
//     pass: no table build, no Y stage, no extra barriers and no Y-stage weight traffic at the pass boundary (13 %
//     of a pass before), and the rgb head shares the last chunk of the alpha / view segment (one chunk entry less).
// Round 4: the joint slots are permuted into LIMBS (pg_layout.h PERM16) and the limbs a wave's points -- or a whole pass --
// are out of cutoff range of are left out (x_segment16: wave-level skip; Stream MASK_NX: the chunk is not in the pass's
// chunk sequence); and for one pose per call without frame codes the kernel needs no records at all (OC, the on-chip
// variant: rows formed a pass ahead from LDS-DMA'd rays, Y by y_segment16 from limb chunks of direction weights).
#ifndef PG_PREFETCH
#define PG_PREFETCH 1
#endif
#ifndef PG_PIPE_H
#define PG_PIPE_H 4           // A-pipe register sets in the hidden layers and heads: reads three units ahead (-0.3 % against two; five: the same)
#endif
#include <type_traits>

#include "pg_eval16_common.h"

// cache policy of the per-ray record fetches (streamed once; must not evict the weight stream from L2)
#ifndef PG_REC_POLICY
#define PG_REC_POLICY ""
#endif

namespace pgd {

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <typename V> struct Op16;
template <> struct Op16<bf16x8> {
    static __device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Op16<f16x8> {
    static __device__ __forceinline__ f32x4 mfma(f16x8 a, f16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

using StreamR = Stream<NWAVE, pgp::R::NCHUNK, PG_DMA_WAVES, pgp::R::NLIMB, 0, pgp::R::C_L5X>;
// on-chip variant (no per-ray records in HBM): the limb chunks of the view layer's direction weights sit behind layer 0
using StreamRO = Stream<NWAVE, pgp::R::NCHUNK_OC, PG_DMA_WAVES, pgp::R::NLIMB, 0, pgp::R::C_L5X_OC, pgp::R::C_Y>;
// where the ring bookkeeping of a chunk entry is static (Stream::plain_ok): not where a maskable chunk or the wrap is within reach
static_assert(StreamRO::plain_ok(14) && StreamRO::plain_ok(30) && !StreamRO::plain_ok(31) && StreamRO::plain_ok(42) && StreamRO::plain_ok(49) &&
              !StreamRO::plain_ok(50) && StreamRO::plain_ok(13) && !StreamRO::plain_ok(5) && !StreamRO::plain_ok(2), "on-chip stream");
static_assert(StreamR::plain_ok(8) && StreamR::plain_ok(24) && !StreamR::plain_ok(25) && StreamR::plain_ok(36) && StreamR::plain_ok(43) && !StreamR::plain_ok(44),
              "record stream");

// LDS carve-up of this kernel (bytes)
constexpr int LDSR_RING = 0;
constexpr int LDSR_BIAS = PG_RING_SLOTS * CHUNK_BYTES;          // BIAS16_FLOATS floats
constexpr int LDSR_CUT = LDSR_BIAS + BIAS16_FLOATS * 4;         // 72 floats, by joint SLOT: cutoff constants of embed_fn, of embeddirs_fn, far^2
constexpr int LDSR_AB = LDSR_CUT + 72 * 4;                      // two buffers of LDS_AB_BYTES: this pass / the next
constexpr int LDSR_Y = LDSR_AB + 2 * LDS_AB_BYTES;              // MAXR_F rays x REC_Y_BYTES
constexpr int LDSR_TOTAL = LDSR_Y + MAXR_F * REC_Y_BYTES;
static_assert(LDSR_BIAS % 16 == 0 && LDSR_CUT % 16 == 0 && LDSR_AB % 16 == 0 && LDSR_Y % 16 == 0, "LDS alignment");
static_assert(LDSR_TOTAL <= 160 * 1024, "LDS budget of one CU");
// on-chip variant: + the pose's bone rows by joint slot (24 x 12 floats) and a staging area for the next pass's rays
// (64 floats of ray_batch rows, 64 floats of first / last depths)
// with frame codes: + 64 floats (one DMA of a wave; 5 used) of the next pass's frame-code indices in the staging area, and the code's part of the view
// layer Yc[code][128 out] (fp32 rows of a host-made table) of this pass's / the next pass's rays
constexpr int LDSR_SK = LDSR_TOTAL;
constexpr int LDSR_STAGE = LDSR_SK + J * 12 * 4;
constexpr int LDSR_YC = LDSR_STAGE + 768;
constexpr int LDSR_YC_BYTES = MAXR_F * VW * 4;
constexpr int LDSR_TOTAL_OC = LDSR_YC + 2 * LDSR_YC_BYTES;
static_assert(LDSR_SK % 16 == 0 && LDSR_STAGE % 16 == 0 && LDSR_YC % 16 == 0 && LDSR_TOTAL_OC <= 160 * 1024, "LDS budget of one CU (on-chip variant)");

__device__ __forceinline__ f32x4 load_bias16(const float* bias, int tile, int g) {
    const float4 b = *reinterpret_cast<const float4*>(bias + tile * 16 + 4 * g);
    f32x4 r = {b.x, b.y, b.z, b.w};
    return r;
}

// out tiles 2u (lo) and 2u+1 (hi) of one column tile -> k-unit u of the next layer
template <typename V>
__device__ __forceinline__ V relu_pack16(const f32x4& lo, const f32x4& hi, bool relu) {
    const float t[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    V f = Op<V>::cvt(t);
    return relu ? relu16<V>(f) : f;
}

// one LDS-DMA piece (1 KiB, lane-linear) from a wave-uniform source to a wave-uniform LDS address; counted by the
// chunk entries' vmcnt like the ring's own pieces (in-order completion: anything issued before a chunk's refill
// pieces has landed by the next entry, and is visible to every wave behind that entry's barrier)
__device__ __forceinline__ void dma_piece(const uint8_t* src, uint32_t lds_dst, uint32_t lane16) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" PG_REC_POLICY :: "s"(lds_dst), "v"(lane16), "s"(src) : "memory");
}

// one k-unit (two B fragments, one per column tile) against NO out tiles of a k-major segment
template <typename V, int NO, int T, int NS, typename ST>
__device__ __forceinline__ void mma_row16(f32x4 (*acc)[2], APipe<V, NS>& p, ST& st, int uu, V b0, V b1) {
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        const V av = next_a<V, T, true, NS>(p, st, uu * NO + o);
        acc[o][0] = Op16<V>::mfma(av, b0, acc[o][0]);
        acc[o][1] = Op16<V>::mfma(av, b1, acc[o][1]);
    }
}

// acc += W[:, x-columns] x for the wave's two column tiles; X16 sequence of pg_layout.h: per joint slot jj of the lane
// group two units of cutoff-weighted values (one chunk of the stream = the four joints of one LIMB), then three units
// of directions.  `farmask` bit jj (wave-uniform): limb jj is out of cutoff range of all 32 points of the wave --
// every value of its two units is w_j(point) x (...) with w_j = 1 / (1 + 2^t), t >= 24, i.e. below 6e-8 (the
// reference's 1 - sigmoid rounds to exactly 0 there) -- so the wave only keeps the ring going for that chunk (entry,
// its share of the refill) and runs neither the embedding math nor the 64 MFMAs.  At tau = 79.6 (a trained model,
// cutoff_embedder.py:181-183) 88 % of the limb chunks of a frame are skipped: a point is near 1-2 limbs.
// `gmask` bit jj (the same in every wave of the workgroup): NO point of the pass is in range of limb jj; its chunk is
// not in this pass's chunk sequence at all (Stream MASK_NX: never fetched, no entry, no barrier).
// `hook` runs once behind the segment's first chunk entry (the per-pass record fetch of layer 0 hangs there).
template <typename V, typename ST, typename HOOK>
__device__ __forceinline__ void x_segment16(f32x4 (*acc)[2], ST& st, const float* abp0, const float* abp1, float z0, float z1,
                                            const float* cutv, float tau, int farmask, int gmask, HOOK hook) {
    APipeX<V> p;
    constexpr int T = XU16 * NT16;
    static_assert(2 * NT16 == pgp::R::UPC, "a limb's two unit rows are exactly one chunk of the stream");
    const QFromAB q0{abp0, z0}, q1{abp1, z1};
    bool hooked = false;                    // wave-uniform
#pragma clang loop unroll(full)
    for (int jj = 0; jj < JG; ++jj) {
        if ((gmask >> jj) & 1) continue;
        if ((farmask >> jj) & 1) {
            st.enter_split();
            if (!hooked) { hook(); hooked = true; }
#pragma unroll
            for (int i = 0; i < ST::PER; ++i) st.piece(i);
            continue;
        }
        float x0[18], x1[18];
        float qx, qy, qz;
        q0(jj, qx, qy, qz);
        joint_values_q<true>(qx, qy, qz, tau, cutv[jj], x0);
        q1(jj, qx, qy, qz);
        joint_values_q<true>(qx, qy, qz, tau, cutv[jj], x1);
        x0[15] = x1[15] = 0.0f;             // (the directions x[15..17] have units of their own)
        mma_row16<V, NT16, T>(acc, p, st, 2 * jj, Op<V>::cvt(x0), Op<V>::cvt(x1));
        if (!hooked) { hook(); hooked = true; }
        mma_row16<V, NT16, T>(acc, p, st, 2 * jj + 1, Op<V>::cvt(x0 + 8), Op<V>::cvt(x1 + 8));
    }
    // r = q / max(|q|, 1e-12) of every joint (VecNormEncoder on the bone-local position, encoders.py:172-193): not
    // cutoff-weighted, always there
    auto dir = [](const QFromAB& q, int jj, float* r3) {
        float qx, qy, qz;
        q(jj, qx, qy, qz);
        const float rinv = __builtin_amdgcn_rsqf(fmaxf(qx * qx + qy * qy + qz * qz, 1e-24f));
        r3[0] = qx * rinv; r3[1] = qy * rinv; r3[2] = qz * rinv;
    };
#pragma clang loop unroll(full)
    for (int pr = 0; pr < JG / 2; ++pr) {
        float v0[8], v1[8];
        dir(q0, 2 * pr, v0); dir(q0, 2 * pr + 1, v0 + 3);
        dir(q1, 2 * pr, v1); dir(q1, 2 * pr + 1, v1 + 3);
        v0[6] = v0[7] = v1[6] = v1[7] = 0.0f;
        mma_row16<V, NT16, T>(acc, p, st, XV16 + pr, Op<V>::cvt(v0), Op<V>::cvt(v1));
        if (!hooked) { hook(); hooked = true; }
    }
}

// On-chip variant: the view layer's direction part Y[ray][joint][out] = sum_k W_vd[out, (joint, k)] T[ray][joint][k]
// (pg_layout.h "factorised view layer") for the limbs in range of the pass, straight into the LDS image the second
// stage reads (y_apply16) -- no per-ray record in HBM.  One stream chunk per limb: 32 A fragments [joint slot 6 g' +
// jj][out tile t]; wave w takes joint group g' = w & 3 and out tiles 4 (w >> 2) .. + 3, with the pass's rays as the 16
// MFMA columns: B = the 27 view values of (ray, joint), from the record's b = R_j d: e = b / |b|, rows (e, sin e,
// cos e, .., sin 8 e, cos 8 e) per component (encoders.py:172-193, cutoff_embedder.py:45-46), hardware sin / cos.
// Limbs out of range of the whole pass keep whatever an earlier pass left (zeros at first): the second stage multiplies
// them by exactly zero.
// the frame code's part of the view layer of the pass's rays: Yc rows (fp32, staged in LDS) -> element 6 of lane group 0 of
// the Y image's A fragments (vy16_slot_joint: the pseudo joint JC), every thread of the workgroup two values per ray pair
template <typename V>
__device__ __forceinline__ void y_code16(const float* yc, uint8_t* ylds, int nrm1, int tid) {
    using E = typename Op<V>::E;
    for (int idx = tid; idx < (nrm1 + 1) * VW; idx += NTHR) {
        const int ray = idx / VW, o = idx - ray * VW;
        *reinterpret_cast<E*>(ylds + ray * REC_Y_BYTES + (o >> 4) * 1024 + (o & 15) * 16 + 2 * JG) = (E)yc[idx];
    }
}

template <typename V, typename ST>
__device__ __forceinline__ void y_segment16(ST& st, int gmask, const uint8_t* ab, uint8_t* ylds, int nrm1, int wave, int lane) {
    using E = typename Op<V>::E;
    const int g = lane >> 4, col = lane & 15;
    const int gj = wave & 3, t0 = 4 * (wave >> 2);
    const uint8_t* row = ab + min(col, nrm1) * REC_AB_BYTES + (JG * gj) * 32 + 16;         // b rows of this wave's joint slots
#pragma clang loop unroll(full)
    for (int jj = 0; jj < JG; ++jj) {
        if ((gmask >> jj) & 1) continue;
        st.enter_split();
#pragma unroll
        for (int i = 0; i < ST::PER; ++i) st.piece(i);
        const float4 b = *reinterpret_cast<const float4*>(row + jj * 32);
        const float inv = __builtin_amdgcn_rsqf(fmaxf(b.x * b.x + b.y * b.y + b.z * b.z, 1e-24f)) * 0.15915494309189535f;
        const float rx = b.x * inv, ry = b.y * inv, rz = b.z * inv;       // e in revolutions
        float tv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 8 * g + i;                    // 0 .. 31 (27 used): component c = k / 9, row r9 = k % 9
            const int c = (k >= 9) + (k >= 18), r9 = k - 9 * c;
            const float ec = c == 0 ? rx : (c == 1 ? ry : rz);
            const int f = (r9 - 1) >> 1;
            const float ang = ec * (float)(1 << (f < 0 ? 0 : f)) + (((r9 - 1) & 1) ? 0.25f : 0.0f);
            const float sv = __builtin_amdgcn_sinf(ang);
            tv[i] = k >= 27 ? 0.0f : (r9 == 0 ? ec * 6.283185307179586f : sv);
        }
        const V bf = Op<V>::cvt(tv);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = t0 + q;
            const V av = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(st.at(0, (gj * NTV16 + t) * UNIT_BYTES)));
            f32x4 c4 = {0.0f, 0.0f, 0.0f, 0.0f};
            c4 = Op16<V>::mfma(av, bf, c4);
            if (col <= nrm1) {
                E* dst = reinterpret_cast<E*>(ylds + col * REC_Y_BYTES + t * 1024 + (gj * 16 + 4 * g) * 16 + 2 * jj);
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r * 8] = (E)c4[r];
            }
        }
    }
}

// Limbs out of cutoff range of EVERY point of a pass, from the per-ray records alone (so that every wave gets the same
// answer, a pass ahead): AB[ray][slot].w = the squared distance of the ray's sampled segment from the joint
// (pg_rayrec.hip).  `ab` = the pass's (a, b) buffer in LDS, nrm1 = its last ray; one ballot per joint slot over the
// four lane groups' joints x the rays of the pass (lane column = ray; columns past the last ray repeat it).
__device__ __forceinline__ int pass_far_mask(const uint8_t* ab, int nrm1, const float* far2g, int g, int col) {
    const float* row = opaque_ptr(reinterpret_cast<const float*>(ab + min(col, nrm1) * REC_AB_BYTES) + JG * g * 8 + 3);
    int m = 0;
#pragma unroll
    for (int jj = 0; jj < JG; ++jj)
        if (__builtin_amdgcn_ballot_w64(row[jj * 8] < far2g[jj]) == 0ull) m |= 1 << jj;
    return __builtin_amdgcn_readfirstlane(m);
}

// The bias tile of the next out tile, fetched beside the ring pipe: fetch() in front of a next_a() whose successor
// still runs before take() (then the read has landed, see lds_async128); take() behind a next_a().  The tile index
// goes into the instruction's offset field and the base is made opaque per layer: with the addresses visible hipcc
// computes all 138 of them ahead of the pass loop and spills them (a scratch reload per tile, each waiting vmcnt(0)).
struct BiasPipe {
    unsigned base;          // LDS byte address of the lane group's rows of the segment's first bias tile
    a128 r;
    __device__ __forceinline__ BiasPipe(unsigned bbase, int tile0) : base(bbase + tile0 * 64) { asm volatile("" : "+v"(base)); }
    __device__ __forceinline__ void fetch(int k) {          // tile0 + k, k < 16 (constant after unrolling)
        switch (k) {
#define PG_BIAS_CASE(K) case K: asm volatile("ds_read_b128 %0, %1 offset:0+" #K "*64" : "=v"(r) : "v"(base)); break;
            PG_BIAS_CASE(0) PG_BIAS_CASE(1) PG_BIAS_CASE(2) PG_BIAS_CASE(3) PG_BIAS_CASE(4) PG_BIAS_CASE(5) PG_BIAS_CASE(6) PG_BIAS_CASE(7)
            PG_BIAS_CASE(8) PG_BIAS_CASE(9) PG_BIAS_CASE(10) PG_BIAS_CASE(11) PG_BIAS_CASE(12) PG_BIAS_CASE(13) PG_BIAS_CASE(14) PG_BIAS_CASE(15)
#undef PG_BIAS_CASE
            default: __builtin_unreachable();
        }
    }
    __device__ __forceinline__ f32x4 take() { lds_landed(r); return __builtin_bit_cast(f32x4, r); }
};

// one out tile of an out-tile-major segment of T units on the activation fin[HU16][2]
template <typename V, int T, int NS, typename ST>
__device__ __forceinline__ void row_tile16(f32x4& acc0, f32x4& acc1, APipe<V, NS>& p, ST& st, int o, const V (*fin)[2]) {
#pragma unroll
    for (int u = 0; u < HU16; ++u) {
        const V av = next_a<V, T, true, NS>(p, st, o * HU16 + u);
        acc0 = Op16<V>::mfma(av, fin[u][0], acc0);
        acc1 = Op16<V>::mfma(av, fin[u][1], acc1);
    }
}

#ifndef PG_R_PACK_AT
#define PG_R_PACK_AT 2        // unit of an even tile behind which the previous tile pair is converted
#endif
#ifndef PG_R_BIAS_AT
#define PG_R_BIAS_AT 4        // unit behind which the next tile's bias is fetched (>= 2 units before the tile ends)
#endif
static_assert(PG_R_BIAS_AT <= HU16 - 3, "the bias fetch needs two more retires of the tile behind it");
// fout = relu(W fin + b): 16 out tiles of 16 channels, out-tile-major.  The ReLU + 16-bit packing of a finished
// tile pair (VALU, needs the pair's last MFMA to retire) is placed behind the first MFMAs of the next tile, and the
// next tile's bias is read mid-tile, so that neither sits at a tile boundary where the matrix pipe would drain.
template <typename V, typename ST>
__device__ __forceinline__ void hidden_layer16(const V (*fin)[2], V (*fout)[2], ST& st, unsigned bbase, int tile0, int c0) {
    APipe<V> p;
    constexpr int T = HU16 * NT16;
    f32x4 lo0, lo1, hi0, hi1;
    BiasPipe bp(bbase, tile0);
    bp.fetch(0);
#pragma unroll
    for (int o = 0; o < NT16; ++o) {
        f32x4 acc0, acc1;
#pragma unroll
        for (int u = 0; u < HU16; ++u) {
            const V av = next_a<V, T, true, PG_PIPE_H>(p, st, o * HU16 + u, c0);
            if (u == 0) acc0 = acc1 = bp.take();
            acc0 = Op16<V>::mfma(av, fin[u][0], acc0);
            acc1 = Op16<V>::mfma(av, fin[u][1], acc1);
            if (u == PG_R_PACK_AT && o >= 2 && (o & 1) == 0) {
                if (PG_PIN_TILE) __builtin_amdgcn_sched_barrier(0);
                fout[o / 2 - 1][0] = relu_pack16<V>(lo0, hi0, true);
                fout[o / 2 - 1][1] = relu_pack16<V>(lo1, hi1, true);
                if (PG_PIN_TILE) __builtin_amdgcn_sched_barrier(0);
            }
            if (u == PG_R_BIAS_AT && o + 1 < NT16) bp.fetch(o + 1);     // in front of the next unit's next_a
        }
        if (o & 1) { hi0 = acc0; hi1 = acc1; } else { lo0 = acc0; lo1 = acc1; }
    }
    fout[NT16 / 2 - 1][0] = relu_pack16<V>(lo0, hi0, true);
    fout[NT16 / 2 - 1][1] = relu_pack16<V>(lo1, hi1, true);
}

// second stage of the factorised view layer: vacc[t][c] += sum_j w_j Y[ray][j][16t..] for the (at most two) rays
// of the wave; Y[ray] = the ray's record as fetched into LDS (A fragments: [out tile16][lane (g, row)] x 16 B)
template <typename V, bool FC>
__device__ __forceinline__ void y_apply16(f32x4 (*vacc)[2], const uint8_t* ylds, const float (*wd)[JG],
                                          const int* myr, int lane) {
    const int g = lane >> 4;
    u32x4 w[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float wx[8];
#pragma unroll
        for (int e = 0; e < JG; ++e) wx[e] = wd[c][e];
        wx[6] = (FC && g == 0) ? 1.0f : 0.0f;
        wx[7] = 0.0f;
        w[c] = __builtin_bit_cast(u32x4, Op<V>::cvt(wx));
    }
    const int ra = __builtin_amdgcn_readfirstlane(myr[0]);
    const int rb = __builtin_amdgcn_readlane(myr[1], 63);
    for (int ray = ra; ray <= rb; ++ray) {
        u32x4 b0, b1;
#pragma unroll
        for (int q = 0; q < 4; ++q) { b0[q] = myr[0] == ray ? w[0][q] : 0u; b1[q] = myr[1] == ray ? w[1][q] : 0u; }
        const uint8_t* yb = ylds + ray * REC_Y_BYTES + lane * 16;
#pragma unroll
        for (int t = 0; t < NTV16; ++t) {
            const V av = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(yb + t * 1024));
            vacc[t][0] = Op16<V>::mfma(av, __builtin_bit_cast(V, b0), vacc[t][0]);
            vacc[t][1] = Op16<V>::mfma(av, __builtin_bit_cast(V, b1), vacc[t][1]);
        }
    }
}

// TAPS = the debug tap of pg_stage_eval (stage 0: pre-activation of density layer 0) compiled in: its own
// instantiation, launched only when a dump is asked for
// OC = the on-chip variant: no per-ray records in HBM and no record kernel in front.  With frame codes (FC; BASELINE config
// 4) the code's part of the view layer is a row of a table made on the host when codes or weights change (a.wy:
// Yc[code][128] = W_view[:, 904:920] codes[code], fp32), staged into LDS a pass ahead and copied into the pseudo-joint
// slot of the Y image.  The (a, b) rows of a pass's rays are formed by the workgroup a pass
// ahead from the rays themselves (LDS-DMA of their ray_batch rows and first / last depths, the pose's bone rows kept in
// LDS), and the view layer's direction part Y by y_segment16 from limb chunks of the weight stream.
// CNT (measurement aid, dbg_stage 97; its own instantiation): passes, limbs left out of whole passes (of 6 per pass) and
// limbs left out per wave (of 48 per pass), summed over the launch into a.dbg[0..2] (unsigned)
// PP (on-chip variant only): per-ray poses (a.pose_stride != 0, as the reference's batchify_rays hands them over,
// core/trainer.py:64-81) -- the bone rows of a pass's rays are read from a.skts instead of the LDS copy of the one pose
template <typename V, bool FC, bool TAPS, bool OC, bool CNT = false, bool PP = false>
__global__ __launch_bounds__(NTHR, 2) void eval16r_kernel(const EvalArgs a) {
    static_assert(!PP || OC, "per-ray poses without records are a form of the on-chip variant");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* bias = reinterpret_cast<float*>(smem + LDSR_BIAS);
    float* cut = reinterpret_cast<float*>(smem + LDSR_CUT);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    using ST = typename std::conditional<OC, StreamRO, StreamR>::type;
    ST st{a.wstream, smem + LDSR_RING, wave, lane, 0u, 0u, 0u, lds0 + LDSR_RING, lane16};
    const uint8_t* rec_ab = reinterpret_cast<const uint8_t*>(a.rec_ab);
    const float* sk_lds = reinterpret_cast<const float*>(smem + LDSR_SK);
    const float* stage = reinterpret_cast<const float*>(smem + LDSR_STAGE);

    for (int i = tid; i < BIAS16_FLOATS; i += NTHR) bias[i] = a.bias[i];
    const float tlv = a.tau_v * 1.4426950408889634f, tld = a.tau_d * 1.4426950408889634f;
    // by joint SLOT (pg_layout.h slot16_joint): the folded sigmoid constants of both embedders, and the squared distance
    // beyond which a joint's cutoff weight 1 / (1 + 2^(v tl + cs)) is below 2^-24
    if (tid < 48) cut[tid] = -a.cutoff[(tid < J ? 0 : J) + slot_joint_dev(tid < J ? tid : tid - J)] * (tid < J ? tlv : tld);
    else if (tid < 72) {    // (of both embedders: the same mask drops a limb's view-direction part)
        const int jt = slot_joint_dev(tid - 48);
        const float far = fmaxf(a.cutoff[jt] + 24.0f / tlv, a.cutoff[J + jt] + 24.0f / tld);
        cut[tid] = far * far;
    }
    if (OC) {       // the pose's bone rows by joint slot; an all-zero Y image (limbs no pass has computed yet)
        if (!PP) for (int i = tid; i < J * 12; i += NTHR) reinterpret_cast<float*>(smem + LDSR_SK)[i] = a.skts[slot_joint_dev(i / 12) * 16 + i % 12];
        for (int i = tid; i < MAXR_F * REC_Y_BYTES / 16; i += NTHR) reinterpret_cast<uint4*>(smem + LDSR_Y)[i] = make_uint4(0u, 0u, 0u, 0u);
    }
#if defined(PG_YOUNG_PRIO)
    // experiment: the second-dispatched wave of each SIMD loses every issue arbitration to the older one and is what
    // the older one waits for at the chunk barriers: one static priority for that half (no per-segment flips)
    if (wave >= NWAVE / 2) __builtin_amdgcn_s_setprio(PG_YOUNG_PRIO);
#endif
    const unsigned bbase = lds_addr_of(bias) + 16 * g;          // this lane group's rows of bias tile 0
    // Ray bookkeeping without a division per pass (a 64-bit divide is ~150 VALU instructions, and both waves of a
    // SIMD would run it at the same time): the pass's first point is sample `off0` of ray `r0`; a pass later both
    // advance by the constant step of the persistent grid.
    const long long step = (long long)PTS * gridDim.x;
    const int dq = __builtin_amdgcn_readfirstlane((int)(step / a.S)), dr = __builtin_amdgcn_readfirstlane((int)(step % a.S));
    long long p0 = (long long)blockIdx.x * PTS;
    int r0 = __builtin_amdgcn_readfirstlane((int)(p0 / a.S));
    int off0 = __builtin_amdgcn_readfirstlane((int)(p0 - (long long)r0 * a.S));
    // (a, b) of the first pass's rays into buffer 0; every later pass finds its own fetched (OC: formed) a pass ahead
    if (OC) {
        lds_barrier();                      // the bone rows are in LDS
        if (tid < MAXR_F * J) {
            const int k = tid / J, sl = tid - k * J;
            const long long ray = min((long long)r0 + k, (long long)a.n_rays - 1);
            const float* skr = PP ? a.skts + ray * a.pose_stride + slot_joint_dev(sl) * 16 : sk_lds + sl * 12;
            ab_row(skr, a.rays + ray * 11, a.z[ray * a.S], a.z[ray * a.S + a.S - 1],
                   reinterpret_cast<float4*>(smem + LDSR_AB + k * REC_AB_BYTES + sl * 32));
        }
    } else if ((int)blockIdx.x < a.n_iters && wave < LDS_AB_BYTES / 1024)
        dma_piece(rec_ab + (long long)r0 * REC_AB_BYTES + wave * 1024, lds0 + LDSR_AB + wave * 1024, lane16);
    if (OC && FC) {     // the code rows of the first pass's rays
        const float* ytab = reinterpret_cast<const float*>(a.wy);
        for (int idx = tid; idx < MAXR_F * VW; idx += NTHR) {
            const int k = idx / VW, o = idx - k * VW;
            const float cf = a.cams ? a.cams[min((long long)r0 + k, (long long)a.n_rays - 1)] : -1.0f;
            const int ci = cf < 0.0f ? a.n_codes : min((int)cf, a.n_codes - 1);
            reinterpret_cast<float*>(smem + LDSR_YC)[idx] = ytab[(long long)ci * VW + o];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    int abuf = 0;
    // limbs no point of the first pass is in range of; every later pass learns its own a pass ahead (below).  The weight
    // ring starts with that mask: such limbs' chunks are never fetched.
    auto rays_of_pass = [&](long long p0_, int off0_) {         // index of the pass's last ray among its (<= 5) rays
        const int last_ = (int)max(0ll, min((long long)PTS - 1, a.n_points - 1 - p0_));
        const int t_ = off0_ + last_;
        return (t_ >= a.S) + (t_ >= 2 * a.S) + (t_ >= 3 * a.S) + (t_ >= 4 * a.S);
    };
    int gmask = 0;
#if !defined(PG_NO_FAR_SKIP)
    gmask = pass_far_mask(smem + LDSR_AB, rays_of_pass(p0, off0), cut + 2 * J + JG * g, g, col);
    if (!a.far_skip) gmask = 0;
#endif
    st.start((uint32_t)gmask);

    // depths of the next pass, fetched a pass ahead (consumed at the top of the pass: one wait finds them there)
    float nx_z[2] = {0.0f, 0.0f};
#define PG_PREFETCH_Z(itn)                                                                         \
    do {                                                                                           \
        const long long p0n_ = (long long)(itn) * PTS + wave * 32 + col;                           \
        nx_z[0] = a.z[min(p0n_, a.n_points - 1)];                                                  \
        nx_z[1] = a.z[min(p0n_ + 16, a.n_points - 1)];                                             \
    } while (0)
    if ((int)blockIdx.x < a.n_iters) PG_PREFETCH_Z(blockIdx.x);

#if defined(PG_STAMPS)
    unsigned long long stamps[12];
#endif
    for (int it = blockIdx.x; it < a.n_iters; it += gridDim.x) {
        PG_STAMP(0);
        // a per-pass copy of the lane index: LDS addresses derived from it are recomputed per pass (a few VALU operations)
        // instead of being hoisted out of the pass loop and spilled -- the kernel sits at exactly 256 registers
        int lane_p = lane;
        asm volatile("" : "+v"(lane_p));
        const int g_p = lane_p >> 4;
        // rays of the pass's points: point i of the pass is sample off0 + i of ray r0, i.e. (S >= 64, 256 points)
        // at most 4 rays on; points past the end of the launch (last pass) take the last valid ray
        const int last = (int)min((long long)PTS - 1, a.n_points - 1 - p0);                 // wave-uniform
        const int S1 = a.S, S2 = 2 * a.S, S3 = 3 * a.S, S4 = 4 * a.S;
        auto ray_of = [&](int t) { return (t >= S1) + (t >= S2) + (t >= S3) + (t >= S4); };
        const int nrm1 = ray_of(off0 + last);
        int myr[2];
        float zz[2];
        const float* abp[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int i = wave * 32 + 16 * c + col;
            myr[c] = min(ray_of(off0 + i), nrm1);
            zz[c] = nx_z[c];
            abp[c] = opaque_ptr(reinterpret_cast<const float*>(smem + LDSR_AB + abuf * LDS_AB_BYTES + myr[c] * REC_AB_BYTES) + JG * g_p * 8);
        }
        // the next pass of this workgroup
        const int itn = min(it + (int)gridDim.x, a.n_iters - 1);
        int off0n = off0 + dr, r0n = r0 + dq;
        if (off0n >= a.S) { off0n -= a.S; ++r0n; }
        const float* cutd = opaque_ptr(cut + J + JG * g_p);
        const QFromAB q0{abp[0], zz[0]}, q1{abp[1], zz[1]};
        float cutv[JG];                         // the lane group's folded cutoff constants (6 registers)
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) cutv[jj] = cut[JG * g_p + jj];
        // limbs out of cutoff range of every point of the wave (x_segment16): one ballot per joint slot over the four
        // lane groups' joints and the 2 x 16 points
        int farmask = 0;
#if !defined(PG_NO_FAR_SKIP)
        {
            const float* far2 = opaque_ptr(cut + 2 * J + JG * g_p);
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) {
                float ax, ay, az, bx, by, bz;
                q0(jj, ax, ay, az);
                q1(jj, bx, by, bz);
                const bool near = fminf(ax * ax + ay * ay + az * az, bx * bx + by * by + bz * bz) < far2[jj];
                if (__builtin_amdgcn_ballot_w64(near) == 0ull) farmask |= 1 << jj;
            }
            farmask = __builtin_amdgcn_readfirstlane(a.far_skip ? farmask : 0);
        }
#endif
        if (CNT && lane == 0) {
            if (wave == 0) { atomicAdd(reinterpret_cast<unsigned*>(a.dbg), 1u); atomicAdd(reinterpret_cast<unsigned*>(a.dbg) + 1, (unsigned)__builtin_popcount(gmask)); }
            atomicAdd(reinterpret_cast<unsigned*>(a.dbg) + 2, (unsigned)__builtin_popcount(farmask));
        }
        // Behind layer 0's first chunk entry every wave is done with the previous pass: its Y records and the
        // (a, b) buffer of the pass before may be overwritten.  Wave w fetches out tile w of this pass's MAXR_F Y
        // records and waves 0..3 a piece of the NEXT pass's (a, b); both are in LDS, and visible, one chunk entry on.
        auto fetch_records = [&]() {
            if (OC) {
                // the NEXT pass's rays: 64 floats of their ray_batch rows from the first one on (wave 0) and their first
                // and last depths (wave 1), lane offsets clamped to the arrays; in LDS one chunk entry on
                const long long rn = min((long long)r0n, (long long)a.n_rays - 1);
                if (wave == 0) dma_dwords(a.rays, (uint32_t)(min(rn * 11 + lane_p, (long long)a.n_rays * 11 - 1) * 4), lds0 + LDSR_STAGE);
                else if (wave == 1) {
                    const long long ray = min(rn + min(lane_p >> 1, MAXR_F - 1), (long long)a.n_rays - 1);
                    dma_dwords(a.z, (uint32_t)((ray * a.S + ((lane_p & 1) ? a.S - 1 : 0)) * 4), lds0 + LDSR_STAGE + 256);
                } else if (FC && wave == 2 && a.cams)         // their frame-code indices
                    dma_dwords(a.cams, (uint32_t)(min(rn + min(lane_p, MAXR_F - 1), (long long)a.n_rays - 1) * 4), lds0 + LDSR_STAGE + 512);
                return;
            }
            const uint8_t* ysrc = a.rec_y + (size_t)r0 * REC_Y_BYTES + wave * 1024;
#pragma unroll
            for (int k = 0; k < MAXR_F; ++k)         // slots past the pass's last ray re-fetch that ray (an L2 hit, not HBM; no branch)
                dma_piece(ysrc + (size_t)min(k, nrm1) * REC_Y_BYTES, lds0 + LDSR_Y + k * REC_Y_BYTES + wave * 1024, lane16);
            if (wave < LDS_AB_BYTES / 1024)
                dma_piece(rec_ab + (long long)min(r0n, a.n_rays - 1) * REC_AB_BYTES + wave * 1024,
                          lds0 + LDSR_AB + (abuf ^ 1) * LDS_AB_BYTES + wave * 1024, lane16);
        };

        PG_STAMP(1);
        // first stream chunk of the segments behind layer 0 (pg_program.h R): where the ring bookkeeping is static (Stream::plain_ok)
        constexpr int CH = pgp::R::CH_HID;
        constexpr int C_L1 = pgp::R::CH_L0X + (OC ? pgp::R::NLIMB : 0), C_L5H = C_L1 + 4 * CH, C_L6 = C_L5H + CH + pgp::R::CH_L0X, C_AV = C_L6 + 2 * CH;
        static_assert(C_L5H + CH == (OC ? pgp::R::C_L5X_OC : pgp::R::C_L5X) && C_AV + pgp::R::CH_AVR == (OC ? pgp::R::NCHUNK_OC : pgp::R::NCHUNK),
                      "chunk bases follow pg_program.h");
        V fa[HU16][2], fb[HU16][2];
        {   // ---- layer 0: K = 432 generated on the fly, all 16 out tiles live ----
            f32x4 acc[NT16][2];
#pragma unroll
            for (int o = 0; o < NT16; ++o) acc[o][0] = acc[o][1] = load_bias16(bias, BS_LAYER0 + o, g);
            x_segment16<V>(acc, st, abp[0], abp[1], zz[0], zz[1], cutv, tlv, farmask, gmask, fetch_records);
            if (TAPS && a.dbg && a.dbg_stage == 0) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int i = wave * 32 + 16 * c + col;
                    if (i <= last) {
#pragma unroll
                        for (int o = 0; o < NT16; ++o)
#pragma unroll
                            for (int r = 0; r < 4; ++r) a.dbg[(p0 + i) * W + 16 * o + 4 * g + r] = acc[o][c][r];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < HU16; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) fa[u][c] = relu_pack16<V>(acc[2 * u][c], acc[2 * u + 1][c], true);
        }
        if constexpr (OC) {
            // the view layer's direction part of this pass's rays, for the limbs in range
            y_segment16<V>(st, gmask, smem + LDSR_AB + abuf * LDS_AB_BYTES, smem + LDSR_Y, nrm1, wave, lane_p);
            if (FC) y_code16<V>(reinterpret_cast<const float*>(smem + LDSR_YC + abuf * LDSR_YC_BYTES), smem + LDSR_Y, nrm1, wave * 64 + lane_p);
        }
        PG_STAMP(2);
        // ---- layers 1..4 ----
        hidden_layer16<V>(fa, fb, st, bbase, BS_LAYER0 + 1 * NT16, C_L1);
        if constexpr (OC) {
            // The NEXT pass's (a, b) rows from the staged rays.  The fetch was issued in layer 0's first chunk, possibly
            // between that chunk's refill pieces: the counted wait of the SECOND entry behind it covers it, and that
            // entry's barrier makes the other wave's share visible -- layer 1's four entries lie in between.  15 of each
            // wave's lanes take one (ray, joint slot) each.
            if (FC && wave < MAXR_F) {      // the NEXT pass's code rows: ray `wave`, two dword DMAs of 64 floats each (landed entries before the pass ends)
                const float cf = a.cams ? stage[128 + wave] : -1.0f;
                const int ci = __builtin_amdgcn_readfirstlane(cf < 0.0f ? a.n_codes : min((int)cf, a.n_codes - 1));
                const uint32_t dst = lds0 + LDSR_YC + (abuf ^ 1) * LDSR_YC_BYTES + wave * (VW * 4);
                dma_dwords(a.wy, (uint32_t)((ci * VW + lane_p) * 4), dst);
                dma_dwords(a.wy, (uint32_t)((ci * VW + 64 + lane_p) * 4), dst + 256);
            }
            const int item = wave * 15 + (lane_p & 15);         // (per pass: addresses derived from it are not hoisted out of the pass loop)
            if (lane_p < 15) {
                const int k = item / J, sl = item - k * J;
                const float* skr = PP ? a.skts + min((long long)r0n + k, (long long)a.n_rays - 1) * a.pose_stride + slot_joint_dev(sl) * 16 : sk_lds + sl * 12;
                ab_row(skr, stage + 11 * k, stage[64 + 2 * k], stage[64 + 2 * k + 1],
                       reinterpret_cast<float4*>(smem + LDSR_AB + (abuf ^ 1) * LDS_AB_BYTES + k * REC_AB_BYTES + sl * 32));
            }
        }
        hidden_layer16<V>(fb, fa, st, bbase, BS_LAYER0 + 2 * NT16, C_L1 + CH);
        hidden_layer16<V>(fa, fb, st, bbase, BS_LAYER0 + 3 * NT16, C_L1 + 2 * CH);
        hidden_layer16<V>(fb, fa, st, bbase, BS_LAYER0 + 4 * NT16, C_L1 + 3 * CH);
        PG_STAMP(3);
        {   // ---- layer 5: [x(432), h4(256)] -> 256 (skip connection, nerf.py:99-101) ----
            f32x4 acc[NT16][2];
            APipeX<V> p5;
            BiasPipe bp(bbase, BS_LAYER0 + 5 * NT16);
            bp.fetch(0);
#pragma unroll
            for (int o = 0; o < NT16; ++o) {
#pragma unroll
                for (int u = 0; u < HU16; ++u) {
                    const V av = next_a<V, HU16 * NT16, true, PG_PIPE_X>(p5, st, o * HU16 + u, C_L5H);
                    if (u == 0) acc[o][0] = acc[o][1] = bp.take();
                    acc[o][0] = Op16<V>::mfma(av, fa[u][0], acc[o][0]);
                    acc[o][1] = Op16<V>::mfma(av, fa[u][1], acc[o][1]);
                    if (u == PG_R_BIAS_AT && o + 1 < NT16) bp.fetch(o + 1);
                }
            }
            x_segment16<V>(acc, st, abp[0], abp[1], zz[0], zz[1], cutv, tlv, farmask, gmask, [] {});
#pragma unroll
            for (int u = 0; u < HU16; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) fb[u][c] = relu_pack16<V>(acc[2 * u][c], acc[2 * u + 1][c], true);
        }
        PG_STAMP(4);
        hidden_layer16<V>(fb, fa, st, bbase, BS_LAYER0 + 6 * NT16, C_L6);
        hidden_layer16<V>(fa, fb, st, bbase, BS_LAYER0 + 7 * NT16, C_L6 + CH);
        PG_STAMP(5);
        // the NEXT pass's limb mask, from its (a, b) records (in LDS since this pass's second chunk entry): the ring's
        // prefetch pointer wraps to the head of the stream two chunk entries from here and must know it by then
        int gmask_n = 0;
#if !defined(PG_NO_FAR_SKIP)
        gmask_n = pass_far_mask(smem + LDSR_AB + (abuf ^ 1) * LDS_AB_BYTES, rays_of_pass(p0 + step, off0n),
                                opaque_ptr(cut + 2 * J + JG * g_p), g_p, lane_p & 15);
        if (!a.far_skip) gmask_n = 0;
#endif
        st.nx_mask = (uint32_t)gmask_n;
        // ---- sigma head + view layer (feature layer folded in, view directions from the Y records) + rgb head,
        // one stream segment: the rgb head's 4 units sit in the chunk the view tiles end in ----
        float sigma[2];
        V fg[NTV16 / 2][2];
        APipe<V> pv;
        {
            f32x4 vacc[NTV16][2];
            constexpr int TAV = pgp::R::U_AV;
            static_assert(BS_VIEWF == BS_ALPHA + 1 && BS_RGB == BS_VIEWF + NTV16, "alpha, view and rgb bias tiles are consecutive");
            BiasPipe bp(bbase, BS_ALPHA);
            bp.fetch(0);
#pragma unroll
            for (int o = 0; o < NTV16 + 1; ++o) {       // tile 0: alpha (row 0), tiles 1..8: the folded view layer
                f32x4 t0, t1;
#pragma unroll
                for (int u = 0; u < HU16; ++u) {
                    const V av = next_a<V, TAV, true, PG_PIPE_H>(pv, st, o * HU16 + u, C_AV);
                    if (u == 0) t0 = t1 = bp.take();
                    t0 = Op16<V>::mfma(av, fb[u][0], t0);
                    t1 = Op16<V>::mfma(av, fb[u][1], t1);
                    if (u == PG_R_BIAS_AT && o < NTV16) bp.fetch(o + 1);
                }
                if (o == 0) { sigma[0] = t0[0]; sigma[1] = t1[0]; }
                else { vacc[o - 1][0] = t0; vacc[o - 1][1] = t1; }
            }
            PG_STAMP(6);
            float wd[2][JG];
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) {
                float qx, qy, qz;
                q0(jj, qx, qy, qz);
                wd[0][jj] = cutoff_weight_fast(__builtin_amdgcn_sqrtf(qx * qx + qy * qy + qz * qz), tld, cutd[jj]);
                q1(jj, qx, qy, qz);
                wd[1][jj] = cutoff_weight_fast(__builtin_amdgcn_sqrtf(qx * qx + qy * qy + qz * qz), tld, cutd[jj]);
                // a limb left out of the pass has weights below 2^-24 in every point: exactly zero instead, so that what the
                // on-chip variant's Y image still holds for the limb from an earlier pass can never reach a result
                if (OC && ((gmask >> jj) & 1)) wd[0][jj] = wd[1][jj] = 0.0f;
            }
            y_apply16<V, FC>(vacc, smem + LDSR_Y, wd, myr, lane_p);
#pragma unroll
            for (int u = 0; u < NTV16 / 2; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) fg[u][c] = relu_pack16<V>(vacc[2 * u][c], vacc[2 * u + 1][c], true);
        }
        PG_STAMP(7);
        // the next pass's depths: in flight through the rgb head and the pass boundary (unconditional: a
        // conditional re-definition would keep registers live through the whole pass)
        PG_PREFETCH_Z(itn);
        f32x4 c0, c1;
        {
            APipe<V> pr;
            BiasPipe bp(bbase, BS_RGB);
            bp.fetch(0);
#pragma unroll
            for (int u = 0; u < pgp::R::U_RGB; ++u) {
                const V av = next_a_cont<V, pgp::R::U_RGB, PG_PIPE_H, pgp::R::U_AV % UPC>(pr, st, u);
                if (u == 0) c0 = c1 = bp.take();
                c0 = Op16<V>::mfma(av, fg[u][0], c0);
                c1 = Op16<V>::mfma(av, fg[u][1], c1);
            }
        }
        if (g == 0) {       // rows 0..2 of the rgb tile and row 0 of the alpha tile live in lane group 0
            const int i = wave * 32 + col;      // (the pass's point index is recomputed here rather than kept for the whole pass)
            if (i <= last) *reinterpret_cast<float4*>(a.raw + (p0 + i) * 4) = make_float4(c0[0], c0[1], c0[2], sigma[0]);
            if (i + 16 <= last) *reinterpret_cast<float4*>(a.raw + (p0 + i + 16) * 4) = make_float4(c1[0], c1[1], c1[2], sigma[1]);
        }
        abuf ^= 1;
        p0 += step; r0 = r0n; off0 = off0n;
        gmask = gmask_n;
        PG_STAMP(8);
#if defined(PG_STAMPS)
        if (a.dbg && a.dbg_stage == 99 && lane == 0 && it < 64) {
            for (int k = 0; k < 9; ++k) reinterpret_cast<unsigned long long*>(a.dbg)[((long long)it * NWAVE + wave) * 16 + k] = stamps[k];
            reinterpret_cast<unsigned long long*>(a.dbg)[((long long)it * NWAVE + wave) * 16 + 9] = st.t_vm;
            reinterpret_cast<unsigned long long*>(a.dbg)[((long long)it * NWAVE + wave) * 16 + 10] = st.t_bar;
            st.t_vm = 0; st.t_bar = 0;
        }
#endif
    }
    st.drain();
}

template <typename V, bool FC, bool TAPS, bool OC, bool CNT = false, bool PP = false>
static hipError_t launch_eval16r(const EvalArgs& a, int grid, hipStream_t stream) {
    auto k = eval16r_kernel<V, FC, TAPS, OC, CNT, PP>;
    constexpr int lds = OC ? LDSR_TOTAL_OC : LDSR_TOTAL;
    static std::atomic<unsigned long long> attr_done{0};       // per device (pg_device.h)
    const hipError_t ae = ensure_lds_attr(reinterpret_cast<const void*>(k), lds, attr_done);
    if (ae != hipSuccess) return ae;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHR), lds, stream, a);
    return hipGetLastError();
}

template <typename V>
static hipError_t dispatch_eval16r(const EvalArgs& a, int framecode, int onchip, int grid, hipStream_t s) {
    if (onchip) {       // (no debug taps in this variant; 99: the stamps of a PG_STAMPS build, 97: the limb-mask counters)
        if (a.dbg && a.dbg_stage != 99 && a.dbg_stage != 97) return hipErrorInvalidValue;
        const bool pp = a.pose_stride != 0;
        if (a.dbg && a.dbg_stage == 97) return (pp || framecode) ? hipErrorInvalidValue : launch_eval16r<V, false, false, true, true>(a, grid, s);
        if (framecode) return pp ? launch_eval16r<V, true, false, true, false, true>(a, grid, s) : launch_eval16r<V, true, false, true>(a, grid, s);
        return pp ? launch_eval16r<V, false, false, true, false, true>(a, grid, s) : launch_eval16r<V, false, false, true>(a, grid, s);
    }
    const bool taps = a.dbg && a.dbg_stage != 99;
    if (taps) return framecode ? launch_eval16r<V, true, true, false>(a, grid, s) : launch_eval16r<V, false, true, false>(a, grid, s);
    return framecode ? launch_eval16r<V, true, false, false>(a, grid, s) : launch_eval16r<V, false, false, false>(a, grid, s);
}

}  // namespace pgd

// needs S >= pgl::FACT_MIN_S, the 16-row bias table (pack_bias_s) and
//   onchip = 0: the R weight stream (pack_stream_r) and the per-ray records of pg_rayrec.hip in a.rec_ab / a.rec_y
//   onchip = 1: the on-chip R stream (pack_stream_r(..., onchip)); one pose for all rays or (a.pose_stride != 0) a pose per ray; with
//               frame codes a.wy = the table Yc[n_codes + 1][128] floats (W_view[:, 904:920] codes[c]) and a.cams the per-ray indices
extern "C" int pg_launch_eval16r(const pgd::EvalArgs* a, int fp16, int framecode, int onchip, int grid, void* stream) {
    using namespace pgd;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (onchip && framecode && !a->wy) return (int)hipErrorInvalidValue;       // (the code table Yc: pg_api.hip ensure_ycode)
    return (int)(fp16 ? dispatch_eval16r<f16x8>(*a, framecode, onchip, grid, s) : dispatch_eval16r<bf16x8>(*a, framecode, onchip, grid, s));
}
