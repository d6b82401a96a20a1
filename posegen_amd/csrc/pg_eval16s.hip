// pg_eval16s.hip -- the fused embed+MLP kernel of pg_eval16.hip on v_mfma_f32_16x16x32
// (bf16 / fp16 operands, fp32 accumulate), factorised view layer only (S >= 64 samples per ray).
//
// Same structure as pg_eval16.hip (8 waves x 32 points, activations in registers, weights
// streamed L2 -> LDS ring, embedding generated as B fragments); the tile shape differs
// (pg_layout.h "small tile"): A = 16 out channels x 32 k, B = 32 k x 16 points.  A wave's 32 points
// are two column tiles c, every A fragment read from the ring feeds two MFMAs, lane group
// g = lane>>4 generates the embedding of joints 6g..6g+5 for the wave's points col and col+16.
// Built to test whether MI355X holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS 7):
// on this kernel it does not -- same wall time as pg_eval16.hip -- so it is opt-in
// (POSEGEN_MFMA=16) and pg_eval16.hip stays the default.
#include "pg_eval16_common.h"

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

using StreamS = Stream<NWAVE, pgp::S::NCHUNK, PG_DMA_WAVES>;
static_assert(BIAS16_FLOATS <= BIAS_FLOATS, "the 16-row bias table shares the LDS region of the 32-row one");

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

// acc += W[:, x-columns] x for the wave's two column tiles; X16 sequence of pg_layout.h:
// joints 0..3 (two units each), their leftovers, joints 4, 5, their leftovers
template <typename V, typename ST>
__device__ __forceinline__ void x_segment16(f32x4 (*acc)[2], ST& st, const QFromAB& q0, const QFromAB& q1,
                                            const float* cutb, float tau) {
    APipeX<V> p;
    constexpr int T = XU16 * NT16;
    float lo0[8], lo1[8];
    int uu = 0;
#pragma clang loop unroll(full)
    for (int jj = 0; jj < JG; ++jj) {
        float x0[18], x1[18], qx, qy, qz;
        q0(jj, qx, qy, qz);
        joint_values_q<true>(qx, qy, qz, tau, cutb[jj], x0);
        q1(jj, qx, qy, qz);
        joint_values_q<true>(qx, qy, qz, tau, cutb[jj], x1);
        const int k = jj < 4 ? jj : jj - 4;
        lo0[2 * k] = x0[16]; lo0[2 * k + 1] = x0[17];
        lo1[2 * k] = x1[16]; lo1[2 * k + 1] = x1[17];
        mma_row16<V, NT16, T>(acc, p, st, uu++, Op<V>::cvt(x0), Op<V>::cvt(x1));
        mma_row16<V, NT16, T>(acc, p, st, uu++, Op<V>::cvt(x0 + 8), Op<V>::cvt(x1 + 8));
        if (jj == 3 || jj == JG - 1) {
            if (jj == JG - 1) {
#pragma unroll
                for (int e = 4; e < 8; ++e) { lo0[e] = 0.0f; lo1[e] = 0.0f; }
            }
            mma_row16<V, NT16, T>(acc, p, st, uu++, Op<V>::cvt(lo0), Op<V>::cvt(lo1));
        }
    }
}

// one out tile of an out-tile-major segment on the activation fin[HU16][2]
template <typename V, int T, int NS, typename ST>
__device__ __forceinline__ void row_tile16(f32x4& acc0, f32x4& acc1, APipe<V, NS>& p, ST& st, int o, const V (*fin)[2]) {
#pragma unroll
    for (int u = 0; u < HU16; ++u) {
        const V av = next_a<V, T, true, NS>(p, st, o * HU16 + u);
        acc0 = Op16<V>::mfma(av, fin[u][0], acc0);
        acc1 = Op16<V>::mfma(av, fin[u][1], acc1);
    }
}

// fout = relu(W fin + b): 16 out tiles, out-tile-major
template <typename V, typename ST>
__device__ __forceinline__ void hidden_layer16(const V (*fin)[2], V (*fout)[2], ST& st, const float* bias, int tile0, int g) {
    APipe<V> p;
    f32x4 e0, e1;
#pragma unroll
    for (int o = 0; o < NT16; ++o) {
        f32x4 acc0 = load_bias16(bias, tile0 + o, g), acc1 = acc0;
        row_tile16<V, HU16 * NT16>(acc0, acc1, p, st, o, fin);
        if (o & 1) {
            fout[o / 2][0] = relu_pack16<V>(e0, acc0, true);
            fout[o / 2][1] = relu_pack16<V>(e1, acc1, true);
        } else {
            e0 = acc0; e1 = acc1;
        }
    }
}

// Y stage (see pg_eval16_common.h y_stage): same 32x32x16 MFMAs with the rays as rows; the
// result goes into A fragments of the 16x16x32 second stage: [ray][out tile16][lane (g, row)]
// x 16 B, slot e = joint 6g+e of lane group g (slot 6 of g = 0: frame code).
template <typename V, bool FC>
__device__ __forceinline__ void y_stage16(const YWeights<V, FC>& yw, uint8_t* rt, int nr, int wave, int lane) {
    constexpr int NE = JH + (FC ? 1 : 0);
    using E = typename Op<V>::E;
    const int t = wave & 3, hw = wave >> 2, hl = lane >> 5, col = lane & 31;
    const uint8_t* trow = rt + min(col, nr - 1) * SLOTF_BYTES + SLOTF_T16 + hl * 16;
    const uint8_t* trow_h = trow + hw * (JH * TK * 2);
    uint8_t* ybase = rt + SLOTF_Y + ((2 * t + (col >> 4)) * 64 + (col & 15)) * 16 + 4 * hl * SLOTF_BYTES;
#pragma clang loop unroll(full)
    for (int e = 0; e < NE; ++e) {
        const uint8_t* tj = e < JH ? trow_h + e * (TK * 2) : trow + JC * (TK * 2);
        const V a0 = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(tj));
        const V a1 = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(tj + 32));
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        acc = Op<V>::mfma(a0, __builtin_bit_cast(V, yw.b[2 * e]), acc);
        acc = Op<V>::mfma(a1, __builtin_bit_cast(V, yw.b[2 * e + 1]), acc);
        // joint 12 hw + e -> lane group 2 hw + e/6, slot e%6; frame code (waves 0..3) -> group 0,
        // slot 6; waves 4..7 hold zero weights there and park the zero in group 2's unused slot 6
        const int gj = e < JH ? 2 * hw + e / JG : 2 * hw, slot = e < JH ? e % JG : JG;
#pragma unroll
        for (int r = 0; r < 4; ++r)        // ray = r + 4 hl; only MAXR_F = 5 slots exist
            if (hl == 0 || r == 0)
                *reinterpret_cast<E*>(ybase + r * SLOTF_BYTES + gj * 256 + slot * 2) = (E)acc[r];
    }
}

// second stage: vacc[t][c] += sum_j w_j Y[ray][j][16t..] for the (at most two) rays of the wave
template <typename V, bool FC>
__device__ __forceinline__ void y_apply16(f32x4 (*vacc)[2], const uint8_t* rt, const float (*wd)[JG],
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
        const uint8_t* yb = rt + ray * SLOTF_BYTES + SLOTF_Y + lane * 16;
#pragma unroll
        for (int t = 0; t < NTV16; ++t) {
            const V av = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(yb + t * 1024));
            vacc[t][0] = Op16<V>::mfma(av, __builtin_bit_cast(V, b0), vacc[t][0]);
            vacc[t][1] = Op16<V>::mfma(av, __builtin_bit_cast(V, b1), vacc[t][1]);
        }
    }
}

template <typename V, bool FC>
__global__ __launch_bounds__(NTHR, 2) void eval16s_kernel(const EvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* bias = reinterpret_cast<float*>(smem + LDS_BIAS);
    float* cut = reinterpret_cast<float*>(smem + LDS_CUT);
    uint8_t* rtf = smem + LDS_RTAB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    StreamS st{a.wstream, smem + LDS_RING, wave, lane, 0u, 0u, 0u,
               (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(smem + LDS_RING), (uint32_t)lane * 16u};

    for (int i = tid; i < BIAS16_FLOATS; i += NTHR) bias[i] = a.bias[i];
    const float tlv = a.tau_v * 1.4426950408889634f, tld = a.tau_d * 1.4426950408889634f;
    if (tid < 48) cut[tid] = -a.cutoff[tid] * (tid < J ? tlv : tld);
    for (int i = tid; i < MAXR_F * 512; i += NTHR)     // pad slots of the Y fragments stay zero
        *reinterpret_cast<uint4*>(rtf + (i / 512) * SLOTF_BYTES + SLOTF_Y + (i % 512) * 16) = make_uint4(0, 0, 0, 0);
    st.start();

    for (int it = blockIdx.x; it < a.n_iters; it += gridDim.x) {
        const long long p0 = (long long)it * PTS;
        const long long plast = min(p0 + PTS - 1, a.n_points - 1);
        const int r0 = (int)(p0 / a.S);
        const int nr = (int)(plast / a.S) - r0 + 1;
        YWeights<V, FC> yw;
        yw.load(a, wave, lane);                 // in flight across the barrier and the table build
        lds_barrier();                          // previous pass is done with the table
        ray_tablef<V, FC, NTHR>(a, rtf, r0, nr);
        lds_barrier();
        y_stage16<V, FC>(yw, rtf, nr, wave, lane);      // visible to all after the next chunk barrier

        long long gp[2];
        bool valid[2];
        int myr[2];
        float zz[2];
        const float* abp[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            gp[c] = p0 + wave * 32 + 16 * c + col;
            valid[c] = gp[c] < a.n_points;
            const long long gpc = valid[c] ? gp[c] : a.n_points - 1;
            myr[c] = (int)(gpc / a.S) - r0;
            zz[c] = a.z[gpc];
            abp[c] = opaque_ptr(reinterpret_cast<const float*>(rtf + myr[c] * SLOTF_BYTES + SLOTF_AB) + JG * g * 8);
        }
        const float* cutv = opaque_ptr(cut + JG * g);
        const float* cutd = opaque_ptr(cut + J + JG * g);
        const QFromAB q0{abp[0], zz[0]}, q1{abp[1], zz[1]};

        V fa[HU16][2], fb[HU16][2];
        {   // ---- layer 0: K = 432 generated on the fly, all 16 out tiles live ----
            f32x4 acc[NT16][2];
#pragma unroll
            for (int o = 0; o < NT16; ++o) acc[o][0] = acc[o][1] = load_bias16(bias, BS_LAYER0 + o, g);
            x_segment16<V>(acc, st, q0, q1, cutv, tlv);
            if (a.dbg && a.dbg_stage == 0) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (valid[c]) {
#pragma unroll
                        for (int o = 0; o < NT16; ++o)
#pragma unroll
                            for (int r = 0; r < 4; ++r) a.dbg[gp[c] * W + 16 * o + 4 * g + r] = acc[o][c][r];
                    }
            }
#pragma unroll
            for (int u = 0; u < HU16; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) fa[u][c] = relu_pack16<V>(acc[2 * u][c], acc[2 * u + 1][c], true);
        }
        // ---- layers 1..4 ----
        hidden_layer16<V>(fa, fb, st, bias, BS_LAYER0 + 1 * NT16, g);
        hidden_layer16<V>(fb, fa, st, bias, BS_LAYER0 + 2 * NT16, g);
        hidden_layer16<V>(fa, fb, st, bias, BS_LAYER0 + 3 * NT16, g);
        hidden_layer16<V>(fb, fa, st, bias, BS_LAYER0 + 4 * NT16, g);
        {   // ---- layer 5: [x(432), h4(256)] -> 256 (skip connection, nerf.py:99-101) ----
            f32x4 acc[NT16][2];
            APipeX<V> p5;
#pragma unroll
            for (int o = 0; o < NT16; ++o) {
                acc[o][0] = acc[o][1] = load_bias16(bias, BS_LAYER0 + 5 * NT16 + o, g);
                row_tile16<V, HU16 * NT16>(acc[o][0], acc[o][1], p5, st, o, fa);
            }
            x_segment16<V>(acc, st, q0, q1, cutv, tlv);
#pragma unroll
            for (int u = 0; u < HU16; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) fb[u][c] = relu_pack16<V>(acc[2 * u][c], acc[2 * u + 1][c], true);
        }
        hidden_layer16<V>(fb, fa, st, bias, BS_LAYER0 + 6 * NT16, g);
        hidden_layer16<V>(fa, fb, st, bias, BS_LAYER0 + 7 * NT16, g);
        // ---- sigma head + view layer (feature layer folded in, view directions factorised) ----
        float sigma[2];
        V fg[NTV16 / 2][2];
        {
            f32x4 vacc[NTV16][2];
            APipe<V> pv;
            constexpr int TAV = HU16 * (NTV16 + 1);
            {
                f32x4 s0 = load_bias16(bias, BS_ALPHA, g), s1 = s0;
                row_tile16<V, TAV>(s0, s1, pv, st, 0, fb);
                sigma[0] = s0[0]; sigma[1] = s1[0];
            }
#pragma unroll
            for (int o = 0; o < NTV16; ++o) {
                vacc[o][0] = vacc[o][1] = load_bias16(bias, BS_VIEWF + o, g);
                row_tile16<V, TAV>(vacc[o][0], vacc[o][1], pv, st, 1 + o, fb);
            }
            float wd[2][JG];
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) {
                float qx, qy, qz;
                q0(jj, qx, qy, qz);
                wd[0][jj] = cutoff_weight_fast(__builtin_amdgcn_sqrtf(qx * qx + qy * qy + qz * qz), tld, cutd[jj]);
                q1(jj, qx, qy, qz);
                wd[1][jj] = cutoff_weight_fast(__builtin_amdgcn_sqrtf(qx * qx + qy * qy + qz * qz), tld, cutd[jj]);
            }
            y_apply16<V, FC>(vacc, rtf, wd, myr, lane);
#pragma unroll
            for (int u = 0; u < NTV16 / 2; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) fg[u][c] = relu_pack16<V>(vacc[2 * u][c], vacc[2 * u + 1][c], true);
        }
        // ---- rgb head ----
        f32x4 c0 = load_bias16(bias, BS_RGB, g), c1 = c0;
        {
            APipe<V> pr;
#pragma unroll
            for (int u = 0; u < NTV16 / 2; ++u) {
                const V av = next_a<V, NTV16 / 2, true, PG_PIPE_H>(pr, st, u);
                c0 = Op16<V>::mfma(av, fg[u][0], c0);
                c1 = Op16<V>::mfma(av, fg[u][1], c1);
            }
        }
        if (g == 0) {       // rows 0..2 of the rgb tile and row 0 of the alpha tile live in lane group 0
            if (valid[0]) *reinterpret_cast<float4*>(a.raw + gp[0] * 4) = make_float4(c0[0], c0[1], c0[2], sigma[0]);
            if (valid[1]) *reinterpret_cast<float4*>(a.raw + gp[1] * 4) = make_float4(c1[0], c1[1], c1[2], sigma[1]);
        }
    }
    st.drain();
}

template <typename V, bool FC>
static hipError_t launch_eval16s(const EvalArgs& a, int grid, hipStream_t stream) {
    auto k = eval16s_kernel<V, FC>;
    static std::atomic<unsigned long long> attr_done{0};       // per device (pg_device.h)
    const hipError_t ae = ensure_lds_attr(reinterpret_cast<const void*>(k), LDS_TOTAL_F, attr_done);
    if (ae != hipSuccess) return ae;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHR), LDS_TOTAL_F, stream, a);
    return hipGetLastError();
}

}  // namespace pgd

// needs S >= pgl::FACT_MIN_S, the S weight stream (pack_stream_s), bias table (pack_bias_s)
// and the Y-stage weights (pack_vy)
extern "C" int pg_launch_eval16s(const pgd::EvalArgs* a, int fp16, int framecode, int grid, void* stream) {
    using namespace pgd;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e;
    if (fp16) e = framecode ? launch_eval16s<f16x8, true>(*a, grid, s) : launch_eval16s<f16x8, false>(*a, grid, s);
    else      e = framecode ? launch_eval16s<bf16x8, true>(*a, grid, s) : launch_eval16s<bf16x8, false>(*a, grid, s);
    return (int)e;
}
