// pg_eval16w.hip -- the factorised fused embed+MLP kernel of pg_eval16.hip with ONE wave per
// SIMD: 4 waves x 64 points, up to 512 registers per wave (bf16 / fp16 operands,
// v_mfma_f32_32x32x16, fp32 accumulate; rays with >= 64 samples).
//
// Same weight stream, bias table, Y-stage weights and per-ray tables as pg_eval16.hip (FACT).
// A wave owns two column tiles c of 32 points, so every weight fragment read from the LDS ring
// feeds two independent MFMAs: half the ring reads, waits and pads per MFMA, and half the LDS
// read bytes per FLOP.  What it gives up is the second wave of the SIMD that overlaps one
// wave's VALU work with the other's MFMAs.
#include "pg_eval16_common.h"

namespace pgd {
using namespace pgp::A;

constexpr int NWAVE_W = 4;
constexpr int NTHR_W = NWAVE_W * 64;
static_assert(NWAVE_W * 64 == PTS, "same 256 points per workgroup pass as pg_eval16.hip (table sizes, grid)");

using StreamW = Stream<NWAVE_W, pgp::AF::NCHUNK, NWAVE_W>;

// one k-unit (two B fragments, one per column tile) against NO out tiles of a k-major segment
template <typename V, int NO, int T, int NS, typename ST>
__device__ __forceinline__ void mma_row_w(f32x16 (*acc)[2], APipe<V, NS>& p, ST& st, int uu, V b0, V b1) {
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        const V av = next_a<V, T, true, NS>(p, st, uu * NO + o);
        acc[o][0] = Op<V>::mfma(av, b0, acc[o][0]);
        acc[o][1] = Op<V>::mfma(av, b1, acc[o][1]);
    }
}

// acc += W[:, x-columns] x for the wave's two column tiles (X sequence of pg_layout.h)
template <typename V, typename ST>
__device__ __forceinline__ void x_segment_w(f32x16 (*acc)[2], ST& st, const QFromAB& q0, const QFromAB& q1,
                                            const float* cutb, float tau) {
    APipeX<V> p;
    constexpr int T = XU * NT;
#pragma clang loop unroll(full)
    for (int sb = 0; sb < 3; ++sb) {
        float lo0[8], lo1[8];
#pragma clang loop unroll(full)
        for (int k = 0; k < 4; ++k) {
            const int jj = 4 * sb + k;
            float x0[18], x1[18], qx, qy, qz;
            q0(jj, qx, qy, qz);
            joint_values_q<true>(qx, qy, qz, tau, cutb[jj], x0);
            q1(jj, qx, qy, qz);
            joint_values_q<true>(qx, qy, qz, tau, cutb[jj], x1);
            lo0[2 * k] = x0[16]; lo0[2 * k + 1] = x0[17];
            lo1[2 * k] = x1[16]; lo1[2 * k + 1] = x1[17];
            mma_row_w<V, NT, T>(acc, p, st, sb * 9 + 2 * k, Op<V>::cvt(x0), Op<V>::cvt(x1));
            mma_row_w<V, NT, T>(acc, p, st, sb * 9 + 2 * k + 1, Op<V>::cvt(x0 + 8), Op<V>::cvt(x1 + 8));
        }
        mma_row_w<V, NT, T>(acc, p, st, sb * 9 + 8, Op<V>::cvt(lo0), Op<V>::cvt(lo1));
    }
}

// one out tile of an out-tile-major segment on the activation fin[HU][2]
template <typename V, int T, int NS, typename ST>
__device__ __forceinline__ void row_tile_w(f32x16& acc0, f32x16& acc1, APipe<V, NS>& p, ST& st, int o, const V (*fin)[2]) {
#pragma unroll
    for (int u = 0; u < HU; ++u) {
        const V av = next_a<V, T, true, NS>(p, st, o * HU + u);
        acc0 = Op<V>::mfma(av, fin[u][0], acc0);
        acc1 = Op<V>::mfma(av, fin[u][1], acc1);
    }
}

// fout = relu(W fin + b), out-tile-major; the conversion of tile o-1 is spread over the first
// MFMAs of tile o, one quarter per unit, so that it issues in their shadow
template <typename V, typename ST>
__device__ __forceinline__ void hidden_layer_w(const V (*fin)[2], V (*fout)[2], ST& st, const float* bias, int tile0, int h) {
    APipe<V> p;
    f32x16 pv0, pv1;
    constexpr int T = HU * NT;
#pragma unroll
    for (int o = 0; o < NT; ++o) {
        f32x16 acc0 = load_bias(bias, tile0 + o, h), acc1 = acc0;
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            const V av = next_a<V, T, true, PG_PIPE_H>(p, st, o * HU + u);
            acc0 = Op<V>::mfma(av, fin[u][0], acc0);
            acc1 = Op<V>::mfma(av, fin[u][1], acc1);
            if (o > 0 && u == 4) relu_pack<V>(pv0, fout[2 * (o - 1)][0], fout[2 * (o - 1) + 1][0], true);
            if (o > 0 && u == 8) relu_pack<V>(pv1, fout[2 * (o - 1)][1], fout[2 * (o - 1) + 1][1], true);
        }
        pv0 = acc0; pv1 = acc1;
    }
    relu_pack<V>(pv0, fout[2 * (NT - 1)][0], fout[2 * (NT - 1) + 1][0], true);
    relu_pack<V>(pv1, fout[2 * (NT - 1)][1], fout[2 * (NT - 1) + 1][1], true);
}

// second stage of the factorised view layer for two column tiles (see y_apply)
template <typename V, bool FC>
__device__ __forceinline__ void y_apply_w(f32x16 (*vacc)[2], const uint8_t* rt, const float (*wd)[JH],
                                          const int* myr, int lane) {
    const int h = lane >> 5;
    u32x4 w0[2], w1[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float wx[16];
#pragma unroll
        for (int e = 0; e < JH; ++e) wx[e] = wd[c][e];
        wx[12] = (FC && h == 0) ? 1.0f : 0.0f;
        wx[13] = wx[14] = wx[15] = 0.0f;
        w0[c] = __builtin_bit_cast(u32x4, Op<V>::cvt(wx));
        w1[c] = __builtin_bit_cast(u32x4, Op<V>::cvt(wx + 8));
    }
    const int ra = __builtin_amdgcn_readfirstlane(myr[0]);
    const int rb = __builtin_amdgcn_readlane(myr[1], 63);
    for (int ray = ra; ray <= rb; ++ray) {
        u32x4 b0[2], b1[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                b0[c][q] = myr[c] == ray ? w0[c][q] : 0u;
                b1[c][q] = myr[c] == ray ? w1[c][q] : 0u;
            }
        const uint8_t* yb = rt + ray * SLOTF_BYTES + SLOTF_Y + lane * 16;
#pragma unroll
        for (int t = 0; t < NTV; ++t) {
            const V a0 = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(yb + (t * 2) * 1024));
            const V a1 = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(yb + (t * 2 + 1) * 1024));
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                vacc[t][c] = Op<V>::mfma(a0, __builtin_bit_cast(V, b0[c]), vacc[t][c]);
                vacc[t][c] = Op<V>::mfma(a1, __builtin_bit_cast(V, b1[c]), vacc[t][c]);
            }
        }
    }
}

template <typename V, bool FC>
__global__ __launch_bounds__(NTHR_W, 1) void eval16w_kernel(const EvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* bias = reinterpret_cast<float*>(smem + LDS_BIAS);
    float* cut = reinterpret_cast<float*>(smem + LDS_CUT);
    uint8_t* rtf = smem + LDS_RTAB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, pt = lane & 31;
    StreamW st{a.wstream, smem + LDS_RING, wave, lane, 0u, 0u, 0u,
               (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(smem + LDS_RING), (uint32_t)lane * 16u};

    for (int i = tid; i < BIAS_FLOATS; i += NTHR_W) bias[i] = a.bias[i];
    const float tlv = a.tau_v * 1.4426950408889634f, tld = a.tau_d * 1.4426950408889634f;
    if (tid < 48) cut[tid] = -a.cutoff[tid] * (tid < J ? tlv : tld);
    for (int i = tid; i < MAXR_F * 512; i += NTHR_W)     // pad slots of the Y fragments stay zero
        *reinterpret_cast<uint4*>(rtf + (i / 512) * SLOTF_BYTES + SLOTF_Y + (i % 512) * 16) = make_uint4(0, 0, 0, 0);
    st.start();

#if defined(PG_STAMPS)
    unsigned long long stamps[12];
#endif
    for (int it = blockIdx.x; it < a.n_iters; it += gridDim.x) {
        PG_STAMP(0);
        const long long p0 = (long long)it * PTS;
        const long long plast = min(p0 + PTS - 1, a.n_points - 1);
        const int r0 = (int)(p0 / a.S);
        const int nr = (int)(plast / a.S) - r0 + 1;
        {   // Y stage: this wave stands in for waves w and w+4 of the 8-wave kernel (tile w, both joint halves)
            YWeights<V, FC> yw0, yw1;
            yw0.load(a, wave, lane);
            yw1.load(a, wave + 4, lane);
            lds_barrier();                          // previous pass is done with the table
            ray_tablef<V, FC, NTHR_W>(a, rtf, r0, nr);
            lds_barrier();
            y_stage<V, FC>(yw0, rtf, nr, wave, lane);
            y_stage<V, FC>(yw1, rtf, nr, wave + 4, lane);
        }

        long long gp[2];
        bool valid[2];
        int myr[2];
        float zz[2];
        const float* abp[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            gp[c] = p0 + wave * 64 + 32 * c + pt;
            valid[c] = gp[c] < a.n_points;
            const long long gpc = valid[c] ? gp[c] : a.n_points - 1;
            myr[c] = (int)(gpc / a.S) - r0;
            zz[c] = a.z[gpc];
            abp[c] = opaque_ptr(reinterpret_cast<const float*>(rtf + myr[c] * SLOTF_BYTES + SLOTF_AB) + JH * h * 8);
        }
        const float* cutv = opaque_ptr(cut + JH * h);
        const float* cutd = opaque_ptr(cut + J + JH * h);
        const QFromAB q0{abp[0], zz[0]}, q1{abp[1], zz[1]};

        PG_STAMP(1);
        V fa[HU][2], fb[HU][2];
        {   // ---- layer 0: K = 432 generated on the fly, all 8 out tiles of both column tiles live ----
            f32x16 acc[NT][2];
#pragma unroll
            for (int o = 0; o < NT; ++o) acc[o][0] = acc[o][1] = load_bias(bias, BT_LAYER0 + o, h);
            x_segment_w<V>(acc, st, q0, q1, cutv, tlv);
            if (a.dbg && a.dbg_stage == 0) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (valid[c]) {
#pragma unroll
                        for (int o = 0; o < NT; ++o)
#pragma unroll
                            for (int r = 0; r < 16; ++r) a.dbg[gp[c] * W + 32 * o + rho(r, h)] = acc[o][c][r];
                    }
            }
#pragma unroll
            for (int o = 0; o < NT; ++o)
#pragma unroll
                for (int c = 0; c < 2; ++c) relu_pack<V>(acc[o][c], fa[2 * o][c], fa[2 * o + 1][c], true);
        }
        PG_STAMP(2);
        // ---- layers 1..4 ----
        hidden_layer_w<V>(fa, fb, st, bias, BT_LAYER0 + 1 * NT, h);
        hidden_layer_w<V>(fb, fa, st, bias, BT_LAYER0 + 2 * NT, h);
        hidden_layer_w<V>(fa, fb, st, bias, BT_LAYER0 + 3 * NT, h);
        hidden_layer_w<V>(fb, fa, st, bias, BT_LAYER0 + 4 * NT, h);
        PG_STAMP(3);
        {   // ---- layer 5: [x(432), h4(256)] -> 256 (skip connection, nerf.py:99-101) ----
            f32x16 acc[NT][2];
            APipeX<V> p5;
#pragma unroll
            for (int o = 0; o < NT; ++o) {
                acc[o][0] = acc[o][1] = load_bias(bias, BT_LAYER0 + 5 * NT + o, h);
                row_tile_w<V, HU * NT>(acc[o][0], acc[o][1], p5, st, o, fa);
            }
            x_segment_w<V>(acc, st, q0, q1, cutv, tlv);
#pragma unroll
            for (int o = 0; o < NT; ++o)
#pragma unroll
                for (int c = 0; c < 2; ++c) relu_pack<V>(acc[o][c], fb[2 * o][c], fb[2 * o + 1][c], true);
        }
        PG_STAMP(4);
        hidden_layer_w<V>(fb, fa, st, bias, BT_LAYER0 + 6 * NT, h);
        hidden_layer_w<V>(fa, fb, st, bias, BT_LAYER0 + 7 * NT, h);
        PG_STAMP(5);
        // ---- sigma head + view layer (feature layer folded in, view directions factorised) ----
        float sigma[2];
        V fg[HU / 2][2];
        {
            f32x16 vacc[NTV][2];
            APipe<V> pv;
            constexpr int TAV = HU * (NTV + 1);
            {
                f32x16 s0 = load_bias(bias, BT_ALPHA, h), s1 = s0;
                row_tile_w<V, TAV>(s0, s1, pv, st, 0, fb);
                sigma[0] = s0[0]; sigma[1] = s1[0];
            }
            PG_STAMP(6);
#pragma unroll
            for (int o = 0; o < NTV; ++o) {
                vacc[o][0] = vacc[o][1] = load_bias(bias, BT_VIEWF + o, h);
                row_tile_w<V, TAV>(vacc[o][0], vacc[o][1], pv, st, 1 + o, fb);
            }
            float wd[2][JH];
#pragma unroll
            for (int jj = 0; jj < JH; ++jj) {
                float qx, qy, qz;
                q0(jj, qx, qy, qz);
                wd[0][jj] = cutoff_weight_fast(__builtin_amdgcn_sqrtf(qx * qx + qy * qy + qz * qz), tld, cutd[jj]);
                q1(jj, qx, qy, qz);
                wd[1][jj] = cutoff_weight_fast(__builtin_amdgcn_sqrtf(qx * qx + qy * qy + qz * qz), tld, cutd[jj]);
            }
            y_apply_w<V, FC>(vacc, rtf, wd, myr, lane);
#pragma unroll
            for (int o = 0; o < NTV; ++o)
#pragma unroll
                for (int c = 0; c < 2; ++c) relu_pack<V>(vacc[o][c], fg[2 * o][c], fg[2 * o + 1][c], true);
        }
        PG_STAMP(7);
        // ---- rgb head ----
        f32x16 c0 = load_bias(bias, BT_RGB, h), c1 = c0;
        {
            APipe<V> pr;
#pragma unroll
            for (int u = 0; u < HU / 2; ++u) {
                const V av = next_a<V, HU / 2, true, PG_PIPE_H>(pr, st, u);
                c0 = Op<V>::mfma(av, fg[u][0], c0);
                c1 = Op<V>::mfma(av, fg[u][1], c1);
            }
        }
        if (h == 0) {
            if (valid[0]) *reinterpret_cast<float4*>(a.raw + gp[0] * 4) = make_float4(c0[0], c0[1], c0[2], sigma[0]);
            if (valid[1]) *reinterpret_cast<float4*>(a.raw + gp[1] * 4) = make_float4(c1[0], c1[1], c1[2], sigma[1]);
        }
        PG_STAMP(8);
#if defined(PG_STAMPS)
        if (a.dbg && a.dbg_stage == 99 && lane == 0 && it < 64) {
            for (int k = 0; k < 9; ++k) reinterpret_cast<unsigned long long*>(a.dbg)[((long long)it * 8 + wave) * 16 + k] = stamps[k];
            reinterpret_cast<unsigned long long*>(a.dbg)[((long long)it * 8 + wave) * 16 + 9] = st.t_vm;
            reinterpret_cast<unsigned long long*>(a.dbg)[((long long)it * 8 + wave) * 16 + 10] = st.t_bar;
            st.t_vm = 0; st.t_bar = 0;
        }
#endif
    }
    st.drain();
}

template <typename V, bool FC>
static hipError_t launch_eval16w(const EvalArgs& a, int grid, hipStream_t stream) {
    auto k = eval16w_kernel<V, FC>;
    static std::atomic<unsigned long long> attr_done{0};       // per device (pg_device.h)
    const hipError_t ae = ensure_lds_attr(reinterpret_cast<const void*>(k), LDS_TOTAL_F, attr_done);
    if (ae != hipSuccess) return ae;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHR_W), LDS_TOTAL_F, stream, a);
    return hipGetLastError();
}

}  // namespace pgd

// needs S >= pgl::FACT_MIN_S and the streams / tables of pg_launch_eval16(fact = 1)
extern "C" int pg_launch_eval16w(const pgd::EvalArgs* a, int fp16, int framecode, int grid, void* stream) {
    using namespace pgd;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e;
    if (fp16) e = framecode ? launch_eval16w<f16x8, true>(*a, grid, s) : launch_eval16w<f16x8, false>(*a, grid, s);
    else      e = framecode ? launch_eval16w<bf16x8, true>(*a, grid, s) : launch_eval16w<bf16x8, false>(*a, grid, s);
    return (int)e;
}
