// pg_eval32.hip -- fused bone-relative embedding + NeRF MLP with fp32-grade operands.
// Shape B of pg_program.h; same algorithm and reference mapping as pg_eval16.hip.
//
//   PG_PREC_FP32   : v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fma chain (parity mode)
//   PG_PREC_BF16X3 : every product as hi*hi + hi*lo + lo*hi of bf16 halves, fp32 accumulate
//   PG_PREC_FP16X3 : the same with fp16 halves
//   PG_PREC_FP16C  : compensated fp16, two products per MAC into one accumulator (PolC below)
//
// Workgroup = 4 waves (one per SIMD, up to 512 registers each): a wave keeps the fp32
// activations of its 32 points (128 registers) plus the 8 out-tile accumulators of the
// layer in flight (128 registers).  Every segment is k-major: one input unit (4 fp32
// values, or 8 split values, per lane) is multiplied into all out tiles, so split
// operands are converted once per unit.
#include "pg_device.h"

namespace pgd {
using namespace pgp::B;

constexpr int NWAVE_B = 4;
constexpr int NTHR_B = NWAVE_B * 64;
constexpr int PTS_B = NWAVE_B * 32;

template <bool FOLD> using Stream32T = Stream<NWAVE_B, (FOLD ? NCHUNK_FOLD : NCHUNK)>;

struct PolF32 {
    static constexpr bool FOLD = false;         // the parity mode composes the network as the reference does
    static constexpr int UE = 4;                // values per lane per unit
    static constexpr int UBYTES = 1024;
    static constexpr int UPC = CHUNK_BYTES / UBYTES;
    struct B { float v[4]; };
    static __device__ __forceinline__ B prep(const float* x) { return B{{x[0], x[1], x[2], x[3]}}; }
    static __device__ __forceinline__ float roundtrip(float x) { return x; }
    static __device__ __forceinline__ f32x16 mma(const uint8_t* u, const B& b, f32x16 acc) {
        const float4 a = *reinterpret_cast<const float4*>(u);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.v[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.v[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.v[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.v[3], acc, 0, 0, 0);
        return acc;
    }
};

template <typename V> struct Split;
template <> struct Split<bf16x8> {
    typedef __bf16 E;
    static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Split<f16x8> {
    typedef _Float16 E;
    static __device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

template <typename V>
struct PolX3 {
    static constexpr bool FOLD = true;          // feature_linear folded into the view layer (pg_pack.cpp)
    static constexpr int UE = 8;
    static constexpr int UBYTES = 2048;          // hi plane then lo plane
    static constexpr int UPC = CHUNK_BYTES / UBYTES;
    struct B { V hi, lo; };
    static __device__ __forceinline__ float roundtrip(float x) {
        const typename Split<V>::E hi = (typename Split<V>::E)x;
        return (float)hi + (float)(typename Split<V>::E)(x - (float)hi);
    }
    static __device__ __forceinline__ B prep(const float* x) {
        B b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const typename Split<V>::E hi = (typename Split<V>::E)x[j];
            b.hi[j] = hi;
            b.lo[j] = (typename Split<V>::E)(x[j] - (float)hi);
        }
        return b;
    }
    static __device__ __forceinline__ f32x16 mma(const uint8_t* u, const B& b, f32x16 acc) {
        const V ahi = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(u));
        const V alo = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(u + 1024));
        acc = Split<V>::mfma(alo, b.hi, acc);
        acc = Split<V>::mfma(ahi, b.lo, acc);
        acc = Split<V>::mfma(ahi, b.hi, acc);
        return acc;
    }
};

// Compensated fp16 (PG_PREC_FP16C): W x = (S-1) w1 x1 + w2 x2 with w = W / S, t1 = f16(t),
// t2 = f16(t1 + S (t - t1)): the second product carries the first-order rounding terms of both
// operands scaled by S, and (S-1) is a power of two, so both go into ONE fp32 accumulator.
// Error per product (S-1) a b + (e_w x + w e_x) / S ~ 2^-17 of |w x| (plain fp16: 2^-11).
struct PolC {
    static constexpr bool FOLD = true;
    static constexpr int UE = 8;
    static constexpr int UBYTES = 2048;          // (S-1) w1 plane then w2 plane (pg_pack.cpp)
    static constexpr int UPC = CHUNK_BYTES / UBYTES;
    struct B { f16x8 x1, x2; };
    static __device__ __forceinline__ float roundtrip(float x) {
        const _Float16 h = (_Float16)x;
        return (float)h + ((float)(_Float16)((float)h + (float)COMP_S * (x - (float)h)) - (float)h) / (float)COMP_S;
    }
    static __device__ __forceinline__ B prep(const float* x) {
        B b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b.x1[j] = (_Float16)x[j];
        // ONE conversion result must feed both the fragment and the residual x - x1: left visible,
        // hipcc (-ffp-contract=on) forms the fragment with v_cvt_pk_f16_f32 from the fp32 value and
        // the residual's copy with v_fma_mixlo_f16 from the exact product that produced x -- two
        // roundings that disagree on ties, and the compensation then has the wrong sign (2^-11).
        asm volatile("" : "+v"(b.x1));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float h = (float)b.x1[j];
            b.x2[j] = (_Float16)(h + (float)COMP_S * (x[j] - h));
        }
        return b;
    }
    static __device__ __forceinline__ f32x16 mma(const uint8_t* u, const B& b, f32x16 acc) {
        const f16x8 a1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(u));
        const f16x8 a2 = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(u + 1024));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b.x1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b.x2, acc, 0, 0, 0);
        return acc;
    }
};

// one input unit (values x[0..UE)) against NO out tiles; unit index uu of a k-major segment
template <typename P, int NO, typename ST>
__device__ __forceinline__ void mma_row(f32x16* acc, ST& st, int cbase, int uu, const float* x) {
    const typename P::B b = P::prep(x);
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        const int L = uu * NO + o;
        if (L % P::UPC == 0) st.enter(cbase + L / P::UPC);
        acc[o] = P::mma(st.at(cbase + L / P::UPC, (L % P::UPC) * P::UBYTES), b, acc[o]);
    }
}

// `nvals` consecutive lane values starting at sequence index i0 (multiples of UE)
template <typename P, int NO, typename ST>
__device__ __forceinline__ void feed(f32x16* acc, ST& st, int cbase, int i0, const float* x, int nvals) {
#pragma unroll
    for (int t = 0; t < nvals / P::UE; ++t) mma_row<P, NO>(acc, st, cbase, i0 / P::UE + t, x + t * P::UE);
}

template <typename P, typename ST>
__device__ __forceinline__ void x_segment(f32x16* acc, ST& st, int cbase, const float* slot,
                                          const float* cut, float tau, float px, float py, float pz, int h) {
#pragma unroll
    for (int sb = 0; sb < 3; ++sb) {
        float lo[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jj = 4 * sb + k;
            float x[18];
            joint_values<false>(slot + SLOT_SKT + (JH * h + jj) * 12, px, py, pz, tau, cut[JH * h + jj], x);
            lo[2 * k] = x[16];
            lo[2 * k + 1] = x[17];
            feed<P, NT>(acc, st, cbase, sb * 72 + k * 16, x, 16);
        }
        feed<P, NT>(acc, st, cbase, sb * 72 + 64, lo, 8);
    }
}

// acc[o] (+)= W * act over the `nvals` hidden values of this lane (k-major)
template <typename P, int NO, typename ST>
__device__ __forceinline__ void hidden_segment(f32x16* acc, ST& st, int cbase, const float* act, int nvals) {
    feed<P, NO>(acc, st, cbase, 0, act, nvals);
}

template <int NO>
__device__ __forceinline__ void store_act(const f32x16* acc, float* act, bool relu) {
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[16 * o + r] = relu ? fmaxf(acc[o][r], 0.0f) : acc[o][r];
}

// debug: activation values of this lane (sequence order, see hseq_channel) -> dbg[pt][256]
template <int NVAL>
__device__ __forceinline__ void dump_act(const EvalArgs& a, int stage, long long gp, bool valid, const float* act, int h) {
    if (a.dbg && a.dbg_stage == stage && valid) {
#pragma unroll
        for (int i = 0; i < NVAL; ++i) a.dbg[gp * W + hseq_channel(i, h)] = act[i];
    }
}

template <typename P, bool FC>
__global__ __launch_bounds__(NTHR_B, 1) void eval32_kernel(const EvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* bias = reinterpret_cast<float*>(smem + LDS_BIAS);
    float* cut = reinterpret_cast<float*>(smem + LDS_CUT);
    float* rtab = reinterpret_cast<float*>(smem + LDS_RTAB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, pt = lane & 31;
    Stream32T<P::FOLD> st{a.wstream, smem + LDS_RING, wave, lane, 0u, 0u, 0u,
               (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(smem + LDS_RING)};

    for (int i = tid; i < BIAS_FLOATS; i += NTHR_B) bias[i] = a.bias[i];
    if (tid < 48) cut[tid] = a.cutoff[tid];
    st.start();

    for (int it = blockIdx.x; it < a.n_iters; it += gridDim.x) {
        const long long p0 = (long long)it * PTS_B;
        const long long plast = min(p0 + PTS_B - 1, a.n_points - 1);
        const int r0 = (int)(p0 / a.S);
        const int nr = (int)(plast / a.S) - r0 + 1;
        lds_barrier();                          // previous pass is done with the table
        ray_table_phase1<NTHR_B>(a, rtab, r0, nr);
        lds_barrier();
        ray_table_phase2<NTHR_B, false>(rtab, nr);
        lds_barrier();

        const long long gp = p0 + wave * 32 + pt;
        const bool valid = gp < a.n_points;
        const long long gpc = valid ? gp : a.n_points - 1;
        const float* slot = rtab + ((int)(gpc / a.S) - r0) * SLOT_FLOATS;
        const float zz = a.pts ? 0.0f : a.z[gpc];
        // explicit points (pg_query_density) or p = o + d z as the reference forms it (mul, then add)
        float px = a.pts ? a.pts[gpc * 3 + 0] : __fadd_rn(slot[SLOT_O + 0], __fmul_rn(slot[SLOT_D + 0], zz));
        float py = a.pts ? a.pts[gpc * 3 + 1] : __fadd_rn(slot[SLOT_O + 1], __fmul_rn(slot[SLOT_D + 1], zz));
        float pz = a.pts ? a.pts[gpc * 3 + 2] : __fadd_rn(slot[SLOT_O + 2], __fmul_rn(slot[SLOT_D + 2], zz));
        if (a.pnoise) {     // pts + randn_like(pts) * ray_noise_std (raycasters.py:660-661)
            px = __fadd_rn(px, a.pnoise[gpc * 3]); py = __fadd_rn(py, a.pnoise[gpc * 3 + 1]); pz = __fadd_rn(pz, a.pnoise[gpc * 3 + 2]);
        }

        float act[HSEQ];
        f32x16 acc[NT];
        // ---- layer 0 ----
#pragma unroll
        for (int o = 0; o < NT; ++o) acc[o] = load_bias(bias, BT_LAYER0 + o, h);
        x_segment<P>(acc, st, C_L0, slot, cut, a.tau_v, px, py, pz, h);
        if (a.dbg && a.dbg_stage == 0 && valid) {
#pragma unroll
            for (int o = 0; o < NT; ++o)
#pragma unroll
                for (int r = 0; r < 16; ++r) a.dbg[gp * W + 32 * o + rho(r, h)] = acc[o][r];
        }
        store_act<NT>(acc, act, true);
        // ---- layers 1..4 ----
#pragma unroll
        for (int l = 1; l <= 4; ++l) {
#pragma unroll
            for (int o = 0; o < NT; ++o) acc[o] = load_bias(bias, BT_LAYER0 + l * NT + o, h);
            hidden_segment<P, NT>(acc, st, C_L1 + (l - 1) * CH_HID, act, HSEQ);
            store_act<NT>(acc, act, true);
            dump_act<HSEQ>(a, l, gp, valid, act, h);
        }
        // ---- layer 5 (skip) ----
#pragma unroll
        for (int o = 0; o < NT; ++o) acc[o] = load_bias(bias, BT_LAYER0 + 5 * NT + o, h);
        hidden_segment<P, NT>(acc, st, C_L5H, act, HSEQ);
        x_segment<P>(acc, st, C_L5X, slot, cut, a.tau_v, px, py, pz, h);
        store_act<NT>(acc, act, true);
        dump_act<HSEQ>(a, 5, gp, valid, act, h);
        // ---- layers 6, 7 ----
#pragma unroll
        for (int l = 6; l <= 7; ++l) {
#pragma unroll
            for (int o = 0; o < NT; ++o) acc[o] = load_bias(bias, BT_LAYER0 + l * NT + o, h);
            hidden_segment<P, NT>(acc, st, C_L6 + (l - 6) * CH_HID, act, HSEQ);
            store_act<NT>(acc, act, true);
            dump_act<HSEQ>(a, l, gp, valid, act, h);
        }
        // ---- feature (no activation) and sigma heads, both on h7; FOLD: no feature segment, the
        // view layer's trunk part multiplies h7 by W_view[:, :256] W_feature ----
        if (!P::FOLD) {
#pragma unroll
            for (int o = 0; o < NT; ++o) acc[o] = load_bias(bias, BT_FEAT + o, h);
            hidden_segment<P, NT>(acc, st, C_F, act, HSEQ);
        }
        f32x16 acc1 = load_bias(bias, BT_ALPHA, h);
        hidden_segment<P, 1>(&acc1, st, C_ALPHA, act, HSEQ);
        const float sigma = acc1[0];
        if (!P::FOLD) {
            store_act<NT>(acc, act, false);
            dump_act<HSEQ>(a, 8, gp, valid, act, h);
        }
        // ---- view layer ----
        f32x16 accv[NTV];
#pragma unroll
        for (int o = 0; o < NTV; ++o) accv[o] = load_bias(bias, (P::FOLD ? BT_VIEWF : BT_VIEW) + o, h);
        hidden_segment<P, NTV>(accv, st, C_VF, act, HSEQ);
        {
            float wd[JH];
#pragma unroll
            for (int jj = 0; jj < JH; ++jj)
                wd[jj] = cutoff_weight<false>(joint_dist<false>(slot + SLOT_SKT + (JH * h + jj) * 12, px, py, pz),
                                       a.tau_d, cut[J + JH * h + jj]);
            if (a.dbg && a.dbg_stage == 10 && valid) {
#pragma unroll
                for (int jj = 0; jj < JH; ++jj) a.dbg[gp * W + JH * h + jj] = wd[jj];
            }
            const float* tab = slot + SLOT_DTAB + h * DSEQ;
#pragma unroll
            for (int uu = 0; uu < DSEQ / 8; ++uu) {
                const float4 t0 = *reinterpret_cast<const float4*>(tab + uu * 8);
                const float4 t1 = *reinterpret_cast<const float4*>(tab + uu * 8 + 4);
                float x[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = uu < JH * 3 ? uu / 3 : (8 * (uu - JH * 3) + e) / 3;
                    x[e] = k < JH ? x[e] * wd[k] : 0.0f;
                }
                if (a.dbg && a.dbg_stage == 11 && valid && uu < 16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) a.dbg[gp * W + h * 128 + uu * 8 + e] = P::roundtrip(x[e]);
                }
                feed<P, NTV>(accv, st, C_VD, uu * 8, x, 8);
            }
            if (FC) {
                const float4 t0 = *reinterpret_cast<const float4*>(slot + SLOT_CODE + 8 * h);
                const float4 t1 = *reinterpret_cast<const float4*>(slot + SLOT_CODE + 8 * h + 4);
                const float x[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                feed<P, NTV>(accv, st, C_VD, DSEQ, x, 8);
            }
        }
        store_act<NTV>(accv, act, true);
        dump_act<VW / 2>(a, 9, gp, valid, act, h);
        // ---- rgb head ----
        f32x16 accr = load_bias(bias, BT_RGB, h);
        hidden_segment<P, 1>(&accr, st, C_RGB, act, VW / 2);
        if (valid && h == 0)
            *reinterpret_cast<float4*>(a.raw + gp * 4) = make_float4(accr[0], accr[1], accr[2], sigma);
        if (a.dbg && a.dbg_stage >= 12 && a.dbg_stage < 12 + DSEQ / 64 + 1 && valid) {
            // the view table as this lane finds it in LDS at the END of the pass: 64 values of its half
            const volatile float* tv = slot + SLOT_DTAB + h * DSEQ + (a.dbg_stage - 12) * 64;
            for (int e = 0; e < 64; ++e)
                if ((a.dbg_stage - 12) * 64 + e < DSEQ) a.dbg[gp * W + h * 64 + e] = tv[e];
        }
    }
    st.drain();
}

template <typename P, bool FC>
static hipError_t launch_eval32(const EvalArgs& a, int grid, hipStream_t stream) {
    auto k = eval32_kernel<P, FC>;
    static std::atomic<unsigned long long> attr_done{0};       // per device (pg_device.h)
    const hipError_t ae = ensure_lds_attr(reinterpret_cast<const void*>(k), LDS_TOTAL, attr_done);
    if (ae != hipSuccess) return ae;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHR_B), LDS_TOTAL, stream, a);
    return hipGetLastError();
}

}  // namespace pgd

extern "C" int pg_launch_eval32(const pgd::EvalArgs* a, int precision, int framecode, int grid, void* stream) {
    using namespace pgd;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipErrorInvalidValue;
    switch (precision) {
        case 0: /* PG_PREC_FP32 */
            e = framecode ? launch_eval32<PolF32, true>(*a, grid, s) : launch_eval32<PolF32, false>(*a, grid, s);
            break;
        case 2: /* PG_PREC_BF16X3 */
            e = framecode ? launch_eval32<PolX3<bf16x8>, true>(*a, grid, s) : launch_eval32<PolX3<bf16x8>, false>(*a, grid, s);
            break;
        case 4: /* PG_PREC_FP16X3 */
            e = framecode ? launch_eval32<PolX3<f16x8>, true>(*a, grid, s) : launch_eval32<PolX3<f16x8>, false>(*a, grid, s);
            break;
        case 5: /* PG_PREC_FP16C */
            e = framecode ? launch_eval32<PolC, true>(*a, grid, s) : launch_eval32<PolC, false>(*a, grid, s);
            break;
        default: break;
    }
    return (int)e;
}

extern "C" int pg_eval32_points_per_pass(void) { return pgd::PTS_B; }
