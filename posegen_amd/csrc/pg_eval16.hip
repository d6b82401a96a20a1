// pg_eval16.hip -- fused bone-relative embedding + NeRF MLP, 16-bit MFMA operands
// (bf16 or fp16, fp32 accumulate).  Shape A of pg_program.h.
//
// Replaces RayCaster.encode_inputs + run_network + NeRF.forward for one net
// (reference core/raycasters.py:476-577, core/networks/nerf.py:90-148,
// core/encoders.py, core/cutoff_embedder.py) on n*S points p = o + d*z.
//
// Workgroup = 8 waves (2 per SIMD), each wave owns 32 consecutive points of the
// flattened [ray][sample] list and keeps their activations in registers across all
// layers (pg_layout.h).  All 8 waves consume the same weight stream, staged
// L2 -> LDS in 32-KiB chunks by global_load_lds_dwordx4 into a 3-slot ring (two chunks
// in flight while one feeds v_mfma_f32_32x32x16).  The 432-wide density input is
// produced on the fly as B fragments (never stored) and recomputed for the skip layer.
// The 648-wide view input is a per-ray sin/cos table in LDS times the per-point cutoff weight (the
// "direct" view layer); feature_linear is folded into the view layer by the host packer.
// This is the 16-bit kernel for rays with < 64 samples, explicit points (density queries) and position
// noise; rays with >= 64 samples run pg_eval16r.hip (16x16x32 MFMAs, per-ray records).
#include "pg_eval16_common.h"

namespace pgd {
__device__ __forceinline__ int h_abl(int lane) { return lane >> 5; }
using namespace pgp::A;

// TAPS = the debug taps of pg_stage_eval compiled in: a separate instantiation, launched only when a dump
// is asked for (the cold dump blocks otherwise cost the production kernel spilled registers, and every
// scratch reload drains the weight DMA with its vmcnt(0))
template <typename V, bool FC, bool TAPS>
__global__ __launch_bounds__(NTHR, 2) void eval16_kernel(const EvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* bias = reinterpret_cast<float*>(smem + LDS_BIAS);
    float* cut = reinterpret_cast<float*>(smem + LDS_CUT);
    float* rtab = reinterpret_cast<float*>(smem + LDS_RTAB);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, pt = lane & 31;
    StreamA st{a.wstream, smem + LDS_RING, wave, lane, 0u, 0u, 0u,
               (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(smem + LDS_RING), (uint32_t)lane * 16u};

    for (int i = tid; i < BIAS_FLOATS; i += NTHR) bias[i] = a.bias[i];
    // cutoff table with the sigmoid constants folded in (cutoff_weight_fast)
    const float tlv = a.tau_v * 1.4426950408889634f, tld = a.tau_d * 1.4426950408889634f;
    if (tid < 48) cut[tid] = -a.cutoff[tid] * (tid < J ? tlv : tld);
    st.start();
#if defined(PG_YOUNG_PRIO)
    // the second-dispatched wave of each SIMD loses every issue arbitration to the older one and is what the
    // older one waits for at the chunk barriers: one static priority for that half (no per-segment flips)
    if (wave >= NWAVE / 2) __builtin_amdgcn_s_setprio(PG_YOUNG_PRIO);
#endif

#if defined(PG_STAMPS)
    unsigned long long stamps[12];
#endif
    for (int it = blockIdx.x; it < a.n_iters; it += gridDim.x) {
        PG_STAMP(0);
        const long long p0 = (long long)it * PTS;
        const long long plast = min(p0 + PTS - 1, a.n_points - 1);
        const int r0 = (int)(p0 / a.S);
        const int nr = (int)(plast / a.S) - r0 + 1;
        lds_barrier();                          // previous pass is done with the table
        ray_table_phase1<NTHR>(a, rtab, r0, nr);
        lds_barrier();
        ray_table_phase2<NTHR, true>(rtab, nr);
        lds_barrier();

        const long long gp = p0 + wave * 32 + pt;
        const bool valid = gp < a.n_points;
        const long long gpc = valid ? gp : a.n_points - 1;
        const int myr = (int)(gpc / a.S) - r0;
        const float* slot = rtab + myr * SLOT_FLOATS;
        const float* od = slot + SLOT_O;
        // Lane-dependent bases made opaque: left visible, hipcc materialises one address
        // register per joint (base + h-dependent offset) and spills them; a scratch reload
        // then waits vmcnt(0), i.e. drains the weight DMA that must stay in flight.
        const float* skb = opaque_ptr(slot + SLOT_SKT + JH * h * 12);
        const float* cutv = opaque_ptr(cut + JH * h);
        const float* cutd = opaque_ptr(cut + J + JH * h);
        const float* tab = opaque_ptr(slot + SLOT_DTAB + h * DSEQ);
        const float zz = a.pts ? 0.0f : a.z[gpc];
        // p = o + d z as the reference forms it (mul, then add; raycasters.py:658)
        float px = 0.0f, py = 0.0f, pz = 0.0f;
        {
            if (a.pts) {        // density query on explicit points (pg_query_density)
                px = a.pts[gpc * 3]; py = a.pts[gpc * 3 + 1]; pz = a.pts[gpc * 3 + 2];
            } else {
                px = __fadd_rn(od[0], __fmul_rn(od[3], zz));
                py = __fadd_rn(od[1], __fmul_rn(od[4], zz));
                pz = __fadd_rn(od[2], __fmul_rn(od[5], zz));
                if (a.pnoise) {     // pts + randn_like(pts) * ray_noise_std (raycasters.py:660-661)
                    px = __fadd_rn(px, a.pnoise[gpc * 3]); py = __fadd_rn(py, a.pnoise[gpc * 3 + 1]); pz = __fadd_rn(pz, a.pnoise[gpc * 3 + 2]);
                }
            }
        }
        const QFromRows q_rows{skb, px, py, pz};

        PG_STAMP(1);
        V fa[HU], fb[HU];
        {   // ---- layer 0: K = 432 generated on the fly, all 8 out tiles live ----
            f32x16 acc[NT];
#pragma unroll
            for (int o = 0; o < NT; ++o) acc[o] = load_bias(bias, BT_LAYER0 + o, h);
            x_segment<V>(acc, st, C_L0, q_rows, cutv, tlv);
            if (TAPS && a.dbg && a.dbg_stage == 0 && valid) {
#pragma unroll
                for (int o = 0; o < NT; ++o)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a.dbg[gp * W + 32 * o + rho(r, h)] = acc[o][r];
            }
#pragma unroll
            for (int o = 0; o < NT; ++o) relu_pack<V>(acc[o], fa[2 * o], fa[2 * o + 1], true);
        }
        PG_STAMP(2);
        // ---- layers 1..4 ----
        hidden_layer<V>(fa, fb, st, C_L1 + 0 * CH_HID, bias, BT_LAYER0 + 1 * NT, h);
        hidden_layer<V>(fb, fa, st, C_L1 + 1 * CH_HID, bias, BT_LAYER0 + 2 * NT, h);
        hidden_layer<V>(fa, fb, st, C_L1 + 2 * CH_HID, bias, BT_LAYER0 + 3 * NT, h);
        hidden_layer<V>(fb, fa, st, C_L1 + 3 * CH_HID, bias, BT_LAYER0 + 4 * NT, h);
        PG_STAMP(3);
        {   // ---- layer 5: [x(432), h4(256)] -> 256 (skip connection, nerf.py:99-101) ----
            f32x16 acc[NT];
            APipeX<V> p5;
#pragma unroll
            for (int o = 0; o < NT; ++o) {
                acc[o] = load_bias(bias, BT_LAYER0 + 5 * NT + o, h);
                row_tile<V, HU * NT, PG_ASYNC_X>(acc[o], p5, st, o, fa);
            }
            x_segment<V>(acc, st, C_L5X, q_rows, cutv, tlv);
#pragma unroll
            for (int o = 0; o < NT; ++o) relu_pack<V>(acc[o], fb[2 * o], fb[2 * o + 1], true);
        }
        PG_STAMP(4);
        hidden_layer<V>(fb, fa, st, C_L6 + 0 * CH_HID, bias, BT_LAYER0 + 6 * NT, h);
        hidden_layer<V>(fa, fb, st, C_L6 + 1 * CH_HID, bias, BT_LAYER0 + 7 * NT, h);
        PG_STAMP(5);
        // ---- sigma head and the view layer's trunk part.  feature_linear has no activation, so
        // the host folds it into the view weights (pg_pack.cpp NetTensors::fold): one segment of
        // 1 + 4 out tiles on the last trunk activation instead of 9 + 4 ----
        float sigma;
        if (TAPS) dump_frags<V, HU>(a, 7, gp, valid, fb, h);
        V fg[HU / 2];
        {
            f32x16 acc[NTV];
            APipe<V> pv;
            APipeX<V> pd;
            constexpr int TVD = (DU + (FC ? 1 : 0)) * NTV;
            constexpr int TAV = HU * (NTV + 1);
            {
                f32x16 a0 = load_bias(bias, BT_ALPHA, h);
                row_tile<V, TAV, true>(a0, pv, st, 0, fb);
                sigma = a0[0];
            }
            PG_STAMP(6);
#pragma unroll
            for (int o = 0; o < NTV; ++o) {
                acc[o] = load_bias(bias, BT_VIEWF + o, h);
                row_tile<V, TAV, true>(acc[o], pv, st, 1 + o, fb);
            }
            float wd[JH];
#pragma unroll
            for (int jj = 0; jj < JH; ++jj)
            {
                float qx, qy, qz;
                q_rows(jj, qx, qy, qz);
                wd[jj] = cutoff_weight_fast(__builtin_amdgcn_sqrtf(qx * qx + qy * qy + qz * qz), tld, cutd[jj]);
            }
            {
    #pragma clang loop unroll(full)
                for (int uu = 0; uu < DU; ++uu) {
                    const float4 t0 = *reinterpret_cast<const float4*>(tab + uu * 8);
                    const float4 t1 = *reinterpret_cast<const float4*>(tab + uu * 8 + 4);
                    float x[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    #pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = uu < JH * 3 ? uu / 3 : (8 * (uu - JH * 3) + e) / 3;
                        x[e] = k < JH ? x[e] * wd[k] : 0.0f;
                    }
                    mma_row<V, NTV, TVD, PG_ASYNC_VD>(acc, pd, st, uu, Op<V>::cvt(x));
                }
                if (FC) {
                    const float4 t0 = *reinterpret_cast<const float4*>(slot + SLOT_CODE + 8 * h);
                    const float4 t1 = *reinterpret_cast<const float4*>(slot + SLOT_CODE + 8 * h + 4);
                    const float x[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                    mma_row<V, NTV, TVD, PG_ASYNC_VD>(acc, pd, st, DU, Op<V>::cvt(x));
                }
            }
#pragma unroll
            for (int o = 0; o < NTV; ++o) relu_pack<V>(acc[o], fg[2 * o], fg[2 * o + 1], true);
        }
        PG_STAMP(7);
        if (TAPS) dump_frags<V, HU / 2>(a, 9, gp, valid, fg, h);
        // ---- rgb head ----
        f32x16 acc = load_bias(bias, BT_RGB, h);
        {
            APipe<V> pr;
#pragma unroll
            for (int u = 0; u < HU / 2; ++u) acc = Op<V>::mfma(next_a<V, HU / 2, true, PG_PIPE_H>(pr, st, u), fg[u], acc);
        }
        if (valid && h == 0)
            *reinterpret_cast<float4*>(a.raw + gp * 4) = make_float4(acc[0], acc[1], acc[2], sigma);
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

template <typename V, bool FC, bool TAPS>
static hipError_t launch_eval16(const EvalArgs& a, int grid, hipStream_t stream) {
    auto k = eval16_kernel<V, FC, TAPS>;
    static std::atomic<unsigned long long> attr_done{0};       // per device (pg_device.h)
    const hipError_t ae = ensure_lds_attr(reinterpret_cast<const void*>(k), LDS_TOTAL, attr_done);
    if (ae != hipSuccess) return ae;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHR), LDS_TOTAL, stream, a);
    return hipGetLastError();
}

}  // namespace pgd

template <typename V>
static hipError_t dispatch_eval16(const pgd::EvalArgs& a, int framecode, int grid, hipStream_t s) {
    using namespace pgd;
    if (a.dbg && a.dbg_stage != 99) return framecode ? launch_eval16<V, true, true>(a, grid, s) : launch_eval16<V, false, true>(a, grid, s);
    return framecode ? launch_eval16<V, true, false>(a, grid, s) : launch_eval16<V, false, false>(a, grid, s);
}

// the direct-view 16-bit kernel: rays with < pgl::FACT_MIN_S samples, explicit points, position noise
extern "C" int pg_launch_eval16(const pgd::EvalArgs* a, int fp16, int framecode, int grid, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    return (int)(fp16 ? dispatch_eval16<pgd::f16x8>(*a, framecode, grid, s) : dispatch_eval16<pgd::bf16x8>(*a, framecode, grid, s));
}

extern "C" int pg_eval16_points_per_pass(void) { return pgd::PTS; }
extern "C" int pg_eval16_wgs_per_cu(void) { return 8 / pgd::NWAVE; }
