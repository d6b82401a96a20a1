// pg_comp.h -- the compensated-fp16 arithmetic (PG_PREC_FP16C) shared by the kernels that compute in it
// (pg_evalc.hip: points over waves, activations in registers; pg_evalc2.hip: out tiles over waves, activations in LDS):
// the fp16 pair of a value, a B-operand fragment pair, and the embedding values of one joint.
#pragma once
#include "pg_eval16_common.h"

namespace pgd {

using VC = f16x8;

// ---- the fp16 pair of a value: x1 = f16(x) (RNE), x2 = f16(x1 + S (x - x1)) ---------------------
// Both halves come from ONE conversion result: left to hipcc under -ffp-contract=on, the fragment
// is formed by v_cvt_pk_f16_f32 from the fp32 value and the residual's copy by v_fma_mixlo_f16
// from the exact product that produced the value -- two roundings that disagree on ties, after
// which the compensation has the wrong sign (an error of a full fp16 ulp).
// step A: two values -> their (optionally ReLU'd) fp32 values and the packed x1 pair
template <bool RELU>
__device__ __forceinline__ void conv_a(float a, float b, float& ra, float& rb, unsigned& h) {
    if (RELU) {
        asm("v_max_f32 %0, 0, %3\n\tv_max_f32 %1, 0, %4\n\tv_cvt_pk_f16_f32 %2, %0, %1"
            : "=&v"(ra), "=&v"(rb), "=&v"(h) : "v"(a), "v"(b));
    } else {
        ra = a; rb = b;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
    }
}
// step B: d = x - x1 (exact), t = S d + x1, x2 pair = f16(t).  v_fma_mix_f32 reads the fp16 halves of
// `h` directly.  The trailing s_nop 1 provides the wait states a VALU write needs before an MFMA may
// read the register (hipcc pads nothing for inline asm).
template <bool NOP = true>
__device__ __forceinline__ unsigned conv_b(float ra, float rb, unsigned h, float s) {
    unsigned x2;
    float da, db;
    if (NOP)
        asm("v_fma_mix_f32 %1, %3, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %2, %3, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %1, %1, %6, %3 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mix_f32 %2, %2, %6, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "v_cvt_pk_f16_f32 %0, %1, %2\n\ts_nop 1"
            : "=&v"(x2), "=&v"(da), "=&v"(db) : "v"(h), "v"(ra), "v"(rb), "s"(s));
    else        // the fragment is consumed a whole unit row later: no wait states needed
        asm("v_fma_mix_f32 %1, %3, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %2, %3, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %1, %1, %6, %3 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mix_f32 %2, %2, %6, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "v_cvt_pk_f16_f32 %0, %1, %2"
            : "=&v"(x2), "=&v"(da), "=&v"(db) : "v"(h), "v"(ra), "v"(rb), "s"(s));
    return x2;
}

struct FragC { unsigned x1[4], x2[4]; };       // one input unit: 8 values per lane as the two MFMA B operands

__device__ __forceinline__ VC frag_v(const unsigned* p) {
    const u32x4 v = {p[0], p[1], p[2], p[3]};
    return __builtin_bit_cast(VC, v);
}

// the 18 density-input values of one joint (joint_values_q of pg_device.h) for the compensated mode:
// hardware transcendentals, but every octave's sin/cos straight from v_sin/v_cos (revolutions, exact
// power-of-two argument scaling: 4.5e-7 absolute) instead of the angle-doubling chain, whose error
// doubles per octave (5e-5 at the 7th: profiles/r2_trig_err.txt) -- too coarse for this mode
__device__ __forceinline__ void joint_values_c(float qx, float qy, float qz, float tl, float cs, float* x) {
    const float d2 = qx * qx + qy * qy + qz * qz;
    const float rinv = __builtin_amdgcn_rsqf(fmaxf(d2, 1e-24f));
    const float v = d2 * rinv;
    const float w = cutoff_weight_fast(v, tl, cs);
    const float rev = v * 0.15915494309189535f;
    x[0] = v * w;
#pragma unroll
    for (int f = 0; f < LV; ++f) {
        const float a = rev * (float)(1 << f);
        x[1 + 2 * f] = __builtin_amdgcn_sinf(a) * w;
        x[2 + 2 * f] = __builtin_amdgcn_cosf(a) * w;
    }
    x[15] = qx * rinv; x[16] = qy * rinv; x[17] = qz * rinv;
}

__device__ __forceinline__ FragC frag_of(const float* x, float s129) {
    FragC f;
    float ra, rb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        conv_a<false>(x[2 * j], x[2 * j + 1], ra, rb, f.x1[j]);
        f.x2[j] = conv_b(ra, rb, f.x1[j], s129);
    }
    return f;
}

}  // namespace pgd
