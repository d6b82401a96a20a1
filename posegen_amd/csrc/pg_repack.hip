// pg_repack.hip -- the packed weight images of the fused inference kernels re-formed ON THE DEVICE from new parameter values
// (pg_load_weights_device: TrainableRayCaster.sync_inference_weights between optimiser steps and a validation render; the
// reference simply renders with the module it trains, core/trainer.py:463).
//
// The host packers (pg_pack.cpp) decide where every weight goes; asked to, they also record, per output element, the
// offset of its source in ONE flat vector of the net's tensors (NetTensors::flat: the 24 tensors in pg_load_weights order,
// then the folded view layer).  That map depends on the configuration only, is built once per handle and image, and a
// re-pack is then: copy the parameters into the flat vector, fold feature_linear into the view layer (the host's sums, in
// double, in its order), gather + convert.  Every conversion repeats the host's arithmetic operation by operation (no
// contraction: __dmul_rn / __dadd_rn), so the images are bitwise what pg_load_weights would have produced
// (tests/test_gpu_train.py::test_device_side_weight_sync_equals_the_host_packing).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pg_layout.h"
#include "pg_pack.h"

namespace pgr {
using namespace pgl;

__device__ __forceinline__ uint16_t f2bf_host(float f) {           // pg_pack.cpp f32_to_bf16
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint16_t f2h_host(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float h2f_host(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }

// the 24 parameter tensors -> the flat source vector, one launch (24 device-to-device copies cost the host a millisecond)
struct Collect { const float* p[24]; long long off[25]; };
__global__ __launch_bounds__(256) void collect_kernel(Collect c, float* __restrict__ dst) {
    const int i = blockIdx.y;
    const long long n = c.off[i + 1] - c.off[i];
    for (long long k = blockIdx.x * 256ll + threadIdx.x; k < n; k += (long long)gridDim.x * 256) dst[c.off[i] + k] = c.p[i][k];
}

// W_view[:, :256] W_feature and b_view + W_view[:, :256] b_feature (NetTensors::fold): block o = output row, thread k = column
__global__ __launch_bounds__(256) void fold_kernel(float* __restrict__ src, long long off_view_w, int vcols, long long off_view_b,
                                                   long long off_feat_w, long long off_feat_b, long long off_fw, long long off_fb) {
    const int o = blockIdx.x, k = threadIdx.x;
    const float* vr = src + off_view_w + (long long)o * vcols;
    double acc = 0.0;
    for (int m = 0; m < W; ++m) acc = __dadd_rn(acc, __dmul_rn((double)vr[m], (double)src[off_feat_w + (long long)m * W + k]));
    src[off_fw + (long long)o * W + k] = (float)acc;
    if (k == 0) {
        double b = (double)src[off_view_b + o];
        for (int m = 0; m < W; ++m) b = __dadd_rn(b, __dmul_rn((double)vr[m], (double)src[off_feat_b + m]));
        src[off_fb + o] = (float)b;
    }
}

// out[i] = convert(src[map[i] >> 2]) by kind map[i] & 3: a plain 16-bit value (bf16 or fp16) or one plane of the compensated
// pair of W = 129 w (pg_pack.cpp comp_pair: plane 0 = 128 w1, plane 1 = f16(w1 + 129 (w - w1)), w1 = f16(w)); -1: zero
__global__ __launch_bounds__(256) void gather16_kernel(const int32_t* __restrict__ map, const float* __restrict__ src, uint16_t* __restrict__ out,
                                                       long long n, int is_bf) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int32_t m = map[i];
        uint16_t r = 0;
        if (m >= 0) {
            const float wv = src[m >> 2];
            const int kind = m & 3;
            if (kind == pgpack::SRC_PLAIN) r = is_bf ? f2bf_host(wv) : f2h_host(wv);
            else {
                const double wd = (double)wv / (double)COMP_S;
                const double w1 = (double)h2f_host(f2h_host((float)wd));
                r = kind == pgpack::SRC_COMP0 ? f2h_host((float)__dmul_rn((double)(COMP_S - 1), w1))
                                              : f2h_host((float)__dadd_rn(w1, __dmul_rn((double)COMP_S, __dsub_rn(wd, w1))));
            }
        }
        out[i] = r;
    }
}
__global__ __launch_bounds__(256) void gather32_kernel(const int32_t* __restrict__ map, const float* __restrict__ src, float* __restrict__ out, long long n) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = map[i] >= 0 ? src[map[i] >> 2] : 0.0f;
}

// frame codes [n_codes + 1][16]: the rows, then their mean summed in row order (pg_set_framecodes, embedding.py:25-26)
__global__ __launch_bounds__(64) void codes_kernel(const float* __restrict__ codes, int n_codes, float* __restrict__ out) {
    const int c = threadIdx.x;
    if (c >= FC_CH) return;
    float s = 0.0f;
    for (int i = 0; i < n_codes; ++i) { const float v = codes[(long long)i * FC_CH + c]; out[(long long)i * FC_CH + c] = v; s += v; }
    out[(long long)n_codes * FC_CH + c] = s / (float)n_codes;
}
// Yc[c][o] = sum_k W_view[o][256 + 648 + k] codes[c][k] (pg_api.hip ensure_ycode), sums in double
__global__ __launch_bounds__(128) void ycode_kernel(const float* __restrict__ view_w, int vcols, const float* __restrict__ codes, float* __restrict__ yc) {
    const int c = blockIdx.x, o = threadIdx.x;
    double s = 0.0;
    for (int k = 0; k < FC_CH; ++k) s = __dadd_rn(s, __dmul_rn((double)view_w[(long long)o * vcols + W + CH_D + k], (double)codes[(long long)c * FC_CH + k]));
    yc[(long long)c * VW + o] = (float)s;
}
}  // namespace pgr

extern "C" {
void pg_launch_collect(const float* const* tensors, const long long* off25, float* dst, void* stream) {
    pgr::Collect c;
    for (int i = 0; i < 24; ++i) c.p[i] = tensors[i];
    for (int i = 0; i < 25; ++i) c.off[i] = off25[i];
    hipLaunchKernelGGL(pgr::collect_kernel, dim3(64, 24), dim3(256), 0, static_cast<hipStream_t>(stream), c, dst);
}
void pg_launch_fold(float* src, long long off_view_w, int vcols, long long off_view_b, long long off_feat_w, long long off_feat_b,
                    long long off_fw, long long off_fb, void* stream) {
    hipLaunchKernelGGL(pgr::fold_kernel, dim3(pgl::VW), dim3(256), 0, static_cast<hipStream_t>(stream), src, off_view_w, vcols, off_view_b, off_feat_w,
                       off_feat_b, off_fw, off_fb);
}
void pg_launch_gather16(const int32_t* map, const float* src, uint16_t* out, long long n, int is_bf, void* stream) {
    hipLaunchKernelGGL(pgr::gather16_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, static_cast<hipStream_t>(stream), map, src, out, n, is_bf);
}
void pg_launch_gather32(const int32_t* map, const float* src, float* out, long long n, void* stream) {
    hipLaunchKernelGGL(pgr::gather32_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, static_cast<hipStream_t>(stream), map, src, out, n);
}
void pg_launch_codes(const float* codes, int n_codes, float* out, void* stream) {
    hipLaunchKernelGGL(pgr::codes_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), codes, n_codes, out);
}
void pg_launch_ycode(const float* view_w, int vcols, const float* codes, int n_codes, float* yc, void* stream) {
    hipLaunchKernelGGL(pgr::ycode_kernel, dim3(n_codes + 1), dim3(128), 0, static_cast<hipStream_t>(stream), view_w, vcols, codes, yc);
}
}
