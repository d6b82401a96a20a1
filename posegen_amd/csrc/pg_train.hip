// pg_train.hip -- the A-NeRF training step on the device: forward with the activations kept, and the backward pass
// through compositing, the 8x256 MLP + heads and the cutoff embedding's inputs (second half).
//
// Replaces, for one ray batch, `render(..., **render_kwargs_train)` + `loss.backward()` of Trainer.train_batch
// (reference core/trainer.py:232-275, 463): the forward of RayCaster.render_rays in training mode
// (core/raycasters.py:361-474 with perturb / raw_noise_std / ray_noise_std) and the gradient of a scalar loss of
// (rgb_map, acc_map, rgb0, acc0) -- what Trainer.compute_loss reads (trainer.py:321-383) -- with respect to every
// tensor of both networks (core/networks/nerf.py:57-88) and the frame codes (core/networks/embedding.py).  The
// importance samples are constants of the backward pass, as in the reference (`z_samples.detach()`,
// core/utils/ray_utils.py:285); poses, rays and the embedder's cutoff parameters get no gradient (the reference's
// cutoff_dist has requires_grad=False; pose optimisation is out of scope).
//
// Training batches are small (N_rand = 2048 rays -> 131 k + 164 k points, configs/surreal/surreal.txt:34), so the
// 1080-wide embedding and the layer activations are MATERIALISED in HBM (the tape: 14 KB per point in fp32) and every layer
// is a plain GEMM; the weight gradients split K over the points and are reduced in a fixed order (bitwise repeatable).
// The fused inference kernels are not involved.  Two modes (pg_set_train_precision; the rendering precision plays no part):
//   * fp32 (parity with the reference's autograd): v_mfma_f32_32x32x2_f32, 128 x 128 x 16 tiles (sgemm128_kernel), a
//     64-tile kernel for small and unaligned shapes (sgemm_kernel);
//   * 16-bit (PG_PREC_BF16, opt-in): the tape's activations and activation gradients are STORED in bf16 (the embedding rows, the
//     layer outputs, dH, dG) and the large GEMMs multiply bf16 operands on v_mfma_f32_32x32x16_bf16 with fp32
//     accumulation (bgemm128_kernel, 128 x 128 x 64 tiles; the weights as bf16 copies made once per step); weights,
//     biases, raw, d_raw, every weight gradient and the partial sums of the two-part layers stay fp32.
// Both: one-pass kernels for the alpha / rgb heads (skinny_fwd_kernel, skinny_dw_kernel).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pg_handle.h"

extern "C" {
int pg_launch_sample_coarse(const float* rays, const float* cyls, long long cyl_stride, long long n, int chunk,
                            int S, int lindisp, float* near_far, float* z, const float* t_rand, double* scratch, void* stream);
int pg_launch_gather_noise(const float* src, long long n, int stride, int S, const int* order, float* dst, void* stream);
int pg_launch_composite(const float* rays, const float* z, const float* raw, long long n, int S,
                        float density_scale, float rgb_eps, int density_act, float act_shift, float* rgb, float* disp, float* acc,
                        float* alpha, float* weights, int n_imp, float* z_fine, const float* noise, const float* u_rand, int* order,
                        void* stream);
int pg_composite_max_samples(void);
int pg_composite_max_importance(void);
}

namespace pgt {
using namespace pgl;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int XW = CH_X + CH_D + FC_CH;       // 1096: row of the materialised input (density | view | frame code or zeros)

// element types of the tape: fp32, or bf16 bits in an unsigned short
typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(bf16_t b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float ld_el(const void* p, long long i, bool bf) {
    return bf ? bf2f(static_cast<const bf16_t*>(p)[i]) : static_cast<const float*>(p)[i];
}
__device__ __forceinline__ void st_el(void* p, long long i, float v, bool bf) {
    if (bf) static_cast<bf16_t*>(p)[i] = f2bf(v); else static_cast<float*>(p)[i] = v;
}
template <typename T> __device__ __forceinline__ void put(T* x, int i, float v);
template <> __device__ __forceinline__ void put<float>(float* x, int i, float v) { x[i] = v; }
template <> __device__ __forceinline__ void put<bf16_t>(bf16_t* x, int i, float v) { x[i] = f2bf(v); }
enum { DT_A = 1, DT_B = 2, DT_C = 4, DT_M = 8 };        // which operands of a GEMM are bf16: A, B, the result, the ReLU mask

// ---- the 1080 (+16) network inputs of every point, one thread per (point, joint) --------------------------------
// RelDist / VecNorm encoders on bone-local coordinates + cutoff positional embedding, channel order of the reference:
// v part row * 24 + j, direction part 360 + 3 j + c, view part 432 + row * 72 + 3 j + c (core/encoders.py:8-37,
// 101-122, 172-193; core/cutoff_embedder.py:111-174).  sin / cos of every octave directly (no doubling chain).
template <typename XT>
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ rays, const float* __restrict__ z,
                                                         const float* __restrict__ pnoise, const float* __restrict__ skts,
                                                         long long pose_stride, const float* __restrict__ cams,
                                                         const float* __restrict__ codes, int n_codes, int fc,
                                                         const float* __restrict__ cutoff, float tau_v, float tau_d,
                                                         long long n_points, int S, XT* __restrict__ X) {
    const long long idx = blockIdx.x * 256ll + threadIdx.x;
    if (idx >= n_points * J) return;
    const long long pt = idx / J;
    const int j = (int)(idx - pt * J);
    const long long ray = pt / S;
    const float* rb = rays + ray * 11;
    const float zz = z[pt];
    float px = __fadd_rn(rb[0], __fmul_rn(rb[3], zz)), py = __fadd_rn(rb[1], __fmul_rn(rb[4], zz)), pz = __fadd_rn(rb[2], __fmul_rn(rb[5], zz));
    if (pnoise) { px = __fadd_rn(px, pnoise[pt * 3]); py = __fadd_rn(py, pnoise[pt * 3 + 1]); pz = __fadd_rn(pz, pnoise[pt * 3 + 2]); }
    const float* sk = skts + ray * pose_stride + j * 16;
    const float qx = fmaf(sk[2], pz, fmaf(sk[1], py, fmaf(sk[0], px, sk[3])));
    const float qy = fmaf(sk[6], pz, fmaf(sk[5], py, fmaf(sk[4], px, sk[7])));
    const float qz = fmaf(sk[10], pz, fmaf(sk[9], py, fmaf(sk[8], px, sk[11])));
    const float v = sqrtf(qx * qx + qy * qy + qz * qz);
    XT* x = X + pt * XW;
    {
        const float w = 1.0f - 1.0f / (1.0f + expf(-tau_v * (v - cutoff[j])));
        put(x, j, v * w);
        float f = 1.0f;
#pragma unroll
        for (int k = 0; k < LV; ++k, f *= 2.0f) {
            float s, c;
            sincosf(f * v, &s, &c);
            put(x, (1 + 2 * k) * J + j, s * w);
            put(x, (2 + 2 * k) * J + j, c * w);
        }
        const float den = fmaxf(v, 1e-12f);
        put(x, CH_V + 3 * j, qx / den); put(x, CH_V + 3 * j + 1, qy / den); put(x, CH_V + 3 * j + 2, qz / den);
    }
    {
        const float dx = rb[3], dy = rb[4], dz = rb[5];
        float e[3] = {fmaf(sk[2], dz, fmaf(sk[1], dy, sk[0] * dx)), fmaf(sk[6], dz, fmaf(sk[5], dy, sk[4] * dx)),
                      fmaf(sk[10], dz, fmaf(sk[9], dy, sk[8] * dx))};
        const float den = fmaxf(sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), 1e-12f);
        const float w = 1.0f - 1.0f / (1.0f + expf(-tau_d * (v - cutoff[J + j])));
        XT* xd = x + CH_X;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ev = e[c] / den;
            put(xd, 3 * j + c, ev * w);
            float f = 1.0f;
#pragma unroll
            for (int k = 0; k < LD; ++k, f *= 2.0f) {
                float s, co;
                sincosf(f * ev, &s, &co);
                put(xd, (1 + 2 * k) * (3 * J) + 3 * j + c, s * w);
                put(xd, (2 + 2 * k) * (3 * J) + 3 * j + c, co * w);
            }
        }
    }
    if (j == 0) {
        XT* xc = x + CH_X + CH_D;
        if (fc) {
            const float cam = cams ? cams[ray] : -1.0f;
            const int ci = cam < 0.0f ? n_codes : min((int)cam, n_codes - 1);      // row n_codes = the mean code
#pragma unroll
            for (int k = 0; k < FC_CH; ++k) put(xc, k, codes[ci * FC_CH + k]);
        } else {
#pragma unroll
            for (int k = 0; k < FC_CH; ++k) put(xc, k, 0.0f);
        }
    }
}

// ---- fp32 GEMM: C[M,N] (op)= A[M,K] B[K,N] with element strides, 64 x 64 x 16 tiles on v_mfma_f32_32x32x2_f32 ------
// A(m,k) = A[m sam + k sak], B(k,n) = B[k sbk + n sbn]; the template flags say which index is contiguous (coalesced
// tile loads).  gridDim.z > 1 splits K: the partial tiles go to a scratch array and are summed in slice order.
constexpr int GB = 64, GK = 16;
enum { GEMM_ACC = 1, GEMM_RELU = 2 };

// dt (DT_*): which of A, B, C are bf16 in HBM (this kernel multiplies in fp32 whatever they are stored as); GEMM_ACC adds
// cin[m ldcin + n] (fp32; the result array itself in the fp32 mode)
template <bool A_KCONT, bool B_KCONT>
__global__ __launch_bounds__(256) void sgemm_kernel(int M, int N, int K, const void* __restrict__ A, long long sam, long long sak,
                                                    const void* __restrict__ B, long long sbk, long long sbn,
                                                    void* __restrict__ C, long long ldc, const float* __restrict__ bias, int flags,
                                                    float* __restrict__ part, int dt, const float* cin, long long ldcin) {
    const bool abf = dt & DT_A, bbf = dt & DT_B;
    // gridDim.z > 1: K is split; slice z writes its tile to part[z][M][N] and reduce_parts_kernel adds the slices in
    // order (bitwise repeatable, unlike float atomics)
    __shared__ float As[GK][GB + 4], Bs[GK][GB + 4];
    const int t = threadIdx.x;
    // wave w owns the 32 x 32 quadrant (w >> 1, w & 1) of the tile: v_mfma_f32_32x32x2_f32, A lane (row li, k kh), B lane
    // (k kh, column li), C lane (column li) x 16 rows rho(r, kh)
    const int lane = t & 63, wv = t >> 6, mq = (wv >> 1) * 32, nq = (wv & 1) * 32, li = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * GB, n0 = blockIdx.x * GB;
    const int nz = gridDim.z;
    const int kper = ((K + nz - 1) / nz + GK - 1) / GK * GK;
    const int k0 = blockIdx.z * kper, k1 = min(K, k0 + kper);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int kb = k0; kb < k1; kb += GK) {
        if (A_KCONT) {
            const int m = t >> 2, kq = (t & 3) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = kb + kq + i;
                As[kq + i][m] = (m0 + m < M && k < k1) ? ld_el(A, (long long)(m0 + m) * sam + (long long)k * sak, abf) : 0.0f;
            }
        } else {
            const int m = (t & 15) * 4, k = kb + (t >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                As[t >> 4][m + i] = (m0 + m + i < M && k < k1) ? ld_el(A, (long long)(m0 + m + i) * sam + (long long)k * sak, abf) : 0.0f;
        }
        if (B_KCONT) {
            const int n = t >> 2, kq = (t & 3) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = kb + kq + i;
                Bs[kq + i][n] = (n0 + n < N && k < k1) ? ld_el(B, (long long)k * sbk + (long long)(n0 + n) * sbn, bbf) : 0.0f;
            }
        } else {
            const int n = (t & 15) * 4, k = kb + (t >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                Bs[t >> 4][n + i] = (n0 + n + i < N && k < k1) ? ld_el(B, (long long)k * sbk + (long long)(n0 + n + i) * sbn, bbf) : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK / 2; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * kk + kh][mq + li], Bs[2 * kk + kh][nq + li], acc, 0, 0, 0);
        __syncthreads();
    }
    const int n = n0 + nq + li;
    if (n < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + mq + rho(r, kh);
            if (m >= M) continue;
            float v = acc[r];
            if (nz > 1) { part[((long long)blockIdx.z * M + m) * N + n] = v; continue; }
            if (flags & GEMM_ACC) v += cin[(long long)m * ldcin + n];
            if (bias) v += bias[n];
            if (flags & GEMM_RELU) v = fmaxf(v, 0.0f);
            st_el(C, (long long)m * ldc + n, v, dt & DT_C);
        }
    }
}

// ---- the same GEMM for the large shapes (M, N >= 64; every stride and pointer a multiple of 4 floats): 128 x 128 x 16
// tiles, a wave owns 64 x 64 (2 x 2 MFMA tiles: one LDS read per MFMA), 16-byte global loads, the next k-step's
// operands in flight in registers while the current one is multiplied.
constexpr int TB = 128;

template <bool KCONT>
__device__ __forceinline__ void tile_fetch(float4 (&v)[2], const float* __restrict__ P, long long s_row, long long s_k,
                                           int row0, int rows, int kb, int k1, int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = t + 256 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KCONT) {            // 4 consecutive k of one row
            const int r = idx >> 2, k = kb + (idx & 3) * 4;
            if (row0 + r < rows && k < k1) v[i] = *reinterpret_cast<const float4*>(P + (long long)(row0 + r) * s_row + k);
        } else {                // 4 consecutive rows of one k
            const int k = kb + (idx >> 5), r = (idx & 31) * 4;
            if (row0 + r < rows && k < k1) v[i] = *reinterpret_cast<const float4*>(P + (long long)k * s_k + row0 + r);
        }
    }
}

template <bool KCONT>
__device__ __forceinline__ void tile_store(float (*T)[TB + 4], const float4 (&v)[2], int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = t + 256 * i;
        if (KCONT) {
            const int r = idx >> 2, kq = (idx & 3) * 4;
            T[kq][r] = v[i].x; T[kq + 1][r] = v[i].y; T[kq + 2][r] = v[i].z; T[kq + 3][r] = v[i].w;
        } else {
            *reinterpret_cast<float4*>(&T[idx >> 5][(idx & 31) * 4]) = v[i];
        }
    }
}

template <bool A_KCONT, bool B_KCONT>
__global__ __launch_bounds__(256) void sgemm128_kernel(int M, int N, int K, const float* __restrict__ A, long long sam, long long sak,
                                                       const float* __restrict__ B, long long sbk, long long sbn,
                                                       float* __restrict__ C, long long ldc, const float* __restrict__ bias, int flags,
                                                       const float* __restrict__ mask, long long ldm, float* __restrict__ rowsum,
                                                       float* __restrict__ part, float* __restrict__ rs_part, const float* cin, long long ldcin) {
    // mask: C(m,n) is zeroed where mask[m ldm + n] <= 0 (the ReLU backward of the layer that consumes C, fused);
    // rowsum (A m-contiguous only): sum_k A(m,k) (the bias gradient beside a weight gradient): block (x, z) writes the
    // share of its k-steps to rs_part[z * gridDim.x + x][M]; reduce_parts_kernel adds the shares in order.
    // gridDim.z > 1: slice z's tile goes to part[z][M][N] (summed in slice order by reduce_parts_kernel)
    __shared__ __attribute__((aligned(16))) float As[GK][TB + 4], Bs[GK][TB + 4];
    float rs = 0.0f;
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6, wm = (wv >> 1) * 64, wn = (wv & 1) * 64, li = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * TB, n0 = blockIdx.x * TB;
    const int nz = gridDim.z;
    const int kper = ((K + nz - 1) / nz + GK - 1) / GK * GK;
    const int k0 = blockIdx.z * kper, k1 = min(K, k0 + kper);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float4 va[2], vb[2];
    tile_fetch<A_KCONT>(va, A, sam, sak, m0, M, k0, k1, t);
    tile_fetch<B_KCONT>(vb, B, sbn, sbk, n0, N, k0, k1, t);
    for (int kb = k0; kb < k1; kb += GK) {
        tile_store<A_KCONT>(As, va, t);
        tile_store<B_KCONT>(Bs, vb, t);
        __syncthreads();
        tile_fetch<A_KCONT>(va, A, sam, sak, m0, M, kb + GK, k1, t);        // (all zeros past the end)
        tile_fetch<B_KCONT>(vb, B, sbn, sbk, n0, N, kb + GK, k1, t);
        // (the block columns of a tile row share the row sums: k-step s belongs to column s mod gridDim.x)
        if (!A_KCONT && rowsum && (unsigned)((kb - k0) / GK) % gridDim.x == blockIdx.x && t < TB) {
#pragma unroll
            for (int kk = 0; kk < GK; ++kk) rs += As[kk][t];
        }
#pragma unroll
        for (int kk = 0; kk < GK / 2; ++kk) {
            const float a0 = As[2 * kk + kh][wm + li], a1 = As[2 * kk + kh][wm + 32 + li];
            const float b0 = Bs[2 * kk + kh][wn + li], b1 = Bs[2 * kk + kh][wn + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + 32 * j + li;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + rho(r, kh);
                if (m >= M) continue;
                float* c = C + (long long)m * ldc + n;
                float v = acc[i][j][r];
                if (nz > 1) { part[((long long)blockIdx.z * M + m) * N + n] = v; continue; }
                if (flags & GEMM_ACC) v += cin[(long long)m * ldcin + n];
                v += bv;
                if (flags & GEMM_RELU) v = fmaxf(v, 0.0f);
                if (mask && !(mask[(long long)m * ldm + n] > 0.0f)) v = 0.0f;
                *c = v;
            }
    }
    if (!A_KCONT && rowsum && t < TB && m0 + t < M) rs_part[((long long)blockIdx.z * gridDim.x + blockIdx.x) * M + m0 + t] = rs;
}

// ---- the large GEMMs of the 16-bit training mode: bf16 operands in HBM (the tape's activations, the step's bf16 copies of
// the weight matrices), fp32 accumulate.  128 x 128 x 64 tiles on v_mfma_f32_32x32x16_bf16, LDS rows [row][64 k + 8 pad] bf16
// (144-byte pitch: the 16-byte fragment reads of a 16-lane group fall on 16 different bank quads), k-contiguous operands
// stored 8 k at a time, k-strided ones transposed as (k, k + 1) pairs.  One k-step (16 KiB per operand) is in flight in
// registers while the current one is multiplied: with ~3 workgroups per CU that is what keeps HBM busy -- at 32 k per step
// the same kernel moved 1.7 TB/s.
constexpr int BK = 64, BPITCH = BK + 8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8t;

template <bool KCONT>
__device__ __forceinline__ void btile_fetch_h(uint4 (&v)[4], const bf16_t* __restrict__ P, long long s_row, long long s_k,
                                              int row0, int rows, int kb, int k1, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = make_uint4(0u, 0u, 0u, 0u);
        if (KCONT) {            // 8 consecutive k of one row
            const int idx = t + 256 * i, r = idx >> 3, k = kb + (idx & 7) * 8;
            if (row0 + r < rows && k < k1) v[i] = *reinterpret_cast<const uint4*>(P + (long long)(row0 + r) * s_row + k);
        } else {                // 8 consecutive rows of k = 2 kp (i even) and 2 kp + 1 (i odd); neighbouring lanes take neighbouring
                                // k pairs, so that the transposing stores below spread over the LDS banks (2-way, not 16-way)
            const int kp = (t & 15) + 16 * (i >> 1), k = kb + 2 * kp + (i & 1), r = (t >> 4) * 8;
            if (row0 + r < rows && k < k1) v[i] = *reinterpret_cast<const uint4*>(P + (long long)k * s_k + row0 + r);
        }
    }
}
template <bool KCONT>
__device__ __forceinline__ void btile_store_h(unsigned short (*T)[BPITCH], const uint4 (&v)[4], int t) {
    if (KCONT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = t + 256 * i;
            *reinterpret_cast<uint4*>(&T[idx >> 3][(idx & 7) * 8]) = v[i];
        }
    } else {                    // row r + j gets the pair (k, k + 1) = (v[2 q].j, v[2 q + 1].j)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int kp = (t & 15) + 16 * q, r = (t >> 4) * 8;
            const unsigned a[4] = {v[2 * q].x, v[2 * q].y, v[2 * q].z, v[2 * q].w}, b[4] = {v[2 * q + 1].x, v[2 * q + 1].y, v[2 * q + 1].z, v[2 * q + 1].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                *reinterpret_cast<unsigned*>(&T[r + 2 * c][2 * kp]) = (a[c] & 0xffffu) | (b[c] << 16);
                *reinterpret_cast<unsigned*>(&T[r + 2 * c + 1][2 * kp]) = (a[c] >> 16) | (b[c] & 0xffff0000u);
            }
        }
    }
}

// dt & DT_C / DT_M: the result / the ReLU mask is bf16; GEMM_ACC adds cin[m ldcin + n] (fp32)
#ifndef PG_BGEMM_WGS
#define PG_BGEMM_WGS 3        // workgroups per CU the register budget is held to (168 VGPRs: three waves per SIMD)
#endif
// A second k segment of a GEMM (a layer on the concatenation of two inputs: the skip layer, the view layer): k >= K of the
// product comes from these operands (k-contiguous both, not split over workgroups); K a multiple of the k-step
struct KSeg2 { const bf16_t* A; const bf16_t* B; long long sam, sbn; int K; };
template <bool A_KCONT, bool B_KCONT>
__global__ __launch_bounds__(256, PG_BGEMM_WGS) void bgemm128_kernel(int M, int N, int K, const bf16_t* __restrict__ A, long long sam, long long sak,
                                                       const bf16_t* __restrict__ B, long long sbk, long long sbn,
                                                       void* __restrict__ C, long long ldc, const float* __restrict__ bias, int flags,
                                                       const void* __restrict__ mask, long long ldm, float* __restrict__ rowsum,
                                                       float* __restrict__ part, float* __restrict__ rs_part, int dt,
                                                       const float* cin, long long ldcin, const KSeg2 seg2) {
    __shared__ __attribute__((aligned(16))) unsigned short tiles[2][TB][BPITCH];
    unsigned short (*As)[BPITCH] = tiles[0], (*Bs)[BPITCH] = tiles[1];
    float rs = 0.0f;
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6, wm = (wv >> 1) * 64, wn = (wv & 1) * 64, li = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * TB, n0 = blockIdx.x * TB;
    const int nz = gridDim.z;
    const int kper = ((K + nz - 1) / nz + BK - 1) / BK * BK;
    const int k0 = blockIdx.z * kper, k1 = min(K, k0 + kper) + (A_KCONT && B_KCONT ? seg2.K : 0);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    uint4 ha[4], hb[4];
    // the operands of the segment the next k-step lies in (switched once, workgroup-uniform, when the steps reach K)
    const bf16_t *Ac = A, *Bc = B;
    long long samc = sam, sbnc = sbn;
    int kbase = 0, kend = (A_KCONT && B_KCONT && seg2.K > 0) ? min(k1, K) : k1;
    auto fetch = [&](int kb_) {
        if (A_KCONT && B_KCONT && seg2.K > 0 && kb_ >= K && kbase == 0) { Ac = seg2.A; Bc = seg2.B; samc = seg2.sam; sbnc = seg2.sbn; kbase = K; kend = k1; }
        btile_fetch_h<A_KCONT>(ha, Ac, samc, sak, m0, M, kb_ - kbase, kend - kbase, t);
        btile_fetch_h<B_KCONT>(hb, Bc, sbnc, sbk, n0, N, kb_ - kbase, kend - kbase, t);
    };
    fetch(k0);
    for (int kb = k0; kb < k1; kb += BK) {
        btile_store_h<A_KCONT>(As, ha, t);
        btile_store_h<B_KCONT>(Bs, hb, t);
        __syncthreads();
        fetch(kb + BK);         // (all zeros past the end)
        if (!A_KCONT && rowsum && (unsigned)((kb - k0) / BK) % gridDim.x == blockIdx.x && t < TB) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) rs += __builtin_bit_cast(float, (unsigned)As[t][kk] << 16);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const bf16x8t a0 = *reinterpret_cast<const bf16x8t*>(&As[wm + li][16 * ks + 8 * kh]);
            const bf16x8t a1 = *reinterpret_cast<const bf16x8t*>(&As[wm + 32 + li][16 * ks + 8 * kh]);
            const bf16x8t b0 = *reinterpret_cast<const bf16x8t*>(&Bs[wn + li][16 * ks + 8 * kh]);
            const bf16x8t b1 = *reinterpret_cast<const bf16x8t*>(&Bs[wn + 32 + li][16 * ks + 8 * kh]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // bf16 result (and mask) with 16-byte rows: the tile leaves through LDS -- the two 64-row halves one after the other in the
    // operand tiles' space -- so that every lane moves 16 contiguous bytes of a row (the register layout of the accumulators
    // would store 2 bytes per lane: 64 store instructions and, with a mask, 64 loads per thread)
    const bool vec = nz == 1 && (dt & DT_C) && ldc % 8 == 0 && reinterpret_cast<uintptr_t>(C) % 16 == 0 && n0 + TB <= N &&
                     (!mask || ((dt & DT_M) && ldm % 8 == 0 && reinterpret_cast<uintptr_t>(mask) % 16 == 0));
    if (vec) {
        constexpr int SP = TB + 8;                                      // staging pitch (bf16 elements): 272-byte rows
        static_assert(64 * SP * 2 <= (int)sizeof(tiles), "a half tile fits the operand tiles' space");
        bf16_t* stage = &tiles[0][0][0];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nl = wn + 32 * j + li;
                const float bv = bias ? bias[n0 + nl] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = (wv >> 1) * 32 + rho(r, kh), m = m0 + wm + 32 * i + rho(r, kh);
                    float v = acc[i][j][r];
                    if ((flags & GEMM_ACC) && m < M) v += cin[(long long)m * ldcin + n0 + nl];
                    v += bv;
                    if (flags & GEMM_RELU) v = fmaxf(v, 0.0f);
                    stage[ml * SP + nl] = f2bf(v);
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {                               // 64 rows x 16 chunks of 16 bytes
                const int idx = t + 256 * q, ml = idx >> 4, ch = idx & 15;
                const int m = m0 + (ml >> 5) * 64 + 32 * i + (ml & 31);
                if (m >= M) continue;
                uint4 v = *reinterpret_cast<const uint4*>(stage + ml * SP + ch * 8);
                if (mask) {
                    const uint4 k = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(mask) + (long long)m * ldm + n0 + ch * 8);
                    // keep where the stored post-activation is > 0: a positive bf16 has its sign bit clear and is not zero
                    auto keep = [](unsigned x, unsigned kk) {
                        const unsigned lo = ((kk & 0x8000u) == 0u && (kk & 0x7fffu) != 0u) ? 0xffffu : 0u;
                        const unsigned hi = ((kk & 0x80000000u) == 0u && (kk & 0x7fff0000u) != 0u) ? 0xffff0000u : 0u;
                        return x & (lo | hi);
                    };
                    v = make_uint4(keep(v.x, k.x), keep(v.y, k.y), keep(v.z, k.z), keep(v.w, k.w));
                }
                *reinterpret_cast<uint4*>(static_cast<bf16_t*>(C) + (long long)m * ldc + n0 + ch * 8) = v;
            }
            __syncthreads();
        }
    } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + 32 * j + li;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + rho(r, kh);
                if (m >= M) continue;
                float v = acc[i][j][r];
                if (nz > 1) { part[((long long)blockIdx.z * M + m) * N + n] = v; continue; }
                if (flags & GEMM_ACC) v += cin[(long long)m * ldcin + n];
                v += bv;
                if (flags & GEMM_RELU) v = fmaxf(v, 0.0f);
                if (mask && !(ld_el(mask, (long long)m * ldm + n, dt & DT_M) > 0.0f)) v = 0.0f;
                st_el(C, (long long)m * ldc + n, v, dt & DT_C);
            }
    }
    }
    if (!A_KCONT && rowsum && t < TB && m0 + t < M) rs_part[((long long)blockIdx.z * gridDim.x + blockIdx.x) * M + m0 + t] = rs;
}

// ---- a 256 x 256 layer as a PERSISTENT kernel (16-bit mode: the forward and dX GEMMs of the trunk's plain layers and of
// feature_linear): C[m][n] = epilogue(sum_k A[m][k] B[n][k]), N = K = 256, both operands bf16 and k-contiguous, bf16 result.
// A 256-wide layer moves 0.34 GB for 43 GFLOP: it is bound by HBM, so the kernel is built around the stream of activation
// rows, not around the MFMAs.
//   * One workgroup of 8 waves per CU strides over 128-row tiles.  The layer's weights live in REGISTERS for the whole launch:
//     wave w holds rows 32 w .. + 31 of B as 16 A-operand fragments of v_mfma_f32_32x32x16_bf16 (64 registers) and computes those
//     32 outputs for all 128 rows of every tile (four accumulator tiles; every wave reads the whole activation tile from LDS:
//     512 KiB per tile and CU = 4 k cycles, as many as the tile's MFMAs and less than half of what its 128 KiB take on HBM).
//   * The activation tile travels HBM -> LDS by LDS-DMA straight into MFMA-fragment order (fragment (row tile, k-step) = 1 KiB =
//     64 lanes x 16 B, lane (row, k half) fetching its own 16 bytes): no register staging, no transposing stores, conflict-free
//     ds_read_b128.  Two 64-KiB buffers; the tile after next is requested when a tile's MFMAs are done, so a tile has a whole
//     iteration (MFMAs + epilogue of the tile in front of it) to land.  One counted wait per tile: vmcnt(8) = "everything
//     but the 8 pieces just requested has arrived" (every wave requests 8 pieces per tile, the last tiles re-request the last one).
//   * The product is formed TRANSPOSED (W as the A operand, activations as B): a lane then holds four runs of 4 consecutive
//     outputs of ONE row per 32 x 32 tile (8-byte LDS stores).  The result tile is staged in the buffer its input has just left
//     (XOR-swizzled 16-byte chunks) and leaves in 512-byte rows, the ReLU mask read the same way (the first build stored 8 bytes
//     per lane straight from the registers: forward 82 - 108 us per layer, but dX -- whose mask loads then scatter too -- 140 - 160
//     against bgemm128_kernel's 130).
// Epilogue like bgemm128_kernel's: + cin (fp32, GEMM_ACC) + bias, ReLU, zero where the stored post-activation `mask` is <= 0.
// Tile geometry: ROWS rows per tile in NBUF buffers.  Measured (profiles/r5_train_lgemm_tiles.txt, 16-bit training step on one
// box): 128 x 2 and 96 x 3 9.06 ms, 64 x 4 8.88 -- the same 128 KiB of LDS, three tiles of requests in flight instead of one.
#ifndef PG_LG_ROWS
#define PG_LG_ROWS 64
#endif
#ifndef PG_LG_NBUF
#define PG_LG_NBUF 4
#endif
#ifndef PG_LG_WGS
#define PG_LG_WGS 1         // workgroups per CU (2: half the LDS and 128 registers each -- measured slower, the kernel spills)
#endif
// Geometry of an instantiation: KS1 + KS2 k-steps of 16 (two input segments: [h | x] of the skip layer; KS2 = 0: one), ROWS rows per
// tile, NBUF buffers.  The weights of both segments stay in registers (4 (KS1 + KS2) of them: 64 for a 256-wide layer, 108 for
// layer 0's K = 432, 172 for the skip layer's 256 + 432).
template <int KS1, int KS2, int ROWS, int NBUF>
struct LG {
    static constexpr int F = KS1 + KS2, MT = ROWS / 32, NFRAG = MT * F, BUF = NFRAG * 1024, LDS = NBUF * BUF + 2048;
    static constexpr int JMAX = (NFRAG + 7) / 8;             // LDS-DMA pieces per wave and tile: JMAX, or JMAX - 1 for the waves past NFRAG % 8
    static_assert(ROWS % 32 == 0 && LDS * PG_LG_WGS <= 160 * 1024 && (ROWS * 32) % 512 == 0 && BUF >= ROWS * 512 && KS1 > 0, "tile geometry");
};
using LG256 = LG<16, 0, PG_LG_ROWS, PG_LG_NBUF>;             // a 256 x 256 layer
using LG432 = LG<27, 0, 32, 4>;                              // layer 0: K = 432
using LGSKIP = LG<16, 27, 32, 3>;                            // the skip layer: [h (256) | x (432)]
__device__ __forceinline__ void lg_dma(const bf16_t* base, unsigned lane_off, unsigned lds_dst) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds_dst), "v"(lane_off), "s"(base) : "memory");
}
template <typename G>
struct LGK;
template <int KS1, int KS2, int ROWS, int NBUF>
struct LGK<LG<KS1, KS2, ROWS, NBUF>> {
    using G = LG<KS1, KS2, ROWS, NBUF>;
    static __device__ __forceinline__ void run(int M, const bf16_t* __restrict__ A, long long lda, const bf16_t* __restrict__ B, long long ldb,
                                               const bf16_t* __restrict__ A2, long long lda2, const bf16_t* __restrict__ B2, long long ldb2,
                                               bf16_t* __restrict__ C, long long ldc, const float* __restrict__ bias, int flags,
                                               const bf16_t* __restrict__ mask, long long ldm, const float* __restrict__ cin, long long ldcin,
                                               const float* __restrict__ r1_row, long long r1_ld, const float* __restrict__ r1_col) {
    // r1_row / r1_col: + r1_row[m r1_ld] r1_col[n], a rank-1 term in fp32 (the alpha head's share of dH7 = d sigma (x) w_alpha: no
    // [P, 256] fp32 array written by one kernel and read back by this one)
    constexpr int LG_ROWS = ROWS, LG_NBUF = NBUF, LG_MT = G::MT, LG_BUF = G::BUF;
    extern __shared__ __attribute__((aligned(16))) uint8_t lg_smem[];
    float* bias_l = reinterpret_cast<float*>(lg_smem + LG_NBUF * LG_BUF);
    float* col_l = bias_l + 256;
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lg_smem;
    if (t < 256) { bias_l[t] = bias ? bias[t] : 0.0f; col_l[t] = r1_col ? r1_col[t] : 0.0f; }
    bf16x8t wf[G::F];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) wf[ks] = *reinterpret_cast<const bf16x8t*>(B + (long long)(32 * wv + li) * ldb + 16 * ks + 8 * kh);
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) wf[KS1 + ks] = *reinterpret_cast<const bf16x8t*>(B2 + (long long)(32 * wv + li) * ldb2 + 16 * ks + 8 * kh);
    const int n_tiles = (M + LG_ROWS - 1) / LG_ROWS;
    // fragment f = (row tile f / F, k-step f % F of the concatenated input) of a tile; wave wv requests f = wv, wv + 8, ..
    const bool full = wv < (G::NFRAG % 8 == 0 ? 8 : G::NFRAG % 8);         // this wave has JMAX pieces per tile (else JMAX - 1)
    auto request = [&](int tile, int buf) {
        tile = min(tile, n_tiles - 1);
        const bf16_t* base1 = A + (long long)tile * LG_ROWS * lda;         // wave-uniform
        const bf16_t* base2 = KS2 ? A2 + (long long)tile * LG_ROWS * lda2 : nullptr;
        const int rows_left = M - tile * LG_ROWS;
#pragma unroll
        for (int j = 0; j < G::JMAX; ++j) {
            const int f = wv + 8 * j;
            if (j + 1 == G::JMAX && !full) break;
            const int mt = f / G::F, fk = f - mt * G::F;
            const int row = min(32 * mt + li, rows_left - 1);            // rows past the end re-read the last one (never stored)
            if (KS2 == 0 || fk < KS1) lg_dma(base1, (unsigned)((row * lda + 16 * fk + 8 * kh) * 2), lds0 + buf * LG_BUF + f * 1024);
            else lg_dma(base2, (unsigned)((row * lda2 + 16 * (fk - KS1) + 8 * kh) * 2), lds0 + buf * LG_BUF + f * 1024);
        }
    };
    int buf = 0;
#pragma unroll
    for (int b = 0; b < LG_NBUF; ++b) request(blockIdx.x + b * (int)gridDim.x, b);
    asm volatile("" :: "v"(wf[0]), "v"(wf[G::F - 1]) : "memory");       // (the weight loads are in front of the counted waits below)
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, buf = buf + 1 == LG_NBUF ? 0 : buf + 1) {
        // everything but the pieces of the LG_NBUF - 1 tiles requested behind this one has arrived
        if (full) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((LG_NBUF - 1) * G::JMAX) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((LG_NBUF - 1) * (G::JMAX - 1)) : "memory");
        __builtin_amdgcn_s_barrier();
        f32x16 acc[LG_MT];
#pragma unroll
        for (int i = 0; i < LG_MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        const uint8_t* fb = lg_smem + buf * LG_BUF + lane * 16;
#pragma unroll
        for (int ks = 0; ks < G::F; ++ks)
#pragma unroll
            for (int mt = 0; mt < LG_MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], *reinterpret_cast<const bf16x8t*>(fb + (G::F * mt + ks) * 1024), acc[mt], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // every wave is done reading the buffer: it becomes the result tile's staging area
        // registers -> LDS: row m of the tile at m * 512 B, its 16-byte chunk c at slot c ^ (m & 31) (a lane column is 32 rows at one
        // n: without the swizzle they would share a bank)
        uint8_t* stage = lg_smem + buf * LG_BUF;
#pragma unroll
        for (int mt = 0; mt < LG_MT; ++mt) {
            const int ml = 32 * mt + li;
            const long long m = (long long)tile * LG_ROWS + ml;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int n0 = 32 * wv + 8 * qd + 4 * kh;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[mt][4 * qd + r];
                if ((flags & GEMM_ACC) && m < M) {
                    const float4 c4 = *reinterpret_cast<const float4*>(cin + m * ldcin + n0);
                    v[0] += c4.x; v[1] += c4.y; v[2] += c4.z; v[3] += c4.w;
                }
                if (r1_row) {
                    const float gr = r1_row[min(m, (long long)M - 1) * r1_ld];
                    const float4 c4 = *reinterpret_cast<const float4*>(col_l + n0);
                    v[0] = fmaf(gr, c4.x, v[0]); v[1] = fmaf(gr, c4.y, v[1]); v[2] = fmaf(gr, c4.z, v[2]); v[3] = fmaf(gr, c4.w, v[3]);
                }
                const float4 b4 = *reinterpret_cast<const float4*>(bias_l + n0);
                v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                if (flags & GEMM_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                }
                const unsigned lo = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), hi = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                *reinterpret_cast<uint2*>(stage + ml * 512 + (((n0 >> 3) ^ (ml & 31)) << 4) + (n0 & 4) * 2) = make_uint2(lo, hi);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // LDS -> HBM: a thread moves chunk t & 31 of rows (t >> 5) + 16 j -- 512 contiguous bytes per row and half-wave; the ReLU
        // mask (the stored post-activation the consumer of the result was ReLU'd to) is read in the same pattern
        {
            const int ch = t & 31, r0 = t >> 5;
            auto keep = [](unsigned x, unsigned k) {    // keep where the stored post-activation is > 0: a positive bf16 has its sign bit clear and is not zero
                const unsigned l = ((k & 0x8000u) == 0u && (k & 0x7fffu) != 0u) ? 0xffffu : 0u;
                const unsigned h = ((k & 0x80000000u) == 0u && (k & 0x7fff0000u) != 0u) ? 0xffff0000u : 0u;
                return x & (l | h);
            };
            constexpr int RPT = LG_ROWS / 16, GRP = RPT % 4 == 0 ? 4 : RPT % 3 == 0 ? 3 : RPT % 2 == 0 ? 2 : 1;      // rows per thread, in groups whose masks are in flight together
            static_assert(RPT % GRP == 0, "rows per thread");
#pragma unroll
            for (int jh = 0; jh < RPT / GRP; ++jh) {
                uint4 kk[GRP];
                if (mask) {
#pragma unroll
                    for (int j = 0; j < GRP; ++j) {
                        const long long m = (long long)tile * LG_ROWS + r0 + 16 * (GRP * jh + j);
                        kk[j] = m < M ? *reinterpret_cast<const uint4*>(mask + m * ldm + ch * 8) : make_uint4(0u, 0u, 0u, 0u);
                    }
                }
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const int ml = r0 + 16 * (GRP * jh + j);
                    const long long m = (long long)tile * LG_ROWS + ml;
                    uint4 v = *reinterpret_cast<const uint4*>(stage + ml * 512 + ((ch ^ (ml & 31)) << 4));
                    if (mask) v = make_uint4(keep(v.x, kk[j].x), keep(v.y, kk[j].y), keep(v.z, kk[j].z), keep(v.w, kk[j].w));
                    if (m < M) *reinterpret_cast<uint4*>(C + m * ldc + ch * 8) = v;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // the staging area has been read: the next request may overwrite it
        asm volatile("" ::: "memory");
        request(tile + LG_NBUF * (int)gridDim.x, buf);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no DMA in flight when the wave exits
    }
};
#define PG_LG_ARGS int M, const bf16_t* __restrict__ A, long long lda, const bf16_t* __restrict__ B, long long ldb, const bf16_t* __restrict__ A2, long long lda2,      \
    const bf16_t* __restrict__ B2, long long ldb2, bf16_t* __restrict__ C, long long ldc, const float* __restrict__ bias, int flags, const bf16_t* __restrict__ mask,   \
    long long ldm, const float* __restrict__ cin, long long ldcin, const float* __restrict__ r1_row, long long r1_ld, const float* __restrict__ r1_col
#define PG_LG_PASS M, A, lda, B, ldb, A2, lda2, B2, ldb2, C, ldc, bias, flags, mask, ldm, cin, ldcin, r1_row, r1_ld, r1_col
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2 * PG_LG_WGS, 2 * PG_LG_WGS))) void lgemm256_kernel(PG_LG_ARGS) { LGK<LG256>::run(PG_LG_PASS); }
__global__ __launch_bounds__(512) void lgemm432_kernel(PG_LG_ARGS) { LGK<LG432>::run(PG_LG_PASS); }
__global__ __launch_bounds__(512) void lgemm_skip_kernel(PG_LG_ARGS) { LGK<LGSKIP>::run(PG_LG_PASS); }

// ---- the heads' skinny products (alpha: 1 output on 256 inputs, rgb: 3 on 128): one pass over the activations at memory
// speed instead of a 64-wide GEMM tile with 1-3 useful columns.  XT = the activations' element type; a row of K elements is
// read by LPR = K / VEC lanes, 16 bytes each (VEC = 8 bf16 or 4 floats).
template <typename XT> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { const float4 a = *reinterpret_cast<const float4*>(p); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __builtin_bit_cast(float, w[i] << 16); v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u); }
    }
};
// Y[p, o] = b[o] + sum_k X[p, k] Wt[o, k], o < NO (Y fp32 with leading dimension ldy)
template <typename XT, int NO>
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const XT* __restrict__ X, long long ldx, long long P, int K, const float* __restrict__ Wt, long long ldw,
                                                         const float* __restrict__ b, float* __restrict__ Y, long long ldy) {
    constexpr int VEC = Vec16<XT>::N;
    const int lpr = K / VEC, t = threadIdx.x, kl = (t % lpr) * VEC, rg = t / lpr, nrg = 256 / lpr;
    float w[NO][VEC];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int e = 0; e < VEC; ++e) w[o][e] = Wt[o * ldw + kl + e];
    for (long long p = (long long)blockIdx.x * nrg + rg; p < P; p += (long long)gridDim.x * nrg) {
        float x[VEC], acc[NO];
        Vec16<XT>::load(X + p * ldx + kl, x);
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            acc[o] = 0.0f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[o] = fmaf(x[e], w[o][e], acc[o]);
        }
        for (int d = lpr >> 1; d > 0; d >>= 1)
#pragma unroll
            for (int o = 0; o < NO; ++o) acc[o] += __shfl_xor(acc[o], d);
        if (t % lpr == 0)
#pragma unroll
            for (int o = 0; o < NO; ++o) Y[p * ldy + o] = acc[o] + (b ? b[o] : 0.0f);
    }
}
// part[block][o][k] = sum over the block's points of dY[p, o] X[p, k] (dY fp32, leading dimension ldy); reduce_parts_kernel
// adds the blocks in order
template <typename XT, int NO>
__global__ __launch_bounds__(256) void skinny_dw_kernel(const XT* __restrict__ X, long long ldx, long long P, int K, const float* __restrict__ dY, long long ldy,
                                                        float* __restrict__ part) {
    constexpr int VEC = Vec16<XT>::N;
    __shared__ float red[256][NO * VEC + 1];
    const int lpr = K / VEC, t = threadIdx.x, kl = (t % lpr) * VEC, rg = t / lpr, nrg = 256 / lpr;
    const long long per = (P + gridDim.x - 1) / gridDim.x, p0 = blockIdx.x * per, p1 = min(P, p0 + per);
    float acc[NO][VEC];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[o][e] = 0.0f;
    for (long long p = p0 + rg; p < p1; p += nrg) {
        float x[VEC];
        Vec16<XT>::load(X + p * ldx + kl, x);
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const float g = dY[p * ldy + o];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[o][e] = fmaf(g, x[e], acc[o][e]);
        }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int e = 0; e < VEC; ++e) red[t][o * VEC + e] = acc[o][e];
    __syncthreads();
    for (int i = t; i < NO * K; i += 256) {             // element (o, k): the row groups' shares, in order
        const int o = i / K, k = i - o * K, lane = k / VEC, e = k - lane * VEC;
        float sum = 0.0f;
        for (int g = 0; g < nrg; ++g) sum += red[g * lpr + lane][o * VEC + e];
        part[(long long)blockIdx.x * NO * K + i] = sum;
    }
}

// dX[p][k] = sum_{o < NO} dY[p][o] W[o][k] (the heads' input gradients: 1 or 3 products per element), eight columns per thread,
// zeroed where the stored post-activation `relu_of` (element type of dX) is <= 0: one pass that writes dX at memory speed
// instead of a 64-wide GEMM tile with K = 1..3 and a masking kernel behind it (33 - 103 + 37 us per call)
template <typename CT, int NO>
__global__ __launch_bounds__(256) void skinny_bwd_x_kernel(const float* __restrict__ dY, long long ldy, long long P, int K, const float* __restrict__ Wt, long long ldw,
                                                           CT* __restrict__ dX, long long ldx, const CT* __restrict__ relu_of) {
    const int tpr = K / 8;                      // threads per row
    const long long i = blockIdx.x * 256ll + threadIdx.x, p = i / tpr;
    const int k0 = (int)(i - p * tpr) * 8;
    if (p >= P) return;
    float g[NO], v[8];
#pragma unroll
    for (int o = 0; o < NO; ++o) g[o] = dY[p * ldy + o];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float s = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o) s = fmaf(g[o], Wt[o * ldw + k0 + e], s);
        v[e] = s;
    }
    if (relu_of) {
        CT hk[8];           // (one or two 16-byte loads)
        if constexpr (sizeof(CT) == 2) *reinterpret_cast<uint4*>(hk) = *reinterpret_cast<const uint4*>(relu_of + p * ldx + k0);
        else { *reinterpret_cast<float4*>(hk) = *reinterpret_cast<const float4*>(relu_of + p * ldx + k0); *reinterpret_cast<float4*>(hk + 4) = *reinterpret_cast<const float4*>(relu_of + p * ldx + k0 + 4); }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float hv;
            if constexpr (sizeof(CT) == 2) hv = bf2f(hk[e]); else hv = hk[e];
            if (!(hv > 0.0f)) v[e] = 0.0f;
        }
    }
    if constexpr (sizeof(CT) == 2) {
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (unsigned)f2bf(v[2 * e]) | ((unsigned)f2bf(v[2 * e + 1]) << 16);
        *reinterpret_cast<uint4*>(dX + p * ldx + k0) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        *reinterpret_cast<float4*>(dX + p * ldx + k0) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dX + p * ldx + k0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// out[i (row-major M x N with leading dimension ldo)] = sum over the nz slices of part[z][M][N], in slice order
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ part, int nz, int M, int N,
                                                          float* __restrict__ out, long long ldo) {
    const long long MN = (long long)M * N;
    if (N % 4 == 0 && ldo % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 && nz >= 8) {
        // four elements per thread; the slices in four consecutive groups, one per wave of the block, each summed in slice
        // order, the four group sums added in a fixed order (bitwise repeatable like the plain loop, four times the loads in
        // flight, four times the blocks)
        __shared__ float4 red[4][64];
        const int e = threadIdx.x & 63, zg = threadIdx.x >> 6;
        const long long i = (blockIdx.x * 64ll + e) * 4;
        const int zper = (nz + 3) / 4, z0 = zg * zper, z1 = min(nz, z0 + zper);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < MN) {
            int z = z0;
            for (; z + 4 <= z1; z += 4) {
                const float4 a = *reinterpret_cast<const float4*>(part + z * MN + i), b = *reinterpret_cast<const float4*>(part + (z + 1) * MN + i);
                const float4 c = *reinterpret_cast<const float4*>(part + (z + 2) * MN + i), d = *reinterpret_cast<const float4*>(part + (z + 3) * MN + i);
                s.x = ((s.x + a.x) + b.x) + c.x + d.x; s.y = ((s.y + a.y) + b.y) + c.y + d.y;
                s.z = ((s.z + a.z) + b.z) + c.z + d.z; s.w = ((s.w + a.w) + b.w) + c.w + d.w;
            }
            for (; z < z1; ++z) {
                const float4 a = *reinterpret_cast<const float4*>(part + z * MN + i);
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
        }
        red[zg][e] = s;
        __syncthreads();
        if (zg == 0 && i < MN) {
            const float4 a = red[0][e], b = red[1][e], c = red[2][e], d = red[3][e];
            const float4 r = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
            *reinterpret_cast<float4*>(out + (i / N) * ldo + (i % N)) = r;
        }
        return;
    }
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= MN) return;
    float s = 0.0f;
    for (int z = 0; z < nz; ++z) s += part[z * MN + i];
    out[(i / N) * ldo + (i % N)] = s;
}

// dH <- dH where H > 0 else 0 (ReLU backward on the stored post-activation)
__global__ __launch_bounds__(256) void relu_mask_kernel(void* __restrict__ d, const void* __restrict__ h, long long count, int bf) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < count; i += (long long)gridDim.x * 256)
        if (!(ld_el(h, i, bf) > 0.0f)) st_el(d, i, 0.0f, bf);
}

// this step's bf16 copies of all weight matrices in ONE launch (38 launches of 3 us each were 0.2 ms of the step's timeline):
// job = a plain copy [rows x cols, contiguous] or a transposed copy of a column block (source leading dimension ld)
struct WJob { const float* src; bf16_t* dst; long long ld; int rows, cols, transpose; };
constexpr int WJOBS_MAX = 64;
struct WJobs { WJob j[WJOBS_MAX]; };
__global__ __launch_bounds__(256) void cvt_weights_kernel(WJobs js) {
    const WJob& j = js.j[blockIdx.y];
    if (!j.transpose) {
        const long long n = (long long)j.rows * j.cols;
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) j.dst[i] = f2bf(j.src[i]);
        return;
    }
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, tcols = (j.cols + 31) / 32, tiles = ((j.rows + 31) / 32) * tcols;
    for (int tl = blockIdx.x; tl < tiles; tl += gridDim.x) {
        const int r0 = (tl / tcols) * 32, c0 = (tl % tcols) * 32;
        for (int i = ty; i < 32; i += 8) tile[i][tx] = (r0 + i < j.rows && c0 + tx < j.cols) ? j.src[(long long)(r0 + i) * j.ld + c0 + tx] : 0.0f;
        __syncthreads();
        for (int i = ty; i < 32; i += 8)
            if (c0 + i < j.cols && r0 + tx < j.rows) j.dst[(long long)(c0 + i) * j.rows + r0 + tx] = f2bf(tile[tx][i]);
        __syncthreads();
    }
}

// part[block][n] = sum over the block's CS_ROWS rows of d[row * ld + n], N <= 4 (the heads' bias gradients: columns of d_raw;
// reduce_parts_kernel adds the blocks in order).  Thread t adds rows r0 + t, r0 + t + 256, .. in that order, the 256 shares are
// added by a fixed tree: bitwise repeatable.  (Until round 5 a block covered 256 rows with ONE thread per column -- 1 to 3 active
// threads per block -- and left 1280 slices to a single reduction block: 70 - 105 us per call, 0.5 ms per step.)
constexpr int CS_ROWS = 4096;
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ d, long long rows, int N, long long ld, float* __restrict__ part, int bf) {
    __shared__ float red[4][256];
    const long long r0 = (long long)blockIdx.x * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (long long r = r0 + threadIdx.x; r < r1; r += 256)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            if (n < N) s[n] += ld_el(d, r * ld + n, bf);
#pragma unroll
    for (int n = 0; n < 4; ++n) red[n][threadIdx.x] = s[n];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
#pragma unroll
            for (int n = 0; n < 4; ++n) red[n][threadIdx.x] += red[n][threadIdx.x + w];
        __syncthreads();
    }
    if ((int)threadIdx.x < N) part[(long long)blockIdx.x * N + threadIdx.x] = red[threadIdx.x][0];
}

// the same for any number of columns, a thread per column over the block's 256 rows (the row sums behind a small-tile GEMM)
__global__ __launch_bounds__(256) void colsum_wide_kernel(const void* __restrict__ d, long long rows, int N, long long ld, float* __restrict__ part, int bf) {
    const long long r0 = blockIdx.x * 256ll, r1 = min(rows, r0 + 256);
    for (int n = threadIdx.x; n < N; n += 256) {
        float s = 0.0f;
        for (long long r = r0; r < r1; ++r) s += ld_el(d, r * ld + n, bf);
        part[(long long)blockIdx.x * N + n] = s;
    }
}

// gradient of the frame codes (embedding.py:19-36), in two ordered steps: per ray the sum of dxc over its S points,
// then per (code, channel) the sum over the rays that index it, in ray order; the mean code of a negative index
// spreads its gradient evenly over all rows, like codes.mean(0)
__global__ __launch_bounds__(256) void code_ray_sum_kernel(const float* __restrict__ dxc, long long n_rays, int S, float* __restrict__ ray_g) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= n_rays * FC_CH) return;
    const long long ray = i / FC_CH;
    const int k = (int)(i - ray * FC_CH);
    float s = 0.0f;
    for (int p = 0; p < S; ++p) s += dxc[(ray * S + p) * FC_CH + k];
    ray_g[i] = s;
}
__global__ __launch_bounds__(64) void code_gather_kernel(const float* __restrict__ ray_g, long long n_rays, const float* __restrict__ cams,
                                                        int n_codes, float* __restrict__ dcodes) {
    const int c = blockIdx.x, k = threadIdx.x;
    if (k >= FC_CH) return;
    float own = 0.0f, mean = 0.0f;
    for (long long r = 0; r < n_rays; ++r) {
        const float cam = cams ? cams[r] : -1.0f;
        const float g = ray_g[r * FC_CH + k];
        if (cam < 0.0f) mean += g / (float)n_codes;
        else if (min((int)cam, n_codes - 1) == c) own += g;
    }
    dcodes[c * FC_CH + k] = own + mean;
}

// ---- backward of raw2outputs (core/networks/nerf.py:150-205), one thread per ray ---------------------------------
// forward: delta_i = (z_{i+1} - z_i) |d| (last 1e10 |d|); c_i = sigmoid(raw_rgb) (1 + 2 eps) - eps;
// s_i = act(raw_sigma / B + noise) (relu or shifted softplus); a_i = 1 - exp(-s_i delta_i); T_i = prod_{k<i} (1 - a_k + 1e-10); w_i = a_i T_i;
// rgb_map = sum w_i c_i; acc_map = min(sum w_i, 1).  Given d_rgb [n,3] and d_acc [n] (or null):
// g_i = dL/dw_i = d_rgb . c_i + d_acc [sum w < 1];  dL/da_i = g_i T_i - (sum_{k>i} g_k w_k) / (1 - a_i + 1e-10).
__global__ __launch_bounds__(64) void composite_bwd_kernel(const float* __restrict__ rays, const float* __restrict__ z,
                                                          const float* __restrict__ raw, const float* __restrict__ noise,
                                                          long long n, int S, float density_scale, float rgb_eps, int act, float act_shift,
                                                          const float* __restrict__ d_rgb, const float* __restrict__ d_acc,
                                                          float* __restrict__ d_raw) {
    // act_fn of raw2outputs and its derivative (get_density_fn, core/raycasters.py:230-238): relu, or
    // softplus(x - shift) with torch's linear branch above 20
    auto actf = [&](float x) { if (act == 0) return fmaxf(x, 0.0f); const float t = x - act_shift; return t > 20.0f ? t : log1pf(expf(t)); };
    auto dact = [&](float x) { if (act == 0) return x > 0.0f ? 1.0f : 0.0f; const float t = x - act_shift; return t > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-t)); };
    const long long r = blockIdx.x * 64ll + threadIdx.x;
    if (r >= n) return;
    const float* rb = rays + r * 11;
    const float dn = sqrtf(rb[3] * rb[3] + rb[4] * rb[4] + rb[5] * rb[5]);
    const float* zr = z + r * S;
    const float* rw = raw + r * S * 4;
    const float* nz = noise ? noise + r * S : nullptr;
    float* dr = d_raw + r * S * 4;
    const float gr = d_rgb ? d_rgb[r * 3] : 0.0f, gg = d_rgb ? d_rgb[r * 3 + 1] : 0.0f, gb = d_rgb ? d_rgb[r * 3 + 2] : 0.0f;
    // pass 1: sum of the weights (for the min(.., 1) of acc_map)
    float T = 1.0f, wsum = 0.0f;
    for (int i = 0; i < S; ++i) {
        const float delta = (i + 1 < S ? zr[i + 1] - zr[i] : 1e10f) * dn;
        const float s = actf(rw[i * 4 + 3] / density_scale + (nz ? nz[i] : 0.0f));
        const float a = 1.0f - expf(-s * delta);
        wsum += a * T;
        T *= 1.0f - a + 1e-10f;
    }
    const float ga = (d_acc && wsum < 1.0f) ? d_acc[r] : 0.0f;
    // pass 2, back to front: T_i by division is unstable, so the transmittances are recomputed front to back in
    // chunks ... S <= 256: keep them in a small local array instead
    float Tl[256];
    T = 1.0f;
    for (int i = 0; i < S; ++i) {
        Tl[i] = T;
        const float delta = (i + 1 < S ? zr[i + 1] - zr[i] : 1e10f) * dn;
        const float s = actf(rw[i * 4 + 3] / density_scale + (nz ? nz[i] : 0.0f));
        T *= 1.0f - (1.0f - expf(-s * delta)) + 1e-10f;
    }
    float suffix = 0.0f;        // sum_{k>i} g_k w_k
    for (int i = S - 1; i >= 0; --i) {
        const float delta = (i + 1 < S ? zr[i + 1] - zr[i] : 1e10f) * dn;
        const float pre = rw[i * 4 + 3] / density_scale + (nz ? nz[i] : 0.0f);
        const float s = actf(pre);
        const float e = expf(-s * delta);
        const float a = 1.0f - e;
        const float w = a * Tl[i];
        const float sr = 1.0f / (1.0f + expf(-rw[i * 4])), sg = 1.0f / (1.0f + expf(-rw[i * 4 + 1])), sb = 1.0f / (1.0f + expf(-rw[i * 4 + 2]));
        const float k = 1.0f + 2.0f * rgb_eps;
        const float g = gr * (sr * k - rgb_eps) + gg * (sg * k - rgb_eps) + gb * (sb * k - rgb_eps) + ga;
        const float dA = g * Tl[i] - suffix / (1.0f - a + 1e-10f);
        dr[i * 4] = gr * w * k * sr * (1.0f - sr);
        dr[i * 4 + 1] = gg * w * k * sg * (1.0f - sg);
        dr[i * 4 + 2] = gb * w * k * sb * (1.0f - sb);
        const float da = dact(pre);
        dr[i * 4 + 3] = da > 0.0f ? dA * delta * e * da / density_scale : 0.0f;
        suffix += g * w;
    }
}

// ---- host side -------------------------------------------------------------------------------------------------
struct Pass {               // one network evaluation kept for the backward pass
    long long P = 0;        // points
    int S = 0;
    void *X = nullptr, *H[DEPTH] = {}, *F = nullptr, *G = nullptr;       // tape element type: fp32, or bf16 in the 16-bit mode
    float *raw = nullptr, *z = nullptr, *noise = nullptr, *pn = nullptr;
};

struct Tape {
    uint8_t* buf = nullptr;
    size_t bytes = 0;
    bool valid = false;
    int64_t generation = 0;     // id of the forward pass the tape holds (pg_train_forward returns it, pg_train_backward checks it)
    long long n = 0;
    int S = 0, N = 0, fc = 0;
    float *rays = nullptr, *cams = nullptr;
    void *tmpA = nullptr, *tmpB = nullptr, *dG = nullptr;       // activation gradients (tape element type)
    bf16_t* wb[2][24] = {};                                     // 16-bit mode: bf16 copies of the weight matrices (even tensor indices)
    bf16_t* wbT[2][24] = {};                                    // ... and transposed copies [in][out] of the blocks the dX GEMMs multiply by
    float *tmpF = nullptr;                                      // 16-bit mode: fp32 partial sums of the two-part layers (skip layer, view layer, dH7)
    float *dC = nullptr, *d_raw = nullptr;
    float *part = nullptr, *rs_part = nullptr, *ray_g = nullptr;      // split-K slices, row / column sum shares, per-ray code gradients
    Pass pass[2];
    pg_net_params params[2];
    bool has_fine = false;
    bool bf16 = false;          // 16-bit training mode (pg_set_train_precision PG_PREC_BF16): bf16 tape, bf16 operands in the large GEMMs
    int es() const { return bf16 ? 2 : 4; }                     // bytes per tape element
};
inline const void* el_off(const void* p, long long elems, int es) { return static_cast<const uint8_t*>(p) + elems * es; }

constexpr size_t PART_FLOATS = 20u << 20;        // split-K scratch: slices x M x N of the largest weight gradient (80 MB)
constexpr size_t RS_FLOATS = 1u << 20;

inline Tape* tape_of(pg_handle* h) {
    if (!h->train) h->train = new Tape();
    return static_cast<Tape*>(h->train);
}

#define PG_LAUNCH_CHECK(h, what)                                                                            \
    do {                                                                                                    \
        hipError_t e_ = hipGetLastError();                                                                  \
        if (e_ != hipSuccess) return pg_fail(h, PG_EHIP, "%s launch failed: %s", what, hipGetErrorString(e_)); \
    } while (0)

// C[M,N] = A B (+ bias, relu, accumulate); ksplit > 1: K in slices, summed into C in slice order (C is overwritten)
// dt (DT_*): which of A, B, C and the mask are bf16 arrays (16-bit mode: the tape's activations); GEMM_ACC adds cin (fp32,
// leading dimension ldcin; null: C itself, fp32)
// C += row (x) col in fp32, row[m ld] a column of another array (gemm(): persistent layer kernel only)
struct Rank1 { const float* row; long long ld; const float* col; };
bool lgemm_enabled() {
    static const bool on = [] { const char* e = std::getenv("POSEGEN_LGEMM"); return !(e && e[0] == '0'); }();
    return on;
}
int gemm(pg_handle* h, hipStream_t s, bool a_kcont, bool b_kcont, int M, int N, int K, const void* A, long long sam, long long sak,
         const void* B, long long sbk, long long sbn, void* C, long long ldc, const float* bias, int flags, int ksplit = 1,
         const void* mask = nullptr, long long ldm = 0, float* rowsum = nullptr, int dt = 0, const float* cin = nullptr, long long ldcin = 0,
         const KSeg2* seg2 = nullptr, const Rank1* r1 = nullptr) {
    if (M <= 0 || N <= 0 || K <= 0) return PG_OK;
    Tape& t = *tape_of(h);
    if ((flags & GEMM_ACC) && !cin) {
        if (dt & DT_C) return pg_fail(h, PG_EINVAL, "accumulating GEMM into a bf16 result needs an fp32 input array");
        cin = static_cast<const float*>(C); ldcin = ldc;
    }
    if (ksplit > 1) {
        const long long fit = (long long)(PART_FLOATS / ((size_t)M * N));
        if (fit < 2) return pg_fail(h, PG_EINVAL, "split-K scratch too small for a %d x %d result", M, N);
        if (ksplit > fit) ksplit = (int)fit;
        if (!t.part) return pg_fail(h, PG_ESTATE, "split-K GEMM without a tape");
        if (dt & DT_C) return pg_fail(h, PG_EINVAL, "split-K GEMM results are fp32");
    }
    auto reduce = [&](const float* part, int nz, int rows, int cols, float* out, long long ldo) {
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)(((long long)rows * cols + 255) / 256)), dim3(256), 0, s, part, nz, rows, cols, out, ldo);
    };
    // 16-byte loads: the contiguous index in runs of 4 floats / 8 bf16, every other stride and the base aligned likewise
    auto aligned = [](const void* p, long long stride, bool bf) { return reinterpret_cast<uintptr_t>(p) % 16 == 0 && stride % (bf ? 8 : 4) == 0; };
    const bool abf = dt & DT_A, bbf = dt & DT_B;
    const int qa = abf ? 8 : 4, qb = bbf ? 8 : 4;
    bool big = M >= 64 && N >= 64 && aligned(A, a_kcont ? sam : sak, abf) && aligned(B, b_kcont ? sbn : sbk, bbf) &&
               (a_kcont ? K % qa == 0 : M % qa == 0) && (b_kcont ? K % qb == 0 : N % qb == 0);
    if ((abf || bbf) && !t.bf16) return pg_fail(h, PG_EINVAL, "bf16 GEMM operands outside the 16-bit mode");
    // a 256-wide layer of the 16-bit mode on the persistent kernel: K = 256 (forward / dX of the trunk's plain layers, feature_linear),
    // K = 432 (layer 0) or the skip layer's two segments [h (256) | x (432)]
    const bool seg_skip = seg2 && K == 256 && seg2->K == 432 && seg2->sam % 8 == 0 && seg2->sbn % 8 == 0 &&
                          reinterpret_cast<uintptr_t>(seg2->A) % 16 == 0 && reinterpret_cast<uintptr_t>(seg2->B) % 16 == 0;
    if (lgemm_enabled() && big && t.bf16 && abf && bbf && (dt & DT_C) && a_kcont && b_kcont && N == 256 && ksplit == 1 && !rowsum &&
        ((!seg2 && (K == 256 || K == 432)) || seg_skip) &&
        sak == 1 && sbk == 1 && ldc % 8 == 0 && reinterpret_cast<uintptr_t>(C) % 16 == 0 &&
        (!mask || ((dt & DT_M) && ldm % 8 == 0 && reinterpret_cast<uintptr_t>(mask) % 16 == 0)) &&
        (!(flags & GEMM_ACC) || (ldcin % 4 == 0 && reinterpret_cast<uintptr_t>(cin) % 16 == 0))) {
        auto launch = [&](auto kern, int lds, int rows, int wgs, std::atomic<unsigned long long>& attr_done) -> int {
            if (!(attr_done.load(std::memory_order_acquire) & (1ull << (h->device & 63)))) {      // the opt-in to > 64 KiB of dynamic LDS is per (kernel, device)
                PG_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                attr_done.fetch_or(1ull << (h->device & 63), std::memory_order_release);
            }
            const int tiles = (M + rows - 1) / rows;
            hipLaunchKernelGGL(kern, dim3(std::min(tiles, wgs * h->n_cu)), dim3(512), lds, s, M, static_cast<const bf16_t*>(A), sam,
                               static_cast<const bf16_t*>(B), sbn, seg2 ? seg2->A : nullptr, seg2 ? seg2->sam : 0, seg2 ? seg2->B : nullptr,
                               seg2 ? seg2->sbn : 0, static_cast<bf16_t*>(C), ldc, bias, flags, static_cast<const bf16_t*>(mask), ldm, cin, ldcin,
                               r1 ? r1->row : nullptr, r1 ? r1->ld : 0, r1 ? r1->col : nullptr);
            PG_LAUNCH_CHECK(h, "persistent layer GEMM");
            return PG_OK;
        };
        static std::atomic<unsigned long long> done256{0}, done432{0}, done_skip{0};
        if (seg2) return launch(lgemm_skip_kernel, LGSKIP::LDS, 32, 1, done_skip);
        if (K == 432) return launch(lgemm432_kernel, LG432::LDS, 32, 1, done432);
        return launch(lgemm256_kernel, LG256::LDS, PG_LG_ROWS, PG_LG_WGS, done256);
    }
    if (r1) return pg_fail(h, PG_EINVAL, "a rank-1 term outside the persistent layer kernel");
    if (big && !(a_kcont == false && b_kcont == true)) {
        const dim3 g((N + TB - 1) / TB, (M + TB - 1) / TB, ksplit);
        if (rowsum && (size_t)ksplit * g.x * M > RS_FLOATS) return pg_fail(h, PG_EINVAL, "row-sum scratch too small");
        bool launched = true;
#define PG_BGEMM(AK, BK_) hipLaunchKernelGGL((bgemm128_kernel<AK, BK_>), g, dim3(256), 0, s, M, N, K, static_cast<const bf16_t*>(A), sam, sak, static_cast<const bf16_t*>(B), sbk, sbn, C, ldc, bias, flags, mask, ldm, rowsum, t.part, t.rs_part, dt, cin, ldcin, sg)
        const KSeg2 sg = seg2 ? *seg2 : KSeg2{nullptr, nullptr, 0, 0, 0};
        if (seg2 && !(t.bf16 && abf && bbf && a_kcont && b_kcont && ksplit == 1 && K % BK == 0 && seg2->K % 8 == 0 && seg2->sam % 8 == 0 &&
                      seg2->sbn % 8 == 0 && reinterpret_cast<uintptr_t>(seg2->A) % 16 == 0 && reinterpret_cast<uintptr_t>(seg2->B) % 16 == 0))
            return pg_fail(h, PG_EINVAL, "two-segment GEMM: bf16 k-contiguous operands, first segment a multiple of the k-step");
        if (t.bf16) {       // the operand layouts the 16-bit training step uses: forward, dX, dW
            if (!(abf && bbf)) launched = false;
            else if (a_kcont && b_kcont) PG_BGEMM(true, true);
            else if (a_kcont) PG_BGEMM(true, false);
            else PG_BGEMM(false, false);
        } else {
            if (dt) return pg_fail(h, PG_EINVAL, "bf16 GEMM operands outside the 16-bit mode");
            float* Cf = static_cast<float*>(C);
            const float* Af = static_cast<const float*>(A);
            const float* Bf = static_cast<const float*>(B);
            const float* mf = static_cast<const float*>(mask);
#define PG_SGEMM(AK, BK_) hipLaunchKernelGGL((sgemm128_kernel<AK, BK_>), g, dim3(256), 0, s, M, N, K, Af, sam, sak, Bf, sbk, sbn, Cf, ldc, bias, flags, mf, ldm, rowsum, t.part, t.rs_part, cin, ldcin)
            if (a_kcont && b_kcont) PG_SGEMM(true, true);
            else if (a_kcont) PG_SGEMM(true, false);
            else PG_SGEMM(false, false);
#undef PG_SGEMM
        }
#undef PG_BGEMM
        if (launched) {
            PG_LAUNCH_CHECK(h, "gemm128");
            if (ksplit > 1) { reduce(t.part, ksplit, M, N, static_cast<float*>(C), ldc); PG_LAUNCH_CHECK(h, "split-K reduction"); }
            if (rowsum && !a_kcont) { reduce(t.rs_part, ksplit * (int)g.x, 1, M, rowsum, M); PG_LAUNCH_CHECK(h, "row-sum reduction"); }
            return PG_OK;
        }
    }
    if (seg2) return pg_fail(h, PG_EINVAL, "two-segment GEMM outside the 128-tile bf16 kernel");
    // small or unaligned shapes (and operand type mixes the 128-tile kernel has no instantiation for): the 64-tile kernel,
    // then the mask / the row sums as kernels of their own
    if (mask && (ldc != N || ldm != N || ((dt & DT_C) != 0) != ((dt & DT_M) != 0)))
        return pg_fail(h, PG_EINVAL, "ReLU mask behind a strided small GEMM result is not supported");
    if (rowsum && a_kcont) return pg_fail(h, PG_EINVAL, "row sums need the m-contiguous A operand");
    const dim3 grid((N + GB - 1) / GB, (M + GB - 1) / GB, ksplit);
#define PG_SMALL(AK, BK_) hipLaunchKernelGGL((sgemm_kernel<AK, BK_>), grid, dim3(256), 0, s, M, N, K, A, sam, sak, B, sbk, sbn, C, ldc, bias, flags, t.part, dt, cin, ldcin)
    if (a_kcont && b_kcont) PG_SMALL(true, true);
    else if (a_kcont) PG_SMALL(true, false);
    else if (!b_kcont) PG_SMALL(false, false);
    else return pg_fail(h, PG_EINVAL, "unsupported GEMM operand layout");
#undef PG_SMALL
    PG_LAUNCH_CHECK(h, "sgemm");
    if (ksplit > 1) { reduce(t.part, ksplit, M, N, static_cast<float*>(C), ldc); PG_LAUNCH_CHECK(h, "split-K reduction"); }
    if (mask) {
        const unsigned blocks = (unsigned)std::min<long long>(((long long)M * N + 255) / 256, 8192);
        hipLaunchKernelGGL(relu_mask_kernel, dim3(blocks), dim3(256), 0, s, C, mask, (long long)M * N, (dt & DT_C) ? 1 : 0);
        PG_LAUNCH_CHECK(h, "relu_mask");
    }
    if (rowsum) {
        const unsigned blocks = (unsigned)((K + 255) / 256);
        if ((size_t)blocks * M > RS_FLOATS) return pg_fail(h, PG_EINVAL, "row-sum scratch too small");
        hipLaunchKernelGGL(colsum_wide_kernel, dim3(blocks), dim3(256), 0, s, A, (long long)K, M, sak, t.rs_part, abf ? 1 : 0);
        PG_LAUNCH_CHECK(h, "colsum");
        reduce(t.rs_part, (int)blocks, 1, M, rowsum, M);
        PG_LAUNCH_CHECK(h, "column-sum reduction");
    }
    return PG_OK;
}
// Y[P,out] = X[P,in] W[out,in]^T (+ b, relu, accumulate)        (nn.Linear forward)
// dt: DT_A = X is bf16, DT_C = Y is bf16; cin: the fp32 array GEMM_ACC adds (null: Y itself)
int linear_fwd(pg_handle* h, hipStream_t s, long long P, int out, int in, const void* X, long long ldx, const void* W, long long ldw,
               void* Y, long long ldy, const float* b, int flags, int dt = 0, const float* cin = nullptr, long long ldcin = 0) {
    // the heads (1 or 3 outputs, fp32 weights and result): one pass over X
    const bool xbf = dt & DT_A;
    const int vec = xbf ? 8 : 4, lpr = in / vec;
    if ((out == 1 || out == 3) && flags == 0 && !(dt & (DT_B | DT_C)) && in % vec == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0 &&
        ldx % vec == 0 && reinterpret_cast<uintptr_t>(X) % 16 == 0) {
        const long long nrg = 256 / lpr;
        const unsigned grid = (unsigned)std::min<long long>((P + nrg - 1) / nrg, 4096);
        const float* Wf = static_cast<const float*>(W);
        float* Yf = static_cast<float*>(Y);
#define PG_SKF(XT, NO) hipLaunchKernelGGL((skinny_fwd_kernel<XT, NO>), dim3(grid), dim3(256), 0, s, static_cast<const XT*>(X), ldx, P, in, Wf, ldw, b, Yf, ldy)
        if (xbf) { if (out == 1) PG_SKF(bf16_t, 1); else PG_SKF(bf16_t, 3); }
        else { if (out == 1) PG_SKF(float, 1); else PG_SKF(float, 3); }
#undef PG_SKF
        PG_LAUNCH_CHECK(h, "skinny forward");
        return PG_OK;
    }
    return gemm(h, s, true, true, (int)P, out, in, X, ldx, 1, W, 1, ldw, Y, ldy, b, flags, 1, nullptr, 0, nullptr, dt, cin, ldcin);
}
// Y[P,out] = [X1 | X2] [W1 | W2]^T + b (relu): a layer on the concatenation of two inputs in ONE pass (16-bit mode: no fp32
// partial sum through HBM); in1 a multiple of the k-step
int linear_fwd2(pg_handle* h, hipStream_t s, long long P, int out, int in1, const void* X1, long long ldx1, const void* W1, long long ldw1,
                int in2, const void* X2, long long ldx2, const void* W2, long long ldw2, void* Y, long long ldy, const float* b, int flags) {
    const KSeg2 sg{static_cast<const bf16_t*>(X2), static_cast<const bf16_t*>(W2), ldx2, ldw2, in2};
    return gemm(h, s, true, true, (int)P, out, in1, X1, ldx1, 1, W1, 1, ldw1, Y, ldy, b, flags, 1, nullptr, 0, nullptr, DT_A | DT_B | DT_C, nullptr, 0, &sg);
}
// dX[P,in] (+)= dY[P,out] W[out,in]
// relu_of: the stored post-activation the consumer of dX was ReLU'd to -- dX is zeroed where it is <= 0 (fused ReLU backward)
// dt: DT_A = dY, DT_C = dX, DT_M = relu_of are bf16
int linear_bwd_x(pg_handle* h, hipStream_t s, long long P, int out, int in, const void* dY, long long ldy, const void* W, long long ldw,
                 void* dX, long long ldx, int flags, const void* relu_of = nullptr, int dt = 0, const float* cin = nullptr, long long ldcin = 0,
                 const bf16_t* WT = nullptr, const Rank1* r1 = nullptr) {
    // the heads (1 or 3 outputs, fp32 dY and weights): one pass that writes dX
    if ((out == 1 || out == 3) && !WT && flags == 0 && !(dt & (DT_A | DT_B)) && in % 8 == 0 && ldx % 8 == 0 && reinterpret_cast<uintptr_t>(dX) % 16 == 0 &&
        (!relu_of || reinterpret_cast<uintptr_t>(relu_of) % 16 == 0) &&
        (!relu_of || ((dt & DT_M) != 0) == ((dt & DT_C) != 0))) {
        const unsigned grid = (unsigned)((P * (in / 8) + 255) / 256);
        const float* dYf = static_cast<const float*>(dY);
        const float* Wf = static_cast<const float*>(W);
#define PG_SKX(CT, NO) hipLaunchKernelGGL((skinny_bwd_x_kernel<CT, NO>), dim3(grid), dim3(256), 0, s, dYf, ldy, P, in, Wf, ldw, static_cast<CT*>(dX), ldx, static_cast<const CT*>(relu_of))
        if (dt & DT_C) { if (out == 1) PG_SKX(bf16_t, 1); else PG_SKX(bf16_t, 3); }
        else { if (out == 1) PG_SKX(float, 1); else PG_SKX(float, 3); }
#undef PG_SKX
        PG_LAUNCH_CHECK(h, "skinny input gradient");
        return PG_OK;
    }
    // WT: the step's transposed bf16 copy [in][out] of W's block -- both operands k-contiguous (16-byte tile stores, no transposing ones)
    if (WT) return gemm(h, s, true, true, (int)P, in, out, dY, ldy, 1, WT, 1, out, dX, ldx, nullptr, flags, 1, relu_of, ldx, nullptr, dt, cin, ldcin, nullptr, r1);
    if (r1) return pg_fail(h, PG_EINVAL, "a rank-1 term needs the transposed weight copy");
    return gemm(h, s, true, false, (int)P, in, out, dY, ldy, 1, W, ldw, 1, dX, ldx, nullptr, flags, 1, relu_of, ldx, nullptr, dt, cin, ldcin);
}
// dW[out,in] = dY[P,out]^T X[P,in] (split-K over the points, slices summed in order); dt: DT_A = dY, DT_B = X are bf16
int linear_bwd_w(pg_handle* h, hipStream_t s, long long P, int out, int in, const void* dY, long long ldy, const void* X, long long ldx,
                 float* dW, long long ldw, float* db = nullptr, int dt = 0) {      // db[out] += column sums of dY (the bias gradient, fused)
    {   // the heads' weight gradients (1 or 3 rows, dY fp32): one pass over X, per-block shares summed in order
        Tape& t = *tape_of(h);
        const bool xbf = dt & DT_B;
        const int vec = xbf ? 8 : 4, lpr = in / vec;
        if ((out == 1 || out == 3) && !db && !(dt & DT_A) && ldw == in && in % vec == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0 &&
            ldx % vec == 0 && reinterpret_cast<uintptr_t>(X) % 16 == 0 && t.part) {
            const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>(512, (P + 511) / 512));
            if ((size_t)grid * out * in > PART_FLOATS) return pg_fail(h, PG_EINVAL, "split-K scratch too small");
            const float* dYf = static_cast<const float*>(dY);
#define PG_SKW(XT, NO) hipLaunchKernelGGL((skinny_dw_kernel<XT, NO>), dim3(grid), dim3(256), 0, s, static_cast<const XT*>(X), ldx, P, in, dYf, ldy, t.part)
            if (xbf) { if (out == 1) PG_SKW(bf16_t, 1); else PG_SKW(bf16_t, 3); }
            else { if (out == 1) PG_SKW(float, 1); else PG_SKW(float, 3); }
#undef PG_SKW
            PG_LAUNCH_CHECK(h, "skinny weight gradient");
            hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((out * in + 255) / 256)), dim3(256), 0, s, t.part, (int)grid, out, in, dW, ldw);
            PG_LAUNCH_CHECK(h, "skinny weight gradient reduction");
            return PG_OK;
        }
    }
    const int tb = (out >= 64 && in >= 64) ? TB : GB;            // the tile gemm() will pick
    const int tiles = ((out + tb - 1) / tb) * ((in + tb - 1) / tb);
    // about two workgroups per CU (measured: 256 / 512 / 768 / 1024 workgroups -> 11.76 / 11.14 / 11.27 / 11.72 ms per 16-bit
    // step; POSEGEN_DW_WGS overrides): the GEMM streams dY and X once whatever the split, every slice costs a tile of partial
    // sums written and read again by the reduction
    static const int wg_target = [] { const char* e = std::getenv("POSEGEN_DW_WGS"); return e ? std::atoi(e) : 512; }();
    int ksplit = (int)std::max<long long>(1, std::min<long long>(wg_target / std::max(tiles, 1), (P + 1023) / 1024));
    return gemm(h, s, false, false, out, in, (int)P, dY, 1, ldy, X, ldx, 1, dW, ldw, nullptr, 0, std::max(ksplit, 2), nullptr, 0, db, dt);
}
int colsum(pg_handle* h, hipStream_t s, const float* d, long long rows, int N, long long ld, float* out) {
    Tape& t = *tape_of(h);
    if (N > 4) return pg_fail(h, PG_EINVAL, "colsum: at most 4 columns (the heads' bias gradients)");
    const unsigned blocks = (unsigned)((rows + CS_ROWS - 1) / CS_ROWS);
    if ((size_t)blocks * N > RS_FLOATS) return pg_fail(h, PG_EINVAL, "column-sum scratch too small");
    hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, s, d, rows, N, ld, t.rs_part, 0);
    PG_LAUNCH_CHECK(h, "colsum");
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, t.rs_part, (int)blocks, 1, N, out, (long long)N);
    PG_LAUNCH_CHECK(h, "column-sum reduction");
    return PG_OK;
}
int relu_mask(pg_handle* h, hipStream_t s, void* d, const void* hh, long long count, int bf) {
    const unsigned blocks = (unsigned)std::min<long long>((count + 255) / 256, 8192);
    hipLaunchKernelGGL(relu_mask_kernel, dim3(blocks), dim3(256), 0, s, d, hh, count, bf);
    PG_LAUNCH_CHECK(h, "relu_mask");
    return PG_OK;
}

#define PG_TRY(call) do { const int rc_ = (call); if (rc_) return rc_; } while (0)

// tensor i of a net in pg_load_weights order: 2l / 2l+1 = pts_linears.l.{weight,bias}; 16,17 alpha; 18,19 feature; 20,21 views; 22,23 rgb
// 16-bit mode: every activation of the tape is a bf16 array (A = DT_A, results DT_C); a layer made of two GEMMs (the skip
// layer, the view layer) sums in the fp32 array tmpF and rounds once
int mlp_forward(pg_handle* h, hipStream_t s, Tape& t, int net, const Pass& p, const pg_net_params& w, int fc) {
    const long long P = p.P;
    const int vk = CH_D + (fc ? FC_CH : 0), vcols = W + vk;
    const int bf = t.bf16 ? 1 : 0, es = t.es();
    const int ABC = bf ? (DT_A | DT_B | DT_C) : 0, AB = bf ? (DT_A | DT_B) : 0, A_ = bf ? DT_A : 0;
    // weight matrix i from column `off` on: the step's bf16 copy in the 16-bit mode (large GEMMs only)
    auto WT = [&](int i, long long off) -> const void* {
        return bf ? static_cast<const void*>(t.wb[net][i] + off) : static_cast<const void*>(w.w[i] + off);
    };
    PG_TRY(linear_fwd(h, s, P, W, CH_X, p.X, XW, WT(0, 0), CH_X, p.H[0], W, w.w[1], GEMM_RELU, ABC));
    for (int l = 1; l < DEPTH; ++l) {
        if (l == SKIP + 1 && bf) {  // h = cat([x, h]) in front of layer 5 (nerf.py:99-101): the h part's 256 columns first (a whole
                                    // number of k-steps), then the x part's 432
            PG_TRY(linear_fwd2(h, s, P, W, W, p.H[l - 1], W, WT(2 * l, CH_X), CH_X + W, CH_X, p.X, XW, WT(2 * l, 0), CH_X + W, p.H[l], W,
                               w.w[2 * l + 1], GEMM_RELU));
        } else if (l == SKIP + 1) {
            PG_TRY(linear_fwd(h, s, P, W, CH_X, p.X, XW, WT(2 * l, 0), CH_X + W, p.H[l], W, nullptr, 0, AB));
            PG_TRY(linear_fwd(h, s, P, W, W, p.H[l - 1], W, WT(2 * l, CH_X), CH_X + W, p.H[l], W, w.w[2 * l + 1], GEMM_ACC | GEMM_RELU, ABC));
        } else {
            PG_TRY(linear_fwd(h, s, P, W, W, p.H[l - 1], W, WT(2 * l, 0), W, p.H[l], W, w.w[2 * l + 1], GEMM_RELU, ABC));
        }
    }
    const void* h7 = p.H[DEPTH - 1];
    PG_TRY(linear_fwd(h, s, P, 1, W, h7, W, w.w[16], W, p.raw + 3, 4, w.w[17], 0, A_));
    PG_TRY(linear_fwd(h, s, P, W, W, h7, W, WT(18, 0), W, p.F, W, w.w[19], 0, ABC));
    if (bf) {
        PG_TRY(linear_fwd2(h, s, P, VW, W, p.F, W, WT(20, 0), vcols, vk, el_off(p.X, CH_X, es), XW, WT(20, W), vcols, p.G, VW, w.w[21], GEMM_RELU));
    } else {
        PG_TRY(linear_fwd(h, s, P, VW, W, p.F, W, WT(20, 0), vcols, p.G, VW, nullptr, 0, AB));
        PG_TRY(linear_fwd(h, s, P, VW, vk, el_off(p.X, CH_X, es), XW, WT(20, W), vcols, p.G, VW, w.w[21], GEMM_ACC | GEMM_RELU, ABC));
    }
    PG_TRY(linear_fwd(h, s, P, 3, VW, p.G, VW, w.w[22], VW, p.raw, 4, w.w[23], 0, A_));
    return PG_OK;
}

int mlp_backward(pg_handle* h, hipStream_t s, Tape& t, int net, const Pass& p, const pg_net_params& w, const pg_net_grads& g) {
    const long long P = p.P;
    const int fc = t.fc, vk = CH_D + (fc ? FC_CH : 0), vcols = W + vk;
    const int bf = t.bf16 ? 1 : 0, es = t.es();
    const int A_ = bf ? DT_A : 0, B_ = bf ? DT_B : 0, C_ = bf ? DT_C : 0, AB = A_ | B_, ABC = AB | C_, ABCM = bf ? (DT_A | DT_B | DT_C | DT_M) : 0;
    auto WT = [&](int i, long long off) -> const void* {        // (see mlp_forward)
        return bf ? static_cast<const void*>(t.wb[net][i] + off) : static_cast<const void*>(w.w[i] + off);
    };
    // (every gradient tensor is written whole below -- the ordered reductions overwrite, nothing accumulates into them)
    for (int i = 0; i < 24; ++i)
        if (!g.w[i]) return pg_fail(h, PG_EINVAL, "pg_train_backward: gradient tensor %d is null", i);
    const float* d_raw = t.d_raw;
    void* dG = t.dG;
    // rgb_linear: raw[:, :3] = G Wr^T + br
    PG_TRY(linear_bwd_x(h, s, P, 3, VW, d_raw, 4, w.w[22], VW, dG, VW, 0, p.G, bf ? (DT_C | DT_M) : 0));       // (masked by [G > 0] in the same pass)
    PG_TRY(linear_bwd_w(h, s, P, 3, VW, d_raw, 4, p.G, VW, g.w[22], VW, nullptr, B_));
    PG_TRY(colsum(h, s, d_raw, P, 3, 4, g.w[23]));
    // views_linears.0 on [feature | view embedding (| frame code)]
    PG_TRY(linear_bwd_w(h, s, P, VW, W, dG, VW, p.F, W, g.w[20], vcols, g.w[21], AB));
    PG_TRY(linear_bwd_w(h, s, P, VW, vk, dG, VW, el_off(p.X, CH_X, es), XW, g.w[20] + W, vcols, nullptr, AB));
    void* dF = t.tmpA;
    PG_TRY(linear_bwd_x(h, s, P, VW, W, dG, VW, WT(20, 0), vcols, dF, W, 0, nullptr, ABC, nullptr, 0, bf ? t.wbT[net][20] : nullptr));
    if (fc && g.codes) {
        PG_HIP(h, hipMemsetAsync(g.codes, 0, (size_t)w.n_codes * FC_CH * sizeof(float), s));
        PG_TRY(linear_bwd_x(h, s, P, VW, FC_CH, dG, VW, w.w[20] + W + CH_D, vcols, t.dC, FC_CH, 0, nullptr, A_));
        hipLaunchKernelGGL(code_ray_sum_kernel, dim3((unsigned)((t.n * FC_CH + 255) / 256)), dim3(256), 0, s, t.dC, (long long)t.n, p.S, t.ray_g);
        PG_LAUNCH_CHECK(h, "code_ray_sum");
        hipLaunchKernelGGL(code_gather_kernel, dim3((unsigned)w.n_codes), dim3(64), 0, s, t.ray_g, (long long)t.n, t.cams, w.n_codes, g.codes);
        PG_LAUNCH_CHECK(h, "code_gather");
    }
    // feature_linear and alpha_linear on the trunk output
    const void* h7 = p.H[DEPTH - 1];
    PG_TRY(linear_bwd_w(h, s, P, W, W, dF, W, h7, W, g.w[18], W, g.w[19], AB));
    void* dH = t.tmpB;              // dH7 = (alpha's part + feature's part) * [H7 > 0]: the mask rides on the second GEMM
    if (bf && lgemm_enabled()) {    // 16-bit mode: alpha's part d sigma (x) w_alpha rides in the feature GEMM's epilogue as a rank-1 term (fp32)
        const Rank1 r1{d_raw + 3, 4, w.w[16]};
        PG_TRY(linear_bwd_x(h, s, P, W, W, dF, W, WT(18, 0), W, dH, W, 0, h7, ABCM, nullptr, 0, t.wbT[net][18], &r1));
    } else {
        void* dpart = bf ? static_cast<void*>(t.tmpF) : dH;      // (16-bit mode: alpha's part in fp32, rounded once with the sum)
        PG_TRY(linear_bwd_x(h, s, P, 1, W, d_raw + 3, 4, w.w[16], W, dpart, W, 0));
        PG_TRY(linear_bwd_x(h, s, P, W, W, dF, W, WT(18, 0), W, dH, W, GEMM_ACC, h7, ABCM, static_cast<const float*>(dpart), W, bf ? t.wbT[net][18] : nullptr));
    }
    PG_TRY(linear_bwd_w(h, s, P, 1, W, d_raw + 3, 4, h7, W, g.w[16], W, nullptr, B_));
    PG_TRY(colsum(h, s, d_raw + 3, P, 1, 4, g.w[17]));
    // the trunk, back to front: dZ_l = dH_l * [H_l > 0]
    void* other = t.tmpA;
    // (dH arrives masked: the GEMM that produced it zeroed it where H_l <= 0; the bias gradient rides on a weight-gradient GEMM)
    for (int l = DEPTH - 1; l >= 0; --l) {
        if (l == 0) {
            PG_TRY(linear_bwd_w(h, s, P, W, CH_X, dH, W, p.X, XW, g.w[0], CH_X, g.w[1], AB));
        } else if (l == SKIP + 1) {
            PG_TRY(linear_bwd_w(h, s, P, W, CH_X, dH, W, p.X, XW, g.w[2 * l], CH_X + W, g.w[2 * l + 1], AB));
            PG_TRY(linear_bwd_w(h, s, P, W, W, dH, W, p.H[l - 1], W, g.w[2 * l] + CH_X, CH_X + W, nullptr, AB));
            PG_TRY(linear_bwd_x(h, s, P, W, W, dH, W, WT(2 * l, CH_X), CH_X + W, other, W, 0, p.H[l - 1], ABCM, nullptr, 0, bf ? t.wbT[net][2 * l] : nullptr));
            std::swap(dH, other);
        } else {
            PG_TRY(linear_bwd_w(h, s, P, W, W, dH, W, p.H[l - 1], W, g.w[2 * l], W, g.w[2 * l + 1], AB));
            PG_TRY(linear_bwd_x(h, s, P, W, W, dH, W, WT(2 * l, 0), W, other, W, 0, p.H[l - 1], ABCM, nullptr, 0, bf ? t.wbT[net][2 * l] : nullptr));
            std::swap(dH, other);
        }
    }
    return PG_OK;
}

}  // namespace pgt

extern "C" {

void pg_train_release(pg_handle* h) {
    if (!h || !h->train) return;
    pgt::Tape* t = static_cast<pgt::Tape*>(h->train);
    if (t->buf) (void)hipFree(t->buf);
    delete t;
    h->train = nullptr;
}

int pg_train_forward(pg_handle* h, void* stream, int64_t n, const float* ray_batch, const float* skts, int64_t pose_stride,
                     const float* cyls, int64_t cyl_stride, const float* cams, int n_samples, int n_importance, int flags,
                     const pg_train_draws* dr, const pg_net_params* coarse, const pg_net_params* fine, const pg_outputs* out,
                     int64_t* tape_id) {
    using namespace pgt;
    if (!h) return pg_fail(nullptr, PG_EINVAL, "null handle");
    if (n <= 0 || !ray_batch || !skts || !cyls || !coarse || !out) return pg_fail(h, PG_EINVAL, "pg_train_forward: null / non-positive argument");
    if (!h->emb_set[0] || !h->emb_set[1]) return pg_fail(h, PG_ESTATE, "embedder state not set (pg_set_embedder)");
    if (pose_stride != 0 && pose_stride != 384) return pg_fail(h, PG_EINVAL, "pose_stride must be 0 (shared) or 384 (per ray)");
    if (cyl_stride != 0 && cyl_stride != 5) return pg_fail(h, PG_EINVAL, "cyl_stride must be 0 (shared) or 5 (per ray)");
    const int S = n_samples, N = n_importance, SF = S + N;
    if (S < 2 || SF > pg_composite_max_samples() || N < 0 || N == 1 || N > pg_composite_max_importance())
        return pg_fail(h, PG_EINVAL, "pg_train_forward: N_samples %d / N_importance %d outside the supported range", S, N);
    if (N > 0 && !fine) return pg_fail(h, PG_EINVAL, "pg_train_forward: importance sampling needs the fine network's parameters");
    const int fc = h->cfg.framecode_ch > 0;
    for (int k = 0; k < (N > 0 ? 2 : 1); ++k) {
        const pg_net_params* p = k ? fine : coarse;
        for (int i = 0; i < 24; ++i) if (!p->w[i]) return pg_fail(h, PG_EINVAL, "pg_train_forward: parameter tensor %d of net %d is null", i, k);
        if (fc && (!p->codes || p->n_codes <= 0)) return pg_fail(h, PG_EINVAL, "pg_train_forward: frame codes of net %d missing", k);
    }
    PG_HIP(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    Tape& t = *tape_of(h);
    t.valid = false;
    // carve the tape
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    const long long Pc = n * S, Pf = N > 0 ? n * SF : 0, Pm = std::max(Pc, Pf);
    const bool rnoise = dr && dr->ray_noise;
    size_t need = al((size_t)n * 44) + al((size_t)n * 4) + al((size_t)n * 8) + al((size_t)n * S * 4) /*w0*/ + al((size_t)n * SF * 4) /*order*/;
    const bool bf = h->train_precision == PG_PREC_BF16;     // 16-bit mode (pg_set_train_precision): the tape's activations and their gradients are bf16 arrays
    const size_t es = bf ? 2 : 4;
    auto pass_bytes = [&](long long P) {
        return al((size_t)P * XW * es) + (DEPTH + 1) * al((size_t)P * W * es) + al((size_t)P * VW * es) + al((size_t)P * 16) + 2 * al((size_t)P * 4) + al((size_t)P * 12);
    };
    need += pass_bytes(Pc) + (N > 0 ? pass_bytes(Pf) : 0);
    need += 2 * al((size_t)Pm * W * es) + al((size_t)Pm * VW * es) + al((size_t)Pm * FC_CH * 4) + al((size_t)Pm * 16) + (bf ? al((size_t)Pm * W * 4) : 0);
    need += al(PART_FLOATS * 4) + al(RS_FLOATS * 4) + al((size_t)n * FC_CH * 4);
    const int vcols_ = W + CH_D + (fc ? FC_CH : 0);
    const size_t wsize[24] = {(size_t)W * CH_X, 0, (size_t)W * W, 0, (size_t)W * W, 0, (size_t)W * W, 0, (size_t)W * W, 0, (size_t)W * (CH_X + W), 0,
                              (size_t)W * W, 0, (size_t)W * W, 0, 0, 0, (size_t)W * W, 0, (size_t)VW * vcols_, 0, 0, 0};     // the large GEMMs' weight matrices
    // blocks of the weight matrices the dX GEMMs multiply by: (rows = out, cols = in, first column); layer 0 has no dX
    struct TB_ { int rows, cols, col0; };
    const TB_ tblock[24] = {{0, 0, 0}, {}, {W, W, 0}, {}, {W, W, 0}, {}, {W, W, 0}, {}, {W, W, 0}, {}, {W, W, CH_X}, {}, {W, W, 0}, {}, {W, W, 0}, {}, {}, {},
                            {W, W, 0}, {}, {VW, W, 0}, {}, {}, {}};
    if (bf) for (int i = 0; i < 24; ++i) need += 2 * al(wsize[i] * 2) + 2 * al((size_t)tblock[i].rows * tblock[i].cols * 2);
    if (need > t.bytes) {
        if (t.buf) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(t.buf)); t.buf = nullptr; t.bytes = 0; }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&t.buf), need);
        if (e != hipSuccess) return pg_fail(h, PG_ENOMEM, "training tape of %zu bytes failed: %s", need, hipGetErrorString(e));
        t.bytes = need;
    }
    uint8_t* q = t.buf;
    auto take = [&](size_t b) { float* r = reinterpret_cast<float*>(q); q += al(b); return r; };
    t.n = n; t.S = S; t.N = N; t.fc = fc; t.has_fine = N > 0;
    t.bf16 = bf;
    t.rays = take((size_t)n * 44);
    t.cams = cams ? take((size_t)n * 4) : (take((size_t)n * 4), nullptr);
    float* nf = take((size_t)n * 8);
    float* w0 = take((size_t)n * S * 4);
    int* order = reinterpret_cast<int*>(take((size_t)n * SF * 4));
    for (int k = 0; k < 2; ++k) {
        Pass& p = t.pass[k];
        p = Pass();
        if (k == 1 && N == 0) break;
        p.P = k ? Pf : Pc; p.S = k ? SF : S;
        p.X = take((size_t)p.P * XW * es);
        for (int l = 0; l < DEPTH; ++l) p.H[l] = take((size_t)p.P * W * es);
        p.F = take((size_t)p.P * W * es);
        p.G = take((size_t)p.P * VW * es);
        p.raw = take((size_t)p.P * 16);
        p.z = take((size_t)p.P * 4);
        p.noise = take((size_t)p.P * 4);
        p.pn = take((size_t)p.P * 12);
    }
    t.tmpA = take((size_t)Pm * W * es); t.tmpB = take((size_t)Pm * W * es);
    t.dG = take((size_t)Pm * VW * es); t.dC = take((size_t)Pm * FC_CH * 4); t.d_raw = take((size_t)Pm * 16);
    t.tmpF = bf ? take((size_t)Pm * W * 4) : nullptr;
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 24; ++i) {
            t.wb[k][i] = (bf && wsize[i]) ? reinterpret_cast<bf16_t*>(take(wsize[i] * 2)) : nullptr;
            t.wbT[k][i] = (bf && tblock[i].rows) ? reinterpret_cast<bf16_t*>(take((size_t)tblock[i].rows * tblock[i].cols * 2)) : nullptr;
        }
    if (bf) {       // this step's bf16 copies of the weight matrices (the parameters do not change between forward and backward): one launch
        WJobs js{};
        int nj = 0;
        for (int k = 0; k < (N > 0 ? 2 : 1); ++k)
            for (int i = 0; i < 24; ++i) {
                if (!wsize[i]) continue;
                const float* src = (k ? fine : coarse)->w[i];
                js.j[nj++] = WJob{src, t.wb[k][i], 0, 1, (int)wsize[i], 0};
                if (tblock[i].rows) {
                    const long long ldw = (long long)(wsize[i] / tblock[i].rows);       // (= the matrix's column count)
                    js.j[nj++] = WJob{src + tblock[i].col0, t.wbT[k][i], ldw, tblock[i].rows, tblock[i].cols, 1};
                }
            }
        if (nj > WJOBS_MAX) return pg_fail(h, PG_EINVAL, "weight conversion: %d jobs", nj);
        hipLaunchKernelGGL(cvt_weights_kernel, dim3(64, nj), dim3(256), 0, s, js);
        PG_LAUNCH_CHECK(h, "weight conversion");
    }
    t.part = take(PART_FLOATS * 4); t.rs_part = take(RS_FLOATS * 4); t.ray_g = take((size_t)n * FC_CH * 4);
    t.params[0] = *coarse;
    if (N > 0) t.params[1] = *fine;
    PG_HIP(h, hipMemcpyAsync(t.rays, ray_batch, (size_t)n * 44, hipMemcpyDeviceToDevice, s));
    if (cams) PG_HIP(h, hipMemcpyAsync(t.cams, cams, (size_t)n * 4, hipMemcpyDeviceToDevice, s));

    // frame codes: [n_codes + 1, 16], the caller appends the mean row (embedding.py:25-26) that a negative index selects
    const float* codes_dev[2] = {coarse->codes, N > 0 ? fine->codes : nullptr};

    Pass& pc = t.pass[0];
    double* scs = nullptr;
    { const int rc_ = pg_sc_scratch(h, n, h->cfg.chunk, &scs); if (rc_) return rc_; }
    int e = pg_launch_sample_coarse(t.rays, cyls, cyl_stride, n, h->cfg.chunk, S, (flags & PG_FLAG_LINDISP) ? 1 : 0, nf, pc.z, dr ? dr->t_rand : nullptr, scs, stream);
    if (e) return pg_fail(h, PG_EHIP, "coarse sampling launch failed: %s", hipGetErrorString((hipError_t)e));
    if (dr && dr->noise0) PG_HIP(h, hipMemcpyAsync(pc.noise, dr->noise0, (size_t)Pc * 4, hipMemcpyDeviceToDevice, s));
    if (rnoise) {
        e = pg_launch_gather_noise(dr->ray_noise, n, SF, S, nullptr, pc.pn, stream);
        if (e) return pg_fail(h, PG_EHIP, "noise gather launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    auto embed = [&](Pass& p, const float* codes, int n_codes) {
        const dim3 grid((unsigned)((p.P * J + 255) / 256));
        if (bf) hipLaunchKernelGGL(embed_rows_kernel<bf16_t>, grid, dim3(256), 0, s, t.rays, p.z, rnoise ? p.pn : nullptr, skts, (long long)pose_stride,
                                   t.cams, codes, n_codes, fc, h->d_cut, h->tau[0], h->tau[1], p.P, p.S, static_cast<bf16_t*>(p.X));
        else hipLaunchKernelGGL(embed_rows_kernel<float>, grid, dim3(256), 0, s, t.rays, p.z, rnoise ? p.pn : nullptr, skts, (long long)pose_stride,
                                t.cams, codes, n_codes, fc, h->d_cut, h->tau[0], h->tau[1], p.P, p.S, static_cast<float*>(p.X));
        return hipGetLastError();
    };
    if (embed(pc, codes_dev[0], coarse->n_codes) != hipSuccess) return pg_fail(h, PG_EHIP, "embedding kernel launch failed");
    PG_TRY(mlp_forward(h, s, t, 0, pc, *coarse, fc));
    const bool hier = N > 0;
    e = pg_launch_composite(t.rays, pc.z, pc.raw, n, S, h->cfg.density_scale, h->cfg.rgb_eps, h->cfg.density_act, h->cfg.softplus_shift, hier ? out->rgb0 : out->rgb_map,
                            hier ? out->disp0 : out->disp_map, hier ? out->acc0 : out->acc_map, hier ? out->alpha0 : out->alpha,
                            out->weights0 ? out->weights0 : w0, N, hier ? t.pass[1].z : nullptr, (dr && dr->noise0) ? pc.noise : nullptr,
                            dr ? dr->u_rand : nullptr, (rnoise && hier) ? order : nullptr, stream);
    if (e) return pg_fail(h, PG_EHIP, "composite launch failed: %s", hipGetErrorString((hipError_t)e));
    if (!(dr && dr->noise0)) pc.noise = nullptr;
    if (hier) {
        Pass& pf = t.pass[1];
        if (dr && dr->noise1) PG_HIP(h, hipMemcpyAsync(pf.noise, dr->noise1, (size_t)Pf * 4, hipMemcpyDeviceToDevice, s));
        else pf.noise = nullptr;
        if (rnoise) {
            e = pg_launch_gather_noise(dr->ray_noise, n, SF, SF, order, pf.pn, stream);
            if (e) return pg_fail(h, PG_EHIP, "noise gather launch failed: %s", hipGetErrorString((hipError_t)e));
        }
        if (embed(pf, codes_dev[1], fine->n_codes) != hipSuccess) return pg_fail(h, PG_EHIP, "embedding kernel launch failed");
        PG_TRY(mlp_forward(h, s, t, 1, pf, *fine, fc));
        e = pg_launch_composite(t.rays, pf.z, pf.raw, n, SF, h->cfg.density_scale, h->cfg.rgb_eps, h->cfg.density_act, h->cfg.softplus_shift, out->rgb_map, out->disp_map, out->acc_map,
                                out->alpha, nullptr, 0, nullptr, pf.noise, nullptr, nullptr, stream);
        if (e) return pg_fail(h, PG_EHIP, "composite launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    if (out->near_far) PG_HIP(h, hipMemcpyAsync(out->near_far, nf, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
    if (out->z_coarse) PG_HIP(h, hipMemcpyAsync(out->z_coarse, pc.z, (size_t)Pc * 4, hipMemcpyDeviceToDevice, s));
    if (out->raw_coarse) PG_HIP(h, hipMemcpyAsync(out->raw_coarse, pc.raw, (size_t)Pc * 16, hipMemcpyDeviceToDevice, s));
    if (hier && out->z_fine) PG_HIP(h, hipMemcpyAsync(out->z_fine, t.pass[1].z, (size_t)Pf * 4, hipMemcpyDeviceToDevice, s));
    if (hier && out->raw_fine) PG_HIP(h, hipMemcpyAsync(out->raw_fine, t.pass[1].raw, (size_t)Pf * 16, hipMemcpyDeviceToDevice, s));
    t.valid = true;
    t.generation += 1;
    if (tape_id) *tape_id = t.generation;
    return PG_OK;
}

int pg_train_backward(pg_handle* h, void* stream, int64_t tape_id, const float* d_rgb_map, const float* d_acc_map, const float* d_rgb0,
                      const float* d_acc0, const pg_net_grads* coarse, const pg_net_grads* fine) {
    using namespace pgt;
    if (!h) return pg_fail(nullptr, PG_EINVAL, "null handle");
    if (!h->train || !static_cast<Tape*>(h->train)->valid) return pg_fail(h, PG_ESTATE, "pg_train_backward: no forward pass on the tape (pg_train_forward)");
    Tape& t = *static_cast<Tape*>(h->train);
    // ONE forward is outstanding per handle: a later pg_train_forward reuses the tape, and its activations must not be
    // taken for those of the pass this backward belongs to (two batches summed into one loss, a delayed backward)
    if (tape_id != t.generation)
        return pg_fail(h, PG_ESTATE, "pg_train_backward: the tape of forward pass %lld has been overwritten by forward pass %lld "
                       "(one outstanding pg_train_forward per handle: run backward before the next forward)",
                       (long long)tape_id, (long long)t.generation);
    if (!coarse || (t.has_fine && !fine)) return pg_fail(h, PG_EINVAL, "pg_train_backward: null gradient struct");
    PG_HIP(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto run = [&](int k, const float* d_rgb, const float* d_acc, const pg_net_grads& g) -> int {
        const Pass& p = t.pass[k];
        hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((t.n + 63) / 64)), dim3(64), 0, s, t.rays, p.z, p.raw, p.noise, (long long)t.n, p.S,
                           h->cfg.density_scale, h->cfg.rgb_eps, h->cfg.density_act, h->cfg.softplus_shift, d_rgb, d_acc, t.d_raw);
        PG_LAUNCH_CHECK(h, "composite backward");
        return mlp_backward(h, s, t, k, p, t.params[k], g);
    };
    if (t.has_fine) {
        PG_TRY(run(1, d_rgb_map, d_acc_map, *fine));
        PG_TRY(run(0, d_rgb0, d_acc0, *coarse));
    } else {
        PG_TRY(run(0, d_rgb_map, d_acc_map, *coarse));
    }
    return PG_OK;
}

}  // extern "C"
