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
// Exact fp32 arithmetic.  Training batches are small (N_rand = 2048 rays -> 131 k + 164 k points,
// configs/surreal/surreal.txt:34), so the 1080-wide embedding and the layer activations are MATERIALISED in HBM
// (14 KB per point, 4 GB per batch -- 1.4 % of the card) and every layer is a plain fp32 GEMM on
// v_mfma_f32_32x32x2_f32 (128 x 128 x 16 tiles for the trunk shapes, a 64-tile kernel for the heads and unaligned
// shapes); the weight gradients split K over the points and are reduced in a fixed order (bitwise repeatable).  The
// fused inference kernels are not involved.
#include <hip/hip_runtime.h>

#include <algorithm>
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

// ---- the 1080 (+16) network inputs of every point, one thread per (point, joint) --------------------------------
// RelDist / VecNorm encoders on bone-local coordinates + cutoff positional embedding, channel order of the reference:
// v part row * 24 + j, direction part 360 + 3 j + c, view part 432 + row * 72 + 3 j + c (core/encoders.py:8-37,
// 101-122, 172-193; core/cutoff_embedder.py:111-174).  sin / cos of every octave directly (no doubling chain).
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ rays, const float* __restrict__ z,
                                                         const float* __restrict__ pnoise, const float* __restrict__ skts,
                                                         long long pose_stride, const float* __restrict__ cams,
                                                         const float* __restrict__ codes, int n_codes, int fc,
                                                         const float* __restrict__ cutoff, float tau_v, float tau_d,
                                                         long long n_points, int S, float* __restrict__ X) {
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
    float* x = X + pt * XW;
    {
        const float w = 1.0f - 1.0f / (1.0f + expf(-tau_v * (v - cutoff[j])));
        x[j] = v * w;
        float f = 1.0f;
#pragma unroll
        for (int k = 0; k < LV; ++k, f *= 2.0f) {
            float s, c;
            sincosf(f * v, &s, &c);
            x[(1 + 2 * k) * J + j] = s * w;
            x[(2 + 2 * k) * J + j] = c * w;
        }
        const float den = fmaxf(v, 1e-12f);
        x[CH_V + 3 * j] = qx / den; x[CH_V + 3 * j + 1] = qy / den; x[CH_V + 3 * j + 2] = qz / den;
    }
    {
        const float dx = rb[3], dy = rb[4], dz = rb[5];
        float e[3] = {fmaf(sk[2], dz, fmaf(sk[1], dy, sk[0] * dx)), fmaf(sk[6], dz, fmaf(sk[5], dy, sk[4] * dx)),
                      fmaf(sk[10], dz, fmaf(sk[9], dy, sk[8] * dx))};
        const float den = fmaxf(sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), 1e-12f);
        const float w = 1.0f - 1.0f / (1.0f + expf(-tau_d * (v - cutoff[J + j])));
        float* xd = x + CH_X;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ev = e[c] / den;
            xd[3 * j + c] = ev * w;
            float f = 1.0f;
#pragma unroll
            for (int k = 0; k < LD; ++k, f *= 2.0f) {
                float s, co;
                sincosf(f * ev, &s, &co);
                xd[(1 + 2 * k) * (3 * J) + 3 * j + c] = s * w;
                xd[(2 + 2 * k) * (3 * J) + 3 * j + c] = co * w;
            }
        }
    }
    if (j == 0) {
        float* xc = x + CH_X + CH_D;
        if (fc) {
            const float cam = cams ? cams[ray] : -1.0f;
            const int ci = cam < 0.0f ? n_codes : min((int)cam, n_codes - 1);      // row n_codes = the mean code
#pragma unroll
            for (int k = 0; k < FC_CH; ++k) xc[k] = codes[ci * FC_CH + k];
        } else {
#pragma unroll
            for (int k = 0; k < FC_CH; ++k) xc[k] = 0.0f;
        }
    }
}

// ---- fp32 GEMM: C[M,N] (op)= A[M,K] B[K,N] with element strides, 64 x 64 x 16 tiles on v_mfma_f32_32x32x2_f32 ------
// A(m,k) = A[m sam + k sak], B(k,n) = B[k sbk + n sbn]; the template flags say which index is contiguous (coalesced
// tile loads).  gridDim.z > 1 splits K: the partial tiles go to a scratch array and are summed in slice order.
constexpr int GB = 64, GK = 16;
enum { GEMM_ACC = 1, GEMM_RELU = 2 };

template <bool A_KCONT, bool B_KCONT>
__global__ __launch_bounds__(256) void sgemm_kernel(int M, int N, int K, const float* __restrict__ A, long long sam, long long sak,
                                                    const float* __restrict__ B, long long sbk, long long sbn,
                                                    float* __restrict__ C, long long ldc, const float* __restrict__ bias, int flags,
                                                    float* __restrict__ part) {
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
                As[kq + i][m] = (m0 + m < M && k < k1) ? A[(long long)(m0 + m) * sam + (long long)k * sak] : 0.0f;
            }
        } else {
            const int m = (t & 15) * 4, k = kb + (t >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                As[t >> 4][m + i] = (m0 + m + i < M && k < k1) ? A[(long long)(m0 + m + i) * sam + (long long)k * sak] : 0.0f;
        }
        if (B_KCONT) {
            const int n = t >> 2, kq = (t & 3) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = kb + kq + i;
                Bs[kq + i][n] = (n0 + n < N && k < k1) ? B[(long long)k * sbk + (long long)(n0 + n) * sbn] : 0.0f;
            }
        } else {
            const int n = (t & 15) * 4, k = kb + (t >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                Bs[t >> 4][n + i] = (n0 + n + i < N && k < k1) ? B[(long long)k * sbk + (long long)(n0 + n + i) * sbn] : 0.0f;
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
            float* c = C + (long long)m * ldc + n;
            float v = acc[r];
            if (nz > 1) { part[((long long)blockIdx.z * M + m) * N + n] = v; continue; }
            if (flags & GEMM_ACC) v += *c;
            if (bias) v += bias[n];
            if (flags & GEMM_RELU) v = fmaxf(v, 0.0f);
            *c = v;
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
                                                       float* __restrict__ part, float* __restrict__ rs_part) {
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
                if (flags & GEMM_ACC) v += *c;
                v += bv;
                if (flags & GEMM_RELU) v = fmaxf(v, 0.0f);
                if (mask && !(mask[(long long)m * ldm + n] > 0.0f)) v = 0.0f;
                *c = v;
            }
    }
    if (!A_KCONT && rowsum && t < TB && m0 + t < M) rs_part[((long long)blockIdx.z * gridDim.x + blockIdx.x) * M + m0 + t] = rs;
}

// ---- the same GEMM with bf16 operands (fp32 in HBM, rounded on the way into LDS; fp32 accumulate): the 16-bit training
// mode.  128 x 128 x 32 tiles on v_mfma_f32_32x32x16_bf16, LDS rows [row][32 k + 8 pad] bf16 (80-byte pitch: the 16-byte
// fragment reads of a 16-lane group fall on 16 different bank quads), k-contiguous operands stored 4 k at a time,
// k-strided ones as (k, k + 1) pairs.
constexpr int BK = 32, BPITCH = BK + 8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8t;

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

template <bool KCONT>
__device__ __forceinline__ void btile_fetch(float4 (&v)[4], const float* __restrict__ P, long long s_row, long long s_k,
                                            int row0, int rows, int kb, int k1, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KCONT) {            // 4 consecutive k of one row
            const int idx = t + 256 * i, r = idx >> 3, k = kb + (idx & 7) * 4;
            if (row0 + r < rows && k < k1) v[i] = *reinterpret_cast<const float4*>(P + (long long)(row0 + r) * s_row + k);
        } else {                // 4 consecutive rows of k = 2 kp and (i odd) 2 kp + 1
            const int kp = (t >> 5) + 8 * (i >> 1), k = kb + 2 * kp + (i & 1), r = (t & 31) * 4;
            if (row0 + r < rows && k < k1) v[i] = *reinterpret_cast<const float4*>(P + (long long)k * s_k + row0 + r);
        }
    }
}

template <bool KCONT>
__device__ __forceinline__ void btile_store(unsigned short (*T)[BPITCH], const float4 (&v)[4], int t) {
    if (KCONT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = t + 256 * i, r = idx >> 3, kq = (idx & 7) * 4;
            *reinterpret_cast<uint2*>(&T[r][kq]) = make_uint2(pack_bf16(v[i].x, v[i].y), pack_bf16(v[i].z, v[i].w));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kp = (t >> 5) + 8 * i, r = (t & 31) * 4;
            const float4 a = v[2 * i], b = v[2 * i + 1];
            *reinterpret_cast<unsigned*>(&T[r][2 * kp]) = pack_bf16(a.x, b.x);
            *reinterpret_cast<unsigned*>(&T[r + 1][2 * kp]) = pack_bf16(a.y, b.y);
            *reinterpret_cast<unsigned*>(&T[r + 2][2 * kp]) = pack_bf16(a.z, b.z);
            *reinterpret_cast<unsigned*>(&T[r + 3][2 * kp]) = pack_bf16(a.w, b.w);
        }
    }
}

template <bool A_KCONT, bool B_KCONT>
__global__ __launch_bounds__(256) void bgemm128_kernel(int M, int N, int K, const float* __restrict__ A, long long sam, long long sak,
                                                       const float* __restrict__ B, long long sbk, long long sbn,
                                                       float* __restrict__ C, long long ldc, const float* __restrict__ bias, int flags,
                                                       const float* __restrict__ mask, long long ldm, float* __restrict__ rowsum,
                                                       float* __restrict__ part, float* __restrict__ rs_part) {
    __shared__ __attribute__((aligned(16))) unsigned short As[TB][BPITCH], Bs[TB][BPITCH];
    float rs = 0.0f;
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6, wm = (wv >> 1) * 64, wn = (wv & 1) * 64, li = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * TB, n0 = blockIdx.x * TB;
    const int nz = gridDim.z;
    const int kper = ((K + nz - 1) / nz + BK - 1) / BK * BK;
    const int k0 = blockIdx.z * kper, k1 = min(K, k0 + kper);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float4 va[4], vb[4];
    btile_fetch<A_KCONT>(va, A, sam, sak, m0, M, k0, k1, t);
    btile_fetch<B_KCONT>(vb, B, sbn, sbk, n0, N, k0, k1, t);
    for (int kb = k0; kb < k1; kb += BK) {
        btile_store<A_KCONT>(As, va, t);
        btile_store<B_KCONT>(Bs, vb, t);
        __syncthreads();
        btile_fetch<A_KCONT>(va, A, sam, sak, m0, M, kb + BK, k1, t);       // (all zeros past the end)
        btile_fetch<B_KCONT>(vb, B, sbn, sbk, n0, N, kb + BK, k1, t);
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
                if (flags & GEMM_ACC) v += *c;
                v += bv;
                if (flags & GEMM_RELU) v = fmaxf(v, 0.0f);
                if (mask && !(mask[(long long)m * ldm + n] > 0.0f)) v = 0.0f;
                *c = v;
            }
    }
    if (!A_KCONT && rowsum && t < TB && m0 + t < M) rs_part[((long long)blockIdx.z * gridDim.x + blockIdx.x) * M + m0 + t] = rs;
}

// out[i (row-major M x N with leading dimension ldo)] = sum over the nz slices of part[z][M][N], in slice order
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ part, int nz, int M, int N,
                                                          float* __restrict__ out, long long ldo) {
    const long long i = blockIdx.x * 256ll + threadIdx.x, MN = (long long)M * N;
    if (i >= MN) return;
    float s = 0.0f;
    for (int z = 0; z < nz; ++z) s += part[z * MN + i];
    out[(i / N) * ldo + (i % N)] = s;
}

// dH <- dH where H > 0 else 0 (ReLU backward on the stored post-activation)
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ d, const float* __restrict__ h, long long count) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < count; i += (long long)gridDim.x * 256)
        if (!(h[i] > 0.0f)) d[i] = 0.0f;
}

// part[block][n] = sum over the block's 256 rows of d[row * ld + n] (bias gradients; reduce_parts_kernel adds the
// blocks in order)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ d, long long rows, int N, long long ld, float* __restrict__ part) {
    const long long r0 = blockIdx.x * 256ll, r1 = min(rows, r0 + 256);
    for (int n = threadIdx.x; n < N; n += 256) {
        float s = 0.0f;
        for (long long r = r0; r < r1; ++r) s += d[r * ld + n];
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
    float *X = nullptr, *H[DEPTH] = {}, *F = nullptr, *G = nullptr, *raw = nullptr, *z = nullptr, *noise = nullptr, *pn = nullptr;
};

struct Tape {
    uint8_t* buf = nullptr;
    size_t bytes = 0;
    bool valid = false;
    int64_t generation = 0;     // id of the forward pass the tape holds (pg_train_forward returns it, pg_train_backward checks it)
    long long n = 0;
    int S = 0, N = 0, fc = 0;
    float *rays = nullptr, *cams = nullptr;
    float *tmpA = nullptr, *tmpB = nullptr, *dG = nullptr, *dC = nullptr, *d_raw = nullptr;
    float *part = nullptr, *rs_part = nullptr, *ray_g = nullptr;      // split-K slices, row / column sum shares, per-ray code gradients
    Pass pass[2];
    pg_net_params params[2];
    bool has_fine = false;
    bool bf16 = false;          // 16-bit training mode (handle precision PG_PREC_BF16): bf16 operands in the large GEMMs
};

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
int gemm(pg_handle* h, hipStream_t s, bool a_kcont, bool b_kcont, int M, int N, int K, const float* A, long long sam, long long sak,
         const float* B, long long sbk, long long sbn, float* C, long long ldc, const float* bias, int flags, int ksplit = 1,
         const float* mask = nullptr, long long ldm = 0, float* rowsum = nullptr) {
    if (M <= 0 || N <= 0 || K <= 0) return PG_OK;
    Tape& t = *tape_of(h);
    if (ksplit > 1) {
        const long long fit = (long long)(PART_FLOATS / ((size_t)M * N));
        if (fit < 2) return pg_fail(h, PG_EINVAL, "split-K scratch too small for a %d x %d result", M, N);
        if (ksplit > fit) ksplit = (int)fit;
        if (!t.part) return pg_fail(h, PG_ESTATE, "split-K GEMM without a tape");
    }
    auto reduce = [&](const float* part, int nz, int rows, int cols, float* out, long long ldo) {
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)(((long long)rows * cols + 255) / 256)), dim3(256), 0, s, part, nz, rows, cols, out, ldo);
    };
    auto al4 = [](const void* p, long long a, long long b) { return reinterpret_cast<uintptr_t>(p) % 16 == 0 && a % 4 == 0 && b % 4 == 0; };
    // (the strides that are not 1 must keep 16-byte alignment of every row / k start; M, N, K multiples of 4)
    const bool big = M >= 64 && N >= 64 && M % 4 == 0 && N % 4 == 0 && K % 4 == 0 &&
                     al4(A, a_kcont ? sam : sak, 4) && al4(B, b_kcont ? sbn : sbk, 4);
    if (big && !(a_kcont == false && b_kcont == true)) {
        const dim3 g((N + TB - 1) / TB, (M + TB - 1) / TB, ksplit);
        if (rowsum && (size_t)ksplit * g.x * M > RS_FLOATS) return pg_fail(h, PG_EINVAL, "row-sum scratch too small");
#define PG_GEMM128(KERNEL, AK, BK_) hipLaunchKernelGGL((KERNEL<AK, BK_>), g, dim3(256), 0, s, M, N, K, A, sam, sak, B, sbk, sbn, C, ldc, bias, flags, mask, ldm, rowsum, t.part, t.rs_part)
        if (t.bf16) {
            if (a_kcont && b_kcont) PG_GEMM128(bgemm128_kernel, true, true);
            else if (a_kcont) PG_GEMM128(bgemm128_kernel, true, false);
            else PG_GEMM128(bgemm128_kernel, false, false);
        } else {
            if (a_kcont && b_kcont) PG_GEMM128(sgemm128_kernel, true, true);
            else if (a_kcont) PG_GEMM128(sgemm128_kernel, true, false);
            else PG_GEMM128(sgemm128_kernel, false, false);
        }
#undef PG_GEMM128
        PG_LAUNCH_CHECK(h, "gemm128");
        if (ksplit > 1) { reduce(t.part, ksplit, M, N, C, ldc); PG_LAUNCH_CHECK(h, "split-K reduction"); }
        if (rowsum && !a_kcont) { reduce(t.rs_part, ksplit * (int)g.x, M, 1, rowsum, 1); PG_LAUNCH_CHECK(h, "row-sum reduction"); }
        return PG_OK;
    }
    // small or unaligned shapes: the 64-tile kernel, then the mask / the row sums as kernels of their own
    if (mask && (ldc != N || ldm != N)) return pg_fail(h, PG_EINVAL, "ReLU mask behind a strided GEMM result is not supported");
    if (rowsum && a_kcont) return pg_fail(h, PG_EINVAL, "row sums need the m-contiguous A operand");
    const dim3 grid((N + GB - 1) / GB, (M + GB - 1) / GB, ksplit);
    if (a_kcont && b_kcont) hipLaunchKernelGGL((sgemm_kernel<true, true>), grid, dim3(256), 0, s, M, N, K, A, sam, sak, B, sbk, sbn, C, ldc, bias, flags, t.part);
    else if (a_kcont) hipLaunchKernelGGL((sgemm_kernel<true, false>), grid, dim3(256), 0, s, M, N, K, A, sam, sak, B, sbk, sbn, C, ldc, bias, flags, t.part);
    else if (!b_kcont) hipLaunchKernelGGL((sgemm_kernel<false, false>), grid, dim3(256), 0, s, M, N, K, A, sam, sak, B, sbk, sbn, C, ldc, bias, flags, t.part);
    else return pg_fail(h, PG_EINVAL, "unsupported GEMM operand layout");
    PG_LAUNCH_CHECK(h, "sgemm");
    if (ksplit > 1) { reduce(t.part, ksplit, M, N, C, ldc); PG_LAUNCH_CHECK(h, "split-K reduction"); }
    if (mask) {
        const unsigned blocks = (unsigned)std::min<long long>(((long long)M * N + 255) / 256, 8192);
        hipLaunchKernelGGL(relu_mask_kernel, dim3(blocks), dim3(256), 0, s, C, mask, (long long)M * N);
        PG_LAUNCH_CHECK(h, "relu_mask");
    }
    if (rowsum) {
        const unsigned blocks = (unsigned)((K + 255) / 256);
        if ((size_t)blocks * M > RS_FLOATS) return pg_fail(h, PG_EINVAL, "row-sum scratch too small");
        hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, s, A, (long long)K, M, sak, t.rs_part);
        PG_LAUNCH_CHECK(h, "colsum");
        reduce(t.rs_part, (int)blocks, 1, M, rowsum, M);
        PG_LAUNCH_CHECK(h, "column-sum reduction");
    }
    return PG_OK;
}
// Y[P,out] = X[P,in] W[out,in]^T (+ b, relu, accumulate)        (nn.Linear forward)
int linear_fwd(pg_handle* h, hipStream_t s, long long P, int out, int in, const float* X, long long ldx, const float* W, long long ldw,
               float* Y, long long ldy, const float* b, int flags) {
    return gemm(h, s, true, true, (int)P, out, in, X, ldx, 1, W, 1, ldw, Y, ldy, b, flags);
}
// dX[P,in] (+)= dY[P,out] W[out,in]
// relu_of: the stored post-activation the consumer of dX was ReLU'd to -- dX is zeroed where it is <= 0 (fused ReLU backward)
int linear_bwd_x(pg_handle* h, hipStream_t s, long long P, int out, int in, const float* dY, long long ldy, const float* W, long long ldw,
                 float* dX, long long ldx, int flags, const float* relu_of = nullptr) {
    return gemm(h, s, true, false, (int)P, in, out, dY, ldy, 1, W, ldw, 1, dX, ldx, nullptr, flags, 1, relu_of, ldx);
}
// dW[out,in] = dY[P,out]^T X[P,in] (split-K over the points, slices summed in order)
int linear_bwd_w(pg_handle* h, hipStream_t s, long long P, int out, int in, const float* dY, long long ldy, const float* X, long long ldx,
                 float* dW, long long ldw, float* db = nullptr) {      // db[out] += column sums of dY (the bias gradient, fused)
    const int tb = (out >= 64 && in >= 64) ? TB : GB;            // the tile gemm() will pick
    const int tiles = ((out + tb - 1) / tb) * ((in + tb - 1) / tb);
    // about one workgroup per CU (the GEMM streams dY and X once whatever the split): every slice costs a tile of partial
    // sums written and read again by the reduction
    int ksplit = (int)std::max<long long>(1, std::min<long long>(256 / std::max(tiles, 1), (P + 2047) / 2048));
    return gemm(h, s, false, false, out, in, (int)P, dY, 1, ldy, X, ldx, 1, dW, ldw, nullptr, 0, std::max(ksplit, 2), nullptr, 0, db);
}
int colsum(pg_handle* h, hipStream_t s, const float* d, long long rows, int N, long long ld, float* out) {
    Tape& t = *tape_of(h);
    const unsigned blocks = (unsigned)((rows + 255) / 256);
    if ((size_t)blocks * N > RS_FLOATS) return pg_fail(h, PG_EINVAL, "column-sum scratch too small");
    hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, s, d, rows, N, ld, t.rs_part);
    PG_LAUNCH_CHECK(h, "colsum");
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, t.rs_part, (int)blocks, 1, N, out, (long long)N);
    PG_LAUNCH_CHECK(h, "column-sum reduction");
    return PG_OK;
}
int relu_mask(pg_handle* h, hipStream_t s, float* d, const float* hh, long long count) {
    const unsigned blocks = (unsigned)std::min<long long>((count + 255) / 256, 8192);
    hipLaunchKernelGGL(relu_mask_kernel, dim3(blocks), dim3(256), 0, s, d, hh, count);
    PG_LAUNCH_CHECK(h, "relu_mask");
    return PG_OK;
}

#define PG_TRY(call) do { const int rc_ = (call); if (rc_) return rc_; } while (0)

// tensor i of a net in pg_load_weights order: 2l / 2l+1 = pts_linears.l.{weight,bias}; 16,17 alpha; 18,19 feature; 20,21 views; 22,23 rgb
int mlp_forward(pg_handle* h, hipStream_t s, const Pass& p, const pg_net_params& w, int fc) {
    const long long P = p.P;
    const int vk = CH_D + (fc ? FC_CH : 0), vcols = W + vk;
    PG_TRY(linear_fwd(h, s, P, W, CH_X, p.X, XW, w.w[0], CH_X, p.H[0], W, w.w[1], GEMM_RELU));
    for (int l = 1; l < DEPTH; ++l) {
        if (l == SKIP + 1) {        // h = cat([x, h]) in front of layer 5 (nerf.py:99-101)
            PG_TRY(linear_fwd(h, s, P, W, CH_X, p.X, XW, w.w[2 * l], CH_X + W, p.H[l], W, nullptr, 0));
            PG_TRY(linear_fwd(h, s, P, W, W, p.H[l - 1], W, w.w[2 * l] + CH_X, CH_X + W, p.H[l], W, w.w[2 * l + 1], GEMM_ACC | GEMM_RELU));
        } else {
            PG_TRY(linear_fwd(h, s, P, W, W, p.H[l - 1], W, w.w[2 * l], W, p.H[l], W, w.w[2 * l + 1], GEMM_RELU));
        }
    }
    const float* h7 = p.H[DEPTH - 1];
    PG_TRY(linear_fwd(h, s, P, 1, W, h7, W, w.w[16], W, p.raw + 3, 4, w.w[17], 0));
    PG_TRY(linear_fwd(h, s, P, W, W, h7, W, w.w[18], W, p.F, W, w.w[19], 0));
    PG_TRY(linear_fwd(h, s, P, VW, W, p.F, W, w.w[20], vcols, p.G, VW, nullptr, 0));
    PG_TRY(linear_fwd(h, s, P, VW, vk, p.X + CH_X, XW, w.w[20] + W, vcols, p.G, VW, w.w[21], GEMM_ACC | GEMM_RELU));
    PG_TRY(linear_fwd(h, s, P, 3, VW, p.G, VW, w.w[22], VW, p.raw, 4, w.w[23], 0));
    return PG_OK;
}

int mlp_backward(pg_handle* h, hipStream_t s, Tape& t, const Pass& p, const pg_net_params& w, const pg_net_grads& g) {
    const long long P = p.P;
    const int fc = t.fc, vk = CH_D + (fc ? FC_CH : 0), vcols = W + vk;
    const size_t sizes[24] = {(size_t)W * CH_X, W, (size_t)W * W, W, (size_t)W * W, W, (size_t)W * W, W, (size_t)W * W, W, (size_t)W * (CH_X + W), W,
                              (size_t)W * W, W, (size_t)W * W, W, W, 1, (size_t)W * W, W, (size_t)VW * vcols, VW, 3 * VW, 3};
    for (int i = 0; i < 24; ++i) {
        if (!g.w[i]) return pg_fail(h, PG_EINVAL, "pg_train_backward: gradient tensor %d is null", i);
        PG_HIP(h, hipMemsetAsync(g.w[i], 0, sizes[i] * sizeof(float), s));
    }
    const float* d_raw = t.d_raw;
    float* dG = t.dG;
    // rgb_linear: raw[:, :3] = G Wr^T + br
    PG_TRY(linear_bwd_x(h, s, P, 3, VW, d_raw, 4, w.w[22], VW, dG, VW, 0));
    PG_TRY(linear_bwd_w(h, s, P, 3, VW, d_raw, 4, p.G, VW, g.w[22], VW));
    PG_TRY(colsum(h, s, d_raw, P, 3, 4, g.w[23]));
    PG_TRY(relu_mask(h, s, dG, p.G, P * VW));
    // views_linears.0 on [feature | view embedding (| frame code)]
    PG_TRY(linear_bwd_w(h, s, P, VW, W, dG, VW, p.F, W, g.w[20], vcols, g.w[21]));
    PG_TRY(linear_bwd_w(h, s, P, VW, vk, dG, VW, p.X + CH_X, XW, g.w[20] + W, vcols));
    float* dF = t.tmpA;
    PG_TRY(linear_bwd_x(h, s, P, VW, W, dG, VW, w.w[20], vcols, dF, W, 0));
    if (fc && g.codes) {
        PG_HIP(h, hipMemsetAsync(g.codes, 0, (size_t)w.n_codes * FC_CH * sizeof(float), s));
        PG_TRY(linear_bwd_x(h, s, P, VW, FC_CH, dG, VW, w.w[20] + W + CH_D, vcols, t.dC, FC_CH, 0));
        hipLaunchKernelGGL(code_ray_sum_kernel, dim3((unsigned)((t.n * FC_CH + 255) / 256)), dim3(256), 0, s, t.dC, (long long)t.n, p.S, t.ray_g);
        PG_LAUNCH_CHECK(h, "code_ray_sum");
        hipLaunchKernelGGL(code_gather_kernel, dim3((unsigned)w.n_codes), dim3(64), 0, s, t.ray_g, (long long)t.n, t.cams, w.n_codes, g.codes);
        PG_LAUNCH_CHECK(h, "code_gather");
    }
    // feature_linear and alpha_linear on the trunk output
    const float* h7 = p.H[DEPTH - 1];
    PG_TRY(linear_bwd_w(h, s, P, W, W, dF, W, h7, W, g.w[18], W, g.w[19]));
    float* dH = t.tmpB;             // dH7 = (alpha's part + feature's part) * [H7 > 0]: the mask rides on the second GEMM
    PG_TRY(linear_bwd_x(h, s, P, 1, W, d_raw + 3, 4, w.w[16], W, dH, W, 0));
    PG_TRY(linear_bwd_x(h, s, P, W, W, dF, W, w.w[18], W, dH, W, GEMM_ACC, h7));
    PG_TRY(linear_bwd_w(h, s, P, 1, W, d_raw + 3, 4, h7, W, g.w[16], W));
    PG_TRY(colsum(h, s, d_raw + 3, P, 1, 4, g.w[17]));
    // the trunk, back to front: dZ_l = dH_l * [H_l > 0]
    float* other = t.tmpA;
    // (dH arrives masked: the GEMM that produced it zeroed it where H_l <= 0; the bias gradient rides on a weight-gradient GEMM)
    for (int l = DEPTH - 1; l >= 0; --l) {
        if (l == 0) {
            PG_TRY(linear_bwd_w(h, s, P, W, CH_X, dH, W, p.X, XW, g.w[0], CH_X, g.w[1]));
        } else if (l == SKIP + 1) {
            PG_TRY(linear_bwd_w(h, s, P, W, CH_X, dH, W, p.X, XW, g.w[2 * l], CH_X + W, g.w[2 * l + 1]));
            PG_TRY(linear_bwd_w(h, s, P, W, W, dH, W, p.H[l - 1], W, g.w[2 * l] + CH_X, CH_X + W));
            PG_TRY(linear_bwd_x(h, s, P, W, W, dH, W, w.w[2 * l] + CH_X, CH_X + W, other, W, 0, p.H[l - 1]));
            std::swap(dH, other);
        } else {
            PG_TRY(linear_bwd_w(h, s, P, W, W, dH, W, p.H[l - 1], W, g.w[2 * l], W, g.w[2 * l + 1]));
            PG_TRY(linear_bwd_x(h, s, P, W, W, dH, W, w.w[2 * l], W, other, W, 0, p.H[l - 1]));
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
    auto pass_bytes = [&](long long P) {
        return al((size_t)P * XW * 4) + (DEPTH + 1) * al((size_t)P * W * 4) + al((size_t)P * VW * 4) + al((size_t)P * 16) + 2 * al((size_t)P * 4) + al((size_t)P * 12);
    };
    need += pass_bytes(Pc) + (N > 0 ? pass_bytes(Pf) : 0);
    need += 2 * al((size_t)Pm * W * 4) + al((size_t)Pm * VW * 4) + al((size_t)Pm * FC_CH * 4) + al((size_t)Pm * 16);
    need += al(PART_FLOATS * 4) + al(RS_FLOATS * 4) + al((size_t)n * FC_CH * 4);
    if (need > t.bytes) {
        if (t.buf) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(t.buf)); t.buf = nullptr; t.bytes = 0; }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&t.buf), need);
        if (e != hipSuccess) return pg_fail(h, PG_ENOMEM, "training tape of %zu bytes failed: %s", need, hipGetErrorString(e));
        t.bytes = need;
    }
    uint8_t* q = t.buf;
    auto take = [&](size_t b) { float* r = reinterpret_cast<float*>(q); q += al(b); return r; };
    t.n = n; t.S = S; t.N = N; t.fc = fc; t.has_fine = N > 0;
    t.bf16 = h->cfg.precision == PG_PREC_BF16;
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
        p.X = take((size_t)p.P * XW * 4);
        for (int l = 0; l < DEPTH; ++l) p.H[l] = take((size_t)p.P * W * 4);
        p.F = take((size_t)p.P * W * 4);
        p.G = take((size_t)p.P * VW * 4);
        p.raw = take((size_t)p.P * 16);
        p.z = take((size_t)p.P * 4);
        p.noise = take((size_t)p.P * 4);
        p.pn = take((size_t)p.P * 12);
    }
    t.tmpA = take((size_t)Pm * W * 4); t.tmpB = take((size_t)Pm * W * 4);
    t.dG = take((size_t)Pm * VW * 4); t.dC = take((size_t)Pm * FC_CH * 4); t.d_raw = take((size_t)Pm * 16);
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
        hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)((p.P * J + 255) / 256)), dim3(256), 0, s, t.rays, p.z, rnoise ? p.pn : nullptr, skts,
                           (long long)pose_stride, t.cams, codes, n_codes, fc, h->d_cut, h->tau[0], h->tau[1], p.P, p.S, p.X);
        return hipGetLastError();
    };
    if (embed(pc, codes_dev[0], coarse->n_codes) != hipSuccess) return pg_fail(h, PG_EHIP, "embedding kernel launch failed");
    PG_TRY(mlp_forward(h, s, pc, *coarse, fc));
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
        PG_TRY(mlp_forward(h, s, pf, *fine, fc));
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
        return mlp_backward(h, s, t, p, t.params[k], g);
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
