// pg_rayrec.hip -- per-ray records of the factorised view layer: ray_records_kernel for the 16-bit path (pg_eval16r.hip),
// ray_records_c_kernel for the compensated-fp16 path (pg_evalc.hip, record variant) -- pg_layout.h "per-ray records".
//
// The 648-wide view-direction input of a point is w_j(point) * T[ray][j][k] (core/encoders.py:25-37, 172-193;
// core/cutoff_embedder.py:111-174 with dist_inputs=True): 27 values per joint that depend on the RAY only, times
// the point's cutoff weight of that joint.  So the view layer's direction part is
//     W_vd xd = sum_j w_j(point) Y[ray][j],   Y[ray][j][o] = sum_k W_vd[o, (j, k)] T[ray][j][k],
// and Y -- like the bone-local ray (a_j, b_j) = (R_j o + t_j, R_j d) with q_j = a_j + z b_j (encoders.py:8-23) --
// is a function of the ray.  This kernel computes both for every ray of a launch, 32 rays per MFMA tile (the
// fused kernel used to do it per workgroup pass for the <= 5 rays the pass touches: 5 of the 32 MFMA rows used,
// the Y-stage weights re-read from L2 every pass), and writes them in the LDS image pg_eval16r.hip fetches with
// LDS-DMA.  HBM-bound: 8 KiB + 768 B written per ray.
//
// Workgroup = 8 waves; wave w owns out tile w&3 (32 of the 128 view channels) and the joints of half w>>2, and
// keeps its Y-stage weights (pack_vy: 24 or 26 B fragments) in registers for the whole launch.
#include "pg_device.h"

namespace pgd {

template <typename V> struct OpR;
template <> struct OpR<bf16x8> {
    using E = __bf16;
    static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct OpR<f16x8> {
    using E = _Float16;
    static __device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// min over z in [z_first, z_last] of |a + z b|^2: the squared distance of the ray's sampled segment from the joint
// (bone-local: q = a + z b, encoders.py:8-23), kept in the pad of the record's a row.  The fused kernel leaves a limb's
// weights out of a pass when this is beyond cutoff range for every ray of the pass (pg_eval16r.hip pass_far_mask).
__device__ __forceinline__ float segment_dist2(const RecArgs& a, long long ray, float ax, float ay, float az, float bx, float by, float bz) {
    if (!a.z) return 0.0f;                  // (no depths given: "near")
    const float z0 = a.z[ray * a.S], z1 = a.z[ray * a.S + a.S - 1];
    const float bb = bx * bx + by * by + bz * bz, ab = ax * bx + ay * by + az * bz;
    float zs = bb > 0.0f ? -ab / bb : z0;
    zs = fminf(fmaxf(zs, fminf(z0, z1)), fmaxf(z0, z1));
    const float qx = fmaf(zs, bx, ax), qy = fmaf(zs, by, ay), qz = fmaf(zs, bz, az);
    const float d2 = qx * qx + qy * qy + qz * qz;
    return d2 == d2 ? d2 : 0.0f;            // NaN depths (never expected): "near"
}

constexpr int REC_THREADS = 512;
constexpr int REC_NJT = J + 1;                  // joints per ray in the T table (24 + the frame-code pseudo joint)
constexpr int REC_TSTRIDE = REC_NJT * TK * 2 + 16;      // bytes per ray (padded: rays 4 apart would share LDS banks)

template <typename V, bool FC>
__global__ __launch_bounds__(REC_THREADS, 2) void ray_records_kernel(const RecArgs a) {
    using E = typename OpR<V>::E;
    constexpr int NJ = J + (FC ? 1 : 0);
    constexpr int NE = JH + (FC ? 1 : 0);       // joints a wave handles: 12 of its half (+ the frame code)
    constexpr int NU = 2 * NE;
    __shared__ __attribute__((aligned(16))) uint8_t t16[REC_TILE_RAYS * REC_TSTRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = wave & 3, hw = wave >> 2, hl = lane >> 5, col = lane & 31;

    uint4 yw[NU];
    {
        const uint4* p = reinterpret_cast<const uint4*>(a.wy) + ((size_t)wave * NU) * 64 + lane;
#pragma unroll
        for (int n = 0; n < NU; ++n) yw[n] = p[n * 64];
    }
    const int n_tiles = (a.n_rays + REC_TILE_RAYS - 1) / REC_TILE_RAYS;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int r0 = tile * REC_TILE_RAYS;
        __syncthreads();                        // the previous tile's table has been read
        // ---- phase 1: one thread per (ray, joint): (a, b) to HBM, the 27 view values T to LDS as MFMA operands ----
        // (records are laid out by SLOT of the 16x16x32 kernel: slot s holds joint slot16_joint(s), pg_layout.h)
        for (int idx = tid; idx < REC_TILE_RAYS * NJ; idx += REC_THREADS) {
            const int rr = idx / NJ, j = idx - rr * NJ;
            const bool live = r0 + rr < a.n_rays;
            const long long ray = live ? r0 + rr : a.n_rays - 1;
            float tv[TK];
#pragma unroll
            for (int k = 0; k < TK; ++k) tv[k] = 0.0f;
            if (j < J) {
                const float4* sk = reinterpret_cast<const float4*>(a.skts + ray * a.pose_stride + slot_joint_dev(j) * 16);
                const float4 ra = sk[0], rb = sk[1], rc = sk[2];
                const float* ry = a.rays + ray * 11;
                const float ox = ry[0], oy = ry[1], oz = ry[2], dx = ry[3], dy = ry[4], dz = ry[5];
                float e[3];
                e[0] = fmaf(ra.z, dz, fmaf(ra.y, dy, ra.x * dx));
                e[1] = fmaf(rb.z, dz, fmaf(rb.y, dy, rb.x * dx));
                e[2] = fmaf(rc.z, dz, fmaf(rc.y, dy, rc.x * dx));
                if (live) {
                    float4* ab = reinterpret_cast<float4*>(a.rec_ab + ray * (REC_AB_BYTES / 4) + j * 8);
                    const float ax = fmaf(ra.z, oz, fmaf(ra.y, oy, fmaf(ra.x, ox, ra.w))),
                                ay = fmaf(rb.z, oz, fmaf(rb.y, oy, fmaf(rb.x, ox, rb.w))),
                                az = fmaf(rc.z, oz, fmaf(rc.y, oy, fmaf(rc.x, ox, rc.w)));
                    ab[0] = make_float4(ax, ay, az, segment_dist2(a, ray, ax, ay, az, e[0], e[1], e[2]));
                    ab[1] = make_float4(e[0], e[1], e[2], 0.0f);
                }
                // e = normalize(R_j d); rows (e, sin e, cos e, sin 2e, .., cos 8e) per component c: k = c * 9 + row
                const float inv = __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), 1e-12f));
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float ev = e[c] * inv;
                    const float rev = ev * 0.15915494309189535f;
                    float sn = __builtin_amdgcn_sinf(rev), co = __builtin_amdgcn_cosf(rev);
                    tv[c * ROWS_D] = ev;
#pragma unroll
                    for (int f = 0; f < LD; ++f) {
                        tv[c * ROWS_D + 1 + 2 * f] = sn;
                        tv[c * ROWS_D + 2 + 2 * f] = co;
                        const float s2 = 2.0f * sn * co;
                        co = (co - sn) * (co + sn);
                        sn = s2;
                    }
                }
            } else {
                const float cam = a.cams ? a.cams[ray] : -1.0f;
                const int ci = cam < 0.0f ? a.n_codes : min((int)cam, a.n_codes - 1);
#pragma unroll
                for (int k = 0; k < FC_CH; ++k) tv[k] = a.codes[ci * FC_CH + k];
            }
            V* dst = reinterpret_cast<V*>(t16 + rr * REC_TSTRIDE + j * (TK * 2));
#pragma unroll
            for (int q = 0; q < TK / 8; ++q) {
                V v;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (E)tv[8 * q + k];
                dst[q] = v;
            }
        }
        __syncthreads();
        // ---- phase 2: Y = T W, rays as MFMA rows; C has the out channel on the lane and 16 rays in registers.
        // Slot e of the wave (slot 12 hw + e of the T table; e = 12: the frame code) is slot e % 6 of lane group 2 hw + e / 6 of
        // the second stage's A fragment (vy16_slot_joint); the code sits in slot 6 of group 2 hw (zero weights, hence
        // a zero, for hw = 1).  One 16-byte store per ray and lane group. ----
        const uint8_t* trow = t16 + col * REC_TSTRIDE + hl * 16;
#pragma clang loop unroll(full)
        for (int half = 0; half < 2; ++half) {
            unsigned pk[16][4];
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) pk[r][q] = 0u;
#pragma clang loop unroll(full)
            for (int s = 0; s < JG + 1; ++s) {
                if (s == JG && !(FC && half == 0)) continue;
                const int e = s < JG ? JG * half + s : JH;                    // joint of the wave, slot s
                const uint8_t* tj = trow + (e < JH ? JH * hw + e : JC) * (TK * 2);
                const V a0 = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(tj));
                const V a1 = __builtin_bit_cast(V, *reinterpret_cast<const uint4*>(tj + 32));
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                acc = OpR<V>::mfma(a0, __builtin_bit_cast(V, yw[2 * e]), acc);
                acc = OpR<V>::mfma(a1, __builtin_bit_cast(V, yw[2 * e + 1]), acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const E v = (E)acc[r];
                    unsigned short u;
                    __builtin_memcpy(&u, &v, 2);
                    pk[r][s / 2] |= (unsigned)u << (16 * (s & 1));
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ray = r0 + rho(r, hl);
                if (ray < a.n_rays)
                    *reinterpret_cast<uint4*>(a.rec_y + (size_t)ray * REC_Y_BYTES + (2 * t + (col >> 4)) * 1024 +
                                              ((2 * hw + half) * 16 + (col & 15)) * 16) = make_uint4(pk[r][0], pk[r][1], pk[r][2], pk[r][3]);
            }
        }
    }
}

// ---- records of the compensated-fp16 kernel (pg_layout.h "per-ray records of the compensated-fp16 kernel") ----
// Y in fp32 (its error must stay below the 2^-17 of the compensated products that consume it, so no 16-bit MFMA
// here): v_mfma_f32_32x32x2_f32 with 32 rays as rows (A = the ray's view table T from LDS, one float per lane),
// the out channels as columns (B = the fp32 weights [joint][k][out] straight from L2, one coalesced float per lane)
// and K = 28 = 14 steps per joint.  Wave w owns out tile w and runs the two lane halves of the second stage's A
// operand (vyc_slot_joint) one after the other; a k-unit's eight joints at a time, then per ray of the C layout the
// eight values are split like a weight, y = Y / S -> ((S-1) f16(y), f16(y1 + S (y - y1))), one 16-byte fragment per plane.
// The view table uses accurate sincosf for the base angle like the direct form of pg_evalc.hip (per ray, its cost
// is nothing).
constexpr int RECC_THREADS = 256;            // one wave per SIMD: the 128 accumulators of a k-unit need the 512-register budget
constexpr int RECC_RAYS = 32;
constexpr int RECC_TSTRIDE = (J + 1) * VYC_K + 1;        // floats per ray in LDS (701, odd: the 32 rays of an A read hit 32 banks)
constexpr int RECC_LDS = RECC_RAYS * RECC_TSTRIDE * 4;

// One k-unit of the second stage's A operand for this wave's out tile and lane half: the joints of its NJ <= 8 slots
// (slot e < NJ: joint jbase + e, or JC for the LAST slot when `code`), all accumulators live, so that a ray's eight
// values leave as ONE 16-byte fragment per plane (32 lanes x 16 B contiguous per store; half fragments of 8 bytes
// ran the 4.3 GB of a 512 x 512 launch at 1.8 TB/s).
template <int NJ, bool CODE>
__device__ __forceinline__ void recc_unit(const RecArgs& a, const float* wy, const float* trow, int ray0, int hl, int jbase, uint8_t* dst0) {
    f32x16 acc[NJ];
#pragma unroll
    for (int e = 0; e < NJ; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[e][r] = 0.0f;
#pragma unroll
    for (int kk = 0; kk < VYC_K / 2; ++kk) {
        if (kk % 2 == 0) asm volatile("" ::: "memory");         // operands of two steps in flight, not of all fourteen
#pragma unroll
        for (int e = 0; e < NJ; ++e) {
            const int j = (CODE && e == NJ - 1) ? JC : jbase + e;
            const float av = trow[j * VYC_K + 2 * kk];
            const float bv = wy[(j * VYC_K + 2 * kk) * VW];
            acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[e], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ray = ray0 + rho(r, hl);
        unsigned p0[4], p1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned short b0[2], b1[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = 2 * q + i;
                float v = e < NJ ? acc[e < NJ ? e : 0][r] * (1.0f / (float)COMP_S) : 0.0f;
                asm volatile("" : "+v"(v));     // one rounded value for both halves (no fused convert of the product)
                const _Float16 y1 = (_Float16)v;
                const float y1f = (float)y1;
                const _Float16 y2 = (_Float16)fmaf((float)COMP_S, v - y1f, y1f);
                const _Float16 ym = (_Float16)((float)(COMP_S - 1) * y1f);
                __builtin_memcpy(&b0[i], &ym, 2);
                __builtin_memcpy(&b1[i], &y2, 2);
            }
            p0[q] = (unsigned)b0[0] | ((unsigned)b0[1] << 16);
            p1[q] = (unsigned)b1[0] | ((unsigned)b1[1] << 16);
        }
        if (ray < a.n_rays) {
            uint8_t* dst = dst0 + (size_t)rho(r, hl) * RECC_Y_BYTES;
            *reinterpret_cast<uint4*>(dst) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
            *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        }
    }
}

template <bool FC>
__global__ __launch_bounds__(RECC_THREADS, 1) void ray_records_c_kernel(const RecArgs a) {
    constexpr int NJ = J + (FC ? 1 : 0);
    extern __shared__ __attribute__((aligned(16))) float tl[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = wave, hl = lane >> 5, col = lane & 31;
    const float* wy = reinterpret_cast<const float*>(a.wy) + hl * VW + 32 * t + col;       // row k = 2 kk + hl, column 32 t + col
    const float* trow = tl + col * RECC_TSTRIDE + hl;                                       // ray `col` of the tile, value 2 kk + hl
    const int n_tiles = (a.n_rays + RECC_RAYS - 1) / RECC_RAYS;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int r0 = tile * RECC_RAYS;
        __syncthreads();                        // the previous tile's table has been read
        for (int idx = tid; idx < RECC_RAYS * NJ; idx += RECC_THREADS) {
            const int rr = idx / NJ, j = idx - rr * NJ;
            const bool live = r0 + rr < a.n_rays;
            const long long ray = live ? r0 + rr : a.n_rays - 1;
            float* tv = tl + rr * RECC_TSTRIDE + j * VYC_K;
            if (j < J) {        // (j = joint SLOT of the compensated kernel's record variant: joint slotc_joint(j), pg_layout.h)
                const float4* sk = reinterpret_cast<const float4*>(a.skts + ray * a.pose_stride + slotc_joint_dev(j) * 16);
                const float4 ra = sk[0], rb = sk[1], rc = sk[2];
                const float* ry = a.rays + ray * 11;
                const float ox = ry[0], oy = ry[1], oz = ry[2], dx = ry[3], dy = ry[4], dz = ry[5];
                float e[3];
                e[0] = fmaf(ra.z, dz, fmaf(ra.y, dy, ra.x * dx));
                e[1] = fmaf(rb.z, dz, fmaf(rb.y, dy, rb.x * dx));
                e[2] = fmaf(rc.z, dz, fmaf(rc.y, dy, rc.x * dx));
                if (live) {
                    float4* ab = reinterpret_cast<float4*>(a.rec_ab + ray * (REC_AB_BYTES / 4) + j * 8);
                    const float ax = fmaf(ra.z, oz, fmaf(ra.y, oy, fmaf(ra.x, ox, ra.w))),
                                ay = fmaf(rb.z, oz, fmaf(rb.y, oy, fmaf(rb.x, ox, rb.w))),
                                az = fmaf(rc.z, oz, fmaf(rc.y, oy, fmaf(rc.x, ox, rc.w)));
                    ab[0] = make_float4(ax, ay, az, segment_dist2(a, ray, ax, ay, az, e[0], e[1], e[2]));
                    ab[1] = make_float4(e[0], e[1], e[2], 0.0f);
                }
                const float den = fmaxf(sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), 1e-12f);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float ev = e[c] / den;
                    float sn, co;
                    sincosf(ev, &sn, &co);
                    tv[c * ROWS_D] = ev;
#pragma unroll
                    for (int f = 0; f < LD; ++f) {
                        tv[c * ROWS_D + 1 + 2 * f] = sn;
                        tv[c * ROWS_D + 2 + 2 * f] = co;
                        const float s2 = 2.0f * sn * co;
                        co = (co - sn) * (co + sn);
                        sn = s2;
                    }
                }
                tv[27] = 0.0f;
            } else {
                const float cam = a.cams ? a.cams[ray] : -1.0f;
                const int ci = cam < 0.0f ? a.n_codes : min((int)cam, a.n_codes - 1);
#pragma unroll
                for (int k = 0; k < VYC_K; ++k) tv[k] = k < FC_CH ? a.codes[ci * FC_CH + k] : 0.0f;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int kh = 0; kh < 2; ++kh) {
            uint8_t* d0 = a.rec_y + (size_t)r0 * RECC_Y_BYTES + (t * 4) * 1024 + (kh * 32 + col) * 16;
            recc_unit<8, false>(a, wy, trow, r0, hl, JH * kh, d0);                            // k-unit 0: joints 12 kh + 0..7
            if (FC && kh == 0) recc_unit<JH - 8 + 1, true>(a, wy, trow, r0, hl, JH * kh + 8, d0 + 2048);   // k-unit 1: joints 12 kh + 8..11
            else recc_unit<JH - 8, false>(a, wy, trow, r0, hl, JH * kh + 8, d0 + 2048);        // (+ the frame code in half 0), zeros
        }
    }
}

template <bool FC>
static hipError_t launch_records_c(const RecArgs& a, int n_cu, hipStream_t stream) {
    auto k = ray_records_c_kernel<FC>;
    static std::atomic<unsigned long long> attr_done{0};       // per device (pg_device.h)
    const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(k), RECC_LDS, attr_done);
    if (e != hipSuccess) return e;
    const int n_tiles = (a.n_rays + RECC_RAYS - 1) / RECC_RAYS;
    hipLaunchKernelGGL(k, dim3(n_tiles < n_cu ? n_tiles : n_cu), dim3(RECC_THREADS), RECC_LDS, stream, a);
    return hipGetLastError();
}

template <typename V, bool FC>
static hipError_t launch_records(const RecArgs& a, int n_cu, hipStream_t stream) {
    const int n_tiles = (a.n_rays + REC_TILE_RAYS - 1) / REC_TILE_RAYS;
    const int grid = n_tiles < n_cu ? n_tiles : n_cu;
    hipLaunchKernelGGL((ray_records_kernel<V, FC>), dim3(grid), dim3(REC_THREADS), 0, stream, a);
    return hipGetLastError();
}

}  // namespace pgd

extern "C" int pg_launch_ray_records_c(const pgd::RecArgs* a, int framecode, int n_cu, void* stream) {
    using namespace pgd;
    if (a->n_rays <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return (int)(framecode ? launch_records_c<true>(*a, n_cu, s) : launch_records_c<false>(*a, n_cu, s));
}

extern "C" int pg_launch_ray_records(const pgd::RecArgs* a, int fp16, int framecode, int n_cu, void* stream) {
    using namespace pgd;
    if (a->n_rays <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e;
    if (fp16) e = framecode ? launch_records<f16x8, true>(*a, n_cu, s) : launch_records<f16x8, false>(*a, n_cu, s);
    else      e = framecode ? launch_records<bf16x8, true>(*a, n_cu, s) : launch_records<bf16x8, false>(*a, n_cu, s);
    return (int)e;
}
