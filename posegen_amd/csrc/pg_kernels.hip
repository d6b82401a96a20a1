// pg_kernels.hip -- the non-MLP kernels of the render path:
//   sample_coarse_kernel : ray/cylinder near-far + per-chunk nanmean patch + coarse depths
//   composite_kernel     : wave-per-ray prefix-product alpha compositing and the
//                          deterministic inverse-CDF importance samples + sorted merge
// All fp32.  These are HBM-bound byte movers: one coalesced pass over their inputs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

namespace pgk {

// ------------------------------------------------------------------------------------
// get_near_far_in_cylinder + sample_from_lineseg
// (reference core/utils/ray_utils.py:292-344 and :204-251, eval mode: perturb = 0).
// One workgroup per `chunk` consecutive rays, because rays that miss the cylinder take
// the nanmean of the hits OF THEIR CHUNK (the reference calls this once per
// batchify_rays slice, core/trainer.py:64-81).
// ------------------------------------------------------------------------------------
constexpr int SC_THREADS = 256;

__device__ __forceinline__ void ray_near_far(const float* rb, const float* cyl, float& nn, float& ff,
                                             bool& q_nan) {
    const float ox = rb[0], oz = rb[2], dx = rb[3], dz = rb[5];
    const float near0 = rb[6], far0 = rb[7];
    // r_near = (o + d*near)[x,z], r_far = (o + d*far)[x,z]   (mul, then add)
    const float nx = __fadd_rn(ox, __fmul_rn(dx, near0)), nz = __fadd_rn(oz, __fmul_rn(dz, near0));
    const float fx = __fadd_rn(ox, __fmul_rn(dx, far0)), fz = __fadd_rn(oz, __fmul_rn(dz, far0));
    const float radius = cyl[2];
    const float cx = __fsub_rn(cyl[0], nx), cz = __fsub_rn(cyl[1], nz);     // near -> centre
    const float sx = __fsub_rn(fx, nx), sz = __fsub_rn(fz, nz);             // near -> far
    const float seg = __fsqrt_rn(__fadd_rn(__fmul_rn(sx, sx), __fmul_rn(sz, sz)));
    const float scale = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dz, dz)));
    const float cross = __fsub_rn(__fmul_rn(cx, sz), __fmul_rn(cz, sx));
    const float dist = __fdiv_rn(fabsf(cross), seg);
    const float q2 = __fsub_rn(__fmul_rn(radius, radius), __fmul_rn(dist, dist));
    const float Q = q2 < 0.0f ? __builtin_nanf("") : __fsqrt_rn(q2);       // pow(0.5) of <0 is NaN
    const float K = __fdiv_rn(__fadd_rn(__fmul_rn(cx, sx), __fmul_rn(cz, sz)), seg);
    const float inside = (Q < K) ? 1.0f : 0.0f;
    nn = __fadd_rn(near0, __fdiv_rn(__fmul_rn(inside, __fsub_rn(K, Q)), scale));
    ff = __fadd_rn(near0, __fdiv_rn(__fadd_rn(K, Q), scale));
    q_nan = isnan(Q);
}

__global__ __launch_bounds__(SC_THREADS) void sample_coarse_kernel(
        const float* __restrict__ rays, const float* __restrict__ cyls, long long cyl_stride,
        long long n, int chunk, int S, int lindisp,
        float* __restrict__ near_far, float* __restrict__ z, const float* __restrict__ t_rand) {
    __shared__ double red[3][SC_THREADS / 64][2];
    __shared__ float fix[2];
    __shared__ int any_nan;
    const long long c0 = (long long)blockIdx.x * chunk;
    const long long c1 = min(c0 + chunk, n);
    const int tid = threadIdx.x;
    if (tid == 0) any_nan = 0;
    __syncthreads();

    // pass 1: near/far, sums of the non-NaN entries (np.nanmean over the chunk)
    double s_near = 0.0, s_far = 0.0, c_near = 0.0, c_far = 0.0;
    bool saw_nan = false;
    for (long long r = c0 + tid; r < c1; r += SC_THREADS) {
        float nn, ff;
        bool qn;
        ray_near_far(rays + r * 11, cyls + r * cyl_stride, nn, ff, qn);
        near_far[r * 2 + 0] = nn;
        near_far[r * 2 + 1] = ff;
        if (!isnan(nn)) { s_near += nn; c_near += 1.0; } else saw_nan = true;
        if (!isnan(ff)) { s_far += ff; c_far += 1.0; }
    }
    if (saw_nan) any_nan = 1;
    // block reduction (wave shuffle, then across the 4 waves)
    double v[6] = {s_near, c_near, s_far, c_far, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) {
        red[0][wave][0] = v[0]; red[0][wave][1] = v[1];
        red[1][wave][0] = v[2]; red[1][wave][1] = v[3];
    }
    __syncthreads();
    if (tid == 0) {
        double sn = 0, cn = 0, sf = 0, cf = 0;
        for (int w = 0; w < SC_THREADS / 64; ++w) {
            sn += red[0][w][0]; cn += red[0][w][1];
            sf += red[1][w][0]; cf += red[1][w][1];
        }
        fix[0] = cn > 0 ? (float)(sn / cn) : __builtin_nanf("");
        fix[1] = cf > 0 ? (float)(sf / cf) : __builtin_nanf("");
    }
    __syncthreads();
    const bool patch = any_nan != 0;            // `if torch.isnan(new_near).any()`
    const float m_near = fix[0], m_far = fix[1];

    // pass 2: patch misses (rows where Q is NaN) and emit the S depths of every ray
    const float step = 1.0f / (float)(S - 1);
    for (long long r = c0 + tid; r < c1; r += SC_THREADS) {
        float nn, ff;
        bool qn;
        ray_near_far(rays + r * 11, cyls + r * cyl_stride, nn, ff, qn);
        if (patch && qn) {
            nn = isnan(m_near) ? rays[r * 11 + 6] : m_near;
            ff = isnan(m_far) ? rays[r * 11 + 7] : m_far;
            near_far[r * 2 + 0] = nn;
            near_far[r * 2 + 1] = ff;
        }
        float* zr = z + r * S;
        for (int s = 0; s < S; ++s) {
            // torch.linspace(0,1,S): start + i*step below the midpoint, end - (S-1-i)*step above
            const float t = s < S / 2 ? __fmul_rn(step, (float)s) : __fsub_rn(1.0f, __fmul_rn(step, (float)(S - 1 - s)));
            float zv;
            if (!lindisp) {
                zv = __fadd_rn(__fmul_rn(nn, __fsub_rn(1.0f, t)), __fmul_rn(ff, t));
            } else {
                const float a = __fmul_rn(__fdiv_rn(1.0f, nn), __fsub_rn(1.0f, t));
                const float b = __fmul_rn(__fdiv_rn(1.0f, ff), t);
                zv = __fdiv_rn(1.0f, __fadd_rn(a, b));
            }
            zr[s] = zv;
        }
        if (t_rand) {
            // perturb > 0 (ray_utils.py:229-246): a stratified sample in [lower, upper] of every depth, the
            // uniform draws t_rand [n,S] supplied by the caller; in place, neighbours read before they change
            const float* tr = t_rand + r * S;
            float prev = zr[0], cur = zr[0];
            for (int s = 0; s < S; ++s) {
                const float nxt = s + 1 < S ? zr[s + 1] : cur;
                const float lower = s == 0 ? cur : __fmul_rn(0.5f, __fadd_rn(cur, prev));
                const float upper = s + 1 < S ? __fmul_rn(0.5f, __fadd_rn(nxt, cur)) : cur;
                zr[s] = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), tr[s]));
                prev = cur; cur = nxt;
            }
        }
    }
}

// The same in two launches for chunks of more than SC_SUB rays (a training batch is ONE chunk of N_rand rays, a direct
// forward call one chunk of the whole batch: a single workgroup per chunk then crawls): blocks of SC_SUB rays write
// their near/far and partial nanmean sums (sc_partial_kernel), then every block adds its chunk's partials in index
// order -- fp64, so the mean is the one the single-block form computes except for the order of a few double additions --
// and emits the depths with a wave per ray (coalesced rows of z).
constexpr int SC_SUB = 256;

__global__ __launch_bounds__(SC_THREADS) void sc_partial_kernel(
        const float* __restrict__ rays, const float* __restrict__ cyls, long long cyl_stride,
        long long n, int chunk, int nsub, float* __restrict__ near_far, double* __restrict__ part) {
    __shared__ double red[SC_THREADS / 64][4];
    __shared__ int any_nan;
    const long long group = blockIdx.x / nsub, sub = blockIdx.x % nsub;
    const long long c0 = group * chunk + sub * SC_SUB;
    const long long c1 = min(min(c0 + SC_SUB, (group + 1) * chunk), n);
    const int tid = threadIdx.x;
    if (tid == 0) any_nan = 0;
    __syncthreads();
    double v[4] = {0.0, 0.0, 0.0, 0.0};         // sum near, count near, sum far, count far
    bool saw_nan = false;
    for (long long r = c0 + tid; r < c1; r += SC_THREADS) {
        float nn, ff;
        bool qn;
        ray_near_far(rays + r * 11, cyls + r * cyl_stride, nn, ff, qn);
        near_far[r * 2 + 0] = nn;
        near_far[r * 2 + 1] = ff;
        if (!isnan(nn)) { v[0] += nn; v[1] += 1.0; } else saw_nan = true;
        if (!isnan(ff)) { v[2] += ff; v[3] += 1.0; }
    }
    if (saw_nan) any_nan = 1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) { red[wave][0] = v[0]; red[wave][1] = v[1]; red[wave][2] = v[2]; red[wave][3] = v[3]; }
    __syncthreads();
    if (tid == 0) {
        double o[4] = {0.0, 0.0, 0.0, 0.0};
        for (int w = 0; w < SC_THREADS / 64; ++w)
            for (int k = 0; k < 4; ++k) o[k] += red[w][k];
        double* dst = part + (long long)blockIdx.x * 5;
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3]; dst[4] = any_nan ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(SC_THREADS) void sc_sample_kernel(
        const float* __restrict__ rays, const float* __restrict__ cyls, long long cyl_stride,
        long long n, int chunk, int nsub, int S, int lindisp,
        float* __restrict__ near_far, float* __restrict__ z, const float* __restrict__ t_rand, const double* __restrict__ part) {
    __shared__ float fix[2];
    __shared__ int any_nan;
    const long long group = blockIdx.x / nsub, sub = blockIdx.x % nsub;
    const long long c0 = group * chunk + sub * SC_SUB;
    const long long c1 = min(min(c0 + SC_SUB, (group + 1) * chunk), n);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) {
        double sn = 0, cn = 0, sf = 0, cf = 0, an = 0;
        for (int k = 0; k < nsub; ++k) {
            const double* p = part + (group * nsub + k) * 5;
            sn += p[0]; cn += p[1]; sf += p[2]; cf += p[3]; an += p[4];
        }
        fix[0] = cn > 0 ? (float)(sn / cn) : __builtin_nanf("");
        fix[1] = cf > 0 ? (float)(sf / cf) : __builtin_nanf("");
        any_nan = an > 0.0;
    }
    __syncthreads();
    const bool patch = any_nan != 0;
    const float m_near = fix[0], m_far = fix[1];
    const float step = 1.0f / (float)(S - 1);
    // a wave takes 64 rays at a time: lane k computes ray k's near/far (and patch), then the wave writes the 64 rows of
    // depths one after the other, a lane per depth (coalesced 4 S-byte rows)
    for (long long base = c0 + wave * 64; base < c1; base += SC_THREADS) {
        const long long rl = base + lane;
        float nn = 1.0f, ff = 2.0f;
        if (rl < c1) {
            bool qn;
            ray_near_far(rays + rl * 11, cyls + rl * cyl_stride, nn, ff, qn);
            if (patch && qn) {
                nn = isnan(m_near) ? rays[rl * 11 + 6] : m_near;
                ff = isnan(m_far) ? rays[rl * 11 + 7] : m_far;
                near_far[rl * 2 + 0] = nn;
                near_far[rl * 2 + 1] = ff;
            }
        }
        const int cnt = (int)min((long long)64, c1 - base);
        for (int k = 0; k < cnt; ++k) {
            const long long r = base + k;
            const float nk = __shfl(nn, k), fk = __shfl(ff, k);
            auto depth = [&](int s) {
                const float t = s < S / 2 ? __fmul_rn(step, (float)s) : __fsub_rn(1.0f, __fmul_rn(step, (float)(S - 1 - s)));
                if (!lindisp) return __fadd_rn(__fmul_rn(nk, __fsub_rn(1.0f, t)), __fmul_rn(fk, t));
                const float a = __fmul_rn(__fdiv_rn(1.0f, nk), __fsub_rn(1.0f, t));
                const float b = __fmul_rn(__fdiv_rn(1.0f, fk), t);
                return __fdiv_rn(1.0f, __fadd_rn(a, b));
            };
            for (int s = lane; s < S; s += 64) {
                float zv = depth(s);
                if (t_rand) {   // stratified jitter (ray_utils.py:229-246): between the midpoints to the neighbours
                    const float cur = zv, prev = s > 0 ? depth(s - 1) : cur, nxt = s + 1 < S ? depth(s + 1) : cur;
                    const float lower = s == 0 ? cur : __fmul_rn(0.5f, __fadd_rn(cur, prev));
                    const float upper = s + 1 < S ? __fmul_rn(0.5f, __fadd_rn(nxt, cur)) : cur;
                    zv = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t_rand[r * S + s]));
                }
                z[r * S + s] = zv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// raw2outputs (reference core/networks/nerf.py:150-205, eval: no noise) and, when
// n_imp > 0, isample_from_lineseg / sample_pdf with det=True
// (core/utils/ray_utils.py:157-201, 255-289).  One wave per ray; sample s lives in lane
// s / E, E = ceil(S/64) consecutive samples per lane; transmittance is an exclusive
// prefix product (lane-local then a 6-step wave scan).
// ------------------------------------------------------------------------------------
constexpr int CP_WAVES = 4;
constexpr int CP_MAXE = 4;          // S <= 256
constexpr int CP_MAXS = 64 * CP_MAXE;
constexpr int CP_MAXI = 64;         // importance samples per ray

// LDS written by some lanes of a wave is read by other lanes of the SAME wave: the LDS
// unit serves one wave's accesses in order, so only compiler reordering must be fenced.
#define PG_WAVE_SYNC()                                               \
    do {                                                             \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       \
        __builtin_amdgcn_wave_barrier();                             \
    } while (0)

__device__ __forceinline__ float wave_excl_scan_mul(float x, int lane) {
    // inclusive Hillis-Steele product, then shift by one lane
    float incl = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(incl, off);
        if (lane >= off) incl *= t;
    }
    const float prev = __shfl_up(incl, 1);
    return lane == 0 ? 1.0f : prev;
}
__device__ __forceinline__ float wave_incl_scan_add(float x, int lane) {
    float incl = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    return incl;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// act_fn of raw2outputs (get_density_fn, core/raycasters.py:230-238): F.relu, or F.softplus(x - shift, beta=1) with
// torch's threshold of 20 (the linear branch above it)
__device__ __forceinline__ float density_act(float x, int act, float shift) {
    if (act == 0) return fmaxf(x, 0.0f);
    const float t = x - shift;
    return t > 20.0f ? t : log1pf(expf(t));
}

// one ray by one wave (sh_*: the wave's rows of the workgroup's LDS arrays)
__device__ __forceinline__ void composite_ray(
        const long long ray, const int lane, const int wave,
        float (*sh_w)[CP_MAXS], float (*sh_z)[CP_MAXS + CP_MAXI], float (*sh_cdf)[CP_MAXS],
        const float* __restrict__ rays, const float* __restrict__ z, const float4* __restrict__ raw,
        int S, float density_scale, float rgb_eps, int act, float act_shift,
        float* __restrict__ rgb_out, float* __restrict__ disp_out, float* __restrict__ acc_out,
        float* __restrict__ alpha_out, float* __restrict__ w_out,
        int n_imp, float* __restrict__ z_fine, const float* __restrict__ noise, const float* __restrict__ u_rand,
        int* __restrict__ order) {
    const int E = (S + 63) >> 6;
    const float* rb = rays + ray * 11;
    const float dnorm = sqrtf(rb[3] * rb[3] + rb[4] * rb[4] + rb[5] * rb[5]);
    const float* zr = z + ray * S;
    const float4* rr = raw + ray * S;

    float a_[CP_MAXE], z_[CP_MAXE], cr[CP_MAXE], cg[CP_MAXE], cb[CP_MAXE];
    float lane_prod = 1.0f;
#pragma unroll
    for (int e = 0; e < CP_MAXE; ++e) {
        const int s = lane * E + e;
        a_[e] = 0.0f; z_[e] = 0.0f; cr[e] = cg[e] = cb[e] = 0.0f;
        if (e < E && s < S) {
            const float4 q = rr[s];
            const float zs = zr[s];
            const float delta = (s + 1 < S ? zr[s + 1] - zs : 1e10f) * dnorm;
            // raw2alpha(raw / B + noise): `noise` [n,S] is the caller's draw (training, nerf.py:175-186), else 0
            const float sig = density_act(q.w / density_scale + (noise ? noise[ray * S + s] : 0.0f), act, act_shift);
            a_[e] = 1.0f - expf(-sig * delta);
            z_[e] = zs;
            const float k = 1.0f + 2.0f * rgb_eps;
            cr[e] = (1.0f / (1.0f + expf(-q.x))) * k - rgb_eps;
            cg[e] = (1.0f / (1.0f + expf(-q.y))) * k - rgb_eps;
            cb[e] = (1.0f / (1.0f + expf(-q.z))) * k - rgb_eps;
            lane_prod *= (1.0f - a_[e] + 1e-10f);
        }
    }
    float T = wave_excl_scan_mul(lane_prod, lane);
    float sr = 0, sg = 0, sb = 0, sd = 0, sw = 0;
#pragma unroll
    for (int e = 0; e < CP_MAXE; ++e) {
        const int s = lane * E + e;
        if (e < E && s < S) {
            const float w = a_[e] * T;
            T *= (1.0f - a_[e] + 1e-10f);
            sr += w * cr[e]; sg += w * cg[e]; sb += w * cb[e];
            sd += w * z_[e]; sw += w;
            if (alpha_out) alpha_out[ray * S + s] = a_[e];
            if (w_out) w_out[ray * S + s] = w;
            sh_w[wave][s] = w;
            sh_z[wave][s] = z_[e];
        }
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sd = wave_sum(sd); sw = wave_sum(sw);
    if (lane == 0) {
        if (rgb_out) { rgb_out[ray * 3 + 0] = sr; rgb_out[ray * 3 + 1] = sg; rgb_out[ray * 3 + 2] = sb; }
        if (disp_out) {
            float disp = 1.0f / fmaxf(1e-10f, sd / (sw + 1e-10f));
            // invalid_mask: torch.isclose(sum w, 0) with default rtol/atol -> |sum w| <= 1e-8
            if (fabsf(sw) <= 1e-8f) disp = 0.0f;
            disp_out[ray] = disp;
        }
        if (acc_out) acc_out[ray] = fminf(sw, 1.0f);
    }
    if (n_imp <= 0 || z_fine == nullptr) return;

    // ---- importance samples: pdf over the S-2 interior weights, bins = S-1 midpoints ----
    PG_WAVE_SYNC();
    const int NB = S - 2;                       // number of pdf entries; cdf has NB+1 = S-1 entries
    float pw[CP_MAXE];
    float part = 0.0f;
#pragma unroll
    for (int e = 0; e < CP_MAXE; ++e) {
        const int i = lane * E + e;
        pw[e] = 0.0f;
        if (e < E && i < NB) { pw[e] = sh_w[wave][i + 1] + 1e-5f; part += pw[e]; }
    }
    const float total = wave_sum(part);
    // cdf[0] = 0, cdf[i+1] = cumsum(pdf)[i]
    float lane_sum = 0.0f;
#pragma unroll
    for (int e = 0; e < CP_MAXE; ++e) { pw[e] = pw[e] / total; lane_sum += pw[e]; }
    float run = wave_incl_scan_add(lane_sum, lane) - lane_sum;
    if (lane == 0) sh_cdf[wave][0] = 0.0f;
#pragma unroll
    for (int e = 0; e < CP_MAXE; ++e) {
        const int i = lane * E + e;
        if (e < E && i < NB) { run += pw[e]; sh_cdf[wave][i + 1] = run; }
    }
    PG_WAVE_SYNC();
    const int NC = NB + 1;
    for (int k = lane; k < n_imp; k += 64) {
        // torch.linspace(0,1,n_imp)
        const float stepu = 1.0f / (float)(n_imp - 1);
        // det = (perturb == 0): linspace; otherwise the caller's uniform draws u_rand [n,n_imp] (ray_utils.py:166-170)
        const float u = u_rand ? u_rand[ray * n_imp + k]
                               : (k < n_imp / 2 ? stepu * (float)k : 1.0f - stepu * (float)(n_imp - 1 - k));
        // searchsorted(cdf, u, right=True): first index with cdf > u
        int lo = 0, hi = NC;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sh_cdf[wave][mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = max(lo - 1, 0), above = min(lo, NC - 1);
        const float c0 = sh_cdf[wave][below], c1 = sh_cdf[wave][above];
        const float b0 = 0.5f * (sh_z[wave][below + 1] + sh_z[wave][below]);
        const float b1 = 0.5f * (sh_z[wave][above + 1] + sh_z[wave][above]);
        float den = c1 - c0;
        if (den < 1e-5f) den = 1.0f;
        const float t = (u - c0) / den;
        sh_z[wave][S + k] = b0 + t * (b1 - b0);
    }
    PG_WAVE_SYNC();
    // ---- stable merge of the S + n_imp depths by rank (== torch.sort of the concat) ----
    // rank of element i = #{j : z_j < z_i or (z_j == z_i and j < i)}.  The coarse depths are increasing and,
    // with deterministic u, so are the new ones (the inverse cdf is monotone): then
    //   rank(coarse i) = i + #{new < z_i},   rank(new k) = k + #{coarse <= z_new[k]}
    // by two binary searches (6 + 5 LDS reads per lane instead of 2 x 80).  Checked per ray (random draws
    // give unsorted samples, a rounding quirk could): anything else takes the all-pairs count.
    const int NTOT = S + n_imp;
    bool sorted = true;
    for (int i = lane; i < NTOT - 1; i += 64)
        if (i != S - 1 && sh_z[wave][i + 1] < sh_z[wave][i]) sorted = false;
    if (__builtin_amdgcn_ballot_w64(!sorted) == 0ull) {
        const float* zc = sh_z[wave];
        const float* zn = sh_z[wave] + S;
        for (int i = lane; i < NTOT; i += 64) {
            const float x = zc[i];
            int lo = 0, hi;
            if (i < S) {            // first new sample >= x
                hi = n_imp;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (zn[mid] < x) lo = mid + 1; else hi = mid; }
            } else {                // first coarse sample > x
                hi = S;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (zc[mid] <= x) lo = mid + 1; else hi = mid; }
            }
            const int rank = (i < S ? i : i - S) + lo;
            z_fine[ray * NTOT + rank] = x;
            if (order) order[ray * NTOT + rank] = i;
        }
        return;
    }
    for (int i = lane; i < NTOT; i += 64) {
        const float x = sh_z[wave][i];
        int rank = 0;
        for (int j = 0; j < NTOT; ++j) {
            const float y = sh_z[wave][j];
            rank += (y < x || (y == x && j < i)) ? 1 : 0;
        }
        z_fine[ray * NTOT + rank] = x;
        if (order) order[ray * NTOT + rank] = i;       // sorted_idxs of torch.sort(cat([z, z_samples]))
    }
}

// A wave takes rays blockIdx.x * CP_WAVES + wave, + gridDim.x * CP_WAVES, ...: a few rays per wave instead of one
// (262 144 one-ray waves per 512 x 512 launch are bound by the rate waves can be dispatched, not by their 2 KB of traffic).
__global__ __launch_bounds__(CP_WAVES * 64) void composite_kernel(
        const float* __restrict__ rays, const float* __restrict__ z, const float4* __restrict__ raw,
        long long n, int S, float density_scale, float rgb_eps, int act, float act_shift,
        float* __restrict__ rgb_out, float* __restrict__ disp_out, float* __restrict__ acc_out,
        float* __restrict__ alpha_out, float* __restrict__ w_out,
        int n_imp, float* __restrict__ z_fine, const float* __restrict__ noise, const float* __restrict__ u_rand,
        int* __restrict__ order) {
    __shared__ float sh_w[CP_WAVES][CP_MAXS];
    __shared__ float sh_z[CP_WAVES][CP_MAXS + CP_MAXI];
    __shared__ float sh_cdf[CP_WAVES][CP_MAXS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long ray = (long long)blockIdx.x * CP_WAVES + wave; ray < n; ray += (long long)gridDim.x * CP_WAVES) {
        composite_ray(ray, lane, wave, sh_w, sh_z, sh_cdf, rays, z, raw, S, density_scale, rgb_eps, act, act_shift, rgb_out, disp_out, acc_out,
                      alpha_out, w_out, n_imp, z_fine, noise, u_rand, order);
        PG_WAVE_SYNC();                         // the wave's LDS rows are reused by its next ray
    }
}

// ray_noise_std > 0 (raycasters.py:660-661, 673-674): the position noise of the points of one pass in the order
// the eval kernel walks them.  src [n,stride,3] holds the caller's draws in the reference's pre-sort order
// (coarse points first, importance points after); order = the sort permutation (or null: identity).
__global__ __launch_bounds__(256) void gather_noise_kernel(const float* __restrict__ src, long long n, int stride, int S,
                                                          const int* __restrict__ order, float* __restrict__ dst) {
    const long long tot = n * S;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const long long ray = i / S;
        const int k = order ? order[i] : (int)(i - ray * S);
        const float* q = src + (ray * stride + k) * 3;
        dst[i * 3] = q[0]; dst[i * 3 + 1] = q[1]; dst[i * 3 + 2] = q[2];
    }
}

}  // namespace pgk

namespace pgk {

// ---- frame front / back end ---------------------------------------
struct FrameGeom {
    int H, W, tlx, tly, bw, bh;        // box = pixels [tly, tly+bh) x [tlx, tlx+bw)
    float fx, fy, cx, cy;
    float R[9], t[3];                   // c2w[:3,:3] row-major, c2w[:3,3]
    float near, far, cam;
};

// ray_batch rows of the box's pixels (get_rays + the bbox gather of kp_to_valid_rays,
// core/utils/ray_utils.py:6-28, 83-136, and the packing of trainer.py:118-137), in the
// reference's row-major pixel order.  Plain fp32 ops in the reference's order, no FMA.
// rays i0 .. i0+n of the box's row-major ray list go to rows 0 .. n of `rays` (a range: one device's share of a frame)
__global__ __launch_bounds__(256) void frame_rays_kernel(const FrameGeom g, long long i0, long long n,
                                                        float* __restrict__ rays, float* __restrict__ cams) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        const long long i = i0 + k;
        const int r = g.tly + (int)(i / g.bw), c = g.tlx + (int)(i % g.bw);
        const float dx = __fdiv_rn(__fsub_rn((float)c, g.cx), g.fx);
        const float dy = -__fdiv_rn(__fsub_rn((float)r, g.cy), g.fy);
        const float dz = -1.0f;
        float d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            d[k] = __fadd_rn(__fadd_rn(__fmul_rn(dx, g.R[3 * k]), __fmul_rn(dy, g.R[3 * k + 1])), __fmul_rn(dz, g.R[3 * k + 2]));
        const float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        float* o = rays + k * 11;
        o[0] = g.t[0]; o[1] = g.t[1]; o[2] = g.t[2];
        o[3] = d[0]; o[4] = d[1]; o[5] = d[2];
        o[6] = g.near; o[7] = g.far;
        o[8] = d[0] * inv; o[9] = d[1] * inv; o[10] = d[2] * inv;      // carried, unused (SURVEY a-5)
        if (cams) cams[k] = g.cam;
    }
}

// Scatter of the rendered box into the frame over the background (run_nerf.py:98-137):
// rgb = rgb_map + (1 - acc) bg, NaN disparity of empty rays -> 0, optional uint8 frame.
__global__ __launch_bounds__(256) void frame_compose_kernel(const FrameGeom g, const float* __restrict__ rgb_map,
                                                           const float* __restrict__ disp_map,
                                                           const float* __restrict__ acc_map,
                                                           const float* __restrict__ bg, float base_bg,
                                                           float* __restrict__ rgb, float* __restrict__ disp,
                                                           float* __restrict__ acc, uint8_t* __restrict__ rgb8) {
    const long long hw = (long long)g.H * g.W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < hw; p += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(p / g.W), c = (int)(p % g.W);
        float b[3] = {base_bg, base_bg, base_bg};
        if (bg) { b[0] = bg[p * 3]; b[1] = bg[p * 3 + 1]; b[2] = bg[p * 3 + 2]; }
        float o[3] = {b[0], b[1], b[2]}, dsp = 0.0f, a = 0.0f;
        if (r >= g.tly && r < g.tly + g.bh && c >= g.tlx && c < g.tlx + g.bw) {
            const long long i = (long long)(r - g.tly) * g.bw + (c - g.tlx);
            a = acc_map[i];
            dsp = disp_map[i];
            if (dsp != dsp) dsp = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) o[k] = __fadd_rn(rgb_map[i * 3 + k], __fmul_rn(__fsub_rn(1.0f, a), b[k]));
        }
        rgb[p * 3] = o[0]; rgb[p * 3 + 1] = o[1]; rgb[p * 3 + 2] = o[2];
        if (disp) disp[p] = dsp;
        if (acc) acc[p] = a;
        if (rgb8) {
#pragma unroll
            for (int k = 0; k < 3; ++k) rgb8[p * 3 + k] = (uint8_t)fminf(fmaxf(o[k] * 255.0f, 0.0f), 255.0f);
        }
    }
}

// ---- bounding cylinder + projected 2-D box per pose, on the device (residue) ------------
// get_kp_bounding_cylinder (core/utils/skeleton_utils.py:635-685, head '-y', the constants of
// kp_to_valid_rays, core/utils/ray_utils.py:89-104) in float32 like numpy on the reference's float32 kp, then
// cylinder_to_box_2d (skeleton_utils.py:700-787) in float64: 50 points per cap, w2c, pinhole, floor / ceil,
// principal-point shift, clip.  One thread per pose.  `ring` = cos / sin of np.linspace(0, 2 pi, 50) as the
// caller's numpy computes them (the box is an integer: everything that feeds floor/ceil is kept in the
// reference's precision and operation order; the 4-term dot products use one fma chain in k order).
__global__ __launch_bounds__(64) void pose_boxes_kernel(const float* __restrict__ kps, long long n, const double* __restrict__ w2c,
                                                       long long w2c_stride, const double* __restrict__ ring, float ext_r,
                                                       float ext_top, float ext_bot, double fx, double fy, int H, int W,
                                                       int offx, int offy, float* __restrict__ cyls, int* __restrict__ boxes) {
    const long long f = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (f >= n) return;
    const float* kp = kps + f * 72;
    const float rx = kp[0], rz = kp[2];
    float reach = 0.0f, hmax = -__FLT_MAX__, hmin = __FLT_MAX__;
    for (int j = 0; j < 24; ++j) {
        const float dx = __fsub_rn(kp[3 * j], rx), dz = __fsub_rn(kp[3 * j + 2], rz);
        reach = fmaxf(reach, sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dz, dz))));   // sqrtf: correctly rounded (__fsqrt_rn is the native approximation here)
        const float hgt = -kp[3 * j + 1];                   // head '-y': height = -y
        hmax = fmaxf(hmax, hgt); hmin = fminf(hmin, hgt);
    }
    const float radius = __fadd_rn(reach, ext_r);
    const float top = -__fadd_rn(hmax, ext_top), bot = -__fsub_rn(hmin, ext_bot);
    float* c = cyls + f * 5;
    c[0] = rx; c[1] = rz; c[2] = radius; c[3] = top; c[4] = bot;
    const double* m = w2c + f * w2c_stride;                  // row-major 4x4 (rows 0..2 used)
    const double cx = rx, cz = rz, cr = radius;
    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
    for (int cap = 0; cap < 2; ++cap) {
        const double y = cap == 0 ? (double)top : (double)bot;
        for (int k = 0; k < 50; ++k) {
            const double px = __dadd_rn(cx, __dmul_rn(ring[2 * k], cr)), pz = __dadd_rn(cz, __dmul_rn(ring[2 * k + 1], cr));
            double q[3];
            for (int r = 0; r < 3; ++r)
                q[r] = fma(m[4 * r + 3], 1.0, fma(m[4 * r + 2], pz, fma(m[4 * r + 1], y, __dmul_rn(m[4 * r], px))));
            const double u = __ddiv_rn(__dmul_rn(q[0], fx), q[2]), v = __ddiv_rn(__dmul_rn(q[1], fy), q[2]);
            lo[0] = fmin(lo[0], u); hi[0] = fmax(hi[0], u);
            lo[1] = fmin(lo[1], v); hi[1] = fmax(hi[1], v);
        }
    }
    int* b = boxes + f * 4;
    const int lim[2] = {W - 1, H - 1}, off[2] = {offx, offy};
    for (int a = 0; a < 2; ++a) {
        int tl = (int)floor(lo[a]) + off[a], br = (int)ceil(hi[a]) + off[a];
        b[a] = min(max(tl, 0), lim[a]);
        b[2 + a] = min(max(br, 0), lim[a]);
    }
}

// ---- batched pose kinematics --------------------------------------
// get_smpl_l2ws (core/utils/skeleton_utils.py:379-463; run_gan.py:2211-2257): axis-angle ->
// rotation (scipy's rotvec -> unit quaternion -> matrix map), chain product down the joint
// tree, kp = l2w[:3,3], skt = l2w^-1 (rigid: [R^T | -R^T t]).  One thread per pose, all in
// float64 like the reference; outputs rounded to float32 once, as the reference's tensors are.
struct PoseConst { double offs[24 * 3]; int parent[24]; };   // offs[j] = rest[j] - rest[parent[j]] (rest[0] for the root)

__global__ __launch_bounds__(64) void pose_kinematics_kernel(const PoseConst pc, const double* __restrict__ bones,
                                                             long long n, float* __restrict__ kps,
                                                             float* __restrict__ skts, double* __restrict__ l2ws_out) {
    const long long f = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (f >= n) return;
    double l2w[24][12];                       // rows 0..2 of each 4x4
    for (int j = 0; j < 24; ++j) {
        const double rx = bones[(f * 24 + j) * 3], ry = bones[(f * 24 + j) * 3 + 1], rz = bones[(f * 24 + j) * 3 + 2];
        const double t2 = rx * rx + ry * ry + rz * rz, th = sqrt(t2);
        const double k = th < 1e-3 ? 0.5 - t2 / 48.0 + t2 * t2 / 3840.0 : sin(0.5 * th) / th;
        double qx = rx * k, qy = ry * k, qz = rz * k, qw = cos(0.5 * th);
        const double nrm = sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
        qx /= nrm; qy /= nrm; qz /= nrm; qw /= nrm;
        const double xx = qx * qx, yy = qy * qy, zz = qz * qz, ww = qw * qw;
        const double xy = qx * qy, zw = qz * qw, xz = qx * qz, yw = qy * qw, yz = qy * qz, xw = qx * qw;
        double rel[12];
        rel[0] = xx - yy - zz + ww; rel[1] = 2.0 * (xy - zw);    rel[2] = 2.0 * (xz + yw);
        rel[4] = 2.0 * (xy + zw);   rel[5] = -xx + yy - zz + ww; rel[6] = 2.0 * (yz - xw);
        rel[8] = 2.0 * (xz - yw);   rel[9] = 2.0 * (yz + xw);    rel[10] = -xx - yy + zz + ww;
        const int p = pc.parent[j];
        for (int c = 0; c < 3; ++c) rel[4 * c + 3] = pc.offs[3 * j + c];
        if (j == 0) {
            for (int e = 0; e < 12; ++e) l2w[0][e] = rel[e];
        } else {
            for (int r = 0; r < 3; ++r) {
                const double a0 = l2w[p][4 * r], a1 = l2w[p][4 * r + 1], a2 = l2w[p][4 * r + 2], a3 = l2w[p][4 * r + 3];
                for (int c = 0; c < 4; ++c)
                    l2w[j][4 * r + c] = a0 * rel[c] + a1 * rel[4 + c] + a2 * rel[8 + c] + (c == 3 ? a3 : 0.0);
            }
        }
    }
    for (int j = 0; j < 24; ++j) {
        const double* m = l2w[j];
        if (kps) for (int c = 0; c < 3; ++c) kps[(f * 24 + j) * 3 + c] = (float)m[4 * c + 3];
        if (l2ws_out) {
            for (int e = 0; e < 12; ++e) l2ws_out[(f * 24 + j) * 16 + e] = m[e];
            l2ws_out[(f * 24 + j) * 16 + 12] = 0.0; l2ws_out[(f * 24 + j) * 16 + 13] = 0.0;
            l2ws_out[(f * 24 + j) * 16 + 14] = 0.0; l2ws_out[(f * 24 + j) * 16 + 15] = 1.0;
        }
        if (skts) {
            float* o = skts + (f * 24 + j) * 16;
            for (int r = 0; r < 3; ++r) {
                o[4 * r] = (float)m[r]; o[4 * r + 1] = (float)m[4 + r]; o[4 * r + 2] = (float)m[8 + r];
                o[4 * r + 3] = (float)(-(m[r] * m[3] + m[4 + r] * m[7] + m[8 + r] * m[11]));
            }
            o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
        }
    }
}

}  // namespace pgk

extern "C" int pg_launch_pose_kinematics(const double* offs72, const int* parents24, const double* bones, long long n,
                                         float* kps, float* skts, double* l2ws, void* stream) {
    if (n <= 0) return 0;
    pgk::PoseConst pc;
    for (int i = 0; i < 72; ++i) pc.offs[i] = offs72[i];
    for (int i = 0; i < 24; ++i) pc.parent[i] = parents24[i];
    hipLaunchKernelGGL(pgk::pose_kinematics_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0,
                       static_cast<hipStream_t>(stream), pc, bones, n, kps, skts, l2ws);
    return (int)hipGetLastError();
}

extern "C" int pg_launch_frame_rays(const pgk::FrameGeom* g, long long i0, long long n, float* rays, float* cams, void* stream) {
    if (n <= 0) return 0;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(pgk::frame_rays_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *g, i0, n, rays, cams);
    return (int)hipGetLastError();
}

extern "C" int pg_launch_pose_boxes(const float* kps, long long n, const double* w2c, long long w2c_stride, const double* ring,
                                    float ext_r, float ext_top, float ext_bot, double fx, double fy, int H, int W, int offx,
                                    int offy, float* cyls, int* boxes, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pgk::pose_boxes_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream),
                       kps, n, w2c, w2c_stride, ring, ext_r, ext_top, ext_bot, fx, fy, H, W, offx, offy, cyls, boxes);
    return (int)hipGetLastError();
}

extern "C" int pg_launch_frame_compose(const pgk::FrameGeom* g, const float* rgb_map, const float* disp_map,
                                       const float* acc_map, const float* bg, float base_bg, float* rgb, float* disp,
                                       float* acc, uint8_t* rgb8, void* stream) {
    const long long hw = (long long)g->H * g->W;
    if (hw <= 0) return 0;
    const unsigned blocks = (unsigned)((hw + 255) / 256 < 4096 ? (hw + 255) / 256 : 4096);
    hipLaunchKernelGGL(pgk::frame_compose_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *g,
                       rgb_map, disp_map, acc_map, bg, base_bg, rgb, disp, acc, rgb8);
    return (int)hipGetLastError();
}

// doubles of scratch the two-launch form needs for n rays in chunks of `chunk` (0: the one-launch form is used)
extern "C" long long pg_sample_coarse_scratch(long long n, int chunk) {
    if (n <= 0 || chunk <= pgk::SC_SUB) return 0;
    const long long groups = (n + chunk - 1) / chunk, nsub = (chunk + pgk::SC_SUB - 1) / pgk::SC_SUB;
    return groups * nsub * 5;
}

extern "C" int pg_launch_sample_coarse(const float* rays, const float* cyls, long long cyl_stride,
                                       long long n, int chunk, int S, int lindisp,
                                       float* near_far, float* z, const float* t_rand, double* scratch, void* stream) {
    if (n <= 0) return 0;
    if (scratch && chunk > pgk::SC_SUB) {
        const long long groups = (n + chunk - 1) / chunk;
        const int nsub = (chunk + pgk::SC_SUB - 1) / pgk::SC_SUB;
        hipStream_t s = static_cast<hipStream_t>(stream);
        hipLaunchKernelGGL(pgk::sc_partial_kernel, dim3((unsigned)(groups * nsub)), dim3(pgk::SC_THREADS), 0, s,
                           rays, cyls, cyl_stride, n, chunk, nsub, near_far, scratch);
        hipLaunchKernelGGL(pgk::sc_sample_kernel, dim3((unsigned)(groups * nsub)), dim3(pgk::SC_THREADS), 0, s,
                           rays, cyls, cyl_stride, n, chunk, nsub, S, lindisp, near_far, z, t_rand, scratch);
        return (int)hipGetLastError();
    }
    const long long blocks = (n + chunk - 1) / chunk;
    hipLaunchKernelGGL(pgk::sample_coarse_kernel, dim3((unsigned)blocks), dim3(pgk::SC_THREADS), 0,
                       static_cast<hipStream_t>(stream), rays, cyls, cyl_stride, n, chunk, S, lindisp,
                       near_far, z, t_rand);
    return (int)hipGetLastError();
}

extern "C" int pg_launch_composite(const float* rays, const float* z, const float* raw, long long n, int S,
                                   float density_scale, float rgb_eps, int density_act, float act_shift, float* rgb, float* disp,
                                   float* acc, float* alpha, float* weights, int n_imp, float* z_fine, const float* noise,
                                   const float* u_rand, int* order, void* stream) {
    if (n <= 0) return 0;
    long long blocks = (n + pgk::CP_WAVES - 1) / pgk::CP_WAVES;
    if (blocks > 16384) blocks = 16384;         // a wave takes several rays (measured flat from 4 k to 32 k blocks, -0.05 ms per frame against one ray per wave)
    hipLaunchKernelGGL(pgk::composite_kernel, dim3((unsigned)blocks), dim3(pgk::CP_WAVES * 64), 0,
                       static_cast<hipStream_t>(stream), rays, z, reinterpret_cast<const float4*>(raw), n, S,
                       density_scale, rgb_eps, density_act, act_shift, rgb, disp, acc, alpha, weights, n_imp, z_fine, noise, u_rand, order);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------
// MFMA calibration: the rate this device SUSTAINS on v_mfma_f32_32x32x16_{bf16,f16} with nothing else
// in the loop (operands in registers, non-trivial values: the clock the chip holds under MFMA load
// depends on the data).  bench.py reports it beside the nominal peak: the fused kernels are priced
// against the nominal figure, this says how much of the gap is the chip's own clock management.
// 8 waves per workgroup (2 per SIMD), one workgroup per CU, 4 independent accumulator chains.
// ------------------------------------------------------------------------------------
namespace pgk {
typedef _Float16 cal_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 cal_b8 __attribute__((ext_vector_type(8)));
typedef float cal_f16v __attribute__((ext_vector_type(16)));

// LDSFED: the A operand of every MFMA comes from LDS (one conflict-free ds_read_b128 per MFMA and wave, a
// lane-linear 1-KiB fragment image like the weight ring's), reads issued one MFMA ahead: the structural
// ceiling of "32 points per wave, weights from LDS" -- what the fused 16-bit kernels could reach if the
// fragment reads were their only cost.
// SMALL: the same FLOPs as v_mfma_f32_16x16x32 (two per unit: one A fragment of 16 out channels x 32 k against the
// two 16-point column tiles of a wave's 32 points).  MI355X_MICROARCH.md (DVFS give-back 7): the chip can hold a
// higher clock on one MFMA shape than on the other at equal cycles per FLOP, so the shapes are ranked by wall time.
typedef float cal_f4v __attribute__((ext_vector_type(4)));
// independent accumulators the loop cycles through, minus one (measurement aid: -DPG_CAL_MASK=1 is the two-accumulator
// chain of a fused kernel's out-tile-major hidden layer: each MFMA accumulates onto the result of the one before the last)
#ifndef PG_CAL_MASK
#define PG_CAL_MASK 7
#endif
template <bool F16, bool LDSFED>
__global__ __launch_bounds__(512) void mfma_rate_small_kernel(int iters, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) unsigned frag[32 * 256];
    const int lane = threadIdx.x & 63;
    float seed = 0.37f + 0.0131f * (float)lane + 0.00071f * (float)(threadIdx.x >> 6);
    cal_f4v acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c][e] = 0.0f;
    float av[8], bv[16];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        seed = seed * 1.6180339f; seed -= floorf(seed);
        av[e] = seed - 0.5f;
        seed = seed * 2.2360679f; seed -= floorf(seed);
        bv[e] = (seed - 0.5f) * 0.25f;
        seed = seed * 1.7320508f; seed -= floorf(seed);
        bv[8 + e] = (seed - 0.5f) * 0.25f;
    }
    typedef unsigned cal_u4 __attribute__((ext_vector_type(4)));
    cal_u4 a, b0, b1;
    if constexpr (F16) {
        cal_h8 ah, bh, ch;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)av[e]; bh[e] = (_Float16)bv[e]; ch[e] = (_Float16)bv[8 + e]; }
        a = __builtin_bit_cast(cal_u4, ah); b0 = __builtin_bit_cast(cal_u4, bh); b1 = __builtin_bit_cast(cal_u4, ch);
    } else {
        cal_b8 ab, bb, cb;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)av[e]; bb[e] = (__bf16)bv[e]; cb[e] = (__bf16)bv[8 + e]; }
        a = __builtin_bit_cast(cal_u4, ab); b0 = __builtin_bit_cast(cal_u4, bb); b1 = __builtin_bit_cast(cal_u4, cb);
    }
    auto mma = [&](cal_f4v& c, const cal_u4& x, const cal_u4& y) {
        if constexpr (F16) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cal_h8, x), __builtin_bit_cast(cal_h8, y), c, 0, 0, 0);
        else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cal_b8, x), __builtin_bit_cast(cal_b8, y), c, 0, 0, 0);
    };
    if constexpr (LDSFED) {
        for (int i = threadIdx.x; i < 32 * 256; i += 512) frag[i] = a[i & 3] ^ (unsigned)(i >> 8) * 0x00010001u;
        __syncthreads();
        const cal_u4* img = reinterpret_cast<const cal_u4*>(frag) + lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const cal_u4 x = img[u * 64];
                mma(acc[(2 * u) & PG_CAL_MASK], x, b0);
                mma(acc[(2 * u + 1) & PG_CAL_MASK], x, b1);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) { mma(acc[(2 * c) & PG_CAL_MASK], a, b0); mma(acc[(2 * c + 1) & PG_CAL_MASK], a, b1); }
        }
    }
    float r = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) r += acc[c][e];
    if (r == 123.456f) sink[0] = r;
}

template <bool F16, bool LDSFED>
__global__ __launch_bounds__(512) void mfma_rate_kernel(int iters, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) unsigned frag[32 * 256];      // 32 fragments of 1 KiB
    const int lane = threadIdx.x & 63;
    // operand values in (-1, 1) with full mantissas, different per lane and element
    float seed = 0.37f + 0.0131f * (float)lane + 0.00071f * (float)(threadIdx.x >> 6);
    cal_f16v acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.0f;
    float av[8], bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        seed = seed * 1.6180339f; seed -= floorf(seed);
        av[e] = seed - 0.5f;
        seed = seed * 2.2360679f; seed -= floorf(seed);
        bv[e] = (seed - 0.5f) * 0.25f;
    }
    typedef unsigned cal_u4 __attribute__((ext_vector_type(4)));
    cal_u4 a, b;
    if constexpr (F16) {
        cal_h8 ah, bh;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)av[e]; bh[e] = (_Float16)bv[e]; }
        a = __builtin_bit_cast(cal_u4, ah); b = __builtin_bit_cast(cal_u4, bh);
    } else {
        cal_b8 ab, bb;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)av[e]; bb[e] = (__bf16)bv[e]; }
        a = __builtin_bit_cast(cal_u4, ab); b = __builtin_bit_cast(cal_u4, bb);
    }
    auto mma = [&](cal_f16v& c, const cal_u4& x) {
        if constexpr (F16) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cal_h8, x), __builtin_bit_cast(cal_h8, b), c, 0, 0, 0);
        else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cal_b8, x), __builtin_bit_cast(cal_b8, b), c, 0, 0, 0);
    };
    if constexpr (LDSFED) {
        for (int i = threadIdx.x; i < 32 * 256; i += 512) frag[i] = a[i & 3] ^ (unsigned)(i >> 8) * 0x00010001u;
        __syncthreads();
        const cal_u4* img = reinterpret_cast<const cal_u4*>(frag) + lane;      // fragment f: img[f * 64]
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const cal_u4 x = img[u * 64];
                mma(acc[u & 3], x);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) mma(acc[c], a);
        }
    }
    float r = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) r += acc[c][e];
    if (r == 123.456f) sink[0] = r;        // never true for these operands; keeps the chains alive
}
}  // namespace pgk

// launches `blocks` workgroups x 8 waves x iters x 32 MFMAs; returns 0 or the hipError
extern "C" int pg_launch_mfma_rate(int f16, int lds_fed, int blocks, int iters, float* sink, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (lds_fed & 2) {      // bit 1: the 16x16x32 shape (same FLOPs per unit)
        const bool fed = (lds_fed & 1) != 0;
        if (f16 && fed) hipLaunchKernelGGL((pgk::mfma_rate_small_kernel<true, true>), dim3(blocks), dim3(512), 0, s, iters, sink);
        else if (f16) hipLaunchKernelGGL((pgk::mfma_rate_small_kernel<true, false>), dim3(blocks), dim3(512), 0, s, iters, sink);
        else if (fed) hipLaunchKernelGGL((pgk::mfma_rate_small_kernel<false, true>), dim3(blocks), dim3(512), 0, s, iters, sink);
        else hipLaunchKernelGGL((pgk::mfma_rate_small_kernel<false, false>), dim3(blocks), dim3(512), 0, s, iters, sink);
        return (int)hipGetLastError();
    }
    if (f16 && lds_fed) hipLaunchKernelGGL((pgk::mfma_rate_kernel<true, true>), dim3(blocks), dim3(512), 0, s, iters, sink);
    else if (f16) hipLaunchKernelGGL((pgk::mfma_rate_kernel<true, false>), dim3(blocks), dim3(512), 0, s, iters, sink);
    else if (lds_fed) hipLaunchKernelGGL((pgk::mfma_rate_kernel<false, true>), dim3(blocks), dim3(512), 0, s, iters, sink);
    else hipLaunchKernelGGL((pgk::mfma_rate_kernel<false, false>), dim3(blocks), dim3(512), 0, s, iters, sink);
    return (int)hipGetLastError();
}

extern "C" int pg_launch_gather_noise(const float* src, long long n, int stride, int S, const int* order, float* dst, void* stream) {
    if (n <= 0) return 0;
    const long long tot = n * S;
    const unsigned blocks = (unsigned)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192);
    hipLaunchKernelGGL(pgk::gather_noise_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, n, stride, S, order, dst);
    return (int)hipGetLastError();
}

extern "C" int pg_composite_max_samples(void) { return pgk::CP_MAXS; }
extern "C" int pg_composite_max_importance(void) { return pgk::CP_MAXI; }
