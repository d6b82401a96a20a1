// pg_device.h -- device-side helpers shared by the fused embed+MLP kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "pg_program.h"

namespace pgd {
using namespace pgl;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Kernel arguments of one fused embed+MLP launch (one net, one pass).
struct EvalArgs {
    const float* rays;        // [n,11] ray_batch rows (o, d, near, far, viewdir): o,d used
    const float* z;           // [n,S]  depths of the points p = o + d z
    const float* pts;         // density query: explicit points [n_points,3] instead of o + d z (else null)
    const float* pnoise;      // training (ray_noise_std): position noise [n_points,3] added to o + d z (else null)
    const float* skts;        // [*,24,4,4]
    const float* cams;        // [n] frame-code index (float) or null
    const float* codes;       // [n_codes+1,16], last row = mean code; null if no frame code
    const uint8_t* wstream;   // packed weights (pg_pack.cpp)
    const uint8_t* wy;        // Y-stage weights of the factorised view layer (pack_vy), or null
    const float* rec_ab;      // per-ray records (pg_rayrec.hip) of the 16x16x32 kernel, or null
    const uint8_t* rec_y;
    const float* bias;        // BIAS_FLOATS
    const float* cutoff;      // [48] = cutoff_dist of embed_fn (24) then embeddirs_fn (24)
    float* raw;               // [n*S,4] (rgb_raw, sigma_raw)
    float* dbg;               // [n*S,256] pre-activation of density layer 0, or null
    long long pose_stride;    // floats between consecutive rays' skts (0 = shared pose)
    long long n_points;       // n*S
    int n_rays;
    int S;
    int n_codes;
    int n_iters;              // workgroup passes = ceil(n_points / points per pass)
    float tau_v, tau_d;
    int dbg_stage;            // which activation `dbg` receives (see pg_stage_eval)
    int far_skip;             // 1: limbs out of cutoff range are skipped (pg_eval16r.hip, pg_evalc.hip REC); 0: every limb computed
};

// Kernel arguments of the per-ray record kernel (pg_rayrec.hip) in front of a factorised 16-bit launch.
struct RecArgs {
    const float* rays;        // [n,11]
    const float* skts;        // [*,24,4,4]
    const float* cams;        // [n] or null
    const float* codes;       // [n_codes+1,16] or null
    const uint8_t* wy;        // Y-stage weights (pack_vy)
    float* rec_ab;            // [n + REC_PAD_RAYS][24][8]
    uint8_t* rec_y;           // [n + REC_PAD_RAYS][REC_Y_BYTES]
    const float* z;           // [n,S] depths of the launch the records are for (the per-joint distance bound below)
    long long pose_stride;
    int n_rays;
    int n_codes;
    int S;
};

// slot -> joint tables of pg_layout.h (PERM16, PERMC) as a device lookup at a run-time index: 24 x 5 bits in two constants
template <const int (&P)[24]>
__device__ __forceinline__ int perm_dev(int s) {
    constexpr unsigned long long lo = [] { unsigned long long v = 0; for (int i = 0; i < 12; ++i) v |= (unsigned long long)P[i] << (5 * i); return v; }();
    constexpr unsigned long long hi = [] { unsigned long long v = 0; for (int i = 0; i < 12; ++i) v |= (unsigned long long)P[12 + i] << (5 * i); return v; }();
    return (int)(((s < 12 ? lo : hi) >> (5 * (s < 12 ? s : s - 12))) & 31);
}
__device__ __forceinline__ int slot_joint_dev(int s) { return perm_dev<PERM16>(s); }
__device__ __forceinline__ int slotc_joint_dev(int s) { return perm_dev<PERMC>(s); }

// The opt-in to > 64 KiB of dynamic LDS is per (kernel, device): set once per device, from any host
// thread (pg_render_frames drives one thread per device).
inline hipError_t ensure_lds_attr(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// Weight-stream consumer state of one wave: an NSLOT-deep LDS ring of 16-KiB chunks filled
// by LDS-DMA (global_load_lds_dwordx4), DEPTH = NSLOT-1 chunks in flight.  Every wave copies
// 16/NWAVE of each chunk's 1-KiB pieces.  `enter()` is the only synchronisation:
//   counted vmcnt (own pieces of the chunk being entered have landed; the younger chunks'
//   DMA stays in flight across the barrier) -> lgkmcnt(0) (own reads of the previous chunk
//   returned) -> raw s_barrier -> the chunk is readable by every wave and the slot of the
//   previous chunk is free -> start the DMA of chunk +DEPTH into it.
// Chunks are consumed strictly in stream order, wrapping from NCHUNK-1 to 0.
#ifndef PG_RING_SLOTS
#define PG_RING_SLOTS 3
#endif
#ifndef PG_DMA_BASE
#define PG_DMA_BASE 0         // first DMA wave when only a subset of the waves refills the ring
#endif
// Only the first NDMA waves of the workgroup issue the DMA (PER pieces each per chunk); the
// others just synchronise.  The oldest wave of each SIMD wins issue arbitration and reaches
// every chunk barrier early: giving those waves all of the refill work (NDMA = NWAVE/2) moves
// its issue cost off the critical wave of the SIMD.
// MASK_NX > 0 (pg_eval16r.hip): chunks MASK_X0 + k and MASK_X1 + k (k < MASK_NX) of the stream hold the weights of one
// LIMB each; chunk k of both ranges is left out of a pass's chunk sequence when bit k of the pass's mask is set (no
// workgroup point is within cutoff range of the limb: nothing would read it); MASK_X2 >= 0: a third such range.  The mask must be the same in every
// wave.  `pf_mask` is the mask of the pass the prefetch pointer is in, `nx_mask` that of the pass behind it (taken
// over when the pointer wraps to the head of the stream).
template <int NWAVE, int NCHUNK_, int NDMA = NWAVE, int MASK_NX = 0, int MASK_X0 = 0, int MASK_X1 = 0, int MASK_X2 = -1>
struct Stream {
    static constexpr int NSLOT = PG_RING_SLOTS;
    static constexpr int DEPTH = NSLOT - 1;
    static constexpr int PER = CHUNK_BYTES / 1024 / NDMA;    // DMA instructions per DMA wave per chunk
    const uint8_t* wstream;
    uint8_t* ring;
    int wave, lane;
    // Loop-carried, optimizer-opaque state (asm "+s"/"+v" below).  Derived from the chunk
    // index these would be per-chunk loop invariants, and hipcc (a) hoists every chunk's
    // 64-bit source address to kernel entry and spills them, (b) does not model
    // global_load_lds as a write to LDS, so it may CSE/hoist ring reads across fills.
    uint32_t next_off;     // byte offset in the stream of the next chunk to fetch
    uint32_t fill_slot;    // ring slot that fetch goes to
    uint32_t rd_off;       // this lane's byte offset into the ring for the current chunk
    uint32_t ring_lds;     // LDS byte address of the ring (for asm ds_read)
    uint32_t lane16;       // lane * 16
    uint32_t pf_mask = 0u, nx_mask = 0u;

    // next_off: one chunk on in the (masked) sequence.  Scalar arithmetic only.
    __device__ __forceinline__ void skip_masked() {
        if constexpr (MASK_NX > 0) {
            const uint32_t c = next_off / CHUNK_BYTES;
            const uint32_t k0 = c - (uint32_t)MASK_X0, k1 = c - (uint32_t)MASK_X1, k2 = c - (uint32_t)MASK_X2;
            const uint32_t k = k0 < (uint32_t)MASK_NX ? k0 : (MASK_X2 >= 0 && k2 < (uint32_t)MASK_NX ? k2 : k1);
            // bits >= MASK_NX of a mask are zero: the run of set bits from k ends inside the limb range
            if (k < (uint32_t)MASK_NX) next_off += (uint32_t)__builtin_ctz(~(pf_mask >> k)) * CHUNK_BYTES;
        }
    }
    __device__ __forceinline__ void advance() {
        next_off = next_off + CHUNK_BYTES == (uint32_t)NCHUNK_ * CHUNK_BYTES ? 0u : next_off + CHUNK_BYTES;
        if constexpr (MASK_NX > 0) {
            if (next_off == 0u) pf_mask = nx_mask;
            skip_masked();
        }
    }

    __device__ __forceinline__ bool dma_wave() const { return NDMA == NWAVE || (wave >= dma_base() && wave < dma_base() + NDMA); }
    static constexpr int dma_base() { return NDMA == NWAVE ? 0 : PG_DMA_BASE; }
    __device__ __forceinline__ void prefetch_next() {
        const uint8_t* src = wstream + next_off + ((wave - dma_base()) * (PER * 1024) + lane * 16);
        uint8_t* dst = ring + fill_slot * CHUNK_BYTES + (wave - dma_base()) * (PER * 1024);
        if (dma_wave()) {
#pragma unroll
            for (int i = 0; i < PER; ++i) glds16(src + i * 1024, dst + i * 1024);
        }
        advance();
        fill_slot = fill_slot + 1 == NSLOT ? 0u : fill_slot + 1;
        asm volatile("" : "+s"(next_off), "+s"(fill_slot));
    }
    // kernel start: DEPTH chunks in flight; the first enter() consumes slot 0
    // first_mask: the mask of the first pass (masked streams)
    __device__ __forceinline__ void start(uint32_t first_mask = 0u) {
        next_off = 0;
        fill_slot = 0;
        pf_mask = nx_mask = first_mask;
        skip_masked();
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) prefetch_next();
        rd_off = (uint32_t)(NSLOT - 1) * CHUNK_BYTES + lane * 16;   // advanced by the first enter()
        rd_nxt = lane * 16;
    }
#if defined(PG_STAMPS)
    unsigned long long t_vm = 0, t_bar = 0;
#endif
    __device__ __forceinline__ void enter(int /*chunk index, documentation only*/) {
#if defined(PG_STAMPS)
        unsigned long long s0, s1, s2;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s0)::"memory");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PER) : "memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s1)::"memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s2)::"memory");
        t_vm += s1 - s0;
        t_bar += s2 - s1;
#else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PER) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#endif
        prefetch_next();
        rd_off = rd_off + CHUNK_BYTES >= (uint32_t)NSLOT * CHUNK_BYTES + lane * 16 ? rd_off - (NSLOT - 1) * CHUNK_BYTES
                                                                                  : rd_off + CHUNK_BYTES;
        asm volatile("" : "+v"(rd_off)::"memory");
    }
    // Split form of enter(): the same waits and barrier, but the refill of the freed slot is
    // left to the caller, who issues its PER pieces one at a time with piece(i) spread over
    // the units of the chunk just entered (all PER before the next enter: the vmcnt count
    // relies on it).  Issued as one burst right after the barrier, the 8 waves' pieces queue up
    // in the texture addresser and every wave stalls on the issue.
    uint32_t cur_src, cur_dst;   // stream byte offset / ring byte offset of this wave's share
    uint32_t rd_nxt;             // this lane's ring offset for the chunk AFTER the current one (enter_ahead / issue_ahead)
    // plain_ok(c): at the entry of stream chunk c the ring bookkeeping is known at compile time -- no masked chunk lies between c
    // and the chunk behind the one fetched now, and the stream does not wrap there -- so `next_off += CHUNK_BYTES` replaces
    // advance() (the wrap test, the mask hand-over and skip_masked: ~20 scalar instructions per entry)
    static constexpr bool plain_ok(int c) {
        if (c < 0) return false;
        const int n = c + DEPTH + 1;
        if (n >= NCHUNK_) return false;
        if (MASK_NX > 0) {
            const int xs[3] = {MASK_X0, MASK_X1, MASK_X2};
            for (int i = 0; i < 3; ++i)
                if (xs[i] >= 0 && c + 1 <= xs[i] + MASK_NX - 1 && n >= xs[i]) return false;
        }
        return true;
    }
    __device__ __forceinline__ void enter_split(bool plain = false) {
#if defined(PG_STAMPS)
        unsigned long long s0, s1, s2;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s0)::"memory");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PER) : "memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s1)::"memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s2)::"memory");
        t_vm += s1 - s0;
        t_bar += s2 - s1;
#else
        if (dma_wave()) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PER) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !defined(PG_ABL_NOBARRIER)      // timing ablation only (racy, wrong results): what ANY scheme without a workgroup barrier per chunk could gain at most
        __builtin_amdgcn_s_barrier();
#endif
#endif
        cur_src = next_off + (wave - dma_base()) * (PER * 1024);
        cur_dst = fill_slot * CHUNK_BYTES + (wave - dma_base()) * (PER * 1024);
        if (plain) next_off += CHUNK_BYTES; else advance();
        fill_slot = fill_slot + 1 == NSLOT ? 0u : fill_slot + 1;
        asm volatile("" : "+s"(next_off), "+s"(fill_slot), "+s"(cur_src), "+s"(cur_dst));
        rd_off = rd_off + CHUNK_BYTES >= (uint32_t)NSLOT * CHUNK_BYTES + lane * 16 ? rd_off - (NSLOT - 1) * CHUNK_BYTES
                                                                                  : rd_off + CHUNK_BYTES;
        asm volatile("" : "+v"(rd_off)::"memory");
    }
    // Chunk entry with the data of the NEXT chunk confirmed as well (needs DEPTH >= 3, i.e. a ring of >= 4
    // slots): a wave waits for its pieces of chunk c+1 too before the barrier, so behind the barrier every wave
    // knows that chunks c and c+1 have landed, and reads of chunk c+1 may be issued BEFORE the barrier of its own
    // entry (issue_ahead).  The A-fragment pipe then runs across chunk boundaries instead of restarting empty
    // behind every barrier (an LDS latency exposed per chunk with one wave per SIMD).  The barrier of entry(c+1)
    // still says "everyone is done reading chunk c" before its slot is refilled; this wave's reads of chunk c
    // are complete by then because its last MFMA of the chunk waited for them (LGKM returns in order), so
    // no lgkmcnt(0) is needed and the reads already in flight for chunk c+1 stay in flight.
    __device__ __forceinline__ void enter_ahead() {
        static_assert(DEPTH >= 3 || NSLOT < 4, "enter_ahead needs three chunks in flight");
        if (dma_wave()) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH >= 3 ? DEPTH - 2 : 0) * PER) : "memory");
        __builtin_amdgcn_s_barrier();
        cur_src = next_off + (wave - dma_base()) * (PER * 1024);
        cur_dst = fill_slot * CHUNK_BYTES + (wave - dma_base()) * (PER * 1024);
        next_off = next_off + CHUNK_BYTES == (uint32_t)NCHUNK_ * CHUNK_BYTES ? 0u : next_off + CHUNK_BYTES;
        fill_slot = fill_slot + 1 == NSLOT ? 0u : fill_slot + 1;
        asm volatile("" : "+s"(next_off), "+s"(fill_slot), "+s"(cur_src), "+s"(cur_dst));
        rd_off = rd_nxt;
        rd_nxt = rd_nxt + CHUNK_BYTES >= (uint32_t)NSLOT * CHUNK_BYTES + lane * 16 ? rd_nxt - (NSLOT - 1) * CHUNK_BYTES
                                                                                  : rd_nxt + CHUNK_BYTES;
        asm volatile("" : "+v"(rd_off), "+v"(rd_nxt)::"memory");
    }
    // One refill piece, issued by asm in the SGPR-base + 32-bit lane offset form: through the
    // builtin hipcc forms a 64-bit VGPR address (a v_lshl_add_u64) for every one of the ~224
    // pieces per pass.  m0 (the LDS destination) is not otherwise used by these kernels.
    __device__ __forceinline__ void piece(int i) const {
#if defined(PG_ABL_NODMA)       // timing ablation only (wrong results): no refill of the ring
        return;
#endif
        if (!dma_wave()) return;
#if defined(PG_VISIBLE_DMA)     // through the builtin: hipcc counts the piece in its own vmcnt bookkeeping, so its
        // waits for loads issued a pass ahead come out COUNTED (pg_eval16.hip) instead of vmcnt(0)
        glds16(wstream + (cur_src + i * 1024) + lane16, ring + cur_dst + i * 1024);
        return;
#endif
        const uint8_t* sbase = wstream + (cur_src + i * 1024);      // wave-uniform
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(ring_lds + cur_dst + i * 1024), "v"(lane16), "s"(sbase) : "memory");
    }
    __device__ __forceinline__ const uint8_t* at(int /*chunk*/, int byte_off) const {
        return ring + byte_off + rd_off;
    }
    // asynchronous ds_read_b128 of unit `pos` (compile-time) of the current chunk; the caller
    // retires it with a counted s_waitcnt lgkmcnt (hipcc does not see this load)
    template <typename R>
    __device__ __forceinline__ void issue(R& dst, int pos) const {
        const unsigned addr = ring_lds + rd_off;
        switch (pos) {   // the DS offset field wants a literal: pos is constant after unrolling
#define PG_ISSUE_CASE(P) case P: asm volatile("ds_read_b128 %0, %1 offset:" #P "*1024" : "=v"(dst) : "v"(addr)); break;
            PG_ISSUE_CASE(0) PG_ISSUE_CASE(1) PG_ISSUE_CASE(2) PG_ISSUE_CASE(3) PG_ISSUE_CASE(4) PG_ISSUE_CASE(5)
            PG_ISSUE_CASE(6) PG_ISSUE_CASE(7) PG_ISSUE_CASE(8) PG_ISSUE_CASE(9) PG_ISSUE_CASE(10) PG_ISSUE_CASE(11)
            PG_ISSUE_CASE(12) PG_ISSUE_CASE(13) PG_ISSUE_CASE(14) PG_ISSUE_CASE(15) PG_ISSUE_CASE(16) PG_ISSUE_CASE(17)
            PG_ISSUE_CASE(18) PG_ISSUE_CASE(19) PG_ISSUE_CASE(20) PG_ISSUE_CASE(21) PG_ISSUE_CASE(22) PG_ISSUE_CASE(23)
            PG_ISSUE_CASE(24) PG_ISSUE_CASE(25) PG_ISSUE_CASE(26) PG_ISSUE_CASE(27) PG_ISSUE_CASE(28) PG_ISSUE_CASE(29)
            PG_ISSUE_CASE(30) PG_ISSUE_CASE(31)
#undef PG_ISSUE_CASE
            default: __builtin_unreachable();
        }
    }
    // the same read from the chunk AFTER the current one (enter_ahead)
    template <typename R>
    __device__ __forceinline__ void issue_ahead(R& dst, int pos) const {
        const unsigned addr = ring_lds + rd_nxt;
        switch (pos) {
#define PG_ISSUE_CASE(P) case P: asm volatile("ds_read_b128 %0, %1 offset:" #P "*1024" : "=v"(dst) : "v"(addr)); break;
            PG_ISSUE_CASE(0) PG_ISSUE_CASE(1) PG_ISSUE_CASE(2) PG_ISSUE_CASE(3) PG_ISSUE_CASE(4) PG_ISSUE_CASE(5)
            PG_ISSUE_CASE(6) PG_ISSUE_CASE(7) PG_ISSUE_CASE(8) PG_ISSUE_CASE(9) PG_ISSUE_CASE(10) PG_ISSUE_CASE(11)
#undef PG_ISSUE_CASE
            default: __builtin_unreachable();
        }
    }
    // before the wave exits: no DMA may be left in flight
    __device__ __forceinline__ void drain() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

__device__ __forceinline__ f32x16 load_bias(const float* bias, int tile, int h) {
    const float4* p = reinterpret_cast<const float4*>(bias + (tile * 2 + h) * 16);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    f32x16 r = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    return r;
}

// Workgroup barrier for LDS traffic only.  __syncthreads() would also wait vmcnt(0), i.e.
// drain the weight-stream DMA that is deliberately in flight across pass boundaries.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// LDS carve-up (bytes) shared by both kernel shapes
constexpr int LDS_RING = 0;                                  // PG_RING_SLOTS x CHUNK_BYTES
constexpr int LDS_BIAS = PG_RING_SLOTS * CHUNK_BYTES;        // BIAS_FLOATS floats
constexpr int LDS_CUT = LDS_BIAS + BIAS_FLOATS * 4;          // 48 floats
constexpr int LDS_RTAB = LDS_CUT + 48 * 4;                   // MAXR slots
constexpr int LDS_TOTAL = LDS_RTAB + MAXR * SLOT_FLOATS * 4;
static_assert(LDS_BIAS % 16 == 0 && LDS_CUT % 16 == 0 && LDS_RTAB % 16 == 0, "LDS alignment");
#if !defined(PG_RING_EXPERIMENT)      // ring-depth experiments of ONE kernel: the others' layouts need not fit
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget of one CU");
#endif

// q = (skt @ [p;1]).xyz with skt rows 0..2 at sk[0..11]      (core/encoders.py:8-23)
__device__ __forceinline__ void bone_local(const float* sk, float px, float py, float pz,
                                           float& qx, float& qy, float& qz) {
    const float4 a = *reinterpret_cast<const float4*>(sk);
    const float4 b = *reinterpret_cast<const float4*>(sk + 4);
    const float4 c = *reinterpret_cast<const float4*>(sk + 8);
    qx = fmaf(a.z, pz, fmaf(a.y, py, fmaf(a.x, px, a.w)));
    qy = fmaf(b.z, pz, fmaf(b.y, py, fmaf(b.x, px, b.w)));
    qz = fmaf(c.z, pz, fmaf(c.y, py, fmaf(c.x, px, c.w)));
}

// Math flavour of the per-point embedding.  ACCURATE: correctly-rounded sqrt / division and
// ocml expf / sincosf (fp32-grade kernels, parity mode).  FAST: the hardware transcendental
// units (v_sqrt / v_rcp / v_exp / v_sin / v_cos, ~1e-6 relative) -- used by the 16-bit
// operand kernels, whose operands are rounded to 8-11 bits right afterwards.
template <bool FAST> __device__ __forceinline__ float pg_sqrt(float x) {
    return FAST ? __builtin_amdgcn_sqrtf(x) : sqrtf(x);
}
template <bool FAST> __device__ __forceinline__ float pg_div(float a, float b) {
    return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b;
}

// cutoff weight 1 - sigmoid(tau (v - c)) = 1 / (1 + exp(t))       (cutoff_embedder.py:139-146)
template <bool FAST>
__device__ __forceinline__ float cutoff_weight(float v, float tau, float c) {
    const float t = tau * (v - c);
    if (FAST) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * 1.4426950408889634f));
    return 1.0f - 1.0f / (1.0f + expf(-t));
}

// FAST form with the constants folded on entry to the kernel: tl = tau log2(e), cs = -c tl
__device__ __forceinline__ float cutoff_weight_fast(float v, float tl, float cs) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(v, tl, cs)));
}

// The 18 density-input values of one joint for one point (RelDist + VecNorm + cutoff
// embedding, encoders.py:101-122,172-193; cutoff_embedder.py:111-174):
//   x[0] = v w, x[1+2f] = sin(2^f v) w, x[2+2f] = cos(2^f v) w (f < 7), x[15..17] = q/|q|
// sin/cos of the octaves by exact angle doubling from one sincos of v.
template <bool FAST>
__device__ __forceinline__ void joint_values_q(float qx, float qy, float qz, float tau, float cut, float* x);

template <bool FAST>
__device__ __forceinline__ void joint_values(const float* sk, float px, float py, float pz,
                                             float tau, float cut, float* x) {
    float qx, qy, qz;
    bone_local(sk, px, py, pz, qx, qy, qz);
    joint_values_q<FAST>(qx, qy, qz, tau, cut, x);
}

template <bool FAST>
__device__ __forceinline__ void joint_values_q(float qx, float qy, float qz, float tau, float cut, float* x) {
    const float d2 = qx * qx + qy * qy + qz * qz;
    // FAST: `tau`/`cut` are the folded constants of cutoff_weight_fast; one v_rsq gives both
    // |q| and 1/max(|q|, 1e-12)
    const float rinv = FAST ? __builtin_amdgcn_rsqf(fmaxf(d2, 1e-24f)) : 0.0f;
    const float v = FAST ? d2 * rinv : sqrtf(d2);
    const float w = FAST ? cutoff_weight_fast(v, tau, cut) : cutoff_weight<false>(v, tau, cut);
    float s, c;
    if (FAST) {
        const float rev = v * 0.15915494309189535f;      // v_sin/v_cos take revolutions
        s = __builtin_amdgcn_sinf(rev);
        c = __builtin_amdgcn_cosf(rev);
    } else {
        sincosf(v, &s, &c);
    }
    x[0] = v * w;
    if (FAST) {
        // weighted doubling: S = w sin, C = w cos carried beside the plain pair, 5 ops per octave
        //   sin 2a = (2 s) c,  cos 2a = 1 - (2 s) s,  w sin 2a = (2 s) C,  w cos 2a = w - (2 s) S
        float S = s * w, C = c * w;
#pragma unroll
        for (int f = 0; f < LV; ++f) {
            x[1 + 2 * f] = S;
            x[2 + 2 * f] = C;
            if (f + 1 < LV) {
                const float t = s + s;
                const float sn = t * c, cn = fmaf(-t, s, 1.0f);
                const float Sn = t * C, Cn = fmaf(-t, S, w);
                s = sn; c = cn; S = Sn; C = Cn;
            }
        }
    } else {
#pragma unroll
        for (int f = 0; f < LV; ++f) {
            x[1 + 2 * f] = s * w;
            x[2 + 2 * f] = c * w;
            const float s2 = 2.0f * s * c;
            c = (c - s) * (c + s);
            s = s2;
        }
    }
    if (FAST) {
        x[15] = qx * rinv; x[16] = qy * rinv; x[17] = qz * rinv;
    } else {
        const float den = fmaxf(v, 1e-12f);
        x[15] = qx / den; x[16] = qy / den; x[17] = qz / den;
    }
}

// distance of the point to one joint (for the view-embedding cutoff weight)
template <bool FAST>
__device__ __forceinline__ float joint_dist(const float* sk, float px, float py, float pz) {
    float qx, qy, qz;
    bone_local(sk, px, py, pz, qx, qy, qz);
    return pg_sqrt<FAST>(qx * qx + qy * qy + qz * qz);
}

// Fill the per-ray LDS table for the rays [r0, r0+nr) this pass touches.
//  phase 1: skt rows, o, d, cam, frame code     phase 2 (after a barrier): view table
template <int NTHREADS>
__device__ __forceinline__ void ray_table_phase1(const EvalArgs& a, float* rtab, int r0, int nr) {
    const int tid = threadIdx.x;
    for (int idx = tid; idx < nr * 288; idx += NTHREADS) {
        const int rr = idx / 288, k = idx - rr * 288;
        const int j = k / 12, e = k - j * 12;
        rtab[rr * SLOT_FLOATS + SLOT_SKT + k] = a.skts[(long long)(r0 + rr) * a.pose_stride + j * 16 + e];
    }
    for (int idx = tid; idx < nr * 24; idx += NTHREADS) {
        const int rr = idx / 24, k = idx - rr * 24;
        float* slot = rtab + rr * SLOT_FLOATS;
        const long long ray = r0 + rr;
        if (k < 6) {
            slot[SLOT_O + k] = a.rays[ray * 11 + k];
        } else if (k == 6) {
            slot[SLOT_CAM] = a.cams ? a.cams[ray] : -1.0f;
        } else if (k >= 8 && a.codes) {
            const float cf = a.cams ? a.cams[ray] : -1.0f;
            const int ci = cf < 0.0f ? a.n_codes : min((int)cf, a.n_codes - 1);
            slot[SLOT_CODE + (k - 8)] = a.codes[ci * FC_CH + (k - 8)];
        }
    }
}

// View-direction table: e = normalize(R_j d) per joint (encoders.py:25-37,172-193), rows
// (e, sin e, cos e, sin 2e, cos 2e, sin 4e, cos 4e, sin 8e | cos 8e) in D-sequence order.
template <int NTHREADS, bool FAST>
__device__ __forceinline__ void ray_table_phase2(float* rtab, int nr) {
    for (int idx = threadIdx.x; idx < nr * J; idx += NTHREADS) {
        const int rr = idx / J, j = idx - rr * J;
        float* slot = rtab + rr * SLOT_FLOATS;
        const float* sk = slot + SLOT_SKT + j * 12;
        const float dx = slot[SLOT_D], dy = slot[SLOT_D + 1], dz = slot[SLOT_D + 2];
        float e[3];
        e[0] = fmaf(sk[2], dz, fmaf(sk[1], dy, sk[0] * dx));
        e[1] = fmaf(sk[6], dz, fmaf(sk[5], dy, sk[4] * dx));
        e[2] = fmaf(sk[10], dz, fmaf(sk[9], dy, sk[8] * dx));
        const float den = fmaxf(pg_sqrt<FAST>(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), 1e-12f);
        const int h = j / JH, jj = j - h * JH;
        float* tab = slot + SLOT_DTAB + h * DSEQ;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ev = pg_div<FAST>(e[c], den);
            float s, co;
            if (FAST) {
                const float rev = ev * 0.15915494309189535f;
                s = __builtin_amdgcn_sinf(rev);
                co = __builtin_amdgcn_cosf(rev);
            } else {
                sincosf(ev, &s, &co);
            }
            float* m = tab + (jj * 3 + c) * 8;
            m[0] = ev;
#pragma unroll
            for (int f = 0; f < LD; ++f) {
                m[1 + 2 * f] = s;
                if (f < LD - 1) m[2 + 2 * f] = co;
                else tab[DSEQ_MAIN + jj * 3 + c] = co;
                const float s2 = 2.0f * s * co;
                co = (co - s) * (co + s);
                s = s2;
            }
        }
        if (jj == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) tab[DSEQ_MAIN + JH * 3 + k] = 0.0f;
        }
    }
}

}  // namespace pgd
