// pg_evalc2.hip -- fused bone-relative embedding + NeRF MLP in COMPENSATED fp16 (PG_PREC_FP16C), second form: the
// OUT TILES of every layer are split over the waves and the activations live in LDS.  Shape T of pg_program.h.
//
// Replaces RayCaster.encode_inputs + run_network + NeRF.forward for one net (reference core/raycasters.py:476-577,
// core/networks/nerf.py:90-148, core/encoders.py, core/cutoff_embedder.py) on n*S points p = o + d*z, like pg_evalc.hip,
// with the same arithmetic: every product W x = (S-1) w1 x1 + w2 x2 as two fp16 MFMAs into one fp32 accumulator
// (pg_comp.h), hardware sin / cos per octave, Y / S split like a weight.
//
// Why a second form.  pg_evalc.hip gives a wave 32 points and all out channels: two accumulator sets of 128 registers,
// hence ONE wave per SIMD, and every MFMA needs its own 1-KiB weight fragment from the LDS ring -- its matrix pipe is
// busy half the time (profiles/r4_fp16c_*).  Here a workgroup of 8 waves (two per SIMD) carries 128 points through the
// net together:
//   * wave w owns out channels 32 w .. 32 w + 31 of every layer (two 16-row tiles of v_mfma_f32_16x16x32_f16) for ALL
//     128 points (8 column tiles): 64 accumulator registers, one set;
//   * its weights come straight from L2 into registers (global_load_dwordx4, one 4-KiB block per k-unit, prefetched a
//     k-unit ahead): every weight fragment feeds 8 MFMAs, nobody else needs it, no ring, no LDS-DMA, no chunk barriers;
//   * the layer's input lives in LDS as ready-made B fragments (the split fp16 pair of every activation: 1 KiB per point
//     and layer, 128 KiB): a B fragment read feeds two MFMAs (the wave's two tiles), i.e. LDS reads at half their peak;
//   * a layer ends with: barrier (everyone has read the input) -> ReLU, split, ds_write of the wave's 32 channels as
//     k-unit w of the next input -> barrier.  Two barriers per layer instead of eight chunk entries.
// The density input (K = 432) is generated into the same LDS region, at most 8 k-units at a time (the three units of
// directions and two limbs; a second round if more than two limbs are in range of the pass); the limb masks of
// pg_eval16r.hip are formed in the pass itself -- there is no stream to plan a pass ahead -- per column tile of 16 points.
// The per-ray part (bone-local rays (a, b), the view layer's direction part Y, the frame code) is formed by the
// workgroup from the rays themselves, so the kernel needs no per-ray records whatever the call: per-ray poses and
// frame codes included (template parameters PP, FC).
#include <type_traits>

#include "pg_comp.h"

namespace pgd {
namespace c2 {
namespace TT = pgp::T;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int NTHR2 = TT::NW * 64;
constexpr int NX = 5;                  // k-units of the density input in the first round through the region: directions + one limb
// ---- LDS carve-up (bytes) ----
constexpr int L_REG = 0;                                    // activations [k-unit 8][column tile 8][plane 2][lane] x 16 B
constexpr int UNIT_LDS = TT::NCT * 2 * 1024;                // 16 KiB: one k-unit of the pass's 128 points
constexpr int L_REG_BYTES = HU16 * UNIT_LDS;                // 128 KiB
constexpr int L_BIAS = L_REG + L_REG_BYTES;                 // 16-row bias tiles (pack_bias_s)
constexpr int L_CUT = L_BIAS + BIAS16_FLOATS * 4;           // 72 floats by joint slot: both embedders' constants, far^2
constexpr int L_ALPHA = L_CUT + 72 * 4;                     // compact A fragments of the alpha row / the rgb rows
constexpr int L_RGB = L_ALPHA + TT::SMALL_ALPHA;
constexpr int L_AB = L_RGB + TT::SMALL_RGB;                 // two buffers of (a, b) rows: this pass's rays / the next's
constexpr int L_ABSZ = TT::MAXR * REC_AB_BYTES;
constexpr int L_CMASK = L_AB + 2 * L_ABSZ;                  // limb masks of the 8 column tiles
constexpr int L_SK = L_CMASK + 64;                          // shared pose: bone rows by joint slot
constexpr int L_TOTAL = L_SK + J * 12 * 4;
static_assert(L_BIAS % 16 == 0 && L_CUT % 16 == 0 && L_ALPHA % 16 == 0 && L_RGB % 16 == 0 && L_AB % 16 == 0 && L_CMASK % 16 == 0 && L_SK % 16 == 0,
              "LDS alignment");
static_assert(L_TOTAL <= 160 * 1024, "LDS budget of one CU");
// images of the pass's tail, inside the activation region once the last trunk activation is dead
constexpr int L_YRAY = NTV16 * 2 * 1024;                    // Y of one ray: A fragments [out tile16 8][plane][lane (g, row)]
constexpr int L_Y = 0;
constexpr int L_WD = TT::MAXR * L_YRAY;                     // cutoff weights of the view embedder [column tile][first / second ray][plane][lane]
constexpr int L_TQ = L_WD + TT::NCT * 4 * 1024;             // T fragment pairs of a round of eight joint slots [slot][plane][lane]
constexpr int L_G = 0;                                      // view activations [k-unit 4][column tile 8][plane][lane] (behind a barrier)
static_assert(L_TQ + TT::NW * 2048 <= L_REG_BYTES, "tail images fit the region");

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- hand-issued loads (hipcc sees neither; every use sits behind a counted wait tied to the registers) ----
template <int OFF>
__device__ __forceinline__ void ds_rd(a128& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void wait_pair(a128& r0, a128& r1) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(N));
}
// the four A fragments of one k-unit of one wave: [tile t][plane], 4 KiB contiguous from `blk` (wave-uniform).
// The leading s_nop 4: `blk` may have just been written by a VALU instruction (v_readlane_b32 of a spilled SGPR,
// v_readfirstlane_b32), and a vector-memory instruction may not read such an SGPR in the next five issue slots -- hipcc
// pads its own instructions, not inline asm (the first build of this kernel faulted on a base with a stale upper half).
struct ASet { a128 f[4]; };
__device__ __forceinline__ void issue_a(ASet& s, unsigned lane16, const uint8_t* blk) {
    asm volatile("s_nop 4\n\t"
                 "global_load_dwordx4 %0, %4, %5 offset:0\n\t"
                 "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
                 "global_load_dwordx4 %3, %4, %5 offset:3072"
                 : "=&v"(s.f[0]), "=&v"(s.f[1]), "=&v"(s.f[2]), "=&v"(s.f[3]) : "v"(lane16), "s"(blk));
}
template <int N>
__device__ __forceinline__ void wait_a(ASet& s) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(s.f[0]), "+v"(s.f[1]), "+v"(s.f[2]), "+v"(s.f[3]) : "n"(N));
}
__device__ __forceinline__ f32x4 mma(const a128& a, const a128& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(VC, a), __builtin_bit_cast(VC, b), c, 0, 0, 0);
}
__device__ __forceinline__ a128 frag_x(const unsigned* p) {
    const a128 v = {p[0], p[1], p[2], p[3]};
    return v;
}
__device__ __forceinline__ void st128(uint8_t* p, const unsigned* v) {
    *reinterpret_cast<uint4*>(p) = make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ a128 ld128(const uint8_t* p) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const a128 r = {v.x, v.y, v.z, v.w};
    return r;
}

// One trunk layer on the activation in LDS: acc[t][c] += W[tile 2 w + t, :] H[:, column tile c], 8 k-units x 8 column
// tiles.  `wl` = this wave's block of k-unit 0; set A[0] already holds (in flight) that block.  The B pairs run through
// a ring of three register pairs, read two column steps ahead; the A block of the next k-unit is requested a whole k-unit
// (32 MFMAs) ahead.  NEXT: at the last k-unit the first block of the NEXT segment (`next`) is requested into A[0].
template <bool NEXT>
__device__ __forceinline__ void hidden_mma(f32x4 (&acc)[2][TT::NCT], ASet (&A)[2], const uint8_t* wl, const uint8_t* next,
                                           unsigned hb_lo, unsigned hb_hi, unsigned lane16) {
    a128 B[3][2];
    constexpr int NS = HU16 * TT::NCT;           // 64 steps (k-unit, column tile)
    auto issue_b = [&](auto ic) {
        constexpr int s = decltype(ic)::value, u = s / TT::NCT, c = s % TT::NCT;
        constexpr int off = (u & 3) * UNIT_LDS + c * 2048;
        ds_rd<off>(B[s % 3][0], u < 4 ? hb_lo : hb_hi);
        ds_rd<off + 1024>(B[s % 3][1], u < 4 ? hb_lo : hb_hi);
    };
    issue_b(std::integral_constant<int, 0>{});
    issue_b(std::integral_constant<int, 1>{});
    static_for<0, NS>([&](auto ic) {
        constexpr int s = decltype(ic)::value, u = s / TT::NCT, c = s % TT::NCT;
        if constexpr (c == 0) {
            if constexpr (u + 1 < HU16) {
                issue_a(A[(u + 1) & 1], lane16, wl + (u + 1) * (TT::NW * TT::KBLK));
                wait_a<4>(A[u & 1]);
            } else if constexpr (NEXT) {
                issue_a(A[0], lane16, next);
                wait_a<4>(A[u & 1]);
            } else {
                wait_a<0>(A[u & 1]);
            }
        }
        if constexpr (s + 2 < NS) issue_b(std::integral_constant<int, s + 2>{});
        constexpr int younger = s + 2 < NS ? 4 : 2 * (NS - 1 - s);
        wait_pair<younger>(B[s % 3][0], B[s % 3][1]);
        const ASet& a = A[u & 1];
        acc[0][c] = mma(a.f[0], B[s % 3][0], acc[0][c]);
        acc[1][c] = mma(a.f[2], B[s % 3][0], acc[1][c]);
        acc[0][c] = mma(a.f[1], B[s % 3][1], acc[0][c]);
        acc[1][c] = mma(a.f[3], B[s % 3][1], acc[1][c]);
    });
}

// Four value pairs -> their (ReLU'd) fp16 pairs in ONE asm block: the same instructions per value as conv_a / conv_b
// (pg_comp.h), interleaved over the eight values so that no instruction reads the result of the one right before it
// (a dependent VALU chain issues at a quarter of the rate; the layer-end conversion is 256 of these per wave).
// PIN: asm volatile -- the block keeps its place among the hand-issued LDS reads it is interleaved with.
#define C2_CONV4_RELU "v_max_f32 %8, 0, %16\n\tv_max_f32 %9, 0, %17\n\tv_max_f32 %10, 0, %18\n\tv_max_f32 %11, 0, %19\n\tv_max_f32 %12, 0, %20\n\tv_max_f32 %13, 0, %21\n\tv_max_f32 %14, 0, %22\n\tv_max_f32 %15, 0, %23\n\tv_cvt_pk_f16_f32 %0, %8, %9\n\tv_cvt_pk_f16_f32 %1, %10, %11\n\tv_cvt_pk_f16_f32 %2, %12, %13\n\tv_cvt_pk_f16_f32 %3, %14, %15\n\tv_fma_mix_f32 %8, %0, -1.0, %8 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %9, %0, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %10, %1, -1.0, %10 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %11, %1, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %12, %2, -1.0, %12 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %13, %2, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %14, %3, -1.0, %14 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %15, %3, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %8, %8, %24, %0 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %9, %9, %24, %0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %10, %10, %24, %1 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %11, %11, %24, %1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %12, %12, %24, %2 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %13, %13, %24, %2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %14, %14, %24, %3 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %15, %15, %24, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_cvt_pk_f16_f32 %4, %8, %9\n\tv_cvt_pk_f16_f32 %5, %10, %11\n\tv_cvt_pk_f16_f32 %6, %12, %13\n\tv_cvt_pk_f16_f32 %7, %14, %15"
#define C2_CONV4_PLAIN "v_cvt_pk_f16_f32 %0, %16, %17\n\tv_cvt_pk_f16_f32 %1, %18, %19\n\tv_cvt_pk_f16_f32 %2, %20, %21\n\tv_cvt_pk_f16_f32 %3, %22, %23\n\tv_fma_mix_f32 %8, %0, -1.0, %16 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %9, %0, -1.0, %17 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %10, %1, -1.0, %18 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %11, %1, -1.0, %19 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %12, %2, -1.0, %20 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %13, %2, -1.0, %21 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %14, %3, -1.0, %22 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %15, %3, -1.0, %23 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %8, %8, %24, %0 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %9, %9, %24, %0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %10, %10, %24, %1 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %11, %11, %24, %1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %12, %12, %24, %2 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %13, %13, %24, %2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %14, %14, %24, %3 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %15, %15, %24, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_cvt_pk_f16_f32 %4, %8, %9\n\tv_cvt_pk_f16_f32 %5, %10, %11\n\tv_cvt_pk_f16_f32 %6, %12, %13\n\tv_cvt_pk_f16_f32 %7, %14, %15"
#define C2_CONV4_OPS                                                                                                        \
    : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]), "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]),                 \
      "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)                                   \
    : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "s"(s)
template <bool RELU, bool PIN = false>
__device__ __forceinline__ void conv4(const float (&v)[8], unsigned (&h)[4], unsigned (&x)[4], float s) {
    float t0, t1, t2, t3, t4, t5, t6, t7;
    if constexpr (PIN) {
        if constexpr (RELU) asm volatile(C2_CONV4_RELU C2_CONV4_OPS);
        else asm volatile(C2_CONV4_PLAIN C2_CONV4_OPS);
    } else {
        if constexpr (RELU) asm(C2_CONV4_RELU C2_CONV4_OPS);
        else asm(C2_CONV4_PLAIN C2_CONV4_OPS);
    }
}

// the wave's 32 channels of all 128 points -> k-unit `wave` of the next layer's input: lane (g, col) of column tile c holds
// rows 4 g .. 4 g + 3 of its two tiles = values 0..3 / 4..7 of the unit (hseq16_channel)
template <bool RELU>
__device__ __forceinline__ void conv_write(const f32x4 (&acc)[2][TT::NCT], uint8_t* hw, float s129) {
    static_for<0, TT::NCT>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        unsigned h[4], x[4];
        const float v[8] = {acc[0][c][0], acc[0][c][1], acc[0][c][2], acc[0][c][3], acc[1][c][0], acc[1][c][1], acc[1][c][2], acc[1][c][3]};
        conv4<RELU>(v, h, x, s129);
        st128(hw + c * 2048, h);
        st128(hw + c * 2048 + 1024, x);
    });
}

// column tiles C0 .. C0 + N - 1 of the wave's 32 channels -> its k-unit of the next layer's input
template <int C0, int N, bool PIN>
__device__ __forceinline__ void conv_cols(const f32x4 (&acc)[2][TT::NCT], uint8_t* hw, float s129) {
    static_for<C0, C0 + N>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        unsigned h[4], x[4];
        const float v[8] = {acc[0][c][0], acc[0][c][1], acc[0][c][2], acc[0][c][3], acc[1][c][0], acc[1][c][1], acc[1][c][2], acc[1][c][3]};
        conv4<true, PIN>(v, h, x, s129);
        st128(hw + c * 2048, h);
        st128(hw + c * 2048 + 1024, x);
    });
}

// An MFMA result may not be read by a VALU instruction in the next 7 issue slots (4-pass XDL write -> VALU read on
// gfx950); hipcc pads its own instructions, not the inline asm of the conversions: the accumulators of column tiles
// C0 .. C0 + N - 1 pass through here in front of a conversion that is not separated from their last MFMA by other work
template <int C0, int N>
__device__ __forceinline__ void settle(f32x4 (&acc)[2][TT::NCT]) {
    if constexpr (N > 0) {
        asm volatile("s_nop 7" : "+v"(acc[0][C0]), "+v"(acc[1][C0]));
        settle<C0 + 1, N - 1>(acc);
    }
}
// one value pair -> its ReLU'd fp16 pair, pinned (asm volatile) between the hand-issued reads of the MFMA steps: the
// instructions of conv_a<true> + conv_b (pg_comp.h)
// Every register the block WRITES is a "+v" operand that lives through the whole run of MFMA steps (CState): a fresh
// temporary could be given a register that an MFMA of the step before has just written (hipcc renames the accumulators
// freely), and a VALU write to it inside the same 7 slots is the same hazard (tools/audit_asm_hazards.py).
struct CState { unsigned h[4], x[4]; float t0, t1; };
__device__ __forceinline__ void cstate_init(CState& cs) {
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "=v"(cs.t0), "=v"(cs.t1));
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "=v"(cs.h[j]), "=v"(cs.x[j]));
}
__device__ __forceinline__ void conv_pair_pin(float a, float b, unsigned& h, unsigned& x, float& t0, float& t1, float s) {
    asm volatile("v_max_f32 %2, 0, %4\n\tv_max_f32 %3, 0, %5\n\tv_cvt_pk_f16_f32 %0, %2, %3\n\t"
                 "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %2, %2, %6, %0 op_sel_hi:[0,0,1]\n\t"
                 "v_fma_mix_f32 %3, %3, %6, %0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                 "v_cvt_pk_f16_f32 %1, %2, %3"
                 : "+v"(h), "+v"(x), "+v"(t0), "+v"(t1) : "v"(a), "v"(b), "s"(s));
}
// pair J (0..3) of column tile C of the wave's two tiles: values 2 J, 2 J + 1 of the k-unit; the tile's two fragments
// are stored behind its last pair
template <int C, int J>
__device__ __forceinline__ void conv_step(const f32x4 (&acc)[2][TT::NCT], CState& cs, uint8_t* hw, float s129) {
    conv_pair_pin(acc[J >> 1][C][2 * (J & 1)], acc[J >> 1][C][2 * (J & 1) + 1], cs.h[J], cs.x[J], cs.t0, cs.t1, s129);
    if constexpr (J == 3) {
        st128(hw + C * 2048, cs.h);
        st128(hw + C * 2048 + 1024, cs.x);
    }
}

// ---- (-DPG_C2_SPLIT, measured and not the default) a trunk layer on the two point HALVES of the pass (alpha = column tiles
// 0..3, beta = 4..7).  Result on MI355X (profiles/r5_c2_split_*.txt): parity-green, and SLOWER than the plain loop -- the
// conversion rides in 16 of a layer's 64 MFMA steps, where its 8 VALU instructions per step (4 cycles each, two waves) need
// exactly the 64 issue cycles the step's 8 MFMAs leave free; the two waves of a SIMD leave every barrier in lockstep, so in
// practice both sit in their VALU block while the matrix pipe idles (65 instead of 28 ticks per MFMA in a0 a1; the layers take
// 88.6 k ticks per pass instead of 81.4 k); delaying waves 4..7 by half a step (PG_C2_STAGGER) changes 1 %. ----
// The layer-end conversion (ReLU, split, 128 KiB of ds_write per layer) is bound by the LDS store path, ~1.7 k cycles in
// which no MFMA runs if every wave converts at the same time (profiles/r5_c2_stamps*.txt: 10 % of a pass).  So a layer
// runs as eight blocks of 32 MFMAs (half, k-pair q = k-units 2 q, 2 q + 1) in the order
//      a0 a1 | b0 a2 b1 a3 | b2 b3          (| = workgroup barrier)
// and the conversion of a half rides in the other half's blocks: beta's output of the PREVIOUS layer in a0 a1 (before
// b0 reads it), alpha's output of THIS layer in b2 b3 (before the next layer's a0 reads it).  A k-pair's weights (two
// 4-KiB blocks: 32 registers) are requested once, a block or two ahead, and kept until the second half has used them: at
// most three pairs live (96 registers); the pairs of the next layer's a0 / a1 are requested in a3 / b3 and RETIRED before
// the layer ends, so that nothing hand-issued is in flight across the layer loop's back edge.
constexpr int sb_h(int b) { return (0xD4 >> b) & 1; }                               // half of block b: 0 0 1 0 1 0 1 1
constexpr int sb_q(int b) { constexpr int q[8] = {0, 1, 0, 2, 1, 3, 2, 3}; return q[b]; }
struct WPair { ASet k[2]; };            // the weights of one k-pair of one wave
__device__ __forceinline__ void issue_pair(WPair& w, unsigned lane16, const uint8_t* wl, int q) {
    issue_a(w.k[0], lane16, wl + (2 * q) * (TT::NW * TT::KBLK));
    issue_a(w.k[1], lane16, wl + (2 * q + 1) * (TT::NW * TT::KBLK));
}
template <int N>
__device__ __forceinline__ void wait_wpair(WPair& w) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(w.k[0].f[0]), "+v"(w.k[0].f[1]), "+v"(w.k[0].f[2]), "+v"(w.k[0].f[3]),
                                          "+v"(w.k[1].f[0]), "+v"(w.k[1].f[1]), "+v"(w.k[1].f[2]), "+v"(w.k[1].f[3]) : "n"(N));
}
// blocks B0 .. B0 + NB - 1 of the schedule as ONE pipelined run of B-fragment reads (three register pairs, two steps
// ahead); hook(step) runs in front of every step (4 MFMAs): weight requests / retires, conversion pieces, the barrier
template <int B0, int NB, typename HOOK>
__device__ __forceinline__ void split_span(f32x4 (&acc)[2][TT::NCT], WPair (&W)[4], unsigned hb_lo, bool stagger, HOOK&& hook) {
    a128 B[3][2];
    const unsigned hb_hi = hb_lo + 4 * UNIT_LDS;      // (k-units 4..7: the DS offset field holds 16 bits)
    constexpr int NS = NB * 8;
    auto issue_b = [&](auto ic) {
        constexpr int s = decltype(ic)::value, b = B0 + s / 8, t = s % 8;
        constexpr int u = 2 * sb_q(b) + t / 4, c = 4 * sb_h(b) + t % 4;
        constexpr int off = (u & 3) * UNIT_LDS + c * 2048;
        ds_rd<off>(B[s % 3][0], u < 4 ? hb_lo : hb_hi);
        ds_rd<off + 1024>(B[s % 3][1], u < 4 ? hb_lo : hb_hi);
    };
#if defined(PG_C2_STAGGER)
    // the two waves of a SIMD leave a barrier in lockstep: both in their conversion block while the matrix pipe idles, then
    // both with MFMAs queued.  Half a step of delay for the second-dispatched half puts one's VALU block beside the other's MFMAs.
    if (stagger) __builtin_amdgcn_s_sleep(PG_C2_STAGGER);
#endif
    issue_b(std::integral_constant<int, 0>{});
    issue_b(std::integral_constant<int, 1>{});
    static_for<0, NS>([&](auto ic) {
        constexpr int s = decltype(ic)::value, b = B0 + s / 8, t = s % 8;
        constexpr int q = sb_q(b), c = 4 * sb_h(b) + t % 4;
        hook(ic);
        if constexpr (s + 2 < NS) issue_b(std::integral_constant<int, s + 2>{});
        constexpr int younger = s + 2 < NS ? 4 : 2 * (NS - 1 - s);
        wait_pair<younger>(B[s % 3][0], B[s % 3][1]);
        const ASet& a = W[q].k[t / 4];
        acc[0][c] = mma(a.f[0], B[s % 3][0], acc[0][c]);
        acc[1][c] = mma(a.f[2], B[s % 3][0], acc[1][c]);
        acc[0][c] = mma(a.f[1], B[s % 3][1], acc[0][c]);
        acc[1][c] = mma(a.f[3], B[s % 3][1], acc[1][c]);
    });
}

__device__ __forceinline__ void zero4(unsigned (&z)[4]) {
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(z[0]), "=v"(z[1]), "=v"(z[2]), "=v"(z[3]));
}
// the two halves of conv_write: the fragments formed in registers (VALU only) / stored
struct HFrags { unsigned h[TT::NCT][4], x[TT::NCT][4]; };
template <bool RELU>
__device__ __forceinline__ void conv_form(const f32x4 (&acc)[2][TT::NCT], HFrags& f, float s129) {
    static_for<0, TT::NCT>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        const float v[8] = {acc[0][c][0], acc[0][c][1], acc[0][c][2], acc[0][c][3], acc[1][c][0], acc[1][c][1], acc[1][c][2], acc[1][c][3]};
        conv4<RELU>(v, f.h[c], f.x[c], s129);
    });
}
__device__ __forceinline__ void conv_store(const HFrags& f, uint8_t* hw) {
    static_for<0, TT::NCT>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        st128(hw + c * 2048, f.h[c]);
        st128(hw + c * 2048 + 1024, f.x[c]);
    });
}

// pins the order: every load named here has been issued (and, as far as hipcc knows, consumed) before anything behind it
__device__ __forceinline__ void pin4(a128& a, a128& b, a128& c, a128& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }

__device__ __forceinline__ f32x4 bias_tile(const float* bias, int tile, int g) {
    const float4 b = *reinterpret_cast<const float4*>(bias + tile * 16 + 4 * g);
    const f32x4 r = {b.x, b.y, b.z, b.w};
    return r;
}

// one (ray, joint slot) row of the (a, b) table from values (ab_row of pg_eval16_common.h reads them through pointers)
__device__ __forceinline__ void ab_row_v(const float* sk, const float* ray, float z0, float z1, uint8_t* dst) {
    ab_row(sk, ray, z0, z1, reinterpret_cast<float4*>(dst));
}

#if defined(PG_STAMPS)
#define C2_STAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamp_acc[k] += t_ - stamp_prev; stamp_prev = t_; } while (0)
#else
#define C2_STAMP(k) do {} while (0)
#endif

// FC: frame codes (view layer K = 920: the code rides as pseudo joint slot 24 of the view layer's second stage)
// PP: per-ray poses (a.pose_stride != 0): the bone rows of a pass's rays are read from a.skts instead of the LDS table
template <bool FC, bool PP>
__global__ __launch_bounds__(NTHR2, 2) void evalc2_kernel(const EvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* bias = reinterpret_cast<float*>(smem + L_BIAS);
    float* cut = reinterpret_cast<float*>(smem + L_CUT);
    int* cmaskp = reinterpret_cast<int*>(smem + L_CMASK);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const uint8_t* wbase = a.wstream;

    // ---- tables ----
    for (int i = tid; i < BIAS16_FLOATS; i += NTHR2) bias[i] = a.bias[i];
    const float tlv = a.tau_v * 1.4426950408889634f, tld = a.tau_d * 1.4426950408889634f;
    if (tid < 48) cut[tid] = -a.cutoff[(tid < J ? 0 : J) + slot_joint_dev(tid < J ? tid : tid - J)] * (tid < J ? tlv : tld);
    else if (tid < 72) {
        const int jt = slot_joint_dev(tid - 48);
        const float far = fmaxf(a.cutoff[jt] + 24.0f / tlv, a.cutoff[J + jt] + 24.0f / tld);
        cut[tid] = far * far;
    }
    for (int i = tid; i < (TT::SMALL_ALPHA + TT::SMALL_RGB) / 16; i += NTHR2)
        reinterpret_cast<uint4*>(smem + L_ALPHA)[i] = reinterpret_cast<const uint4*>(wbase + TT::OFF_SMALL)[i];
    if (!PP)
        for (int i = tid; i < J * 12; i += NTHR2) reinterpret_cast<float*>(smem + L_SK)[i] = a.skts[slot_joint_dev(i / 12) * 16 + i % 12];
    float s129 = (float)COMP_S;
    asm volatile("" : "+s"(s129));

    // ---- ray bookkeeping (no division per pass): the pass's first point is sample off0 of ray r0 ----
    const long long step = (long long)TT::PTS * gridDim.x;
    const int dq = __builtin_amdgcn_readfirstlane((int)(step / a.S)), dr = __builtin_amdgcn_readfirstlane((int)(step % a.S));
    long long p0 = (long long)blockIdx.x * TT::PTS;
    int r0 = __builtin_amdgcn_readfirstlane((int)(p0 / a.S));
    int off0 = __builtin_amdgcn_readfirstlane((int)(p0 - (long long)r0 * a.S));
    const int S1 = a.S, S2 = 2 * a.S, S3 = 3 * a.S, S4 = 4 * a.S;
    auto ray_of = [&](int t) { return (t >= S1) + (t >= S2) + (t >= S3) + (t >= S4); };
    // one (ray k of the pass that starts at ray rr, joint slot sl) row per item
    auto load_ray_item = [&](int item, int rr, float* ray6, float& z0, float& z1, float* sk12) {
        const int k = item / J, sl = item - k * J;
        const long long ray = min((long long)rr + k, (long long)a.n_rays - 1);
#pragma unroll
        for (int e = 0; e < 6; ++e) ray6[e] = a.rays[ray * 11 + e];
        z0 = a.z[ray * a.S];
        z1 = a.z[ray * a.S + a.S - 1];
        if (PP) {
            const float4* skp = reinterpret_cast<const float4*>(a.skts + ray * a.pose_stride + slot_joint_dev(sl) * 16);
            const float4 r0_ = skp[0], r1_ = skp[1], r2_ = skp[2];
            sk12[0] = r0_.x; sk12[1] = r0_.y; sk12[2] = r0_.z; sk12[3] = r0_.w;
            sk12[4] = r1_.x; sk12[5] = r1_.y; sk12[6] = r1_.z; sk12[7] = r1_.w;
            sk12[8] = r2_.x; sk12[9] = r2_.y; sk12[10] = r2_.z; sk12[11] = r2_.w;
        }
    };
    auto store_ray_item = [&](int item, int buf, const float* ray6, float z0, float z1, const float* sk12) {
        const int k = item / J, sl = item - k * J;
        const float* sk = PP ? sk12 : reinterpret_cast<const float*>(smem + L_SK) + sl * 12;
        ab_row_v(sk, ray6, z0, z1, smem + L_AB + buf * L_ABSZ + k * REC_AB_BYTES + sl * 32);
    };
    lds_barrier();                              // the tables are in LDS
    if (tid < TT::MAXR * J) {                   // (a, b) of the first pass's rays
        float ray6[6], z0, z1, sk12[12];
        load_ray_item(tid, r0, ray6, z0, z1, sk12);
        store_ray_item(tid, 0, ray6, z0, z1, sk12);
    }
    int abuf = 0;
    float nx_z = 0.0f;
    if ((int)blockIdx.x < a.n_iters) nx_z = a.z[min(p0 + 16 * wave + col, a.n_points - 1)];
    lds_barrier();

    // dbg_stage 97: passes, limbs left out of whole passes (of 6 per pass), (column tile, limb) pairs left out (of 48 per pass),
    // summed over the launch into a.dbg[0..2] (unsigned)
    const bool count_skips = a.dbg != nullptr && a.dbg_stage == 97;
    unsigned n_cnt[3] = {0u, 0u, 0u};
#if defined(PG_C2_YOUNG_PRIO)
    if (wave >= 4) __builtin_amdgcn_s_setprio(PG_C2_YOUNG_PRIO);       // (experiment: the second-dispatched half loses every issue arbitration)
#endif
    // per-wave constants of the pass loop
    const uint8_t* wx_wave = wbase + wave * TT::KBLK;       // + section + k-unit * NW * KBLK
    WPair W[4];
    ASet X[NX];

#if defined(PG_STAMPS)
    unsigned long long stamp_acc[12] = {}, stamp_prev = 0;
#endif
    for (int it = blockIdx.x; it < a.n_iters; it += gridDim.x) {
#if defined(PG_STAMPS)
        { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamp_prev = t_; }
#endif
        int lane_p;                                 // the lane index, formed per pass: addresses derived from it are not hoisted out of
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_p));      // the pass loop and spilled, and no copy of it lives through the pass
        // likewise the weight bases: with the plain pointers hipcc forms every section / k-unit address of the pass ahead of the pass
        // loop -- dozens of 64-bit constants, 170 spilled SGPRs and pointer pairs parked in VGPRs
        const uint8_t* wx_p = wx_wave;
        const uint8_t* wb_p = wbase;
        asm volatile("" : "+s"(wx_p), "+s"(wb_p));
        const int g_p = lane_p >> 4, col_p = lane_p & 15;
        const unsigned lane16 = (unsigned)lane_p * 16u;
        const unsigned hb_lo = lds0 + L_REG + lane16;
        uint8_t* const lane_reg = smem + L_REG + lane_p * 16;        // this lane's 16 bytes of fragment 0 of the region
        const int last = (int)min((long long)TT::PTS - 1, a.n_points - 1 - p0);       // wave-uniform
        const int nrm1 = ray_of(off0 + last);
        const int i_pt = 16 * wave + col_p;
        const int myr = min(ray_of(off0 + i_pt), nrm1);
        const float zz = nx_z;
        const float* abp = opaque_ptr(reinterpret_cast<const float*>(smem + L_AB + abuf * L_ABSZ + myr * REC_AB_BYTES) + JG * g_p * 8);
        int off0n = off0 + dr, r0n = r0 + dq;
        if (off0n >= a.S) { off0n -= a.S; ++r0n; }
        auto local = [&](int jj, float& qx, float& qy, float& qz) {
            const float4 lo = *reinterpret_cast<const float4*>(abp + jj * 8);
            const float4 hi = *reinterpret_cast<const float4*>(abp + jj * 8 + 4);
            qx = fmaf(zz, hi.x, lo.x); qy = fmaf(zz, hi.y, lo.y); qz = fmaf(zz, hi.z, lo.z);
        };

        // ---- limbs out of cutoff range of the 16 points of this wave's column tile (pg_eval16r.hip explains the test) ----
        int wm = 0;
#if !defined(PG_NO_FAR_SKIP)
        {
            const float* far2 = cut + 2 * J + JG * g_p;
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) {
                float qx, qy, qz;
                local(jj, qx, qy, qz);
                if (__builtin_amdgcn_ballot_w64(qx * qx + qy * qy + qz * qz < far2[jj]) == 0ull) wm |= 1 << jj;
            }
            wm = __builtin_amdgcn_readfirstlane(a.far_skip ? wm : 0);
        }
#endif
        if (lane_p == 0) cmaskp[wave] = wm;
        // the NEXT pass's rays: requested here, turned into its (a, b) rows behind layer 0 (15 lanes of every wave)
        float nray[6], nz0, nz1, nsk[12];
        const int nitem = wave * 15 + min(lane_p, 14);
        load_ray_item(nitem, r0n, nray, nz0, nz1, nsk);
        lds_barrier();                                                  // B1: masks visible; the previous pass is over
        int cm[TT::NCT];
#pragma unroll
        for (int c = 0; c < TT::NCT; ++c) cm[c] = __builtin_amdgcn_readfirstlane(cmaskp[c]);
        int gmask = 63;
#pragma unroll
        for (int c = 0; c < TT::NCT; ++c) gmask &= cm[c];
        if (count_skips && tid == 0) {       // (measurement aid, dbg_stage 97: what the limb masks leave out)
            n_cnt[0] += 1;
            n_cnt[1] += __builtin_popcount(gmask);
#pragma unroll
            for (int c = 0; c < TT::NCT; ++c) n_cnt[2] += __builtin_popcount(cm[c]);
        }
        // the limbs in range of the pass, in order: nibble k of `ll`
        int nl = 0;
        unsigned ll = 0;
#pragma unroll
        for (int jj = 0; jj < JG; ++jj)
            if (!((gmask >> jj) & 1)) { ll |= (unsigned)jj << (4 * nl); ++nl; }
        const int nu = 3 + 2 * nl;              // k-units of the density input this pass: directions, then two per limb
        // unit i of the pass: k-unit `xu` of the X16 sequence (weights), slot of the region, limb (-1: directions)
        auto unit_desc = [&](int i, int& xu, int& slot, int& jj) {
            if (i < 3) { xu = XV16 + i; slot = i; jj = -1; }
            else {
                jj = (int)((ll >> (4 * ((i - 3) >> 1))) & 15u);
                xu = 2 * jj + ((i - 3) & 1);
                slot = i < NX ? i : (i - NX) & 3;
            }
        };
        // the wave's column tile of the density input, round 0: directions and limb 0 of the pass; round r: limbs 2 r - 1, 2 r
        auto xgen = [&](int round) {
            uint8_t* xw = lane_reg + wave * 2048;
            if (round == 0) {
#pragma unroll
                for (int pr = 0; pr < JG / 2; ++pr) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        float qx, qy, qz;
                        local(2 * pr + k, qx, qy, qz);
                        const float rinv = __builtin_amdgcn_rsqf(fmaxf(qx * qx + qy * qy + qz * qz, 1e-24f));
                        v[3 * k] = qx * rinv; v[3 * k + 1] = qy * rinv; v[3 * k + 2] = qz * rinv;
                    }
                    v[6] = v[7] = 0.0f;
                    const FragC f = frag_of(v, s129);
                    st128(xw + pr * UNIT_LDS, f.x1);
                    st128(xw + pr * UNIT_LDS + 1024, f.x2);
                }
            }
            const int k0 = round == 0 ? 0 : 2 * round - 1, k1 = min(nl, 2 * round + 1);
#pragma unroll 1
            for (int k = k0; k < k1; ++k) {
                const int jj = (int)((ll >> (4 * k)) & 15u);
                if ((wm >> jj) & 1) continue;               // out of range of this column tile: nobody reads its fragments
                float x[18], qx, qy, qz;
                local(jj, qx, qy, qz);
                joint_values_c(qx, qy, qz, tlv, cut[JG * g_p + jj], x);
                x[15] = 0.0f;
                const FragC f0 = frag_of(x, s129), f1 = frag_of(x + 8, s129);
                uint8_t* d = xw + (round == 0 ? 3 + 2 * k : 2 * (k - k0)) * UNIT_LDS;
                st128(d, f0.x1); st128(d + 1024, f0.x2);
                st128(d + UNIT_LDS, f1.x1); st128(d + UNIT_LDS + 1024, f1.x2);
            }
        };
        auto x_block = [&](int sec_off, int xu) { return wx_p + sec_off + xu * (TT::NW * TT::KBLK); };
        // acc += W[:, x-columns of the unit] x for the column tiles in range of the unit's limb
        auto unit_mma = [&](f32x4 (&acc)[2][TT::NCT], const ASet& s, int slot, int jj) {
            const uint8_t* xb = lane_reg + slot * UNIT_LDS;
            static_for<0, 2>([&](auto hc) {                 // four column tiles at a time: their fragments in flight before the first MFMA
                constexpr int c0 = 4 * decltype(hc)::value;
                a128 b[4][2];
                static_for<0, 4>([&](auto ic) {
                    constexpr int c = c0 + decltype(ic)::value;
                    if (jj < 0 || !((cm[c] >> jj) & 1)) { b[c - c0][0] = ld128(xb + c * 2048); b[c - c0][1] = ld128(xb + c * 2048 + 1024); }
                });
                static_for<0, 4>([&](auto ic) {
                    constexpr int c = c0 + decltype(ic)::value;
                    if (jj < 0 || !((cm[c] >> jj) & 1)) {
                        acc[0][c] = mma(s.f[0], b[c - c0][0], acc[0][c]);
                        acc[1][c] = mma(s.f[2], b[c - c0][0], acc[1][c]);
                        acc[0][c] = mma(s.f[1], b[c - c0][1], acc[0][c]);
                        acc[1][c] = mma(s.f[3], b[c - c0][1], acc[1][c]);
                    }
                });
            });
        };
        // The density input against the wave's two tiles, in rounds through the region: round 0 = the three units of
        // directions and the pass's first limb (NX = 5 units), every further round two more limbs (4 units).  All weight
        // blocks of a round are requested up front -- they land while the fragments are generated -- and retired by ONE
        // wait behind the barrier: no hand-issued load is in flight across a loop back-edge or a branch that redefines
        // it (hipcc copies such registers at the merge; tools/audit_asm_loads.py).
        auto x_round = [&](f32x4 (&acc)[2][TT::NCT], int sec_off, int round) {
            const int i0 = round == 0 ? 0 : 1 + 4 * round;          // first unit of the round
            static_for<0, NX>([&](auto ic) {
                constexpr int j = decltype(ic)::value;
                if (j < (round == 0 ? NX : 4) && i0 + j < nu) {
                    int xu, slot, jj;
                    unit_desc(i0 + j, xu, slot, jj);
                    issue_a(X[j], lane16, x_block(sec_off, xu));
                }
            });
            xgen(round);
            lds_barrier();
            static_for<0, NX>([&](auto ic) { wait_a<0>(X[decltype(ic)::value]); });
            static_for<0, NX>([&](auto ic) {
                constexpr int j = decltype(ic)::value;
                if (j < (round == 0 ? NX : 4) && i0 + j < nu) {
                    int xu, slot, jj;
                    unit_desc(i0 + j, xu, slot, jj);
                    unit_mma(acc, X[j], j, jj);
                }
            });
        };
        auto x_phase = [&](f32x4 (&acc)[2][TT::NCT], int sec_off) {
            x_round(acc, sec_off, 0);
#pragma unroll 1
            for (int round = 1; 1 + 4 * round < nu; ++round) {
                lds_barrier();                                       // everyone is done with the previous round's fragments
                x_round(acc, sec_off, round);
            }
        };
        C2_STAMP(0);

        // ---- layer 0: K = 432 generated into the region ----
        f32x4 acc[2][TT::NCT];
        {
            const f32x4 b0 = bias_tile(bias, BS_LAYER0 + 2 * wave, g_p), b1 = bias_tile(bias, BS_LAYER0 + 2 * wave + 1, g_p);
#pragma unroll
            for (int c = 0; c < TT::NCT; ++c) { acc[0][c] = b0; acc[1][c] = b1; }
        }
        x_phase(acc, TT::OFF_X0);
        C2_STAMP(1);
        // the NEXT pass's (a, b) rows (requested at the top of the pass)
        if (lane_p < 15) store_ray_item(nitem, abuf ^ 1, nray, nz0, nz1, nsk);
#if defined(PG_C2_SPLIT)
        // ---- layers 1..7 on the two point halves (split_span above): layers 1..5, the skip connection's x part, layers 6, 7 ----
        uint8_t* const hw = lane_reg + wave * UNIT_LDS;                 // this wave's k-unit of the next input
        auto layer_wl = [&](int hs) { return wx_p + TT::OFF_HID(0) + hs * TT::SEC_H + (hs >= 5 ? TT::SEC_X : 0); };
        // the first two k-pairs of layer hs + 1 requested, alpha's half of the previous output written (nobody may still
        // read the region's old content: barrier first), the pairs retired -- in front of a run of layers
        auto run_prologue = [&](int hs) {
            issue_pair(W[0], lane16, layer_wl(hs), 0);
            issue_pair(W[1], lane16, layer_wl(hs), 1);
            lds_barrier();                                              // everyone is done reading the x fragments
            settle<0, 4>(acc);
            conv_cols<0, 4, false>(acc, hw, s129);
            wait_wpair<0>(W[0]);
            wait_wpair<0>(W[1]);
            lds_barrier();                                              // alpha's input is complete
        };
        // one layer: on entry W[0], W[1] hold its k-pairs 0, 1, alpha's input is in LDS and beta's previous output still in
        // acc[.][4..7].  CONV_A = this layer's alpha output is converted in b2 b3 (not for layer 5: its x part follows),
        // MORE = another layer follows directly (its k-pairs 0, 1 are requested in a3 / b3): compile-time, so that no
        // hand-issued load is defined on one side of a branch only (hipcc copies such registers at the merge)
        auto layer_s = [&](int hs, auto conv_alpha_c, auto more_c) {
            constexpr bool CONV_A = decltype(conv_alpha_c)::value, MORE = decltype(more_c)::value;
            const uint8_t* wl = layer_wl(hs);
            CState cs;                                                  // the fragments of the column tile being converted
            cstate_init(cs);
            {
                const f32x4 b0 = bias_tile(bias, BS_LAYER0 + (hs + 1) * NT16 + 2 * wave, g_p);
                const f32x4 b1 = bias_tile(bias, BS_LAYER0 + (hs + 1) * NT16 + 2 * wave + 1, g_p);
#pragma unroll
                for (int c = 0; c < 4; ++c) { acc[0][c] = b0; acc[1][c] = b1; }
            }
            split_span<0, 2>(acc, W, hb_lo, wave >= 4, [&](auto ic) {
                constexpr int s = decltype(ic)::value;
                if constexpr (s == 8) issue_pair(W[2], lane16, wl, 2);
                conv_step<4 + s / 4, s % 4>(acc, cs, hw, s129);                          // beta's previous output: one value pair per step
            });
            C2_STAMP(3);
            lds_barrier();                                              // X: beta's input is complete
            {   // (the bias tiles are read again rather than kept: eight registers through a0 a1)
                int wv = wave;
                asm volatile("" : "+s"(wv));
                const f32x4 b0 = bias_tile(bias, BS_LAYER0 + (hs + 1) * NT16 + 2 * wv, g_p);
                const f32x4 b1 = bias_tile(bias, BS_LAYER0 + (hs + 1) * NT16 + 2 * wv + 1, g_p);
#pragma unroll
                for (int c = 4; c < TT::NCT; ++c) { acc[0][c] = b0; acc[1][c] = b1; }
            }
            C2_STAMP(4);
            split_span<2, 6>(acc, W, hb_lo, wave >= 4, [&](auto ic) {
                constexpr int s = decltype(ic)::value;
                if constexpr (s == 8) { issue_pair(W[3], lane16, wl, 3); wait_wpair<8>(W[2]); }            // block a2
                if constexpr (s == 24) {                                                                    // block a3
                    if constexpr (MORE) { issue_pair(W[0], lane16, wl + TT::SEC_H, 0); wait_wpair<8>(W[3]); }
                    else wait_wpair<0>(W[3]);
                }
                if constexpr (s == 32) __builtin_amdgcn_s_barrier();    // Y: everyone is done reading alpha's input (its last
                                                                        // reads fed the MFMAs of a3; those in flight are beta's)
                if constexpr (s == 40 && MORE) issue_pair(W[1], lane16, wl + TT::SEC_H, 1);               // block b3
                if constexpr (CONV_A && s >= 32) conv_step<(s - 32) / 4, s % 4>(acc, cs, hw, s129);      // alpha's output: one value pair per step
            });
            C2_STAMP(6);
            if constexpr (MORE) {
                wait_wpair<0>(W[0]);                                    // (requested blocks ago: landed; nothing in flight at the back edge)
                wait_wpair<0>(W[1]);
            }
            lds_barrier();                                              // Z: alpha's next input is complete, beta's input is free
            C2_STAMP(5);
        };
        C2_STAMP(2);
        run_prologue(0);
        constexpr std::true_type yes{};
        constexpr std::false_type no{};
#pragma unroll 1
        for (int i = 0; i < 5; ++i) {           // layers 1..4, 6 by one copy of the layer's code; layer 5 and the x part in front of 6
            if (i == 4) {
                layer_s(4, no, no);
                x_phase(acc, TT::OFF_X5);      // layer 5: the skip connection's x part behind the trunk part (nerf.py:99-101)
                C2_STAMP(7);
                run_prologue(5);
            }
            layer_s(i < 4 ? i : 5, yes, yes);
        }
        layer_s(6, yes, no);
#else
        // ---- layers 1..7.  An iteration = [request the layer's first weight block; barrier; write the previous layer's
        // output as this layer's input; barrier; the layer's MFMAs]: the request is covered by the conversion and retired
        // inside the iteration, so nothing hand-issued is in flight across the loop's back edge ----
        uint8_t* const hw = lane_reg + wave * UNIT_LDS;                 // this wave's k-unit of the next input
        ASet (&A)[2] = W[0].k;
        C2_STAMP(2);
#pragma unroll 1
        for (int hs = 0; hs < 7; ++hs) {
            const uint8_t* wl = wx_p + TT::OFF_HID(0) + hs * TT::SEC_H + (hs >= 5 ? TT::SEC_X : 0);
            issue_a(A[0], lane16, wl);
#if defined(PG_C2_EARLY_CONV)
            // The first-dispatched wave of a SIMD wins the MFMA arbitration, is through a layer's MFMAs ~40 % earlier than its
            // partner and then waits at this barrier (profiles/r5_c2_plain_stamps.txt: 22 k ticks per pass): it forms its
            // fragments (VALU only, no LDS) in that time, beside the partner's MFMAs; the partner converts behind the barrier.
            HFrags hf;
            settle<0, TT::NCT>(acc);
            if (wave < 4) conv_form<true>(acc, hf, s129);
            lds_barrier();                                              // everyone is done reading the previous input
            C2_STAMP(3);
            if (wave >= 4) conv_form<true>(acc, hf, s129);
            conv_store(hf, hw);
#else
            lds_barrier();                                              // everyone is done reading the previous input
            C2_STAMP(3);
            settle<0, TT::NCT>(acc);
            conv_write<true>(acc, hw, s129);
#endif
            {
                const f32x4 b0 = bias_tile(bias, BS_LAYER0 + (hs + 1) * NT16 + 2 * wave, g_p);
                const f32x4 b1 = bias_tile(bias, BS_LAYER0 + (hs + 1) * NT16 + 2 * wave + 1, g_p);
#pragma unroll
                for (int c = 0; c < TT::NCT; ++c) { acc[0][c] = b0; acc[1][c] = b1; }
            }
            C2_STAMP(4);
            lds_barrier();                                              // the layer's input is complete
            C2_STAMP(5);
            hidden_mma<false>(acc, A, wl, nullptr, hb_lo, hb_lo + 4 * UNIT_LDS, lane16);
            C2_STAMP(6);
            if (hs == 4) {      // layer 5: the skip connection's x part behind the trunk part (nerf.py:99-101)
                lds_barrier();                                          // everyone is done reading h4
                x_phase(acc, TT::OFF_X5);
                C2_STAMP(7);
            }
        }
#endif

        // ---- sigma head and the view layer's trunk part (feature layer folded in, NetTensors::fold): wave w takes view tiles
        // 2 v, 2 v + 1 (v = w & 3) for the column tiles 4 (w >> 2) .. + 3, and the alpha row for column tile w ----
        const int v4 = wave & 3, chalf = wave >> 2;
        f32x4 av[2][4], al;
        {
            const f32x4 b0 = bias_tile(bias, BS_VIEWF + 2 * v4, g_p), b1 = bias_tile(bias, BS_VIEWF + 2 * v4 + 1, g_p);
#pragma unroll
            for (int c = 0; c < 4; ++c) { av[0][c] = b0; av[1][c] = b1; }
            al = bias_tile(bias, BS_ALPHA, g_p);
        }
        {
            // alpha A fragments: only the lanes of row 0 hold weights, the others read the zero entry
            a128 alA[HU16][2];
            const uint8_t* ap = smem + L_ALPHA + ((lane_p & 15) == 0 ? g_p * 16 : 64);
#pragma unroll
            for (int u = 0; u < HU16; ++u) {
                alA[u][0] = ld128(ap + (2 * u) * TT::ALPHA_STRIDE);
                alA[u][1] = ld128(ap + (2 * u + 1) * TT::ALPHA_STRIDE);
            }
            const uint8_t* wl = wb_p + TT::OFF_AV + v4 * TT::KBLK;
#if defined(PG_C2_SPLIT)
            ASet (&A)[2] = W[0].k;
            issue_a(A[0], lane16, wl);
            settle<4, 4>(acc);
            conv_cols<4, 4, false>(acc, hw, s129);                      // beta's half of h7 (alpha's went out under layer 7's b2 b3)
#else
            issue_a(A[0], lane16, wl);
            lds_barrier();                                              // everyone is done reading layer 7's input
            settle<0, TT::NCT>(acc);
            conv_write<true>(acc, hw, s129);
#endif
            lds_barrier();                                              // h7 is complete (and the alpha fragments have landed)
            const unsigned vb_lo = hb_lo + chalf * (4 * 2048), vb_hi = vb_lo + 4 * UNIT_LDS;
            a128 B[3][2];
            constexpr int NS = HU16 * 4;
            auto issue_b = [&](auto ic) {
                constexpr int s = decltype(ic)::value, u = s / 4, c = s % 4;
                constexpr int off = (u & 3) * UNIT_LDS + c * 2048;
                ds_rd<off>(B[s % 3][0], u < 4 ? vb_lo : vb_hi);
                ds_rd<off + 1024>(B[s % 3][1], u < 4 ? vb_lo : vb_hi);
            };
            issue_b(std::integral_constant<int, 0>{});
            issue_b(std::integral_constant<int, 1>{});
            static_for<0, NS>([&](auto ic) {
                constexpr int s = decltype(ic)::value, u = s / 4, c = s % 4;
                if constexpr (c == 0) {
                    if constexpr (u + 1 < HU16) {
                        issue_a(A[(u + 1) & 1], lane16, wl + (u + 1) * (4 * TT::KBLK));
                        wait_a<4>(A[u & 1]);
                    } else wait_a<0>(A[u & 1]);
                }
                if constexpr (s + 2 < NS) issue_b(std::integral_constant<int, s + 2>{});
                constexpr int younger = s + 2 < NS ? 4 : 2 * (NS - 1 - s);
                wait_pair<younger>(B[s % 3][0], B[s % 3][1]);
                const ASet& w4 = A[u & 1];
                av[0][c] = mma(w4.f[0], B[s % 3][0], av[0][c]);
                av[1][c] = mma(w4.f[2], B[s % 3][0], av[1][c]);
                av[0][c] = mma(w4.f[1], B[s % 3][1], av[0][c]);
                av[1][c] = mma(w4.f[3], B[s % 3][1], av[1][c]);
                if (v4 == c) {                  // (wave-uniform) column tile 4 chalf + v4 = wave: the alpha row
                    al = mma(alA[u][0], B[s % 3][0], al);
                    al = mma(alA[u][1], B[s % 3][1], al);
                }
            });
        }
        C2_STAMP(8);
        lds_barrier();                                                  // everyone is done reading h7: the region is free

        // ---- the pass's tail: Y of the pass's rays and the view embedder's cutoff weights into the region ----
        {
            // cutoff weights of this wave's column tile (cutoff_embedder.py:139-146 with the view embedder's tau), the frame
            // code's pseudo joint with weight 1; a point belongs to the tile's first or second ray
            float wdv[8];
#pragma unroll
            for (int jj = 0; jj < JG; ++jj) {
                float qx, qy, qz;
                local(jj, qx, qy, qz);
                const float wv = cutoff_weight_fast(__builtin_amdgcn_sqrtf(qx * qx + qy * qy + qz * qz), tld, cut[J + JG * g_p + jj]);
                wdv[jj] = ((gmask >> jj) & 1) ? 0.0f : wv;     // (a limb left out of the pass: below 2^-24 in every point, and no Y formed)
            }
            wdv[6] = (FC && g_p == 0) ? 1.0f : 0.0f;
            wdv[7] = 0.0f;
            const FragC f = frag_of(wdv, s129);
            const int ra_own = min(ray_of(off0 + 16 * wave), nrm1);
            int ipt2 = 16 * wave + col_p;
            asm volatile("" : "+v"(ipt2));            // (the point's ray formed again rather than kept through the pass)
            const bool first = min(ray_of(off0 + ipt2), nrm1) == ra_own;
            unsigned z4[4];
                zero4(z4);        // (formed here: hoisted out of the pass loop, hipcc spills the four zeros)
            uint8_t* wp = lane_reg + L_WD + wave * 4096;
            st128(wp, first ? f.x1 : z4);
            st128(wp + 1024, first ? f.x2 : z4);
            st128(wp + 2048, first ? z4 : f.x1);
            st128(wp + 3072, first ? z4 : f.x2);
        }
        {
            // Y[ray][slot][out] = sum_k W_vd[out, (joint, k)] T[ray][slot][k] (pg_layout.h "factorised view layer") for the
            // slots of the limbs in range: A = the rays' 27 view values (row = ray; hardware sin / cos, split like an
            // activation), B = the weights (column = out channel) -> lane (g, col) holds Y[ray 4 g + i][slot][16 t + col].
            // In rounds of eight slots: wave w forms the T fragment pair of the round's w-th slot ONCE (into L_TQ) and, behind
            // a barrier, out tile t = w of all eight -- the same work in every wave whatever limbs are in range.  A wave
            // zeroes its tile's image first and scatters into it: LDS operations of one wave stay in order.
            const int rowr = min(col_p, nrm1);
            const bool live = col_p <= nrm1;
            const uint8_t* abr = smem + L_AB + abuf * L_ABSZ + rowr * REC_AB_BYTES;
            {
                unsigned z4[4];
                zero4(z4);        // (formed here: hoisted out of the pass loop, hipcc spills the four zeros)
#pragma unroll
                for (int ray = 0; ray < TT::MAXR; ++ray)
                    if (ray <= nrm1) { st128(lane_reg + L_Y + ray * L_YRAY + wave * 2048, z4); st128(lane_reg + L_Y + ray * L_YRAY + wave * 2048 + 1024, z4); }
            }
            const uint8_t* wy_w = wb_p + TT::OFF_Y + wave * 2048 + lane_p * 16;         // + slot * 16 KiB: this wave's tile, both planes
            const int nsl = 4 * nl + (FC ? 1 : 0);                                      // slots to form: four per limb in range, the frame code
            auto slot_of = [&](int q) {                                                 // q-th slot of the pass
                return q < 4 * nl ? JG * (q & 3) + (int)((ll >> (4 * (q >> 2))) & 15u) : J;
            };
#pragma unroll 1
            for (int q0 = 0; q0 < nsl; q0 += TT::NW) {
                // this wave's out tile of the round's weights: requested first, they land while the T fragments are formed
                a128 w[TT::NW][2];
                static_for<0, TT::NW>([&](auto ic) {
                    constexpr int j = decltype(ic)::value;
                    const int s = slot_of(min(q0 + j, nsl - 1));
                    w[j][0] = ld128(wy_w + (size_t)s * L_YRAY);
                    w[j][1] = ld128(wy_w + (size_t)s * L_YRAY + 1024);
                });
                if (q0 > 0) lds_barrier();                              // everyone is done with the previous round's T fragments
                if (q0 + wave < nsl) {
                    const int s = slot_of(q0 + wave);
                    float tv[8];
                    if (s < J) {
                        const float4 b = *reinterpret_cast<const float4*>(abr + s * 32 + 16);
                        const float inv = __builtin_amdgcn_rsqf(fmaxf(b.x * b.x + b.y * b.y + b.z * b.z, 1e-24f));
                        const float ex = b.x * inv, ey = b.y * inv, ez = b.z * inv;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int k = 8 * g_p + i;                  // 0 .. 31 (27 used): component c = k / 9, row r9 = k % 9
                            const int c = (k >= 9) + (k >= 18), r9 = k - 9 * c;
                            const float ec = c == 0 ? ex : (c == 1 ? ey : ez);
                            const int f = (r9 - 1) >> 1;
                            const float ang = ec * 0.15915494309189535f * (float)(1 << (f < 0 ? 0 : f)) + (((r9 - 1) & 1) ? 0.25f : 0.0f);
                            const float sv = __builtin_amdgcn_sinf(ang);
                            tv[i] = k >= 27 ? 0.0f : (r9 == 0 ? ec : sv);
                        }
                    } else {
                        const float cf = a.cams ? a.cams[min((long long)r0 + rowr, (long long)a.n_rays - 1)] : -1.0f;
                        const int ci = cf < 0.0f ? a.n_codes : min((int)cf, a.n_codes - 1);
#pragma unroll
                        for (int i = 0; i < 8; ++i) tv[i] = g_p < 2 ? a.codes[ci * FC_CH + 8 * g_p + i] : 0.0f;
                    }
                    // (trans forwarding: a VALU instruction may not read a v_sin result in the next issue slot; conv_a is inline asm)
                    asm volatile("s_nop 0" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]), "+v"(tv[4]), "+v"(tv[5]), "+v"(tv[6]), "+v"(tv[7]));
                    FragC f = frag_of(tv, s129);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { f.x1[q] = live ? f.x1[q] : 0u; f.x2[q] = live ? f.x2[q] : 0u; }
                    st128(lane_reg + L_TQ + wave * 2048, f.x1);
                    st128(lane_reg + L_TQ + wave * 2048 + 1024, f.x2);
                }
                lds_barrier();                                          // the round's T fragments are complete
                static_for<0, TT::NW>([&](auto ic) {
                    constexpr int j = decltype(ic)::value;
                    if (q0 + j < nsl) {
                        const int s = slot_of(q0 + j);
                        const a128 t1 = ld128(lane_reg + L_TQ + j * 2048), t2 = ld128(lane_reg + L_TQ + j * 2048 + 1024);
                        f32x4 c4 = {0.0f, 0.0f, 0.0f, 0.0f};
                        c4 = mma(t1, w[j][0], c4);
                        c4 = mma(t2, w[j][1], c4);
                        const int gq = s < J ? s / JG : 0, eq = s < J ? s % JG : JG;
                        uint8_t* yd = smem + L_Y + wave * 2048 + (gq * 16 + col_p) * 16 + eq * 2;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int ray = 4 * g_p + i;
                            if (ray <= nrm1) {
                                float v = c4[i] * (1.0f / (float)COMP_S);
                                asm volatile("" : "+v"(v));             // one rounded value for both halves
                                const _Float16 y1 = (_Float16)v;
                                const float y1f = (float)y1;
                                *reinterpret_cast<_Float16*>(yd + ray * L_YRAY) = (_Float16)((float)(COMP_S - 1) * y1f);
                                *reinterpret_cast<_Float16*>(yd + ray * L_YRAY + 1024) = (_Float16)fmaf((float)COMP_S, v - y1f, y1f);
                            }
                        }
                    }
                });
            }
        }
        C2_STAMP(9);
        lds_barrier();                                                  // Y and the cutoff weights are complete
        // ---- second stage of the factorised view layer: av += Y[ray] w for the two rays a column tile can touch (the weights
        // of the second are all zero where the tile lies on one ray) ----
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int c = 4 * chalf + cc;
            const int ra = min(ray_of(off0 + 16 * c), nrm1), rb = min(ra + 1, nrm1);
            const uint8_t* wp = lane_reg + L_WD + c * 4096;
            a128 b0 = ld128(wp), b1 = ld128(wp + 1024), b2 = ld128(wp + 2048), b3 = ld128(wp + 3072);
            const uint8_t* ya = lane_reg + L_Y + ra * L_YRAY + (2 * v4) * 2048;
            const uint8_t* yb = lane_reg + L_Y + rb * L_YRAY + (2 * v4) * 2048;
            a128 a0 = ld128(ya), a1 = ld128(ya + 1024), a2 = ld128(ya + 2048), a3 = ld128(ya + 3072);
            a128 e0 = ld128(yb), e1 = ld128(yb + 1024), e2 = ld128(yb + 2048), e3 = ld128(yb + 3072);
            pin4(b0, b1, b2, b3); pin4(a0, a1, a2, a3); pin4(e0, e1, e2, e3);
            av[0][cc] = mma(a0, b0, av[0][cc]);
            av[1][cc] = mma(a2, b0, av[1][cc]);
            av[0][cc] = mma(a1, b1, av[0][cc]);
            av[1][cc] = mma(a3, b1, av[1][cc]);
            av[0][cc] = mma(e0, b2, av[0][cc]);
            av[1][cc] = mma(e2, b2, av[1][cc]);
            av[0][cc] = mma(e1, b3, av[0][cc]);
            av[1][cc] = mma(e3, b3, av[1][cc]);
        }
        nx_z = a.z[min(p0 + step + 16 * wave + col_p, a.n_points - 1)];     // the next pass's depth: in flight through the rgb head
        // rgb A fragments: the lanes of rows 0..2 hold weights, the others read the zero entry
        a128 rgA[VW / 32][2];
        {
            const uint8_t* rp = smem + L_RGB + ((lane_p & 15) < 3 ? (3 * g_p + (lane_p & 15)) * 16 : 12 * 16);
#pragma unroll
            for (int u = 0; u < VW / 32; ++u) {
                rgA[u][0] = ld128(rp + (2 * u) * TT::RGB_STRIDE);
                rgA[u][1] = ld128(rp + (2 * u + 1) * TT::RGB_STRIDE);
            }
        }
        lds_barrier();                                                  // everyone is done with Y and the weights
        // view activations -> k-unit v4 of the rgb head's input, the wave's four column tiles
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            unsigned h[4], x[4];
            const float v[8] = {av[0][cc][0], av[0][cc][1], av[0][cc][2], av[0][cc][3], av[1][cc][0], av[1][cc][1], av[1][cc][2], av[1][cc][3]};
            conv4<true>(v, h, x, s129);
            uint8_t* gp = lane_reg + L_G + v4 * UNIT_LDS + (4 * chalf + cc) * 2048;
            st128(gp, h);
            st128(gp + 1024, x);
        }
        lds_barrier();                                                  // the view activations are complete
        // ---- rgb head: wave w takes column tile w ----
        f32x4 c3 = bias_tile(bias, BS_RGB, g_p);
        {
            const uint8_t* gp = lane_reg + L_G + wave * 2048;
            a128 g0 = ld128(gp), g1 = ld128(gp + 1024), g2 = ld128(gp + UNIT_LDS), g3 = ld128(gp + UNIT_LDS + 1024);
            a128 g4 = ld128(gp + 2 * UNIT_LDS), g5 = ld128(gp + 2 * UNIT_LDS + 1024), g6 = ld128(gp + 3 * UNIT_LDS), g7 = ld128(gp + 3 * UNIT_LDS + 1024);
            pin4(g0, g1, g2, g3); pin4(g4, g5, g6, g7);
            c3 = mma(rgA[0][0], g0, c3); c3 = mma(rgA[0][1], g1, c3);
            c3 = mma(rgA[1][0], g2, c3); c3 = mma(rgA[1][1], g3, c3);
            c3 = mma(rgA[2][0], g4, c3); c3 = mma(rgA[2][1], g5, c3);
            c3 = mma(rgA[3][0], g6, c3); c3 = mma(rgA[3][1], g7, c3);
        }
        {
            int ipt3 = 16 * wave + col_p;       // (the point's index formed again rather than kept through the pass)
            asm volatile("" : "+v"(ipt3));
            if (g_p == 0 && ipt3 <= last)       // rows 0..2 of the rgb tile and row 0 of the alpha tile live in lane group 0
                *reinterpret_cast<float4*>(a.raw + (p0 + ipt3) * 4) = make_float4(c3[0], c3[1], c3[2], al[0]);
        }
        abuf ^= 1;
        p0 += step; r0 = r0n; off0 = off0n;
        C2_STAMP(10);
#if defined(PG_STAMPS)
        stamp_acc[11] += 1;
#endif
    }
#if defined(PG_STAMPS)
    if (a.dbg && a.dbg_stage == 99 && lane == 0)
        for (int k = 0; k < 12; ++k) reinterpret_cast<unsigned long long*>(a.dbg)[((long long)blockIdx.x * TT::NW + wave) * 16 + k] = stamp_acc[k];
#endif
    if (count_skips && tid == 0)
        for (int k = 0; k < 3; ++k) atomicAdd(reinterpret_cast<unsigned*>(a.dbg) + k, n_cnt[k]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool FC, bool PP>
static hipError_t launch(const EvalArgs& a, int grid, hipStream_t stream) {
    auto k = evalc2_kernel<FC, PP>;
    static std::atomic<unsigned long long> attr_done{0};
    const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(k), L_TOTAL, attr_done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHR2), L_TOTAL, stream, a);
    return hipGetLastError();
}

}  // namespace c2
}  // namespace pgd

// needs S >= pgp::T::MIN_S, rays (no explicit points, no position noise), the shape-T weights (pack_c2) in a.wstream and the
// 16-row bias table (pack_bias_s) in a.bias; any pose stride; frame codes when `framecode`; no debug taps (a.dbg only
// receives the stamps of a PG_STAMPS build, stage 99, or the limb-mask counters, stage 97)
extern "C" int pg_launch_evalc2(const pgd::EvalArgs* a, int framecode, int grid, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a->S < pgp::T::MIN_S || a->pts || a->pnoise || (a->dbg && a->dbg_stage != 99 && a->dbg_stage != 97)) return (int)hipErrorInvalidValue;
    const bool pp = a->pose_stride != 0;
    if (framecode) return (int)(pp ? pgd::c2::launch<true, true>(*a, grid, s) : pgd::c2::launch<true, false>(*a, grid, s));
    return (int)(pp ? pgd::c2::launch<false, true>(*a, grid, s) : pgd::c2::launch<false, false>(*a, grid, s));
}

extern "C" int pg_evalc2_points_per_pass(void) { return pgp::T::PTS; }
