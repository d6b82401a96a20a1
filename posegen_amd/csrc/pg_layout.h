// pg_layout.h -- weight-stream / lane-value layouts shared by the host packer
// (pg_pack.cpp) and the fused embed+MLP kernels (pg_eval*.hip).
//
// Design ("points on lanes, channels in registers"):
//   Every layer is computed transposed, Y^T[out, pt] = W[out, k] * X^T[k, pt], with
//   v_mfma_f32_32x32x{16_bf16,16_f16,2_f32}: A = weights (32 out-channels x K),
//   B = activations (K x 32 points), C/D = 32 out-channels x 32 points.  The C/D
//   layout puts the POINT on the lane (col = lane&31) and the out-channel in the
//   register (row = (r&3) + 8*(r>>2) + 4*(lane>>5)), which is exactly the B-operand
//   layout of the next layer up to a fixed permutation of k.  So a wave owns 32 points
//   and carries their whole activation vector through all layers in registers: no
//   LDS or cross-lane traffic for activations.  The k permutation is folded into the
//   weights by the host packer, which emits the weights as a linear stream of 1-KiB
//   "units" (64 lanes x 16 B = one A fragment) in the exact order the kernel consumes
//   them; the kernel streams them L2 -> LDS with global_load_lds_dwordx4 (lane-linear,
//   conflict-free ds_read_b128) and all waves of a workgroup share each unit.
//
// Reference semantics of the channels: SURVEY.md section 8 (a'), i.e.
// core/cutoff_embedder.py:111-174 (channel = row*24+j / row*72+3j+c),
// core/networks/nerf.py:94-148 (layer structure).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PG_HD __host__ __device__
#else
#define PG_HD
#endif

namespace pgl {

constexpr int J = 24;            // joints
constexpr int JH = 12;           // joints handled by one lane half (h = lane>>5)
constexpr int LV = 7;            // multires (distance frequencies)
constexpr int LD = 4;            // multires_views
constexpr int ROWS_V = 1 + 2 * LV;   // 15
constexpr int ROWS_D = 1 + 2 * LD;   // 9
constexpr int CH_V = J * ROWS_V;     // 360
constexpr int CH_R = J * 3;          // 72
constexpr int CH_X = CH_V + CH_R;    // 432  density-net input
constexpr int CH_D = J * 3 * ROWS_D; // 648  view input
constexpr int W = 256;           // trunk width
constexpr int NT = W / 32;       // 8 out tiles
constexpr int VW = 128;          // view layer width
constexpr int NTV = VW / 32;     // 4
constexpr int DEPTH = 8;
constexpr int SKIP = 4;          // concat after layer 4 -> layer 5 has K = 432+256
constexpr int FC_CH = 16;        // frame code channels (when enabled)
constexpr int COMP_S = 129;      // compensated fp16 (PG_PREC_FP16C): t2 = f16(t1 + COMP_S (t - t1)), COMP_S - 1 a power of two

#ifndef PG_CHUNK_KB
#define PG_CHUNK_KB 32
#endif
constexpr int CHUNK_BYTES = PG_CHUNK_KB * 1024;   // LDS ring slot (16 or 32 units of 1 KiB)
constexpr int UNIT_BYTES = 1024;

// lane-value sequences ------------------------------------------------------------
// X: 216 values per lane half: three superblocks of 4 joints; per joint 16 values
// (q = 0..15) then one leftover unit holding q = 16,17 of the 4 joints.
//   q = 0: v*w   q = 1+2f: sin(2^f v)*w   q = 2+2f: cos(2^f v)*w   q = 15..17: r_xyz
constexpr int XSEQ = 216;
// D: 328 values per lane half: (jj,c) blocks of 8 (rows 0..7) then 36 leftovers
// (row 8 = cos(8e)) and 4 zero pads.
constexpr int DSEQ_MAIN = JH * 3 * 8;   // 288
constexpr int DSEQ = DSEQ_MAIN + 40;    // 328
constexpr int HSEQ = W / 2;             // 128 hidden values per lane half

// row of a 32-row C/D tile held in accumulator register r by lane half h
PG_HD constexpr int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Input column of the density-net input (0..431) for X-sequence index i of half h.
PG_HD constexpr int xseq_channel(int i, int h) {
    int sb = i / 72, w = i % 72;
    int jj = 0, q = 0;
    if (w < 64) { jj = 4 * sb + w / 16; q = w % 16; }
    else        { jj = 4 * sb + (w - 64) / 2; q = 16 + (w - 64) % 2; }
    int j = JH * h + jj;
    return q < ROWS_V ? q * J + j : CH_V + 3 * j + (q - ROWS_V);
}
// View-embedding channel (0..647, -1 = zero pad) for D-sequence index i of half h.
PG_HD constexpr int dseq_channel(int i, int h) {
    if (i < DSEQ_MAIN) {
        int blk = i / 8, row = i % 8;
        int j = JH * h + blk / 3, c = blk % 3;
        return row * (3 * J) + 3 * j + c;
    }
    int k = i - DSEQ_MAIN;
    if (k >= JH * 3) return -1;
    int j = JH * h + k / 3, c = k % 3;
    return 8 * (3 * J) + 3 * j + c;
}
// Hidden channel for H-sequence index i (= 16*tile + r) of half h.
PG_HD constexpr int hseq_channel(int i, int h) { return 32 * (i / 16) + rho(i % 16, h); }

// per-ray LDS table slot (floats)
constexpr int SLOT_SKT = 0;                 // 24 x 12: rows 0..2 of each 4x4 (R|t)
constexpr int SLOT_DTAB = 288;              // 2 halves x DSEQ
constexpr int SLOT_O = SLOT_DTAB + 2 * DSEQ;  // 944
constexpr int SLOT_D = SLOT_O + 3;            // 947
constexpr int SLOT_CAM = SLOT_D + 3;          // 950
constexpr int SLOT_CODE = 952;                // 16 floats
constexpr int SLOT_FLOATS = 976;              // multiple of 4
#ifndef PG_MAXR
#define PG_MAXR 9
#endif
constexpr int MAXR = PG_MAXR;                 // rays overlapped by one workgroup pass

// ---- factorised view layer (16-bit kernels, >= 64 samples per ray) ---------------------
// The 648 view inputs of a point are w_j(point) * T[ray][j][k]: 27 values k = c*9 + row per
// joint that depend on the RAY only, times the per-point cutoff weight.  So
//     W_vd xd = sum_j w_j(point) Y[ray][j][:],   Y[ray][j][o] = sum_k W_vd[o,(j,k)] T[ray][j][k]
// Y is computed once per RAY by pg_rayrec.hip ("per-ray records" below: wave w does out tile w&3 for
// the 12 joints of half w>>2, 2 MFMAs per joint with 32 rays as rows), and every wave of the fused
// kernel contracts over the joints of its points' ray(s) with one K=32 MFMA per out tile16.  The frame
// code rides along as pseudo joint JC with weight 1.  Exact in real arithmetic.
constexpr int TK = 32;                  // k padded per joint (27 used; frame code 16)
constexpr int JC = J;                   // pseudo joint carrying the frame code
constexpr int MAXR_F = 5;               // rays overlapped by one 256-point pass when S >= 64
constexpr int FACT_MIN_S = 64;
// view-input column (0 .. 647 + 16) multiplied by value k of joint j; -1 = zero pad
PG_HD constexpr int vd_channel(int j, int k) {
    if (j < J) return k < 3 * ROWS_D ? (k % ROWS_D) * (3 * J) + 3 * j + k / ROWS_D : -1;
    return k < FC_CH ? CH_D + k : -1;
}
// joint the Y-stage wave w (half w>>2) handles as its e-th (0..12): the joint of SLOT 12 (w>>2) + e of the 16x16x32
// kernel's slot order (slot16_joint below); -1 = none
PG_HD constexpr int vy_slot(int w, int e, bool fc) {
    return e < JH ? JH * (w >> 2) + e : ((fc && (w >> 2) == 0 && e == JH) ? JC : -1);
}

// ---- per-ray LDS slot of the compensated-fp16 kernel (pg_evalc.hip), floats -----------------
// AB: per joint a = R_j o + t_j and b = R_j d (8 floats, 2 pads): q = a + z b; DTAB: the view table
// of the classic slot (2 halves x DSEQ); CODE: the ray's frame code
constexpr int SLOTC_AB = 0;                       // 24 x 8
constexpr int SLOTC_DTAB = J * 8;                 // 192
constexpr int SLOTC_CODE = SLOTC_DTAB + 2 * DSEQ; // 848
constexpr int SLOTC_FLOATS = SLOTC_CODE + FC_CH;  // 864, multiple of 4
constexpr int MAXR_C = 5;                         // rays overlapped by a 128-point pass when S >= 32
constexpr int COMP_MIN_S = 32;
static_assert(SLOTC_FLOATS % 4 == 0 && SLOTC_DTAB % 4 == 0 && SLOTC_CODE % 4 == 0, "LDS alignment");

// ---- "small tile" layout of pg_eval16r.hip (v_mfma_f32_16x16x32, factorised view layer) ----
// A = weights 16 out channels x 32 k, B = activations 32 k x 16 points, C = 16 x 16: lane
// (g = lane>>4, col = lane&15) holds rows 4g..4g+3.  A wave still owns 32 points = two column
// tiles c (points 16c + col), every A fragment feeding both; k-unit u of the next layer is
// out tiles 2u (values 0..3) and 2u+1 (values 4..7) of lane group g.
constexpr int G16 = 4;                  // lane groups
constexpr int JG = J / G16;             // 6 joints per lane group
constexpr int NT16 = W / 16;            // 16 out tiles of the trunk
constexpr int NTV16 = VW / 16;          // 8 of the view layer
constexpr int HU16 = W / 32;            // 8 k-units of a trunk activation
constexpr int XU16 = 15;                // k-units of the density input: 6 joint slots x 16 (15 used) + 3 units of directions
constexpr int XSEQ16 = XU16 * 8;
PG_HD constexpr int hseq16_channel(int i, int g) { return 16 * (2 * (i / 8) + ((i % 8) >> 2)) + 4 * g + (i & 3); }
// Joint of slot s = 6 g + jj (lane group g, its jj-th joint).  A k-unit of the density input holds slot jj of all
// four lane groups, i.e. the four joints {PERM16[jj], PERM16[6 + jj], PERM16[12 + jj], PERM16[18 + jj]}: they are
// chosen as one LIMB each (left leg, right leg, left arm, right arm, spine, head / collars), because a wave skips the
// units of a limb that is out of cutoff range of all its 32 points (pg_eval16r.hip: the cutoff weight
// 1 - sigmoid(tau (v - c)), cutoff_embedder.py:139-146, is exactly zero in fp32 beyond c + 24 / (tau log2 e)) --
// with SMPL's joint order (pelvis, l/r hip, spine1, l/r knee, spine2, l/r ankle, spine3, l/r foot, neck, l/r collar,
// head, l/r shoulder, l/r elbow, l/r wrist, l/r hand) as the slot order a unit would mix four body regions.
constexpr int PERM16[24] = {1, 2, 16, 17, 0, 12,  4, 5, 18, 19, 3, 13,  7, 8, 20, 21, 6, 14,  10, 11, 22, 23, 9, 15};
PG_HD constexpr int slot16_joint(int s) { return PERM16[s]; }
constexpr int XV16 = 2 * JG;            // units 0 .. 11: the cutoff-weighted values (skippable per limb)
// density-input column of X16-sequence index i of lane group g, in generation order: units 2 jj, 2 jj + 1 = values
// q = 0..14 of slot jj (v w, sin / cos of the 7 octaves times w) and one zero pad; units 12 + p = the directions
// r_xyz (values 15..17, not cutoff-weighted) of slots 2 p and 2 p + 1 and two zero pads
PG_HD constexpr int xseq16_channel(int i, int g) {
    int u = i / 8, e = i % 8;
    if (u < XV16) {
        int q = 8 * (u % 2) + e;
        if (q >= ROWS_V) return -1;
        return q * J + slot16_joint(JG * g + u / 2);
    }
    if (e >= 6) return -1;
    return CH_V + 3 * slot16_joint(JG * g + 2 * (u - XV16) + e / 3) + e % 3;
}
// second stage of the factorised view layer: joint in slot e of lane group g (-1 = zero)
PG_HD constexpr int vy16_slot_joint(int g, int e, bool fc) {
    return e < JG ? slot16_joint(JG * g + e) : ((fc && g == 0 && e == JG) ? JC : -1);
}
// bias tiles of 16 rows, [tile][g][4]: L0..L7 (16 each), alpha, folded view (8), rgb
constexpr int BS_LAYER0 = 0;
constexpr int BS_ALPHA = 128;
constexpr int BS_VIEWF = 129;
constexpr int BS_RGB = 137;
constexpr int BS_COUNT = 138;
constexpr int BIAS16_FLOATS = BS_COUNT * 16;

// ---- per-ray records of the factorised 16-bit path (pg_rayrec.hip writes, pg_eval16r.hip reads) ----------
// Everything of the embedding that depends on the RAY only is computed once per ray by a small kernel in front of
// the fused one and handed over through HBM in the exact LDS image the fused kernel wants, so that a workgroup
// pass fetches it with a handful of LDS-DMA pieces instead of building it (table build + Y stage + two barriers
// per pass, and 192 KiB of Y-stage weights through the vector memory path per pass, were 13 % of a pass):
//   AB[ray][slot] = (a = R_j o + t_j, pad, b = R_j d, pad)  8 floats: q = a + z b, j = slot16_joint(slot)
//                                                                                          (REC_AB_BYTES per ray)
//   Y [ray][out tile16 t][lane (g, row)] x 16 B = the 8 joint slots of lane group g (vy16_slot_joint) of
//       Y[ray][j][16 t + row] = sum_k W_vd[16 t + row, (j, k)] T[ray][j][k]                     (REC_Y_BYTES per ray)
// Both arrays carry REC_PAD_RAYS rays of slack at the end: a pass always fetches MAXR_F rays.
constexpr int REC_AB_BYTES = J * 32;              // 768
constexpr int REC_Y_BYTES = (VW / 16) * 64 * 16;  // 8192
constexpr int REC_PAD_RAYS = 8;
constexpr int REC_TILE_RAYS = 32;                 // rays per MFMA tile of the record kernel (rows of a 32x32x16 MFMA)
constexpr int LDS_AB_BYTES = 4096;                // one AB buffer in LDS: MAXR_F x 768 = 3840 -> 4 DMA pieces of 1 KiB
static_assert(MAXR_F * REC_AB_BYTES <= LDS_AB_BYTES && LDS_AB_BYTES / REC_AB_BYTES + 1 <= REC_PAD_RAYS, "AB fetch stays inside the padded array");

// ---- per-ray records of the compensated-fp16 kernel (pg_rayrec.hip writes, pg_evalc.hip reads; >= 64 samples per ray) ----
// Same idea for PG_PREC_FP16C: (a, b) as above, and the view layer's direction part as A operands of the
// compensated product (S-1) y1 w1 + y2 w2 (y = Y / S split like a weight, w = the point's cutoff weights split
// like an activation):
//   Yc[ray][out tile32 t][k-unit u][plane][lane (h, row)] x 16 B = the 8 joint slots (vyc_slot_joint) of
//       plane 0: (S-1) f16(y),  plane 1: f16(y1 + S (y - y1)),  y = Y[ray][j][32 t + row] / S,  Y in fp32
// A 128-point pass touches <= MAXR_CR rays.  The fp32 Y-stage weights are [joint 0..24][VYC_K][128 out] floats.
constexpr int RECC_Y_BYTES = (VW / 32) * 2 * 2 * 1024;    // 16384
constexpr int MAXR_CR = 3;
constexpr int LDS_ABC_BYTES = 3072;               // one AB buffer in LDS: MAXR_CR x 768 = 2304 -> 3 DMA pieces of 1 KiB
constexpr int VYC_K = 28;                         // 27 view values per joint (16 of the frame code), padded
constexpr int VYC_FLOATS = (J + 1) * VYC_K * VW;
// SLOT (slotc_joint below; JC = the frame code) whose weight is value e of k-unit u in lane half h of the second-stage
// B operand (-1 = zero)
PG_HD constexpr int vyc_slot(int u, int h, int e, bool fc) {
    if (u == 0) return JH * h + e;
    if (e < JH - 8) return JH * h + 8 + e;
    return (fc && h == 0 && e == JH - 8) ? JC : -1;
}
// ---- density input of the compensated kernel's RECORD variant (pg_evalc.hip REC): like the 16x16x32 kernel's X16
// sequence, with the TWO lane halves' joints of a unit forming one limb segment, so that a wave (and a pass) can leave
// out the units of a joint pair that is out of cutoff range (pg_eval16r.hip explains the test).  Slot s = 12 h + jj of
// lane half h holds joint PERMC[s]; per slot two units of cutoff-weighted values (one chunk of 16 unit pairs per joint
// pair), then six units of directions (two slots each).  The (a, b) records, the cutoff tables and the Y records of
// this variant are in slot order too.
constexpr int PERMC[24] = {1, 7, 2, 8, 16, 20, 17, 21, 0, 6, 12, 13,   4, 10, 5, 11, 18, 22, 19, 23, 3, 9, 15, 14};
PG_HD constexpr int slotc_joint(int s) { return PERMC[s]; }
constexpr int XVC = 2 * JH;             // 24 units of cutoff-weighted values
constexpr int XUC = XVC + JH / 2;       // + 6 units of directions = 30
PG_HD constexpr int xseqc_channel(int i, int h) {
    int u = i / 8, e = i % 8;
    if (u < XVC) {
        int q = 8 * (u % 2) + e;
        if (q >= ROWS_V) return -1;
        return q * J + slotc_joint(JH * h + u / 2);
    }
    if (e >= 6) return -1;
    return CH_V + 3 * slotc_joint(JH * h + 2 * (u - XVC) + e / 3) + e % 3;
}
static_assert(MAXR_CR * REC_AB_BYTES <= LDS_ABC_BYTES && LDS_ABC_BYTES / REC_AB_BYTES + 1 <= REC_PAD_RAYS && MAXR_CR <= REC_PAD_RAYS,
              "record fetches stay inside the padded arrays");
// compacted bias table of the record variant of pg_evalc.hip: L0..L7 (64 tiles), alpha, folded view (4), rgb
constexpr int BTC_ALPHA = 64;
constexpr int BTC_VIEWF = 65;
constexpr int BTC_RGB = 69;
constexpr int BTC_COUNT = 70;

// bias tiles: L0..L7 (8 each), feature (8), alpha (1), view (4), rgb (1), folded view (4)
constexpr int BT_LAYER0 = 0;
constexpr int BT_FEAT = 64;
constexpr int BT_ALPHA = 72;
constexpr int BT_VIEW = 73;
constexpr int BT_RGB = 77;
constexpr int BT_VIEWF = 78;    // 4 tiles: b_view + W_view[:, :256] b_feature (feature layer folded, 16-bit kernels)
constexpr int BT_COUNT = 82;
constexpr int BIAS_FLOATS = BT_COUNT * 32;   // [tile][h][r]

}  // namespace pgl
