// pg_program.h -- the fixed "program" of weight units one 32-point group consumes.
//
// A program is a list of segments; a segment multiplies one weight matrix by one or
// more lane-value sequences (pg_layout.h) for NO out tiles, either out-tile-major
// ([o][u]) or k-major ([u][o]); its units are laid out linearly in consumption order
// and the segment is padded to a whole number of 16-KiB chunks.
//
//   shape A  (pg_eval16.hip: bf16 / fp16, one MFMA per unit, 8 waves x 32 points, direct view layer)
//   shape R  (pg_eval16r.hip: bf16 / fp16, 16x16x32 tiles, per-ray records: rays with >= 64 samples)
//   shape B  (pg_eval32.hip: fp32 / bf16x3 / fp16x3, k-major everywhere, 4 waves)
#pragma once
#include "pg_layout.h"

namespace pgp {
using namespace pgl;

enum Seq { SEQ_X = 0, SEQ_H = 1, SEQ_D = 2, SEQ_CODE = 3, SEQ_XC = 4 /* record variant of the compensated kernel */ };
enum Mat { MAT_L0 = 0, /* .. MAT_L7 = 7 */ MAT_FEAT = 8, MAT_ALPHA = 9, MAT_VIEW = 10, MAT_RGB = 11,
           MAT_FEAT_ALPHA = 12 /* tiles 0..7 feature_linear, tile 8 row 0 alpha_linear */,
           MAT_ALPHA_VIEWF = 13 /* tile 0 row 0 alpha_linear, tiles 1..4 W_view[:, :256] W_feature */,
           MAT_VIEWF = 14 /* W_view[:, :256] W_feature alone (split-operand kernels) */ };

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------- shape A: 16-bit operands, 8 values per lane per unit -------------
namespace A {
constexpr int UE = 8;                         // values per lane per unit
constexpr int UPC = CHUNK_BYTES / UNIT_BYTES; // units per chunk
constexpr int XU = XSEQ / UE;                 // 27
constexpr int HU = HSEQ / UE;                 // 16
constexpr int DU = DSEQ / UE;                 // 41
constexpr int CH_L0X = cdiv(XU * NT, UPC);
constexpr int CH_HID = cdiv(HU * NT, UPC);
constexpr int C_L0 = 0;
constexpr int C_L1 = C_L0 + CH_L0X;           // L1..L4 consecutive
constexpr int C_L5H = C_L1 + 4 * CH_HID;
constexpr int C_L5X = C_L5H + CH_HID;
constexpr int C_L6 = C_L5X + CH_L0X;          // L6, L7 consecutive
// feature_linear has no activation (nerf.py:103-110), so it is folded into the view layer on the
// host: one out-tile-major segment [alpha | W_view[:, :256] W_feature] on the last trunk output
constexpr int C_AV = C_L6 + 2 * CH_HID;
constexpr int CH_AV = cdiv(HU * (NTV + 1), UPC);
constexpr int C_VD = C_AV + CH_AV;
constexpr int CH_VD = cdiv((DU + 1) * NTV, UPC);
constexpr int C_RGB = C_VD + CH_VD;
constexpr int NCHUNK = C_RGB + 1;
static_assert(cdiv(DU * NTV, UPC) == CH_VD, "view segment must take the same chunks with and without frame code");
// MFMAs issued per 32-point group (for pg_query / roofline bookkeeping)
constexpr int MFMA_PER_GROUP(bool fc) {
    return XU * NT * 2 + 6 * HU * NT + HU * NT /*L5h*/ + HU * (NTV + 1) + (DU + (fc ? 1 : 0)) * NTV + 8;
}
}  // namespace A

// ---------------- shape R: 16x16x32 tiles, per-ray records (pg_eval16r.hip) ---------------------
// [L0 k-major][L1..4][L5h][L5x k-major][L6][L7][alpha | folded view: 9 tiles, then the rgb head in the same chunk]
// The view-direction part of the view layer comes from the per-ray Y records (pg_layout.h), not from the stream.
namespace R {
constexpr int UPC = CHUNK_BYTES / UNIT_BYTES;
constexpr int CH_L0X = cdiv(XU16 * NT16, UPC);          // 8: one chunk per limb (6), then the directions
constexpr int CH_HID = cdiv(HU16 * NT16, UPC);          // 4
constexpr int U_AV = HU16 * (NTV16 + 1);                // 72 units: alpha tile + 8 folded view tiles
constexpr int U_RGB = VW / 32;                          // 4 units, directly behind (no chunk padding in between)
constexpr int CH_AVR = cdiv(U_AV + U_RGB, UPC);         // 3
constexpr int NCHUNK = 2 * CH_L0X + 7 * CH_HID + CH_AVR;        // 47
constexpr int C_L5X = CH_L0X + 5 * CH_HID;              // 28: first chunk of the skip layer's x part (its limb chunks, like layer 0's from 0)
constexpr int NLIMB = JG;                               // limb chunks at the head of both x segments
// on-chip variant (pg_eval16r.hip OC): + one chunk per limb of the view layer's direction weights, [joint slot 6 g + jj][out
// tile16 t] fragments (k = the 27 view values of the joint), right behind layer 0
constexpr int C_Y = CH_L0X;                             // 8
constexpr int NCHUNK_OC = NCHUNK + NLIMB;               // 53
constexpr int C_L5X_OC = C_L5X + NLIMB;                 // 34
static_assert(G16 * NTV16 == UPC, "a limb's direction weights (4 joints x 8 out tiles) are exactly one chunk");
// 16x16x32 MFMAs per 32-point group: two per unit, plus the second stage of the view layer for one ray
constexpr int MFMA16_PER_GROUP = 2 * (2 * XU16 * NT16 + 7 * HU16 * NT16 + U_AV + U_RGB) + 2 * NTV16;
// Y-stage weights of the record kernel (pg_rayrec.hip): per wave (out tile w&3, joint half w>>2) two B fragments
// of 1 KiB per joint (k-units 0, 1), the frame-code pseudo joint behind the 12 joints of the half
constexpr int VY_UNITS(bool fc) { return 2 * (JH + (fc ? 1 : 0)); }                // 24 / 26
constexpr int VY_BYTES(bool fc) { return 8 * VY_UNITS(fc) * UNIT_BYTES; }
}  // namespace R

// ---------------- shape B: fp32 (UE 4, 1-KiB units) or split 16-bit (UE 8, 2-KiB) ---
// bytes per sequence position per out tile are the same for both (256 B), so are the chunk counts
namespace B {
constexpr int VPC = CHUNK_BYTES / 256;        // (sequence position, tile) pairs per chunk
constexpr int CH_L0X = cdiv(XSEQ * NT, VPC);
constexpr int CH_HID = cdiv(HSEQ * NT, VPC);
constexpr int C_L0 = 0;
constexpr int C_L1 = C_L0 + CH_L0X;
constexpr int C_L5H = C_L1 + 4 * CH_HID;
constexpr int C_L5X = C_L5H + CH_HID;
constexpr int C_L6 = C_L5X + CH_L0X;
constexpr int C_F = C_L6 + 2 * CH_HID;
constexpr int C_ALPHA = C_F + CH_HID;
constexpr int CH_ALPHA = cdiv(HSEQ, VPC);
constexpr int C_VF = C_ALPHA + CH_ALPHA;
constexpr int CH_VF = cdiv(HSEQ * NTV, VPC);
constexpr int C_VD = C_VF + CH_VF;
constexpr int CH_VD = cdiv((DSEQ + 8) * NTV, VPC);
constexpr int C_RGB = C_VD + CH_VD;
constexpr int NCHUNK = C_RGB + 1;
// split-operand precisions fold feature_linear into the view layer (as the 16-bit kernels do):
// the feature segment disappears, the view layer's trunk part reads the last trunk activation
constexpr int NCHUNK_FOLD = NCHUNK - CH_HID;
static_assert(cdiv(DSEQ * NTV, VPC) == CH_VD, "view segment must take the same chunks with and without frame code");
}  // namespace B

// ---------------- shape C: compensated fp16 (pg_evalc.hip), k-major everywhere ----------------
// A unit pair = 2 KiB: plane (S-1) w1 then plane w2 (pg_pack.cpp); one pair per (sequence unit, out tile).
// [L0x][L1..L4][L5h][L5x][L6][L7][alpha | folded view trunk: 5 tiles][view directions (+code)][rgb]
namespace C {
constexpr int PPC = CHUNK_BYTES / 2048;       // unit pairs per chunk
constexpr int XU = XSEQ / 8;                  // 27
constexpr int HU = HSEQ / 8;                  // 16
constexpr int DU = DSEQ / 8;                  // 41
constexpr int CH_L0X = cdiv(XU * NT, PPC);    // 14
constexpr int CH_HID = cdiv(HU * NT, PPC);    // 8
constexpr int CH_AV = cdiv(HU * (NTV + 1), PPC);        // 5
constexpr int CH_VD = cdiv((DU + 1) * NTV, PPC);        // 11
constexpr int NCHUNK = 2 * CH_L0X + 7 * CH_HID + CH_AV + CH_VD + 1;
static_assert(cdiv(DU * NTV, PPC) == CH_VD, "view segment must take the same chunks with and without frame code");
constexpr int MFMA_PER_GROUP(bool fc) {
    return 2 * (2 * XU * NT + 7 * HU * NT + HU * (NTV + 1) + (DU + (fc ? 1 : 0)) * NTV + HU / 2);
}
// record variant (>= 64 samples per ray): no view-direction segment, its second stage is 2 k-units x NTV tiles per ray;
// the density input in the XC sequence (pg_layout.h): one chunk per joint pair (12), then the directions (3 chunks)
constexpr int CH_L0XR = cdiv(XUC * NT, PPC);            // 15
constexpr int NCHUNK_R = 2 * CH_L0XR + 7 * CH_HID + CH_AV + 1;
constexpr int C_L5XR = CH_L0XR + 5 * CH_HID;            // 55: first chunk of the skip layer's x part
constexpr int NPAIRJ = JH;                              // joint-pair chunks at the head of both x segments
static_assert(2 * NT == PPC, "a joint pair's two unit rows are exactly one chunk");
constexpr int MFMA_PER_GROUP_R = 2 * (2 * XUC * NT + 7 * HU * NT + HU * (NTV + 1) + HU / 2) + 2 * 2 * NTV;
// on-chip variant (pg_evalc.hip OC: one pose per launch, no frame codes): no per-ray records; + one chunk per joint pair of the
// view layer's direction weights right behind layer 0: unit pairs [k-unit u of 8 view values (4)][out tile o (NTV)], lane
// (h, col) = out channel 32 o + col, values 8 u + e of joint slot 12 h + p (vd_channel order, 27 used)
constexpr int C_YC = CH_L0XR;                           // 15
constexpr int NCHUNK_OC = NCHUNK_R + NPAIRJ;            // 104
constexpr int C_L5XR_OC = C_L5XR + NPAIRJ;              // 67
static_assert(4 * NTV == PPC, "a joint pair's direction weights (4 k-units x 4 out tiles) are exactly one chunk");
}  // namespace C

// ---------------- shape T: compensated fp16 with the OUT TILES split over the waves (pg_evalc2.hip) ----------------
// No stream and no ring: a workgroup of NW = 8 waves (two per SIMD) carries PTS = 128 points through the net together,
// wave w computing out channels 32 w .. 32 w + 31 (two 16-row tiles of v_mfma_f32_16x16x32_f16) of every layer for all
// of them; the layer's input -- the split fp16 pairs of the activations -- lives in LDS as B fragments
// [k-unit u][column tile c][plane][lane (g, col)] x 16 B (1 KiB each; k-unit u of the next layer = wave u's two tiles,
// hseq16_channel), and every wave reads its OWN quarter-KiB-per-lane of weights straight from L2 into registers:
// fragment (1 KiB: lane (g, row) = out row, k = 8 g + e) [k-unit][wave][tile t of the wave (2)][plane (2)], so that one
// k-unit of one wave is KBLK = 4 KiB contiguous.  Plane 0 = (S-1) w1, plane 1 = w2 (pg_pack.cpp comp_pair).
//   X0  | H1 H2 H3 H4 H5 | X5 | H6 H7 | AV | Y | SMALL
//   X*: the density input in the X16 sequence of pg_layout.h (15 k-units: 6 limbs x 2, then 3 of directions)
//   H*: trunk layers 1..7 on the previous activation (8 k-units)
//   AV: the folded view layer's trunk part, [k-unit][tile pair v (4)][t][plane]: waves v and v + 4 (column tiles 0..3 / 4..7)
//   Y : the view layer's direction (and frame-code) weights as B fragments [joint slot s (24 + code)][out tile16 (8)][plane]:
//       lane (g, col) = out channel 16 t + col, k = 8 g + e of the slot's 27 (16) values (vd_channel)
//   SMALL: the alpha row and the three rgb rows as compact A fragments (only lanes of rows 0 / 0..2 are non-zero):
//       alpha [k-unit 8][plane][5 x 16 B: g = 0..3, zero], rgb [k-unit 4][plane][13 x 16 B: (g, row < 3), zero]
namespace T {
constexpr int NW = 8;
constexpr int PTS = 128;
constexpr int NCT = PTS / 16;                 // 8 column tiles
constexpr int MAXR = 5;                       // rays a pass can touch (samples per ray >= MIN_S)
constexpr int MIN_S = 32;
constexpr int FRAG = 1024;
constexpr int KBLK = 4 * FRAG;
constexpr int SEC_X = XU16 * NW * KBLK;       // 480 KiB
constexpr int SEC_H = HU16 * NW * KBLK;       // 256 KiB
constexpr int SEC_AV = HU16 * 4 * KBLK;       // 128 KiB
constexpr int NSLOT_Y = J + 1;
constexpr int SEC_Y = NSLOT_Y * NTV16 * 2 * FRAG;       // 400 KiB
constexpr int ALPHA_STRIDE = 5 * 16, RGB_STRIDE = 13 * 16;
constexpr int SMALL_ALPHA = HU16 * 2 * ALPHA_STRIDE;    // 1280
constexpr int SMALL_RGB = (VW / 32) * 2 * RGB_STRIDE;   // 1664
constexpr int OFF_X0 = 0;
constexpr int OFF_HID(int hs) { return SEC_X + hs * SEC_H + (hs >= 5 ? SEC_X : 0); }   // hs = 0..6: layers 1..7
constexpr int OFF_X5 = SEC_X + 5 * SEC_H;
constexpr int OFF_AV = OFF_HID(7);
constexpr int OFF_Y = OFF_AV + SEC_AV;
constexpr int OFF_SMALL = OFF_Y + SEC_Y;
constexpr int TOTAL = OFF_SMALL + SMALL_ALPHA + SMALL_RGB;
// 16x16x32 MFMAs per 128-point pass with every limb in range (pg_query / roofline bookkeeping)
constexpr int MFMA16_PER_PASS = NW * (2 * XU16 * 32 + 7 * HU16 * 32 + HU16 * 16 + HU16 * 2 /*alpha*/ + 32 /*second stage*/ + 8 /*rgb*/ + 2 * NSLOT_Y /*Y*/);
}  // namespace T

}  // namespace pgp
