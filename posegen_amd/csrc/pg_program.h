// pg_program.h -- the fixed "program" of weight units one 32-point group consumes.
//
// A program is a list of segments; a segment multiplies one weight matrix by one or
// more lane-value sequences (pg_layout.h) for NO out tiles, either out-tile-major
// ([o][u]) or k-major ([u][o]); its units are laid out linearly in consumption order
// and the segment is padded to a whole number of 16-KiB chunks.
//
//   shape A  (pg_eval16.hip: bf16 / fp16, one MFMA per unit, 8 waves x 32 points)
//   shape B  (pg_eval32.hip: fp32 / bf16x3 / fp16x3, k-major everywhere, 4 waves)
#pragma once
#include "pg_layout.h"

namespace pgp {
using namespace pgl;

enum Seq { SEQ_X = 0, SEQ_H = 1, SEQ_D = 2, SEQ_CODE = 3 };
enum Mat { MAT_L0 = 0, /* .. MAT_L7 = 7 */ MAT_FEAT = 8, MAT_ALPHA = 9, MAT_VIEW = 10, MAT_RGB = 11,
           MAT_FEAT_ALPHA = 12 /* tiles 0..7 feature_linear, tile 8 row 0 alpha_linear */ };

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------- shape A: 16-bit operands, 8 values per lane per unit -------------
namespace A {
constexpr int UE = 8;                         // values per lane per unit
constexpr int UPC = 16;                       // units per chunk
constexpr int XU = XSEQ / UE;                 // 27
constexpr int HU = HSEQ / UE;                 // 16
constexpr int DU = DSEQ / UE;                 // 41
constexpr int CH_L0X = cdiv(XU * NT, UPC);    // 14
constexpr int CH_HID = HU * NT / UPC;         // 8
constexpr int C_L0 = 0;
constexpr int C_L1 = C_L0 + CH_L0X;           // 14  (L1..L4 consecutive)
constexpr int C_L5H = C_L1 + 4 * CH_HID;      // 46
constexpr int C_L5X = C_L5H + CH_HID;         // 54
constexpr int C_L6 = C_L5X + CH_L0X;          // 68  (L6, L7 consecutive)
constexpr int C_FA = C_L6 + 2 * CH_HID;       // 84
constexpr int C_VF = C_FA + 9;                // 93
constexpr int C_VD = C_VF + NTV;              // 97
constexpr int CH_VD = 11;                     // cdiv((41 [+1]) * 4, 16)
constexpr int C_RGB = C_VD + CH_VD;           // 108
constexpr int NCHUNK = 110;                   // 109 rounded up to even (static ring parity)
static_assert(cdiv((DU + 1) * NTV, UPC) == CH_VD && cdiv(DU * NTV, UPC) == CH_VD, "view chunks");
// MFMAs issued per 32-point group (for pg_query / roofline bookkeeping)
constexpr int MFMA_PER_GROUP(bool fc) {
    return XU * NT * 2 + 6 * HU * NT + HU * NT /*L5h*/ + HU * 9 + HU * NTV + (DU + (fc ? 1 : 0)) * NTV + 8;
}
}  // namespace A

// ---------------- shape B: fp32 (UE 4, 1-KiB units) or split 16-bit (UE 8, 2-KiB) ---
namespace B {
constexpr int CH_L0X = 27;
constexpr int CH_HID = 16;
constexpr int C_L0 = 0;
constexpr int C_L1 = C_L0 + CH_L0X;           // 27
constexpr int C_L5H = C_L1 + 4 * CH_HID;      // 91
constexpr int C_L5X = C_L5H + CH_HID;         // 107
constexpr int C_L6 = C_L5X + CH_L0X;          // 134
constexpr int C_F = C_L6 + 2 * CH_HID;        // 166
constexpr int C_ALPHA = C_F + CH_HID;         // 182
constexpr int C_VF = C_ALPHA + 2;             // 184
constexpr int C_VD = C_VF + 8;                // 192
constexpr int CH_VD = 21;
constexpr int C_RGB = C_VD + CH_VD;           // 213
constexpr int NCHUNK = 214;
}  // namespace B

}  // namespace pgp
