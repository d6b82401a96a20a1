// pg_pack.h -- host packer interface (see pg_pack.cpp)
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/posegen_hip.h"
#include "pg_program.h"

namespace pgpack {

enum { SHAPE_A = 0, SHAPE_B = 1, SHAPE_C = 2 };

// Borrowed host pointers to one net's tensors (reference checkpoint layout).
struct NetTensors {
    const float* lw[pgl::DEPTH] = {};
    const float* lb[pgl::DEPTH] = {};
    int lcols[pgl::DEPTH] = {};
    const float* alpha_w = nullptr; const float* alpha_b = nullptr;
    const float* feat_w = nullptr;  const float* feat_b = nullptr;
    const float* view_w = nullptr;  const float* view_b = nullptr;
    int view_cols = 0;
    const float* rgb_w = nullptr;   const float* rgb_b = nullptr;
    // W_view[:, :256] W_feature [128,256] and b_view + W_view[:, :256] b_feature [128] (fold())
    std::vector<float> viewf_w, viewf_b;
    void fold();
    float w(int mat, int row, int col) const;
    // The same tensors as ONE flat vector (the device-side source of pg_load_weights_device): the 24 tensors in
    // pg_load_weights order, then the folded view layer's weights and bias.  flat(): the offset of the element w() returns
    // (-1 where w() returns 0 by range); the packers below record it per output element when asked (`src`), so that a packed
    // image can be re-formed from new weight values by a gather.
    static constexpr int N_SRC = 26, SRC_VIEWF_W = 24, SRC_VIEWF_B = 25;
    long long off[N_SRC + 1] = {};
    void layout(int framecode_ch);
    long long flat(int mat, int row, int col) const;
    long long flat_bias(int tensor, int row) const { return off[tensor] + row; }
};
// one entry of a source map: (flat offset << 2) | kind, or -1 (a zero of the layout)
enum { SRC_PLAIN = 0, SRC_COMP0 = 1, SRC_COMP1 = 2 };

// Packs the weight stream for `precision`; returns 0, or <0 on an internal layout error.
// `fact`: for PG_PREC_FP16C, the program of the dedicated kernel pg_evalc.hip (pg_program.h C); the 16-bit
// precisions' second program (rays with >= 64 samples, pg_eval16r.hip) is pack_stream_r.
// `rec` (with fact, PG_PREC_FP16C only): the record variant of that kernel -- no view-direction segment, the view
// directions arrive as per-ray Y records (pack_vyc, pg_rayrec.hip).
// `onchip` (with rec, no frame codes): the variant without per-ray records (pg_evalc.hip OC): + one chunk per joint pair of the
// view layer's direction weights behind layer 0 (pg_program.h C::C_YC).
int pack_stream(const NetTensors& t, int precision, bool framecode, bool fact, std::vector<uint8_t>& out,
                std::vector<int>* seg_chunk_base = nullptr, bool rec = false, bool onchip = false);
void pack_bias(const NetTensors& t, std::vector<float>& out);
// stream and bias table of the 16x16x32 kernel (pg_program.h R, pg_layout.h "small tile")
// `onchip`: the variant without per-ray records (pg_eval16r.hip OC): + one chunk per limb of the view layer's direction
// weights behind layer 0 ([joint slot 6 g + jj][out tile16 t], k = the joint's 27 view values: vd_channel)
int pack_stream_r(const NetTensors& t, int precision, std::vector<uint8_t>& out, bool onchip = false, std::vector<int32_t>* src = nullptr);
void pack_bias_s(const NetTensors& t, std::vector<float>& out, std::vector<int32_t>* src = nullptr);
// weights of the compensated kernel with the out tiles split over the waves (pg_evalc2.hip; pg_program.h T): fragments
// addressed directly, no stream; the 16-row bias table (pack_bias_s) goes with it
int pack_c2(const NetTensors& t, bool framecode, std::vector<uint8_t>& out, std::vector<int32_t>* src = nullptr);
// Y-stage weights of the record kernel (pg_rayrec.hip): [wave 8][unit n][64 lanes x 16 B]; unit n of
// wave w = (joint slot16_joint(vy_slot(w, n/2)), k-unit n%2) of out tile w&3 as an MFMA B operand.
int pack_vy(const NetTensors& t, int precision, bool framecode, std::vector<uint8_t>& out);
// fp32 Y-stage weights of the compensated-fp16 record kernel: [joint 0..24][VYC_K][128 out channels],
// W_vd[o, (j, k)] for value k = c * 9 + row of joint j (vd_channel), joint 24 = the frame code
void pack_vyc(const NetTensors& t, bool framecode, std::vector<float>& out);

}  // namespace pgpack
