// pg_pack.cpp -- host-side packing of the reference's nn.Linear tensors into the
// per-precision weight streams and the bias table the kernels consume.
//
// Input tensors follow the reference checkpoint layout (core/networks/nerf.py:57-88,
// [out,in] row-major).  Output: a linear stream of A-operand units in the exact
// consumption order of pg_program.h, with the k permutations of pg_layout.h folded in.
#include "pg_pack.h"

#include <cmath>
#include <cstring>

namespace pgpack {
using namespace pgp;

static inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f32_to_f16(float f) {
    _Float16 h = (_Float16)f;      // round-to-nearest-even
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
static inline float f16_to_f32(uint16_t b) {
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return (float)h;
}

struct InUnits { int seq; int n_values; int colbase; };
struct Segment {
    int mat;          // MAT_*
    int no;           // out tiles
    bool kmajor;
    std::vector<InUnits> inputs;
};

static int seq_channel(int seq, int i, int h) {
    switch (seq) {
        case SEQ_X: return xseq_channel(i, h);
        case SEQ_H: return hseq_channel(i, h);
        case SEQ_D: return dseq_channel(i, h);
        case SEQ_CODE: return 8 * h + i;
        case SEQ_XC: return xseqc_channel(i, h);
    }
    return -1;
}

// weight element of matrix `mat`, out row `row`, input column `col` (0 outside)
float NetTensors::w(int mat, int row, int col) const {
    const float* p = nullptr;
    int rows = 0, cols = 0;
    if (mat >= MAT_L0 && mat < MAT_L0 + DEPTH) { p = lw[mat]; rows = W; cols = lcols[mat]; }
    else if (mat == MAT_FEAT) { p = feat_w; rows = W; cols = W; }
    else if (mat == MAT_ALPHA) { p = alpha_w; rows = 1; cols = W; }
    else if (mat == MAT_VIEW) { p = view_w; rows = VW; cols = view_cols; }
    else if (mat == MAT_RGB) { p = rgb_w; rows = 3; cols = VW; }
    else if (mat == MAT_VIEWF) { p = viewf_w.data(); rows = VW; cols = W; }
    else if (mat == MAT_ALPHA_VIEWF) {
        if (row < 32) { p = alpha_w; rows = 1; cols = W; }
        else { p = viewf_w.data(); rows = VW; cols = W; row -= 32; }
    }
    else if (mat == MAT_FEAT_ALPHA) {
        if (row < W) { p = feat_w; rows = W; cols = W; }
        else { p = alpha_w; rows = 1; cols = W; row -= W; }
    }
    if (!p || row < 0 || row >= rows || col < 0 || col >= cols) return 0.f;
    return p[(size_t)row * cols + col];
}

void NetTensors::layout(int framecode_ch) {
    long long o = 0;
    for (int l = 0; l < DEPTH; ++l) {
        const int cols = l == 0 ? CH_X : (l == SKIP + 1 ? CH_X + W : W);
        off[2 * l] = o; o += (long long)W * cols;
        off[2 * l + 1] = o; o += W;
    }
    off[16] = o; o += W;            off[17] = o; o += 1;
    off[18] = o; o += (long long)W * W;    off[19] = o; o += W;
    off[20] = o; o += (long long)VW * (W + CH_D + framecode_ch);   off[21] = o; o += VW;
    off[22] = o; o += 3 * VW;       off[23] = o; o += 3;
    off[SRC_VIEWF_W] = o; o += (long long)VW * W;
    off[SRC_VIEWF_B] = o; o += VW;
    off[N_SRC] = o;
}

long long NetTensors::flat(int mat, int row, int col) const {
    int id = -1, rows = 0, cols = 0;
    if (mat >= MAT_L0 && mat < MAT_L0 + DEPTH) { id = 2 * mat; rows = W; cols = lcols[mat]; }
    else if (mat == MAT_FEAT) { id = 18; rows = W; cols = W; }
    else if (mat == MAT_ALPHA) { id = 16; rows = 1; cols = W; }
    else if (mat == MAT_VIEW) { id = 20; rows = VW; cols = view_cols; }
    else if (mat == MAT_RGB) { id = 22; rows = 3; cols = VW; }
    else if (mat == MAT_VIEWF) { id = SRC_VIEWF_W; rows = VW; cols = W; }
    else if (mat == MAT_ALPHA_VIEWF) {
        if (row < 32) { id = 16; rows = 1; cols = W; }
        else { id = SRC_VIEWF_W; rows = VW; cols = W; row -= 32; }
    }
    else if (mat == MAT_FEAT_ALPHA) {
        if (row < W) { id = 18; rows = W; cols = W; }
        else { id = 16; rows = 1; cols = W; row -= W; }
    }
    if (id < 0 || row < 0 || row >= rows || col < 0 || col >= cols) return -1;
    return off[id] + (long long)row * cols + col;
}

// feature = W_f h + b_f enters the view layer linearly (no activation in between), so
// W_view[:, :256] feature = (W_view[:, :256] W_f) h + W_view[:, :256] b_f.  Products in double.
void NetTensors::fold() {
    viewf_w.assign((size_t)VW * W, 0.f);
    viewf_b.assign(VW, 0.f);
    for (int o = 0; o < VW; ++o) {
        const float* vr = view_w + (size_t)o * view_cols;
        double bsum = view_b ? view_b[o] : 0.0;
        for (int m = 0; m < W; ++m) bsum += (double)vr[m] * (feat_b ? feat_b[m] : 0.f);
        viewf_b[o] = (float)bsum;
        // (row o of W_view[:, :256] W_feature: the sums run over m in the same order as before, k innermost for contiguous reads)
        double acc[W];
        for (int k = 0; k < W; ++k) acc[k] = 0.0;
        for (int m = 0; m < W; ++m) {
            const double v = vr[m];
            const float* fr = feat_w + (size_t)m * W;
            for (int k = 0; k < W; ++k) acc[k] += v * fr[k];
        }
        for (int k = 0; k < W; ++k) viewf_w[(size_t)o * W + k] = (float)acc[k];
    }
}

static std::vector<Segment> program(int shape, bool fc, bool fact, bool fold_b = false, bool rec = false, bool onchip = false) {
    std::vector<Segment> s;
    auto hid = [](int cb) { return InUnits{SEQ_H, HSEQ, cb}; };
    if (shape == SHAPE_C) {         // compensated fp16 kernel (pg_evalc.hip): k-major everywhere, direct view layer
        const InUnits xin = rec ? InUnits{SEQ_XC, XUC * 8, 0} : InUnits{SEQ_X, XSEQ, 0};     // record variant: limb-wise x sequence
        s.push_back({MAT_L0, NT, true, {xin}});
        if (onchip) s.push_back({MAT_VIEW, NTV, true, {}});       // (marker: the joint-pair chunks of direction weights, packed in pack_stream)
        for (int l = 1; l <= 4; ++l) s.push_back({MAT_L0 + l, NT, true, {hid(0)}});
        s.push_back({MAT_L0 + 5, NT, true, {hid(CH_X)}});
        s.push_back({MAT_L0 + 5, NT, true, {xin}});
        for (int l = 6; l <= 7; ++l) s.push_back({MAT_L0 + l, NT, true, {hid(0)}});
        s.push_back({MAT_ALPHA_VIEWF, NTV + 1, true, {hid(0)}});
        if (!rec) {                 // (record variant: the view directions arrive as per-ray Y records)
            Segment v{MAT_VIEW, NTV, true, {{SEQ_D, DSEQ, W}}};
            if (fc) v.inputs.push_back({SEQ_CODE, 8, W + CH_D});
            s.push_back(v);
        }
        s.push_back({MAT_RGB, 1, true, {{SEQ_H, VW / 2, 0}}});
        return s;
    }
    const bool km = (shape == SHAPE_B);
    s.push_back({MAT_L0, NT, true, {{SEQ_X, XSEQ, 0}}});
    for (int l = 1; l <= 4; ++l) s.push_back({MAT_L0 + l, NT, km, {hid(0)}});
    s.push_back({MAT_L0 + 5, NT, km, {hid(CH_X)}});
    s.push_back({MAT_L0 + 5, NT, true, {{SEQ_X, XSEQ, 0}}});
    for (int l = 6; l <= 7; ++l) s.push_back({MAT_L0 + l, NT, km, {hid(0)}});
    if (shape == SHAPE_A) {
        s.push_back({MAT_ALPHA_VIEWF, NTV + 1, false, {hid(0)}});
    } else {
        if (!fold_b) s.push_back({MAT_FEAT, NT, true, {hid(0)}});
        s.push_back({MAT_ALPHA, 1, true, {hid(0)}});
        s.push_back({fold_b ? MAT_VIEWF : MAT_VIEW, NTV, km, {hid(0)}});
    }
    {
        Segment v{MAT_VIEW, NTV, true, {{SEQ_D, DSEQ, W}}};
        if (fc) v.inputs.push_back({SEQ_CODE, 8, W + CH_D});
        s.push_back(v);
    }
    s.push_back({MAT_RGB, 1, km, {{SEQ_H, VW / 2, 0}}});
    return s;
}

int pack_stream(const NetTensors& t, int precision, bool fc, bool fact, std::vector<uint8_t>& out,
                std::vector<int>* seg_chunk_base, bool rec, bool onchip) {
    const int shape = (precision == PG_PREC_BF16 || precision == PG_PREC_FP16) ? SHAPE_A
                    : (precision == PG_PREC_FP16C && fact) ? SHAPE_C : SHAPE_B;
    const bool is_f32 = precision == PG_PREC_FP32;
    const bool is_bf = precision == PG_PREC_BF16 || precision == PG_PREC_BF16X3;
    const bool comp = precision == PG_PREC_FP16C;
    const bool split = precision == PG_PREC_BF16X3 || precision == PG_PREC_FP16X3 || comp;     // two planes per unit
    if (precision < 0 || precision >= PG_PREC_COUNT) return -1;
    if (onchip && (!rec || fc)) return -3;                  // the on-chip variant is a form of the record program, without frame codes
    if ((fact || rec) && shape != SHAPE_C) return -3;       // only PG_PREC_FP16C has a second program here (16-bit rays with >= 64 samples: pack_stream_r)
    if ((shape == SHAPE_A || split) && t.viewf_w.size() != (size_t)VW * W) return -4;   // NetTensors::fold() not called
    const int ue = is_f32 ? 4 : 8;
    const size_t unit_bytes = split ? 2048 : 1024;
    out.clear();
    if (seg_chunk_base) seg_chunk_base->clear();
    // the compensated pair of a weight W = 129 w: plane 0 = 128 w1 (exact), plane 1 = f16(w1 + 129 (w - w1)), w1 = f16(w)
    auto comp_pair = [](float wv, uint16_t& p0, uint16_t& p1) {
        const double wd = (double)wv / COMP_S;
        const double w1 = f16_to_f32(f32_to_f16((float)wd));
        p0 = f32_to_f16((float)((COMP_S - 1) * w1));
        p1 = f32_to_f16((float)(w1 + COMP_S * (wd - w1)));
    };
    for (const Segment& sg : program(shape, fc, fact, split, rec, onchip)) {
        if (seg_chunk_base) seg_chunk_base->push_back((int)(out.size() / CHUNK_BYTES));
        if (sg.inputs.empty()) {        // on-chip variant of shape C: chunk p = joint pair p (slots p and 12 + p), pairs [u][o]
            for (int pj = 0; pj < JH; ++pj)
                for (int u = 0; u < 4; ++u)
                    for (int o = 0; o < NTV; ++o) {
                        const size_t base = out.size();
                        out.resize(base + 2048, 0);
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int ch = vd_channel(slotc_joint(JH * (lane >> 5) + pj), 8 * u + e);
                                if (ch < 0) continue;
                                uint16_t p0, p1;
                                comp_pair(t.w(MAT_VIEW, 32 * o + (lane & 31), W + ch), p0, p1);
                                std::memcpy(&out[base + lane * 16 + e * 2], &p0, 2);
                                std::memcpy(&out[base + 1024 + lane * 16 + e * 2], &p1, 2);
                            }
                    }
            continue;
        }
        // flatten the input units of this segment
        struct U { int seq, u, colbase; };
        std::vector<U> us;
        for (const InUnits& in : sg.inputs)
            for (int u = 0; u < in.n_values / ue; ++u) us.push_back({in.seq, u, in.colbase});
        const int nu = (int)us.size();
        for (int L = 0; L < nu * sg.no; ++L) {
            const int ui = sg.kmajor ? L / sg.no : L % nu;
            const int o = sg.kmajor ? L % sg.no : L / nu;
            const U& un = us[ui];
            const size_t base = out.size();
            out.resize(base + unit_bytes, 0);
            for (int lane = 0; lane < 64; ++lane) {
                const int row = lane & 31, h = lane >> 5;
                for (int e = 0; e < ue; ++e) {
                    const int ch = seq_channel(un.seq, un.u * ue + e, h);
                    const float wv = ch < 0 ? 0.f : t.w(sg.mat, 32 * o + row, un.colbase + ch);
                    if (is_f32) {
                        std::memcpy(&out[base + lane * 16 + e * 4], &wv, 4);
                    } else {
                        if (comp) {
                            uint16_t p0, p1;
                            comp_pair(wv, p0, p1);
                            std::memcpy(&out[base + lane * 16 + e * 2], &p0, 2);
                            std::memcpy(&out[base + 1024 + lane * 16 + e * 2], &p1, 2);
                            continue;
                        }
                        uint16_t hi = is_bf ? f32_to_bf16(wv) : f32_to_f16(wv);
                        std::memcpy(&out[base + lane * 16 + e * 2], &hi, 2);
                        if (split) {
                            const float rem = wv - (is_bf ? bf16_to_f32(hi) : f16_to_f32(hi));
                            uint16_t lo = is_bf ? f32_to_bf16(rem) : f32_to_f16(rem);
                            std::memcpy(&out[base + 1024 + lane * 16 + e * 2], &lo, 2);
                        }
                    }
                }
            }
        }
        out.resize((out.size() + CHUNK_BYTES - 1) / CHUNK_BYTES * CHUNK_BYTES, 0);
    }
    const size_t nchunk = shape == SHAPE_A ? (size_t)A::NCHUNK
                        : shape == SHAPE_C ? (size_t)(onchip ? C::NCHUNK_OC : rec ? C::NCHUNK_R : C::NCHUNK) : (split ? B::NCHUNK_FOLD : B::NCHUNK);
    if (out.size() != nchunk * CHUNK_BYTES) return -2;   // packer and kernel programs disagree
    return 0;
}

void pack_vyc(const NetTensors& t, bool fc, std::vector<float>& out) {
    // row block s = joint SLOT s of the record variant (slotc_joint), then the frame code
    out.assign((size_t)VYC_FLOATS, 0.f);
    for (int sl = 0; sl < J + (fc ? 1 : 0); ++sl)
        for (int k = 0; k < VYC_K; ++k) {
            const int ch = vd_channel(sl < J ? slotc_joint(sl) : JC, k);
            if (ch < 0) continue;
            for (int o = 0; o < VW; ++o) out[((size_t)sl * VYC_K + k) * VW + o] = t.w(MAT_VIEW, o, W + ch);
        }
}

int pack_vy(const NetTensors& t, int precision, bool fc, std::vector<uint8_t>& out) {
    if (precision != PG_PREC_BF16 && precision != PG_PREC_FP16) return -3;
    const bool is_bf = precision == PG_PREC_BF16;
    out.assign((size_t)R::VY_BYTES(fc), 0);
    for (int w = 0; w < 8; ++w)
        for (int n = 0; n < R::VY_UNITS(fc); ++n) {
            const int sl = vy_slot(w, n / 2, fc), ku = n % 2;
            if (sl < 0) continue;
            const int j = sl == JC ? JC : slot16_joint(sl);
            const size_t base = ((size_t)w * R::VY_UNITS(fc) + n) * UNIT_BYTES;
            // lane (hl, col) = out channel 32(w&3)+col, values k = 16ku + 8hl + 0..7 of joint j
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int ch = vd_channel(j, 16 * ku + 8 * (lane >> 5) + e);
                    if (ch < 0) continue;
                    const float wv = t.w(MAT_VIEW, 32 * (w & 3) + (lane & 31), W + ch);
                    const uint16_t hi = is_bf ? f32_to_bf16(wv) : f32_to_f16(wv);
                    std::memcpy(&out[base + lane * 16 + e * 2], &hi, 2);
                }
        }
    return 0;
}

// ---- 16x16x32 kernel (pg_eval16r.hip): units are 16 out rows x 32 k; lane (g, row) holds k = 32u + 8g + 0..7.
// Segments start on chunk boundaries, except the rgb head, which follows the alpha / view tiles directly. ----
static inline int32_t src_entry(long long flat, int kind) { return flat < 0 ? -1 : (int32_t)((flat << 2) | kind); }

int pack_stream_r(const NetTensors& t, int precision, std::vector<uint8_t>& out, bool onchip, std::vector<int32_t>* src) {
    if (precision != PG_PREC_BF16 && precision != PG_PREC_FP16) return -3;
    if (src) src->clear();
    if (t.viewf_w.size() != (size_t)VW * W) return -4;
    const bool is_bf = precision == PG_PREC_BF16;
    struct Seg { int mat, no, nu; bool kmajor, xseq; int colbase; bool pad; };
    std::vector<Seg> prog;
    prog.push_back({MAT_L0, NT16, XU16, true, true, 0, true});
    if (onchip) prog.push_back({MAT_VIEW, NTV16, G16 * JG, false, false, W, true});      // (marker: the limb chunks, packed below)
    for (int l = 1; l <= 4; ++l) prog.push_back({MAT_L0 + l, NT16, HU16, false, false, 0, true});
    prog.push_back({MAT_L0 + 5, NT16, HU16, false, false, CH_X, true});
    prog.push_back({MAT_L0 + 5, NT16, XU16, true, true, 0, true});
    for (int l = 6; l <= 7; ++l) prog.push_back({MAT_L0 + l, NT16, HU16, false, false, 0, true});
    prog.push_back({MAT_ALPHA_VIEWF, NTV16 + 1, HU16, false, false, 0, false});
    prog.push_back({MAT_RGB, 1, VW / 32, false, false, 0, true});
    out.clear();
    for (const Seg& sg : prog) {
        if (sg.mat == MAT_VIEW) {       // chunk jj = limb jj: unit [g' (joint slot 6 g' + jj)][out tile t], lane (g, row): k = 8 g + e
            for (int jj = 0; jj < JG; ++jj)
                for (int gp = 0; gp < G16; ++gp)
                    for (int tt = 0; tt < NTV16; ++tt) {
                        const size_t base = out.size();
                        out.resize(base + UNIT_BYTES, 0);
                        const int j = slot16_joint(JG * gp + jj);
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int ch = vd_channel(j, 8 * (lane >> 4) + e);
                                if (ch < 0) continue;
                                const float wv = t.w(MAT_VIEW, 16 * tt + (lane & 15), W + ch);
                                const uint16_t hi = is_bf ? f32_to_bf16(wv) : f32_to_f16(wv);
                                std::memcpy(&out[base + lane * 16 + e * 2], &hi, 2);
                                if (src) { src->resize(out.size() / 2, -1); (*src)[(base + lane * 16 + e * 2) / 2] = src_entry(t.flat(MAT_VIEW, 16 * tt + (lane & 15), W + ch), SRC_PLAIN); }
                            }
                    }
            continue;
        }
        for (int L = 0; L < sg.nu * sg.no; ++L) {
            const int u = sg.kmajor ? L / sg.no : L % sg.nu;
            const int o = sg.kmajor ? L % sg.no : L / sg.nu;
            const size_t base = out.size();
            out.resize(base + UNIT_BYTES, 0);
            for (int lane = 0; lane < 64; ++lane) {
                const int g = lane >> 4, row = lane & 15;
                for (int e = 0; e < 8; ++e) {
                    const int ch = sg.xseq ? xseq16_channel(8 * u + e, g) : hseq16_channel(8 * u + e, g);
                    if (ch < 0) continue;
                    // MAT_ALPHA_VIEWF keeps its 32-row tile 0 for alpha: rows of 16-tile o are
                    // alpha (o = 0, row 0) or folded-view row 16 (o - 1) + row
                    const int wrow = sg.mat == MAT_ALPHA_VIEWF ? (o == 0 ? row : 32 + 16 * (o - 1) + row) : 16 * o + row;
                    if (sg.mat == MAT_ALPHA_VIEWF && o == 0 && row > 0) continue;
                    const float wv = t.w(sg.mat, wrow, sg.colbase + ch);
                    const uint16_t hi = is_bf ? f32_to_bf16(wv) : f32_to_f16(wv);
                    std::memcpy(&out[base + lane * 16 + e * 2], &hi, 2);
                    if (src) { src->resize(out.size() / 2, -1); (*src)[(base + lane * 16 + e * 2) / 2] = src_entry(t.flat(sg.mat, wrow, sg.colbase + ch), SRC_PLAIN); }
                }
            }
        }
        if (sg.pad) out.resize((out.size() + CHUNK_BYTES - 1) / CHUNK_BYTES * CHUNK_BYTES, 0);
    }
    if (src) src->resize(out.size() / 2, -1);
    return out.size() == (size_t)(onchip ? R::NCHUNK_OC : R::NCHUNK) * CHUNK_BYTES ? 0 : -2;
}

// ---- compensated kernel with the out tiles split over the waves (pg_evalc2.hip, pg_program.h T): no stream, one
// 4-KiB block of four A fragments per (k-unit, wave): [tile t of the wave][plane], lane (g, row): k = 8 g + e ----
int pack_c2(const NetTensors& t, bool fc, std::vector<uint8_t>& out, std::vector<int32_t>* src) {
    if (t.viewf_w.size() != (size_t)VW * W) return -4;
    out.assign((size_t)T::TOTAL, 0);
    if (src) src->assign((size_t)T::TOTAL / 2, -1);
    auto comp_pair = [](float wv, uint16_t& p0, uint16_t& p1) {
        const double wd = (double)wv / COMP_S;
        const double w1 = f16_to_f32(f32_to_f16((float)wd));
        p0 = f32_to_f16((float)((COMP_S - 1) * w1));
        p1 = f32_to_f16((float)(w1 + COMP_S * (wd - w1)));
    };
    auto put = [&](size_t frag0, int lane, int e, int mat, int row, int col) {       // planes at frag0 and frag0 + FRAG
        uint16_t p0, p1;
        comp_pair(t.w(mat, row, col), p0, p1);
        std::memcpy(&out[frag0 + lane * 16 + e * 2], &p0, 2);
        std::memcpy(&out[frag0 + T::FRAG + lane * 16 + e * 2], &p1, 2);
        if (src) {
            (*src)[(frag0 + lane * 16 + e * 2) / 2] = src_entry(t.flat(mat, row, col), SRC_COMP0);
            (*src)[(frag0 + T::FRAG + lane * 16 + e * 2) / 2] = src_entry(t.flat(mat, row, col), SRC_COMP1);
        }
    };
    // trunk sections: `xseq` = the density input in X16 order, else the previous activation (hseq16_channel) at column `colbase`
    auto trunk = [&](size_t off, int mat, int nu, bool xseq, int colbase) {
        for (int u = 0; u < nu; ++u)
            for (int w = 0; w < T::NW; ++w)
                for (int tt = 0; tt < 2; ++tt) {
                    const size_t f0 = off + ((size_t)(u * T::NW + w) * 4 + tt * 2) * T::FRAG;
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int g = lane >> 4, row = lane & 15;
                            const int ch = xseq ? xseq16_channel(8 * u + e, g) : hseq16_channel(8 * u + e, g);
                            if (ch < 0) continue;
                            put(f0, lane, e, mat, 16 * (2 * w + tt) + row, colbase + ch);
                        }
                }
    };
    trunk(T::OFF_X0, MAT_L0, XU16, true, 0);
    for (int hs = 0; hs < 7; ++hs) trunk(T::OFF_HID(hs), MAT_L0 + 1 + hs, HU16, false, hs == 4 ? CH_X : 0);
    trunk(T::OFF_X5, MAT_L0 + 5, XU16, true, 0);
    for (int u = 0; u < HU16; ++u)                      // folded view layer, trunk part: tile pair v = tiles 2 v, 2 v + 1
        for (int v = 0; v < 4; ++v)
            for (int tt = 0; tt < 2; ++tt) {
                const size_t f0 = T::OFF_AV + ((size_t)(u * 4 + v) * 4 + tt * 2) * T::FRAG;
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e)
                        put(f0, lane, e, MAT_VIEWF, 16 * (2 * v + tt) + (lane & 15), hseq16_channel(8 * u + e, lane >> 4));
            }
    for (int s = 0; s < J + (fc ? 1 : 0); ++s)          // direction part: B fragments [slot][out tile16]
        for (int tt = 0; tt < NTV16; ++tt) {
            const size_t f0 = T::OFF_Y + ((size_t)(s * NTV16 + tt) * 2) * T::FRAG;
            const int j = s < J ? slot16_joint(s) : JC;
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int ch = vd_channel(j, 8 * (lane >> 4) + e);
                    if (ch < 0) continue;
                    put(f0, lane, e, MAT_VIEW, 16 * tt + (lane & 15), W + ch);
                }
        }
    auto put_small = [&](size_t base, size_t plane_stride, int e, int mat, int row, int col) {
        uint16_t p0, p1;
        comp_pair(t.w(mat, row, col), p0, p1);
        std::memcpy(&out[base + e * 2], &p0, 2);
        std::memcpy(&out[base + plane_stride + e * 2], &p1, 2);
        if (src) {
            (*src)[(base + e * 2) / 2] = src_entry(t.flat(mat, row, col), SRC_COMP0);
            (*src)[(base + plane_stride + e * 2) / 2] = src_entry(t.flat(mat, row, col), SRC_COMP1);
        }
    };
    for (int u = 0; u < HU16; ++u)                      // alpha: entry g of [u][plane] = row 0, k = 8 g + e
        for (int g = 0; g < 4; ++g)
            for (int e = 0; e < 8; ++e)
                put_small(T::OFF_SMALL + (size_t)(u * 2) * T::ALPHA_STRIDE + g * 16, T::ALPHA_STRIDE, e, MAT_ALPHA, 0, hseq16_channel(8 * u + e, g));
    for (int u = 0; u < VW / 32; ++u)                   // rgb: entry 3 g + row of [u][plane]
        for (int g = 0; g < 4; ++g)
            for (int row = 0; row < 3; ++row)
                for (int e = 0; e < 8; ++e)
                    put_small(T::OFF_SMALL + T::SMALL_ALPHA + (size_t)(u * 2) * T::RGB_STRIDE + (3 * g + row) * 16, T::RGB_STRIDE, e,
                              MAT_RGB, row, hseq16_channel(8 * u + e, g));
    return 0;
}

void pack_bias_s(const NetTensors& t, std::vector<float>& out, std::vector<int32_t>* src) {
    out.assign(BIAS16_FLOATS, 0.f);
    if (src) src->assign(BIAS16_FLOATS, -1);
    auto put = [&](int tile, const float* b, int tensor, int n, int row0) {
        for (int g = 0; g < G16; ++g)
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * g + r;
                out[(size_t)tile * 16 + 4 * g + r] = (b && row < n) ? b[row] : 0.f;
                if (src && b && row < n) (*src)[(size_t)tile * 16 + 4 * g + r] = src_entry(t.flat_bias(tensor, row), SRC_PLAIN);
            }
    };
    for (int l = 0; l < DEPTH; ++l)
        for (int o = 0; o < NT16; ++o) put(BS_LAYER0 + l * NT16 + o, t.lb[l], 2 * l + 1, W, 16 * o);
    put(BS_ALPHA, t.alpha_b, 17, 1, 0);
    for (int o = 0; o < NTV16; ++o) put(BS_VIEWF + o, t.viewf_b.data(), NetTensors::SRC_VIEWF_B, VW, 16 * o);
    put(BS_RGB, t.rgb_b, 23, 3, 0);
}

void pack_bias(const NetTensors& t, std::vector<float>& out) {
    out.assign(BIAS_FLOATS, 0.f);
    auto put = [&](int tile, const float* b, int n, int row0) {
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) {
                int row = row0 + rho(r, h);
                out[(size_t)(tile * 2 + h) * 16 + r] = (b && row < n) ? b[row] : 0.f;
            }
    };
    for (int l = 0; l < DEPTH; ++l)
        for (int o = 0; o < NT; ++o) put(BT_LAYER0 + l * NT + o, t.lb[l], W, 32 * o);
    for (int o = 0; o < NT; ++o) put(BT_FEAT + o, t.feat_b, W, 32 * o);
    put(BT_ALPHA, t.alpha_b, 1, 0);
    for (int o = 0; o < NTV; ++o) put(BT_VIEW + o, t.view_b, VW, 32 * o);
    put(BT_RGB, t.rgb_b, 3, 0);
    if (t.viewf_b.size() == (size_t)VW)
        for (int o = 0; o < NTV; ++o) put(BT_VIEWF + o, t.viewf_b.data(), VW, 32 * o);
}

}  // namespace pgpack
