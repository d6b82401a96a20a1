// pg_api.hip -- C ABI of libposegen_hip.so (include/posegen_hip.h): handle, weight
// packing / upload, workspace, and the launch sequence of one render_rays call.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <thread>
#include <new>
#include <utility>
#include <vector>

#include "../../include/posegen_hip.h"
#include "pg_device.h"
#include "pg_handle.h"
#include "pg_pack.h"

extern "C" {
int pg_launch_eval16(const pgd::EvalArgs* a, int fp16, int framecode, int grid, void* stream);
int pg_launch_eval16r(const pgd::EvalArgs* a, int fp16, int framecode, int onchip, int grid, void* stream);
int pg_launch_ray_records(const pgd::RecArgs* a, int fp16, int framecode, int n_cu, void* stream);
int pg_eval16_points_per_pass(void);
int pg_eval16_wgs_per_cu(void);
int pg_launch_eval32(const pgd::EvalArgs* a, int precision, int framecode, int grid, void* stream);
int pg_eval32_points_per_pass(void);
int pg_launch_evalc(const pgd::EvalArgs* a, int framecode, int rec, int grid, void* stream);
int pg_launch_ray_records_c(const pgd::RecArgs* a, int framecode, int n_cu, void* stream);
int pg_evalc_points_per_pass(void);
int pg_launch_evalc2(const pgd::EvalArgs* a, int framecode, int grid, void* stream);
int pg_evalc2_points_per_pass(void);
int pg_launch_sample_coarse(const float* rays, const float* cyls, long long cyl_stride, long long n, int chunk,
                            int S, int lindisp, float* near_far, float* z, const float* t_rand, double* scratch, void* stream);
long long pg_sample_coarse_scratch(long long n, int chunk);
int pg_launch_gather_noise(const float* src, long long n, int stride, int S, const int* order, float* dst, void* stream);
int pg_launch_mfma_rate(int f16, int lds_fed, int blocks, int iters, float* sink, void* stream);
int pg_launch_composite(const float* rays, const float* z, const float* raw, long long n, int S,
                        float density_scale, float rgb_eps, int density_act, float act_shift, float* rgb, float* disp, float* acc,
                        float* alpha, float* weights, int n_imp, float* z_fine, const float* noise, const float* u_rand, int* order,
                        void* stream);
int pg_composite_max_samples(void);
int pg_composite_max_importance(void);
}
namespace pgk {
struct FrameGeom {
    int H, W, tlx, tly, bw, bh;
    float fx, fy, cx, cy;
    float R[9], t[3];
    float near, far, cam;
};
}
extern "C" {
int pg_launch_frame_rays(const pgk::FrameGeom* g, long long i0, long long n, float* rays, float* cams, void* stream);
int pg_launch_pose_kinematics(const double* offs72, const int* parents24, const double* bones, long long n,
                              float* kps, float* skts, double* l2ws, void* stream);
int pg_launch_pose_boxes(const float* kps, long long n, const double* w2c, long long w2c_stride, const double* ring,
                         float ext_r, float ext_top, float ext_bot, double fx, double fy, int H, int W, int offx, int offy,
                         float* cyls, int* boxes, void* stream);
int pg_launch_frame_compose(const pgk::FrameGeom* g, const float* rgb_map, const float* disp_map, const float* acc_map,
                            const float* bg, float base_bg, float* rgb, float* disp, float* acc, uint8_t* rgb8,
                            void* stream);
// pg_repack.hip: packed images re-formed on the device (pg_load_weights_device)
void pg_launch_collect(const float* const* tensors, const long long* off25, float* dst, void* stream);
void pg_launch_fold(float* src, long long off_view_w, int vcols, long long off_view_b, long long off_feat_w, long long off_feat_b,
                    long long off_fw, long long off_fb, void* stream);
void pg_launch_gather16(const int32_t* map, const float* src, uint16_t* out, long long n, int is_bf, void* stream);
void pg_launch_gather32(const int32_t* map, const float* src, float* out, long long n, void* stream);
void pg_launch_codes(const float* codes, int n_codes, float* out, void* stream);
void pg_launch_ycode(const float* view_w, int vcols, const float* codes, int n_codes, float* yc, void* stream);
}

namespace {

using namespace pgl;

thread_local char g_last_error[512] = "";      // pg_last_error(NULL): per host thread

void frames_cache_release(pg_handle* h);       // per-device buffers of pg_render_frames (defined beside it)

}  // namespace

int pg_fail(pg_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    std::snprintf(g_last_error, sizeof g_last_error, "%s", buf);
    if (h) std::snprintf(h->err, sizeof h->err, "%s", buf);
    return code;
}

namespace {

// state setters are forwarded to the sub-handles of a multi-device handle (message of a failing one is kept)
#define PG_FORWARD(h, call)                                                                  \
    do {                                                                                     \
        for (pg_handle* sub_ : (h)->peers) {                                                 \
            pg_handle* hh = sub_;                                                            \
            const int rc_ = (call);                                                          \
            if (rc_) return pg_fail(h, rc_, "device %d: %s", hh->device, hh->err);              \
        }                                                                                    \
    } while (0)

bool is_shape_a(int prec) { return prec == PG_PREC_BF16 || prec == PG_PREC_FP16; }

// bf16x3 (every product as hi*hi + hi*lo + lo*hi of bf16 halves) is the 1e-4-grade mode at MFMA
// speed.  fp16x3 stays EXPERIMENTAL: the low half of a small value underflows fp16's exponent
// range, so it is no better than plain fp16 there (DESIGN.md "Known issues"); opt-in only.
bool x3_allowed() {
    const char* e = std::getenv("POSEGEN_EXPERIMENTAL_X3");
    return e && e[0] == '1';
}
bool is_x3(int prec) { return prec == PG_PREC_FP16X3; }

pgpack::NetTensors tensors_of(const NetState& ns, const pg_config& cfg) {
    pgpack::NetTensors t;
    for (int l = 0; l < DEPTH; ++l) {
        t.lw[l] = ns.host[2 * l].data();
        t.lb[l] = ns.host[2 * l + 1].data();
        t.lcols[l] = l == 0 ? CH_X : (l == SKIP + 1 ? CH_X + W : W);
    }
    t.alpha_w = ns.host[16].data(); t.alpha_b = ns.host[17].data();
    t.feat_w = ns.host[18].data();  t.feat_b = ns.host[19].data();
    t.view_w = ns.host[20].data();  t.view_b = ns.host[21].data();
    t.view_cols = W + CH_D + cfg.framecode_ch;
    t.rgb_w = ns.host[22].data();   t.rgb_b = ns.host[23].data();
    if (ns.fold_w.empty()) { t.fold(); ns.fold_w = t.viewf_w; ns.fold_b = t.viewf_b; }      // (every packer of the net shares it)
    else { t.viewf_w = ns.fold_w; t.viewf_b = ns.fold_b; }
    return t;
}

// ... of a handle's net: first the host copies are brought up to date if the last weights came from the device
// (pg_load_weights_device leaves them stale: only the images it re-forms itself are current)
int refresh_host(pg_handle* h, NetState& ns) {
    if (!ns.host_stale) return PG_OK;
    pgpack::NetTensors lay;
    lay.layout(h->cfg.framecode_ch);
    PG_HIP(h, hipSetDevice(h->device));
    PG_HIP(h, hipDeviceSynchronize());
    for (int i = 0; i < 24; ++i)
        PG_HIP(h, hipMemcpy(ns.host[i].data(), ns.d_src + lay.off[i], ns.host[i].size() * sizeof(float), hipMemcpyDeviceToHost));
    ns.fold_w.clear(); ns.fold_b.clear();
    if (ns.d_codes && !ns.codes_host.empty())
        PG_HIP(h, hipMemcpy(ns.codes_host.data(), ns.d_codes, ns.codes_host.size() * sizeof(float), hipMemcpyDeviceToHost));
    ns.host_stale = false;
    std::vector<float> bias;
    pgpack::pack_bias(tensors_of(ns, h->cfg), bias);
    if (!ns.d_bias) PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_bias), BIAS_FLOATS * sizeof(float)));
    PG_HIP(h, hipMemcpy(ns.d_bias, bias.data(), BIAS_FLOATS * sizeof(float), hipMemcpyHostToDevice));
    return PG_OK;
}

// The 16-bit precisions factorise the view layer over rays when a pass cannot touch more than MAXR_F rays
// (pg_layout.h): per-ray records (pg_rayrec.hip) + the 16x16x32 kernel (pg_eval16r.hip); POSEGEN_VIEW_FACT=0
// forces the direct 32x32x16 kernel (pg_eval16.hip) everywhere (A/B, debugging).
bool use_fact(int prec, int S) {
    static const bool allowed = [] { const char* e = std::getenv("POSEGEN_VIEW_FACT"); return !(e && e[0] == '0'); }();
    return allowed && is_shape_a(prec) && S >= FACT_MIN_S;
}

// PG_PREC_FP16C runs in its dedicated kernel (pg_evalc.hip) when a ray has >= COMP_MIN_S samples (then a
// 128-point pass touches <= MAXR_C rays) and the points come from rays; otherwise (and with
// POSEGEN_COMP_KERNEL=0, for A/B) in the k-major kernel of pg_eval32.hip, same arithmetic.
bool use_comp_kernel(int prec, int S, bool points) {
    static const bool allowed = [] { const char* e = std::getenv("POSEGEN_COMP_KERNEL"); return !(e && e[0] == '0'); }();
    return allowed && prec == PG_PREC_FP16C && !points && S >= COMP_MIN_S;
}

// ... and in that kernel's record variant when a ray has >= FACT_MIN_S samples (per-ray records of
// ray_records_c_kernel instead of the direct view layer); POSEGEN_COMP_REC=0 forces the direct form (A/B).
bool use_comp_rec(int S) {
    static const bool allowed = [] { const char* e = std::getenv("POSEGEN_COMP_REC"); return !(e && e[0] == '0'); }();
    return allowed && S >= FACT_MIN_S;
}

int ensure_rec(pg_handle* h, int64_t n, int y_bytes) {
    const size_t need = (size_t)(n + REC_PAD_RAYS) * ((size_t)y_bytes + REC_AB_BYTES);
    if (need <= h->rec_bytes) return PG_OK;
    PG_HIP(h, hipSetDevice(h->device));
    if (h->rec) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(h->rec)); h->rec = nullptr; h->rec_bytes = 0; }
    const size_t want = need + need / 16;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->rec), want);
    if (e != hipSuccess) return pg_fail(h, PG_ENOMEM, "ray record buffer of %zu bytes failed: %s", want, hipGetErrorString(e));
    h->rec_bytes = want;
    h->rec_pad_n = -1;
    return PG_OK;
}

int ensure_stream_r(pg_handle* h, int which, int prec) {
    NetState& ns = h->net[which];
    if (!ns.loaded) return pg_fail(h, PG_ESTATE, "weights of net %d not loaded", which);
    if (ns.d_stream_r[prec] && ns.d_bias_s && ns.d_vy[prec]) return PG_OK;
    if (const int rr_ = refresh_host(h, ns)) return rr_;        // (the last weights may have come from the device)
    const pgpack::NetTensors t = tensors_of(ns, h->cfg);        // (folds feature_linear into the view layer: milliseconds of host work)
    PG_HIP(h, hipSetDevice(h->device));
    if (!ns.d_stream_r[prec]) {
        std::vector<uint8_t> packed;
        const int rc = pgpack::pack_stream_r(t, prec, packed);
        if (rc != 0) return pg_fail(h, PG_EINVAL, "16x16x32 weight stream packing failed (%d) for precision %d", rc, prec);
        PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_stream_r[prec]), packed.size()));
        PG_HIP(h, hipMemcpy(ns.d_stream_r[prec], packed.data(), packed.size(), hipMemcpyHostToDevice));
    }
    if (!ns.d_bias_s) {
        std::vector<float> b;
        pgpack::pack_bias_s(t, b);
        PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_bias_s), b.size() * sizeof(float)));
        PG_HIP(h, hipMemcpy(ns.d_bias_s, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (!ns.d_vy[prec]) {
        std::vector<uint8_t> vy;
        if (pgpack::pack_vy(t, prec, h->cfg.framecode_ch > 0, vy) != 0)
            return pg_fail(h, PG_EINVAL, "Y-stage weight packing failed for precision %d", prec);
        PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_vy[prec]), vy.size()));
        PG_HIP(h, hipMemcpy(ns.d_vy[prec], vy.data(), vy.size(), hipMemcpyHostToDevice));
    }
    return PG_OK;
}

// the on-chip variant of the 16x16x32 kernel (no per-ray records): its stream; the bias table is shared
int ensure_stream_ro(pg_handle* h, int which, int prec) {
    NetState& ns = h->net[which];
    if (!ns.loaded) return pg_fail(h, PG_ESTATE, "weights of net %d not loaded", which);
    if (ns.d_stream_ro[prec] && ns.d_bias_s) return PG_OK;
    if (const int rr_ = refresh_host(h, ns)) return rr_;        // (the last weights may have come from the device)
    const pgpack::NetTensors t = tensors_of(ns, h->cfg);
    PG_HIP(h, hipSetDevice(h->device));
    if (!ns.d_stream_ro[prec]) {
        std::vector<uint8_t> packed;
        const int rc = pgpack::pack_stream_r(t, prec, packed, true);
        if (rc != 0) return pg_fail(h, PG_EINVAL, "on-chip 16x16x32 weight stream packing failed (%d) for precision %d", rc, prec);
        PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_stream_ro[prec]), packed.size()));
        PG_HIP(h, hipMemcpy(ns.d_stream_ro[prec], packed.data(), packed.size(), hipMemcpyHostToDevice));
    }
    if (!ns.d_bias_s) {
        std::vector<float> b;
        pgpack::pack_bias_s(t, b);
        PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_bias_s), b.size() * sizeof(float)));
        PG_HIP(h, hipMemcpy(ns.d_bias_s, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return PG_OK;
}

// ... which runs for rays of at most ONCHIP_MAX_S samples.  The 16x16x32 kernel's on-chip variant takes per-ray poses and
// frame codes too; pg_evalc.hip's on-chip form needs one pose per launch and no frame codes (`plain_only`) and has no
// sample-count rule.  The on-chip variant forms a ray's rows in every pass the ray has points in, the record variant once
// per ray in a kernel in front: measured on one box (profiles/r5_ab_onchip_by_samples.txt, bf16 512 x 512 frames) the two
// tie at 64 + 16 samples (31.7 / 31.8 ms), on-chip wins at 96 + 16 (43.3 / 43.6) and records win from 128 + 16 on (59.3 /
// 58.0; with frame codes 59.7 / 58.2) -- at a cost of 8.75 KiB of HBM per ray.  pg_set_onchip (initial value: POSEGEN_ONCHIP = 0 / 1 / 2)
// forces the record variants (0) or the on-chip ones whatever the sample count (2).
constexpr int ONCHIP_MAX_S = 112;
int onchip_mode_from_env() {
    const char* e = std::getenv("POSEGEN_ONCHIP");
    return e && e[0] == '0' ? PG_ONCHIP_RECORDS : e && e[0] == '2' ? PG_ONCHIP_ALWAYS : PG_ONCHIP_AUTO;
}
bool use_onchip(const pg_handle* h, bool fc, long long pose_stride, int S, bool plain_only = false) {
    const int mode = h->onchip_mode;
    if (plain_only) return mode != PG_ONCHIP_RECORDS && !fc && pose_stride == 0;
    return mode == PG_ONCHIP_ALWAYS || (mode == PG_ONCHIP_AUTO && S <= ONCHIP_MAX_S);
}

// The frame code's part of the view layer for every code (and the mean row, embedding.py:25-26), as the on-chip variant reads it:
// Yc[c][o] = sum_k W_view[o][256 + 648 + k] codes[c][k] in fp32 (sums in double) -- 16 products per value once per
// pg_set_framecodes / pg_load_weights instead of once per ray.
int ensure_ycode(pg_handle* h, int which) {
    NetState& ns = h->net[which];
    if (ns.d_ycode) return PG_OK;
    if (!ns.loaded || ns.codes_host.empty()) return pg_fail(h, PG_ESTATE, "frame codes of net %d not set (pg_set_framecodes)", which);
    if (const int rr_ = refresh_host(h, ns)) return rr_;        // (the last weights may have come from the device)
    const int vcols = W + CH_D + FC_CH;
    const std::vector<float>& wv = ns.host[20];
    std::vector<float> yc((size_t)(ns.n_codes + 1) * VW);
    for (int c = 0; c <= ns.n_codes; ++c)
        for (int o = 0; o < VW; ++o) {
            double s = 0.0;
            for (int k = 0; k < FC_CH; ++k) s += (double)wv[(size_t)o * vcols + W + CH_D + k] * (double)ns.codes_host[(size_t)c * FC_CH + k];
            yc[(size_t)c * VW + o] = (float)s;
        }
    PG_HIP(h, hipSetDevice(h->device));
    PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_ycode), yc.size() * sizeof(float)));
    PG_HIP(h, hipMemcpy(ns.d_ycode, yc.data(), yc.size() * sizeof(float), hipMemcpyHostToDevice));
    return PG_OK;
}

int ensure_stream_cr(pg_handle* h, int which) {
    NetState& ns = h->net[which];
    if (!ns.loaded) return pg_fail(h, PG_ESTATE, "weights of net %d not loaded", which);
    if (ns.d_stream_cr && ns.d_vyc) return PG_OK;
    if (const int rr_ = refresh_host(h, ns)) return rr_;        // (the last weights may have come from the device)
    const pgpack::NetTensors t = tensors_of(ns, h->cfg);
    PG_HIP(h, hipSetDevice(h->device));
    if (!ns.d_stream_cr) {
        std::vector<uint8_t> packed;
        const int rc = pgpack::pack_stream(t, PG_PREC_FP16C, h->cfg.framecode_ch > 0, true, packed, nullptr, true);
        if (rc != 0) return pg_fail(h, PG_EINVAL, "compensated-fp16 record-variant stream packing failed (%d)", rc);
        PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_stream_cr), packed.size()));
        PG_HIP(h, hipMemcpy(ns.d_stream_cr, packed.data(), packed.size(), hipMemcpyHostToDevice));
    }
    if (!ns.d_vyc) {
        std::vector<float> vy;
        pgpack::pack_vyc(t, h->cfg.framecode_ch > 0, vy);
        PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_vyc), vy.size() * sizeof(float)));
        PG_HIP(h, hipMemcpy(ns.d_vyc, vy.data(), vy.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return PG_OK;
}

// the on-chip form of that variant (one pose per launch, no frame codes): its stream
int ensure_stream_co(pg_handle* h, int which) {
    NetState& ns = h->net[which];
    if (!ns.loaded) return pg_fail(h, PG_ESTATE, "weights of net %d not loaded", which);
    if (ns.d_stream_co) return PG_OK;
    std::vector<uint8_t> packed;
    if (const int rr_ = refresh_host(h, ns)) return rr_;        // (the last weights may have come from the device)
    const int rc = pgpack::pack_stream(tensors_of(ns, h->cfg), PG_PREC_FP16C, false, true, packed, nullptr, true, true);
    if (rc != 0) return pg_fail(h, PG_EINVAL, "compensated-fp16 on-chip stream packing failed (%d)", rc);
    PG_HIP(h, hipSetDevice(h->device));
    PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_stream_co), packed.size()));
    PG_HIP(h, hipMemcpy(ns.d_stream_co, packed.data(), packed.size(), hipMemcpyHostToDevice));
    return PG_OK;
}

// PG_PREC_FP16C with >= pgp::T::MIN_S samples per ray runs in the kernel that splits the OUT TILES over the waves
// (pg_evalc2.hip: two waves per SIMD, activations in LDS, weights straight from L2) -- whatever the pose stride, with or
// without frame codes; POSEGEN_EVALC2=0 keeps the calls on pg_evalc.hip (A/B, and the record / on-chip forms' tests)
bool use_evalc2(int S) {
    static const bool allowed = [] { const char* e = std::getenv("POSEGEN_EVALC2"); return !(e && e[0] == '0'); }();
    return allowed && S >= pgp::T::MIN_S;
}

// its weights (pg_program.h T) and the 16-row bias table it shares with the 16x16x32 kernel
int ensure_c2(pg_handle* h, int which) {
    NetState& ns = h->net[which];
    if (!ns.loaded) return pg_fail(h, PG_ESTATE, "weights of net %d not loaded", which);
    if (ns.d_c2 && ns.d_bias_s) return PG_OK;
    if (const int rr_ = refresh_host(h, ns)) return rr_;        // (the last weights may have come from the device)
    const pgpack::NetTensors t = tensors_of(ns, h->cfg);
    PG_HIP(h, hipSetDevice(h->device));
    if (!ns.d_c2) {
        std::vector<uint8_t> packed;
        const int rc = pgpack::pack_c2(t, h->cfg.framecode_ch > 0, packed);
        if (rc != 0) return pg_fail(h, PG_EINVAL, "compensated-fp16 tile-split weight packing failed (%d)", rc);
        PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_c2), packed.size()));
        PG_HIP(h, hipMemcpy(ns.d_c2, packed.data(), packed.size(), hipMemcpyHostToDevice));
    }
    if (!ns.d_bias_s) {
        std::vector<float> b;
        pgpack::pack_bias_s(t, b);
        PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_bias_s), b.size() * sizeof(float)));
        PG_HIP(h, hipMemcpy(ns.d_bias_s, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return PG_OK;
}

int ensure_stream(pg_handle* h, int which, int prec, bool fact) {
    NetState& ns = h->net[which];
    if (!ns.loaded) return pg_fail(h, PG_ESTATE, "weights of net %d not loaded", which);
    if (ns.d_stream[prec][fact]) return PG_OK;
    std::vector<uint8_t> packed;
    if (const int rr_ = refresh_host(h, ns)) return rr_;        // (the last weights may have come from the device)
    const int rc = pgpack::pack_stream(tensors_of(ns, h->cfg), prec, h->cfg.framecode_ch > 0, fact, packed);
    if (rc != 0) return pg_fail(h, PG_EINVAL, "weight stream packing failed (%d) for precision %d", rc, prec);
    PG_HIP(h, hipSetDevice(h->device));
    PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_stream[prec][fact]), packed.size()));
    PG_HIP(h, hipMemcpy(ns.d_stream[prec][fact], packed.data(), packed.size(), hipMemcpyHostToDevice));
    ns.stream_bytes[prec][fact] = packed.size();
    return PG_OK;
}

// the packed weight stream a precision mode will use for net `which` in the usual call (rays with >= 64 samples; one pose
// per call unless the config has frame codes), built ahead of the first render; the other forms of the mode (per-ray poses,
// short rays, explicit points) are packed by the first call that needs them (launch_eval_one)
int ensure_mode_streams(pg_handle* h, int which, int mode) {
    auto one = [&](int prec) {
        const bool fc = h->cfg.framecode_ch > 0;
        if (is_shape_a(prec) && use_fact(prec, FACT_MIN_S)) return use_onchip(h, fc, 0, FACT_MIN_S) ? ensure_stream_ro(h, which, prec) : ensure_stream_r(h, which, prec);
        if (prec == PG_PREC_FP16C && use_evalc2(FACT_MIN_S)) return ensure_c2(h, which);
        if (prec == PG_PREC_FP16C && use_comp_rec(FACT_MIN_S)) return use_onchip(h, fc, 0, FACT_MIN_S, true) ? ensure_stream_co(h, which) : ensure_stream_cr(h, which);
        return ensure_stream(h, which, prec, use_fact(prec, FACT_MIN_S));
    };
    if (mode != PG_PREC_FP16M) return one(mode);
    int rc = one(PG_PREC_FP16C);
    if (!rc && which == 0) rc = one(PG_PREC_FP16);
    return rc;
}

int ensure_ws(pg_handle* h, size_t bytes) {
    if (bytes <= h->ws_bytes) return PG_OK;
    PG_HIP(h, hipSetDevice(h->device));
    if (h->ws) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(h->ws)); h->ws = nullptr; h->ws_bytes = 0; }
    const size_t want = bytes + bytes / 8;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->ws), want);
    if (e != hipSuccess) return pg_fail(h, PG_ENOMEM, "workspace allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
    h->ws_bytes = want;
    return PG_OK;
}

int check_ready(pg_handle* h, bool need_fine) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "null handle");
    if (!h->net[0].loaded) return pg_fail(h, PG_ESTATE, "coarse network weights not loaded (pg_load_weights)");
    if (need_fine && !h->net[1].loaded) return pg_fail(h, PG_ESTATE, "fine network weights not loaded (pg_load_weights)");
    if (!h->emb_set[0] || !h->emb_set[1]) return pg_fail(h, PG_ESTATE, "embedder state not set (pg_set_embedder)");
    return PG_OK;
}

int launch_eval_one(pg_handle* h, void* stream, int which, long long n, int S, const float* rays, const float* z,
                    const float* skts, long long pose_stride, const float* cams, float* raw, float* dbg, int dbg_stage,
                    const float* points, const float* pnoise, bool guide_pass) {
    // PG_PREC_FP16M: the coarse pass of a hierarchical render only places the importance samples (and fills
    // rgb0/acc0): plain fp16 there, compensated fp16 wherever the pass produces the returned maps
    const int prec = h->cfg.precision == PG_PREC_FP16M ? (guide_pass ? PG_PREC_FP16 : PG_PREC_FP16C) : h->cfg.precision;
    // explicit points and position noise need q = R p + t per point: the direct kernels (no per-ray a + z b table)
    const bool compk = !pnoise && use_comp_kernel(prec, S, points != nullptr);
    const bool fact = compk || (!points && !pnoise && use_fact(prec, S));
    const bool sa = is_shape_a(prec);
    const bool fc = h->cfg.framecode_ch > 0;
    const bool onchip = sa && fact && (!dbg || (dbg_stage == 97 && pose_stride == 0 && !fc)) && use_onchip(h, fc, pose_stride, S);     // the 16x16x32 kernel without per-ray records (97: its limb-mask counters)
    const bool recs = sa && fact && !onchip;                      // per-ray records + the 16x16x32 kernel
    const bool c2 = compk && (!dbg || dbg_stage == 99 || dbg_stage == 97) && use_evalc2(S);      // out tiles over the waves (pg_evalc2.hip): any pose stride, frame codes or not
    const bool conchip = !c2 && compk && use_comp_rec(S) && (!dbg || dbg_stage == 98 || dbg_stage == 99) && use_onchip(h, fc, pose_stride, S, true);   // (98 / 99: diagnosis builds' dumps)   // the record variant of pg_evalc.hip without per-ray records
    const bool crec = !c2 && compk && use_comp_rec(S) && !conchip;       // per-ray records + the record variant of pg_evalc.hip
    int rc = onchip ? ensure_stream_ro(h, which, prec) : recs ? ensure_stream_r(h, which, prec) : c2 ? ensure_c2(h, which)
           : conchip ? ensure_stream_co(h, which) : crec ? ensure_stream_cr(h, which) : ensure_stream(h, which, prec, fact);
    if (rc) return rc;
    if (onchip && fc && (rc = ensure_ycode(h, which))) return rc;
    const int y_bytes = crec ? RECC_Y_BYTES : REC_Y_BYTES;
    if ((recs || crec) && (rc = ensure_rec(h, n, y_bytes))) return rc;
    NetState& ns = h->net[which];
    if (fc && !ns.d_codes) return pg_fail(h, PG_ESTATE, "frame codes of net %d not set (pg_set_framecodes)", which);
    pgd::EvalArgs a{};
    a.rays = rays; a.z = z; a.pts = points; a.pnoise = pnoise; a.skts = skts; a.cams = cams;
    a.codes = fc ? ns.d_codes : nullptr;
    a.wstream = onchip ? ns.d_stream_ro[prec] : recs ? ns.d_stream_r[prec] : c2 ? ns.d_c2 : conchip ? ns.d_stream_co : crec ? ns.d_stream_cr
              : ns.d_stream[prec][fact];
    a.wy = recs ? ns.d_vy[prec] : crec ? reinterpret_cast<const uint8_t*>(ns.d_vyc) : (onchip && fc) ? reinterpret_cast<const uint8_t*>(ns.d_ycode) : nullptr;
    a.bias = (recs || onchip || c2) ? ns.d_bias_s : ns.d_bias;
    if (recs || crec) {
        a.rec_y = h->rec;
        a.rec_ab = reinterpret_cast<const float*>(h->rec + (size_t)(n + REC_PAD_RAYS) * y_bytes);
        // the padding rays behind the last record are fetched by the last passes (their values are multiplied by
        // zero weights at most): keep them finite whatever the buffer held before.  The record kernels write rays
        // < n only, so the padding of an (n, record size) pair stays zero until another pair moves it.
        // (every writer of h->rec -- the two record kernels -- stores rays < n only; anything else that is ever handed the
        // buffer must reset rec_pad_n to -1, as ensure_rec does)
        if (h->rec_pad_n != n || h->rec_pad_y != y_bytes || h->rec_pad_stream != stream) {
            PG_HIP(h, hipMemsetAsync(h->rec + (size_t)n * y_bytes, 0, (size_t)REC_PAD_RAYS * y_bytes, static_cast<hipStream_t>(stream)));
            PG_HIP(h, hipMemsetAsync(h->rec + (size_t)(n + REC_PAD_RAYS) * y_bytes + (size_t)n * REC_AB_BYTES, 0,
                                     (size_t)REC_PAD_RAYS * REC_AB_BYTES, static_cast<hipStream_t>(stream)));
            h->rec_pad_n = n; h->rec_pad_y = y_bytes; h->rec_pad_stream = stream;
        }
    }
    a.cutoff = h->d_cut;
    a.raw = raw; a.dbg = dbg;
    a.pose_stride = pose_stride;
    a.n_points = n * S;
    a.n_rays = (int)n;
    a.S = S;
    a.n_codes = ns.n_codes;
    a.tau_v = h->tau[0];
    a.tau_d = h->tau[1];
    a.dbg_stage = dbg_stage;
    a.far_skip = h->far_skip ? 1 : 0;
    const int pts = sa ? pg_eval16_points_per_pass() : c2 ? pg_evalc2_points_per_pass() : compk ? pg_evalc_points_per_pass() : pg_eval32_points_per_pass();
    if (!points && S < pts / (MAXR - 1))      // explicit points are one pseudo ray: a pass touches one slot
        return pg_fail(h, PG_EINVAL, "N_samples=%d too small: the fused kernel needs >= %d samples per ray", S, pts / (MAXR - 1));
    const long long iters = (a.n_points + pts - 1) / pts;
    a.n_iters = (int)iters;
    // POSEGEN_MAX_WG (measurement aid): fewer persistent workgroups than CUs, to see how much of a pass's time
    // is contention between CUs (the weight stream is pulled from L2 by every CU) rather than its own work
    static const long long wg_cap = [] { const char* e = std::getenv("POSEGEN_MAX_WG"); return e ? std::atoll(e) : 0ll; }();
    long long max_wg = (long long)h->n_cu * (sa ? pg_eval16_wgs_per_cu() : 1);
    if (wg_cap > 0 && wg_cap < max_wg) max_wg = wg_cap;
    const int grid = (int)(iters < max_wg ? iters : max_wg);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto get = [&](hipEvent_t& ev) {
        if (!h->ev_free.empty()) { ev = h->ev_free.back(); h->ev_free.pop_back(); return hipSuccess; }
        return hipEventCreate(&ev);
    };
    if (recs || crec) {     // what depends on the ray only, once per ray, in front of the fused kernel (pg_rayrec.hip)
        pgd::RecArgs ra{};
        ra.rays = rays; ra.skts = skts; ra.cams = cams; ra.codes = a.codes; ra.wy = a.wy;
        ra.rec_ab = const_cast<float*>(a.rec_ab); ra.rec_y = const_cast<uint8_t*>(a.rec_y);
        ra.pose_stride = pose_stride; ra.n_rays = (int)n; ra.n_codes = ns.n_codes;
        ra.z = z; ra.S = S;
        hipEvent_t x0 = nullptr, x1 = nullptr;
        if (h->profiling) { PG_HIP(h, get(x0)); PG_HIP(h, get(x1)); PG_HIP(h, hipEventRecord(x0, static_cast<hipStream_t>(stream))); }
        const int er = crec ? pg_launch_ray_records_c(&ra, fc, h->n_cu, stream) : pg_launch_ray_records(&ra, prec == PG_PREC_FP16, fc, h->n_cu, stream);
        if (h->profiling) {
            PG_HIP(h, hipEventRecord(x1, static_cast<hipStream_t>(stream)));
            if (er) { h->ev_free.push_back(x0); h->ev_free.push_back(x1); }      // a failed launch is not a sample
            else h->ev_aux.emplace_back(x0, x1);
        }
        if (er) return pg_fail(h, PG_EHIP, "ray record kernel launch failed: %s", hipGetErrorString((hipError_t)er));
    }
    if (h->profiling) {
        PG_HIP(h, get(e0));
        PG_HIP(h, get(e1));
        PG_HIP(h, hipEventRecord(e0, static_cast<hipStream_t>(stream)));
    }
    int e = (recs || onchip) ? pg_launch_eval16r(&a, prec == PG_PREC_FP16, fc, onchip, grid, stream)
          : sa ? pg_launch_eval16(&a, prec == PG_PREC_FP16, fc, grid, stream)
          : c2 ? pg_launch_evalc2(&a, fc, grid, stream)
          : compk ? pg_launch_evalc(&a, fc, conchip ? 2 : crec ? 1 : 0, grid, stream)
               : pg_launch_eval32(&a, prec, fc, grid, stream);
    if (h->profiling) {
        PG_HIP(h, hipEventRecord(e1, static_cast<hipStream_t>(stream)));
        h->ev_used.emplace_back(e0, e1);
        h->prof_points += a.n_points;
    }
    if (e) return pg_fail(h, PG_EHIP, "fused embed+MLP kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return PG_OK;
}

// One net on n rays x S samples.  Calls with more than REC_BATCH_RAYS rays run as consecutive launches over ray
// ranges (rays are independent): the per-ray records (8.75 / 16.75 KiB per ray) then need 4.6 / 9 GB at most instead
// of growing with the call (a 2048 x 2048 frame would ask for 72 GB).  POSEGEN_REC_BATCH overrides the size (tests).
int launch_eval(pg_handle* h, void* stream, int which, long long n, int S, const float* rays, const float* z,
                const float* skts, long long pose_stride, const float* cams, float* raw, float* dbg, int dbg_stage = 0,
                const float* points = nullptr, const float* pnoise = nullptr, bool guide_pass = false) {
    long long batch = 1ll << 19;
    if (const char* e = std::getenv("POSEGEN_REC_BATCH")) { const long long v = std::atoll(e); if (v >= 64) batch = v; }
    if (points || dbg || n <= batch)
        return launch_eval_one(h, stream, which, n, S, rays, z, skts, pose_stride, cams, raw, dbg, dbg_stage, points, pnoise, guide_pass);
    for (long long r0 = 0; r0 < n; r0 += batch) {
        const long long m = std::min(batch, n - r0);
        const int rc = launch_eval_one(h, stream, which, m, S, rays + r0 * 11, z + r0 * S, skts + r0 * pose_stride, pose_stride,
                                       cams ? cams + r0 : nullptr, raw + r0 * S * 4, nullptr, 0, nullptr,
                                       pnoise ? pnoise + r0 * S * 3 : nullptr, guide_pass);
        if (rc) return rc;
    }
    return PG_OK;
}

}  // namespace

int pg_sc_scratch(pg_handle* h, long long n, int chunk, double** out) {
    *out = nullptr;
    const long long need = pg_sample_coarse_scratch(n, chunk);
    if (need <= 0) return PG_OK;
    if ((size_t)need > h->sc_part_cap) {
        PG_HIP(h, hipSetDevice(h->device));
        if (h->sc_part) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(h->sc_part)); h->sc_part = nullptr; h->sc_part_cap = 0; }
        const size_t want = (size_t)need * 2;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->sc_part), want * sizeof(double));
        if (e != hipSuccess) return pg_fail(h, PG_ENOMEM, "coarse sampler scratch of %zu doubles failed: %s", want, hipGetErrorString(e));
        h->sc_part_cap = want;
    }
    *out = h->sc_part;
    return PG_OK;
}

extern "C" {

int pg_abi_version(void) { return PG_ABI_VERSION; }

const char* pg_last_error(const pg_handle* h) { return h ? h->err : g_last_error; }

int pg_create(const pg_config* cfg, int n_devices, const int* device_ids, pg_handle** out) {
    if (!cfg || !out) return pg_fail(nullptr, PG_EINVAL, "pg_create: null argument");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return pg_fail(nullptr, PG_EINVAL, "pg_create: n_devices must be 1..64, got %d", n_devices);
    if (n_devices > 1 && !device_ids) return pg_fail(nullptr, PG_EINVAL, "pg_create: device_ids required for n_devices > 1");
    if (cfg->n_joints != J || cfg->multires != LV || cfg->multires_views != LD || cfg->multires_bones != 0 ||
        cfg->net_depth != DEPTH || cfg->net_width != W || cfg->skip_layer != SKIP || cfg->view_width != VW ||
        (cfg->framecode_ch != 0 && cfg->framecode_ch != FC_CH))
        return pg_fail(nullptr, PG_EINVAL,
                    "pg_create: unsupported architecture (kernels are built for 24 joints, multires 7/4/0, "
                    "8x256 trunk, skip 4, view width 128, frame code 0|16)");
    if (cfg->precision < 0 || cfg->precision >= PG_PREC_MODES) return pg_fail(nullptr, PG_EINVAL, "pg_create: bad precision %d", cfg->precision);
    if (is_x3(cfg->precision) && !x3_allowed())
        return pg_fail(nullptr, PG_EINVAL, "pg_create: split-operand precision %d is experimental (set POSEGEN_EXPERIMENTAL_X3=1)", cfg->precision);
    if (cfg->chunk <= 0) return pg_fail(nullptr, PG_EINVAL, "pg_create: chunk must be positive");
    if (!(cfg->density_scale > 0.f)) return pg_fail(nullptr, PG_EINVAL, "pg_create: density_scale must be positive");
    if (cfg->density_act != PG_ACT_RELU && cfg->density_act != PG_ACT_SOFTPLUS)
        return pg_fail(nullptr, PG_EINVAL, "pg_create: density_act must be PG_ACT_RELU or PG_ACT_SOFTPLUS, got %d", cfg->density_act);
    pg_handle* h = new (std::nothrow) pg_handle();
    if (h) h->onchip_mode = onchip_mode_from_env();
    if (!h) return pg_fail(nullptr, PG_ENOMEM, "pg_create: out of host memory");
    h->cfg = *cfg;
    h->device = device_ids ? device_ids[0] : 0;
    for (int i = 0; i < 48; ++i) h->cut[i] = cfg->cutoff_dist;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { delete h; return pg_fail(nullptr, PG_EHIP, "pg_create: no HIP device available (%s)", hipGetErrorString(e)); }
    if (h->device < 0 || h->device >= ndev) { delete h; return pg_fail(nullptr, PG_EINVAL, "pg_create: device %d out of range (%d devices)", cfg ? device_ids ? device_ids[0] : 0 : 0, ndev); }
    hipDeviceProp_t prop;
    if (hipSetDevice(h->device) != hipSuccess || hipGetDeviceProperties(&prop, h->device) != hipSuccess) {
        delete h;
        return pg_fail(nullptr, PG_EHIP, "pg_create: cannot query device");
    }
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->clock_khz = prop.clockRate;
    if (hipMalloc(reinterpret_cast<void**>(&h->d_cut), 48 * sizeof(float)) != hipSuccess) {
        delete h;
        return pg_fail(nullptr, PG_ENOMEM, "pg_create: device allocation failed");
    }
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&h->d_pose), (24 * 16 + 8) * sizeof(float)) != hipSuccess) {
        pg_destroy(h);
        return pg_fail(nullptr, PG_ENOMEM, "pg_create: stream / pose buffer creation failed");
    }
    // further devices: one sub-handle each (the same device may be listed twice: two workers on one GPU)
    for (int i = 1; i < n_devices; ++i) {
        pg_handle* sub = nullptr;
        const int rc = pg_create(cfg, 1, &device_ids[i], &sub);
        if (rc) { pg_destroy(h); return rc; }
        h->peers.push_back(sub);
    }
    // The gather of a cut frame (pg_render_frames phase B) is a device-to-device copy: it goes over xGMI only
    // with peer access enabled in both directions, otherwise the runtime stages it through the host -- refused
    // here rather than done silently (POSEGEN_ALLOW_STAGED_PEER=1 accepts the staged copies).
    if (n_devices > 1) {
        const char* ev = std::getenv("POSEGEN_ALLOW_STAGED_PEER");
        const bool allow_staged = ev && ev[0] == '1';
        for (int i = 0; i < n_devices; ++i)
            for (int j = 0; j < n_devices; ++j) {
                const int di = device_ids[i], dj = device_ids[j];
                if (di == dj) continue;
                int can = 0;
                hipError_t pe = hipDeviceCanAccessPeer(&can, di, dj);
                if (pe == hipSuccess && can) {
                    pe = hipSetDevice(di);
                    if (pe == hipSuccess) pe = hipDeviceEnablePeerAccess(dj, 0);
                    if (pe == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); pe = hipSuccess; }
                }
                if ((pe != hipSuccess || !can) && !allow_staged) {
                    const int rc = pg_fail(nullptr, PG_EHIP, "pg_create: device %d cannot access device %d directly (%s); frame gathers would be "
                                        "staged through the host (set POSEGEN_ALLOW_STAGED_PEER=1 to accept that)", di, dj,
                                        pe != hipSuccess ? hipGetErrorString(pe) : "hipDeviceCanAccessPeer = 0");
                    pg_destroy(h);
                    return rc;
                }
            }
        (void)hipSetDevice(h->device);
    }
    *out = h;
    return PG_OK;
}

void pg_destroy(pg_handle* h) {
    if (!h) return;
    for (pg_handle* sub : h->peers) pg_destroy(sub);
    h->peers.clear();
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    if (h->d_pose) (void)hipFree(h->d_pose);
    if (h->fws) (void)hipFree(h->fws);
    if (h->rec) (void)hipFree(h->rec);
    if (h->sc_part) (void)hipFree(h->sc_part);
    pg_train_release(h);
    frames_cache_release(h);
    for (auto& pr : h->ev_aux) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (NetState& ns : h->net) {
        for (auto& p : ns.d_stream_r) if (p) (void)hipFree(p);
        for (auto& p : ns.d_stream_ro) if (p) (void)hipFree(p);
        for (auto& pp : ns.d_stream) for (auto& p : pp) if (p) (void)hipFree(p);
        for (auto& p : ns.d_vy) if (p) (void)hipFree(p);
        if (ns.d_bias_s) (void)hipFree(ns.d_bias_s);
        if (ns.d_ycode) (void)hipFree(ns.d_ycode);
        if (ns.d_stream_cr) (void)hipFree(ns.d_stream_cr);
        if (ns.d_stream_co) (void)hipFree(ns.d_stream_co);
        if (ns.d_c2) (void)hipFree(ns.d_c2);
        if (ns.d_vyc) (void)hipFree(ns.d_vyc);
        if (ns.d_bias) (void)hipFree(ns.d_bias);
        if (ns.d_codes) (void)hipFree(ns.d_codes);
        if (ns.d_src) (void)hipFree(ns.d_src);
        if (ns.d_map_ro) (void)hipFree(ns.d_map_ro);
        if (ns.d_map_c2) (void)hipFree(ns.d_map_c2);
        if (ns.d_map_bias_s) (void)hipFree(ns.d_map_bias_s);
    }
    for (auto& pr : h->ev_used) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto& ev : h->ev_free) (void)hipEventDestroy(ev);
    if (h->d_cut) (void)hipFree(h->d_cut);
    if (h->ws) (void)hipFree(h->ws);
    delete h;
}

int pg_load_weights(pg_handle* h, int which, const float* const* tensors, const int64_t* shapes, int n_tensors) {
    if (!h || !tensors || !shapes) return pg_fail(h, PG_EINVAL, "pg_load_weights: null argument");
    if (which < 0 || which > 1) return pg_fail(h, PG_EINVAL, "pg_load_weights: which_net must be 0 or 1");
    if (n_tensors != 24) return pg_fail(h, PG_EINVAL, "pg_load_weights: expected 24 tensors, got %d", n_tensors);
    const int vcols = W + CH_D + h->cfg.framecode_ch;
    int64_t want[24][2];
    for (int l = 0; l < DEPTH; ++l) {
        want[2 * l][0] = W; want[2 * l][1] = l == 0 ? CH_X : (l == SKIP + 1 ? CH_X + W : W);
        want[2 * l + 1][0] = W; want[2 * l + 1][1] = 1;
    }
    want[16][0] = 1; want[16][1] = W;       want[17][0] = 1; want[17][1] = 1;
    want[18][0] = W; want[18][1] = W;       want[19][0] = W; want[19][1] = 1;
    want[20][0] = VW; want[20][1] = vcols;  want[21][0] = VW; want[21][1] = 1;
    want[22][0] = 3; want[22][1] = VW;      want[23][0] = 3; want[23][1] = 1;
    for (int i = 0; i < 24; ++i) {
        if (!tensors[i]) return pg_fail(h, PG_EINVAL, "pg_load_weights: tensor %d is null", i);
        if (shapes[2 * i] != want[i][0] || shapes[2 * i + 1] != want[i][1])
            return pg_fail(h, PG_EINVAL, "pg_load_weights: tensor %d has shape [%lld,%lld], expected [%lld,%lld]", i,
                        (long long)shapes[2 * i], (long long)shapes[2 * i + 1], (long long)want[i][0], (long long)want[i][1]);
    }
    NetState& ns = h->net[which];
    static const bool time_load = std::getenv("POSEGEN_TIME_LOAD") != nullptr;
    const auto tl0 = std::chrono::steady_clock::now();
    ns.host.assign(24, {});
    ns.fold_w.clear(); ns.fold_b.clear();
    for (int i = 0; i < 24; ++i) ns.host[i].assign(tensors[i], tensors[i] + want[i][0] * want[i][1]);
    ns.loaded = true;
    ns.host_stale = false;
    PG_HIP(h, hipSetDevice(h->device));
    for (int p = 0; p < PG_PREC_COUNT; ++p)
        for (int f = 0; f < 2; ++f)
            if (ns.d_stream[p][f]) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(ns.d_stream[p][f])); ns.d_stream[p][f] = nullptr; }
    for (int p = 0; p < PG_PREC_COUNT; ++p) {
        if (ns.d_vy[p]) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(ns.d_vy[p])); ns.d_vy[p] = nullptr; }
        if (ns.d_stream_r[p]) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(ns.d_stream_r[p])); ns.d_stream_r[p] = nullptr; }
        if (ns.d_stream_ro[p]) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(ns.d_stream_ro[p])); ns.d_stream_ro[p] = nullptr; }
    }
    if (ns.d_bias_s) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(ns.d_bias_s)); ns.d_bias_s = nullptr; }
    if (ns.d_ycode) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(ns.d_ycode)); ns.d_ycode = nullptr; }
    if (ns.d_stream_cr) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(ns.d_stream_cr)); ns.d_stream_cr = nullptr; }
    if (ns.d_stream_co) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(ns.d_stream_co)); ns.d_stream_co = nullptr; }
    if (ns.d_c2) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(ns.d_c2)); ns.d_c2 = nullptr; }
    if (ns.d_vyc) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(ns.d_vyc)); ns.d_vyc = nullptr; }
    std::vector<float> bias;
    pgpack::pack_bias(tensors_of(ns, h->cfg), bias);
    if (!ns.d_bias) PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_bias), BIAS_FLOATS * sizeof(float)));
    PG_HIP(h, hipMemcpy(ns.d_bias, bias.data(), BIAS_FLOATS * sizeof(float), hipMemcpyHostToDevice));
    const auto tl1 = std::chrono::steady_clock::now();
    const int rc0 = ensure_mode_streams(h, which, h->cfg.precision);
    if (rc0) return rc0;
    if (time_load) {
        const auto tl2 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "pg_load_weights net %d: copy + free + fold + bias %.2f ms, stream of the mode %.2f ms\n", which,
                     std::chrono::duration<double, std::milli>(tl1 - tl0).count(), std::chrono::duration<double, std::milli>(tl2 - tl1).count());
    }
    PG_FORWARD(h, pg_load_weights(hh, which, tensors, shapes, n_tensors));
    return PG_OK;
}

// New values for a loaded net's tensors (and frame codes) from DEVICE memory -- what sits between optimiser steps and a
// validation render (TrainableRayCaster.sync_inference_weights; the reference renders with the module it trains,
// core/trainer.py:463).  The images of the fast paths -- the on-chip stream of the 16x16x32 kernel (bf16 / fp16), pg_evalc2.hip's
// weight image, their bias table, the frame-code tables -- are re-formed on the device, bitwise as pg_load_weights would pack
// them (pg_repack.hip); every other image is dropped and re-packed from the host copies, which are refreshed from the device,
// by the first call that needs it.  Enqueued on `stream`; the tensors may be reused as soon as the call returns in stream order.
int pg_load_weights_device(pg_handle* h, void* stream, int which, const float* const* d_tensors, int n_tensors, const float* d_codes, int n_codes) {
    if (!h || !d_tensors) return pg_fail(h, PG_EINVAL, "pg_load_weights_device: null argument");
    if (which < 0 || which > 1) return pg_fail(h, PG_EINVAL, "pg_load_weights_device: which_net must be 0 or 1");
    if (n_tensors != 24) return pg_fail(h, PG_EINVAL, "pg_load_weights_device: expected 24 tensors, got %d", n_tensors);
    if (!h->peers.empty()) return pg_fail(h, PG_EINVAL, "pg_load_weights_device: a multi-device handle takes its weights from the host (pg_load_weights)");
    NetState& ns = h->net[which];
    if (!ns.loaded) return pg_fail(h, PG_ESTATE, "pg_load_weights_device: net %d has no weights yet (the first load is pg_load_weights: it fixes the shapes)", which);
    const bool fc = h->cfg.framecode_ch > 0;
    if (fc && (!d_codes || n_codes != ns.n_codes || !ns.d_codes))
        return pg_fail(h, PG_EINVAL, "pg_load_weights_device: frame codes [%d,16] expected (set once by pg_set_framecodes)", ns.n_codes);
    for (int i = 0; i < 24; ++i)
        if (!d_tensors[i]) return pg_fail(h, PG_EINVAL, "pg_load_weights_device: tensor %d is null", i);
    PG_HIP(h, hipSetDevice(h->device));
    pgpack::NetTensors lay;                                 // (offsets only; a source map is built from the shapes, see map_tensors)
    lay.layout(h->cfg.framecode_ch);
    auto map_tensors = [&]() {      // the packers in index mode: pointers and shapes as usual, the (possibly stale) values are not what is recorded
        pgpack::NetTensors t = tensors_of(ns, h->cfg);
        t.layout(h->cfg.framecode_ch);
        return t;
    };
    if (!ns.d_src) PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_src), (size_t)lay.off[pgpack::NetTensors::N_SRC] * sizeof(float)));
    pg_launch_collect(d_tensors, lay.off, ns.d_src, stream);        // (off[24] = the end of tensor 23: the folded view layer follows)
    const int vcols = W + CH_D + h->cfg.framecode_ch;
    pg_launch_fold(ns.d_src, lay.off[20], vcols, lay.off[21], lay.off[18], lay.off[19], lay.off[pgpack::NetTensors::SRC_VIEWF_W],
                   lay.off[pgpack::NetTensors::SRC_VIEWF_B], stream);
    auto upload_map = [&](const std::vector<int32_t>& m, int32_t** d, size_t* n) -> int {
        PG_HIP(h, hipMalloc(reinterpret_cast<void**>(d), m.size() * sizeof(int32_t)));
        PG_HIP(h, hipMemcpy(*d, m.data(), m.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        if (n) *n = m.size();
        return PG_OK;
    };
    // images re-formed here (those that exist: the others are built by the first call that needs them)
    bool any_ro = false;
    for (int prec : {PG_PREC_BF16, PG_PREC_FP16}) any_ro = any_ro || ns.d_stream_ro[prec];
    if (any_ro && !ns.d_map_ro) {
        std::vector<uint8_t> img; std::vector<int32_t> m;
        if (pgpack::pack_stream_r(map_tensors(), PG_PREC_BF16, img, true, &m) != 0) return pg_fail(h, PG_EINVAL, "pg_load_weights_device: source map of the on-chip stream");
        if (const int rc = upload_map(m, &ns.d_map_ro, &ns.n_map_ro)) return rc;
    }
    for (int prec : {PG_PREC_BF16, PG_PREC_FP16})
        if (ns.d_stream_ro[prec])
            pg_launch_gather16(ns.d_map_ro, ns.d_src, reinterpret_cast<uint16_t*>(ns.d_stream_ro[prec]), (long long)ns.n_map_ro, prec == PG_PREC_BF16, stream);
    if (ns.d_c2) {
        if (!ns.d_map_c2) {
            std::vector<uint8_t> img; std::vector<int32_t> m;
            if (pgpack::pack_c2(map_tensors(), fc, img, &m) != 0) return pg_fail(h, PG_EINVAL, "pg_load_weights_device: source map of the tile-split image");
            if (const int rc = upload_map(m, &ns.d_map_c2, &ns.n_map_c2)) return rc;
        }
        pg_launch_gather16(ns.d_map_c2, ns.d_src, reinterpret_cast<uint16_t*>(ns.d_c2), (long long)ns.n_map_c2, 0, stream);
    }
    if (ns.d_bias_s) {
        if (!ns.d_map_bias_s) {
            std::vector<float> b; std::vector<int32_t> m;
            pgpack::pack_bias_s(map_tensors(), b, &m);
            if (const int rc = upload_map(m, &ns.d_map_bias_s, nullptr)) return rc;
        }
        pg_launch_gather32(ns.d_map_bias_s, ns.d_src, ns.d_bias_s, (long long)BIAS16_FLOATS, stream);
    }
    if (fc) {
        pg_launch_codes(d_codes, ns.n_codes, ns.d_codes, stream);
        if (ns.d_ycode) pg_launch_ycode(ns.d_src + lay.off[20], vcols, ns.d_codes, ns.n_codes, ns.d_ycode, stream);
    }
    PG_HIP(h, hipGetLastError());
    // everything else: dropped (hipFree waits for the device: only forms the run has used beside the fast paths pay it)
    auto drop = [&](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    for (int p = 0; p < PG_PREC_COUNT; ++p) {
        for (int f = 0; f < 2; ++f) drop(ns.d_stream[p][f]);
        drop(ns.d_vy[p]); drop(ns.d_stream_r[p]);
        if (p != PG_PREC_BF16 && p != PG_PREC_FP16) drop(ns.d_stream_ro[p]);
    }
    drop(ns.d_stream_cr); drop(ns.d_stream_co); drop(ns.d_vyc);
    ns.fold_w.clear(); ns.fold_b.clear();
    ns.host_stale = true;               // (refresh_host also re-forms d_bias, the bias table of the other kernels)
    return ensure_mode_streams(h, which, h->cfg.precision);
}

int pg_set_embedder(pg_handle* h, int which, const float* cutoff_dist, float tau) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "pg_set_embedder: null handle");
    if (which < 0 || which > 1) return pg_fail(h, PG_EINVAL, "pg_set_embedder: which must be 0 (embed_fn) or 1 (embeddirs_fn)");
    if (cutoff_dist) std::memcpy(h->cut + 24 * which, cutoff_dist, 24 * sizeof(float));
    h->tau[which] = tau;
    h->emb_set[which] = true;
    PG_HIP(h, hipSetDevice(h->device));
    PG_HIP(h, hipMemcpy(h->d_cut, h->cut, sizeof h->cut, hipMemcpyHostToDevice));
    PG_FORWARD(h, pg_set_embedder(hh, which, cutoff_dist, tau));
    return PG_OK;
}

int pg_set_framecodes(pg_handle* h, int which, const float* codes, int n_codes) {
    if (!h || !codes) return pg_fail(h, PG_EINVAL, "pg_set_framecodes: null argument");
    if (which < 0 || which > 1) return pg_fail(h, PG_EINVAL, "pg_set_framecodes: which_net must be 0 or 1");
    if (h->cfg.framecode_ch != FC_CH) return pg_fail(h, PG_EINVAL, "pg_set_framecodes: handle was created without frame codes");
    if (n_codes <= 0) return pg_fail(h, PG_EINVAL, "pg_set_framecodes: n_codes must be positive");
    NetState& ns = h->net[which];
    ns.codes_host.assign(codes, codes + (size_t)n_codes * FC_CH);
    ns.codes_host.resize((size_t)(n_codes + 1) * FC_CH, 0.f);
    for (int c = 0; c < FC_CH; ++c) {       // mean row (embedding.py:25-26), summed in row order
        float s = 0.f;
        for (int i = 0; i < n_codes; ++i) s += codes[(size_t)i * FC_CH + c];
        ns.codes_host[(size_t)n_codes * FC_CH + c] = s / (float)n_codes;
    }
    ns.n_codes = n_codes;
    PG_HIP(h, hipSetDevice(h->device));
    if (ns.d_codes) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(ns.d_codes)); ns.d_codes = nullptr; }
    if (ns.d_ycode) { PG_HIP(h, hipFree(ns.d_ycode)); ns.d_ycode = nullptr; }
    PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&ns.d_codes), ns.codes_host.size() * sizeof(float)));
    PG_HIP(h, hipMemcpy(ns.d_codes, ns.codes_host.data(), ns.codes_host.size() * sizeof(float), hipMemcpyHostToDevice));
    PG_FORWARD(h, pg_set_framecodes(hh, which, codes, n_codes));
    return PG_OK;
}

int pg_set_precision(pg_handle* h, int precision) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "pg_set_precision: null handle");
    if (precision < 0 || precision >= PG_PREC_MODES) return pg_fail(h, PG_EINVAL, "pg_set_precision: bad precision %d", precision);
    if (is_x3(precision) && !x3_allowed())
        return pg_fail(h, PG_EINVAL, "pg_set_precision: split-operand precision %d is experimental (set POSEGEN_EXPERIMENTAL_X3=1)", precision);
    h->cfg.precision = precision;
    for (int w = 0; w < 2; ++w)
        if (h->net[w].loaded) { int rc = ensure_mode_streams(h, w, precision); if (rc) return rc; }
    PG_FORWARD(h, pg_set_precision(hh, precision));
    return PG_OK;
}

int pg_set_chunk(pg_handle* h, int chunk) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "pg_set_chunk: null handle");
    if (chunk <= 0) return pg_fail(h, PG_EINVAL, "pg_set_chunk: chunk must be positive, got %d", chunk);
    h->cfg.chunk = chunk;
    PG_FORWARD(h, pg_set_chunk(hh, chunk));
    return PG_OK;
}

int pg_set_far_skip(pg_handle* h, int on) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "pg_set_far_skip: null handle");
    h->far_skip = on != 0;
    PG_FORWARD(h, pg_set_far_skip(hh, on));
    return PG_OK;
}

int pg_set_onchip(pg_handle* h, int mode) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "pg_set_onchip: null handle");
    if (mode < PG_ONCHIP_RECORDS || mode > PG_ONCHIP_ALWAYS) return pg_fail(h, PG_EINVAL, "pg_set_onchip: mode must be 0 (records), 1 (by sample count) or 2 (on chip), got %d", mode);
    h->onchip_mode = mode;
    PG_FORWARD(h, pg_set_onchip(hh, mode));
    return PG_OK;
}

int pg_set_train_precision(pg_handle* h, int precision) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "null handle");
    if (precision != PG_PREC_FP32 && precision != PG_PREC_BF16)
        return pg_fail(h, PG_EINVAL, "pg_set_train_precision: the training step computes in fp32 (PG_PREC_FP32) or on a bf16 tape (PG_PREC_BF16), not %d", precision);
    h->train_precision = precision;
    for (pg_handle* p : h->peers) p->train_precision = precision;
    return PG_OK;
}

int pg_profile_enable(pg_handle* h, int on) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "pg_profile_enable: null handle");
    h->profiling = on != 0;
    return PG_OK;
}

// the record-kernel events recorded so far -> the handle's running totals; the events go back to the pool (called by
// both reads, so that a caller that only ever reads the fused kernel's numbers does not pile events up)
static int fold_aux(pg_handle* h) {
    for (auto& pr : h->ev_aux) {
        PG_HIP(h, hipEventSynchronize(pr.second));
        float t = 0.f;
        PG_HIP(h, hipEventElapsedTime(&t, pr.first, pr.second));
        h->aux_ms += t;
        h->aux_n += 1;
        h->ev_free.push_back(pr.first);
        h->ev_free.push_back(pr.second);
    }
    h->ev_aux.clear();
    return PG_OK;
}

int pg_profile_read_aux(pg_handle* h, int64_t* n_launches, double* total_ms) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "pg_profile_read_aux: null handle");
    PG_HIP(h, hipSetDevice(h->device));
    if (const int rc = fold_aux(h)) return rc;
    if (n_launches) *n_launches = h->aux_n;
    if (total_ms) *total_ms = h->aux_ms;
    h->aux_n = 0;
    h->aux_ms = 0.0;
    return PG_OK;
}

int pg_profile_read(pg_handle* h, int64_t* n_launches, double* total_ms, int64_t* n_points) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "pg_profile_read: null handle");
    PG_HIP(h, hipSetDevice(h->device));
    double ms = 0.0;
    for (auto& pr : h->ev_used) {
        PG_HIP(h, hipEventSynchronize(pr.second));
        float t = 0.f;
        PG_HIP(h, hipEventElapsedTime(&t, pr.first, pr.second));
        ms += t;
        h->ev_free.push_back(pr.first);
        h->ev_free.push_back(pr.second);
    }
    if (n_launches) *n_launches = (int64_t)h->ev_used.size();
    if (total_ms) *total_ms = ms;
    if (n_points) *n_points = h->prof_points;
    h->ev_used.clear();
    h->prof_points = 0;
    return fold_aux(h);             // (kept for the next pg_profile_read_aux)
}

int pg_debug_pack(const float* const* tensors, const int64_t* shapes, int n_tensors, int framecode_ch,
                  int precision, int view_fact, uint8_t* stream_out, int64_t stream_cap, int64_t* stream_bytes, float* bias_out,
                  int32_t* chunk_bytes) {
    if (!tensors || !shapes || n_tensors != 24) return pg_fail(nullptr, PG_EINVAL, "pg_debug_pack: need 24 tensors");
    if (precision < 0 || precision >= PG_PREC_COUNT) return pg_fail(nullptr, PG_EINVAL, "pg_debug_pack: bad precision");
    NetState ns;
    ns.host.assign(24, {});
    for (int i = 0; i < 24; ++i) ns.host[i].assign(tensors[i], tensors[i] + shapes[2 * i] * shapes[2 * i + 1]);
    pg_config cfg{};
    cfg.framecode_ch = framecode_ch;
    std::vector<uint8_t> packed;
    const bool rprog = view_fact != 0 && is_shape_a(precision);       // the 16x16x32 program of pg_eval16r.hip
    const bool crec = view_fact >= 2 && precision == PG_PREC_FP16C;     // record variant of pg_evalc.hip (3: its on-chip form)
    const bool c2 = view_fact == 4 && precision == PG_PREC_FP16C;       // the weight image of pg_evalc2.hip (pg_program.h T)
    const int rc = c2 ? pgpack::pack_c2(tensors_of(ns, cfg), framecode_ch > 0, packed)
                 : rprog ? pgpack::pack_stream_r(tensors_of(ns, cfg), precision, packed, view_fact == 3)
                         : pgpack::pack_stream(tensors_of(ns, cfg), precision, framecode_ch > 0, view_fact != 0, packed, nullptr, crec,
                                               crec && view_fact == 3);
    if (rc != 0) return pg_fail(nullptr, PG_EINVAL, "pg_debug_pack: packing failed (%d)", rc);
    if (stream_bytes) *stream_bytes = (int64_t)packed.size();
    if (chunk_bytes) *chunk_bytes = CHUNK_BYTES;
    if (stream_out) {
        if ((int64_t)packed.size() > stream_cap) return pg_fail(nullptr, PG_EINVAL, "pg_debug_pack: buffer too small");
        std::memcpy(stream_out, packed.data(), packed.size());
    }
    if (bias_out) {
        std::vector<float> bias;
        if (rprog || c2) pgpack::pack_bias_s(tensors_of(ns, cfg), bias);
        else pgpack::pack_bias(tensors_of(ns, cfg), bias);
        std::memcpy(bias_out, bias.data(), bias.size() * sizeof(float));
    }
    return PG_OK;
}

// Host-only: the source map of a packed image (pg_load_weights_device re-forms the image from it by a gather) and the flat
// source vector it indexes.  form 0: on-chip stream of the 16x16x32 kernel, 1: pg_evalc2.hip's image, 2: the 16-row bias table.
int pg_debug_pack_map(const float* const* tensors, const int64_t* shapes, int n_tensors, int framecode_ch, int form,
                      int32_t* map_out, int64_t map_cap, int64_t* map_n, float* src_out, int64_t src_cap, int64_t* src_n) {
    if (!tensors || !shapes || n_tensors != 24) return pg_fail(nullptr, PG_EINVAL, "pg_debug_pack_map: need 24 tensors");
    NetState ns;
    ns.host.assign(24, {});
    for (int i = 0; i < 24; ++i) ns.host[i].assign(tensors[i], tensors[i] + shapes[2 * i] * shapes[2 * i + 1]);
    pg_config cfg{};
    cfg.framecode_ch = framecode_ch;
    pgpack::NetTensors t = tensors_of(ns, cfg);
    t.layout(framecode_ch);
    std::vector<int32_t> m;
    std::vector<uint8_t> img;
    std::vector<float> b;
    int rc = 0;
    if (form == 0) rc = pgpack::pack_stream_r(t, PG_PREC_BF16, img, true, &m);
    else if (form == 1) rc = pgpack::pack_c2(t, framecode_ch > 0, img, &m);
    else if (form == 2) pgpack::pack_bias_s(t, b, &m);
    else return pg_fail(nullptr, PG_EINVAL, "pg_debug_pack_map: form 0, 1 or 2");
    if (rc != 0) return pg_fail(nullptr, PG_EINVAL, "pg_debug_pack_map: packing failed (%d)", rc);
    const int64_t ns_ = t.off[pgpack::NetTensors::N_SRC];
    if (map_n) *map_n = (int64_t)m.size();
    if (src_n) *src_n = ns_;
    if (map_out) {
        if ((int64_t)m.size() > map_cap) return pg_fail(nullptr, PG_EINVAL, "pg_debug_pack_map: map buffer too small");
        std::memcpy(map_out, m.data(), m.size() * sizeof(int32_t));
    }
    if (src_out) {
        if (ns_ > src_cap) return pg_fail(nullptr, PG_EINVAL, "pg_debug_pack_map: source buffer too small");
        for (int i = 0; i < 24; ++i) std::memcpy(src_out + t.off[i], ns.host[i].data(), ns.host[i].size() * sizeof(float));
        std::memcpy(src_out + t.off[pgpack::NetTensors::SRC_VIEWF_W], t.viewf_w.data(), t.viewf_w.size() * sizeof(float));
        std::memcpy(src_out + t.off[pgpack::NetTensors::SRC_VIEWF_B], t.viewf_b.data(), t.viewf_b.size() * sizeof(float));
    }
    return PG_OK;
}

int pg_debug_pack_vy(const float* const* tensors, const int64_t* shapes, int n_tensors, int framecode_ch,
                     int precision, uint8_t* out, int64_t cap, int64_t* out_bytes) {
    if (!tensors || !shapes || n_tensors != 24) return pg_fail(nullptr, PG_EINVAL, "pg_debug_pack_vy: need 24 tensors");
    NetState ns;
    ns.host.assign(24, {});
    for (int i = 0; i < 24; ++i) ns.host[i].assign(tensors[i], tensors[i] + shapes[2 * i] * shapes[2 * i + 1]);
    pg_config cfg{};
    cfg.framecode_ch = framecode_ch;
    std::vector<uint8_t> vy;
    if (precision == PG_PREC_FP16C) {       // the fp32 Y-stage weights of ray_records_c_kernel
        std::vector<float> f;
        pgpack::pack_vyc(tensors_of(ns, cfg), framecode_ch > 0, f);
        vy.resize(f.size() * sizeof(float));
        std::memcpy(vy.data(), f.data(), vy.size());
    } else if (pgpack::pack_vy(tensors_of(ns, cfg), precision, framecode_ch > 0, vy) != 0)
        return pg_fail(nullptr, PG_EINVAL, "pg_debug_pack_vy: 16-bit and compensated-fp16 precisions only");
    if (out_bytes) *out_bytes = (int64_t)vy.size();
    if (out) {
        if ((int64_t)vy.size() > cap) return pg_fail(nullptr, PG_EINVAL, "pg_debug_pack_vy: buffer too small");
        std::memcpy(out, vy.data(), vy.size());
    }
    return PG_OK;
}

int pg_device_info(const pg_handle* h, int32_t* n_cu, int32_t* clock_khz) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "pg_device_info: null handle");
    if (n_cu) *n_cu = h->n_cu;
    if (clock_khz) *clock_khz = h->clock_khz;
    return PG_OK;
}

int pg_calibrate_mfma(pg_handle* h, int f16, int lds_fed, double min_ms, double* tflops, double* ms_out) {
    if (!h || !tflops) return pg_fail(h, PG_EINVAL, "pg_calibrate_mfma: null argument");
    PG_HIP(h, hipSetDevice(h->device));
    int rc = ensure_ws(h, 256);
    if (rc) return rc;
    hipEvent_t e0, e1;
    PG_HIP(h, hipEventCreate(&e0));
    PG_HIP(h, hipEventCreate(&e1));
    hipStream_t s = h->own_stream ? h->own_stream : nullptr;
    PG_HIP(h, hipDeviceSynchronize());                  // own_stream is not ordered behind the caller's stream, and ws is shared
    int blocks = h->n_cu;                               // POSEGEN_MAX_WG: fewer CUs (how the clock answers to the load)
    if (const char* e = std::getenv("POSEGEN_MAX_WG")) { const int c = std::atoi(e); if (c > 0 && c < blocks) blocks = c; }
    int iters = 2000;                                   // ~1 ms per 1000 iterations of 32 MFMAs at 2 waves/SIMD
    int timed_iters = 0;                                // iterations of the launch `ms` belongs to
    float ms = 0.0f;
    int err = 0;
    hipError_t herr = hipSuccess;
    for (int round = 0; round < 6; ++round) {           // grow until one launch lasts min_ms: the clock settles in ms
        (void)hipEventRecord(e0, s);
        err = pg_launch_mfma_rate(f16, lds_fed, blocks, iters, reinterpret_cast<float*>(h->ws), s);
        (void)hipEventRecord(e1, s);
        if (err) break;
        herr = hipEventSynchronize(e1);
        if (herr == hipSuccess) herr = hipEventElapsedTime(&ms, e0, e1);
        if (herr != hipSuccess) break;
        timed_iters = iters;
        if (ms >= min_ms && round > 0) break;
        if (ms < min_ms) iters = (int)(iters * (ms > 0.05 ? 1.25 * min_ms / ms : 8.0)) + 1;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (err) return pg_fail(h, PG_EHIP, "calibration launch failed: %s", hipGetErrorString((hipError_t)err));
    if (herr != hipSuccess) return pg_fail(h, PG_EHIP, "calibration kernel failed: %s", hipGetErrorString(herr));
    if (!(ms > 0.0f) || timed_iters <= 0) return pg_fail(h, PG_EHIP, "calibration measured no time (%.3f ms over %d iterations)", ms, timed_iters);
    const double flop = (double)blocks * 8.0 * (double)timed_iters * 32.0 * 32768.0;
    *tflops = flop / (ms * 1e-3) / 1e12;
    if (ms_out) *ms_out = ms;
    return PG_OK;
}

int pg_query(const pg_handle* h, int precision, int64_t* stream_bytes, int64_t* mfma_per_group) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "pg_query: null handle");
    if (precision < 0 || precision >= PG_PREC_MODES) return PG_EINVAL;
    if (precision == PG_PREC_FP16M) precision = PG_PREC_FP16C;      // the pass that produces the returned maps
    const bool fc = h->cfg.framecode_ch > 0;
    const bool sa = is_shape_a(precision);
    // 16-bit kernels: the factorised-view program (rays with >= 64 samples, the usual case)
    const bool fact = use_fact(precision, FACT_MIN_S);
    const bool compk = use_comp_kernel(precision, FACT_MIN_S, false);
    if (sa && fact) {                        // 16x16x32 kernel: reported in 32x32x16 equivalents (32 768 FLOP each)
        if (stream_bytes) *stream_bytes = (int64_t)(use_onchip(h, fc, 0, FACT_MIN_S) ? pgp::R::NCHUNK_OC : pgp::R::NCHUNK) * CHUNK_BYTES;
        if (mfma_per_group) *mfma_per_group = pgp::R::MFMA16_PER_GROUP / 2;
        return PG_OK;
    }
    if (compk && use_evalc2(FACT_MIN_S)) {       // out tiles over the waves (pg_evalc2.hip): no stream; 16x16x32 MFMAs in 32x32x16 equivalents
        if (stream_bytes) *stream_bytes = (int64_t)pgp::T::TOTAL;
        if (mfma_per_group) *mfma_per_group = pgp::T::MFMA16_PER_PASS / 2 / (pgp::T::PTS / 32);
        return PG_OK;
    }
    if (compk && use_comp_rec(FACT_MIN_S)) {     // record variant of the compensated kernel (the usual case)
        if (stream_bytes) *stream_bytes = (int64_t)(use_onchip(h, fc, 0, FACT_MIN_S, true) ? pgp::C::NCHUNK_OC : pgp::C::NCHUNK_R) * CHUNK_BYTES;
        if (mfma_per_group) *mfma_per_group = pgp::C::MFMA_PER_GROUP_R;
        return PG_OK;
    }
    if (stream_bytes)
        *stream_bytes = (int64_t)(compk ? pgp::C::NCHUNK : sa ? pgp::A::NCHUNK
                                     : (precision == PG_PREC_FP32 ? pgp::B::NCHUNK : pgp::B::NCHUNK_FOLD)) * CHUNK_BYTES;
    if (mfma_per_group) {
        // the fp32 / split kernels keep feature_linear and the direct view layer (13 + 4 out tiles)
        const int64_t direct = pgp::A::MFMA_PER_GROUP(fc) + (NT + 1 + NTV - (NTV + 1)) * pgp::A::HU;
        *mfma_per_group = compk ? pgp::C::MFMA_PER_GROUP(fc) : sa ? pgp::A::MFMA_PER_GROUP(fc)
                             : (precision == PG_PREC_FP32 ? direct * 8 : (direct - NT * pgp::A::HU) * (precision == PG_PREC_FP16C ? 2 : 3));
    }
    return PG_OK;
}

int pg_stage_sample_coarse(pg_handle* h, void* stream, int64_t n, const float* ray_batch, const float* cyls,
                           int64_t cyl_stride, int n_samples, int flags, float* near_far, float* z) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "null handle");
    if (n < 0 || !ray_batch || !cyls || !near_far || !z) return pg_fail(h, PG_EINVAL, "pg_stage_sample_coarse: null/negative argument");
    if (n_samples < 2) return pg_fail(h, PG_EINVAL, "pg_stage_sample_coarse: N_samples must be >= 2");
    if (cyl_stride != 0 && cyl_stride != 5) return pg_fail(h, PG_EINVAL, "cyl_stride must be 0 or 5");
    PG_HIP(h, hipSetDevice(h->device));
    double* scs = nullptr;
    if (int rc = pg_sc_scratch(h, n, h->cfg.chunk, &scs)) return rc;
    int e = pg_launch_sample_coarse(ray_batch, cyls, cyl_stride, n, h->cfg.chunk, n_samples,
                                    (flags & PG_FLAG_LINDISP) ? 1 : 0, near_far, z, nullptr, scs, stream);
    if (e) return pg_fail(h, PG_EHIP, "sample_coarse launch failed: %s", hipGetErrorString((hipError_t)e));
    return PG_OK;
}

int pg_stage_eval(pg_handle* h, void* stream, int which, int64_t n, int n_samples, const float* ray_batch,
                  const float* z, const float* skts, int64_t pose_stride, const float* cams, float* raw, float* dbg,
                  int dbg_stage) {
    int rc = check_ready(h, which == 1);
    if (rc) return rc;
    if (which < 0 || which > 1) return pg_fail(h, PG_EINVAL, "pg_stage_eval: which_net must be 0 or 1");
    if (n < 0 || !ray_batch || !z || !skts || !raw) return pg_fail(h, PG_EINVAL, "pg_stage_eval: null/negative argument");
    if (pose_stride != 0 && pose_stride != 384) return pg_fail(h, PG_EINVAL, "pose_stride must be 0 or 384");
    if (n == 0) return PG_OK;
    PG_HIP(h, hipSetDevice(h->device));
    return launch_eval(h, stream, which, n, n_samples, ray_batch, z, skts, pose_stride, cams, raw, dbg, dbg_stage);
}

int pg_query_density(pg_handle* h, void* stream, int which, int64_t n_points, const float* pts, const float* skts,
                     float* raw) {
    int rc = check_ready(h, which == 1);
    if (rc) return rc;
    if (which < 0 || which > 1) return pg_fail(h, PG_EINVAL, "pg_query_density: which_net must be 0 or 1");
    if (n_points < 0 || !pts || !skts || !raw) return pg_fail(h, PG_EINVAL, "pg_query_density: null/negative argument");
    if (n_points > 0x7fffffffLL) return pg_fail(h, PG_EINVAL, "pg_query_density: at most 2^31-1 points per call");
    if (n_points == 0) return PG_OK;
    PG_HIP(h, hipSetDevice(h->device));
    // one pseudo ray (o = d = 0) that owns all points: the kernels take the pose and the view table
    // from the ray slot, the position from `pts`
    rc = ensure_ws(h, 256);
    if (rc) return rc;
    PG_HIP(h, hipMemsetAsync(h->ws, 0, 64, static_cast<hipStream_t>(stream)));
    return launch_eval(h, stream, which, 1, (int)n_points, reinterpret_cast<const float*>(h->ws), nullptr, skts, 0,
                       nullptr, raw, nullptr, 0, pts);
}

int pg_stage_composite(pg_handle* h, void* stream, int64_t n, int n_samples, const float* ray_batch, const float* z,
                       const float* raw, float* rgb, float* disp, float* acc, float* alpha, float* weights,
                       int n_importance, float* z_fine) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "null handle");
    if (n < 0 || !ray_batch || !z || !raw) return pg_fail(h, PG_EINVAL, "pg_stage_composite: null/negative argument");
    if (n_samples < 2 || n_samples > pg_composite_max_samples()) return pg_fail(h, PG_EINVAL, "pg_stage_composite: N_samples %d outside [2,%d]", n_samples, pg_composite_max_samples());
    if (n_importance < 0 || n_importance > pg_composite_max_importance() || n_importance == 1)
        return pg_fail(h, PG_EINVAL, "pg_stage_composite: N_importance %d outside {0, 2..%d}", n_importance, pg_composite_max_importance());
    if (n_importance > 0 && n_samples < 3) return pg_fail(h, PG_EINVAL, "importance sampling needs N_samples >= 3");
    PG_HIP(h, hipSetDevice(h->device));
    int e = pg_launch_composite(ray_batch, z, raw, n, n_samples, h->cfg.density_scale, h->cfg.rgb_eps, h->cfg.density_act, h->cfg.softplus_shift, rgb, disp, acc,
                                alpha, weights, n_importance, z_fine, nullptr, nullptr, nullptr, stream);
    if (e) return pg_fail(h, PG_EHIP, "composite launch failed: %s", hipGetErrorString((hipError_t)e));
    return PG_OK;
}

namespace {
int render_rays_impl(pg_handle* h, void* stream, int64_t n, const float* ray_batch, const float* skts,
                     int64_t pose_stride, const float* cyls, int64_t cyl_stride, const float* cams, int n_samples,
                     int n_importance, int flags, const pg_train_draws* dr, const pg_outputs* out);
}

int pg_render_rays(pg_handle* h, void* stream, int64_t n, const float* ray_batch, const float* skts,
                   int64_t pose_stride, const float* cyls, int64_t cyl_stride, const float* cams, int n_samples,
                   int n_importance, int flags, const pg_outputs* out) {
    return render_rays_impl(h, stream, n, ray_batch, skts, pose_stride, cyls, cyl_stride, cams, n_samples, n_importance,
                            flags, nullptr, out);
}

int pg_render_rays_train(pg_handle* h, void* stream, int64_t n, const float* ray_batch, const float* skts,
                         int64_t pose_stride, const float* cyls, int64_t cyl_stride, const float* cams, int n_samples,
                         int n_importance, int flags, const pg_train_draws* draws, const pg_outputs* out) {
    if (!draws) return pg_fail(h, PG_EINVAL, "pg_render_rays_train: null draws (use pg_render_rays for eval mode)");
    return render_rays_impl(h, stream, n, ray_batch, skts, pose_stride, cyls, cyl_stride, cams, n_samples, n_importance,
                            flags, draws, out);
}

namespace {
int render_rays_impl(pg_handle* h, void* stream, int64_t n, const float* ray_batch, const float* skts,
                     int64_t pose_stride, const float* cyls, int64_t cyl_stride, const float* cams, int n_samples,
                     int n_importance, int flags, const pg_train_draws* dr, const pg_outputs* out) {
    int rc = check_ready(h, n_importance > 0);
    if (rc) return rc;
    if (n < 0 || !ray_batch || !skts || !cyls || !out) return pg_fail(h, PG_EINVAL, "pg_render_rays: null/negative argument");
    if (pose_stride != 0 && pose_stride != 384) return pg_fail(h, PG_EINVAL, "pose_stride must be 0 (shared) or 384 (per ray)");
    if (cyl_stride != 0 && cyl_stride != 5) return pg_fail(h, PG_EINVAL, "cyl_stride must be 0 (shared) or 5 (per ray)");
    if (n_samples < 2 || n_samples > pg_composite_max_samples()) return pg_fail(h, PG_EINVAL, "N_samples %d outside [2,%d]", n_samples, pg_composite_max_samples());
    if (n_importance < 0 || n_importance == 1 || n_importance > pg_composite_max_importance())
        return pg_fail(h, PG_EINVAL, "N_importance %d outside {0, 2..%d}", n_importance, pg_composite_max_importance());
    if (n_importance > 0 && n_samples + n_importance > pg_composite_max_samples())
        return pg_fail(h, PG_EINVAL, "N_samples + N_importance exceeds %d", pg_composite_max_samples());
    if (n == 0) return PG_OK;
    PG_HIP(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int S = n_samples, SF = n_samples + n_importance;
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t b_nf = al((size_t)n * 2 * 4), b_zc = al((size_t)n * S * 4), b_rc = al((size_t)n * S * 16),
                 b_w0 = al((size_t)n * S * 4), b_zf = al((size_t)n * SF * 4), b_rf = al((size_t)n * SF * 16);
    const bool hier = n_importance > 0;
    const bool rnoise = dr && dr->ray_noise;
    const size_t b_ord = rnoise && hier ? al((size_t)n * SF * 4) : 0, b_pn = rnoise ? al((size_t)n * SF * 12) : 0;
    rc = ensure_ws(h, b_nf + b_zc + b_rc + b_w0 + (hier ? b_zf + b_rf : 0) + b_ord + b_pn);
    if (rc) return rc;
    uint8_t* p = h->ws;
    float* nf = reinterpret_cast<float*>(p); p += b_nf;
    float* zc = reinterpret_cast<float*>(p); p += b_zc;
    float* rawc = reinterpret_cast<float*>(p); p += b_rc;
    float* w0 = reinterpret_cast<float*>(p); p += b_w0;
    float* zf = reinterpret_cast<float*>(p); p += hier ? b_zf : 0;
    float* rawf = reinterpret_cast<float*>(p); p += hier ? b_rf : 0;
    int* order = b_ord ? reinterpret_cast<int*>(p) : nullptr; p += b_ord;
    float* pn = b_pn ? reinterpret_cast<float*>(p) : nullptr;

    // near/far + coarse depths; with perturb the stratified jitter from the caller's draws (ray_utils.py:229-246)
    {
        double* scs = nullptr;
        if (int rc = pg_sc_scratch(h, n, h->cfg.chunk, &scs)) return rc;
        int e0 = pg_launch_sample_coarse(ray_batch, cyls, cyl_stride, n, h->cfg.chunk, S, (flags & PG_FLAG_LINDISP) ? 1 : 0, nf, zc,
                                         dr ? dr->t_rand : nullptr, scs, stream);
        if (e0) return pg_fail(h, PG_EHIP, "coarse sampling launch failed: %s", hipGetErrorString((hipError_t)e0));
    }
    if (rnoise) {       // position noise of the coarse points: rows [:S] of every ray's draws
        int e0 = pg_launch_gather_noise(dr->ray_noise, n, SF, S, nullptr, pn, stream);
        if (e0) return pg_fail(h, PG_EHIP, "noise gather launch failed: %s", hipGetErrorString((hipError_t)e0));
    }
    rc = launch_eval(h, stream, 0, n, S, ray_batch, zc, skts, pose_stride, cams, rawc, nullptr, 0, nullptr, rnoise ? pn : nullptr, hier);
    if (rc) return rc;
    int e = pg_launch_composite(ray_batch, zc, rawc, n, S, h->cfg.density_scale, h->cfg.rgb_eps, h->cfg.density_act, h->cfg.softplus_shift,
                                hier ? out->rgb0 : out->rgb_map, hier ? out->disp0 : out->disp_map,
                                hier ? out->acc0 : out->acc_map, hier ? out->alpha0 : out->alpha,
                                out->weights0 ? out->weights0 : w0, n_importance, hier ? zf : nullptr,
                                dr ? dr->noise0 : nullptr, dr ? dr->u_rand : nullptr, order, stream);
    if (e) return pg_fail(h, PG_EHIP, "composite launch failed: %s", hipGetErrorString((hipError_t)e));
    if (hier) {
        if (rnoise) {   // every fine point keeps the noise of the stage it came from, in sorted order (raycasters.py:666-686)
            e = pg_launch_gather_noise(dr->ray_noise, n, SF, SF, order, pn, stream);
            if (e) return pg_fail(h, PG_EHIP, "noise gather launch failed: %s", hipGetErrorString((hipError_t)e));
        }
        rc = launch_eval(h, stream, 1, n, SF, ray_batch, zf, skts, pose_stride, cams, rawf, nullptr, 0, nullptr, rnoise ? pn : nullptr);
        if (rc) return rc;
        e = pg_launch_composite(ray_batch, zf, rawf, n, SF, h->cfg.density_scale, h->cfg.rgb_eps, h->cfg.density_act, h->cfg.softplus_shift, out->rgb_map,
                                out->disp_map, out->acc_map, out->alpha, nullptr, 0, nullptr, dr ? dr->noise1 : nullptr,
                                nullptr, nullptr, stream);
        if (e) return pg_fail(h, PG_EHIP, "composite launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    // optional intermediates
    if (out->near_far) PG_HIP(h, hipMemcpyAsync(out->near_far, nf, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
    if (out->z_coarse) PG_HIP(h, hipMemcpyAsync(out->z_coarse, zc, (size_t)n * S * 4, hipMemcpyDeviceToDevice, s));
    if (out->raw_coarse) PG_HIP(h, hipMemcpyAsync(out->raw_coarse, rawc, (size_t)n * S * 16, hipMemcpyDeviceToDevice, s));
    if (hier && out->z_fine) PG_HIP(h, hipMemcpyAsync(out->z_fine, zf, (size_t)n * SF * 4, hipMemcpyDeviceToDevice, s));
    if (hier && out->raw_fine) PG_HIP(h, hipMemcpyAsync(out->raw_fine, rawf, (size_t)n * SF * 16, hipMemcpyDeviceToDevice, s));
    return PG_OK;
}
}  // namespace

int pg_pose_kinematics(pg_handle* h, void* stream, int64_t n_poses, const double* bones, const double* bone_offsets,
                       const int32_t* parents, float* kps, float* skts, double* l2ws) {
    const double* rest_pose = bone_offsets;
    if (!h) return pg_fail(nullptr, PG_EINVAL, "null handle");
    if (n_poses < 0 || !bones || !rest_pose || !parents) return pg_fail(h, PG_EINVAL, "pg_pose_kinematics: null/negative argument");
    if (h->cfg.n_joints != 24) return pg_fail(h, PG_EINVAL, "pg_pose_kinematics: 24-joint SMPL skeleton only");
    for (int j = 0; j < 24; ++j)
        if (parents[j] < 0 || parents[j] > j || (j > 0 && parents[j] == j))
            return pg_fail(h, PG_EINVAL, "pg_pose_kinematics: joint %d must come after its parent (%d)", j, parents[j]);
    PG_HIP(h, hipSetDevice(h->device));
    int e = pg_launch_pose_kinematics(rest_pose, parents, bones, n_poses, kps, skts, l2ws, stream);
    if (e) return pg_fail(h, PG_EHIP, "pose kinematics kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return PG_OK;
}

// ---- frame front / back end: helpers shared by pg_render_frame (one device) and pg_render_frames ----
namespace {

struct FrameMaps { float *rgb_map, *disp_map, *acc_map; };      // [n_box,3], [n_box], [n_box]: the whole box

int frame_geom(pg_handle* h, int H, int W, const float* c2w, const float* intrinsics, const int* box, float near,
               float far, float cam, pgk::FrameGeom* g) {
    if (H <= 0 || W <= 0 || !c2w || !intrinsics || !box) return pg_fail(h, PG_EINVAL, "frame: null/non-positive argument");
    g->H = H; g->W = W;
    g->tlx = box[0] < 0 ? 0 : box[0]; g->tly = box[1] < 0 ? 0 : box[1];
    const int brx = box[2] > W ? W : box[2], bry = box[3] > H ? H : box[3];
    g->bw = brx > g->tlx ? brx - g->tlx : 0; g->bh = bry > g->tly ? bry - g->tly : 0;
    g->fx = intrinsics[0]; g->fy = intrinsics[1]; g->cx = intrinsics[2]; g->cy = intrinsics[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) g->R[3 * r + c] = c2w[4 * r + c];
        g->t[r] = c2w[4 * r + 3];
    }
    g->near = near; g->far = far; g->cam = cam;
    return PG_OK;
}

// Frame workspace of `h`: ray_batch rows and cams of a range of n_range rays, the maps of a WHOLE box of
// n rays (n = 0: none -- the range's maps live in a buffer of the caller), coarse-pass scratch of the range.
int frame_ws(pg_handle* h, int64_t n, int64_t n_range, float** rays, float** cams, FrameMaps* maps, pg_outputs* scratch) {
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t b_rays = al((size_t)n_range * 44), b_cam = al((size_t)n_range * 4);
    const size_t b_rgb = al((size_t)n * 12), b_1 = al((size_t)n * 4), r_rgb = al((size_t)n_range * 12), r_1 = al((size_t)n_range * 4);
    const size_t need = b_rays + b_cam + b_rgb + 2 * b_1 + r_rgb + 2 * r_1;
    if (need > h->fws_bytes) {
        if (h->fws) { PG_HIP(h, hipDeviceSynchronize()); PG_HIP(h, hipFree(h->fws)); h->fws = nullptr; h->fws_bytes = 0; }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->fws), need + need / 8);
        if (e != hipSuccess) return pg_fail(h, PG_ENOMEM, "frame workspace allocation of %zu bytes failed", need);
        h->fws_bytes = need + need / 8;
    }
    uint8_t* p = h->fws;
    *rays = reinterpret_cast<float*>(p); p += b_rays;
    *cams = reinterpret_cast<float*>(p); p += b_cam;
    maps->rgb_map = reinterpret_cast<float*>(p); p += b_rgb;
    maps->disp_map = reinterpret_cast<float*>(p); p += b_1;
    maps->acc_map = reinterpret_cast<float*>(p); p += b_1;
    scratch->rgb0 = reinterpret_cast<float*>(p); p += r_rgb;
    scratch->disp0 = reinterpret_cast<float*>(p); p += r_1;
    scratch->acc0 = reinterpret_cast<float*>(p);
    return PG_OK;
}

// rays [r0, r1) of the box (row-major ray list of kp_to_valid_rays).  r0 must be a multiple of the nanmean
// group size (`chunk`) unless it is 0, so that the groups are those of the whole frame.  ext == nullptr:
// the maps of the whole box are carved from the frame workspace and the range is written at its offset
// (returned in *maps); otherwise the range's maps go to ext (rgb [r1-r0,3], disp, acc [r1-r0]).
int frame_render_range(pg_handle* h, void* stream, const pgk::FrameGeom& g, int64_t r0, int64_t r1, const float* skts,
                       const float* cyl, int n_samples, int n_importance, int flags, FrameMaps* maps, const FrameMaps* ext = nullptr) {
    const int64_t n = (int64_t)g.bw * g.bh;
    if (r0 < 0 || r1 > n || r0 > r1) return pg_fail(h, PG_EINVAL, "frame range [%lld, %lld) outside the box of %lld rays", (long long)r0, (long long)r1, (long long)n);
    if (r0 % h->cfg.chunk != 0) return pg_fail(h, PG_EINVAL, "frame range must start on a nanmean group boundary (chunk %d)", h->cfg.chunk);
    PG_HIP(h, hipSetDevice(h->device));
    float *rays, *cams;
    pg_outputs out{};
    FrameMaps own{};
    int rc = frame_ws(h, ext ? 0 : n, r1 - r0, &rays, &cams, &own, &out);
    if (rc) return rc;
    if (maps) *maps = ext ? *ext : own;
    if (r1 == r0) return PG_OK;
    if (ext) { out.rgb_map = ext->rgb_map; out.disp_map = ext->disp_map; out.acc_map = ext->acc_map; }
    else { out.rgb_map = own.rgb_map + r0 * 3; out.disp_map = own.disp_map + r0; out.acc_map = own.acc_map + r0; }
    const bool fc = h->cfg.framecode_ch > 0;
    int e = pg_launch_frame_rays(&g, r0, r1 - r0, rays, fc ? cams : nullptr, stream);
    if (e) return pg_fail(h, PG_EHIP, "frame ray kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return pg_render_rays(h, stream, r1 - r0, rays, skts, 0, cyl, 0, fc ? cams : nullptr, n_samples, n_importance, flags, &out);
}

int frame_compose(pg_handle* h, void* stream, const pgk::FrameGeom& g, const FrameMaps& maps, const float* bg, float base_bg,
                  float* rgb, float* disp, float* acc, uint8_t* rgb8) {
    int e = pg_launch_frame_compose(&g, maps.rgb_map, maps.disp_map, maps.acc_map, bg, base_bg, rgb, disp, acc, rgb8, stream);
    if (e) return pg_fail(h, PG_EHIP, "frame compose kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return PG_OK;
}

}  // namespace

int pg_render_frame(pg_handle* h, void* stream, int H, int W, const float* c2w, const float* intrinsics,
                    const int* box, float near, float far, const float* skts, const float* cyl, float cam,
                    int n_samples, int n_importance, int flags, const float* bg, float base_bg,
                    float* rgb, float* disp, float* acc, uint8_t* rgb8) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "null handle");
    if (!skts || !cyl || !rgb) return pg_fail(h, PG_EINVAL, "pg_render_frame: null argument");
    pgk::FrameGeom g{};
    int rc = frame_geom(h, H, W, c2w, intrinsics, box, near, far, cam, &g);
    if (rc) return rc;
    FrameMaps maps{};
    rc = frame_render_range(h, stream, g, 0, (int64_t)g.bw * g.bh, skts, cyl, n_samples, n_importance, flags, &maps);
    if (rc) return rc;
    return frame_compose(h, stream, g, maps, bg, base_bg, rgb, disp, acc, rgb8);
}

int pg_render_frame_range(pg_handle* h, void* stream, int H, int W, const float* c2w, const float* intrinsics,
                          const int* box, float near, float far, const float* skts, const float* cyl, float cam,
                          int n_samples, int n_importance, int flags, int64_t ray_begin, int64_t ray_end,
                          float* rgb_map, float* disp_map, float* acc_map) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "null handle");
    if (!skts || !cyl || !rgb_map || !disp_map || !acc_map) return pg_fail(h, PG_EINVAL, "pg_render_frame_range: null argument");
    pgk::FrameGeom g{};
    int rc = frame_geom(h, H, W, c2w, intrinsics, box, near, far, cam, &g);
    if (rc) return rc;
    const FrameMaps ext{rgb_map, disp_map, acc_map};
    return frame_render_range(h, stream, g, ray_begin, ray_end, skts, cyl, n_samples, n_importance, flags, nullptr, &ext);
}

int pg_compose_frame(pg_handle* h, void* stream, int H, int W, const int* box, const float* rgb_map, const float* disp_map,
                     const float* acc_map, const float* bg, float base_bg, float* rgb, float* disp, float* acc, uint8_t* rgb8) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "null handle");
    if (H <= 0 || W <= 0 || !box || !rgb) return pg_fail(h, PG_EINVAL, "pg_compose_frame: null/non-positive argument");
    pgk::FrameGeom g{};
    const float c2w[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, intr[4] = {1.f, 1.f, 0.f, 0.f};
    int rc = frame_geom(h, H, W, c2w, intr, box, 0.f, 1.f, -1.f, &g);
    if (rc) return rc;
    if ((int64_t)g.bw * g.bh > 0 && (!rgb_map || !disp_map || !acc_map)) return pg_fail(h, PG_EINVAL, "pg_compose_frame: null map of a non-empty box");
    PG_HIP(h, hipSetDevice(h->device));
    const FrameMaps maps{const_cast<float*>(rgb_map), const_cast<float*>(disp_map), const_cast<float*>(acc_map)};
    return frame_compose(h, stream, g, maps, bg, base_bg, rgb, disp, acc, rgb8);
}

int pg_pose_boxes(pg_handle* h, void* stream, int64_t n_poses, const float* kps, const double* w2c, int64_t w2c_stride,
                  const double* ring, double extension, double top_extension, double bot_extension, double fx, double fy,
                  int H, int W, int off_x, int off_y, float* cyls, int32_t* boxes) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "null handle");
    if (n_poses < 0 || !kps || !w2c || !ring || !cyls || !boxes || H <= 0 || W <= 0)
        return pg_fail(h, PG_EINVAL, "pg_pose_boxes: null/negative argument");
    if (w2c_stride != 0 && w2c_stride != 16) return pg_fail(h, PG_EINVAL, "pg_pose_boxes: w2c_stride must be 0 (one camera) or 16");
    PG_HIP(h, hipSetDevice(h->device));
    int e = pg_launch_pose_boxes(kps, n_poses, w2c, w2c_stride, ring, (float)extension, (float)top_extension, (float)bot_extension,
                                 fx, fy, H, W, off_x, off_y, cyls, boxes, stream);
    if (e) return pg_fail(h, PG_EHIP, "pose box kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return PG_OK;
}

int pg_device_count(const pg_handle* h) { return h ? 1 + (int)h->peers.size() : 0; }

// ---- in-process multi-device frame rendering -------------------------------------------------------
namespace {

struct FrameTask { int frame; int64_t r0, r1; int worker; int owner; };

// Work plan of a frame batch on G workers (the call pattern of run_gan.py:2042-2047 is
// 20 frames per call: whole frames alone would leave 3:2 loads on 8 GPUs).  The unit of work is a nanmean
// group (`chunk` consecutive rays of a frame's box).  Frames go to workers whole, largest first, to the
// least loaded worker, as long as they fit under the per-worker target (total rays / G, 2 % slack); the
// frames that do not fit (the tail of a batch with F mod G != 0, or every frame when F < G) are cut into
// contiguous runs of whole groups that fill the least loaded workers up to the target.  Every cut falls
// on a multiple of `chunk`, so every group is rendered exactly as on one device.  A cut frame is composed
// by its owner (the worker of its first run).  Deterministic; dist.plan_tasks is the same algorithm.
void plan_frames(const std::vector<int64_t>& n_rays, int G, int chunk, std::vector<FrameTask>* tasks) {
    const int F = (int)n_rays.size();
    tasks->clear();
    if (F == 0) return;
    std::vector<int> order(F);
    for (int f = 0; f < F; ++f) order[f] = f;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return n_rays[a] > n_rays[b]; });
    int64_t total = 0;
    for (int64_t n : n_rays) total += n > 0 ? n : 0;
    const int64_t target = (total + G - 1) / G;
    std::vector<int64_t> load(G, 0);
    auto least = [&] {
        int w = 0;
        for (int k = 1; k < G; ++k) if (load[k] < load[w]) w = k;
        return w;
    };
    std::vector<int> tail;
    for (int f : order) {
        const int w = least();
        if (n_rays[f] <= chunk || load[w] + n_rays[f] <= target + target / 50) {
            load[w] += n_rays[f] > 0 ? n_rays[f] : 0;
            tasks->push_back({f, 0, n_rays[f] > 0 ? n_rays[f] : 0, w, w});
        } else {
            tail.push_back(f);
        }
    }
    for (int f : tail) {
        const int64_t groups = (n_rays[f] + chunk - 1) / chunk;
        int64_t g = 0;
        int owner = -1;
        while (g < groups) {
            const int w = least();
            const int64_t cap = target - load[w];
            int64_t take = cap > 0 ? (cap + chunk / 2) / chunk : 0;     // nearest whole number of groups
            if (take < 1) take = 1;
            if (take > groups - g) take = groups - g;
            const int64_t rest = groups - g - take;
            if (rest > 0 && rest * chunk <= std::max<int64_t>(chunk, target / 32)) take += rest;   // no sliver of a run for yet another worker
            const int64_t r0 = g * chunk, r1 = std::min((g + take) * chunk, n_rays[f]);
            if (owner < 0) owner = w;
            if (!tasks->empty() && tasks->back().frame == f && tasks->back().worker == w && tasks->back().r1 == r0) tasks->back().r1 = r1;
            else tasks->push_back({f, r0, r1, w, owner});
            load[w] += r1 - r0;
            g += take;
        }
    }
}

}  // namespace

int pg_plan_frames(int n_frames, const int64_t* n_rays, int n_workers, int chunk, int32_t* out_tasks /*[cap,5]*/, int cap, int* n_tasks) {
    if (n_frames < 0 || !n_rays || n_workers < 1 || chunk < 1 || !n_tasks) return pg_fail(nullptr, PG_EINVAL, "pg_plan_frames: bad argument");
    std::vector<FrameTask> t;
    plan_frames(std::vector<int64_t>(n_rays, n_rays + n_frames), n_workers, chunk, &t);
    *n_tasks = (int)t.size();
    if (out_tasks) {
        if ((int)t.size() > cap) return pg_fail(nullptr, PG_EINVAL, "pg_plan_frames: %zu tasks exceed the capacity %d", t.size(), cap);
        for (size_t i = 0; i < t.size(); ++i) {
            out_tasks[5 * i] = t[i].frame; out_tasks[5 * i + 1] = (int32_t)t[i].r0; out_tasks[5 * i + 2] = (int32_t)t[i].r1;
            out_tasks[5 * i + 3] = t[i].worker; out_tasks[5 * i + 4] = t[i].owner;
        }
    }
    return PG_OK;
}

// ---- pg_render_frames: per-device resources kept on the handle between calls -----------------------------------
}  // extern "C"

namespace {

constexpr int NBUF = pg_handle::FramesCache::NBUF;

void frames_cache_release(pg_handle* h) {
    auto& c = h->fc;
    (void)hipSetDevice(h->device);
    for (int b = 0; b < NBUF; ++b) {
        if (c.d_frame[b]) (void)hipFree(c.d_frame[b]);
        if (c.copied[b]) (void)hipEventDestroy(c.copied[b]);
        if (c.composed[b]) (void)hipEventDestroy(c.composed[b]);
        c.d_frame[b] = nullptr; c.copied[b] = nullptr; c.composed[b] = nullptr;
    }
    if (c.copy_stream) (void)hipStreamDestroy(c.copy_stream);
    if (c.d_bg) (void)hipFree(c.d_bg);
    if (c.d_poses) (void)hipFree(c.d_poses);
    if (c.d_part) (void)hipFree(c.d_part);
    if (c.h_stage) (void)hipHostFree(c.h_stage);
    c = pg_handle::FramesCache();
}

// bytes of one frame buffer: rgb [hw,3] | disp [hw] | acc [hw] floats, then the uint8 frame
size_t frame_bytes(size_t hw) { return hw * 20 + ((hw * 3 + 255) & ~size_t(255)); }

// grow-only: nothing is allocated or freed by a call whose sizes an earlier call has seen
int frames_cache_ensure(pg_handle* h, size_t hw, int n_frames, size_t part_rays, const float* bg_host, bool staged) {
    auto& c = h->fc;
    PG_HIP(h, hipSetDevice(h->device));
    if (!c.copy_stream) PG_HIP(h, hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < NBUF; ++b) {
        if (!c.copied[b]) PG_HIP(h, hipEventCreateWithFlags(&c.copied[b], hipEventDisableTiming));
        if (!c.composed[b]) PG_HIP(h, hipEventCreateWithFlags(&c.composed[b], hipEventDisableTiming));
    }
    if (hw > c.hw) {
        PG_HIP(h, hipDeviceSynchronize());
        for (int b = 0; b < NBUF; ++b) {
            if (c.d_frame[b]) { PG_HIP(h, hipFree(c.d_frame[b])); c.d_frame[b] = nullptr; }
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&c.d_frame[b]), frame_bytes(hw));
            if (e != hipSuccess) { c.hw = 0; return pg_fail(h, PG_ENOMEM, "frame buffer of %zu bytes failed: %s", frame_bytes(hw), hipGetErrorString(e)); }
        }
        c.hw = hw;
    }
    if (bg_host) {
        if (hw > c.bg_hw) {
            PG_HIP(h, hipDeviceSynchronize());
            if (c.d_bg) { PG_HIP(h, hipFree(c.d_bg)); c.d_bg = nullptr; c.bg_hw = 0; }
            PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&c.d_bg), hw * 12));
            c.bg_hw = hw;
        }
        PG_HIP(h, hipMemcpy(c.d_bg, bg_host, hw * 12, hipMemcpyHostToDevice));
    }
    if ((size_t)n_frames > c.poses_cap) {
        PG_HIP(h, hipDeviceSynchronize());
        if (c.d_poses) { PG_HIP(h, hipFree(c.d_poses)); c.d_poses = nullptr; c.poses_cap = 0; }
        const size_t cap = (size_t)n_frames + (size_t)n_frames / 2 + 8;
        PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&c.d_poses), cap * (384 + 8) * sizeof(float)));
        c.poses_cap = cap;
    }
    if (part_rays > c.part_cap) {
        PG_HIP(h, hipDeviceSynchronize());
        if (c.d_part) { PG_HIP(h, hipFree(c.d_part)); c.d_part = nullptr; c.part_cap = 0; }
        const size_t cap = part_rays + part_rays / 4;
        PG_HIP(h, hipMalloc(reinterpret_cast<void**>(&c.d_part), cap * 20));
        c.part_cap = cap;
    }
    if (staged && NBUF * frame_bytes(hw) > c.stage_bytes) {
        PG_HIP(h, hipDeviceSynchronize());
        if (c.h_stage) { PG_HIP(h, hipHostFree(c.h_stage)); c.h_stage = nullptr; c.stage_bytes = 0; }
        PG_HIP(h, hipHostMalloc(&c.h_stage, NBUF * frame_bytes(hw), hipHostMallocDefault));
        c.stage_bytes = NBUF * frame_bytes(hw);
    }
    return PG_OK;
}

// page-locked host memory (hipHostMalloc / hipHostRegister, e.g. a torch tensor with pin_memory=True)?
bool host_pinned(const void* p) {
    if (!p) return true;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// Output pipeline of one worker: frame k of the worker composes into device buffer k % NBUF on the render stream
// while frame k - 1 is copied to the host on the copy stream; nothing blocks the host thread before the call's end
// (results in pageable memory go through pinned staging and are moved by the worker thread one frame later).
struct FrameOut {
    pg_handle* h;
    hipStream_t st;
    size_t hw;
    float* rgbs; float* disps; float* accs; uint8_t* rgb8;
    bool staged;
    const float* d_bg;
    float base_bg;
    int k = 0;
    int pend[NBUF];
    FrameOut(pg_handle* h_, hipStream_t st_, size_t hw_, float* r, float* d, float* a, uint8_t* u8, bool stg, const float* bg, float bb)
        : h(h_), st(st_), hw(hw_), rgbs(r), disps(d), accs(a), rgb8(u8), staged(stg), d_bg(bg), base_bg(bb) {
        for (int b = 0; b < NBUF; ++b) pend[b] = -1;
    }
    uint8_t* stage(int b) const { return static_cast<uint8_t*>(h->fc.h_stage) + (size_t)b * frame_bytes(hw); }
    int drain(int b) {              // the staged frame of buffer b -> the caller's (pageable) arrays
        const int f = pend[b];
        if (f < 0) return PG_OK;
        PG_HIP(h, hipEventSynchronize(h->fc.copied[b]));
        const uint8_t* sp = stage(b);
        if (rgbs) std::memcpy(rgbs + (size_t)f * hw * 3, sp, hw * 12);
        if (disps) std::memcpy(disps + (size_t)f * hw, sp + hw * 12, hw * 4);
        if (accs) std::memcpy(accs + (size_t)f * hw, sp + hw * 16, hw * 4);
        if (rgb8) std::memcpy(rgb8 + (size_t)f * hw * 3, sp + hw * 20, hw * 3);
        pend[b] = -1;
        return PG_OK;
    }
    int put(int f, const pgk::FrameGeom& g, const FrameMaps& maps) {
        auto& c = h->fc;
        const int b = k++ % NBUF;
        if (staged) { const int rc = drain(b); if (rc) return rc; }
        PG_HIP(h, hipStreamWaitEvent(st, c.copied[b], 0));          // buffer b's previous copy-out (no-op before the first)
        uint8_t* base = reinterpret_cast<uint8_t*>(c.d_frame[b]);
        float* d_rgb = reinterpret_cast<float*>(base);
        float* d_disp = reinterpret_cast<float*>(base + hw * 12);
        float* d_acc = reinterpret_cast<float*>(base + hw * 16);
        uint8_t* d_u8 = rgb8 ? base + hw * 20 : nullptr;
        int rc = frame_compose(h, st, g, maps, d_bg, base_bg, d_rgb, d_disp, d_acc, d_u8);
        if (rc) return rc;
        PG_HIP(h, hipEventRecord(c.composed[b], st));
        PG_HIP(h, hipStreamWaitEvent(c.copy_stream, c.composed[b], 0));
        uint8_t* sp = staged ? stage(b) : nullptr;
        if (rgbs) PG_HIP(h, hipMemcpyAsync(staged ? (void*)sp : (void*)(rgbs + (size_t)f * hw * 3), d_rgb, hw * 12, hipMemcpyDeviceToHost, c.copy_stream));
        if (disps) PG_HIP(h, hipMemcpyAsync(staged ? (void*)(sp + hw * 12) : (void*)(disps + (size_t)f * hw), d_disp, hw * 4, hipMemcpyDeviceToHost, c.copy_stream));
        if (accs) PG_HIP(h, hipMemcpyAsync(staged ? (void*)(sp + hw * 16) : (void*)(accs + (size_t)f * hw), d_acc, hw * 4, hipMemcpyDeviceToHost, c.copy_stream));
        if (rgb8) PG_HIP(h, hipMemcpyAsync(staged ? (void*)(sp + hw * 20) : (void*)(rgb8 + (size_t)f * hw * 3), d_u8, hw * 3, hipMemcpyDeviceToHost, c.copy_stream));
        PG_HIP(h, hipEventRecord(c.copied[b], c.copy_stream));
        if (staged) pend[b] = f;
        return PG_OK;
    }
    int finish() {
        if (staged)
            for (int b = 0; b < NBUF; ++b) { const int rc = drain(b); if (rc) return rc; }
        PG_HIP(h, hipStreamSynchronize(h->fc.copy_stream));
        return PG_OK;
    }
};

}  // namespace

extern "C" {

int pg_render_frames(pg_handle* h, int n_frames, int H, int W, const float* c2ws, const float* intrinsics, const int* boxes,
                     float near, float far, const float* skts, const float* cyls, const float* cams, int n_samples,
                     int n_importance, int flags, const float* bg, float base_bg, float* rgbs, float* disps, float* accs,
                     uint8_t* rgb8) {
    if (!h) return pg_fail(nullptr, PG_EINVAL, "null handle");
    if (n_frames < 0 || H <= 0 || W <= 0 || !c2ws || !intrinsics || !boxes || !skts || !cyls || (!rgbs && !rgb8))
        return pg_fail(h, PG_EINVAL, "pg_render_frames: null/negative argument");
    if (n_frames == 0) return PG_OK;
    // Worker 0 runs on the primary handle's own stream over the primary's workspaces: everything the caller
    // queued on ITS stream (pg_render_rays / pg_render_frame are asynchronous and use the same buffers) must
    // have finished first.  The call is synchronous anyway.
    PG_HIP(h, hipSetDevice(h->device));
    PG_HIP(h, hipDeviceSynchronize());
    std::vector<pg_handle*> wk;
    wk.push_back(h);
    for (pg_handle* s : h->peers) wk.push_back(s);
    const int G = (int)wk.size();
    const size_t hw = (size_t)H * W;
    std::vector<pgk::FrameGeom> geo(n_frames);
    std::vector<int64_t> nr(n_frames);
    for (int f = 0; f < n_frames; ++f) {
        const int rc = frame_geom(h, H, W, c2ws + 12 * f, intrinsics + 4 * f, boxes + 4 * f, near, far, cams ? cams[f] : -1.0f, &geo[f]);
        if (rc) return rc;
        nr[f] = (int64_t)geo[f].bw * geo[f].bh;
    }
    std::vector<FrameTask> tasks;
    plan_frames(nr, G, h->cfg.chunk, &tasks);
    auto whole = [&](const FrameTask& tk) { return tk.r0 == 0 && tk.r1 == nr[tk.frame]; };
    // Everything a worker needs is set up BEFORE its first launch and kept on its handle between calls (no allocation
    // in a call whose sizes have been seen): the poses of all frames (one upload), two frame buffers in rotation, the
    // background, one buffer for the packed maps of the cut-frame ranges it renders, pinned staging when the
    // caller's result arrays are pageable.
    const bool staged = !(host_pinned(rgbs) && host_pinned(disps) && host_pinned(accs) && host_pinned(rgb8));
    std::vector<size_t> part_off(tasks.size(), 0), part_rays(G, 0);
    std::vector<char> composes(G, 0);
    for (size_t t = 0; t < tasks.size(); ++t) {
        const FrameTask& tk = tasks[t];
        if (whole(tk)) { composes[tk.worker] = 1; continue; }
        composes[tk.owner] = 1;
        part_off[t] = part_rays[tk.worker];
        part_rays[tk.worker] += (size_t)(tk.r1 - tk.r0);
    }
    for (int k = 0; k < G; ++k) {
        const int rc = frames_cache_ensure(wk[k], composes[k] ? hw : 0, n_frames, part_rays[k], composes[k] ? bg : nullptr, staged && composes[k]);
        if (rc) { (void)hipSetDevice(h->device); return wk[k] == h ? rc : pg_fail(h, rc, "device %d: %s", wk[k]->device, wk[k]->err); }
    }
    struct Worker { pg_handle* h; int rc = PG_OK; };
    std::vector<Worker> ws(G);
    for (int k = 0; k < G; ++k) ws[k].h = wk[k];
    std::vector<FrameMaps> maps_of(tasks.size());
    std::vector<FrameOut> outs;
    outs.reserve(G);
    for (int k = 0; k < G; ++k)
        outs.emplace_back(wk[k], wk[k]->own_stream, hw, rgbs, disps, accs, rgb8, staged, bg ? wk[k]->fc.d_bg : nullptr, base_bg);

    auto checker = [&](Worker& w) {
        return [&w](hipError_t e, const char* what) {
            if (e != hipSuccess && w.rc == PG_OK) w.rc = pg_fail(w.h, PG_EHIP, "%s failed on device %d: %s", what, w.h->device, hipGetErrorString(e));
            return e == hipSuccess;
        };
    };
    // phase A: every worker renders its tasks; whole frames are composed and handed to the output pipeline at once,
    // the runs of cut frames stay in the worker's range buffer for the owner
    auto phase_a = [&](int k) {
        Worker& w = ws[k];
        pg_handle* hh = w.h;
        auto check = checker(w);
        if (!check(hipSetDevice(hh->device), "hipSetDevice")) return;
        hipStream_t st = hh->own_stream;
        float* d_skts = hh->fc.d_poses;
        float* d_cyls = hh->fc.d_poses + (size_t)n_frames * 384;
        // (pageable sources: the runtime stages them before the calls return; the kernels are ordered behind on `st`)
        if (!check(hipMemcpyAsync(d_skts, skts, (size_t)n_frames * 384 * sizeof(float), hipMemcpyHostToDevice, st), "pose upload")) return;
        if (!check(hipMemcpyAsync(d_cyls, cyls, (size_t)n_frames * 5 * sizeof(float), hipMemcpyHostToDevice, st), "cylinder upload")) return;
        for (size_t t = 0; t < tasks.size() && w.rc == PG_OK; ++t) {
            const FrameTask& tk = tasks[t];
            if (tk.worker != k) continue;
            const int f = tk.frame;
            if (whole(tk)) {
                w.rc = frame_render_range(hh, st, geo[f], tk.r0, tk.r1, d_skts + (size_t)f * 384, d_cyls + (size_t)f * 5, n_samples, n_importance, flags, &maps_of[t]);
                if (w.rc) return;
                w.rc = outs[k].put(f, geo[f], maps_of[t]);
            } else {
                const size_t n = (size_t)(tk.r1 - tk.r0);
                if (n == 0) continue;
                float* part = hh->fc.d_part + part_off[t] * 5;
                const FrameMaps ext{part, part + n * 3, part + n * 4};
                w.rc = frame_render_range(hh, st, geo[f], tk.r0, tk.r1, d_skts + (size_t)f * 384, d_cyls + (size_t)f * 5, n_samples, n_importance, flags, &maps_of[t], &ext);
            }
        }
        if (w.rc == PG_OK) check(hipStreamSynchronize(st), "hipStreamSynchronize");      // phase B reads other workers' range buffers
    };
    // phase B: the owner of a cut frame gathers all its runs (device to device), composes, copies out
    auto phase_b = [&](int k) {
        Worker& w = ws[k];
        pg_handle* hh = w.h;
        auto check = checker(w);
        if (!check(hipSetDevice(hh->device), "hipSetDevice")) return;
        hipStream_t st = hh->own_stream;
        for (int f = 0; f < n_frames && w.rc == PG_OK; ++f) {
            bool mine = false;
            for (const FrameTask& tk : tasks) mine = mine || (tk.frame == f && tk.owner == k && !whole(tk));
            if (!mine) continue;
            float *rays, *cams_d;
            FrameMaps box{};
            pg_outputs scratch{};
            w.rc = frame_ws(hh, nr[f], 0, &rays, &cams_d, &box, &scratch);     // phase A is over: the workspace is free
            if (w.rc) return;
            for (size_t u = 0; u < tasks.size(); ++u) {
                const FrameTask& pt = tasks[u];
                if (pt.frame != f || pt.r1 == pt.r0) continue;
                const int src_dev = wk[pt.worker]->device;
                const size_t n = (size_t)(pt.r1 - pt.r0);
                if (!check(hipMemcpyPeerAsync(box.rgb_map + pt.r0 * 3, hh->device, maps_of[u].rgb_map, src_dev, n * 12, st), "peer copy") ||
                    !check(hipMemcpyPeerAsync(box.disp_map + pt.r0, hh->device, maps_of[u].disp_map, src_dev, n * 4, st), "peer copy") ||
                    !check(hipMemcpyPeerAsync(box.acc_map + pt.r0, hh->device, maps_of[u].acc_map, src_dev, n * 4, st), "peer copy")) return;
            }
            w.rc = outs[k].put(f, geo[f], box);
        }
    };
    auto finish = [&](int k) {
        Worker& w = ws[k];
        if (w.rc != PG_OK || !composes[k]) return;
        if (!checker(w)(hipSetDevice(w.h->device), "hipSetDevice")) return;
        w.rc = outs[k].finish();
    };
    auto run = [&](auto&& fn) {
        std::vector<std::thread> th;
        for (int k = 1; k < G; ++k) th.emplace_back(fn, k);
        fn(0);
        for (auto& t : th) t.join();
    };
    run(phase_a);
    bool split = false;
    for (const FrameTask& tk : tasks) split = split || !whole(tk);
    bool ok = true;
    for (const Worker& w : ws) ok = ok && w.rc == PG_OK;
    if (ok && split) run(phase_b);
    run(finish);
    int rc = PG_OK;
    for (Worker& w : ws) {
        if (w.rc != PG_OK) {        // nothing of a failed call may still be in flight when the caller's arrays go away
            (void)hipSetDevice(w.h->device);
            (void)hipDeviceSynchronize();
            if (rc == PG_OK) rc = (w.h == h) ? w.rc : pg_fail(h, w.rc, "device %d: %s", w.h->device, w.h->err);
        }
    }
    (void)hipSetDevice(h->device);
    return rc;
}

}  // extern "C"
