// pg_handle.h -- the renderer handle behind the C ABI (include/posegen_hip.h), shared by the translation units
// that implement its entry points (pg_api.hip: rendering; pg_train.hip: the training step).
#pragma once
#include <hip/hip_runtime.h>

#include <utility>
#include <vector>

#include "../../include/posegen_hip.h"
#include "pg_layout.h"

struct NetState {
    bool loaded = false;
    std::vector<std::vector<float>> host;      // 24 tensors, reference order (see header)
    std::vector<float> codes_host;             // [n_codes+1,16]
    mutable std::vector<float> fold_w, fold_b; // W_view[:, :256] W_feature and its bias (NetTensors::fold), formed once per pg_load_weights
    int n_codes = 0;
    uint8_t* d_stream[PG_PREC_COUNT][2] = {};     // [precision][factorised view layer]
    uint8_t* d_vy[PG_PREC_COUNT] = {};            // Y-stage weights of the per-ray record kernel (pg_rayrec.hip)
    uint8_t* d_stream_ro[PG_PREC_COUNT] = {};     // 16x16x32 kernel, on-chip variant (no per-ray records): stream
    uint8_t* d_stream_r[PG_PREC_COUNT] = {};      // 16x16x32 kernel with per-ray records (pg_eval16r.hip): stream,
    float* d_bias_s = nullptr;                    // ... and its 16-row bias table
    float* d_ycode = nullptr;                     // on-chip variant with frame codes: Yc[n_codes + 1][128] = W_view[:, 904:920] codes[c] (ensure_ycode)
    uint8_t* d_c2 = nullptr;                      // compensated-fp16 kernel with the out tiles over the waves (pg_evalc2.hip): weights (pg_program.h T)
    uint8_t* d_stream_co = nullptr;               // compensated-fp16 kernel, on-chip form of the record variant (pg_evalc.hip OC): stream
    uint8_t* d_stream_cr = nullptr;               // compensated-fp16 kernel, record variant (pg_evalc.hip REC): stream,
    float* d_vyc = nullptr;                       // ... and the fp32 Y-stage weights of its record kernel
    size_t stream_bytes[PG_PREC_COUNT][2] = {};
    float* d_bias = nullptr;
    float* d_codes = nullptr;
    // pg_load_weights_device: the net's tensors as one flat device vector (NetTensors::layout; + the folded view layer), the
    // source maps of the images that are re-formed by a gather, and whether `host` lags the device copy
    float* d_src = nullptr;
    int32_t* d_map_ro = nullptr;   size_t n_map_ro = 0;      // on-chip stream of the 16x16x32 kernel (one map for bf16 and fp16)
    int32_t* d_map_c2 = nullptr;   size_t n_map_c2 = 0;      // pg_evalc2.hip's weight image
    int32_t* d_map_bias_s = nullptr;
    bool host_stale = false;
};


struct pg_handle {
    pg_config cfg;
    int device = 0;
    int n_cu = 256;
    int clock_khz = 0;
    char err[512] = "";
    NetState net[2];
    float cut[48];
    float tau[2] = {20.f, 20.f};
    bool emb_set[2] = {false, false};
    float* d_cut = nullptr;
    uint8_t* ws = nullptr;
    size_t ws_bytes = 0;
    uint8_t* fws = nullptr;          // frame front/back end: ray_batch, cams, rgb/disp/acc maps of the box
    size_t fws_bytes = 0;
    double* sc_part = nullptr;       // partial nanmean sums of the two-launch coarse sampler (pg_kernels.hip)
    size_t sc_part_cap = 0;          // ... doubles
    uint8_t* rec = nullptr;          // per-ray records of the factorised 16-bit path: Y [n + pad, 8 KiB] then (a, b) [n + pad, 768 B]
    size_t rec_bytes = 0;
    long long rec_pad_n = -1;        // (n, Y bytes per ray) whose padding records are currently zero (-1: none)
    int rec_pad_y = 0;
    void* rec_pad_stream = nullptr;  // the stream that memset was issued on: a call on another stream zeroes again (no ordering between streams)
    // in-process multi-device rendering (pg_render_frames): the primary handle owns one sub-handle per
    // further device; every handle has a stream and a small pose buffer of its own for that path
    std::vector<pg_handle*> peers;
    hipStream_t own_stream = nullptr;
    float* d_pose = nullptr;         // [24*16 + 5 + pad] skts + cyl of the frame being rendered
    bool far_skip = true;            // pg_set_far_skip (test / measurement aid)
    int onchip_mode = PG_ONCHIP_AUTO;        // pg_set_onchip (initial value: POSEGEN_ONCHIP)
    int train_precision = PG_PREC_FP32;      // pg_set_train_precision: arithmetic of the training step (fp32 like the reference, or bf16)
    bool profiling = false;
    std::vector<hipEvent_t> ev_free;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_used;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_aux;      // the record kernel in front of a factorised launch
    int64_t aux_n = 0;               // record launches / their device time already folded in by a pg_profile_read
    double aux_ms = 0.0;
    int64_t prof_points = 0;
    void* train = nullptr;           // the training tape (pg_train.hip): activations of the last pg_train_forward
    // pg_render_frames: per-device buffers kept between calls (frames of H x W pixels, background, pinned staging)
    struct FramesCache {
        size_t hw = 0;               // pixels the frame buffers were sized for
        bool has_u8 = false;
        static constexpr int NBUF = 2;               // frame buffers in rotation: frame k+1 composes while frame k copies out
        float* d_frame[NBUF] = {};                   // rgb [hw,3] | disp [hw] | acc [hw] (| rgb8 [hw,3] bytes behind)
        hipEvent_t composed[NBUF] = {};              // buffer b holds a finished frame (render stream)
        hipEvent_t copied[NBUF] = {};                // the device-to-host copy out of buffer b has finished (copy stream)
        hipStream_t copy_stream = nullptr;
        float* d_bg = nullptr;                       // background [hw,3] (uploaded per call when given)
        size_t bg_hw = 0;
        float* d_poses = nullptr;                    // skts + cyls of all frames of the call: [F, 384 + 8]
        size_t poses_cap = 0;                        // ... frames
        float* d_part = nullptr;                     // packed maps (20 B per ray) of the ray ranges of cut frames this device renders
        size_t part_cap = 0;                         // ... rays
        void* h_stage = nullptr;                     // pinned host staging of NBUF frames (results that land in pageable memory)
        size_t stage_bytes = 0;
    } fc;
};

extern "C" void pg_train_release(pg_handle* h);
// scratch of pg_launch_sample_coarse for n rays in chunks of `chunk` (null when the one-launch form runs)
int pg_sc_scratch(pg_handle* h, long long n, int chunk, double** out);

// records the message (handle and thread-local "last error") and returns `code`
int pg_fail(pg_handle* h, int code, const char* fmt, ...);

#define PG_HIP(h, call)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return pg_fail(h, PG_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
