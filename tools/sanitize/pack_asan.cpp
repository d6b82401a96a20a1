// AddressSanitizer / UBSan run of the host packer (pg_pack.cpp) -- the only host-side native code with
// non-trivial index arithmetic -- on the CPU (GPU sanitizers are not available on the pool).
//   /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I posegen_amd/csrc -I include \
//       tools/sanitize/pack_asan.cpp posegen_amd/csrc/pg_pack.cpp -o /tmp/pack_asan && /tmp/pack_asan
#include <cstdio>
#include <random>
#include <vector>

#include "pg_pack.h"

int main() {
    using namespace pgpack;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-0.5f, 0.5f);
    int fails = 0;
    for (int fc = 0; fc < 2; ++fc) {
        const int in0 = 432, skip_in = 432 + 256, view_in = 256 + 648 + (fc ? 16 : 0);
        std::vector<std::vector<float>> store;
        auto mk = [&](size_t n) { store.emplace_back(n); for (auto& v : store.back()) v = u(rng); return store.back().data(); };
        NetTensors t;
        for (int l = 0; l < pgl::DEPTH; ++l) {
            t.lcols[l] = l == 0 ? in0 : (l == 5 ? skip_in : 256);
            t.lw[l] = mk((size_t)256 * t.lcols[l]);
            t.lb[l] = mk(256);
        }
        t.alpha_w = mk(256); t.alpha_b = mk(1);
        t.feat_w = mk(256 * 256); t.feat_b = mk(256);
        t.view_cols = view_in;
        t.view_w = mk((size_t)128 * view_in); t.view_b = mk(128);
        t.rgb_w = mk(3 * 128); t.rgb_b = mk(3);
        t.fold();
        for (int prec = 0; prec < PG_PREC_COUNT; ++prec)
            for (int fact = 0; fact < 2; ++fact) {
                std::vector<uint8_t> out;
                std::vector<int> base;
                const int rc = pack_stream(t, prec, fc != 0, fact != 0, out, &base);
                std::printf("fc=%d prec=%d fact=%d: rc=%d, %zu bytes, %zu segments\n", fc, prec, fact, rc, out.size(), base.size());
                if (rc != 0 && !(fact && prec != PG_PREC_FP16C)) ++fails;       // only fp16c has a second program here
            }
        for (int prec : {PG_PREC_BF16, PG_PREC_FP16}) {
            std::vector<uint8_t> s, vy;
            std::vector<float> b;
            if (pack_stream_r(t, prec, s) != 0) ++fails;
            pack_bias_s(t, b);
            if (pack_vy(t, prec, fc != 0, vy) != 0) ++fails;
            std::printf("fc=%d prec=%d: small-tile stream %zu bytes, vy %zu bytes\n", fc, prec, s.size(), vy.size());
        }
        {   // record variant of the compensated-fp16 kernel: stream without the view-direction segment + fp32 Y-stage weights
            std::vector<uint8_t> s;
            std::vector<float> vyc;
            if (pack_stream(t, PG_PREC_FP16C, fc != 0, true, s, nullptr, true) != 0) ++fails;
            pack_vyc(t, fc != 0, vyc);
            std::printf("fc=%d fp16c record variant: stream %zu bytes, vyc %zu floats\n", fc, s.size(), vyc.size());
        }
        std::vector<float> bias;
        pack_bias(t, bias);
    }
    std::printf(fails ? "FAILED %d\n" : "packer clean under ASan/UBSan\n", fails);
    return fails ? 1 : 0;
}
