// Hardware probe: is "v_mfma reads SrcB, next VALU overwrites that VGPR" (WAR) safe on gfx950
// for v_mfma_f32_32x32x16_bf16 when the matrix pipe is busy with earlier MFMAs?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned* in, unsigned* out, int variant) {
    const int lane = threadIdx.x & 63;
    unsigned a0 = in[lane * 8 + 0], a1 = in[lane * 8 + 1], a2 = in[lane * 8 + 2], a3 = in[lane * 8 + 3];
    unsigned b0 = in[lane * 8 + 4], b1 = in[lane * 8 + 5], b2 = in[lane * 8 + 6], b3 = in[lane * 8 + 7];
    unsigned garbage = 0x7f7f7f7fu;
    unsigned r0, r1, r2, r3, q0, q1, q2, q3;
    if (variant == 0) {
    asm volatile(
        "v_mov_b32 v20, %8\n v_mov_b32 v21, %9\n v_mov_b32 v22, %10\n v_mov_b32 v23, %11\n"
        "v_mov_b32 v24, %12\n v_mov_b32 v25, %13\n v_mov_b32 v26, %14\n v_mov_b32 v27, %15\n"
        "v_mov_b32 v28, %16\n"
        "v_mov_b32 v29, 0\n"
        "v_accvgpr_write_b32 a0, v29\n v_accvgpr_write_b32 a1, v29\n v_accvgpr_write_b32 a2, v29\n v_accvgpr_write_b32 a3, v29\n"
        "v_accvgpr_write_b32 a4, v29\n v_accvgpr_write_b32 a5, v29\n v_accvgpr_write_b32 a6, v29\n v_accvgpr_write_b32 a7, v29\n"
        "v_accvgpr_write_b32 a8, v29\n v_accvgpr_write_b32 a9, v29\n v_accvgpr_write_b32 a10, v29\n v_accvgpr_write_b32 a11, v29\n"
        "v_accvgpr_write_b32 a12, v29\n v_accvgpr_write_b32 a13, v29\n v_accvgpr_write_b32 a14, v29\n v_accvgpr_write_b32 a15, v29\n"
        "v_accvgpr_write_b32 a16, v29\n v_accvgpr_write_b32 a17, v29\n v_accvgpr_write_b32 a18, v29\n v_accvgpr_write_b32 a19, v29\n"
        "v_accvgpr_write_b32 a20, v29\n v_accvgpr_write_b32 a21, v29\n v_accvgpr_write_b32 a22, v29\n v_accvgpr_write_b32 a23, v29\n"
        "v_accvgpr_write_b32 a24, v29\n v_accvgpr_write_b32 a25, v29\n v_accvgpr_write_b32 a26, v29\n v_accvgpr_write_b32 a27, v29\n"
        "v_accvgpr_write_b32 a28, v29\n v_accvgpr_write_b32 a29, v29\n v_accvgpr_write_b32 a30, v29\n v_accvgpr_write_b32 a31, v29\n"
        "v_accvgpr_write_b32 a32, v29\n v_accvgpr_write_b32 a33, v29\n v_accvgpr_write_b32 a34, v29\n v_accvgpr_write_b32 a35, v29\n"
        "v_accvgpr_write_b32 a36, v29\n v_accvgpr_write_b32 a37, v29\n v_accvgpr_write_b32 a38, v29\n v_accvgpr_write_b32 a39, v29\n"
        "v_accvgpr_write_b32 a40, v29\n v_accvgpr_write_b32 a41, v29\n v_accvgpr_write_b32 a42, v29\n v_accvgpr_write_b32 a43, v29\n"
        "v_accvgpr_write_b32 a44, v29\n v_accvgpr_write_b32 a45, v29\n v_accvgpr_write_b32 a46, v29\n v_accvgpr_write_b32 a47, v29\n"
        "s_nop 7\n"
        "v_mfma_f32_32x32x16_bf16 a[16:31], v[20:23], v[24:27], a[16:31]\n"   // reference result
        "v_mfma_f32_32x32x16_bf16 a[32:47], v[20:23], v[24:27], a[32:47]\n"   // filler keeps pipe busy
        "v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[24:27], a[0:15]\n"     // victim
        "v_mov_b32 v24, v28\n v_mov_b32 v25, v28\n v_mov_b32 v26, v28\n v_mov_b32 v27, v28\n"   // WAR on SrcB
        "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
        "v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a5\n v_accvgpr_read_b32 %2, a10\n v_accvgpr_read_b32 %3, a15\n"
        "v_accvgpr_read_b32 %4, a16\n v_accvgpr_read_b32 %5, a21\n v_accvgpr_read_b32 %6, a26\n v_accvgpr_read_b32 %7, a31\n"
        : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(garbage)
        : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29",
          "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15",
          "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31",
          "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47");
    } else {
    // same with SrcA overwritten
    asm volatile(
        "v_mov_b32 v20, %8\n v_mov_b32 v21, %9\n v_mov_b32 v22, %10\n v_mov_b32 v23, %11\n"
        "v_mov_b32 v24, %12\n v_mov_b32 v25, %13\n v_mov_b32 v26, %14\n v_mov_b32 v27, %15\n"
        "v_mov_b32 v28, %16\n"
        "v_mov_b32 v29, 0\n"
        "v_accvgpr_write_b32 a0, v29\n v_accvgpr_write_b32 a1, v29\n v_accvgpr_write_b32 a2, v29\n v_accvgpr_write_b32 a3, v29\n"
        "v_accvgpr_write_b32 a4, v29\n v_accvgpr_write_b32 a5, v29\n v_accvgpr_write_b32 a6, v29\n v_accvgpr_write_b32 a7, v29\n"
        "v_accvgpr_write_b32 a8, v29\n v_accvgpr_write_b32 a9, v29\n v_accvgpr_write_b32 a10, v29\n v_accvgpr_write_b32 a11, v29\n"
        "v_accvgpr_write_b32 a12, v29\n v_accvgpr_write_b32 a13, v29\n v_accvgpr_write_b32 a14, v29\n v_accvgpr_write_b32 a15, v29\n"
        "v_accvgpr_write_b32 a16, v29\n v_accvgpr_write_b32 a17, v29\n v_accvgpr_write_b32 a18, v29\n v_accvgpr_write_b32 a19, v29\n"
        "v_accvgpr_write_b32 a20, v29\n v_accvgpr_write_b32 a21, v29\n v_accvgpr_write_b32 a22, v29\n v_accvgpr_write_b32 a23, v29\n"
        "v_accvgpr_write_b32 a24, v29\n v_accvgpr_write_b32 a25, v29\n v_accvgpr_write_b32 a26, v29\n v_accvgpr_write_b32 a27, v29\n"
        "v_accvgpr_write_b32 a28, v29\n v_accvgpr_write_b32 a29, v29\n v_accvgpr_write_b32 a30, v29\n v_accvgpr_write_b32 a31, v29\n"
        "v_accvgpr_write_b32 a32, v29\n v_accvgpr_write_b32 a33, v29\n v_accvgpr_write_b32 a34, v29\n v_accvgpr_write_b32 a35, v29\n"
        "v_accvgpr_write_b32 a36, v29\n v_accvgpr_write_b32 a37, v29\n v_accvgpr_write_b32 a38, v29\n v_accvgpr_write_b32 a39, v29\n"
        "v_accvgpr_write_b32 a40, v29\n v_accvgpr_write_b32 a41, v29\n v_accvgpr_write_b32 a42, v29\n v_accvgpr_write_b32 a43, v29\n"
        "v_accvgpr_write_b32 a44, v29\n v_accvgpr_write_b32 a45, v29\n v_accvgpr_write_b32 a46, v29\n v_accvgpr_write_b32 a47, v29\n"
        "s_nop 7\n"
        "v_mfma_f32_32x32x16_bf16 a[16:31], v[20:23], v[24:27], a[16:31]\n"
        "v_mfma_f32_32x32x16_bf16 a[32:47], v[20:23], v[24:27], a[32:47]\n"
        "v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[24:27], a[0:15]\n"
        "v_mov_b32 v20, v28\n v_mov_b32 v21, v28\n v_mov_b32 v22, v28\n v_mov_b32 v23, v28\n"   // WAR on SrcA
        "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
        "v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a5\n v_accvgpr_read_b32 %2, a10\n v_accvgpr_read_b32 %3, a15\n"
        "v_accvgpr_read_b32 %4, a16\n v_accvgpr_read_b32 %5, a21\n v_accvgpr_read_b32 %6, a26\n v_accvgpr_read_b32 %7, a31\n"
        : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(garbage)
        : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29",
          "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15",
          "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31",
          "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47");
    }
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid * 8 + 0] = r0; out[gid * 8 + 1] = r1; out[gid * 8 + 2] = r2; out[gid * 8 + 3] = r3;
    out[gid * 8 + 4] = q0; out[gid * 8 + 5] = q1; out[gid * 8 + 6] = q2; out[gid * 8 + 7] = q3;
}
int main() {
    std::vector<unsigned> in(64 * 8);
    unsigned s = 12345;
    for (auto& v : in) { s = s * 1664525u + 1013904223u; unsigned hi = 0x3f80u + ((s >> 9) & 0x7f), lo = 0x3f80u + ((s >> 20) & 0x7f); v = (hi << 16) | lo; }
    unsigned *din, *dout;
    const int blocks = 1024, threads = 512;
    hipMalloc(&din, in.size() * 4); hipMalloc(&dout, (size_t)blocks * threads * 8 * 4);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> out((size_t)blocks * threads * 8);
    for (int variant = 0; variant < 2; ++variant) {
        long bad = 0; long badlanes[64] = {0};
        for (int rep = 0; rep < 20; ++rep) {
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, din, dout, variant);
            hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
            for (size_t t = 0; t < (size_t)blocks * threads; ++t)
                for (int k = 0; k < 4; ++k)
                    if (out[t * 8 + k] != out[t * 8 + 4 + k]) { ++bad; ++badlanes[t & 63]; }
        }
        printf("variant %d (%s overwritten right after issue): mismatches %ld\n", variant, variant ? "SrcA" : "SrcB", bad);
        if (bad) { printf("  per-lane:"); for (int l = 0; l < 64; ++l) printf(" %ld", badlanes[l]); printf("\n"); }
    }
    return 0;
}
