// Hardware probe: VALU writes an MFMA source VGPR and the MFMA issues right after (RAW, no wait
// states).  Which VALU opcodes need padding on gfx950 for v_mfma_f32_32x32x16_bf16?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ZERO16(base) \
    "v_accvgpr_write_b32 a" #base "+0, v29\n"
template <int OP>
__global__ void probe(const unsigned* in, unsigned* out) {
    const int lane = threadIdx.x & 63;
    unsigned a0 = in[lane * 8 + 0], a1 = in[lane * 8 + 1], a2 = in[lane * 8 + 2], a3 = in[lane * 8 + 3];
    unsigned b0 = in[lane * 8 + 4], b1 = in[lane * 8 + 5], b2 = in[lane * 8 + 6], b3 = in[lane * 8 + 7];
    float f0 = __uint_as_float(in[lane * 8 + 1] & 0x3fffffffu), f1 = __uint_as_float(in[lane * 8 + 2] & 0x3fffffffu);
    unsigned r[8];
    // v24..v27 = B operand.  Padded reference first (a[16:31]), then unpadded victim (a[0:15]).
#define PROLOGUE \
        "v_mov_b32 v20, %8\n v_mov_b32 v21, %9\n v_mov_b32 v22, %10\n v_mov_b32 v23, %11\n" \
        "v_mov_b32 v30, %12\n v_mov_b32 v31, %13\n v_mov_b32 v32, %14\n v_mov_b32 v33, %15\n" \
        "v_mov_b32 v34, %16\n v_mov_b32 v35, %17\n v_mov_b32 v29, 0\n" \
        "v_accvgpr_write_b32 a0, v29\n v_accvgpr_write_b32 a1, v29\n v_accvgpr_write_b32 a2, v29\n v_accvgpr_write_b32 a3, v29\n" \
        "v_accvgpr_write_b32 a4, v29\n v_accvgpr_write_b32 a5, v29\n v_accvgpr_write_b32 a6, v29\n v_accvgpr_write_b32 a7, v29\n" \
        "v_accvgpr_write_b32 a8, v29\n v_accvgpr_write_b32 a9, v29\n v_accvgpr_write_b32 a10, v29\n v_accvgpr_write_b32 a11, v29\n" \
        "v_accvgpr_write_b32 a12, v29\n v_accvgpr_write_b32 a13, v29\n v_accvgpr_write_b32 a14, v29\n v_accvgpr_write_b32 a15, v29\n" \
        "v_accvgpr_write_b32 a16, v29\n v_accvgpr_write_b32 a17, v29\n v_accvgpr_write_b32 a18, v29\n v_accvgpr_write_b32 a19, v29\n" \
        "v_accvgpr_write_b32 a20, v29\n v_accvgpr_write_b32 a21, v29\n v_accvgpr_write_b32 a22, v29\n v_accvgpr_write_b32 a23, v29\n" \
        "v_accvgpr_write_b32 a24, v29\n v_accvgpr_write_b32 a25, v29\n v_accvgpr_write_b32 a26, v29\n v_accvgpr_write_b32 a27, v29\n" \
        "v_accvgpr_write_b32 a28, v29\n v_accvgpr_write_b32 a29, v29\n v_accvgpr_write_b32 a30, v29\n v_accvgpr_write_b32 a31, v29\n" \
        "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n s_nop 7\n"
#define EPILOGUE \
        "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n" \
        "v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a5\n v_accvgpr_read_b32 %2, a10\n v_accvgpr_read_b32 %3, a15\n" \
        "v_accvgpr_read_b32 %4, a16\n v_accvgpr_read_b32 %5, a21\n v_accvgpr_read_b32 %6, a26\n v_accvgpr_read_b32 %7, a31\n"
#define OPERANDS \
        : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) \
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(f0), "v"(f1) \
        : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v29", "v30", "v31", "v32", "v33", "v34", "v35", \
          "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15", \
          "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"
#define WRITES_MOV   "v_mov_b32 v24, v30\n v_mov_b32 v25, v31\n v_mov_b32 v26, v32\n v_mov_b32 v27, v33\n"
#define WRITES_CVT   "v_cvt_pk_bf16_f32 v24, v34, v35\n v_cvt_pk_bf16_f32 v25, v35, v34\n v_cvt_pk_bf16_f32 v26, v34, v34\n v_cvt_pk_bf16_f32 v27, v35, v35\n"
#define WRITES_PERM  "v_perm_b32 v24, v30, v31, v32\n v_alignbit_b32 v25, v31, v30, 16\n v_alignbit_b32 v26, v32, v31, 16\n v_alignbit_b32 v27, v33, v32, 16\n"
#define WRITES_PK    "v_pk_add_f32 v[24:25], v[34:35], v[34:35]\n v_pk_add_f32 v[26:27], v[34:35], v[34:35]\n"
#define RESET        "s_nop 15\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n s_nop 7\n"
#define BODY(W) \
    asm volatile(PROLOGUE W "s_nop 15\n" \
        "v_mfma_f32_32x32x16_bf16 a[16:31], v[20:23], v[24:27], a[16:31]\n" RESET \
        W "v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[24:27], a[0:15]\n" EPILOGUE OPERANDS)
    if (OP == 0) BODY(WRITES_MOV);
    if (OP == 1) BODY(WRITES_CVT);
    if (OP == 2) BODY(WRITES_PERM);
    if (OP == 3) BODY(WRITES_PK);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < 8; ++k) out[gid * 8 + k] = r[k];
}
template <int OP> void run(const char* name, unsigned* din, unsigned* dout, std::vector<unsigned>& out, int blocks, int threads) {
    long bad = 0; long badl[64] = {0};
    for (int rep = 0; rep < 10; ++rep) {
        hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, din, dout);
        (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        for (size_t t = 0; t < (size_t)blocks * threads; ++t)
            for (int k = 0; k < 4; ++k) if (out[t * 8 + k] != out[t * 8 + 4 + k]) { ++bad; ++badl[t & 63]; }
    }
    printf("%-28s mismatches %ld\n", name, bad);
    if (bad) { printf("  per-lane:"); for (int l = 0; l < 64; ++l) printf(" %ld", badl[l]); printf("\n"); }
}
int main() {
    std::vector<unsigned> in(64 * 8);
    unsigned s = 12345;
    for (auto& v : in) { s = s * 1664525u + 1013904223u; unsigned hi = 0x3f80u + ((s >> 9) & 0x7f), lo = 0x3f80u + ((s >> 20) & 0x7f); v = (hi << 16) | lo; }
    unsigned *din, *dout; const int blocks = 1024, threads = 256;
    (void)hipMalloc(&din, in.size() * 4); (void)hipMalloc(&dout, (size_t)blocks * threads * 8 * 4);
    (void)hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> out((size_t)blocks * threads * 8);
    run<0>("v_mov_b32 -> mfma SrcB", din, dout, out, blocks, threads);
    run<1>("v_cvt_pk_bf16_f32 -> SrcB", din, dout, out, blocks, threads);
    run<2>("v_perm/v_alignbit -> SrcB", din, dout, out, blocks, threads);
    run<3>("v_pk_add_f32 -> SrcB", din, dout, out, blocks, threads);
    return 0;
}
