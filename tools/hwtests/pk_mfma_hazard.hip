// Hardware probe (gfx950): a packed-f32 VALU op (v_pk_mul_f32) immediately followed by a v_mfma,
// then a VALU read of the packed result -- the instruction sequence to which round 1's intermittent
// wrong results of the split-operand kernels were traced (SLP-vectorised code of pg_eval32.hip:
//   ds_read2_b32 v[38:39] .. s_waitcnt lgkmcnt(0) / v_pk_mul_f32 v[38:39], v[38:39], v[8:9] op_sel:[0,1] /
//   v_mfma_f32_32x32x16_bf16 / v_cvt_pk_bf16_f32 v43, v38, v39:  the LOW result came back stale in
//   lanes 48..63; one s_nop between the packed op and the MFMA removes it).
// The probe rebuilds that context: operands fresh from LDS, a dependent MFMA chain in flight.
//   variant 0: as compiled      1: s_nop 0 between v_pk_mul_f32 and the MFMA      2: two v_mul_f32 instead
//   variant 3: as 0 without the MFMAs in flight before      4: as 0 with the result read by v_mov_b32
//   5 / 6: s_nop 1 / s_nop 7 between the packed op and the MFMA    7: no MFMA after the packed op    8: s_nop 7 between the
//   LDS wait and the packed op    9: v_pk_add_f32    10: s_nop 7 between the MFMA and the read    11: 128 wait states before the packed op
//   12: no op_sel (low x low, high x high)    13: op_sel_hi:[1,0] (both results use the LOW half of the second source)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int VAR>
__global__ void probe(const float* in, float* out, int reps) {
    __shared__ float tab[2 * 64 * 16];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * 64 * 16; i += blockDim.x) tab[i] = in[i % 256] + (float)(i / 256);
    __syncthreads();
    const float w0 = in[lane * 4 + 2], w1 = in[lane * 4 + 3];
    float bad = 0.f, what[1] = {0.f};
    for (int r = 0; r < reps; ++r) {
        const int idx = ((lane >> 5) * 64 + (r & 15)) * 2 + 3;           // 4-byte aligned pair straddling 16 B
        const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(tab + idx);
        unsigned packed;
        float lo, hi;
#define CLOB "v2", "v3", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
#define HEAD "v_mov_b32 v6, %3\n v_mov_b32 v7, %4\n v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n" \
             "v_cvt_pk_bf16_f32 v2, v6, v7\n v_mov_b32 v3, v2\n s_nop 7\n"                 \
             "ds_read2_b32 v[2:3], %5 offset1:1\n"
#define MF "v_mfma_f32_32x32x16_bf16 a[0:15], v[8:11], v[8:11], a[0:15]\n"
#define TAIL "v_cvt_pk_bf16_f32 v12, v2, v3\n s_nop 7\n s_nop 7\n v_mov_b32 %0, v12\n v_mov_b32 %1, v2\n v_mov_b32 %2, v3\n"
        if (VAR == 0) asm volatile(HEAD MF MF MF "s_waitcnt lgkmcnt(0)\n v_pk_mul_f32 v[2:3], v[2:3], v[6:7] op_sel:[0,1]\n" MF TAIL
                                   : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
        if (VAR == 1) asm volatile(HEAD MF MF MF "s_waitcnt lgkmcnt(0)\n v_pk_mul_f32 v[2:3], v[2:3], v[6:7] op_sel:[0,1]\n s_nop 0\n" MF TAIL
                                   : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
        if (VAR == 2) asm volatile(HEAD MF MF MF "s_waitcnt lgkmcnt(0)\n v_mul_f32 v2, v2, v7\n v_mul_f32 v3, v3, v7\n" MF TAIL
                                   : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
        if (VAR == 3) asm volatile(HEAD "s_waitcnt lgkmcnt(0)\n v_pk_mul_f32 v[2:3], v[2:3], v[6:7] op_sel:[0,1]\n" MF TAIL
                                   : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
        if (VAR == 4) asm volatile(HEAD MF MF MF "s_waitcnt lgkmcnt(0)\n v_pk_mul_f32 v[2:3], v[2:3], v[6:7] op_sel:[0,1]\n" MF
                                   "v_mov_b32 v12, v2\n s_nop 7\n s_nop 7\n v_mov_b32 %0, v12\n v_mov_b32 %1, v2\n v_mov_b32 %2, v3\n"
                                   : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
#define PKM "s_waitcnt lgkmcnt(0)\n v_pk_mul_f32 v[2:3], v[2:3], v[6:7] op_sel:[0,1]\n"
        if (VAR == 5) asm volatile(HEAD MF MF MF PKM "s_nop 1\n" MF TAIL : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
        if (VAR == 6) asm volatile(HEAD MF MF MF PKM "s_nop 7\n" MF TAIL : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
        if (VAR == 7) asm volatile(HEAD MF MF MF PKM TAIL : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
        if (VAR == 8) asm volatile(HEAD MF MF MF "s_waitcnt lgkmcnt(0)\n s_nop 7\n v_pk_mul_f32 v[2:3], v[2:3], v[6:7] op_sel:[0,1]\n" MF TAIL
                                   : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
        if (VAR == 9) asm volatile(HEAD MF MF MF "s_waitcnt lgkmcnt(0)\n v_pk_add_f32 v[2:3], v[2:3], v[6:7] op_sel:[0,1]\n" MF TAIL
                                   : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
        if (VAR == 10) asm volatile(HEAD MF MF MF PKM MF "s_nop 7\n" TAIL : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
        if (VAR == 11) asm volatile(HEAD MF MF MF "s_waitcnt lgkmcnt(0)\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n v_pk_mul_f32 v[2:3], v[2:3], v[6:7] op_sel:[0,1]\n" MF TAIL
                                   : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
        if (VAR == 12) asm volatile(HEAD MF MF MF "s_waitcnt lgkmcnt(0)\n v_pk_mul_f32 v[2:3], v[2:3], v[6:7]\n" MF TAIL
                                   : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
        if (VAR == 13) asm volatile(HEAD MF MF MF "s_waitcnt lgkmcnt(0)\n v_pk_mul_f32 v[2:3], v[2:3], v[6:7] op_sel_hi:[1,0]\n" MF TAIL
                                   : "=v"(packed), "=v"(lo), "=v"(hi) : "v"(w0), "v"(w1), "v"(addr) : CLOB);
        const float elo = VAR == 12 || VAR == 13 ? tab[idx] * w0 : (VAR == 9 ? tab[idx] + w1 : tab[idx] * w1), ehi = VAR == 13 ? tab[idx + 1] * w0 : (VAR == 9 ? tab[idx + 1] + w1 : tab[idx + 1] * w1);
        if (VAR == 4) { if (__uint_as_float(packed) != elo) bad += 1.f; }
        else {
            const unsigned e = (__float_as_uint((float)(__bf16)elo) >> 16) | (__float_as_uint((float)(__bf16)ehi) & 0xffff0000u);
            if ((packed & 0xffffu) != (e & 0xffffu)) {
                bad += 1.f;
                if (r == 0 || what[0] == 0.f) {          // classify one wrong value per lane
                    const unsigned raw = __float_as_uint((float)(__bf16)tab[idx]) >> 16;
                    const unsigned old = __float_as_uint((float)(__bf16)(VAR == 9 ? tab[idx] + w0 : tab[idx] * (VAR == 12 || VAR == 13 ? w1 : w0))) >> 16;
                    what[0] = (packed & 0xffffu) == raw ? 1.f : (packed & 0xffffu) == old ? 2.f : 3.f;
                }
            }
            if ((packed >> 16) != (e >> 16)) bad += 65536.f;
        }
        (void)lo; (void)hi;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = bad;
    out[gridDim.x * blockDim.x + blockIdx.x * blockDim.x + threadIdx.x] = what[0];
}
template <int VAR> void run(const float* din, float* dout, int blocks, int reps) {
    probe<VAR><<<blocks, 256>>>(din, dout, reps);
    std::vector<float> o(2 * blocks * 256);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    long lo_bad[4] = {0, 0, 0, 0}, hi_bad[4] = {0, 0, 0, 0};
    long cls[4] = {0, 0, 0, 0};
    for (size_t i = o.size() / 2; i < o.size(); ++i) cls[(int)o[i]]++;
    for (size_t i = 0; i < o.size() / 2; ++i) {
        const long v = (long)o[i];
        lo_bad[(i % 64) / 16] += v % 65536; hi_bad[(i % 64) / 16] += v / 65536;
    }
    printf("variant %d: wrong LOW results by lane quarter [%ld %ld %ld %ld], wrong HIGH results [%ld %ld %ld %ld] of %ld per quarter\n",
           VAR, lo_bad[0], lo_bad[1], lo_bad[2], lo_bad[3], hi_bad[0], hi_bad[1], hi_bad[2], hi_bad[3], (long)blocks * 64 * reps);
    printf("           a wrong LOW value is: the raw LDS value (multiplication lost) in %ld lanes, the LOW source times/plus the OTHER half of the second source (op_sel not applied) in %ld, something else in %ld\n", cls[1], cls[2], cls[3]);
}
int main() {
    const int blocks = 1024, reps = 64;
    std::vector<float> in(256);
    for (int i = 0; i < 256; ++i) in[i] = 1.0f + 0.37f * i;
    float *din, *dout;
    (void)hipMalloc(&din, 1024); (void)hipMalloc(&dout, 2 * blocks * 256 * 4);
    (void)hipMemcpy(din, in.data(), 1024, hipMemcpyHostToDevice);
    run<0>(din, dout, blocks, reps); run<1>(din, dout, blocks, reps); run<2>(din, dout, blocks, reps);
    run<3>(din, dout, blocks, reps); run<4>(din, dout, blocks, reps);
    run<5>(din, dout, blocks, reps); run<6>(din, dout, blocks, reps); run<7>(din, dout, blocks, reps); run<8>(din, dout, blocks, reps);
    run<9>(din, dout, blocks, reps); run<10>(din, dout, blocks, reps); run<11>(din, dout, blocks, reps);
    run<12>(din, dout, blocks, reps); run<13>(din, dout, blocks, reps);
    return 0;
}
