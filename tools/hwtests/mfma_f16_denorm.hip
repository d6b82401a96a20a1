// Hardware probe: does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL operands (A, B) or flush them?
// And the precision of v_sqrt_f32 / v_rsq_f32 (the compensated-fp16 mode's |q|).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void probe(float* out, const float* in) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    const _Float16 sub = (_Float16)3.0e-6f;            // subnormal in fp16 (min normal 6.1e-5)
    for (int j = 0; j < 8; ++j) { a[j] = (lane < 32 && j == 0) ? sub : (_Float16)0.f; b[j] = (lane < 32 && j == 0) ? (_Float16)1024.f : (_Float16)0.f; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);      // C[i][j] = sub * 1024 for all i, j
    f32x16 d = {0};
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, d, 0, 0, 0);      // subnormal on the B side
    if (lane == 0) { out[0] = c[0]; out[1] = d[0]; out[2] = (float)sub * 1024.f; }
    const float x = in[lane];
    out[8 + lane] = __builtin_amdgcn_sqrtf(x);
    out[72 + lane] = x * __builtin_amdgcn_rsqf(x);
}
int main() {
    float *d, *din, h[160], hin[64];
    for (int i = 0; i < 64; ++i) hin[i] = 0.013f + 0.41f * i * (1.f + 0.01f * i);
    (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&din, sizeof hin);
    (void)hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, din);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("fp16 subnormal 3e-6 x 1024: A-side %.6e  B-side %.6e  expected %.6e\n", h[0], h[1], h[2]);
    double es = 0, er = 0;
    for (int i = 0; i < 64; ++i) {
        const double r = sqrt((double)hin[i]);
        es = fmax(es, fabs(h[8 + i] - r) / r); er = fmax(er, fabs(h[72 + i] - r) / r);
    }
    printf("max relative error: v_sqrt_f32 %.3e   x*v_rsq_f32(x) %.3e  (1 ulp = 6e-8)\n", es, er);
    return 0;
}
