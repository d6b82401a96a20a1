// Hardware probe: absolute error of the transcendental units the fused kernels' FAST embedding uses
// (v_sin/v_cos in revolutions, v_exp, v_rcp, v_rsq) against double precision, and of the angle-
// doubling chain that produces the 7 octaves of the cutoff positional embedding from one sin/cos.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void probe(const float* v, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = v[i];
    const float rev = x * 0.15915494309189535f;
    float s = __builtin_amdgcn_sinf(rev), c = __builtin_amdgcn_cosf(rev);
    out[i * 8 + 0] = s; out[i * 8 + 1] = c;
    for (int f = 0; f < 6; ++f) { const float t = s + s; const float sn = t * c, cn = fmaf(-t, s, 1.0f); s = sn; c = cn; }
    out[i * 8 + 2] = s; out[i * 8 + 3] = c;                 // sin/cos(64 x) by doubling from the hw pair
    float s2, c2; sincosf(x, &s2, &c2);
    for (int f = 0; f < 6; ++f) { const float t = 2.0f * s2 * c2; c2 = (c2 - s2) * (c2 + s2); s2 = t; }
    out[i * 8 + 4] = s2; out[i * 8 + 5] = c2;               // the same from libm's sincosf (exact-mode kernels)
    out[i * 8 + 6] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(79.6f * 1.4426950408889634f * (x - 0.5f)));
    out[i * 8 + 7] = x * x * __builtin_amdgcn_rsqf(fmaxf(x * x, 1e-24f));
}
int main() {
    const int n = 1 << 20;
    std::vector<float> v(n), o(n * 8);
    for (int i = 0; i < n; ++i) v[i] = 6.0f * (i + 0.5f) / n;
    float *dv, *dout;
    hipMalloc(&dv, n * 4); hipMalloc(&dout, n * 32);
    hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<n / 256, 256>>>(dv, dout, n);
    hipMemcpy(o.data(), dout, n * 32, hipMemcpyDeviceToHost);
    double e[8] = {0};
    for (int i = 0; i < n; ++i) {
        const double x = v[i];
        const double ref[8] = {sin(x), cos(x), sin(64 * x), cos(64 * x), sin(64 * x), cos(64 * x),
                               1.0 / (1.0 + exp(79.6 * (x - 0.5))), x};
        for (int k = 0; k < 8; ++k) e[k] = fmax(e[k], fabs(o[i * 8 + k] - ref[k]));
    }
    printf("max abs err over v in (0,6): v_sin %.3e v_cos %.3e | sin64 (hw, doubled) %.3e cos64 %.3e | sin64 (libm, doubled) %.3e cos64 %.3e | cutoff weight %.3e | |q| via rsq %.3e\n",
           e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
    return 0;
}
