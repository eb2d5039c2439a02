// bf16 MFMA (v_mfma_f32_16x16x32_bf16) rate, and whether it overlaps with VALU work of another wave on the same SIMD (f32 MFMA does not).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
__global__ void __launch_bounds__(512) k(float* out, int mode, int n_mfma, int n_valu) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
            bf16x8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)2.f; }
            for (int i = 0; i < n_mfma; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a2, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2];
        }
    } else if (mode & 2) {
        float x0 = threadIdx.x * 1e-3f, x1 = 0.5f, x2 = 0.25f, x3 = 0.125f;
        for (int i = 0; i < n_valu; ++i) {
            x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 0.9999f, 0.25f); x2 = fmaf(x2, 1.0002f, 0.125f); x3 = fmaf(x3, 0.9998f, 0.75f);
            x0 = fmaf(x0, x1, 0.5f); x1 = fmaf(x1, x2, 0.25f); x2 = fmaf(x2, x3, 0.125f); x3 = fmaf(x3, x0, 0.75f);
        }
        r = x0 + x1 + x2 + x3;
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
float run(float* out, int mode, int nm, int nv) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<256, 512>>>(out, mode, 10, 10); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<<<256, 512>>>(out, mode, nm, nv); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int nm = 40000, nv = 40000;
    const float t1 = run(out, 1, nm, nv), t2 = run(out, 2, nm, nv), t3 = run(out, 3, nm, nv);
    const double flops = 256.0 * 4 * nm * 3 * (16.0 * 16 * 32 * 2);
    printf("bf16 MFMA only %.3f ms (%.0f TFLOP/s with 1 wave per SIMD), VALU only %.3f ms, both %.3f ms (sum %.3f, max %.3f)\n", t1, flops / t1 / 1e9, t2, t3,
           t1 + t2, t1 > t2 ? t1 : t2);
    return 0;
}
