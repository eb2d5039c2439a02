// Do MFMA work of one wave and VALU work of another wave on the same SIMD overlap?  Workgroup = 8 waves: waves 0-3 run an MFMA loop, waves 4-7
// a VALU loop (mode 3), or only one of the two kinds does work (modes 1, 2).  If they overlap, t(3) ~ max(t(1), t(2)); if they serialise, t(3) ~ sum.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void __launch_bounds__(512) k(float* out, int mode, int n_mfma, int n_valu, int trans) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
            const float a = threadIdx.x, b = 2.f;
            for (int i = 0; i < n_mfma; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a2, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2];
        }
        if (mode == 4) {      // ONE wave interleaves: per iteration 3 MFMAs (96 matrix-pipe cycles) + 16 independent VALU fmas (64 VALU cycles)
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
            const float a = threadIdx.x, b = 2.f;
            float x0 = threadIdx.x * 1e-3f, x1 = 0.5f, x2 = 0.25f, x3 = 0.125f;
            for (int i = 0; i < n_mfma; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a0, 0, 0, 0);
                x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 0.9999f, 0.25f); x2 = fmaf(x2, 1.0002f, 0.125f); x3 = fmaf(x3, 0.9998f, 0.75f);
                x0 = fmaf(x0, x1, 0.5f);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a1, 0, 0, 0);
                x1 = fmaf(x1, x2, 0.25f); x2 = fmaf(x2, x3, 0.125f); x3 = fmaf(x3, x0, 0.75f);
                x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 0.9999f, 0.25f);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a2, 0, 0, 0);
                x2 = fmaf(x2, 1.0002f, 0.125f); x3 = fmaf(x3, 0.9998f, 0.75f);
                x0 = fmaf(x0, x1, 0.5f); x1 = fmaf(x1, x2, 0.25f); x2 = fmaf(x2, x3, 0.125f); x3 = fmaf(x3, x0, 0.75f);
            }
            r = a0[0] + a1[1] + a2[2] + x0 + x1 + x2 + x3;
        }
    } else if (mode & 2) {
        float x0 = threadIdx.x * 1e-3f, x1 = 0.5f, x2 = 0.25f, x3 = 0.125f;
        for (int i = 0; i < n_valu; ++i) {
            if (trans) { x0 = __expf(x0) * 0.3f; x1 = __expf(x1) * 0.3f; x2 = __builtin_amdgcn_rcpf(x2 + 2.f); x3 = __builtin_amdgcn_rcpf(x3 + 2.f); }
            x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 0.9999f, 0.25f); x2 = fmaf(x2, 1.0002f, 0.125f); x3 = fmaf(x3, 0.9998f, 0.75f);
            x0 = fmaf(x0, x1, 0.5f); x1 = fmaf(x1, x2, 0.25f); x2 = fmaf(x2, x3, 0.125f); x3 = fmaf(x3, x0, 0.75f);
        }
        r = x0 + x1 + x2 + x3;
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
float run(float* out, int mode, int nm, int nv, int trans) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<256, 512>>>(out, mode, 10, 10, trans); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<<<256, 512>>>(out, mode, nm, nv, trans); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    for (int trans = 0; trans < 2; ++trans) {
        const int nm = 20000, nv = trans ? 12000 : 40000;
        const float t1 = run(out, 1, nm, nv, trans), t2 = run(out, 2, nm, nv, trans), t3 = run(out, 3, nm, nv, trans);
        if (!trans) { const float t4 = run(out, 4, nm, nv, trans); printf("one wave, MFMAs interleaved with 16 fma per 3 MFMA: %.3f ms (MFMA-only %.3f)\n", t4, t1); }
        printf("%s VALU loop: MFMA only %.3f ms, VALU only %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n", trans ? "transcendental" : "fma", t1, t2, t3,
               t1 + t2, t1 > t2 ? t1 : t2);
    }
    return 0;
}
