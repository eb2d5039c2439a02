// MFMA issue-rate probe for gfx950: v_mfma_f32_32x32x2_f32 with 1 / 2 / 4 independent accumulators per wave, W waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 mfma.hip -o mfma ; run: ./mfma
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC> void run(int blocks_per_cu, float* out) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<grid, 256>>>(out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<grid, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 16 * NACC * 4096.0;
    printf("acc/wave %d  waves/SIMD %d : %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flops / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 4; w *= 2) { run<1>(w, out); run<2>(w, out); run<4>(w, out); }
    return 0;
}
