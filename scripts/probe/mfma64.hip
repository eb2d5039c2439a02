// float64 MFMA issue-rate probe for gfx950: v_mfma_f64_16x16x4_f64 with 1 / 2 / 4 independent accumulators per wave, W waves per SIMD.
// (The local microarchitecture guide lists no f64 MFMA peak; SURVEY 8d asks for a measured one.)   ./mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <int NACC>
__global__ void __launch_bounds__(256) k(double* out, int iters, double a0, double b0) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.0;
    double a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC> void run(int blocks_per_cu, double* out) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NACC><<<grid, 256>>>(out, 10, 1.0, 2.0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NACC><<<grid, 256>>>(out, iters, 1.0, 2.0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 16 * NACC * (16.0 * 16 * 4 * 2);
    printf("acc/wave %d  waves/SIMD %d : %.3f ms  %.1f TFLOP/s (f64)\n", NACC, blocks_per_cu, ms, flops / ms / 1e9);
}

int main() {
    double* out; (void)hipMalloc(&out, 256 * 8 * 256 * 8);
    for (int w = 1; w <= 4; w *= 2) { run<1>(w, out); run<2>(w, out); run<4>(w, out); }
    return 0;
}
