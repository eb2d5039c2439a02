// Does bf16 MFMA (v_mfma_f32_16x16x32_bf16) work overlap with VALU work on one CDNA4 SIMD?
//   mode 0: MFMA only, 1 wave per SIMD (4 independent accumulators)             -> cycles per MFMA
//   mode 1: same wave, F independent VALU fillers (v_fma_f32 / v_exp_f32) pinned behind every MFMA with sched_group_barrier
//   mode 2: VALU only (the same fillers, no MFMA)
//   mode 3: 2 waves per SIMD: waves 0-3 MFMA only, waves 4-7 VALU only (both loops sized to the same stand-alone time)
// Prints wall time and per-iteration shader cycles (s_memtime of wave 0 / wave 4 of block 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int F, bool EXP, bool MF>
__device__ __forceinline__ float body(int n, float seed) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)2.f; }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed * 1e-3f + 0.1f * i;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MF) {
                if (u == 0) a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a0, 0, 0, 0);
                if (u == 1) a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a1, 0, 0, 0);
                if (u == 2) a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a2, 0, 0, 0);
                if (u == 3) a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, a3, 0, 0, 0);
            }
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int j = (u * F + f) & 7;
                if (EXP && (f & 1)) x[j] = __expf(x[j]) * 0.3f; else x[j] = fmaf(x[j], 0.9999f, 0.25f);
            }
            if (MF) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (F > 0) __builtin_amdgcn_sched_group_barrier(0x002, EXP ? F + F / 2 : F, 0);
        }
    }
    float r = a0[0] + a1[1] + a2[2] + a3[3];
    for (int i = 0; i < 8; ++i) r += x[i];
    return r;
}

template <int F, bool EXP>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int mode, int n_m, int n_v) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) r = body<0, EXP, true>(n_m, (float)threadIdx.x);
    else if (mode == 1) r = body<F, EXP, true>(n_m, (float)threadIdx.x);
    else if (mode == 2) r = body<F, EXP, false>(n_v, (float)threadIdx.x);
    else if (wave < 4) r = body<0, EXP, true>(n_m, (float)threadIdx.x);
    else r = body<F, EXP, false>(n_v, (float)threadIdx.x);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int F, bool EXP>
void run(float* out, long long* cyc, int mode, int threads, int nm, int nv, const char* what) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<F, EXP><<<256, threads>>>(out, cyc, mode, 10, 10); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<F, EXP><<<256, threads>>>(out, cyc, mode, nm, nv); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("F=%d %s %-34s %8.3f ms   wave0 %6.1f cyc/MFMA-slot", F, EXP ? "fma+exp" : "fma    ", what, ms, (double)h[0] / (4.0 * (mode == 2 ? nv : nm)));
    if (threads == 512) printf("   wave4 %6.1f cyc/slot", (double)h[4] / (4.0 * nv));
    printf("\n");
}

template <int F, bool EXP> void suite(float* out, long long* cyc) {
    const int n = 20000;
    run<F, EXP>(out, cyc, 0, 256, n, n, "MFMA only, 1 wave/SIMD");
    run<F, EXP>(out, cyc, 1, 256, n, n, "MFMA + F fillers each, same wave");
    run<F, EXP>(out, cyc, 2, 256, n, n, "fillers only, 1 wave/SIMD");
    run<F, EXP>(out, cyc, 3, 512, n, n, "MFMA waves 0-3 | VALU waves 4-7");
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
    suite<1, false>(out, cyc); suite<2, false>(out, cyc); suite<3, false>(out, cyc); suite<4, false>(out, cyc); suite<6, false>(out, cyc);
    suite<2, true>(out, cyc); suite<4, true>(out, cyc);
    return 0;
}
