// Round-3 re-measurement of the co-issue question: how many independent VALU instructions hide behind one bf16 MFMA on a
// gfx950 SIMD, for both tile shapes, from the same wave and from a partner wave?  Unlike bf16co2.hip every instruction is its own
// `asm volatile` statement (program order = issue order, nothing for the scheduler to move), fillers are F independent chains per MFMA slot
// drawn from 12 registers (a chain is touched again only after >= 12 other fillers), and the MFMAs rotate over 6 accumulators.
//   mode 0: MFMA + F fillers per MFMA, same wave, 1 wave / SIMD        mode 1: the same stream, 2 waves / SIMD
//   mode 2: fillers only (F per slot), 1 wave / SIMD                   mode 3: waves 0-3 MFMA only | waves 4-7 fillers only (F per slot)
//   mode 4: fillers only, 2 waves / SIMD
// Output: shader cycles per MFMA slot (s_memtime); with two waves per SIMD the SLOWER of waves 0 and 4 (the older wave wins the arbitration
// and finishes first: its time alone says nothing about the SIMD's throughput).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

enum { FMA = 0, EXP = 1, MIX = 2 };       // MIX: the flow arithmetic's ratio, 1 transcendental in 4

template <int FT> __device__ __forceinline__ void filler(float& x, int idx) {
    if (FT == FMA || (FT == MIX && (idx & 3) != 3)) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(0.9999f), "v"(0.25f));
    else asm volatile("v_exp_f32 %0, %0" : "+v"(x));
}

template <int SHAPE, int F, int FT, bool MF, bool VA>
__device__ __forceinline__ float body(int n, float seed) {
    f32x4 a[6]; f32x16 c[4];
    for (int i = 0; i < 6; ++i) a[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    bf16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(seed * 1e-3f + i); B[i] = (__bf16)0.5f; }
    float x[12];
    for (int i = 0; i < 12; ++i) x[i] = seed * 1e-4f - 0.1f * i;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            if (MF) {
                if (SHAPE == 16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a[u % 6]) : "v"(A), "v"(B));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[u % 4]) : "v"(A), "v"(B));
            }
            if (VA) {
#pragma unroll
                for (int f = 0; f < F; ++f) filler<FT>(x[(u * F + f) % 12], u * F + f);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float r = 0.f;
    for (int i = 0; i < 6; ++i) r += a[i][0];
    for (int i = 0; i < 4; ++i) r += c[i][3];
    for (int i = 0; i < 12; ++i) r += x[i];
    return r;
}

template <int SHAPE, int F, int FT>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int mode, int n) {
    const int wave = threadIdx.x >> 6;
    float r;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0 || mode == 1) r = body<SHAPE, F, FT, true, true>(n, (float)threadIdx.x);
    else if (mode == 2 || mode == 4) r = body<SHAPE, F, FT, false, true>(n, (float)threadIdx.x);
    else if (wave < 4) r = body<SHAPE, F, FT, true, false>(n, (float)threadIdx.x);
    else r = body<SHAPE, F, FT, false, true>(n, (float)threadIdx.x);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int SHAPE, int F, int FT> void run(float* out, long long* cyc) {
    const int n = 4000;
    const char* ft = FT == FMA ? "fma" : FT == EXP ? "exp" : "mix";
    printf("%dx%d  F=%d %s :", SHAPE, SHAPE, F, ft);
    for (int mode = 0; mode < 5; ++mode) {
        const int threads = (mode == 1 || mode == 3 || mode == 4) ? 512 : 256;
        k<SHAPE, F, FT><<<256, threads>>>(out, cyc, mode, 10); (void)hipDeviceSynchronize();
        k<SHAPE, F, FT><<<256, threads>>>(out, cyc, mode, n); (void)hipDeviceSynchronize();
        long long h[8]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double per = (double)h[0] / (12.0 * n);
        if (mode == 1 || mode == 4) { const double p4 = (double)h[4] / (12.0 * n); per = per > p4 ? per : p4; }
        if (mode == 0) printf("  same wave %6.1f", per);
        if (mode == 1) printf("  | 2 waves/SIMD same stream %6.1f (slower wave; %5.1f per MFMA of the SIMD)", per, per / 2);
        if (mode == 4) printf("  | 2 waves/SIMD fillers only %6.1f (slower wave)", per);
        if (mode == 2) printf("  | fillers alone %6.1f", per);
        if (mode == 3) printf("  | split roles: MFMA wave %6.1f, filler wave %6.1f", per, (double)h[4] / (12.0 * n));
    }
    printf("  cyc/slot\n");
}

template <int SHAPE, int FT> void sweep(float* out, long long* cyc) {
    run<SHAPE, 0, FT>(out, cyc); run<SHAPE, 1, FT>(out, cyc); run<SHAPE, 2, FT>(out, cyc); run<SHAPE, 3, FT>(out, cyc);
    run<SHAPE, 4, FT>(out, cyc); run<SHAPE, 5, FT>(out, cyc); run<SHAPE, 6, FT>(out, cyc); run<SHAPE, 8, FT>(out, cyc);
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
    sweep<16, FMA>(out, cyc); sweep<16, MIX>(out, cyc); sweep<16, EXP>(out, cyc);
    sweep<32, FMA>(out, cyc); sweep<32, MIX>(out, cyc); sweep<32, EXP>(out, cyc);
    return 0;
}
