// Issue cost of the float64 vector instructions the mixture kernels are made of (MI355X, wave64): N independent chains per lane of ONE
// instruction, every SIMD of the chip busy with W waves -> cycles per wave instruction = W x cycles / (instructions per wave).
// Answers: is v_fma_f64 a 4-cycle instruction like v_fma_f32?  what do v_ldexp_f64 / v_rndne_f64 / v_cvt_i32_f64 / v_frexp / v_rcp_f64 cost?
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/f64_rates.hip -o scripts/probe/f64_rates && scripts/probe/f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define ITER 512

#define KERNEL(NAME, DECL, BODY)                                                                  \
    __global__ void __launch_bounds__(256) NAME(double* out, long long* clk, double seed) {      \
        DECL;                                                                                     \
        const long long t0 = __builtin_readcyclecounter();                                       \
        for (int it = 0; it < ITER; ++it) {                                                       \
            _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) { BODY; }                         \
        }                                                                                         \
        const long long t1 = __builtin_readcyclecounter();                                       \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;              \
        if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                          \
    }
#define F3 float f0 = (float)seed; float f1 = 1.0001f; float f2 = 0.5f; float g0 = f0, g1 = f0 + 1, g2 = f0 + 2, g3 = f0 + 3, g4 = f0 + 4, g5 = f0 + 5, g6 = f0 + 6, g7 = f0 + 7
#define D8 double a0 = seed; double a1 = seed + 1; double a2 = seed + 2; double a3 = seed + 3; double a4 = seed + 4; double a5 = seed + 5; double a6 = seed + 6; double a7 = seed + 7; const double b = seed * 0.5; const double c = seed * 0.25
#define EACH(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

#define OP_FMA64(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_ADD64(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MUL64(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_LDEXP64(x) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(ei));
#define OP_RNDNE64(x) asm volatile("v_rndne_f64 %0, %0" : "+v"(x));
#define OP_RCP64(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
#define OP_FREXPM64(x) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(x));
#define OP_CVTI(x) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ti) : "v"(x)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x) : "v"(ti));
#define OP_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f0) : "v"(f1));
#define OP_FMA32(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
#define OP_EXP32(x) asm volatile("v_exp_f32 %0, %0" : "+v"(f0));
#define OP_FMA32I(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(g##x) : "v"(f1), "v"(f2));
#define OP_EXP32I(x) asm volatile("v_exp_f32 %0, %0" : "+v"(g##x));
#define OP_RCP32I(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(g##x));
#define OP_CND32I(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(g##x) : "v"(f1));
#define OP_PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define EACHI(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define OP_CMP64(x) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define OP_MAX64(x) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));

KERNEL(k_fma64, D8, EACH(OP_FMA64))
KERNEL(k_add64, D8, EACH(OP_ADD64))
KERNEL(k_mul64, D8, EACH(OP_MUL64))
KERNEL(k_ldexp64, D8; int ei = (int)seed & 1, EACH(OP_LDEXP64))
KERNEL(k_rndne64, D8, EACH(OP_RNDNE64))
KERNEL(k_rcp64, D8, EACH(OP_RCP64))
KERNEL(k_frexpm64, D8, EACH(OP_FREXPM64))
KERNEL(k_cvt_i32_f64_pair, D8; int ti = 0, EACH(OP_CVTI))
KERNEL(k_cndmask, D8; F3, EACH(OP_CNDMASK))
KERNEL(k_fma32, D8; F3, EACH(OP_FMA32))
KERNEL(k_exp32, D8; F3, EACH(OP_EXP32))
KERNEL(k_fma32_indep, D8; F3, EACHI(OP_FMA32I); a0 += g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7)
KERNEL(k_exp32_indep, D8; F3, EACHI(OP_EXP32I); a0 += g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7)
KERNEL(k_rcp32_indep, D8; F3, EACHI(OP_RCP32I); a0 += g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7)
KERNEL(k_cnd32_indep, D8; F3, EACHI(OP_CND32I); a0 += g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7)
KERNEL(k_pkfma32, D8, EACH(OP_PKFMA))
KERNEL(k_cmp64, D8, EACH(OP_CMP64))
KERNEL(k_max64, D8, EACH(OP_MAX64))

template <typename K> void run(const char* name, K k, int per_body, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;                 // 256 CUs x (4 waves per block = one per SIMD) x waves_per_simd
    double* out; long long* clk;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&clk, sizeof(long long) * blocks);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, clk, 1.0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, clk, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), clk, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    const double insts = (double)ITER * (REP / 8) * 8 * per_body;     // wave instructions per wave
    // s_memtime / readcyclecounter ticks at a constant 100 MHz on gfx9: use wall time and the instruction count instead
    const double inst_per_s_per_simd = insts * waves_per_simd / (ms * 1e-3);
    printf("%-22s waves/SIMD %d  %8.3f ms  %6.2f G wave-inst/s per SIMD  -> %5.2f cycles per instruction at 2.4 GHz (%5.2f at 1.6)\n", name, waves_per_simd, ms,
           inst_per_s_per_simd / 1e9, 2.4e9 / inst_per_s_per_simd, 1.6e9 / inst_per_s_per_simd);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int w : {1, 4}) {
        run("v_fma_f64", k_fma64, 1, w);
        run("v_add_f64", k_add64, 1, w);
        run("v_mul_f64", k_mul64, 1, w);
        run("v_max_f64", k_max64, 1, w);
        run("v_ldexp_f64", k_ldexp64, 1, w);
        run("v_rndne_f64", k_rndne64, 1, w);
        run("v_frexp_mant_f64", k_frexpm64, 1, w);
        run("v_rcp_f64", k_rcp64, 1, w);
        run("v_cvt_i32_f64+back", k_cvt_i32_f64_pair, 2, w);
        run("v_cmp_gt_f64", k_cmp64, 1, w);
        run("v_cndmask_b32", k_cndmask, 1, w);
        run("v_fma_f32", k_fma32, 1, w);
        run("v_exp_f32", k_exp32, 1, w);
        run("v_fma_f32 indep", k_fma32_indep, 1, w);
        run("v_exp_f32 indep", k_exp32_indep, 1, w);
        run("v_rcp_f32 indep", k_rcp32_indep, 1, w);
        run("v_cndmask_b32 indep", k_cnd32_indep, 1, w);
        run("v_pk_fma_f32 (2 lanes)", k_pkfma32, 1, w);
    }
    return 0;
}
