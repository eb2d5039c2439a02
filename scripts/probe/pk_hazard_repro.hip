// Attempt at an ISOLATED reproducer of DESIGN.md 3.9 (round 4).  What the bisect of round 4 established inside the real kernel
// (scripts/probe/hazard/build_variants.py): with packed f32 enabled, the wrong log-dets come from ONE instruction encoding,
//     v_pk_add_f32 v[a:b], v[c:d], v[e:f] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]          (a = c - f,  b = d - e)
// -- every other packed instruction may stay; wait states before / after it change nothing; the same values from the commuted operands
// (crossing on src0), from v_pk_fma_f32, from a straight v_pk_add_f32 between two v_swap_b32, or from two v_add_f32 are all correct.
// This probe issues that instruction with the kernel's neighbourhood in VICTIM waves and checks its low result (the one that was wrong in the
// kernel, lanes 48..63) against plain v_sub_f32, while the other wave of each SIMD (512-thread workgroups: waves w and w + 4 share a SIMD)
// runs an AGGRESSOR instruction mix: 0 the same code, 1 f16 MFMAs fed from LDS, 2 transcendentals, 3 other packed f32 with different
// op_sel / neg patterns, 4 LDS traffic, 5 everything in turn.
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/pk_hazard_repro.hip -o scripts/probe/pk_hazard_repro && scripts/probe/pk_hazard_repro [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__device__ __forceinline__ unsigned victim(float x0, float x1, float x2) {
    float got, ref;
    asm volatile(
        "s_mov_b32 s40, 0x3f317218\n\ts_mov_b32 s41, 0x3f317218\n\tv_mov_b32 v70, 0x3f317218\n\tv_mov_b32 v71, 1.0\n\t"
        "v_log_f32 v64, %2\n\tv_mov_b32 v65, %3\n\tv_log_f32 v67, %4\n\t"
        "v_pk_mul_f32 v[64:65], v[64:65], v[70:71]\n\ts_nop 0\n\tv_log_f32 v66, v65\n\ts_nop 0\n\t"
        "v_pk_mul_f32 v[66:67], v[66:67], s[40:41] op_sel_hi:[1,0]\n\t"
        "s_mov_b32 s42, 0\n\ts_cmp_lt_i32 s42, 3\n\t"
        "v_pk_add_f32 v[76:77], v[64:65], v[66:67] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "s_nop 0\n\tv_mov_b32 v77, v67\n\t"
        "v_pk_add_f32 v[72:73], v[76:77], v[66:67] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "s_nop 7\n\tv_sub_f32 v78, v64, v67\n\tv_sub_f32 v78, v78, v66\n\t"
        "v_mov_b32 %0, v72\n\tv_mov_b32 %1, v78\n\t"
        : "=v"(got), "=v"(ref) : "v"(x0), "v"(x1), "v"(x2)
        : "v64", "v65", "v66", "v67", "v70", "v71", "v72", "v73", "v76", "v77", "v78", "s40", "s41", "s42", "scc");
    return got != ref;
}

template <int AGG> __global__ void __launch_bounds__(512) probe(const float* in, unsigned* bad, float* sink, int iters) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 48 * 1024 / 16; i += 512) reinterpret_cast<f32x4*>(lds)[i] = f32x4{(float)(i & 7), 1.f, 2.f, 3.f};
    __syncthreads();
    float x0 = in[(blockIdx.x * 512 + tid) & 4095] + 1.5f, x1 = x0 * 1.25f + 0.5f, x2 = x0 * 0.75f + 2.0f;
    f32x4 acc[6] = {};
    float t0 = x0, t1 = x1, t2 = x2, t3 = 1.f;
    unsigned miss = 0;
    const bool is_victim = AGG == 0 || wave < 4;
    for (int it = 0; it < iters; ++it) {
        if (is_victim) {
            miss += victim(x0, x1, x2);
            x0 += 0.001f; x1 += 0.002f; x2 += 0.003f;
            continue;
        }
        const int what = AGG == 5 ? 1 + (it & 3) : AGG;
        if (what == 1) {
            const f16x8 a = *reinterpret_cast<const f16x8*>(lds + (it & 31) * 1024 + lane * 16);
            const f16x8 b = *reinterpret_cast<const f16x8*>(lds + 32768 + (it & 7) * 1024 + lane * 16);
#pragma unroll
            for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
        } else if (what == 2) {
            asm volatile("v_exp_f32 %0, %0\n\tv_log_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_log_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_rcp_f32 %2, %2" : "+v"(t0), "+v"(t1), "+v"(t2));
        } else if (what == 3) {
            asm volatile(
                "v_mov_b32 v80, %0\n\tv_mov_b32 v81, %1\n\tv_mov_b32 v82, %2\n\tv_mov_b32 v83, %3\n\t"
                "v_pk_mul_f32 v[84:85], v[80:81], v[82:83]\n\t"
                "v_pk_add_f32 v[84:85], v[84:85], v[82:83] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_fma_f32 v[86:87], v[80:81], v[82:83], v[84:85] op_sel_hi:[1,0,1]\n\t"
                "v_pk_add_f32 v[86:87], v[86:87], v[80:81] op_sel:[1,0] op_sel_hi:[0,1]\n\t"
                "v_pk_mul_f32 v[84:85], v[86:87], v[82:83] op_sel_hi:[0,1]\n\t"
                "v_mov_b32 %0, v84\n\tv_mov_b32 %1, v85\n\t"
                : "+v"(t0), "+v"(t1) : "v"(t2), "v"(t3) : "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87");
        } else if (what == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((it * 64 + lane) & 2047) * 16);
            *reinterpret_cast<f32x4*>(lds + 40960 + ((it + wave) & 7) * 1024 + lane * 16) = v;
            t0 += v[0];
        }
    }
    float s = t0 + t1 + t2;
    for (int t = 0; t < 6; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (s == 12345.678f) sink[0] = s;
    if (miss) atomicAdd(&bad[lane >> 4], miss);
}

template <int AGG> static void run(const float* in, unsigned* bad, float* sink, int iters) {
    (void)hipMemset(bad, 0, 16);
    (void)hipFuncSetAttribute((const void*)probe<AGG>, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
    hipLaunchKernelGGL(probe<AGG>, dim3(256 * 3 * 4), dim3(512), 48 * 1024, 0, in, bad, sink, iters);
    unsigned b[4];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(b, bad, 16, hipMemcpyDeviceToHost);
    static const char* names[] = {"the same code", "f16 MFMAs fed from LDS", "transcendentals", "other packed f32 encodings", "LDS traffic", "all in turn"};
    printf("partner wave runs %-28s mismatches per 16-lane quarter of the victim waves: %u %u %u %u  (%s)\n", names[AGG], b[0], b[1], b[2], b[3],
           hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* in; unsigned* bad; float* sink;
    (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&bad, 16); (void)hipMalloc(&sink, 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)(i % 977) * 0.01f;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(in, bad, sink, iters); run<1>(in, bad, sink, iters); run<2>(in, bad, sink, iters);
    run<3>(in, bad, sink, iters); run<4>(in, bad, sink, iters); run<5>(in, bad, sink, iters);
    return 0;
}
