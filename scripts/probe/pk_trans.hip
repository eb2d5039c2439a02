// Does a packed-f32 instruction see the result of a transcendental instruction issued a few instructions earlier?
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/pk_trans.hip -o scripts/probe/pk_trans && scripts/probe/pk_trans
// Per iteration and lane: three back-to-back v_log_f32 (quarter rate), one VALU, then a consumer of the third result -- either
// v_pk_mul_f32 (mode 0) or v_mul_f32 (mode 1) -- compared with the same product taken 16+ wait states later.  Mismatches are counted per
// 16-lane quarter of the wave.  Several waves per SIMD keep the transcendental unit contended.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE> __global__ void __launch_bounds__(256) probe(const float* in, unsigned* bad, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float a = in[tid & 4095] + 1.5f, b = a * 1.25f, c = a * 0.75f + 2.0f;
    unsigned miss = 0;
    for (int it = 0; it < iters; ++it) {
        float early, late;
        // the instruction sequence of the plain mixture path of cond_gf_split_kernel as hipcc (ROCm 7.2) emitted it with packed f32 enabled:
        //   log, mov, mul, log, pk_mul (in place), nop, log (of the pk_mul's high half), nop, pk_mul (in place, SGPR-pair scale)
        if (MODE == 0) {
            asm volatile(
                "s_mov_b32 s40, 0x3f317218\n\t"
                "s_mov_b32 s41, 0x3f317218\n\t"
                "v_mov_b32 v20, 0x3f317218\n\t"
                "v_mov_b32 v21, 0x40400000\n\t"
                "s_nop 4\n\t"
                "v_log_f32 v10, %2\n\t"
                "v_mov_b32 v11, %3\n\t"
                "v_mul_f32 v30, %3, %4\n\t"
                "v_log_f32 v13, v30\n\t"
                "v_pk_mul_f32 v[10:11], v[10:11], v[20:21]\n\t"
                "s_nop 0\n\t"
                "v_mov_b32 v31, v11\n\t"                     // keep the argument of the next log for the late recomputation (extra VALU: see MODE 2 without it)
                "v_log_f32 v12, v11\n\t"
                "s_nop 0\n\t"
                "v_pk_mul_f32 v[12:13], v[12:13], s[40:41] op_sel_hi:[1,0]\n\t"
                "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                "v_log_f32 v14, v31\n\t"
                "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                "v_mul_f32 v15, v14, v20\n\t"
                "v_mov_b32 %0, v12\n\t"
                "v_mov_b32 %1, v15\n\t"
                : "=v"(early), "=v"(late) : "v"(a), "v"(b), "v"(c)
                : "v10", "v11", "v12", "v13", "v14", "v15", "v20", "v21", "v30", "v31", "s40", "s41");
        } else {
            asm volatile(
                "v_mov_b32 v20, 0x3f317218\n\t"
                "v_mov_b32 v21, 0x40400000\n\t"
                "s_nop 4\n\t"
                "v_log_f32 v10, %2\n\t"
                "v_mov_b32 v11, %3\n\t"
                "v_mul_f32 v30, %3, %4\n\t"
                "v_log_f32 v13, v30\n\t"
                "v_mul_f32 v10, v10, v20\n\t"
                "v_mul_f32 v11, v11, v21\n\t"
                "s_nop 0\n\t"
                "v_mov_b32 v31, v11\n\t"
                "v_log_f32 v12, v11\n\t"
                "s_nop 0\n\t"
                "v_mul_f32 v12, v12, v20\n\t"
                "v_mul_f32 v13, v13, v20\n\t"
                "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                "v_log_f32 v14, v31\n\t"
                "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                "v_mul_f32 v15, v14, v20\n\t"
                "v_mov_b32 %0, v12\n\t"
                "v_mov_b32 %1, v15\n\t"
                : "=v"(early), "=v"(late) : "v"(a), "v"(b), "v"(c)
                : "v10", "v11", "v12", "v13", "v14", "v15", "v20", "v21", "v30", "v31");
        }
        miss += (early != late);
        a += 0.001f; b += 0.002f; c += 0.003f;
    }
    if (miss) atomicAdd(&bad[(threadIdx.x & 63) >> 4], miss);
}

// the same packed sequence in waves 4..7 of a 512-thread workgroup while waves 0..3 (their SIMD partners) issue bf16 MFMAs back to back
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__global__ void __launch_bounds__(512) probe_mixed(const float* in, unsigned* bad, float* sink, int iters) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        bf16x8_t a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
        f32x4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters * 2; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
        if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[tid] = c0[0];
        return;
    }
    float a = in[tid & 4095] + 1.5f, b = a * 1.25f, c = a * 0.75f + 2.0f;
    unsigned miss = 0;
    for (int it = 0; it < iters; ++it) {
        float early, late;
        asm volatile(
            "s_mov_b32 s40, 0x3f317218\n\t"
            "s_mov_b32 s41, 0x3f317218\n\t"
            "v_mov_b32 v20, 0x3f317218\n\t"
            "v_mov_b32 v21, 0x40400000\n\t"
            "s_nop 4\n\t"
            "v_log_f32 v10, %2\n\t"
            "v_mov_b32 v11, %3\n\t"
            "v_mul_f32 v30, %3, %4\n\t"
            "v_log_f32 v13, v30\n\t"
            "v_pk_mul_f32 v[10:11], v[10:11], v[20:21]\n\t"
            "s_nop 0\n\t"
            "v_mov_b32 v31, v11\n\t"
            "v_log_f32 v12, v11\n\t"
            "s_nop 0\n\t"
            "v_pk_mul_f32 v[12:13], v[12:13], s[40:41] op_sel_hi:[1,0]\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
            "v_log_f32 v14, v31\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
            "v_mul_f32 v15, v14, v20\n\t"
            "v_mov_b32 %0, v12\n\t"
            "v_mov_b32 %1, v15\n\t"
            : "=v"(early), "=v"(late) : "v"(a), "v"(b), "v"(c)
            : "v10", "v11", "v12", "v13", "v14", "v15", "v20", "v21", "v30", "v31", "s40", "s41");
        miss += (early != late);
        a += 0.001f; b += 0.002f; c += 0.003f;
    }
    if (miss) atomicAdd(&bad[(threadIdx.x & 63) >> 4], miss);
}

int main() {
    float* in; unsigned* bad;
    hipMalloc(&in, 4096 * 4); hipMalloc(&bad, 16);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 0.5f + (i % 97) * 0.01f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(bad, 0, 16);
        const int iters = 2000, blocks = 4096;
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, in, bad, iters);
        else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, in, bad, iters);
        hipDeviceSynchronize();
        unsigned hb[4]; hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
        printf("%s consumer one wait state after v_log_f32: stale reads per lane quarter [0-15 16-31 32-47 48-63] = %u %u %u %u of %.3g lane-iterations each\n",
               mode == 0 ? "v_pk_mul_f32" : "v_mul_f32   ", hb[0], hb[1], hb[2], hb[3], (double)blocks * 64 * iters);
    }
    {
        float* sink; hipMalloc(&sink, 4096 * 512 * 4);
        hipMemset(bad, 0, 16);
        hipLaunchKernelGGL(probe_mixed, dim3(2048), dim3(512), 0, 0, in, bad, sink, 2000);
        hipDeviceSynchronize();
        unsigned hb[4]; hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
        printf("v_pk_mul_f32 consumer beside a partner wave issuing bf16 MFMAs: stale reads per lane quarter = %u %u %u %u\n", hb[0], hb[1], hb[2], hb[3]);
    }
    return 0;
}
