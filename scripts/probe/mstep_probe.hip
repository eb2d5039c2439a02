// Stand-alone timing of the k-step block of cond_pp_kernels.hip (30 MFMAs 32x32x16 over five accumulators + 15 ds_read_b128 of the next
// k-step's fragments): cycles per MFMA with (0) the block as shipped, (1) no LDS reads, (2) reads into a second register set (no
// write-after-read on the MFMA's A operand), one wave per SIMD, all CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int MODE> __global__ void __launch_bounds__(256, 2) k(float* out, long long* cyc, int n) {
    extern __shared__ __align__(16) unsigned char smem[];
    for (int i = threadIdx.x; i < 61440 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[5]; bf16x8 A[3][5], N[3][5], hB[3];
    for (int t = 0; t < 5; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    for (int p = 0; p < 3; ++p) for (int j = 0; j < 8; ++j) hB[p][j] = (__bf16)(0.25f * (p + 1));
    const unsigned nx = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem + lane * 16;
    for (int p = 2; p >= 0; --p) for (int t = 0; t < 5; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(A[p][t]) : "v"(nx));
    for (int p = 2; p >= 0; --p) for (int t = 0; t < 5; ++t) N[p][t] = A[p][t];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        if (MODE == 0)
            asm volatile(
                "s_waitcnt lgkmcnt(10)\n\t"
                "s_nop 1\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a20], %[h0], %[c0]\n\t"
                "ds_read_b128 %[a20], %[nx] offset:2048\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a21], %[h0], %[c1]\n\t"
                "ds_read_b128 %[a21], %[nx] offset:14336\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a22], %[h0], %[c2]\n\t"
                "ds_read_b128 %[a22], %[nx] offset:26624\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a23], %[h0], %[c3]\n\t"
                "ds_read_b128 %[a23], %[nx] offset:38912\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a24], %[h0], %[c4]\n\t"
                "ds_read_b128 %[a24], %[nx] offset:51200\n\t"
                "s_waitcnt lgkmcnt(10)\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a10], %[h1], %[c0]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a11], %[h1], %[c1]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a12], %[h1], %[c2]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a13], %[h1], %[c3]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a14], %[h1], %[c4]\n\t"
                "s_waitcnt lgkmcnt(5)\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a00], %[h2], %[c0]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a01], %[h2], %[c1]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a02], %[h2], %[c2]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a03], %[h2], %[c3]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a04], %[h2], %[c4]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a10], %[h0], %[c0]\n\t"
                "ds_read_b128 %[a10], %[nx] offset:1024\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a11], %[h0], %[c1]\n\t"
                "ds_read_b128 %[a11], %[nx] offset:13312\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a12], %[h0], %[c2]\n\t"
                "ds_read_b128 %[a12], %[nx] offset:25600\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a13], %[h0], %[c3]\n\t"
                "ds_read_b128 %[a13], %[nx] offset:37888\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a14], %[h0], %[c4]\n\t"
                "ds_read_b128 %[a14], %[nx] offset:50176\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a00], %[h1], %[c0]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a01], %[h1], %[c1]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a02], %[h1], %[c2]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a03], %[h1], %[c3]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a04], %[h1], %[c4]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a00], %[h0], %[c0]\n\t"
                "ds_read_b128 %[a00], %[nx] offset:0\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a01], %[h0], %[c1]\n\t"
                "ds_read_b128 %[a01], %[nx] offset:12288\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a02], %[h0], %[c2]\n\t"
                "ds_read_b128 %[a02], %[nx] offset:24576\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a03], %[h0], %[c3]\n\t"
                "ds_read_b128 %[a03], %[nx] offset:36864\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a04], %[h0], %[c4]\n\t"
                "ds_read_b128 %[a04], %[nx] offset:49152"
                : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]),
              [a20] "+v"(A[2][0]), [a21] "+v"(A[2][1]), [a22] "+v"(A[2][2]), [a23] "+v"(A[2][3]), [a24] "+v"(A[2][4]), [a10] "+v"(A[1][0]), [a11] "+v"(A[1][1]), [a12] "+v"(A[1][2]), [a13] "+v"(A[1][3]), [a14] "+v"(A[1][4]), [a00] "+v"(A[0][0]), [a01] "+v"(A[0][1]), [a02] "+v"(A[0][2]), [a03] "+v"(A[0][3]), [a04] "+v"(A[0][4])
                : [h0] "v"(hB[0]), [h1] "v"(hB[1]), [h2] "v"(hB[2]), [nx] "v"(nx));
        else if (MODE == 1)
            asm volatile(
                "s_waitcnt lgkmcnt(10)\n\t"
                "s_nop 1\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a20], %[h0], %[c0]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a21], %[h0], %[c1]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a22], %[h0], %[c2]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a23], %[h0], %[c3]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a24], %[h0], %[c4]\n\t"
                "s_waitcnt lgkmcnt(10)\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a10], %[h1], %[c0]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a11], %[h1], %[c1]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a12], %[h1], %[c2]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a13], %[h1], %[c3]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a14], %[h1], %[c4]\n\t"
                "s_waitcnt lgkmcnt(5)\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a00], %[h2], %[c0]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a01], %[h2], %[c1]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a02], %[h2], %[c2]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a03], %[h2], %[c3]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a04], %[h2], %[c4]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a10], %[h0], %[c0]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a11], %[h0], %[c1]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a12], %[h0], %[c2]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a13], %[h0], %[c3]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a14], %[h0], %[c4]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a00], %[h1], %[c0]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a01], %[h1], %[c1]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a02], %[h1], %[c2]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a03], %[h1], %[c3]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a04], %[h1], %[c4]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a00], %[h0], %[c0]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a01], %[h0], %[c1]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a02], %[h0], %[c2]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a03], %[h0], %[c3]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a04], %[h0], %[c4]"
                : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]),
              [a20] "+v"(A[2][0]), [a21] "+v"(A[2][1]), [a22] "+v"(A[2][2]), [a23] "+v"(A[2][3]), [a24] "+v"(A[2][4]), [a10] "+v"(A[1][0]), [a11] "+v"(A[1][1]), [a12] "+v"(A[1][2]), [a13] "+v"(A[1][3]), [a14] "+v"(A[1][4]), [a00] "+v"(A[0][0]), [a01] "+v"(A[0][1]), [a02] "+v"(A[0][2]), [a03] "+v"(A[0][3]), [a04] "+v"(A[0][4])
                : [h0] "v"(hB[0]), [h1] "v"(hB[1]), [h2] "v"(hB[2]), [nx] "v"(nx));
        else
            asm volatile(
                "s_waitcnt lgkmcnt(10)\n\t"
                "s_nop 1\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a20], %[h0], %[c0]\n\t"
                "ds_read_b128 %[na20], %[nx] offset:2048\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a21], %[h0], %[c1]\n\t"
                "ds_read_b128 %[na21], %[nx] offset:14336\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a22], %[h0], %[c2]\n\t"
                "ds_read_b128 %[na22], %[nx] offset:26624\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a23], %[h0], %[c3]\n\t"
                "ds_read_b128 %[na23], %[nx] offset:38912\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a24], %[h0], %[c4]\n\t"
                "ds_read_b128 %[na24], %[nx] offset:51200\n\t"
                "s_waitcnt lgkmcnt(10)\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a10], %[h1], %[c0]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a11], %[h1], %[c1]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a12], %[h1], %[c2]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a13], %[h1], %[c3]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a14], %[h1], %[c4]\n\t"
                "s_waitcnt lgkmcnt(5)\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a00], %[h2], %[c0]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a01], %[h2], %[c1]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a02], %[h2], %[c2]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a03], %[h2], %[c3]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a04], %[h2], %[c4]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a10], %[h0], %[c0]\n\t"
                "ds_read_b128 %[na10], %[nx] offset:1024\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a11], %[h0], %[c1]\n\t"
                "ds_read_b128 %[na11], %[nx] offset:13312\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a12], %[h0], %[c2]\n\t"
                "ds_read_b128 %[na12], %[nx] offset:25600\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a13], %[h0], %[c3]\n\t"
                "ds_read_b128 %[na13], %[nx] offset:37888\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a14], %[h0], %[c4]\n\t"
                "ds_read_b128 %[na14], %[nx] offset:50176\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a00], %[h1], %[c0]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a01], %[h1], %[c1]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a02], %[h1], %[c2]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a03], %[h1], %[c3]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a04], %[h1], %[c4]\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c0], %[a00], %[h0], %[c0]\n\t"
                "ds_read_b128 %[na00], %[nx] offset:0\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c1], %[a01], %[h0], %[c1]\n\t"
                "ds_read_b128 %[na01], %[nx] offset:12288\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c2], %[a02], %[h0], %[c2]\n\t"
                "ds_read_b128 %[na02], %[nx] offset:24576\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c3], %[a03], %[h0], %[c3]\n\t"
                "ds_read_b128 %[na03], %[nx] offset:36864\n\t"
                "v_mfma_f32_32x32x16_bf16 %[c4], %[a04], %[h0], %[c4]\n\t"
                "ds_read_b128 %[na04], %[nx] offset:49152"
                : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]),
              [a20] "+v"(A[2][0]), [a21] "+v"(A[2][1]), [a22] "+v"(A[2][2]), [a23] "+v"(A[2][3]), [a24] "+v"(A[2][4]), [a10] "+v"(A[1][0]), [a11] "+v"(A[1][1]), [a12] "+v"(A[1][2]), [a13] "+v"(A[1][3]), [a14] "+v"(A[1][4]), [a00] "+v"(A[0][0]), [a01] "+v"(A[0][1]), [a02] "+v"(A[0][2]), [a03] "+v"(A[0][3]), [a04] "+v"(A[0][4]),
              [na20] "=&v"(N[2][0]), [na21] "=&v"(N[2][1]), [na22] "=&v"(N[2][2]), [na23] "=&v"(N[2][3]), [na24] "=&v"(N[2][4]), [na10] "=&v"(N[1][0]), [na11] "=&v"(N[1][1]), [na12] "=&v"(N[1][2]), [na13] "=&v"(N[1][3]), [na14] "=&v"(N[1][4]), [na00] "=&v"(N[0][0]), [na01] "=&v"(N[0][1]), [na02] "=&v"(N[0][2]), [na03] "=&v"(N[0][3]), [na04] "=&v"(N[0][4])
                : [h0] "v"(hB[0]), [h1] "v"(hB[1]), [h2] "v"(hB[2]), [nx] "v"(nx));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && lane == 0) cyc[threadIdx.x >> 6] = t1 - t0;
    float r = 0.f;
    for (int t = 0; t < 5; ++t) r += acc[t][3];
    for (int p = 0; p < 3; ++p) for (int t = 0; t < 5; ++t) r += (float)N[p][t][1];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE> void run(float* out, long long* cyc, const char* what) {
    const int n = 2000;
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 61440);
    k<MODE><<<256, 256, 61440>>>(out, cyc, 10); (void)hipDeviceSynchronize();
    k<MODE><<<256, 256, 61440>>>(out, cyc, n); (void)hipDeviceSynchronize();
    long long h[4]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-52s %6.1f cycles per MFMA (wave 0), %6.1f (wave 3)\n", what, (double)h[0] / (30.0 * n), (double)h[3] / (30.0 * n));
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 64);
    run<0>(out, cyc, "k-step block as shipped (reads overwrite A operands)");
    run<1>(out, cyc, "MFMAs only");
    run<2>(out, cyc, "reads into a second register set");
    return 0;
}
