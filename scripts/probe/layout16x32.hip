// Checks, on the device, the assumptions cond_split_kernels.hip rests on:
//  (1) v_mfma_f32_16x16x32_bf16 operand / result layout: A lane l = (m = l % 16, k = 8 (l / 16) + i), B lane l = (n = l % 16, same k),
//      D lane l register r = (m = 4 (l / 16) + r, n = l % 16)
//  (2) v_permlane16_swap / v_permlane32_swap butterfly = sum over lanes {l, l^16, l^32, l^48}
//  (3) global_load_lds_dwordx4 lands lane i's 16 bytes at (wave-uniform LDS base) + 16 i
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
typedef __attribute__((address_space(3))) void* lptr;
typedef const __attribute__((address_space(1))) void* gptr;

__global__ void k(const float* A, const float* B, float* D, float* S, const unsigned char* src, unsigned char* dst) {
    __shared__ __align__(16) unsigned char sm[2048];
    const int l = threadIdx.x, m = l & 15, q = l >> 4;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)A[m * 32 + 8 * q + i]; b[i] = (__bf16)B[(8 * q + i) * 16 + m]; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + m] = acc[r];
    const float v = (float)(l * l % 37);
    float a1 = v, b1 = v;                     // inline asm: hipcc (ROCm 7.2) turns builtin_permlane16_swap(v, v) -> r[0] + r[1] into r[0] + r[0]
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a1), "+v"(b1));
    float c1 = a1 + b1, e1 = c1;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c1), "+v"(e1));
    S[l] = c1 + e1;
    __builtin_amdgcn_global_load_lds((gptr)(src + l * 16), (lptr)(sm + 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 16; ++i) dst[l * 16 + i] = sm[1024 + l * 16 + i];
}
int main() {
    float hA[16 * 32], hB[32 * 16], hD[256], hS[64];
    unsigned char hs[1024], hd[1024];
    srand(1);
    for (auto& v : hA) v = (float)(rand() % 17 - 8);
    for (auto& v : hB) v = (float)(rand() % 13 - 6);
    for (int i = 0; i < 1024; ++i) hs[i] = (unsigned char)(i * 7 + 3);
    float *A, *B, *D, *S; unsigned char *s, *d;
    (void)hipMalloc(&A, sizeof(hA)); (void)hipMalloc(&B, sizeof(hB)); (void)hipMalloc(&D, sizeof(hD)); (void)hipMalloc(&S, sizeof(hS));
    (void)hipMalloc(&s, 1024); (void)hipMalloc(&d, 1024);
    (void)hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(B, hB, sizeof(hB), hipMemcpyHostToDevice);
    (void)hipMemcpy(s, hs, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(A, B, D, S, s, d);
    (void)hipMemcpy(hD, D, sizeof(hD), hipMemcpyDeviceToHost); (void)hipMemcpy(hS, S, sizeof(hS), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    double e = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { double r = 0; for (int kk = 0; kk < 32; ++kk) r += hA[m * 32 + kk] * hB[kk * 16 + n]; e = fmax(e, fabs(r - hD[m * 16 + n])); }
    double es = 0;
    for (int l = 0; l < 64; ++l) { double r = 0; for (int g = 0; g < 4; ++g) { const int j = (l & 15) + 16 * g; r += (double)(j * j % 37); } es = fmax(es, fabs(r - hS[l])); }
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += hd[i] != hs[i];
    printf("mfma 16x16x32 bf16 layout: max err %g | permlane butterfly: max err %g | global_load_lds: %d bytes differ\n", e, es, bad);
    return 0;
}
