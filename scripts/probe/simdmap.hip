// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 256 * 8 * 4);
    k<<<256, 512, 150000>>>(d);
    unsigned h[256 * 8]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 6; ++b) {
        printf("block %d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d simd %u slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    int same = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 4; ++w) same += ((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3);
    printf("waves w and w+4 on the same SIMD: %d of %d\n", same, 256 * 4);
    int distinct = 0;
    for (int b = 0; b < 256; ++b) { unsigned m = 0; for (int w = 0; w < 4; ++w) m |= 1u << ((h[b * 8 + w] >> 4) & 3); distinct += m == 15; }
    printf("waves 0-3 on four distinct SIMDs: %d of 256 blocks\n", distinct);
    return 0;
}
