// prints what v_permlane16_swap / v_permlane32_swap return for vdst = lane, src = 100 + lane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    const unsigned l = threadIdx.x;
    const auto a = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
    o[l] = a[0]; o[64 + l] = a[1]; o[128 + l] = b[0]; o[192 + l] = b[1];
}
int main() {
    unsigned* d; unsigned h[256];
    (void)hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane16_swap r[0]", "permlane16_swap r[1]", "permlane32_swap r[0]", "permlane32_swap r[1]"};
    for (int j = 0; j < 4; ++j) { printf("%s:", names[j]); for (int l = 0; l < 64; l += 8) printf(" [%u]=%u", l, h[j * 64 + l]); printf("\n"); }
    return 0;
}
