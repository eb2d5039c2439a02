// layout check of v_mfma_f32_16x16x4_f32: A[m][k] lane = (m = l%16, k = l/16), B[k][n] lane = (n = l%16, k = l/16); which (m, n) does D register v of lane l hold?
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void k(float* out) {
    const int l = threadIdx.x;
    // A[m][k] = m + 1 for k == 0 else 0;  B[k][n] = 100 * (n + 1) for k == 0 else 0  ->  D[m][n] = (m + 1) * 100 * (n + 1)
    const float a = (l / 16 == 0) ? float(l % 16 + 1) : 0.f;
    const float b = (l / 16 == 0) ? 100.f * float(l % 16 + 1) : 0.f;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int ok_a = 1, ok_b = 1;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        const int n = l % 16, q = l / 16;
        const float want_a = (4 * q + v + 1) * 100.f * (n + 1);      // m = 4q + v
        const float want_b = (q + 4 * v + 1) * 100.f * (n + 1);      // m = q + 4v
        if (h[l * 4 + v] != want_a) ok_a = 0;
        if (h[l * 4 + v] != want_b) ok_b = 0;
    }
    printf("m = 4*(lane/16) + v : %s\nm = (lane/16) + 4*v : %s\n", ok_a ? "YES" : "no", ok_b ? "YES" : "no");
    return 0;
}
