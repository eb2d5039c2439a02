// probe: HBM streaming rate of the per-sample staging patterns (64-thread workgroups, 35.8 KB LDS each => 4 waves/CU)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE, int U>
__global__ void __launch_bounds__(64) k(const float4* __restrict__ src, float* __restrict__ out, int nblk) {
    extern __shared__ float4 lds[];
    const int tid = threadIdx.x;
    const long row0 = (long)blockIdx.x * 64;
    float acc = 0.f;
    for (int layer = 0; layer < 4; ++layer) {
        __syncthreads();
        const int nv = (layer == 3) ? 35 : 34;               // 16-byte pieces per row of this layer
        const int col0 = layer * 34;
        const int total = 64 * nv;
        for (int base = 0; base < total; base += U * 64) {
            float4 v[U]; int off[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * 64 + tid;
                const int r = idx / nv, c = idx - r * nv;
                off[u] = idx < total ? r * 35 + c : -1;
                if (MODE == 0) { if (idx < total) v[u] = src[(row0 + r) * 137 + col0 + c]; }          // strided segments of full rows (stride 548 floats)
                else { if (idx < total) v[u] = src[(long)blockIdx.x * 64 * 137 + (long)layer * 64 * 34 + idx]; }   // contiguous
            }
#pragma unroll
            for (int u = 0; u < U; ++u) if (off[u] >= 0) lds[off[u]] = v[u];
        }
        __syncthreads();
        const float4 t = lds[tid * 35 + (tid & 31)];
        acc += t.x + t.w;
    }
    out[blockIdx.x * 64 + tid] = acc;
}
template <int MODE, int U> float run(const float4* src, float* out, int nblk, size_t lds) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<MODE, U>), dim3(nblk), dim3(64), lds, 0, src, out, nblk);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<MODE, U>), dim3(nblk), dim3(64), lds, 0, src, out, nblk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}
int main() {
    const long B = 1 << 20; const int nblk = B / 64;
    float4* src; float* out;
    hipMalloc(&src, B * 137 * sizeof(float4)); hipMalloc(&out, B * sizeof(float));
    hipMemset(src, 0, B * 137 * sizeof(float4));
    const double gb = B * 137.0 * 16 / 1e9;
    for (size_t lds : {(size_t)35840, (size_t)17920}) {
        printf("LDS %zu: strided U=8 %.3f ms | strided U=36 %.3f ms | contiguous U=8 %.3f ms | contiguous U=36 %.3f ms   (%.2f GB per launch)\n", lds,
               run<0, 8>(src, out, nblk, lds), run<0, 36>(src, out, nblk, lds), run<1, 8>(src, out, nblk, lds), run<1, 36>(src, out, nblk, lds), gb);
    }
    return 0;
}
