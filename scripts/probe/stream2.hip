// probe 2: HBM streaming rate, one wave per workgroup, LDS-resident slabs: register staging vs LDS-DMA (global_load_lds), and a plain read-sum baseline
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
// MODE 0: global_load_lds, contiguous 35 KB slab per phase.  MODE 1: plain loads summed in registers (no LDS), same data
template <int MODE>
__global__ void __launch_bounds__(64) k(const float4* __restrict__ src, float* __restrict__ out) {
    extern __shared__ float4 lds[];
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int layer = 0; layer < 4; ++layer) {
        const float4* p = src + (long)blockIdx.x * 64 * 137 + (long)layer * 64 * 34;
        const int n = 34;        // 34 x (64 lanes x 16 B) = 34 KB
        if (MODE == 0) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < n; ++i)
                __builtin_amdgcn_global_load_lds((const void*)(p + i * 64 + tid), (lds_void*)(lds + i * 64), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const float4 t = lds[tid * 34 + (tid & 31)];
            acc += t.x + t.w;
        } else if (MODE == 2) {      // strided row segments (34 x 16 B of every 137 x 16 B row), straight-line loads, summed
            float4 s = {0, 0, 0, 0};
            const float4* q = src + (long)blockIdx.x * 64 * 137 + layer * 34;
            int r = tid / 34, c = tid - r * 34;
#pragma unroll
            for (int i = 0; i < n; ++i) {
                const float4 v = q[r * 137 + c];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                r += 1; c += 30; if (c >= 34) { c -= 34; r += 1; }
            }
            acc += s.x + s.y + s.z + s.w;
        } else {
            float4 s = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < n; ++i) { const float4 v = p[i * 64 + tid]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            acc += s.x + s.y + s.z + s.w;
        }
    }
    out[blockIdx.x * 64 + tid] = acc;
}
// classic streaming read: 256 threads, grid-stride, many waves per CU
__global__ void __launch_bounds__(256) kread(const float4* __restrict__ src, float* __restrict__ out, long n) {
    float4 s = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float4 v = src[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}
int main() {
    const long B = 1 << 20; const int nblk = B / 64;
    float4* src; float* out;
    hipMalloc(&src, B * 137 * sizeof(float4)); hipMalloc(&out, B * sizeof(float));
    hipMemset(src, 0, B * 137 * sizeof(float4));
    const double gb = B * 136.0 * 16 / 1e9;
    for (size_t lds : {(size_t)35840, (size_t)17920 * 1}) {
        float a = timeit([&] { hipLaunchKernelGGL((k<0>), dim3(nblk), dim3(64), lds, 0, src, out); });
        float b = timeit([&] { hipLaunchKernelGGL((k<1>), dim3(nblk), dim3(64), lds, 0, src, out); });
        float c2 = timeit([&] { hipLaunchKernelGGL((k<2>), dim3(nblk), dim3(64), lds, 0, src, out); });
        printf("LDS %zu: LDS-DMA %.3f ms (%.0f GB/s) | register read-sum %.3f ms (%.0f GB/s) | strided row segments %.3f ms (%.0f GB/s)\n", lds, a, gb / a * 1e3, b, gb / b * 1e3, c2, gb / c2 * 1e3);
    }
    float c = timeit([&] { hipLaunchKernelGGL(kread, dim3(2048), dim3(256), 0, 0, src, out, B * 137); });
    printf("grid-stride float4 read, 2048x256: %.3f ms (%.0f GB/s)\n", c, B * 137.0 * 16 / 1e9 / c * 1e3);
    return 0;
}
