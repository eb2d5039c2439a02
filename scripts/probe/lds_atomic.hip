// probe: throughput of LDS floating-point atomics (ds_add_f32 / ds_add_f64) against plain ds_write / ds_read+ds_write, per address pattern
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/lds_atomic.hip -o /tmp/lds_atomic && /tmp/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int MODE, int PAT> __global__ void __launch_bounds__(256) k(T* out, int iters) {
    __shared__ T acc[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) acc[i] = T(0);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int idx = PAT == 0 ? lane : PAT == 1 ? (lane >> 1) : PAT == 2 ? ((lane & 3) << 4) + (lane >> 2) : lane * 2;   // 0 consecutive, 1 pairs, 2 (d, slot), 3 stride 2
    T v = T(threadIdx.x) * T(1e-3);
    for (int it = 0; it < iters; ++it) {
        T* p = acc + ((it * 64) & 4095) + idx;
        if (MODE == 0) atomicAdd(p, v);
        else if (MODE == 1) *(volatile T*)p = v;
        else { T o = *(volatile T*)p; *(volatile T*)p = o + v; }
        v += T(1e-6);
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x];
}
template <typename T, int MODE, int PAT> void run(const char* name) {
    T* out; hipMalloc(&out, 1024 * 256 * sizeof(T));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4096;
    k<T, MODE, PAT><<<1024, 256>>>(out, iters);
    hipEventRecord(a);
    k<T, MODE, PAT><<<1024, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // 1024 blocks x 4 waves x iters wave-instructions over 256 CUs
    const double per_cu = 1024.0 * 4 * iters / 256;
    printf("%-28s %8.3f ms  %7.1f cycles per wave-instruction per CU (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / per_cu);
    hipFree(out);
}
int main() {
    run<float, 0, 0>("ds_add_f32 consecutive");
    run<float, 0, 1>("ds_add_f32 pairs");
    run<float, 0, 2>("ds_add_f32 (d,slot)");
    run<float, 0, 3>("ds_add_f32 stride 2");
    run<double, 0, 0>("ds_add_f64 consecutive");
    run<double, 0, 1>("ds_add_f64 pairs");
    run<float, 1, 0>("ds_write_b32 consecutive");
    run<float, 2, 0>("read+add+write f32");
    run<double, 2, 0>("read+add+write f64");
    run<int, 0, 0>("ds_add_u32 consecutive");
    return 0;
}
