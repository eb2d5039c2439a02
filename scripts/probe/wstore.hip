// HBM write-pattern probe: how fast can a [B x 548] f32 matrix be written (a) linearly with 16-byte stores, (b) in the tile order of a
// GEMM epilogue (workgroup = 128 rows, 64-column tiles, each wave-store = 2 rows x 128 B), (c) the same with 4-column (16 B) per lane
// stores (lane = row, 32 rows x 32 B per wave-store), (d) like (b) but every wave writes one full 256-B row piece per store (LDS-staged order).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int N = 548, BROWS = 128;

__global__ void __launch_bounds__(256) k_linear(float4* out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// (b) lane: col = lane%32, rows (r&3)+8(r>>2)+4(lane/32) -- the MFMA 32x32 result layout
__global__ void __launch_bounds__(256) k_tile(float* out, int64_t stride, int64_t B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lq = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * BROWS + wave * 32;
    for (int t = 0; t < (N + 63) / 64; ++t)
        for (int ct = 0; ct < 2; ++ct) {
            const int gc = t * 64 + ct * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gr = row0 + (r & 3) + 8 * (r >> 2) + 4 * lq;
                if (gc < N && gr < B) out[gr * stride + gc] = (float)r;
            }
        }
}
// (c) lane = row (lane%32), 4 consecutive columns per lane: 16-byte stores
__global__ void __launch_bounds__(256) k_tile4(float* out, int64_t stride, int64_t B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lq = lane >> 5;
    const int64_t gr = (int64_t)blockIdx.x * BROWS + wave * 32 + li;
    for (int t = 0; t < (N + 63) / 64; ++t)
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int gc = t * 64 + ct * 32 + g * 8 + lq * 4;
                if (gc + 3 < N && gr < B) *reinterpret_cast<float4*>(out + gr * stride + gc) = make_float4(1.f, 2.f, 3.f, 4.f);
            }
}
// (d) 16 lanes x 16 B = one 256-B row piece; a wave-store covers 4 rows x 256 B
__global__ void __launch_bounds__(256) k_rows(float* out, int64_t stride, int64_t B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * BROWS + wave * 32;
    for (int t = 0; t < (N + 63) / 64; ++t)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int64_t gr = row0 + g * 4 + (lane >> 4);
            const int gc = t * 64 + (lane & 15) * 4;
            if (gc + 3 < N && gr < B) *reinterpret_cast<float4*>(out + gr * stride + gc) = make_float4(1.f, 2.f, 3.f, 4.f);
        }
}
// (e) each workgroup writes its 128 rows completely, row by row, 16 B per lane, fully contiguous (the ideal epilogue order)
__global__ void __launch_bounds__(256) k_contig(float* out, int64_t stride, int64_t B) {
    const int64_t row0 = (int64_t)blockIdx.x * BROWS;
    float4* p = reinterpret_cast<float4*>(out + row0 * stride);
    const int n4 = BROWS * N / 4;
    for (int i = threadIdx.x; i < n4; i += 256) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <typename F> void timeit(const char* name, F f, double bytes) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-28s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
}

int main() {
    const int64_t B = 1 << 20;
    float* out; (void)hipMalloc(&out, (size_t)B * 576 * 4);
    const double bytes = (double)B * N * 4;
    const unsigned grid = (unsigned)(B / BROWS);
    timeit("linear float4", [&] { k_linear<<<4096, 256>>>((float4*)out, (size_t)B * N / 4); }, bytes);
    timeit("tile dword (mfma layout)", [&] { k_tile<<<grid, 256>>>(out, N, B); }, bytes);
    timeit("tile dword, stride 576", [&] { k_tile<<<grid, 256>>>(out, 576, B); }, bytes);
    timeit("tile float4 lane=row", [&] { k_tile4<<<grid, 256>>>(out, N, B); }, bytes);
    timeit("tile float4 row pieces", [&] { k_rows<<<grid, 256>>>(out, N, B); }, bytes);
    timeit("workgroup-contiguous", [&] { k_contig<<<grid, 256>>>(out, N, B); }, bytes);
    return 0;
}
