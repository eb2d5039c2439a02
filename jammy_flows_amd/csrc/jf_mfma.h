// MFMA instruction traits (gfx950).  Fragment layouts (lane l of the 64-lane wave):
//   A[m][k]: m = l % MT, k-slot = l / MT;   B[k][n]: n = l % MT, k-slot = l / MT;   D[m][n]: n = l % MT, m = row_of(register, l)
// (row_of verified on the device: scripts/probe/mfma16.hip and the dense-kernel tests).
#pragma once
#include <hip/hip_runtime.h>

namespace jf {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <typename T> struct Mfma;
template <> struct Mfma<float> {
    static constexpr int MT = 32, KS = 2, NREG = 16, RUN = 4;   // RUN: consecutive registers hold consecutive rows
    using Acc = f32x16;
    static __device__ __forceinline__ Acc mma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
};
template <> struct Mfma<double> {
    static constexpr int MT = 16, KS = 4, NREG = 4, RUN = 1;
    using Acc = f64x4;
    static __device__ __forceinline__ Acc mma(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int reg, int lane) { return (lane >> 4) + 4 * reg; }
};

// 16 x 16 x 4 tiles for both precisions: the fused MLP + flow kernel wants 16-row waves (one row group pass of the flow per MFMA row tile)
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <typename T> struct Mfma16;
template <> struct Mfma16<float> {
    static constexpr int MT = 16, KS = 4, NREG = 4, RUN = 4;     // RUN: consecutive registers hold consecutive rows
    using Acc = f32x4;
    static __device__ __forceinline__ Acc mma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int reg, int lane) { return 4 * (lane >> 4) + reg; }
};
template <> struct Mfma16<double> {
    static constexpr int MT = 16, KS = 4, NREG = 4, RUN = 1;
    using Acc = f64x4;
    static __device__ __forceinline__ Acc mma(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int reg, int lane) { return (lane >> 4) + 4 * reg; }
};

// workgroup barrier that waits for this wave's LDS traffic only (a __syncthreads() also drains the wave's outstanding global stores)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace jf
