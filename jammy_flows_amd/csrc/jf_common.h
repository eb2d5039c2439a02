// Shared device/host helpers for the jammy_flows MI355X hot path (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/jammy_hip.h"

#define JF_WAVE 64

namespace jf {

// ---------------------------------------------------------------------------------------------
// vector types: 16-byte accesses are what both HBM (global_load_dwordx4) and LDS (ds_read_b128) want
// ---------------------------------------------------------------------------------------------
template <typename T> struct Vec16;
template <> struct Vec16<float> { using type = float4; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = double2; static constexpr int N = 2; };

// LDS row stride (in elements) for a lane-per-row tile read with 16-byte ds_read_b128:
// stride = N*odd elements => the 16 lanes of each b128 service group hit 16 distinct 16-byte slots.
template <typename T> __host__ __device__ inline int padded_stride(int n) {
    constexpr int N = Vec16<T>::N;
    int s = (n + N - 1) / N;  // in 16-byte slots
    if ((s & 1) == 0) s += 1;
    return s * N;
}

// ---------------------------------------------------------------------------------------------
// Cooperative, coalesced staging of a [rows x ncols] slab of a row-major parameter matrix into an LDS tile
// (lane-per-row consumption afterwards).  Consecutive lanes read consecutive 16-byte pieces of a row and
// then continue with the next row, so every wave load instruction covers whole contiguous 1 KiB pieces of
// at most a few rows.  Falls back to element-wise copies when the slab is not 16-byte aligned.
// `nthreads` threads (tid in [0,nthreads)) cooperate; rows beyond `valid_rows` replicate the last valid row.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ inline void stage_rows(T* __restrict__ tile, int tile_stride, const T* __restrict__ src, int64_t src_stride,
                                  int ncols, int rows, int valid_rows, int tid, int nthreads, bool vec_ok) {
    constexpr int N = Vec16<T>::N;
    using V = typename Vec16<T>::type;
    if (vec_ok) {
        const int nv = ncols / N;  // vectors per row (ncols is a multiple of N when vec_ok)
        const int total = rows * nv;
        int r = tid / nv, c = tid - r * nv;
        const int dr = nthreads / nv, dc = nthreads - dr * nv;
        for (int idx = tid; idx < total; idx += nthreads) {
            const int rs = r < valid_rows ? r : valid_rows - 1;
            const V v = *reinterpret_cast<const V*>(src + (int64_t)rs * src_stride + c * N);
            *reinterpret_cast<V*>(tile + r * tile_stride + c * N) = v;
            r += dr;
            c += dc;
            if (c >= nv) { c -= nv; r += 1; }
        }
    } else {
        const int total = rows * ncols;
        for (int idx = tid; idx < total; idx += nthreads) {
            const int r = idx / ncols, c = idx - r * ncols;
            const int rs = r < valid_rows ? r : valid_rows - 1;
            tile[r * tile_stride + c] = src[(int64_t)rs * src_stride + c];
        }
    }
}

template <typename T> __host__ inline bool aligned16(const void* p, int64_t stride_elems, int64_t col0_elems) {
    constexpr int N = Vec16<T>::N;
    return ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) && (stride_elems % N == 0) && (col0_elems % N == 0);
}

// load D consecutive elements from an LDS row; section offsets are multiples of D and the row base is 16-byte aligned
template <typename T, int D> __device__ inline void load_d(const T* __restrict__ p, T (&v)[D]) {
    constexpr int N = Vec16<T>::N;
    using V = typename Vec16<T>::type;
    if constexpr (D % N == 0) {
#pragma unroll
        for (int i = 0; i < D / N; ++i) {
            const V t = reinterpret_cast<const V*>(p)[i];
            if constexpr (N == 4) { v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
            else { v[2 * i] = t.x; v[2 * i + 1] = t.y; }
        }
    } else {
#pragma unroll
        for (int i = 0; i < D; ++i) v[i] = p[i];
    }
}

// wave-aggregated status counter bump (one atomic per wave)
__device__ inline void status_add(int32_t* status, int which, bool flag) {
    if (status == nullptr) return;
    const unsigned long long m = __ballot(flag);
    if (m != 0ull && (threadIdx.x & (JF_WAVE - 1)) == (unsigned)(__ffsll((long long)m) - 1)) atomicAdd(status + which, (int32_t)__popcll(m));
}

inline int check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? JF_OK : JF_ERR_LAUNCH;
}

}  // namespace jf
