// Shared device/host helpers for the jammy_flows MI355X hot path (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <tuple>
#include <type_traits>

#include "../../include/jammy_hip.h"

#define JF_WAVE 64

namespace jf {

// host-side argument sanity: matrix widths / row counts beyond these are refused (JF_ERR_BADARG) before any size arithmetic
constexpr int64_t JF_MAX_WIDTH = 1 << 24, JF_MAX_ROWS = (int64_t)1 << 40;
inline bool width_ok(int64_t v) { return v >= 1 && v <= JF_MAX_WIDTH; }
inline bool rows_ok(int64_t v) { return v >= 0 && v <= JF_MAX_ROWS; }

// hipFuncAttributeMaxDynamicSharedMemorySize is an attribute of a kernel PER DEVICE: a `static bool` guard sets it on the first device a process
// touches only, and a later launch on another GPU fails.  One of these per call site: a bit per device, set once, safe from several
// host threads (at worst two threads set the same value).
struct LdsAttrOnce {
    std::atomic<uint64_t> done{0};
    void set(const void* fn, int bytes) {
        int dev = 0;
        const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
        if (known && (done.load(std::memory_order_acquire) >> dev & 1)) return;
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (known) done.fetch_or(1ull << dev, std::memory_order_release);
    }
};


// ---------------------------------------------------------------------------------------------
// Kernel launches.  Every entry point launches through jf::launch(kernel, grid, block, lds, stream, args...).  Normally that is
// hipLaunchKernel; while a step plan is being RECORDED on this thread (jf_plan_record_begin .. _end, plan.hip) the launch is not issued but
// copied into the plan -- kernel address, geometry and the bytes of every argument -- so that jf_plan_launch can re-issue the whole step from
// C in one call, with the device pointers into the caller's input / output buffers rebound (include/jammy_hip.h, "step plans").
// ---------------------------------------------------------------------------------------------
struct PlanSink {
    virtual void add_launch(const void* fn, dim3 grid, dim3 block, size_t lds, void** args, const size_t* sizes, const size_t* aligns, int n) = 0;
    virtual void forget(const PlanSink*) {}                        // `dead` is being destroyed: a sink that forwards to it must stop doing so
    virtual ~PlanSink() {}
};
PlanSink*& plan_sink();                                            // thread-local (plan.hip); nullptr = launch for real

template <typename... KArgs, typename... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
    std::tuple<std::remove_cv_t<KArgs>...> vals{static_cast<KArgs>(args)...};      // converted exactly as a <<<>>> launch would convert them
    void* ptrs[sizeof...(KArgs) ? sizeof...(KArgs) : 1];
    std::apply([&](auto&... v) { int i = 0; ((ptrs[i++] = (void*)&v), ...); }, vals);
    if (PlanSink* sink = plan_sink()) {
        static constexpr size_t sizes[] = {sizeof(KArgs)..., 0}, aligns[] = {alignof(KArgs)..., 0};
        sink->add_launch((const void*)kernel, grid, block, lds, ptrs, sizes, aligns, (int)sizeof...(KArgs));
        return;
    }
    (void)hipLaunchKernel((const void*)kernel, grid, block, ptrs, lds, st);
}

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
// Loads are issued in batches of up to 16 per lane (measured on MI355X, scripts/probe/stream2.hip: HBM streams at full rate once a few
// tens of 16-byte loads per lane are in flight per SIMD, which occupancy supplies here).  A batch is straight-line code: piece coordinates advance
// incrementally (no per-piece division), out-of-range lanes re-read a valid piece and only skip the LDS write, so the U
// global_load_dwordx4 are issued back to back and waited for once (any branch between the loads makes hipcc wait after each of them).
template <typename T> struct StageCursor {
    const T* src; int64_t src_stride; T* tile; int tile_stride;
    int nv, rows, last_row, total, nthreads, tid;
    int r, c, dq, dr, it;
};

// One batch = U loads then U LDS writes, written as a template recursion that carries the loaded pieces as a parameter pack, so that
// every piece lives in a plain local: an array `E v[U]` is not promoted to registers by hipcc here and bounces through scratch memory.
template <typename E> struct StagePiece { E v; int off; };

template <typename T, int U, bool VEC> struct StageRec {
    template <typename... P> static __device__ __forceinline__ void run(StageCursor<T>& s, P... done) {
        constexpr int N = Vec16<T>::N;
        using V = typename Vec16<T>::type;
        using E = typename std::conditional<VEC, V, T>::type;
        constexpr int W = VEC ? N : 1;
        // pieces past the end of the slab (last batch only) fall on row >= rows: they are clamped to the last row and simply rewrite one
        // of its pieces with the same data -- no predicate, hence no branch for the compiler to sink the load into
        const int rr = s.r < s.rows ? s.r : s.rows - 1;
        const int rs = rr < s.last_row ? rr : s.last_row;
        StagePiece<E> p;
        p.v = *reinterpret_cast<const E*>(s.src + (int64_t)rs * s.src_stride + s.c * W);
        p.off = rr * s.tile_stride + s.c * W;
        s.r += s.dq; s.c += s.dr;
        if (s.c >= s.nv) { s.c -= s.nv; s.r += 1; }
        StageRec<T, U - 1, VEC>::run(s, done..., p);
    }
};
template <typename T, bool VEC> struct StageRec<T, 0, VEC> {
    template <typename... P> static __device__ __forceinline__ void run(StageCursor<T>& s, P... done) {
        __builtin_amdgcn_sched_barrier(0);   // keep every load of the batch ahead of the first LDS write (the scheduler otherwise caps ~8 in flight)
        ((*reinterpret_cast<decltype(done.v)*>(s.tile + done.off) = done.v), ...);   // in load order: the waits count down vmcnt(U-1) .. vmcnt(0)
    }
};
template <typename T, int U, bool VEC> __device__ __forceinline__ void stage_batch(StageCursor<T>& s) {
    StageRec<T, U, VEC>::run(s);
    s.it += U;
}

// All pieces of a slab: batches of 16 while more than 16 remain, then ONE batch sized to the remainder rounded up to even (the extra
// piece, if any, is a clamped duplicate), so a slab of <= 16 pieces per lane costs a single HBM round trip.
template <typename T, bool VEC> __device__ __forceinline__ void stage_all(StageCursor<T>& s, int iters) {
    while (iters - s.it > 16) stage_batch<T, 16, VEC>(s);
    switch ((iters - s.it + 1) >> 1) {
        case 1: stage_batch<T, 2, VEC>(s); break;
        case 2: stage_batch<T, 4, VEC>(s); break;
        case 3: stage_batch<T, 6, VEC>(s); break;
        case 4: stage_batch<T, 8, VEC>(s); break;
        case 5: stage_batch<T, 10, VEC>(s); break;
        case 6: stage_batch<T, 12, VEC>(s); break;
        case 7: stage_batch<T, 14, VEC>(s); break;
        case 8: stage_batch<T, 16, VEC>(s); break;
        default: break;
    }
}

template <typename T>
__device__ __forceinline__ void stage_rows(T* __restrict__ tile, int tile_stride, const T* __restrict__ src, int64_t src_stride,
                                           int ncols, int rows, int valid_rows, int tid, int nthreads, bool vec_ok) {
    constexpr int N = Vec16<T>::N;
    StageCursor<T> s;
    s.src = src; s.src_stride = src_stride; s.tile = tile; s.tile_stride = tile_stride;
    s.nv = vec_ok ? ncols / N : ncols;                     // pieces per row (16-byte vectors or single elements)
    s.rows = rows; s.last_row = valid_rows - 1; s.total = rows * s.nv; s.nthreads = nthreads; s.tid = tid;
    s.r = tid / s.nv; s.c = tid - s.r * s.nv;              // piece idx = it * nthreads + tid  ->  (row r, piece c)
    s.dq = nthreads / s.nv; s.dr = nthreads - s.dq * s.nv; s.it = 0;
    const int iters = (s.total + nthreads - 1) / nthreads;
    if (vec_ok) stage_all<T, true>(s, iters);
    else stage_all<T, false>(s, iters);
}

// ---------------------------------------------------------------------------------------------
// Staging for tiles of few rows (rows * LPR == 64 lanes, LPR a power of two >= 4): LPR neighbouring lanes share a row and walk it in
// 16-byte pieces j, j + LPR, j + 2 LPR, ...  Every address of a lane is its first one plus a compile-time constant, so a piece costs
// no VALU instruction at all (immediate offsets of global_load_dwordx4 / ds_write_b128); the generic cursor above spends ~24 per
// piece.  Per instruction the wave reads `rows` segments of 16 LPR bytes (64 B for D = 3, 4; 128 B for D = 5..8); the other half of each
// 128-byte line is served by the L1 hit of the next instruction.  Only the last piece of a lane can fall past the row: it is clamped
// onto the row's last piece (duplicate load and store of the same data).  Requires 16-byte aligned rows (vec_ok).
// ---------------------------------------------------------------------------------------------
template <typename T, int LPR, int I, int NIT> struct GroupStage {
    using V = typename Vec16<T>::type;
    template <typename... P> static __device__ __forceinline__ void run(const V* __restrict__ g, V* __restrict__ l, int last_off, P... done) {
        if constexpr (I + 1 < NIT) {
            const V v = g[I * LPR];
            GroupStage<T, LPR, I + 1, NIT>::run(g, l, last_off, done..., v);
        } else {
            const V v = g[last_off];                                  // (NIT-1) * LPR, or clamped onto the row's last piece
            __builtin_amdgcn_sched_barrier(0);                       // all loads ahead of the first LDS write
            int i = 0;
            ((l[i++ * LPR] = done), ...);
            l[last_off] = v;
        }
    }
};

template <typename T, int LPR>
__device__ __forceinline__ void stage_rows_grouped(T* __restrict__ tile, int tile_stride, const T* __restrict__ src, int64_t src_stride, int ncols,
                                                   int valid_rows, int lane) {
    using V = typename Vec16<T>::type;
    constexpr int N = Vec16<T>::N;
    const int r = lane / LPR, j = lane % LPR;
    const int rs = r < valid_rows ? r : valid_rows - 1;              // rows past the end replicate the last valid row
    const int nv = ncols / N;
    const int nit = (nv + LPR - 1) / LPR;                            // pieces per lane (uniform)
    const int jj = j < nv ? j : nv - 1;
    const V* g = reinterpret_cast<const V*>(src + (int64_t)rs * src_stride) + jj;
    V* l = reinterpret_cast<V*>(tile + r * tile_stride) + jj;
    const int c_last = jj + (nit - 1) * LPR;
    const int last_off = (c_last < nv ? c_last : nv - 1) - jj;
    switch (nit) {
#define JF_GS(n) case n: GroupStage<T, LPR, 0, n>::run(g, l, last_off); break;
        JF_GS(1) JF_GS(2) JF_GS(3) JF_GS(4) JF_GS(5) JF_GS(6) JF_GS(7) JF_GS(8) JF_GS(9) JF_GS(10) JF_GS(11) JF_GS(12)
        JF_GS(13) JF_GS(14) JF_GS(15) JF_GS(16) JF_GS(17) JF_GS(18) JF_GS(19) JF_GS(20)
#undef JF_GS
        default: break;                                              // callers fall back to stage_rows for longer rows
    }
}
constexpr int JF_GROUP_STAGE_MAX_PIECES = 20;

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

template <typename T, int D> __device__ inline void store_d(T* __restrict__ p, const T (&v)[D]) {
    constexpr int N = Vec16<T>::N;
    using V = typename Vec16<T>::type;
    if constexpr (D % N == 0) {
#pragma unroll
        for (int i = 0; i < D / N; ++i) {
            V t;
            if constexpr (N == 4) { t.x = v[4 * i]; t.y = v[4 * i + 1]; t.z = v[4 * i + 2]; t.w = v[4 * i + 3]; }
            else { t.x = v[2 * i]; t.y = v[2 * i + 1]; }
            reinterpret_cast<V*>(p)[i] = t;
        }
    } else {
#pragma unroll
        for (int i = 0; i < D; ++i) p[i] = v[i];
    }
}

// wave-aggregated status counter bump (one atomic per wave)
__device__ inline void status_add(int32_t* status, int which, bool flag) {
    if (status == nullptr) return;
    const unsigned long long m = __ballot(flag);
    if (m != 0ull && (threadIdx.x & (JF_WAVE - 1)) == (unsigned)(__ffsll((long long)m) - 1)) atomicAdd(status + which, (int32_t)__popcll(m));
}

// launch status of the kernel just enqueued.  hipPeekAtLastError does not clear the sticky error state, so an error left behind by an
// unrelated earlier call of the application is neither swallowed nor misattributed silently: it keeps surfacing until the caller handles it.
inline int check_launch() {
    hipError_t e = hipPeekAtLastError();
    return e == hipSuccess ? JF_OK : JF_ERR_LAUNCH;
}

}  // namespace jf
