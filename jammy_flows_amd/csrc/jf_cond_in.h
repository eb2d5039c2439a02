// MLP inputs read where they are.  The input row of sub-pdf i's amortisation MLP is cat[conditional_input, embed(x_0), ..., embed(x_{i-1})]
// (jammy_flows/main/default.py:946-962; embed = identity for Euclidean / interval targets, (cos, sin) for S1, (x, y, z) for S2:
// sphere_base.py:305-332, 786-794).  The reference builds it with torch.cat; round 1-3 of this library built it with one launch
// (jf_conditioning_rows) and re-read it from HBM.  A consumer that takes a CondIn reads the SEGMENTS themselves while it stages its input tile:
// no launch, no (B, K1) round trip.  cond_in_value evaluates exactly what conditioning_kernel writes (same operations, same values).
#pragma once
#include "jf_common.h"
#include "jf_math.h"
#include "jf_sphere.h"

namespace jf {

constexpr int JF_COND_IN_MAX = 4;
struct CondIn { int n; jf_cond_segment s[JF_COND_IN_MAX]; };      // n == 0: the plain (in, in_stride) matrix is the input

// FAST (float32 consumers that stage their own input tile): the embedding's sines / cosines on the hardware unit, v_sin_f32 / v_cos_f32 of
// angle / 2 pi (angles lie in [0, 2 pi]: no range reduction; absolute error ~1e-6 measured, tests/test_gpu_plan.py) instead of OCML's sinf / cosf
// (~60 instructions each) -- the staging is a serial prologue of a workgroup whose kernel is already bound by vector issue: with sinf the fused
// read cost as much as the launch it replaced (cond_gf_split_kernel 0.501 -> 0.529 ms per 2^20 rows).
template <bool FAST> struct CondTrig {
    template <typename T> static __device__ __forceinline__ T sin(T x) { return M<T>::sin(x); }
    template <typename T> static __device__ __forceinline__ T cos(T x) { return M<T>::cos(x); }
};
template <> struct CondTrig<true> {
    static __device__ __forceinline__ float sin(float x) { return __builtin_amdgcn_sinf(x * 0.15915494309189535f); }
    static __device__ __forceinline__ float cos(float x) { return __builtin_amdgcn_cosf(x * 0.15915494309189535f); }
};
template <typename T, bool FAST = false> __device__ __forceinline__ T cond_seg_value(const jf_cond_segment& g, int64_t row, int col) {
    using TR = CondTrig<FAST && std::is_same<T, float>::value>;
    const T* r = static_cast<const T*>(g.src) + row * g.stride;
    if (g.kind == 0) return r[col];
    // only the column's own factors of s1_to_eucl / s2_to_eucl
    if (g.kind == 1) return col == 0 ? TR::cos(r[0]) : TR::sin(r[0]);
    if (col == 2) return TR::cos(safe_angle_pi(r[0]));
    return TR::sin(safe_angle_pi(r[0])) * (col == 0 ? TR::cos(r[1]) : TR::sin(r[1]));
}
template <typename T, bool FAST = false> __device__ __forceinline__ T cond_in_value(const CondIn& a, int64_t row, int col) {
    T val = T(0);
#pragma unroll
    for (int i = 0; i < JF_COND_IN_MAX; ++i) {
        if (i < a.n) {
            const int w = a.s[i].kind == 0 ? a.s[i].n_in : a.s[i].kind + 1;
            if (col >= 0 && col < w) val = cond_seg_value<T, FAST>(a.s[i], row, col);
            col -= w;
        }
    }
    return val;
}

// The same value in three steps, for staging loops that want ALL their global loads in flight before the first use (a load inside a branch is
// waited for inside the branch: four elements per thread became four serial HBM round trips with cond_in_value -- 0.501 -> 0.520 ms per 2^20 rows
// of the fused block): cond_in_locate (arithmetic + kernel-argument reads only) -> two unconditional loads -> cond_in_finish (selects).
template <typename T> struct CondLoc { const T* pa; const T* pb; int kind, col; };
template <typename T> __device__ __forceinline__ CondLoc<T> cond_in_locate(const CondIn& a, int64_t row, int col) {
    CondLoc<T> l;
    const T* r0 = static_cast<const T*>(a.s[0].src) + row * a.s[0].stride;
    l.pa = r0; l.pb = r0; l.kind = 0; l.col = 0;                      // (a column past the row's width reads element 0 of segment 0; the caller discards it)
#pragma unroll
    for (int i = 0; i < JF_COND_IN_MAX; ++i) {
        if (i < a.n) {                                               // uniform
            const int kind = a.s[i].kind;
            const int w = kind == 0 ? a.s[i].n_in : kind + 1;
            const bool here = col >= 0 && col < w;
            const T* r = static_cast<const T*>(a.s[i].src) + row * a.s[i].stride;
            const T* pa = r + (kind == 0 ? col : 0);
            const T* pb = r + (kind == 2 ? 1 : (kind == 0 ? col : 0));
            l.pa = here ? pa : l.pa; l.pb = here ? pb : l.pb; l.kind = here ? kind : l.kind; l.col = here ? col : l.col;
            col -= w;
        }
    }
    return l;
}
template <typename T, bool FAST> __device__ __forceinline__ T cond_in_finish(const CondLoc<T>& l, T va, T vb, bool any_embed) {
    if (!any_embed) return va;                                       // uniform: plain column ranges only
    using TR = CondTrig<FAST && std::is_same<T, float>::value>;
    const T th = l.kind == 2 ? safe_angle_pi(va) : va;
    const T s1 = TR::sin(th), c1 = TR::cos(th), s2 = TR::sin(vb), c2 = TR::cos(vb);
    const T e1 = l.col == 0 ? c1 : s1;                               // S1: (cos, sin)
    const T e2 = l.col == 2 ? c1 : s1 * (l.col == 0 ? c2 : s2);      // S2: (sin th cos ph, sin th sin ph, cos th)
    return l.kind == 0 ? va : (l.kind == 1 ? e1 : e2);
}
__device__ __forceinline__ bool cond_in_any_embed(const CondIn& a) {
    bool e = false;
#pragma unroll
    for (int i = 0; i < JF_COND_IN_MAX; ++i) e = e || (i < a.n && a.s[i].kind != 0);
    return e;
}

// host: segments -> CondIn; their widths must add up to the MLP's input width K1
static inline int cond_in_make(const jf_cond_segment* segs, int32_t n, int32_t K1, CondIn& out) {
    if (!segs || n < 1 || n > JF_COND_IN_MAX) return n > JF_COND_IN_MAX ? JF_ERR_UNSUPPORTED : JF_ERR_BADARG;
    int w = 0;
    out.n = n;
    for (int i = 0; i < n; ++i) {
        if (!segs[i].src || segs[i].kind < 0 || segs[i].kind > 2 || segs[i].n_in < 0 || segs[i].n_in > JF_MAX_WIDTH) return JF_ERR_BADARG;
        out.s[i] = segs[i];
        w += segs[i].kind == 0 ? segs[i].n_in : segs[i].kind + 1;
    }
    return w == K1 ? JF_OK : JF_ERR_BADARG;
}

}  // namespace jf
