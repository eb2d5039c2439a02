// 'g' layer device code for the lane = (row, dimension) layout used by the chain kernels.
//
// A wave holds 64/G rows; the G = next-power-of-two(D) neighbouring lanes of a group own the D coordinates of one row
// (lanes g >= D shadow coordinate D-1 and never store).  Everything per coordinate (mixture sums, inverse-CDF stage, bisection)
// is plain scalar code, so the register footprint does not grow with D; the three places the reference reduces over the
// coordinates (Householder dot products, sum of log-derivatives, the Newton stopping rule) become DPP butterflies inside the group.
// Same arithmetic as jf_gf.h (which documents the reference lines); only the data distribution differs.
#pragma once
#include "jf_gf.h"

namespace jf {

// ---------------------------------------------------------------------------------------------------------- DPP group reductions
template <int CTRL> __device__ __forceinline__ float dpp_swap(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_swap(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane i <-> 7-i inside each 8 lanes (after the quad steps both quads hold their sums)

template <typename T, int G> __device__ __forceinline__ T group_sum(T v) {
    if constexpr (G >= 2) v += dpp_swap<DPP_XOR1>(v);
    if constexpr (G >= 4) v += dpp_swap<DPP_XOR2>(v);
    if constexpr (G >= 8) v += dpp_swap<DPP_HALF_MIRROR>(v);
    return v;
}
template <typename T, int G> __device__ __forceinline__ T group_max(T v) {
    if constexpr (G >= 2) v = M<T>::max(v, dpp_swap<DPP_XOR1>(v));
    if constexpr (G >= 4) v = M<T>::max(v, dpp_swap<DPP_XOR2>(v));
    if constexpr (G >= 8) v = M<T>::max(v, dpp_swap<DPP_HALF_MIRROR>(v));
    return v;
}

// ---------------------------------------------------------------------------------------------------------- Householder rotations
// p = row + d (the lane's coordinate), live = lane owns a real coordinate.  RAW rows hold the reference's unnormalised vectors
// (H = I - 2 v v^T / |v|^2, gaussianization_flow.py:457-471); derived rows hold sqrt(2) v / |v|.
template <typename T, int G, bool RAW> __device__ __forceinline__ T gfg_reflect(const T* __restrict__ p, int off, bool live, T x) {
    const T v = live ? p[off] : T(0);
    if constexpr (RAW) {
        const T n2 = group_sum<T, G>(v * v), dot = group_sum<T, G>(v * x);
        return x - T(2) * dot * M<T>::rcp(n2) * v;
    } else {
        return x - v * group_sum<T, G>(v * x);
    }
}
template <typename T, int G, bool RAW> __device__ __forceinline__ T gfg_rotate_inv(const T* __restrict__ p, const GfLayerDev<T>& o, int D, bool live, T x) {
    for (int i = 0; i < o.hh; ++i) x = gfg_reflect<T, G, RAW>(p, o.off_rot + i * D, live, x);      // x <- Q^T x (:1038)
    return x;
}
template <typename T, int G, bool RAW> __device__ __forceinline__ T gfg_rotate_fwd(const T* __restrict__ p, const GfLayerDev<T>& o, int D, bool live, T x) {
    for (int i = o.hh - 1; i >= 0; --i) x = gfg_reflect<T, G, RAW>(p, o.off_rot + i * D, live, x);  // x <- Q x (:975)
    return x;
}

// ---------------------------------------------------------------------------------------------------------- mixture, one coordinate
// derived row (mean, 1/width, normalised pi): linear-space sums, log-space redo for tails (see jf_gf.h gf_mixture)
template <typename T> __device__ __forceinline__ MixQ<T> gfg_mixture(const T* __restrict__ p, const GfLayerDev<T>& o, int D, T x) {
    T C = T(0), S = T(0), P = T(0);
    const T uniform_w = M<T>::rcp(T(o.K));
#pragma unroll 2
    for (int k = 0; k < o.K; ++k) {
        const T mu = p[o.off_mean + k * D], iw = p[o.off_lw + k * D];
        const T wk = o.fit_norm ? p[o.off_ln + k * D] : uniform_w;
        const T u = (x - mu) * iw;
        const T t = M<T>::exp_fast(-M<T>::abs(u));
        const T hi = M<T>::rcp(T(1) + t);
        const T lo = t * hi;
        const bool pos = u >= T(0);
        C += wk * (pos ? hi : lo);
        S += wk * (pos ? lo : hi);
        P += wk * hi * lo * iw;
    }
    MixQ<T> q;
    if (C > M<T>::TINY && S > M<T>::TINY && P > M<T>::TINY) {
        q.lc = M<T>::log_fast(C); q.ls = M<T>::log_fast(S); q.lp = M<T>::log_fast(P);
        q.cdf = C; q.sf = S;
    } else {
        const Log3<T> r = gf_logspace_dim<T>(p + o.off_mean, p + o.off_lw, o.fit_norm ? p + o.off_ln : nullptr, o.K, D, x);
        q.lc = r.lc; q.ls = r.ls; q.lp = r.lp;
        q.cdf = M<T>::exp(q.lc); q.sf = M<T>::exp(q.ls);
    }
    return q;
}

// raw row: width / weight regulation fused into the loop (log-prob direction, every parameter is used exactly once)
template <typename T> __device__ __forceinline__ MixQ<T> gfg_mixture_raw(const T* __restrict__ p, const GfLayerDev<T>& o, int D, T x) {
    T C = T(0), S = T(0), P = T(0), Nn = T(0), shift = T(0);
    if (o.fit_norm && !o.reg_norm) {
        shift = p[o.off_ln];
        for (int k = 1; k < o.K; ++k) shift = M<T>::max(shift, p[o.off_ln + k * D]);
    }
#pragma unroll 2
    for (int k = 0; k < o.K; ++k) {
        const T mu = p[o.off_mean + k * D], lw = p[o.off_lw + k * D];
        const T iw = M<T>::rcp(gf_width(o, lw));
        const T wk = o.fit_norm ? gf_weight(o, p[o.off_ln + k * D], shift) : T(1);
        const T u = (x - mu) * iw;
        const T t = M<T>::exp_fast(-M<T>::abs(u));
        const T hi = M<T>::rcp(T(1) + t);
        const T lo = t * hi;
        const bool pos = u >= T(0);
        C += wk * (pos ? hi : lo);
        S += wk * (pos ? lo : hi);
        P += wk * hi * lo * iw;
        Nn += wk;
    }
    const T inv = M<T>::rcp(Nn);
    C *= inv; S *= inv; P *= inv;
    MixQ<T> q;
    q.lc = M<T>::log_fast(C); q.ls = M<T>::log_fast(S); q.lp = M<T>::log_fast(P);
    q.cdf = C; q.sf = S;
    const bool slow = !(C > M<T>::TINY && S > M<T>::TINY && P > M<T>::TINY);
    if (__any(slow)) {   // rare (tails): the reference's log-space arithmetic (:389-454), wave-uniform branch
        Lse<T> c, s, pp;
        const T lnN = M<T>::log(Nn);
        for (int k = 0; k < o.K; ++k) {
            const T w = gf_width(o, p[o.off_lw + k * D]);
            const T u = (x - p[o.off_mean + k * D]) / w;
            const T sp = softplus(-u);
            const T lnpi = (o.fit_norm ? M<T>::log(gf_weight(o, p[o.off_ln + k * D], shift)) : T(0)) - lnN;
            c.add(-sp + lnpi);
            s.add(-u - sp + lnpi);
            pp.add(-u - M<T>::log(w) - T(2) * sp + lnpi);
        }
        if (slow) {
            q.lc = c.value(); q.ls = s.value(); q.lp = pp.value();
            q.cdf = M<T>::exp(q.lc); q.sf = M<T>::exp(q.ls);
        }
    }
    return q;
}

// in-place derive of a staged raw row by the G lanes of its group: lane d < D takes column d, reflections are dealt round-robin
template <typename T, int G> __device__ __forceinline__ void gfg_derive(T* __restrict__ row, const GfLayerDev<T>& o, int D, int g) {
    if (g < D) gf_derive_column<T>(row, o, D, g);
    for (int i = g; i < o.hh; i += G) gf_derive_reflection<T>(row, o, D, i);
}

// ---------------------------------------------------------------------------------------------------------- sampling direction
// bisection + Newton (layers/bisection_n_newton.py:11-135, called with 25 / 20 iterations on [-1e5, 1e5], :921) for one coordinate;
// the Newton stopping rule sums |update| over the row's coordinates (group butterfly), so the D lanes of a row stop together.
template <typename T, int G> __device__ __forceinline__ T gfg_solve(const T* __restrict__ p, const GfLayerDev<T>& o, int D, bool live, T z,
                                                                     bool row_valid, bool leader, int32_t* status) {
    T lo = T(-1e5), hi = T(1e5), x = T(0);
    for (int it = 0; it < 25; ++it) {
        x = (hi + lo) * T(0.5);
        const T y = gf_icdf<T>(o.inv_type, gfg_mixture<T>(p, o, D, x)).y;
        const bool ok = M<T>::abs(y - z) <= T(1e-6) * M<T>::abs(z);
        if (ok) { lo = x; hi = x; }
        else if (y < z) lo = x;
        else hi = x;
    }
    bool active = row_valid;
    T ferr = T(0);
    bool nonfinite = false;
    for (int it = 0; it < 20 && __any(active); ++it) {
        const IcdfOut<T> s = gf_icdf<T>(o.inv_type, gfg_mixture<T>(p, o, D, x));
        const T f = s.y - z;
        const T upd = f / M<T>::exp(s.logd);
        const T usum = group_sum<T, G>(live ? M<T>::abs(upd) : T(0));
        if (active) {
            const T nx = x - upd;
            if (M<T>::finite(nx)) x = nx; else nonfinite = nonfinite || live;     // keep the previous iterate (:84-91)
            ferr = M<T>::abs(f);
            active = usum >= T(1e-14);
        }
    }
    const T prec = sizeof(T) == 8 ? T(1e-7) : T(1e-4);
    const T ferr_row = group_max<T, G>(live ? ferr : T(0));
    const T nf_row = group_max<T, G>(nonfinite ? T(1) : T(0));
    status_add(status, JF_STATUS_NONCONVERGED, row_valid && leader && (ferr_row > prec));
    status_add(status, JF_STATUS_NONFINITE, row_valid && leader && (nf_row > T(0)));
    return x;
}

}  // namespace jf
