// Device code of the Gaussianization-flow layer 'g' (logistic-mixture CDF + inverse-CDF stage + Householder rotation).
// Restates jammy_flows/layers/euclidean/gaussianization_flow.py:389-454 (mixture), :480-671 (inverse-CDF stage and its
// log-derivative), :699-861 (parameter regulation), :457-471 (Householder) for a lane-per-sample CDNA4 kernel.
//
// Data flow per layer and lane:   raw parameter row (LDS)  --derive-->  (mean, 1/width, pi_k) row (same LDS slots)
//                                 x[D] --offset, reflections--> mixture sums in LINEAR space (1 v_exp + 1 v_rcp per (k,d))
//                                 --> log cdf / log sf / log pdf --> inverse-CDF stage.
// A (lane, d) whose cdf or sf falls below M<T>::TINY is redone in log space (online log-sum-exp), i.e. exactly the
// reference's arithmetic; that branch is what the +-50 sigma rows of the golden fixtures exercise.
#pragma once
#include "jf_common.h"
#include "jf_math.h"

namespace jf {

template <typename T> struct GfLayerDev {
    int K, hh, model_offset, fit_norm, reg_norm, inv_type, width_mode, clamp_widths;
    int n_params;                            // raw row length of this layer
    int col0;                                // first column of the layer inside the chain's parameter row
    int off_rot, off_mean, off_lw, off_ln;   // section offsets inside the layer row (elements)
    int vec_ok;                              // 16-byte staging possible
    T wmin, wmax, inv_wmax, nmin, nmax, lw_lo, lw_hi;
};

constexpr double PADE_BOUND = 0.5e-7;   // gaussianization_flow.py:140
constexpr double PADE_A = 0.147;        // gaussianization_flow.py:143

// ----------------------------------------------------------------------------------------------------------
// parameter regulation (gaussianization_flow.py:269-317, 342, 406) in closed linear form:
//   smooth saturation  logw' = LSE( ln wmax - softplus(ln wmax - x), ln wmin )   <=>   w = wmin + 1 / (1/wmax + e^-x)
//   norm regulator     logn' = LSE( ln nmax - softplus(-x), ln nmin )            <=>   n = nmin + nmax / (1 + e^-x)
// ----------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T gf_width(const GfLayerDev<T>& o, T x) {
    if (o.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION) {
        if (o.clamp_widths) x = clampv(x, o.lw_lo, o.lw_hi);
        return o.wmin + M<T>::rcp(o.inv_wmax + M<T>::exp_fast(-x));
    }
    if (o.clamp_widths) x = clampv(x, o.lw_lo, o.lw_hi);
    if (o.width_mode == JF_GF_WIDTH_EXP) return M<T>::exp(x) + o.wmin;
    return softplus(x) + o.wmin;
}

// derive one column d of a layer row in place:  log_width slot -> 1/width,  log_weight slot -> normalised pi_k
template <typename T> __device__ __forceinline__ void gf_derive_column(T* __restrict__ row, const GfLayerDev<T>& o, int D, int d) {
    const int K = o.K;
    T nsum = T(0), nmax = T(0);
    if (o.fit_norm && !o.reg_norm) {   // unbounded log-weights: shift by the max before exponentiating
        nmax = row[o.off_ln + d];
        for (int k = 1; k < K; ++k) nmax = M<T>::max(nmax, row[o.off_ln + k * D + d]);
    }
    for (int k = 0; k < K; ++k) {
        const int i = k * D + d;
        row[o.off_lw + i] = M<T>::rcp(gf_width(o, row[o.off_lw + i]));
        if (o.fit_norm) {
            const T xn = row[o.off_ln + i];
            const T w = o.reg_norm ? o.nmin + o.nmax * M<T>::rcp(T(1) + M<T>::exp_fast(-xn)) : M<T>::exp(xn - nmax);
            row[o.off_ln + i] = w;
            nsum += w;
        }
    }
    if (o.fit_norm) {
        const T inv = M<T>::rcp(nsum);
        for (int k = 0; k < K; ++k) row[o.off_ln + k * D + d] *= inv;
    }
}

// pi_k weight before normalisation (gaussianization_flow.py:342, 406)
template <typename T> __device__ __forceinline__ T gf_weight(const GfLayerDev<T>& o, T xn, T shift) {
    return o.reg_norm ? o.nmin + o.nmax * M<T>::rcp(T(1) + M<T>::exp_fast(-xn)) : M<T>::exp(xn - shift);
}

// derive a whole row in place with D-wide vector LDS accesses and D independent dependency chains per k
template <typename T, int D> __device__ __forceinline__ void gf_derive_row(T* __restrict__ row, const GfLayerDev<T>& o) {
    const int K = o.K;
    T shift[D], nsum[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { shift[d] = T(0); nsum[d] = T(0); }
    if (o.fit_norm && !o.reg_norm) {
        load_d<T, D>(row + o.off_ln, shift);
        for (int k = 1; k < K; ++k) {
            T v[D];
            load_d<T, D>(row + o.off_ln + k * D, v);
#pragma unroll
            for (int d = 0; d < D; ++d) shift[d] = M<T>::max(shift[d], v[d]);
        }
    }
    for (int k = 0; k < K; ++k) {
        T lw[D], ln[D];
        load_d<T, D>(row + o.off_lw + k * D, lw);
#pragma unroll
        for (int d = 0; d < D; ++d) lw[d] = M<T>::rcp(gf_width(o, lw[d]));
        store_d<T, D>(row + o.off_lw + k * D, lw);
        if (o.fit_norm) {
            load_d<T, D>(row + o.off_ln + k * D, ln);
#pragma unroll
            for (int d = 0; d < D; ++d) { ln[d] = gf_weight(o, ln[d], shift[d]); nsum[d] += ln[d]; }
            store_d<T, D>(row + o.off_ln + k * D, ln);
        }
    }
    if (o.fit_norm) {
#pragma unroll
        for (int d = 0; d < D; ++d) nsum[d] = M<T>::rcp(nsum[d]);
        for (int k = 0; k < K; ++k) {
            T ln[D];
            load_d<T, D>(row + o.off_ln + k * D, ln);
#pragma unroll
            for (int d = 0; d < D; ++d) ln[d] *= nsum[d];
            store_d<T, D>(row + o.off_ln + k * D, ln);
        }
    }
    for (int i = 0; i < o.hh; ++i) {
        T v[D];
        load_d<T, D>(row + o.off_rot + i * D, v);
        T n2 = T(0);
#pragma unroll
        for (int d = 0; d < D; ++d) n2 += v[d] * v[d];
        const T sc = M<T>::SQRT2 / M<T>::sqrt(n2);
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] *= sc;
        store_d<T, D>(row + o.off_rot + i * D, v);
    }
}

// Householder vector i -> sqrt(2) v/|v| so that a reflection is x -= v (v.x)   (H = I - 2 v v^T/|v|^2)
template <typename T> __device__ __forceinline__ void gf_derive_reflection(T* __restrict__ row, const GfLayerDev<T>& o, int D, int i) {
    T n2 = T(0);
    for (int d = 0; d < D; ++d) { const T v = row[o.off_rot + i * D + d]; n2 += v * v; }
    const T s = M<T>::SQRT2 / M<T>::sqrt(n2);
    for (int d = 0; d < D; ++d) row[o.off_rot + i * D + d] *= s;
}

template <typename T, int D> __device__ __forceinline__ void gf_reflect(const T* __restrict__ v, T (&x)[D]) {
    T vv[D];
    load_d<T, D>(v, vv);
    T dot = T(0);
#pragma unroll
    for (int d = 0; d < D; ++d) dot += vv[d] * x[d];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] -= vv[d] * dot;
}

// x <- Q^T x  (inverse / log-prob direction: H_0 first)   gaussianization_flow.py:1038
template <typename T, int D> __device__ __forceinline__ void gf_rotate_inv(const T* __restrict__ row, const GfLayerDev<T>& o, T (&x)[D]) {
    for (int i = 0; i < o.hh; ++i) gf_reflect<T, D>(row + o.off_rot + i * D, x);
}
// x <- Q x  (sampling direction: H_{n-1} first)            gaussianization_flow.py:975
template <typename T, int D> __device__ __forceinline__ void gf_rotate_fwd(const T* __restrict__ row, const GfLayerDev<T>& o, T (&x)[D]) {
    for (int i = o.hh - 1; i >= 0; --i) gf_reflect<T, D>(row + o.off_rot + i * D, x);
}

// ----------------------------------------------------------------------------------------------------------
// mixture quantities
// ----------------------------------------------------------------------------------------------------------
template <typename T> struct Lse {   // online log-sum-exp
    T m, s;
    __device__ __forceinline__ Lse() : m(-INFINITY), s(T(0)) {}
    __device__ __forceinline__ void add(T a) {
        if (a > m) { s = s * M<T>::exp(m - a) + T(1); m = a; }
        else s += M<T>::exp(a - m);
    }
    __device__ __forceinline__ T value() const { return m + M<T>::log(s); }
};

// faithful log-space evaluation of one dimension (gaussianization_flow.py:389-454) from a derived row
template <typename T> struct Log3 { T lc, ls, lp; };
template <typename T> __device__ __noinline__ Log3<T> gf_logspace_dim(const T* __restrict__ mean, const T* __restrict__ invw, const T* __restrict__ pi,
                                                                     int K, int D, T xd) {
    Lse<T> c, s, p;
    const T uniform_ln = -M<T>::log(T(K));
    for (int k = 0; k < K; ++k) {
        const T iw = invw[k * D];
        const T u = (xd - mean[k * D]) * iw;
        const T sp = softplus(-u);
        const T lnpi = pi ? M<T>::log(pi[k * D]) : uniform_ln;
        c.add(-sp + lnpi);
        s.add(-u - sp + lnpi);
        p.add(-u + M<T>::log(iw) - T(2) * sp + lnpi);
    }
    Log3<T> r;
    r.lc = c.value();
    r.ls = s.value();
    r.lp = p.value();
    return r;
}

template <typename T> struct MixQ { T lc, ls, lp, cdf, sf; };   // log cdf, log sf, log pdf, cdf, sf of one dimension

// all D dimensions at once, linear space with per-(lane,d) log-space fallback
template <typename T, int D> __device__ __forceinline__ void gf_mixture(const T* __restrict__ row, const GfLayerDev<T>& o, const T (&x)[D], MixQ<T> (&q)[D]) {
    T C[D], S[D], P[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { C[d] = T(0); S[d] = T(0); P[d] = T(0); }
    const T uniform_w = M<T>::rcp(T(o.K));
    for (int k = 0; k < o.K; ++k) {
        T mu[D], iw[D], w[D];
        load_d<T, D>(row + o.off_mean + k * D, mu);
        load_d<T, D>(row + o.off_lw + k * D, iw);
        if (o.fit_norm) load_d<T, D>(row + o.off_ln + k * D, w);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const T u = (x[d] - mu[d]) * iw[d];
            const T t = M<T>::exp_fast(-M<T>::abs(u));
            const T hi = M<T>::rcp(T(1) + t);      // sigma(|u|)
            const T lo = t * hi;                   // sigma(-|u|)
            const T wk = o.fit_norm ? w[d] : uniform_w;
            const bool pos = u >= T(0);
            C[d] += wk * (pos ? hi : lo);
            S[d] += wk * (pos ? lo : hi);
            P[d] += wk * hi * lo * iw[d];
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (C[d] > M<T>::TINY && S[d] > M<T>::TINY && P[d] > M<T>::TINY) {
            q[d].lc = M<T>::log_fast(C[d]);
            q[d].ls = M<T>::log_fast(S[d]);
            q[d].lp = M<T>::log_fast(P[d]);
            q[d].cdf = C[d];
            q[d].sf = S[d];
        } else {
            const Log3<T> r = gf_logspace_dim<T>(row + o.off_mean + d, row + o.off_lw + d, o.fit_norm ? row + o.off_ln + d : nullptr, o.K, D, x[d]);
            q[d].lc = r.lc; q[d].ls = r.ls; q[d].lp = r.lp;
            q[d].cdf = M<T>::exp(q[d].lc);
            q[d].sf = M<T>::exp(q[d].ls);
        }
    }
}

// Log-prob direction, per-sample regime: the lane's row holds RAW parameters; width / weight regulation is fused into the
// mixture loop (no LDS write-back, D independent chains per k).  Reflections use the raw Householder vectors.
template <typename T, int D> __device__ __forceinline__ void gf_rotate_inv_raw(const T* __restrict__ row, const GfLayerDev<T>& o, T (&x)[D]) {
    for (int i = 0; i < o.hh; ++i) {
        T v[D];
        load_d<T, D>(row + o.off_rot + i * D, v);
        T n2 = T(0), dot = T(0);
#pragma unroll
        for (int d = 0; d < D; ++d) { n2 += v[d] * v[d]; dot += v[d] * x[d]; }
        const T f = T(2) * dot * M<T>::rcp(n2);
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] -= f * v[d];
    }
}

template <typename T, int D> __device__ __forceinline__ void gf_mixture_raw(const T* __restrict__ row, const GfLayerDev<T>& o, const T (&x)[D], MixQ<T> (&q)[D]) {
    T C[D], S[D], P[D], Nn[D], shift[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { C[d] = T(0); S[d] = T(0); P[d] = T(0); Nn[d] = T(0); shift[d] = T(0); }
    if (o.fit_norm && !o.reg_norm) {
        load_d<T, D>(row + o.off_ln, shift);
        for (int k = 1; k < o.K; ++k) {
            T v[D];
            load_d<T, D>(row + o.off_ln + k * D, v);
#pragma unroll
            for (int d = 0; d < D; ++d) shift[d] = M<T>::max(shift[d], v[d]);
        }
    }
    for (int k = 0; k < o.K; ++k) {
        T mu[D], lw[D], ln[D];
        load_d<T, D>(row + o.off_mean + k * D, mu);
        load_d<T, D>(row + o.off_lw + k * D, lw);
        if (o.fit_norm) load_d<T, D>(row + o.off_ln + k * D, ln);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const T iw = M<T>::rcp(gf_width(o, lw[d]));
            const T wk = o.fit_norm ? gf_weight(o, ln[d], shift[d]) : T(1);
            const T u = (x[d] - mu[d]) * iw;
            const T t = M<T>::exp_fast(-M<T>::abs(u));
            const T hi = M<T>::rcp(T(1) + t);
            const T lo = t * hi;
            const bool pos = u >= T(0);
            C[d] += wk * (pos ? hi : lo);
            S[d] += wk * (pos ? lo : hi);
            P[d] += wk * hi * lo * iw;
            Nn[d] += wk;
        }
    }
    bool slow = false;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const T inv = M<T>::rcp(Nn[d]);
        C[d] *= inv; S[d] *= inv; P[d] *= inv;
        q[d].lc = M<T>::log_fast(C[d]);
        q[d].ls = M<T>::log_fast(S[d]);
        q[d].lp = M<T>::log_fast(P[d]);
        q[d].cdf = C[d];
        q[d].sf = S[d];
        slow = slow || !(C[d] > M<T>::TINY && S[d] > M<T>::TINY && P[d] > M<T>::TINY);
    }
    if (__any(slow)) {   // rare (tails): redo the flagged dimensions in log space = the reference's arithmetic (:389-454)
        Lse<T> c[D], s[D], p[D];
        for (int k = 0; k < o.K; ++k) {
            T mu[D], lw[D], ln[D];
            load_d<T, D>(row + o.off_mean + k * D, mu);
            load_d<T, D>(row + o.off_lw + k * D, lw);
            if (o.fit_norm) load_d<T, D>(row + o.off_ln + k * D, ln);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const T w = gf_width(o, lw[d]);
                const T u = (x[d] - mu[d]) / w;
                const T sp = softplus(-u);
                const T lnpi = (o.fit_norm ? M<T>::log(gf_weight(o, ln[d], shift[d])) : T(0)) - M<T>::log(Nn[d]);
                c[d].add(-sp + lnpi);
                s[d].add(-u - sp + lnpi);
                p[d].add(-u - M<T>::log(w) - T(2) * sp + lnpi);
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const bool bad = !(C[d] > M<T>::TINY && S[d] > M<T>::TINY && P[d] > M<T>::TINY);
            if (bad) {
                q[d].lc = c[d].value(); q[d].ls = s[d].value(); q[d].lp = p[d].value();
                q[d].cdf = M<T>::exp(q[d].lc);
                q[d].sf = M<T>::exp(q[d].ls);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------
// inverse-CDF stage  (gaussianization_flow.py:480-560 value, :568-671 log-derivative)
// ----------------------------------------------------------------------------------------------------------
template <typename T> struct Pade { T F2, F2mF; };   // F2 = sqrt(F^2 - ln_fac/a),  F2mF = F2 - F (cancellation-free)

template <typename T> __device__ __forceinline__ Pade<T> pade_terms(const MixQ<T>& q) {
    const T a = T(PADE_A);
    const T c = T(2.0 / (3.14159265358979323846 * PADE_A));
    // ln(4 cdf sf) = log1p(-(sf-cdf)^2) in the centre (no cancellation), log-space sum in the tails
    const T dlt = q.sf - q.cdf;
    const T ln_fac = (M<T>::min(q.cdf, q.sf) > T(0.01)) ? M<T>::log1p(-dlt * dlt) : q.lc + q.ls + T(1.38629436111989061883);
    const T F = ln_fac * T(0.5) + c;
    const T rad = -ln_fac / a;
    Pade<T> p;
    p.F2 = M<T>::sqrt(F * F + rad);
    p.F2mF = F > T(0) ? rad / (p.F2 + F) : p.F2 - F;
    return p;
}
template <typename T> __device__ __forceinline__ T pade_value(const Pade<T>& p) {       // sqrt(2 (F2 - F)), clamped at 0 (:517-522)
    return M<T>::sqrt(M<T>::max(T(2) * p.F2mF, T(0)));
}
template <typename T> __device__ __forceinline__ T pade_logderiv(const Pade<T>& p, const MixQ<T>& q) {   // (:597-619) without + log_pdf
    const T log_num = M<T>::log(p.F2mF + T(1.0 / PADE_A));
    const T log_den = T(1.03972077083991796413) + T(0.5) * M<T>::log(p.F2mF) + M<T>::log(p.F2);   // 0.5 ln 8
    return log_num - log_den - q.ls - q.lc + M<T>::log(M<T>::abs(q.sf - q.cdf));
}

// returns y, writes the log-derivative d y / d x
template <typename T> __device__ __forceinline__ T gf_inverse_cdf(int inv_type, const MixQ<T>& q, T& logd) {
    if (inv_type == JF_GF_ISIGMOID) {
        logd = q.lp - q.lc - q.ls;          // = LSE(-log sf, -log cdf) + log pdf   since cdf + sf = 1
        return q.lc - q.ls;
    }
    const T bound = T(PADE_BOUND);
    if (inv_type == JF_GF_INORMAL_FULL_PADE) {
        const Pade<T> p = pade_terms(q);
        const T tot = pade_value(p);
        const bool centre = (q.cdf > T(0.49999)) && (q.cdf < T(0.50001));
        logd = centre ? T(0.91893852361801185) + q.lp : pade_logderiv(p, q) + q.lp;     // ln 2.506628 (:654)
        return q.cdf <= q.sf ? -tot : tot;
    }
    const bool left = q.cdf <= bound, right = q.sf <= bound;
    if (!left && !right) {   // central region: exact inverse normal CDF, evaluated from the smaller of cdf / sf
        const T e = M<T>::erfcinv(T(2) * M<T>::min(q.cdf, q.sf));
        logd = M<T>::HALF_LN_2PI + e * e + q.lp;
        return (q.cdf < q.sf ? -M<T>::SQRT2 : M<T>::SQRT2) * e;
    }
    T tot;
    if (inv_type == JF_GF_INORMAL_PARTLY_CRUDE) {
        const T lsum = q.lc + q.ls;
        tot = M<T>::sqrt(T(-2) * lsum) - T(0.4717);
        logd = T(-0.5) * M<T>::log(T(-2) * lsum) - lsum + q.lp;
    } else {
        const Pade<T> p = pade_terms(q);
        tot = pade_value(p);
        logd = pade_logderiv(p, q) + q.lp;
    }
    return right ? tot : -tot;
}

// Call form used by the lane = (row, coordinate) kernels.  In float64 the stage is an out-of-line function: inlined, the several dozen
// double-precision polynomial constants of erfcinv / log / log1p / sqrt are hoisted out of the layer loop and pin > 250 VGPRs.
template <typename T> struct IcdfOut { T y, logd; };
template <typename T> __device__ __forceinline__ IcdfOut<T> gf_icdf(int inv_type, MixQ<T> q) {
    IcdfOut<T> r;
    r.y = gf_inverse_cdf<T>(inv_type, q, r.logd);
    return r;
}
template <> __device__ __noinline__ IcdfOut<double> gf_icdf<double>(int inv_type, MixQ<double> q) {
    IcdfOut<double> r;
    r.y = gf_inverse_cdf<double>(inv_type, q, r.logd);
    return r;
}

// one full layer evaluation at x (already offset-shifted and rotated): y[d], sum_d log dy/dx
template <typename T, int D> __device__ __forceinline__ T gf_stage(const T* __restrict__ row, const GfLayerDev<T>& o, const T (&x)[D], T (&y)[D]) {
    MixQ<T> q[D];
    gf_mixture<T, D>(row, o, x, q);
    T sum = T(0);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        T ld;
        y[d] = gf_inverse_cdf<T>(o.inv_type, q[d], ld);
        sum += ld;
    }
    return sum;
}
template <typename T, int D> __device__ __forceinline__ T gf_stage_raw(const T* __restrict__ row, const GfLayerDev<T>& o, const T (&x)[D], T (&y)[D]) {
    MixQ<T> q[D];
    gf_mixture_raw<T, D>(row, o, x, q);
    T sum = T(0);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        T ld;
        y[d] = gf_inverse_cdf<T>(o.inv_type, q[d], ld);
        sum += ld;
    }
    return sum;
}
template <typename T, int D> __device__ __forceinline__ void gf_stage_deriv(const T* __restrict__ row, const GfLayerDev<T>& o, const T (&x)[D], T (&y)[D], T (&logd)[D]) {
    MixQ<T> q[D];
    gf_mixture<T, D>(row, o, x, q);
#pragma unroll
    for (int d = 0; d < D; ++d) y[d] = gf_inverse_cdf<T>(o.inv_type, q[d], logd[d]);
}

// ----------------------------------------------------------------------------------------------------------
// sampling direction: bisection + Newton per row (layers/bisection_n_newton.py:11-135; called with 25 / 20 iterations
// and [-1e5, 1e5] from gaussianization_flow.py:921).  A row keeps iterating while sum_d |update| >= 1e-14; the wave leaves
// the Newton loop when no lane is active any more (ballot).
// ----------------------------------------------------------------------------------------------------------
template <typename T, int D> __device__ __forceinline__ void gf_solve(const T* __restrict__ row, const GfLayerDev<T>& o, const T (&z)[D], T (&x)[D],
                                                             bool lane_valid, int32_t* status) {
    T lo[D], hi[D], y[D], logd[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { lo[d] = T(-1e5); hi[d] = T(1e5); x[d] = T(0); }
    for (int it = 0; it < 25; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = (hi[d] + lo[d]) * T(0.5);
        gf_stage_deriv<T, D>(row, o, x, y, logd);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const bool ok = M<T>::abs(y[d] - z[d]) <= T(1e-6) * M<T>::abs(z[d]);
            const bool right = y[d] < z[d];
            if (ok) { lo[d] = x[d]; hi[d] = x[d]; }
            else if (right) lo[d] = x[d];
            else hi[d] = x[d];
        }
    }
    bool active = lane_valid;
    T ferr = T(0);
    bool nonfinite = false;
    for (int it = 0; it < 20 && __any(active); ++it) {
        gf_stage_deriv<T, D>(row, o, x, y, logd);
        if (active) {
            T usum = T(0);
            ferr = T(0);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const T f = y[d] - z[d];
                const T upd = f / M<T>::exp(logd[d]);
                const T nx = x[d] - upd;
                if (M<T>::finite(nx)) x[d] = nx; else nonfinite = true;      // keep the previous iterate (:84-91)
                usum += M<T>::abs(upd);
                ferr = M<T>::max(ferr, M<T>::abs(f));
            }
            active = usum >= T(1e-14);
        }
    }
    const T prec = sizeof(T) == 8 ? T(1e-7) : T(1e-4);
    status_add(status, JF_STATUS_NONCONVERGED, lane_valid && (ferr > prec));
    status_add(status, JF_STATUS_NONFINITE, lane_valid && nonfinite);
}

}  // namespace jf
