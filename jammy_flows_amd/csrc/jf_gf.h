// Device code of the Gaussianization-flow layer 'g' (logistic-mixture CDF + inverse-CDF stage + Householder rotation).
// Restates jammy_flows/layers/euclidean/gaussianization_flow.py:389-454 (mixture), :480-671 (inverse-CDF stage and its
// log-derivative), :699-861 (parameter regulation), :457-471 (Householder) for a lane-per-sample CDNA4 kernel.
//
// Work distribution: lane = (row, coordinate).  A wave holds 64/G rows; the G = next-power-of-two(D) neighbouring lanes of a group own
// the D coordinates of one row (lanes g >= D shadow coordinate D-1 and never store).  Everything per coordinate (mixture sums,
// inverse-CDF stage, bisection) is scalar code per lane, so the register footprint does not grow with D; the three places the
// reference reduces over the coordinates (Householder dot products, sum of log-derivatives, the Newton stopping rule) are DPP
// butterflies inside the group.
//
// Data flow per layer and lane:   parameter row of the lane's sample (LDS)
//                                 x --offset, reflections--> mixture sums in LINEAR space (1 v_exp + 1 v_rcp per k)
//                                 --> log cdf / log sf / log pdf --> inverse-CDF stage.
// When a cdf, sf or pdf of some lane underflows (M<T>::TINY) the wave re-evaluates the mixture with every sum scaled by e^{m},
// m = min_k |u_k| (gfg_mixture_scaled): same relative accuracy as the reference's log-sum-exp arithmetic at any distance from the
// components; that branch is what the +-50 sigma rows of the golden fixtures exercise.
#pragma once
#include "jf_common.h"
#include "jf_math.h"
#include "jf_spline.h"

namespace jf {

template <typename T> struct GfLayerDev {
    int K, hh, model_offset, fit_norm, reg_norm, inv_type, width_mode, clamp_widths;
    int fast;                                // all options at the reference's defaults (smooth-saturation widths without clamping, fitted and
                                             //   regulated weights): the mixture loop then runs a branch-free specialisation
    int stretch;                             // JF_GF_STRETCH_*: for RQ_SPLINES off_mean / off_lw / off_ln hold the log_w / log_h / log_d sections
    int off_box;                             //   (each d-major, K / K / K+1 values per coordinate) and off_box the 4 box values per coordinate
    int n_params;                            // raw row length of this layer
    int col0;                                // first column of the layer inside the chain's parameter row
    int off_rot, off_mean, off_lw, off_ln;   // section offsets inside the layer row (elements)
    int vec_ok;                              // 16-byte staging possible
    int rot_mode, center_mean, skew;         // JF_GF_ROT_* / center_mean / add_skewness: any of them non-zero -> general-option kernel (jf_gf_ext.h)
    int off_skew;                            // log-exponent section of the skewed components
    T wmin, wmax, inv_wmax, nmin, nmax, lw_lo, lw_hi;
};

constexpr double PADE_BOUND = 0.5e-7;   // gaussianization_flow.py:140
constexpr double PADE_A = 0.147;        // gaussianization_flow.py:143

// ----------------------------------------------------------------------------------------------------------
// parameter regulation (gaussianization_flow.py:269-317, 342, 406) in closed linear form:
//   smooth saturation  logw' = LSE( ln wmax - softplus(ln wmax - x), ln wmin )   <=>   w = wmin + 1 / (1/wmax + e^-x)
//   norm regulator     logn' = LSE( ln nmax - softplus(-x), ln nmin )            <=>   n = nmin + nmax / (1 + e^-x)
// ----------------------------------------------------------------------------------------------------------
// 1 / width directly: smooth saturation  1/w = (a + e) / (wmin (a + e) + 1),  a = 1/wmax, e = e^-x   (one rcp instead of two)
template <typename T> __device__ __forceinline__ T gf_inv_width(const GfLayerDev<T>& o, T x) {
    if (o.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION) {
        if (o.clamp_widths) x = clampv(x, o.lw_lo, o.lw_hi);
        const T ae = o.inv_wmax + M<T>::exp_fast(-x);
        return ae * M<T>::rcp(o.wmin * ae + T(1));
    }
    if (o.clamp_widths) x = clampv(x, o.lw_lo, o.lw_hi);
    if (o.width_mode == JF_GF_WIDTH_EXP) return M<T>::rcp(M<T>::exp(x) + o.wmin);
    return M<T>::rcp(softplus(x) + o.wmin);
}
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

// Householder vector i -> sqrt(2) v/|v| so that a reflection is x -= v (v.x)   (H = I - 2 v v^T/|v|^2)
template <typename T> __device__ __forceinline__ void gf_derive_reflection(T* __restrict__ row, const GfLayerDev<T>& o, int D, int i) {
    T n2 = T(0);
    for (int d = 0; d < D; ++d) { const T v = row[o.off_rot + i * D + d]; n2 += v * v; }
    const T s = M<T>::SQRT2 / M<T>::sqrt(n2);
    for (int d = 0; d < D; ++d) row[o.off_rot + i * D + d] *= s;
}

template <typename T> struct MixQ { T lc, ls, lp, cdf, sf; };   // log cdf, log sf, log pdf, cdf, sf of one coordinate

// ----------------------------------------------------------------------------------------------------------
// inverse-CDF stage  (gaussianization_flow.py:480-560 value, :568-671 log-derivative)
// ----------------------------------------------------------------------------------------------------------
template <typename T> struct Pade { T F2, F2mF; };   // F2 = sqrt(F^2 - ln_fac/a),  F2mF = F2 - F (cancellation-free)

template <typename T> __device__ __forceinline__ Pade<T> pade_terms(const MixQ<T>& q) {
    const T a = T(PADE_A);
    const T c = T(2.0 / (3.14159265358979323846 * PADE_A));
    // ln(4 cdf sf) = log1p(-(sf-cdf)^2) in the centre (no cancellation), log-space sum in the tails
    const T dlt = q.sf - q.cdf;
    Pade<T> p;
    if constexpr (sizeof(T) == 4) {
        // float32: hardware sqrt / rcp (1 ulp) and the log1p only for waves that hold a centre lane at all -- the tail branch of the partly
        // precise inverse (the only caller with rows beyond the bound) never does, and the select below would otherwise evaluate the ~25
        // instructions of log1pf for every lane.  (Round 3: this stage had grown to half of the flow's vector instructions on inputs whose
        // waves mix centre and tail rows -- both sides of the region branch run there.)
        const bool centre = M<T>::min(q.cdf, q.sf) > T(0.01);
        T ln_fac = q.lc + q.ls + T(1.38629436111989061883);
        if (__any(centre)) {
            const T l1p = M<T>::log1p(-dlt * dlt);
            ln_fac = centre ? l1p : ln_fac;
        }
        const T F = ln_fac * T(0.5) + c;
        const T rad = -ln_fac * T(1.0 / PADE_A);
        p.F2 = M<T>::sqrt_fast(F * F + rad);
        p.F2mF = F > T(0) ? rad * M<T>::rcp(p.F2 + F) : p.F2 - F;
        return p;
    }
    const T ln_fac = (M<T>::min(q.cdf, q.sf) > T(0.01)) ? M<T>::log1p(-dlt * dlt) : q.lc + q.ls + T(1.38629436111989061883);
    const T F = ln_fac * T(0.5) + c;
    const T rad = -ln_fac / a;
    p.F2 = M<T>::sqrt(F * F + rad);
    p.F2mF = F > T(0) ? rad / (p.F2 + F) : p.F2 - F;
    return p;
}
template <typename T> __device__ __forceinline__ T pade_value(const Pade<T>& p) {       // sqrt(2 (F2 - F)), clamped at 0 (:517-522)
    if constexpr (sizeof(T) == 4) return M<T>::sqrt_fast(M<T>::max(T(2) * p.F2mF, T(0)));
    return M<T>::sqrt(M<T>::max(T(2) * p.F2mF, T(0)));
}
template <typename T> __device__ __forceinline__ T pade_logderiv(const Pade<T>& p, const MixQ<T>& q) {   // (:597-619) without + log_pdf
    if constexpr (sizeof(T) == 4) {
        // v_log_f32 (1 ulp on normal inputs; all three arguments are: F2mF >= 3e-10 outside the pinned centre window, F2 >= sqrt(rad))
        const T log_num = M<T>::log_fast(p.F2mF + T(1.0 / PADE_A));
        const T log_den = T(1.03972077083991796413) + T(0.5) * M<T>::log_fast(p.F2mF) + M<T>::log_fast(p.F2);
        return log_num - log_den - q.ls - q.lc + M<T>::log_fast(M<T>::abs(q.sf - q.cdf));
    }
    const T log_num = M<T>::log(p.F2mF + T(1.0 / PADE_A));
    const T log_den = T(1.03972077083991796413) + T(0.5) * M<T>::log(p.F2mF) + M<T>::log(p.F2);   // 0.5 ln 8
    return log_num - log_den - q.ls - q.lc + M<T>::log(M<T>::abs(q.sf - q.cdf));
}

// float32 inverse normal CDF of the central region (5e-8 < cdf, sf):  y = sqrt(2) erfinv(cdf - sf) with M. Giles' single-precision
// erfinv ("Approximating the erfinv function", GPU Computing Gems 2011): erfinv(x) = x p(w), w = -ln((1-x)(1+x)) = -ln(4 cdf sf), which
// the mixture already delivers without cancellation as -(log cdf + log sf + ln 4).  Both branches of the approximation are 9-term Horner
// forms, evaluated branch-free with selected coefficients.  |error| < 2e-6 over the region (checked against scipy ndtri), ~25 VALU
// instructions where OCML erfcinvf needs ~140.
__device__ __forceinline__ float inormal_central_f32(const MixQ<float>& q) {
    const float w = fmaxf(-(q.lc + q.ls + 1.38629436112f), 0.0f);
    const bool c = w < 5.0f;
    const float t = c ? w - 2.5f : M<float>::sqrt_fast(w) - 3.0f;
    float p = c ? 2.81022636e-08f : -0.000200214257f;
    p = fmaf(p, t, c ? 3.43273939e-07f : 0.000100950558f);
    p = fmaf(p, t, c ? -3.5233877e-06f : 0.00134934322f);
    p = fmaf(p, t, c ? -4.39150654e-06f : -0.00367342844f);
    p = fmaf(p, t, c ? 0.00021858087f : 0.00573950773f);
    p = fmaf(p, t, c ? -0.00125372503f : -0.0076224613f);
    p = fmaf(p, t, c ? -0.00417768164f : 0.00943887047f);
    p = fmaf(p, t, c ? 0.246640727f : 1.00167406f);
    p = fmaf(p, t, c ? 1.50140941f : 2.83297682f);
    return 1.41421356237f * p * (q.cdf - q.sf);
}
template <typename T> __device__ __forceinline__ T inormal_central_f32(const MixQ<T>&) { return T(0); }   // never called for double

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
    if (!left && !right) {   // central region: exact inverse normal CDF
        if constexpr (sizeof(T) == 4) {
            const T y = inormal_central_f32(q);
            logd = M<T>::HALF_LN_2PI + T(0.5) * y * y + q.lp;
            return y;
        } else {             // evaluated from the smaller of cdf / sf
            const T e = M<T>::erfcinv(T(2) * M<T>::min(q.cdf, q.sf));
            logd = M<T>::HALF_LN_2PI + e * e + q.lp;
            return (q.cdf < q.sf ? -M<T>::SQRT2 : M<T>::SQRT2) * e;
        }
    }
    T tot;
    if (inv_type == JF_GF_INORMAL_PARTLY_CRUDE) {
        const T lsum = q.lc + q.ls;
        if constexpr (sizeof(T) == 4) {
            tot = M<T>::sqrt_fast(T(-2) * lsum) - T(0.4717);
            logd = T(-0.5) * M<T>::log_fast(T(-2) * lsum) - lsum + q.lp;
        } else {
            tot = M<T>::sqrt(T(-2) * lsum) - T(0.4717);
            logd = T(-0.5) * M<T>::log(T(-2) * lsum) - lsum + q.lp;
        }
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

constexpr int DPP_ROW_ROR8 = 0x128;     // row_ror:8: lane i <-> i ^ 8 inside each row of 16 lanes (both 8-lane halves hold their sums by then)

// lanes 16 apart (G = 32: coordinates 16..31 of a row): no DPP form crosses the 16-lane rows; ds_bpermute does
template <typename T> __device__ __forceinline__ T xor16(T v) { return __shfl_xor(v, 16, 64); }
// ... and the two halves of the wave (G = 64: a whole wave per row, 33 .. 64 coordinates, round 6)
template <typename T> __device__ __forceinline__ T xor32(T v) { return __shfl_xor(v, 32, 64); }

template <typename T, int G> __device__ __forceinline__ T group_sum(T v) {
    if constexpr (G >= 2) v += dpp_swap<DPP_XOR1>(v);
    if constexpr (G >= 4) v += dpp_swap<DPP_XOR2>(v);
    if constexpr (G >= 8) v += dpp_swap<DPP_HALF_MIRROR>(v);
    if constexpr (G >= 16) v += dpp_swap<DPP_ROW_ROR8>(v);
    if constexpr (G >= 32) v += xor16(v);
    if constexpr (G >= 64) v += xor32(v);
    return v;
}
template <typename T, int G> __device__ __forceinline__ T group_max(T v) {
    if constexpr (G >= 2) v = M<T>::max(v, dpp_swap<DPP_XOR1>(v));
    if constexpr (G >= 4) v = M<T>::max(v, dpp_swap<DPP_XOR2>(v));
    if constexpr (G >= 8) v = M<T>::max(v, dpp_swap<DPP_HALF_MIRROR>(v));
    if constexpr (G >= 16) v = M<T>::max(v, dpp_swap<DPP_ROW_ROR8>(v));
    if constexpr (G >= 32) v = M<T>::max(v, xor16(v));
    if constexpr (G >= 64) v = M<T>::max(v, xor32(v));
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
// Logistic mixture (gaussianization_flow.py:389-454):  with u_k = (x - mu_k)/w_k, a_k = |u_k|, hi_k = sigma(a_k), lo_k = e^{-a_k} hi_k
//   cdf = sum_k pi_k sigma(u_k),  sf = sum_k pi_k sigma(-u_k),  pdf = sum_k pi_k hi_k lo_k / w_k.
// RAW rows hold (mean, log-width, log-weight) as the amortisation MLP emits them (width / weight regulation fused into the loop, the
// weights are normalised at the end); derived rows hold (mean, 1/width, pi).

// Underflow-proof evaluation: every e^{-a_k} is factored as e^{-m} e^{-(a_k - m)}, m = min_k a_k, so each sum splits into an unscaled
// part (terms of order pi_k) and a part scaled by e^{-m} whose leading term is of order pi_k as well:
//   cdf = Cu + e^{-m} Cs,  sf = Su + e^{-m} Ss,  pdf = e^{-m} Ps      ->  log cdf = log(Cs) - m when no component lies left of x, ...
// Equivalent to the reference's log-sum-exp over components to rounding, at any distance from the components.
template <typename T, bool RAW> __device__ __forceinline__ MixQ<T> gfg_mixture_scaled(const T* __restrict__ p, const GfLayerDev<T>& o, int D, T x, T shift) {
    const T uniform_w = RAW ? T(1) : M<T>::rcp(T(o.K));
    T m = T(INFINITY);
    for (int k = 0; k < o.K; ++k) {
        const T iw = RAW ? gf_inv_width(o, p[o.off_lw + k * D]) : p[o.off_lw + k * D];
        m = M<T>::min(m, M<T>::abs((x - p[o.off_mean + k * D]) * iw));
    }
    const T em = M<T>::exp_fast(-m);                         // may underflow to 0: the unscaled parts then stand alone
    T Cu = T(0), Cs = T(0), Su = T(0), Ss = T(0), Ps = T(0), Nn = T(0);
    for (int k = 0; k < o.K; ++k) {
        const T iw = RAW ? gf_inv_width(o, p[o.off_lw + k * D]) : p[o.off_lw + k * D];
        const T wk = o.fit_norm ? (RAW ? gf_weight(o, p[o.off_ln + k * D], shift) : p[o.off_ln + k * D]) : uniform_w;
        const T u = (x - p[o.off_mean + k * D]) * iw;
        const T t = M<T>::exp_fast(m - M<T>::abs(u));      // <= 1, equals 1 for the nearest component (argument <= 0: v_exp is accurate here)
        const T hi = M<T>::rcp(T(1) + t * em);
        const T c1 = wk * hi, c2 = c1 * t;
        if (u >= T(0)) { Cu += c1; Ss += c2; }
        else { Su += c1; Cs += c2; }
        Ps += c2 * hi * iw;
        Nn += wk;
    }
    if constexpr (RAW) {
        const T inv = M<T>::rcp(Nn);
        Cu *= inv; Cs *= inv; Su *= inv; Ss *= inv; Ps *= inv;
    }
    MixQ<T> q;
    q.cdf = Cu + em * Cs;
    q.sf = Su + em * Ss;
    q.lc = Cu > T(0) ? M<T>::log_fast(q.cdf) : M<T>::log_fast(Cs) - m;
    q.ls = Su > T(0) ? M<T>::log_fast(q.sf) : M<T>::log_fast(Ss) - m;
    q.lp = M<T>::log_fast(Ps) - m;
    return q;
}

// FAST: the layer has the reference's default options (o.fast), which turns every option test inside the loop into a compile-time
// constant -- without it the k loop keeps ~10 scalar branches and a wait after every LDS read.
template <typename T, bool RAW, bool FAST> __device__ __forceinline__ MixQ<T> gfg_mixture_impl(const T* __restrict__ p, const GfLayerDev<T>& o, int D, T x) {
    T C = T(0), S = T(0), P = T(0), Nn = T(0), shift = T(0);
    const bool fit_norm = FAST ? true : (o.fit_norm != 0);
    const T uniform_w = RAW ? T(1) : M<T>::rcp(T(o.K));
    if (!FAST && RAW && o.fit_norm && !o.reg_norm) {       // unbounded log-weights: shift by the max before exponentiating
        shift = p[o.off_ln];
        for (int k = 1; k < o.K; ++k) shift = M<T>::max(shift, p[o.off_ln + k * D]);
    }
    const T* pm = p + o.off_mean;
    const T* pw = p + o.off_lw;
    const T* pn = p + o.off_ln;
#pragma unroll 2
    for (int k = 0; k < o.K; ++k) {
        const T mu = pm[k * D], rw = pw[k * D];
        const T rn = fit_norm ? pn[k * D] : T(0);
        T iw, wk;
        if constexpr (RAW) {
            if constexpr (FAST) {
                const T ae = o.inv_wmax + M<T>::exp_fast(-rw);
                iw = ae * M<T>::rcp(o.wmin * ae + T(1));
                wk = o.nmin + o.nmax * M<T>::rcp(T(1) + M<T>::exp_fast(-rn));
            } else {
                iw = gf_inv_width(o, rw);
                wk = fit_norm ? gf_weight(o, rn, shift) : T(1);
            }
        } else {
            iw = rw;
            wk = fit_norm ? rn : uniform_w;
        }
        const T u = (x - mu) * iw;
        const T t = M<T>::exp_fast(-M<T>::abs(u));
        const T hi = M<T>::rcp(T(1) + t);                  // sigma(|u|)
        const T lo = t * hi;                               // sigma(-|u|)
        const bool pos = u >= T(0);
        C += wk * (pos ? hi : lo);
        S += wk * (pos ? lo : hi);
        P += wk * hi * lo * iw;
        if constexpr (RAW) Nn += wk;
    }
    if constexpr (RAW) {
        const T inv = M<T>::rcp(Nn);
        C *= inv; S *= inv; P *= inv;
    }
    MixQ<T> q;
    q.lc = M<T>::log_fast(C); q.ls = M<T>::log_fast(S); q.lp = M<T>::log_fast(P);
    q.cdf = C; q.sf = S;
    const bool under = !(C > M<T>::TINY && S > M<T>::TINY && P > M<T>::TINY);
    if (__any(under)) {                                    // wave-uniform branch
        const MixQ<T> qs = gfg_mixture_scaled<T, RAW>(p, o, D, x, shift);
        if (under) q = qs;
    }
    return q;
}
template <typename T, bool RAW> __device__ __forceinline__ MixQ<T> gfg_mixture(const T* __restrict__ p, const GfLayerDev<T>& o, int D, T x) {
    return o.fast ? gfg_mixture_impl<T, RAW, true>(p, o, D, x) : gfg_mixture_impl<T, RAW, false>(p, o, D, x);
}

// in-place derive of a staged raw row by the G lanes of its group: lane d < D takes column d, reflections are dealt round-robin
template <typename T, int G> __device__ __forceinline__ void gfg_derive(T* __restrict__ row, const GfLayerDev<T>& o, int D, int g, bool columns) {
    if (columns && g < D) gf_derive_column<T>(row, o, D, g);
    for (int i = g; i < o.hh; i += G) gf_derive_reflection<T>(row, o, D, i);
}

// log Phi(-a) = log(erfc(a / sqrt 2) / 2), a >= 0: right-hand side of the log-space equation of gf_approach.  That phase only
// has to land within ~1e-3 of the root (the Newton stage then solves the layer's own inverse-CDF equation), so float32 and a fractional
// error of 1.2e-7 are plenty: erfc(x) = t exp(-x^2 + c0 + c1 t + .. + c9 t^9), t = 1 / (1 + x / 2) -- the Chebyshev fit of Numerical Recipes
// ("erfcc", Press et al., section 6.2) -- taken in LOG space, where it needs no exponential and cannot underflow at any a: 1 reciprocal,
// 1 logarithm, 12 FMAs (rounds 1-3: OCML erfc + log + log1p + two divisions, exact to an ulp).
__device__ __forceinline__ float gf_log_ndtr_neg(float a) {
    const float x = a * 0.70710678118654752440f;
    const float t = M<float>::rcp(1.0f + 0.5f * x);
    float p = 0.17087277f;
    p = fmaf(p, t, -0.82215223f); p = fmaf(p, t, 1.48851587f); p = fmaf(p, t, -1.13520398f); p = fmaf(p, t, 0.27886807f);
    p = fmaf(p, t, -0.18628806f); p = fmaf(p, t, 0.09678418f); p = fmaf(p, t, 0.37409196f); p = fmaf(p, t, 1.00002368f);
    p = fmaf(p, t, -1.26551223f);
    return M<float>::log_fast(t) - x * x + p - 0.69314718055994530942f;
}


// ---- approach phase of the sampling solvers.  The reference brackets the root of stage(mixture(x)) = z with 25 bisections of [-1e5, 1e5]
// (bisection_n_newton.py:11-72) before its Newton stage; rounds 1-3 restated that (~17 mixture evaluations per solve outside the far-midpoint
// skips -- 0.8 of the sampling kernels' time).  Round 4: a SAFEGUARDED NEWTON iteration on the stage's equation in the space where it is nearly
// linear:
//   isigmoid stages     g(x) = log cdf - log sf - z            slope pdf / cdf + pdf / sf   (exactly linear for one component)
//   normal-type stages  g(x) = log cdf - log Phi(z)  (z <= 0)  slope pdf / cdf              (the side that does not cancel; linear in the
//                       g(x) = log Phi(-z) - log sf  (z > 0)   slope pdf / sf                tail the root of a small z lies in)
// g is increasing; where the mixture's cdf is log-concave a tangent's zero lands left of the root and the iterates then approach it monotonically
// and quadratically.  A proposal outside the bracket of the signs seen so far (flat stretches between distant components) is replaced by the
// bracket's midpoint, which keeps bisection's guarantee; so is -- rtsafe's rule -- a step that is not at most half the one before it once both
// ends are known (Newton's two-cycle around an inflection: both proposals inside a bracket that then shrinks by 1e-4 per step).  It ends where
// the bisection ended: with the root inside a bracket of 6e-3, or on Newton's quadratic tail (a step below 2e-3 that is at most a quarter of the
// one before it); any other short step says "the root is here" only if g is as linear as its tangent, so the iterate steps 4e-3 PAST the
// proposal and the next evaluation either closes the bracket around the root or shows that it lies further on.  Typically 4-6 evaluations.
// eval(x) -> MixQ<F> (log cdf, log sf, log pdf); x0: the mixture's mean.  The Newton stage after it is the reference's, unchanged.
template <typename F, typename EVAL> __device__ __forceinline__ F gf_approach(EVAL eval, bool proxy, F z, F x0, bool live, int* n_evals = nullptr) {
    const bool neg = z <= F(0);
    const F tz = proxy ? (F)gf_log_ndtr_neg((float)M<F>::abs(z)) : F(0);
    F xf = x0, blo = F(-1e5), bhi = F(1e5), dxold = F(2e5);
    bool act = live;
    if (newton_reference_rule()) {                                 // audit switch (jf_common.h): the reference's 25 bisections on [-1e5, 1e5] (bisection_n_newton.py:11-60)
        for (int it = 0; it < 25; ++it) {
            xf = F(0.5) * (blo + bhi);
            if (n_evals != nullptr) *n_evals += 1;
            const MixQ<F> q = eval(xf);
            const F g = proxy ? (neg ? q.lc - tz : tz - q.ls) : q.lc - q.ls - z;
            if (g < F(0)) blo = xf; else bhi = xf;
        }
        return xf;
    }
    for (int it = 0; it < 40 && __any(act); ++it) {
        if (n_evals != nullptr) *n_evals += 1;                     // (probe builds: evaluations the WAVE makes, JF_PROBE_COUNT_APPROACH)
        const MixQ<F> q = eval(xf);
        F g, sl;
        if (proxy) {
            g = neg ? q.lc - tz : tz - q.ls;
            sl = M<F>::exp_fast(q.lp - (neg ? q.lc : q.ls));
        } else {
            g = q.lc - q.ls - z;
            sl = M<F>::exp_fast(q.lp - q.lc) + M<F>::exp_fast(q.lp - q.ls);
        }
        if (act) {
            if (g < F(0)) blo = xf; else bhi = xf;
            F xn = xf - g * M<F>::rcp(sl);
            const bool inside = xn >= blo && xn <= bhi;                      // false also for a non-finite proposal
            if (bhi - blo <= F(6e-3)) {                                      // the root is bracketed as tightly as the 25 bisections did: done
                xf = inside ? xn : F(0.5) * (blo + bhi);
                act = false;
            } else {
                const bool slow = blo > F(-1e5) && bhi < F(1e5) && M<F>::abs(xn - xf) > F(0.5) * M<F>::abs(dxold);
                const F step = M<F>::abs(xn - xf);
                if (!inside || slow) xn = F(0.5) * (blo + bhi);
                else if (step < F(2e-3) && step <= F(0.25) * M<F>::abs(dxold)) act = false;
                else if (step < F(2e-3)) {
                    const F over = xn + (xn >= xf ? F(4e-3) : F(-4e-3));
                    xn = (over > blo && over < bhi) ? over : F(0.5) * (blo + bhi);
                }
                dxold = xn - xf;
                xf = xn;
            }
        }
    }
    // safety net: a lane still active after 40 evaluations (one bracket end never left +-1e5: the `slow` rule was off, e.g. narrow
    // components far apart with |z| near 8) finishes with the reference's plain bisection of what it has -- at most 25 halvings, the guarantee
    // of bisection_n_newton.py:11-60 -- so the Newton stage never starts from an arbitrary point.  Wave-uniform: skipped when no lane needs it.
    for (int it = 0; it < 25 && __any(act); ++it) {
        const F mid = F(0.5) * (blo + bhi);
        const MixQ<F> q = eval(mid);
        const F g = proxy ? (neg ? q.lc - tz : tz - q.ls) : q.lc - q.ls - z;
        if (act) {
            if (g < F(0)) blo = mid; else bhi = mid;
            xf = F(0.5) * (blo + bhi);
            if (bhi - blo <= F(6e-3)) act = false;
        }
    }
    return xf;
}

// ---------------------------------------------------------------------------------------------------------- sampling direction
// approach phase + the reference's Newton stage (layers/bisection_n_newton.py:11-135, called with 25 / 20 iterations on [-1e5, 1e5], :921) for
// one coordinate;
// the Newton stopping rule sums |update| over the row's coordinates (group butterfly), so the D lanes of a row stop together.
template <typename T, int G> __device__ __forceinline__ T gfg_solve(const T* __restrict__ p, const GfLayerDev<T>& o, int D, bool live, T z,
                                                                     bool row_valid, bool leader, int32_t* status) {
    // approach phase (gf_approach above) from the mixture's mean, on the derived row in LDS
    T x = T(0);
    for (int k = 0; k < o.K; ++k) x += (o.fit_norm ? p[o.off_ln + k * D] : T(1) / T(o.K)) * p[o.off_mean + k * D];
    x = gf_approach<T>([&](T xx) { return gfg_mixture<T, false>(p, o, D, xx); }, o.inv_type != JF_GF_ISIGMOID, z, x, live);
    bool active = row_valid;
    T ferr = T(0), prev = T(INFINITY);
    bool nonfinite = false;
    for (int it = 0; it < 20 && __any(active); ++it) {
        const IcdfOut<T> s = gf_icdf<T>(o.inv_type, gfg_mixture<T, false>(p, o, D, x));
        const T f = s.y - z;
        const T upd = f / M<T>::exp(s.logd);
        const T usum = group_sum<T, G>(live ? M<T>::abs(upd) : T(0));
        status_add(status, JF_STATUS_NEWTON_STEPS, active && leader);
        if (active) {
            const T nx = x - upd;
            if (M<T>::finite(nx)) x = nx; else nonfinite = nonfinite || live;     // keep the previous iterate (:84-91)
            ferr = M<T>::abs(f);
            active = usum >= newton_tol<T>();
        }
        if (sizeof(T) == 4 && !newton_reference_rule()) {
            // float32: the reference's absolute 1e-14 fires only on an exactly zero update, i.e. its float32 runs do all 20 steps and the last
            // ~16 of them move the iterate by rounding noise.  A row is at that floor when its update has reached the resolution of its
            // coordinates, or has stopped shrinking while already below 1e-4 of them; further steps cannot improve it.
            const T xs = group_sum<T, G>(live ? M<T>::max(M<T>::abs(x), T(1)) : T(0));
            const bool done = usum < T(2.5e-7) * xs || (usum < T(JF_F32_NEWTON_FLOOR) * xs && group_max<T, G>(live ? M<T>::abs(f) : T(0)) <= T(1e-4));   // (jf_math.h)
            if (done || (usum >= T(0.5) * prev && usum < T(1e-4) * xs)) active = false;
            prev = usum;
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
