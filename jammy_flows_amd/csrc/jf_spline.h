// Rational-quadratic spline family (jammy_flows/layers/spline_fns.py) for a lane-per-sample kernel.
//
// Each lane owns a small knot table in LDS (cumulative widths, cumulative heights, derivatives: 3*(nb+1) values, row stride
// odd => conflict-free ds_read_b32 with lane-dependent dynamic indices).  The table is built from the lane's parameter row
// exactly in the reference's order of operations (softmax -> min-size mix -> cumsum -> affine map -> pinned ends), the bin is
// found by the reference's counting rule ("#(x >= knot) - 1 with the last knot bumped by eps", spline_fns.py:13-19), then the
// closed-form forward map / quadratic-root inverse is evaluated (spline_fns.py:127-186).
#pragma once
#include "jf_common.h"
#include "jf_math.h"

namespace jf {

// Arithmetic of the table build and of the closed-form evaluation.  float64: OCML functions and IEEE divisions in the reference's order of
// operations -- the searchsorted results are bit-exact against it (tests/test_gpu_parity.py: test_spline_bins_bit_exact_float64).  float32 (bar
// |dlogp| < 1e-2): the hardware exponential / logarithm / reciprocal and ONE reciprocal of the softmax sum instead of a division per bin -- a
// table costs ~3 nb transcendentals (softmax of widths and heights, softplus of the derivatives) and with OCML's expf / log1pf and correctly
// rounded divisions that was ~600 instructions per row; the spline is C1, so a knot that moves by 1e-6 moves neither y nor log|dy/dx| by more.
template <typename T> struct SM {
    static __device__ __forceinline__ T exp(T x) { return M<T>::exp(x); }
    static __device__ __forceinline__ T log(T x) { return M<T>::log(x); }
    static __device__ __forceinline__ T sqrt(T x) { return M<T>::sqrt(x); }
    static __device__ __forceinline__ T div(T a, T b) { return a / b; }
    static __device__ __forceinline__ T softplus(T x) { return jf::softplus<T>(x); }
};
template <> struct SM<float> {
    static __device__ __forceinline__ float exp(float x) { return M<float>::exp_fast(x); }
    static __device__ __forceinline__ float log(float x) { return M<float>::log_fast(x); }
    static __device__ __forceinline__ float sqrt(float x) { return M<float>::sqrt_fast(x); }
    static __device__ __forceinline__ float div(float a, float b) { return a * M<float>::rcp(b); }
    static __device__ __forceinline__ float softplus(float x) { return fmaxf(x, 0.f) + M<float>::log_fast(1.0f + M<float>::exp_fast(-fabsf(x))); }
};

constexpr int JF_SPLINE_MAX_BINS = 16;                            // the potentials of 'v' (10 bins): tables at a fixed stride
constexpr int JF_SPLINE_TAB = 3 * (JF_SPLINE_MAX_BINS + 1) + 2;   // 53 (odd)
constexpr int JF_SPLINE_CAP = 64;                                 // 'r', 'o', the splines nested in 'f' and 'g' with the rq_splines stretch: a lane's table has the chain's own bin count
                                                                  // (spline_tab_words) and the launch sizes its row tile to the LDS that takes

// public option block shared by 'r', 'o' and the Euclidean rq_splines stretch (mirrors jf_spline_opts of the C header)
template <typename T> struct SplineDev {
    int nb, smooth, fix_first, fix_second, independent, fix_bd;
    int n_w, n_h, n_d;
    T fix_bd_value, min_w, min_h, min_d, ratio;
};

template <typename T> struct SplineOut { T y, lad; int bin; };

// the three sections follow each other at the spline's own bin count: a lane's table is 3 (nb + 1) words, not 3 x 17 -- the LDS a lane takes bounds the
// resident waves of the lane-per-row kernels (spline_tab_words)
template <typename T> struct KnotTab {
    T* cw; T* ch; T* d;   // nb+1 entries each
    __device__ __forceinline__ KnotTab(T* base, int nb) : cw(base), ch(base + nb + 1), d(base + 2 * (nb + 1)) {}
};
// per-lane table stride (elements) for splines of at most nb bins: odd, so that lane-dependent indices stay conflict-free
__host__ __device__ inline int spline_tab_words(int nb) { const int w = 3 * (nb + 1); return (w & 1) ? w : w + 1; }

// unnormalised widths / heights of the lane with the option handling of rational_quadratic_spline.py:200-230 / splines_1d.py:136-156
template <typename T> __device__ inline void spline_unpack_wh(const T* __restrict__ p, const SplineDev<T>& o, KnotTab<T>& t) {
    const int nb = o.nb;
    int wi = 0, hi = 0;
    int k = 0;
    if (o.fix_first) {
        t.ch[0] = T(0);
        t.cw[0] = T(0);
        k = 1;
        if (o.fix_second) { t.cw[1] = T(0); }
    }
    const int w_start = o.fix_first ? (o.fix_second ? 2 : 1) : 0;
    const int sym3 = (o.smooth == 1 && nb == 3) ? 1 : 0;        // symmetric 3-bin case: last bin mirrors the first
    for (int j = w_start; j < nb - sym3; ++j) t.cw[j] = p[wi++];
    for (int j = k; j < nb - sym3; ++j) t.ch[j] = p[o.n_w + hi++];
    if (o.independent) for (int j = 0; j < nb - sym3; ++j) t.ch[j] = t.cw[j] + t.ch[j];
    if (sym3) { t.cw[nb - 1] = t.cw[0]; t.ch[nb - 1] = t.ch[0]; }
    if (o.ratio > T(0)) {   // restrict_max_min_width_height_ratio (spline_fns.py:80-85)
        const T ln_max = (M<T>::log(o.ratio) - M<T>::log(T(nb - 1))) * T(0.5);
        for (int j = 0; j < nb; ++j) {
            t.cw[j] = T(2) * ln_max / (T(1) + M<T>::exp(-t.cw[j])) - ln_max;
            t.ch[j] = T(2) * ln_max / (T(1) + M<T>::exp(-t.ch[j])) - ln_max;
        }
    }
}

// in place: unnormalised[0..nb) -> cumulative knots[0..nb]   (spline_fns.py:88-98)
template <typename T> __device__ inline void spline_cum_knots(T* __restrict__ a, int nb, T lo, T hi, T rel_min, bool pin) {
    // (the three loops are chains of LDS round trips when taken one element at a time: unrolled four-fold, the loads of a group are in flight
    //  together -- these kernels sit on that latency, not on arithmetic)
    T m = a[0];
#pragma unroll 4
    for (int j = 1; j < nb; ++j) m = M<T>::max(m, a[j]);
    T s = T(0);
#pragma unroll 4
    for (int j = 0; j < nb; ++j) { const T e = SM<T>::exp(a[j] - m); a[j] = e; s += e; }
    const T scale = T(1) - rel_min * T(nb);
    T cum = T(0);
    T prev = (hi - lo) * T(0) + lo;
    const T inv_s = sizeof(T) == 4 ? SM<T>::div(T(1), s) : T(1);       // float32: one reciprocal for the whole softmax
#pragma unroll 4
    for (int j = 0; j < nb; ++j) {
        const T frac = rel_min + scale * (sizeof(T) == 4 ? a[j] * inv_s : a[j] / s);
        cum += frac;
        const T knot = (hi - lo) * cum + lo;
        a[j] = prev;          // shift: a[j] becomes knot j, carry knot j+1
        prev = knot;
    }
    a[nb] = prev;
    if (pin) { a[0] = lo; a[nb] = hi; }
}

template <typename T> __device__ __forceinline__ int spline_search(const T* __restrict__ knots, int nb, T x, T eps) {
    int c = 0;
    for (int j = 0; j < nb; ++j) c += (x >= knots[j]) ? 1 : 0;
    c += (x >= knots[nb] + eps) ? 1 : 0;
    return c - 1;
}

// closed-form evaluation in a bin given its six knot values (spline_fns.py:127-186); T may carry tangents (the adjoint of the spline layers
// evaluates this on DualN<T, 7>: the input and the six knot values, jf_spline_adj.h)
template <typename T> __device__ inline SplineOut<T> spline_core_vals(T in_cw, T cw1, T in_ch, T ch1, T d0, T d1, int b, T x, bool inverse) {
    const T in_w = cw1 - in_cw;
    const T in_h = ch1 - in_ch;
    const T delta = SM<T>::div(in_h, in_w);
    const T s = d0 + d1 - T(2) * delta;
    SplineOut<T> r;
    r.bin = b;
    T theta;
    if (inverse) {
        const T dy = x - in_ch;
        const T a = dy * s + in_h * (delta - d0);
        const T bq = in_h * d0 - dy * s;
        const T c = -delta * dy;
        const T disc = bq * bq - T(4) * a * c;
        theta = SM<T>::div(T(2) * c, -bq - SM<T>::sqrt(disc));
        r.y = theta * in_w + in_cw;
    } else {
        theta = SM<T>::div(x - in_cw, in_w);
    }
    const T t1mt = theta * (T(1) - theta);
    const T den = delta + s * t1mt;
    const T num = delta * delta * (d1 * theta * theta + T(2) * delta * t1mt + d0 * (T(1) - theta) * (T(1) - theta));
    const T lad = SM<T>::log(num) - T(2) * SM<T>::log(den);
    if (inverse) {
        r.lad = -lad;
    } else {
        r.y = in_ch + SM<T>::div(in_h * (delta * theta * theta + d0 * t1mt), den);
        r.lad = lad;
    }
    return r;
}
// closed-form evaluation in bin `b` of a knot table
template <typename T> __device__ inline SplineOut<T> spline_core(const KnotTab<T>& t, int b, T x, bool inverse) {
    return spline_core_vals<T>(t.cw[b], t.cw[b + 1], t.ch[b], t.ch[b + 1], t.d[b], t.d[b + 1], b, x, inverse);
}

// interior derivatives of the C2-smooth variants in closed form (spline_fns.py:431-484), boundary ones given
template <typename T> __device__ inline void spline_smooth_derivs(KnotTab<T>& t, int nb, T bd0, T bd1) {
    if (nb == 1) { t.d[0] = bd0; t.d[1] = bd1; return; }
    const T w1 = t.cw[1] - t.cw[0], w2 = t.cw[2] - t.cw[1];
    const T h1 = t.ch[1] - t.ch[0], h2 = t.ch[2] - t.ch[1];
    if (nb == 2) {
        const T hs = h1 + h2;
        const T lo_p = h1 / hs, hi_p = h2 / hs;
        const T neg_p_half = T(0.5) * (lo_p * ((h2 / w2) - bd1) + hi_p * ((h1 / w1) - bd0));
        const T q = -(h1 * h2) * (lo_p * (T(1) / (w1 * w1)) + hi_p * (T(1) / (w2 * w2)));
        const T res = neg_p_half + M<T>::sqrt(neg_p_half * neg_p_half - q);
        t.d[0] = bd0; t.d[1] = res; t.d[2] = bd1;
    } else {   // nb == 3, symmetric solution
        const T cden = w1 * w2 * (T(2) * h1 + h2);
        const T p = h2 * (bd0 * w1 * w2 - h1 * (w1 + w2)) / cden;
        const T q = -h1 * h2 * (h1 * w2 * w2 + h2 * w1 * w1) / (cden * w1 * w2);
        const T neg_p_half = -p * T(0.5);
        const T res = neg_p_half + M<T>::sqrt(neg_p_half * neg_p_half - q);
        t.d[0] = bd0; t.d[1] = res; t.d[2] = res; t.d[3] = bd1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// interval spline used by 'r' (plain or smooth) on [lo, hi]  (rational_quadratic_spline.py:232-280 -> spline_fns.py:45-186 / 361-558)
// ---------------------------------------------------------------------------------------------------------------
// build: the knot table of one parameter row (independent of x) -- per lane for per-sample parameters, once per workgroup for broadcast ones
template <typename T> __device__ inline void spline_interval_build(const T* __restrict__ p, const SplineDev<T>& o, T* __restrict__ tab, T lo, T hi) {
    const int nb = o.nb;
    KnotTab<T> t(tab, nb);
    spline_unpack_wh<T>(p, o, t);
    spline_cum_knots<T>(t.cw, nb, lo, hi, o.min_w, true);
    spline_cum_knots<T>(t.ch, nb, lo, hi, o.min_h, true);
    const T* pd = p + o.n_w + o.n_h;
    if (o.smooth == 0) {
        if (o.fix_bd) {
            const T fixed = o.min_d + SM<T>::softplus(o.fix_bd_value);
            t.d[0] = fixed; t.d[nb] = fixed;
            for (int j = 1; j < nb; ++j) t.d[j] = o.min_d + SM<T>::softplus(pd[j - 1]);
        } else {
            for (int j = 0; j <= nb; ++j) t.d[j] = o.min_d + SM<T>::softplus(pd[j]);
        }
    } else {
        const T b0 = o.min_d + SM<T>::softplus(o.fix_bd ? o.fix_bd_value : pd[0]);
        const T b1 = o.min_d + SM<T>::softplus(o.fix_bd ? o.fix_bd_value : pd[1]);
        spline_smooth_derivs<T>(t, nb, b0, b1);
    }
}
template <typename T> __device__ inline SplineOut<T> spline_interval_eval(const SplineDev<T>& o, const T* __restrict__ tab, T x, bool inverse, T lo, T hi,
                                                                         bool& out_of_range) {
    const int nb = o.nb;
    const KnotTab<T> t(const_cast<T*>(tab), nb);
    out_of_range = (x < lo) || (x > hi);
    const T eps = T(1e-6);
    int b = spline_search<T>(inverse ? t.ch : t.cw, nb, x, eps);
    const int raw = b;
    b = b < 0 ? 0 : (b > nb - 1 ? nb - 1 : b);     // only reachable for out-of-range inputs (flagged)
    SplineOut<T> r = spline_core<T>(t, b, x, inverse);
    r.bin = raw;
    return r;
}
template <typename T> __device__ inline SplineOut<T> spline_interval(const T* __restrict__ p, const SplineDev<T>& o, T* __restrict__ tab, T x, bool inverse,
                                                                    T lo, T hi, bool& out_of_range, bool built = false) {
    if (!built) spline_interval_build<T>(p, o, tab, lo, hi);
    return spline_interval_eval<T>(o, tab, x, inverse, lo, hi, out_of_range);
}

// The C2-smooth circular spline behind its knot table: two bins on [0, 2 pi] whose only free knot values are cw1 and ch1 (spline_fns.py:628-668).
// T may carry tangents (the adjoint evaluates this on DualN<T, 3>: x, cw1, ch1 -- jf_manifold_adj.h); the bin search looks at values only.
template <typename T> __device__ inline SplineOut<T> spline_circular_smooth_vals(T cw1, T ch1, T x, bool inverse, int& raw_bin) {
    const T TWO_PI = M<T>::TWO_PI;
    const T cw0 = T(0), cw2 = TWO_PI, ch0 = T(0), ch2 = TWO_PI;
    const T w1 = cw1 - cw0, w2 = cw2 - cw1;
    const T h1 = ch1 - ch0, h2 = ch2 - ch1;
    const T hp = h1 * h2, wp = w1 * w2;
    const T sq = M<T>::sqrt(hp * (T(8) * ((h2 * w1) * (h2 * w1) + (h1 * w2) * (h1 * w2)) + (T(9) * (w1 + w2) * (w1 + w2) - T(16) * wp) * hp));
    const T res = (hp * (w1 + w2) + sq) / (T(4) * (h1 + h2) * wp);
    const T w1mx = -M<T>::PI + w1 * T(0.5);
    const T w1mx_p_w2 = w1mx + w2;
    const T nom = h2 * w1mx * (w1mx * h1 - res * w1 * w1mx_p_w2);
    const T den = h1 * w2 * w2 + T(2) * (h1 - res * w1) * w1mx * w1mx_p_w2;
    const T corr = TWO_PI - (h1 + nom / den);
    const T mid = M<T>::PI - w1 * T(0.5);
    T used = inverse ? x - corr : x - mid;
    if (used < T(0)) used += TWO_PI;
    const T eps = T(1e-6);
    // the counting rule of spline_search on the three knots of the searched axis
    const T k1 = inverse ? ch1 : cw1;
    int b = ((used >= T(0)) ? 1 : 0) + ((used >= k1) ? 1 : 0) + ((used >= TWO_PI + eps) ? 1 : 0) - 1;
    raw_bin = b;
    b = b < 0 ? 0 : (b > 1 ? 1 : b);
    SplineOut<T> r = b == 0 ? spline_core_vals<T>(cw0, cw1, ch0, ch1, res, res, 0, used, inverse) : spline_core_vals<T>(cw1, cw2, ch1, ch2, res, res, 1, used, inverse);
    T y = r.y + (inverse ? mid : corr);
    if (y > TWO_PI) y -= TWO_PI;
    if (x == T(0)) y = T(0);
    if (x == TWO_PI) y = TWO_PI;
    r.y = y;
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// circular splines used by 'o': plain periodic (derivative at 0 == derivative at 2pi) or the smooth 2-bin variant
// (splines_1d.py:162-194 -> spline_fns.py:45-186 / 561-760)
// ---------------------------------------------------------------------------------------------------------------
template <typename T> __device__ inline SplineOut<T> spline_circular(const T* __restrict__ p, const SplineDev<T>& o, T* __restrict__ tab, T x, bool inverse,
                                                                    T scale, bool& out_of_range) {
    const int nb = o.nb;
    KnotTab<T> t(tab, nb);
    const T TWO_PI = M<T>::TWO_PI;
    out_of_range = (x < T(0)) || (x > TWO_PI);
    spline_unpack_wh<T>(p, o, t);
    if (scale != T(1)) {     // fvm_2d.py:416-427: the azimuthal flow's parameters are scaled by a smooth step in cos(theta)
        for (int j = 0; j < nb; ++j) { t.cw[j] *= scale; t.ch[j] *= scale; }
    }
    spline_cum_knots<T>(t.cw, nb, T(0), TWO_PI, o.min_w, true);
    spline_cum_knots<T>(t.ch, nb, T(0), TWO_PI, o.min_h, true);
    const T eps = T(1e-6);
    if (o.smooth == 0) {
        const T* pd = p + o.n_w + o.n_h;
        if (o.fix_bd) {
            const T fixed = o.min_d + SM<T>::softplus(o.fix_bd_value);
            t.d[0] = fixed; t.d[nb] = fixed;
            for (int j = 1; j < nb; ++j) t.d[j] = o.min_d + SM<T>::softplus(pd[j - 1] * scale);
        } else {
            for (int j = 0; j < nb; ++j) t.d[j] = o.min_d + SM<T>::softplus(pd[j] * scale);
            t.d[nb] = t.d[0];
        }
        int b = spline_search<T>(inverse ? t.ch : t.cw, nb, x, eps);
        const int raw = b;
        b = b < 0 ? 0 : (b > nb - 1 ? nb - 1 : b);
        SplineOut<T> r = spline_core<T>(t, b, x, inverse);
        r.bin = raw;
        return r;
    }
    // smooth circular: two bins, one shared derivative, seam shifted to mid-bin (spline_fns.py:628-668)
    int raw;
    SplineOut<T> r = spline_circular_smooth_vals<T>(t.cw[1], t.ch[1], x, inverse, raw);
    r.bin = raw;
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// Euclidean spline with learnable box and linear tails ('g' with nonlinear_stretch_type="rq_splines",
// spline_fns.py:188-358): eps = 0 search, clamped index, linear extension outside the box.
// un_w / un_h / un_d point at this dimension's K / K / K+1 raw values; box = (left, ln(width-0.5), bottom, ln(height-0.5)).
// ---------------------------------------------------------------------------------------------------------------
// (ROW: anything indexable that yields T -- plain pointers, or the backward kernel's seeded view of a row of plain values)
template <typename T, typename ROW = const T*> __device__ inline SplineOut<T> spline_linext(ROW un_w, ROW un_h, ROW un_d, ROW box, int nb, T* __restrict__ tab,
                                                                                        T x, bool inverse) {
    KnotTab<T> t(tab, nb);
    const T left = box[0], right = left + M<T>::exp(box[1]) + T(0.5);      // gaussianization_flow.py:901-907
    const T bottom = box[2], top = bottom + M<T>::exp(box[3]) + T(0.5);
    for (int j = 0; j < nb; ++j) { t.cw[j] = un_w[j]; t.ch[j] = un_h[j]; }
    spline_cum_knots<T>(t.cw, nb, left, right, T(1e-3), false);
    spline_cum_knots<T>(t.ch, nb, bottom, top, T(1e-3), false);
    for (int j = 0; j <= nb; ++j) t.d[j] = T(1e-3) + SM<T>::softplus(un_d[j]);
    int b = spline_search<T>(inverse ? t.ch : t.cw, nb, x, T(0));
    const int raw = b;
    b = b < 0 ? 0 : (b > nb - 1 ? nb - 1 : b);
    SplineOut<T> r = spline_core<T>(t, b, x, inverse);
    r.bin = raw;
    const T d0 = t.d[0], dl = t.d[nb];
    if (inverse) {
        if (x <= bottom) { r.y = x / d0 + (t.cw[0] - t.ch[0] / d0); r.lad = -M<T>::log(d0); }
        if (x >= top) { r.y = x / dl + (t.cw[nb] - t.ch[nb] / dl); r.lad = -M<T>::log(dl); }
    } else {
        if (x <= left) { r.y = x * d0 + (t.ch[0] - t.cw[0] * d0); r.lad = M<T>::log(d0); }
        if (x >= right) { r.y = x * dl + (t.ch[nb] - t.cw[nb] * dl); r.lad = M<T>::log(dl); }
    }
    return r;
}

}  // namespace jf
