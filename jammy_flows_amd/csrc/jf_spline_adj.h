// Reverse-mode adjoint of the rational-quadratic spline family (the splines of 'r', 'o' and the ones nested in 'f'), round 6.
//
// What torch.autograd returns for spline_fns.rational_quadratic_spline (spline_fns.py:45-186) behind the option handling of
// rational_quadratic_spline.py:200-280 / splines_1d.py:136-194 / fvm_2d.py:416-432.  Until round 5 the backward kernels replayed the whole
// chain on dual numbers once per parameter group (3 nb + 1 directions per spline, each with its own table build: 34 x ... 360 x the forward).
// A spline's output depends on its parameter row only through the SIX knot values of the bin the input falls into, so the adjoint is staged:
//   (1) the closed-form bin evaluation on DualN<T, 7> -- tangents for the input and the bin's six knot values, through the very function
//       the forward kernels evaluate (spline_core_vals, jf_spline.h) -- contracted with the upstream gradients of (y, log|dy/dx|);
//   (2) the table build reversed by hand, O(bins) multiply-adds per row and two logistic functions: cumulative sums -> softmax -> min-size
//       mix -> option handling (fixed first / second knots, dependent heights, the width / height ratio squashing, the azimuthal scale of
//       fvm_2d) and softplus for the derivatives.  The softmax weights are read back off the knots (soft_i = ((knot_{i+1} - knot_i) / span -
//       rel_min) / (1 - nb rel_min)), the unnormalised values off the parameter row: a lane needs no scratch beyond the forward's 3 (nb + 1)
//       knot words.
// Step (2) is linear in the knot adjoints: chains with permanent (broadcast) parameters whose tables do not depend on the row ('r') add the
// rows' knot adjoints up per workgroup (LDS atomics on 3 (nb + 1) accumulators) and reverse the table ONCE (spline_adj_table_reverse_dense;
// mchain_rev_kernel, manifold_rev_kernels.hip).
// The C2-smooth circular spline (two bins: everything behind its table is a function of x and the two free knot values) takes step (1) on
// three tangents through spline_circular_smooth_vals and the same step (2); the C2-smooth interval variants (<= 3 bins, <= 8 parameters;
// derivatives that are closed-form functions of all knots) keep a dual-number pass over the layer's own parameters (stage_dual_adjoint,
// jf_manifold_adj.h).
// Checked against the reference's autograd (tests/golden/grads/) and against the dual-number replay (JF_M_BWD_DUAL=1, scripts/probe/m_adjoint_check.py).
#pragma once
#include "jf_dual.h"
#include "jf_spline.h"

namespace jf {

// ---- forward: the knot table of an interval (lo, hi pinned) or circular (0 .. 2 pi, periodic derivative, `scale` = fvm_2d's azimuthal scaling)
// spline, smooth == 0: the forward kernels' own steps (spline_interval_build / the table part of spline_circular)
template <typename T> __device__ inline void spline_adj_build(const T* __restrict__ p, const SplineDev<T>& o, T* __restrict__ tab, T lo, T hi, bool circular, T scale) {
    const int nb = o.nb;
    if (!circular) { spline_interval_build<T>(p, o, tab, lo, hi); return; }
    KnotTab<T> t(tab, nb);
    spline_unpack_wh<T>(p, o, t);
    if (scale != T(1)) {
        for (int j = 0; j < nb; ++j) { t.cw[j] *= scale; t.ch[j] *= scale; }
    }
    spline_cum_knots<T>(t.cw, nb, lo, hi, o.min_w, true);
    spline_cum_knots<T>(t.ch, nb, lo, hi, o.min_h, true);
    if (o.smooth) return;                                         // (the C2-smooth two-bin variant: its one derivative follows from the knots)
    const T* pd = p + o.n_w + o.n_h;
    if (o.fix_bd) {
        const T fixed = o.min_d + SM<T>::softplus(o.fix_bd_value);
        t.d[0] = fixed; t.d[nb] = fixed;
        for (int j = 1; j < nb; ++j) t.d[j] = o.min_d + SM<T>::softplus(pd[j - 1] * scale);
    } else {
        for (int j = 0; j < nb; ++j) t.d[j] = o.min_d + SM<T>::softplus(pd[j] * scale);
        t.d[nb] = t.d[0];
    }
}

// logistic function of the softplus' argument, accurate (a backward pass is not the place for the hardware approximations)
template <typename T> __device__ __forceinline__ T adj_sigmoid(T x) { return T(1) / (T(1) + M<T>::exp(-x)); }

// the unnormalised width / height j as spline_unpack_wh leaves it (before the azimuthal scale), and d (that value) / d (its raw sum)
template <typename T> struct SplineUnpack {
    const T* p; int n_w, w_start, k, independent; T ln_max; bool ratio;
    __device__ __forceinline__ SplineUnpack(const T* p_, const SplineDev<T>& o) : p(p_), n_w(o.n_w), w_start(o.fix_first ? (o.fix_second ? 2 : 1) : 0), k(o.fix_first ? 1 : 0),
        independent(o.independent), ln_max(T(0)), ratio(o.ratio > T(0)) {
        if (ratio) ln_max = (M<T>::log(o.ratio) - M<T>::log(T(o.nb - 1))) * T(0.5);
    }
    __device__ __forceinline__ T raw_w(int j) const { return j >= w_start ? p[j - w_start] : T(0); }
    __device__ __forceinline__ T raw_h(int j) const { return (j >= k ? p[n_w + j - k] : T(0)) + (independent ? raw_w(j) : T(0)); }
    __device__ __forceinline__ T squash(T v) const { return ratio ? T(2) * ln_max / (T(1) + M<T>::exp(-v)) - ln_max : v; }
    // d squash / d v from the squashed value u: u = 2 L sigma(v) - L  =>  2 L sigma (1 - sigma), sigma = (u + L) / (2 L)
    __device__ __forceinline__ T dsquash(T u) const {
        if (!ratio) return T(1);
        const T sg = (u + ln_max) / (T(2) * ln_max);
        return T(2) * ln_max * sg * (T(1) - sg);
    }
};

// ---- reverse of the table build, one row: the adjoints gk of the six knot values (cw_b, cw_{b+1}, ch_b, ch_{b+1}, d_b, d_{b+1}) of bin b -> the
// parameter row's gradient (ADDED to gp); returns d S / d scale (circular).  tab: the row's knot table (read only).
//   knot_j = lo + span sum_{i<j} frac_i (j = 1 .. nb - 1; knots 0 and nb are pinned), frac_i = rel_min + mix soft_i, mix = 1 - nb rel_min
//   A_i = d S / d frac_i = span (gk_b [i < b] + gk_{b+1} [i < b + 1]),  d S / d a_i = mix soft_i (A_i - sum_m soft_m A_m),
//   sum_m soft_m A_m = span ((gk_b + gk_{b+1}) S_b + gk_{b+1} soft_b),  S_b = sum_{i<b} soft_i = ((knot_b - lo) / span - b rel_min) / mix
template <typename T> __device__ inline T spline_adj_table_reverse(const T* __restrict__ p, T* __restrict__ gp, const SplineDev<T>& o, const T* __restrict__ tab, int b,
                                                                  const T (&gk)[6], T lo, T hi, bool circular, T scale) {
    const int nb = o.nb;
    const KnotTab<T> t(const_cast<T*>(tab), nb);
    const T span = hi - lo, inv_span = T(1) / span;
    const SplineUnpack<T> un(p, o);
    const bool scaled = circular && scale != T(1);
    T g_scale = T(0);
    // pinned knots take no gradient
    const T gw0 = b == 0 ? T(0) : gk[0], gw1 = b + 1 == nb ? T(0) : gk[1];
    const T gh0 = b == 0 ? T(0) : gk[2], gh1 = b + 1 == nb ? T(0) : gk[3];
    const T mixw = T(1) - o.min_w * T(nb), mixh = T(1) - o.min_h * T(nb);
    const T inv_mw = T(1) / mixw, inv_mh = T(1) / mixh;
    const T softw_b = ((t.cw[b + 1] - t.cw[b]) * inv_span - o.min_w) * inv_mw, softh_b = ((t.ch[b + 1] - t.ch[b]) * inv_span - o.min_h) * inv_mh;
    const T Sw_b = ((t.cw[b] - lo) * inv_span - o.min_w * T(b)) * inv_mw, Sh_b = ((t.ch[b] - lo) * inv_span - o.min_h * T(b)) * inv_mh;
    const T dotw = span * ((gw0 + gw1) * Sw_b + gw1 * softw_b), doth = span * ((gh0 + gh1) * Sh_b + gh1 * softh_b);
    T cwj = t.cw[0], chj = t.ch[0];
    for (int j = 0; j < nb; ++j) {
        const T cwn = t.cw[j + 1], chn = t.ch[j + 1];
        const T sw = ((cwn - cwj) * inv_span - o.min_w) * inv_mw, sh = ((chn - chj) * inv_span - o.min_h) * inv_mh;
        cwj = cwn; chj = chn;
        const T Aw = span * (j < b ? gw0 + gw1 : (j == b ? gw1 : T(0))), Ah = span * (j < b ? gh0 + gh1 : (j == b ? gh1 : T(0)));
        T gw = mixw * sw * (Aw - dotw), gh = mixh * sh * (Ah - doth);
        if (scaled || un.ratio) {
            const T uw = un.squash(un.raw_w(j)), uh = un.squash(un.raw_h(j));
            if (scaled) { g_scale += gw * uw + gh * uh; gw *= scale; gh *= scale; }
            gw *= un.dsquash(uw); gh *= un.dsquash(uh);
        }
        if (o.independent) gw += gh;
        if (j >= un.k) gp[o.n_w + j - un.k] += gh;
        if (j >= un.w_start) gp[j - un.w_start] += gw;
    }
    if (o.smooth) return g_scale;                                 // (no derivative parameters)
    // derivatives: d_j = min_d + softplus(raw * scale)
    T* gpd = gp + o.n_w + o.n_h;
    const T* pd = p + o.n_w + o.n_h;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int j = b + e;
        const T g = gk[4 + e];
        int idx;
        if (o.fix_bd) {
            if (j < 1 || j > nb - 1) continue;
            idx = j - 1;
        } else if (circular) {
            idx = j == nb ? 0 : j;
        } else {
            idx = j;
        }
        const T raw = pd[idx];
        const T gs = g * adj_sigmoid<T>(circular ? raw * scale : raw);
        if (circular) { gpd[idx] += gs * scale; g_scale += gs * raw; }
        else gpd[idx] += gs;
    }
    return g_scale;
}

// ---- the same from DENSE knot adjoints (gcw, gch, gd: nb + 1 each; entries of pinned knots are ignored), interval splines: what a workgroup
// does once per layer with the sums of its rows' knot adjoints when the table does not depend on the row (the values in g* are used up)
template <typename T> __device__ inline void spline_adj_table_reverse_dense(const T* __restrict__ p, T* __restrict__ gp, const SplineDev<T>& o, const T* __restrict__ tab,
                                                                           T* __restrict__ gcw, T* __restrict__ gch, const T* __restrict__ gd, T lo, T hi) {
    const int nb = o.nb;
    const KnotTab<T> t(const_cast<T*>(tab), nb);
    const T span = hi - lo, inv_span = T(1) / span;
    const SplineUnpack<T> un(p, o);
    for (int which = 1; which >= 0; --which) {                    // heights first: with `independent` their adjoints flow into the widths'
        T* gk = which ? gch : gcw;
        const T* knots = which ? t.ch : t.cw;
        const T rel_min = which ? o.min_h : o.min_w;
        const T mix = T(1) - rel_min * T(nb), inv_mix = T(1) / mix;
        T suffix = T(0), dot = T(0);
        for (int i = nb - 1; i >= 0; --i) {                       // A_i = span * sum_{j = i+1 .. nb-1} gknot_j, kept in gk[i]
            const T soft = ((knots[i + 1] - knots[i]) * inv_span - rel_min) * inv_mix;
            const T A = span * suffix;
            dot += soft * A;
            if (i >= 1) suffix += gk[i];
            gk[i] = A;
        }
        for (int i = 0; i < nb; ++i) {
            const T soft = ((knots[i + 1] - knots[i]) * inv_span - rel_min) * inv_mix;
            T g = mix * soft * (gk[i] - dot);
            if (un.ratio) g *= un.dsquash(un.squash(which ? un.raw_h(i) : un.raw_w(i)));
            gk[i] = g;
        }
        if (which) {
            for (int j = un.k; j < nb; ++j) gp[o.n_w + (j - un.k)] += gk[j];
        } else {
            if (o.independent) for (int j = 0; j < nb; ++j) gk[j] += gch[j];
            for (int j = un.w_start; j < nb; ++j) gp[j - un.w_start] += gk[j];
        }
    }
    T* gpd = gp + o.n_w + o.n_h;
    const T* pd = p + o.n_w + o.n_h;
    if (o.fix_bd) {
        for (int j = 1; j < nb; ++j) gpd[j - 1] += gd[j] * adj_sigmoid<T>(pd[j - 1]);
    } else {
        for (int j = 0; j <= nb; ++j) gpd[j] += gd[j] * adj_sigmoid<T>(pd[j]);
    }
}

// bin of a table (eps 1e-6 on the last knot, clamped for out-of-range inputs as spline_interval_eval does)
template <typename T> __device__ __forceinline__ int spline_adj_bin(const T* cw, const T* ch, int nb, T x, bool inverse) {
    int b = spline_search<T>(inverse ? ch : cw, nb, x, T(1e-6));
    return b < 0 ? 0 : (b > nb - 1 ? nb - 1 : b);
}

}  // namespace jf
