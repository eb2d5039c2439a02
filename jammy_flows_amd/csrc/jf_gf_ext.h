// General-option device code of the 'g' layer: the rotation parametrisations other than Householder reflections ("angles" = Givens
// rotations, "cayley", "triangular_combination"; gaussianization_flow.py:711-799, 942-989, 1004-1049), center_mean (:846-852) and the
// skewed-logistic components (add_skewness, :352-368, 411-442 with extra_functions.log_one_plus_exp_x_to_a_minus_1, extra_functions.py:14-61).
//
// None of these is a default of the reference and none lies on the benchmarked path, so this code trades speed for generality: one lane per
// row, the row's coordinates in a lane-private LDS column (dynamic indexing without scratch), the parameters read straight from the row in
// HBM, the mixture in log space exactly as the reference writes it (stream log-sum-exp over the components).  It is correct for EVERY option
// combination of the classic stretch, so the kernel built on it (gfx_chain_kernel, gf_kernels.hip) also serves as the catch-all.
#pragma once
#include "jf_gf.h"

namespace jf {

constexpr int GX_THREADS = 128;
constexpr int JF_MAX_D_GF = 8;           // the general-option kernel (this file) takes D <= 8
constexpr int JF_MAX_D_G = 64;           // the lane = (row, coordinate) kernels: groups of up to 64 lanes (a whole wave) per row

template <typename T> struct XCol {           // the lane's coordinate vector: element d at b[d * GX_THREADS]
    T* b;
    __device__ __forceinline__ T& operator[](int d) const { return b[d * GX_THREADS]; }
};

__host__ __device__ inline int gx_lower_index(int D, int i, int j) { const int ind = D - 1 - (i - j); return ind * (ind + 1) / 2 + j; }   // matrix_fns.py:33-49

__host__ inline int gx_rot_len(int mode, int hh, int D) {
    if (mode == JF_GF_ROT_HOUSEHOLDER) return hh * D;
    if (D < 2) return 0;
    if (mode == JF_GF_ROT_ANGLES) return D * (D - 1) / 2;
    if (mode == JF_GF_ROT_CAYLEY) return 1;
    return D - 1 + D * (D - 1);              // triangular_combination (:161)
}

// The parameter row is a template parameter PA of everything below: `const T*` in the forward kernels, and in the backward kernel a proxy
// (SeededRow, gf_bwd_kernels.hip) that hands out dual numbers whose tangent is 1 at the seeded parameter -- the only operations used on it are
// `p[i]` and `p + offset`.

// x <- R x (inverse == false, sampling direction :942-987) or x <- R^{-1} x (log-prob direction :1004-1049)
template <typename T, typename PA> __device__ inline void gx_rotate(const GfLayerDev<T>& o, PA p, XCol<T> x, int D, bool inverse) {
    const PA rp = p + o.off_rot;
    if (o.rot_mode == JF_GF_ROT_HOUSEHOLDER) {
        // Q = H_0 H_1 ... (:457-471): Q^T x applies H_0 first, Q x applies H_{n-1} first
        for (int it = 0; it < o.hh; ++it) {
            const PA v = rp + (inverse ? it : o.hh - 1 - it) * D;
            T n2 = T(0), dot = T(0);
            for (int d = 0; d < D; ++d) { n2 += v[d] * v[d]; dot += v[d] * x[d]; }
            const T f = T(2) * dot / n2;
            for (int d = 0; d < D; ++d) x[d] -= f * v[d];
        }
        return;
    }
    if (D < 2) return;
    if (o.rot_mode == JF_GF_ROT_ANGLES) {
        // R = G_{n-1} ... G_0 over itertools.combinations(range(D), 2); G[a][a] = G[b][b] = cos, G[a][b] = sin, G[b][a] = -sin (:760-780)
        const int n = D * (D - 1) / 2;
        for (int it = 0; it < n; ++it) {
            const int ind = inverse ? n - 1 - it : it;
            int a = 0, rem = ind;
            while (rem >= D - 1 - a) { rem -= D - 1 - a; ++a; }
            const int b = a + 1 + rem;
            const T c = M<T>::cos(rp[ind]);
            const T s = inverse ? -M<T>::sin(rp[ind]) : M<T>::sin(rp[ind]);
            const T xa = x[a], xb = x[b];
            x[a] = c * xa + s * xb;
            x[b] = c * xb - s * xa;
        }
        return;
    }
    if (o.rot_mode == JF_GF_ROT_CAYLEY) {     // R = [[c, -s], [s, c]], c = (1 - t^2)/(1 + t^2), s = 2t/(1 + t^2)  (:793-798)
        const T t = rp[0], m = T(1) / (T(1) + t * t);
        const T c = (T(1) - t * t) * m, s = (inverse ? T(-2) : T(2)) * t * m;
        const T x0 = x[0], x1 = x[1];
        x[0] = c * x0 - s * x1;
        x[1] = s * x0 + c * x1;
        return;
    }
    // triangular_combination: x <- L diag(e^d) U x with unit-diagonal L (lower) and U (upper = transposed lower layout), sum(d) = 0
    const int nt = D * (D - 1) / 2;
    const PA lower = rp;
    const PA diag = rp + nt;
    const PA upper = rp + (nt + D - 1);
    T dsum = T(0);
    if (!inverse) {
        for (int i = 0; i < D; ++i) {                      // U x (row i uses x_j, j > i: ascending i reads not-yet-overwritten entries)
            T acc = x[i];
            for (int j = i + 1; j < D; ++j) acc += upper[gx_lower_index(D, j, i)] * x[j];
            x[i] = acc;
        }
        for (int i = 0; i < D; ++i) {
            const T dv = i < D - 1 ? diag[i] : -dsum;
            if (i < D - 1) dsum += dv;
            x[i] *= M<T>::exp(dv);
        }
        for (int i = D - 1; i >= 0; --i) {                 // L x
            T acc = x[i];
            for (int j = 0; j < i; ++j) acc += lower[gx_lower_index(D, i, j)] * x[j];
            x[i] = acc;
        }
    } else {
        for (int i = 0; i < D; ++i) {                      // L^{-1} x: forward substitution
            T acc = x[i];
            for (int j = 0; j < i; ++j) acc -= lower[gx_lower_index(D, i, j)] * x[j];
            x[i] = acc;
        }
        for (int i = 0; i < D; ++i) {
            const T dv = i < D - 1 ? diag[i] : -dsum;
            if (i < D - 1) dsum += dv;
            x[i] *= M<T>::exp(-dv);
        }
        for (int i = D - 1; i >= 0; --i) {                 // U^{-1} x: back substitution
            T acc = x[i];
            for (int j = i + 1; j < D; ++j) acc -= upper[gx_lower_index(D, j, i)] * x[j];
            x[i] = acc;
        }
    }
}

// generate_log_function_bounded_in_logspace (gaussianization_flow.py:23-47)
template <typename T> __device__ __forceinline__ T gx_bounded_log(T x, T ln_min, T ln_max, bool center) {
    const T first = ln_max - softplus<T>(-x + (center ? ln_max : T(0)));
    return logaddexp<T>(first, ln_min);
}

// log( ((1 + e^x)^a - 1) / (1 + e^x)^a ), branch for branch as extra_functions.py:31-61
template <typename T> __device__ __forceinline__ T gx_log_one_minus_pow(T x, T a) {
    const T sp = a * softplus<T>(x);
    T res;
    if (x <= T(-20)) res = M<T>::log(a) + x;
    else if (sp > T(20)) res = sp;
    else if (sp < T(1e-8)) res = M<T>::log(sp);
    else res = M<T>::log(M<T>::expm1(sp));
    return res - sp;
}

template <typename T> struct Lse {            // stream log-sum-exp
    T m, s;
    __device__ __forceinline__ Lse() : m(T(-INFINITY)), s(T(0)) {}
    __device__ __forceinline__ void add(T v) {
        if (v > m) { s = s * M<T>::exp(m - v) + T(1); m = v; }
        else s += M<T>::exp(v - m);
    }
    __device__ __forceinline__ T value() const { return m + M<T>::log(s); }
};

// regulated log-weight of component k (:840-844), 0 without fit_normalization
template <typename T, typename PA> __device__ __forceinline__ T gx_log_weight(const GfLayerDev<T>& o, PA p, int D, int k, int d) {
    if (!o.fit_norm) return T(0);
    const T raw = p[o.off_ln + k * D + d];
    return o.reg_norm ? M<T>::log(o.nmin + o.nmax / (T(1) + M<T>::exp(-raw))) : raw;
}

// per-coordinate quantities that do not depend on x: log-sum-exp of the log-weights and, with center_mean, the dependent last mean
template <typename T> struct GxCoord { T lse_w, last_mean; };
template <typename T, typename PA> __device__ inline GxCoord<T> gx_prepare(const GfLayerDev<T>& o, PA p, int D, int d) {
    GxCoord<T> c;
    Lse<T> l;
    T acc = T(0), wl = T(1);
    for (int k = 0; k < o.K; ++k) {
        const T lw = gx_log_weight<T, PA>(o, p, D, k, d);
        l.add(lw);
        if (o.center_mean) {
            const T w = M<T>::exp(lw);
            if (k < o.K - 1) acc += p[o.off_mean + k * D + d] * w; else wl = w;
        }
    }
    c.lse_w = l.value();
    c.last_mean = -acc / wl;
    return c;
}

// one component's terms of the three log-sum-exps (log cdf, log sf, log pdf) of logistic_kernel_log_pdf_quantities (:389-454); T may carry tangents
// (the reverse sweep evaluates a component on DualN<T, 4>: x and its mean, raw log-width, raw log-exponent -- gf_rev_kernels.hip)
template <typename T> __device__ __forceinline__ void gx_component(const GfLayerDev<T>& o, T x, T mu, T raw_lw, T raw_skew, bool pos, T ln_pi, T& tc, T& ts, T& tp) {
    const T ln9 = T(2.19722457733621938279), ln01 = T(-2.30258509299404568402);
    const T w = gf_width<T>(o, raw_lw);
    const T logw = M<T>::log(w);
    const T u = (x - mu) / w;
    if (o.skew) {
        const T log_a = gx_bounded_log<T>(raw_skew, ln01, ln9, true);      // exponent regulator (:367)
        const T a = M<T>::exp(log_a);
        const T su = pos ? u : -u;
        tp = -su - logw + log_a - (a + T(1)) * softplus<T>(-su) + ln_pi;
        if (pos) {
            tc = -a * softplus<T>(-u) + ln_pi;
            ts = gx_log_one_minus_pow<T>(-u, a) + ln_pi;
        } else {
            tc = gx_log_one_minus_pow<T>(u, a) + ln_pi;
            ts = -a * softplus<T>(u) + ln_pi;
        }
    } else {
        const T sp = softplus<T>(-u);
        tp = -u - logw - T(2) * sp + ln_pi;
        tc = -sp + ln_pi;
        ts = -u - sp + ln_pi;
    }
}
// the inverse-CDF stage's input from the three log-sum-exps
template <typename T> __device__ __forceinline__ MixQ<T> gx_mixq(const GfLayerDev<T>& o, T lc, T ls, T lp) {
    MixQ<T> q;
    q.lc = lc; q.ls = ls; q.lp = lp;
    q.cdf = M<T>::exp(q.lc); q.sf = M<T>::exp(q.ls);
    // The skewed components' cdf and sf come from two different closed forms whose shortcuts (extra_functions.py:40-45: the "- 1" is dropped
    // beyond softplus > 20) leave cdf + sf = 1 + O(2e-9).  The reference's central inverse-normal branch reads the cdf only (erfinv(2 cdf - 1),
    // :505-515), this code reads the smaller of the two: hand it the reference's complement so that both see the same number.
    if (o.skew && q.cdf > T(0.5)) q.sf = T(1) - q.cdf;
    return q;
}

// logistic_kernel_log_pdf_quantities (:389-454) for one coordinate
template <typename T, typename PA> __device__ inline MixQ<T> gx_mixture(const GfLayerDev<T>& o, PA p, int D, int d, const GxCoord<T>& c, T x) {
    Lse<T> lc, ls, lp;
    const int n_pos = o.K / 2;                                                   // (:356-359): the first int(K/2) components keep sign +1
    for (int k = 0; k < o.K; ++k) {
        const T mu = (o.center_mean && k == o.K - 1) ? c.last_mean : p[o.off_mean + k * D + d];
        const T ln_pi = gx_log_weight<T, PA>(o, p, D, k, d) - c.lse_w;
        T tc, ts, tp;
        gx_component<T>(o, x, mu, p[o.off_lw + k * D + d], o.skew ? T(p[o.off_skew + k * D + d]) : T(0), k < n_pos, ln_pi, tc, ts, tp);
        lp.add(tp);
        lc.add(tc);
        ls.add(ts);
    }
    return gx_mixq<T>(o, lc.value(), ls.value(), lp.value());
}

// approach + Newton of the sampling direction (layers/bisection_n_newton.py:11-135; 25 / 20 iterations on [-1e5, 1e5], :921) for one row:
// z holds the targets, x receives the solution; the Newton stopping rule sums |update| over the row's coordinates.  The approach phase is
// gf_approach (jf_gf.h) on this mixture's log cdf / log sf / log pdf from the mixture's mean -- 4-6 evaluations where the reference's 25
// bisections (restated here through round 5, and still what the audit library runs) made 25: the general-option fixtures sampled at 15-22 x
// their forward pass (profiles/r06_scan_fixtures.md), this phase being 25 of ~31 evaluations.
template <typename T, typename PA> __device__ inline void gx_solve(const GfLayerDev<T>& o, PA p, int D, XCol<T> z, XCol<T> x, bool row_valid,
                                                                   int32_t* status) {
    for (int d = 0; d < D; ++d) {
        const GxCoord<T> c = gx_prepare<T, PA>(o, p, D, d);
        const T zd = z[d];
        if (newton_reference_rule()) {
            T lo = T(-1e5), hi = T(1e5), xm = T(0);
            for (int it = 0; it < 25; ++it) {
                xm = (hi + lo) * T(0.5);
                const T y = gf_icdf<T>(o.inv_type, gx_mixture<T, PA>(o, p, D, d, c, xm)).y;
                if (M<T>::abs(y - zd) <= T(1e-6) * M<T>::abs(zd)) { lo = xm; hi = xm; }
                else if (y < zd) lo = xm;
                else hi = xm;
            }
            x[d] = xm;
        } else {
            T x0 = T(0);
            for (int k = 0; k < o.K; ++k) {
                const T mu = (o.center_mean && k == o.K - 1) ? c.last_mean : T(p[o.off_mean + k * D + d]);
                x0 += M<T>::exp(gx_log_weight<T, PA>(o, p, D, k, d) - c.lse_w) * mu;
            }
            if (!M<T>::finite(x0)) x0 = T(0);
            x[d] = gf_approach<T>([&](T xx) { return gx_mixture<T, PA>(o, p, D, d, c, xx); }, o.inv_type != JF_GF_ISIGMOID, zd, x0, row_valid);
        }
    }
    T ferr = T(0);
    bool nonfinite = false;
    bool active = row_valid;
    T prev = T(INFINITY);
    for (int it = 0; it < 20 && __any(active); ++it) {
        status_add(status, JF_STATUS_NEWTON_STEPS, active);
        if (!active) continue;
        T usum = T(0), xs = T(0);
        ferr = T(0);
        for (int d = 0; d < D; ++d) {
            const GxCoord<T> c = gx_prepare<T, PA>(o, p, D, d);
            const IcdfOut<T> s = gf_icdf<T>(o.inv_type, gx_mixture<T, PA>(o, p, D, d, c, x[d]));
            const T f = s.y - z[d];
            const T upd = f / M<T>::exp(s.logd);
            usum += M<T>::abs(upd);
            const T nx = x[d] - upd;
            if (M<T>::finite(nx)) x[d] = nx; else nonfinite = true;     // keep the previous iterate (:84-91)
            ferr = M<T>::max(ferr, M<T>::abs(f));
            xs += M<T>::max(M<T>::abs(x[d]), T(1));
        }
        active = usum >= newton_tol<T>();
        if (sizeof(T) == 4 && !newton_reference_rule()) {          // float32 rounding floor, as in gfg_solve (jf_gf.h)
            const bool done = usum < T(2.5e-7) * xs || (usum < T(JF_F32_NEWTON_FLOOR) * xs && ferr <= T(1e-4));                  // (jf_math.h)
            if (done || (usum >= T(0.5) * prev && usum < T(1e-4) * xs)) active = false;
            prev = usum;
        }
    }
    const T prec = sizeof(T) == 8 ? T(1e-7) : T(1e-4);
    status_add(status, JF_STATUS_NONCONVERGED, row_valid && (ferr > prec));
    status_add(status, JF_STATUS_NONFINITE, row_valid && nonfinite);
}

}  // namespace jf
