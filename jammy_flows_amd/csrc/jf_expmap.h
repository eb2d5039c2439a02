// Exponential-map flow on S2, layer 'v' (jammy_flows/layers/spheres/exponential_map_s2.py) for a lane-per-sample kernel.
//   x' = exp_x( grad phi(x) ),  phi = sum_k a_k f(mu_k . x)   (exponential / linear / quadratic potentials)
//   analytic 3x3 Jacobian in embedding space, projected on the tangent basis (t, x cross t), log|det| = 1/2 log det(J_p^T J_p)
//   inverse by damped Newton iterations on the sphere (layers/bisection_n_newton.py:330-465).
// The reference asserts float64 for this layer (exponential_map_s2.py:450, 493); only the double instantiation is exported.
#pragma once
#include "jf_manifold.h"

namespace jf {

template <typename T> struct Mat3 { T m[3][3]; };

template <typename T> struct ExpMapOut { T y[3]; T logdet_half; Mat3<T> jac; };

// mu_norm_function of the "old" mean parametrisation: generate_normalization_function(stretch 10, max 1)  (exponential_map_s2.py:32-43, 118)
template <typename T> __device__ __forceinline__ T v_mu_norm(T n) { return -M<T>::log(T(1) + T(1.718281828459045) * M<T>::exp(-n / T(10))) + T(1); }

constexpr int JF_V_SPLINE_BINS = 10;    // exponential_map_s2.py:111 (num_spline_basis_functions)

// get_exp_map_and_jacobian (exponential_map_s2.py:248-442).  pp: (n_pot, nc) row-major for this lane; tab: the lane's spline knot table
// (only touched by the "splines" potential: rows 4.. hold 10 widths, 10 heights, 11 derivatives per component, :346-388); oob: spline input
// outside [-1, 1] (the reference raises, spline_fns.py:57-59)
//
// Two stages (the backward kernel differentiates them separately, manifold_bwd_kernels.hip):
//   v_potential       (parameters, x) -> g = grad phi (3), gj = its Jacobian (3 x 3): a sum over the components, cheap
//   v_exp_geometry    (x, g, gj)      -> exp_x(g), its 3 x 3 Jacobian, 1/2 log det of the projected Jacobian: no parameters, the expensive part
template <typename T> struct VPotential { T g[3]; T gj[3][3]; };

// log-sum-exp of the components' log-weights (row 3)
// (ROW: anything indexable that yields T -- a plain pointer, or the backward kernel's seeded accessor over a row of plain values)
template <typename T, typename ROW> __device__ inline T v_lse(ROW pp, int nc) {
    const int w_row = 3;
    T lmax = pp[w_row * nc];
    for (int k = 1; k < nc; ++k) lmax = M<T>::max(lmax, pp[w_row * nc + k]);
    T lse = T(0);
    for (int k = 0; k < nc; ++k) lse += M<T>::exp(pp[w_row * nc + k] - lmax);
    return lmax + M<T>::log(lse);
}

// component k's term of grad phi and of its Jacobian, ADDED to P (the weight normaliser lse is an input: the backward kernel differentiates a
// single component with lse held fixed and adds the softmax coupling in closed form)
template <typename T, typename ROW> __device__ inline void v_component(ROW pp, int nc, int k, int kind, T lse, const T (&x)[3], VPotential<T>& P,
                                                                      T* __restrict__ tab, bool& oob) {
    const int w_row = 3, b_row = 4;
    const T m0 = pp[k], m1 = pp[nc + k], m2 = pp[2 * nc + k];
    const T nrm = M<T>::sqrt(m0 * m0 + m1 * m1 + m2 * m2);
    const T mu[3] = {m0 / nrm, m1 / nrm, m2 / nrm};
    const T w = M<T>::exp(pp[w_row * nc + k] - lse + M<T>::log(v_mu_norm<T>(nrm)));                 // :288-289
    const T xmu = x[0] * mu[0] + x[1] * mu[1] + x[2] * mu[2];
    T f, fp;   // grad contribution w * mu * f, Jacobian contribution w * fp * mu mu^T
    if (kind == JF_V_EXPONENTIAL) {
        const T beta = M<T>::exp(pp[b_row * nc + k]);
        f = M<T>::exp(beta * (xmu - T(1)));                                                    // :301
        fp = beta * f;                                                                       // :306
    } else if (kind == JF_V_LINEAR) {
        f = T(1); fp = T(0);
    } else if (kind == JF_V_SPLINES) {
        // the potential's derivative is a monotone rational-quadratic spline [-1, 1] -> [-1, 1] of mu . x (rational_quadratic_spline with
        // rel_min_bin_width = rel_min_bin_height = min_derivative = 1e-3, :354-362); f = spline value, f' = exp(logabsdet)
        constexpr int NB = JF_V_SPLINE_BINS;
        KnotTab<T> t(tab, NB);
        for (int j = 0; j < NB; ++j) { t.cw[j] = pp[(4 + j) * nc + k]; t.ch[j] = pp[(4 + NB + j) * nc + k]; }
        for (int j = 0; j <= NB; ++j) t.d[j] = T(1e-3) + softplus<T>(pp[(4 + 2 * NB + j) * nc + k]);
        spline_cum_knots<T>(t.cw, NB, T(-1), T(1), T(1e-3), true);
        spline_cum_knots<T>(t.ch, NB, T(-1), T(1), T(1e-3), true);
        oob = oob || (xmu < T(-1)) || (xmu > T(1));
        int b = spline_search<T>(t.cw, NB, xmu, T(1e-6));
        b = b < 0 ? 0 : (b > NB - 1 ? NB - 1 : b);
        const SplineOut<T> r = spline_core<T>(t, b, xmu, false);
        f = r.y; fp = M<T>::exp(r.lad);
    } else {
        f = xmu; fp = T(1);                                                                  // :332-335
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        P.g[i] += w * mu[i] * f;
#pragma unroll
        for (int j = 0; j < 3; ++j) P.gj[i][j] += w * fp * mu[i] * mu[j];
    }
}

template <typename T, typename ROW> __device__ inline void v_potential(ROW pp, int nc, int kind, const T (&x)[3], VPotential<T>& P, T* __restrict__ tab,
                                                                      bool& oob) {
    const T lse = v_lse<T>(pp, nc);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        P.g[i] = T(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) P.gj[i][j] = T(0);
    }
    for (int k = 0; k < nc; ++k) v_component<T>(pp, nc, k, kind, lse, x, P, tab, oob);
}

template <typename T> __device__ inline void v_exp_geometry(int kind, const T (&x)[3], const VPotential<T>& P, ExpMapOut<T>& o) {
    const T (&g)[3] = P.g;
    const T (&gj)[3][3] = P.gj;
    // unnormalized_logarithmic_map with Jacobians (:163-219)
    const T tn = M<T>::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    T nt[3], tv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) nt[i] = g[i] / tn;
    const T ca = nt[0] * x[0] + nt[1] * x[1] + nt[2] * x[2];
    const T alpha = M<T>::acos(ca);
    const T sa = M<T>::sin(alpha);
#pragma unroll
    for (int i = 0; i < 3; ++i) tv[i] = (nt[i] - x[i] * ca) / sa;
    const T proj = g[0] * tv[0] + g[1] * tv[1] + g[2] * tv[2];
    const T inv_sq = T(-1) / M<T>::sqrt(T(1) - ca * ca);
    T jt[3][3], jp[3];
    // d tangent / d base (+ chain through theta)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T dth = (x[i] - nt[i] * ca) / (sa * sa);
#pragma unroll
        for (int j = 0; j < 3; ++j) jt[i][j] = (i == j ? -ca / sa : T(0)) + dth * (inv_sq * nt[j]);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) jp[j] = jt[0][j] * g[0] + jt[1][j] * g[1] + jt[2][j] * g[2];
    if (kind != JF_V_LINEAR) {
        // d normalised target / d unnormalised target, then the two chain terms through the gradient's own Jacobian gj
        T dn[3][3], a[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) dn[i][j] = (-g[i] / (tn * tn)) * nt[j] + (i == j ? T(1) / tn : T(0));
        // a = dn @ gj
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) a[i][j] = dn[i][0] * gj[0][j] + dn[i][1] * gj[1][j] + dn[i][2] * gj[2][j];
        // row vector r = (inv_sq * x) @ a ; jt += dth_i * r_j + a_ij / sa
        T r[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) r[j] = inv_sq * (x[0] * a[0][j] + x[1] * a[1][j] + x[2] * a[2][j]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const T dth = (x[i] - nt[i] * ca) / (sa * sa);
#pragma unroll
            for (int j = 0; j < 3; ++j) jt[i][j] += dth * r[j] + a[i][j] / sa;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) jp[j] += tv[0] * gj[0][j] + tv[1] * gj[1][j] + tv[2] * gj[2][j];
    }
    const T cp = M<T>::cos(proj), sp = M<T>::sin(proj);
#pragma unroll
    for (int i = 0; i < 3; ++i) o.y[i] = x[i] * cp + tv[i] * sp;                                 // basic_exponential_map (:153-161)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o.jac.m[i][j] = (i == j ? cp : T(0)) + (-x[i] * sp) * jp[j] + jt[i][j] * sp + (tv[i] * cp) * jp[j];      // :419-427
    // project on the tangent basis (tv, x cross tv) and take 1/2 log det(P^T P)  (:431-442, 474-478)
    const T t2[3] = {x[1] * tv[2] - x[2] * tv[1], x[2] * tv[0] - x[0] * tv[2], x[0] * tv[1] - x[1] * tv[0]};
    T c1[3], c2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        c1[i] = o.jac.m[i][0] * tv[0] + o.jac.m[i][1] * tv[1] + o.jac.m[i][2] * tv[2];
        c2[i] = o.jac.m[i][0] * t2[0] + o.jac.m[i][1] * t2[1] + o.jac.m[i][2] * t2[2];
    }
    const T a11 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
    const T a22 = c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2];
    const T a12 = c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2];
    o.logdet_half = T(0.5) * M<T>::log(M<T>::abs(a11 * a22 - a12 * a12));
}

template <typename T> __device__ inline void v_exp_map(const T* __restrict__ pp, int nc, int kind, const T (&x)[3], ExpMapOut<T>& o, T* __restrict__ tab,
                                                      bool& oob) {
    VPotential<T> P;
    v_potential<T>(pp, nc, kind, x, P, tab, oob);
    v_exp_geometry<T>(kind, x, P, o);
}

// basic_logarithmic_map (exponential_map_s2.py:221-244): unit tangent at `base` towards `target`, angle alpha (0 when already there)
template <typename T> __device__ __forceinline__ void v_log_map(const T (&base)[3], const T (&target)[3], T (&tv)[3], T& alpha) {
    T ca = target[0] * base[0] + target[1] * base[1] + target[2] * base[2];
    const bool conv = ca >= T(1);
    T b[3] = {base[0], base[1], base[2]};
    if (conv) { b[0] = T(1); b[1] = T(0); b[2] = T(0); ca = target[0]; }
    alpha = M<T>::acos(ca);
    const T sa = M<T>::sin(alpha);
#pragma unroll
    for (int i = 0; i < 3; ++i) tv[i] = (target[i] - b[i] * ca) / sa;
    if (conv) alpha = T(0);
}

// Newton on the sphere: solve exp-map(x) = target.
//   fast = inverse_bisection_n_newton_sphere_fast (bisection_n_newton.py:394-465): damping 0.4, per-row stop |step| < 1e-12
//   slow = inverse_bisection_n_newton_sphere      (:330-391): damping 0.1, stops when max over the BATCH < 1e-12 -- a batch-global
//          criterion; here every row stops on its own step (rows that would have kept iterating only because another row of the batch
//          was still moving change by < 1e-12 per further step).
template <typename T> __device__ inline void v_newton(const T* __restrict__ pp, int nc, int kind, const T (&target)[3], int max_iter, bool fast,
                                                     bool lane_valid, T (&x)[3], T* __restrict__ tab, bool& oob) {
    // start AT the target: the layer is exp_x(grad phi(x)) with a bounded potential gradient, i.e. a perturbation of the identity, so the image of
    // the target is usually already within the Gauss-Newton radius (fn < 0.1) and the damped approach from the south pole (the reference's
    // start: ~10 evaluations until fn < 0.1) is skipped; the map is a diffeomorphism of the sphere, the root is the same (round 4: 1.29 -> see
    // DESIGN 3.3b)
    {
        const T tn2 = target[0] * target[0] + target[1] * target[1] + target[2] * target[2];
        const bool ok = tn2 > T(0.25) && tn2 < T(4);
        const T inv = ok ? T(1) / M<T>::sqrt(tn2) : T(0);
        x[0] = ok ? target[0] * inv : T(0); x[1] = ok ? target[1] * inv : T(0); x[2] = ok ? target[2] * inv : T(-1);
    }
    bool active = lane_valid;
    const T damp = fast ? T(0.4) : T(0.1);
    bool gn_ok = true, last_gn = false;
    T fn_prev = T(INFINITY);
    ExpMapOut<T> o;
    for (int it = 0; it < max_iter && __any(active); ++it) {
        v_exp_map<T>(pp, nc, kind, x, o, tab, oob);
        if (active) {
            const T fn = T(1) - (o.y[0] * target[0] + o.y[1] * target[1] + o.y[2] * target[2]);
            T rv[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) rv[j] = -(o.jac.m[0][j] * target[0] + o.jac.m[1][j] * target[1] + o.jac.m[2][j] * target[2]);
            const T gn = M<T>::sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
            const T tg[3] = {-rv[0] / gn, -rv[1] / gn, -rv[2] / gn};
            T nv[3], alpha;
            v_log_map<T>(x, tg, nv, alpha);
            const T gp = nv[0] * rv[0] + nv[1] * rv[1] + nv[2] * rv[2];
            T step = -(fn / gp);
            if (fast && alpha == T(0)) step = T(0);
            if (!fast && !(step >= T(1e-12))) { active = false; continue; }      // reference breaks BEFORE applying the step (:384-389)
            // The objective fn = 1 - cos(angle(y, target)) has a double root, so the reference's update (a damped gradient step of Polyak
            // length: 0.4 / 0.1 x fn / |grad|) shrinks the error by ~0.8 / ~0.95 per evaluation: 120 / 540 evaluations of the map until its
            // 1e-12 rule fires.  Same iteration here until the image is within 0.45 rad of the target (fn < 0.1); from there ONE
            // Gauss-Newton step per evaluation on the 2-d system y(x) = target (least squares of J delta = target - y over the tangent plane
            // at x) converges quadratically to the same root -- 3 evaluations instead of ~100 / ~430 -- and the reference's stopping rule then
            // ends the row.  A step that would be long (> 0.5 rad) falls back to the damped update, and a row whose Gauss-Newton step did not reduce fn
            // keeps the reference's iteration for good (none of the fixtures / fuzz cases needs either guard).
            T arc = damp * step;
            bool gn_done = false;
            // a Gauss-Newton step that did not bring the image closer (above the rounding noise of fn): damped updates from here on
            if (last_gn && !(fn < fn_prev) && fn > T(1e-12)) gn_ok = false;
            fn_prev = fn;
            last_gn = false;
            if (gn_ok && fn < T(1e-1)) {
                const int ax = (M<T>::abs(x[0]) <= M<T>::abs(x[1]) && M<T>::abs(x[0]) <= M<T>::abs(x[2])) ? 0 : (M<T>::abs(x[1]) <= M<T>::abs(x[2]) ? 1 : 2);
                const T a3[3] = {ax == 0 ? T(1) : T(0), ax == 1 ? T(1) : T(0), ax == 2 ? T(1) : T(0)};
                T e1[3] = {a3[1] * x[2] - a3[2] * x[1], a3[2] * x[0] - a3[0] * x[2], a3[0] * x[1] - a3[1] * x[0]};
                const T n1 = T(1) / M<T>::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
#pragma unroll
                for (int i = 0; i < 3; ++i) e1[i] *= n1;
                const T e2[3] = {x[1] * e1[2] - x[2] * e1[1], x[2] * e1[0] - x[0] * e1[2], x[0] * e1[1] - x[1] * e1[0]};
                T A1[3], A2[3], r[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    A1[i] = o.jac.m[i][0] * e1[0] + o.jac.m[i][1] * e1[1] + o.jac.m[i][2] * e1[2];
                    A2[i] = o.jac.m[i][0] * e2[0] + o.jac.m[i][1] * e2[1] + o.jac.m[i][2] * e2[2];
                    r[i] = target[i] - o.y[i];
                }
                const T g11 = A1[0] * A1[0] + A1[1] * A1[1] + A1[2] * A1[2], g12 = A1[0] * A2[0] + A1[1] * A2[1] + A1[2] * A2[2];
                const T g22 = A2[0] * A2[0] + A2[1] * A2[1] + A2[2] * A2[2];
                const T b1 = A1[0] * r[0] + A1[1] * r[1] + A1[2] * r[2], b2 = A2[0] * r[0] + A2[1] * r[1] + A2[2] * r[2];
                const T det = g11 * g22 - g12 * g12;
                const T c1 = (g22 * b1 - g12 * b2) / det, c2 = (g11 * b2 - g12 * b1) / det;
                const T len = M<T>::sqrt(c1 * c1 + c2 * c2);
                if (len > T(0) && len < T(0.5)) {                  // (NaN fails both comparisons: damped update)
                    arc = len;
                    last_gn = true;
#pragma unroll
                    for (int i = 0; i < 3; ++i) nv[i] = (c1 * e1[i] + c2 * e2[i]) / len;
                    // a Gauss-Newton step below the reference's own 1e-12 threshold ends the row: at the root fn = 1 - y.t is rounding noise
                    // (+-1 ulp), and fn / |grad| -- the reference's criterion, which its slow approach reaches with fn rounding to exactly 0
                    // -- no longer measures the distance
                    if (len < T(1e-12)) gn_done = true;
                } else if (len == T(0)) {
                    gn_done = true;
                }
            }
            const T cs = M<T>::cos(arc), sn = M<T>::sin(arc);
#pragma unroll
            for (int i = 0; i < 3; ++i) x[i] = x[i] * cs + nv[i] * sn;
            if (fast) active = M<T>::abs(step) >= T(1e-12);
            if (gn_done) active = false;
        }
    }
}

struct VFam {
    using CLayer = jf_v_layer;
    static constexpr int DIM = 2;
    static __host__ int n_pot(const CLayer& L) {
        return L.exp_map_type == JF_V_SPLINES ? 4 + 3 * JF_V_SPLINE_BINS + 1 : 3 + (L.exp_map_type == JF_V_EXPONENTIAL ? 2 : 1);
    }
    static __host__ bool sane(const CLayer& L) { return L.num_components >= 1 && L.num_components <= 4096 && sane_hh(L.hh_iter); }
    static __host__ int row_len(const CLayer& L) { return rot_len(L.hh_iter, 3) + n_pot(L) * L.num_components; }
    static __host__ int n_bins(const CLayer&) { return 0; }
    static __host__ bool needs_tab(const CLayer& L) { return L.exp_map_type == JF_V_SPLINES; }

    // the two ends of the log-prob direction around the exponential map (exponential_map_s2.py:446-487): rotation + angles -> embedding, and
    // embedding -> angles (+ the first layer's chart).  Separate functions because the backward kernel differentiates the three stages apart.
    template <typename T> static __device__ __forceinline__ void inv_pre(const CLayer& L, const T* __restrict__ p, T (&x)[3], T& ld, T (&e)[3]) {
        if (L.hh_iter != 0) s2_rotate<T>(p, L.hh_iter, x, ld, true);
        s2_to_eucl<T>(x[0], x[1], e, ld);
    }
    template <typename T> static __device__ __forceinline__ void inv_post(const CLayer& L, const T (&y)[3], T (&x)[3], T& ld) {
        T th, ph;
        eucl_to_s2<T>(y, th, ph, ld);
        if (L.first) {
            T pl[3];
            s2_to_plane<T>(th, ph, pl, ld);
            x[0] = pl[0]; x[1] = pl[1];
        } else { x[0] = th; x[1] = ph; }
    }

    template <typename T, bool FWD> static __device__ __forceinline__ void apply(const CLayer& L, const T* __restrict__ p, T (&x)[3], T& ld, LaneCtx<T>& c) {
        const T* pp = p + rot_len(L.hh_iter, 3);
        bool oob = false;
        const int nc = L.num_components, kind = L.exp_map_type;
        T e[3], th, ph;
        ExpMapOut<T> o;
        if constexpr (FWD) {
            if (L.first) {
                T pl[3] = {x[0], x[1], T(0)};
                plane_to_s2<T>(pl, x[0], x[1], ld);
            }
            s2_to_eucl<T>(x[0], x[1], e, ld);                                        // exponential_map_s2.py:495-499
            if (L.natural_direction) {
                v_exp_map<T>(pp, nc, kind, e, o, c.tab, oob);
                ld += o.logdet_half;
#pragma unroll
                for (int i = 0; i < 3; ++i) e[i] = o.y[i];
            } else {
                T r[3];
                v_newton<T>(pp, nc, kind, e, L.max_newton_iter, true, c.lane_valid, r, c.tab, oob);
                v_exp_map<T>(pp, nc, kind, r, o, c.tab, oob);
                ld -= o.logdet_half;
#pragma unroll
                for (int i = 0; i < 3; ++i) e[i] = r[i];
            }
            eucl_to_s2<T>(e, th, ph, ld);
            x[0] = th; x[1] = ph;
            if (L.hh_iter != 0) s2_rotate<T>(p, L.hh_iter, x, ld, false);
        } else {
            inv_pre<T>(L, p, x, ld, e);                                              // :459-460
            if (L.natural_direction) {
                T r[3];
                v_newton<T>(pp, nc, kind, e, L.max_newton_iter, false, c.lane_valid, r, c.tab, oob);
                v_exp_map<T>(pp, nc, kind, r, o, c.tab, oob);
                ld -= o.logdet_half;
#pragma unroll
                for (int i = 0; i < 3; ++i) e[i] = r[i];
            } else {
                v_exp_map<T>(pp, nc, kind, e, o, c.tab, oob);
                ld += o.logdet_half;
#pragma unroll
                for (int i = 0; i < 3; ++i) e[i] = o.y[i];
            }
            inv_post<T>(L, e, x, ld);
        }
        bool bad = !M<T>::finite(x[0]) || !M<T>::finite(x[1]);
        c.nonfinite = c.nonfinite || bad;
        c.oob = c.oob || oob;
    }
};

}  // namespace jf
