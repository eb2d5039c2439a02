// Lane-per-sample device code of the manifold flow layers:
//   'r'  jammy_flows/layers/intervals/rational_quadratic_spline.py:180-400  (+ interval_base.py:61-79)
//   'o'  jammy_flows/layers/spheres/splines_1d.py:111-306
//   'm'  jammy_flows/layers/spheres/moebius_1d.py:57-259  (+ layers/bisection_n_newton.py:137-256)
//   'f'  jammy_flows/layers/spheres/fvm_2d.py:273-726
//   'v'  jammy_flows/layers/spheres/exponential_map_s2.py:153-528  (+ layers/bisection_n_newton.py:330-465)
// wrapped by the sphere_base rotation / chart steps (jammy_flows/layers/spheres/sphere_base.py:601-695).
// Every family exposes  apply<FWD>(layer, row, x[3], log_det, ctx)  on INTRINSIC coordinates.
#pragma once
#include "jf_sphere.h"
#include "jf_spline.h"

namespace jf {

constexpr int JF_CORR_SCRATCH = 81;   // odd stride; emitted parameters + rank accumulators of the correlated MLP

template <typename T> struct LaneCtx {
    T* tab;              // lane-private LDS scratch (JF_SPLINE_TAB elements)
    T* corr;             // lane-private LDS scratch for per-sample emitted parameters (JF_CORR_SCRATCH elements, 'f' correlated only)
    int64_t* bins;       // this row's bin slots (nullable)
    int bin_i;
    bool oob, nonconv, nonfinite;
    bool lane_valid;
    bool tab_built = false;   // `tab` is the WORKGROUP's table of the current layer, already built (broadcast parameters: Fam::build, manifold_kernels.hip)
    __device__ __forceinline__ void put_bin(int b) {
        if (bins) bins[bin_i] = (int64_t)b;
        ++bin_i;
    }
};

template <typename T> __device__ __forceinline__ SplineDev<T> to_dev(const jf_spline_opts& s) {
    SplineDev<T> o;
    o.nb = s.num_bins; o.smooth = s.smooth; o.fix_first = s.fix_first; o.fix_second = s.fix_second; o.independent = s.independent;
    o.fix_bd = s.fix_bd; o.n_w = s.n_w; o.n_h = s.n_h; o.n_d = s.n_d;
    o.fix_bd_value = (T)s.fix_bd_value; o.min_w = (T)s.min_w; o.min_h = (T)s.min_h; o.min_d = (T)s.min_d; o.ratio = (T)s.ratio;
    return o;
}
__host__ __device__ inline int spline_row_len(const jf_spline_opts& s) { return s.n_w + s.n_h + s.n_d; }

// descriptor sanity (host): field ranges checked BEFORE any row-length arithmetic, so that nonsense descriptors (a fuzzer's, a corrupted one's)
// end in JF_ERR_BADARG instead of integer overflow or an out-of-bounds index into the nested layer arrays (tests/test_abi_asan.py)
__host__ inline bool sane_hh(int hh) { return hh >= JF_ROT_QUATERNION && hh <= 64; }
__host__ inline bool sane_spline(const jf_spline_opts& s) {
    return s.num_bins >= 1 && s.num_bins <= JF_SPLINE_CAP && s.n_w >= 0 && s.n_w <= 4 * JF_SPLINE_CAP && s.n_h >= 0 &&
           s.n_h <= 4 * JF_SPLINE_CAP && s.n_d >= 0 && s.n_d <= 4 * JF_SPLINE_CAP;
}

// per-lane knot-table elements a layer needs: the family's own count where it states one ('r', 'o', 'f': 3 (bins + 1)), else the fixed stride
template <class Fam, class = void> struct fam_tab_words { static int of(const typename Fam::CLayer&) { return JF_SPLINE_TAB; } };
template <class Fam> struct fam_tab_words<Fam, std::void_t<decltype(&Fam::tab_words)>> { static int of(const typename Fam::CLayer& L) { return Fam::tab_words(L); } };

// =================================================================================================  'r'
template <typename T> __device__ __forceinline__ T r_core(const jf_r_layer& L, const T* __restrict__ p, T x, T& ld, LaneCtx<T>& c, bool inverse) {
    x = x > T(1) ? T(1) : (x < T(-1) ? T(-1) : x);                         // rational_quadratic_spline.py:185-186, 295-296
    bool oob;
    const SplineDev<T> o = to_dev<T>(L.sp);
    const SplineOut<T> r = spline_interval<T>(p, o, c.tab, x, inverse, (T)L.lo, (T)L.hi, oob, c.tab_built);
    c.oob = c.oob || oob;
    c.put_bin(r.bin);
    ld += r.lad;
    return r.y > T(1) ? T(1) : (r.y < T(-1) ? T(-1) : r.y);
}
struct RFam {
    using CLayer = jf_r_layer;
    static constexpr int DIM = 1;
    // the layer's knot table does not depend on the row: with broadcast parameters one lane builds it for the workgroup
    static constexpr bool HAS_BUILD = true;
    template <typename T> static __device__ __forceinline__ void build(const CLayer& L, const T* __restrict__ p, T* __restrict__ tab) {
        spline_interval_build<T>(p, to_dev<T>(L.sp), tab, (T)L.lo, (T)L.hi);
    }
    static __host__ bool sane(const CLayer& L) { return sane_spline(L.sp); }
    static __host__ int row_len(const CLayer& L) { return spline_row_len(L.sp); }
    static __host__ int n_bins(const CLayer&) { return 1; }
    static __host__ bool needs_tab(const CLayer&) { return true; }       // lane-private knot tables (tab_words elements of LDS per lane)
    static __host__ int tab_words(const CLayer& L) { return spline_tab_words(L.sp.num_bins); }
    template <typename T, bool FWD> static __device__ __forceinline__ void apply(const CLayer& L, const T* __restrict__ p, T (&x)[3], T& ld, LaneCtx<T>& c) {
        if constexpr (FWD) {
            if (L.first) x[0] = real_line_to_interval<T>(x[0], (T)L.lo, (T)L.hi, ld);        // interval_base.py:71-79
            x[0] = r_core<T>(L, p, x[0], ld, c, false);
        } else {
            x[0] = r_core<T>(L, p, x[0], ld, c, true);
            if (L.first) x[0] = interval_to_real_line<T>(x[0], (T)L.lo, (T)L.hi, ld);        // interval_base.py:61-69
        }
    }
};

// =================================================================================================  S1 helpers
template <typename T> __device__ __forceinline__ T s1_rotate(const T* __restrict__ vs, int hh, T phi, bool transpose) {
    T e[3];
    s1_to_eucl<T>(phi, e);
    rotate_embed<T, 2>(vs, hh, e, transpose);
    return eucl_to_s1<T>(e);
}

template <typename T> __device__ __forceinline__ void s2_rotate(const T* __restrict__ vs, int hh, T (&x)[3], T& ld, bool transpose) {
    T e[3];
    s2_to_eucl<T>(x[0], x[1], e, ld);
    rotate_embed<T, 3>(vs, hh, e, transpose);
    eucl_to_s2<T>(e, x[0], x[1], ld);
}

// =================================================================================================  'o'
template <typename T> __device__ __forceinline__ T o_core(const jf_o_layer& L, const T* __restrict__ p, T x, T& ld, LaneCtx<T>& c, bool fwd, T scale) {
    // splines_1d.py:119-120/198-199 (inverse: safe clamp) vs :219-221/298-300 (forward: hard clip)
    if (fwd) x = x >= M<T>::TWO_PI ? M<T>::TWO_PI : (x < T(0) ? T(0) : x);
    else x = safe_angle_2pi<T>(x);
    const bool use_inverse = fwd ? (L.natural_direction == 0) : (L.natural_direction != 0);
    bool oob;
    const SplineDev<T> o = to_dev<T>(L.sp);
    const SplineOut<T> r = spline_circular<T>(p, o, c.tab, x, use_inverse, scale, oob);
    c.oob = c.oob || oob;
    c.put_bin(r.bin);
    ld += r.lad;
    if (fwd) return r.y >= M<T>::TWO_PI ? M<T>::TWO_PI : (r.y < T(0) ? T(0) : r.y);
    return safe_angle_2pi<T>(r.y);
}
struct OFam {
    using CLayer = jf_o_layer;
    static constexpr int DIM = 1;
    static __host__ bool sane(const CLayer& L) { return sane_spline(L.sp) && sane_hh(L.hh_iter); }
    static __host__ int row_len(const CLayer& L) { return rot_len(L.hh_iter, 2) + spline_row_len(L.sp); }
    static __host__ int n_bins(const CLayer&) { return 1; }
    static __host__ bool needs_tab(const CLayer&) { return true; }       // lane-private knot tables (tab_words elements of LDS per lane)
    static __host__ int tab_words(const CLayer& L) { return spline_tab_words(L.sp.num_bins); }
    template <typename T, bool FWD> static __device__ __forceinline__ void apply(const CLayer& L, const T* __restrict__ p, T (&x)[3], T& ld, LaneCtx<T>& c) {
        const T* sp = p + rot_len(L.hh_iter, 2);
        if constexpr (FWD) {
            if (L.first) x[0] = plane_to_s1<T>(x[0], ld);
            x[0] = o_core<T>(L, sp, x[0], ld, c, true, T(1));
            if (L.hh_iter != 0) x[0] = s1_rotate<T>(p, L.hh_iter, x[0], false);
        } else {
            if (L.hh_iter != 0) x[0] = s1_rotate<T>(p, L.hh_iter, x[0], true);
            x[0] = o_core<T>(L, sp, x[0], ld, c, false, T(1));
            if (L.first) x[0] = s1_to_plane<T>(x[0], ld);
        }
    }
};

// =================================================================================================  'm'
// moebius_1d.py:140-259.  x in [-pi, pi].  value: sum_k pi_k * arg(Moebius_k(e^{ix})) normalised so that -pi -> -pi; deriv: sum_k pi_k (1-|w|^2)/|e^{ix}-w|^2
// np: parameters per component, 4 = (omega_x, omega_y, logit length, log weight), 3 = (omega angle, logit length, log weight) (:175-178)
// (ROW: anything indexable that yields T -- a plain pointer, or the values of a dual row, DualValues)
// one component at e^{ix} = (cx, sx) given its (np - 1) shape parameters: the normalised arc and the derivative's ratio (1 - |w|^2) / |e^{ix} - w|^2
// (T may carry tangents: the adjoint evaluates a component on DualN<T, 4> -- x and its three parameters, jf_manifold_adj.h)
template <typename T> __device__ __forceinline__ void moebius_component(int np, T q0, T q1, T logit_len, T cx, T sx, T& arc, T& ratio) {
    const T cmp = T(-1), smp = (T)(-1.2246467991473532e-16);          // numpy.cos(-pi), numpy.sin(-pi)
    const T denom = logaddexp<T>(T(0), -logit_len);
    const T len = T(0.001) + M<T>::exp(T(-0.0020020026706730793) - denom);      // ln(0.999 - 0.001)
    T ox, oy;
    if (np == 4) {
        const T nrm = len / M<T>::sqrt(q0 * q0 + q1 * q1);
        ox = q0 * nrm; oy = q1 * nrm;
    } else {
        ox = M<T>::cos(q0) * len; oy = M<T>::sin(q0) * len;
    }
    const T omo = T(1) - len * len;
    const T opo = T(1) + len * len - T(2) * (cx * ox + sx * oy);
    const T opo_mp = T(1) + len * len - T(2) * (cmp * ox + smp * oy);
    const T y_mp = omo * (smp - oy) - oy * opo_mp;
    const T x_mp = omo * (cmp - ox) - ox * opo_mp;
    const T rot = -M<T>::PI - M<T>::atan2(y_mp, x_mp);
    const T yv = omo * (sx - oy) - oy * opo;
    const T xv = omo * (cx - ox) - ox * opo;
    const T cr = M<T>::cos(rot), sr = M<T>::sin(rot);
    arc = M<T>::atan2(sr * xv + cr * yv, cr * xv - sr * yv) + M<T>::PI;
    ratio = omo / opo;
}
template <typename T, typename ROW> __device__ inline void moebius_eval(ROW p, int nc, int np, T x, T& val, T& deriv) {
    const T cx = M<T>::cos(x), sx = M<T>::sin(x);
    T lmax = p[np - 1];
    for (int k = 1; k < nc; ++k) lmax = M<T>::max(lmax, p[np * k + np - 1]);
    T wsum = T(0), vsum = T(0), dsum = T(0);
    for (int k = 0; k < nc; ++k) {
        const int q = np * k;
        T arc, ratio;
        moebius_component<T>(np, p[q], np == 4 ? T(p[q + 1]) : T(0), p[q + np - 2], cx, sx, arc, ratio);
        const T w = M<T>::exp(p[q + np - 1] - lmax);
        wsum += w;
        vsum += w * arc;
        dsum += w * ratio;
    }
    val = vsum / wsum - M<T>::PI;
    deriv = dsum / wsum;
}
// the iteration of bisection_n_newton.py:171-238 on plain values
template <typename T, typename ROW> __device__ inline T moebius_solve_values(ROW p, int nc, int np, T z, bool lane_valid, bool& nonconv, bool& nonfinite) {
    T lo = -M<T>::PI, hi = M<T>::PI, x = T(0), f, d;
    for (int it = 0; it < 20; ++it) {                                     // bisection_n_newton.py:171-182
        x = (hi + lo) * T(0.5);
        moebius_eval<T>(p, nc, np, x, f, d);
        if (M<T>::abs(f - z) <= T(1e-6) * M<T>::abs(z)) { lo = x; hi = x; }
        else if (f < z) lo = x;
        else hi = x;
    }
    bool active = lane_valid;
    T ferr = T(0);
    for (int it = 0; it < 20 && __any(active); ++it) {                    // :192-238
        moebius_eval<T>(p, nc, np, x, f, d);
        if (active) {
            const T upd = (f - z) / d;
            x -= upd;
            ferr = M<T>::abs(f - z);
            active = M<T>::abs(upd) >= newton_tol<T>();
            // float32 (the absolute 1e-14 only fires on an exactly zero update: all 20 steps ran, 19 of them on rounding noise): the floor of
            // the 'g' solvers (jf_math.h) -- at the coordinate's resolution, or below 1e-5 of it with the residual inside the 1e-4 threshold
            if (sizeof(T) == 4 && !newton_reference_rule()) {
                const T xs = M<T>::max(M<T>::abs(x), T(1));
                if (M<T>::abs(upd) < T(2.5e-7) * xs || (M<T>::abs(upd) < T(JF_F32_NEWTON_FLOOR) * xs && ferr <= T(1e-4))) active = false;
            }
        }
    }
    nonconv = nonconv || (ferr > (sizeof(T) == 8 ? T(1e-7) : T(1e-4)));
    nonfinite = nonfinite || !M<T>::finite(x);
    return x;
}
// Scalar types with tangents (the backward kernels' dual numbers): the iteration runs on the VALUES, and the solution's tangents follow from
// ONE evaluation on dual numbers at the solution -- f(x, p) = z  =>  dx = (dz - df|_x) / f'(x) -- instead of 40 evaluations that drag the
// tangents through every bisection and Newton step (round 5: the per-sample `m` adjoint took 34 x its forward).
template <typename T> __device__ inline T moebius_solve(const T* __restrict__ p, int nc, int np, T z, LaneCtx<T>& c) {
    if constexpr (DualTraits<T>::is_dual) {
        using V = typename DualTraits<T>::value_type;
        const V xv = moebius_solve_values<V>(DualValues<T>{p}, nc, np, DualTraits<T>::value(z), c.lane_valid, c.nonconv, c.nonfinite);
        T f, d;
        moebius_eval<T>(p, nc, np, T(xv), f, d);
        return DualTraits<T>::implicit(xv, z, f, DualTraits<T>::value(d));
    } else {
        return moebius_solve_values<T>(p, nc, np, z, c.lane_valid, c.nonconv, c.nonfinite);
    }
}
struct MFam {
    using CLayer = jf_m_layer;
    static constexpr int DIM = 1;
    static __host__ __device__ int omega_pars(const CLayer& L) { return L.omega_pars == 3 ? 3 : 4; }
    static __host__ bool sane(const CLayer& L) { return L.num_components >= 1 && L.num_components <= 4096 && sane_hh(L.hh_iter); }
    static __host__ int row_len(const CLayer& L) { return rot_len(L.hh_iter, 2) + omega_pars(L) * L.num_components; }
    static __host__ int n_bins(const CLayer&) { return 0; }
    static __host__ bool needs_tab(const CLayer&) { return false; }
    template <typename T> static __device__ __forceinline__ T core(const CLayer& L, const T* __restrict__ mp, T x, T& ld, LaneCtx<T>& c, bool direct) {
        x = x > M<T>::PI ? x - M<T>::TWO_PI : x;                           // moebius_1d.py:73-74
        T val, d;
        if (direct) {
            moebius_eval<T>(mp, L.num_components, omega_pars(L), x, val, d);
            ld += M<T>::log(d);
            x = val;
        } else {
            x = moebius_solve<T>(mp, L.num_components, omega_pars(L), x, c);
            moebius_eval<T>(mp, L.num_components, omega_pars(L), x, val, d);
            ld -= M<T>::log(d);
        }
        return x < T(0) ? M<T>::TWO_PI + x : x;
    }
    template <typename T, bool FWD> static __device__ __forceinline__ void apply(const CLayer& L, const T* __restrict__ p, T (&x)[3], T& ld, LaneCtx<T>& c) {
        const T* mp = p + rot_len(L.hh_iter, 2);
        if constexpr (FWD) {
            if (L.first) x[0] = plane_to_s1<T>(x[0], ld);
            x[0] = core<T>(L, mp, x[0], ld, c, L.natural_direction != 0);
            if (L.hh_iter != 0) x[0] = s1_rotate<T>(p, L.hh_iter, x[0], false);
        } else {
            if (L.hh_iter != 0) x[0] = s1_rotate<T>(p, L.hh_iter, x[0], true);
            x[0] = core<T>(L, mp, x[0], ld, c, L.natural_direction == 0);
            if (L.first) x[0] = s1_to_plane<T>(x[0], ld);
        }
    }
};

// =================================================================================================  base-class steps only
// rotation (sphere_base.py:607-624 / 678-693) and / or the first-layer chart (interval_base.py:61-79, sphere_base.py:641-667): the identity
// layers 'y' / 'z' and the generic sphere_base / interval_base wrappers around third-party subclasses.
struct CFam {
    using CLayer = jf_c_layer;
    static constexpr int DIM = 2;      // interval / S1 use column 0 only (the host passes dim)
    static __host__ bool sane(const CLayer& L) { return L.kind >= 0 && L.kind <= 2 && sane_hh(L.hh_iter); }
    static __host__ int row_len(const CLayer& L) { return rot_len(L.hh_iter, L.kind == 2 ? 3 : 2); }
    static __host__ int n_bins(const CLayer&) { return 0; }
    static __host__ bool needs_tab(const CLayer&) { return false; }
    template <typename T, bool FWD> static __device__ __forceinline__ void apply(const CLayer& L, const T* __restrict__ p, T (&x)[3], T& ld, LaneCtx<T>& c) {
        if constexpr (FWD) {
            if (L.first) {
                if (L.kind == 0) x[0] = real_line_to_interval<T>(x[0], (T)L.lo, (T)L.hi, ld);
                else if (L.kind == 1) x[0] = plane_to_s1<T>(x[0], ld);
                else { T pl[3] = {x[0], x[1], T(0)}; plane_to_s2<T>(pl, x[0], x[1], ld); }
            }
            if (L.hh_iter != 0) {
                if (L.kind == 1) x[0] = s1_rotate<T>(p, L.hh_iter, x[0], false);
                else if (L.kind == 2) s2_rotate<T>(p, L.hh_iter, x, ld, false);
            }
        } else {
            if (L.hh_iter != 0) {
                if (L.kind == 1) x[0] = s1_rotate<T>(p, L.hh_iter, x[0], true);
                else if (L.kind == 2) s2_rotate<T>(p, L.hh_iter, x, ld, true);
            }
            if (L.first) {
                if (L.kind == 0) x[0] = interval_to_real_line<T>(x[0], (T)L.lo, (T)L.hi, ld);
                else if (L.kind == 1) x[0] = s1_to_plane<T>(x[0], ld);
                else { T pl[3]; s2_to_plane<T>(x[0], x[1], pl, ld); x[0] = pl[0]; x[1] = pl[1]; }
            }
        }
    }
};

// =================================================================================================  'f'
template <typename T> __device__ __forceinline__ T azimuthal_scaling(T c) {     // fvm_2d.py:267-271
    const T c3 = c * c * c, c4 = c3 * c, c5 = c4 * c;
    return c <= T(0) ? T(6) * c5 + T(15) * c4 + T(10) * c3 + T(1) : T(-6) * c5 + T(15) * c4 - T(10) * c3 + T(1);
}
struct FFam {
    using CLayer = jf_f_layer;
    static constexpr int DIM = 2;
    static __host__ __device__ int corr_out(const CLayer& L) {      // parameters the correlated MLP emits = rows of the circular layers incl. their rotations
        int n = 0;
        for (int i = 0; i < L.n_circular; ++i) n += rot_len(L.circular[i].hh_iter, 2) + spline_row_len(L.circular[i].sp);
        return n;
    }
    static __host__ __device__ int corr_len(const CLayer& L) {      // U/V/b vector of the MLP 1 -> H -> n_out (amortizable_mlp.py:284-375)
        const int H = L.corr_hidden, n = corr_out(L);
        return 2 * H + (L.corr_full2 ? n * H : L.corr_rank * (n + H)) + n;
    }
    // add_extra_rotation_inbetween (fvm_2d.py:381-402, 664-688): e -> M e (sampling) / M^T e (log-prob) with M = [[0,0,1],[0,1,0],[-1,0,0]], taken
    // through the angle <-> embedding conversions and their log-dets exactly as the reference does
    template <typename T> static __device__ __forceinline__ void inbetween(T& cos_theta, T& angle, T& ld, bool inverse) {
        T th = M<T>::acos(cos_theta);
        ld -= M<T>::log(M<T>::sin(safe_angle_pi<T>(th)));
        T e[3];
        s2_to_eucl<T>(th, angle, e, ld);
        T r[3];
        if (inverse) { r[0] = -e[2]; r[1] = e[1]; r[2] = e[0]; }
        else { r[0] = e[2]; r[1] = e[1]; r[2] = -e[0]; }
        eucl_to_s2<T>(r, th, angle, ld);
        cos_theta = M<T>::cos(th);
        ld += M<T>::log(M<T>::sin(safe_angle_pi<T>(th)));
    }
    // kappa of the von-Mises-Fisher step (fvm_2d.py:105-139, 289-330): from its own parameter (modes 0-2, optionally clamped) or from the
    // length of the layer's rotation parameters ("mu" / "mu_squared" with rotation_mode xyz, "quatvec" / "quatvec_squared" with quaternion)
    static __host__ __device__ int n_kappa(const CLayer& L) { return L.kappa_mode <= JF_F_KAPPA_LOG_BOUNDED ? 1 : 0; }
    template <typename T> static __device__ __forceinline__ T kappa_of(const CLayer& L, const T* __restrict__ rot, const T* __restrict__ fp) {
        const T mk = (T)L.min_kappa;
        switch (L.kappa_mode) {
            case JF_F_KAPPA_DIRECT_LOG: { const T v = L.kappa_clamping ? M<T>::max(fp[0], T(-5)) : fp[0]; return M<T>::exp(v) + mk; }
            case JF_F_KAPPA_SOFTPLUS: { const T v = L.kappa_clamping ? M<T>::max(fp[0], T(-5)) : fp[0]; return softplus(v) + mk; }
            case JF_F_KAPPA_LOG_BOUNDED: {
                T v = softplus(fp[0]);
                if (L.kappa_clamping) v = M<T>::max(v, T(-5));
                return M<T>::exp(v + M<T>::log(mk));
            }
            case JF_F_KAPPA_MU: return M<T>::sqrt(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2]);
            case JF_F_KAPPA_MU_SQUARED: return rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2];
            case JF_F_KAPPA_QUATVEC: return M<T>::sqrt(rot[1] * rot[1] + rot[2] * rot[2] + rot[3] * rot[3]);
            default: return rot[1] * rot[1] + rot[2] * rot[2] + rot[3] * rot[3];
        }
    }
    static __host__ bool sane(const CLayer& L) {
        if (!sane_hh(L.hh_iter) || L.n_vertical < 0 || L.n_vertical > JF_MAX_NESTED || L.n_circular < 0 || L.n_circular > JF_MAX_NESTED) return false;
        if (L.corr_hidden < 0 || L.corr_hidden > 65536 || L.corr_rank < 0 || L.corr_rank > 65536) return false;
        for (int i = 0; i < L.n_vertical; ++i) if (!sane_spline(L.vertical[i].sp)) return false;
        for (int i = 0; i < L.n_circular; ++i) if (!sane_spline(L.circular[i].sp) || !sane_hh(L.circular[i].hh_iter)) return false;
        return true;
    }
    static __host__ int row_len(const CLayer& L) {
        int n = rot_len(L.hh_iter, 3) + n_kappa(L);
        for (int i = 0; i < L.n_vertical; ++i) n += spline_row_len(L.vertical[i].sp);
        if (L.correlated) return n + corr_len(L);
        for (int i = 0; i < L.n_circular; ++i) n += spline_row_len(L.circular[i].sp);
        return n;
    }
    static __host__ int n_bins(const CLayer& L) { return L.n_vertical + L.n_circular; }
    static __host__ bool needs_tab(const CLayer& L) { return L.n_vertical + L.n_circular > 0; }
    static __host__ int tab_words(const CLayer& L) {                    // the nested splines are evaluated one after the other on the lane's one table
        int w = 1;
        for (int i = 0; i < L.n_vertical && i < JF_MAX_NESTED; ++i) { const int t = spline_tab_words(L.vertical[i].sp.num_bins); w = t > w ? t : w; }
        for (int i = 0; i < L.n_circular && i < JF_MAX_NESTED; ++i) { const int t = spline_tab_words(L.circular[i].sp.num_bins); w = t > w ? t : w; }
        return w;
    }
    static __host__ int scratch(const CLayer& L) { return L.correlated ? JF_CORR_SCRATCH : 0; }

    // the per-sample MLP of the correlated variant: out[0..n_out) = W2 tanh(W1 z + b1) + b2 with this row's own weights
    // (_apply_amortized_mlp, amortizable_mlp.py:508-578); out / the rank accumulators live in the lane's LDS scratch
    template <typename T> static __device__ inline void corr_mlp(const CLayer& L, const T* __restrict__ mp, T z, T* __restrict__ out) {
        const int H = L.corr_hidden, R = L.corr_rank, n = corr_out(L);
        const T* W1 = mp;
        const T* b1 = mp + H;
        const T* s2 = mp + 2 * H;
        if (L.corr_full2) {
            const T* b2 = s2 + n * H;
            for (int i = 0; i < n; ++i) out[i] = b2[i];
            for (int j = 0; j < H; ++j) {
                const T h = M<T>::tanh(W1[j] * z + b1[j]);
                for (int i = 0; i < n; ++i) out[i] += s2[i * H + j] * h;
            }
        } else {
            const T* V = s2 + n * R;
            const T* b2 = V + R * H;
            T* t = out + n;
            for (int r = 0; r < R; ++r) t[r] = T(0);
            for (int j = 0; j < H; ++j) {
                const T h = M<T>::tanh(W1[j] * z + b1[j]);
                for (int r = 0; r < R; ++r) t[r] += V[r * H + j] * h;
            }
            for (int i = 0; i < n; ++i) {
                T acc = T(0);
                for (int r = 0; r < R; ++r) acc += s2[i * R + r] * t[r];
                out[i] = acc + b2[i];
            }
        }
    }

    // log-prob direction, head: rotation, the von-Mises-Fisher z step and the optional quarter turn -> (cos theta, azimuth) that the nested spline
    // flows act on (fvm_2d.py:336-402); tail: back to angles and, for the first layer of its block, the plane chart (:434-470).  Separate
    // functions because the adjoint (jf_manifold_adj.h) differentiates them stage by stage.
    template <typename T> static __device__ __forceinline__ void inv_head(const CLayer& L, const T* __restrict__ p, T (&x)[3], T& ld, T& ret_out, T& angle_out) {
        const T* fp = p + rot_len(L.hh_iter, 3);
        const T zs = (T)L.z_sign;
        const T kappa = kappa_of<T>(L, p, fp);                                               // fvm_2d.py:105-139, 289-330
        if (L.hh_iter != 0) s2_rotate<T>(p, L.hh_iter, x, ld, true);
        const T prev = M<T>::cos(x[0]);
        ld += M<T>::log(M<T>::sin(safe_angle_pi<T>(x[0])));
        const T e2k = M<T>::exp(T(-2) * kappa);
        const T safe = kappa < T(100) ? M<T>::log(M<T>::expm1(T(2) * kappa)) : T(2) * kappa;          // :352-357 (expm1: a kappa of 1e-10
                                                                                                              // -- log_bounded with the default min_kappa -- survives float32)
        ld += M<T>::log(T(2) * kappa) + kappa * (zs * prev + T(1)) - safe;
        T ret = zs * ((T(1) + e2k - T(2) * M<T>::exp(kappa * (zs * prev - T(1)))) / (T(-1) + e2k));       // :361
        if (kappa < M<T>::KAPPA_ID) ret = prev;
        ret = safe_cos<T>(ret, M<T>::EPS_COS);
        T angle = x[1];
        if (L.extra_rotation) inbetween<T>(ret, angle, ld, true);
        ret_out = ret; angle_out = angle;
    }
    template <typename T> static __device__ __forceinline__ void inv_tail(const CLayer& L, T ret, T angle, T (&x)[3], T& ld) {
        ret = safe_cos<T>(ret, M<T>::EPS_COS);
        const T th = M<T>::acos(ret);
        ld -= M<T>::log(M<T>::sin(safe_angle_pi<T>(th)));
        if (L.first) {
            T pl[3];
            s2_to_plane<T>(th, angle, pl, ld);
            x[0] = pl[0]; x[1] = pl[1];
        } else { x[0] = th; x[1] = angle; }
    }

    template <typename T, bool FWD> static __device__ __forceinline__ void apply(const CLayer& L, const T* __restrict__ p, T (&x)[3], T& ld, LaneCtx<T>& c) {
        const T* fp = p + rot_len(L.hh_iter, 3);
        const T zs = (T)L.z_sign, region = (T)L.identity_region;
        const T kappa = kappa_of<T>(L, p, fp);                                               // fvm_2d.py:105-139, 289-330
        int nv = 0;
        for (int i = 0; i < L.n_vertical; ++i) nv += spline_row_len(L.vertical[i].sp);
        const T* vert = fp + n_kappa(L);
        const T* circ = vert + nv;
        if constexpr (!FWD) {
            T ret, angle;
            inv_head<T>(L, p, x, ld, ret, angle);
            const bool inside = (region == T(0)) || ((ret > T(-1) + region) && (ret < T(1) - region));
            if (L.correlated) {                                                             // :406-409: nested i1+s1 passthrough pdf, inverse direction
                if (inside) {
                    const T z_in = ret;                                                     // block 1 is conditioned on block 0's TARGET value
                    int off = nv;
                    for (int i = L.n_vertical - 1; i >= 0; --i) {
                        off -= spline_row_len(L.vertical[i].sp);
                        ret = r_core<T>(L.vertical[i], vert + off, ret, ld, c, true);
                    }
                    corr_mlp<T>(L, vert + nv, z_in, c.corr);
                    int coff = corr_out(L);
                    for (int i = L.n_circular - 1; i >= 0; --i) {
                        coff -= rot_len(L.circular[i].hh_iter, 2) + spline_row_len(L.circular[i].sp);
                        T xx[3] = {angle, T(0), T(0)};
                        OFam::apply<T, false>(L.circular[i], c.corr + coff, xx, ld, c);
                        angle = xx[0];
                    }
                } else {
                    for (int i = 0; i < L.n_vertical + L.n_circular; ++i) c.put_bin(-2);
                }
            } else {
            if (L.n_circular > 0) {                                                         // :416-427 (layers in reverse, tail-first)
                const T sc = azimuthal_scaling<T>(ret);
                int off = 0;
                for (int i = 0; i < L.n_circular; ++i) off += spline_row_len(L.circular[i].sp);
                for (int i = L.n_circular - 1; i >= 0; --i) {
                    off -= spline_row_len(L.circular[i].sp);
                    if (inside) angle = o_core<T>(L.circular[i], circ + off, angle, ld, c, false, sc);
                    else c.put_bin(-2);
                }
            }
            if (L.n_vertical > 0) {                                                         // :430-432
                int off = nv;
                for (int i = L.n_vertical - 1; i >= 0; --i) {
                    off -= spline_row_len(L.vertical[i].sp);
                    if (inside) ret = r_core<T>(L.vertical[i], vert + off, ret, ld, c, true);
                    else c.put_bin(-2);
                }
            }
            }
            inv_tail<T>(L, ret, angle, x, ld);
        } else {
            if (L.first) {
                T pl[3] = {x[0], x[1], T(0)};
                plane_to_s2<T>(pl, x[0], x[1], ld);
            }
            T prev = M<T>::cos(x[0]);
            ld += M<T>::log(M<T>::sin(safe_angle_pi<T>(x[0])));
            T angle = x[1];
            const bool inside = (region == T(0)) || ((prev > T(-1) + region) && (prev < T(1) - region));
            if (L.correlated) {                                                             // :575-578: sampling direction
                if (inside) {
                    int off = 0;
                    for (int i = 0; i < L.n_vertical; ++i) {
                        prev = r_core<T>(L.vertical[i], vert + off, prev, ld, c, false);
                        off += spline_row_len(L.vertical[i].sp);
                    }
                    corr_mlp<T>(L, vert + nv, prev, c.corr);                                // conditioned on the PRODUCED target of block 0
                    int coff = 0;
                    for (int i = 0; i < L.n_circular; ++i) {
                        T xx[3] = {angle, T(0), T(0)};
                        OFam::apply<T, true>(L.circular[i], c.corr + coff, xx, ld, c);
                        angle = xx[0];
                        coff += rot_len(L.circular[i].hh_iter, 2) + spline_row_len(L.circular[i].sp);
                    }
                } else {
                    for (int i = 0; i < L.n_vertical + L.n_circular; ++i) c.put_bin(-2);
                }
            } else {
            if (L.n_vertical > 0) {                                                         // :591-592 (layers in order, head-first)
                int off = 0;
                for (int i = 0; i < L.n_vertical; ++i) {
                    if (inside) prev = r_core<T>(L.vertical[i], vert + off, prev, ld, c, false);
                    else c.put_bin(-2);
                    off += spline_row_len(L.vertical[i].sp);
                }
            }
            if (L.n_circular > 0) {                                                         // :596-607
                const T sc = azimuthal_scaling<T>(prev);
                int off = 0;
                for (int i = 0; i < L.n_circular; ++i) {
                    if (inside) angle = o_core<T>(L.circular[i], circ + off, angle, ld, c, true, sc);
                    else c.put_bin(-2);
                    off += spline_row_len(L.circular[i].sp);
                }
            }
            }
            if (L.extra_rotation) inbetween<T>(prev, angle, ld, false);
            ld -= M<T>::log(kappa * zs * prev + kappa / M<T>::tanh(kappa));                   // :698
            T ret = zs * (T(1) + (T(1) / kappa) * M<T>::log(T(0.5) * (T(1) + zs * prev) + (T(0.5) - T(0.5) * zs * prev) * M<T>::exp(T(-2) * kappa)));
            if (kappa < M<T>::KAPPA_ID) ret = prev;
            ret = safe_cos<T>(ret, M<T>::EPS_COS);
            x[0] = M<T>::acos(ret);
            ld -= M<T>::log(M<T>::sin(safe_angle_pi<T>(x[0])));
            x[1] = angle;
            if (L.hh_iter != 0) s2_rotate<T>(p, L.hh_iter, x, ld, false);
        }
    }
};

}  // namespace jf
