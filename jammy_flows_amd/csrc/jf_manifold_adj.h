// Reverse-mode adjoints of the manifold layers in the log-prob direction ('r', 'o', 'm', 'f'), round 6.
//
// What torch.autograd returns for one layer's inv_flow_mapping (rational_quadratic_spline.py:282-400, splines_1d.py:111-194,
// moebius_1d.py:57-139, fvm_2d.py:273-483 + the sphere_base / interval_base wrappers, sphere_base.py:601-695): given d S / d (x_out, log_det)
// -> d S / d x_in and the gradient of the layer's parameter row.
//
// Until round 5 the backward kernel replayed the WHOLE chain on dual numbers once per group of four input directions (every parameter of
// every layer is a direction): fine for the dozen parameters of a default 'f', 34 x ... 800 x the forward for parameter-rich layers
// (splines, Moebius mixtures, the correlated 'f').  Here a chain is differentiated layer by layer in reverse (mchain_rev_kernel keeps every
// layer's input from a plain forward sweep), and a layer stage by stage:
//   * stages with few inputs and few parameters (rotations, charts, the von-Mises-Fisher z step, C2-smooth splines of <= 3 bins) keep the
//     dual-number evaluation of the forward kernels' own functions, but over THAT stage's directions only (stage_dual_adjoint);
//   * parameter-rich stages are reversed by hand: the spline family (jf_spline_adj.h), the Moebius mixture (a component's parameters
//     enter through that component's term alone: one 4-tangent evaluation per component), the per-sample MLP of the correlated 'f'.
// The dual-number replay of the whole chain stays available (JF_M_BWD_DUAL=1) as the check of everything in this file.
#pragma once
#include "jf_manifold.h"
#include "jf_spline_adj.h"

namespace jf {

// tangents per pass of the local dual-number stages: six in float32 (the S1 rotation with its input -- 1 + 4 directions --, a C2-smooth circular
// spline with its input and scale -- 2 + 4 --, the head of a default 'f' -- 2 + 10 -- take one / one / two passes), four in float64 (registers)
template <typename T> struct AdjN { static constexpr int value = sizeof(T) == 4 ? 6 : 4; };
#define ADJ_N (AdjN<T>::value)

// a lane's working memory (LDS) and flags
template <typename T> struct AdjLane {
    T* scr;                       // plain: one spline's knot table (spline_tab_words of the chain's largest bin count)
    T* corr;                      // plain: correlated 'f' -- emitted parameters + rank accumulators (JF_CORR_SCRATCH), then their gradients (JF_CORR_SCRATCH)
    DualN<T, ADJ_N>* drow;        // dual: a stage's parameter row
    DualN<T, ADJ_N>* dtab;        // dual: knot table of a C2-smooth spline (spline_tab_words(3))
    bool lane_valid, oob, nonconv, nonfinite;
};

// ---- generic stage: S = sum_d gy[d] y_d + gld * ld with (y, ld) = f(x_in[0 .. dim_in), p[0 .. n));  gx_in[j] = d S / d x_in[j], gp[j] += d S / d p[j].
// f(const Du* row, Du (&x)[3], Du& ld, LaneCtx<Du>& c): reads x as its inputs, leaves its outputs there, ADDS to ld.
template <typename T, class F>
__device__ inline void stage_dual_adjoint(F&& f, const T* __restrict__ p, int n, T* __restrict__ gp, int dim_in, const T (&xin)[3], const T (&gy)[3], T gld,
                                          T (&gxin)[3], AdjLane<T>& A) {
    using Du = DualN<T, ADJ_N>;
    Du* drow = A.drow;
    for (int j = 0; j < n; ++j) drow[j] = Du(p[j]);
    LaneCtx<Du> dc;
    dc.tab = A.dtab; dc.corr = nullptr; dc.bins = nullptr; dc.bin_i = 0;
    dc.oob = dc.nonconv = dc.nonfinite = false;
    dc.lane_valid = A.lane_valid;
    const int n_dir = dim_in + n;
#pragma unroll 1
    for (int j0 = 0; j0 < n_dir; j0 += ADJ_N) {
#pragma unroll
        for (int c = 0; c < ADJ_N; ++c) { const int j = j0 + c; if (j >= dim_in && j < n_dir) drow[j - dim_in].d[c] = T(1); }
        Du x[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            x[d] = Du(xin[d]);
#pragma unroll
            for (int c = 0; c < ADJ_N; ++c) if (d < dim_in && d == j0 + c) x[d].d[c] = T(1);
        }
        Du ld(T(0));
        f(const_cast<const Du*>(drow), x, ld, dc);
#pragma unroll
        for (int c = 0; c < ADJ_N; ++c) {
            const int j = j0 + c;
            if (j >= n_dir) break;
            const T g = gld * ld.d[c] + gy[0] * x[0].d[c] + gy[1] * x[1].d[c] + gy[2] * x[2].d[c];
            if (j < dim_in) gxin[j] = g;
            else gp[j - dim_in] += g;
        }
#pragma unroll
        for (int c = 0; c < ADJ_N; ++c) { const int j = j0 + c; if (j >= dim_in && j < n_dir) drow[j - dim_in].d[c] = T(0); }
    }
    A.oob = A.oob || dc.oob; A.nonconv = A.nonconv || dc.nonconv; A.nonfinite = A.nonfinite || dc.nonfinite;
}

// ================================================================================================= spline cores
// r_core (inverse direction) and o_core (log-prob direction, fwd = false) in two phases, so that what follows them in the layer (charts, further
// splines) can be differentiated in between: *_fwd builds the table in A.scr and evaluates the bin on seven tangents, *_bwd contracts and
// reverses the table.  Nothing else may use A.scr between the two.
template <typename T> struct SplineTape {
    T dy[7], dl[7];                // d (y, lad) / d (x, cw_b, cw_{b+1}, ch_b, ch_{b+1}, d_b, d_{b+1})
    int b;
    bool cl_in, cl_out;
    T y;                           // the core's (clamped) output
};

template <typename T> __device__ inline T spline_adj_core7(const T* cw, const T* ch, const T* d, int b, T x, bool inverse, SplineTape<T>& tp) {
    using D7 = DualN<T, 7>;
    D7 in[7] = {D7(x), D7(cw[b]), D7(cw[b + 1]), D7(ch[b]), D7(ch[b + 1]), D7(d[b]), D7(d[b + 1])};
#pragma unroll
    for (int c = 0; c < 7; ++c) in[c].d[c] = T(1);
    const SplineOut<D7> r = spline_core_vals<D7>(in[1], in[2], in[3], in[4], in[5], in[6], b, in[0], inverse);
#pragma unroll
    for (int c = 0; c < 7; ++c) { tp.dy[c] = r.y.d[c]; tp.dl[c] = r.lad.d[c]; }
    tp.b = b;
    return r.y.v;
}
// the C2-smooth circular spline (two bins): everything behind the knot table is a function of (x, cw_1, ch_1) -- three tangents through
// spline_circular_smooth_vals (jf_spline.h); in the tape's layout cw_1 / ch_1 are the upper knots of bin 0
template <typename T> __device__ inline T spline_adj_smooth_circular3(T cw1, T ch1, T x, bool inverse, SplineTape<T>& tp) {
    using D3 = DualN<T, 3>;
    D3 xd(x), cwd(cw1), chd(ch1);
    xd.d[0] = T(1); cwd.d[1] = T(1); chd.d[2] = T(1);
    int raw;
    const SplineOut<D3> r = spline_circular_smooth_vals<D3>(cwd, chd, xd, inverse, raw);
#pragma unroll
    for (int c = 0; c < 7; ++c) { tp.dy[c] = T(0); tp.dl[c] = T(0); }
    tp.dy[0] = r.y.d[0]; tp.dl[0] = r.lad.d[0];
    tp.dy[2] = r.y.d[1]; tp.dl[2] = r.lad.d[1];
    tp.dy[4] = r.y.d[2]; tp.dl[4] = r.lad.d[2];
    tp.b = 0;
    return r.y.v;
}
// the six knot adjoints of the bin from the tape and the upstream gradients of (y, lad); returns d S / d x of the spline
template <typename T> __device__ __forceinline__ T spline_tape_contract(const SplineTape<T>& tp, T gy, T glad, T (&gk)[6]) {
#pragma unroll
    for (int c = 0; c < 6; ++c) gk[c] = gy * tp.dy[c + 1] + glad * tp.dl[c + 1];
    return gy * tp.dy[0] + glad * tp.dl[0];
}
// contraction + table reverse; returns d S / d x of the spline, g_scale receives d S / d scale (circular)
template <typename T> __device__ inline T spline_tape_reverse(const T* __restrict__ p, T* __restrict__ gp, const SplineDev<T>& o, const T* __restrict__ tab, const SplineTape<T>& tp,
                                                             T gy, T glad, T lo, T hi, bool circular, T scale, T& g_scale) {
    T gk[6];
    const T gx = spline_tape_contract<T>(tp, gy, glad, gk);
    g_scale = spline_adj_table_reverse<T>(p, gp, o, tab, tp.b, gk, lo, hi, circular, scale);
    return gx;
}

// ---- 'r' core, non-smooth (rational_quadratic_spline.py:282-296 -> spline_fns.py:45-186)
// (tab: the lane's scratch, or -- built = true -- the workgroup's table of this layer)
template <typename T> __device__ inline void r_core_adj_fwd(const jf_r_layer& L, const T* __restrict__ p, T xin, T* __restrict__ tab, bool built, AdjLane<T>& A,
                                                           SplineTape<T>& tp) {
    tp.cl_in = xin > T(1) || xin < T(-1);
    const T x = xin > T(1) ? T(1) : (xin < T(-1) ? T(-1) : xin);
    const SplineDev<T> o = to_dev<T>(L.sp);
    if (!built) spline_adj_build<T>(p, o, tab, (T)L.lo, (T)L.hi, false, T(1));
    const KnotTab<T> t(tab, o.nb);
    A.oob = A.oob || (x < (T)L.lo) || (x > (T)L.hi);
    const T y = spline_adj_core7<T>(t.cw, t.ch, t.d, spline_adj_bin<T>(t.cw, t.ch, o.nb, x, true), x, true, tp);
    tp.cl_out = y > T(1) || y < T(-1);
    tp.y = y > T(1) ? T(1) : (y < T(-1) ? T(-1) : y);
}
// g: d S / d (core output) in, d S / d (core input) out; the log-det's gradient gld enters through lad
template <typename T> __device__ inline void r_core_adj_bwd(const jf_r_layer& L, const T* __restrict__ p, T* __restrict__ gp, AdjLane<T>& A, const SplineTape<T>& tp, T& g, T gld) {
    const SplineDev<T> o = to_dev<T>(L.sp);
    T gs;
    const T gx = spline_tape_reverse<T>(p, gp, o, A.scr, tp, tp.cl_out ? T(0) : g, gld, (T)L.lo, (T)L.hi, false, T(1), gs);
    g = tp.cl_in ? T(0) : gx;
}
// the smooth variants and anything else: the core on dual numbers over its own parameters
template <typename T> __device__ inline void r_core_adj_dual(const jf_r_layer& L, const T* __restrict__ p, T* __restrict__ gp, T xin, AdjLane<T>& A, T& g, T gld) {
    const T xi[3] = {xin, T(0), T(0)};
    const T gy[3] = {g, T(0), T(0)};
    T gx[3] = {T(0), T(0), T(0)};
    stage_dual_adjoint<T>([&](const DualN<T, ADJ_N>* row, DualN<T, ADJ_N> (&x)[3], DualN<T, ADJ_N>& ld, LaneCtx<DualN<T, ADJ_N>>& c) {
        x[0] = r_core<DualN<T, ADJ_N>>(L, row, x[0], ld, c, true);
    }, p, spline_row_len(L.sp), gp, 1, xi, gy, gld, gx, A);
    g = gx[0];
}
__host__ __device__ inline bool spline_hand_adjoint(const jf_spline_opts& s) { return s.smooth == 0; }
__host__ __device__ inline bool circ_hand_adjoint(const jf_spline_opts& s) { return s.smooth == 0 || s.num_bins == 2; }      // (the C2-smooth circular spline has two bins)

// ---- 'o' core (splines_1d.py:111-194 -> spline_fns.py:45-186), log-prob direction; scale: fvm_2d's azimuthal scaling (1 for a layer of its own)
template <typename T> __device__ inline void o_core_adj_fwd(const jf_o_layer& L, const T* __restrict__ p, T xin, T scale, AdjLane<T>& A, SplineTape<T>& tp) {
    const T lo = T(1e-7), hi = T(2.0 * PI_D - 1e-7);
    tp.cl_in = xin > hi || xin < lo;
    const T x = safe_angle_2pi<T>(xin);
    const bool use_inverse = L.natural_direction != 0;
    const SplineDev<T> o = to_dev<T>(L.sp);
    spline_adj_build<T>(p, o, A.scr, T(0), M<T>::TWO_PI, true, scale);
    const KnotTab<T> t(A.scr, o.nb);
    A.oob = A.oob || (x < T(0)) || (x > M<T>::TWO_PI);
    const T y = o.smooth ? spline_adj_smooth_circular3<T>(t.cw[1], t.ch[1], x, use_inverse, tp)
                         : spline_adj_core7<T>(t.cw, t.ch, t.d, spline_adj_bin<T>(t.cw, t.ch, o.nb, x, use_inverse), x, use_inverse, tp);
    tp.cl_out = y > hi || y < lo;
    tp.y = safe_angle_2pi<T>(y);
}
template <typename T> __device__ inline void o_core_adj_bwd(const jf_o_layer& L, const T* __restrict__ p, T* __restrict__ gp, T scale, AdjLane<T>& A, const SplineTape<T>& tp,
                                                           T& g, T gld, T& g_scale) {
    const SplineDev<T> o = to_dev<T>(L.sp);
    T gs;
    const T gx = spline_tape_reverse<T>(p, gp, o, A.scr, tp, tp.cl_out ? T(0) : g, gld, T(0), M<T>::TWO_PI, true, scale, gs);
    g = tp.cl_in ? T(0) : gx;
    g_scale += gs;
}
template <typename T> __device__ inline void o_core_adj_dual(const jf_o_layer& L, const T* __restrict__ p, T* __restrict__ gp, T xin, T scale, AdjLane<T>& A, T& g, T gld,
                                                            T& g_scale) {
    const T xi[3] = {xin, scale, T(0)};
    const T gy[3] = {g, T(0), T(0)};
    T gx[3] = {T(0), T(0), T(0)};
    stage_dual_adjoint<T>([&](const DualN<T, ADJ_N>* row, DualN<T, ADJ_N> (&x)[3], DualN<T, ADJ_N>& ld, LaneCtx<DualN<T, ADJ_N>>& c) {
        x[0] = o_core<DualN<T, ADJ_N>>(L, row, x[0], ld, c, false, x[1]);
        x[1] = DualN<T, ADJ_N>(T(0));
    }, p, spline_row_len(L.sp), gp, 2, xi, gy, gld, gx, A);
    g = gx[0];
    g_scale += gx[1];
}

// ================================================================================================= layers
// Every family: adjoint(L, p, gp, xin, g, gld, gblp, A) -- p / gp the layer's parameter row and its gradient (added to), xin the layer's input,
// g: d S / d (layer output) in, d S / d (layer input) out.  gblp (non-zero for the LAST layer applied only): the upstream gradient of the base
// log-prob sum_d -1/2 out_d^2, i.e. g_d -= out_d gblp once the layer's own forward part knows its output -- the chain's forward sweep
// (mchain_rev_kernel) then never evaluates that layer: a one-layer chain is evaluated once, inside its adjoint.

struct RAdj {
    static constexpr bool HAS_SHARED = true;                              // adjoint_shared below
    static __host__ int dual_row(const jf_r_layer& L) { return spline_hand_adjoint(L.sp) ? 0 : spline_row_len(L.sp); }
    static __host__ int dual_tab(const jf_r_layer& L) { return spline_hand_adjoint(L.sp) ? 0 : spline_tab_words(L.sp.num_bins); }
    static __host__ int scr_words(const jf_r_layer& L) { return spline_tab_words(L.sp.num_bins); }
    static __host__ int corr_words(const jf_r_layer&) { return 0; }
    // forward part up to the layer's output and d S / d (core output); tab / built as in r_core_adj_fwd
    template <typename T> static __device__ __forceinline__ T head(const jf_r_layer& L, const T* __restrict__ p, T xin, T* __restrict__ tab, bool built, const T (&g)[3], T gld,
                                                                  T gblp, AdjLane<T>& A, SplineTape<T>& tp, bool chart = true) {
        r_core_adj_fwd<T>(L, p, xin, tab, built, A, tp);
        if (L.first && chart) {                                                    // interval_base.py:61-69 behind the core
            Dual<T> ldd(T(0));
            const Dual<T> o = interval_to_real_line<Dual<T>>(Dual<T>(tp.y, T(1)), Dual<T>((T)L.lo), Dual<T>((T)L.hi), ldd);
            return (g[0] - o.v * gblp) * o.d + gld * ldd.d;
        }
        return g[0] - tp.y * gblp;
    }
    // chart = false: the spline core alone (the vertical flows nested in 'f': fvm_2d.py:430-432 calls the core, not the layer)
    template <typename T> static __device__ inline void adjoint(const jf_r_layer& L, const T* __restrict__ p, T* __restrict__ gp, const T (&xin)[3], T (&g)[3], T gld, T gblp,
                                                                AdjLane<T>& A, bool chart = true) {
        if (!spline_hand_adjoint(L.sp)) {                                 // few parameters (<= 8): the whole layer on dual numbers
            if (!chart) { T gc = g[0]; r_core_adj_dual<T>(L, p, gp, xin[0], A, gc, gld); g[0] = gc; return; }
            T gy[3] = {g[0], T(0), T(0)};
            if (gblp != T(0)) {
                LaneCtx<T> c;
                c.tab = A.scr; c.corr = nullptr; c.bins = nullptr; c.bin_i = 0; c.oob = c.nonconv = c.nonfinite = false; c.lane_valid = A.lane_valid;
                T xo[3] = {xin[0], T(0), T(0)}, ldv = T(0);
                RFam::apply<T, false>(L, p, xo, ldv, c);
                gy[0] -= xo[0] * gblp;
            }
            stage_dual_adjoint<T>([&](const DualN<T, ADJ_N>* row, DualN<T, ADJ_N> (&x)[3], DualN<T, ADJ_N>& ld, LaneCtx<DualN<T, ADJ_N>>& c) {
                RFam::apply<DualN<T, ADJ_N>, false>(L, row, x, ld, c);
            }, p, spline_row_len(L.sp), gp, 1, xin, gy, gld, g, A);
            return;
        }
        SplineTape<T> tp;
        T gc = head<T>(L, p, xin[0], A.scr, false, g, gld, gblp, A, tp, chart);
        r_core_adj_bwd<T>(L, p, gp, A, tp, gc, gld);
        g[0] = gc;
    }
    // permanent parameters: the workgroup's table of the layer (`tab`, built once) and its knot-adjoint accumulators acc = [gcw | gch | gd]
    // (nb + 1 each, LDS atomics); the table is reversed once per workgroup (spline_adj_table_reverse_dense)
    template <typename T> static __device__ inline void adjoint_shared(const jf_r_layer& L, const T* __restrict__ p, T* __restrict__ tab, T* __restrict__ acc, const T (&xin)[3],
                                                                       T (&g)[3], T gld, T gblp, bool active, AdjLane<T>& A) {
        SplineTape<T> tp;
        const T gc = head<T>(L, p, xin[0], tab, true, g, gld, gblp, A, tp);
        T gk[6];
        const T gx = spline_tape_contract<T>(tp, tp.cl_out ? T(0) : gc, gld, gk);
        g[0] = tp.cl_in ? T(0) : gx;
        if (active) {
            const int nb = L.sp.num_bins, b = tp.b;
            if (b != 0) { atomicAdd(acc + b, gk[0]); atomicAdd(acc + (nb + 1) + b, gk[2]); }                 // (knots 0 and nb are pinned)
            if (b + 1 != nb) { atomicAdd(acc + b + 1, gk[1]); atomicAdd(acc + (nb + 1) + b + 1, gk[3]); }
            atomicAdd(acc + 2 * (nb + 1) + b, gk[4]);
            atomicAdd(acc + 2 * (nb + 1) + b + 1, gk[5]);
        }
    }
};

struct OAdj {
    static __host__ int dual_row(const jf_o_layer& L) {
        const int r = rot_len(L.hh_iter, 2), s = circ_hand_adjoint(L.sp) ? 0 : spline_row_len(L.sp);
        return r > s ? r : s;
    }
    static __host__ int dual_tab(const jf_o_layer& L) { return circ_hand_adjoint(L.sp) ? 0 : spline_tab_words(L.sp.num_bins); }
    static __host__ int scr_words(const jf_o_layer& L) { return spline_tab_words(L.sp.num_bins); }
    static __host__ int corr_words(const jf_o_layer&) { return 0; }
    template <typename T> static __device__ inline void adjoint(const jf_o_layer& L, const T* __restrict__ p, T* __restrict__ gp, const T (&xin)[3], T (&g)[3], T gld, T gblp,
                                                                AdjLane<T>& A) {
        const int n_rot = rot_len(L.hh_iter, 2);
        const T* sp = p + n_rot;
        T* gsp = gp + n_rot;
        const T x1 = L.hh_iter != 0 ? s1_rotate<T>(p, L.hh_iter, xin[0], true) : xin[0];
        T gc = g[0], gs = T(0);
        if (circ_hand_adjoint(L.sp)) {
            SplineTape<T> tp;
            o_core_adj_fwd<T>(L, sp, x1, T(1), A, tp);
            if (L.first) {                                                // sphere_base.py:460-480 behind the core
                Dual<T> ldd(T(0));
                const Dual<T> o = s1_to_plane<Dual<T>>(Dual<T>(tp.y, T(1)), ldd);
                gc = (g[0] - o.v * gblp) * o.d + gld * ldd.d;
            } else {
                gc = g[0] - tp.y * gblp;
            }
            o_core_adj_bwd<T>(L, sp, gsp, T(1), A, tp, gc, gld, gs);
        } else {
            if (L.first || gblp != T(0)) {
                T ldv = T(0);
                LaneCtx<T> c;
                c.tab = A.scr; c.corr = nullptr; c.bins = nullptr; c.bin_i = 0; c.oob = c.nonconv = c.nonfinite = false; c.lane_valid = A.lane_valid;
                const T y = o_core<T>(L, sp, x1, ldv, c, false, T(1));
                if (L.first) {
                    Dual<T> ldd(T(0));
                    const Dual<T> o = s1_to_plane<Dual<T>>(Dual<T>(y, T(1)), ldd);
                    gc = (g[0] - o.v * gblp) * o.d + gld * ldd.d;
                } else {
                    gc = g[0] - y * gblp;
                }
            }
            o_core_adj_dual<T>(L, sp, gsp, x1, T(1), A, gc, gld, gs);
        }
        if (L.hh_iter != 0) {                                             // the rotation in front: x and its <= hh * 2 parameters on dual numbers
            const T gy[3] = {gc, T(0), T(0)};
            T gx[3] = {T(0), T(0), T(0)};
            const int hh = L.hh_iter;
            stage_dual_adjoint<T>([&](const DualN<T, ADJ_N>* row, DualN<T, ADJ_N> (&x)[3], DualN<T, ADJ_N>&, LaneCtx<DualN<T, ADJ_N>>&) {
                x[0] = s1_rotate<DualN<T, ADJ_N>>(row, hh, x[0], true);
            }, p, n_rot, gp, 1, xin, gy, T(0), gx, A);
            gc = gx[0];
        }
        g[0] = gc;
    }
};

// ---- 'm': the Moebius mixture (moebius_1d.py:140-259).  val = V / W - pi, deriv = D / W with W = sum w_k, V = sum w_k arc_k, D = sum w_k ratio_k.
// S = g_val * val + g_logd * log deriv.  A component's shape parameters reach S through its own (arc_k, ratio_k) only, its log-weight through
// the softmax: one 4-tangent evaluation per component (x and its <= 3 shape parameters).  Returns d S / d x; gp == nullptr: only that.
template <typename T> __device__ inline T moebius_adjoint(const T* __restrict__ p, T* __restrict__ gp, int nc, int np, T x, T g_val, T g_logd) {
    using D4 = DualN<T, 4>;
    T lmax = p[np - 1];
    for (int k = 1; k < nc; ++k) lmax = M<T>::max(lmax, p[np * k + np - 1]);
    T W = T(0), V = T(0), Dm = T(0);
    {
        const T cx = M<T>::cos(x), sx = M<T>::sin(x);
        for (int k = 0; k < nc; ++k) {
            const int q = np * k;
            T arc, ratio;
            moebius_component<T>(np, p[q], np == 4 ? p[q + 1] : T(0), p[q + np - 2], cx, sx, arc, ratio);
            const T w = M<T>::exp(p[q + np - 1] - lmax);
            W += w; V += w * arc; Dm += w * ratio;
        }
    }
    D4 xd(x);
    xd.d[0] = T(1);
    const D4 cx = M<D4>::cos(xd), sx = M<D4>::sin(xd);
    T gx = T(0);
    for (int k = 0; k < nc; ++k) {
        const int q = np * k;
        D4 q0(p[q]), q1(np == 4 ? p[q + 1] : T(0)), ll(p[q + np - 2]);
        q0.d[1] = T(1);
        if (np == 4) { q1.d[2] = T(1); ll.d[3] = T(1); } else { ll.d[2] = T(1); }
        D4 arc, ratio;
        moebius_component<D4>(np, q0, q1, ll, cx, sx, arc, ratio);
        const T w = M<T>::exp(p[q + np - 1] - lmax);
        const T c_arc = g_val * w / W, c_ratio = g_logd * w / Dm;
        gx += c_arc * arc.d[0] + c_ratio * ratio.d[0];
        if (gp == nullptr) continue;
        gp[q] += c_arc * arc.d[1] + c_ratio * ratio.d[1];
        if (np == 4) {
            gp[q + 1] += c_arc * arc.d[2] + c_ratio * ratio.d[2];
            gp[q + 2] += c_arc * arc.d[3] + c_ratio * ratio.d[3];
        } else {
            gp[q + 1] += c_arc * arc.d[2] + c_ratio * ratio.d[2];
        }
        gp[q + np - 1] += g_val * (w / W) * (arc.v - V / W) + g_logd * (w * ratio.v / Dm - w / W);
    }
    return gx;
}

struct MAdj {
    static __host__ int dual_row(const jf_m_layer& L) { return rot_len(L.hh_iter, 2); }
    static __host__ int dual_tab(const jf_m_layer&) { return 0; }
    static __host__ int scr_words(const jf_m_layer&) { return 0; }
    static __host__ int corr_words(const jf_m_layer&) { return 0; }
    template <typename T> static __device__ inline void adjoint(const jf_m_layer& L, const T* __restrict__ p, T* __restrict__ gp, const T (&xin)[3], T (&g)[3], T gld, T gblp,
                                                                AdjLane<T>& A) {
        const int n_rot = rot_len(L.hh_iter, 2), nc = L.num_components, np = MFam::omega_pars(L);
        const T* mp = p + n_rot;
        T* gmp = gp + n_rot;
        const T x1 = L.hh_iter != 0 ? s1_rotate<T>(p, L.hh_iter, xin[0], true) : xin[0];
        const T xi = x1 > M<T>::PI ? x1 - M<T>::TWO_PI : x1;              // moebius_1d.py:73-74 (slope 1 on both branches)
        const bool direct = L.natural_direction == 0;
        T xs = xi, val, d;
        if (!direct) xs = moebius_solve_values<T>(mp, nc, np, xi, A.lane_valid, A.nonconv, A.nonfinite);
        moebius_eval<T>(mp, nc, np, xs, val, d);
        const T out = direct ? (val < T(0) ? M<T>::TWO_PI + val : val) : (xs < T(0) ? M<T>::TWO_PI + xs : xs);
        T gc = g[0] - out * gblp;
        if (L.first) {
            Dual<T> ldd(T(0));
            const Dual<T> o = s1_to_plane<Dual<T>>(Dual<T>(out, T(1)), ldd);
            gc = (g[0] - o.v * gblp) * o.d + gld * ldd.d;
        }
        if (direct) {                                                     // out = val(x, p), ld += log deriv(x, p)
            gc = moebius_adjoint<T>(mp, gmp, nc, np, xs, gc, gld);
        } else {
            // out = x with val(x, p) = z, ld -= log deriv(x, p).  d x / d z = 1 / deriv, d x / d p = -(d val / d p) / deriv, so with
            // lambda = (gc - gld d log deriv / d x) / deriv:  d S / d z = lambda,  d S / d p = -gld d log deriv / d p - lambda d val / d p
            const T ld_x = moebius_adjoint<T>(mp, nullptr, nc, np, xs, T(0), T(1));      // d log deriv / d x (parameter gradients dropped)
            const T lambda = (gc - gld * ld_x) / d;
            (void)moebius_adjoint<T>(mp, gmp, nc, np, xs, -lambda, -gld);
            gc = lambda;
        }
        if (L.hh_iter != 0) {
            const T gy[3] = {gc, T(0), T(0)};
            T gx[3] = {T(0), T(0), T(0)};
            const int hh = L.hh_iter;
            stage_dual_adjoint<T>([&](const DualN<T, ADJ_N>* row, DualN<T, ADJ_N> (&x)[3], DualN<T, ADJ_N>&, LaneCtx<DualN<T, ADJ_N>>&) {
                x[0] = s1_rotate<DualN<T, ADJ_N>>(row, hh, x[0], true);
            }, p, n_rot, gp, 1, xin, gy, T(0), gx, A);
            gc = gx[0];
        }
        g[0] = gc;
    }
};

// ---- 'f' (fvm_2d.py:273-483): head (rotation, von-Mises-Fisher z step, optional quarter turn) and tail (angles, first-layer chart) on dual
// numbers over their own <= 13 parameters; the nested spline flows through the spline adjoints above; the per-sample MLP of the correlated
// variant (amortizable_mlp.py:508-578) reversed by hand.
struct FAdj {
    static __host__ int dual_row(const jf_f_layer& L) {
        int r = rot_len(L.hh_iter, 3) + FFam::n_kappa(L);
        for (int i = 0; i < L.n_vertical; ++i) { const int v = RAdj::dual_row(L.vertical[i]); r = v > r ? v : r; }
        for (int i = 0; i < L.n_circular; ++i) { const int v = OAdj::dual_row(L.circular[i]); r = v > r ? v : r; }
        return r;
    }
    static __host__ int dual_tab(const jf_f_layer& L) {
        int r = 0;
        for (int i = 0; i < L.n_vertical; ++i) { const int v = RAdj::dual_tab(L.vertical[i]); r = v > r ? v : r; }
        for (int i = 0; i < L.n_circular; ++i) { const int v = OAdj::dual_tab(L.circular[i]); r = v > r ? v : r; }
        return r;
    }
    static __host__ int scr_words(const jf_f_layer& L) {
        int r = 0;
        for (int i = 0; i < L.n_vertical; ++i) { const int v = RAdj::scr_words(L.vertical[i]); r = v > r ? v : r; }
        for (int i = 0; i < L.n_circular; ++i) { const int v = OAdj::scr_words(L.circular[i]); r = v > r ? v : r; }
        return r;
    }
    static __host__ int corr_words(const jf_f_layer& L) { return L.correlated ? 2 * JF_CORR_SCRATCH : 0; }

    // gradient of the per-sample MLP's weights and of its input z from the gradient of its outputs (FFam::corr_mlp)
    template <typename T> static __device__ inline T corr_mlp_reverse(const jf_f_layer& L, const T* __restrict__ mp, T* __restrict__ gmp, T z, const T* __restrict__ tacc,
                                                                     const T* __restrict__ gout, T* __restrict__ gt) {
        const int H = L.corr_hidden, R = L.corr_rank, n = FFam::corr_out(L);
        const T* W1 = mp;
        const T* b1 = mp + H;
        const T* s2 = mp + 2 * H;
        T* gW1 = gmp;
        T* gb1 = gmp + H;
        T* gs2 = gmp + 2 * H;
        T gz = T(0);
        if (L.corr_full2) {
            T* gb2 = gs2 + n * H;
            for (int i = 0; i < n; ++i) gb2[i] += gout[i];
            for (int j = 0; j < H; ++j) {
                const T h = M<T>::tanh(W1[j] * z + b1[j]);
                T gh = T(0);
                for (int i = 0; i < n; ++i) { gs2[i * H + j] += gout[i] * h; gh += gout[i] * s2[i * H + j]; }
                const T gpre = gh * (T(1) - h * h);
                gW1[j] += gpre * z; gb1[j] += gpre; gz += gpre * W1[j];
            }
        } else {
            const T* V = s2 + n * R;
            T* gV = gs2 + n * R;
            T* gb2 = gV + R * H;
            for (int r = 0; r < R; ++r) gt[r] = T(0);
            for (int i = 0; i < n; ++i) {
                gb2[i] += gout[i];
                for (int r = 0; r < R; ++r) { gs2[i * R + r] += gout[i] * tacc[r]; gt[r] += gout[i] * s2[i * R + r]; }
            }
            for (int j = 0; j < H; ++j) {
                const T h = M<T>::tanh(W1[j] * z + b1[j]);
                T gh = T(0);
                for (int r = 0; r < R; ++r) { gV[r * H + j] += gt[r] * h; gh += gt[r] * V[r * H + j]; }
                const T gpre = gh * (T(1) - h * h);
                gW1[j] += gpre * z; gb1[j] += gpre; gz += gpre * W1[j];
            }
        }
        return gz;
    }

    template <typename T> static __device__ inline void adjoint(const jf_f_layer& L, const T* __restrict__ p, T* __restrict__ gp, const T (&xin)[3], T (&g)[3], T gld, T gblp,
                                                                AdjLane<T>& A) {
        using Du = DualN<T, ADJ_N>;
        const int n_head = rot_len(L.hh_iter, 3) + FFam::n_kappa(L);
        int nv = 0;
        for (int i = 0; i < L.n_vertical; ++i) nv += spline_row_len(L.vertical[i].sp);
        const T* vert = p + n_head;
        T* gvert = gp + n_head;
        const T* circ = vert + nv;
        T* gcirc = gvert + nv;
        const T region = (T)L.identity_region;
        LaneCtx<T> c;
        c.tab = A.scr; c.corr = A.corr; c.bins = nullptr; c.bin_i = 0; c.oob = c.nonconv = c.nonfinite = false; c.lane_valid = A.lane_valid;
        // ---- forward values: the inputs of every nested spline
        T retB, angB;
        {
            T x[3] = {xin[0], xin[1], T(0)};
            T ld = T(0);
            FFam::inv_head<T>(L, p, x, ld, retB, angB);
        }
        const bool inside = (region == T(0)) || ((retB > T(-1) + region) && (retB < T(1) - region));
        T ang_in[JF_MAX_NESTED], ret_in[JF_MAX_NESTED];
        T ret = retB, angle = angB, sc = T(1);
        if (inside) {
            T ld = T(0);
            if (L.correlated) {
                int off = nv;
                for (int i = L.n_vertical - 1; i >= 0; --i) {
                    off -= spline_row_len(L.vertical[i].sp);
                    ret_in[i] = ret;
                    ret = r_core<T>(L.vertical[i], vert + off, ret, ld, c, true);
                }
                FFam::corr_mlp<T>(L, vert + nv, retB, A.corr);
                int coff = FFam::corr_out(L);
                for (int i = L.n_circular - 1; i >= 0; --i) {
                    coff -= rot_len(L.circular[i].hh_iter, 2) + spline_row_len(L.circular[i].sp);
                    ang_in[i] = angle;
                    T xx[3] = {angle, T(0), T(0)};
                    OFam::apply<T, false>(L.circular[i], A.corr + coff, xx, ld, c);
                    angle = xx[0];
                }
            } else {
                if (L.n_circular > 0) {
                    sc = azimuthal_scaling<T>(retB);
                    int off = 0;
                    for (int i = 0; i < L.n_circular; ++i) off += spline_row_len(L.circular[i].sp);
                    for (int i = L.n_circular - 1; i >= 0; --i) {
                        off -= spline_row_len(L.circular[i].sp);
                        ang_in[i] = angle;
                        angle = o_core<T>(L.circular[i], circ + off, angle, ld, c, false, sc);
                    }
                }
                int off = nv;
                for (int i = L.n_vertical - 1; i >= 0; --i) {
                    off -= spline_row_len(L.vertical[i].sp);
                    ret_in[i] = ret;
                    ret = r_core<T>(L.vertical[i], vert + off, ret, ld, c, true);
                }
            }
        }
        A.oob = A.oob || c.oob;
        // ---- tail: d S / d (ret, angle) behind the nested flows
        T g_ret, g_ang;
        {
            const T xi[3] = {ret, angle, T(0)};
            T gy[3] = {g[0], g[1], T(0)};
            if (gblp != T(0)) {                                           // the chain's output
                T xo[3] = {T(0), T(0), T(0)}, ldv = T(0);
                FFam::inv_tail<T>(L, ret, angle, xo, ldv);
                gy[0] -= xo[0] * gblp; gy[1] -= xo[1] * gblp;
            }
            T gx[3] = {T(0), T(0), T(0)};
            stage_dual_adjoint<T>([&](const Du*, Du (&x)[3], Du& ld, LaneCtx<Du>&) {
                const Du r = x[0], a = x[1];
                FFam::inv_tail<Du>(L, r, a, x, ld);
            }, p, 0, gp, 2, xi, gy, gld, gx, A);
            g_ret = gx[0]; g_ang = gx[1];
        }
        // ---- nested flows in reverse of their order of application
        if (inside) {
            if (L.correlated) {
                int off = 0;
                for (int i = 0; i < L.n_vertical; ++i) {                  // (applied last-to-first: reversed first-to-last)
                    T gi[3] = {g_ret, T(0), T(0)};
                    const T xi[3] = {ret_in[i], T(0), T(0)};
                    RAdj::adjoint<T>(L.vertical[i], vert + off, gvert + off, xi, gi, gld, T(0), A, false);
                    g_ret = gi[0];
                    off += spline_row_len(L.vertical[i].sp);
                }
                const int n_out = FFam::corr_out(L);
                T* gout = A.corr + JF_CORR_SCRATCH;
                for (int i = 0; i < n_out; ++i) gout[i] = T(0);
                int coff = 0;
                for (int i = 0; i < L.n_circular; ++i) {
                    T gi[3] = {g_ang, T(0), T(0)};
                    const T xi[3] = {ang_in[i], T(0), T(0)};
                    OAdj::adjoint<T>(L.circular[i], A.corr + coff, gout + coff, xi, gi, gld, T(0), A);
                    g_ang = gi[0];
                    coff += rot_len(L.circular[i].hh_iter, 2) + spline_row_len(L.circular[i].sp);
                }
                g_ret += corr_mlp_reverse<T>(L, vert + nv, gvert + nv, retB, A.corr + n_out, gout, gout + n_out);
            } else {
                int off = 0;
                for (int i = 0; i < L.n_vertical; ++i) {
                    T gi[3] = {g_ret, T(0), T(0)};
                    const T xi[3] = {ret_in[i], T(0), T(0)};
                    RAdj::adjoint<T>(L.vertical[i], vert + off, gvert + off, xi, gi, gld, T(0), A, false);
                    g_ret = gi[0];
                    off += spline_row_len(L.vertical[i].sp);
                }
                if (L.n_circular > 0) {
                    T g_sc = T(0);
                    int coff = 0;
                    for (int i = 0; i < L.n_circular; ++i) {
                        const jf_o_layer& Lc = L.circular[i];
                        if (circ_hand_adjoint(Lc.sp)) {
                            SplineTape<T> tp;
                            o_core_adj_fwd<T>(Lc, circ + coff, ang_in[i], sc, A, tp);
                            o_core_adj_bwd<T>(Lc, circ + coff, gcirc + coff, sc, A, tp, g_ang, gld, g_sc);
                        } else {
                            o_core_adj_dual<T>(Lc, circ + coff, gcirc + coff, ang_in[i], sc, A, g_ang, gld, g_sc);
                        }
                        coff += spline_row_len(Lc.sp);
                    }
                    const Dual<T> scd = azimuthal_scaling<Dual<T>>(Dual<T>(retB, T(1)));
                    g_ret += g_sc * scd.d;
                }
            }
        }
        // ---- head: (x_in, rotation, kappa) -> (ret, angle)
        {
            const T gy[3] = {g_ret, g_ang, T(0)};
            T gx[3] = {T(0), T(0), T(0)};
            stage_dual_adjoint<T>([&](const Du* row, Du (&x)[3], Du& ld, LaneCtx<Du>&) {
                Du r, a;
                FFam::inv_head<Du>(L, row, x, ld, r, a);
                x[0] = r; x[1] = a;
            }, p, n_head, gp, 2, xin, gy, gld, gx, A);
            g[0] = gx[0]; g[1] = gx[1];
        }
    }
};

}  // namespace jf
