// Reverse-mode backward of 'g' chains with general options -- rotations other than Householder reflections ("angles", "cayley",
// "triangular_combination"), center_mean, add_skewness, the rq_splines stretch (gaussianization_flow.py:699-909) -- round 6.
//
// What torch.autograd returns for the layer loop of gf_block.inv_flow_mapping (gaussianization_flow.py:992-1114) given upstream gradients of
// (x_out, log_det_out, base_logp_out).  Until round 5 these chains were differentiated by replaying the WHOLE chain on single-tangent dual
// numbers once per input direction (gfx_chain_bwd_kernel, gf_bwd_kernels.hip; D + P replays: 18 ... 140 ms per 2^16 rows against 0.25 ... 0.55
// forward).  Here: one lane per row, a plain forward sweep that keeps every layer's input, then the layers in reverse, each layer stage by stage:
//   * the mixture of one coordinate: its three log-sum-exps (log cdf, log sf, log pdf) are sums over the components, a component's
//     parameters (mean, raw log-width, raw log-exponent) reach them through that component's term alone and its log-weight through the
//     softmax -- ONE evaluation per component on four tangents (x and its three parameters; gx_component, jf_gf_ext.h); the inverse-CDF
//     stage behind the sums on three tangents (gf_icdf on (lc, ls, lp)); center_mean's dependent last mean in closed form;
//   * the rq_splines stretch: the spline adjoint of jf_spline_adj.h extended to the learnable box (unpinned end knots) and the linear tails;
//   * the rotation and the offset: dual-number passes over the rotation's own directions only (D coordinates + its parameters, four per pass).
// Per-sample parameters: every gradient is stored once; permanent parameters: wave sums into the workgroup's row in LDS, written as the
// workgroup's partial row (summed by the caller, as for gf_chain_bwd_kernel).
// JF_G_BWD_DUAL=1 selects the dual-number replay: the check of this file (scripts/probe/m_adjoint_check.py covers the 'g' option fixtures too).
#include <cstdlib>

#include "jf_dual.h"
#include "jf_gf_bwd.h"
#include "jf_gf_ext.h"
#include "jf_spline_adj.h"

namespace jf {

constexpr int GXR_N = 4;

// the layer descriptor with its T-valued fields as constants of another scalar type
template <typename T, typename S> __device__ inline GfLayerDev<S> gx_layer_as(const GfLayerDev<T>& o) {
    GfLayerDev<S> r;
    r.K = o.K; r.hh = o.hh; r.model_offset = o.model_offset; r.fit_norm = o.fit_norm; r.reg_norm = o.reg_norm; r.inv_type = o.inv_type;
    r.width_mode = o.width_mode; r.clamp_widths = o.clamp_widths; r.fast = o.fast; r.stretch = o.stretch; r.off_box = o.off_box;
    r.n_params = o.n_params; r.col0 = o.col0; r.off_rot = o.off_rot; r.off_mean = o.off_mean; r.off_lw = o.off_lw; r.off_ln = o.off_ln;
    r.vec_ok = o.vec_ok; r.rot_mode = o.rot_mode; r.center_mean = o.center_mean; r.skew = o.skew; r.off_skew = o.off_skew;
    r.wmin = S(o.wmin); r.wmax = S(o.wmax); r.inv_wmax = S(o.inv_wmax); r.nmin = S(o.nmin); r.nmax = S(o.nmax); r.lw_lo = S(o.lw_lo); r.lw_hi = S(o.lw_hi);
    return r;
}

// parameter row as N-tangent dual numbers: tangent c is 1 at row index seed0 + c
template <typename T, int N> struct SeededRowN {
    const T* p; int seed0;
    __device__ __forceinline__ DualN<T, N> operator[](int i) const {
        DualN<T, N> r(p[i]);
#pragma unroll
        for (int c = 0; c < N; ++c) if (i == seed0 + c) r.d[c] = T(1);
        return r;
    }
    __device__ __forceinline__ SeededRowN operator+(int k) const { return SeededRowN{p + k, seed0 - k}; }
};

template <typename T> __device__ __forceinline__ T gxr_wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// where a lane's parameter gradients go: its own row of g_params (per-sample parameters), or -- wave sums -- the workgroup's row in LDS
template <typename T> struct GradSink {
    T* row;            // per-sample: g_params + row * gps (nullptr for inactive lanes)
    T* acc;            // permanent parameters: the workgroup's accumulators in LDS (else nullptr)
    bool active, bad;
    __device__ __forceinline__ void emit(int col, T v) {       // called by ALL lanes of the wave with the same col (wave-uniform control flow)
        if (!active) v = T(0);
        bad = bad || !M<T>::finite(v);
        if (acc) {
            const T s = gxr_wave_sum<T>(v);
            if ((threadIdx.x & 63) == 0) atomicAdd(acc + col, s);
        } else if (active) {
            row[col] = v;
        }
    }
};

// ---- one coordinate of a classic-stretch layer: x -> y = icdf(mixture(x)), ld += logd.  gy / gld: upstream; returns d S / d x.
template <typename T>
__device__ inline T gxr_coordinate(const GfLayerDev<T>& o, const T* __restrict__ p, int D, int d, T x, T gy, T gld, T gblp, GradSink<T>& sink) {
    using D4 = DualN<T, 4>;
    using D3 = DualN<T, 3>;
    const int K = o.K, n_pos = K / 2;
    // plain pass: the weights' log-sum-exp, the dependent mean, the three sums
    const GxCoord<T> c = gx_prepare<T, const T*>(o, p, D, d);
    const MixQ<T> q = gx_mixture<T, const T*>(o, p, D, d, c, x);
    // the inverse-CDF stage on (lc, ls, lp)
    T g_lc, g_ls, g_lp;
    {
        D3 lc(q.lc), ls(q.ls), lp(q.lp);
        lc.d[0] = T(1); ls.d[1] = T(1); lp.d[2] = T(1);
        const GfLayerDev<D3> o3 = gx_layer_as<T, D3>(o);
        const IcdfOut<D3> s = gf_icdf<D3>(o.inv_type, gx_mixq<D3>(o3, lc, ls, lp));
        const T gyo = gy - s.y.v * gblp;                           // (last layer applied: the base log-prob term)
        g_lc = gyo * s.y.d[0] + gld * s.logd.d[0];
        g_ls = gyo * s.y.d[1] + gld * s.logd.d[1];
        g_lp = gyo * s.y.d[2] + gld * s.logd.d[2];
    }
    const T g_sum = g_lc + g_ls + g_lp;                            // sum_k t_k (the softmax weights of each sum add up to one)
    const GfLayerDev<D4> o4 = gx_layer_as<T, D4>(o);
    T gx = T(0), g_mu_last = T(0);
    T w_last = T(1);
    if (o.center_mean) w_last = M<T>::exp(gx_log_weight<T, const T*>(o, p, D, K - 1, d));
    // components: the last one first (center_mean: its mean's gradient is spread over the others)
    for (int kk = 0; kk < K; ++kk) {
        const int k = kk == 0 ? K - 1 : kk - 1;
        const bool dep = o.center_mean && k == K - 1;
        const T mu_v = dep ? c.last_mean : p[o.off_mean + k * D + d];
        const T raw_lw = p[o.off_lw + k * D + d];
        const T raw_sk = o.skew ? p[o.off_skew + k * D + d] : T(0);
        const T lwt = gx_log_weight<T, const T*>(o, p, D, k, d);
        const T ln_pi = lwt - c.lse_w;
        D4 xd(x), mud(mu_v), lwd(raw_lw), skd(raw_sk);
        xd.d[0] = T(1); mud.d[1] = T(1); lwd.d[2] = T(1); skd.d[3] = T(1);
        D4 tc, ts, tp;
        gx_component<D4>(o4, xd, mud, lwd, skd, k < n_pos, D4(ln_pi), tc, ts, tp);
        // d S / d (term of component k) = upstream of the sum x the term's softmax weight
        const T ac = g_lc * M<T>::exp(tc.v - q.lc), as = g_ls * M<T>::exp(ts.v - q.ls), ap = g_lp * M<T>::exp(tp.v - q.lp);
        gx += ac * tc.d[0] + as * ts.d[0] + ap * tp.d[0];
        T g_mu = ac * tc.d[1] + as * ts.d[1] + ap * tp.d[1];
        const T g_lw = ac * tc.d[2] + as * ts.d[2] + ap * tp.d[2];
        const T g_sk = ac * tc.d[3] + as * ts.d[3] + ap * tp.d[3];
        // the regulated log-weight: through ln_pi = lwt - lse_w (t_k - pi_k sum t) and, with center_mean, through the dependent mean
        T g_lwt = (ac + as + ap) - M<T>::exp(ln_pi) * g_sum;
        if (dep) {
            g_mu_last = g_mu;                                      // mu_last = -sum_{k<K-1} m_k w_k / w_last
            g_lwt += g_mu_last * (-c.last_mean);                   // d mu_last / d lwt_last = -mu_last
        } else {
            if (o.center_mean) {
                const T wk = M<T>::exp(lwt);
                g_mu += g_mu_last * (-wk / w_last);
                g_lwt += g_mu_last * (-mu_v * wk / w_last);
            }
            sink.emit(o.col0 + o.off_mean + k * D + d, g_mu);
        }
        sink.emit(o.col0 + o.off_lw + k * D + d, g_lw);
        if (o.fit_norm) {
            T dreg = T(1);
            if (o.reg_norm) {
                const Dual<T> raw(p[o.off_ln + k * D + d], T(1));
                const Dual<T> reg = M<Dual<T>>::log(Dual<T>(o.nmin) + Dual<T>(o.nmax) / (Dual<T>(T(1)) + M<Dual<T>>::exp(-raw)));
                dreg = reg.d;
            }
            sink.emit(o.col0 + o.off_ln + k * D + d, g_lwt * dreg);
        }
        if (o.skew) sink.emit(o.col0 + o.off_skew + k * D + d, g_sk);
    }
    return gx;
}

// ---- one coordinate of an rq_splines layer (gaussianization_flow.py:863-909 -> spline_fns.py:188-358): spline with a learnable box, linear tails.
// row sections (d-major): widths un_w[K], heights un_h[K], derivatives un_d[K + 1], box = (left, ln(width - 0.5), bottom, ln(height - 0.5))
template <typename T>
__device__ inline T gxr_spline_coordinate(const GfLayerDev<T>& o, const T* __restrict__ p, int d, T x, T gy, T gld, T gblp, T* __restrict__ tab, GradSink<T>& sink) {
    const int nb = o.K;
    const T* un_w = p + o.off_mean + d * nb;
    const T* un_h = p + o.off_lw + d * nb;
    const T* un_d = p + o.off_ln + d * (nb + 1);
    const T* box = p + o.off_box + d * 4;
    const int c_w = o.col0 + o.off_mean + d * nb, c_h = o.col0 + o.off_lw + d * nb, c_d = o.col0 + o.off_ln + d * (nb + 1), c_box = o.col0 + o.off_box + d * 4;
    const SplineOut<T> r0 = spline_linext<T>(un_w, un_h, un_d, box, nb, tab, x, false);          // builds the table; value of y for the base log-prob term
    const KnotTab<T> t(tab, nb);
    const T left = box[0], wspan = M<T>::exp(box[1]) + T(0.5), bottom = box[2], hspan = M<T>::exp(box[3]) + T(0.5);
    const T right = left + wspan;
    const T gyo = gy - r0.y * gblp;
    T gx, g_left = T(0), g_wspan = T(0), g_bottom = T(0), g_hspan = T(0);
    T gk[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};                // adjoints of (cw_b, cw_{b+1}, ch_b, ch_{b+1}, d_b, d_{b+1})
    int b = r0.bin < 0 ? 0 : (r0.bin > nb - 1 ? nb - 1 : r0.bin);
    if (x <= left) {                                               // y = x d0 + (bottom - left d0), lad = log d0
        const T d0 = t.d[0];
        b = 0;
        gx = gyo * d0;
        gk[4] = gyo * (x - left) + gld / d0;
        g_left -= gyo * d0; g_bottom += gyo;
    } else if (x >= right) {                                       // y = x dl + (ch_nb - cw_nb dl), lad = log dl; cw_nb = left + wspan, ch_nb = bottom + hspan (sum of the bins' shares = 1)
        const T dl = t.d[nb];
        b = nb - 1;
        gx = gyo * dl;
        gk[5] = gyo * (x - t.cw[nb]) + gld / dl;
        gk[1] = -gyo * dl; gk[3] = gyo;
    } else {
        using D7 = DualN<T, 7>;
        D7 in[7] = {D7(x), D7(t.cw[b]), D7(t.cw[b + 1]), D7(t.ch[b]), D7(t.ch[b + 1]), D7(t.d[b]), D7(t.d[b + 1])};
#pragma unroll
        for (int c = 0; c < 7; ++c) in[c].d[c] = T(1);
        const SplineOut<D7> r = spline_core_vals<D7>(in[1], in[2], in[3], in[4], in[5], in[6], b, in[0], false);
        gx = gyo * r.y.d[0] + gld * r.lad.d[0];
#pragma unroll
        for (int c = 0; c < 6; ++c) gk[c] = gyo * r.y.d[c + 1] + gld * r.lad.d[c + 1];
    }
    // knots: knot_j = lo + span cum_j, cum_0 = 0, cum_nb = 1 (nothing pinned: the box is a parameter)
    const T mix = T(1) - T(1e-3) * T(nb), inv_mix = T(1) / mix;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const T* knots = which ? t.ch : t.cw;
        const T lo = which ? bottom : left, span = which ? hspan : wspan, inv_span = T(1) / span;
        const T g0 = gk[2 * which], g1 = gk[2 * which + 1];
        const T cum_b = (knots[b] - lo) * inv_span, cum_b1 = (knots[b + 1] - lo) * inv_span;
        if (which) { g_bottom += g0 + g1; g_hspan += g0 * cum_b + g1 * cum_b1; }
        else { g_left += g0 + g1; g_wspan += g0 * cum_b + g1 * cum_b1; }
        const T soft_b = ((knots[b + 1] - knots[b]) * inv_span - T(1e-3)) * inv_mix;
        const T S_b = (cum_b - T(1e-3) * T(b)) * inv_mix;
        const T dot = span * ((g0 + g1) * S_b + g1 * soft_b);
        T kj = knots[0];
        for (int j = 0; j < nb; ++j) {
            const T kn = knots[j + 1];
            const T soft = ((kn - kj) * inv_span - T(1e-3)) * inv_mix;
            kj = kn;
            const T A = span * (j < b ? g0 + g1 : (j == b ? g1 : T(0)));
            sink.emit((which ? c_h : c_w) + j, mix * soft * (A - dot));
        }
    }
    for (int j = 0; j <= nb; ++j) {
        const T g = j == b ? gk[4] : (j == b + 1 ? gk[5] : T(0));
        sink.emit(c_d + j, g == T(0) ? T(0) : g * adj_sigmoid<T>(un_d[j]));
    }
    sink.emit(c_box + 0, g_left);
    sink.emit(c_box + 1, g_wspan * M<T>::exp(box[1]));
    sink.emit(c_box + 2, g_bottom);
    sink.emit(c_box + 3, g_hspan * M<T>::exp(box[3]));
    return gx;
}

template <typename T>
__global__ void __launch_bounds__(GX_THREADS) gfx_chain_rev_kernel(const GfBwdArgs<T> a, const int64_t pstep, const int64_t tiles_total) {
    using Du = DualN<T, GXR_N>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, D = a.D;
    // LDS (offset 16: a pointer to LDS offset 0 must not reach a function that is not inlined, manifold_rev_kernels.hip)
    T* xin0 = reinterpret_cast<T*>(smem_raw + 16);                                  // [n_layers][D][GX_THREADS] every layer's input
    T* xw0 = xin0 + (size_t)a.n_layers * D * GX_THREADS;                            // [D][GX_THREADS] working column (values)
    T* gc0 = xw0 + (size_t)D * GX_THREADS;                                          // [D][GX_THREADS] gradient column
    T* gn0 = gc0 + (size_t)D * GX_THREADS;                                          // [D][GX_THREADS] gradient column (next)
    T* tab0 = gn0 + (size_t)D * GX_THREADS;                                         // [GX_THREADS][spline_tab]
    T* accp = tab0 + (size_t)GX_THREADS * a.spline_tab;                            // [P] (permanent parameters)
    const bool bcast = pstep == 0;
    size_t off_d = (size_t)((accp + (bcast ? a.n_params_total : 0)) - reinterpret_cast<T*>(smem_raw)) * sizeof(T);
    off_d = (off_d + 15) & ~(size_t)15;
    Du* xd0 = reinterpret_cast<Du*>(smem_raw + off_d);                              // [D][GX_THREADS] the rotation's dual column
    const XCol<T> xw{xw0 + tid}, gc{gc0 + tid}, gn{gn0 + tid};
    const XCol<Du> xd{xd0 + tid};
    T* tab = tab0 + tid * a.spline_tab;
    if (bcast) {
        for (int j = tid; j < a.n_params_total; j += GX_THREADS) accp[j] = T(0);
        __syncthreads();
    }
    bool bad_any = false;
    bool first_tile = true;
    for (int64_t tile = blockIdx.x; tile < tiles_total || (bcast && first_tile); tile += gridDim.x) {
        first_tile = false;
        const int64_t row = tile * GX_THREADS + tid;
        const bool active = row < a.B && tile < tiles_total;
        const int64_t rrow = active ? row : a.B - 1;
        const T gld = (a.g_ld && active) ? a.g_ld[rrow] : T(0);
        const T gblp = (a.g_blp && active) ? a.g_blp[rrow] : T(0);
        const T* prow = a.params + rrow * pstep;
        GradSink<T> sink;
        sink.row = (!bcast && active) ? a.g_params + row * a.gps : nullptr;
        sink.acc = bcast ? accp : nullptr;
        sink.active = active; sink.bad = false;
        // ---- forward sweep: every layer's input (layer 0, the last one applied, is evaluated by its own reverse step)
        for (int d = 0; d < D; ++d) xw[d] = a.x[rrow * a.xs + d];
        for (int l = a.n_layers - 1; l >= 0; --l) {
            const GfLayerDev<T>& o = a.L[l];
            const XCol<T> xi{xin0 + (size_t)l * D * GX_THREADS + tid};
            for (int d = 0; d < D; ++d) xi[d] = xw[d];
            if (l == 0) break;
            const T* p = prow + o.col0;
            if (o.model_offset) for (int d = 0; d < D; ++d) xw[d] = xw[d] - p[d];
            gx_rotate<T, const T*>(o, p, xw, D, true);
            for (int d = 0; d < D; ++d) {
                if (o.stretch == JF_GF_STRETCH_RQ_SPLINES) {
                    xw[d] = spline_linext<T>(p + (o.off_mean + d * o.K), p + (o.off_lw + d * o.K), p + (o.off_ln + d * (o.K + 1)), p + (o.off_box + d * 4), o.K, tab, xw[d], false).y;
                } else {
                    const GxCoord<T> c = gx_prepare<T, const T*>(o, p, D, d);
                    xw[d] = gf_icdf<T>(o.inv_type, gx_mixture<T, const T*>(o, p, D, d, c, xw[d])).y;
                }
            }
        }
        for (int d = 0; d < D; ++d) gc[d] = (a.g_xout && active) ? a.g_xout[rrow * a.gxos + d] : T(0);
        // ---- reverse sweep
        for (int l = 0; l < a.n_layers; ++l) {
            const GfLayerDev<T>& o = a.L[l];
            const T* p = prow + o.col0;
            const XCol<T> xi{xin0 + (size_t)l * D * GX_THREADS + tid};
            // this layer's stretch input: offset, rotation (values)
            for (int d = 0; d < D; ++d) xw[d] = o.model_offset ? xi[d] - p[d] : xi[d];
            gx_rotate<T, const T*>(o, p, xw, D, true);
            // the stretch, coordinate by coordinate: gc (d S / d y) -> gn (d S / d rotated input)
            for (int d = 0; d < D; ++d) {
                if (o.stretch == JF_GF_STRETCH_RQ_SPLINES) gn[d] = gxr_spline_coordinate<T>(o, p, d, xw[d], gc[d], gld, l == 0 ? gblp : T(0), tab, sink);
                else gn[d] = gxr_coordinate<T>(o, p, D, d, xw[d], gc[d], gld, l == 0 ? gblp : T(0), sink);
            }
            // the rotation: directions = its D inputs and its parameters, GXR_N per pass; gn -> gc
            const int n_rot = o.off_mean - o.off_rot;
            const bool rotates = o.rot_mode == JF_GF_ROT_HOUSEHOLDER ? o.hh > 0 : (D >= 2);
            if (!rotates) {
                for (int d = 0; d < D; ++d) gc[d] = gn[d];
                for (int i = 0; i < n_rot; ++i) sink.emit(o.col0 + o.off_rot + i, T(0));
            } else {
                const GfLayerDev<Du> od = gx_layer_as<T, Du>(o);
                const int n_dir = D + n_rot;
                for (int j0 = 0; j0 < n_dir; j0 += GXR_N) {
                    for (int d = 0; d < D; ++d) {
                        Du v(o.model_offset ? xi[d] - p[d] : xi[d]);
#pragma unroll
                        for (int c = 0; c < GXR_N; ++c) if (d == j0 + c) v.d[c] = T(1);
                        xd[d] = v;
                    }
                    gx_rotate<Du, SeededRowN<T, GXR_N>>(od, SeededRowN<T, GXR_N>{p, o.off_rot + j0 - D}, xd, D, true);
                    T gj[GXR_N];
#pragma unroll
                    for (int c = 0; c < GXR_N; ++c) gj[c] = T(0);
                    for (int d = 0; d < D; ++d) {
                        const Du v = xd[d];
                        const T g = gn[d];
#pragma unroll
                        for (int c = 0; c < GXR_N; ++c) gj[c] += g * v.d[c];
                    }
#pragma unroll
                    for (int c = 0; c < GXR_N; ++c) {
                        const int j = j0 + c;
                        if (j >= n_dir) break;
                        if (j < D) gc[j] = gj[c];
                        else sink.emit(o.col0 + o.off_rot + (j - D), gj[c]);
                    }
                }
            }
            if (o.model_offset) for (int d = 0; d < D; ++d) sink.emit(o.col0 + d, -gc[d]);
        }
        bool bad = sink.bad;
        for (int d = 0; d < D; ++d) {
            const T v = gc[d];
            bad = bad || (active && !M<T>::finite(v));
            if (active) a.g_x[row * a.gxs + d] = v;
        }
        bad_any = bad_any || bad;
    }
    if (bcast) {                                                   // this workgroup's partial row
        __syncthreads();
        for (int j = tid; j < a.n_params_total; j += GX_THREADS) a.g_params[(int64_t)blockIdx.x * a.gps + j] = accp[j];
    }
    status_add(a.status, JF_STATUS_NONFINITE, bad_any);
}

template <typename T> int64_t gfx_chain_rev_lds_bytes(const GfBwdArgs<T>& a, bool bcast) {
    const size_t plain = 16 + ((size_t)(a.n_layers + 3) * a.D * GX_THREADS + (size_t)GX_THREADS * a.spline_tab + (bcast ? (size_t)a.n_params_total : 0)) * sizeof(T);
    return (int64_t)(((plain + 15) & ~(size_t)15) + (size_t)a.D * GX_THREADS * sizeof(DualN<T, GXR_N>));
}

template <typename T> int gfx_chain_rev_launch(GfBwdArgs<T> a, bool bcast, int64_t ps, int64_t blocks, int64_t tiles, void* stream) {
    const size_t lds = (size_t)gfx_chain_rev_lds_bytes<T>(a, bcast);
    if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;
    auto k = gfx_chain_rev_kernel<T>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    jf::launch(k, dim3((unsigned)blocks), dim3(GX_THREADS), lds, (hipStream_t)stream, a, bcast ? (int64_t)0 : ps, tiles);
    return check_launch();
}

template int gfx_chain_rev_launch<float>(GfBwdArgs<float>, bool, int64_t, int64_t, int64_t, void*);
template int gfx_chain_rev_launch<double>(GfBwdArgs<double>, bool, int64_t, int64_t, int64_t, void*);
template int64_t gfx_chain_rev_lds_bytes<float>(const GfBwdArgs<float>&, bool);
template int64_t gfx_chain_rev_lds_bytes<double>(const GfBwdArgs<double>&, bool);

}  // namespace jf
