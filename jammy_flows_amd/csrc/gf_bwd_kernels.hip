// Backward (vector-Jacobian product) of a chain of 'g' layers in the log-prob direction: what torch.autograd produces for
// gf_block._inv_flow_mapping + the euclidean_base offset (gaussianization_flow.py:995-1114, euclidean_base.py:34-51) when the loss is a
// function of (x_out, log_det_out, base_logp_out) -- the training step of the reference (examples/jammy_flows.py:381-412,
// docs/source/usage/training.rst:24-44) replays ~560 eager ops per layer for it; here it is ONE launch per e-block.
//
//   jf_gf_chain_inv_bwd_*   inputs : x, params (as in jf_gf_chain_inv), upstream gradients g_x_out (B, D), g_log_det (B), g_base_logp (B)
//                           outputs: g_x (B, D); g_params: per-sample regime (B, P) -- one row per sample, the layout of `params`, what the
//                                    amortisation MLP's backward consumes --, broadcast regime (n_partials, P) partial sums, one row per
//                                    workgroup, that the caller adds up (deterministic: no floating-point atomics)
//   (g_log_det_in = g_log_det and g_base_logp_in = g_base_logp: both are accumulated, the caller passes them through.)
//
// Nothing is saved by the forward launch: the kernel first re-runs the chain keeping each layer's input coordinate (one register per layer
// and lane), then walks the layers backwards, re-evaluating each layer's mixture at its input.  Work distribution as in the forward
// kernels (jf_gf.h): lane = (row, coordinate), reductions over a row's coordinates are DPP butterflies.
//
// Derivatives are taken in LOG SPACE so that one code path is valid at any distance from the mixture components (the forward pass needs a
// scaled re-evaluation there): with u_k = (x - mu_k)/w_k, s_k = sigma(u_k) and the responsibilities
//     rC_k = pi_k s_k / cdf,   rS_k = pi_k (1 - s_k) / sf,   rP_k = pi_k s_k (1 - s_k) / (w_k pdf)         (each = exp(log term - log total))
// one has  d log cdf / du_k = rC_k (1 - s_k),  d log sf / du_k = -rS_k s_k,  d log pdf / du_k = rP_k (1 - 2 s_k),
//          d log {cdf, sf, pdf} / d log pi_k = {rC_k, rS_k, rP_k},  d log pdf / d log(1/w_k) += rP_k,
// and the inverse-CDF stage contributes (dy, d logd) = (A_y, A_H) d log cdf + (B_y, B_H) d log sf + (0, 1) d log pdf  (gf_icdf_coeffs below).
// Only sums of the form  sum_k (g . d log(.)/d theta)  are formed, never a quotient of two underflowing sums.
#include "jf_gf.h"
#include "jf_dual.h"
#include "jf_gf_ext.h"

namespace jf {

constexpr int GB_MAX_HH = 8;

template <typename T> struct GfBwdArgs {
    const T* x; int64_t xs;
    const T* params; int64_t ps;
    int64_t B;
    int D, n_layers, tile_stride, tiles_per_block;
    GfLayerDev<T> L[JF_MAX_CHAIN];
    int n_params_total;
    const T* g_xout; int64_t gxos;
    const T* g_ld;
    const T* g_blp;
    T* g_x; int64_t gxs;
    T* g_params; int64_t gps;
    int32_t* status;
};

// coefficients of the inverse-CDF stage in log space: dy = Ay dlc + By dls,  d(logd - lp) = AH dlc + BH dls
// (lc = log cdf, ls = log sf; any pair that reproduces the total derivative along cdf + sf = 1 is valid -- the better conditioned one is used)
template <typename T> struct IcdfCoef { T Ay, By, AH, BH; };

template <typename T> __device__ __forceinline__ IcdfCoef<T> pade_coeffs(const MixQ<T>& q, T y, bool centre_window) {
    const T a = T(PADE_A);
    const T c = T(2.0 / (3.14159265358979323846 * PADE_A));
    const T dlt = q.sf - q.cdf;                              // ln(4 cdf sf) without cancellation near the centre, as in the forward (pade_terms)
    const T L = (M<T>::min(q.cdf, q.sf) > T(0.01)) ? M<T>::log1p(-dlt * dlt) : q.lc + q.ls + T(1.38629436111989061883);
    const T F = L * T(0.5) + c;
    const T rad = -L / a;
    const T F2 = M<T>::sqrt(F * F + rad);
    const T G = F > T(0) ? rad / (F2 + F) : F2 - F;
    const T f2L = (T(0.5) * F - T(0.5) / a) / F2;            // dF2/dL
    const T gL = f2L - T(0.5);                               // dG/dL
    IcdfCoef<T> k;
    k.Ay = k.By = gL / y;                                    // y = +-sqrt(2 G)
    const T hL = gL / (G + T(1) / a) - T(0.5) * gL / G - f2L / F2 - T(1);
    const T dsc = q.sf - q.cdf;
    k.AH = hL - q.cdf / dsc;                                 // + d log|sf - cdf|
    k.BH = hL + q.sf / dsc;
    if (centre_window) { k.AH = T(0); k.BH = T(0); }         // the reference pins the log-derivative there (gaussianization_flow.py:623-625)
    return k;
}

template <typename T> __device__ __forceinline__ IcdfCoef<T> gf_icdf_coeffs(int inv_type, const MixQ<T>& q, T y) {
    IcdfCoef<T> k;
    if (inv_type == JF_GF_ISIGMOID) { k.Ay = T(1); k.By = T(-1); k.AH = T(-1); k.BH = T(-1); return k; }
    const T bound = T(PADE_BOUND);
    if (inv_type == JF_GF_INORMAL_FULL_PADE) return pade_coeffs(q, y, (q.cdf > T(0.49999)) && (q.cdf < T(0.50001)));
    const bool left = q.cdf <= bound, right = q.sf <= bound;
    if (!left && !right) {                                   // exact inverse normal CDF: dy/dcdf = sqrt(2 pi) e^{y^2/2}
        const T e = M<T>::HALF_LN_2PI + T(0.5) * y * y;
        if (q.cdf <= q.sf) { k.Ay = M<T>::exp(q.lc + e); k.By = T(0); }
        else { k.Ay = T(0); k.By = -M<T>::exp(q.ls + e); }
        k.AH = y * k.Ay; k.BH = y * k.By;
        return k;
    }
    if (inv_type == JF_GF_INORMAL_PARTLY_CRUDE) {
        const T lsum = q.lc + q.ls;
        const T r = M<T>::sqrt(T(-2) * lsum);
        k.Ay = k.By = (right ? T(-1) : T(1)) / r;
        k.AH = k.BH = T(-0.5) / lsum - T(1);
        return k;
    }
    return pade_coeffs(q, y, false);
}

// d log(1/w) / d(raw log-width) and 1/w for one component (gaussianization_flow.py:269-317)
template <typename T> __device__ __forceinline__ void gf_inv_width_grad(const GfLayerDev<T>& o, T rw, T& iw, T& dliw) {
    bool inside = true;
    if (o.clamp_widths) { inside = rw >= o.lw_lo && rw <= o.lw_hi; rw = clampv(rw, o.lw_lo, o.lw_hi); }
    if (o.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION) {
        const T e = M<T>::exp(-rw);
        const T ae = o.inv_wmax + e;
        const T den = o.wmin * ae + T(1);
        iw = ae / den;
        dliw = -e / (ae * den);
    } else if (o.width_mode == JF_GF_WIDTH_EXP) {
        const T e = M<T>::exp(rw);
        iw = T(1) / (e + o.wmin);
        dliw = -e * iw;
    } else {
        iw = T(1) / (softplus(rw) + o.wmin);
        dliw = -iw / (T(1) + M<T>::exp(-rw));
    }
    if (!inside) dliw = T(0);                                // torch.clamp passes no gradient outside its bounds
}

// backward of one layer for the lane's coordinate.  p / gp: parameter row and gradient row (+ d), both in LDS; gy: upstream gradient of this
// layer's output coordinate, gl: upstream gradient of log_det (the same for every layer); returns the gradient of the layer's input coordinate.
// ACC: add into the gradient row (broadcast regime: rows of the tile share it) instead of overwriting.
template <typename T, int G, bool ACC>
__device__ __forceinline__ T gf_layer_bwd(const T* __restrict__ p, T* __restrict__ gp, const GfLayerDev<T>& o, int D, bool live, T x_in, T gy, T gl) {
    // ACC (broadcast parameters): the rows of a wave that share this lane's coordinate all add to the SAME accumulator -- one LDS atomic per
    // lane would serialise 64 / G ways on that address.  The wave sums over its rows first (butterfly over the lanes with equal lane % G), then
    // one lane per coordinate adds.  Every lane takes part in the shuffles (put is only called under wave-uniform conditions).
    auto put = [&](int off, T v) {
        if constexpr (ACC) {
            T s = live ? v : T(0);
#pragma unroll
            for (int sh = G; sh < 64; sh <<= 1) s += __shfl_xor(s, sh, 64);
            if (live && (int)(threadIdx.x & 63) < G) atomicAdd(gp + off, s);
        } else {
            if (live) gp[off] = v;
        }
    };
    // ---- recompute: offset, reflections (keeping the vector before each one), mixture
    T xr[GB_MAX_HH];
    T x = x_in;
    if (o.model_offset) x -= p[0];
#pragma unroll
    for (int i = 0; i < GB_MAX_HH; ++i) {
        xr[i] = x;
        if (i < o.hh) x = gfg_reflect<T, G, true>(p, o.off_rot + i * D, live, x);
    }
    const MixQ<T> q = gfg_mixture<T, true>(p, o, D, x);
    const IcdfOut<T> s = gf_icdf<T>(o.inv_type, q);
    const IcdfCoef<T> c = gf_icdf_coeffs<T>(o.inv_type, q, s.y);
    const T g_lc = gy * c.Ay + gl * c.AH, g_ls = gy * c.By + gl * c.BH, g_lp = gl;
    const T Gsum = g_lc + g_ls + g_lp;

    // normaliser of the weights
    const bool fit = o.fit_norm != 0;
    T shift = T(0), Nn = T(0);
    if (fit && !o.reg_norm) {
        shift = p[o.off_ln];
        for (int k = 1; k < o.K; ++k) shift = M<T>::max(shift, p[o.off_ln + k * D]);
    }
    if (fit) for (int k = 0; k < o.K; ++k) Nn += gf_weight(o, p[o.off_ln + k * D], shift);
    const T lN = fit ? M<T>::log(Nn) : M<T>::log(T(o.K));

    T gx = T(0);
    for (int k = 0; k < o.K; ++k) {
        const T mu = p[o.off_mean + k * D];
        T iw, dliw;
        gf_inv_width_grad<T>(o, p[o.off_lw + k * D], iw, dliw);
        T lpi = -lN, dlnn = T(0), pik = T(0);
        if (fit) {
            const T rn = p[o.off_ln + k * D];
            const T nk = gf_weight(o, rn, shift);
            lpi = M<T>::log(nk) - lN;
            pik = nk / Nn;
            if (o.reg_norm) { const T sg = T(1) / (T(1) + M<T>::exp(-rn)); dlnn = o.nmax * sg * (T(1) - sg) / nk; }
            else dlnn = T(1);
        }
        const T u = (x - mu) * iw;
        const T t = M<T>::exp(-M<T>::abs(u));
        const T hi = T(1) / (T(1) + t), lo = t * hi;
        const bool pos = u >= T(0);
        const T sg = pos ? hi : lo, sgc = pos ? lo : hi;     // sigma(u), sigma(-u)
        const T l1p = M<T>::log1p(t);
        const T lsp = (pos ? T(0) : u) - l1p;                // log sigma(u)
        const T lsm = (pos ? -u : T(0)) - l1p;               // log sigma(-u)
        const T rC = M<T>::exp(lpi + lsp - q.lc);
        const T rS = M<T>::exp(lpi + lsm - q.ls);
        const T rP = M<T>::exp(lpi + lsp + lsm + M<T>::log(iw) - q.lp);
        const T gu = g_lc * rC * sgc - g_ls * rS * sg + g_lp * rP * (sgc - sg);
        gx += gu * iw;
        put(o.off_mean + k * D, -gu * iw);
        put(o.off_lw + k * D, (gu * u + g_lp * rP) * dliw);
        if (fit) put(o.off_ln + k * D, ((g_lc * rC + g_ls * rS + g_lp * rP) - pik * Gsum) * dlnn);
    }
    // ---- reflections, last first:  y = x - c v, c = 2 (v.x)/(v.v):  g_x = H g,  g_v = -c g - (2 (v.g)/n) x + (4 (v.x)(v.g)/n^2) v
    T g = gx;
#pragma unroll
    for (int i = GB_MAX_HH - 1; i >= 0; --i) {
        if (i < o.hh) {
            const T v = live ? p[o.off_rot + i * D] : T(0);
            const T n = group_sum<T, G>(v * v), sx = group_sum<T, G>(v * xr[i]), vg = group_sum<T, G>(v * (live ? g : T(0)));
            const T rn = T(1) / n;
            put(o.off_rot + i * D, -T(2) * sx * rn * g - T(2) * vg * rn * xr[i] + T(4) * sx * vg * rn * rn * v);
            g -= T(2) * vg * rn * v;
        }
    }
    if (o.model_offset) put(0, -g);
    return g;
}

template <typename T, int G, bool BCAST>
__global__ void __launch_bounds__(BCAST ? 256 : 64) gf_chain_bwd_kernel(const GfBwdArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    constexpr int NT = BCAST ? 256 : 64;
    constexpr int R = NT / G;
    const int tid = threadIdx.x;
    constexpr int LG = G == 1 ? 0 : G == 2 ? 1 : G == 4 ? 2 : 3;
    const int g = tid & (G - 1), r = tid >> LG;
    const int D = a.D;
    const bool live = g < D, leader = g == 0;
    const int d = live ? g : D - 1;
    // LDS: BCAST: [raw rows of all layers: n_layers x tile_stride][gradient accumulators: n_layers x tile_stride]
    //      per-sample: [parameter tile R x tile_stride][gradient tile R x tile_stride]
    T* ptile = lds;
    T* gtile = lds + (BCAST ? a.n_layers : R) * a.tile_stride;
    if constexpr (BCAST) {
        for (int l = 0; l < a.n_layers; ++l) {
            const GfLayerDev<T> o = a.L[l];
            for (int j = tid; j < a.tile_stride; j += NT) {
                ptile[l * a.tile_stride + j] = j < o.n_params ? a.params[o.col0 + j] : T(0);
                gtile[l * a.tile_stride + j] = T(0);
            }
        }
        __syncthreads();
    }
    const int tiles = BCAST ? a.tiles_per_block : 1;
    for (int t = 0; t < tiles; ++t) {
        const int64_t row0 = ((int64_t)blockIdx.x * tiles + t) * R;
        if (row0 >= a.B) break;
        const int64_t row = row0 + r;
        const bool row_valid = row < a.B;
        const int64_t rrow = row_valid ? row : a.B - 1;
        const int valid_rows = (int)((a.B - row0) < R ? (a.B - row0) : R);

        // ---- forward sweep (layers n-1 .. 0), keeping every layer's input
        T xin[JF_MAX_CHAIN];
        T x = a.x[rrow * a.xs + d];
#pragma unroll
        for (int li = 0; li < JF_MAX_CHAIN; ++li) {
            xin[li] = x;
            if (li < a.n_layers) {
                const int l = a.n_layers - 1 - li;
                const GfLayerDev<T> o = a.L[l];
                const T* p;
                if constexpr (BCAST) p = ptile + l * a.tile_stride + d;
                else {
                    __syncthreads();
                    stage_rows<T>(ptile, a.tile_stride, a.params + row0 * a.ps + o.col0, a.ps, o.n_params, R, valid_rows, tid, NT, o.vec_ok != 0);
                    __syncthreads();
                    p = ptile + r * a.tile_stride + d;
                }
                if (o.model_offset) x -= p[0];
                x = gfg_rotate_inv<T, G, true>(p, o, D, live, x);
                x = gf_icdf<T>(o.inv_type, gfg_mixture<T, true>(p, o, D, x)).y;
            }
        }
        // ---- upstream gradients: base log-prob = sum_d -x^2/2 - ...  =>  d/dx_out = -x_out
        const T gl = (a.g_ld && row_valid) ? a.g_ld[rrow] : T(0);
        T gy = (a.g_xout && row_valid) ? a.g_xout[rrow * a.gxos + d] : T(0);
        if (a.g_blp && row_valid) gy -= x * a.g_blp[rrow];
        if (!live || !row_valid) gy = T(0);
        const T glr = row_valid ? gl : T(0);

        // ---- backward sweep (layers 0 .. n-1)
#pragma unroll
        for (int li = JF_MAX_CHAIN - 1; li >= 0; --li) {
            if (li < a.n_layers) {
                const int l = a.n_layers - 1 - li;
                const GfLayerDev<T> o = a.L[l];
                const T xi = xin[li];
                if constexpr (BCAST) {
                    gy = gf_layer_bwd<T, G, true>(ptile + l * a.tile_stride + d, gtile + l * a.tile_stride + d, o, D, live && row_valid, xi, gy, glr);
                } else {
                    __syncthreads();
                    stage_rows<T>(ptile, a.tile_stride, a.params + row0 * a.ps + o.col0, a.ps, o.n_params, R, valid_rows, tid, NT, o.vec_ok != 0);
                    __syncthreads();
                    gy = gf_layer_bwd<T, G, false>(ptile + r * a.tile_stride + d, gtile + r * a.tile_stride + d, o, D, live, xi, gy, glr);
                    __syncthreads();
                    // gradient tile -> HBM, row by row (consecutive lanes = consecutive columns)
                    for (int rr2 = 0; rr2 < valid_rows; ++rr2)
                        for (int j = tid; j < o.n_params; j += NT)
                            a.g_params[(row0 + rr2) * a.gps + o.col0 + j] = gtile[rr2 * a.tile_stride + j];
                }
            }
        }
        if (row_valid && live) a.g_x[row * a.gxs + d] = gy;
        const T bad = group_max<T, G>((live && !M<T>::finite(gy)) ? T(1) : T(0));
        status_add(a.status, JF_STATUS_NONFINITE, row_valid && leader && bad > T(0));
    }
    if constexpr (BCAST) {
        __syncthreads();
        for (int l = 0; l < a.n_layers; ++l) {
            const GfLayerDev<T> o = a.L[l];
            for (int j = tid; j < o.n_params; j += NT) a.g_params[(int64_t)blockIdx.x * a.gps + o.col0 + j] = gtile[l * a.tile_stride + j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- general-option layers
// Chains with a layer that uses add_skewness / center_mean / a non-Householder rotation (jf_gf_ext.h) are differentiated in FORWARD mode, like
// the manifold layers (manifold_bwd_kernels.hip): the very device code of the forward kernel, instantiated on dual numbers, once per input
// direction (D target coordinates + every parameter of the row), contracted with the upstream gradients.  O(P) replays of the chain per row --
// these options are off the benchmarked path; what matters is that training with them works and is exact.
template <typename T> struct SeededRow {               // parameter row as dual numbers: tangent 1 at index `seed`
    const T* p; int seed;
    __device__ __forceinline__ Dual<T> operator[](int i) const { return Dual<T>(p[i], i == seed ? T(1) : T(0)); }
    __device__ __forceinline__ SeededRow operator+(int k) const { return SeededRow{p + k, seed - k}; }
};

template <typename T> __device__ inline GfLayerDev<Dual<T>> gx_dual_layer(const GfLayerDev<T>& o) {
    GfLayerDev<Dual<T>> r;
    r.K = o.K; r.hh = o.hh; r.model_offset = o.model_offset; r.fit_norm = o.fit_norm; r.reg_norm = o.reg_norm; r.inv_type = o.inv_type;
    r.width_mode = o.width_mode; r.clamp_widths = o.clamp_widths; r.fast = o.fast; r.stretch = o.stretch; r.off_box = o.off_box;
    r.n_params = o.n_params; r.col0 = o.col0; r.off_rot = o.off_rot; r.off_mean = o.off_mean; r.off_lw = o.off_lw; r.off_ln = o.off_ln;
    r.vec_ok = o.vec_ok; r.rot_mode = o.rot_mode; r.center_mean = o.center_mean; r.skew = o.skew; r.off_skew = o.off_skew;
    r.wmin = o.wmin; r.wmax = o.wmax; r.inv_wmax = o.inv_wmax; r.nmin = o.nmin; r.nmax = o.nmax; r.lw_lo = o.lw_lo; r.lw_hi = o.lw_hi;
    return r;
}

template <typename T>
__global__ void __launch_bounds__(GX_THREADS) gfx_chain_bwd_kernel(const GfBwdArgs<T> a, const int64_t pstep, const int64_t tiles_total) {
    using Du = Dual<T>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Du* lds = reinterpret_cast<Du*>(smem_raw);
    T* red = reinterpret_cast<T*>(lds + JF_MAX_D_GF * GX_THREADS);       // one partial per wave (broadcast regime)
    const int tid = threadIdx.x, D = a.D;
    const XCol<Du> x{lds + tid};
    const bool bcast = pstep == 0;
    const int n_dir = D + a.n_params_total;
    bool first_tile = true;
    for (int64_t tile = blockIdx.x; tile < tiles_total || (bcast && first_tile); tile += gridDim.x) {
        const int64_t row = tile * GX_THREADS + tid;
        const bool active = row < a.B && tile < tiles_total;
        const int64_t rrow = active ? row : a.B - 1;
        const T gld = (a.g_ld && active) ? a.g_ld[rrow] : T(0);
        const T gblp = (a.g_blp && active) ? a.g_blp[rrow] : T(0);
        const T* prow = a.params + rrow * pstep;
        bool bad = false;
        for (int j = 0; j < n_dir; ++j) {
            for (int d = 0; d < D; ++d) x[d] = Du(a.x[rrow * a.xs + d], d == j ? T(1) : T(0));
            Du ld(T(0));
            for (int l = a.n_layers - 1; l >= 0; --l) {
                const GfLayerDev<Du> o = gx_dual_layer<T>(a.L[l]);
                const SeededRow<T> p{prow + o.col0, j - D - o.col0};
                if (o.model_offset) for (int d = 0; d < D; ++d) x[d] = x[d] - p[d];
                gx_rotate<Du, SeededRow<T>>(o, p, x, D, true);
                for (int d = 0; d < D; ++d) {
                    const GxCoord<Du> c = gx_prepare<Du, SeededRow<T>>(o, p, D, d);
                    const IcdfOut<Du> s = gf_icdf<Du>(o.inv_type, gx_mixture<Du, SeededRow<T>>(o, p, D, d, c, x[d]));
                    x[d] = s.y;
                    ld = ld + s.logd;
                }
            }
            T gj = gld * ld.d;
            for (int d = 0; d < D; ++d) {
                const Du v = x[d];
                const T gxo = (a.g_xout && active) ? a.g_xout[rrow * a.gxos + d] : T(0);
                gj += (gxo - v.v * gblp) * v.d;
            }
            if (!active) gj = T(0);
            bad = bad || !M<T>::finite(gj);
            if (j < D) {
                if (active) a.g_x[row * a.gxs + j] = gj;
            } else if (!bcast) {
                if (active) a.g_params[row * a.gps + (j - D)] = gj;
            } else {                                               // this workgroup's partial row: sum over its lanes, over its tiles
                T s = gj;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
                __syncthreads();
                if ((tid & 63) == 0) red[tid >> 6] = s;
                __syncthreads();
                if (tid == 0) {
                    T tot = T(0);
                    for (int w = 0; w < GX_THREADS / 64; ++w) tot += red[w];
                    T* dst = a.g_params + (int64_t)blockIdx.x * a.gps + (j - D);
                    *dst = first_tile ? tot : *dst + tot;
                }
            }
        }
        status_add(a.status, JF_STATUS_NONFINITE, active && bad);
        first_tile = false;
    }
}

// ---------------------------------------------------------------------------------------------------------- host side
static inline int gb_group_width(int D) { return D <= 1 ? 1 : D <= 2 ? 2 : D <= 4 ? 4 : 8; }

template <typename T> static int gb_fill(GfBwdArgs<T>& a, const T* params, int64_t ps, bool bcast, int32_t D, int32_t n_layers, const jf_gf_layer* layers,
                                         bool& ext) {
    int col = 0, maxp = 0;
    ext = false;
    for (int l = 0; l < n_layers; ++l) {
        const jf_gf_layer& h = layers[l];
        GfLayerDev<T>& o = a.L[l];
        if (h.num_kde < 1 || h.hh_iter < 0 || h.width_min <= 0) return JF_ERR_BADARG;
        const bool ext_layer = h.rotation_mode != JF_GF_ROT_HOUSEHOLDER || h.center_mean || h.add_skewness;     // general-option layer (jf_gf_ext.h)
        if (h.rotation_mode < JF_GF_ROT_HOUSEHOLDER || h.rotation_mode > JF_GF_ROT_TRIANGULAR || (h.rotation_mode == JF_GF_ROT_CAYLEY && D > 2)) return JF_ERR_BADARG;
        if (h.add_skewness && sizeof(T) != 8) return JF_ERR_UNSUPPORTED;                                        // float64 only, as the forward
        if (ext_layer) ext = true;
        if (h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION && h.width_max <= 0) return JF_ERR_BADARG;
        if (h.nonlinear_stretch_type != JF_GF_STRETCH_CLASSIC) return JF_ERR_UNSUPPORTED;
        o.K = h.num_kde; o.hh = h.hh_iter; o.model_offset = h.model_offset; o.fit_norm = h.fit_normalization;
        o.reg_norm = h.regulate_normalization; o.inv_type = h.inverse_function_type; o.width_mode = h.width_mode;
        o.clamp_widths = h.clamp_widths;
        o.fast = (h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION && !h.clamp_widths && h.fit_normalization && h.regulate_normalization) ? 1 : 0;
        o.stretch = JF_GF_STRETCH_CLASSIC; o.off_box = 0;
        const int kd = h.num_kde * D;
        o.rot_mode = h.rotation_mode; o.center_mean = h.center_mean ? 1 : 0; o.skew = h.add_skewness ? 1 : 0;
        if (o.rot_mode != JF_GF_ROT_HOUSEHOLDER) o.hh = 0;
        o.off_rot = h.model_offset ? D : 0;
        o.off_mean = o.off_rot + gx_rot_len(o.rot_mode, o.hh, D);
        o.off_lw = o.off_mean + kd - (o.center_mean ? D : 0);
        o.off_ln = o.off_lw + kd;
        o.off_skew = o.off_ln + (h.fit_normalization ? kd : 0);
        o.n_params = o.off_skew + (o.skew ? kd : 0);
        o.col0 = col;
        o.vec_ok = (!bcast && aligned16<T>(params, ps, col) && (o.n_params % Vec16<T>::N == 0)) ? 1 : 0;
        o.wmin = (T)h.width_min; o.wmax = (T)h.width_max; o.inv_wmax = h.width_max > 0 ? (T)(1.0 / h.width_max) : T(0);
        o.nmin = (T)h.norm_min; o.nmax = (T)h.norm_max;
        o.lw_lo = (T)log(0.01 * h.width_min);
        if (h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION) o.lw_hi = (T)(3.0 * log(h.width_max));
        else o.lw_hi = h.width_max > 0 ? (T)log(h.width_max) : (T)INFINITY;
        col += o.n_params;
        if (o.n_params > maxp) maxp = o.n_params;
    }
    a.n_params_total = col;
    a.tile_stride = padded_stride<T>(maxp);
    return JF_OK;
}

// number of partial-sum rows a broadcast launch writes for B rows (the caller allocates (n, P) and sums over n)
static int64_t gb_partials(int64_t B, int D) {
    const int G = gb_group_width(D);
    const int64_t n_tiles = (B + 256 / G - 1) / (256 / G);
    const int64_t blocks = n_tiles < 1024 ? (n_tiles < 1 ? 1 : n_tiles) : 1024;
    return blocks;
}

template <typename T, int G>
static int gb_launch(GfBwdArgs<T> a, bool bcast, hipStream_t st) {
    if (bcast) {
        const int64_t n_tiles = (a.B + 256 / G - 1) / (256 / G);
        const int64_t blocks = gb_partials(a.B, a.D);
        a.tiles_per_block = (int)((n_tiles + blocks - 1) / blocks);
        const size_t lds = (size_t)2 * a.n_layers * a.tile_stride * sizeof(T);
        auto k = gf_chain_bwd_kernel<T, G, true>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, st, a);
    } else {
        a.tiles_per_block = 1;
        const size_t lds = (size_t)2 * (64 / G) * a.tile_stride * sizeof(T);
        auto k = gf_chain_bwd_kernel<T, G, false>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3((unsigned)((a.B + 64 / G - 1) / (64 / G))), dim3(64), lds, st, a);
    }
    return check_launch();
}

template <typename T>
static int gf_chain_inv_bwd(const T* x, int64_t xs, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n_layers,
                            const jf_gf_layer* layers, const T* g_xout, int64_t gxos, const T* g_ld, const T* g_blp, T* g_x, int64_t gxs, T* g_params,
                            int64_t gps, int32_t* status, void* stream) {
    if (!x || !params || !g_x || !g_params || !layers) return JF_ERR_BADARG;
    if (n_layers < 1 || n_layers > JF_MAX_CHAIN || D < 1 || B < 0) return JF_ERR_BADARG;
    if (D > 8) return JF_ERR_UNSUPPORTED;
    if (pb != 1 && pb != B) return JF_ERR_BADARG;
    const bool bcast = pb == 1;
    GfBwdArgs<T> a{};
    bool ext = false;
    const int rc = gb_fill<T>(a, params, ps, bcast, D, n_layers, layers, ext);
    if (rc != JF_OK) return rc;
    if (B == 0) return JF_OK;
    a.x = x; a.xs = xs; a.params = params; a.ps = ps; a.B = B; a.D = D; a.n_layers = n_layers;
    a.g_xout = g_xout; a.gxos = gxos; a.g_ld = g_ld; a.g_blp = g_blp; a.g_x = g_x; a.gxs = gxs; a.g_params = g_params; a.gps = gps; a.status = status;
    if (ext) {                                                     // forward-mode kernel; broadcast: one partial row per workgroup, as the adjoint kernel
        const int64_t tiles = (B + GX_THREADS - 1) / GX_THREADS;
        const int64_t blocks = bcast ? gb_partials(B, D) : tiles;
        const size_t lds = (size_t)JF_MAX_D_GF * GX_THREADS * sizeof(Dual<T>) + 16 * sizeof(T);
        auto k = gfx_chain_bwd_kernel<T>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(GX_THREADS), lds, (hipStream_t)stream, a, bcast ? (int64_t)0 : ps, tiles);
        return check_launch();
    }
    for (int l = 0; l < n_layers; ++l) if (layers[l].hh_iter > GB_MAX_HH) return JF_ERR_UNSUPPORTED;
    switch (gb_group_width(D)) {
        case 1: return gb_launch<T, 1>(a, bcast, (hipStream_t)stream);
        case 2: return gb_launch<T, 2>(a, bcast, (hipStream_t)stream);
        case 4: return gb_launch<T, 4>(a, bcast, (hipStream_t)stream);
        default: return gb_launch<T, 8>(a, bcast, (hipStream_t)stream);
    }
}

}  // namespace jf

extern "C" {
int64_t jf_gf_chain_inv_bwd_partials(int64_t B, int32_t D) { return jf::gb_partials(B, D); }
int jf_gf_chain_inv_bwd_f32(const float* x, int64_t xs, const float* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n, const jf_gf_layer* L,
                            const float* gxo, int64_t gxos, const float* gld, const float* gblp, float* gx, int64_t gxs, float* gp, int64_t gps,
                            int32_t* st, void* s) {
    return jf::gf_chain_inv_bwd<float>(x, xs, p, ps, pb, B, D, n, L, gxo, gxos, gld, gblp, gx, gxs, gp, gps, st, s);
}
int jf_gf_chain_inv_bwd_f64(const double* x, int64_t xs, const double* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n, const jf_gf_layer* L,
                            const double* gxo, int64_t gxos, const double* gld, const double* gblp, double* gx, int64_t gxs, double* gp, int64_t gps,
                            int32_t* st, void* s) {
    return jf::gf_chain_inv_bwd<double>(x, xs, p, ps, pb, B, D, n, L, gxo, gxos, gld, gblp, gx, gxs, gp, gps, st, s);
}
}
