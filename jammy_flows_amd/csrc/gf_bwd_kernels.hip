// Backward (vector-Jacobian product) of a chain of 'g' layers in the log-prob direction: what torch.autograd produces for
// gf_block._inv_flow_mapping + the euclidean_base offset (gaussianization_flow.py:995-1114, euclidean_base.py:34-51) when the loss is a
// function of (x_out, log_det_out, base_logp_out) -- the training step of the reference (examples/jammy_flows.py:381-412,
// docs/source/usage/training.rst:24-44) replays ~560 eager ops per layer for it; here it is ONE launch per e-block.
//
//   jf_gf_chain_inv_bwd_*   inputs : x, params (as in jf_gf_chain_inv), upstream gradients g_x_out (B, D), g_log_det (B), g_base_logp (B)
//                           outputs: g_x (B, D); g_params: per-sample regime (B, P) -- one row per sample, the layout of `params`, what the
//                                    amortisation MLP's backward consumes --, broadcast regime (n_partials, P) partial sums, one row per
//                                    workgroup, that the caller adds up (inside a workgroup: float64 LDS atomics, see gf_chain_bwd_kernel)
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
#include <cstdlib>
#include "jf_gf_bwd.h"

namespace jf {

constexpr int GB_MAX_HH = 8;                // reflections kept in registers for groups of up to 8 lanes; wider groups keep G (gb_max_hh)
template <int G> constexpr int gb_max_hh() { return G > GB_MAX_HH ? G : GB_MAX_HH; }
constexpr int GB_NT = 512;                  // threads of a broadcast-regime workgroup: the derived rows / accumulators / records in LDS (~40 KB) are
                                            //   per workgroup, so wider workgroups mean more resident waves per CU (256: 2 per SIMD, 512: 4)

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
template <typename T, int G, bool ACC, typename GP>
__device__ __forceinline__ T gf_layer_bwd(const T* __restrict__ p, GP* __restrict__ gp, const GfLayerDev<T>& o, int D, bool live, T x_in, T gy, T gl, int slsh) {
    // ACC (broadcast parameters): gp points at this lane's accumulator SLOT of the coordinate's first parameter; the 2^slsh slots of a
    // parameter are dealt to the rows of a wave, so the lanes of one LDS atomic hit distinct addresses (see gf_chain_bwd_kernel).
    auto put = [&](int off, T v) {
        if constexpr (ACC) {
            if (live) atomicAdd(gp + (off << slsh), (GP)v);
        } else {
            if (live) gp[off] = v;
        }
    };
    // ---- recompute: offset, reflections (keeping the vector before each one), mixture
    T xr[gb_max_hh<G>()];
    T x = x_in;
    if (o.model_offset) x -= p[0];
#pragma unroll
    for (int i = 0; i < gb_max_hh<G>(); ++i) {
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
    for (int i = gb_max_hh<G>() - 1; i >= 0; --i) {
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

// ---------------------------------------------------------------------------------------------------------- linear-space fast path
// The log-space loop above spends ~8 exp / log and ~8 divisions per component.  Wherever the linear-space mixture of the forward kernels is
// valid (cdf, sf, pdf far from underflow -- every row but the deep tails) the responsibilities are plain products,
//     g_lc rC_k = pi_k sigma(u_k) (g_lc / cdf),  g_ls rS_k = pi_k sigma(-u_k) (g_ls / sf),  g_lp rP_k = pi_k sigma(u_k) sigma(-u_k) (g_lp / (w_k pdf)),
// with three reciprocals per COORDINATE instead of per component.  A wave in which some live lane is outside that range takes the log-space
// function for the layer (wave-uniform branch), so both paths see exactly the rows the forward's two paths see.
// Householder part shared by the two fast paths: reflections last first, g_v written through `put`
template <typename T, int G, typename PUT>
__device__ __forceinline__ T gf_reflections_bwd(const T* __restrict__ p, const GfLayerDev<T>& o, int D, bool live, const T* xr, T g, PUT put) {
#pragma unroll
    for (int i = gb_max_hh<G>() - 1; i >= 0; --i) {
        if (i < o.hh) {
            const T v = live ? p[o.off_rot + i * D] : T(0);
            const T n = group_sum<T, G>(v * v), sx = group_sum<T, G>(v * xr[i]), vg = group_sum<T, G>(v * (live ? g : T(0)));
            const T rn = M<T>::rcp(n);
            put(o.off_rot + i * D, -T(2) * sx * rn * g - T(2) * vg * rn * xr[i] + T(4) * sx * vg * rn * rn * v);
            g -= T(2) * vg * rn * v;
        }
    }
    return g;
}

// per-sample regime, layers with the reference's default options (o.fast): raw row p, gradient row gp (this lane's column of the tile).
// The forward sweep of the kernel keeps the layer's normalised linear-space sums (cdf, sf, pdf) and 1/N in registers (MixSums, 4 per layer),
// so the backward sweep does not evaluate the mixture again: its single loop regulates width and weight of a component (2 exp + 2 rcp),
// evaluates the logistic (1 exp + 1 rcp) and forms the three gradient values.  (Round 2 first re-ran the mixture loop here and parked
// 1/w, its derivative and the weight sigmoid in the gradient slots for a second loop: one more exp + rcp, ~15 more VALU instructions and six
// LDS accesses per component.)

template <typename T> __device__ __forceinline__ MixQ<T> gfb_mixture_fast_raw(const T* __restrict__ p, const GfLayerDev<T>& o, int D, T x, MixSums<T>& m) {
    const T* pm = p + o.off_mean;
    const T* pw = p + o.off_lw;
    const T* pn = p + o.off_ln;
    T C = T(0), S = T(0), P = T(0), Nn = T(0);
#pragma unroll 2
    for (int k = 0; k < o.K; ++k) {
        const T mu = pm[k * D], rw = pw[k * D], rn = pn[k * D];
        const T ae = o.inv_wmax + M<T>::exp_fast(-rw);
        const T iw = ae * M<T>::rcp(o.wmin * ae + T(1));
        const T wk = o.nmin + o.nmax * M<T>::rcp(T(1) + M<T>::exp_fast(-rn));
        const T u = (x - mu) * iw;
        const T t = M<T>::exp_fast(-M<T>::abs(u));
        const T hi = M<T>::rcp(T(1) + t), lo = t * hi;
        const bool pos = u >= T(0);
        C += wk * (pos ? hi : lo);
        S += wk * (pos ? lo : hi);
        P += wk * hi * lo * iw;
        Nn += wk;
    }
    const T invN = M<T>::rcp(Nn);
    C *= invN; S *= invN; P *= invN;
    m.C = C; m.S = S; m.P = P; m.invN = invN;
    MixQ<T> q;
    q.lc = M<T>::log_fast(C); q.ls = M<T>::log_fast(S); q.lp = M<T>::log_fast(P); q.cdf = C; q.sf = S;
    const bool under = !(C > M<T>::TINY && S > M<T>::TINY && P > M<T>::TINY);
    if (__any(under)) {                                            // wave-uniform; the backward sweep sees C = 0 and takes the log-space function
        const MixQ<T> qs = gfg_mixture_scaled<T, true>(p, o, D, x, T(0));
        if (under) { q = qs; m.C = T(0); }
    }
    return q;
}

template <typename T, int G>
__device__ __forceinline__ T gf_layer_bwd_fast(const T* __restrict__ p, T* __restrict__ gp, const GfLayerDev<T>& o, int D, bool live, T x_in, T gy, T gl,
                                               const MixSums<T>& m) {
    const bool ok = m.C > LinRange<T>::lo && m.S > LinRange<T>::lo && m.P > LinRange<T>::lo && m.P < LinRange<T>::hi;
    if (!__all(ok)) return gf_layer_bwd<T, G, false, T>(p, gp, o, D, live, x_in, gy, gl, 0);
    T xr[gb_max_hh<G>()];
    T x = x_in;
    if (o.model_offset) x -= p[0];
#pragma unroll
    for (int i = 0; i < gb_max_hh<G>(); ++i) {
        xr[i] = x;
        if (i < o.hh) x = gfg_reflect<T, G, true>(p, o.off_rot + i * D, live, x);
    }
    const T* pm = p + o.off_mean;
    const T* pw = p + o.off_lw;
    const T* pn = p + o.off_ln;
    T* gm = gp + o.off_mean;
    T* gw = gp + o.off_lw;
    T* gn = gp + o.off_ln;
    const T C = m.C, S = m.S, P = m.P, invN = m.invN;
    MixQ<T> q;
    q.lc = M<T>::log_fast(C); q.ls = M<T>::log_fast(S); q.lp = M<T>::log_fast(P); q.cdf = C; q.sf = S;
    const IcdfOut<T> s = gf_icdf<T>(o.inv_type, q);
    const IcdfCoef<T> c = gf_icdf_coeffs<T>(o.inv_type, q, s.y);
    const T g_lc = gy * c.Ay + gl * c.AH, g_ls = gy * c.By + gl * c.BH, g_lp = gl;
    const T Gsum = g_lc + g_ls + g_lp;
    const T icg = g_lc * M<T>::rcp(C), isg = g_ls * M<T>::rcp(S), ipg = g_lp * M<T>::rcp(P);
    T gx = T(0);
#pragma unroll 2
    for (int k = 0; k < o.K; ++k) {
        const T mu = pm[k * D], rw = pw[k * D], rn = pn[k * D];
        const T e = M<T>::exp_fast(-rw);
        const T ae = o.inv_wmax + e;
        const T r2 = M<T>::rcp(ae * (o.wmin * ae + T(1)));
        const T iw = ae * ae * r2, dliw = -e * r2;               // 1 / w,  d log(1 / w) / d raw
        const T sgn = M<T>::rcp(T(1) + M<T>::exp_fast(-rn));
        const T pik = (o.nmin + o.nmax * sgn) * invN;
        const T u = (x - mu) * iw;
        const T t = M<T>::exp_fast(-M<T>::abs(u));
        const T hi = M<T>::rcp(T(1) + t), lo = t * hi;
        const bool pos = u >= T(0);
        const T sg = pos ? hi : lo, sgc = pos ? lo : hi;
        const T a = sg * icg, b = sgc * isg, cp = sg * sgc * iw * ipg;           // g . responsibility / pi_k
        const T gu = pik * (a * sgc - b * sg + cp * (sgc - sg));
        gx += gu * iw;
        if (live) {
            gm[k * D] = -gu * iw;
            gw[k * D] = (gu * u + pik * cp) * dliw;
            gn[k * D] = (a + b + cp - Gsum) * (o.nmax * sgn * (T(1) - sgn) * invN);
        }
    }
    const T g = gf_reflections_bwd<T, G>(p, o, D, live, xr, gx, [&](int off, T v) { if (live) gp[off] = v; });
    if (o.model_offset && live) gp[0] = -g;
    return g;
}

// broadcast regime, any options.  Per (component, coordinate) the prologue of the kernel packs one 8-word record
//     c = {mean, 1/w, pi_k, d log(1/w)/d raw, pi_k d log n_k/d raw, -, -, -}
// so that a component costs two 16-byte LDS reads, issued one component ahead of their use; the three gradient values of a component go
// straight to this lane's accumulator slots (unconditional LDS atomics: lanes without a row add 0, nothing in the loop waits for them).
// v: derived row (the forward's layout: fallback of the mixture, reflections as sqrt(2) v/|v|), p: raw row (offsets, raw Householder vectors).
template <typename T> __device__ __forceinline__ MixQ<T> gfb_mixture_pk(const T* __restrict__ c0, const T* __restrict__ v, const GfLayerDev<T>& o, int D, T x,
                                                                  T& Cs, T& Ss, T& Ps) {
    const T* c = (const T*)__builtin_assume_aligned(c0, 16);
    T C = T(0), S = T(0), P = T(0);
#pragma unroll 2
    for (int k = 0; k < o.K; ++k, c += 8 * D) {
        const T mu = c[0], iw = c[1], wk = c[2];
        const T u = (x - mu) * iw;
        const T t = M<T>::exp_fast(-M<T>::abs(u));
        const T hi = M<T>::rcp(T(1) + t), lo = t * hi;
        const bool pos = u >= T(0);
        C += wk * (pos ? hi : lo);
        S += wk * (pos ? lo : hi);
        P += wk * hi * lo * iw;
    }
    MixQ<T> q;
    q.lc = M<T>::log_fast(C); q.ls = M<T>::log_fast(S); q.lp = M<T>::log_fast(P);
    q.cdf = C; q.sf = S;
    Cs = C; Ss = S; Ps = P;
    const bool under = !(C > M<T>::TINY && S > M<T>::TINY && P > M<T>::TINY);
    if (__any(under)) {
        const MixQ<T> qs = gfg_mixture_scaled<T, false>(v, o, D, x, T(0));
        if (under) { q = qs; Cs = T(0); }                         // the backward sweep takes the log-space function for this wave
    }
    return q;
}

template <typename T, int G>
__device__ __forceinline__ T gf_layer_bwd_bcast(const T* __restrict__ p, const T* __restrict__ v, const T* __restrict__ c0, double* __restrict__ gp,
                                                const GfLayerDev<T>& o, int D, bool live, T x_in, T gy, T gl, int slsh, T Cs, T Ss, T Ps) {
    // Cs, Ss, Ps: the layer's linear-space sums from the forward sweep (Cs = 0: the wave needed the scaled evaluation there)
    const bool ok = Cs > LinRange<T>::lo && Ss > LinRange<T>::lo && Ps > LinRange<T>::lo && Ps < LinRange<T>::hi;
    T xr[gb_max_hh<G>()];
    T x = x_in;
    if (o.model_offset) x -= p[0];
#pragma unroll
    for (int i = 0; i < gb_max_hh<G>(); ++i) {
        xr[i] = x;
        if (i < o.hh) x = gfg_reflect<T, G, false>(v, o.off_rot + i * D, live, x);
    }
    if (!__all(ok)) {
        // A lane of the wave sits where the linear-space sums under- or overflow (a target tens of widths from every component).  Rounds 2-3 sent
        // the whole wave through the log-space loop of gf_layer_bwd (8 exp / log and 8 divisions per component); measured in round 4 on the
        // SURVEY inputs of C3's block 0 that was HALF of this kernel (0.265 ms against 0.121 on the model's own samples: one tail row per 16
        // is enough to send every wave there).  The responsibilities only need the sums of gfg_mixture_scaled -- everything scaled by e^{m},
        // m = distance to the nearest component -- because every ratio that occurs has e^{-m} on both sides:
        //     s (1 - s) = e^{-m} t' h^2        (t' = e^{m - |u|}, h = sigma(|u|)),     e^{-m} / cdf = em / (Cu + em Cs)  or  1 / Cs  when Cu = 0,
        //     pdf = e^{-m} Ps   =>   s (1 - s) / (w pdf) = t' h^2 / (w Ps),           the same for sf.
        // Three passes over the component records (distance, sums, gradients): ~55 instructions per component.
        const T* c = (const T*)__builtin_assume_aligned(c0, 16);
        T m = T(INFINITY);
        for (int k = 0; k < o.K; ++k) m = M<T>::min(m, M<T>::abs((x - c[8 * D * k]) * c[8 * D * k + 1]));
        const T em = M<T>::exp_fast(-m);
        T Cu = T(0), Cq = T(0), Su = T(0), Sq = T(0), Pq = T(0);
        for (int k = 0; k < o.K; ++k) {
            const T iwk = c[8 * D * k + 1], pk = c[8 * D * k + 2];
            const T u = (x - c[8 * D * k]) * iwk;
            const T tp = M<T>::exp_fast(m - M<T>::abs(u));
            const T h = M<T>::rcp(T(1) + tp * em);
            const T c1 = pk * h, c2 = c1 * tp;
            if (u >= T(0)) { Cu += c1; Sq += c2; } else { Su += c1; Cq += c2; }
            Pq += c2 * h * iwk;
        }
        MixQ<T> q;
        q.cdf = Cu + em * Cq;
        q.sf = Su + em * Sq;
        q.lc = Cu > T(0) ? M<T>::log_fast(q.cdf) : M<T>::log_fast(Cq) - m;
        q.ls = Su > T(0) ? M<T>::log_fast(q.sf) : M<T>::log_fast(Sq) - m;
        q.lp = M<T>::log_fast(Pq) - m;
        const IcdfOut<T> s = gf_icdf<T>(o.inv_type, q);
        const IcdfCoef<T> cf = gf_icdf_coeffs<T>(o.inv_type, q, s.y);
        const T g_lc = gy * cf.Ay + gl * cf.AH, g_ls = gy * cf.By + gl * cf.BH, g_lp = gl;
        const T Gsum = g_lc + g_ls + g_lp;
        const T r1c = Cu > T(0) ? M<T>::rcp(q.cdf) : T(0), r1s = Su > T(0) ? M<T>::rcp(q.sf) : T(0);   // 1 / cdf, 1 / sf: for components on their side
        const T a2c = Cu > T(0) ? em * r1c : M<T>::rcp(Cq), a2s = Su > T(0) ? em * r1s : M<T>::rcp(Sq);  // e^{-m} / cdf, e^{-m} / sf
        const T ap = M<T>::rcp(Pq);
        const bool fit = o.fit_norm != 0;
        double* am = gp + (o.off_mean << slsh);
        double* aw = gp + (o.off_lw << slsh);
        double* an = gp + ((fit ? o.off_ln : o.off_lw) << slsh);
        const int step = D << slsh;
        T gx = T(0);
        for (int k = 0; k < o.K; ++k, c += 8 * D) {
            const T mu = c[0], iwk = c[1], pk = c[2], flw = c[3], fln = c[4];
            const T u = (x - mu) * iwk;
            const T tp = M<T>::exp_fast(m - M<T>::abs(u));
            const T h = M<T>::rcp(T(1) + tp * em);
            const bool pos = u >= T(0);
            const T th = tp * h, w2 = th * h;                    // s (1 - s) = em w2
            const T sk = pos ? h : em * th;                      // sigma(u)
            const T pp = w2 * iwk * ap;                          // s (1 - s) / (w pdf)
            const T sc = pos ? h * r1c : th * a2c;               // s / cdf
            const T ss = pos ? th * a2s : h * r1s;               // (1 - s) / sf
            const T gu = pk * (w2 * (g_lc * a2c - g_ls * a2s) + g_lp * pp * (T(1) - T(2) * sk));
            gx += gu * iwk;
            atomicAdd(am, live ? (double)(-gu * iwk) : 0.0);
            atomicAdd(aw, live ? (double)((gu * u + g_lp * pk * pp) * flw) : 0.0);
            atomicAdd(an, live ? (double)((g_lc * sc + g_ls * ss + g_lp * pp - Gsum) * fln) : 0.0);     // (fln carries pi_k d log n_k / d raw)
            am += step; aw += step; an += step;
        }
        auto put = [&](int off, T val) { atomicAdd(gp + (off << slsh), live ? (double)val : 0.0); };
        const T g = gf_reflections_bwd<T, G>(p, o, D, live, xr, gx, put);
        if (o.model_offset) put(0, -g);
        return g;
    }
    MixQ<T> q;
    q.lc = M<T>::log_fast(Cs); q.ls = M<T>::log_fast(Ss); q.lp = M<T>::log_fast(Ps); q.cdf = Cs; q.sf = Ss;
    const IcdfOut<T> s = gf_icdf<T>(o.inv_type, q);
    const IcdfCoef<T> cf = gf_icdf_coeffs<T>(o.inv_type, q, s.y);
    const T g_lc = gy * cf.Ay + gl * cf.AH, g_ls = gy * cf.By + gl * cf.BH, g_lp = gl;
    const T Gsum = g_lc + g_ls + g_lp;
    const T icg = g_lc * M<T>::rcp(Cs), isg = g_ls * M<T>::rcp(Ss), ipg = g_lp * M<T>::rcp(Ps);
    const bool fit = o.fit_norm != 0;
    double* am = gp + (o.off_mean << slsh);
    double* aw = gp + (o.off_lw << slsh);
    double* an = gp + ((fit ? o.off_ln : o.off_lw) << slsh);          // not fitted: the (zero) weight term lands on the width accumulator
    const int step = D << slsh;
    const T* c = (const T*)__builtin_assume_aligned(c0, 16);
    T mu = c[0], iw = c[1], pik = c[2], flw = c[3], fln = c[4];
    T gx = T(0);
    for (int k = 0; k < o.K; ++k) {
        c += (k + 1 < o.K) ? 8 * D : 0;
        const T mu_n = c[0], iw_n = c[1], pik_n = c[2], flw_n = c[3], fln_n = c[4];
        __builtin_amdgcn_sched_barrier(0);                     // keep the reads of the next record ahead of this component's arithmetic
        const T u = (x - mu) * iw;
        const T t = M<T>::exp_fast(-M<T>::abs(u));
        const T hi = M<T>::rcp(T(1) + t), lo = t * hi;
        const bool pos = u >= T(0);
        const T sg = pos ? hi : lo, sgc = pos ? lo : hi;
        const T a = sg * icg, b = sgc * isg, cp = sg * sgc * iw * ipg;
        const T gu = pik * (a * sgc - b * sg + cp * (sgc - sg));
        gx += gu * iw;
        // (round 4: summing the wave's rows in registers first -- rotations + permlane swaps, 3 values per component -- and leaving the atomics to
        // the first row's lanes made this kernel SLOWER, 0.264 -> 0.291 ms at 2^18 rows of C3's block 0; summing only the 4 rows of a 16-lane DPP
        // row and issuing the three values as ONE atomic instruction -- a third of the instructions, a quarter of the lanes -- left it at 0.2646:
        // the slotted atomics below are not what it waits for)
        atomicAdd(am, live ? (double)(-gu * iw) : 0.0);         // selected, not multiplied: a shadow lane (g >= D) may hold inf / nan
        atomicAdd(aw, live ? (double)((gu * u + pik * cp) * flw) : 0.0);
        atomicAdd(an, live ? (double)((a + b + cp - Gsum) * fln) : 0.0);
        am += step; aw += step; an += step;
        mu = mu_n; iw = iw_n; pik = pik_n; flw = flw_n; fln = fln_n;
    }
    auto put = [&](int off, T val) { atomicAdd(gp + (off << slsh), live ? (double)val : 0.0); };
    const T g = gf_reflections_bwd<T, G>(p, o, D, live, xr, gx, put);
    if (o.model_offset) put(0, -g);
    return g;
}

// broadcast regime, prologue: derived values into v (1/w, pi_k), gradient factors into f (see gf_layer_bwd_bcast); one (layer, component,
// coordinate) per thread, the normalisers of the weights through aux[(l * 8 + d) * 2 + {0: shift, 1: 1/N}]
template <typename T> __device__ __forceinline__ void gf_derive_bwd_item(const T* __restrict__ raw, T* __restrict__ v, T* __restrict__ f, const GfLayerDev<T>& o,
                                                                         int i, T shift) {
    T iw, dliw;
    gf_inv_width_grad<T>(o, raw[o.off_lw + i], iw, dliw);
    v[o.off_lw + i] = iw;
    f[o.off_lw + i] = dliw;
    if (o.fit_norm) {
        const T rn = raw[o.off_ln + i];
        const T nk = gf_weight(o, rn, shift);
        T ndl = nk;                                              // n_k d log n_k / d raw
        if (o.reg_norm) { const T sg = M<T>::rcp(T(1) + M<T>::exp(-rn)); ndl = o.nmax * sg * (T(1) - sg); }
        v[o.off_ln + i] = nk;
        f[o.off_ln + i] = ndl;
    }
}

// DIRECT (per-sample regime): the lanes store their gradient values straight to the (B, P) rows instead of through a gradient tile in LDS.
// Chosen when the D coordinate lanes of a row cover >= 32 contiguous bytes per store (float64 D >= 4, float32 D = 8): the tile was half of the
// workgroup's LDS, which bounds the resident waves of this kernel (float64 D = 8: 39 KB per wave -> one wave per SIMD).
template <typename T, int G, bool BCAST, bool DIRECT = false>
__global__ void __launch_bounds__(BCAST ? GB_NT : 64) gf_chain_bwd_kernel(const GfBwdArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    constexpr int NT = BCAST ? GB_NT : 64;
    constexpr int R = NT / G;
    const int tid = threadIdx.x;
    constexpr int LG = G == 1 ? 0 : G == 2 ? 1 : G == 4 ? 2 : G == 8 ? 3 : G == 16 ? 4 : G == 32 ? 5 : 6;
    constexpr int DM = G > 8 ? G : 8, LDM = G > 8 ? LG : 3;      // coordinate slots per layer of the normaliser table (aux)
    const int g = tid & (G - 1), r = tid >> LG;
    const int D = a.D;
    const bool live = g < D, leader = g == 0;
    const int d = live ? g : D - 1;
    const int ts = a.tile_stride, nl = a.n_layers;
    // LDS: BCAST: [raw rows of all layers: n_layers x tile_stride][derived rows][gradient factors][accumulators: n_layers x tile_stride x SL]
    //             [layer inputs and the forward sweep's mixture sums: 4 x n_layers x NT][normalisers: n_layers x 16][packed component records]
    //      per-sample: [parameter tile R x tile_stride][gradient tile R x tile_stride]
    // BCAST: the layer inputs of the forward sweep live in LDS (one word per layer and lane) so that both sweeps are real loops over the layers: the
    // unrolled form (a register array indexed by the layer) replicated the two layer bodies JF_MAX_CHAIN times -- 190 KB of code.
    // BCAST accumulators: every parameter has SL = 2^slsh slots and row r of the tile adds into slot r % SL.  With SL = 64 / G the lanes of a
    // wave own distinct words (lane -> (coordinate, slot) is a bijection onto 64 consecutive words: no bank conflict, no serialised atomic);
    // only the waves of the workgroup share an address, which is what the LDS atomic is for.  Round 2 first summed over the rows of a
    // wave with a 4..6-step butterfly per value (3 values per component) and then issued one atomic per coordinate.
    // The accumulators are float64 in BOTH precisions: ds_add_f32 costs ~190 cycles per wave instruction on gfx950, ds_add_f64 9 (20 with two
    // lanes per address), ds_add_u32 3.4 (scripts/probe/lds_atomic.hip) -- the float32 chain with float32 accumulators spent 0.6 of its
    // 1.04 ms per 2^18 rows inside those atomics.
    T* ptile = lds;
    T* gtile = lds + (BCAST ? nl : R) * ts;
    T* vtile = gtile;                                            // BCAST only
    T* ftile = vtile + nl * ts;
    double* acc = reinterpret_cast<double*>(ftile + nl * ts);    // float64 accumulators for both precisions, see below
    const int slsh = BCAST ? a.slsh : 0;
    T* xin = reinterpret_cast<T*>(acc + ((nl * ts) << slsh)) + tid;   // BCAST only
    T* pk = xin - tid + nl * (4 * NT + 2 * DM);                      // BCAST only: packed component records (16-byte aligned: every term is a multiple of 4)
    if constexpr (BCAST) {
        if ((int)blockIdx.x >= a.active_blocks) {                // more partial rows than resident workgroups: zero rows
            for (int j = tid; j < a.n_params_total; j += NT) a.g_params[(int64_t)blockIdx.x * a.gps + j] = T(0);
            return;
        }
        T* aux = xin - tid + nl * 4 * NT;                        // (xin: layer input and the three mixture sums per layer and lane)
        for (int l = 0; l < nl; ++l) {
            const GfLayerDev<T> o = a.L[l];
            for (int j = tid; j < ts; j += NT) {
                const T raw = j < o.n_params ? a.params[o.col0 + j] : T(0);
                ptile[l * ts + j] = raw;
                vtile[l * ts + j] = raw;
                ftile[l * ts + j] = T(0);
            }
        }
        for (int j = tid; j < ((nl * ts) << slsh); j += NT) acc[j] = 0.0;
        __syncthreads();
        for (int w = tid; w < nl * DM; w += NT) {                // unbounded log-weights: shift by the column's maximum
            const int l = w >> LDM, dd = w & (DM - 1);
            const GfLayerDev<T> o = a.L[l];
            T shift = T(0);
            if (dd < D && o.fit_norm && !o.reg_norm) {
                shift = ptile[l * ts + o.off_ln + dd];
                for (int k = 1; k < o.K; ++k) shift = M<T>::max(shift, ptile[l * ts + o.off_ln + k * D + dd]);
            }
            aux[w * 2] = shift;
        }
        for (int w = tid; w < nl * gb_max_hh<G>(); w += NT) {
            const int l = w / gb_max_hh<G>(), i = w - l * gb_max_hh<G>();
            if (i < a.L[l].hh) gf_derive_reflection<T>(vtile + l * ts, a.L[l], D, i);
        }
        __syncthreads();
        for (int l = 0; l < nl; ++l) {
            const GfLayerDev<T> o = a.L[l];
            for (int i = tid; i < o.K * D; i += NT)
                gf_derive_bwd_item<T>(ptile + l * ts, vtile + l * ts, ftile + l * ts, o, i, aux[(l * DM + i % D) * 2]);
        }
        __syncthreads();
        for (int w = tid; w < nl * DM; w += NT) {
            const int l = w >> LDM, dd = w & (DM - 1);
            const GfLayerDev<T> o = a.L[l];
            T Nn = T(0);
            if (dd < D && o.fit_norm) for (int k = 0; k < o.K; ++k) Nn += vtile[l * ts + o.off_ln + k * D + dd];
            aux[w * 2 + 1] = T(1) / Nn;
        }
        __syncthreads();
        for (int l = 0; l < nl; ++l) {
            const GfLayerDev<T> o = a.L[l];
            if (o.fit_norm) for (int i = tid; i < o.K * D; i += NT) {
                const T invN = aux[(l * DM + i % D) * 2 + 1];
                vtile[l * ts + o.off_ln + i] *= invN;
                ftile[l * ts + o.off_ln + i] *= invN;
            }
        }
        __syncthreads();
        for (int l = 0; l < nl; ++l) {
            const GfLayerDev<T> o = a.L[l];
            for (int i = tid; i < o.K * D; i += NT) {
                T* c = pk + (a.pk0[l] + i) * 8;
                c[0] = ptile[l * ts + o.off_mean + i];
                c[1] = vtile[l * ts + o.off_lw + i];
                c[2] = o.fit_norm ? vtile[l * ts + o.off_ln + i] : T(1) / T(o.K);
                c[3] = ftile[l * ts + o.off_lw + i];
                c[4] = o.fit_norm ? ftile[l * ts + o.off_ln + i] : T(0);
            }
        }
        __syncthreads();
    }
    const int tiles = BCAST ? a.tiles_per_block : 1;
    for (int t = 0; t < tiles; ++t) {
        const int64_t row0 = (BCAST ? (int64_t)blockIdx.x + (int64_t)t * a.active_blocks : (int64_t)blockIdx.x) * R;
        if (row0 >= a.B) break;
        const int64_t row = row0 + r;
        const bool row_valid = row < a.B;
        const int64_t rrow = row_valid ? row : a.B - 1;
        const int valid_rows = (int)((a.B - row0) < R ? (a.B - row0) : R);

        // ---- forward sweep (layers n-1 .. 0), keeping every layer's input
        T x = a.x[rrow * a.xs + d];
        T msC[JF_MAX_CHAIN], msS[JF_MAX_CHAIN], msP[JF_MAX_CHAIN], msN[JF_MAX_CHAIN];   // per-sample regime: the forward sweep's mixture sums (fast layers)
        T xreg[JF_MAX_CHAIN];                                    // per-sample regime: layer inputs in registers, both sweeps unrolled (measured faster
        if constexpr (BCAST) {                                   //   there: 1.58 vs 1.96 ms for the float64 D = 8 chain of C5)
#pragma unroll 1
            for (int l = nl - 1; l >= 0; --l) {
                xin[l * NT] = x;
                const GfLayerDev<T> o = a.L[l];
                const T* p = vtile + l * ts + d;
                if (o.model_offset) x -= p[0];
                x = gfg_rotate_inv<T, G, false>(p, o, D, live, x);
                T Cs, Ss, Ps;
                x = gf_icdf<T>(o.inv_type, gfb_mixture_pk<T>(pk + (a.pk0[l] + d) * 8, p, o, D, x, Cs, Ss, Ps)).y;
                xin[(nl + l) * NT] = Cs; xin[(2 * nl + l) * NT] = Ss; xin[(3 * nl + l) * NT] = Ps;
            }
        } else {
#pragma unroll
            for (int li = 0; li < JF_MAX_CHAIN; ++li) {
                xreg[li] = x;
                if (li < nl) {
                    const GfLayerDev<T> o = a.L[nl - 1 - li];
                    __syncthreads();
                    stage_rows<T>(ptile, ts, a.params + row0 * a.ps + o.col0, a.ps, o.n_params, R, valid_rows, tid, NT, o.vec_ok != 0);
                    __syncthreads();
                    const T* p = ptile + r * ts + d;
                    if (o.model_offset) x -= p[0];
                    x = gfg_rotate_inv<T, G, true>(p, o, D, live, x);
                    if (o.fast) {
                        MixSums<T> m;
                        x = gf_icdf<T>(o.inv_type, gfb_mixture_fast_raw<T>(p, o, D, x, m)).y;
                        msC[li] = m.C; msS[li] = m.S; msP[li] = m.P; msN[li] = m.invN;
                    } else x = gf_icdf<T>(o.inv_type, gfg_mixture<T, true>(p, o, D, x)).y;
                }
            }
        }
        // ---- upstream gradients: base log-prob = sum_d -x^2/2 - ...  =>  d/dx_out = -x_out
        const T gl = (a.g_ld && row_valid) ? a.g_ld[rrow] : T(0);
        T gy = (a.g_xout && row_valid) ? a.g_xout[rrow * a.gxos + d] : T(0);
        if (a.g_blp && row_valid) gy -= x * a.g_blp[rrow];
        if (!live || !row_valid) gy = T(0);
        const T glr = row_valid ? gl : T(0);

        // ---- backward sweep (layers 0 .. n-1)
        if constexpr (BCAST) {
#pragma unroll 1
            for (int l = 0; l < nl; ++l) {
                const GfLayerDev<T> o = a.L[l];
                const int c0 = l * ts + d;
                gy = gf_layer_bwd_bcast<T, G>(ptile + c0, vtile + c0, pk + (a.pk0[l] + d) * 8, acc + (c0 << slsh) + (r & ((1 << slsh) - 1)), o, D, live && row_valid,
                                              xin[l * NT], gy, glr, slsh, xin[(nl + l) * NT], xin[(2 * nl + l) * NT], xin[(3 * nl + l) * NT]);
            }
        } else {
#pragma unroll
            for (int li = JF_MAX_CHAIN - 1; li >= 0; --li) {
                if (li < nl) {
                    const GfLayerDev<T> o = a.L[nl - 1 - li];
                    const T xi = xreg[li];
                    __syncthreads();
                    stage_rows<T>(ptile, ts, a.params + row0 * a.ps + o.col0, a.ps, o.n_params, R, valid_rows, tid, NT, o.vec_ok != 0);
                    __syncthreads();
                    T* gp = DIRECT ? a.g_params + rrow * a.gps + o.col0 + d : gtile + r * ts + d;
                    const bool lw = DIRECT ? (live && row_valid) : live;       // DIRECT: rows past B must not store
                    if (o.fast) gy = gf_layer_bwd_fast<T, G>(ptile + r * ts + d, gp, o, D, lw, xi, gy, glr, MixSums<T>{msC[li], msS[li], msP[li], msN[li]});
                    else gy = gf_layer_bwd<T, G, false, T>(ptile + r * ts + d, gp, o, D, lw, xi, gy, glr, 0);
                    if constexpr (!DIRECT) {
                        __syncthreads();
                        // gradient tile -> HBM, row by row (consecutive lanes = consecutive columns)
                        for (int rr2 = 0; rr2 < valid_rows; ++rr2)
                            for (int j = tid; j < o.n_params; j += NT)
                                a.g_params[(row0 + rr2) * a.gps + o.col0 + j] = gtile[rr2 * ts + j];
                    }
                }
            }
        }
        if (row_valid && live) a.g_x[row * a.gxs + d] = gy;
        const T bad = group_max<T, G>((live && !M<T>::finite(gy)) ? T(1) : T(0));
        status_add(a.status, JF_STATUS_NONFINITE, row_valid && leader && bad > T(0));
    }
    if constexpr (BCAST) {                                       // this workgroup's partial row: the slots of every parameter summed (rotated start: the
        __syncthreads();                                         //   threads of a wave read different banks)
        const int SL = 1 << slsh;
        for (int l = 0; l < nl; ++l) {
            const GfLayerDev<T> o = a.L[l];
            for (int j = tid; j < o.n_params; j += NT) {
                const double* base = acc + ((l * ts + j) << slsh);
                double sum = 0.0;
                for (int s2 = 0; s2 < SL; ++s2) sum += base[(s2 + j) & (SL - 1)];
                a.g_params[(int64_t)blockIdx.x * a.gps + o.col0 + j] = (T)sum;
            }
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
    Du* tab = reinterpret_cast<Du*>(red + 16) + tid * a.spline_tab;      // lane-private knot table (chains with a spline stretch)
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
                if (o.stretch == JF_GF_STRETCH_RQ_SPLINES) {       // gaussianization_flow.py:863-909 (the log-prob direction evaluates the spline itself)
                    for (int d = 0; d < D; ++d) {
                        const SplineOut<Du> r = spline_linext<Du, SeededRow<T>>(p + (o.off_mean + d * o.K), p + (o.off_lw + d * o.K), p + (o.off_ln + d * (o.K + 1)),
                                                                                p + (o.off_box + d * 4), o.K, tab, x[d], false);
                        x[d] = r.y;
                        ld = ld + r.lad;
                    }
                    continue;
                }
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
static inline int gb_group_width(int D) { return D <= 1 ? 1 : D <= 2 ? 2 : D <= 4 ? 4 : D <= 8 ? 8 : D <= 16 ? 16 : D <= 32 ? 32 : 64; }

template <typename T> static int gb_fill(GfBwdArgs<T>& a, const T* params, int64_t ps, bool bcast, int32_t D, int32_t n_layers, const jf_gf_layer* layers,
                                         bool& ext) {
    int col = 0, maxp = 0;
    ext = false;
    for (int l = 0; l < n_layers; ++l) {
        const jf_gf_layer& h = layers[l];
        GfLayerDev<T>& o = a.L[l];
        if (h.num_kde < 1 || h.num_kde > (1 << 16) || h.hh_iter < 0 || h.hh_iter > (1 << 16) || h.width_min <= 0) return JF_ERR_BADARG;   // (bounded: the column offsets below are ints)
        const bool ext_layer = h.rotation_mode != JF_GF_ROT_HOUSEHOLDER || h.center_mean || h.add_skewness;     // general-option layer (jf_gf_ext.h)
        if (h.rotation_mode < JF_GF_ROT_HOUSEHOLDER || h.rotation_mode > JF_GF_ROT_TRIANGULAR || (h.rotation_mode == JF_GF_ROT_CAYLEY && D > 2)) return JF_ERR_BADARG;
        if (h.add_skewness && sizeof(T) != 8) return JF_ERR_UNSUPPORTED;                                        // float64 only, as the forward
        if (ext_layer) ext = true;
        if (ext_layer && D > JF_MAX_D_GF) return JF_ERR_UNSUPPORTED;
        if (h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION && h.width_max <= 0) return JF_ERR_BADARG;
        if (h.nonlinear_stretch_type != JF_GF_STRETCH_CLASSIC && h.nonlinear_stretch_type != JF_GF_STRETCH_RQ_SPLINES) return JF_ERR_BADARG;
        const bool rq = h.nonlinear_stretch_type == JF_GF_STRETCH_RQ_SPLINES;      // spline stretch: the general-option kernel (one lane per row, lane-private knot table)
        if (rq && (h.center_mean || h.add_skewness)) return JF_ERR_BADARG;
        if (rq && (h.num_kde > JF_SPLINE_CAP || D > JF_MAX_D_GF)) return JF_ERR_UNSUPPORTED;
        if (rq) { ext = true; if (spline_tab_words(h.num_kde) > a.spline_tab) a.spline_tab = spline_tab_words(h.num_kde); }
        o.K = h.num_kde; o.hh = h.hh_iter; o.model_offset = h.model_offset; o.fit_norm = h.fit_normalization;
        o.reg_norm = h.regulate_normalization; o.inv_type = h.inverse_function_type; o.width_mode = h.width_mode;
        o.clamp_widths = h.clamp_widths;
        o.fast = (h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION && !h.clamp_widths && h.fit_normalization && h.regulate_normalization) ? 1 : 0;
        o.stretch = h.nonlinear_stretch_type; o.off_box = 0;
        const int kd = h.num_kde * D;
        o.rot_mode = h.rotation_mode; o.center_mean = h.center_mean ? 1 : 0; o.skew = h.add_skewness ? 1 : 0;
        if (o.rot_mode != JF_GF_ROT_HOUSEHOLDER) o.hh = 0;
        o.off_rot = h.model_offset ? D : 0;
        o.off_mean = o.off_rot + gx_rot_len(o.rot_mode, o.hh, D);
        o.off_lw = o.off_mean + kd - (o.center_mean ? D : 0);
        o.off_ln = o.off_lw + kd;
        o.off_skew = o.off_ln + (h.fit_normalization ? kd : 0);
        o.n_params = o.off_skew + (o.skew ? kd : 0);
        if (rq) {                                                  // row layout of a spline layer as in gf_kernels.hip: widths, heights (K D each), derivatives ((K + 1) D), box (4 D)
            o.off_skew = 0;
            o.off_box = o.off_ln + (h.num_kde + 1) * D;
            o.n_params = o.off_box + 4 * D;
        }
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
    const int64_t n_tiles = (B + GB_NT / G - 1) / (GB_NT / G);
    const int64_t blocks = n_tiles < 1024 ? (n_tiles < 1 ? 1 : n_tiles) : 1024;
    return blocks;
}

template <typename T, int G>
static int gb_launch(GfBwdArgs<T> a, bool bcast, hipStream_t st) {
    if (bcast) {
        const int64_t n_tiles = (a.B + GB_NT / G - 1) / (GB_NT / G);
        const int64_t blocks = gb_partials(a.B, a.D);
        a.tiles_per_block = (int)((n_tiles + blocks - 1) / blocks);
        int slsh = G == 1 ? 6 : G == 2 ? 5 : G == 4 ? 4 : G == 8 ? 3 : G == 16 ? 2 : G == 32 ? 1 : 0;       // one slot per row of a wave, fewer when the accumulators would not fit
        const size_t cell = (size_t)a.n_layers * a.tile_stride * sizeof(T), acell = (size_t)a.n_layers * a.tile_stride * sizeof(double);
        while (slsh > 0 && 3 * cell + (acell << slsh) > 28 * 1024) --slsh;   // measured flat between 2 and 8 slots (conflicting ds_add_f64 are cheap); occupancy matters more
        a.slsh = slsh;
        int n_rec = 0;
        for (int l = 0; l < a.n_layers; ++l) { a.pk0[l] = n_rec; n_rec += a.L[l].K * a.D; }
        const size_t lds = 3 * cell + (acell << slsh) + ((size_t)a.n_layers * (4 * GB_NT + 2 * (G > 8 ? G : 8)) + (size_t)n_rec * 8) * sizeof(T);
        if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;
        auto k = gf_chain_bwd_kernel<T, G, true>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        // one resident round of workgroups takes all tiles (grid stride): each pays the prologue / epilogue once, no tail round
        int dev = 0, cus = 256, occ = 1;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k, GB_NT, lds) != hipSuccess || occ < 1) occ = 1;
        int64_t active = (int64_t)cus * occ;
        if (active > blocks) active = blocks;
        if (active > n_tiles) active = n_tiles < 1 ? 1 : n_tiles;
        a.active_blocks = (int)active;
        a.tiles_per_block = (int)((n_tiles + active - 1) / active);
        jf::launch(k, dim3((unsigned)blocks), dim3(GB_NT), lds, st, a);
    } else {
        a.tiles_per_block = 1;
        const bool direct = (size_t)a.D * sizeof(T) >= 32;
        const size_t lds = (size_t)(direct ? 1 : 2) * (64 / G) * a.tile_stride * sizeof(T);
        auto k = direct ? gf_chain_bwd_kernel<T, G, false, true> : gf_chain_bwd_kernel<T, G, false, false>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        jf::launch(k, dim3((unsigned)((a.B + 64 / G - 1) / (64 / G))), dim3(64), lds, st, a);
    }
    return check_launch();
}

template <typename T>
static int gf_chain_inv_bwd(const T* x, int64_t xs, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n_layers,
                            const jf_gf_layer* layers, const T* g_xout, int64_t gxos, const T* g_ld, const T* g_blp, T* g_x, int64_t gxs, T* g_params,
                            int64_t gps, int32_t* status, void* stream) {
    if (!x || !params || !g_x || !g_params || !layers) return JF_ERR_BADARG;
    if (n_layers < 1 || n_layers > JF_MAX_CHAIN || D < 1 || B < 0) return JF_ERR_BADARG;
    if (D > JF_MAX_D_G) return JF_ERR_UNSUPPORTED;
    if (pb != 1 && pb != B) return JF_ERR_BADARG;
    const bool bcast = pb == 1;
    GfBwdArgs<T> a{};
    bool ext = false;
    const int rc = gb_fill<T>(a, params, ps, bcast, D, n_layers, layers, ext);
    if (rc != JF_OK) return rc;
    if (B == 0) return JF_OK;
    a.x = x; a.xs = xs; a.params = params; a.ps = ps; a.B = B; a.D = D; a.n_layers = n_layers;
    a.g_xout = g_xout; a.gxos = gxos; a.g_ld = g_ld; a.g_blp = g_blp; a.g_x = g_x; a.gxs = gxs; a.g_params = g_params; a.gps = gps; a.status = status;
    if (ext) {                                                     // general-option chains; broadcast: one partial row per workgroup, as the adjoint kernel
        const int64_t tiles = (B + GX_THREADS - 1) / GX_THREADS;
        const int64_t blocks = bcast ? gb_partials(B, D) : tiles;
        // reverse sweep (gf_rev_kernels.hip); JF_G_BWD_DUAL=1: the dual-number replay below, its check
        static const int dual_replay = getenv("JF_G_BWD_DUAL") ? atoi(getenv("JF_G_BWD_DUAL")) : 0;
        if (dual_replay <= 0) return gfx_chain_rev_launch<T>(a, bcast, ps, blocks, tiles, stream);
        const size_t lds = (size_t)JF_MAX_D_GF * GX_THREADS * sizeof(Dual<T>) + 16 * sizeof(T) + (size_t)GX_THREADS * a.spline_tab * sizeof(Dual<T>);
        if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;
        auto k = gfx_chain_bwd_kernel<T>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        jf::launch(k, dim3((unsigned)blocks), dim3(GX_THREADS), lds, (hipStream_t)stream, a, bcast ? (int64_t)0 : ps, tiles);
        return check_launch();
    }
    for (int l = 0; l < n_layers; ++l) if (layers[l].hh_iter > (gb_group_width(D) > GB_MAX_HH ? gb_group_width(D) : GB_MAX_HH)) return JF_ERR_UNSUPPORTED;
    switch (gb_group_width(D)) {
        case 1: return gb_launch<T, 1>(a, bcast, (hipStream_t)stream);
        case 2: return gb_launch<T, 2>(a, bcast, (hipStream_t)stream);
        case 4: return gb_launch<T, 4>(a, bcast, (hipStream_t)stream);
        case 8: return gb_launch<T, 8>(a, bcast, (hipStream_t)stream);
        case 16: return gb_launch<T, 16>(a, bcast, (hipStream_t)stream);
        case 32: return gb_launch<T, 32>(a, bcast, (hipStream_t)stream);
        default: return gb_launch<T, 64>(a, bcast, (hipStream_t)stream);
    }
}

// LDS bytes a backward launch of this chain needs (smallest accumulator-slot count), or a negative JF_ERR_*: what the host uses to cut a long
// chain of wide layers into launches that fit the 160 KB of a CU
template <typename T> static int64_t gb_lds_query(int32_t D, int32_t n_layers, const jf_gf_layer* layers, int bcast) {
    if (!layers || n_layers < 1 || n_layers > JF_MAX_CHAIN || D < 1) return JF_ERR_BADARG;
    if (D > JF_MAX_D_G) return JF_ERR_UNSUPPORTED;
    GfBwdArgs<T> a{};
    bool ext = false;
    const int rc = gb_fill<T>(a, nullptr, 0, bcast != 0, D, n_layers, layers, ext);
    if (rc != JF_OK) return rc;
    if (ext) {
        a.D = D; a.n_layers = n_layers;
        const int64_t dual = (int64_t)((size_t)JF_MAX_D_GF * GX_THREADS * sizeof(Dual<T>) + 16 * sizeof(T) + (size_t)GX_THREADS * a.spline_tab * sizeof(Dual<T>));
        const int64_t rev = gfx_chain_rev_lds_bytes<T>(a, bcast != 0);
        return rev > dual ? rev : dual;                            // (either kernel may take the launch: JF_G_BWD_DUAL)
    }
    const int G = gb_group_width(D);
    for (int l = 0; l < n_layers; ++l) if (layers[l].hh_iter > (G > GB_MAX_HH ? G : GB_MAX_HH)) return JF_ERR_UNSUPPORTED;
    if (!bcast) return (int64_t)((size_t)(((size_t)D * sizeof(T) >= 32) ? 1 : 2) * (64 / G) * a.tile_stride * sizeof(T));
    const size_t cell = (size_t)n_layers * a.tile_stride * sizeof(T), acell = (size_t)n_layers * a.tile_stride * sizeof(double);
    size_t n_rec = 0;
    for (int l = 0; l < n_layers; ++l) n_rec += (size_t)a.L[l].K * D;
    return (int64_t)(3 * cell + acell + ((size_t)n_layers * (4 * GB_NT + 2 * (G > 8 ? G : 8)) + n_rec * 8) * sizeof(T));
}

}  // namespace jf

extern "C" {
int64_t jf_gf_chain_inv_bwd_partials(int64_t B, int32_t D) { return jf::gb_partials(B, D); }
int64_t jf_gf_chain_inv_bwd_lds_bytes_f32(int32_t D, int32_t n, const jf_gf_layer* L, int32_t pb1) { return jf::gb_lds_query<float>(D, n, L, pb1); }
int64_t jf_gf_chain_inv_bwd_lds_bytes_f64(int32_t D, int32_t n, const jf_gf_layer* L, int32_t pb1) { return jf::gb_lds_query<double>(D, n, L, pb1); }
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
