// Pieces of the 'g' layer's adjoint shared by gf_bwd_kernels.hip (parameter rows in LDS) and cond_bwd_kernels.hip (parameter rows in MFMA result
// registers): the inverse-CDF stage's log-space coefficients, the validity range of the linear-space responsibilities, the mixture sums the
// forward sweep hands to the backward sweep.
#pragma once
#include "jf_gf.h"

namespace jf {

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

template <typename T> struct LinRange;
template <> struct LinRange<float> { static constexpr float lo = 1e-35f, hi = 1e30f, llo = -80.0f, lhi = 69.0f; };      // lo = M<T>::TINY: the forward's switch
template <> struct LinRange<double> { static constexpr double lo = 1e-280, hi = 1e280, llo = -644.0, lhi = 644.0; };


template <typename T> struct MixSums { T C, S, P, invN; };

// arguments of the backward kernels of a 'g' chain (gf_bwd_kernels.hip; the general-option reverse sweep gf_rev_kernels.hip)
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
    int active_blocks;                        // broadcast regime: workgroups that take tiles (one resident round); the others write a zero row
    int pk0[JF_MAX_CHAIN];                    // broadcast regime: first packed component record of every layer (gf_chain_bwd_kernel)
    int slsh;                                 // broadcast regime: log2 of the accumulator slots per parameter (see gf_chain_bwd_kernel)
    int spline_tab;                           // general-option kernel: words of a lane's knot table (0: no spline stretch in the chain)
};

// general-option chains (jf_gf_ext.h), reverse mode (gf_rev_kernels.hip): launch and the LDS bytes it takes
template <typename T> int gfx_chain_rev_launch(GfBwdArgs<T> a, bool bcast, int64_t ps, int64_t blocks, int64_t tiles, void* stream);
template <typename T> int64_t gfx_chain_rev_lds_bytes(const GfBwdArgs<T>& a, bool bcast);

}  // namespace jf
