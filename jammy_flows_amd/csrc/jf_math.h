// Scalar math policy for the flow kernels.
//
//  float : hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, quarter-rate on CDNA4) on the hot
//          linear-space path -- the fp32 bar is |dlogp| < 1e-2 against the fp64 reference, these are ~1e-6;
//          accurate OCML functions on the rare log-space (tail) path and in the iterative solvers.
//  double: OCML double functions, except log_fast, rcp and tanh_fast (below);
//          fp64 bar: |dlogp| < 1e-4, measured ~2e-8.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "jf_tanh_table.h"

namespace jf {

template <typename T> struct M;

// scalar types that carry tangents (jf_dual.h specialises this): the iterative solvers of the layer code run on the VALUES and give the
// solution's tangents by the implicit-function theorem instead of dragging tangents through every iteration
template <typename T> struct DualTraits { static constexpr bool is_dual = false; using value_type = T; };
template <typename D> struct DualValues;           // a row of dual numbers read as its values (jf_dual.h)



template <> struct M<float> {
    static constexpr float PI = 3.14159265358979323846f;
    static constexpr float TWO_PI = 6.28318530717958647692f;
    static constexpr float HALF_LN_2PI = 0.91893853320467274178f;
    static constexpr float SQRT2 = 1.41421356237309504880f;
    static constexpr float TINY = 1e-35f;       // below this a linear-space cdf / sf / pdf is re-evaluated with scaled sums
    static constexpr float EPS_COS = 1e-7f;     // sphere_base.return_safe_costheta float32 margin
    static constexpr float EPS_S1 = 1e-5f;      // sphere_base.sphere_to_plane float32 clamp
    static constexpr float KAPPA_ID = 1e-4f;    // fvm_2d small-kappa identity switch (float32)
    // (JF_PROBE_ACCURATE_*: probe builds that put the correctly rounded library function in place of ONE hardware approximation -- the float32 error
    //  budget of DESIGN section 4, scripts/probe/f32_error_budget.sh; never defined in the shipped library)
#ifdef JF_PROBE_ACCURATE_EXP
    static __device__ __forceinline__ float exp_fast(float x) { return expf(x); }
#else
    static __device__ __forceinline__ float exp_fast(float x) { return __expf(x); }
#endif
#ifdef JF_PROBE_ACCURATE_LOG
    static __device__ __forceinline__ float log_fast(float x) { return logf(x); }
#else
    static __device__ __forceinline__ float log_fast(float x) { return 0.69314718056f * __builtin_amdgcn_logf(x); }   // v_log_f32 (normal-range inputs only)
#endif
#ifdef JF_PROBE_ACCURATE_RCP
    static __device__ __forceinline__ float sqrt_fast(float x) { return sqrtf(x); }
    static __device__ __forceinline__ float rcp(float x) { return __fdiv_rn(1.0f, x); }
#else
    static __device__ __forceinline__ float sqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif
    static __device__ __forceinline__ float exp(float x) { return expf(x); }
    static __device__ __forceinline__ float log(float x) { return logf(x); }
    static __device__ __forceinline__ float log1p(float x) { return log1pf(x); }
    static __device__ __forceinline__ float expm1(float x) { return expm1f(x); }
    static __device__ __forceinline__ float sqrt(float x) { return sqrtf(x); }
    static __device__ __forceinline__ float erf(float x) { return erff(x); }
    static __device__ __forceinline__ float erfinv(float x) { return erfinvf(x); }
    static __device__ __forceinline__ float erfcinv(float x) { return erfcinvf(x); }
    static __device__ __forceinline__ float sin(float x) { return sinf(x); }
    static __device__ __forceinline__ float cos(float x) { return cosf(x); }
    static __device__ __forceinline__ float acos(float x) { return acosf(x); }
    static __device__ __forceinline__ float atan2(float y, float x) { return atan2f(y, x); }
    static __device__ __forceinline__ float tanh(float x) { return tanhf(x); }
    // 1 - 2/(e^{2x}+1) on v_exp_f32 / v_rcp_f32: absolute error ~1e-7 (relative accuracy is lost only where |tanh| < 1e-3), saturates cleanly
    // (v_rcp_f32 directly: __frcp_rn expands to the 10-instruction correctly-rounded division sequence, 32 times per lane and row tile)
    static __device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }
    static __device__ __forceinline__ float abs(float x) { return fabsf(x); }
    static __device__ __forceinline__ float max(float a, float b) { return fmaxf(a, b); }
    static __device__ __forceinline__ float min(float a, float b) { return fminf(a, b); }
    static __device__ __forceinline__ bool finite(float x) { return isfinite(x); }
};

template <> struct M<double> {
    static constexpr double PI = 3.14159265358979323846;
    static constexpr double TWO_PI = 6.28318530717958647692;
    static constexpr double HALF_LN_2PI = 0.91893853320467274178;
    static constexpr double SQRT2 = 1.41421356237309504880;
    static constexpr double TINY = 1e-280;
    static constexpr double EPS_COS = 1e-10;
    static constexpr double EPS_S1 = 1e-8;
    static constexpr double KAPPA_ID = 1e-8;
    // (round 4 tried a bar-aware replacement -- Cody-Waite reduction + degree-9 polynomial + v_ldexp_f64, 1.8e-14 relative, ~20 instructions
    // against OCML's 42, no table: scripts/probe/exp_poly.py -- together with a one-Newton-step reciprocal.  The float64 kernels did not move
    // (per-sample g chain 1.335 -> 1.36 ms, broadcast 0.545 -> 0.56, C5 block 1.25 -> 1.21 per 2^19 rows): both exponentials carry the same
    // ~12 float64 FMAs, which issue at half rate on MI355X, and those -- not the instruction count -- are the cycles.  OCML's stays: < 1 ulp.)
    static __device__ __forceinline__ double exp_fast(double x) { return ::exp(x); }
    // Natural logarithm for the mixture sums (three per coordinate and layer; OCML's log is 98 VALU instructions, this one ~40): the classic
    // reduction x = 2^e m, m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f), log m = f - f^2/2 + s (f^2/2 + R(s^2)) with the degree-7 even
    // minimax polynomial R of FreeBSD msun's e_log.c (coefficients Lg1..Lg7), < 1 ulp.  The division is rcp() above.
    // The coefficients come from e_log.c, which carries this notice (kept here as it asks):
    //   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.  Developed at SunPro, a Sun Microsystems, Inc. business.
    //   Permission to use, copy, modify, and distribute this software is freely granted, provided that this notice is preserved.
    // Exact special values: log(0) = -inf, log(inf) = inf, log(x < 0) = log(nan) = nan; denormals go through v_frexp like everything else.
    static __device__ __forceinline__ double log_fast(double x) {
        int e = __builtin_amdgcn_frexp_exp(x);                         // x = m 2^e, m in [1/2, 1)
        double m = __builtin_amdgcn_frexp_mant(x);
        const bool low = m < 0.70710678118654752440;
        m = low ? m + m : m;
        e = low ? e - 1 : e;
        const double f = m - 1.0;
        const double s = f * rcp(2.0 + f);
        const double z = s * s, w = z * z;
        const double t1 = w * ::fma(w, ::fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
        const double t2 = z * ::fma(w, ::fma(w, ::fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                                    6.666666666666735130e-01);
        const double R = t1 + t2, hfsq = 0.5 * f * f, de = (double)e;
        double r = ::fma(de, 6.93147180369123816490e-01, -((hfsq - ::fma(s, hfsq + R, de * 1.90821492927058770002e-10)) - f));
        r = x == 0.0 ? -INFINITY : r;
        return (x > 0.0 && x < INFINITY) || x == 0.0 ? r : (x < 0.0 ? NAN : x);
    }
    static __device__ __forceinline__ double sqrt_fast(double x) { return ::sqrt(x); }
    // v_rcp_f64 (~2^-26) + two Newton steps: 1 ulp, 7 instructions where the IEEE division sequence has ~25 (the float64 mixtures take three
    // reciprocals per component).  rcp(inf) = 0 and rcp(0) = inf survive: a non-finite refinement falls back to the hardware value.
    static __device__ __forceinline__ double rcp(double x) {
        const double r0 = __builtin_amdgcn_rcp(x);
        double r = ::fma(::fma(-x, r0, 1.0), r0, r0);
        r = ::fma(::fma(-x, r, 1.0), r, r);
        return isfinite(r) ? r : r0;
    }
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double log1p(double x) { return ::log1p(x); }
    static __device__ __forceinline__ double expm1(double x) { return ::expm1(x); }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double erf(double x) { return ::erf(x); }
    static __device__ __forceinline__ double erfinv(double x) { return ::erfinv(x); }
    static __device__ __forceinline__ double erfcinv(double x) { return ::erfcinv(x); }
    static __device__ __forceinline__ double sin(double x) { return ::sin(x); }
    static __device__ __forceinline__ double cos(double x) { return ::cos(x); }
    static __device__ __forceinline__ double acos(double x) { return ::acos(x); }
    static __device__ __forceinline__ double atan2(double y, double x) { return ::atan2(y, x); }
    static __device__ __forceinline__ double tanh(double x) { return ::tanh(x); }
    // tanh of a hidden layer: (1 - t) / (1 + t), t = e^{-2 |x|}, to an ABSOLUTE error of ~3e-16.  Every use feeds a linear layer (sum_k w_k h_k),
    // where the absolute error of h is what counts; the earlier expm1(2x) / (expm1(2x) + 2) kept full RELATIVE accuracy for tiny arguments at
    // twice the instructions (expm1 + an IEEE division: ~120 against ~55) -- the float64 MLP kernels are bound by exactly this function
    // (jf_mlp2_f64 of the 4 -> 128 -> 10 head: 0.37 ms per 2^20 rows).  tanh(+-inf) = +-1, tanh(nan) = nan.
    static __device__ __forceinline__ double tanh_fast(double x) {
        const double t = ::exp(-2.0 * ::fabs(x));
        return ::copysign((1.0 - t) * rcp(1.0 + t), x);
    }
    static __device__ __forceinline__ double abs(double x) { return ::fabs(x); }
    static __device__ __forceinline__ double max(double a, double b) { return ::fmax(a, b); }
    static __device__ __forceinline__ double min(double a, double b) { return ::fmin(a, b); }
    static __device__ __forceinline__ bool finite(double x) { return isfinite(x); }
};

// tanh of a float64 hidden layer from a TABLE in LDS.  M<double>::tanh_fast is an exponential (OCML: ~42 instructions) and a reciprocal: ~55 float64
// instructions per hidden unit, 128 units per row -- the float64 MLP kernels are bound by exactly this (jf_mlp2_f64 of the 4 -> 128 -> 10 head:
// 0.37 ms per 2^20 rows for 0.05 ms of matrix arithmetic).  The units of a row are independent, so a table pays here (it lost for the mixture's
// exponentials, a dependent chain): |x| = a + r with a = k / 32 on a 2^-5 grid (k <= 608: tanh(19) rounds to 1 - 2^-53) and |r| <= 2^-6,
//   tanh(a + r) = (T_a + p) / (1 + T_a p),  T_a = tanh(a) from the table (609 doubles, 4.9 KB),  p = tanh r = r - r^3/3 + 2 r^5/15 - 17 r^7/315
// (next term 62 r^9 / 2835 < 1e-18), the quotient by v_rcp_f64 + one Newton step + one residual correction (the denominator lies in [1, 2]).
// ~23 instructions; absolute error <= 3e-16 and relative error <= 5e-16 against a 40-digit tanh over [-20, 20] and around 0 (numpy emulation
// of exactly these operations, scripts/probe/tanh_tab_check.py) -- the accuracy of tanh_fast.  tanh(+-inf) = +-1, tanh(nan) = nan.
__device__ __forceinline__ void tanh_tab_load(double* __restrict__ lds_tab, int tid, int nthreads) {
    for (int i = tid; i < JF_TANH_TAB_N; i += nthreads) lds_tab[i] = JF_TANH_TAB[i];
}
__device__ __forceinline__ double tanh_tab(const double* __restrict__ lds_tab, double x) {
#ifdef JF_PROBE_NO_TANH_TAB                                              // A/B builds only (scripts/probe/tanh_ab.sh)
    return M<double>::tanh_fast(x);
#endif
    const double ax = ::fmin(::fabs(x), 19.0);
    const double k = ::rint(ax * 32.0);
    const double r = ::fma(k, -0.03125, ax);                             // exact
    const double T = lds_tab[(int)k];
    const double r2 = r * r;
    const double p = r * ::fma(r2, ::fma(r2, ::fma(r2, -17.0 / 315.0, 2.0 / 15.0), -1.0 / 3.0), 1.0);
    const double n = T + p, d = ::fma(T, p, 1.0);
    const double r0 = __builtin_amdgcn_rcp(d);
    const double r1 = ::fma(::fma(-d, r0, 1.0), r0, r0);
    double q = n * r1;
    q = ::fma(::fma(-d, q, n), r1, q);
    return x != x ? x : ::copysign(q, x);
}
// (float: the hardware exponential is cheaper than any table)
template <typename T> __device__ __forceinline__ T tanh_hidden(const double* lds_tab, T x) {
    if constexpr (sizeof(T) == 8) return tanh_tab(lds_tab, x);
    else return M<T>::tanh_fast(x);
}

// softplus(x) = log(1 + e^x), overflow-free (torch F.softplus agrees to < 2.1e-9 with its threshold=20 shortcut)
template <typename T> __device__ __forceinline__ T softplus(T x) {
    return M<T>::max(x, T(0)) + M<T>::log1p(M<T>::exp(-M<T>::abs(x)));
}
template <typename T> __device__ __forceinline__ T logaddexp(T a, T b) {
    const T m = M<T>::max(a, b);
    return m + M<T>::log1p(M<T>::exp(-M<T>::abs(a - b)));
}
template <typename T> __device__ __forceinline__ T clampv(T x, T lo, T hi) { return M<T>::min(M<T>::max(x, lo), hi); }

// Stopping rule of the Newton stages (sampling direction of 'g', the solve of 'm').  The reference ends a row when the sum of its |updates| falls
// below 1e-14 (bisection_n_newton.py:68-115, called with newton_tolerance 1e-14): in float64 that costs one evaluation that only CONFIRMS -- near the
// simple root of a smooth increasing function the update after one of size u is (f'' / 2 f') u^2, so a row whose update fell below 1e-9 already
// sits within 1e-16 x (curvature ratio, <= ~1e2 for widths >= 0.01) of the point at which the reference's iteration ends, and applying that update is
// the last thing the reference itself does before its confirming step.  float64 rows therefore stop at 1e-9 (one float64 mixture evaluation per
// coordinate and layer less: C5 sampling 3.2 -> see DESIGN); float32 rows keep their own rounding-floor rule below.
template <typename T> struct NewtonTol { static constexpr double value = sizeof(T) == 8 ? 1e-9 : 1e-14; };
// The audit switch.  -DJF_NEWTON_RULE_REFERENCE builds the library whose solvers follow the reference's own iteration: 25 bisections on
// [-1e5, 1e5], then Newton until the row's update sum falls below 1e-14 or 20 steps are done (bisection_n_newton.py:11-135), no float32 floor, the
// sphere Newton of 'v' until 1e-12 (:330-465).  csrc/Makefile builds it as libjammy_hip_audit.so beside the product library; the Python package
// loads it when JF_NEWTON_RULE=reference is set (jammy_flows_amd/_hip.py), jf_get_newton_rule() says which one is loaded.  A compile-time
// constant: the product kernels carry no trace of it.  (A run-time flag in a __device__ word per translation unit was built first: kernels of
// gf_kernels.hip then hung under rocprofv3 --kernel-trace -- never without the profiler; a constant cannot do that.)
#ifdef JF_NEWTON_RULE_REFERENCE
__device__ __forceinline__ constexpr bool newton_reference_rule() { return true; }
#else
__device__ __forceinline__ constexpr bool newton_reference_rule() { return false; }
#endif
// ... unless the audit switch asks for the reference's own rule (jf_common.h: newton_reference_rule)
template <typename T> __device__ __forceinline__ T newton_tol() { return newton_reference_rule() ? T(1e-14) : T(NewtonTol<T>::value); }
// ... and the float32 rows' floor (the reference's absolute 1e-14 never fires in float32: its float32 runs do all 20 steps on rounding noise).  A row
// stops when the sum of its |updates| is below this fraction of the sum of max(|x|, 1) over its coordinates.  Rounds 1-4 used the resolution of the
// coordinates (2.5e-7), which -- like 1e-14 in float64 -- made most waves spend a second evaluation confirming a first update of 1e-6 .. 1e-5: the
// update after one of relative size 1e-5 is 1e-10 x (curvature ratio) and the stage's log-derivative read at the last evaluated point moves by
// 1e-5 x its slope, both below float32 resolution of the results.  A row only stops this way when the residual of the evaluation just made is
// inside the reference's float32 convergence threshold (1e-4) already -- on steep stretches (narrow components) a 1e-5 update still is a residual
// of 1e-3, and those rows take their confirming evaluation as before, so the "did not converge" count means what it meant.  Measured: C3 sampling 1.37 -> 1.19 ms per 2^20 rows, C2 0.258 -> 0.173, with
// max |dx| 3.5e-4 / 1.6e-6 and max |d log p| 2.4e-4 / 4.1e-6 against the float64 oracle (before: 3.5e-4 / 1.4e-6 and 2.4e-4 / 3.6e-6).
#ifndef JF_F32_NEWTON_FLOOR
#define JF_F32_NEWTON_FLOOR 1e-5
#endif

}  // namespace jf
