// Pieces shared by the fused conditional-block kernels that keep the parameter block in MFMA result registers
// (cond_split_kernels.hip: 16x16x32 tiles, lane = (row, coordinate); cond_pp_kernels.hip: 32x32x16 tiles, lane = (row, coordinate pair)):
// the parameter-slot numbering of one coordinate, the exact 3-way bf16 split, and the logistic mixture evaluated on a register row.
#pragma once
#include <type_traits>
#include "jf_gf.h"

namespace jf {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
typedef __attribute__((address_space(3))) void* cs_lptr;
typedef const __attribute__((address_space(1))) void* cs_gptr;

constexpr int CS_K = 10;                           // mixture components (parameter registers are indexed statically)
constexpr int CS_HH = 4;                           // Householder slots
constexpr int CS_SLOTS = 36;                       // parameter slots of one coordinate and layer (35 used)
constexpr int CS_SLOT_MEAN = 0, CS_SLOT_LW = CS_K, CS_SLOT_LN = 2 * CS_K, CS_SLOT_ROT = 3 * CS_K, CS_SLOT_OFF = 3 * CS_K + CS_HH;
struct CsLayer { int hh, model_offset, inv_type; float wmin, inv_wmax, nmin, nmax; };

struct CsPackLayer { int col0, off_rot, off_mean, off_lw, off_ln, hh, model_offset; };
// original column (inside the layer's row) of parameter slot `slot` for coordinate d, or -1
__device__ __forceinline__ int cs_slot_column(const CsPackLayer& o, int D, int slot, int d) {
    if (d >= D) return -1;
    if (slot < CS_SLOT_LW) return o.off_mean + slot * D + d;
    if (slot < CS_SLOT_LN) return o.off_lw + (slot - CS_SLOT_LW) * D + d;
    if (slot < CS_SLOT_ROT) return o.off_ln + (slot - CS_SLOT_LN) * D + d;
    if (slot < CS_SLOT_OFF) return (slot - CS_SLOT_ROT) < o.hh ? o.off_rot + (slot - CS_SLOT_ROT) * D + d : -1;
    if (slot == CS_SLOT_OFF) return o.model_offset ? d : -1;
    return -1;
}

__device__ __forceinline__ void cs_split(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)v;
    const float r1 = v - (float)hi;                // exact
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);                // exact difference, rounded once
}

// ---------------------------------------------------------------------------------------------------------- mixture on register rows
// gfg_mixture_impl<float, RAW, FAST> / gfg_mixture_scaled (jf_gf.h) with the lane's parameters in registers P[slot].
// The regulated 1/width and weight of every component and its u_k = (x - mu_k)/w_k are computed ONCE (2 exp + 2 rcp per component); the
// distance m = min_k |u_k| to the nearest component then decides -- per wave -- which ONE of the two summations runs (1 exp + 1 rcp per
// component each): the plain linear-space sums, or the sums scaled by e^{m} when some lane sits further than CS_M_SCALED widths from every
// component (below that the plain sums cannot underflow: cdf, sf >= pi_min sigma(-m) >= 1e-2 e^{-60}, pdf >= that / (2 w_max)).
// The first version ran the plain pass always and the scaled pass on top of it whenever a lane underflowed, regulating the parameters again
// in each (7 exp + 7 rcp per component on the benchmark inputs, two thirds of whose rows sit beyond 12 sigma after three layers).
constexpr float CS_M_SCALED = 60.0f;

// sums (optional): the normalised linear-space cdf, sf, pdf and 1 / N -- what the adjoint kernel's linear-space path starts from (a pdf that
// underflows there sends the row's wave to the log-space path)
struct CsSums { float C, S, P, invN; };
__device__ __forceinline__ MixQ<float> cs_mixture(const float (&P)[CS_SLOTS], const CsLayer& o, float x, bool live, CsSums* sums = nullptr) {
    using Mf = M<float>;
    float iw[CS_K], wk[CS_K], u[CS_K];
    float m = INFINITY, Nn = 0.f;
#pragma unroll
    for (int k = 0; k < CS_K; ++k) {
        const float ae = o.inv_wmax + Mf::exp_fast(-P[CS_SLOT_LW + k]);
        iw[k] = ae * Mf::rcp(o.wmin * ae + 1.0f);
        wk[k] = o.nmin + o.nmax * Mf::rcp(1.0f + Mf::exp_fast(-P[CS_SLOT_LN + k]));
        u[k] = (x - P[CS_SLOT_MEAN + k]) * iw[k];
        m = fminf(m, fabsf(u[k]));
        Nn += wk[k];
    }
    const float inv = Mf::rcp(Nn);
    MixQ<float> q;
    {   // plain sums: every lane.  (Until round 5 a wave with ONE far lane sent all its lanes through the scaled sums below, which round
        // differently: a row's bits depended on the 15 rows it shared a wave with -- 12 % of the rows of the SURVEY inputs differed between a
        // 2^20-row batch and the same rows evaluated alone, tests/test_gpu_fullsize.py.  The scaled sums are taken per LANE now; 3 % of the
        // calls see a far lane and pay for both forms.)
        float C = 0.f, S = 0.f, Pd = 0.f;
#pragma unroll
        for (int k = 0; k < CS_K; ++k) {
            // s = sigma(u) = 1 / (1 + e^{-u}), sigma(-u) = e^{-u} s: no |u|, no compare, no selects (jf_gfb.h: gfb_mix_sums).  The exponent is
            // capped so that e^{-u} stays finite (87 < ln FLT_MAX): such a lane is `far` (m > CS_M_SCALED) and takes the scaled sums below
            const float t = Mf::exp_fast(fminf(-u[k], 87.0f));
            const float s = Mf::rcp(1.0f + t);
            const float ts = t * s;
            C += wk[k] * s;
            S += wk[k] * ts;
            Pd += wk[k] * s * ts * iw[k];
        }
        C *= inv; S *= inv; Pd *= inv;
        q.lc = Mf::log_fast(C); q.ls = Mf::log_fast(S); q.lp = Mf::log_fast(Pd);
        q.cdf = C; q.sf = S;
        if (sums) *sums = CsSums{C, S, Pd, inv};
    }
    const bool far = m > CS_M_SCALED;                              // this lane's target is far from every component: plain sums underflow
    if (!__any(live && far)) return q;                             // wave-uniform branch
    const float em = Mf::exp_fast(-m);                             // may underflow to 0: the unscaled parts then stand alone
    float Cu = 0.f, Cs = 0.f, Su = 0.f, Ss = 0.f, Ps = 0.f;
#pragma unroll
    for (int k = 0; k < CS_K; ++k) {
        const float t = Mf::exp_fast(m - fabsf(u[k]));
        const float hi = Mf::rcp(1.0f + t * em);
        const float c1 = wk[k] * hi, c2 = c1 * t;
        if (u[k] >= 0.f) { Cu += c1; Ss += c2; }
        else { Su += c1; Cs += c2; }
        Ps += c2 * hi * iw[k];
    }
    Cu *= inv; Cs *= inv; Su *= inv; Ss *= inv; Ps *= inv;
    if (far) {
        q.cdf = Cu + em * Cs;
        q.sf = Su + em * Ss;
        q.lc = Cu > 0.f ? Mf::log_fast(q.cdf) : Mf::log_fast(Cs) - m;
        q.ls = Su > 0.f ? Mf::log_fast(q.sf) : Mf::log_fast(Ss) - m;
        q.lp = Mf::log_fast(Ps) - m;
        if (sums) *sums = CsSums{q.cdf, q.sf, Ps * em, inv};
    }
    return q;
}

// ---------------------------------------------------------------------------------------------------------- sampling direction on register rows
// gf_derive_column / gfg_mixture_impl<float, false, true> / gfg_mixture_scaled<float, false> / gfg_solve (jf_gf.h) with the lane's parameters in
// registers: the raw row P is regulated ONCE in place (log-width slots -> 1 / width, log-weight slots -> normalised weight), then the 25
// bisection + <= 20 Newton evaluations of a layer (bisection_n_newton.py:11-135, called from gaussianization_flow.py:921) read registers only
// -- the staged-row kernel reads three LDS words per component and evaluation.
__device__ __forceinline__ void cs_derive(float (&P)[CS_SLOTS], const CsLayer& o) {
    using Mf = M<float>;
    float Nn = 0.f;
#pragma unroll
    for (int k = 0; k < CS_K; ++k) {
        const float ae = o.inv_wmax + Mf::exp_fast(-P[CS_SLOT_LW + k]);
        P[CS_SLOT_LW + k] = ae * Mf::rcp(o.wmin * ae + 1.0f);
        const float w = o.nmin + o.nmax * Mf::rcp(1.0f + Mf::exp_fast(-P[CS_SLOT_LN + k]));
        P[CS_SLOT_LN + k] = w;
        Nn += w;
    }
    const float inv = Mf::rcp(Nn);
#pragma unroll
    for (int k = 0; k < CS_K; ++k) P[CS_SLOT_LN + k] *= inv;
}

template <typename T> __device__ __forceinline__ MixQ<T> cs_mixture_derived(const T (&P)[CS_SLOTS], T x) {
    using Mf = M<T>;
    T C = T(0), S = T(0), Pd = T(0);
#pragma unroll
    for (int k = 0; k < CS_K; ++k) {
        const T iw = P[CS_SLOT_LW + k], wk = P[CS_SLOT_LN + k];
        const T u = (x - P[CS_SLOT_MEAN + k]) * iw;
        if constexpr (sizeof(T) == 4) {                            // select-free form (cs_mixture above): sigma(u) = 1 / (1 + e^{-u}), sigma(-u) = e^{-u} sigma(u)
            const T t = Mf::exp_fast(Mf::min(-u, T(87)));
            const T s = Mf::rcp(T(1) + t);
            const T ts = t * s;
            C += wk * s;
            S += wk * ts;
            Pd += wk * s * ts * iw;
        } else {
            const T t = Mf::exp_fast(-Mf::abs(u));
            const T hi = Mf::rcp(T(1) + t);
            const T lo = t * hi;
            const bool pos = u >= T(0);
            C += wk * (pos ? hi : lo);
            S += wk * (pos ? lo : hi);
            Pd += wk * hi * lo * iw;
        }
    }
    MixQ<T> q;
    q.lc = Mf::log_fast(C); q.ls = Mf::log_fast(S); q.lp = Mf::log_fast(Pd);
    q.cdf = C; q.sf = S;
    const bool under = !(C > Mf::TINY && S > Mf::TINY && Pd > Mf::TINY);
    if (__any(under)) {                                            // wave-uniform: sums scaled by e^{m}, m = distance to the nearest component
        T m = T(INFINITY);
#pragma unroll
        for (int k = 0; k < CS_K; ++k) m = Mf::min(m, Mf::abs((x - P[CS_SLOT_MEAN + k]) * P[CS_SLOT_LW + k]));
        const T em = Mf::exp_fast(-m);
        T Cu = T(0), Cs = T(0), Su = T(0), Ss = T(0), Ps = T(0);
#pragma unroll
        for (int k = 0; k < CS_K; ++k) {
            const T iw = P[CS_SLOT_LW + k], wk = P[CS_SLOT_LN + k];
            const T u = (x - P[CS_SLOT_MEAN + k]) * iw;
            const T t = Mf::exp_fast(m - Mf::abs(u));
            const T hi = Mf::rcp(T(1) + t * em);
            const T c1 = wk * hi, c2 = c1 * t;
            if (u >= T(0)) { Cu += c1; Ss += c2; }
            else { Su += c1; Cs += c2; }
            Ps += c2 * hi * iw;
        }
        if (under) {
            q.cdf = Cu + em * Cs;
            q.sf = Su + em * Ss;
            q.lc = Cu > T(0) ? Mf::log_fast(q.cdf) : Mf::log_fast(Cs) - m;
            q.ls = Su > T(0) ? Mf::log_fast(q.sf) : Mf::log_fast(Ss) - m;
            q.lp = Mf::log_fast(Ps) - m;
        }
    }
    return q;
}

// x with stage(mixture(x)) = z: an approach phase (below) + gfg_solve's Newton stage step for step (stopping rules, status counters); RSUM /
// RMAX reduce over the lanes that hold the coordinates of one row.  P: DERIVED row (mean, 1 / width, normalised weight per component).
// info (optional): the caller solves the coordinates of a row in more than one call (two coordinates per lane) and books the row's status
// itself -- Newton row-steps = the longest of the calls, non-converged / non-finite = any of them; the counters are then not touched here.
struct CsSolveInfo { int steps; bool nonconv, nonfinite; };
template <typename T, typename RSUM, typename RMAX>
__device__ __forceinline__ T cs_solve(const T (&P)[CS_SLOTS], int inv_type, bool live, T z, bool row_valid, bool leader, int32_t* status,
                                      RSUM rsum, RMAX rmax, CsSolveInfo* info = nullptr, T* logd_out = nullptr, bool have_start = false, T start = T(0)) {
    using Mf = M<T>;
    // ---- approach phase (gf_approach, jf_gf.h): float64 rows run it in float32 on a float copy of the derived row, as they ran the bisection
    // have_start (per lane): the caller knows the solution to ~1e-3 already (broadcast parameters: the interpolated table of jf_gf_chain_fwd_tab,
    // gf_kernels.hip) -- the lane skips the approach phase; a wave none of whose lanes needs it skips the phase altogether
    using F = typename std::conditional<sizeof(T) == 8, float, T>::type;
    T x = have_start ? start : T(0);                               // (a lane without a start may have been handed a NaN: shadow lanes keep 0)
    if (__any(live && !have_start)) {
        F PF[CS_SLOTS];
#pragma unroll
        for (int k = 0; k < CS_SLOTS; ++k) PF[k] = k < 3 * CS_K ? (F)P[k] : F(0);
        // start: the mixture's mean moved by z mean-widths -- exact for one component of an isigmoid stage (x = mu + w z), and for a normal-type
        // stage with the classic logistic / normal match x = mu + 1.702 w z
        // ... widened by the spread of the component means (round 5): the single logistic / normal with the mixture's mean whose scale is the
        // within-component scale and the between-component standard deviation added in quadrature -- sqrt(wbar^2 + 3 var_b / pi^2) for the
        // logistic, sqrt((1.702 wbar)^2 + var_b) for the normal.  With var_b = 0 this is the start above; for components that lie apart
        // (what a trained or a scaled-up random amortisation MLP emits) the within-width alone put the start several widths short of the root.
        F xf = F(0), wbar = F(0), m2 = F(0);
#pragma unroll
        for (int k = 0; k < CS_K; ++k) {
            const F pm = PF[CS_SLOT_LN + k] * PF[CS_SLOT_MEAN + k];
            xf += pm;
            m2 += pm * PF[CS_SLOT_MEAN + k];
            wbar += PF[CS_SLOT_LN + k] * M<F>::rcp(PF[CS_SLOT_LW + k]);
        }
        {
            const F zc = M<F>::min(M<F>::max((F)z, F(-8)), F(8));
            const F var_b = M<F>::max(m2 - xf * xf, F(0));
            const F sc = inv_type != JF_GF_ISIGMOID ? M<F>::sqrt(F(2.896804) * wbar * wbar + var_b) : M<F>::sqrt(wbar * wbar + F(0.30396355) * var_b);
            xf += sc * zc;
        }
#ifdef JF_PROBE_COUNT_APPROACH
        int n_ev = 0;
        xf = gf_approach<F>([&](F xx) { return cs_mixture_derived<F>(PF, xx); }, inv_type != JF_GF_ISIGMOID, (F)z, xf, live && !have_start, &n_ev);
        if (info == nullptr) { for (int i = 0; i < n_ev; ++i) status_add(status, JF_STATUS_NEWTON_STEPS, row_valid && leader); }
#else
        xf = gf_approach<F>([&](F xx) { return cs_mixture_derived<F>(PF, xx); }, inv_type != JF_GF_ISIGMOID, (F)z, xf, live && !have_start);
#endif
        if (!have_start) x = (T)xf;
    }
    bool active = row_valid;
    T ferr = T(0), prev = T(INFINITY);
    bool nonfinite = false;
    int n_steps = 0;
    T last_logd = T(0);
    bool coarse_end = false;
    for (int it = 0; it < 20 && __any(active); ++it) {
        const IcdfOut<T> s = gf_icdf<T>(inv_type, cs_mixture_derived<T>(P, x));
        const T f = s.y - z;
        T upd;
        if constexpr (sizeof(T) == 4) upd = f * Mf::exp_fast(-s.logd);      // (v_exp_f32: the step's relative error ~1e-6 does not move a Newton iterate's limit)
        else upd = f / Mf::exp(s.logd);
        const T usum = rsum(live ? Mf::abs(upd) : T(0));
        if (info == nullptr) status_add(status, JF_STATUS_NEWTON_STEPS, active && leader);
        n_steps += active ? 1 : 0;
        if (active) {
            const T nx = x - upd;
            if (Mf::finite(nx)) x = nx; else nonfinite = nonfinite || live;
            ferr = Mf::abs(f);
            last_logd = s.logd;
            active = usum >= newton_tol<T>();
        }
        if (sizeof(T) == 4 && !newton_reference_rule()) {
            // float32 floor of the update (see gfg_solve): the reference's float32 runs spend their last ~16 Newton steps on rounding noise
            const T xs = rsum(live ? Mf::max(Mf::abs(x), T(1)) : T(0));
            // at the coordinates' resolution, or (jf_math.h) below JF_F32_NEWTON_FLOOR of them with the residual of the evaluation just made
            // already inside the reference's own convergence threshold (1e-4 in float32: what its report looks at, bisection_n_newton.py:118-131)
            const bool done = usum < T(2.5e-7) * xs || (usum < T(JF_F32_NEWTON_FLOOR) * xs && rmax(live ? Mf::abs(f) : T(0)) <= T(1e-4));
            if (active && usum >= T(0.5) * prev && usum < T(1e-4) * xs && !done) coarse_end = true;   // stopped on stagnation
            if (done || (usum >= T(0.5) * prev && usum < T(1e-4) * xs)) active = false;
            prev = usum;
        }
    }
    if (logd_out != nullptr) {
        // the stage's log-derivative at the solution (the caller's log-det term).  A converged row's last update was below 1e-9 (NewtonTol; float32: below
        // JF_F32_NEWTON_FLOOR = 1e-5 of the coordinate, jf_math.h), so the value of its last evaluation IS the value at the returned point to rounding: one
        // evaluation (mixture + inverse CDF; a fifth of a float64 solve) saved.  A wave with a row that ran out of iterations, or -- float32 --
        // stopped on the stagnation rule (updates up to 1e-4 of the coordinate), evaluates at the returned point as the reference does
        // (gaussianization_flow.py:922-924)
        const bool stale = row_valid && (active || n_steps == 0 || coarse_end);
        if (__any(stale)) *logd_out = gf_icdf<T>(inv_type, cs_mixture_derived<T>(P, x)).logd;
        else *logd_out = last_logd;
    }
    const T prec = sizeof(T) == 8 ? T(1e-7) : T(1e-4);
    const T ferr_row = rmax(live ? ferr : T(0));
    const T nf_row = rmax(nonfinite ? T(1) : T(0));
    if (info != nullptr) {
        info->steps = n_steps; info->nonconv = ferr_row > prec; info->nonfinite = nf_row > T(0);
        return x;
    }
    status_add(status, JF_STATUS_NONCONVERGED, row_valid && leader && (ferr_row > prec));
    status_add(status, JF_STATUS_NONFINITE, row_valid && leader && (nf_row > T(0)));
    return x;
}

}  // namespace jf
