// Broadcast-parameter g chain, log-prob direction, lane = ROW (gaussianization_flow.py:995-1114 per layer, main/default.py:998-1031 loop): the
// kernel arguments of every g-chain kernel and the body of gfb_chain_inv_kernel as a device function, so that the stand-alone kernel
// (gf_kernels.hip) and the merged log-prob step (merged_kernels.hip) run the same code.
#pragma once
#include "jf_gf.h"
#include "jf_gf_ext.h"
#include "jf_cond_regs.h"

namespace jf {

template <typename T> struct GfChainArgs {
    const T* x; int64_t xs;
    const T* ld_in;
    const T* params; int64_t ps;
    int64_t B;
    int D;
    int n_layers;
    int tiles_per_block;     // broadcast kernels: row tiles walked by one workgroup
    int tile_stride;         // per-sample: LDS row stride (elements); broadcast: row capacity per layer
    int tab_offset;          // element offset of the spline knot tables behind the parameter tile
    int spline_tab;          // words of a lane's knot table: spline_tab_words of the chain's largest rq_splines bin count (0: no spline stretch)
    GfLayerDev<T> L[JF_MAX_CHAIN];
    T* x_out; int64_t xos;
    T* ld_out;
    const T* blp_in; T* blp_out;
    int64_t* bins; int64_t bins_stride;
    int32_t* status;
    // log-prob direction only: a co-vector carried along with x -- v <- J_l^{-T} v per layer (J_l = diag(dy/dx) Q_l^T: the same reflections as x,
    // then a division by the stage's derivative), i.e. cot_out = J^{-T} cot_in for the chain's Jacobian J = dy/dx (jf_gf_chain_inv_cot)
    const T* cot_in; int64_t cis;
    T* cot_out; int64_t cos;
    T* total;                // log-prob direction, nullable (needs blp_out): total[b] = blp_out[b] + ld_out[b] -- log_prob = log_prob_base + log_det
                             // (main/default.py:1110-1117) written by the chain launch itself instead of a launch of its own
    // sampling direction, broadcast parameters only: interpolation table of every (layer, coordinate)'s inverse x(z) (gf_fwd_table_kernel), or null
    T* table;
};

template <int G> struct Log2 { static constexpr int v = (G == 1) ? 0 : (G == 2) ? 1 : (G == 4) ? 2 : (G == 8) ? 3 : (G == 16) ? 4 : (G == 32) ? 5 : 6; };

// broadcast regime: raw rows -> LDS, then wave w derives layers w, w+4, ... (columns on lanes 0..D-1, reflections on lanes 32..)
template <typename T> __device__ __forceinline__ void derive_broadcast(T* lds, const GfChainArgs<T>& a) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int l = 0; l < a.n_layers; ++l) {
        const GfLayerDev<T> o = a.L[l];
        for (int j = tid; j < o.n_params; j += blockDim.x) lds[l * a.tile_stride + j] = a.params[o.col0 + j];
    }
    __syncthreads();
    for (int l = wave; l < a.n_layers; l += 4) {
        const GfLayerDev<T> o = a.L[l];      // wave-uniform index
        T* row = lds + l * a.tile_stride;
        if (a.D <= 32) {
            if (lane < a.D) { if (o.stretch == JF_GF_STRETCH_CLASSIC) gf_derive_column<T>(row, o, a.D, lane); }
            else if (lane >= 32 && lane - 32 < o.hh) gf_derive_reflection<T>(row, o, a.D, lane - 32);
        } else {                                         // 33 .. 64 coordinates: columns, then reflections (different words of the row), on all lanes
            if (lane < a.D && o.stretch == JF_GF_STRETCH_CLASSIC) gf_derive_column<T>(row, o, a.D, lane);
            for (int i = lane; i < o.hh; i += 64) gf_derive_reflection<T>(row, o, a.D, i);
        }
    }
    __syncthreads();
}

// Broadcast regime, log-prob direction, classic stretch: lane = ROW, the row's D coordinates in registers.  Every parameter is then the same
// for all 64 lanes of a wave: the derived (mean, 1/width, pi, pi/width) of a component come from ONE uniform 16-byte (float64: 32-byte) LDS
// read per 64 rows (the lane = (row, coordinate) kernel above spends 3 ds_read_b32 + 3 address adds per component on 16 rows), the
// Householder dot products and the sum of the log-derivatives are plain register arithmetic (no DPP butterflies), and x is one row-contiguous
// load per lane.  Same arithmetic per coordinate as gfg_mixture_impl / gfg_mixture_scaled.
template <typename T> struct __attribute__((aligned(16))) GfPack { T mean, iw, pi, piw; };

// Rows of a wave whose plain sums underflowed (a target tens of widths from every component), lane = row kernel.  On the SURVEY inputs that is
// 1 % of the (row, coordinate, layer) evaluations but a lane in 22 % of the waves -- 2.5 lanes of 64 on average -- and rounds 1-3 sent the whole
// wave through gfg_mixture_scaled for them (a second walk over the components with a distance pass in front: +30 % on the kernel).  Here the
// wave turns ITS LANES to those few rows instead: four rows per pass, one per 16-lane DPP row, lane k of a row takes component k (K <= 16) of
// the row's target, and the distance minimum and the five scaled sums are row all-reductions by rotation (row_ror 8 / 4 / 2 / 1).  Same
// arithmetic per component as gfg_mixture_scaled, the sums in tree order: a row's result still depends on nothing but its own target.
constexpr int DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR2 = 0x122, DPP_ROW_ROR1 = 0x121;
template <typename T> __device__ __forceinline__ T row16_sum(T v) {
    v += dpp_swap<DPP_ROW_ROR8>(v); v += dpp_swap<DPP_ROW_ROR4>(v); v += dpp_swap<DPP_ROW_ROR2>(v); v += dpp_swap<DPP_ROW_ROR1>(v);
    return v;
}
template <typename T> __device__ __forceinline__ T row16_min(T v) {
    v = M<T>::min(v, dpp_swap<DPP_ROW_ROR8>(v)); v = M<T>::min(v, dpp_swap<DPP_ROW_ROR4>(v));
    v = M<T>::min(v, dpp_swap<DPP_ROW_ROR2>(v)); v = M<T>::min(v, dpp_swap<DPP_ROW_ROR1>(v));
    return v;
}
template <typename T> __device__ __forceinline__ void gfb_scaled_rows(const GfPack<T>* __restrict__ pd, int K, T xd, bool under, MixQ<T>& q) {
    unsigned long long mask = __ballot(under);
    const int lane = threadIdx.x & 63, grp = lane >> 4, k = lane & 15;
    const bool comp = k < K;
    GfPack<T> e = pd[comp ? k : 0];
    if constexpr (sizeof(T) == 4) e.iw *= T(-0.6931471805599453);     // the float32 records carry -log2(e) / width (see the pack loop)
    const T pk = comp ? e.pi : T(0);
    while (mask != 0ull) {                                      // wave-uniform
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));     // this lane's row is the rank-th underflowed row still to do
        unsigned long long mm = mask;
        int s[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { s[g] = mm != 0ull ? __ffsll((long long)mm) - 1 : 0; mm &= mm - 1ull; }    // (no row left: lane 0's target, result unused)
        const int src = grp == 0 ? s[0] : grp == 1 ? s[1] : grp == 2 ? s[2] : s[3];
        const T xs = __shfl(xd, src, 64);
        const T u = (xs - e.mean) * e.iw;
        const T au = comp ? M<T>::abs(u) : T(INFINITY);
        const T m = row16_min<T>(au);
        const T em = M<T>::exp_fast(-m);                       // may underflow to 0: the unscaled parts then stand alone
        const T tp = M<T>::exp_fast(m - au);                   // <= 1; 0 for the lanes without a component
        const T hi = M<T>::rcp(T(1) + tp * em);
        const T c1 = pk * hi, c2 = c1 * tp;
        const bool pos = u >= T(0);
        const T Cu = row16_sum<T>(pos ? c1 : T(0)), Su = row16_sum<T>(pos ? T(0) : c1);
        const T Ss = row16_sum<T>(pos ? c2 : T(0)), Cs = row16_sum<T>(pos ? T(0) : c2);
        const T Ps = row16_sum<T>(c2 * hi * e.iw);
        MixQ<T> r;
        r.cdf = Cu + em * Cs;
        r.sf = Su + em * Ss;
        r.lc = Cu > T(0) ? M<T>::log_fast(r.cdf) : M<T>::log_fast(Cs) - m;
        r.ls = Su > T(0) ? M<T>::log_fast(r.sf) : M<T>::log_fast(Ss) - m;
        r.lp = M<T>::log_fast(Ps) - m;
        // the row groups hand their results to the rows they worked for
        const int from = 16 * (rank < 4 ? rank : 0);
        const T lc = __shfl(r.lc, from, 64), ls = __shfl(r.ls, from, 64), lp = __shfl(r.lp, from, 64), cd = __shfl(r.cdf, from, 64), sf = __shfl(r.sf, from, 64);
        if (under && ((mask >> lane) & 1ull) != 0ull && rank < 4) { q.lc = lc; q.ls = ls; q.lp = lp; q.cdf = cd; q.sf = sf; }
        mask = mm;                                              // the four lowest rows are done
    }
}

// plain mixture sums of ONE coordinate over its K packed components (both lane layouts run this very code: bit-identical per coordinate)
template <typename T> __device__ __forceinline__ void gfb_mix_sums(const GfPack<T>* __restrict__ pd, const int K, const T xd, T& C, T& S, T& P) {
    C = T(0); S = T(0); P = T(0);
    constexpr int GFB_UNROLL = sizeof(T) == 4 ? 5 : 2;
#pragma unroll GFB_UNROLL                                    // (float64 at 5: 132 -> 163 VGPRs, 0.57 -> 0.60 ms per 2^20 rows)
    for (int k = 0; k < K; ++k) {
        const GfPack<T> e = pd[k];
        if constexpr (sizeof(T) == 4) {
            // s = sigma(u) = 1 / (1 + 2^a), a = -u log2(e) (the record's iw carries -log2(e) / width); sigma(-u) = 2^a s.  No |u|, no compare, no
            // selects: 9 vector + 2 transcendental instructions per component instead of 11 + 2.  a is capped at 126 -- 2^a stays finite, s
            // bottoms out at 2^-126 (below TINY: the row goes to the scaled sums exactly as before) and 2^a s = 1.
            const T a = M<T>::min((xd - e.mean) * e.iw, T(126));
            const T t = __builtin_amdgcn_exp2f(a);
            const T s = M<T>::rcp(T(1) + t);               // sigma(u)
            const T ts = t * s;                            // sigma(-u)
            C += e.pi * s;
            S += e.pi * ts;
            P += e.piw * (s * ts);
        } else {
            const T u = (xd - e.mean) * e.iw;
            const T tt = M<T>::exp_fast(-M<T>::abs(u));
            const T hi = M<T>::rcp(T(1) + tt);             // sigma(|u|)
            const T lo = tt * hi;                          // sigma(-|u|)
            const bool pos = u >= T(0);
            C += e.pi * (pos ? hi : lo);
            S += e.pi * (pos ? lo : hi);
            P += e.piw * (hi * lo);
        }
    }
}
// the cross-coordinate arithmetic of a row, shared by both lane layouts (same expressions -> same contraction into fused multiply-adds)
template <typename T, int D> __device__ __forceinline__ T gfb_hh_dot(const T* __restrict__ v, const T (&x)[D]) {
    T dot = T(0);
#pragma unroll
    for (int d = 0; d < D; ++d) dot += v[d] * x[d];
    return dot;
}
template <typename T> __device__ __forceinline__ T gfb_hh_apply(T xd, T vd, T dot) { return xd - vd * dot; }
template <typename T> __device__ __forceinline__ T gfb_base_term(T xd) { return T(-0.5) * xd * xd - M<T>::HALF_LN_2PI; }

// `block`: the workgroup's index among the chain's workgroups
template <typename T, int D>
__device__ __forceinline__ void gfb_chain_inv_body(const GfChainArgs<T>& a, const int block, unsigned char* smem_raw) {
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x;
    derive_broadcast<T>(lds, a);
    // the layer count as a scalar register: derive_broadcast reads it under divergent control flow, and the value the compiler then reuses
    // lives in a vector register -- which turned the layer loop, its descriptor loads and the component loop into divergent (exec-masked) code
    const int n_layers = __builtin_amdgcn_readfirstlane(a.n_layers);
    int max_k = 1;
    for (int l = 0; l < n_layers; ++l) max_k = a.L[l].K > max_k ? a.L[l].K : max_k;
    const int pstride = __builtin_amdgcn_readfirstlane(max_k) * D;
    GfPack<T>* pack = reinterpret_cast<GfPack<T>*>(lds + a.tab_offset);
    for (int l = 0; l < n_layers; ++l) {
        const GfLayerDev<T> o = a.L[l];
        const T* row = lds + l * a.tile_stride;
        for (int j = tid; j < o.K * D; j += 256) {
            const int d = j / o.K, k = j - d * o.K;
            GfPack<T> e;
            e.mean = row[o.off_mean + k * D + d];
            e.iw = row[o.off_lw + k * D + d];
            e.pi = o.fit_norm ? row[o.off_ln + k * D + d] : M<T>::rcp(T(o.K));
            e.piw = e.pi * e.iw;
            // float32: the record carries -log2(e) / width, so that e^{-|u|} = 2^{|x - mean| iw} is one multiply and v_exp_f32 (the sign of u
            // is the sign of x - mean): one VALU instruction less per component (14.5 -> 13.5)
            if constexpr (sizeof(T) == 4) e.iw *= T(-1.4426950408889634);
            pack[l * pstride + j] = e;
        }
    }
    __syncthreads();

    for (int t = 0; t < a.tiles_per_block; ++t) {
        const int64_t row0 = ((int64_t)block * a.tiles_per_block + t) * 256;
        if (row0 >= a.B) break;                          // block-uniform
        const int64_t row = row0 + tid;
        const bool row_valid = row < a.B;
        const int64_t rrow = row_valid ? row : a.B - 1;
        T x[D];
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = a.x[rrow * a.xs + d];
        T ld = a.ld_in ? a.ld_in[rrow] : T(0);
        for (int l = n_layers - 1; l >= 0; --l) {
            const GfLayerDev<T> o = a.L[l];              // uniform index: scalar loads from the kernarg segment
            const T* prow = lds + l * a.tile_stride;
            if (o.model_offset) {                        // euclidean_base.py:40-45
#pragma unroll
                for (int d = 0; d < D; ++d) x[d] -= prow[d];
            }
            for (int i = 0; i < o.hh; ++i) {             // x <- Q^T x (:1038); derived rows hold sqrt(2) v / |v|
                const T* v = prow + o.off_rot + i * D;
                const T dot = gfb_hh_dot<T, D>(v, x);
#pragma unroll
                for (int d = 0; d < D; ++d) x[d] = gfb_hh_apply<T>(x[d], v[d], dot);
            }
            const GfPack<T>* pk = pack + l * pstride;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const GfPack<T>* pd = pk + d * o.K;
                T C, S, P;
                gfb_mix_sums<T>(pd, o.K, x[d], C, S, P);
                MixQ<T> q;
                q.lc = M<T>::log_fast(C); q.ls = M<T>::log_fast(S); q.lp = M<T>::log_fast(P);
                q.cdf = C; q.sf = S;
                const bool under = !(C > M<T>::TINY && S > M<T>::TINY && P > M<T>::TINY);
                if (__any(under)) {                        // wave-uniform branch
                    if (o.K <= 16) {
                        gfb_scaled_rows<T>(pd, o.K, x[d], under, q);
                    } else {
                        const MixQ<T> qs = gfg_mixture_scaled<T, false>(prow + d, o, D, x[d], T(0));
                        if (under) q = qs;
                    }
                }
                const IcdfOut<T> sy = gf_icdf<T>(o.inv_type, q);
                x[d] = sy.y;
                ld += sy.logd;
            }
        }
        T sb = T(0);
        bool bad = !M<T>::finite(ld);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (row_valid) a.x_out[row * a.xos + d] = x[d];
            sb += gfb_base_term<T>(x[d]);
            bad = bad || !M<T>::finite(x[d]);
        }
        if (row_valid) {
            a.ld_out[row] = ld;
            const T bv = sb + (a.blp_in ? a.blp_in[row] : T(0));
            if (a.blp_out) a.blp_out[row] = bv;
            if (a.total) a.total[row] = bv + ld;
        }
        status_add(a.status, JF_STATUS_NONFINITE, row_valid && bad);
    }
}

// The same chain with G = 2 or 4 LANES PER ROW (lane = (row, coordinate)), for batches that leave the lane = row form short of waves: at 2^17
// rows lane = row is 2 waves per SIMD, each walking layers x D x K components alone (~20 us however few the rows); here a wave carries 64 / G
// rows and a lane one coordinate's K components per layer -- G times the waves, 1 / G of the dependent chain.  Bit for bit the lane = row
// results: a coordinate's mixture runs the same code (gfb_mix_sums), and what couples the coordinates of a row -- the Householder dot
// products, the sums of the log-derivatives and of the base log-probabilities -- is evaluated by EVERY lane of the row on the row's gathered
// values in the lane = row kernel's order (quad-permute DPP gathers; no butterfly sums, whose rounding order would differ).  So the kernel
// choice may follow the batch size without a row's result depending on the batch it sits in.
template <int G, int J> struct QuadBcast;                             // DPP quad_perm control: every lane of a group reads the group's lane J
template <> struct QuadBcast<4, 0> { static constexpr int v = 0x00; };
template <> struct QuadBcast<4, 1> { static constexpr int v = 0x55; };
template <> struct QuadBcast<4, 2> { static constexpr int v = 0xAA; };
template <> struct QuadBcast<4, 3> { static constexpr int v = 0xFF; };
template <> struct QuadBcast<2, 0> { static constexpr int v = 0xA0; };   // [0,0,2,2]
template <> struct QuadBcast<2, 1> { static constexpr int v = 0xF5; };   // [1,1,3,3]
template <typename T, int D, int G, int J = 0> __device__ __forceinline__ void gfb_gather(T v, T (&out)[D]) {
    if constexpr (J < D) {
        out[J] = dpp_swap<QuadBcast<G, J>::v>(v);
        gfb_gather<T, D, G, J + 1>(v, out);
    }
}

template <typename T, int D, int G>
__device__ __forceinline__ void gfbg_chain_inv_body(const GfChainArgs<T>& a, const int block, unsigned char* smem_raw) {
    static_assert((G == 2 || G == 4) && D <= G && 2 * D > G, "G = the power of two that holds D coordinates");
    constexpr int ROWS = 256 / G;
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x;
    derive_broadcast<T>(lds, a);
    const int n_layers = __builtin_amdgcn_readfirstlane(a.n_layers);
    int max_k = 1;
    for (int l = 0; l < n_layers; ++l) max_k = a.L[l].K > max_k ? a.L[l].K : max_k;
    const int pstride = __builtin_amdgcn_readfirstlane(max_k) * D;
    GfPack<T>* pack = reinterpret_cast<GfPack<T>*>(lds + a.tab_offset);
    for (int l = 0; l < n_layers; ++l) {
        const GfLayerDev<T> o = a.L[l];
        const T* row = lds + l * a.tile_stride;
        for (int j = tid; j < o.K * D; j += 256) {
            const int d = j / o.K, k = j - d * o.K;
            GfPack<T> e;
            e.mean = row[o.off_mean + k * D + d];
            e.iw = row[o.off_lw + k * D + d];
            e.pi = o.fit_norm ? row[o.off_ln + k * D + d] : M<T>::rcp(T(o.K));
            e.piw = e.pi * e.iw;
            if constexpr (sizeof(T) == 4) e.iw *= T(-1.4426950408889634);
            pack[l * pstride + j] = e;
        }
    }
    __syncthreads();

    const int g = tid & (G - 1), r = tid / G;
    const bool live = g < D;
    const int d = live ? g : D - 1;                       // (a spare lane of a 3-coordinate row shadows the last coordinate; it stores nothing)
    for (int t = 0; t < a.tiles_per_block; ++t) {
        const int64_t row0 = ((int64_t)block * a.tiles_per_block + t) * ROWS;
        if (row0 >= a.B) break;                          // block-uniform
        const int64_t row = row0 + r;
        const bool row_valid = row < a.B;
        const int64_t rrow = row_valid ? row : a.B - 1;
        T xd = a.x[rrow * a.xs + d];
        T ld = a.ld_in ? a.ld_in[rrow] : T(0);
        for (int l = n_layers - 1; l >= 0; --l) {
            const GfLayerDev<T> o = a.L[l];
            const T* prow = lds + l * a.tile_stride;
            if (o.model_offset) xd -= prow[d];            // euclidean_base.py:40-45
            for (int i = 0; i < o.hh; ++i) {             // x <- Q^T x (:1038)
                const T* v = prow + o.off_rot + i * D;
                T x[D];
                gfb_gather<T, D, G>(xd, x);
                const T dot = gfb_hh_dot<T, D>(v, x);
                xd = gfb_hh_apply<T>(xd, v[d], dot);
            }
            const GfPack<T>* pk = pack + l * pstride;
            const GfPack<T>* pd = pk + d * o.K;
            T C, S, P;
            gfb_mix_sums<T>(pd, o.K, xd, C, S, P);
            MixQ<T> q;
            q.lc = M<T>::log_fast(C); q.ls = M<T>::log_fast(S); q.lp = M<T>::log_fast(P);
            q.cdf = C; q.sf = S;
            const bool under = !(C > M<T>::TINY && S > M<T>::TINY && P > M<T>::TINY);
            if (__any(under)) {                            // wave-uniform branch
                if (o.K <= 16) {
                    for (int dd = 0; dd < D; ++dd) {       // the wave's lanes turn to the underflowed (row, coordinate dd) evaluations, coordinate by coordinate
                        const bool mine = under && d == dd && live;
                        if (__any(mine)) gfb_scaled_rows<T>(pk + dd * o.K, o.K, xd, mine, q);
                    }
                    if (under && !live) {                  // the shadow lane repeats its coordinate's lane (keeps its values finite; never stored)
                        const MixQ<T> qs = gfg_mixture_scaled<T, false>(prow + d, o, D, xd, T(0));
                        q = qs;
                    }
                } else {
                    const MixQ<T> qs = gfg_mixture_scaled<T, false>(prow + d, o, D, xd, T(0));
                    if (under) q = qs;
                }
            }
            const IcdfOut<T> sy = gf_icdf<T>(o.inv_type, q);
            xd = sy.y;
            T lg[D];
            gfb_gather<T, D, G>(sy.logd, lg);
#pragma unroll
            for (int dd = 0; dd < D; ++dd) ld += lg[dd];   // coordinate order, as lane = row
        }
        T xs[D];
        gfb_gather<T, D, G>(xd, xs);
        T sb = T(0);
        bool bad = !M<T>::finite(ld);
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            sb += gfb_base_term<T>(xs[dd]);
            bad = bad || !M<T>::finite(xs[dd]);
        }
        if (row_valid && live) a.x_out[row * a.xos + d] = xd;
        if (row_valid && g == 0) {
            a.ld_out[row] = ld;
            const T bv = sb + (a.blp_in ? a.blp_in[row] : T(0));
            if (a.blp_out) a.blp_out[row] = bv;
            if (a.total) a.total[row] = bv + ld;
        }
        status_add(a.status, JF_STATUS_NONFINITE, row_valid && g == 0 && bad);
    }
}

}  // namespace jf
