// Training step of an e-block whose parameter rows come out of a LOW-RANK last MLP stage (BASELINE configs[4], float64):
//     params[row] = U2 t2[row] + b2,   t2 (B, r2 <= 8),  U2 (N, r2)        (amortizable_mlp.py:508-578)
// The round-3 training step materialised the (B, N) block three times over: written by the last dense launch, read by the chain's forward and
// by its adjoint, whose (B, N) gradient block was then read by two more dense launches (g U2 and g^T t2) -- 9.8 KB per row each time, 2.4 of
// the 4.3 ms of a C5 step at 2^17 rows.  Here neither block exists:
//   lr_gf_fwd_kernel        the chain's log-prob direction with every parameter regenerated from t2 on the f64 matrix cores (the layout of
//                           jf_amlp_mfma.h: a wave owns 16 rows, lane = (row n, group q), group q owns coordinates q and q + 4); it also keeps
//                           each layer's rotated coordinates (behind its offset and reflections) and linear-space mixture sums (5 doubles per
//                           coordinate and layer: `aux`)
//   lr_gf_bwd_layer_kernel  ONE layer's adjoint per launch (layer 0 -- the last one applied -- first): parameters regenerated the same way, the
//                           gradient of a parameter tile stays in the MFMA result registers and is contracted twice right there:
//                             g_t2^T (r2 x rows)  += U2'^T G       4 products per tile, A gathered from the same LDS image of U2
//                             g_U2' (cols x r2|1) += G [t2 | 1]    4 products per tile over the wave's 16 rows (G transposed through a 2 KB
//                                                                  LDS scratch), summed over the workgroup's rows with float64 LDS atomics
//                           (the column of ones makes g_b2 the 9th column).  Persistent workgroups; one partial image per workgroup.
//   lr_reduce_kernel        partial images -> g_U2 (N, r2), g_b2 (N) in natural column order, fixed summation order
// Per-layer launches keep the LDS at the image and the accumulators of ONE layer (67 KB + 4 KB of scratch per wave): one workgroup of 8 waves
// per CU, 256 workgroups -- at 2^17 rows every wave walks exactly 4 row tiles (230 VGPRs; 12 waves at 168 VGPRs spilled 70 and left the waves
// with 2 or 3 tiles: 0.84 instead of 0.70 ms).  The upstream gradient of the coordinates travels through g_x between the launches (64 bytes
// per row), g_t2 is accumulated in place.
// Where the 0.70 ms of the four layer launches + reduction go at 2^17 rows of C5 (parts switched off one at a time, round 4): LDS atomics
// 0.19 (36 lanes x float64 per instruction at about one lane per clock), the mixture arithmetic 0.23, the reflections 0.17, the products
// ~0.06, launches + staging + partial images 0.06, reduction 0.01 (0.13 as one thread per entry walking 512 images), the rest is the latency
// of the row tiles' first loads.  The (B, P)-block kernel it replaces took 1.21 ms, the three dense launches around it another 0.93.
// Supported: what jf_amlp_gf_chain_inv_f64's matrix-core kernel supports (float64, r2 <= 8, D <= 8, default layer options).
#pragma once
#include "jf_gf_bwd.h"

namespace jf {

constexpr int LR_NW = 8;                        // waves of a backward workgroup (one workgroup per CU: three waves per SIMD)
constexpr int LR_RS = AM_R + 1;                  // accumulator row: r2 <= 8 rank columns + the bias column
constexpr int LR_TILES = AG_HH + 1 + 2 * AG_K;   // tiles of a layer in the BACKWARD kernel's order (lr_col)
constexpr int LR_PSZ = LR_TILES * 16 * LR_RS;    // doubles of one layer's partial image
constexpr int LR_MAX_WG = 256;                   // one resident workgroup per CU of an MI355X
constexpr int LR_AUX = 5;                        // per layer and coordinate: rotated input, cdf, sf, pdf sums (normalised), 1 / sum of weights

template <typename T> struct LrFwdArgs {
    const T* t2; int64_t t2s; const T* U2; const T* b2; int r2;
    const T* x; int64_t xs; const T* ld_in; int64_t B; int D, n_layers;
    AgLayer<T> L[JF_MAX_CHAIN];
    T* x_out; int64_t xos; T* ld_out; const T* blp_in; T* blp_out;
    T* aux;                                      // (n_layers, LR_AUX, 2, B, 4) or null
    int32_t* status;
};

template <typename T> struct LrBwdArgs {
    const T* t2; int64_t t2s; const T* U2; const T* b2; int r2;
    const T* aux; const T* x_out; int64_t xos;
    int64_t B, n_row_tiles; int D, layer, n_layers, first;
    AgLayer<T> L;
    const T* g_xout; int64_t gxos; const T* g_ld; const T* g_blp;
    T* g_x; int64_t gxs;
    T* g_t2;                                     // (B, AM_R)
    T* partial;                                  // (gridDim.x, LR_PSZ) of this layer
    int32_t* status;
};

// ag_mixture (amlp_gf_kernels.hip) that also hands out the normalised linear-space sums; C = 0 marks a wave that took the scaled evaluation
template <typename T> __device__ __forceinline__ MixQ<T> ag_mixture_sums(const T (&P)[AG_SLOTS], const AgLayer<T>& o, T x, bool live, MixSums<T>& m) {
    T C = T(0), S = T(0), Pd = T(0), Nn = T(0);
#pragma unroll
    for (int k = 0; k < AG_K; ++k) {
        const T ae = o.inv_wmax + M<T>::exp_fast(-P[AG_LW + k]);
        const T iw = ae * M<T>::rcp(o.wmin * ae + T(1));
        const T wk = o.nmin + o.nmax * M<T>::rcp(T(1) + M<T>::exp_fast(-P[AG_LN + k]));
        const T u = (x - P[AG_MEAN + k]) * iw;
        const T t = M<T>::exp_fast(-M<T>::abs(u));
        const T hi = M<T>::rcp(T(1) + t);
        const T lo = t * hi;
        const bool pos = u >= T(0);
        C += wk * (pos ? hi : lo);
        S += wk * (pos ? lo : hi);
        Pd += wk * hi * lo * iw;
        Nn += wk;
    }
    const T inv = M<T>::rcp(Nn);
    C *= inv; S *= inv; Pd *= inv;
    m.C = C; m.S = S; m.P = Pd; m.invN = inv;
    MixQ<T> q;
    q.lc = M<T>::log_fast(C); q.ls = M<T>::log_fast(S); q.lp = M<T>::log_fast(Pd);
    q.cdf = C; q.sf = S;
    const bool under = live && !(C > M<T>::TINY && S > M<T>::TINY && Pd > M<T>::TINY);
    if (__any(under)) {
        const MixQ<T> qs = ag_mixture_scaled<T>(P, o, x);
        if (under) { q = qs; m.C = T(0); }
    }
    return q;
}

// the permuted fragment image of one layer's rows of U2 / b2 (jf_amlp_mfma.h: am_col) -> LDS
template <typename T, typename LAYER>
__device__ __forceinline__ void lr_stage_layer(const LAYER& o, int D, int r2, const T* __restrict__ U2, const T* __restrict__ b2, T* fU2, T* sb2, int tid, int nt) {
    for (int e = tid; e < AM_TILES * 2 * 64; e += nt) {
        const int f = e >> 6, l = e & 63, m = l & 15, k = 4 * (f & 1) + (l >> 4), tt = f >> 1;
        const int col = am_col(o, D, tt, m & 3, m >> 2);
        fU2[e] = (col >= 0 && k < r2) ? U2[(int64_t)col * r2 + k] : T(0);
    }
    for (int e = tid; e < AM_TILES * 16; e += nt) {
        const int tt = e >> 4, m = e & 15;
        const int col = am_col(o, D, tt, m & 3, m >> 2);
        sb2[e] = col >= 0 ? b2[col] : T(0);
    }
}

template <typename Args>
__global__ void __launch_bounds__(AM_THREADS) lr_gf_fwd_kernel(const Args a) {
    using T = double;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int D = a.D;
    T* fU2 = reinterpret_cast<T*>(smem_raw);
    T* sb2 = fU2 + a.n_layers * AM_TILES * 2 * 64;
    for (int l = 0; l < a.n_layers; ++l) lr_stage_layer<T>(a.L[l], D, a.r2, a.U2, a.b2, fU2 + l * AM_TILES * 2 * 64, sb2 + l * AM_TILES * 16, tid, (int)blockDim.x);
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * am_rows((int)blockDim.x) + wave * 16 + n;
    const bool row_valid = row < a.B;
    const int64_t rrow = row_valid ? row : a.B - 1;
    const T t2a = q < a.r2 ? a.t2[rrow * a.t2s + q] : T(0);
    const T t2b = q + 4 < a.r2 ? a.t2[rrow * a.t2s + q + 4] : T(0);
    const bool v0 = q < D, v1 = q + 4 < D;
    T x0 = v0 ? a.x[rrow * a.xs + q] : T(0);
    T x1 = v1 ? a.x[rrow * a.xs + q + 4] : T(0);
    T ld = a.ld_in ? a.ld_in[rrow] : T(0);
    auto tile = [&](int layer, int tt) -> f64x4_t {
        const int tl = layer * AM_TILES + tt;
        const T* b = sb2 + tl * 16 + q;
        f64x4_t p = {b[0], b[4], b[8], b[12]};
        p = __builtin_amdgcn_mfma_f64_16x16x4f64(fU2[(2 * tl) * 64 + lane], t2a, p, 0, 0, 0);
        return __builtin_amdgcn_mfma_f64_16x16x4f64(fU2[(2 * tl + 1) * 64 + lane], t2b, p, 0, 0, 0);
    };
    for (int l = a.n_layers - 1; l >= 0; --l) {
        const auto o = a.L[l];
        // aux[l][slot][half][row][q]: every store of a wave covers 512 contiguous bytes
        T* ax = a.aux ? a.aux + (int64_t)l * LR_AUX * 2 * a.B * 4 + row * 4 + q : nullptr;
        const int64_t hst = a.B * 4;                                  // stride between the two coordinate halves; 2 hst between slots
        {
            T R[AM_TILES_R * 4];
#pragma unroll
            for (int tt = 0; tt < AM_TILES_R; ++tt) {
                const f64x4_t p = tile(l, tt);
#pragma unroll
                for (int r = 0; r < 4; ++r) R[4 * tt + r] = p[r];
            }
            x0 -= R[16]; x1 -= R[17];
#pragma unroll
            for (int i = 0; i < AG_HH; ++i) {
                if (i < o.hh) {
                    const T va = R[i], vb = R[8 + i];
                    const T n2 = am_xsum(va * va + vb * vb), dot = am_xsum(va * x0 + vb * x1);
                    const T f = T(2) * dot / n2;
                    x0 -= f * va; x1 -= f * vb;
                }
            }
        }
        if (ax && row_valid) {                                        // slot 0: the coordinates BEHIND the offset and the reflections -- what the
            if (v0) ax[0] = x0;                                       //   mixture stage sees; the adjoint walks the reflections back from here
            if (v1) ax[hst] = x1;
        }
        T logd = T(0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            T P[AG_SLOTS];
#pragma unroll
            for (int tt = 0; tt < AM_TILES_M; ++tt) {
                const f64x4_t p = tile(l, AM_TILES_R + half * AM_TILES_M + tt);
#pragma unroll
                for (int r = 0; r < 4; ++r) if (4 * tt + r < 30) P[4 * tt + r] = p[r];
            }
            const bool live = half == 0 ? v0 : v1;
            MixSums<T> m;
            const MixQ<T> mq = ag_mixture_sums<T>(P, o, half == 0 ? x0 : x1, live, m);
            if (ax && row_valid && live) {
                T* s = ax + half * hst;
                s[2 * hst] = m.C; s[4 * hst] = m.S; s[6 * hst] = m.P; s[8 * hst] = m.invN;
            }
            const IcdfOut<T> s = gf_icdf<T>(o.inv_type, mq);
            if (half == 0) x0 = s.y; else x1 = s.y;
            logd += live ? s.logd : T(0);
        }
        ld += am_xsum(logd);
    }
    if (row_valid && v0) a.x_out[row * a.xos + q] = x0;
    if (row_valid && v1) a.x_out[row * a.xos + q + 4] = x1;
    T sb = T(0);
    if (a.blp_out) sb = am_xsum((v0 ? T(-0.5) * x0 * x0 - M<T>::HALF_LN_2PI : T(0)) + (v1 ? T(-0.5) * x1 * x1 - M<T>::HALF_LN_2PI : T(0)));
    if (row_valid && q == 0) {
        a.ld_out[row] = ld;
        if (a.blp_out) a.blp_out[row] = sb + (a.blp_in ? a.blp_in[row] : T(0));
    }
    const bool badx = (v0 && !M<T>::finite(x0)) || (v1 && !M<T>::finite(x1));
    const T bad = am_xsum(badx ? T(1) : T(0));
    status_add(a.status, JF_STATUS_NONFINITE, row_valid && q == 0 && (bad > T(0) || !M<T>::finite(ld)));
}

// ---- backward: its own tile order, one small tile per unit of work so that the adjoint is a sequence of loops over tiles (generate: 2 products,
// differentiate, contract) with nothing of the parameter row kept in registers:
//   tiles 0..7   reflection i: register 0 / 1 = component q / q + 4 of the Householder vector
//   tile  8      offset of coordinates q / q + 4
//   tiles 9..28  (coordinate half, component k): registers 0..2 = mean, log-width, log-weight of component k of coordinate q + 4 half
constexpr int LR_TILES_R = AG_HH + 1;
static_assert(LR_TILES == LR_TILES_R + 2 * AG_K, "tile count");
template <typename L> __device__ __forceinline__ int lr_col(const L& o, int D, int tt, int q, int reg) {
    if (tt < LR_TILES_R) {
        const int d = q + 4 * reg;
        if (reg > 1 || d >= D) return -1;
        if (tt < AG_HH) return tt < o.hh ? o.col0 + o.off_rot + tt * D + d : -1;
        return o.model_offset ? o.col0 + d : -1;
    }
    const int t = tt - LR_TILES_R, half = t / AG_K, k = t - half * AG_K, d = q + 4 * half;
    if (reg > 2 || d >= D) return -1;
    return o.col0 + (reg == 0 ? o.off_mean : reg == 1 ? o.off_lw : o.off_ln) + k * D + d;
}

__global__ void __launch_bounds__(LR_NW * 64) lr_gf_bwd_layer_kernel(const LrBwdArgs<double> a) {
    using T = double;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int D = a.D;
    T* fU2 = reinterpret_cast<T*>(smem_raw);                 // LR_TILES * 2 fragments
    T* sb2 = fU2 + LR_TILES * 2 * 64;                       // LR_TILES * 16
    T* acc = sb2 + LR_TILES * 16;                           // LR_PSZ: g_U2' | g_b2' of this workgroup's rows, tile order
    T* scr = acc + LR_PSZ + wave * (2 * 16 * 17);           // this wave's two transposition scratch tiles
    const auto o = a.L;
    for (int e = tid; e < LR_TILES * 2 * 64; e += LR_NW * 64) {
        const int f = e >> 6, l = e & 63, mm = l & 15, k = 4 * (f & 1) + (l >> 4), tt = f >> 1;
        const int col = lr_col(o, D, tt, mm & 3, mm >> 2);
        fU2[e] = (col >= 0 && k < a.r2) ? a.U2[(int64_t)col * a.r2 + k] : T(0);
    }
    for (int e = tid; e < LR_TILES * 16; e += LR_NW * 64) {
        const int tt = e >> 4, mm = e & 15;
        const int col = lr_col(o, D, tt, mm & 3, mm >> 2);
        sb2[e] = col >= 0 ? a.b2[col] : T(0);
    }
    for (int e = tid; e < LR_PSZ; e += LR_NW * 64) acc[e] = T(0);
    __syncthreads();
    const bool v0 = q < D, v1 = q + 4 < D;
    const int64_t hst = a.B * 4;                            // aux[l][slot][half][row][q]
    // A operand of U2'^T G for register r of a tile: lane (i = n: rank, k = q) takes U2'[tile row 4 r + k][rank i] out of the forward image,
    // where rank i of tile row mm sits in fragment (i / 4) at lane (i % 4) * 16 + mm
    const int ga = ((n >> 2) & 1) * 64 + (n & 3) * 16 + q;
    const bool ga_ok = n < AM_R;
    const bool acc_lane = n <= a.r2;
    for (int64_t rt = (int64_t)blockIdx.x * LR_NW + wave; rt < a.n_row_tiles; rt += (int64_t)gridDim.x * LR_NW) {
        const int64_t row = rt * 16 + n;
        const bool row_valid = row < a.B;
        const int64_t rrow = row_valid ? row : a.B - 1;
        const T t2a = q < a.r2 ? a.t2[rrow * a.t2s + q] : T(0);
        const T t2b = q + 4 < a.r2 ? a.t2[rrow * a.t2s + q + 4] : T(0);
        T tb[4];                                            // B operands of G [t2 | 1]: lane (j = n, k = q) of step s: row 4 s + k, column j
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t rr = rt * 16 + 4 * s + q;
            const bool okr = rr < a.B;
            const T v = (okr && n < a.r2) ? a.t2[rr * a.t2s + n] : T(0);
            tb[s] = (okr && n == a.r2) ? T(1) : v;
        }
        const T* ax = a.aux + (int64_t)a.layer * LR_AUX * 2 * a.B * 4 + rrow * 4 + q;
        T x0 = v0 ? ax[0] : T(0), x1 = v1 ? ax[hst] : T(0);
        const T gl = (a.g_ld && row_valid) ? a.g_ld[rrow] : T(0);
        T gy0, gy1;
        if (a.first) {
            gy0 = (a.g_xout && v0) ? a.g_xout[rrow * a.gxos + q] : T(0);
            gy1 = (a.g_xout && v1) ? a.g_xout[rrow * a.gxos + q + 4] : T(0);
            if (a.g_blp) {                                  // base log-prob = sum_d -x^2 / 2 - ...: d / dx_out = -x_out
                const T gb = a.g_blp[rrow];
                if (v0) gy0 -= a.x_out[rrow * a.xos + q] * gb;
                if (v1) gy1 -= a.x_out[rrow * a.xos + q + 4] * gb;
            }
        } else {
            gy0 = v0 ? a.g_x[rrow * a.gxs + q] : T(0);
            gy1 = v1 ? a.g_x[rrow * a.gxs + q + 4] : T(0);
        }
        if (!row_valid) { gy0 = T(0); gy1 = T(0); }
        const bool w0 = v0 && row_valid, w1 = v1 && row_valid;

        auto tile = [&](int tt) -> f64x4_t {
            const T* b = sb2 + tt * 16 + q;
            f64x4_t p = {b[0], b[4], b[8], b[12]};
            p = __builtin_amdgcn_mfma_f64_16x16x4f64(fU2[(2 * tt) * 64 + lane], t2a, p, 0, 0, 0);
            return __builtin_amdgcn_mfma_f64_16x16x4f64(fU2[(2 * tt + 1) * 64 + lane], t2b, p, 0, 0, 0);
        };
        // g_t2^T of this layer (registers 0, 1 = ranks q, q + 4 of row n): two accumulators so that the products of two tiles interleave
        f64x4_t dta = {0.0, 0.0, 0.0, 0.0}, dtb = {0.0, 0.0, 0.0, 0.0};
        // the gradient tiles GA / GB (register r <-> tile row q + 4 r, column = row n; NR registers in use) of tiles ta / tb2: both contractions
        // of both tiles, interleaved (independent chains).  Rows of the scratch tile a product does not write are garbage that only reaches
        // result registers >= NR, which nobody reads.
        auto contract2 = [&](int ta, const f64x4_t GA, int tb2, const f64x4_t GB, const int NR, const bool two) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < NR) {
                    const T ua = ga_ok ? fU2[(2 * ta) * 64 + ga + 4 * r] : T(0);
                    dta = __builtin_amdgcn_mfma_f64_16x16x4f64(ua, GA[r], dta, 0, 0, 0);
                    scr[(q + 4 * r) * 17 + n] = GA[r];
                    if (two) {
                        const T ub = ga_ok ? fU2[(2 * tb2) * 64 + ga + 4 * r] : T(0);
                        dtb = __builtin_amdgcn_mfma_f64_16x16x4f64(ub, GB[r], dtb, 0, 0, 0);
                        scr[16 * 17 + (q + 4 * r) * 17 + n] = GB[r];
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f64x4_t wa = {0.0, 0.0, 0.0, 0.0}, wb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                wa = __builtin_amdgcn_mfma_f64_16x16x4f64(scr[n * 17 + 4 * s + q], tb[s], wa, 0, 0, 0);
                if (two) wb = __builtin_amdgcn_mfma_f64_16x16x4f64(scr[16 * 17 + n * 17 + 4 * s + q], tb[s], wb, 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (acc_lane) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (r < NR) {
                    atomicAdd(acc + (ta * 16 + q + 4 * r) * LR_RS + n, wa[r]);
                    if (two) atomicAdd(acc + (tb2 * 16 + q + 4 * r) * LR_RS + n, wb[r]);
                }
            }
        };

        // ---- mixture + inverse-CDF stage of the two coordinates: linear-space responsibilities (gf_layer_bwd_fast, gf_bwd_kernels.hip) when every
        //      live lane of the wave is in their range, else log space (gf_layer_bwd) -- the switch of the (B, P)-block kernel on the same saved sums
        T gx0 = T(0), gx1 = T(0);
        {
            const T C0 = v0 ? ax[2 * hst] : T(0.5), S0 = v0 ? ax[4 * hst] : T(0.5), P0 = v0 ? ax[6 * hst] : T(0.1), N0 = v0 ? ax[8 * hst] : T(1);
            const T C1 = v1 ? ax[3 * hst] : T(0.5), S1 = v1 ? ax[5 * hst] : T(0.5), P1 = v1 ? ax[7 * hst] : T(0.1), N1 = v1 ? ax[9 * hst] : T(1);
            const T gl0 = w0 ? gl : T(0), gl1 = w1 ? gl : T(0);
            const bool ok = (!w0 || (C0 > LinRange<T>::lo && S0 > LinRange<T>::lo && P0 > LinRange<T>::lo && P0 < LinRange<T>::hi)) &&
                            (!w1 || (C1 > LinRange<T>::lo && S1 > LinRange<T>::lo && P1 > LinRange<T>::lo && P1 < LinRange<T>::hi));
            if (__all(ok)) {
                T icg[2], isg[2], ipg[2], Gs[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const T C = h ? C1 : C0, S = h ? S1 : S0, Pd = h ? P1 : P0, gy = h ? gy1 : gy0, glc = h ? gl1 : gl0;
                    MixQ<T> mq;
                    mq.lc = M<T>::log_fast(C); mq.ls = M<T>::log_fast(S); mq.lp = M<T>::log_fast(Pd); mq.cdf = C; mq.sf = S;
                    const IcdfOut<T> s = gf_icdf<T>(o.inv_type, mq);
                    const IcdfCoef<T> c = gf_icdf_coeffs<T>(o.inv_type, mq, s.y);
                    const T g_lc = gy * c.Ay + glc * c.AH, g_ls = gy * c.By + glc * c.BH, g_lp = glc;
                    Gs[h] = g_lc + g_ls + g_lp;
                    icg[h] = g_lc * M<T>::rcp(C); isg[h] = g_ls * M<T>::rcp(S); ipg[h] = g_lp * M<T>::rcp(Pd);
                }
#pragma unroll 1
                for (int k = 0; k < AG_K; ++k) {
                    f64x4_t G[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f64x4_t p = tile(LR_TILES_R + h * AG_K + k);
                        const T x = h ? x1 : x0, invN = h ? N1 : N0;
                        const bool live = h ? w1 : w0;
                        const T mu = p[0], rw = p[1], rn = p[2];
                        const T e = M<T>::exp_fast(-rw);
                        const T ae = o.inv_wmax + e;
                        const T r2 = M<T>::rcp(ae * (o.wmin * ae + T(1)));
                        const T iw = ae * ae * r2, dliw = -e * r2;
                        const T sgn = M<T>::rcp(T(1) + M<T>::exp_fast(-rn));
                        const T pik = (o.nmin + o.nmax * sgn) * invN;
                        const T u = (x - mu) * iw;
                        const T t = M<T>::exp_fast(-M<T>::abs(u));
                        const T hi = M<T>::rcp(T(1) + t), lo = t * hi;
                        const bool pos = u >= T(0);
                        const T sg = pos ? hi : lo, sgc = pos ? lo : hi;
                        const T aa = sg * icg[h], bb = sgc * isg[h], cp = sg * sgc * iw * ipg[h];
                        const T gu = pik * (aa * sgc - bb * sg + cp * (sgc - sg));
                        if (h) gx1 += gu * iw; else gx0 += gu * iw;
                        G[h][0] = live ? -gu * iw : T(0);
                        G[h][1] = live ? (gu * u + pik * cp) * dliw : T(0);
                        G[h][2] = live ? (aa + bb + cp - Gs[h]) * (o.nmax * sgn * (T(1) - sgn) * invN) : T(0);
                        G[h][3] = T(0);
                    }
                    contract2(LR_TILES_R + k, G[0], LR_TILES_R + AG_K + k, G[1], 3, true);
                }
            } else {
#pragma unroll 1
                for (int half = 0; half < 2; ++half) {
                    const bool live = half ? w1 : w0;
                    const T x = half ? x1 : x0, gy = half ? gy1 : gy0, glc = half ? gl1 : gl0;
                    const int t0 = LR_TILES_R + half * AG_K;
                    // scaled evaluation of the mixture (ag_mixture_scaled): two passes over the component tiles
                    T mmin = T(INFINITY);
#pragma unroll 1
                    for (int k = 0; k < AG_K; ++k) {
                        const f64x4_t p = tile(t0 + k);
                        const T ae = o.inv_wmax + M<T>::exp(-p[1]);
                        mmin = M<T>::min(mmin, M<T>::abs((x - p[0]) * (ae / (o.wmin * ae + T(1)))));
                    }
                    const T em = M<T>::exp(-mmin);
                    T Cu = T(0), Cs = T(0), Su = T(0), Ss = T(0), Ps = T(0), Nn = T(0);
#pragma unroll 1
                    for (int k = 0; k < AG_K; ++k) {
                        const f64x4_t p = tile(t0 + k);
                        const T ae = o.inv_wmax + M<T>::exp(-p[1]);
                        const T iw = ae / (o.wmin * ae + T(1));
                        const T u = (x - p[0]) * iw;
                        const T wk = o.nmin + o.nmax / (T(1) + M<T>::exp(-p[2]));
                        const T t = M<T>::exp(mmin - M<T>::abs(u));
                        const T hi = T(1) / (T(1) + t * em);
                        const T c1 = wk * hi, c2 = c1 * t;
                        if (u >= T(0)) { Cu += c1; Ss += c2; }
                        else { Su += c1; Cs += c2; }
                        Ps += c2 * hi * iw;
                        Nn += wk;
                    }
                    const T inv = T(1) / Nn;
                    Cu *= inv; Cs *= inv; Su *= inv; Ss *= inv; Ps *= inv;
                    MixQ<T> mq;
                    mq.cdf = Cu + em * Cs;
                    mq.sf = Su + em * Ss;
                    mq.lc = Cu > T(0) ? M<T>::log(mq.cdf) : M<T>::log(Cs) - mmin;
                    mq.ls = Su > T(0) ? M<T>::log(mq.sf) : M<T>::log(Ss) - mmin;
                    mq.lp = M<T>::log(Ps) - mmin;
                    const IcdfOut<T> s = gf_icdf<T>(o.inv_type, mq);
                    const IcdfCoef<T> c = gf_icdf_coeffs<T>(o.inv_type, mq, s.y);
                    const T g_lc = gy * c.Ay + glc * c.AH, g_ls = gy * c.By + glc * c.BH, g_lp = glc;
                    const T Gsum = g_lc + g_ls + g_lp;
                    const T lN = M<T>::log(Nn);
                    T gx = T(0);
#pragma unroll 1
                    for (int k = 0; k < AG_K; ++k) {
                        const f64x4_t p = tile(t0 + k);
                        const T mu = p[0], rw = p[1], rn = p[2];
                        const T e = M<T>::exp(-rw);
                        const T ae = o.inv_wmax + e;
                        const T den = o.wmin * ae + T(1);
                        const T iw = ae / den, dliw = -e / (ae * den);
                        const T sgn = T(1) / (T(1) + M<T>::exp(-rn));
                        const T nk = o.nmin + o.nmax * sgn;
                        const T lpi = M<T>::log(nk) - lN, pik = nk * inv, dlnn = o.nmax * sgn * (T(1) - sgn) / nk;
                        const T u = (x - mu) * iw;
                        const T t = M<T>::exp(-M<T>::abs(u));
                        const T hi = T(1) / (T(1) + t), lo = t * hi;
                        const bool pos = u >= T(0);
                        const T sg = pos ? hi : lo, sgc = pos ? lo : hi;
                        const T l1p = M<T>::log1p(t);
                        const T lsp = (pos ? T(0) : u) - l1p, lsm = (pos ? -u : T(0)) - l1p;
                        const T rC = M<T>::exp(lpi + lsp - mq.lc), rS = M<T>::exp(lpi + lsm - mq.ls), rP = M<T>::exp(lpi + lsp + lsm + M<T>::log(iw) - mq.lp);
                        const T gu = g_lc * rC * sgc - g_ls * rS * sg + g_lp * rP * (sgc - sg);
                        gx += gu * iw;
                        f64x4_t G;
                        G[0] = live ? -gu * iw : T(0);
                        G[1] = live ? (gu * u + g_lp * rP) * dliw : T(0);
                        G[2] = live ? ((g_lc * rC + g_ls * rS + g_lp * rP) - pik * Gsum) * dlnn : T(0);
                        G[3] = T(0);
                        contract2(t0 + k, G, 0, G, 3, false);
                    }
                    if (half) gx1 = gx; else gx0 = gx;
                }
            }
            if (!w0) gx0 = T(0);
            if (!w1) gx1 = T(0);
        }
        // ---- reflections, last first (y = x - c v, c = 2 (v.x) / (v.v):  g_x = H g,  g_v = -c g - (2 (v.g) / n) x + (4 (v.x)(v.g) / n^2) v), the
        //      vector in front of each reflection brought back by applying it again (H is an involution; v . x_before = -v . x_after); the offset
        T g0 = gx0, g1 = gx1;
#pragma unroll 1
        for (int i = o.hh - 1; i >= 0; --i) {
            const f64x4_t p = tile(i);
            const T va = p[0], vb = p[1];
            const T nn = am_xsum(va * va + vb * vb), da = am_xsum(va * x0 + vb * x1), vg = am_xsum(va * g0 + vb * g1);
            const T rn = M<T>::rcp(nn);
            x0 -= T(2) * da * rn * va; x1 -= T(2) * da * rn * vb;
            const T sx = -da;
            f64x4_t G;
            G[0] = w0 ? -T(2) * sx * rn * g0 - T(2) * vg * rn * x0 + T(4) * sx * vg * rn * rn * va : T(0);
            G[1] = w1 ? -T(2) * sx * rn * g1 - T(2) * vg * rn * x1 + T(4) * sx * vg * rn * rn * vb : T(0);
            G[2] = T(0); G[3] = T(0);
            g0 -= T(2) * vg * rn * va; g1 -= T(2) * vg * rn * vb;
            contract2(i, G, 0, G, 2, false);
        }
        if (o.model_offset) {
            f64x4_t G;
            G[0] = w0 ? -g0 : T(0); G[1] = w1 ? -g1 : T(0); G[2] = T(0); G[3] = T(0);
            contract2(AG_HH, G, 0, G, 2, false);
        }
        if (w0) a.g_x[row * a.gxs + q] = g0;
        if (w1) a.g_x[row * a.gxs + q + 4] = g1;
        {
            const bool badg = (w0 && !M<T>::finite(g0)) || (w1 && !M<T>::finite(g1));
            const T bad = am_xsum(badg ? T(1) : T(0));
            status_add(a.status, JF_STATUS_NONFINITE, row_valid && q == 0 && bad > T(0));
        }
        if (row_valid) {
            T* gt = a.g_t2 + row * AM_R;
            const T d0 = dta[0] + dtb[0], d1 = dta[1] + dtb[1];
            if (a.first) { gt[q] = d0; gt[q + 4] = d1; }
            else { gt[q] += d0; gt[q + 4] += d1; }
        }
    }
    __syncthreads();
    T* out = a.partial + (int64_t)blockIdx.x * LR_PSZ;
    for (int e = tid; e < LR_PSZ; e += LR_NW * 64) out[e] = acc[e];
}

struct LrReduceArgs {
    const double* partial; int n_wg, n_layers, D, r2;
    AgLayer<double> L[JF_MAX_CHAIN];
    double* g_U2; double* g_b2;
};
// 64 entries x 4 slices of the partial images per workgroup: slice p sums images p, p + 4, ... (8 loads in flight), the slices meet in LDS --
// always the same order of additions
__global__ void __launch_bounds__(256) lr_reduce_kernel(const LrReduceArgs a) {
    __shared__ double part[4][64];
    const int le = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + le;
    const bool in = e < a.n_layers * LR_PSZ;
    const int l = in ? e / LR_PSZ : 0, w = in ? e - l * LR_PSZ : 0;
    const double* p = a.partial + (int64_t)l * a.n_wg * LR_PSZ + w;
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (in) {
        int g = sl;
        for (; g + 28 < a.n_wg; g += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += p[(int64_t)(g + 4 * u) * LR_PSZ];
        }
        for (; g < a.n_wg; g += 4) s[0] += p[(int64_t)g * LR_PSZ];
    }
    part[sl][le] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (sl != 0 || !in) return;
    const double sum = (part[0][le] + part[1][le]) + (part[2][le] + part[3][le]);
    const int tt = w / (16 * LR_RS), m = (w / LR_RS) & 15, j = w % LR_RS;
    const int col = lr_col(a.L[l], a.D, tt, m & 3, m >> 2);
    if (col < 0 || j > a.r2) return;
    if (j < a.r2) a.g_U2[(int64_t)col * a.r2 + j] = sum;
    else a.g_b2[col] = sum;
}

inline int lr_n_wg(int64_t B) {
    const int64_t tiles = (B + 15) / 16, wgs = (tiles + LR_NW - 1) / LR_NW;
    return (int)(wgs < LR_MAX_WG ? (wgs < 1 ? 1 : wgs) : LR_MAX_WG);
}

}  // namespace jf
