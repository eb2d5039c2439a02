// Conditional e-block whose parameters come from a LOW-RANK AmortizableMLP, in ONE launch (BASELINE configs[4]: e8 / gggg, 16 conditioning
// inputs, hidden 128, rank 8, float64):
//     params = U2 (V2 tanh(U1 (V1 c) + b1)) + b2          (amortizable_mlp.py:508-578, two low-rank stages of rank r1, r2)
//     followed by the layer loop of all_layer_inverse on those per-sample parameters (main/default.py:998-1031).
//
// With a rank-r2 last stage every parameter of a row is  b2[j] + <U2[j, :], t2>  with ONE r2-vector t2 per row: 8 fused multiply-adds
// regenerate a parameter, against 8 bytes of HBM write + 8 bytes of HBM read when the (B, 1224) block is materialised (9.8 KB per row each
// way -- the two launches that did that took 2.1 + 1.4 ms per 2^19 rows).  So nothing of the block ever exists: the kernel computes t2 per
// row (the four small products, cooperatively by the G lanes of a row) and every lane generates the ~39 parameters of its coordinate per
// layer straight into registers from U2 / b2, which sit in LDS for the whole workgroup (1224 x 8 doubles = 78 KB).  No matrix cores: the
// contraction length is 8.  HBM traffic: K1 + 2 D + 2 scalars per row.
// Work distribution as in jf_gf.h: lane = (row, coordinate), G = 4 or 8 lanes per row, DPP group reductions; 512-thread workgroups.
//
// Supported: 2-stage AmortizableMLP (highway_mode 0), first stage full or low-rank (r1 <= 16, K1 <= 32), hidden width <= 128 (multiple
// of G), second stage low-rank r2 <= 16; layers with the reference's default options (K = 10, <= 8 Householder reflections); D <= 8.
#include "jf_gf.h"

namespace jf {

constexpr int AG_K = 10, AG_HH = 8;
constexpr int AG_MEAN = 0, AG_LW = AG_K, AG_LN = 2 * AG_K, AG_ROT = 3 * AG_K, AG_OFF = 3 * AG_K + AG_HH, AG_SLOTS = 3 * AG_K + AG_HH + 1;
constexpr int AG_HMAX = 128, AG_RMAX = 16, AG_K1MAX = 32;
constexpr int AG_THREADS = 768;                      // 12 waves share one LDS copy of the weights: 3 waves per SIMD (165 VGPRs in float64)

template <typename T> struct AgLayer { int hh, model_offset, inv_type, col0, off_rot, off_mean, off_lw, off_ln; T wmin, inv_wmax, nmin, nmax; };

template <typename T> struct AgArgs {
    const T* in; int64_t in_stride;
    const T* V1; const T* U1; const T* b1; const T* V2; const T* U2; const T* b2;    // V1 == nullptr: first stage is a full matrix U1 (H x K1)
    int K1, H, r1, r2, N;
    const T* x; int64_t xs;
    const T* ld_in;
    int64_t B;
    int D, n_layers;
    AgLayer<T> L[JF_MAX_CHAIN];
    T* x_out; int64_t xos;
    T* ld_out;
    const T* blp_in; T* blp_out;
    int32_t* status;
    T* params_out; int64_t pos;           // non-null: MLP only -- write the (B, N) parameter block and skip the flow (jf_amlp2)
};

// mixture on a register row (cf. gfg_mixture_impl<T, RAW, FAST> / gfg_mixture_scaled in jf_gf.h)
template <typename T> __device__ __forceinline__ MixQ<T> ag_mixture_scaled(const T (&P)[AG_SLOTS], const AgLayer<T>& o, T x) {
    T iw[AG_K], u[AG_K];
    T m = T(INFINITY);
#pragma unroll
    for (int k = 0; k < AG_K; ++k) {
        const T ae = o.inv_wmax + M<T>::exp_fast(-P[AG_LW + k]);
        iw[k] = ae * M<T>::rcp(o.wmin * ae + T(1));
        u[k] = (x - P[AG_MEAN + k]) * iw[k];
        m = M<T>::min(m, M<T>::abs(u[k]));
    }
    const T em = M<T>::exp_fast(-m);
    T Cu = T(0), Cs = T(0), Su = T(0), Ss = T(0), Ps = T(0), Nn = T(0);
#pragma unroll
    for (int k = 0; k < AG_K; ++k) {
        const T wk = o.nmin + o.nmax * M<T>::rcp(T(1) + M<T>::exp_fast(-P[AG_LN + k]));
        const T t = M<T>::exp_fast(m - M<T>::abs(u[k]));
        const T hi = M<T>::rcp(T(1) + t * em);
        const T c1 = wk * hi, c2 = c1 * t;
        if (u[k] >= T(0)) { Cu += c1; Ss += c2; }
        else { Su += c1; Cs += c2; }
        Ps += c2 * hi * iw[k];
        Nn += wk;
    }
    const T inv = M<T>::rcp(Nn);
    Cu *= inv; Cs *= inv; Su *= inv; Ss *= inv; Ps *= inv;
    MixQ<T> q;
    q.cdf = Cu + em * Cs;
    q.sf = Su + em * Ss;
    q.lc = Cu > T(0) ? M<T>::log_fast(q.cdf) : M<T>::log_fast(Cs) - m;
    q.ls = Su > T(0) ? M<T>::log_fast(q.sf) : M<T>::log_fast(Ss) - m;
    q.lp = M<T>::log_fast(Ps) - m;
    return q;
}

template <typename T> __device__ __forceinline__ MixQ<T> ag_mixture(const T (&P)[AG_SLOTS], const AgLayer<T>& o, T x, bool live) {
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
    MixQ<T> q;
    q.lc = M<T>::log_fast(C); q.ls = M<T>::log_fast(S); q.lp = M<T>::log_fast(Pd);
    q.cdf = C; q.sf = S;
    const bool under = live && !(C > M<T>::TINY && S > M<T>::TINY && Pd > M<T>::TINY);
    if (__any(under)) {
        const MixQ<T> qs = ag_mixture_scaled<T>(P, o, x);
        if (under) q = qs;
    }
    return q;
}

// lane `src` of the caller's G-lane row group
template <typename T, int G> __device__ __forceinline__ T group_bcast(T v, int src) { return __shfl(v, (threadIdx.x & 63 & ~(G - 1)) + src, 64); }

}  // namespace jf
#include "jf_amlp_mfma.h"      // the float64 matrix-core version of the same block (needs AgArgs / ag_mixture above)
#include "jf_lowrank_gf.h"     // training: the chain on a low-rank last stage, forward with saved layer inputs + per-layer adjoint launches
#include "jf_lowrank_mlp.h"    // training: the MLP in front of that stage, forward and backward in one launch each
namespace jf {

// RM: compiled rank bound (8 or 16): the rank loops are fully unrolled over it
// MLP_ONLY: the jf_amlp2 instantiation (writes the parameter block, no flow) -- a kernel of its own so that profiles tell the two apart
template <typename T, int G, int RM, bool MLP_ONLY>
__global__ void __launch_bounds__(AG_THREADS) amlp_gf_kernel(const AgArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* sV1 = reinterpret_cast<T*>(smem_raw);                       // r1 x K1   (or H x K1 when the first stage is full)
    const int n1 = a.V1 ? a.r1 * a.K1 : a.H * a.K1;
    T* sU1 = sV1 + n1;                                             // H x r1    (unused when full)
    T* sb1 = sU1 + (a.V1 ? a.H * a.r1 : 0);                        // H
    T* sV2 = sb1 + a.H;                                            // r2 x H
    T* sU2 = sV2 + a.r2 * a.H;                                     // N x r2, rows padded to an odd length `us`: the G lanes of a row read G
    const int us = a.r2 | 1;                                       //   consecutive rows at the same column -> G distinct banks
    T* sb2 = sU2 + (int64_t)a.N * us;                              // N
    const int tid = threadIdx.x;
    {   // weights -> LDS once per workgroup
        const T* src1 = a.V1 ? a.V1 : a.U1;
        for (int i = tid; i < n1; i += AG_THREADS) sV1[i] = src1[i];
        if (a.V1) for (int i = tid; i < a.H * a.r1; i += AG_THREADS) sU1[i] = a.U1[i];
        for (int i = tid; i < a.H; i += AG_THREADS) sb1[i] = a.b1[i];
        for (int i = tid; i < a.r2 * a.H; i += AG_THREADS) sV2[i] = a.V2[i];
        for (int i = tid; i < a.N * a.r2; i += AG_THREADS) sU2[(i / a.r2) * us + i % a.r2] = a.U2[i];
        for (int i = tid; i < a.N; i += AG_THREADS) sb2[i] = a.b2[i];
    }
    __syncthreads();
    constexpr int LG = G == 4 ? 2 : 3;
    constexpr int R = AG_THREADS / G;                               // rows per workgroup
    const int g = tid & (G - 1), r = tid >> LG;
    const int D = a.D;
    const bool live = g < D, leader = g == 0;
    const int d = live ? g : D - 1;
    const int64_t row = (int64_t)blockIdx.x * R + r;
    const bool row_valid = row < a.B;
    const int64_t rrow = row_valid ? row : a.B - 1;

    // ---- the row's rank-r2 vector t2 = V2 tanh(W1 c + b1), computed by the G lanes of the row (lane g: hidden units g, g + G, ...)
    T t2[RM];
    {
        const T* c = a.in + rrow * a.in_stride;
        // the row's K1 inputs in registers; every loop over them is fully unrolled (a runtime-indexed copy would live in scratch memory, and
        // reading c[i] from global memory inside the loops serialised ~190 L1 round trips per lane)
        T cin[AG_K1MAX];
#pragma unroll
        for (int i = 0; i < AG_K1MAX; ++i) cin[i] = i < a.K1 ? c[i] : T(0);
        T t1[RM];
        if (a.V1) {                                                // lane g computes t1[g], t1[g + G], ...; the group then exchanges them
            constexpr int PER = RM / G > 0 ? RM / G : 1;
            T mine[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int q = g + u * G;
                const int qq = q < a.r1 ? q : 0;
                T acc = T(0);
#pragma unroll
                for (int i = 0; i < AG_K1MAX; ++i) if (i < a.K1) acc += sV1[qq * a.K1 + i] * cin[i];
                mine[u] = acc;
            }
#pragma unroll
            for (int q = 0; q < RM; ++q) t1[q] = group_bcast<T, G>(mine[(q / G) < PER ? q / G : 0], q % G);
        }
#pragma unroll
        for (int q = 0; q < RM; ++q) t2[q] = T(0);
        // four hidden units per iteration: their pre-activation chains and tanh evaluations are independent, which is the instruction-level
        // parallelism a 2-waves-per-SIMD kernel needs (one unit at a time ran 2.4x slower)
        for (int j0 = g; j0 < a.H; j0 += 4 * G) {
            T pre[4];
            int jj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * G;
                jj[u] = j < a.H ? j : a.H - 1;
                pre[u] = sb1[jj[u]];
            }
            if (a.V1) {
#pragma unroll
                for (int q = 0; q < RM; ++q) if (q < a.r1) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) pre[u] += sU1[jj[u] * a.r1 + q] * t1[q];
                }
            } else {
#pragma unroll
                for (int i = 0; i < AG_K1MAX; ++i) if (i < a.K1) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) pre[u] += sV1[jj[u] * a.K1 + i] * cin[i];
                }
            }
            T h[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) h[u] = (j0 + u * G < a.H) ? M<T>::tanh_fast(pre[u]) : T(0);
#pragma unroll
            for (int q = 0; q < RM; ++q) if (q < a.r2) {
#pragma unroll
                for (int u = 0; u < 4; ++u) t2[q] += sV2[q * a.H + jj[u]] * h[u];
            }
        }
#pragma unroll
        for (int q = 0; q < RM; ++q) if (q < a.r2) t2[q] = group_sum<T, G>(t2[q]);
    }

    auto gen = [&](int col) -> T {                                 // parameter `col` of this row: b2[col] + <U2[col, :], t2>
        const T* u = sU2 + (int64_t)col * us;
        T acc = sb2[col];
#pragma unroll
        for (int q = 0; q < RM; ++q) if (q < a.r2) acc += u[q] * t2[q];
        return acc;
    };
    if constexpr (MLP_ONLY) {                                      // the G lanes of a row write its N outputs
        if (row_valid) for (int j = g; j < a.N; j += G) a.params_out[row * a.pos + j] = gen(j);
        return;
    }
    T x = a.x[rrow * a.xs + d];
    T ld = a.ld_in ? a.ld_in[rrow] : T(0);
    for (int l = a.n_layers - 1; l >= 0; --l) {
        const AgLayer<T> o = a.L[l];
        T P[AG_SLOTS];
#pragma unroll
        for (int k = 0; k < AG_K; ++k) {
            P[AG_MEAN + k] = gen(o.col0 + o.off_mean + k * D + d);
            P[AG_LW + k] = gen(o.col0 + o.off_lw + k * D + d);
            P[AG_LN + k] = gen(o.col0 + o.off_ln + k * D + d);
        }
#pragma unroll
        for (int i = 0; i < AG_HH; ++i) P[AG_ROT + i] = i < o.hh ? gen(o.col0 + o.off_rot + i * D + d) : T(0);
        P[AG_OFF] = o.model_offset ? gen(o.col0 + d) : T(0);
        x -= P[AG_OFF];
#pragma unroll
        for (int i = 0; i < AG_HH; ++i) {
            if (i < o.hh) {
                const T v = live ? P[AG_ROT + i] : T(0);
                const T n2 = group_sum<T, G>(v * v), dot = group_sum<T, G>(v * x);
                x -= T(2) * dot * M<T>::rcp(n2) * v;
            }
        }
        const MixQ<T> q = ag_mixture<T>(P, o, x, live);
        const IcdfOut<T> s = gf_icdf<T>(o.inv_type, q);
        x = s.y;
        ld += group_sum<T, G>(live ? s.logd : T(0));
    }
    if (row_valid && live) a.x_out[row * a.xos + d] = x;
    T sb = T(0);
    if (a.blp_out) sb = group_sum<T, G>(live ? T(-0.5) * x * x - M<T>::HALF_LN_2PI : T(0));
    if (row_valid && leader) {
        a.ld_out[row] = ld;
        if (a.blp_out) a.blp_out[row] = sb + (a.blp_in ? a.blp_in[row] : T(0));
    }
    const T bad = group_max<T, G>((live && !M<T>::finite(x)) ? T(1) : T(0));
    status_add(a.status, JF_STATUS_NONFINITE, row_valid && leader && (bad > T(0) || !M<T>::finite(ld)));
}

template <typename T> static int ag_launch(const AgArgs<T>& a, int D, hipStream_t st) {
    const size_t elems = (size_t)(a.V1 ? a.r1 * a.K1 + a.H * a.r1 : a.H * a.K1) + a.H + (size_t)a.r2 * a.H + (size_t)a.N * (a.r2 | 1) + a.N;
    const size_t lds = elems * sizeof(T);
    if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;
    const int G = D <= 4 ? 4 : 8;
    const bool small = a.r2 <= 8 && a.r1 <= 8;
    const unsigned grid = (unsigned)((a.B + AG_THREADS / G - 1) / (AG_THREADS / G));
#define JF_AG_GO(G_, RM_)                                                                                                      \
    {                                                                                                                          \
        auto k = a.params_out ? amlp_gf_kernel<T, G_, RM_, true> : amlp_gf_kernel<T, G_, RM_, false>;                            \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
        jf::launch(k, dim3(grid), dim3(AG_THREADS), lds, st, a);                                                         \
    }
    if (G == 4) { if (small) JF_AG_GO(4, 8) else JF_AG_GO(4, 16) }
    else { if (small) JF_AG_GO(8, 8) else JF_AG_GO(8, 16) }
#undef JF_AG_GO
    return check_launch();
}

// the two-stage low-rank MLP alone, one launch: out (B, N) = U2 (V2 tanh(W1 in + b1)) + b2
template <typename T>
static int amlp2(const T* in, int64_t in_stride, const T* V1, const T* U1, const T* b1, const T* V2, const T* U2, const T* b2, int64_t B, int32_t K1,
                 int32_t H, int32_t r1, int32_t r2, int32_t N, T* out, int64_t out_stride, void* stream) {
    if (!in || !U1 || !b1 || !V2 || !U2 || !b2 || !out) return JF_ERR_BADARG;
    if (!width_ok(K1) || !width_ok(H) || !width_ok(r2) || !width_ok(N) || !rows_ok(B) || (V1 && r1 < 1)) return JF_ERR_BADARG;
    if (K1 > AG_K1MAX || H > AG_HMAX || r2 > AG_RMAX || (V1 && r1 > AG_RMAX)) return JF_ERR_UNSUPPORTED;
    if (B == 0) return JF_OK;
    AgArgs<T> a{};
    a.in = in; a.in_stride = in_stride; a.V1 = V1; a.U1 = U1; a.b1 = b1; a.V2 = V2; a.U2 = U2; a.b2 = b2;
    a.K1 = K1; a.H = H; a.r1 = V1 ? r1 : 0; a.r2 = r2; a.N = N; a.B = B; a.D = 8; a.n_layers = 0;
    a.params_out = out; a.pos = out_stride;
    if constexpr (sizeof(T) == 8) {                                // float64, ranks <= 8: matrix-core version (jf_amlp_mfma.h)
        const size_t lds = am_lds_mlp_only(K1, H, V1 != nullptr, N) * sizeof(T);
        if (r2 <= AM_R && (!V1 || r1 <= AM_R) && H % 16 == 0 && lds <= 160 * 1024) {
            auto k = amlp2_mfma_kernel<AgArgs<T>>;
            if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            jf::launch(k, dim3((unsigned)((B + am_rows(AM_THREADS) - 1) / am_rows(AM_THREADS))), dim3(AM_THREADS), lds, (hipStream_t)stream, a);
            return check_launch();
        }
    }
    return ag_launch<T>(a, 8, (hipStream_t)stream);
}

// the chain's layer table (offsets of every layer's sections inside the parameter row) -> L; returns the row length or JF_ERR_UNSUPPORTED
template <typename T> static int ag_make_layers(const jf_gf_layer* layers, int n_layers, int D, AgLayer<T>* L) {
    int col = 0;
    for (int l = 0; l < n_layers; ++l) {
        const jf_gf_layer& h = layers[l];
        AgLayer<T>& o = L[l];
        if (h.num_kde != AG_K || h.hh_iter < 0 || h.hh_iter > AG_HH || h.nonlinear_stretch_type != JF_GF_STRETCH_CLASSIC ||
            h.width_mode != JF_GF_WIDTH_SMOOTH_SATURATION || h.clamp_widths || !h.fit_normalization || !h.regulate_normalization ||
            h.width_min <= 0 || h.width_max <= 0 || h.rotation_mode != JF_GF_ROT_HOUSEHOLDER || h.center_mean || h.add_skewness)
            return JF_ERR_UNSUPPORTED;
        const int kd = h.num_kde * D;
        o.hh = h.hh_iter; o.model_offset = h.model_offset; o.inv_type = h.inverse_function_type; o.col0 = col;
        o.off_rot = h.model_offset ? D : 0;
        o.off_mean = o.off_rot + h.hh_iter * D;
        o.off_lw = o.off_mean + kd;
        o.off_ln = o.off_lw + kd;
        o.wmin = (T)h.width_min; o.inv_wmax = (T)(1.0 / h.width_max); o.nmin = (T)h.norm_min; o.nmax = (T)h.norm_max;
        col += o.off_ln + kd;
    }
    return col;
}

template <typename T, bool FWD = false>
static int amlp_gf_chain_inv(const T* in, int64_t in_stride, const T* V1, const T* U1, const T* b1, const T* V2, const T* U2, const T* b2, int32_t K1,
                             int32_t H, int32_t r1, int32_t r2, const T* x, int64_t xs, const T* ld_in, int64_t B, int32_t D, int32_t n_layers,
                             const jf_gf_layer* layers, T* x_out, int64_t xos, T* ld_out, const T* blp_in, T* blp_out, int32_t* status, void* stream) {
    if (!in || !U1 || !b1 || !V2 || !U2 || !b2 || !x || !x_out || !ld_out || !layers) return JF_ERR_BADARG;
    if (!width_ok(K1) || !width_ok(H) || !width_ok(r2) || !rows_ok(B) || n_layers < 1 || n_layers > JF_MAX_CHAIN || D < 1 || (V1 && r1 < 1)) return JF_ERR_BADARG;
    if (K1 > AG_K1MAX || H > AG_HMAX || r2 > AG_RMAX || (V1 && r1 > AG_RMAX) || D > 8) return JF_ERR_UNSUPPORTED;
    AgArgs<T> a{};
    const int col = ag_make_layers<T>(layers, n_layers, D, a.L);
    if (col < 0) return col;
    if (B == 0) return JF_OK;
    a.in = in; a.in_stride = in_stride; a.V1 = V1; a.U1 = U1; a.b1 = b1; a.V2 = V2; a.U2 = U2; a.b2 = b2;
    a.K1 = K1; a.H = H; a.r1 = V1 ? r1 : 0; a.r2 = r2; a.N = col;
    a.x = x; a.xs = xs; a.ld_in = ld_in; a.B = B; a.D = D; a.n_layers = n_layers;
    a.x_out = x_out; a.xos = xos; a.ld_out = ld_out; a.blp_in = blp_in; a.blp_out = blp_out; a.status = status;
    if constexpr (sizeof(T) == 8) {
        // float64 with ranks <= 8: the whole block as a chain of v_mfma_f64_16x16x4 products (jf_amlp_mfma.h) -- same arithmetic, 16x less LDS
        // traffic; everything else (float32, larger ranks, odd hidden widths) stays on amlp_gf_kernel
        const size_t lds = am_lds_doubles(K1, H, V1 != nullptr, n_layers) * sizeof(T);
        if (r2 <= AM_R && (!V1 || r1 <= AM_R) && H % 16 == 0 && lds <= 160 * 1024) {
            auto k = amlp_gf_mfma_kernel<AgArgs<T>, FWD>;
            constexpr int NT = FWD ? AM_THREADS_FWD : AM_THREADS, ROWS = am_rows(NT);
            if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            jf::launch(k, dim3((unsigned)((B + ROWS - 1) / ROWS)), dim3(NT), lds, (hipStream_t)stream, a);
            return check_launch();
        }
    }
    if constexpr (FWD) return JF_ERR_UNSUPPORTED;                  // sampling direction: matrix-core variant only (float64, ranks <= 8)
    else return ag_launch<T>(a, D, (hipStream_t)stream);
}

static int lowrank_gf_chain_inv(const double* t2, int64_t t2s, const double* U2, const double* b2, int32_t r2, const double* x, int64_t xs,
                               const double* ld_in, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, double* x_out, int64_t xos,
                               double* ld_out, const double* blp_in, double* blp_out, double* aux, int32_t* status, void* stream) {
    if (!t2 || !U2 || !b2 || !x || !x_out || !ld_out || !layers) return JF_ERR_BADARG;
    if (!rows_ok(B) || n_layers < 1 || n_layers > JF_MAX_CHAIN || D < 1 || r2 < 1) return JF_ERR_BADARG;
    if (r2 > AM_R || D > 8) return JF_ERR_UNSUPPORTED;
    LrFwdArgs<double> a{};
    const int col = ag_make_layers<double>(layers, n_layers, D, a.L);
    if (col < 0) return col;
    const size_t lds = (size_t)n_layers * (AM_TILES * 2 * 64 + AM_TILES * 16) * sizeof(double);
    if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;
    if (B == 0) return JF_OK;
    a.t2 = t2; a.t2s = t2s; a.U2 = U2; a.b2 = b2; a.r2 = r2; a.x = x; a.xs = xs; a.ld_in = ld_in; a.B = B; a.D = D; a.n_layers = n_layers;
    a.x_out = x_out; a.xos = xos; a.ld_out = ld_out; a.blp_in = blp_in; a.blp_out = blp_out; a.aux = aux; a.status = status;
    auto k = lr_gf_fwd_kernel<LrFwdArgs<double>>;
    static LdsAttrOnce attr;
    if (lds > 48 * 1024) attr.set((const void*)k, (int)lds);
    constexpr int ROWS = am_rows(AM_THREADS);
    jf::launch(k, dim3((unsigned)((B + ROWS - 1) / ROWS)), dim3(AM_THREADS), lds, (hipStream_t)stream, a);
    return check_launch();
}

static int lowrank_gf_chain_inv_bwd(const double* t2, int64_t t2s, const double* U2, const double* b2, int32_t r2, const double* aux, const double* x_out,
                                   int64_t xos, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, const double* g_xout, int64_t gxos,
                                   const double* g_ld, const double* g_blp, double* g_x, int64_t gxs, double* g_t2, double* g_U2, double* g_b2,
                                   double* workspace, int32_t* status, void* stream) {
    if (!t2 || !U2 || !b2 || !aux || !g_x || !g_t2 || !g_U2 || !g_b2 || !workspace || !layers || (g_blp && !x_out)) return JF_ERR_BADARG;
    if (!rows_ok(B) || B < 1 || n_layers < 1 || n_layers > JF_MAX_CHAIN || D < 1 || r2 < 1) return JF_ERR_BADARG;
    if (r2 > AM_R || D > 8) return JF_ERR_UNSUPPORTED;
    LrReduceArgs red{};
    const int col = ag_make_layers<double>(layers, n_layers, D, red.L);
    if (col < 0) return col;
    const int n_wg = lr_n_wg(B);
    const size_t lds = (size_t)(LR_TILES * 2 * 64 + LR_TILES * 16 + LR_PSZ + LR_NW * 2 * 16 * 17) * sizeof(double);
    static LdsAttrOnce attr;
    attr.set((const void*)lr_gf_bwd_layer_kernel, (int)lds);
    for (int l = 0; l < n_layers; ++l) {
        LrBwdArgs<double> a{};
        a.t2 = t2; a.t2s = t2s; a.U2 = U2; a.b2 = b2; a.r2 = r2; a.aux = aux; a.x_out = x_out; a.xos = xos;
        a.B = B; a.n_row_tiles = (B + 15) / 16; a.D = D; a.layer = l; a.n_layers = n_layers; a.first = l == 0;
        a.L = red.L[l];
        a.g_xout = g_xout; a.gxos = gxos; a.g_ld = g_ld; a.g_blp = g_blp; a.g_x = g_x; a.gxs = gxs; a.g_t2 = g_t2;
        a.partial = workspace + (size_t)l * n_wg * LR_PSZ; a.status = status;
        jf::launch(lr_gf_bwd_layer_kernel, dim3((unsigned)n_wg), dim3(LR_NW * 64), lds, (hipStream_t)stream, a);
    }
    red.partial = workspace; red.n_wg = n_wg; red.n_layers = n_layers; red.D = D; red.r2 = r2; red.g_U2 = g_U2; red.g_b2 = g_b2;
    jf::launch(lr_reduce_kernel, dim3((unsigned)((n_layers * LR_PSZ + 63) / 64)), dim3(256), 0, (hipStream_t)stream, red);
    return check_launch();
}

static bool lrm_shape_ok(int32_t K1, int32_t H, int32_t r1, int32_t r2) {
    return K1 >= 1 && K1 <= AG_K1MAX && H >= 16 && H <= AG_HMAX && H % 16 == 0 && r1 >= 1 && r1 <= AM_R && r2 >= 1 && r2 <= AM_R;
}

static int lowrank_head(const double* in, int64_t is, const double* V1, const double* U1, const double* b1, const double* V2, int64_t B, int32_t K1,
                        int32_t H, int32_t r1, int32_t r2, double* t1, double* h, double* t2, void* stream) {
    if (!in || !V1 || !U1 || !b1 || !V2 || !t1 || !h || !t2 || !rows_ok(B)) return JF_ERR_BADARG;
    if (!lrm_shape_ok(K1, H, r1, r2)) return JF_ERR_UNSUPPORTED;
    if (B == 0) return JF_OK;
    LrmFwdArgs a{};
    a.in = in; a.in_stride = is; a.V1 = V1; a.U1 = U1; a.b1 = b1; a.V2 = V2; a.K1 = K1; a.H = H; a.r1 = r1; a.r2 = r2; a.B = B;
    a.t1 = t1; a.h = h; a.t2 = t2;
    const size_t lds = am_lds_mlp(K1, H, true) * sizeof(double);
    constexpr int ROWS = am_rows(AM_THREADS);
    jf::launch(lrm_head_fwd_kernel, dim3((unsigned)((B + ROWS - 1) / ROWS)), dim3(AM_THREADS), lds, (hipStream_t)stream, a);
    return check_launch();
}

static int lowrank_head_bwd(const double* in, int64_t is, const double* V1, const double* U1, const double* V2, int64_t B, int32_t K1, int32_t H,
                            int32_t r1, int32_t r2, const double* t1, const double* h, const double* g_t2, int64_t gs, double* g_c, int64_t gcs,
                            double* g_V1, double* g_U1, double* g_b1, double* g_V2, double* workspace, void* stream) {
    if (!in || !V1 || !U1 || !V2 || !t1 || !h || !g_t2 || !g_V1 || !g_U1 || !g_b1 || !g_V2 || !workspace || !rows_ok(B) || B < 1) return JF_ERR_BADARG;
    if (!lrm_shape_ok(K1, H, r1, r2)) return JF_ERR_UNSUPPORTED;
    const int KT = (K1 + 15) / 16, psz = lrm_psz(H, KT), n_wg = lrm_n_wg(B);
    LrmBwdArgs a{};
    a.in = in; a.in_stride = is; a.V1 = V1; a.U1 = U1; a.V2 = V2; a.K1 = K1; a.H = H; a.r1 = r1; a.r2 = r2; a.B = B; a.n_row_tiles = (B + 15) / 16;
    a.t1 = t1; a.h = h; a.g_t2 = g_t2; a.gs = gs; a.g_c = g_c; a.gcs = gcs; a.partial = workspace;
    const size_t lds = ((size_t)(H / 16) * 2 * 64 + (size_t)(H / 4) * 64 + (size_t)KT * 2 * 64 + psz + (size_t)LRM_NW * 2 * 16 * 17) * sizeof(double);
    static LdsAttrOnce attr1, attr2;
    if (KT == 1) {
        attr1.set((const void*)lrm_head_bwd_kernel<1>, (int)lds);
        jf::launch(lrm_head_bwd_kernel<1>, dim3((unsigned)n_wg), dim3(LRM_NW * 64), lds, (hipStream_t)stream, a);
    } else {
        attr2.set((const void*)lrm_head_bwd_kernel<2>, (int)lds);
        jf::launch(lrm_head_bwd_kernel<2>, dim3((unsigned)n_wg), dim3(LRM_NW * 64), lds, (hipStream_t)stream, a);
    }
    LrmReduceArgs red{};
    red.partial = workspace; red.n_wg = n_wg; red.H = H; red.K1 = K1; red.r1 = r1; red.r2 = r2;
    red.g_V1 = g_V1; red.g_U1 = g_U1; red.g_b1 = g_b1; red.g_V2 = g_V2;
    jf::launch(lrm_reduce_kernel, dim3((unsigned)((psz + 63) / 64)), dim3(256), 0, (hipStream_t)stream, red);
    return check_launch();
}

}  // namespace jf

extern "C" {
int jf_lowrank_head_f64(const double* in, int64_t is, const double* V1, const double* U1, const double* b1, const double* V2, int64_t B, int32_t K1,
                        int32_t H, int32_t r1, int32_t r2, double* t1, double* h, double* t2, void* s) {
    return jf::lowrank_head(in, is, V1, U1, b1, V2, B, K1, H, r1, r2, t1, h, t2, s);
}
int64_t jf_lowrank_head_workspace_doubles(int64_t B, int32_t K1, int32_t H) {
    if (B < 1 || K1 < 1 || K1 > jf::AG_K1MAX || H < 16 || H > jf::AG_HMAX) return 0;
    return (int64_t)jf::lrm_n_wg(B) * jf::lrm_psz(H, (K1 + 15) / 16);
}
int jf_lowrank_head_bwd_f64(const double* in, int64_t is, const double* V1, const double* U1, const double* V2, int64_t B, int32_t K1, int32_t H,
                            int32_t r1, int32_t r2, const double* t1, const double* h, const double* g_t2, int64_t gs, double* g_c, int64_t gcs,
                            double* g_V1, double* g_U1, double* g_b1, double* g_V2, double* workspace, void* s) {
    return jf::lowrank_head_bwd(in, is, V1, U1, V2, B, K1, H, r1, r2, t1, h, g_t2, gs, g_c, gcs, g_V1, g_U1, g_b1, g_V2, workspace, s);
}
int jf_lowrank_gf_chain_inv_f64(const double* t2, int64_t t2s, const double* U2, const double* b2, int32_t r2, const double* x, int64_t xs,
                                const double* ld_in, int64_t B, int32_t D, int32_t n, const jf_gf_layer* L, double* xo, int64_t xos, double* ldo,
                                const double* bi, double* bo, double* aux, int32_t* st, void* s) {
    return jf::lowrank_gf_chain_inv(t2, t2s, U2, b2, r2, x, xs, ld_in, B, D, n, L, xo, xos, ldo, bi, bo, aux, st, s);
}
int64_t jf_lowrank_gf_workspace_doubles(int64_t B, int32_t n_layers) {
    if (B < 1 || n_layers < 1 || n_layers > JF_MAX_CHAIN) return 0;
    return (int64_t)n_layers * jf::lr_n_wg(B) * jf::LR_PSZ;
}
int jf_lowrank_gf_chain_inv_bwd_f64(const double* t2, int64_t t2s, const double* U2, const double* b2, int32_t r2, const double* aux, const double* x_out,
                                    int64_t xos, int64_t B, int32_t D, int32_t n, const jf_gf_layer* L, const double* g_xout, int64_t gxos,
                                    const double* g_ld, const double* g_blp, double* g_x, int64_t gxs, double* g_t2, double* g_U2, double* g_b2,
                                    double* workspace, int32_t* st, void* s) {
    return jf::lowrank_gf_chain_inv_bwd(t2, t2s, U2, b2, r2, aux, x_out, xos, B, D, n, L, g_xout, gxos, g_ld, g_blp, g_x, gxs, g_t2, g_U2, g_b2,
                                        workspace, st, s);
}
int jf_amlp2_f32(const float* in, int64_t is, const float* V1, const float* U1, const float* b1, const float* V2, const float* U2, const float* b2,
                 int64_t B, int32_t K1, int32_t H, int32_t r1, int32_t r2, int32_t N, float* out, int64_t os, void* s) {
    return jf::amlp2<float>(in, is, V1, U1, b1, V2, U2, b2, B, K1, H, r1, r2, N, out, os, s);
}
int jf_amlp2_f64(const double* in, int64_t is, const double* V1, const double* U1, const double* b1, const double* V2, const double* U2,
                 const double* b2, int64_t B, int32_t K1, int32_t H, int32_t r1, int32_t r2, int32_t N, double* out, int64_t os, void* s) {
    return jf::amlp2<double>(in, is, V1, U1, b1, V2, U2, b2, B, K1, H, r1, r2, N, out, os, s);
}
int jf_amlp_gf_chain_inv_f32(const float* in, int64_t is, const float* V1, const float* U1, const float* b1, const float* V2, const float* U2,
                             const float* b2, int32_t K1, int32_t H, int32_t r1, int32_t r2, const float* x, int64_t xs, const float* ld_in, int64_t B,
                             int32_t D, int32_t n, const jf_gf_layer* L, float* xo, int64_t xos, float* ldo, const float* bi, float* bo, int32_t* st,
                             void* s) {
    return jf::amlp_gf_chain_inv<float>(in, is, V1, U1, b1, V2, U2, b2, K1, H, r1, r2, x, xs, ld_in, B, D, n, L, xo, xos, ldo, bi, bo, st, s);
}
int jf_amlp_gf_chain_fwd_f64(const double* in, int64_t is, const double* V1, const double* U1, const double* b1, const double* V2, const double* U2,
                             const double* b2, int32_t K1, int32_t H, int32_t r1, int32_t r2, const double* z, int64_t zs, const double* ld_in,
                             int64_t B, int32_t D, int32_t n, const jf_gf_layer* L, double* xo, int64_t xos, double* ldo, int32_t* st, void* s) {
    return jf::amlp_gf_chain_inv<double, true>(in, is, V1, U1, b1, V2, U2, b2, K1, H, r1, r2, z, zs, ld_in, B, D, n, L, xo, xos, ldo, nullptr, nullptr,
                                               st, s);
}
int jf_amlp_gf_chain_inv_f64(const double* in, int64_t is, const double* V1, const double* U1, const double* b1, const double* V2, const double* U2,
                             const double* b2, int32_t K1, int32_t H, int32_t r1, int32_t r2, const double* x, int64_t xs, const double* ld_in,
                             int64_t B, int32_t D, int32_t n, const jf_gf_layer* L, double* xo, int64_t xos, double* ldo, const double* bi, double* bo,
                             int32_t* st, void* s) {
    return jf::amlp_gf_chain_inv<double>(in, is, V1, U1, b1, V2, U2, b2, K1, H, r1, r2, x, xs, ld_in, B, D, n, L, xo, xos, ldo, bi, bo, st, s);
}
}
