// Conditional e-block in ONE launch: amortisation MLP (Linear -> tanh -> Linear) + the chain of 'g' layers it parametrises.
//
//   jf_cond_gf_chain_inv_*   log-prob direction of an autoregressive / conditional Euclidean block
//                            = mlp_predictors[i](cat(conditional_input, embeddings))  (main/default.py:656-670, 946-962)
//                              followed by the per-block layer loop of all_layer_inverse (main/default.py:998-1031)
//
// The per-sample parameter block (548 floats per row for e4 / gggg) never reaches HBM (note on fused accounting):
// a wave owns 16 rows; it keeps their hidden activations in registers (as jf_mlp2), and for each layer -- in the order the inverse
// direction consumes them, last layer first -- multiplies them with that layer's slice of W2 on the matrix cores straight into its
// own LDS tile [16 rows x P_layer], then evaluates the layer on that tile with the lane = (row, coordinate) code of jf_gf.h.
// W2 is streamed through one LDS chunk (48 output columns) shared by the 4 waves of the workgroup; the chunk for the next MFMA pass is
// prefetched into registers while the current one is multiplied, so its L2 latency is never exposed.
// Measured on MI355X (scripts/probe/coexec.hip): f32 MFMA and VALU work do NOT overlap on a SIMD, neither across waves nor inside one
// wave, so the floor of this kernel is (MFMA cycles + VALU cycles); what fusion removes is the 2 x 2.3 GB HBM round trip of the
// parameter block and one kernel's worth of latency-bound phases.
//
// MFMA: v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 (16-row tiles = one row-group pass of the flow for D = 3, 4).
// Both products transposed (h^T = W1 x^T, params^T = W2 h^T), see mlp_kernels.hip: the first result is the second's B operand.
#include "jf_gf.h"
#include "jf_mfma.h"

namespace jf {

constexpr int CG_HMAX = 128, CG_K1MAX = 32;
constexpr int CG_CT = 3;                         // 16-column MFMA tiles per W2 chunk
constexpr int CG_CHUNK = 16 * CG_CT;             // 48 output columns per chunk
constexpr int CG_ROWS = 64;                      // rows per workgroup (4 waves x 16)
template <typename T> struct CgCfg;
template <> struct CgCfg<float> { static constexpr int LDW = CG_HMAX + 4; };     // aligned, conflict-free ds_read_b128 of 4 consecutive k
template <> struct CgCfg<double> { static constexpr int LDW = CG_HMAX + 1; };

template <typename T> struct CondArgs {
    // MLP
    const T* in; int64_t in_stride;
    const T* W1; int64_t w1s; const T* b1;
    const T* W2; int64_t w2s; const T* b2;
    int K1, H, N;
    // flow
    const T* x; int64_t xs;
    const T* ld_in;
    int64_t B;
    int D, n_layers, tile_stride;
    GfLayerDev<T> L[JF_MAX_CHAIN];
    T* x_out; int64_t xos;
    T* ld_out;
    const T* blp_in; T* blp_out;
    int32_t* status;
};

template <typename T, int JH>
__global__ void __launch_bounds__(256, 2) cond_gf_chain_kernel(const CondArgs<T> a) {
    using MF = Mfma16<T>;
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    constexpr int MT = 16, KS = 4, NREG = 4;
    constexpr int LDW = CgCfg<T>::LDW;
    constexpr int HP = JH * MT;
    constexpr int WPT = CG_CHUNK * CG_HMAX / VN / 256;             // 16-byte pieces of a W2 chunk per thread (f32: 6, f64: 12)
    static_assert(CG_CHUNK * CG_HMAX / VN % 256 == 0, "chunk must be a whole number of thread passes");
    constexpr int G = 4;                                           // lanes per row in the flow phase (D = 3, 4)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* Ws = reinterpret_cast<T*>(smem_raw);                        // [CG_CHUNK][LDW]  W2 chunk
    T* Bs = Ws + CG_CHUNK * LDW;                                   // [CG_CHUNK]       its bias
    T* tiles = Bs + CG_CHUNK;                                      // 4 x [16][tile_stride] parameter tiles, one per wave
    const int k1p = (a.K1 + KS - 1) / KS * KS, ldk = k1p + 1;
    T* Xs = Ws;                                                    // phase 1 only (overlays the W2 chunk)
    T* W1s = Xs + CG_ROWS * ldk;
    T* b1s = W1s + HP * ldk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * CG_ROWS;
    const int64_t last = a.B - 1;
    const int D = a.D, S = a.tile_stride;

    // ---- phase 1: h^T = tanh(W1 x^T + b1) for the wave's 16 rows; rows past B replicate row B-1
    {
        const int nx = CG_ROWS * k1p, nw = HP * k1p;
        for (int base = 0; base < nx; base += 4 * 256) {
            T v[4]; int o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256 + tid;
                const int r = idx / k1p, c = idx - r * k1p;
                const int64_t gr = row0 + r;
                const T t = a.in[(gr <= last ? gr : last) * a.in_stride + (c < a.K1 ? c : 0)];
                v[u] = c < a.K1 ? t : T(0);
                o[u] = idx < nx ? r * ldk + c : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (o[u] >= 0) Xs[o[u]] = v[u];
        }
        for (int base = 0; base < nw; base += 4 * 256) {
            T v[4]; int o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256 + tid;
                const int r = idx / k1p, c = idx - r * k1p;
                const T t = a.W1[(int64_t)(r < a.H ? r : a.H - 1) * a.w1s + (c < a.K1 ? c : 0)];
                v[u] = (r < a.H && c < a.K1) ? t : T(0);
                o[u] = idx < nw ? r * ldk + c : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (o[u] >= 0) W1s[o[u]] = v[u];
        }
        if (tid < HP) b1s[tid] = tid < a.H ? a.b1[tid < a.H ? tid : 0] : T(0);
    }
    __syncthreads();
    T hreg[JH][NREG];
    {
        typename MF::Acc acc[JH];
#pragma unroll
        for (int j = 0; j < JH; ++j)
#pragma unroll
            for (int r = 0; r < NREG; ++r) acc[j][r] = T(0);
        for (int s = 0; s < k1p / KS; ++s) {
            const int kk = s * KS + lq;
            const T xb = Xs[(wave * MT + li) * ldk + kk];
#pragma unroll
            for (int j = 0; j < JH; ++j) acc[j] = MF::mma(W1s[(j * MT + li) * ldk + kk], xb, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < JH; ++j)
#pragma unroll
            for (int r = 0; r < NREG; ++r) hreg[j][r] = M<T>::tanh_fast(acc[j][r] + b1s[j * MT + MF::row_of(r, lane)]);
    }

    // ---- flow state: lane = (row r of the wave's 16, coordinate g)
    const int g = lane & (G - 1), rr = lane >> 2;
    const bool live = g < D, leader = g == 0;
    const int d = live ? g : D - 1;
    const int64_t row = row0 + wave * MT + rr;
    const bool row_valid = row <= last;
    const int64_t rrow = row_valid ? row : last;
    T x = a.x[rrow * a.xs + d];
    T ld = a.ld_in ? a.ld_in[rrow] : T(0);
    T* ptile = tiles + wave * MT * S;

    // ---- W2 chunk streaming: (layer, chunk) pairs in consumption order
    V wreg[WPT];
    T breg = T(0);
    // per-thread constants of the chunk copy: piece u of thread tid is row u*RPP + r_t, 16-byte column c_t of the chunk, so the global address
    // is (uniform chunk/pass base) + (one 32-bit lane offset) and the LDS address a constant -- no address arithmetic per piece
    constexpr int PPR = CG_HMAX / VN;                             // 16-byte pieces per W2 row
    constexpr int RPP = 256 / PPR;                                // rows covered by one pass of the 256 threads
    const int r_t = tid / PPR, c_t = (tid % PPR) * VN;
    const unsigned voff = (unsigned)((r_t * a.w2s + c_t) * (int64_t)sizeof(T));
    T* const lbase = Ws + r_t * LDW + c_t;
    const bool h_full = a.H == CG_HMAX;                            // block-uniform
    auto fetch = [&](int col0) {                                   // straight-line, loads issued back to back
        if (h_full && col0 + CG_CHUNK <= a.N) {                    // block-uniform fast path: whole chunk inside W2, no clamps, no selects
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                const char* base = reinterpret_cast<const char*>(a.W2 + (int64_t)(col0 + u * RPP) * a.w2s);    // uniform
                wreg[u] = *reinterpret_cast<const V*>(base + voff);
            }
        } else {                                                   // clamped addresses + selects (rows past N replicate row N-1)
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                const int gc = col0 + u * RPP + r_t;
                const bool ok = c_t < a.H;
                const V v = *reinterpret_cast<const V*>(a.W2 + (int64_t)(gc < a.N ? gc : a.N - 1) * a.w2s + (ok ? c_t : 0));
                wreg[u].x = ok ? v.x : T(0); wreg[u].y = ok ? v.y : T(0);
                if constexpr (VN == 4) { wreg[u].z = ok ? v.z : T(0); wreg[u].w = ok ? v.w : T(0); }
            }
        }
        const int bc = col0 + (tid < CG_CHUNK ? tid : 0);
        breg = (a.b2 != nullptr) ? a.b2[bc < a.N ? bc : a.N - 1] : T(0);
    };
    auto put = [&]() {
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            T* dd = lbase + u * RPP * LDW;
            if constexpr (VN == 4) { *reinterpret_cast<V*>(dd) = wreg[u]; }
            else { dd[0] = wreg[u].x; dd[1] = wreg[u].y; }
        }
        if (tid < CG_CHUNK) Bs[tid] = breg;
    };
    using F = typename std::conditional<VN == 4, V, T>::type;
    constexpr int NGRP = VN == 4 ? 1 : NREG;                       // fragment reads per hidden tile (f32: one b128 = 4 k; f64: 4 x b64)
    constexpr int KPG = NREG / NGRP;
    const T* wb = Ws + li * LDW;
    auto elem = [](const F& f, int e) -> T {
        if constexpr (VN == 4) return e == 0 ? f.x : e == 1 ? f.y : e == 2 ? f.z : f.w;
        else return f;
    };

    int l = a.n_layers - 1, chunk = 0;
    fetch(a.L[l].col0);
    lds_barrier();                                                 // every wave is done with Xs / W1s / b1s
    put();
    lds_barrier();
    while (l >= 0) {
        const GfLayerDev<T> o = a.L[l];                            // uniform index: scalar loads from the kernarg segment
        const int nchunks = (o.n_params + CG_CHUNK - 1) / CG_CHUNK;
        const bool layer_done = chunk + 1 == nchunks;
        const int nl = layer_done ? l - 1 : l, nc = layer_done ? 0 : chunk + 1;
        if (nl >= 0) fetch(a.L[nl].col0 + nc * CG_CHUNK);          // next chunk: in flight while this one is multiplied
        // ---- matrix phase: parameters [16 rows] x [48 columns of this chunk] -> the wave's tile
        typename MF::Acc acc[CG_CT];
#pragma unroll
        for (int ct = 0; ct < CG_CT; ++ct)
#pragma unroll
            for (int r = 0; r < NREG; ++r) acc[ct][r] = T(0);
#pragma unroll
        for (int j = 0; j < JH; ++j)
#pragma unroll
            for (int gq = 0; gq < NGRP; ++gq) {
                F frag[CG_CT];
#pragma unroll
                for (int ct = 0; ct < CG_CT; ++ct) frag[ct] = *reinterpret_cast<const F*>(wb + ct * MT * LDW + j * MT + MF::row_of(gq * KPG, lane));
#pragma unroll
                for (int e = 0; e < KPG; ++e)
#pragma unroll
                    for (int ct = 0; ct < CG_CT; ++ct) acc[ct] = MF::mma(elem(frag[ct], e), hreg[j][gq * KPG + e], acc[ct]);
            }
        {   // result: acc[ct][v] = param (chunk*48 + 16 ct + row_of(v, lane)) of row li  -> tile [row][param]
            T* trow = ptile + li * S + chunk * CG_CHUNK;
#pragma unroll
            for (int ct = 0; ct < CG_CT; ++ct) {
                if constexpr (MF::RUN == 4) {
                    const int c0 = ct * MT + MF::row_of(0, lane);
                    V ov;                                          // bias as four b32 broadcasts (a b128 read of one address by 16 lanes serialises)
                    ov.x = acc[ct][0] + Bs[c0]; ov.y = acc[ct][1] + Bs[c0 + 1]; ov.z = acc[ct][2] + Bs[c0 + 2]; ov.w = acc[ct][3] + Bs[c0 + 3];
                    *reinterpret_cast<V*>(trow + c0) = ov;
                } else {
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        const int c0 = ct * MT + MF::row_of(r, lane);
                        trow[c0] = acc[ct][r] + Bs[c0];
                    }
                }
            }
        }
        lds_barrier();                                             // every wave has read the chunk (and written its tile part)
        if (nl >= 0) put();
        if (layer_done) {
            // ---- flow phase on the wave's own tile (raw parameters; jf_gf.h)
            const T* p = ptile + rr * S + d;
            if (o.model_offset) x -= p[0];                                                   // euclidean_base.py:40-45
            x = gfg_rotate_inv<T, G, true>(p, o, D, live, x);
            const MixQ<T> q = gfg_mixture<T, true>(p, o, D, x);
            const IcdfOut<T> s = gf_icdf<T>(o.inv_type, q);
            x = s.y;
            ld += group_sum<T, G>(live ? s.logd : T(0));
        }
        lds_barrier();
        l = nl; chunk = nc;
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

// ----------------------------------------------------------------------------------------------------------
template <typename T, int JH> static int cond_launch(const CondArgs<T>& a, size_t lds, hipStream_t st) {
    auto k = cond_gf_chain_kernel<T, JH>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    jf::launch(k, dim3((unsigned)((a.B + CG_ROWS - 1) / CG_ROWS)), dim3(256), lds, st, a);
    return check_launch();
}

template <typename T>
static int cond_gf_chain_inv(const T* in, int64_t in_stride, const T* W1, int64_t w1s, const T* b1, const T* W2, int64_t w2s, const T* b2, int32_t K1,
                             int32_t H, const T* x, int64_t xs, const T* ld_in, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers,
                             T* x_out, int64_t xos, T* ld_out, const T* blp_in, T* blp_out, int32_t* status, void* stream) {
    if (!in || !W1 || !b1 || !W2 || !x || !x_out || !ld_out || !layers) return JF_ERR_BADARG;
    if (!width_ok(K1) || !width_ok(H) || !rows_ok(B) || n_layers < 1 || n_layers > JF_MAX_CHAIN) return JF_ERR_BADARG;
    if (K1 > CG_K1MAX || H > CG_HMAX || D < 3 || D > 4) return JF_ERR_UNSUPPORTED;          // 4-lane row groups only (16 rows per MFMA tile)
    if ((H % Vec16<T>::N) || (w2s % Vec16<T>::N) || (reinterpret_cast<uintptr_t>(W2) & 15u)) return JF_ERR_UNSUPPORTED;
    if (B == 0) return JF_OK;
    CondArgs<T> a{};
    int col = 0, maxp = 0;
    for (int l = 0; l < n_layers; ++l) {
        const jf_gf_layer& h = layers[l];
        GfLayerDev<T>& o = a.L[l];
        if (h.num_kde < 1 || h.num_kde > (1 << 16) || h.hh_iter < 0 || h.hh_iter > (1 << 16) || h.width_min <= 0) return JF_ERR_BADARG;   // (bounded: the column offsets below are ints)
        if (h.rotation_mode != JF_GF_ROT_HOUSEHOLDER || h.center_mean || h.add_skewness) return JF_ERR_UNSUPPORTED;   // general-option layers: jf_gf_chain_inv only
        if (h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION && h.width_max <= 0) return JF_ERR_BADARG;
        if (h.nonlinear_stretch_type != JF_GF_STRETCH_CLASSIC) return JF_ERR_UNSUPPORTED;     // per-lane knot tables do not fit beside the tiles
        o.K = h.num_kde; o.hh = h.hh_iter; o.model_offset = h.model_offset; o.fit_norm = h.fit_normalization;
        o.reg_norm = h.regulate_normalization; o.inv_type = h.inverse_function_type; o.width_mode = h.width_mode;
        o.clamp_widths = h.clamp_widths;
        o.fast = (h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION && !h.clamp_widths && h.fit_normalization && h.regulate_normalization) ? 1 : 0; o.stretch = JF_GF_STRETCH_CLASSIC; o.off_box = 0;
        const int kd = h.num_kde * D;
        o.off_rot = h.model_offset ? D : 0;
        o.off_mean = o.off_rot + h.hh_iter * D;
        o.off_lw = o.off_mean + kd;
        o.off_ln = o.off_lw + kd;
        o.n_params = o.off_ln + (h.fit_normalization ? kd : 0);
        o.col0 = col; o.vec_ok = 0;
        o.wmin = (T)h.width_min; o.wmax = (T)h.width_max; o.inv_wmax = h.width_max > 0 ? (T)(1.0 / h.width_max) : T(0);
        o.nmin = (T)h.norm_min; o.nmax = (T)h.norm_max;
        o.lw_lo = (T)log(0.01 * h.width_min);
        if (h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION) o.lw_hi = (T)(3.0 * log(h.width_max));
        else o.lw_hi = h.width_max > 0 ? (T)log(h.width_max) : (T)INFINITY;
        col += o.n_params;
        const int padded = (o.n_params + CG_CHUNK - 1) / CG_CHUNK * CG_CHUNK;                // the matrix phase writes whole chunks
        if (padded > maxp) maxp = padded;
    }
    a.in = in; a.in_stride = in_stride; a.W1 = W1; a.w1s = w1s; a.b1 = b1; a.W2 = W2; a.w2s = w2s; a.b2 = b2; a.K1 = K1; a.H = H; a.N = col;
    a.x = x; a.xs = xs; a.ld_in = ld_in; a.B = B; a.D = D; a.n_layers = n_layers;
    a.tile_stride = padded_stride<T>(maxp);
    a.x_out = x_out; a.xos = xos; a.ld_out = ld_out; a.blp_in = blp_in; a.blp_out = blp_out; a.status = status;
    constexpr int KS = 4;
    const int k1p = (K1 + KS - 1) / KS * KS, ldk = k1p + 1;
    const size_t phase1 = (size_t)CG_ROWS * ldk + (size_t)CG_HMAX * ldk + CG_HMAX;
    const size_t chunk = (size_t)CG_CHUNK * CgCfg<T>::LDW + CG_CHUNK;
    if (phase1 > chunk) return JF_ERR_UNSUPPORTED;
    const size_t lds = (chunk + (size_t)4 * 16 * a.tile_stride) * sizeof(T);
    if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;
    const int tiles = (H + 15) / 16;
    if (tiles <= 2) return cond_launch<T, 2>(a, lds, (hipStream_t)stream);
    if (tiles <= 4) return cond_launch<T, 4>(a, lds, (hipStream_t)stream);
    return cond_launch<T, 8>(a, lds, (hipStream_t)stream);
}

}  // namespace jf

extern "C" {
int jf_cond_gf_chain_inv_f32(const float* in, int64_t is, const float* W1, int64_t w1s, const float* b1, const float* W2, int64_t w2s, const float* b2,
                             int32_t K1, int32_t H, const float* x, int64_t xs, const float* ld_in, int64_t B, int32_t D, int32_t n,
                             const jf_gf_layer* L, float* xo, int64_t xos, float* ldo, const float* bi, float* bo, int32_t* st, void* s) {
    return jf::cond_gf_chain_inv<float>(in, is, W1, w1s, b1, W2, w2s, b2, K1, H, x, xs, ld_in, B, D, n, L, xo, xos, ldo, bi, bo, st, s);
}
int jf_cond_gf_chain_inv_f64(const double* in, int64_t is, const double* W1, int64_t w1s, const double* b1, const double* W2, int64_t w2s,
                             const double* b2, int32_t K1, int32_t H, const double* x, int64_t xs, const double* ld_in, int64_t B, int32_t D,
                             int32_t n, const jf_gf_layer* L, double* xo, int64_t xos, double* ldo, const double* bi, double* bo, int32_t* st,
                             void* s) {
    return jf::cond_gf_chain_inv<double>(in, is, W1, w1s, b1, W2, w2s, b2, K1, H, x, xs, ld_in, B, D, n, L, xo, xos, ldo, bi, bo, st, s);
}
}
