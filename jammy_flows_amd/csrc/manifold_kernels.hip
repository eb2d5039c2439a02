// Kernels + C-ABI launchers for chains of manifold layers ('r', 'o', 'm', 'f', 'v') and the sphere <-> embedding conversions.
//
// One wave (64 lanes) per workgroup, one sample per lane.  For each layer of the chain the 64 x n_params slab of per-sample
// parameters is staged from HBM into an LDS tile with coalesced 16-byte loads (stage_rows, jf_common.h) and consumed
// lane-per-row; broadcast parameters (param_batch == 1) are staged once as a single row that every lane reads (LDS broadcast).
// A second LDS region holds the lane-private spline knot tables (jf_spline.h).
#include <type_traits>

#include "jf_expmap.h"
#include "jf_manifold.h"

namespace jf {

constexpr int MC_MAX_PRE = 4;
template <typename T, typename CLayer> struct MChainArgs {
    const T* x; int64_t xs;
    const T* ld_in;
    const T* params; int64_t ps;
    int bcast;
    int64_t B;
    int n_layers;
    int dim;                 // columns of x actually used (<= Fam::DIM)
    int tile_stride;
    int scratch;             // per-lane elements of the emitted-parameter scratch behind the knot tables (0 = none)
    int tab;                 // per-lane elements of the knot tables: JF_SPLINE_TAB, or 0 for chains without a spline (default 'f', 'm', 'v', 'c'):
                             //   53 words of LDS per lane that bound the resident workgroups of these latency-bound kernels
    int rows;                // rows per workgroup (64 unless the parameter tile of 64 rows would not fit in LDS)
    int shared_tab;          // broadcast parameters + a family whose knot table does not depend on the row: ONE table per workgroup, built by lane 0
    int vec_ok[JF_MAX_MCHAIN];
    int col0[JF_MAX_MCHAIN];
    int ncols[JF_MAX_MCHAIN];
    CLayer L[JF_MAX_MCHAIN];
    T* x_out; int64_t xos;
    T* ld_out;
    const T* blp_in; T* blp_out;
    int64_t* bins; int64_t bins_stride;
    int32_t* status;
    // log-prob direction, the LAST block of a pdf (jf_<fam>_chain_inv_sum): the earlier blocks' per-row log-dets / base log-probs, added in list
    // order in front of this block's; ld_out / blp_out then hold the pdf's totals and total = blp_out + ld_out (what jf_combine_rows returns)
    const T* ld_pre[MC_MAX_PRE]; const T* blp_pre[MC_MAX_PRE];
    int n_ld_pre, n_blp_pre;
    T* total;
};

// (fam_tab_words: jf_manifold.h)
template <class Fam, class = void> struct fam_has_build : std::false_type {};
template <class Fam> struct fam_has_build<Fam, std::void_t<decltype(Fam::HAS_BUILD)>> : std::true_type {};

template <typename T, class Fam, bool FWD>
__global__ void __launch_bounds__(64) mchain_kernel(const MChainArgs<T, typename Fam::CLayer> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x;
    const int rows = a.rows;
    const int tile_rows = a.bcast ? 1 : rows;
    const bool lane_in = tid < rows;                     // lanes beyond the workgroup's rows idle (they own no scratch)
    const int slot = lane_in ? tid : 0;
    constexpr bool CAN_SHARE = fam_has_build<Fam>::value;
    const bool shared = CAN_SHARE && a.shared_tab != 0;  // uniform
    T* tab = tile + tile_rows * a.tile_stride + (shared ? 0 : slot * a.tab);
    const int64_t row0 = (int64_t)blockIdx.x * rows;
    const int64_t row = row0 + tid;
    const bool active = lane_in && row < a.B;
    const int64_t rrow = active ? row : a.B - 1;
    const int valid_rows = (int)((a.B - row0) < rows ? (a.B - row0) : rows);

    T x[3] = {T(0), T(0), T(0)};
#pragma unroll
    for (int d = 0; d < Fam::DIM; ++d) if (d < a.dim) x[d] = a.x[rrow * a.xs + d];
    T ld = a.ld_in ? a.ld_in[rrow] : T(0);
    LaneCtx<T> ctx;
    ctx.tab = tab;
    ctx.corr = tile + tile_rows * a.tile_stride + (shared ? 1 : rows) * a.tab + slot * a.scratch;
    ctx.bins = (a.bins && active) ? a.bins + row * a.bins_stride : nullptr;
    ctx.bin_i = 0;
    ctx.oob = ctx.nonconv = ctx.nonfinite = false;
    ctx.lane_valid = active;

    for (int i = 0; i < a.n_layers; ++i) {
        const int l = FWD ? i : a.n_layers - 1 - i;
        __syncthreads();
        if (a.bcast) {
            for (int j = tid; j < a.ncols[l]; j += 64) tile[j] = a.params[a.col0[l] + j];
        } else {
            stage_rows<T>(tile, a.tile_stride, a.params + row0 * a.ps + a.col0[l], a.ps, a.ncols[l], rows, valid_rows, tid, 64, a.vec_ok[l] != 0);
        }
        __syncthreads();
        const T* prow = tile + (a.bcast ? 0 : slot * a.tile_stride);
        if constexpr (CAN_SHARE) {
            if (shared) {                                // the layer's table, once (it was 64 identical builds: ~3 nb transcendentals per lane)
                if (tid == 0) Fam::template build<T>(a.L[l], prow, tab);
                __syncthreads();
                ctx.tab_built = true;
            }
        }
        if (lane_in) Fam::template apply<T, FWD>(a.L[l], prow, x, ld, ctx);
    }
    bool bad = !M<T>::finite(ld);
#pragma unroll
    for (int d = 0; d < Fam::DIM; ++d) bad = bad || !M<T>::finite(x[d]);
    if (active) {
#pragma unroll
        for (int d = 0; d < Fam::DIM; ++d) if (d < a.dim) a.x_out[row * a.xos + d] = x[d];
        T ldv = ld;
        if (a.n_ld_pre > 0) {                              // uniform: list order, this block last (the bits of jf_combine_rows)
            T t = a.ld_pre[0][row];
            for (int i = 1; i < a.n_ld_pre; ++i) t += a.ld_pre[i][row];
            ldv = t + ld;
        }
        a.ld_out[row] = ldv;
        if (a.blp_out) {
            T s = a.blp_in ? a.blp_in[row] : T(0);
#pragma unroll
            for (int d = 0; d < Fam::DIM; ++d) if (d < a.dim) s += T(-0.5) * x[d] * x[d] - M<T>::HALF_LN_2PI;
            if (a.n_blp_pre > 0) {
                T t = a.blp_pre[0][row];
                for (int i = 1; i < a.n_blp_pre; ++i) t += a.blp_pre[i][row];
                s = t + s;
            }
            a.blp_out[row] = s;
            if (a.total) a.total[row] = s + ldv;
        }
    }
    status_add(a.status, JF_STATUS_NONFINITE, active && (bad || ctx.nonfinite));
    status_add(a.status, JF_STATUS_OUT_OF_RANGE, active && ctx.oob);
    status_add(a.status, JF_STATUS_NONCONVERGED, active && ctx.nonconv);
}

template <typename T, class Fam, bool FWD>
static int mchain(const T* x, int64_t xs, const T* ld_in, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t n_layers,
                  const typename Fam::CLayer* layers, T* x_out, int64_t xos, T* ld_out, const T* blp_in, T* blp_out, int64_t* bins,
                  int64_t bins_stride, int32_t* status, void* stream, const jf_row_list* ld_pre = nullptr, const jf_row_list* blp_pre = nullptr,
                  T* total = nullptr) {
    if (!x || !x_out || !ld_out || !layers || n_layers < 1 || n_layers > JF_MAX_MCHAIN || B < 0) return JF_ERR_BADARG;
    if ((ld_pre || blp_pre || total) && (FWD || !blp_out)) return JF_ERR_BADARG;
    if ((ld_pre && (ld_pre->n < 0 || ld_pre->n > MC_MAX_PRE)) || (blp_pre && (blp_pre->n < 0 || blp_pre->n > MC_MAX_PRE))) return JF_ERR_UNSUPPORTED;
    if (ld_pre) for (int i = 0; i < ld_pre->n; ++i) if (!ld_pre->p[i]) return JF_ERR_BADARG;
    if (blp_pre) for (int i = 0; i < blp_pre->n; ++i) if (!blp_pre->p[i]) return JF_ERR_BADARG;
    if (pb != 1 && pb != B) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    MChainArgs<T, typename Fam::CLayer> a{};
    a.x = x; a.xs = xs; a.ld_in = ld_in; a.params = params; a.ps = ps; a.bcast = (pb == 1) ? 1 : 0; a.B = B; a.n_layers = n_layers;
    int col = 0, maxp = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (!Fam::sane(layers[l])) return JF_ERR_BADARG;
        a.L[l] = layers[l];
        const int n = Fam::row_len(layers[l]);
        a.col0[l] = col; a.ncols[l] = n;
        a.vec_ok[l] = (!a.bcast && n > 0 && aligned16<T>(params, ps, col) && (n % Vec16<T>::N == 0)) ? 1 : 0;
        col += n;
        if (n > maxp) maxp = n;
    }
    if (col > 0 && !params) return JF_ERR_BADARG;
    a.tile_stride = padded_stride<T>(maxp > 0 ? maxp : 1);
    a.dim = Fam::DIM;
    if constexpr (std::is_same<Fam, CFam>::value) a.dim = layers[0].kind == 2 ? 2 : 1;
    a.x_out = x_out; a.xos = xos; a.ld_out = ld_out; a.blp_in = blp_in; a.blp_out = blp_out; a.bins = bins; a.bins_stride = bins_stride; a.status = status;
    if (ld_pre) { for (int i = 0; i < ld_pre->n; ++i) a.ld_pre[i] = static_cast<const T*>(ld_pre->p[i]); a.n_ld_pre = ld_pre->n; }
    if (blp_pre) { for (int i = 0; i < blp_pre->n; ++i) a.blp_pre[i] = static_cast<const T*>(blp_pre->p[i]); a.n_blp_pre = blp_pre->n; }
    a.total = total;
    a.scratch = 0;
    if constexpr (std::is_same<Fam, FFam>::value) {
        for (int l = 0; l < n_layers; ++l) {
            if (!layers[l].correlated) continue;
            if (layers[l].corr_hidden < 1 || layers[l].corr_rank < 0 || FFam::corr_out(layers[l]) + layers[l].corr_rank > JF_CORR_SCRATCH - 1)
                return JF_ERR_UNSUPPORTED;
            a.scratch = JF_CORR_SCRATCH;
        }
    }
    a.tab = 0;
    for (int l = 0; l < n_layers; ++l)
        if (Fam::needs_tab(layers[l])) { const int w = fam_tab_words<Fam>::of(layers[l]); a.tab = w > a.tab ? w : a.tab; }
    a.rows = 64;
    a.shared_tab = (fam_has_build<Fam>::value && a.bcast && a.tab > 0) ? 1 : 0;
    size_t lds = 0;
    for (;;) {
        lds = ((size_t)(a.bcast ? 1 : a.rows) * a.tile_stride + (size_t)(a.shared_tab ? 1 : a.rows) * a.tab + (size_t)a.rows * a.scratch) * sizeof(T);
        if (lds <= 160 * 1024 || a.rows == 8) break;
        a.rows >>= 1;
    }
    if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;
    auto k = mchain_kernel<T, Fam, FWD>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    jf::launch(k, dim3((unsigned)((B + a.rows - 1) / a.rows)), dim3(64), lds, (hipStream_t)stream, a);
    return check_launch();
}

static bool spline_ok(const jf_spline_opts& s) {
    return s.num_bins >= 1 && s.num_bins <= JF_SPLINE_CAP && s.n_w >= 0 && s.n_h >= 0 && s.n_d >= 0 && (s.smooth == 0 || s.num_bins <= 3);
}

// ---- sphere <-> embedding
template <typename T, bool TO_EMB>
__global__ void __launch_bounds__(256) embed_kernel(const T* __restrict__ x, int64_t xs, const T* __restrict__ ld_in, int64_t B, int dim,
                                                    T* __restrict__ x_out, int64_t xos, T* __restrict__ ld_out) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    T ld = ld_in ? ld_in[row] : T(0);
    const T* r = x + row * xs;
    T* o = x_out + row * xos;
    T e[3];
    if (TO_EMB) {
        if (dim == 1) { s1_to_eucl<T>(r[0], e); o[0] = e[0]; o[1] = e[1]; }
        else { s2_to_eucl<T>(r[0], r[1], e, ld); o[0] = e[0]; o[1] = e[1]; o[2] = e[2]; }
    } else {
        if (dim == 1) { e[0] = r[0]; e[1] = r[1]; o[0] = eucl_to_s1<T>(e); }
        else { e[0] = r[0]; e[1] = r[1]; e[2] = r[2]; T th, ph; eucl_to_s2<T>(e, th, ph, ld); o[0] = th; o[1] = ph; }
    }
    if (ld_out) ld_out[row] = ld;
}
template <typename T, bool TO_EMB>
static int embed(const T* x, int64_t xs, const T* ld_in, int64_t B, int32_t dim, T* x_out, int64_t xos, T* ld_out, void* stream) {
    if (!x || !x_out || (dim != 1 && dim != 2) || B < 0) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    jf::launch((embed_kernel<T, TO_EMB>), dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, xs, ld_in, B, (int)dim, x_out,
                       xos, ld_out);
    return check_launch();
}

}  // namespace jf

using namespace jf;

#define JF_DEFINE_MCHAIN(fam, Fam, T, suffix, CHECK)                                                                                           \
    extern "C" int jf_##fam##_chain_inv_##suffix(const T* x, int64_t xs, const T* ld_in, const T* p, int64_t ps, int32_t pb, int64_t B, int32_t n, \
                                                 const jf_##fam##_layer* L, T* xo, int64_t xos, T* ldo, const T* bi, T* bo, int64_t* bins,     \
                                                 int64_t bst, int32_t* st, void* s) {                                                          \
        CHECK                                                                                                                                  \
        return mchain<T, Fam, false>(x, xs, ld_in, p, ps, pb, B, n, L, xo, xos, ldo, bi, bo, bins, bst, st, s);                                \
    }                                                                                                                                          \
    extern "C" int jf_##fam##_chain_inv_sum_##suffix(const T* x, int64_t xs, const T* ld_in, const T* p, int64_t ps, int32_t pb, int64_t B,     \
                                                     int32_t n, const jf_##fam##_layer* L, T* xo, int64_t xos, T* ldo, const T* bi, T* bo,      \
                                                     const jf_row_list* ldp, const jf_row_list* blpp, T* tot, int64_t* bins, int64_t bst,       \
                                                     int32_t* st, void* s) {                                                                    \
        CHECK                                                                                                                                  \
        return mchain<T, Fam, false>(x, xs, ld_in, p, ps, pb, B, n, L, xo, xos, ldo, bi, bo, bins, bst, st, s, ldp, blpp, tot);                \
    }                                                                                                                                          \
    extern "C" int jf_##fam##_chain_fwd_##suffix(const T* x, int64_t xs, const T* ld_in, const T* p, int64_t ps, int32_t pb, int64_t B, int32_t n, \
                                                 const jf_##fam##_layer* L, T* xo, int64_t xos, T* ldo, const T* bi, T* bo, int64_t* bins,     \
                                                 int64_t bst, int32_t* st, void* s) {                                                          \
        CHECK                                                                                                                                  \
        return mchain<T, Fam, true>(x, xs, ld_in, p, ps, pb, B, n, L, xo, xos, ldo, bi, bo, bins, bst, st, s);                                 \
    }

#define CHECK_R if (L) for (int i = 0; i < n && i < JF_MAX_MCHAIN; ++i) if (!spline_ok(L[i].sp)) return JF_ERR_UNSUPPORTED;
#define CHECK_O CHECK_R
#define CHECK_M
#define CHECK_F                                                                                          \
    if (L) for (int i = 0; i < n && i < JF_MAX_MCHAIN; ++i) {                                                \
        if (L[i].n_vertical < 0 || L[i].n_vertical > JF_MAX_NESTED || L[i].n_circular < 0 || L[i].n_circular > JF_MAX_NESTED) return JF_ERR_UNSUPPORTED; \
        for (int j = 0; j < L[i].n_vertical; ++j) if (!spline_ok(L[i].vertical[j].sp)) return JF_ERR_UNSUPPORTED;   \
        for (int j = 0; j < L[i].n_circular; ++j) if (!spline_ok(L[i].circular[j].sp)) return JF_ERR_UNSUPPORTED;   \
    }
#define CHECK_V if (L) for (int i = 0; i < n && i < JF_MAX_MCHAIN; ++i) if (L[i].exp_map_type < 0 || L[i].exp_map_type > JF_V_SPLINES) return JF_ERR_UNSUPPORTED;

JF_DEFINE_MCHAIN(r, RFam, float, f32, CHECK_R)
JF_DEFINE_MCHAIN(r, RFam, double, f64, CHECK_R)
JF_DEFINE_MCHAIN(o, OFam, float, f32, CHECK_O)
JF_DEFINE_MCHAIN(o, OFam, double, f64, CHECK_O)
JF_DEFINE_MCHAIN(m, MFam, float, f32, CHECK_M)
JF_DEFINE_MCHAIN(m, MFam, double, f64, CHECK_M)
JF_DEFINE_MCHAIN(f, FFam, float, f32, CHECK_F)
JF_DEFINE_MCHAIN(f, FFam, double, f64, CHECK_F)
JF_DEFINE_MCHAIN(v, VFam, double, f64, CHECK_V)
#define CHECK_C if (L) for (int i = 0; i < n && i < JF_MAX_MCHAIN; ++i) if (L[i].kind < 0 || L[i].kind > 2 || L[i].kind != L[0].kind) return JF_ERR_BADARG;
JF_DEFINE_MCHAIN(c, CFam, float, f32, CHECK_C)
JF_DEFINE_MCHAIN(c, CFam, double, f64, CHECK_C)

// the reference asserts float64 for 'v' (exponential_map_s2.py:450, 493): there is no float32 oracle, hence no float32 kernel
extern "C" int jf_v_chain_inv_f32(const float*, int64_t, const float*, const float*, int64_t, int32_t, int64_t, int32_t, const jf_v_layer*, float*, int64_t,
                                  float*, const float*, float*, int64_t*, int64_t, int32_t*, void*) { return JF_ERR_UNSUPPORTED; }
extern "C" int jf_v_chain_fwd_f32(const float*, int64_t, const float*, const float*, int64_t, int32_t, int64_t, int32_t, const jf_v_layer*, float*, int64_t,
                                  float*, const float*, float*, int64_t*, int64_t, int32_t*, void*) { return JF_ERR_UNSUPPORTED; }
extern "C" int jf_v_chain_inv_sum_f32(const float*, int64_t, const float*, const float*, int64_t, int32_t, int64_t, int32_t, const jf_v_layer*, float*, int64_t,
                                      float*, const float*, float*, const jf_row_list*, const jf_row_list*, float*, int64_t*, int64_t, int32_t*, void*) {
    return JF_ERR_UNSUPPORTED;
}

extern "C" {
int jf_sphere_to_embedding_f32(const float* x, int64_t xs, const float* li, int64_t B, int32_t dim, float* xo, int64_t xos, float* lo, void* s) {
    return embed<float, true>(x, xs, li, B, dim, xo, xos, lo, s);
}
int jf_sphere_to_embedding_f64(const double* x, int64_t xs, const double* li, int64_t B, int32_t dim, double* xo, int64_t xos, double* lo, void* s) {
    return embed<double, true>(x, xs, li, B, dim, xo, xos, lo, s);
}
int jf_sphere_from_embedding_f32(const float* x, int64_t xs, const float* li, int64_t B, int32_t dim, float* xo, int64_t xos, float* lo, void* s) {
    return embed<float, false>(x, xs, li, B, dim, xo, xos, lo, s);
}
int jf_sphere_from_embedding_f64(const double* x, int64_t xs, const double* li, int64_t B, int32_t dim, double* xo, int64_t xos, double* lo, void* s) {
    return embed<double, false>(x, xs, li, B, dim, xo, xos, lo, s);
}
}
