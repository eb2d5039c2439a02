// Backward (vector-Jacobian product) of the manifold-layer chains in the log-prob direction: jf_{r,o,m,f,v,c}_chain_inv_bwd_*.
//
// What torch.autograd returns for the per-block layer loop of all_layer_inverse (main/default.py:998-1031) over 'r' / 'o' / 'm' / 'f' /
// 'v' layers and the sphere / interval base-class steps, given upstream gradients of (x_out, log_det_out, base_logp_out).
//
// These layers carry few parameters per sample (8 ... 60; 426 for the correlated 'f'), so the backward is taken in FORWARD mode: the
// kernel evaluates the chain once per input direction -- each target coordinate and each parameter of the row -- on dual numbers
// (jf_dual.h), i.e. through the SAME device code the forward kernels instantiate (splines with their bin searches, Moebius / exponential
// map Newton iterations, charts, clamps), and contracts the output tangents with the upstream gradients:
//     g_in[j] = sum_d g_x_out[d] dx_out[d]/d in_j + g_log_det d log_det/d in_j       (+ the base log-prob term -x_out g_base_logp).
// One wave per workgroup, one sample per lane, parameters of ALL layers of the chain staged once as dual rows in LDS.
// Per-sample parameters: g_params (B, P).  Broadcast parameters: the row sums are accumulated into g_params (1, P) with one atomic add per
// workgroup and parameter (the caller zero-initialises it).
#include <type_traits>

#include "jf_dual.h"
#include "jf_expmap.h"
#include "jf_manifold.h"

namespace jf {

template <typename T, typename CLayer> struct MBwdArgs {
    const T* x; int64_t xs;
    const T* params; int64_t ps;
    int bcast;
    int64_t B;
    int n_layers, dim, P, tile_stride, scratch, rows;
    int col0[JF_MAX_MCHAIN];
    CLayer L[JF_MAX_MCHAIN];
    const T* g_xout; int64_t gxos;
    const T* g_ld; const T* g_blp;
    T* g_x; int64_t gxs;
    T* g_params; int64_t gps;
    int32_t* status;
};

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename T, class Fam>
__global__ void __launch_bounds__(64) mchain_bwd_kernel(const MBwdArgs<T, typename Fam::CLayer> a) {
    using Du = Dual<T>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Du* tile = reinterpret_cast<Du*>(smem_raw);
    const int tid = threadIdx.x;
    const int rows = a.rows;
    const int tile_rows = a.bcast ? 1 : rows;
    const bool lane_in = tid < rows;
    const int slot = lane_in ? tid : 0;
    Du* tab = tile + tile_rows * a.tile_stride + slot * JF_SPLINE_TAB;
    Du* corr = tile + tile_rows * a.tile_stride + rows * JF_SPLINE_TAB + slot * a.scratch;
    const int64_t row0 = (int64_t)blockIdx.x * rows;
    const int64_t row = row0 + tid;
    const bool active = lane_in && row < a.B;
    const int64_t rrow = active ? row : a.B - 1;

    // parameters of all layers -> dual rows (tangent 0)
    if (a.bcast) {
        for (int j = tid; j < a.P; j += 64) tile[j] = Du(a.params[j]);
    } else {
        for (int r = 0; r < rows; ++r) {
            const int64_t gr = (row0 + r) < a.B ? (row0 + r) : a.B - 1;
            for (int j = tid; j < a.P; j += 64) tile[r * a.tile_stride + j] = Du(a.params[gr * a.ps + j]);
        }
    }
    __syncthreads();
    Du* prow = tile + (a.bcast ? 0 : slot * a.tile_stride);

    T x0[3] = {T(0), T(0), T(0)};
#pragma unroll
    for (int d = 0; d < Fam::DIM; ++d) if (d < a.dim) x0[d] = a.x[rrow * a.xs + d];
    T gxo[3] = {T(0), T(0), T(0)};
#pragma unroll
    for (int d = 0; d < Fam::DIM; ++d) if (d < a.dim && a.g_xout && active) gxo[d] = a.g_xout[rrow * a.gxos + d];
    const T gld = (a.g_ld && active) ? a.g_ld[rrow] : T(0);
    const T gblp = (a.g_blp && active) ? a.g_blp[rrow] : T(0);

    bool bad = false;
    const int n_dir = a.dim + a.P;
    for (int j = 0; j < n_dir; ++j) {
        // ---- seed direction j
        if (j >= a.dim) {
            if (a.bcast) { if (tid == 0) prow[j - a.dim].d = T(1); }
            else if (lane_in) prow[j - a.dim].d = T(1);
        }
        __syncthreads();
        Du x[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) x[d] = Du(x0[d], (d == j) ? T(1) : T(0));
        Du ld(T(0));
        LaneCtx<Du> ctx;
        ctx.tab = tab; ctx.corr = corr; ctx.bins = nullptr; ctx.bin_i = 0;
        ctx.oob = ctx.nonconv = ctx.nonfinite = false;
        ctx.lane_valid = active;
        for (int i = 0; i < a.n_layers; ++i) {
            const int l = a.n_layers - 1 - i;
            if (lane_in) Fam::template apply<Du, false>(a.L[l], prow + a.col0[l], x, ld, ctx);
        }
        T gj = gld * ld.d;
#pragma unroll
        for (int d = 0; d < Fam::DIM; ++d) if (d < a.dim) gj += (gxo[d] - x[d].v * gblp) * x[d].d;
        if (!active) gj = T(0);
        bad = bad || !M<T>::finite(gj);
        if (j < a.dim) {
            if (active) a.g_x[row * a.gxs + j] = gj;
        } else if (a.bcast) {
            const T s = wave_sum<T>(gj);
            if (tid == 0) atomicAdd(a.g_params + (j - a.dim), s);
        } else if (active) {
            a.g_params[row * a.gps + (j - a.dim)] = gj;
        }
        __syncthreads();
        // ---- unseed
        if (j >= a.dim) {
            if (a.bcast) { if (tid == 0) prow[j - a.dim].d = T(0); }
            else if (lane_in) prow[j - a.dim].d = T(0);
        }
    }
    status_add(a.status, JF_STATUS_NONFINITE, active && bad);
}

template <typename T, class Fam>
static int mchain_bwd(const T* x, int64_t xs, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t n_layers, const typename Fam::CLayer* layers,
                      const T* g_xout, int64_t gxos, const T* g_ld, const T* g_blp, T* g_x, int64_t gxs, T* g_params, int64_t gps, int32_t* status,
                      void* stream) {
    if (!x || !g_x || !layers || n_layers < 1 || n_layers > JF_MAX_MCHAIN || B < 0) return JF_ERR_BADARG;
    if (pb != 1 && pb != B) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    MBwdArgs<T, typename Fam::CLayer> a{};
    int col = 0;
    for (int l = 0; l < n_layers; ++l) {
        a.L[l] = layers[l];
        a.col0[l] = col;
        col += Fam::row_len(layers[l]);
    }
    if (col > 0 && (!params || !g_params)) return JF_ERR_BADARG;
    a.x = x; a.xs = xs; a.params = params; a.ps = ps; a.bcast = (pb == 1) ? 1 : 0; a.B = B; a.n_layers = n_layers; a.P = col;
    a.tile_stride = col > 0 ? col : 1;
    a.dim = Fam::DIM;
    if constexpr (std::is_same<Fam, CFam>::value) a.dim = layers[0].kind == 2 ? 2 : 1;
    a.g_xout = g_xout; a.gxos = gxos; a.g_ld = g_ld; a.g_blp = g_blp; a.g_x = g_x; a.gxs = gxs; a.g_params = g_params; a.gps = gps; a.status = status;
    a.scratch = 0;
    if constexpr (std::is_same<Fam, FFam>::value) {
        for (int l = 0; l < n_layers; ++l) {
            if (!layers[l].correlated) continue;
            if (layers[l].corr_hidden < 1 || layers[l].corr_rank < 0 || FFam::corr_out(layers[l]) + layers[l].corr_rank > JF_CORR_SCRATCH - 1)
                return JF_ERR_UNSUPPORTED;
            a.scratch = JF_CORR_SCRATCH;
        }
    }
    a.rows = 64;
    size_t lds = 0;
    for (;;) {
        lds = ((size_t)(a.bcast ? 1 : a.rows) * a.tile_stride + (size_t)a.rows * (JF_SPLINE_TAB + a.scratch)) * sizeof(Dual<T>);
        if (lds <= 160 * 1024 || a.rows == 4) break;
        a.rows >>= 1;
    }
    if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;
    auto k = mchain_bwd_kernel<T, Fam>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)((B + a.rows - 1) / a.rows)), dim3(64), lds, (hipStream_t)stream, a);
    return check_launch();
}

}  // namespace jf

using namespace jf;

#define JF_DEFINE_MCHAIN_BWD(fam, Fam, T, suffix)                                                                                              \
    extern "C" int jf_##fam##_chain_inv_bwd_##suffix(const T* x, int64_t xs, const T* p, int64_t ps, int32_t pb, int64_t B, int32_t n,             \
                                                     const jf_##fam##_layer* L, const T* gxo, int64_t gxos, const T* gld, const T* gblp, T* gx,   \
                                                     int64_t gxs, T* gp, int64_t gps, int32_t* st, void* s) {                                   \
        return mchain_bwd<T, Fam>(x, xs, p, ps, pb, B, n, L, gxo, gxos, gld, gblp, gx, gxs, gp, gps, st, s);                                     \
    }
JF_DEFINE_MCHAIN_BWD(r, RFam, float, f32)
JF_DEFINE_MCHAIN_BWD(r, RFam, double, f64)
JF_DEFINE_MCHAIN_BWD(o, OFam, float, f32)
JF_DEFINE_MCHAIN_BWD(o, OFam, double, f64)
JF_DEFINE_MCHAIN_BWD(m, MFam, float, f32)
JF_DEFINE_MCHAIN_BWD(m, MFam, double, f64)
JF_DEFINE_MCHAIN_BWD(f, FFam, float, f32)
JF_DEFINE_MCHAIN_BWD(f, FFam, double, f64)
JF_DEFINE_MCHAIN_BWD(v, VFam, double, f64)
JF_DEFINE_MCHAIN_BWD(c, CFam, float, f32)
JF_DEFINE_MCHAIN_BWD(c, CFam, double, f64)
extern "C" int jf_v_chain_inv_bwd_f32(const float*, int64_t, const float*, int64_t, int32_t, int64_t, int32_t, const jf_v_layer*, const float*, int64_t,
                                      const float*, const float*, float*, int64_t, float*, int64_t, int32_t*, void*) { return JF_ERR_UNSUPPORTED; }
