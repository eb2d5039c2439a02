// 't' -- affine flow / multivariate normal (jammy_flows/layers/euclidean/multivariate_normal.py:54-300, layers/matrix_fns.py:4-146)
// with the euclidean_base offset (euclidean_base.py:34-76):
//   sampling  x = L z + offset,   log_det += sum_i log L_ii
//   log-prob  z = L^-1 (x - offset),  log_det -= sum_i log L_ii
// L lower triangular: log L_ii = make_log_positive(raw_i) (the same width regulators as 'g': smooth saturation / exp / softplus, optional
// clamps), strictly-lower entries stored sub-diagonal by sub-diagonal starting from the bottom-left corner (matrix_fns.py:36-50).
// The reference multiplies by an explicit inverse built from sub-determinants (matrix_fns.py:88-141); here the triangular system is solved by
// forward substitution in registers (same result; D <= 32, kernels instantiated for 8 / 16 / 32 coordinates).  One sample per lane; row: [offset D if model_offset][log-diagonal 1 | D][lower D(D-1)/2].
//   jf_t_layer_inv_* / jf_t_layer_fwd_* / jf_t_layer_inv_bwd_* (backward in forward mode on dual numbers, like the manifold chains)
#include "jf_dual.h"
#include "jf_gf.h"

namespace jf {

constexpr int T_MAXD = 32;                  // the kernels are instantiated for register arrays of 8, 16 and 32 coordinates (MD)

template <typename T> struct TDev {
    int cov, model_offset, D;
    GfLayerDev<T> w;                       // width regulator fields only (width_mode, clamp_widths, wmin, inv_wmax, lw_lo, lw_hi)
};

template <typename T> __device__ __forceinline__ T t_log_diag(const TDev<T>& o, T raw) { return M<T>::log(gf_width<T>(o.w, raw)); }

// index of L[i][j] (i > j) inside the strictly-lower block
__host__ __device__ inline int t_lower_index(int D, int i, int j) { const int ind = D - 1 - (i - j); return ind * (ind + 1) / 2 + j; }

template <typename T, bool FWD, int MD> __device__ __forceinline__ void t_apply(const TDev<T>& o, const T* __restrict__ p, int64_t pstep, T (&x)[MD], T& ld) {
    const int D = o.D;
    auto P = [&](int i) -> T { return p[i * pstep]; };
    int c = 0;
    T off[MD];
#pragma unroll
    for (int d = 0; d < MD; ++d) off[d] = (o.model_offset && d < D) ? P(d) : T(0);
    if (o.model_offset) c = D;
    if constexpr (!FWD) {
#pragma unroll
        for (int d = 0; d < MD; ++d) x[d] = x[d] - off[d];
    }
    if (o.cov == JF_T_DIAGONAL_SYMMETRIC) {
        const T s = t_log_diag<T>(o, P(c));
        const T f = M<T>::exp(FWD ? s : -s);
#pragma unroll
        for (int d = 0; d < MD; ++d) if (d < D) x[d] = x[d] * f;
        ld = FWD ? ld + s * T(D) : ld - s * T(D);
    } else if (o.cov == JF_T_DIAGONAL) {
#pragma unroll
        for (int d = 0; d < MD; ++d) if (d < D) {
            const T s = t_log_diag<T>(o, P(c + d));
            x[d] = x[d] * M<T>::exp(FWD ? s : -s);
            ld = FWD ? ld + s : ld - s;
        }
    } else if (o.cov == JF_T_FULL) {
        T s[MD];
#pragma unroll
        for (int d = 0; d < MD; ++d) s[d] = d < D ? t_log_diag<T>(o, P(c + d)) : T(0);
        const int lo = c + D;
        if constexpr (FWD) {               // x = L z, bottom row first so that z is still intact
#pragma unroll
            for (int i = MD - 1; i >= 0; --i) if (i < D) {
                T acc = x[i] * M<T>::exp(s[i]);
#pragma unroll
                for (int j = 0; j < MD; ++j) if (j < i) acc = acc + P(lo + t_lower_index(D, i, j)) * x[j];
                x[i] = acc;
                ld = ld + s[i];
            }
        } else {                           // forward substitution L z = x
#pragma unroll
            for (int i = 0; i < MD; ++i) if (i < D) {
                T acc = x[i];
#pragma unroll
                for (int j = 0; j < MD; ++j) if (j < i) acc = acc - P(lo + t_lower_index(D, i, j)) * x[j];
                x[i] = acc * M<T>::exp(-s[i]);
                ld = ld - s[i];
            }
        }
    }
    if constexpr (FWD) {
#pragma unroll
        for (int d = 0; d < MD; ++d) x[d] = x[d] + off[d];
    }
}

template <typename T> struct TArgs {
    const T* x; int64_t xs; const T* ld_in; const T* params; int64_t ps; int bcast; int64_t B; int P;
    TDev<T> o;
    T* x_out; int64_t xos; T* ld_out; const T* blp_in; T* blp_out;
    // backward
    const T* g_xout; int64_t gxos; const T* g_ld; const T* g_blp; T* g_x; int64_t gxs; T* g_params; int64_t gps;
    int32_t* status;
};

template <typename T, bool FWD, int MD> __global__ void __launch_bounds__(256) t_kernel(const TArgs<T> a) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.B) return;
    T x[MD];
#pragma unroll
    for (int d = 0; d < MD; ++d) x[d] = d < a.o.D ? a.x[row * a.xs + d] : T(0);
    T ld = a.ld_in ? a.ld_in[row] : T(0);
    const T* p = a.params ? a.params + (a.bcast ? 0 : row * a.ps) : nullptr;
    if (a.o.cov != JF_T_IDENTITY || a.o.model_offset) t_apply<T, FWD, MD>(a.o, p, 1, x, ld);
    bool bad = !M<T>::finite(ld);
    T s = a.blp_in ? a.blp_in[row] : T(0);
#pragma unroll
    for (int d = 0; d < MD; ++d) if (d < a.o.D) {
        a.x_out[row * a.xos + d] = x[d];
        bad = bad || !M<T>::finite(x[d]);
        s += T(-0.5) * x[d] * x[d] - M<T>::HALF_LN_2PI;
    }
    a.ld_out[row] = ld;
    if (a.blp_out) a.blp_out[row] = s;
    status_add(a.status, JF_STATUS_NONFINITE, bad);
}

// backward of the log-prob direction, forward mode: one pass per input direction (D coordinates + P parameters) on dual numbers
template <typename T, int MD> __global__ void __launch_bounds__(64) t_bwd_kernel(const TArgs<T> a) {
    using Du = Dual<T>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Du* tile = reinterpret_cast<Du*>(smem_raw);                       // [P][64] (per-sample) or [P] (broadcast): lane-contiguous columns
    const int tid = threadIdx.x;
    const int64_t row = (int64_t)blockIdx.x * 64 + tid;
    const bool active = row < a.B;
    const int64_t rrow = active ? row : a.B - 1;
    const int D = a.o.D;
    for (int j = 0; j < a.P; ++j) {
        if (a.bcast) { if (tid == 0) tile[j] = Du(a.params[j]); }
        else tile[j * 64 + tid] = Du(a.params[rrow * a.ps + j]);
    }
    __syncthreads();
    TDev<Du> o;
    o.cov = a.o.cov; o.model_offset = a.o.model_offset; o.D = D;
    o.w.width_mode = a.o.w.width_mode; o.w.clamp_widths = a.o.w.clamp_widths;
    o.w.wmin = Du(a.o.w.wmin); o.w.inv_wmax = Du(a.o.w.inv_wmax); o.w.lw_lo = Du(a.o.w.lw_lo); o.w.lw_hi = Du(a.o.w.lw_hi);
    T x0[MD], gxo[MD];
#pragma unroll
    for (int d = 0; d < MD; ++d) {
        x0[d] = d < D ? a.x[rrow * a.xs + d] : T(0);
        gxo[d] = (d < D && a.g_xout && active) ? a.g_xout[rrow * a.gxos + d] : T(0);
    }
    const T gld = (a.g_ld && active) ? a.g_ld[rrow] : T(0);
    const T gblp = (a.g_blp && active) ? a.g_blp[rrow] : T(0);
    const Du* p = a.bcast ? tile : tile + tid;
    const int64_t pstep = a.bcast ? 1 : 64;
    for (int j = 0; j < D + a.P; ++j) {
        if (j >= D) { if (a.bcast) { if (tid == 0) tile[j - D].d = T(1); } else tile[(j - D) * 64 + tid].d = T(1); }
        __syncthreads();
        Du x[MD];
#pragma unroll
        for (int d = 0; d < MD; ++d) x[d] = Du(x0[d], d == j ? T(1) : T(0));
        Du ld(T(0));
        if (a.o.cov != JF_T_IDENTITY || a.o.model_offset) t_apply<Du, false, MD>(o, p, pstep, x, ld);
        T gj = gld * ld.d;
#pragma unroll
        for (int d = 0; d < MD; ++d) if (d < D) gj += (gxo[d] - x[d].v * gblp) * x[d].d;
        if (!active) gj = T(0);
        if (j < D) { if (active) a.g_x[row * a.gxs + j] = gj; }
        else if (a.bcast) {
            T s = gj;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (tid == 0) atomicAdd(a.g_params + (j - D), s);
        } else if (active) a.g_params[row * a.gps + (j - D)] = gj;
        __syncthreads();
        if (j >= D) { if (a.bcast) { if (tid == 0) tile[j - D].d = T(0); } else tile[(j - D) * 64 + tid].d = T(0); }
    }
}

template <typename T> static int t_fill(TArgs<T>& a, const jf_t_layer* L, int32_t D) {
    if (!L || D < 1) return JF_ERR_BADARG;
    if (D > T_MAXD) return JF_ERR_UNSUPPORTED;
    if (L->cov_type < 0 || L->cov_type > 3 || L->width_min <= 0) return JF_ERR_BADARG;
    if (L->width_mode == JF_GF_WIDTH_SMOOTH_SATURATION && L->width_max <= 0) return JF_ERR_BADARG;
    a.o.cov = L->cov_type; a.o.model_offset = L->model_offset; a.o.D = D;
    a.o.w.width_mode = L->width_mode; a.o.w.clamp_widths = L->clamp_widths;
    a.o.w.wmin = (T)L->width_min; a.o.w.inv_wmax = L->width_max > 0 ? (T)(1.0 / L->width_max) : T(0);
    a.o.w.lw_lo = (T)log(0.01 * L->width_min);
    if (L->width_mode == JF_GF_WIDTH_SMOOTH_SATURATION) a.o.w.lw_hi = (T)(3.0 * log(L->width_max));
    else a.o.w.lw_hi = L->width_max > 0 ? (T)log(L->width_max) : (T)INFINITY;
    const int own = L->cov_type == JF_T_IDENTITY ? 0 : L->cov_type == JF_T_DIAGONAL_SYMMETRIC ? 1 : L->cov_type == JF_T_DIAGONAL ? D : D + D * (D - 1) / 2;
    a.P = own + (L->model_offset ? D : 0);
    return JF_OK;
}

template <typename T, bool FWD>
static int t_layer(const T* x, int64_t xs, const T* ld_in, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t D, const jf_t_layer* L, T* x_out,
                   int64_t xos, T* ld_out, const T* blp_in, T* blp_out, int32_t* status, void* stream) {
    if (!x || !x_out || !ld_out || B < 0 || (pb != 1 && pb != B)) return JF_ERR_BADARG;
    TArgs<T> a{};
    const int rc = t_fill<T>(a, L, D);
    if (rc != JF_OK) return rc;
    if (a.P > 0 && !params) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    a.x = x; a.xs = xs; a.ld_in = ld_in; a.params = params; a.ps = ps; a.bcast = pb == 1; a.B = B;
    a.x_out = x_out; a.xos = xos; a.ld_out = ld_out; a.blp_in = blp_in; a.blp_out = blp_out; a.status = status;
    if (D <= 8) jf::launch((t_kernel<T, FWD, 8>), dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    else if (D <= 16) jf::launch((t_kernel<T, FWD, 16>), dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    else jf::launch((t_kernel<T, FWD, 32>), dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch();
}

template <typename T>
static int t_layer_bwd(const T* x, int64_t xs, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t D, const jf_t_layer* L, const T* g_xout,
                       int64_t gxos, const T* g_ld, const T* g_blp, T* g_x, int64_t gxs, T* g_params, int64_t gps, int32_t* status, void* stream) {
    if (!x || !g_x || B < 0 || (pb != 1 && pb != B)) return JF_ERR_BADARG;
    TArgs<T> a{};
    const int rc = t_fill<T>(a, L, D);
    if (rc != JF_OK) return rc;
    if (a.P > 0 && (!params || !g_params)) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    a.x = x; a.xs = xs; a.params = params; a.ps = ps; a.bcast = pb == 1; a.B = B;
    a.g_xout = g_xout; a.gxos = gxos; a.g_ld = g_ld; a.g_blp = g_blp; a.g_x = g_x; a.gxs = gxs; a.g_params = g_params; a.gps = gps; a.status = status;
    const size_t lds = (size_t)(a.P > 0 ? a.P : 1) * (a.bcast ? 1 : 64) * sizeof(Dual<T>);
    if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;
    auto k = D <= 8 ? t_bwd_kernel<T, 8> : D <= 16 ? t_bwd_kernel<T, 16> : t_bwd_kernel<T, 32>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    jf::launch(k, dim3((unsigned)((B + 63) / 64)), dim3(64), lds, (hipStream_t)stream, a);
    return check_launch();
}

}  // namespace jf

extern "C" {
#define JF_T_DEF(T, suffix)                                                                                                                          \
    int jf_t_layer_inv_##suffix(const T* x, int64_t xs, const T* ld, const T* p, int64_t ps, int32_t pb, int64_t B, int32_t D, const jf_t_layer* L,      \
                                T* xo, int64_t xos, T* ldo, const T* bi, T* bo, int32_t* st, void* s) {                                                \
        return jf::t_layer<T, false>(x, xs, ld, p, ps, pb, B, D, L, xo, xos, ldo, bi, bo, st, s);                                                      \
    }                                                                                                                                                \
    int jf_t_layer_fwd_##suffix(const T* x, int64_t xs, const T* ld, const T* p, int64_t ps, int32_t pb, int64_t B, int32_t D, const jf_t_layer* L,      \
                                T* xo, int64_t xos, T* ldo, const T* bi, T* bo, int32_t* st, void* s) {                                                \
        return jf::t_layer<T, true>(x, xs, ld, p, ps, pb, B, D, L, xo, xos, ldo, bi, bo, st, s);                                                       \
    }                                                                                                                                                \
    int jf_t_layer_inv_bwd_##suffix(const T* x, int64_t xs, const T* p, int64_t ps, int32_t pb, int64_t B, int32_t D, const jf_t_layer* L, const T* gxo, \
                                    int64_t gxos, const T* gld, const T* gblp, T* gx, int64_t gxs, T* gp, int64_t gps, int32_t* st, void* s) {         \
        return jf::t_layer_bwd<T>(x, xs, p, ps, pb, B, D, L, gxo, gxos, gld, gblp, gx, gxs, gp, gps, st, s);                                           \
    }
JF_T_DEF(float, f32)
JF_T_DEF(double, f64)
}
