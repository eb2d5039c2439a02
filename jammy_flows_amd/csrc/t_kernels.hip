// 't' -- affine flow / multivariate normal (jammy_flows/layers/euclidean/multivariate_normal.py:54-300, layers/matrix_fns.py:4-146)
// with the euclidean_base offset (euclidean_base.py:34-76):
//   sampling  x = L z + offset,   log_det += sum_i log L_ii
//   log-prob  z = L^-1 (x - offset),  log_det -= sum_i log L_ii
// L lower triangular: log L_ii = make_log_positive(raw_i) (the same width regulators as 'g': smooth saturation / exp / softplus, optional
// clamps), strictly-lower entries stored sub-diagonal by sub-diagonal starting from the bottom-left corner (matrix_fns.py:36-50).
// The reference multiplies by an explicit inverse built from sub-determinants (matrix_fns.py:88-141); here the triangular system is solved by
// forward substitution in registers (same result; D <= 32, kernels instantiated for 8 / 16 / 32 coordinates).  One sample per lane; row: [offset D if model_offset][log-diagonal 1 | D][lower D(D-1)/2].
//   jf_t_layer_inv_* / jf_t_layer_fwd_* / jf_t_layer_inv_bwd_* (backward: one forward + one backward substitution, see t_bwd_kernel)
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

// backward of the log-prob direction in REVERSE mode (round 5; until then one dual-number pass per input direction: 75 passes for a full
// 10 x 10 covariance, 50 x the forward).  z = L^-1 (x - offset) by forward substitution, then one backward substitution with the upstream
// gradients zb_d = g_x_out[d] - z_d g_base_logp and g_log_det:
//     z_i = acc_i e^{-s_i},  acc_i = x_i - off_i - sum_{j<i} L_ij z_j,  log_det -= s_i      (s_i = log of the regulated diagonal)
//     acc_b_i = zb_i e^{-s_i};   s_b_i = -zb_i z_i - g_log_det;   x_b_i = acc_b_i;   off_b_i = -acc_b_i;   L_b_ij = -acc_b_i z_j;   zb_j -= acc_b_i L_ij
// walked from the last row up; d s_i / d raw_i comes from one single-tangent evaluation of the width regulator per diagonal entry.
// Per-sample parameters: g_params (B, P) rows.  Broadcast parameters: the workgroup's sums over ALL its row tiles in LDS, added to g_params
// once at the end -- a resident set of workgroups walks the tiles (manifold_bwd_kernels.hip: one atomic per 64-row tile and parameter was
// most of such a kernel's time).
template <typename T, int MD> __global__ void __launch_bounds__(64) t_bwd_kernel(const TArgs<T> a) {
    using Du = Dual<T>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);                         // [P][64] (per-sample) or [P] (broadcast): lane-contiguous columns
    T* accp = tile + (a.P > 0 ? a.P : 1) * (a.bcast ? 1 : 64);         // broadcast: [P] gradient sums
    const int tid = threadIdx.x;
    const int D = a.o.D, cov = a.o.cov;
    if (a.bcast) {
        for (int j = tid; j < a.P; j += 64) { tile[j] = a.params[j]; accp[j] = T(0); }
    } else {
        const int64_t row = (int64_t)blockIdx.x * 64 + tid;
        const int64_t rrow = row < a.B ? row : a.B - 1;
        for (int j = 0; j < a.P; ++j) tile[j * 64 + tid] = a.params[rrow * a.ps + j];
    }
    __syncthreads();
    TDev<Du> od;                                                      // the width regulator on single duals: d log(diagonal) / d raw
    od.cov = cov; od.model_offset = a.o.model_offset; od.D = D;
    od.w.width_mode = a.o.w.width_mode; od.w.clamp_widths = a.o.w.clamp_widths;
    od.w.wmin = Du(a.o.w.wmin); od.w.inv_wmax = Du(a.o.w.inv_wmax); od.w.lw_lo = Du(a.o.w.lw_lo); od.w.lw_hi = Du(a.o.w.lw_hi);
    const T* p = a.bcast ? tile : tile + tid;
    const int64_t pstep = a.bcast ? 1 : 64;
    auto P = [&](int i) -> T { return p[i * pstep]; };
    const int c0 = a.o.model_offset ? D : 0;                          // first diagonal parameter
    const int n_diag = cov == JF_T_IDENTITY ? 0 : cov == JF_T_DIAGONAL_SYMMETRIC ? 1 : D;
    const int lo = c0 + D;                                            // first strictly-lower entry (full covariance)
    const int64_t n_tiles = (a.B + 63) / 64;
    const int64_t t_end = a.bcast ? n_tiles : (int64_t)blockIdx.x + 1;    // per-sample parameters: the tile staged above
    for (int64_t tile_i = blockIdx.x; tile_i < t_end; tile_i += gridDim.x) {
        const int64_t row = tile_i * 64 + tid;
        const bool active = row < a.B;
        const int64_t rrow = active ? row : a.B - 1;
        // a parameter's gradient of this row: stored (per-sample) or summed over the wave into the workgroup's accumulators (broadcast)
        auto emit = [&](int j, T v) {
            if (!active) v = T(0);
            if (a.bcast) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                if (tid == 0) accp[j] += v;
            } else if (active) {
                a.g_params[row * a.gps + j] = v;
            }
        };
        T z[MD], zb[MD];
#pragma unroll
        for (int d = 0; d < MD; ++d) z[d] = d < D ? a.x[rrow * a.xs + d] : T(0);
        T ld = T(0);
        if (cov != JF_T_IDENTITY || a.o.model_offset) t_apply<T, false, MD>(a.o, p, pstep, z, ld);
        const T gld = (a.g_ld && active) ? a.g_ld[rrow] : T(0);
        const T gblp = (a.g_blp && active) ? a.g_blp[rrow] : T(0);
#pragma unroll
        for (int d = 0; d < MD; ++d) zb[d] = d < D ? ((a.g_xout && active) ? a.g_xout[rrow * a.gxos + d] : T(0)) - z[d] * gblp : T(0);
        // d log(diagonal_i) / d raw_i and e^{-s_i}
        T sb_sym = T(0);
        if (cov == JF_T_FULL) {
#pragma unroll
            for (int i = MD - 1; i >= 0; --i) if (i < D) {
                const Du s = t_log_diag<Du>(od, Du(P(c0 + i), T(1)));
                const T ab = zb[i] * M<T>::exp(-s.v);              // acc_b_i = x_b_i
                emit(c0 + i, (-zb[i] * z[i] - gld) * s.d);
#pragma unroll
                for (int j = 0; j < MD; ++j) if (j < i) {
                    const int li = lo + t_lower_index(D, i, j);
                    emit(li, -ab * z[j]);
                    zb[j] -= ab * P(li);
                }
                zb[i] = ab;
            }
        } else if (cov == JF_T_DIAGONAL) {
#pragma unroll
            for (int i = 0; i < MD; ++i) if (i < D) {
                const Du s = t_log_diag<Du>(od, Du(P(c0 + i), T(1)));
                emit(c0 + i, (-zb[i] * z[i] - gld) * s.d);
                zb[i] *= M<T>::exp(-s.v);
            }
        } else if (cov == JF_T_DIAGONAL_SYMMETRIC) {
            const Du s = t_log_diag<Du>(od, Du(P(c0), T(1)));
            const T f = M<T>::exp(-s.v);
#pragma unroll
            for (int i = 0; i < MD; ++i) if (i < D) { sb_sym -= zb[i] * z[i]; zb[i] *= f; }
            emit(c0, (sb_sym - gld * T(D)) * s.d);
        }
        (void)n_diag;
        // zb is now d S / d (x - offset)
#pragma unroll
        for (int d = 0; d < MD; ++d) if (d < D) {
            if (active) a.g_x[row * a.gxs + d] = zb[d];
            if (a.o.model_offset) emit(d, -zb[d]);
        }
    }
    if (a.bcast) {
        __syncthreads();
        for (int j = tid; j < a.P; j += 64) atomicAdd(a.g_params + j, accp[j]);
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
    const size_t lds = ((size_t)(a.P > 0 ? a.P : 1) * (a.bcast ? 1 : 64) + (a.bcast ? (size_t)a.P : 0)) * sizeof(T);
    if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;
    auto k = D <= 8 ? t_bwd_kernel<T, 8> : D <= 16 ? t_bwd_kernel<T, 16> : t_bwd_kernel<T, 32>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int64_t grid = (B + 63) / 64;
    if (a.bcast) {                                                 // a resident set of workgroups walks the tiles (eight per CU: the kernel is light)
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (grid > (int64_t)cus * 8) grid = (int64_t)cus * 8;
    }
    jf::launch(k, dim3((unsigned)grid), dim3(64), lds, (hipStream_t)stream, a);
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
