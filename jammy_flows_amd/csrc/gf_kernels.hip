// Kernels + C-ABI launchers for chains of 'g' layers (one e-block of a jammy_flows pdf in ONE launch).
//
//   jf_gf_chain_inv_*  log-prob direction   (gaussianization_flow.py:995-1114 per layer, main/default.py:998-1031 loop)
//   jf_gf_chain_fwd_*  sampling direction   (gaussianization_flow.py:911-989,  main/default.py:1482-1506 loop)
//
// Two parameter regimes:
//   broadcast (param_batch == 1, unconditional first sub-pdf): 256-thread workgroups; the chain's derived parameters
//       (<= a few KB) are prepared once per workgroup in LDS and read by every lane as LDS broadcasts.
//   per-sample (param_batch == B, the autoregressive / conditional blocks): one wave per workgroup; each layer's slab of
//       64 rows x n_params is fetched from HBM with coalesced 16-byte loads into an LDS tile (row stride = 4*odd dwords, so the
//       lane-per-row ds_read_b128 that follow are bank-conflict free), derived in place, then consumed lane-per-row.
//       This is the "coalesced HBM reads of the per-sample autoregressive parameter blocks" path; its HBM traffic is the
//       algorithmic minimum (every parameter byte is read exactly once).
#include "jf_gf.h"

namespace jf {

template <typename T> struct GfChainArgs {
    const T* x; int64_t xs;
    const T* ld_in;
    const T* params; int64_t ps;
    int64_t B;
    int n_layers;
    int rows_per_block;      // per-sample kernels: rows (<= 64) handled by one wave
    int tile_stride;         // per-sample: LDS row stride (elements); broadcast: row capacity per layer
    GfLayerDev<T> L[JF_MAX_CHAIN];
    T* x_out; int64_t xos;
    T* ld_out;
    const T* blp_in; T* blp_out;
    int32_t* status;
};

template <typename T, int D> __device__ __forceinline__ void load_x(const T* __restrict__ p, int64_t stride, int64_t row, T (&x)[D]) {
    const T* r = p + row * stride;
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = r[d];
}
template <typename T, int D> __device__ __forceinline__ void store_x(T* __restrict__ p, int64_t stride, int64_t row, const T (&x)[D]) {
    T* r = p + row * stride;
#pragma unroll
    for (int d = 0; d < D; ++d) r[d] = x[d];
}

// cooperative derive of broadcast rows: thread t handles column t (t < D) and reflection t (t < hh) of every layer
template <typename T> __device__ __forceinline__ void derive_broadcast(T* lds, const GfChainArgs<T>& a, int D) {
    const int tid = threadIdx.x;
    for (int l = 0; l < a.n_layers; ++l) {
        const GfLayerDev<T> o = a.L[l];
        for (int j = tid; j < o.n_params; j += blockDim.x) lds[l * a.tile_stride + j] = a.params[o.col0 + j];
    }
    __syncthreads();
    for (int l = 0; l < a.n_layers; ++l) {
        const GfLayerDev<T> o = a.L[l];
        T* row = lds + l * a.tile_stride;
        if (tid < D) gf_derive_column<T>(row, o, D, tid);
        else if (tid >= 64 && tid - 64 < o.hh) gf_derive_reflection<T>(row, o, D, tid - 64);
    }
    __syncthreads();
}

// stage + derive the slab of one layer for the rows of this wave (per-sample regime)
template <typename T, int D, bool DERIVE> __device__ __forceinline__ const T* stage_layer(T* lds, const GfChainArgs<T>& a, const GfLayerDev<T>& o, int64_t row0,
                                                             int valid_rows, bool lane_active) {
    const int tid = threadIdx.x;
    __syncthreads();   // previous layer's reads are done
    stage_rows<T, 18>(lds, a.tile_stride, a.params + row0 * a.ps + o.col0, a.ps, o.n_params, a.rows_per_block, valid_rows, tid, blockDim.x,
                      o.vec_ok != 0);
    __syncthreads();
    T* row = lds + (lane_active ? tid : 0) * a.tile_stride;   // idle lanes (tid >= rows_per_block) shadow row 0, read-only
    if constexpr (DERIVE) {
        if (lane_active) gf_derive_row<T, D>(row, o);
    }
    return row;
}

template <typename T, int D, bool BCAST>
__global__ void __launch_bounds__(BCAST ? 256 : 64) gf_chain_inv_kernel(const GfChainArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x;
    const int rpb = BCAST ? (int)blockDim.x : a.rows_per_block;
    const int64_t row0 = (int64_t)blockIdx.x * rpb;
    const int64_t row = row0 + tid;
    const bool active = (tid < rpb) && (row < a.B);
    const int64_t rrow = active ? row : a.B - 1;
    const int valid_rows = (int)((a.B - row0) < rpb ? (a.B - row0) : rpb);

    T x[D], y[D];
    load_x<T, D>(a.x, a.xs, rrow, x);
    T ld = a.ld_in ? a.ld_in[rrow] : T(0);
    if constexpr (BCAST) derive_broadcast<T>(lds, a, D);

    for (int l = a.n_layers - 1; l >= 0; --l) {
        const GfLayerDev<T> o = a.L[l];      // uniform index: scalar loads from the kernarg segment
        const T* prow;
        if constexpr (BCAST) prow = lds + l * a.tile_stride;
        else prow = stage_layer<T, D, false>(lds, a, o, row0, valid_rows, tid < rpb);     // raw rows: regulation fused into the mixture loop
        if (o.model_offset) {
#pragma unroll
            for (int d = 0; d < D; ++d) x[d] -= prow[d];                 // euclidean_base.py:40-45
        }
        if constexpr (BCAST) {
            gf_rotate_inv<T, D>(prow, o, x);
            ld += gf_stage<T, D>(prow, o, x, y);
        } else {
            gf_rotate_inv_raw<T, D>(prow, o, x);
            ld += gf_stage_raw<T, D>(prow, o, x, y);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = y[d];
    }
    if (active) {
        store_x<T, D>(a.x_out, a.xos, row, x);
        a.ld_out[row] = ld;
        if (a.blp_out) {
            T s = a.blp_in ? a.blp_in[row] : T(0);
#pragma unroll
            for (int d = 0; d < D; ++d) s += T(-0.5) * x[d] * x[d] - M<T>::HALF_LN_2PI;
            a.blp_out[row] = s;
        }
    }
    bool bad = !M<T>::finite(ld);
#pragma unroll
    for (int d = 0; d < D; ++d) bad = bad || !M<T>::finite(x[d]);
    status_add(a.status, JF_STATUS_NONFINITE, active && bad);
}

template <typename T, int D, bool BCAST>
__global__ void __launch_bounds__(BCAST ? 256 : 64) gf_chain_fwd_kernel(const GfChainArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x;
    const int rpb = BCAST ? (int)blockDim.x : a.rows_per_block;
    const int64_t row0 = (int64_t)blockIdx.x * rpb;
    const int64_t row = row0 + tid;
    const bool active = (tid < rpb) && (row < a.B);
    const int64_t rrow = active ? row : a.B - 1;
    const int valid_rows = (int)((a.B - row0) < rpb ? (a.B - row0) : rpb);

    T z[D], x[D], y[D], logd[D];
    load_x<T, D>(a.x, a.xs, rrow, z);
    T ld = a.ld_in ? a.ld_in[rrow] : T(0);
    if constexpr (BCAST) derive_broadcast<T>(lds, a, D);

    for (int l = 0; l < a.n_layers; ++l) {
        const GfLayerDev<T> o = a.L[l];
        const T* prow;
        if constexpr (BCAST) prow = lds + l * a.tile_stride;
        else prow = stage_layer<T, D, true>(lds, a, o, row0, valid_rows, tid < rpb);
        gf_solve<T, D>(prow, o, z, x, active, a.status);
        gf_stage_deriv<T, D>(prow, o, x, y, logd);                       // gaussianization_flow.py:922-924
#pragma unroll
        for (int d = 0; d < D; ++d) ld -= logd[d];
        gf_rotate_fwd<T, D>(prow, o, x);
        if (o.model_offset) {
#pragma unroll
            for (int d = 0; d < D; ++d) x[d] += prow[d];                 // euclidean_base.py:63-68
        }
#pragma unroll
        for (int d = 0; d < D; ++d) z[d] = x[d];
    }
    if (active) {
        store_x<T, D>(a.x_out, a.xos, row, z);
        a.ld_out[row] = ld;
    }
}

// ----------------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------------
constexpr int LDS_LIMIT = 160 * 1024;

template <typename T> static int fill_args(GfChainArgs<T>& a, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n_layers,
                                           const jf_gf_layer* layers, size_t& lds_bytes, bool& bcast) {
    if (n_layers < 1 || n_layers > JF_MAX_CHAIN || D < 1 || B < 0 || layers == nullptr) return JF_ERR_BADARG;
    if (pb != 1 && pb != B) return JF_ERR_BADARG;
    bcast = (pb == 1);
    int col = 0, maxp = 0;
    for (int l = 0; l < n_layers; ++l) {
        const jf_gf_layer& h = layers[l];
        GfLayerDev<T>& o = a.L[l];
        if (h.num_kde < 1 || h.hh_iter < 0 || h.width_min <= 0) return JF_ERR_BADARG;
        if (h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION && h.width_max <= 0) return JF_ERR_BADARG;
        o.K = h.num_kde; o.hh = h.hh_iter; o.model_offset = h.model_offset; o.fit_norm = h.fit_normalization;
        o.reg_norm = h.regulate_normalization; o.inv_type = h.inverse_function_type; o.width_mode = h.width_mode;
        o.clamp_widths = h.clamp_widths;
        const int kd = h.num_kde * D;
        o.off_rot = h.model_offset ? D : 0;
        o.off_mean = o.off_rot + h.hh_iter * D;
        o.off_lw = o.off_mean + kd;
        o.off_ln = o.off_lw + kd;
        o.n_params = o.off_ln + (h.fit_normalization ? kd : 0);
        o.col0 = col;
        o.vec_ok = (!bcast && aligned16<T>(params, ps, col) && (o.n_params % Vec16<T>::N == 0)) ? 1 : 0;
        o.wmin = (T)h.width_min; o.wmax = (T)h.width_max; o.inv_wmax = h.width_max > 0 ? (T)(1.0 / h.width_max) : T(0);
        o.nmin = (T)h.norm_min; o.nmax = (T)h.norm_max;
        o.lw_lo = (T)log(0.01 * h.width_min);                                  // gaussianization_flow.py:129
        if (h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION) o.lw_hi = (T)(3.0 * log(h.width_max));   // :121
        else o.lw_hi = h.width_max > 0 ? (T)log(h.width_max) : (T)INFINITY;   // :275, :290
        col += o.n_params;
        if (o.n_params > maxp) maxp = o.n_params;
    }
    a.params = params; a.ps = ps; a.B = B; a.n_layers = n_layers;
    if (bcast) {
        a.tile_stride = padded_stride<T>(maxp);
        a.rows_per_block = 256;
        lds_bytes = (size_t)n_layers * a.tile_stride * sizeof(T);
    } else {
        a.tile_stride = padded_stride<T>(maxp);
        int rows = 64;
        while (rows > 8 && (size_t)rows * a.tile_stride * sizeof(T) > (size_t)LDS_LIMIT) rows >>= 1;
        a.rows_per_block = rows;
        lds_bytes = (size_t)rows * a.tile_stride * sizeof(T);
    }
    if (lds_bytes > (size_t)LDS_LIMIT) return JF_ERR_UNSUPPORTED;
    return JF_OK;
}

template <typename T, int D, bool FWD> static int launch_d(const GfChainArgs<T>& a, bool bcast, size_t lds_bytes, hipStream_t st) {
    if (a.B == 0) return JF_OK;
    if (bcast) {
        auto k = FWD ? gf_chain_fwd_kernel<T, D, true> : gf_chain_inv_kernel<T, D, true>;
        if (lds_bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        const unsigned grid = (unsigned)((a.B + 255) / 256);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, st, a);
    } else {
        auto k = FWD ? gf_chain_fwd_kernel<T, D, false> : gf_chain_inv_kernel<T, D, false>;
        if (lds_bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        const unsigned grid = (unsigned)((a.B + a.rows_per_block - 1) / a.rows_per_block);
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds_bytes, st, a);
    }
    return check_launch();
}

template <typename T, bool FWD> static int launch(const GfChainArgs<T>& a, int D, bool bcast, size_t lds_bytes, hipStream_t st) {
    switch (D) {
        case 1: return launch_d<T, 1, FWD>(a, bcast, lds_bytes, st);
        case 2: return launch_d<T, 2, FWD>(a, bcast, lds_bytes, st);
        case 3: return launch_d<T, 3, FWD>(a, bcast, lds_bytes, st);
        case 4: return launch_d<T, 4, FWD>(a, bcast, lds_bytes, st);
        case 5: return launch_d<T, 5, FWD>(a, bcast, lds_bytes, st);
        case 6: return launch_d<T, 6, FWD>(a, bcast, lds_bytes, st);
        case 7: return launch_d<T, 7, FWD>(a, bcast, lds_bytes, st);
        case 8: return launch_d<T, 8, FWD>(a, bcast, lds_bytes, st);
        default: return JF_ERR_UNSUPPORTED;
    }
}

template <typename T>
static int gf_chain_inv(const T* x, int64_t xs, const T* ld_in, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n_layers,
                        const jf_gf_layer* layers, T* x_out, int64_t xos, T* ld_out, const T* blp_in, T* blp_out, int32_t* status, void* stream) {
    if (!x || !params || !x_out || !ld_out) return JF_ERR_BADARG;
    GfChainArgs<T> a{};
    size_t lds = 0; bool bcast = false;
    int rc = fill_args<T>(a, params, ps, pb, B, D, n_layers, layers, lds, bcast);
    if (rc != JF_OK) return rc;
    a.x = x; a.xs = xs; a.ld_in = ld_in; a.x_out = x_out; a.xos = xos; a.ld_out = ld_out; a.blp_in = blp_in; a.blp_out = blp_out; a.status = status;
    return launch<T, false>(a, D, bcast, lds, (hipStream_t)stream);
}
template <typename T>
static int gf_chain_fwd(const T* z, int64_t zs, const T* ld_in, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n_layers,
                        const jf_gf_layer* layers, T* x_out, int64_t xos, T* ld_out, int32_t* status, void* stream) {
    if (!z || !params || !x_out || !ld_out) return JF_ERR_BADARG;
    GfChainArgs<T> a{};
    size_t lds = 0; bool bcast = false;
    int rc = fill_args<T>(a, params, ps, pb, B, D, n_layers, layers, lds, bcast);
    if (rc != JF_OK) return rc;
    a.x = z; a.xs = zs; a.ld_in = ld_in; a.x_out = x_out; a.xos = xos; a.ld_out = ld_out; a.blp_in = nullptr; a.blp_out = nullptr; a.status = status;
    return launch<T, true>(a, D, bcast, lds, (hipStream_t)stream);
}

}  // namespace jf

extern "C" {
int jf_abi_version(void) { return 1; }

int jf_gf_chain_inv_f32(const float* x, int64_t xs, const float* ld_in, const float* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n,
                        const jf_gf_layer* L, float* xo, int64_t xos, float* ldo, const float* bi, float* bo, int32_t* st, void* s) {
    return jf::gf_chain_inv<float>(x, xs, ld_in, p, ps, pb, B, D, n, L, xo, xos, ldo, bi, bo, st, s);
}
int jf_gf_chain_inv_f64(const double* x, int64_t xs, const double* ld_in, const double* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n,
                        const jf_gf_layer* L, double* xo, int64_t xos, double* ldo, const double* bi, double* bo, int32_t* st, void* s) {
    return jf::gf_chain_inv<double>(x, xs, ld_in, p, ps, pb, B, D, n, L, xo, xos, ldo, bi, bo, st, s);
}
int jf_gf_chain_fwd_f32(const float* z, int64_t zs, const float* ld_in, const float* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n,
                        const jf_gf_layer* L, float* xo, int64_t xos, float* ldo, int32_t* st, void* s) {
    return jf::gf_chain_fwd<float>(z, zs, ld_in, p, ps, pb, B, D, n, L, xo, xos, ldo, st, s);
}
int jf_gf_chain_fwd_f64(const double* z, int64_t zs, const double* ld_in, const double* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n,
                        const jf_gf_layer* L, double* xo, int64_t xos, double* ldo, int32_t* st, void* s) {
    return jf::gf_chain_fwd<double>(z, zs, ld_in, p, ps, pb, B, D, n, L, xo, xos, ldo, st, s);
}
}
