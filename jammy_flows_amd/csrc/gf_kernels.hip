// Kernels + C-ABI launchers for chains of 'g' layers (one e-block of a jammy_flows pdf in ONE launch).
//
//   jf_gf_chain_inv_*  log-prob direction   (gaussianization_flow.py:995-1114 per layer, main/default.py:998-1031 loop)
//   jf_gf_chain_fwd_*  sampling direction   (gaussianization_flow.py:911-989,  main/default.py:1482-1506 loop)
//
// Work distribution (jf_gf.h): lane = (row, coordinate); the G = next-power-of-two(D) lanes of a group own one row, a wave
// owns 64/G rows.  Per-coordinate arithmetic is scalar code per lane, reductions over the coordinates are DPP butterflies.
//
// Two parameter regimes:
//   broadcast (param_batch == 1, unconditional first sub-pdf): 256-thread workgroups; the chain's derived parameters
//       (<= a few KB) are prepared once per workgroup in LDS (one wave per layer) and the workgroup then walks several row tiles.
//   per-sample (param_batch == B, the autoregressive / conditional blocks): one wave per workgroup, no workgroup barrier coupling
//       different waves; each layer's slab of (64/G rows) x n_params is fetched from HBM with coalesced 16-byte loads issued back to
//       back (stage_rows) into an LDS tile whose row stride is 4*odd dwords, then read by the lanes of each row.  A wave's tile is a few
//       KB, so 16+ waves per CU are resident and the HBM latency of one wave's slab is covered by the arithmetic of the others.
//       HBM traffic is the algorithmic minimum (every parameter byte is read exactly once).
#include "jf_gfb.h"
#include "jf_merge.h"

namespace jf {


// ---- the start table of the broadcast sampler.  With row-independent parameters the solve of (layer l, coordinate d) inverts ONE fixed monotone
// function for every row: x_{l,d}(z).  gf_fwd_table_kernel solves it on GT_N + 1 knots of z in [-zmax, zmax] (plus the interval midpoints),
// keeps each interval's cubic Hermite polynomial (values and slopes dx/dz = 1 / (dy/dx) at its ends) and marks the intervals whose polynomial
// misses the solved midpoint by more than 2e-5, or over which x(z) is far from linear (gaps between distant components, where x(z) is nearly a step).  A lane of the sampler whose z
// falls into a good interval starts the reference's Newton stage from the polynomial's value and skips the approach phase (4-6 of its ~9
// mixture evaluations); every other lane takes the approach phase as before.  16 B per interval (float32): 128 KB for C3's block 0, L2-resident.
constexpr int GT_N = 512;
template <typename T> __host__ __device__ inline T gt_zmax(int inv_type) { return inv_type == JF_GF_ISIGMOID ? T(20) : T(8); }
template <typename T> __host__ __device__ inline bool gt_layer_ok(const GfLayerDev<T>& o) {
    return o.K == CS_K && o.fit_norm && o.stretch == JF_GF_STRETCH_CLASSIC;
}

// grid (layer x coordinate, GT_N / GT_CH): a workgroup of 192 threads solves the 65 knots (wave 0 + the first lane of wave 2) and 64 midpoints
// (wave 1) of GT_CH = 64 intervals -- one solve per thread, so the table costs one solve's latency, not a serial walk over the knots
constexpr int GT_CH = 64;
template <typename T> __global__ void __launch_bounds__(192) gf_fwd_table_kernel(const GfChainArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* row = reinterpret_cast<T*>(smem_raw);
    T* xs = row + a.tile_stride;                          // [GT_CH + 1] solutions at the knots
    T* hm = xs + GT_CH + 1;                               // [GT_CH + 1] h dx/dz at the knots
    T* xm = hm + GT_CH + 1;                               // [GT_CH] solutions at the interval midpoints
    const int tid = threadIdx.x;
    const int D = a.D, l = blockIdx.x / D, d = blockIdx.x - l * D, i0 = blockIdx.y * GT_CH;
    const GfLayerDev<T> o = a.L[l];
    T* out = a.table + ((size_t)blockIdx.x * GT_N + i0) * 4;
    if (!gt_layer_ok<T>(o)) {                              // block-uniform: the sampler does not consult the table for this layer
        for (int i = tid; i < GT_CH * 4; i += 192) out[i] = T(NAN);
        return;
    }
    for (int j = tid; j < o.n_params; j += 192) row[j] = a.params[o.col0 + j];
    __syncthreads();
    if (tid == 0) gf_derive_column<T>(row, o, D, d);
    __syncthreads();
    const T zmax = gt_zmax<T>(o.inv_type), h = T(2) * zmax / T(GT_N);
    const bool knot = tid < GT_CH || tid == 2 * GT_CH, mid = tid >= GT_CH && tid < 2 * GT_CH;
    if (knot || mid) {                                     // wave-uniform except for the last wave's single lane
        T R[CS_SLOTS];
#pragma unroll
        for (int k = 0; k < CS_K; ++k) {
            R[CS_SLOT_MEAN + k] = row[o.off_mean + k * D + d];
            R[CS_SLOT_LW + k] = row[o.off_lw + k * D + d];
            R[CS_SLOT_LN + k] = row[o.off_ln + k * D + d];
        }
        const int i = knot ? (tid < GT_CH ? tid : GT_CH) : tid - GT_CH;
        const T z = -zmax + (T(i0 + i) + (mid ? T(0.5) : T(0))) * h;
        CsSolveInfo info;
        T logd;
        T x = cs_solve<T>(R, o.inv_type, true, z, true, false, nullptr, [](T v) { return v; }, [](T v) { return v; }, &info, &logd);
        if (info.nonconv || info.nonfinite) x = T(NAN);
        if (knot) { xs[i] = x; hm[i] = h * M<T>::exp(-logd); }
        else xm[i] = x;
    }
    __syncthreads();
    if (tid < GT_CH) {
        const int i = tid;
        const T dx = xs[i + 1] - xs[i];
        T c0 = xs[i];
        const T c1 = hm[i], c2 = T(3) * dx - T(2) * hm[i] - hm[i + 1], c3 = T(-2) * dx + hm[i] + hm[i + 1];
        const T pm = c0 + T(0.5) * c1 + T(0.25) * c2 + T(0.125) * c3;
        const T tol = T(2e-5) * M<T>::max(T(1), M<T>::abs(xm[i]));
        // usable: the polynomial reproduces the solved midpoint AND x(z) is close to linear over the interval (both end slopes within a factor
        // 1.25 of the secant) -- on the flank of a plateau (a gap between distant components, where x(z) is nearly a step) a cubic can hit the
        // midpoint and still be off elsewhere, and a Newton stage started there ended above the reference's convergence threshold on 11 rows of
        // 40 000 in the rough-mixture test; those intervals go to the approach phase
        const bool regular = dx > T(0) && hm[i] > T(0) && hm[i + 1] > T(0) && hm[i] < T(1.25) * dx && dx < T(1.25) * hm[i] && hm[i + 1] < T(1.25) * dx &&
                             dx < T(1.25) * hm[i + 1];
        if (!(M<T>::abs(pm - xm[i]) <= tol) || !regular || !M<T>::finite(c2) || !M<T>::finite(c3)) c0 = T(NAN);     // (false also for a NaN anywhere)
        out[4 * i + 0] = c0; out[4 * i + 1] = c1; out[4 * i + 2] = c2; out[4 * i + 3] = c3;
    }
}


template <typename T, int G, bool BCAST, bool FWD>
__global__ void __launch_bounds__(BCAST ? 256 : 64, sizeof(T) == 8 ? 2 : 4) gf_chain_kernel(const GfChainArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    constexpr int NT = BCAST ? 256 : 64;
    constexpr int R = NT / G;                           // rows per tile
    const int tid = threadIdx.x;
    const int g = tid & (G - 1), r = tid >> Log2<G>::v;
    const int D = a.D;
    const bool live = g < D, leader = g == 0;
    const int d = live ? g : D - 1;
    T* spl_tab = lds + a.tab_offset;                    // lane-private knot tables (only sized when a layer uses rq_splines)
    if constexpr (BCAST) derive_broadcast<T>(lds, a);

    const int tiles = BCAST ? a.tiles_per_block : 1;
    for (int t = 0; t < tiles; ++t) {
        const int64_t row0 = ((int64_t)blockIdx.x * tiles + t) * R;
        if (row0 >= a.B) break;                          // block-uniform
        const int64_t row = row0 + r;
        const bool row_valid = row < a.B;
        const int64_t rrow = row_valid ? row : a.B - 1;
        const int valid_rows = (int)((a.B - row0) < R ? (a.B - row0) : R);

        T x = a.x[rrow * a.xs + d];
        T ld = a.ld_in ? a.ld_in[rrow] : T(0);
        T cv = T(0);
        if constexpr (!FWD) { if (a.cot_in) cv = live ? a.cot_in[rrow * a.cis + d] : T(0); }

        int spline_calls = 0;
        for (int li = 0; li < a.n_layers; ++li) {
            const int l = FWD ? li : a.n_layers - 1 - li;
            const GfLayerDev<T> o = a.L[l];              // uniform index: scalar loads from the kernarg segment
            const T* p;
            if constexpr (BCAST) {
                p = lds + l * a.tile_stride + d;
            } else {
                __syncthreads();                         // single-wave workgroup: orders the previous layer's LDS reads before the refill
                if (G >= 4 && o.vec_ok == 2)                   // aligned rows of <= 20 pieces per lane: constant-offset staging
                    stage_rows_grouped<T, (G >= 4 ? G : 4)>(lds, a.tile_stride, a.params + row0 * a.ps + o.col0, a.ps, o.n_params, valid_rows, tid);
                else
                    stage_rows<T>(lds, a.tile_stride, a.params + row0 * a.ps + o.col0, a.ps, o.n_params, R, valid_rows, tid, NT, o.vec_ok != 0);
                __syncthreads();
                if constexpr (FWD) {                     // 45 evaluations per layer follow: regulate the row once, in place
                    gfg_derive<T, G>(lds + r * a.tile_stride, o, D, g, o.stretch == JF_GF_STRETCH_CLASSIC);
                    __syncthreads();
                }
                p = lds + r * a.tile_stride + d;
            }
            if (o.stretch == JF_GF_STRETCH_RQ_SPLINES) {
                // per-dimension spline with learnable box and linear tails (gaussianization_flow.py:926-940, 1060-1068); lane-private knot table
                const T* pr = p - d;                                                         // start of the lane's row
                T* tab = spl_tab + tid * a.spline_tab;
                if constexpr (!FWD) {
                    if (o.model_offset) x -= p[0];
                    x = gfg_rotate_inv<T, G, !BCAST>(p, o, D, live, x);
                    if (a.cot_in) cv = gfg_rotate_inv<T, G, !BCAST>(p, o, D, live, cv);
                }
                const SplineOut<T> r = spline_linext<T>(pr + o.off_mean + d * o.K, pr + o.off_lw + d * o.K, pr + o.off_ln + d * (o.K + 1),
                                                        pr + o.off_box + d * 4, o.K, tab, x, FWD);
                x = r.y;
                ld += group_sum<T, G>(live ? r.lad : T(0));
                if constexpr (!FWD) { if (a.cot_in) cv *= M<T>::exp(-r.lad); }
                if (a.bins != nullptr && row_valid && live) a.bins[row * a.bins_stride + spline_calls * D + d] = (int64_t)r.bin;
                ++spline_calls;
                if constexpr (FWD) {
                    x = gfg_rotate_fwd<T, G, false>(p, o, D, live, x);
                    if (o.model_offset) x += p[0];
                }
            } else if constexpr (!FWD) {
                if (o.model_offset) x -= p[0];                                               // euclidean_base.py:40-45
                x = gfg_rotate_inv<T, G, !BCAST>(p, o, D, live, x);
                const MixQ<T> q = gfg_mixture<T, !BCAST>(p, o, D, x);
                const IcdfOut<T> s = gf_icdf<T>(o.inv_type, q);
                x = s.y;
                ld += group_sum<T, G>(live ? s.logd : T(0));
                if (a.cot_in) cv = gfg_rotate_inv<T, G, !BCAST>(p, o, D, live, cv) * M<T>::exp(-s.logd);
            } else {
                if (o.K == CS_K && o.fit_norm) {
                    // ten components (the reference's default): the lane's derived column -- mean, 1 / width, weight of every component -- goes
                    // into registers once, and the 25 + <= 20 mixture evaluations of the solve read registers instead of three LDS words per
                    // component and evaluation (same arithmetic: cs_solve restates gfg_solve on a register row, jf_cond_regs.h)
                    T R[CS_SLOTS];
#pragma unroll
                    for (int k = 0; k < CS_K; ++k) {
                        R[CS_SLOT_MEAN + k] = p[o.off_mean + k * D];
                        R[CS_SLOT_LW + k] = p[o.off_lw + k * D];
                        R[CS_SLOT_LN + k] = p[o.off_ln + k * D];
                    }
                    T slogd;
                    bool have = false;
                    T xstart = T(0);
                    if constexpr (BCAST) {
                        if (a.table != nullptr) {                  // uniform
                            const T zmax = gt_zmax<T>(o.inv_type);
                            const T tq = (x + zmax) * (T(GT_N) / (T(2) * zmax));
                            if (tq >= T(0) && tq < T(GT_N)) {
                                const int iq = (int)tq;
                                const T fr = tq - T(iq);
                                const T* c = a.table + ((size_t)(l * D + d) * GT_N + iq) * 4;
                                xstart = c[0] + fr * (c[1] + fr * (c[2] + fr * c[3]));
                                have = live && M<T>::finite(xstart);
                            }
                        }
                    }
                    x = cs_solve<T>(R, o.inv_type, live, x, row_valid, leader, a.status, [](T v) { return group_sum<T, G>(v); },
                                    [](T v) { return group_max<T, G>(v); }, nullptr, &slogd, have, xstart);
                    ld -= group_sum<T, G>(live ? slogd : T(0));
                } else {
                    x = gfg_solve<T, G>(p, o, D, live, x, row_valid, leader, a.status);
                    const MixQ<T> q = gfg_mixture<T, false>(p, o, D, x);                            // gaussianization_flow.py:922-924
                    ld -= group_sum<T, G>(live ? gf_icdf<T>(o.inv_type, q).logd : T(0));
                }
                x = gfg_rotate_fwd<T, G, false>(p, o, D, live, x);
                if (o.model_offset) x += p[0];                                               // euclidean_base.py:63-68
            }
        }
        if (row_valid && live) a.x_out[row * a.xos + d] = x;
        if constexpr (!FWD) {
            if (a.cot_out && row_valid && live) a.cot_out[row * a.cos + d] = cv;
            T s = T(0);
            if (a.blp_out) s = group_sum<T, G>(live ? T(-0.5) * x * x - M<T>::HALF_LN_2PI : T(0));
            if (row_valid && leader) {
                a.ld_out[row] = ld;
                const T bv = s + (a.blp_in ? a.blp_in[row] : T(0));
                if (a.blp_out) a.blp_out[row] = bv;
                if (a.total) a.total[row] = bv + ld;
            }
            const T bad = group_max<T, G>((live && !M<T>::finite(x)) ? T(1) : T(0));
            status_add(a.status, JF_STATUS_NONFINITE, row_valid && leader && (bad > T(0) || !M<T>::finite(ld)));
        } else {
            if (row_valid && leader) a.ld_out[row] = ld;
        }
    }
}

// the lane = row broadcast kernel (body: jf_gfb.h)
template <typename T, int D> __global__ void __launch_bounds__(256) gfb_chain_inv_kernel(const GfChainArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    gfb_chain_inv_body<T, D>(a, (int)blockIdx.x, smem_raw);
}
// the same chain with G lanes per row (small batches; bit-identical results: jf_gfb.h)
template <typename T, int D, int G> __global__ void __launch_bounds__(256) gfbg_chain_inv_kernel(const GfChainArgs<T> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    gfbg_chain_inv_body<T, D, G>(a, (int)blockIdx.x, smem_raw);
}


// General-option chain (jf_gf_ext.h): one lane per row, every option of the layer; launched when a layer of the chain uses a rotation other
// than Householder reflections, center_mean or add_skewness.  LDS: two coordinate columns per lane (+ the lane's spline knot table).
template <typename T, bool FWD> __global__ void __launch_bounds__(GX_THREADS) gfx_chain_kernel(const GfChainArgs<T> a, const int64_t pstep) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x, D = a.D;
    const XCol<T> x{lds + tid}, z{lds + JF_MAX_D_GF * GX_THREADS + tid};
    T* tab = lds + 2 * JF_MAX_D_GF * GX_THREADS + tid * a.spline_tab;
    const int64_t row = (int64_t)blockIdx.x * GX_THREADS + tid;
    const bool row_valid = row < a.B;
    const int64_t rrow = row_valid ? row : a.B - 1;
    for (int d = 0; d < D; ++d) x[d] = a.x[rrow * a.xs + d];
    T ld = a.ld_in ? a.ld_in[rrow] : T(0);
    int spline_calls = 0;
    for (int li = 0; li < a.n_layers; ++li) {
        const int l = FWD ? li : a.n_layers - 1 - li;
        const GfLayerDev<T> o = a.L[l];
        const T* p = a.params + rrow * pstep + o.col0;
        if constexpr (!FWD) {
            if (o.model_offset) for (int d = 0; d < D; ++d) x[d] -= p[d];                       // euclidean_base.py:40-45
            gx_rotate<T, const T*>(o, p, x, D, true);
        }
        if (o.stretch == JF_GF_STRETCH_RQ_SPLINES) {
            for (int d = 0; d < D; ++d) {
                const SplineOut<T> r = spline_linext<T>(p + o.off_mean + d * o.K, p + o.off_lw + d * o.K, p + o.off_ln + d * (o.K + 1),
                                                        p + o.off_box + d * 4, o.K, tab, x[d], FWD);
                x[d] = r.y;
                ld += r.lad;
                if (a.bins != nullptr && row_valid) a.bins[row * a.bins_stride + spline_calls * D + d] = (int64_t)r.bin;
            }
            ++spline_calls;
        } else if constexpr (!FWD) {
            for (int d = 0; d < D; ++d) {
                const GxCoord<T> c = gx_prepare<T, const T*>(o, p, D, d);
                const IcdfOut<T> s = gf_icdf<T>(o.inv_type, gx_mixture<T, const T*>(o, p, D, d, c, x[d]));
                x[d] = s.y;
                ld += s.logd;
            }
        } else {
            for (int d = 0; d < D; ++d) z[d] = x[d];
            gx_solve<T, const T*>(o, p, D, z, x, row_valid, a.status);
            for (int d = 0; d < D; ++d) {                                                          // gaussianization_flow.py:922-924
                const GxCoord<T> c = gx_prepare<T, const T*>(o, p, D, d);
                ld -= gf_icdf<T>(o.inv_type, gx_mixture<T, const T*>(o, p, D, d, c, x[d])).logd;
            }
        }
        if constexpr (FWD) {
            gx_rotate<T, const T*>(o, p, x, D, false);
            if (o.model_offset) for (int d = 0; d < D; ++d) x[d] += p[d];                       // euclidean_base.py:63-68
        }
    }
    bool bad = !M<T>::finite(ld);
    T blp = T(0);
    for (int d = 0; d < D; ++d) {
        const T v = x[d];
        if (row_valid) a.x_out[row * a.xos + d] = v;
        bad = bad || !M<T>::finite(v);
        blp += T(-0.5) * v * v - M<T>::HALF_LN_2PI;
    }
    if (row_valid) {
        a.ld_out[row] = ld;
        if (!FWD && a.blp_out) {
            const T bv = blp + (a.blp_in ? a.blp_in[row] : T(0));
            a.blp_out[row] = bv;
            if (a.total) a.total[row] = bv + ld;
        }
    }
    if constexpr (!FWD) status_add(a.status, JF_STATUS_NONFINITE, row_valid && bad);
}

// ----------------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------------
constexpr int LDS_LIMIT = 160 * 1024;

static inline int group_width(int D) { return D <= 1 ? 1 : D <= 2 ? 2 : D <= 4 ? 4 : D <= 8 ? 8 : D <= 16 ? 16 : D <= 32 ? 32 : 64; }

template <typename T> static int fill_args(GfChainArgs<T>& a, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n_layers,
                                           const jf_gf_layer* layers, size_t& lds_bytes, bool& bcast, bool& ext) {
    ext = false;
    if (n_layers < 1 || n_layers > JF_MAX_CHAIN || D < 1 || B < 0 || layers == nullptr) return JF_ERR_BADARG;
    if (D > JF_MAX_D_G) return JF_ERR_UNSUPPORTED;
    if (pb != 1 && pb != B) return JF_ERR_BADARG;
    bcast = (pb == 1);
    int col = 0, maxp = 0;
    bool any_spline = false;
    for (int l = 0; l < n_layers; ++l) {
        const jf_gf_layer& h = layers[l];
        GfLayerDev<T>& o = a.L[l];
        if (h.num_kde < 1 || h.num_kde > (1 << 16) || h.hh_iter < 0 || h.hh_iter > (1 << 16) || h.width_min <= 0) return JF_ERR_BADARG;   // (bounded: the column offsets below are ints)
        if (h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION && h.width_max <= 0) return JF_ERR_BADARG;
        o.K = h.num_kde; o.hh = h.hh_iter; o.model_offset = h.model_offset; o.fit_norm = h.fit_normalization;
        o.reg_norm = h.regulate_normalization; o.inv_type = h.inverse_function_type; o.width_mode = h.width_mode;
        o.clamp_widths = h.clamp_widths;
        o.fast = (h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION && !h.clamp_widths && h.fit_normalization && h.regulate_normalization) ? 1 : 0;
        const int kd = h.num_kde * D;
        o.stretch = h.nonlinear_stretch_type;
        if (o.stretch != JF_GF_STRETCH_CLASSIC && o.stretch != JF_GF_STRETCH_RQ_SPLINES) return JF_ERR_BADARG;
        o.rot_mode = h.rotation_mode; o.center_mean = h.center_mean ? 1 : 0; o.skew = h.add_skewness ? 1 : 0; o.off_skew = 0;
        if (o.rot_mode < JF_GF_ROT_HOUSEHOLDER || o.rot_mode > JF_GF_ROT_TRIANGULAR) return JF_ERR_BADARG;
        if (o.rot_mode == JF_GF_ROT_CAYLEY && D > 2) return JF_ERR_BADARG;                       // "Cayley requires 2 dims at the moment" (:220)
        if (o.rot_mode != JF_GF_ROT_HOUSEHOLDER) o.hh = 0;
        if (o.skew && sizeof(T) != 8) return JF_ERR_UNSUPPORTED;                                 // double precision only (extra_functions.py:28)
        if ((o.center_mean || o.skew) && o.stretch != JF_GF_STRETCH_CLASSIC) return JF_ERR_BADARG;
        if (o.center_mean && h.num_kde < 2) return JF_ERR_BADARG;
        if (o.rot_mode != JF_GF_ROT_HOUSEHOLDER || o.center_mean || o.skew) ext = true;
        if (ext && D > JF_MAX_D_GF) return JF_ERR_UNSUPPORTED;                                 // the general-option kernel keeps a row's coordinates in an LDS column of 8
        o.off_rot = h.model_offset ? D : 0;
        o.off_mean = o.off_rot + gx_rot_len(o.rot_mode, o.hh, D);
        o.off_lw = o.off_mean + kd - (o.center_mean ? D : 0);
        o.off_ln = o.off_lw + kd;
        if (o.stretch == JF_GF_STRETCH_RQ_SPLINES) {
            if (h.num_kde > JF_SPLINE_CAP) return JF_ERR_UNSUPPORTED;           // (round 6: 16 -> 64 bins; a lane's table follows the chain's own bin count)
            o.off_box = o.off_ln + (h.num_kde + 1) * D;
            o.n_params = o.off_box + 4 * D;
            any_spline = true;
            if (spline_tab_words(h.num_kde) > a.spline_tab) a.spline_tab = spline_tab_words(h.num_kde);
        } else {
            o.off_box = 0;
            o.off_skew = o.off_ln + (h.fit_normalization ? kd : 0);
            o.n_params = o.off_skew + (o.skew ? kd : 0);
        }
        o.col0 = col;
        o.vec_ok = (!bcast && aligned16<T>(params, ps, col) && (o.n_params % Vec16<T>::N == 0)) ? 1 : 0;
        {   // 2: the grouped constant-offset staging applies (G lanes per row, at most JF_GROUP_STAGE_MAX_PIECES pieces per lane)
            const int Gw = group_width(D), nvp = o.n_params / Vec16<T>::N;
            if (o.vec_ok && Gw >= 4 && (nvp + Gw - 1) / Gw <= JF_GROUP_STAGE_MAX_PIECES) o.vec_ok = 2;
        }
        o.wmin = (T)h.width_min; o.wmax = (T)h.width_max; o.inv_wmax = h.width_max > 0 ? (T)(1.0 / h.width_max) : T(0);
        o.nmin = (T)h.norm_min; o.nmax = (T)h.norm_max;
        o.lw_lo = (T)log(0.01 * h.width_min);                                  // gaussianization_flow.py:129
        if (h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION) o.lw_hi = (T)(3.0 * log(h.width_max));   // :121
        else o.lw_hi = h.width_max > 0 ? (T)log(h.width_max) : (T)INFINITY;   // :275, :290
        col += o.n_params;
        if (o.n_params > maxp) maxp = o.n_params;
    }
    a.params = params; a.ps = ps; a.B = B; a.n_layers = n_layers; a.D = D;
    a.tile_stride = padded_stride<T>(maxp);
    const int G = group_width(D);
    size_t elems = bcast ? (size_t)n_layers * a.tile_stride : (size_t)(64 / G) * a.tile_stride;
    a.tab_offset = (int)elems;
    if (any_spline) elems += (size_t)(bcast ? 256 : 64) * a.spline_tab;
    if (bcast && any_spline && !ext && elems * sizeof(T) > (size_t)LDS_LIMIT) {
        // permanent parameters with many spline bins: the broadcast kernel's 256 lane-private knot tables do not fit a CU (32 bins in float64:
        // 203 KB).  The per-sample kernel (64 lanes per workgroup) takes the launch with a parameter row stride of 0: every row reads the one row.
        bcast = false;
        a.ps = 0;
        for (int l = 0; l < n_layers; ++l) a.L[l].vec_ok = 0;
        elems = (size_t)(64 / G) * a.tile_stride;
        a.tab_offset = (int)elems;
        elems += (size_t)64 * a.spline_tab;
    }
    lds_bytes = elems * sizeof(T);
    if (ext) lds_bytes = ((size_t)2 * JF_MAX_D_GF * GX_THREADS + (any_spline ? (size_t)GX_THREADS * a.spline_tab : 0)) * sizeof(T);
    a.tiles_per_block = 1;                               // broadcast: set by launch_g from the kernel's occupancy
    if (lds_bytes > (size_t)LDS_LIMIT) return JF_ERR_UNSUPPORTED;
    return JF_OK;
}

// resident workgroups of a broadcast kernel on the whole device (occupancy x CUs), queried once per kernel
template <typename K> static int resident_blocks(K k, size_t lds_bytes) {
    int dev = 0, cus = 256, per_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 256, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 4;
    return cus * per_cu;
}

template <typename T, int G, bool FWD> static int launch_g(GfChainArgs<T> a, bool bcast, size_t lds_bytes, hipStream_t st) {
    if (bcast) {
        auto k = gf_chain_kernel<T, G, true, FWD>;
        if (lds_bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        // one wave of workgroups: every workgroup derives the parameters once and walks ceil(tiles / resident) row tiles, so the grid has
        // no partially filled last round (a 2.3-round grid idles a quarter of the chip in its tail)
        int resident = resident_blocks(k, lds_bytes);       // for the CURRENT device; two cheap runtime queries, no cross-thread cache
        if (resident < 1) resident = 1;
        const int64_t n_tiles = (a.B + 256 / G - 1) / (256 / G);
        const int64_t tpb = (n_tiles + resident - 1) / resident;
        a.tiles_per_block = (int)(tpb < 1 ? 1 : tpb);
        const unsigned grid = (unsigned)((n_tiles + a.tiles_per_block - 1) / a.tiles_per_block);
        jf::launch(k, dim3(grid), dim3(256), lds_bytes, st, a);
    } else {
        auto k = gf_chain_kernel<T, G, false, FWD>;
        if (lds_bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        const unsigned grid = (unsigned)((a.B + 64 / G - 1) / (64 / G));
        jf::launch(k, dim3(grid), dim3(64), lds_bytes, st, a);
    }
    return check_launch();
}

// rows below which the broadcast log-prob chain runs with one lane per (row, coordinate) instead of one lane per row (D = 2 .. 4): the two
// give bit-identical rows (jf_gfb.h), the first has G times the waves and 1 / G of the dependent chain per lane -- C3's block 0 at 2^15 rows
// 0.029 -> 0.022 ms, at 2^13 0.032 -> 0.023; from 2^16 rows on lane = row wins (its component records are wave-uniform LDS reads, G times
// fewer of them: 0.030 vs 0.035 ms at 2^17 rows, 0.123 vs 0.146 at 2^20).  JF_GFB_LANE_ROWS overrides (0: always lane = row)
static int64_t gfbg_forced_rows = -1;                   // jf_gf_bcast_lane_rows()
static int64_t gfbg_max_rows() {
    static const int64_t v = getenv("JF_GFB_LANE_ROWS") ? atoll(getenv("JF_GFB_LANE_ROWS")) : ((int64_t)1 << 16);
    return gfbg_forced_rows >= 0 ? gfbg_forced_rows : v;
}
template <typename T, int D, int G> static int launch_rows_g(GfChainArgs<T> a, size_t lds_bytes, hipStream_t st) {
    auto k = gfbg_chain_inv_kernel<T, D, G>;
    if (lds_bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    int resident = resident_blocks(k, lds_bytes);
    if (resident < 1) resident = 1;
    const int64_t n_tiles = (a.B + 256 / G - 1) / (256 / G);
    const int64_t tpb = (n_tiles + resident - 1) / resident;
    a.tiles_per_block = (int)(tpb < 1 ? 1 : tpb);
    jf::launch(k, dim3((unsigned)((n_tiles + a.tiles_per_block - 1) / a.tiles_per_block)), dim3(256), lds_bytes, st, a);
    return check_launch();
}

template <typename T, int D> static int launch_rows(GfChainArgs<T> a, size_t lds_bytes, hipStream_t st) {
    if constexpr (D >= 2 && D <= 4) {
        if (a.B < gfbg_max_rows()) return launch_rows_g<T, D, (D == 2 ? 2 : 4)>(a, lds_bytes, st);
    }
    auto k = gfb_chain_inv_kernel<T, D>;
    if (lds_bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    int resident = resident_blocks(k, lds_bytes);           // one wave of workgroups, as for the lane = (row, coordinate) broadcast kernel
    if (resident < 1) resident = 1;
    const int64_t n_tiles = (a.B + 255) / 256;
    const int64_t tpb = (n_tiles + resident - 1) / resident;
    a.tiles_per_block = (int)(tpb < 1 ? 1 : tpb);
    jf::launch(k, dim3((unsigned)((n_tiles + a.tiles_per_block - 1) / a.tiles_per_block)), dim3(256), lds_bytes, st, a);
    return check_launch();
}

template <typename T, bool FWD> static int launch(const GfChainArgs<T>& a, int D, bool bcast, bool ext, size_t lds_bytes, hipStream_t st) {
    if (a.B == 0) return JF_OK;
    if constexpr (!FWD) {
        bool classic = bcast && !ext && a.cot_in == nullptr;       // (the lane = row kernel does not carry the co-vector)
        int pack_elems = 0;
        for (int l = 0; l < a.n_layers; ++l) {
            classic = classic && a.L[l].stretch == JF_GF_STRETCH_CLASSIC;
            pack_elems = a.L[l].K > pack_elems ? a.L[l].K : pack_elems;
        }
        const size_t lds_rows = ((size_t)a.tab_offset + (size_t)a.n_layers * pack_elems * D * 4) * sizeof(T);
        // lane = row (gfb_chain_inv_kernel) or, for small batches of 2 .. 4 dimensions, one lane per (row, coordinate) with bit-identical rows
        // (gfbg_chain_inv_kernel: launch_rows picks, jf_gfb.h explains)
        if (classic && D <= 8 && lds_rows <= (size_t)LDS_LIMIT) {
            switch (D) {
                case 1: return launch_rows<T, 1>(a, lds_rows, st);
                case 2: return launch_rows<T, 2>(a, lds_rows, st);
                case 3: return launch_rows<T, 3>(a, lds_rows, st);
                case 4: return launch_rows<T, 4>(a, lds_rows, st);
                case 5: return launch_rows<T, 5>(a, lds_rows, st);
                case 6: return launch_rows<T, 6>(a, lds_rows, st);
                case 7: return launch_rows<T, 7>(a, lds_rows, st);
                default: return launch_rows<T, 8>(a, lds_rows, st);
            }
        }
    }
    if (ext && a.cot_in != nullptr) return JF_ERR_UNSUPPORTED;
    if (ext) {
        auto k = gfx_chain_kernel<T, FWD>;
        if (lds_bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        jf::launch(k, dim3((unsigned)((a.B + GX_THREADS - 1) / GX_THREADS)), dim3(GX_THREADS), lds_bytes, st, a, bcast ? (int64_t)0 : a.ps);
        return check_launch();
    }
    switch (group_width(D)) {
        case 1: return launch_g<T, 1, FWD>(a, bcast, lds_bytes, st);
        case 2: return launch_g<T, 2, FWD>(a, bcast, lds_bytes, st);
        case 4: return launch_g<T, 4, FWD>(a, bcast, lds_bytes, st);
        case 8: return launch_g<T, 8, FWD>(a, bcast, lds_bytes, st);
        case 16: return launch_g<T, 16, FWD>(a, bcast, lds_bytes, st);
        case 32: return launch_g<T, 32, FWD>(a, bcast, lds_bytes, st);
        default: return launch_g<T, 64, FWD>(a, bcast, lds_bytes, st);
    }
}

template <typename T>
static int gf_chain_inv(const T* x, int64_t xs, const T* ld_in, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n_layers,
                        const jf_gf_layer* layers, T* x_out, int64_t xos, T* ld_out, const T* blp_in, T* blp_out, int64_t* bins, int64_t bins_stride,
                        int32_t* status, void* stream, T* total = nullptr) {
    if (!x || !params || !x_out || !ld_out || (total && !blp_out)) return JF_ERR_BADARG;
    GfChainArgs<T> a{};
    size_t lds = 0; bool bcast = false, ext = false;
    int rc = fill_args<T>(a, params, ps, pb, B, D, n_layers, layers, lds, bcast, ext);
    if (rc != JF_OK) return rc;
    a.x = x; a.xs = xs; a.ld_in = ld_in; a.x_out = x_out; a.xos = xos; a.ld_out = ld_out; a.blp_in = blp_in; a.blp_out = blp_out; a.status = status;
    a.bins = bins; a.bins_stride = bins_stride; a.total = total;
    return launch<T, false>(a, D, bcast, ext, lds, (hipStream_t)stream);
}
// J^{-T} cot for the Jacobian J = d x_out / d x of the log-prob direction (the chain is evaluated along the way: x_out, ld_out are scratch outputs)
template <typename T>
static int gf_chain_inv_cot(const T* x, int64_t xs, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n_layers,
                            const jf_gf_layer* layers, const T* cot_in, int64_t cis, T* cot_out, int64_t cos, T* x_out, int64_t xos, T* ld_out,
                            void* stream) {
    if (!x || !params || !x_out || !ld_out || !cot_in || !cot_out) return JF_ERR_BADARG;
    GfChainArgs<T> a{};
    size_t lds = 0; bool bcast = false, ext = false;
    int rc = fill_args<T>(a, params, ps, pb, B, D, n_layers, layers, lds, bcast, ext);
    if (rc != JF_OK) return rc;
    a.x = x; a.xs = xs; a.x_out = x_out; a.xos = xos; a.ld_out = ld_out;
    a.cot_in = cot_in; a.cis = cis; a.cot_out = cot_out; a.cos = cos;
    return launch<T, false>(a, D, bcast, ext, lds, (hipStream_t)stream);
}
template <typename T>
static int gf_chain_fwd(const T* z, int64_t zs, const T* ld_in, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n_layers,
                        const jf_gf_layer* layers, T* x_out, int64_t xos, T* ld_out, int64_t* bins, int64_t bins_stride, int32_t* status,
                        void* stream, T* table = nullptr) {
    if (!z || !params || !x_out || !ld_out) return JF_ERR_BADARG;
    GfChainArgs<T> a{};
    size_t lds = 0; bool bcast = false, ext = false;
    int rc = fill_args<T>(a, params, ps, pb, B, D, n_layers, layers, lds, bcast, ext);
    if (rc != JF_OK) return rc;
    a.x = z; a.xs = zs; a.ld_in = ld_in; a.x_out = x_out; a.xos = xos; a.ld_out = ld_out; a.blp_in = nullptr; a.blp_out = nullptr; a.status = status;
    a.bins = bins; a.bins_stride = bins_stride;
    if (table != nullptr && bcast && !ext && B > 0) {          // (per-sample parameters / the general-option kernel: the table is not used)
        bool any = false;
        for (int l = 0; l < n_layers; ++l) any = any || gt_layer_ok<T>(a.L[l]);
        if (any) {
            a.table = table;
            const size_t tl = ((size_t)a.tile_stride + 3 * (GT_CH + 1)) * sizeof(T);
            jf::launch(gf_fwd_table_kernel<T>, dim3((unsigned)(n_layers * D), GT_N / GT_CH), dim3(192), tl, (hipStream_t)stream, a);
            rc = check_launch();
            if (rc != JF_OK) return rc;
        }
    }
    return launch<T, true>(a, D, bcast, ext, lds, (hipStream_t)stream);
}

const void* gfbg_inv_kernel_f32(int D) {
    switch (D) {
        case 2: return (const void*)gfbg_chain_inv_kernel<float, 2, 2>;
        case 3: return (const void*)gfbg_chain_inv_kernel<float, 3, 4>;
        case 4: return (const void*)gfbg_chain_inv_kernel<float, 4, 4>;
        default: return nullptr;
    }
}
const void* gfb_inv_kernel_f32(int D) {
    switch (D) {
        case 1: return (const void*)gfb_chain_inv_kernel<float, 1>;
        case 2: return (const void*)gfb_chain_inv_kernel<float, 2>;
        case 3: return (const void*)gfb_chain_inv_kernel<float, 3>;
        case 4: return (const void*)gfb_chain_inv_kernel<float, 4>;
        case 5: return (const void*)gfb_chain_inv_kernel<float, 5>;
        case 6: return (const void*)gfb_chain_inv_kernel<float, 6>;
        case 7: return (const void*)gfb_chain_inv_kernel<float, 7>;
        case 8: return (const void*)gfb_chain_inv_kernel<float, 8>;
        default: return nullptr;
    }
}

// LDS bytes a log-prob / sampling launch of this chain needs, or a negative JF_ERR_* (JF_ERR_UNSUPPORTED: more than a CU has -- cut the chain)
template <typename T> static int64_t gf_lds_query(int32_t D, int32_t n_layers, const jf_gf_layer* layers, int bcast) {
    GfChainArgs<T> a{};
    size_t lds = 0; bool b = false, ext = false;
    const int rc = fill_args<T>(a, nullptr, 0, bcast ? 1 : 2, 2, D, n_layers, layers, lds, b, ext);
    return rc != JF_OK ? (int64_t)rc : (int64_t)lds;
}

}  // namespace jf

extern "C" {
int jf_abi_version(void) { return 8; }   // v8 (round 6): jf_get_newton_rule (which solver rule the library was built with: libjammy_hip_audit.so); v7: jf_merge_*, jf_gf_bcast_lane_rows
int64_t jf_gf_bcast_lane_rows(int64_t rows) {
    const int64_t prev = jf::gfbg_max_rows();
    jf::gfbg_forced_rows = rows;
    return prev;
}
int64_t jf_gf_chain_lds_bytes_f32(int32_t D, int32_t n, const jf_gf_layer* L, int32_t pb1) { return jf::gf_lds_query<float>(D, n, L, pb1); }
int64_t jf_gf_chain_lds_bytes_f64(int32_t D, int32_t n, const jf_gf_layer* L, int32_t pb1) { return jf::gf_lds_query<double>(D, n, L, pb1); }

int jf_gf_chain_inv_f32(const float* x, int64_t xs, const float* ld_in, const float* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n,
                        const jf_gf_layer* L, float* xo, int64_t xos, float* ldo, const float* bi, float* bo, int64_t* bins, int64_t bs, int32_t* st,
                        void* s) {
    return jf::gf_chain_inv<float>(x, xs, ld_in, p, ps, pb, B, D, n, L, xo, xos, ldo, bi, bo, bins, bs, st, s);
}
int jf_gf_chain_inv_f64(const double* x, int64_t xs, const double* ld_in, const double* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n,
                        const jf_gf_layer* L, double* xo, int64_t xos, double* ldo, const double* bi, double* bo, int64_t* bins, int64_t bs,
                        int32_t* st, void* s) {
    return jf::gf_chain_inv<double>(x, xs, ld_in, p, ps, pb, B, D, n, L, xo, xos, ldo, bi, bo, bins, bs, st, s);
}
int jf_gf_chain_inv_total_f32(const float* x, int64_t xs, const float* ld_in, const float* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n,
                              const jf_gf_layer* L, float* xo, int64_t xos, float* ldo, const float* bi, float* bo, float* total, int64_t* bins,
                              int64_t bs, int32_t* st, void* s) {
    if (!total) return JF_ERR_BADARG;
    return jf::gf_chain_inv<float>(x, xs, ld_in, p, ps, pb, B, D, n, L, xo, xos, ldo, bi, bo, bins, bs, st, s, total);
}
int jf_gf_chain_inv_total_f64(const double* x, int64_t xs, const double* ld_in, const double* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n,
                              const jf_gf_layer* L, double* xo, int64_t xos, double* ldo, const double* bi, double* bo, double* total, int64_t* bins,
                              int64_t bs, int32_t* st, void* s) {
    if (!total) return JF_ERR_BADARG;
    return jf::gf_chain_inv<double>(x, xs, ld_in, p, ps, pb, B, D, n, L, xo, xos, ldo, bi, bo, bins, bs, st, s, total);
}
int jf_gf_chain_inv_cot_f32(const float* x, int64_t xs, const float* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n, const jf_gf_layer* L,
                            const float* ci, int64_t cis, float* co, int64_t cos, float* xo, int64_t xos, float* ldo, void* s) {
    return jf::gf_chain_inv_cot<float>(x, xs, p, ps, pb, B, D, n, L, ci, cis, co, cos, xo, xos, ldo, s);
}
int jf_gf_chain_inv_cot_f64(const double* x, int64_t xs, const double* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n, const jf_gf_layer* L,
                            const double* ci, int64_t cis, double* co, int64_t cos, double* xo, int64_t xos, double* ldo, void* s) {
    return jf::gf_chain_inv_cot<double>(x, xs, p, ps, pb, B, D, n, L, ci, cis, co, cos, xo, xos, ldo, s);
}
int jf_gf_chain_fwd_f32(const float* z, int64_t zs, const float* ld_in, const float* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n,
                        const jf_gf_layer* L, float* xo, int64_t xos, float* ldo, int64_t* bins, int64_t bs, int32_t* st, void* s) {
    return jf::gf_chain_fwd<float>(z, zs, ld_in, p, ps, pb, B, D, n, L, xo, xos, ldo, bins, bs, st, s);
}
int jf_gf_chain_fwd_f64(const double* z, int64_t zs, const double* ld_in, const double* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n,
                        const jf_gf_layer* L, double* xo, int64_t xos, double* ldo, int64_t* bins, int64_t bs, int32_t* st, void* s) {
    return jf::gf_chain_fwd<double>(z, zs, ld_in, p, ps, pb, B, D, n, L, xo, xos, ldo, bins, bs, st, s);
}
int64_t jf_gf_chain_fwd_table_elems(int32_t D, int32_t n) {
    if (D < 1 || D > jf::JF_MAX_D_G || n < 1 || n > JF_MAX_CHAIN) return JF_ERR_BADARG;
    return (int64_t)D * n * jf::GT_N * 4;
}
int jf_gf_chain_fwd_tab_f32(const float* z, int64_t zs, const float* ld_in, const float* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n,
                            const jf_gf_layer* L, float* xo, int64_t xos, float* ldo, int64_t* bins, int64_t bs, int32_t* st, float* table, void* s) {
    if (!table) return JF_ERR_BADARG;
    return jf::gf_chain_fwd<float>(z, zs, ld_in, p, ps, pb, B, D, n, L, xo, xos, ldo, bins, bs, st, s, table);
}
int jf_gf_chain_fwd_tab_f64(const double* z, int64_t zs, const double* ld_in, const double* p, int64_t ps, int32_t pb, int64_t B, int32_t D, int32_t n,
                            const jf_gf_layer* L, double* xo, int64_t xos, double* ldo, int64_t* bins, int64_t bs, int32_t* st, double* table, void* s) {
    if (!table) return JF_ERR_BADARG;
    return jf::gf_chain_fwd<double>(z, zs, ld_in, p, ps, pb, B, D, n, L, xo, xos, ldo, bins, bs, st, s, table);
}
}
