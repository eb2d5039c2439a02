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
#include <cstdlib>
#include <type_traits>

#include "jf_dual.h"
#include "jf_expmap.h"
#include "jf_manifold.h"
#include "jf_manifold_rev.h"
#include "jf_spline_adj.h"

namespace jf {

template <typename T, typename CLayer> struct MBwdArgs {
    const T* x; int64_t xs;
    const T* params; int64_t ps;
    int bcast;
    int64_t B;
    int n_layers, dim, P, tile_stride, scratch, rows, tab;       // tab: JF_SPLINE_TAB or 0 (no spline in the chain), as in manifold_kernels.hip
    int rot_max;                                                 // staged 'v' kernel: longest rotation row of the chain (lane-private dual copy)
    int v_dual;                                                  // staged 'v' kernel: dual-number replay for EVERY potential (JF_V_BWD_DUAL: the check of the closed form)
    int shared_tab;                                              // generic kernel, broadcast parameters, families with a build(): the layers' knot tables once per workgroup and pass
    int col0[JF_MAX_MCHAIN];
    CLayer L[JF_MAX_MCHAIN];
    const T* g_xout; int64_t gxos;
    const T* g_ld; const T* g_blp;
    T* g_x; int64_t gxs;
    T* g_params; int64_t gps;
    int32_t* status;
};

__device__ __forceinline__ int VFam_row_len_dev(const jf_v_layer& L) {
    const int n_pot = L.exp_map_type == JF_V_SPLINES ? 4 + 3 * JF_V_SPLINE_BINS + 1 : 3 + (L.exp_map_type == JF_V_EXPONENTIAL ? 2 : 1);
    return rot_len(L.hh_iter, 3) + n_pot * L.num_components;
}

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <class Fam, class = void> struct bwd_has_build : std::false_type {};
template <class Fam> struct bwd_has_build<Fam, std::void_t<decltype(Fam::HAS_BUILD)>> : std::true_type {};

// Broadcast parameters of a family whose knot table does not depend on the row ('r'): a pass's dual tables are the same for EVERY row, and
// building them (softmax + cumulative sums + softplus on N-tangent duals: ~3 bins transcendentals per layer) is most of a pass.  The resident
// workgroup therefore takes the directions OUTSIDE its row tiles: seed, build each layer's table once (lane l builds layer l), then walk the
// tiles with the closed-form bin evaluation only.  (Per lane the saving is the build of every tile but the first; with four tiles per
// workgroup at 2^18 rows: C4's `r` adjoint 0.246 -> see DESIGN 3.9.)
template <typename T, class Fam, int N>
__device__ __forceinline__ void mchain_bwd_shared_tab(const MBwdArgs<T, typename Fam::CLayer>& a, unsigned char* smem_raw) {
    using Du = DualN<T, N>;
    Du* tile = reinterpret_cast<Du*>(smem_raw);                      // [tile_stride] dual parameter row (all layers)
    Du* stab = tile + a.tile_stride;                                 // [n_layers][a.tab] the pass's knot tables
    const int tid = threadIdx.x;
    const int rows = a.rows;
    const bool lane_in = tid < rows;
    const int slot = lane_in ? tid : 0;
    Du* corr = stab + a.n_layers * a.tab + slot * a.scratch;
    T* accp = reinterpret_cast<T*>(stab + a.n_layers * a.tab + rows * a.scratch);
    for (int j = tid; j < a.P; j += 64) { tile[j] = Du(a.params[j]); accp[j] = T(0); }
    __syncthreads();
    bool bad_any = false;
    const int64_t n_tiles = (a.B + rows - 1) / rows;
    const int n_dir = a.dim + a.P;
    for (int j0 = 0; j0 < n_dir; j0 += N) {
        if (tid == 0) {
#pragma unroll
            for (int c = 0; c < N; ++c) if (j0 + c >= a.dim && j0 + c < n_dir) tile[j0 + c - a.dim].d[c] = T(1);
        }
        __syncthreads();
        if (tid < a.n_layers) Fam::template build<Du>(a.L[tid], tile + a.col0[tid], stab + tid * a.tab);
        __syncthreads();
        for (int64_t tile_i = blockIdx.x; tile_i < n_tiles; tile_i += gridDim.x) {
            const int64_t row = tile_i * rows + tid;
            const bool active = lane_in && row < a.B;
            const int64_t rrow = active ? row : a.B - 1;
            T gxo[3] = {T(0), T(0), T(0)};
            Du x[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                x[d] = Du((d < Fam::DIM && d < a.dim) ? a.x[rrow * a.xs + d] : T(0));
                if (d < Fam::DIM && d < a.dim && a.g_xout && active) gxo[d] = a.g_xout[rrow * a.gxos + d];
#pragma unroll
                for (int c = 0; c < N; ++c) if (d == j0 + c) x[d].d[c] = T(1);
            }
            const T gld = (a.g_ld && active) ? a.g_ld[rrow] : T(0);
            const T gblp = (a.g_blp && active) ? a.g_blp[rrow] : T(0);
            Du ld(T(0));
            LaneCtx<Du> ctx;
            ctx.corr = corr; ctx.bins = nullptr; ctx.bin_i = 0;
            ctx.oob = ctx.nonconv = ctx.nonfinite = false;
            ctx.lane_valid = active;
            ctx.tab_built = true;
            for (int i = 0; i < a.n_layers; ++i) {
                const int l = a.n_layers - 1 - i;
                ctx.tab = stab + l * a.tab;
                if (lane_in) Fam::template apply<Du, false>(a.L[l], tile + a.col0[l], x, ld, ctx);
            }
            bool bad = false;
#pragma unroll
            for (int c = 0; c < N; ++c) {
                const int j = j0 + c;
                if (j >= n_dir) break;
                T gj = gld * ld.d[c];
#pragma unroll
                for (int d = 0; d < Fam::DIM; ++d) if (d < a.dim) gj += (gxo[d] - x[d].v * gblp) * x[d].d[c];
                if (!active) gj = T(0);
                bad = bad || !M<T>::finite(gj);
                if (j < a.dim) {
                    if (active) a.g_x[row * a.gxs + j] = gj;
                } else {
                    const T s = wave_sum<T>(gj);
                    if (tid == 0) accp[j - a.dim] += s;
                }
            }
            bad_any = bad_any || (active && bad);
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int c = 0; c < N; ++c) if (j0 + c >= a.dim && j0 + c < n_dir) tile[j0 + c - a.dim].d[c] = T(0);
        }
    }
    __syncthreads();
    for (int j = tid; j < a.P; j += 64) atomicAdd(a.g_params + j, accp[j]);
    status_add(a.status, JF_STATUS_NONFINITE, bad_any);
}

// N: input directions per pass (DualN<T, N>, jf_dual.h): the value part of the chain is evaluated once per pass
template <typename T, class Fam, int N>
__global__ void __launch_bounds__(64) mchain_bwd_kernel(const MBwdArgs<T, typename Fam::CLayer> a) {
    using Du = DualN<T, N>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    if constexpr (bwd_has_build<Fam>::value) {
        if (a.shared_tab) { mchain_bwd_shared_tab<T, Fam, N>(a, smem_raw); return; }      // (uniform)
    }
    Du* tile = reinterpret_cast<Du*>(smem_raw);
    const int tid = threadIdx.x;
    const int rows = a.rows;
    const int tile_rows = a.bcast ? 1 : rows;
    const bool lane_in = tid < rows;
    const int slot = lane_in ? tid : 0;
    Du* tab = tile + tile_rows * a.tile_stride + slot * a.tab;
    Du* corr = tile + tile_rows * a.tile_stride + rows * a.tab + slot * a.scratch;
    // broadcast parameters: the workgroup's sums of the parameter gradients over ALL its row tiles, added to g_params once at the end.  (Until
    // round 5 every 64-row workgroup added its sums straight to g_params: 4096 workgroups x 33 atomics on the same 33 words at 2^18 rows of
    // C4's `r` layer -- 0.86 ms, of which the chain replays are ~0.1.)  A resident set of workgroups walks the tiles instead.
    T* accp = reinterpret_cast<T*>(tile + tile_rows * a.tile_stride + rows * (a.tab + a.scratch));
    const int64_t row0_first = (int64_t)blockIdx.x * rows;

    // parameters of all layers -> dual rows (tangents 0)
    if (a.bcast) {
        for (int j = tid; j < a.P; j += 64) { tile[j] = Du(a.params[j]); accp[j] = T(0); }
    } else {
        const int64_t row0 = row0_first;
        // eight loads in flight per lane (a row at a time is one dependent load after the other: 64 x ~1 us)
        const int total = rows * a.P;
        for (int i0 = 0; i0 < total; i0 += 64 * 8) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 64 + tid;
                const int rr = i / a.P, j = i - rr * a.P;
                const int64_t gr = (row0 + rr) < a.B ? (row0 + rr) : a.B - 1;
                v[u] = i < total ? a.params[gr * a.ps + j] : T(0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 64 + tid;
                const int rr = i / a.P, j = i - rr * a.P;
                if (i < total) tile[rr * a.tile_stride + j] = Du(v[u]);
            }
        }
    }
    __syncthreads();
    Du* prow = tile + (a.bcast ? 0 : slot * a.tile_stride);
    bool bad_any = false;
    const int64_t n_tiles = (a.B + rows - 1) / rows;
    const int64_t t_end = a.bcast ? n_tiles : (int64_t)blockIdx.x + 1;      // per-sample parameters: the tile whose rows were staged above
    for (int64_t tile_i = blockIdx.x; tile_i < t_end; tile_i += gridDim.x) {
    const int64_t row0 = tile_i * rows;
    const int64_t row = row0 + tid;
    const bool active = lane_in && row < a.B;
    const int64_t rrow = active ? row : a.B - 1;

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
    for (int j0 = 0; j0 < n_dir; j0 += N) {
        // ---- seed directions j0 .. j0 + N - 1 (tangent c = direction j0 + c)
#pragma unroll
        for (int c = 0; c < N; ++c) {
            const int j = j0 + c;
            if (j >= a.dim && j < n_dir) {
                if (a.bcast) { if (tid == 0) prow[j - a.dim].d[c] = T(1); }
                else if (lane_in) prow[j - a.dim].d[c] = T(1);
            }
        }
        __syncthreads();
        Du x[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            x[d] = Du(x0[d]);
#pragma unroll
            for (int c = 0; c < N; ++c) if (d == j0 + c) x[d].d[c] = T(1);
        }
        Du ld(T(0));
        LaneCtx<Du> ctx;
        ctx.tab = tab; ctx.corr = corr; ctx.bins = nullptr; ctx.bin_i = 0;
        ctx.oob = ctx.nonconv = ctx.nonfinite = false;
        ctx.lane_valid = active;
        for (int i = 0; i < a.n_layers; ++i) {
            const int l = a.n_layers - 1 - i;
            if (lane_in) Fam::template apply<Du, false>(a.L[l], prow + a.col0[l], x, ld, ctx);
        }
#pragma unroll
        for (int c = 0; c < N; ++c) {
            const int j = j0 + c;
            if (j >= n_dir) break;
            T gj = gld * ld.d[c];
#pragma unroll
            for (int d = 0; d < Fam::DIM; ++d) if (d < a.dim) gj += (gxo[d] - x[d].v * gblp) * x[d].d[c];
            if (!active) gj = T(0);
            bad = bad || !M<T>::finite(gj);
            if (j < a.dim) {
                if (active) a.g_x[row * a.gxs + j] = gj;
            } else if (a.bcast) {
                const T s = wave_sum<T>(gj);
                if (tid == 0) accp[j - a.dim] += s;
            } else if (active) {
                a.g_params[row * a.gps + (j - a.dim)] = gj;
            }
        }
        __syncthreads();
        // ---- unseed
#pragma unroll
        for (int c = 0; c < N; ++c) {
            const int j = j0 + c;
            if (j >= a.dim && j < n_dir) {
                if (a.bcast) { if (tid == 0) prow[j - a.dim].d[c] = T(0); }
                else if (lane_in) prow[j - a.dim].d[c] = T(0);
            }
        }
    }
    bad_any = bad_any || (active && bad);
    }
    if (a.bcast) {
        __syncthreads();
        for (int j = tid; j < a.P; j += 64) atomicAdd(a.g_params + j, accp[j]);
    }
    status_add(a.status, JF_STATUS_NONFINITE, bad_any);
}

// ---------------------------------------------------------------------------------------------------------- 'v' chains, staged
// The exponential-map layer (natural_direction 0, the default: log-prob direction = one direct evaluation) factors into
//     pre  : (x, rotation parameters) -> e in R^3                       (rotation, angles -> embedding; cheap)
//     pot  : (e, potential parameters) -> u = (grad phi (3), its Jacobian (3 x 3))          (a sum over the components)
//     geo  : (e, u) -> exp_e(grad phi), 1/2 log det of the projected Jacobian               (no parameters; the expensive part)
//     post : embedding -> angles (+ chart of the first layer)                               (cheap)
// The generic kernel above replays ALL of it once per input direction: 2 + 50 passes for the default layer.  Here (round 5) the two expensive
// stages are taken in REVERSE mode, written out by hand (jf_expmap.h):
//   (1) geo: v_geo_forward keeps the intermediates, v_geo_reverse takes (d S / d y, d S / d logdet) back to G = d S / d (e, u); `post` gives
//       d S / d y from three dual-number directions;
//   (2) pot: v_component_adjoint, one evaluation per component -> its five parameters and the potential's share of d S / d e.  Spline
//       potentials (31 parameters per component behind a bin search) replay the component on dual numbers instead, one parameter at a time,
//       contracted with G;
//   (3) pre on dual numbers for the layer's input and its rotation parameters, contracted with d S / d e.
// Layers of a chain are walked in reverse with the (2 + 1)-component upstream gradient, after a forward sweep that records every layer's
// input (the last layer applied is evaluated by its own backward step only).  C5's block, 2^17 rows: 0.495 ms (round 4: 15 geometry
// directions in five 3-tangent passes + one dual replay of a component per parameter) -> 0.348 (2) -> 0.195 ms (1).
// JF_V_BWD_DUAL=1 selects the dual-number replay of BOTH stages through v_exp_geometry / v_component themselves: the check of the hand-written
// adjoints (scripts/probe/v_adjoint_check.py, tests/test_gpu_grad.py), not a product path.
//
// LV lanes per row: the components of (2), the directions of (3) -- and, in the check build, the 15 geometry directions -- are independent, so
// LV lanes of a row can take them side by side (k = lane, lane + LV, ...) and exchange sums by lane shuffles: LV times the waves for a kernel
// whose 2^17 rows are two waves per SIMD at one wave's worth of registers each (vector unit 38 % busy).  Measured with LV = 4 / 8 and the
// register caps that let the extra waves be resident (256 / 168 / 128 registers): dual-number geometry 0.53 / 0.9 / 1.1 ms against 0.35 for
// one lane per row, reverse-mode geometry 1.07 (LV 4, 256 registers) against 0.195 -- the geometry's intermediates (~100 doubles) spill to
// scratch under every cap.  One lane per row is what is instantiated; the code is kept LV-generic (checked at 4 and 8).
// Parameter rows are kept as PLAIN values in LDS; the potential's functions read them through SeededVals (value + a unit tangent at one index),
// the <= 12 rotation parameters of the layer at hand are copied into a lane-private dual row.
template <typename T> struct SeededVals {
    const T* p; int seed;
    __device__ __forceinline__ Dual<T> operator[](int i) const { return Dual<T>(p[i], i == seed ? T(1) : T(0)); }
};
constexpr int JF_V_ROT_MAX = 12;
constexpr int JF_V_G = 16;                                           // G's slot per row (15 used)

template <typename T, int LV> __device__ __forceinline__ T group_sum(T v) {         // over the LV lanes of a row
#pragma unroll
    for (int m = 1; m < LV; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}
template <typename T, int LV> __device__ __forceinline__ T group_max(T v) {
#pragma unroll
    for (int m = 1; m < LV; m <<= 1) v = M<T>::max(v, __shfl_xor(v, m, 64));
    return v;
}
template <typename T, int LV> __device__ __forceinline__ T rows_sum(T v) {          // over the rows of the wave, for each lane-of-row
#pragma unroll
    for (int m = LV; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// log-sum-exp of the components' log-weights, the components dealt to the LV lanes of the row
template <typename T, int LV> __device__ inline T v_lse_lanes(const T* __restrict__ pp, int nc, int g) {
    T lmax = T(-INFINITY);
    for (int k = g; k < nc; k += LV) lmax = M<T>::max(lmax, pp[3 * nc + k]);
    lmax = group_max<T, LV>(lmax);
    T se = T(0);
    for (int k = g; k < nc; k += LV) se += M<T>::exp(pp[3 * nc + k] - lmax);
    return lmax + M<T>::log(group_sum<T, LV>(se));
}

// v_potential with the components dealt to the LV lanes of the row (g: this lane's index in its row); every lane returns the full sums
template <typename T, int LV> __device__ inline void v_potential_lanes(const T* __restrict__ pp, int nc, int kind, const T (&x)[3], VPotential<T>& P, T& lse,
                                                                       T* __restrict__ tab, bool& oob, int g) {
    lse = v_lse_lanes<T, LV>(pp, nc, g);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        P.g[i] = T(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) P.gj[i][j] = T(0);
    }
    for (int k = g; k < nc; k += LV) v_component<T>(pp, nc, k, kind, lse, x, P, tab, oob);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        P.g[i] = group_sum<T, LV>(P.g[i]);
#pragma unroll
        for (int j = 0; j < 3; ++j) P.gj[i][j] = group_sum<T, LV>(P.gj[i][j]);
    }
}

// closed-form potentials: the sums over this lane's components, TWO components per trip -- their chains of transcendental functions are
// independent, and a kernel that runs one wave per SIMD has nothing else to cover their latency with (KIND as a template parameter keeps the
// pair in one basic block for the scheduler)
template <typename T, int LV, int KIND>
__device__ inline void v_closed_potential(const T* __restrict__ ppv, int nc, T lse, const T (&e)[3], VPotential<T>& P, int g) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        P.g[i] = T(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) P.gj[i][j] = T(0);
    }
    int k = g;
#pragma unroll 1
    for (; k + LV < nc; k += 2 * LV) {
        const VCompVals<T> v0 = v_component_vals<T, KIND>(ppv, nc, k, lse, e), v1 = v_component_vals<T, KIND>(ppv, nc, k + LV, lse, e);
        v_component_add<T, KIND>(ppv, nc, k, v0, P);
        v_component_add<T, KIND>(ppv, nc, k + LV, v1, P);
    }
    if (k < nc) v_component_add<T, KIND>(ppv, nc, k, v_component_vals<T, KIND>(ppv, nc, k, lse, e), P);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        P.g[i] = group_sum<T, LV>(P.g[i]);
#pragma unroll
        for (int j = 0; j < 3; ++j) P.gj[i][j] = group_sum<T, LV>(P.gj[i][j]);
    }
}

// ... and their reverse pass: the gradients of every component's parameters go to g_params (per-sample rows: stored; permanent parameters:
// summed over the wave's rows, one atomic add per parameter), the potential's share of d S / d e is added to Ge (this lane's components only)
template <typename T, int LV, int KIND>
__device__ inline void v_closed_adjoint(const MBwdArgs<T, jf_v_layer>& a, const T* __restrict__ ppv, int nc, T lse, const T (&e)[3], const T (&Gg)[3],
                                        const T (&Gj)[3][3], T GP0, T (&Ge)[3], int g, int tid, bool active, int64_t row, int col0, bool& bad,
                                        T* __restrict__ accp) {
    constexpr int n_prow = KIND == JF_V_EXPONENTIAL ? 5 : 4;
#pragma unroll 1
    for (int k0 = 0; k0 < nc; k0 += 2 * LV) {
        const int kk[2] = {k0 + g, k0 + LV + g};
        T gp[2][5];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = kk[u] < nc ? kk[u] : nc - 1;           // (a lane without a component in this trip repeats the last one; its results are dropped)
            const VCompVals<T> v = v_component_vals<T, KIND>(ppv, nc, k, lse, e);
            T gm[3], gx[3] = {T(0), T(0), T(0)};
            v_component_adjoint<T, KIND>(ppv, nc, k, v, e, Gg, Gj, gx, gm, gp[u][3], gp[u][4]);
            gp[u][0] = gm[0]; gp[u][1] = gm[1]; gp[u][2] = gm[2];
            gp[u][3] -= M<T>::exp(ppv[3 * nc + k] - lse) * GP0;                                 // softmax coupling of the log-weights
            if (kk[u] < nc) { Ge[0] += gx[0]; Ge[1] += gx[1]; Ge[2] += gx[2]; }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = kk[u];
            if (k0 + u * LV >= nc) break;                          // (uniform)
#pragma unroll
            for (int q = 0; q < n_prow; ++q) {
                const T v = (active && k < nc) ? gp[u][q] : T(0);
                bad = bad || !M<T>::finite(v);
                const int col = col0 + q * nc + k;
                if (a.bcast) {
                    const T s = rows_sum<T, LV>(v);
                    if (tid < LV && k < nc) accp[col] += s;          // the workgroup's sums (one wave: no atomic), added to g_params at the end
                } else if (active && k < nc) {
                    a.g_params[row * a.gps + col] = v;
                }
            }
        }
    }
}

// ... and the spline potentials (round 6): the component's f / f' are a monotone rational-quadratic spline of mu . x and its derivative
// (v_component, jf_expmap.h), so S_k = w (f A + f' Q) reaches the component's 31 spline parameters through the six knot values of one bin:
// the seven-tangent bin evaluation and the hand-written reverse of the knot table of jf_spline_adj.h (until round 5: one dual-number replay of
// the component per parameter -- 31 per component, each with its own table build: v_s2_splines_cond 37 ms per 2^16 rows against 0.9 forward).
// scr: the lane's scratch, >= 33 + 31 + 31 words (knot table, the component's raw spline parameters made contiguous, their gradients)
template <typename T, int LV>
__device__ inline void v_spline_adjoint(const MBwdArgs<T, jf_v_layer>& a, const T* __restrict__ ppv, int nc, T lse, const T (&e)[3], const T (&Gg)[3],
                                        const T (&Gj)[3][3], T GP0, T (&Ge)[3], int g, int tid, bool active, int64_t row, int col0, bool& bad,
                                        T* __restrict__ accp, T* __restrict__ scr) {
    constexpr int NB = JF_V_SPLINE_BINS, NSP = 3 * NB + 1, n_prow = 4 + NSP;
    using D7 = DualN<T, 7>;
    SplineDev<T> o;
    o.nb = NB; o.smooth = 0; o.fix_first = 0; o.fix_second = 0; o.independent = 0; o.fix_bd = 0; o.n_w = NB; o.n_h = NB; o.n_d = NB + 1;
    o.fix_bd_value = T(0); o.min_w = T(1e-3); o.min_h = T(1e-3); o.min_d = T(1e-3); o.ratio = T(-1);
    T* tab = scr;
    T* ploc = scr + 3 * (NB + 1);
    T* gloc = ploc + NSP;
#pragma unroll 1
    for (int k0 = 0; k0 < nc; k0 += LV) {
        const int kk = k0 + g;
        const int k = kk < nc ? kk : nc - 1;                       // (a lane without a component in this trip repeats the last one; its results are dropped)
        const VCompVals<T> v = v_component_vals<T, JF_V_LINEAR>(ppv, nc, k, lse, e);      // weight, d w / d |m|, 1 / |m|  (f of the linear kind: unused)
        const T mu[3] = {ppv[k] * v.inv_nrm, ppv[nc + k] * v.inv_nrm, ppv[2 * nc + k] * v.inv_nrm};
        const T xmu = e[0] * mu[0] + e[1] * mu[1] + e[2] * mu[2];
        for (int j = 0; j < NSP; ++j) { ploc[j] = ppv[(4 + j) * nc + k]; gloc[j] = T(0); }
        spline_interval_build<T>(ploc, o, tab, T(-1), T(1));
        const KnotTab<T> t(tab, NB);
        const int b = spline_adj_bin<T>(t.cw, t.ch, NB, xmu, false);
        D7 in[7] = {D7(xmu), D7(t.cw[b]), D7(t.cw[b + 1]), D7(t.ch[b]), D7(t.ch[b + 1]), D7(t.d[b]), D7(t.d[b + 1])};
#pragma unroll
        for (int c = 0; c < 7; ++c) in[c].d[c] = T(1);
        const SplineOut<D7> r = spline_core_vals<D7>(in[1], in[2], in[3], in[4], in[5], in[6], b, in[0], false);
        const T f = r.y.v, fp = M<T>::exp(r.lad.v);
        // S_k = w (f A + fp Q),  A = Gg . mu,  Q = mu^T Gj mu;  h = (Gj + Gj^T) mu = dQ / d mu
        T h[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) h[i] = (Gj[i][0] + Gj[0][i]) * mu[0] + (Gj[i][1] + Gj[1][i]) * mu[1] + (Gj[i][2] + Gj[2][i]) * mu[2];
        const T A = Gg[0] * mu[0] + Gg[1] * mu[1] + Gg[2] * mu[2];
        const T Q = T(0.5) * (h[0] * mu[0] + h[1] * mu[1] + h[2] * mu[2]);
        const T w = v.w;
        const T dS_dw = f * A + fp * Q;
        const T gy = w * A, glad = w * Q * fp;                     // d fp = fp d lad
        const T gxmu = gy * r.y.d[0] + glad * r.lad.d[0];
        T gk[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) gk[c] = gy * r.y.d[c + 1] + glad * r.lad.d[c + 1];
        (void)spline_adj_table_reverse<T>(ploc, gloc, o, tab, b, gk, T(-1), T(1), false, T(1));
        T dmu[3], gm[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) dmu[i] = w * (f * Gg[i] + fp * h[i]) + gxmu * e[i];
        if (kk < nc) { Ge[0] += gxmu * mu[0]; Ge[1] += gxmu * mu[1]; Ge[2] += gxmu * mu[2]; }
        const T radial = dmu[0] * mu[0] + dmu[1] * mu[1] + dmu[2] * mu[2];
        const T dS_dnrm = dS_dw * v.dwdn;
#pragma unroll
        for (int i = 0; i < 3; ++i) gm[i] = (dmu[i] - mu[i] * radial) * v.inv_nrm + dS_dnrm * mu[i];
        const T glw = dS_dw * w - M<T>::exp(ppv[3 * nc + k] - lse) * GP0;      // softmax coupling of the log-weights
        if (k0 >= nc) break;                                       // (uniform)
#pragma unroll 1
        for (int q = 0; q < n_prow; ++q) {
            T val = q < 3 ? gm[q] : (q == 3 ? glw : gloc[q - 4]);
            if (!(active && kk < nc)) val = T(0);
            bad = bad || !M<T>::finite(val);
            const int col = col0 + q * nc + k;
            if (a.bcast) {
                const T sum = rows_sum<T, LV>(val);
                if (tid < LV && kk < nc) accp[col] += sum;
            } else if (active && kk < nc) {
                a.g_params[row * a.gps + col] = val;
            }
        }
    }
}

// the potential's share of d S / d e for given (Gg, Gj), no parameter gradients: what the implicit-function adjoint of natural_direction = 1
// (below) applies three times per row before it knows the multiplier of the solve.  Every potential kind; scr as in v_spline_adjoint.
template <typename T>
__device__ inline void v_pot_x_adjoint(const T* __restrict__ ppv, int nc, int kind, T lse, const T (&e)[3], const T (&Gg)[3], const T (&Gj)[3][3], T (&Ge)[3],
                                       T* __restrict__ scr) {
    for (int k = 0; k < nc; ++k) {
        T gx[3] = {T(0), T(0), T(0)}, gm[3], glw, glb;
        if (kind == JF_V_EXPONENTIAL) {
            v_component_adjoint<T, JF_V_EXPONENTIAL>(ppv, nc, k, v_component_vals<T, JF_V_EXPONENTIAL>(ppv, nc, k, lse, e), e, Gg, Gj, gx, gm, glw, glb);
        } else if (kind == JF_V_LINEAR) {
            v_component_adjoint<T, JF_V_LINEAR>(ppv, nc, k, v_component_vals<T, JF_V_LINEAR>(ppv, nc, k, lse, e), e, Gg, Gj, gx, gm, glw, glb);
        } else if (kind == JF_V_QUADRATIC) {
            v_component_adjoint<T, JF_V_QUADRATIC>(ppv, nc, k, v_component_vals<T, JF_V_QUADRATIC>(ppv, nc, k, lse, e), e, Gg, Gj, gx, gm, glw, glb);
        } else {                                                   // splines: S_k = w (f A + f' Q) through the spline's input mu . e only
            constexpr int NB = JF_V_SPLINE_BINS, NSP = 3 * NB + 1;
            using D1 = Dual<T>;
            SplineDev<T> o;
            o.nb = NB; o.smooth = 0; o.fix_first = 0; o.fix_second = 0; o.independent = 0; o.fix_bd = 0; o.n_w = NB; o.n_h = NB; o.n_d = NB + 1;
            o.fix_bd_value = T(0); o.min_w = T(1e-3); o.min_h = T(1e-3); o.min_d = T(1e-3); o.ratio = T(-1);
            T* tab = scr;
            T* ploc = scr + 3 * (NB + 1);
            const VCompVals<T> v = v_component_vals<T, JF_V_LINEAR>(ppv, nc, k, lse, e);
            const T mu[3] = {ppv[k] * v.inv_nrm, ppv[nc + k] * v.inv_nrm, ppv[2 * nc + k] * v.inv_nrm};
            const T xmu = e[0] * mu[0] + e[1] * mu[1] + e[2] * mu[2];
            for (int j = 0; j < NSP; ++j) ploc[j] = ppv[(4 + j) * nc + k];
            spline_interval_build<T>(ploc, o, tab, T(-1), T(1));
            const KnotTab<T> t(tab, NB);
            const int b = spline_adj_bin<T>(t.cw, t.ch, NB, xmu, false);
            const SplineOut<D1> r = spline_core_vals<D1>(D1(t.cw[b]), D1(t.cw[b + 1]), D1(t.ch[b]), D1(t.ch[b + 1]), D1(t.d[b]), D1(t.d[b + 1]), b, D1(xmu, T(1)), false);
            const T fp = M<T>::exp(r.lad.v);
            T h[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) h[i] = (Gj[i][0] + Gj[0][i]) * mu[0] + (Gj[i][1] + Gj[1][i]) * mu[1] + (Gj[i][2] + Gj[2][i]) * mu[2];
            const T A = Gg[0] * mu[0] + Gg[1] * mu[1] + Gg[2] * mu[2];
            const T Q = T(0.5) * (h[0] * mu[0] + h[1] * mu[1] + h[2] * mu[2]);
            const T gxmu = v.w * (A * r.y.d + Q * fp * r.lad.d);
#pragma unroll
            for (int i = 0; i < 3; ++i) gx[i] = gxmu * mu[i];
        }
        Ge[0] += gx[0]; Ge[1] += gx[1]; Ge[2] += gx[2];
    }
}

template <typename T, int LV, int WPE, bool DUALGEO>
__global__ void __launch_bounds__(64, WPE) vchain_bwd_kernel(const MBwdArgs<T, jf_v_layer> a) {
    using Du = Dual<T>;
    constexpr int NG = (15 + LV - 1) / LV;                         // geometry directions per lane
    using DG = DualN<T, NG>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);                      // [tile_rows][tile_stride] plain parameter values
    const int tid = threadIdx.x, r = tid / LV, g = tid % LV;
    const int rows = a.rows;                                       // rows of this workgroup (<= 64 / LV)
    const int tile_rows = a.bcast ? 1 : rows;
    const bool lane_in = r < rows;
    const int tile_elems = (tile_rows * a.tile_stride + 1) & ~1;   // the dual regions stay 16-byte aligned
    Du* dual0 = reinterpret_cast<Du*>(tile + tile_elems);
    Du* tab = dual0 + tid * (a.tab + a.rot_max);                   // lane-private: knot table (spline potentials only), then the rotation row
    Du* rot = tab + a.tab;
    // the row's scratch in LDS: every layer's input (2 per layer) and G (15)
    T* row_mem = reinterpret_cast<T*>(dual0 + 64 * (a.tab + a.rot_max)) + (lane_in ? r : 0) * a.scratch;
    T* xin = row_mem;                                              // [layer][2]
    T* Gm = row_mem + 2 * a.n_layers;                              // [JF_V_G]
    // broadcast parameters: a resident set of workgroups walks the row tiles and adds its gradient sums to g_params once (mchain_bwd_kernel above)
    T* accp = row_mem - (lane_in ? r : 0) * a.scratch + (size_t)rows * a.scratch;      // behind the rows' scratch
    const int64_t row0_first = (int64_t)blockIdx.x * rows;
    if (a.bcast) {
        for (int j = tid; j < a.P; j += 64) { tile[j] = a.params[j]; accp[j] = T(0); }
    } else {
        const int64_t row0 = row0_first;
        // the rows' parameters, eight loads in flight per lane (a row at a time is one dependent load after the other: 64 x ~1 us)
        const int total = rows * a.P;
        for (int i0 = 0; i0 < total; i0 += 64 * 8) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 64 + tid;
                const int rr = i / a.P, j = i - rr * a.P;
                const int64_t gr = (row0 + rr) < a.B ? (row0 + rr) : a.B - 1;
                v[u] = i < total ? a.params[gr * a.ps + j] : T(0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 64 + tid;
                const int rr = i / a.P, j = i - rr * a.P;
                if (i < total) tile[rr * a.tile_stride + j] = v[u];
            }
        }
    }
    __syncthreads();
    const T* prow = tile + (a.bcast ? 0 : (lane_in ? r : 0) * a.tile_stride);
    bool bad_any = false;
    const int64_t n_tiles = (a.B + rows - 1) / rows;
    const int64_t t_end = a.bcast ? n_tiles : (int64_t)blockIdx.x + 1;
    for (int64_t tile_i = blockIdx.x; tile_i < t_end; tile_i += gridDim.x) {
    const int64_t row0 = tile_i * rows;
    const int64_t row = row0 + r;
    const bool active = lane_in && row < a.B;
    const int64_t rrow = active ? row : a.B - 1;
    const T gld = (a.g_ld && active) ? a.g_ld[rrow] : T(0);
    const T gblp = (a.g_blp && active) ? a.g_blp[rrow] : T(0);
    const T gxo[2] = {(a.g_xout && active) ? a.g_xout[rrow * a.gxos + 0] : T(0), (a.g_xout && active) ? a.g_xout[rrow * a.gxos + 1] : T(0)};
    bool oob = false;

    // ---- forward sweep (plain values): the input of every layer (the last one applied, layer 0, is evaluated by its own backward step)
    {
        T x[3] = {a.x[rrow * a.xs + 0], a.x[rrow * a.xs + 1], T(0)};
        T ld = T(0);
#pragma unroll 1
        for (int l = a.n_layers - 1; l >= 0; --l) {
            if (lane_in && g == 0) { xin[2 * l] = x[0]; xin[2 * l + 1] = x[1]; }
            if (l == 0) break;
            const jf_v_layer L = a.L[l];
            const T* pv = prow + a.col0[l];
            T e[3];
            VPotential<T> P;
            ExpMapOut<T> o;
            T lse;
            VFam::template inv_pre<T>(L, pv, x, ld, e);
            if (L.natural_direction) {                             // the layer is the INVERSE of the exponential map: its output solves exp map(r) = e
                T rr[3];
                v_newton<T>(pv + rot_len(L.hh_iter, 3), L.num_components, L.exp_map_type, e, L.max_newton_iter, false, active, rr, reinterpret_cast<T*>(tab), oob);
                VFam::template inv_post<T>(L, rr, x, ld);
            } else {
                v_potential_lanes<T, LV>(pv + rot_len(L.hh_iter, 3), L.num_components, L.exp_map_type, e, P, lse, reinterpret_cast<T*>(tab), oob, g);
                v_exp_geometry<T>(L.exp_map_type, e, P, o);
                VFam::template inv_post<T>(L, o.y, x, ld);
            }
        }
    }
    __syncthreads();

    T up[2] = {T(0), T(0)};                                        // d S / d (x after the layer being differentiated)
    bool bad = false;
#pragma unroll 1
    for (int l = 0; l < a.n_layers; ++l) {                        // reverse of the order of application (layer n-1 is applied first)
        const jf_v_layer L = a.L[l];
        const T* pv = prow + a.col0[l];                            // this layer's plain row: rotation parameters, then the potential's
        const int n_rot = rot_len(L.hh_iter, 3);
        const T* ppv = pv + n_rot;
        const int nc = L.num_components, kind = L.exp_map_type;
        if (lane_in) for (int i = 0; i < n_rot; ++i) rot[i] = Du(pv[i]);
        const Du* p = rot;
        const T xl[2] = {xin[2 * l], xin[2 * l + 1]};
        // values of the intermediates at this layer's input
        T e0[3] = {T(0), T(0), T(0)};
        VPotential<T> P0;
        T lse0;
        {
            T x[3] = {xl[0], xl[1], T(0)};
            T ld = T(0);
            VFam::template inv_pre<T>(L, pv, x, ld, e0);
        }
        // natural_direction = 1 (the log-prob direction SOLVES exp map(r; p) = e, exponential_map_s2.py:459-489): the solution's dependence on e and
        // on the parameters by the implicit-function theorem, from the same hand-written adjoints evaluated AT the solution r.  With
        // A(yb, gl) = the pull-back of (d S / d y, d S / d logdet_half) through geometry + potential to (r, parameters):
        //     the layer's output is r and its log-det term is -logdet_half(r, p), so with rb = d S / d r from the steps behind the layer,
        //     w = rb - gld grad_r logdet_half,   J^T lambda = w  (J = d exp map / d r between the tangent planes at r and at e),
        //     d S / d e = lambda,   d S / d p = A(-lambda, -gld).p
        // J^T is assembled from two pull-backs of tangent vectors at e, grad_r logdet_half is a third; a 2 x 2 solve; the fourth pull-back
        // carries the parameters.  (Until round 6 these chains went to the generic kernel, which replays the Newton iteration on dual numbers
        // once per direction: v_s2_nat1_rot 11.6 ms per 2^16 rows against 0.17 forward.)
        if (L.natural_direction) {
            if constexpr (!DUALGEO) {
            T* scrT = reinterpret_cast<T*>(tab);
            T r[3];
            v_newton<T>(ppv, nc, kind, e0, L.max_newton_iter, false, active, r, scrT, oob);
            VPotential<T> Pr;
            T lser;
            v_potential_lanes<T, LV>(ppv, nc, kind, r, Pr, lser, scrT, oob, g);
            VGeoTape<T> tape;
            T y[3], ldh;
            v_geo_forward<T>(kind, r, Pr, tape, y, ldh);
            // d S / d r from what follows the layer (post: embedding -> angles, chart)
            using D3 = DualN<T, 3>;
            D3 rd[3] = {D3(r[0]), D3(r[1]), D3(r[2])}, xd[3], ldd(T(0));
            rd[0].d[0] = T(1); rd[1].d[1] = T(1); rd[2].d[2] = T(1);
            VFam::template inv_post<D3>(L, rd, xd, ldd);
            if (l == 0) { up[0] = gxo[0] - xd[0].v * gblp; up[1] = gxo[1] - xd[1].v * gblp; }
            T rb[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) rb[c] = up[0] * xd[0].d[c] + up[1] * xd[1].d[c] + gld * ldd.d[c];
            auto pull = [&](const T (&yb)[3], T gl, T (&out)[3], T (&Gg)[3], T (&Gj)[3][3]) {       // A(yb, gl).r; (Gg, Gj) for the parameter part
                T Ge0[3];
                v_geo_reverse<T>(kind, r, Pr, tape, yb, gl, Ge0, Gg, Gj);
                out[0] = Ge0[0]; out[1] = Ge0[1]; out[2] = Ge0[2];
                v_pot_x_adjoint<T>(ppv, nc, kind, lser, r, Gg, Gj, out, scrT);
            };
            // orthonormal tangent bases: (s1, s2) at r, (t1, t2) at the image y = e
            auto basis = [](const T (&n)[3], T (&b1)[3], T (&b2)[3]) {
                const int m = (M<T>::abs(n[0]) <= M<T>::abs(n[1]) && M<T>::abs(n[0]) <= M<T>::abs(n[2])) ? 0 : (M<T>::abs(n[1]) <= M<T>::abs(n[2]) ? 1 : 2);
                T a[3] = {T(0), T(0), T(0)};
                a[m] = T(1);
                const T dot = n[m];
                T nn = T(0);
#pragma unroll
                for (int i = 0; i < 3; ++i) { b1[i] = a[i] - dot * n[i]; nn += b1[i] * b1[i]; }
                const T inv = T(1) / M<T>::sqrt(nn);
#pragma unroll
                for (int i = 0; i < 3; ++i) b1[i] *= inv;
                b2[0] = n[1] * b1[2] - n[2] * b1[1]; b2[1] = n[2] * b1[0] - n[0] * b1[2]; b2[2] = n[0] * b1[1] - n[1] * b1[0];
            };
            T s1[3], s2[3], t1[3], t2[3];
            basis(r, s1, s2);
            basis(y, t1, t2);
            T Gg[3], Gj[3][3], c1[3], c2[3], a0[3];
            const T zero3[3] = {T(0), T(0), T(0)};
            pull(zero3, T(1), a0, Gg, Gj);
            pull(t1, T(0), c1, Gg, Gj);
            pull(t2, T(0), c2, Gg, Gj);
            auto dot3 = [](const T (&u)[3], const T (&v)[3]) { return u[0] * v[0] + u[1] * v[1] + u[2] * v[2]; };
            T w[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) w[c] = rb[c] - gld * a0[c];
            const T m11 = dot3(s1, c1), m12 = dot3(s1, c2), m21 = dot3(s2, c1), m22 = dot3(s2, c2);
            const T w1 = dot3(s1, w), w2 = dot3(s2, w);
            const T det = m11 * m22 - m12 * m21;
            const T l1 = (w1 * m22 - m12 * w2) / det, l2 = (m11 * w2 - m21 * w1) / det;
            T lam[3], mlam[3], dummy[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { lam[c] = l1 * t1[c] + l2 * t2[c]; mlam[c] = -lam[c]; }
            // the parameters: A(-lambda, -gld)
            pull(mlam, -gld, dummy, Gg, Gj);
            T GP0 = T(0);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                GP0 += Gg[c] * Pr.g[c];
#pragma unroll
                for (int d = 0; d < 3; ++d) GP0 += Gj[c][d] * Pr.gj[c][d];
            }
            T Gex[3] = {T(0), T(0), T(0)};
            const int colp = a.col0[l] + n_rot;
            if (kind == JF_V_SPLINES) v_spline_adjoint<T, LV>(a, ppv, nc, lser, r, Gg, Gj, GP0, Gex, g, tid, active, row, colp, bad, accp, scrT);
            else if (kind == JF_V_EXPONENTIAL) v_closed_adjoint<T, LV, JF_V_EXPONENTIAL>(a, ppv, nc, lser, r, Gg, Gj, GP0, Gex, g, tid, active, row, colp, bad, accp);
            else if (kind == JF_V_LINEAR) v_closed_adjoint<T, LV, JF_V_LINEAR>(a, ppv, nc, lser, r, Gg, Gj, GP0, Gex, g, tid, active, row, colp, bad, accp);
            else v_closed_adjoint<T, LV, JF_V_QUADRATIC>(a, ppv, nc, lser, r, Gg, Gj, GP0, Gex, g, tid, active, row, colp, bad, accp);
            // the layer's input and rotation parameters: pre on dual numbers, contracted with d S / d e = lambda
            T nup[2] = {T(0), T(0)};
            const int n_dir = 2 + n_rot;
#pragma unroll 1
            for (int j = 0; j < n_dir; ++j) {
                if (j >= 2 && lane_in) rot[j - 2].d = T(1);
                Du x[3] = {Du(xl[0], j == 0 ? T(1) : T(0)), Du(xl[1], j == 1 ? T(1) : T(0)), Du(T(0))};
                Du ld(T(0)), e[3];
                VFam::template inv_pre<Du>(L, p, x, ld, e);
                T gj_ = gld * ld.d + lam[0] * e[0].d + lam[1] * e[1].d + lam[2] * e[2].d;
                if (j >= 2 && lane_in) rot[j - 2].d = T(0);
                if (!active) gj_ = T(0);
                bad = bad || !M<T>::finite(gj_);
                if (j < 2) nup[j] = gj_;
                else if (a.bcast) {
                    const T sum = rows_sum<T, LV>(gj_);
                    if (tid < LV) accp[a.col0[l] + (j - 2)] += sum;
                } else if (active) {
                    a.g_params[row * a.gps + a.col0[l] + (j - 2)] = gj_;
                }
            }
            up[0] = nup[0]; up[1] = nup[1];
            }
            continue;
        }
        const bool spl = kind == JF_V_SPLINES && !a.v_dual;             // spline potentials: v_spline_adjoint (seven-tangent bin evaluation + table reverse)
        const bool closed = (kind != JF_V_SPLINES && !a.v_dual) || spl;
        if (closed && !spl) {
            lse0 = v_lse_lanes<T, LV>(ppv, nc, g);
            if (kind == JF_V_EXPONENTIAL) v_closed_potential<T, LV, JF_V_EXPONENTIAL>(ppv, nc, lse0, e0, P0, g);
            else if (kind == JF_V_LINEAR) v_closed_potential<T, LV, JF_V_LINEAR>(ppv, nc, lse0, e0, P0, g);
            else v_closed_potential<T, LV, JF_V_QUADRATIC>(ppv, nc, lse0, e0, P0, g);
        } else {
            v_potential_lanes<T, LV>(ppv, nc, kind, e0, P0, lse0, reinterpret_cast<T*>(tab), oob, g);
        }
        // (1) d S / d (e, g, gj) through geo + post
        T Gg[3], Gj[3][3], Ge0[3];
        if constexpr (!DUALGEO) {
            // reverse mode (v_geo_forward / v_geo_reverse, jf_expmap.h); the three directions of `post` on dual numbers.  Every lane of the row
            // evaluates it (nothing to deal out)
            VGeoTape<T> tape;
            T y[3], ldh;
            v_geo_forward<T>(kind, e0, P0, tape, y, ldh);
            using D3 = DualN<T, 3>;
            D3 yd[3] = {D3(y[0]), D3(y[1]), D3(y[2])}, xd[3], ldd(T(0));
            yd[0].d[0] = T(1); yd[1].d[1] = T(1); yd[2].d[2] = T(1);
            VFam::template inv_post<D3>(L, yd, xd, ldd);
            if (l == 0) { up[0] = gxo[0] - xd[0].v * gblp; up[1] = gxo[1] - xd[1].v * gblp; }    // the chain's output: base log-prob term
            T yb[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) yb[c] = up[0] * xd[0].d[c] + up[1] * xd[1].d[c] + gld * ldd.d[c];
            v_geo_reverse<T>(kind, e0, P0, tape, yb, gld, Ge0, Gg, Gj);
        } else {
            // the check of the above (JF_V_BWD_DUAL): this lane's NG of the 15 directions of (e, g, gj) on dual numbers through v_exp_geometry
            // itself, exchanged through the row's slot in LDS
            DG e[3];
            VPotential<DG> P;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                e[c] = DG(e0[c]);
                P.g[c] = DG(P0.g[c]);
#pragma unroll
                for (int d = 0; d < 3; ++d) P.gj[c][d] = DG(P0.gj[c][d]);
            }
#pragma unroll
            for (int t = 0; t < NG; ++t) {
                const int i = g * NG + t;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (i == c) e[c].d[t] = T(1);
                    if (i == 3 + c) P.g[c].d[t] = T(1);
#pragma unroll
                    for (int d = 0; d < 3; ++d) if (i == 6 + 3 * c + d) P.gj[c][d].d[t] = T(1);
                }
            }
            ExpMapOut<DG> o;
            DG x[3], ld(T(0));
            v_exp_geometry<DG>(kind, e, P, o);
            ld = ld + o.logdet_half;
            VFam::template inv_post<DG>(L, o.y, x, ld);
            if (l == 0) { up[0] = gxo[0] - x[0].v * gblp; up[1] = gxo[1] - x[1].v * gblp; }
#pragma unroll
            for (int t = 0; t < NG; ++t)
                if (lane_in && g * NG + t < 15) Gm[g * NG + t] = up[0] * x[0].d[t] + up[1] * x[1].d[t] + gld * ld.d[t];
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Ge0[c] = Gm[c];
                Gg[c] = Gm[3 + c];
#pragma unroll
                for (int d = 0; d < 3; ++d) Gj[c][d] = Gm[6 + 3 * c + d];
            }
            __syncthreads();                                       // (Gm is rewritten by the next layer)
        }
        T Ge[3] = {T(0), T(0), T(0)};
        T GP0 = T(0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            GP0 += Gg[c] * P0.g[c];
#pragma unroll
            for (int d = 0; d < 3; ++d) GP0 += Gj[c][d] * P0.gj[c][d];
        }
        const int n_row = VFam_row_len_dev(L);
        // (2) the potential's parameters
        if (spl) {
            v_spline_adjoint<T, LV>(a, ppv, nc, lse0, e0, Gg, Gj, GP0, Ge, g, tid, active, row, a.col0[l] + n_rot, bad, accp, reinterpret_cast<T*>(tab));
#pragma unroll
            for (int c = 0; c < 3; ++c) Ge[c] = Ge0[c] + group_sum<T, LV>(Ge[c]);
        } else if (closed) {
            const int col = a.col0[l] + n_rot;
            if (kind == JF_V_EXPONENTIAL) v_closed_adjoint<T, LV, JF_V_EXPONENTIAL>(a, ppv, nc, lse0, e0, Gg, Gj, GP0, Ge, g, tid, active, row, col, bad, accp);
            else if (kind == JF_V_LINEAR) v_closed_adjoint<T, LV, JF_V_LINEAR>(a, ppv, nc, lse0, e0, Gg, Gj, GP0, Ge, g, tid, active, row, col, bad, accp);
            else v_closed_adjoint<T, LV, JF_V_QUADRATIC>(a, ppv, nc, lse0, e0, Gg, Gj, GP0, Ge, g, tid, active, row, col, bad, accp);
#pragma unroll
            for (int c = 0; c < 3; ++c) Ge[c] = Ge0[c] + group_sum<T, LV>(Ge[c]);
        }
        // (3) the layer's input and rotation parameters (closed-form potentials); every direction on dual numbers through pre + pot otherwise.
        //     A potential parameter belongs to ONE component, whose term alone carries a tangent -- except the log-weights, whose softmax
        //     normaliser couples all components: d w_m / d lw_k = w_m (delta_mk - s_k), i.e. the single-component tangent (normaliser held
        //     fixed) minus s_k times the totals.
        T nup[2] = {T(0), T(0)};
        const int n_dir = closed ? 2 + n_rot : 2 + n_row;
#pragma unroll 1
        for (int j0 = 0; j0 < n_dir; j0 += LV) {
            const int j = j0 + g;
            const bool jin = j < n_dir;
            T gj_ = T(0);
            if (jin && j < 2 + n_rot) {
                if (j >= 2 && lane_in) rot[j - 2].d = T(1);        // lane-private row: no barrier
                Du x[3] = {Du(xl[0], j == 0 ? T(1) : T(0)), Du(xl[1], j == 1 ? T(1) : T(0)), Du(T(0))};
                Du ld(T(0)), e[3];
                VFam::template inv_pre<Du>(L, p, x, ld, e);
                gj_ = gld * ld.d;
                if (closed) {
                    gj_ += Ge[0] * e[0].d + Ge[1] * e[1].d + Ge[2] * e[2].d;
                } else {
                    VPotential<Du> P;
                    v_potential<Du>(SeededVals<T>{ppv, -1}, nc, kind, e, P, tab, oob);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        gj_ += Ge0[c] * e[c].d + Gg[c] * P.g[c].d;
#pragma unroll
                        for (int d = 0; d < 3; ++d) gj_ += Gj[c][d] * P.gj[c][d].d;
                    }
                }
                if (j >= 2 && lane_in) rot[j - 2].d = T(0);
            } else if (jin) {
                const int jp = j - 2 - n_rot, k = jp % nc, prow_i = jp / nc;
                const Du e[3] = {Du(e0[0]), Du(e0[1]), Du(e0[2])};
                VPotential<Du> P;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    P.g[c] = Du(T(0));
#pragma unroll
                    for (int d = 0; d < 3; ++d) P.gj[c][d] = Du(T(0));
                }
                v_component<Du>(SeededVals<T>{ppv, jp}, nc, k, kind, Du(lse0), e, P, tab, oob);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    gj_ += Gg[c] * P.g[c].d;
#pragma unroll
                    for (int d = 0; d < 3; ++d) gj_ += Gj[c][d] * P.gj[c][d].d;
                }
                if (prow_i == 3) gj_ -= M<T>::exp(ppv[3 * nc + k] - lse0) * GP0;                 // softmax coupling of the log-weights
            }
            if (!active || !jin) gj_ = T(0);
            bad = bad || !M<T>::finite(gj_);
#pragma unroll
            for (int t = 0; t < 2; ++t)                            // the layer's input: to every lane of the row
                if (j0 <= t && t < j0 + LV) nup[t] = __shfl(gj_, (tid & ~(LV - 1)) + (t - j0), 64);
            if (a.bcast) {
                const T s = rows_sum<T, LV>(gj_);
                if (tid < LV && jin && j >= 2) accp[a.col0[l] + (j - 2)] += s;
            } else if (active && jin && j >= 2) {
                a.g_params[row * a.gps + a.col0[l] + (j - 2)] = gj_;
            }
        }
        up[0] = nup[0]; up[1] = nup[1];
    }
    if (active && g == 0) { a.g_x[row * a.gxs + 0] = up[0]; a.g_x[row * a.gxs + 1] = up[1]; }
    bad_any = bad_any || (active && bad);
    __syncthreads();                                               // (the rows' scratch is rewritten by the next tile)
    }
    if (a.bcast) {
        __syncthreads();
        for (int j = tid; j < a.P; j += 64) atomicAdd(a.g_params + j, accp[j]);
    }
    status_add(a.status, JF_STATUS_NONFINITE, bad_any);
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
        if (!Fam::sane(layers[l])) return JF_ERR_BADARG;
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
    bool staged = false;
    constexpr int lv = 1;                                          // staged 'v' kernel: lanes per row (the kernel is written for 1, 4, 8; see its header)
    if constexpr (std::is_same<Fam, VFam>::value) {               // all layers in the default direction: the staged kernel
        staged = true;
        static const int v_dual = getenv("JF_V_BWD_DUAL") ? atoi(getenv("JF_V_BWD_DUAL")) : 0;
        // natural_direction = 1: the implicit-function adjoint of the staged kernel; under JF_V_BWD_DUAL (the check) the generic kernel, which
        // replays the Newton iteration on dual numbers
        for (int l = 0; l < n_layers; ++l)
            staged = staged && (layers[l].natural_direction == 0 || !v_dual) && rot_len(layers[l].hh_iter, 3) <= JF_V_ROT_MAX;
        if (staged) {
            a.v_dual = v_dual;
            a.scratch = (2 * n_layers + JF_V_G + 1) & ~1;          // per ROW (T units): layer inputs, G
            a.rot_max = 0;
            for (int l = 0; l < n_layers; ++l) a.rot_max = rot_len(layers[l].hh_iter, 3) > a.rot_max ? rot_len(layers[l].hh_iter, 3) : a.rot_max;
        }
    }
    a.tab = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (!Fam::needs_tab(layers[l])) continue;
        const int w = fam_tab_words<Fam>::of(layers[l]);           // families that state their bin count ('r', 'o', 'f'): 3 (bins + 1) words per lane, as in
                                                                   // the forward kernels (35 for 10 bins, not 53)
        a.tab = w > a.tab ? w : a.tab;
    }
    a.rows = 64 / lv;
    size_t lds = 0;
    // generic kernel: four directions per pass when a full wave of rows still fits the LDS with the wider dual rows, else one
    constexpr int NW = std::is_same<Fam, FFam>::value && sizeof(T) == 4 ? 6 : 4;    // 'f' float32 (2 + 10 directions by default): two passes
    bool wide = false;
    if (!staged) {
        a.shared_tab = (bwd_has_build<Fam>::value && a.bcast && a.tab > 0) ? 1 : 0;      // ('r' with permanent parameters: mchain_bwd_shared_tab)
        const size_t lds4 = ((size_t)(a.bcast ? 1 : 64) * a.tile_stride + (size_t)(a.shared_tab ? n_layers : 64) * a.tab + (size_t)64 * a.scratch) * sizeof(DualN<T, NW>) +
                            (a.bcast ? (size_t)a.P * sizeof(T) : 0);
        wide = lds4 <= 64 * 1024;                                  // (also leaves room for two workgroups per CU)
    }
    for (;;) {
        const size_t accp = a.bcast ? (size_t)a.P * sizeof(T) : 0;            // generic kernel, broadcast parameters: the workgroup's gradient sums
        const size_t tabs = (size_t)(a.shared_tab ? n_layers : a.rows) * a.tab;
        if (wide) { lds = ((size_t)(a.bcast ? 1 : a.rows) * a.tile_stride + tabs + (size_t)a.rows * a.scratch) * sizeof(DualN<T, NW>) + accp; break; }
        if (staged) {                                              // plain-value parameter tile + per lane: knot table, rotation row (duals) + per row: scratch (values)
            const size_t tile_elems = (((size_t)(a.bcast ? 1 : a.rows) * a.tile_stride) + 1) & ~(size_t)1;
            lds = tile_elems * sizeof(T) + (size_t)64 * (size_t)(a.tab + a.rot_max) * sizeof(Dual<T>) + (size_t)a.rows * (size_t)a.scratch * sizeof(T) + accp;
        } else
        lds = ((size_t)(a.bcast ? 1 : a.rows) * a.tile_stride + tabs + (size_t)a.rows * a.scratch) * sizeof(DualN<T, 1>) + accp;
        if (lds <= 160 * 1024 || a.rows == 4) break;
        a.rows >>= 1;
    }
    if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;
    if constexpr (std::is_same<Fam, VFam>::value) {
        if (staged) {
            auto kv = a.v_dual ? vchain_bwd_kernel<T, 1, 1, true> : vchain_bwd_kernel<T, 1, 1, false>;
            if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            int64_t grid = (B + a.rows - 1) / a.rows;
            if (a.bcast) {                                         // a resident set of workgroups walks the tiles (four per CU)
                int dev = 0, cus = 256;
                (void)hipGetDevice(&dev);
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                if (grid > (int64_t)cus * 4) grid = (int64_t)cus * 4;
            }
            jf::launch(kv, dim3((unsigned)grid), dim3(64), lds, (hipStream_t)stream, a);
            return check_launch();
        }
    }
    // generic kernel: one workgroup per 64-row tile; broadcast parameters: a resident set of workgroups walks the tiles (four per CU)
    auto generic_grid = [&]() -> int64_t {
        const int64_t tiles = (B + a.rows - 1) / a.rows;
        if (!a.bcast) return tiles;
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int64_t resident = (int64_t)cus * 4;
        return tiles < resident ? tiles : resident;
    };
    if (wide) {
        auto k4 = mchain_bwd_kernel<T, Fam, NW>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        jf::launch(k4, dim3((unsigned)generic_grid()), dim3(64), lds, (hipStream_t)stream, a);
        return check_launch();
    }
    auto k = mchain_bwd_kernel<T, Fam, 1>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    jf::launch(k, dim3((unsigned)generic_grid()), dim3(64), lds, (hipStream_t)stream, a);
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
// 'r' / 'o' / 'm' / 'f': the C entry points are the reverse-mode kernels (manifold_rev_kernels.hip); this replay is their check (JF_M_BWD_DUAL=1)
#define JF_DEFINE_MCHAIN_BWD_DUAL(fam, Fam, T, suffix)                                                                                         \
    int jf::dual_##fam##_chain_inv_bwd_##suffix(const T* x, int64_t xs, const T* p, int64_t ps, int32_t pb, int64_t B, int32_t n,                  \
                                                const jf_##fam##_layer* L, const T* gxo, int64_t gxos, const T* gld, const T* gblp, T* gx,        \
                                                int64_t gxs, T* gp, int64_t gps, int32_t* st, void* s) {                                        \
        return mchain_bwd<T, Fam>(x, xs, p, ps, pb, B, n, L, gxo, gxos, gld, gblp, gx, gxs, gp, gps, st, s);                                     \
    }
JF_DEFINE_MCHAIN_BWD_DUAL(r, RFam, float, f32)
JF_DEFINE_MCHAIN_BWD_DUAL(r, RFam, double, f64)
JF_DEFINE_MCHAIN_BWD_DUAL(o, OFam, float, f32)
JF_DEFINE_MCHAIN_BWD_DUAL(o, OFam, double, f64)
JF_DEFINE_MCHAIN_BWD_DUAL(m, MFam, float, f32)
JF_DEFINE_MCHAIN_BWD_DUAL(m, MFam, double, f64)
JF_DEFINE_MCHAIN_BWD_DUAL(f, FFam, float, f32)
JF_DEFINE_MCHAIN_BWD_DUAL(f, FFam, double, f64)
JF_DEFINE_MCHAIN_BWD(v, VFam, double, f64)
JF_DEFINE_MCHAIN_BWD(c, CFam, float, f32)
JF_DEFINE_MCHAIN_BWD(c, CFam, double, f64)
extern "C" int jf_v_chain_inv_bwd_f32(const float*, int64_t, const float*, int64_t, int32_t, int64_t, int32_t, const jf_v_layer*, const float*, int64_t,
                                      const float*, const float*, float*, int64_t, float*, int64_t, int32_t*, void*) { return JF_ERR_UNSUPPORTED; }
