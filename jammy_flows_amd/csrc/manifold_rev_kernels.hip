// Reverse-mode backward of the manifold-layer chains in the log-prob direction: jf_{r,o,m,f}_chain_inv_bwd_* (round 6).
//
// What torch.autograd returns for the per-block layer loop of all_layer_inverse (main/default.py:998-1031) over 'r' / 'o' / 'm' / 'f' layers,
// given upstream gradients of (x_out, log_det_out, base_logp_out).  One wave per workgroup, one sample per lane:
//   (1) a forward sweep on plain values through the forward kernels' own layer functions (Fam::apply) keeps every layer's input;
//   (2) the layers are walked in reverse with the (<= 2 + 1)-component upstream gradient; a layer's adjoint (jf_manifold_adj.h) adds its
//       parameters' gradients to the lane's gradient row in LDS;
//   (3) per-sample parameters: the gradient rows are written out coalesced; permanent (broadcast) parameters: wave sums, kept in LDS by a
//       resident set of workgroups that walks the row tiles, one atomic add per parameter and workgroup at the end.
// Cost per row: one forward chain + per layer roughly two evaluations of the layer, independent of the number of parameters up to the O(P)
// reverse of the knot tables -- the dual-number replay (manifold_bwd_kernels.hip; JF_M_BWD_DUAL=1 selects it: the check of this file) took
// (dim + P) / 4 passes over the WHOLE chain.
#include <cstdlib>
#include <type_traits>

#include "jf_manifold_adj.h"
#include "jf_manifold_rev.h"

namespace jf {

template <typename T, typename CLayer> struct MRevArgs {
    const T* x; int64_t xs;
    const T* params; int64_t ps;
    int bcast;
    int64_t B;
    int n_layers, dim, P, tile_stride, gstride, rows;
    int scr, corr, drow, dtab;                                  // per lane: plain scratch words (knot table, correlated MLP), dual elements (stage row, smooth table)
    int shared;                                                 // 'r' with permanent parameters: one knot table + one set of knot-adjoint accumulators per layer and workgroup
    int col0[JF_MAX_MCHAIN];
    CLayer L[JF_MAX_MCHAIN];
    const T* g_xout; int64_t gxos;
    const T* g_ld; const T* g_blp;
    T* g_x; int64_t gxs;
    T* g_params; int64_t gps;
    int32_t* status;
};

template <typename T> __device__ __forceinline__ T rev_wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

constexpr int REV_XIN = 2 * JF_MAX_MCHAIN;                      // per lane: every layer's input (<= 2 coordinates)

// The layer descriptors are an array INSIDE the by-value kernel argument (the C ABI hands host pointers: nothing to dereference on the device).
// Indexing that array with the loop variable made the compiler copy the whole argument struct (3.6 KB for 'f': four layers with their nested
// spline descriptors) into scratch memory at kernel entry and read every descriptor field back from there (220 scratch loads in the float32
// 'f' kernel, 3.7 KB of private memory per lane: C3's 'f' backward 0.12 -> 0.36 ms).  Read through the kernarg segment pointer instead: a
// uniform, dynamically indexed load from constant memory (scalar loads), no copy.  The argument struct is the kernel's first parameter.
template <typename A> __device__ __forceinline__ const A& kernarg_view() {
    return *reinterpret_cast<const A*>((const void*)__builtin_amdgcn_kernarg_segment_ptr());
}

template <class Adj, class = void> struct adj_has_shared : std::false_type {};
template <class Adj> struct adj_has_shared<Adj, std::void_t<decltype(Adj::HAS_SHARED)>> : std::true_type {};

// ---- 'r' chains with permanent parameters: the knot tables do not depend on the row.  Lane l builds layer l's table once; a row costs the bin
// search, the seven-tangent bin evaluation and six LDS atomic adds per layer; the tables are reversed once per workgroup at the end.  Four waves
// per workgroup and two workgroups per CU: what the launch costs beyond the rows is one global atomic per parameter and workgroup, all on
// the same few words (4096 one-wave workgroups x 16 parameters at 2^18 rows: 0.08 ms of atomics for 0.01 ms of work).
constexpr int REV_SHARED_THREADS = 256;
template <typename T, class Fam, class Adj>
__global__ void __launch_bounds__(REV_SHARED_THREADS) mchain_rev_shared_kernel(const MRevArgs<T, typename Fam::CLayer> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const MRevArgs<T, typename Fam::CLayer>& ka = kernarg_view<MRevArgs<T, typename Fam::CLayer>>();
    const int tid = threadIdx.x, nt = REV_SHARED_THREADS;
    T* tile = reinterpret_cast<T*>(smem_raw + 16);                              // [P] the parameter row (offset 16: see mchain_rev_kernel)
    T* accp = tile + a.tile_stride;                                             // [P] the workgroup's parameter gradients
    T* stab = accp + a.tile_stride;                                             // [n_layers][scr] knot tables
    T* sacc = stab + a.n_layers * a.scr;                                        // [n_layers][scr] knot-adjoint accumulators (gcw | gch | gd)
    for (int j = tid; j < a.P; j += nt) { tile[j] = a.params[j]; accp[j] = T(0); }
    for (int j = tid; j < a.n_layers * a.scr; j += nt) sacc[j] = T(0);
    __syncthreads();
    if (tid < a.n_layers) Fam::template build<T>(ka.L[tid], tile + ka.col0[tid], stab + tid * a.scr);
    __syncthreads();
    AdjLane<T> A;
    A.scr = nullptr; A.corr = nullptr; A.drow = nullptr; A.dtab = nullptr;
    bool bad_any = false;
    const int64_t n_tiles = (a.B + nt - 1) / nt;
    for (int64_t tile_i = blockIdx.x; tile_i < n_tiles; tile_i += gridDim.x) {
        const int64_t row = tile_i * nt + tid;
        const bool active = row < a.B;
        const int64_t rrow = active ? row : a.B - 1;
        T x[3] = {a.x[rrow * a.xs], T(0), T(0)};
        T g[3] = {(a.g_xout && active) ? a.g_xout[rrow * a.gxos] : T(0), T(0), T(0)};
        const T gld = (a.g_ld && active) ? a.g_ld[rrow] : T(0);
        const T gblp = (a.g_blp && active) ? a.g_blp[rrow] : T(0);
        A.lane_valid = active;
        A.oob = A.nonconv = A.nonfinite = false;
        T xin[JF_MAX_MCHAIN];
        {
            LaneCtx<T> ctx;
            ctx.corr = nullptr; ctx.bins = nullptr; ctx.bin_i = 0;
            ctx.oob = ctx.nonconv = ctx.nonfinite = false;
            ctx.lane_valid = active;
            ctx.tab_built = true;
            T ld = T(0);
#pragma unroll
            for (int i = 0; i < JF_MAX_MCHAIN; ++i) {
                const int l = a.n_layers - 1 - i;
                if (l < 0) break;
                xin[l] = x[0];
                if (l == 0) break;                                 // the last layer applied is evaluated by its adjoint
                ctx.tab = stab + l * a.scr;
                Fam::template apply<T, false>(ka.L[l], tile + ka.col0[l], x, ld, ctx);
            }
        }
#pragma unroll
        for (int l = 0; l < JF_MAX_MCHAIN; ++l) {
            if (l >= a.n_layers) break;
            const T xi[3] = {xin[l], T(0), T(0)};
            Adj::template adjoint_shared<T>(ka.L[l], tile + ka.col0[l], stab + l * a.scr, sacc + l * a.scr, xi, g, gld, l == 0 ? gblp : T(0), active, A);
        }
        bad_any = bad_any || (active && !M<T>::finite(g[0]));
        if (active) a.g_x[row * a.gxs] = g[0];
    }
    __syncthreads();
    if (tid < a.n_layers) {
        const int l = tid, nb = ka.L[l].sp.num_bins;
        T* acc = sacc + l * a.scr;
        spline_adj_table_reverse_dense<T>(tile + ka.col0[l], accp + ka.col0[l], to_dev<T>(ka.L[l].sp), stab + l * a.scr, acc, acc + (nb + 1), acc + 2 * (nb + 1), (T)ka.L[l].lo,
                                          (T)ka.L[l].hi);
    }
    __syncthreads();
    for (int j = tid; j < a.P; j += nt) {
        const T v = accp[j];
        bad_any = bad_any || !M<T>::finite(v);
        atomicAdd(a.g_params + j, v);
    }
    status_add(a.status, JF_STATUS_NONFINITE, bad_any);
}

template <typename T, class Fam, class Adj>
__global__ void __launch_bounds__(64) mchain_rev_kernel(const MRevArgs<T, typename Fam::CLayer> a) {
    using Du = DualN<T, ADJ_N>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const MRevArgs<T, typename Fam::CLayer>& ka = kernarg_view<MRevArgs<T, typename Fam::CLayer>>();
    const int tid = threadIdx.x;
    const int rows = a.rows;
    const int tile_rows = a.bcast ? 1 : rows;
    const bool lane_in = tid < rows;
    const int slot = lane_in ? tid : 0;
    // ---- LDS carve-up (T units, then the dual region 16-byte aligned)
    // (the first 16 bytes stay unused: a parameter row at LDS offset 0 handed to a function the compiler did not inline -- moebius_adjoint took
    //  `const T*` as a generic pointer -- faulted with a memory aperture violation: the local -> flat cast reads offset 0 as the null pointer)
    T* tile = reinterpret_cast<T*>(smem_raw + 16);                              // [tile_rows][tile_stride] parameter rows
    T* gtile = tile + tile_rows * a.tile_stride;                                // [rows][gstride] gradient rows
    T* scr0 = gtile + rows * a.gstride;                                         // [rows][scr + corr + REV_XIN]
    const int lane_words = a.scr + a.corr + REV_XIN;
    T* accp = scr0 + rows * lane_words;                                         // [P] (broadcast parameters)
    size_t off_d = (size_t)((accp + (a.bcast ? a.P : 0)) - reinterpret_cast<T*>(smem_raw)) * sizeof(T);
    off_d = (off_d + 15) & ~(size_t)15;
    Du* dual0 = reinterpret_cast<Du*>(smem_raw + off_d);                        // [rows][drow + dtab]
    AdjLane<T> A;
    A.scr = scr0 + slot * lane_words;
    A.corr = A.scr + a.scr;
    T* xin_l = A.corr + a.corr;
    A.drow = dual0 + slot * (a.drow + a.dtab);
    A.dtab = A.drow + a.drow;
    T* grow = gtile + slot * a.gstride;

    if (a.bcast) {
        for (int j = tid; j < a.P; j += 64) { tile[j] = a.params[j]; accp[j] = T(0); }
    } else {
        const int64_t row0 = (int64_t)blockIdx.x * rows;
        const int total = rows * a.P;
        for (int i0 = 0; i0 < total; i0 += 64 * 8) {             // eight loads in flight per lane
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
    const T* prow = tile + (a.bcast ? 0 : slot * a.tile_stride);
    bool bad_any = false;
    const int64_t n_tiles = (a.B + rows - 1) / rows;
    const int64_t t_end = a.bcast ? n_tiles : (int64_t)blockIdx.x + 1;
    for (int64_t tile_i = blockIdx.x; tile_i < t_end; tile_i += gridDim.x) {
        const int64_t row0 = tile_i * rows;
        const int64_t row = row0 + tid;
        const bool active = lane_in && row < a.B;
        const int64_t rrow = active ? row : a.B - 1;
        if (lane_in) for (int j = 0; j < a.P; ++j) grow[j] = T(0);
        T x[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int d = 0; d < Fam::DIM; ++d) if (d < a.dim) x[d] = a.x[rrow * a.xs + d];
        T g[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int d = 0; d < Fam::DIM; ++d) if (d < a.dim && a.g_xout && active) g[d] = a.g_xout[rrow * a.gxos + d];
        const T gld = (a.g_ld && active) ? a.g_ld[rrow] : T(0);
        const T gblp = (a.g_blp && active) ? a.g_blp[rrow] : T(0);
        A.lane_valid = active;
        A.oob = A.nonconv = A.nonfinite = false;
        // (1) forward sweep: every layer's input (the last layer applied, layer 0, is evaluated by its own adjoint)
        {
            LaneCtx<T> ctx;
            ctx.tab = A.scr; ctx.corr = A.corr; ctx.bins = nullptr; ctx.bin_i = 0;
            ctx.oob = ctx.nonconv = ctx.nonfinite = false;
            ctx.lane_valid = active;
            T ld = T(0);
#pragma unroll 1
            for (int l = a.n_layers - 1; l >= 0; --l) {
                if (lane_in) { xin_l[2 * l] = x[0]; xin_l[2 * l + 1] = x[1]; }
                if (l == 0) break;
                if (lane_in) Fam::template apply<T, false>(ka.L[l], prow + ka.col0[l], x, ld, ctx);
            }
        }
        // (2) reverse sweep (layer n-1 was applied first: layer 0 last; it also takes the base log-prob term -1/2 out^2)
#pragma unroll 1
        for (int l = 0; l < a.n_layers; ++l) {
            if (lane_in) {
                const T xi[3] = {xin_l[2 * l], xin_l[2 * l + 1], T(0)};
                Adj::template adjoint<T>(ka.L[l], prow + ka.col0[l], grow + ka.col0[l], xi, g, gld, l == 0 ? gblp : T(0), A);
            }
        }
        bool bad = false;
#pragma unroll
        for (int d = 0; d < Fam::DIM; ++d) {
            if (d < a.dim) {
                bad = bad || !M<T>::finite(g[d]);
                if (active) a.g_x[row * a.gxs + d] = g[d];
            }
        }
        // (3) the parameters' gradients
        if (a.bcast) {
            for (int j = 0; j < a.P; ++j) {
                const T v = active ? grow[j] : T(0);
                bad = bad || !M<T>::finite(v);
                const T s = rev_wave_sum<T>(v);
                if (tid == 0) accp[j] += s;
            }
        } else {
            __syncthreads();
            const int total = rows * a.P;
            for (int i = tid; i < total; i += 64) {
                const int rr = i / a.P, j = i - rr * a.P;
                if (row0 + rr < a.B) {
                    const T v = gtile[rr * a.gstride + j];
                    bad = bad || !M<T>::finite(v);
                    a.g_params[(row0 + rr) * a.gps + j] = v;
                }
            }
            __syncthreads();
        }
        bad_any = bad_any || bad || (active && A.nonfinite);
    }
    if (a.bcast) {
        __syncthreads();
        for (int j = tid; j < a.P; j += 64) atomicAdd(a.g_params + j, accp[j]);
    }
    status_add(a.status, JF_STATUS_NONFINITE, bad_any);       // (out-of-range inputs / non-convergence: reported by the forward call of the same step)
}

template <typename T, class Fam, class Adj>
static int mchain_rev(const T* x, int64_t xs, const T* params, int64_t ps, int32_t pb, int64_t B, int32_t n_layers, const typename Fam::CLayer* layers,
                      const T* g_xout, int64_t gxos, const T* g_ld, const T* g_blp, T* g_x, int64_t gxs, T* g_params, int64_t gps, int32_t* status,
                      void* stream) {
    if (!x || !g_x || !layers || n_layers < 1 || n_layers > JF_MAX_MCHAIN || B < 0) return JF_ERR_BADARG;
    if (pb != 1 && pb != B) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    MRevArgs<T, typename Fam::CLayer> a{};
    int col = 0;
    bool all_hand = true;
    for (int l = 0; l < n_layers; ++l) {
        if (!Fam::sane(layers[l])) return JF_ERR_BADARG;
        a.L[l] = layers[l];
        a.col0[l] = col;
        col += Fam::row_len(layers[l]);
        const int s = Adj::scr_words(layers[l]), c = Adj::corr_words(layers[l]), dr = Adj::dual_row(layers[l]), dt = Adj::dual_tab(layers[l]);
        a.scr = s > a.scr ? s : a.scr;
        a.corr = c > a.corr ? c : a.corr;
        a.drow = dr > a.drow ? dr : a.drow;
        a.dtab = dt > a.dtab ? dt : a.dtab;
        all_hand = all_hand && dr == 0 && dt == 0;
    }
    if (col > 0 && (!params || !g_params)) return JF_ERR_BADARG;
    if constexpr (std::is_same<Fam, FFam>::value) {
        for (int l = 0; l < n_layers; ++l) {
            if (!layers[l].correlated) continue;
            if (layers[l].corr_hidden < 1 || layers[l].corr_rank < 0 || FFam::corr_out(layers[l]) + layers[l].corr_rank > JF_CORR_SCRATCH - 1)
                return JF_ERR_UNSUPPORTED;
        }
    }
    // the forward sweep's knot tables live in the same scratch
    for (int l = 0; l < n_layers; ++l)
        if (Fam::needs_tab(layers[l])) { const int w = fam_tab_words<Fam>::of(layers[l]); a.scr = w > a.scr ? w : a.scr; }
    if ((a.scr + a.corr) % 2 == 0) a.scr += 1;                     // odd lane stride (REV_XIN is even)
    a.x = x; a.xs = xs; a.params = params; a.ps = ps; a.bcast = (pb == 1) ? 1 : 0; a.B = B; a.n_layers = n_layers; a.P = col;
    a.tile_stride = col > 0 ? (col | 1) : 1;
    a.gstride = col > 0 ? (col | 1) : 1;
    a.dim = Fam::DIM;
    a.g_xout = g_xout; a.gxos = gxos; a.g_ld = g_ld; a.g_blp = g_blp; a.g_x = g_x; a.gxs = gxs; a.g_params = g_params; a.gps = gps; a.status = status;
    a.rows = 64;
    a.shared = (adj_has_shared<Adj>::value && a.bcast && all_hand && col > 0) ? 1 : 0;
    size_t lds = 0;
    auto k = mchain_rev_kernel<T, Fam, Adj>;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if constexpr (adj_has_shared<Adj>::value) {
        if (a.shared) {
            lds = 16 + ((size_t)2 * a.tile_stride + (size_t)2 * n_layers * a.scr) * sizeof(T);
            int64_t grid = (B + REV_SHARED_THREADS - 1) / REV_SHARED_THREADS;
            if (grid > (int64_t)cus * 2) grid = (int64_t)cus * 2;       // resident workgroups walk the tiles
            jf::launch(mchain_rev_shared_kernel<T, Fam, Adj>, dim3((unsigned)grid), dim3(REV_SHARED_THREADS), lds, (hipStream_t)stream, a);
            return check_launch();
        }
    }
    for (;;) {
        const size_t plain = 16 + ((size_t)(a.bcast ? 1 : a.rows) * a.tile_stride + (size_t)a.rows * a.gstride + (size_t)a.rows * (a.scr + a.corr + REV_XIN) +
                                   (a.bcast ? (size_t)a.P : 0)) * sizeof(T);
        lds = ((plain + 15) & ~(size_t)15) + (size_t)a.rows * (a.drow + a.dtab) * sizeof(DualN<T, ADJ_N>);
        if (lds <= 40 * 1024 || a.rows == 4) break;               // (<= 40 KB: four workgroups per CU, one wave per SIMD)
        if (lds <= 64 * 1024 && a.rows <= 32) break;
        a.rows >>= 1;
    }
    if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int64_t grid = (B + a.rows - 1) / a.rows;
    if (a.bcast) {                                                 // a resident set of workgroups walks the tiles
        const int64_t per_cu = lds > 0 ? (int64_t)(160 * 1024 / lds) : 8;
        const int64_t resident = (int64_t)cus * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
        if (grid > resident) grid = resident;
    }
    jf::launch(k, dim3((unsigned)grid), dim3(64), lds, (hipStream_t)stream, a);
    return check_launch();
}

static bool use_dual_replay() {
    static const int v = getenv("JF_M_BWD_DUAL") ? atoi(getenv("JF_M_BWD_DUAL")) : 0;
    return v > 0;
}
// chains the dual-number replay serves better: 'f' layers without nested flows carry a dozen parameters -- two six-tangent passes over the
// whole layer (C3's 'f' backward: 0.12 ms per 2^18 rows) against three stage passes + a plain evaluation here (0.17); JF_M_BWD_DUAL=-1 sends
// them through the reverse sweep all the same (its check runs on every fixture)
template <class CLayer> static bool prefers_dual(const CLayer*, int) { return false; }
template <> bool prefers_dual<jf_f_layer>(const jf_f_layer* L, int n) {
    static const int v = getenv("JF_M_BWD_DUAL") ? atoi(getenv("JF_M_BWD_DUAL")) : 0;
    if (v < 0 || !L) return false;
    for (int l = 0; l < n && l < JF_MAX_MCHAIN; ++l)
        if (L[l].n_vertical != 0 || L[l].n_circular != 0 || L[l].correlated) return false;
    return true;
}

}  // namespace jf

using namespace jf;

#define JF_DEFINE_MCHAIN_REV(fam, Fam, Adj, T, suffix)                                                                                         \
    extern "C" int jf_##fam##_chain_inv_bwd_##suffix(const T* x, int64_t xs, const T* p, int64_t ps, int32_t pb, int64_t B, int32_t n,             \
                                                     const jf_##fam##_layer* L, const T* gxo, int64_t gxos, const T* gld, const T* gblp, T* gx,   \
                                                     int64_t gxs, T* gp, int64_t gps, int32_t* st, void* s) {                                   \
        if (use_dual_replay() || prefers_dual(L, n))                                                                                           \
            return jf::dual_##fam##_chain_inv_bwd_##suffix(x, xs, p, ps, pb, B, n, L, gxo, gxos, gld, gblp, gx, gxs, gp, gps, st, s);             \
        return mchain_rev<T, Fam, Adj>(x, xs, p, ps, pb, B, n, L, gxo, gxos, gld, gblp, gx, gxs, gp, gps, st, s);                                \
    }
JF_DEFINE_MCHAIN_REV(r, RFam, RAdj, float, f32)
JF_DEFINE_MCHAIN_REV(r, RFam, RAdj, double, f64)
JF_DEFINE_MCHAIN_REV(o, OFam, OAdj, float, f32)
JF_DEFINE_MCHAIN_REV(o, OFam, OAdj, double, f64)
JF_DEFINE_MCHAIN_REV(m, MFam, MAdj, float, f32)
JF_DEFINE_MCHAIN_REV(m, MFam, MAdj, double, f64)
JF_DEFINE_MCHAIN_REV(f, FFam, FAdj, float, f32)
JF_DEFINE_MCHAIN_REV(f, FFam, FAdj, double, f64)
