// Weight / bias gradient of a dense layer:  g_W (N, K) = sum_b g[b, :]^T in[b, :],   g_b (N) = sum_b g[b, :]   (B rows, B >> N, K).
//
// This is the one product of a linear layer's backward that reduces over the BATCH: an (N x B) (B x K) GEMM whose output is tiny (C5: 1224 x 8,
// C3: 548 x 128) and whose inner dimension is 1e5..1e6.  rocBLAS picks a 128 x 128 output tiling for it, i.e. a handful of workgroups walking
// the whole batch serially: 14 ms per call in float64 at 2^17 rows, 8 calls per C5 training step (profiles/r02_train.md).  Here the batch is
// split over the grid: wave (n-tile, s) reduces rows [s Bc, (s+1) Bc) of a 32..64-column tile of g against all K <= 128 input columns into
// partials[s] (N, K) / bias_partials[s] (N); the caller adds the few partial slabs (deterministic: no atomics).
//
// On the matrix cores (f32 32x32x2 / f64 16x16x4): 2.9e11 flop for C3 at 2^20 rows would be 3.7 ms on the VALU, 1.9 ms at the f32 MFMA peak.
// Algorithmic bytes = s (B N + ceil(N / 64) B K + S N K): g is read once, `in` once per n-tile.
#include "jf_common.h"
#include "jf_mfma.h"
#include "jf_math.h"

namespace jf {

// One wave per workgroup.  The wave owns NA x MT consecutive columns n of g (MFMA A operand: A[m = n][slot = row]) and all K <= KT x MT input
// columns (B operand: B[slot = row][n = k]); every MFMA step consumes KS rows of the chunk (f32: 32x32x2, f64: 16x16x4).  Both operand loads are
// coalesced row segments (lane -> consecutive column).  D[m][n] = g_W[n0 + row_of(reg, lane)][k0 + lane % MT]: stores are coalesced along k.
template <typename T, int NA, int KT>
__global__ void __launch_bounds__(64) wgrad_kernel(const T* __restrict__ g, int64_t gs, const T* __restrict__ in, int64_t is, int64_t B, int K, int N,
                                                   int64_t rows_per_split, T* __restrict__ pw, T* __restrict__ pb) {
    using MM = Mfma<T>;
    constexpr int MT = MM::MT, KS = MM::KS;
    const int lane = threadIdx.x, c = lane % MT, slot = lane / MT;
    const int n0 = blockIdx.x * NA * MT;
    const int64_t b0 = (int64_t)blockIdx.y * rows_per_split;
    const int64_t b1 = b0 + rows_per_split < B ? b0 + rows_per_split : B;
    int nidx[NA], kidx[KT];
#pragma unroll
    for (int a = 0; a < NA; ++a) { const int n = n0 + a * MT + c; nidx[a] = n < N ? n : N - 1; }
#pragma unroll
    for (int t = 0; t < KT; ++t) { const int k = t * MT + c; kidx[t] = k < K ? k : K - 1; }
    typename MM::Acc acc[NA][KT];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < MM::NREG; ++r) acc[a][t][r] = T(0);
    T bsum[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) bsum[a] = T(0);
    // operands of step i + 1 are requested before the MFMAs of step i are issued (one step = 6 dependent-free loads and 8 MFMAs for the widest
    // shape: without the prefetch every step waited a full HBM round trip in front of its MFMAs)
    T av[NA], bv[KT], an[NA], bn[KT];
    auto fetch = [&](int64_t b, T (&a_)[NA], T (&b_)[KT]) {
        const int64_t row = b + slot;
        const bool valid = row < b1;
        const int64_t rr = valid ? row : (b1 > b0 ? b1 - 1 : 0);
#pragma unroll
        for (int a = 0; a < NA; ++a) a_[a] = g[rr * gs + nidx[a]];
#pragma unroll
        for (int t = 0; t < KT; ++t) b_[t] = in[rr * is + kidx[t]];
#pragma unroll
        for (int a = 0; a < NA; ++a) a_[a] = valid ? a_[a] : T(0);
    };
    if (b0 < b1) fetch(b0, av, bv);
    for (int64_t b = b0; b < b1; b += KS) {
        const bool more = b + KS < b1;
        if (more) fetch(b + KS, an, bn);
#pragma unroll
        for (int a = 0; a < NA; ++a) bsum[a] += av[a];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int t = 0; t < KT; ++t) acc[a][t] = MM::mma(av[a], bv[t], acc[a][t]);
        if (more) {
#pragma unroll
            for (int a = 0; a < NA; ++a) av[a] = an[a];
#pragma unroll
            for (int t = 0; t < KT; ++t) bv[t] = bn[t];
        }
    }
    T* slab = pw + (int64_t)blockIdx.y * N * K;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const int k = t * MT + c;
#pragma unroll
            for (int r = 0; r < MM::NREG; ++r) {
                const int n = n0 + a * MT + MM::row_of(r, lane);
                if (n < N && k < K) slab[(int64_t)n * K + k] = acc[a][t][r];
            }
        }
    if (pb != nullptr) {
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            T v = bsum[a];
            for (int off = MT; off < 64; off <<= 1) v += __shfl_xor(v, off);        // the KS slots of a column sit MT lanes apart
            const int n = n0 + a * MT + c;
            if (slot == 0 && n < N) pb[(int64_t)blockIdx.y * N + n] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- one tiny dimension
// min(N, K) <= 16 (the rank-8 stages of AmortizableMLP, the 4 / 7 / 10-wide layers of the default MLPs): a pure stream over the wide operand.
// thread = column c of the WIDE matrix (coalesced loads), the S <= 16 values of the narrow row are wave-uniform (scalar loads), S accumulators
// per thread; a workgroup (64 .. 256 columns) walks its row range, one partial slab per range.  G_WIDE: the wide operand is g (N = C), the
// narrow one `in` (K = S); else the wide operand is `in` (K = C) and g the narrow one (N = S).
template <typename T, int S, bool G_WIDE>
__global__ void __launch_bounds__(256) wgrad_skinny_kernel(const T* __restrict__ wide, int64_t ws, const T* __restrict__ narrow, int64_t ns, int64_t B, int C,
                                                           int K, int N, int64_t rows_per_split, T* __restrict__ pw, T* __restrict__ pb) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = c < C;
    const int cc = live ? c : C - 1;
    const int64_t b0 = (int64_t)blockIdx.y * rows_per_split;
    const int64_t b1 = b0 + rows_per_split < B ? b0 + rows_per_split : B;
    T acc[S], bsum = T(0), nsum[S];
#pragma unroll
    for (int s = 0; s < S; ++s) { acc[s] = T(0); nsum[s] = T(0); }
#pragma unroll 8
    for (int64_t b = b0; b < b1; ++b) {
        const T v = wide[b * ws + cc];
        const T* x = narrow + b * ns;                            // uniform: scalar loads
        bsum += v;
#pragma unroll
        for (int s = 0; s < S; ++s) { const T xv = x[s]; acc[s] += v * xv; nsum[s] += xv; }
    }
    T* slab = pw + (int64_t)blockIdx.y * N * K;
    if (live) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (G_WIDE) slab[(int64_t)c * K + s] = acc[s];       // g_W[n = c][k = s]
            else slab[(int64_t)s * K + c] = acc[s];              // g_W[n = s][k = c]
        }
    }
    if (pb != nullptr) {
        if (G_WIDE) { if (live) pb[(int64_t)blockIdx.y * N + c] = bsum; }
        else if (c == 0) {
#pragma unroll
            for (int s = 0; s < S; ++s) pb[(int64_t)blockIdx.y * N + s] = nsum[s];
        }
    }
}

constexpr int WGS_MAX = 16;
static inline bool wgrad_is_skinny(int32_t K, int32_t N) { return (K <= WGS_MAX || N <= WGS_MAX) && (K >= 32 || N >= 32 || (K <= WGS_MAX && N <= WGS_MAX)); }
static inline int wgrad_skinny_threads(int C) { return C <= 64 ? 64 : C <= 128 ? 128 : 256; }
static inline int64_t wgrad_skinny_splits(int64_t B, int32_t K, int32_t N) {
    const int C = (K <= WGS_MAX && N >= K) ? N : K;              // the wide side
    const int threads = wgrad_skinny_threads(C);
    const int64_t waves = (int64_t)((C + threads - 1) / threads) * (threads / 64);
    int64_t s = (8192 + waves - 1) / waves;                      // ~8 waves per SIMD: every thread walks its rows serially, latency is hidden by waves
    const int64_t max_s = (B + 31) / 32;                         // at least 32 rows per split
    if (s > max_s) s = max_s;
    if (s > 65535) s = 65535;
    return s < 1 ? 1 : s;
}

template <typename T, bool G_WIDE>
static bool wgrad_skinny_go(int S, dim3 grid, dim3 block, hipStream_t st, const T* wide, int64_t ws, const T* narrow, int64_t ns, int64_t B, int C, int K, int N,
                            int64_t rps, T* pw, T* pb) {
#define JF_WS(S_) case S_: jf::launch((wgrad_skinny_kernel<T, S_, G_WIDE>), grid, block, 0, st, wide, ws, narrow, ns, B, C, K, N, rps, pw, pb); return true;
    switch (S) { JF_WS(1) JF_WS(2) JF_WS(3) JF_WS(4) JF_WS(5) JF_WS(6) JF_WS(7) JF_WS(8) JF_WS(9) JF_WS(10) JF_WS(11) JF_WS(12) JF_WS(13) JF_WS(14) JF_WS(15)
                 JF_WS(16) default: return false; }
#undef JF_WS
}

template <typename T> static bool wgrad_skinny(const T* g, int64_t gs, const T* in, int64_t is, int64_t B, int32_t K, int32_t N, T* pw, T* pb, hipStream_t st) {
    if (!wgrad_is_skinny(K, N)) return false;
    const bool g_wide = (K <= WGS_MAX && N >= K);
    const int C = g_wide ? N : K, S = g_wide ? K : N;
    const int64_t splits = wgrad_skinny_splits(B, K, N);
    const int64_t rps = (B + splits - 1) / splits;
    const int threads = wgrad_skinny_threads(C);
    const dim3 grid((unsigned)((C + threads - 1) / threads), (unsigned)splits), block(threads);
    return g_wide ? wgrad_skinny_go<T, true>(S, grid, block, st, g, gs, in, is, B, C, K, N, rps, pw, pb)
                  : wgrad_skinny_go<T, false>(S, grid, block, st, in, is, g, gs, B, C, K, N, rps, pw, pb);
}

// ---------------------------------------------------------------------------------------------------------- narrow heads, whole backward
// Backward of a Linear - tanh - Linear head with few inputs and few outputs (K1 <= 32, N <= 16, or K1 <= 8, N <= 64; H hidden units: the
// 4 -> 128 -> 10 MLP that parametrises an 'f' layer, 4 -> 128 -> 46 with the spline options of c3b) in ONE launch: what autograd runs as tanh', two weight gradients, two bias sums and grad_output @ W2.
// thread = hidden unit j.  Everything a hidden unit needs is its own: W1[j][:], W2[:][j] in registers; the input row and the upstream row are
// wave-uniform (scalar loads).  Per row: h_j = tanh(W1[j] . x + b1[j]) recomputed (the forward kept nothing), g_j = (g_out . W2[:, j])(1 - h_j^2),
// and the accumulators  g_W1[j][k] += g_j x[k],  g_b1[j] += g_j,  g_W2[n][j] += g_out[n] h_j  -- no communication between threads at all.
// A workgroup walks its row range and writes one partial slab [H][K1 + 1 + N] (+ [N] for g_b2), summed by the caller.  No gradient with respect
// to the input rows (they are data; the caller takes the layer-by-layer path when they require grad).
constexpr int MS_K1MAX = 32, MS_NMAX = 16, MS_NWIDE = 64, MS_NWIDE64 = 48, MS_K1WIDE = 8;      // N <= 16, or N <= 64 (float64: 48, the register file) with K1 <= 8
// KB / NB: compile-time bounds of the input / output loops (the sizes rounded up to 1, 2, 4, 8, 16, 32 / 4, 8, 12, 16, 32, 48, 64).  Slots beyond the real sizes carry
// zero weights and read a clamped (duplicate) element, so the row loop has no size-dependent branch and its scalar loads batch up.
template <typename T, int KB, int NB>
__global__ void __launch_bounds__(128) mlp2_small_bwd_kernel(const T* __restrict__ x, int64_t xs, const T* __restrict__ W1, int64_t w1s, const T* __restrict__ b1,
                                                             const T* __restrict__ W2, int64_t w2s, const T* __restrict__ g, int64_t gs, int64_t B, int K1, int H,
                                                             int N, int64_t rows_per_block, T* __restrict__ slab, T* __restrict__ slab_b2, int dense) {
    const int j = threadIdx.x;
    const bool live = j < H;
    const int jj = live ? j : H - 1;
    T w1[KB], w2[NB], a1[KB], a2[NB];
    int kx[KB], nx[NB];
#pragma unroll
    for (int k = 0; k < KB; ++k) { w1[k] = k < K1 ? W1[(int64_t)jj * w1s + k] : T(0); a1[k] = T(0); kx[k] = k < K1 ? k : K1 - 1; }
#pragma unroll
    for (int n = 0; n < NB; ++n) { w2[n] = n < N ? W2[(int64_t)n * w2s + jj] : T(0); a2[n] = T(0); nx[n] = n < N ? n : N - 1; }
    const T bj = b1[jj];
    T ab1 = T(0);
    // g_b2 = the column sums of g: thread j keeps column j % N, read with a vector load of its own (the upstream row sits in scalar registers
    // for the products below; summing all NB columns there in every thread was a quarter of the loop's vector instructions)
    const int nj = j % N;
    T gown = T(0);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < B ? r0 + rows_per_block : B;
    // rows whose KB / NB-element reads stay inside the arrays are read CONTIGUOUSLY (one or two wide scalar loads per row; the slots beyond K1 / N
    // then hold the NEXT ROW's first values, which meet zero weights); only the last rows of the arrays take the clamped indices.  `dense` (host:
    // every padded array has stride = width): the values in the padded slots are rows of the same array, so a non-finite one poisons the same
    // batch sums through its own row anyway; with gaps between the rows (views of wider tensors) the gap's content is unknown (0 * NaN = NaN)
    // and every row takes the clamped reads.  (Selecting zero per padded slot in this loop cost 40 % of the kernel: 0.088 -> 0.125 ms.)
    const int64_t safe = B - 1 - ((KB - K1 + xs - 1) / xs > (NB - N + gs - 1) / gs ? (KB - K1 + xs - 1) / xs : (NB - N + gs - 1) / gs);
    const int64_t rm = !dense ? r0 : (r1 < safe ? r1 : (safe > r0 ? safe : r0));
    auto one_row = [&](const T (&xv)[KB], const T (&gv)[NB], T gcol) {
        gown += gcol;
        T pre = bj;
#pragma unroll
        for (int k = 0; k < KB; ++k) pre += w1[k] * xv[k];
        const T h = M<T>::tanh_fast(pre);                          // the forward kernels' tanh (jf_mlp2 / jf_linear)
        T gh = T(0);
#pragma unroll
        for (int n = 0; n < NB; ++n) { gh += gv[n] * w2[n]; a2[n] += gv[n] * h; }
        gh *= T(1) - h * h;
        ab1 += gh;
#pragma unroll
        for (int k = 0; k < KB; ++k) a1[k] += gh * xv[k];
    };
    constexpr int RU = NB > 16 ? 1 : 4;                            // rows in flight: their NB upstream values sit in scalar registers (~100)
#pragma unroll RU
    for (int64_t r = r0; r < rm; ++r) {
        const T* xr = x + r * xs;                                 // uniform addresses: scalar loads
        const T* gr = g + r * gs;
        T xv[KB], gv[NB];
#pragma unroll
        for (int k = 0; k < KB; ++k) xv[k] = xr[k];
#pragma unroll
        for (int n = 0; n < NB; ++n) gv[n] = gr[n];
        one_row(xv, gv, gr[nj]);
    }
    for (int64_t r = rm; r < r1; ++r) {
        const T* xr = x + r * xs;
        const T* gr = g + r * gs;
        T xv[KB], gv[NB];
#pragma unroll
        for (int k = 0; k < KB; ++k) xv[k] = xr[kx[k]];
#pragma unroll
        for (int n = 0; n < NB; ++n) gv[n] = gr[nx[n]];
        one_row(xv, gv, gr[nj]);
    }
    if (live) {
        T* row = slab + ((int64_t)blockIdx.x * H + j) * (K1 + 1 + N);
#pragma unroll
        for (int k = 0; k < KB; ++k) if (k < K1) row[k] = a1[k];
        row[K1] = ab1;
#pragma unroll
        for (int n = 0; n < NB; ++n) if (n < N) row[K1 + 1 + n] = a2[n];
    }
    if (j < N) slab_b2[(int64_t)blockIdx.x * N + j] = gown;
}

// the first layer alone: the gradient with respect to the hidden activations (B, H) is given (the second layer's input-gradient product), the
// tanh derivative and the first layer's weight / bias gradient follow in one launch (slab [H][K1 + 1]); same thread = hidden unit scheme
template <typename T, int KB>
__global__ void __launch_bounds__(128) mlp_hidden_bwd_kernel(const T* __restrict__ x, int64_t xs, const T* __restrict__ W1, int64_t w1s, const T* __restrict__ b1,
                                                             const T* __restrict__ gh, int64_t ghs, int64_t B, int K1, int H, int64_t rows_per_block,
                                                             T* __restrict__ slab, int dense) {
    const int j = threadIdx.x;
    const bool live = j < H;
    const int jj = live ? j : H - 1;
    T w1[KB], a1[KB];
    int kx[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) { w1[k] = k < K1 ? W1[(int64_t)jj * w1s + k] : T(0); a1[k] = T(0); kx[k] = k < K1 ? k : K1 - 1; }
    const T bj = b1[jj];
    T ab1 = T(0);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < B ? r0 + rows_per_block : B;
    const int64_t safe = B - 1 - (KB - K1 + xs - 1) / xs;          // contiguous KB-element reads of x stay inside the array up to this row
    const int64_t rm = !dense ? r0 : (r1 < safe ? r1 : (safe > r0 ? safe : r0));   // dense: see mlp2_small_bwd_kernel
    auto one_row = [&](const T (&xv)[KB], T gr) {
        T pre = bj;
#pragma unroll
        for (int k = 0; k < KB; ++k) pre += w1[k] * xv[k];
        const T h = M<T>::tanh_fast(pre);
        const T g = gr * (T(1) - h * h);
        ab1 += g;
#pragma unroll
        for (int k = 0; k < KB; ++k) a1[k] += g * xv[k];
    };
#pragma unroll 4
    for (int64_t r = r0; r < rm; ++r) {
        const T* xr = x + r * xs;                                 // uniform addresses: scalar loads
        T xv[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) xv[k] = xr[k];
        one_row(xv, gh[r * ghs + jj]);
    }
    for (int64_t r = rm; r < r1; ++r) {
        const T* xr = x + r * xs;
        T xv[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) xv[k] = xr[kx[k]];
        one_row(xv, gh[r * ghs + jj]);
    }
    if (live) {
        T* row = slab + ((int64_t)blockIdx.x * H + j) * (K1 + 1);
#pragma unroll
        for (int k = 0; k < KB; ++k) if (k < K1) row[k] = a1[k];
        row[K1] = ab1;
    }
}

static int64_t mlp2_small_slabs(int64_t B) {
    int64_t s = (B + 63) / 64;                                     // >= 64 rows per workgroup, <= 4096 workgroups
    if (s > 4096) s = 4096;                                        // (round 4: 2048 / 1024 / 512 slabs make both kernels slower -- 0.081 -> 0.097 / 0.110 /
    return s < 1 ? 1 : s;                                          //  0.163 ms at 2^18 rows -- and leave the two slab-sum launches where they are)
}

template <typename T>
static int mlp2_small_bwd(const T* x, int64_t xs, const T* W1, int64_t w1s, const T* b1, const T* W2, int64_t w2s, const T* g, int64_t gs, int64_t B, int32_t K1,
                          int32_t H, int32_t N, T* slab, T* slab_b2, void* stream) {
    if (!x || !W1 || !b1 || !W2 || !g || !slab || !slab_b2 || B < 0) return JF_ERR_BADARG;
    if (K1 < 1 || K1 > MS_K1MAX || N < 1 || N > (K1 <= MS_K1WIDE ? (sizeof(T) == 8 ? MS_NWIDE64 : MS_NWIDE) : MS_NMAX) || H < 1 || H > 128) return JF_ERR_UNSUPPORTED;
    if (B == 0) return JF_OK;
    const int64_t S = mlp2_small_slabs(B);
    const int64_t rpb = (B + S - 1) / S;
    const dim3 grid((unsigned)S), block(128);
    hipStream_t st = (hipStream_t)stream;
    // (K1 = 1, 2 get their own instantiations since round 6: the 1 -> 128 -> 8 head of C4 spent 6 of its 35 vector instructions per row and hidden
    //  unit on the three zero-weight slots of a four-wide input loop)
    const int kb = K1 <= 1 ? 1 : K1 <= 2 ? 2 : K1 <= 4 ? 4 : K1 <= 8 ? 8 : K1 <= 16 ? 16 : 32, nb = N <= 4 ? 4 : N <= 8 ? 8 : N <= 12 ? 12 : N <= 16 ? 16 : N <= 32 ? 32 : N <= 48 ? 48 : 64;
    const int dense = (kb == K1 || xs == K1) && (nb == N || gs == N);           // padded slots read rows of the same array, never a gap
#define JF_MS(KB_, NB_) jf::launch((mlp2_small_bwd_kernel<T, KB_, NB_>), grid, block, 0, st, x, xs, W1, w1s, b1, W2, w2s, g, gs, B, (int)K1, (int)H, (int)N, rpb, slab, slab_b2, dense)
#define JF_MS_N(KB_) { if (N <= 4) JF_MS(KB_, 4); else if (N <= 8) JF_MS(KB_, 8); else if (N <= 12) JF_MS(KB_, 12); else JF_MS(KB_, 16); }
    // (17 .. 64 outputs, round 6: the 4 -> 128 -> 46 head of c3b went through the per-layer path -- a library GEMM, tanh', two weight-gradient
    //  launches: 0.45 ms per 2^18 rows)
#define JF_MS_W(KB_) { if (N <= 16) JF_MS_N(KB_) else if (N <= 32) JF_MS(KB_, 32); else if (N <= 48) JF_MS(KB_, 48); else JF_MS(KB_, 64); }
    if (K1 <= 1) JF_MS_W(1) else if (K1 <= 2) JF_MS_W(2) else if (K1 <= 4) JF_MS_W(4) else if (K1 <= 8) JF_MS_W(8) else if (K1 <= 16) JF_MS_N(16) else JF_MS_N(32)
#undef JF_MS_W
#undef JF_MS_N
#undef JF_MS
    return check_launch();
}

template <typename T>
static int mlp_hidden_bwd(const T* x, int64_t xs, const T* W1, int64_t w1s, const T* b1, const T* gh, int64_t ghs, int64_t B, int32_t K1, int32_t H, T* slab,
                          void* stream) {
    if (!x || !W1 || !b1 || !gh || !slab || B < 0) return JF_ERR_BADARG;
    if (K1 < 1 || K1 > MS_K1MAX || H < 1 || H > 128) return JF_ERR_UNSUPPORTED;
    if (B == 0) return JF_OK;
    const int64_t S = mlp2_small_slabs(B);
    const int64_t rpb = (B + S - 1) / S;
    const dim3 grid((unsigned)S), block(128);
    hipStream_t st = (hipStream_t)stream;
    const int kb = K1 <= 4 ? 4 : K1 <= 8 ? 8 : K1 <= 16 ? 16 : 32;
    const int dense = kb == K1 || xs == K1;
#define JF_MH(KB_) jf::launch((mlp_hidden_bwd_kernel<T, KB_>), grid, block, 0, st, x, xs, W1, w1s, b1, gh, ghs, B, (int)K1, (int)H, rpb, slab, dense)
    if (K1 <= 4) JF_MH(4); else if (K1 <= 8) JF_MH(8); else if (K1 <= 16) JF_MH(16); else JF_MH(32);
#undef JF_MH
    return check_launch();
}

template <typename T> static int wgrad_na(int32_t N) { return N > Mfma<T>::MT ? 2 : 1; }

template <typename T> static int64_t wgrad_splits_t(int64_t B, int32_t N) {        // (tiled MFMA kernel)
    const int per_wave = wgrad_na<T>(N) * Mfma<T>::MT;
    const int64_t tiles = (N + per_wave - 1) / per_wave;
    int64_t s = (4096 + tiles - 1) / tiles;                    // ~4 waves per SIMD over the n-tiles x splits grid
    const int64_t max_s = (B + 127) / 128;                      // at least 128 rows per split
    if (s > max_s) s = max_s;
    if (s > 65535) s = 65535;
    return s < 1 ? 1 : s;
}

template <typename T, int NA, int KT>
static void wgrad_go(const T* g, int64_t gs, const T* in, int64_t is, int64_t B, int32_t K, int32_t N, int64_t S, T* pw, T* pb, hipStream_t st) {
    const int64_t rps = (((B + S - 1) / S) + Mfma<T>::KS - 1) / Mfma<T>::KS * Mfma<T>::KS;
    jf::launch((wgrad_kernel<T, NA, KT>), dim3((unsigned)((N + NA * Mfma<T>::MT - 1) / (NA * Mfma<T>::MT)), (unsigned)S), dim3(64), 0, st, g, gs, in,
                       is, B, K, N, rps, pw, pb);
}

template <typename T>
static int wgrad(const T* g, int64_t gs, const T* in, int64_t is, int64_t B, int32_t K, int32_t N, T* pw, T* pb, void* stream) {
    if (!g || !in || !pw || !rows_ok(B) || !width_ok(K) || !width_ok(N)) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    hipStream_t st = (hipStream_t)stream;
    if (wgrad_skinny<T>(g, gs, in, is, B, K, N, pw, pb, st)) return check_launch();
    if (K > 128) return JF_ERR_UNSUPPORTED;
    const int64_t S = wgrad_splits_t<T>(B, N);
    const int kt = (K + Mfma<T>::MT - 1) / Mfma<T>::MT;        // f32: 1..4, f64: 1..8
    const bool two = wgrad_na<T>(N) == 2;
#define JF_WG(KT_)                                                                   \
    { if (two) wgrad_go<T, 2, KT_>(g, gs, in, is, B, K, N, S, pw, pb, st);           \
      else wgrad_go<T, 1, KT_>(g, gs, in, is, B, K, N, S, pw, pb, st); }
    if (kt <= 1) JF_WG(1)
    else if (kt <= 2) JF_WG(2)
    else if (kt <= 4) JF_WG(4)
    else { if constexpr (sizeof(T) == 8) JF_WG(8) else return JF_ERR_UNSUPPORTED; }
#undef JF_WG
    return check_launch();
}

}  // namespace jf

extern "C" {
int64_t jf_linear_wgrad_splits_f32(int64_t B, int32_t K, int32_t N) { if (!jf::width_ok(K) || !jf::width_ok(N) || !jf::rows_ok(B)) return JF_ERR_BADARG; return jf::wgrad_is_skinny(K, N) ? jf::wgrad_skinny_splits(B, K, N) : jf::wgrad_splits_t<float>(B, N); }
int64_t jf_linear_wgrad_splits_f64(int64_t B, int32_t K, int32_t N) { if (!jf::width_ok(K) || !jf::width_ok(N) || !jf::rows_ok(B)) return JF_ERR_BADARG; return jf::wgrad_is_skinny(K, N) ? jf::wgrad_skinny_splits(B, K, N) : jf::wgrad_splits_t<double>(B, N); }
int jf_linear_wgrad_f32(const float* g, int64_t gs, const float* in, int64_t is, int64_t B, int32_t K, int32_t N, float* pw, float* pb, void* s) {
    return jf::wgrad<float>(g, gs, in, is, B, K, N, pw, pb, s);
}
int jf_linear_wgrad_f64(const double* g, int64_t gs, const double* in, int64_t is, int64_t B, int32_t K, int32_t N, double* pw, double* pb, void* s) {
    return jf::wgrad<double>(g, gs, in, is, B, K, N, pw, pb, s);
}
int64_t jf_mlp2_small_bwd_slabs(int64_t B) { return jf::rows_ok(B) ? jf::mlp2_small_slabs(B) : (int64_t)JF_ERR_BADARG; }
int jf_mlp2_small_bwd_f32(const float* x, int64_t xs, const float* W1, int64_t w1s, const float* b1, const float* W2, int64_t w2s, const float* g, int64_t gs,
                          int64_t B, int32_t K1, int32_t H, int32_t N, float* slab, float* slab_b2, void* s) {
    return jf::mlp2_small_bwd<float>(x, xs, W1, w1s, b1, W2, w2s, g, gs, B, K1, H, N, slab, slab_b2, s);
}
int jf_mlp2_small_bwd_f64(const double* x, int64_t xs, const double* W1, int64_t w1s, const double* b1, const double* W2, int64_t w2s, const double* g, int64_t gs,
                          int64_t B, int32_t K1, int32_t H, int32_t N, double* slab, double* slab_b2, void* s) {
    return jf::mlp2_small_bwd<double>(x, xs, W1, w1s, b1, W2, w2s, g, gs, B, K1, H, N, slab, slab_b2, s);
}
int jf_mlp_hidden_bwd_f32(const float* x, int64_t xs, const float* W1, int64_t w1s, const float* b1, const float* gh, int64_t ghs, int64_t B, int32_t K1,
                          int32_t H, float* slab, void* s) {
    return jf::mlp_hidden_bwd<float>(x, xs, W1, w1s, b1, gh, ghs, B, K1, H, slab, s);
}
int jf_mlp_hidden_bwd_f64(const double* x, int64_t xs, const double* W1, int64_t w1s, const double* b1, const double* gh, int64_t ghs, int64_t B, int32_t K1,
                          int32_t H, double* slab, void* s) {
    return jf::mlp_hidden_bwd<double>(x, xs, W1, w1s, b1, gh, ghs, B, K1, H, slab, s);
}
}
