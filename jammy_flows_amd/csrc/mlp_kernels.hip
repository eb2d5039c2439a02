// Dense layers of the parameter-emitting MLPs on the matrix cores:  out = act(in @ W^T + bias).
//
// Replaces torch.nn.Linear (+ tanh) of the default amortisation MLP (jammy_flows/main/default.py:656-670) and the
// U / V^T products of AmortizableMLP with permanent parameters (jammy_flows/amortizable_mlp.py:508-578).
// Shapes on the hot path: M = batch (2^20), K <= 128 (inputs / hidden / rank), N = 8 ... 1224 (parameter block width).
//
// f32: v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD); f64: v_mfma_f64_16x16x4_f64.
//
//   mlp2_kernel    Linear -> tanh -> Linear in one launch, and (without its first layer) every single dense layer with K <= 128:
//                  the operand that is reused for all output tiles lives in registers, W streams through LDS (see below).
//   linear_kernel  generic fall-back (K > 128 or unaligned weight rows): workgroup = 4 waves stacked along M, tile 128 x 64, K walked in
//                  chunks of 32 staged through LDS (rows padded by one element: conflict-free ds_read_b32 fragment reads).
#include "jf_common.h"
#include "jf_math.h"
#include <cstdlib>

#include "jf_mfma.h"

namespace jf {

constexpr int BM = 128, BN = 64, KC = 32, WM = 32, WN = 64, LDP = KC + 1;

// [ROWS x KC] chunk of a row-major matrix -> LDS rows padded to LDP; rows >= max_row and columns >= kc are zero filled
template <typename T, int ROWS>
__device__ __forceinline__ void stage_chunk(T* __restrict__ dst, const T* __restrict__ src, int64_t stride, int64_t row0, int64_t max_row, int k0,
                                            int kc, int tid, bool vec_ok) {
    constexpr int N = Vec16<T>::N;
    using V = typename Vec16<T>::type;
    // loads are unconditional (addresses clamped into the matrix, results zeroed by a select): a branch around a load makes hipcc
    // sink the load into it and wait for each one separately
    const int64_t last = max_row - 1;
    if (vec_ok) {
        constexpr int PER_ROW = KC / N;
        constexpr int CNT = ROWS * PER_ROW / 256;
        V v[CNT];
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
            const int idx = u * 256 + tid;
            const int r = idx / PER_ROW, c = (idx % PER_ROW) * N;
            const int64_t gr = row0 + r;
            const bool ok = (gr <= last) && (c < kc);                                     // kc % N == 0 when vec_ok
            const V t = *reinterpret_cast<const V*>(src + (gr <= last ? gr : last) * stride + k0 + (c < kc ? c : 0));
            v[u].x = ok ? t.x : T(0); v[u].y = ok ? t.y : T(0);
            if constexpr (N == 4) { v[u].z = ok ? t.z : T(0); v[u].w = ok ? t.w : T(0); }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
            const int idx = u * 256 + tid;
            const int r = idx / PER_ROW, c = (idx % PER_ROW) * N;
            T* d = dst + r * LDP + c;
            d[0] = v[u].x; d[1] = v[u].y;
            if constexpr (N == 4) { d[2] = v[u].z; d[3] = v[u].w; }
        }
    } else {
        constexpr int CNT = ROWS * KC / 256;
        T v[CNT];
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
            const int idx = u * 256 + tid;
            const int r = idx / KC, c = idx % KC;
            const int64_t gr = row0 + r;
            const T t = src[(gr <= last ? gr : last) * stride + k0 + (c < kc ? c : 0)];
            v[u] = ((gr <= last) && (c < kc)) ? t : T(0);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
            const int idx = u * 256 + tid;
            dst[(idx / KC) * LDP + (idx % KC)] = v[u];
        }
    }
}

// BNT: output columns per workgroup tile.  BN (64) in general; ONE mfma column tile (16 for float64, 32 for float32) when the whole output is
// that narrow -- g_params (B, 1224) @ U2 (1224, 8) of the low-rank AmortizableMLP's backward spent 8x the float64 matrix work on padding
// columns and was bound by it (0.445 ms per 2^17 rows; the 1.28 GB it reads take 0.25 ms).
template <typename T, int BNT = BN>
__global__ void __launch_bounds__(256) linear_kernel(const T* __restrict__ in, int64_t in_stride, const T* __restrict__ W, int64_t w_stride,
                                                     const T* __restrict__ bias, int64_t B, int K, int N, int act, T* __restrict__ out,
                                                     int64_t out_stride) {
    using MF = Mfma<T>;
    constexpr int MT = MF::MT, KS = MF::KS, NREG = MF::NREG;
    constexpr int TM = WM / MT, TN = BNT / MT;   // mfma tiles per wave
    static_assert(BNT % MT == 0 && BNT >= MT, "whole mfma column tiles");
    __shared__ T As[BM * LDP];
    __shared__ T Ws[BNT * LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = (N + BNT - 1) / BNT;               // 1-D grid, N tile is the fast index: the blocks that share an activation
    const int col0 = (int)(blockIdx.x % n_tiles) * BNT;    // tile are dispatched back to back and find it in L2
    const int64_t row0 = (int64_t)(blockIdx.x / n_tiles) * BM;
    const bool vec_in = (K % Vec16<T>::N == 0) && (in_stride % Vec16<T>::N == 0) && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0);
    const bool vec_w = (K % Vec16<T>::N == 0) && (w_stride % Vec16<T>::N == 0) && ((reinterpret_cast<uintptr_t>(W) & 15u) == 0);

    typename MF::Acc acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < NREG; ++r) acc[i][j][r] = T(0);

    const int li = lane % MT, lk = lane / MT;
    for (int k0 = 0; k0 < K; k0 += KC) {
        const int kc = (K - k0) < KC ? (K - k0) : KC;
        __syncthreads();
        // stage the A chunk [BM x KC] and the W chunk [BN x KC] (zero padded); all global loads of a thread are issued before its LDS writes
        stage_chunk<T, BM>(As, in, in_stride, row0, B, k0, kc, tid, vec_in);
        stage_chunk<T, BNT>(Ws, W, w_stride, (int64_t)col0, (int64_t)N, k0, kc, tid, vec_w);
        __syncthreads();
        const int ksteps = (kc + KS - 1) / KS;
        for (int s = 0; s < ksteps; ++s) {
            const int kk = s * KS + lk;
            T a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(wave * WM + i * MT + li) * LDP + kk];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Ws[(j * MT + li) * LDP + kk];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(a[i], b[j], acc[i][j]);
        }
    }
    // epilogue
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gc = col0 + j * MT + li;
        const T bv = (bias != nullptr && gc < N) ? bias[gc] : T(0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const int64_t gr = row0 + wave * WM + i * MT + MF::row_of(r, lane);
                if (gr < B && gc < N) {
                    T v = acc[i][j][r] + bv;
                    if (act == 1) v = M<T>::tanh(v);
                    out[gr * out_stride + gc] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused two-layer amortisation MLP:  out = (tanh(in @ W1^T + b1)) @ W2^T + b2      (main/default.py:656-670 with one hidden layer)
//
// Workgroup = 4 waves; wave w owns MT rows (f32: 32, f64: 16) of the row tile and keeps their hidden activations IN REGISTERS for the
// whole kernel.  Both products are computed transposed (h^T = W1 x^T, out^T = W2 h^T): the MFMA result layout of the first
// (lane = row, registers = hidden units) is exactly the B-operand layout of the second (lane = row, k-slot = lane group), so h never
// touches LDS, and the second result has lane = row with groups of 4 consecutive output columns per lane: 16-byte result stores.
// The k index a (register, lane group) pair stands for is row_of(v, lane); the W2 fragment is read from LDS at that same k, so the
// contraction pairs up correctly whatever the order.  W2 is streamed through one LDS tile of BN output columns shared by the 4 waves
// (f32: rows padded to 132 floats -> aligned, conflict-free ds_read_b128 of four consecutive k), read one MFMA group ahead.
// No predicates anywhere: rows past B duplicate row B-1 and columns past N duplicate column N-1 (same inputs -> same values ->
// benign duplicate stores), so the instruction stream between two barriers is one straight scheduling region.
// Barriers wait for LDS traffic only; result stores drain behind the next tile's loads and MFMAs.  Requires K1 <= 32, H <= 128.
// ------------------------------------------------------------------------------------------------------------------
constexpr int HMAX = 128, K1MAX = 32;
template <typename T> struct Mlp2Cfg;
template <> struct Mlp2Cfg<float> { static constexpr int LDW = HMAX + 4; };
template <> struct Mlp2Cfg<double> { static constexpr int LDW = HMAX + 1; };

// L1 = false: no first layer -- the "hidden activations" are the input itself (single dense layer out = act(in W2^T + b2) with K = H <= 128
// input columns, jf_linear); act: 0 identity, 1 tanh on the output.
template <typename T, int JH, int TN, bool VECROW, bool L1>
__global__ void __launch_bounds__(256, 3) mlp2_kernel(const T* __restrict__ in, int64_t in_stride, const T* __restrict__ W1, int64_t w1_stride,
                                                      const T* __restrict__ b1, const T* __restrict__ W2, int64_t w2_stride, const T* __restrict__ b2,
                                                      int64_t B, int K1, int H, int N, T* __restrict__ out, int64_t out_stride, int act) {
    using MF = Mfma<T>;
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    constexpr int MT = MF::MT, KS = MF::KS, NREG = MF::NREG;
    constexpr int BN2 = TN * MT, LDW = Mlp2Cfg<T>::LDW;      // TN mfma column tiles per W2 tile (1 for narrow outputs)
    constexpr int BMR = 4 * MT;                              // rows per workgroup
    constexpr int HP = JH * MT;                              // hidden width padded to whole mfma tiles
    constexpr int WPT = BN2 * HMAX / VN / 256;               // 16-byte pieces of a W2 tile per thread
    constexpr int RG = MF::RUN;                              // consecutive result registers = consecutive columns (f32: 4, f64: 1)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* Ws = reinterpret_cast<T*>(smem_raw);                  // [BN2][LDW]   W2 tile (phase 2)
    T* Bs = Ws + BN2 * LDW;                                  // [BN2]        bias of the tile's columns
    const int k1p = L1 ? (K1 + KS - 1) / KS * KS : MT, ldk = k1p + 1;   // (!L1: the input is staged in MT-column chunks)
    T* Xs = Bs + BN2;                                        // [BMR][ldk]   input tile
    T* W1s = Xs + BMR * ldk;                                 // [HP][ldk]    W1   } staged once per workgroup
    T* b1s = W1s + HP * ldk;                                 // [HP]         b1   }
    const double* ttab = reinterpret_cast<const double*>(b1s + HP);   // float64 with a first layer: tanh(k / 32) (jf_math.h: tanh_tab)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane % MT, lq = lane / MT;
    const int64_t last = B - 1;
    const int n_tiles = (N + BN2 - 1) / BN2;
    const int64_t n_row_tiles = (B + BMR - 1) / BMR;
    if constexpr (L1 && sizeof(T) == 8) tanh_tab_load(const_cast<double*>(ttab), tid, 256);

    // ---- W1, b1: once per workgroup.  Staging in straight-line batches of 4 loads per thread (clamped addresses + selects): issued
    // back to back, one round trip per batch.
    if constexpr (L1) {
        const int nw = HP * k1p;
        for (int base = 0; base < nw; base += 4 * 256) {
            T v[4]; int o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256 + tid;
                const int r = idx / k1p, c = idx - r * k1p;
                const T t = W1[(int64_t)(r < H ? r : H - 1) * w1_stride + (c < K1 ? c : 0)];
                v[u] = (r < H && c < K1) ? t : T(0);
                o[u] = idx < nw ? r * ldk + c : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (o[u] >= 0) W1s[o[u]] = v[u];
        }
        if (tid < HP) b1s[tid] = tid < H ? b1[tid < H ? tid : 0] : T(0);
    }
    // ---- phase 2: out^T = W2 h^T + b2, W2 streamed in BN2-column tiles
    // per-thread constants of the tile copy: piece u of thread tid is row u*RPP + r_t, 16-byte column c_t of the tile, so the global address is
    // (uniform tile/pass base) + (one 32-bit lane offset) and the LDS address a constant -- no address arithmetic per piece
    constexpr int PPR = HMAX / VN, RPP = 256 / PPR;
    const int r_t = tid / PPR, c_t = (tid % PPR) * VN;
    const unsigned voff = (unsigned)((r_t * w2_stride + c_t) * (int64_t)sizeof(T));
    T* const lbase = Ws + r_t * LDW + c_t;
    const bool h_full = H == HMAX;                           // block-uniform
    auto load_tile = [&](int t) {                            // straight-line, loads issued back to back
        V wreg[WPT];
        const int bc = t * BN2 + (tid < BN2 ? tid : 0);
        const T bval = (b2 != nullptr) ? b2[bc < N ? bc : N - 1] : T(0);
        if (h_full && (t + 1) * BN2 <= N) {                  // block-uniform fast path: whole tile inside W2, no clamps, no selects
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                const char* base = reinterpret_cast<const char*>(W2 + (int64_t)(t * BN2 + u * RPP) * w2_stride);    // uniform
                wreg[u] = *reinterpret_cast<const V*>(base + voff);
            }
        } else {                                             // clamped addresses + selects (columns past N replicate column N-1)
#pragma unroll
            for (int u = 0; u < WPT; ++u) {
                const int gc = t * BN2 + u * RPP + r_t;
                const bool ok = c_t < H;
                const V v = *reinterpret_cast<const V*>(W2 + (int64_t)(gc < N ? gc : N - 1) * w2_stride + (ok ? c_t : 0));
                wreg[u].x = ok ? v.x : T(0); wreg[u].y = ok ? v.y : T(0);
                if constexpr (VN == 4) { wreg[u].z = ok ? v.z : T(0); wreg[u].w = ok ? v.w : T(0); }
            }
        }
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            T* d = lbase + u * RPP * LDW;
            if constexpr (VN == 4) { *reinterpret_cast<V*>(d) = wreg[u]; }
            else { d[0] = wreg[u].x; d[1] = wreg[u].y; }
        }
        if (tid < BN2) Bs[tid] = bval;
    };
    const T* wb = Ws + li * LDW;
    constexpr int NGRP = VN == 4 ? NREG / 4 : NREG;          // W2 fragment reads per hidden tile (f32: one b128 = 4 k, f64: one b64 = 1 k)
    constexpr int KPG = NREG / NGRP;                         // MFMAs (k values) per read
    using F = typename std::conditional<VN == 4, V, T>::type;
    auto read_frag = [&](F (&f)[TN], int j, int g) {
#pragma unroll
        for (int ct = 0; ct < TN; ++ct) f[ct] = *reinterpret_cast<const F*>(wb + ct * MT * LDW + j * MT + MF::row_of(g * KPG, lane));
    };
    auto elem = [](const F& f, int e) -> T {
        if constexpr (VN == 4) return e == 0 ? f.x : e == 1 ? f.y : e == 2 ? f.z : f.w;
        else return f;
    };

    // ---- row tiles: one per workgroup, or a grid-stride walk when the output is a single W2 tile wide (narrow outputs: the launch is
    // then dominated by per-workgroup fixed latency, so a resident set of workgroups keeps W1 / b1 / W2 in LDS and only streams x)
    __syncthreads();                                         // W1s / b1s staged
    load_tile(0);
    for (int64_t rt = blockIdx.x; rt < n_row_tiles; rt += gridDim.x) {
        const int64_t row0 = rt * BMR;
        T hreg[JH][NREG];
        if constexpr (L1) {
            {
                const int nx = BMR * k1p;
                for (int base = 0; base < nx; base += 4 * 256) {
                    T v[4]; int o[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = base + u * 256 + tid;
                        const int r = idx / k1p, c = idx - r * k1p;
                        const int64_t gr = row0 + r;
                        const T t = in[(gr <= last ? gr : last) * in_stride + (c < K1 ? c : 0)];      // rows past B replicate row B-1
                        v[u] = c < K1 ? t : T(0);
                        o[u] = idx < nx ? r * ldk + c : -1;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (o[u] >= 0) Xs[o[u]] = v[u];
                }
            }
            __syncthreads();                                 // x tile (and, first pass, the W2 tile) visible
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
                for (int r = 0; r < NREG; ++r) hreg[j][r] = tanh_hidden<T>(ttab, acc[j][r] + b1s[j * MT + MF::row_of(r, lane)]);   // padded units: tanh(0) = 0
        } else {
            // the input itself in the B-operand layout (lane = row, register r of tile j <-> column j*MT + row_of(r, lane)), staged MT columns at a time
#pragma unroll
            for (int j = 0; j < JH; ++j) {
                if (j > 0) lds_barrier();                    // previous chunk consumed
                constexpr int PER = BMR * MT / 256;          // elements per thread and chunk
                T v[PER];
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const int idx = u * 256 + tid;
                    const int r = idx / MT, c = j * MT + idx % MT;
                    const int64_t gr = row0 + r;
                    const T t = in[(gr <= last ? gr : last) * in_stride + (c < K1 ? c : 0)];
                    v[u] = c < K1 ? t : T(0);
                }
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const int idx = u * 256 + tid;
                    Xs[(idx / MT) * ldk + idx % MT] = v[u];
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < NREG; ++r) hreg[j][r] = Xs[(wave * MT + li) * ldk + MF::row_of(r, lane)];
            }
        }
        // result addressing: lane = row (clamped to B-1); per (tile, column tile, register group) a uniform column offset
        const int64_t grow = row0 + wave * MT + li;
        T* const orow = out + (grow <= last ? grow : last) * out_stride;
        for (int t = 0; t < n_tiles; ++t) {
            typename MF::Acc acc[TN];
    #pragma unroll
            for (int ct = 0; ct < TN; ++ct)
    #pragma unroll
                for (int r = 0; r < NREG; ++r) acc[ct][r] = T(0);
            F frag[2][TN];
            read_frag(frag[0], 0, 0);
    #pragma unroll
            for (int j = 0; j < JH; ++j)
    #pragma unroll
                for (int g = 0; g < NGRP; ++g) {
                    const int cur = (j * NGRP + g) & 1;
                    if (j * NGRP + g + 1 < JH * NGRP) read_frag(frag[cur ^ 1], (j * NGRP + g + 1) / NGRP, (j * NGRP + g + 1) % NGRP);   // one group ahead
    #pragma unroll
                    for (int e = 0; e < KPG; ++e)
    #pragma unroll
                        for (int ct = 0; ct < TN; ++ct) {
                            if constexpr (RG == 4) acc[ct] = MF::mma(elem(frag[cur][ct], e), hreg[j][g * KPG + e], acc[ct]);   // out^T tile: lane = row
                            else acc[ct] = MF::mma(hreg[j][g * KPG + e], elem(frag[cur][ct], e), acc[ct]);                    // out tile: lane = column
                        }
                }
            if constexpr (RG == 1) {
                // float64: the 16x16x4 result registers hold rows q + 4v, so the product is taken un-transposed (hreg is equally valid as the A
                // operand) and each store instruction writes 4 rows x 16 consecutive columns (128-byte segments)
#pragma unroll
                for (int ct = 0; ct < TN; ++ct) {
                    const int lc = ct * MT + li;
                    const int gc = t * BN2 + lc;
                    const T bb = Bs[lc];
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        const int64_t gr = row0 + wave * MT + MF::row_of(r, lane);
                        T v = acc[ct][r] + bb;
                        if (act == 1) v = M<T>::tanh(v);
                        out[(gr <= last ? gr : last) * out_stride + (gc < N ? gc : N - 1)] = v;      // duplicates carry identical values
                    }
                }
            } else {
            // results: acc[ct][r] = out[row = lane's row][col = t*BN2 + ct*MT + row_of(r, lane)]; bias from the LDS tile
    #pragma unroll
                for (int ct = 0; ct < TN; ++ct)
    #pragma unroll
                    for (int r = 0; r < NREG; ++r) acc[ct][r] += Bs[ct * MT + MF::row_of(r, lane)];
                if (act == 1) {                                      // block-uniform
    #pragma unroll
                    for (int ct = 0; ct < TN; ++ct)
    #pragma unroll
                        for (int r = 0; r < NREG; ++r) acc[ct][r] = M<T>::tanh(acc[ct][r]);
                }
                const bool edge = (t + 1) * BN2 > N;                 // block-uniform: only the last tile can reach past N
                if (VECROW && RG == 4 && !edge) {
        #pragma unroll
                    for (int ct = 0; ct < TN; ++ct)
        #pragma unroll
                        for (int r0 = 0; r0 < NREG; r0 += RG) {
                            const int lc = ct * MT + MF::row_of(r0, lane);
                            V o;
                            o.x = acc[ct][r0]; o.y = acc[ct][r0 + 1];
                            if constexpr (VN == 4) { o.z = acc[ct][r0 + 2]; o.w = acc[ct][r0 + 3]; }
                            *reinterpret_cast<V*>(orow + t * BN2 + lc) = o;
                        }
                } else {
                    // edge tile: 16-byte stores for the register groups that lie inside N, scalar stores (last column duplicated) for the group that
                    // straddles N, nothing for groups past N -- a narrow output (N = 10) would otherwise issue 16 scattered dword stores per lane
    #pragma unroll
                    for (int ct = 0; ct < TN; ++ct)
    #pragma unroll
                        for (int r0 = 0; r0 < NREG; r0 += RG) {
                            const int lc = ct * MT + MF::row_of(r0, lane);
                            const int gc = t * BN2 + lc;
                            if (VECROW && RG == 4 && gc + RG <= N) {
                                V o;
                                o.x = acc[ct][r0]; o.y = acc[ct][r0 + 1];
                                if constexpr (VN == 4) { o.z = acc[ct][r0 + 2]; o.w = acc[ct][r0 + 3]; }
                                *reinterpret_cast<V*>(orow + gc) = o;
                            } else if (gc < N) {
    #pragma unroll
                                for (int e = 0; e < RG; ++e)
                                    if (gc + e < N) orow[gc + e] = acc[ct][r0 + e];
                            }
                        }
                }
            }
            if (n_tiles > 1 && (t + 1 < n_tiles || rt + gridDim.x < n_row_tiles)) {   // block-uniform; a single tile stays resident
                lds_barrier();                               // every wave has read the tile
                load_tile(t + 1 < n_tiles ? t + 1 : 0);      // (after the last one: tile 0 for the workgroup's next row tile)
                lds_barrier();
            }
        }

        lds_barrier();                                       // every wave is done with Xs before the next row tile overwrites it
    }
}

template <typename T, int JH, int TN, bool L1>
static int mlp2_launch(const T* in, int64_t in_stride, const T* W1, int64_t w1_stride, const T* b1, const T* W2, int64_t w2_stride, const T* b2, int64_t B,
                       int32_t K1, int32_t H, int32_t N, T* out, int64_t out_stride, int act, void* stream) {
    constexpr int MT = Mfma<T>::MT, KS = Mfma<T>::KS, BMR = 4 * MT, HP = JH * MT;
    const int k1p = L1 ? (K1 + KS - 1) / KS * KS : MT, ldk = k1p + 1;
    const size_t lds = ((size_t)BMR * ldk + (L1 ? (size_t)HP * ldk + HP : 0) + (size_t)(TN * MT) * Mlp2Cfg<T>::LDW + TN * MT) * sizeof(T) +
                       ((L1 && sizeof(T) == 8) ? (size_t)JF_TANH_TAB_N * sizeof(double) : 0);
    // 16-byte result stores need 16-byte aligned rows
    const bool vecrow = (out_stride % Vec16<T>::N == 0) && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0);
    auto k = vecrow ? mlp2_kernel<T, JH, TN, true, L1> : mlp2_kernel<T, JH, TN, false, L1>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int64_t grid = (B + BMR - 1) / BMR;
    if (N <= TN * MT) {                                      // single W2 tile: resident workgroups walk the row tiles
        int dev = 0, cus = 256, per_cu = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 256, lds) != hipSuccess || per_cu < 1) per_cu = 2;
        const int64_t resident = (int64_t)cus * per_cu;
        if (grid > resident) grid = resident;
    }
    jf::launch(k, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, in, in_stride, W1, w1_stride, b1, W2, w2_stride, b2, B,
                       (int)K1, (int)H, (int)N, out, out_stride, act);
    return check_launch();
}

// hidden width (L1) / input width (!L1) in mfma tiles, rounded up to an instantiated count; N <= MT takes the single-tile variant
template <typename T, bool L1>
static int mlp2_dispatch(const T* in, int64_t in_stride, const T* W1, int64_t w1_stride, const T* b1, const T* W2, int64_t w2_stride, const T* b2, int64_t B,
                         int32_t K1, int32_t H, int32_t N, T* out, int64_t out_stride, int act, void* stream) {
    constexpr int MT = Mfma<T>::MT;
    const int tiles = (H + MT - 1) / MT;
    constexpr int Q = HMAX / MT / 4;                         // f32: 1, f64: 2
#define JF_MLP2_GO(JH_) \
    return (N <= MT) ? mlp2_launch<T, JH_, 1, L1>(in, in_stride, W1, w1_stride, b1, W2, w2_stride, b2, B, K1, H, N, out, out_stride, act, stream) \
                     : mlp2_launch<T, JH_, 2, L1>(in, in_stride, W1, w1_stride, b1, W2, w2_stride, b2, B, K1, H, N, out, out_stride, act, stream)
    if (tiles <= 1 * Q) JF_MLP2_GO(1 * Q);
    if (tiles <= 2 * Q) JF_MLP2_GO(2 * Q);
    JF_MLP2_GO(4 * Q);
#undef JF_MLP2_GO
}

// narrow float32 MLPs (<= 4 inputs, <= 16 outputs): mlp_narrow_kernels.hip
int mlp2_narrow_f32(const float* in, int64_t is, const float* W1, int64_t w1s, const float* b1, const float* W2, int64_t w2s, const float* b2, int64_t B,
                    int32_t K1, int32_t H, int32_t N, float* out, int64_t os, void* stream);

template <typename T>
static int mlp2(const T* in, int64_t in_stride, const T* W1, int64_t w1_stride, const T* b1, const T* W2, int64_t w2_stride, const T* b2, int64_t B,
                int32_t K1, int32_t H, int32_t N, T* out, int64_t out_stride, void* stream) {
    if (!in || !W1 || !b1 || !W2 || !out || !width_ok(K1) || !width_ok(N) || !width_ok(H) || !rows_ok(B)) return JF_ERR_BADARG;
    if (K1 > K1MAX || H > HMAX) return JF_ERR_UNSUPPORTED;
    if constexpr (sizeof(T) == 4) {
        static const bool narrow_off = getenv("JF_MLP2_NARROW_OFF") != nullptr;
        if (!narrow_off) {
            const int rc = mlp2_narrow_f32(in, in_stride, W1, w1_stride, b1, W2, w2_stride, b2, B, K1, H, N, out, out_stride, stream);
            if (rc != JF_ERR_UNSUPPORTED) return rc;
        }
    }
    if ((H % Vec16<T>::N) || (w2_stride % Vec16<T>::N) || (reinterpret_cast<uintptr_t>(W2) & 15u)) return JF_ERR_UNSUPPORTED;   // 16-byte W2 rows
    if (B == 0) return JF_OK;
    return mlp2_dispatch<T, true>(in, in_stride, W1, w1_stride, b1, W2, w2_stride, b2, B, K1, H, N, out, out_stride, 0, stream);
}

// ---------------------------------------------------------------------------------------------------------- skinny products
// The low-rank stages of AmortizableMLP (amortizable_mlp.py:508-578) and their backward are products with ONE tiny dimension (rank 8):
// (B x 8) (8 x 1224), (B x 1224) (1224 x 8), ...  They are pure HBM streams -- 8 FMAs per written / read element -- and a 16 x 16 MFMA
// tiling wastes most of every tile on them (jf_linear_f64[K8_N1224] 0.51 ms, [K1224_N8] 0.44 ms per 2^17 rows; 0.17 ms of traffic each).
// skinny_k: K <= 16.  thread = output column n (W[n][:] in registers, coalesced stores), a workgroup walks SK_ROWS rows whose K inputs are
//           wave-uniform (scalar loads).  K8_N1224 0.51 -> 0.33 ms, K8_N128 0.078 -> 0.040 ms per 2^17 rows (float64).
// (The mirrored shape, N <= 16 with a long K, stays on the tiled MFMA kernel below: a lane-per-column streaming version with a wave
//  reduce-scatter measured slower, 0.54 .. 0.74 vs 0.44 ms -- 32 .. 64 float64 accumulators per lane leave one wave per SIMD.)
constexpr int SK_ROWS = 32, SK_KMAX = 16;
template <typename T, int K>
__global__ void __launch_bounds__(256) skinny_k_kernel(const T* __restrict__ in, int64_t is, const T* __restrict__ W, int64_t ws, const T* __restrict__ bias,
                                                       int64_t B, int N, int act, T* __restrict__ out, int64_t os, int cb_shift, int rows) {
    // CB = 2^cb_shift (64, 128 or 256) columns per workgroup; the 256 / CB wave groups take interleaved rows
    const int CB = 1 << cb_shift;
    const int n = blockIdx.x * CB + (threadIdx.x & (CB - 1));
    const int rg = __builtin_amdgcn_readfirstlane(threadIdx.x >> cb_shift), nrg = 256 >> cb_shift;
    const bool live = n < N;
    T w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) w[k] = live ? W[(int64_t)n * ws + k] : T(0);
    const T b = (live && bias) ? bias[n] : T(0);
    const int64_t r0 = (int64_t)blockIdx.y * rows;
    const int64_t r1 = r0 + rows < B ? r0 + rows : B;
    for (int64_t r = r0 + rg; r < r1; r += nrg) {
        const T* x = in + r * is;                                // wave-uniform address: scalar loads
        T acc = b;
#pragma unroll
        for (int k = 0; k < K; ++k) acc += x[k] * w[k];
        if (act) acc = M<T>::tanh_fast(acc);
        if (live) out[r * os + n] = acc;
    }
}

template <typename T> static bool skinny_linear(const T* in, int64_t is, const T* W, int64_t ws, const T* bias, int64_t B, int32_t K, int32_t N, int32_t act,
                                                T* out, int64_t os, hipStream_t st) {
    if (K <= SK_KMAX && N >= 32) {
        const int cb_shift = N <= 64 ? 6 : N <= 128 ? 7 : 8;
        int64_t rows = SK_ROWS;                                    // rows per workgroup: 32, more when the grid's y extent (65535) would overflow
        if ((B + rows - 1) / rows > 65535) rows = (B + 65534) / 65535;
        if (rows > 0x7fffffff) return false;
        const dim3 grid((unsigned)((N + (1 << cb_shift) - 1) >> cb_shift), (unsigned)((B + rows - 1) / rows));
#define JF_SK(K_) case K_: jf::launch((skinny_k_kernel<T, K_>), grid, dim3(256), 0, st, in, is, W, ws, bias, B, (int)N, (int)act, out, os, cb_shift, (int)rows); return true;
        switch (K) { JF_SK(1) JF_SK(2) JF_SK(3) JF_SK(4) JF_SK(5) JF_SK(6) JF_SK(7) JF_SK(8) JF_SK(9) JF_SK(10) JF_SK(11) JF_SK(12) JF_SK(13) JF_SK(14) JF_SK(15)
                     JF_SK(16) default: return false; }
#undef JF_SK
    }
    return false;
}

template <typename T>
static int linear(const T* in, int64_t in_stride, const T* W, int64_t w_stride, const T* bias, int64_t B, int32_t K, int32_t N, int32_t act, T* out,
                  int64_t out_stride, void* stream) {
    if (!in || !W || !out || !width_ok(K) || !width_ok(N) || !rows_ok(B) || (act != 0 && act != 1)) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    if (skinny_linear<T>(in, in_stride, W, w_stride, bias, B, K, N, act, out, out_stride, (hipStream_t)stream)) return check_launch();
    // K <= 128 with 16-byte aligned weight rows: the mlp2 machinery without a first layer (input held in registers as the B operand,
    // transposed product, 16-byte predicate-free stores, resident workgroups for narrow outputs)
    if (K <= HMAX && (K % Vec16<T>::N == 0) && (w_stride % Vec16<T>::N == 0) && ((reinterpret_cast<uintptr_t>(W) & 15u) == 0))
        return mlp2_dispatch<T, false>(in, in_stride, nullptr, 0, nullptr, W, w_stride, bias, B, K, K, N, out, out_stride, act, stream);
    constexpr int NARROW = Mfma<T>::MT;                          // one mfma column tile
    const int bn = N <= NARROW ? NARROW : BN;
    const int64_t blocks = ((B + BM - 1) / BM) * ((N + bn - 1) / bn);
    if (blocks > 0x7fffffffLL) return JF_ERR_UNSUPPORTED;
    dim3 grid((unsigned)blocks);
    if (N <= NARROW)
        jf::launch((linear_kernel<T, NARROW>), grid, dim3(256), 0, (hipStream_t)stream, in, in_stride, W, w_stride, bias, B, (int)K, (int)N,
                           (int)act, out, out_stride);
    else
        jf::launch((linear_kernel<T, BN>), grid, dim3(256), 0, (hipStream_t)stream, in, in_stride, W, w_stride, bias, B, (int)K, (int)N, (int)act,
                           out, out_stride);
    return check_launch();
}

}  // namespace jf

extern "C" {
int jf_linear_f32(const float* in, int64_t is, const float* W, int64_t ws, const float* b, int64_t B, int32_t K, int32_t N, int32_t act, float* out,
                  int64_t os, void* s) {
    return jf::linear<float>(in, is, W, ws, b, B, K, N, act, out, os, s);
}
int jf_linear_f64(const double* in, int64_t is, const double* W, int64_t ws, const double* b, int64_t B, int32_t K, int32_t N, int32_t act, double* out,
                  int64_t os, void* s) {
    return jf::linear<double>(in, is, W, ws, b, B, K, N, act, out, os, s);
}
int jf_mlp2_f32(const float* in, int64_t is, const float* W1, int64_t w1s, const float* b1, const float* W2, int64_t w2s, const float* b2, int64_t B,
                int32_t K1, int32_t H, int32_t N, float* out, int64_t os, void* s) {
    return jf::mlp2<float>(in, is, W1, w1s, b1, W2, w2s, b2, B, K1, H, N, out, os, s);
}
int jf_mlp2_f64(const double* in, int64_t is, const double* W1, int64_t w1s, const double* b1, const double* W2, int64_t w2s, const double* b2, int64_t B,
                int32_t K1, int32_t H, int32_t N, double* out, int64_t os, void* s) {
    return jf::mlp2<double>(in, is, W1, w1s, b1, W2, w2s, b2, B, K1, H, N, out, os, s);
}
}
