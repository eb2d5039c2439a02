// The default amortisation MLP (Linear -> tanh -> Linear, main/default.py:656-670) in FLOAT64 with the wide second product on the INT8 matrix cores.
//
//   jf_mlp2_i8_pack_f64   W2 / b2 -> the digit image the kernel streams (once per weight version)
//   jf_mlp2_i8_f64        out = tanh(in @ W1^T + b1) @ W2^T + b2, same contract as jf_mlp2_f64
//
// Why.  The float64 matrix cores of MI355X run at the float64 VECTOR rate (78.6 TFLOP/s): the 128 -> 548 product of the C3 parameter block costs
// 1.9 ms per 2^20 rows at that peak (jf_mlp2_f64: 2.94 ms measured) -- 60 % of the float64 log-prob step.  The int8 matrix cores are 50x faster
// (v_mfma_i32_16x16x64_i8: 32768 multiply-adds in 16 cycles) and accumulate EXACTLY in int32, so the float64 product is evaluated as an
// error-free sum of integer products (the "Ozaki scheme"):
//     h_k  = sum_i 2^-(6 + 7 i) a_ik,            a_ik in [-64, 64]    (h = tanh(..) in [-1, 1]: no scaling needed)
//     w_nk = 2^e_n sum_j 2^-(6 + 7 j) b_njk,     b_njk in [-64, 64]   (e_n: the row's own exponent, |w_nk| 2^-e_n < 1)
//     sum_k w_nk h_k = 2^(e_n - 12) sum_d 2^(-7 d) [ sum_{i + j = d} sum_k a_ik b_njk ]
// Every bracket is an int32 (<= 6 * 128 * 2^12 < 2^22): ALL slice pairs of one level d share one accumulator, so S slices cost S (S + 1) / 2
// products (21 for S = 6) but only S accumulators per 16 x 16 tile.  Dropped: the digits of h and w beyond 2^-(6 + 7 (S - 1)) (S = 6: 2^-41) and
// the levels d >= S -- about 2e-12 |w|_max per output in the typical case (random digit signs), 2e-10 in the worst; nothing is rounded in
// between (the int32 sums are exact, the float64 Horner recombination rounds at 2^-53).  Measured on the C3 fixture: log p agrees with the
// float64-MFMA path to 3e-10 (S = 6) / 3e-8 (S = 5); the float64 parity bar is 1e-7 relative, the north-star bar 1e-4.
// The first layer (K1 <= 28 inputs, 1 % of the work) stays on exact f64 MFMA; its result layout (unit 16 j + q + 4 r in register r of tile j of
// lane group q) fixes which hidden unit each operand byte of the int8 products carries (ci_unit) -- A and B use the same rule, so the
// hardware's own k order inside a lane never matters.
//
// A fused variant (this product + the flow of cond_split_kernels.hip in float64 on the MFMA result registers) was built first and dropped: with
// 72 VGPRs of float64 parameters + 48 of digits only two waves fit a SIMD, the float64 flow arithmetic then runs at 31 % VALU utilisation, and
// the block took 4.7 ms per 2^20 rows against 1.2 (this kernel) + 1.2 ms (jf_gf_chain_inv_f64 on the materialised block), DESIGN.md 3.12.
//
// Work distribution: 512-thread workgroups, a wave owns 16 rows (128 rows per workgroup); W2's digit image is streamed in chunks of three
// 16-column tiles by LDS-DMA into a double buffer shared by the 8 waves; the result tile (16 rows x 16 columns) of a wave goes straight from
// the accumulators to HBM, whole 128-byte lines per store instruction.  <= 128 VGPRs: four waves per SIMD.
//
// Supported: float64, H <= 128, K1 <= 28.
#include "jf_cond_in.h"
#include "jf_cond_regs.h"
#include "jf_mfma.h"

namespace jf {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
constexpr int CI_CT = 3;                                           // 16-column tiles per chunk
constexpr int CI_FRAG = 1024;                                      // bytes of one MFMA operand fragment (64 lanes x 16 int8)
constexpr int CI_HMAX = 128, CI_K1MAX = 28;                        // hidden units (two k-steps of 64), inputs
constexpr int CI_WAVES = 8, CI_THREADS = 64 * CI_WAVES, CI_ROWS = 16 * CI_WAVES;
constexpr int CI_TAIL = CI_CT * 16 * 2 * 8;                        // per chunk: 48 scales 2^(e_n - 19) + 48 biases, float64
__host__ __device__ constexpr int ci_w_bytes(int S) { return CI_CT * 2 * S * CI_FRAG; }         // 3 tiles x 2 k-steps x S slices x 1 KiB
__host__ __device__ constexpr int ci_chunk_bytes(int S) { return ci_w_bytes(S) + CI_TAIL; }
__host__ __device__ constexpr int ci_chunks(int N) { return ((N + 15) / 16 + CI_CT - 1) / CI_CT; }
static bool ci_slices_ok(int S) { return S == 5 || S == 6; }

// v in [-1, 1] -> S balanced base-128 digits: v = sum_i 2^-(6 + 7 i) d[i] + r, |r| <= 2^-(7 + 7 (S - 1)); every step is exact in float64
template <int S> __device__ __forceinline__ void ci_digits(double v, int (&d)[S]) {
    double r = v * 64.0;
#pragma unroll
    for (int i = 0; i < S; ++i) {
        const double q = __builtin_rint(r);
        d[i] = (int)q;
        r = (r - q) * 128.0;
    }
}
__device__ __forceinline__ int ci_pack4(int a, int b, int c, int d) {
    return (a & 0xff) | ((b & 0xff) << 8) | ((c & 0xff) << 16) | (d << 24);
}
// hidden unit of byte i of lane group q in k-step s (bytes 4 g + r of k-step s = register r of phase-1 tile 4 s + g)
__host__ __device__ constexpr int ci_unit(int s, int i, int q) { return 16 * (4 * s + (i >> 2)) + q + 4 * (i & 3); }

// ---------------------------------------------------------------------------------------------------------- packing
struct CiPackArgs {
    const double* W2; int64_t w2s; const double* b2;
    int H, N;
    unsigned char* out;
};

// one thread per (chunk, tile, lane): the row's exponent, then the 2 x S fragments (16 bytes each) of its 32 hidden units; lanes 0..15 also write
// the row's scale and bias.  fragment = the B operand of one MFMA (W2^T: matrix column = output column): lane (m, q) carries output column 16 tile + m, hidden units ci_unit(s, 0..15, q).
template <int S> __global__ void __launch_bounds__(256) ci_pack_kernel(const CiPackArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = idx & 63;
    const int tc = (idx >> 6) % CI_CT, chunk = (idx >> 6) / CI_CT;
    if (chunk >= ci_chunks(a.N)) return;
    const int m = lane & 15, q = lane >> 4;
    const int col = 16 * (chunk * CI_CT + tc) + m;
    const double* wrow = col < a.N ? a.W2 + (int64_t)col * a.w2s : nullptr;
    double wmax = 0.0;
    if (wrow)
        for (int k = 0; k < a.H; ++k) wmax = fmax(wmax, fabs(wrow[k]));
    // |w| 2^-e < 1 strictly; a zero, non-finite or absent row gets e = 0 (non-finite weights then give garbage digits: the exact paths report
    // such weights, this one is only chosen for finite ones -- see jf_mlp2_i8_pack_f64)
    int e = (wmax > 0.0 && wmax < INFINITY) ? ilogb(wmax) + 1 : 0;
    e = e < -1000 ? -1000 : e;                                     // (e <= 1024 for every finite double; below -1000 the row is zero to 2^-1000 anyway, and 2^(e - 19) stays normal)
    unsigned char* base = a.out + (size_t)chunk * ci_chunk_bytes(S);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        i32x4 f[S];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int d[4][S];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = ci_unit(s, 4 * g + r, q);
                const double w = (wrow && k < a.H) ? wrow[k] : 0.0;
                ci_digits<S>(ldexp(w, -e), d[r]);
            }
#pragma unroll
            for (int j = 0; j < S; ++j) f[j][g] = ci_pack4(d[0][j], d[1][j], d[2][j], d[3][j]);
        }
#pragma unroll
        for (int j = 0; j < S; ++j) *reinterpret_cast<i32x4*>(base + (size_t)((tc * 2 + s) * S + j) * CI_FRAG + lane * 16) = f[j];
    }
    if (lane < 16) {
        double* tail = reinterpret_cast<double*>(base + ci_w_bytes(S));
        tail[tc * 16 + m] = ldexp(1.0, e - 19);                    // 2^(e - 12) for the product, 2^-7: the kernel sums level PAIRS 128 acc_d + acc_{d+1}
        tail[CI_CT * 16 + tc * 16 + m] = (wrow && a.b2 != nullptr) ? a.b2[col] : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------------------- the kernel
struct CiArgs {
    const double* in; int64_t in_stride;
    const double* W1; int64_t w1s; const double* b1;
    const unsigned char* packed;
    int K1, H, N;
    int64_t B;
    double* out; int64_t os;
    CondIn cin;                                                    // n > 0: the input rows are these segments (jf_cond_in.h)
};

// one chunk of the image -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: 1 KiB per wave instruction, no register hop): wave w of the 8 moves
// KiB pieces w, w + 8, ...; the tail goes with wave 0.  A plain template function, not a lambda inside the kernel (see cs_dma_chunk, jf_cond_split.h).
template <int S>
__device__ __forceinline__ void ci_dma_chunk(__amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int g, int wave, int lane) {
    constexpr int PIECES = ci_w_bytes(S) / 1024;
#pragma unroll
    for (int u = 0; u < (PIECES + CI_WAVES - 1) / CI_WAVES; ++u)
        if (u * CI_WAVES + wave < PIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (cs_lptr)(dst + (u * CI_WAVES + wave) * 1024), 16, wave * 1024 + lane * 16, g + u * CI_WAVES * 1024, 0, 0);
    if (wave == 0 && lane < CI_TAIL / 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (cs_lptr)(dst + ci_w_bytes(S)), 16, lane * 16, g + ci_w_bytes(S), 0, 0);
}

// phase 1: h^T = tanh(W1 x^T + b1) for the wave's 16 rows on exact f64 MFMA (rows past B replicate row B-1), returned as the B operands of
// the int8 products: hd[s][i] = digit slice i of the 16 hidden units ci_unit(s, 0..15, lq).  Xs: LDS scratch of (CI_ROWS + CI_HMAX) (k1p + 1) +
// CI_HMAX doubles.  Ends past the barrier that follows the staging, not past one after the MFMA reads (the caller's next barrier covers those).
// Returns (wave-uniform) the mask of the wave's 16 rows with a NaN hidden activation: NaN cannot be cut into digits ((int)rint(NaN) = 0 would
// turn the row into out = b2), so the caller stores NaN for those rows -- what jf_mlp2_f64 and the reference's nn.Linear propagate.
template <int S>
__device__ __forceinline__ unsigned ci_hidden(const CiArgs& a, int64_t row0, int64_t last, double* Xs, i32x4 (&hd)[2][S]) {
    using MF = Mfma16<double>;
    constexpr int MT = 16, KS = 4, JH = CI_HMAX / MT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int K1 = a.K1, H = a.H;
    const int k1p = (K1 + KS - 1) / KS * KS, ldk = k1p + 1;
    double* W1s = Xs + CI_ROWS * ldk;
    double* b1s = W1s + CI_HMAX * ldk;
    // (the table form of the hidden layer's tanh, jf_math.h: tanh_tab, LOSES here -- 1.50 -> 1.68 ms per 2^20 rows, scripts/probe/tanh_ab.sh: the
    //  32 dependent LDS lookups per lane wait behind the fragment traffic of the int8 products; the narrow-output jf_mlp2_f64 gains, 0.30 -> 0.25)
    {
        const int nx = CI_ROWS * k1p, nw = CI_HMAX * k1p;
        if (a.cin.n) {
            const bool any_embed = cond_in_any_embed(a.cin);
            for (int base = 0; base < nx; base += 4 * CI_THREADS) {
                CondLoc<double> loc[4]; double va[4], vb[4]; int o[4]; bool keep[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + u * CI_THREADS + tid;
                    const int r = idx / k1p, c = idx - r * k1p;
                    const int64_t gr = row0 + r;
                    loc[u] = cond_in_locate<double>(a.cin, gr <= last ? gr : last, c < K1 ? c : 0);
                    keep[u] = c < K1;
                    o[u] = idx < nx ? r * ldk + c : -1;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { va[u] = *loc[u].pa; vb[u] = *loc[u].pb; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double t = cond_in_finish<double, false>(loc[u], va[u], vb[u], any_embed);
                    if (o[u] >= 0) Xs[o[u]] = keep[u] ? t : 0.0;
                }
            }
        } else
        for (int base = 0; base < nx; base += 4 * CI_THREADS) {
            double v[4]; int o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * CI_THREADS + tid;
                const int r = idx / k1p, c = idx - r * k1p;
                const int64_t gr = row0 + r;
                const double t = a.in[(gr <= last ? gr : last) * a.in_stride + (c < K1 ? c : 0)];
                v[u] = c < K1 ? t : 0.0;
                o[u] = idx < nx ? r * ldk + c : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (o[u] >= 0) Xs[o[u]] = v[u];
        }
        for (int base = 0; base < nw; base += 4 * CI_THREADS) {
            double v[4]; int o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * CI_THREADS + tid;
                const int r = idx / k1p, c = idx - r * k1p;
                const double t = a.W1[(int64_t)(r < H ? r : H - 1) * a.w1s + (c < K1 ? c : 0)];
                v[u] = (r < H && c < K1) ? t : 0.0;
                o[u] = idx < nw ? r * ldk + c : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (o[u] >= 0) W1s[o[u]] = v[u];
        }
        if (tid < CI_HMAX) b1s[tid] = tid < H ? a.b1[tid < H ? tid : 0] : 0.0;
    }
    __syncthreads();
    // two halves of 4 unit tiles (= the two k-steps of the int8 products): acc[g][r] = pre-activation of hidden unit 16 (4 s + g) + lq + 4 r for
    // row li (f64 C/D layout) = byte 4 g + r of k-step s
    bool nan_h = false;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        typename MF::Acc acc[JH / 2];
#pragma unroll
        for (int g = 0; g < JH / 2; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[g][r] = 0.0;
        for (int ks = 0; ks < k1p / KS; ++ks) {
            const int kk = ks * KS + lq;
            const double xb = Xs[(wave * MT + li) * ldk + kk];
#pragma unroll
            for (int g = 0; g < JH / 2; ++g) acc[g] = MF::mma(W1s[((4 * s + g) * MT + li) * ldk + kk], xb, acc[g]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j = 4 * s + g;
            int d[4][S];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double h = M<double>::tanh_fast(acc[g][r] + b1s[j * MT + lq + 4 * r]);
                nan_h |= h != h;                                   // (an infinite pre-activation is a legitimate h = +-1)
                ci_digits<S>(h, d[r]);
            }
#pragma unroll
            for (int i = 0; i < S; ++i) hd[s][i][g] = ci_pack4(d[0][i], d[1][i], d[2][i], d[3][i]);
        }
    }
    // lane (li, lq) holds 32 of row li's 128 units: a row is bad when any of its four lanes saw a NaN
    const unsigned long long m = __ballot(nan_h);
    return (unsigned)((m | (m >> 16) | (m >> 32) | (m >> 48)) & 0xffffu);
}

template <int S>
__global__ void __launch_bounds__(CI_THREADS, 4) mlp2_i8_kernel(const CiArgs a) {
    constexpr int W = ci_w_bytes(S), CHUNK = ci_chunk_bytes(S);
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned char* Ws0 = smem_raw;                                 // two packed chunks (double buffer)
    double* Xs = reinterpret_cast<double*>(smem_raw + CHUNK);      // phase 1 only (overlays buffer 1 while chunk 0 lands in buffer 0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * CI_ROWS;
    const int64_t last = a.B - 1;
    const int n_chunks = ci_chunks(a.N);
    const __amdgpu_buffer_rsrc_t packed_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.packed), 0, n_chunks * CHUNK, 0x00027000);
    ci_dma_chunk<S>(packed_rsrc, Ws0, 0, wave, lane);              // lands in buffer 0 while phase 1 works in buffer 1

    i32x4 hd[2][S];
    const unsigned nan_rows = ci_hidden<S>(a, row0, last, Xs, hd);

    // result layout: the digit slices of h are the A operand (matrix rows = the wave's 16 batch rows), W2's the B operand (matrix columns =
    // the tile's 16 output columns), so lane (n = lane % 16, lq) holds output column n of batch rows 4 lq + r, r = 0..3: every store
    // instruction of the wave writes four whole 128-byte lines (16 lanes x 8 bytes each).  The transposed assignment (4 consecutive columns
    // of one row per lane) wrote 16-byte pieces 32 bytes apart and ran at 2.8 TB/s of HBM writes -- the same 1.71 ms for 5 and for 6 slices.
    // Stores: raw buffer stores through a resource that covers exactly this workgroup's rows (rows past B and columns past N fall outside it /
    // get an out-of-range offset and are dropped by the hardware), so that every wave issues EXACTLY 4 stores per tile, masked or not -- the
    // chunk barrier can then wait for the DMA of the next chunk alone (s_waitcnt vmcnt(12): vector memory operations retire in order, the 12
    // stores of this chunk were issued after its DMA).  With conditional stores and vmcnt(0) every chunk waited for its stores to reach L2.
    const int rows_here = (int)(last - row0 + 1 < CI_ROWS ? last - row0 + 1 : CI_ROWS);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.out + row0 * a.os, 0, (int)((((int64_t)rows_here - 1) * a.os + a.N) * 8), 0x00027000);
    const unsigned obase = (unsigned)(((wave * 16 + 4 * lq) * a.os + li) * 8);       // byte offset of (first of this lane's 4 rows, column li)
    const unsigned ostep = (unsigned)(a.os * 8);
    const unsigned my_nan = (nan_rows >> (4 * lq)) & 0xfu;         // this lane stores rows 4 lq + r of the wave's 16
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto next_landed = [&]() { asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    landed();                                                      // chunk 0 is in buffer 0 and every wave is done with Xs / W1s / b1s (buffer 1)
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        if (chunk + 1 < n_chunks) ci_dma_chunk<S>(packed_rsrc, Ws0 + ((chunk + 1) & 1) * CHUNK, (chunk + 1) * CHUNK, wave, lane);   // in flight while this one is multiplied
        const unsigned char* Ws = Ws0 + (chunk & 1) * CHUNK;
        const double* tail = reinterpret_cast<const double*>(Ws + W);
        // one column tile at a time (S accumulators); the W fragments run two (tile, k-step, slice) steps ahead of the MFMAs that consume them.
        // Slice j of W meets the slices i <= S - 1 - j of h, and consecutive MFMAs go to different levels i + j (no back-to-back dependent
        // accumulators).
        constexpr int NF = CI_CT * 2 * S;
        i32x4 Wf[3];
        auto load_w = [&](int f) { Wf[f % 3] = *reinterpret_cast<const i32x4*>(Ws + f * CI_FRAG + lane * 16); };   // fragment order (tile, k-step, slice)
        load_w(0);
        load_w(1);
#pragma unroll
        for (int t = 0; t < CI_CT; ++t) {
            i32x4 acc[S];
#pragma unroll
            for (int lv = 0; lv < S; ++lv) acc[lv] = i32x4{0, 0, 0, 0};
#pragma unroll
            for (int sj = 0; sj < 2 * S; ++sj) {
                const int f = t * 2 * S + sj, s = sj / S, j = sj % S;
                if (f + 2 < NF) load_w(f + 2);
#pragma unroll
                for (int i = 0; i + j < S; ++i) acc[i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(hd[s][i], Wf[f % 3], acc[i + j], 0, 0, 0);
            }
            // float64 Horner over the levels (each int32 -> float64 conversion is exact), then the column's scale and bias
            const int col = 16 * (chunk * CI_CT + t) + li;
            const double scale = tail[t * 16 + li], bias = tail[CI_CT * 16 + t * 16 + li];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // levels pairwise in int32 first (|acc_d| <= (d + 1) 2^19, so 128 acc_d + acc_{d+1} < 2^31): half the conversions and multiply-adds
                double v = 0.0;
#pragma unroll
                for (int lv = (S - 1) & ~1; lv >= 0; lv -= 2) {
                    const int pair = lv + 1 < S ? acc[lv][r] * 128 + acc[lv + 1][r] : acc[lv][r] * 128;
                    v = v * 6.103515625e-05 + (double)pair;                                   // 2^-14 per pair of levels
                }
                const unsigned off = col < a.N ? obase + r * ostep + 128u * (chunk * CI_CT + t) : 0xfffffff0u;
                const double res = (my_nan >> r & 1u) ? __builtin_nan("") : v * scale + bias;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, res), out_rsrc, off, 0, 0);
            }
        }
        static_assert(CI_CT * 4 == 12, "next_landed waits for everything but this chunk's 12 stores");
        next_landed();                                             // next chunk in place, every wave has read this one
    }
}

// ---------------------------------------------------------------------------------------------------------- host side
static int ci_pack(const double* W2, int64_t w2s, const double* b2, int32_t H, int32_t N, int S, void* packed, void* stream) {
    if (!W2 || !packed || !ci_slices_ok(S)) return JF_ERR_BADARG;
    if (!width_ok(H) || !width_ok(N)) return JF_ERR_BADARG;
    if (H > CI_HMAX || (reinterpret_cast<uintptr_t>(packed) & 15u)) return JF_ERR_UNSUPPORTED;
    CiPackArgs a{W2, w2s, b2, H, N, static_cast<unsigned char*>(packed)};
    const int threads = ci_chunks(N) * CI_CT * 64;
    hipStream_t st = (hipStream_t)stream;
    if (S == 6) jf::launch(ci_pack_kernel<6>, dim3((threads + 255) / 256), dim3(256), 0, st, a);
    else jf::launch(ci_pack_kernel<5>, dim3((threads + 255) / 256), dim3(256), 0, st, a);
    return hipPeekAtLastError() == hipSuccess ? JF_OK : JF_ERR_LAUNCH;
}

template <int S> static int ci_launch(const CiArgs& a, hipStream_t st) {
    // phase 1's scratch overlays chunk buffer 1 and may be larger than it (K1 = 28: 60 KB)
    const int k1p = (a.K1 + 3) / 4 * 4;
    const size_t scratch = ((size_t)(CI_ROWS + CI_HMAX) * (k1p + 1) + CI_HMAX) * 8;
    const size_t second = scratch > (size_t)ci_chunk_bytes(S) ? (scratch + 15) / 16 * 16 : (size_t)ci_chunk_bytes(S);
    const size_t lds = (size_t)ci_chunk_bytes(S) + second;
    static LdsAttrOnce attr;                                       // per template instance (S) and per device
    attr.set((const void*)mlp2_i8_kernel<S>, 128 * 1024);
    jf::launch((mlp2_i8_kernel<S>), dim3((unsigned)((a.B + CI_ROWS - 1) / CI_ROWS)), dim3(CI_THREADS), lds, st, a);
    return hipPeekAtLastError() == hipSuccess ? JF_OK : JF_ERR_LAUNCH;
}

static int ci_mlp2(const double* in, int64_t in_stride, const double* W1, int64_t w1s, const double* b1, const void* packed, int64_t B, int32_t K1,
                   int32_t H, int32_t N, int S, double* out, int64_t os, void* stream, const jf_cond_segment* segs = nullptr, int32_t n_segs = 0) {
    CondIn cin{};
    if (segs || n_segs) {
        const int rc = cond_in_make(segs, n_segs, K1, cin);
        if (rc != JF_OK) return rc;
        in = static_cast<const double*>(cin.s[0].src);
    }
    if (!in || !W1 || !b1 || !packed || !out || !ci_slices_ok(S)) return JF_ERR_BADARG;
    if (!width_ok(K1) || !width_ok(H) || !width_ok(N) || !rows_ok(B)) return JF_ERR_BADARG;
    if (K1 > CI_K1MAX || H > CI_HMAX || (reinterpret_cast<uintptr_t>(packed) & 15u)) return JF_ERR_UNSUPPORTED;
    if (os < N || os > (1 << 20)) return JF_ERR_UNSUPPORTED;       // 31-bit byte counts inside a workgroup's 128 rows
    if (B == 0) return JF_OK;
    const CiArgs a{in, in_stride, W1, w1s, b1, static_cast<const unsigned char*>(packed), K1, H, N, B, out, os, cin};
    return S == 6 ? ci_launch<6>(a, (hipStream_t)stream) : ci_launch<5>(a, (hipStream_t)stream);
}

}  // namespace jf

extern "C" {
int64_t jf_mlp2_i8_packed_bytes(int32_t N, int32_t slices) {
    if (!jf::width_ok(N) || !jf::ci_slices_ok(slices)) return JF_ERR_BADARG;
    return (int64_t)jf::ci_chunks(N) * jf::ci_chunk_bytes(slices);
}
int jf_mlp2_i8_pack_f64(const double* W2, int64_t w2s, const double* b2, int32_t H, int32_t N, int32_t slices, void* packed, void* s) {
    return jf::ci_pack(W2, w2s, b2, H, N, slices, packed, s);
}
int jf_mlp2_i8_f64(const double* in, int64_t is, const double* W1, int64_t w1s, const double* b1, const void* packed, int64_t B, int32_t K1, int32_t H,
                   int32_t N, int32_t slices, double* out, int64_t os, void* s) {
    return jf::ci_mlp2(in, is, W1, w1s, b1, packed, B, K1, H, N, slices, out, os, s);
}
int jf_mlp2_i8_seg_f64(const jf_cond_segment* segs, int32_t n_segs, const double* W1, int64_t w1s, const double* b1, const void* packed, int64_t B, int32_t K1,
                       int32_t H, int32_t N, int32_t slices, double* out, int64_t os, void* s) {
    if (!segs || n_segs < 1) return JF_ERR_BADARG;
    return jf::ci_mlp2(nullptr, 0, W1, w1s, b1, packed, B, K1, H, N, slices, out, os, s, segs, n_segs);
}
}
