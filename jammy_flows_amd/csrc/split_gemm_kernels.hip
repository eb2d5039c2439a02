// Dense layer on split-bf16 matrix arithmetic:  out (B, N) = X (B, K) W^T (N, K) + bias, float32 in and out, float32-equivalent accuracy.
//
//   jf_linear_split_packed_bytes / jf_linear_split_pack_f32   W -> the image of MFMA A-fragments the kernel streams (once per weight version)
//   jf_linear_split_f32                                        the product
//
// Why: f32-input MFMA runs at the vector rate on CDNA4 (157 TFLOP/s), so the three B x 548 x 128 products of the conditional block's
// backward -- the parameter block recomputed from the hidden activations, g_params W2, and (wgrad_kernels.hip) g_params^T h -- cost 0.35 .. 0.6 ms
// per 2^18 rows each, a third of the training step.  As in cond_split_kernels.hip every f32 operand is split into three bf16 pieces
// (v = hi + mid + lo exactly) and the product is the six v_mfma_f32_16x16x32_bf16 whose piece indices sum to <= 2, accumulated in f32: the
// dropped terms are <= 3 * 2^-24 |w||x|, the size of one f32 rounding; six passes at 16x the f32 matrix rate.
//
// Work distribution: a workgroup (4 waves) owns 64 RG rows; a wave holds its 16 RG rows of X as MFMA B operands in registers (split on the
// fly from coalesced 16-byte loads), W arrives as ready-made A fragments through a double-buffered LDS chunk (buffer_load ... lds), so the
// accumulator of lane (row = lane % 16, q = lane / 16) holds out[row][16 t + 4 q .. + 3]: one 16-byte store per tile.
// Two shapes of the loop nest (template NG = column tiles per chunk, KC = k-steps of 32 per chunk):
//   wide  <3, 4>: N large, K <= 128 per chunk (the recomputed parameter block: N = 548, K = 128) -- X is split once, 12 column groups stream by;
//   deep  <8, 1>: N <= 128, K large (g_params W2: N = 128, K = 548) -- all accumulators resident, X streams by in 32-column slices.
#include "jf_common.h"
#include "jf_mfma.h"
#include <cstdint>

namespace jf {

using sg_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
typedef __attribute__((address_space(3))) void* sg_lptr;

constexpr int SG_FRAG = 1024;                      // bytes of one A fragment (64 lanes x 8 bf16)
constexpr int SG_NP = 3;                           // bf16 pieces per f32 operand

__device__ __forceinline__ void sg_split(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)v;
    const float r1 = v - (float)hi;                // exact
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);                // exact difference, rounded once
}

// The same split by TRUNCATION, two values at a time, as packed bf16 pairs: a float32 has 24 significant bits, each piece takes the next 8, so
// v = hi + mid + lo holds exactly here as well.  and / sub / and / sub per value + one v_perm_b32 per packed pair: full-rate VALU only -- the
// RNE form needs three v_cvt_pk_bf16_f32 per value, which dominated the staging of the weight-gradient kernel.
__device__ __forceinline__ void sg_split2(float v0, float v1, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned a0 = __builtin_bit_cast(unsigned, v0), a1 = __builtin_bit_cast(unsigned, v1);
    const float r0 = v0 - __builtin_bit_cast(float, a0 & 0xffff0000u), r1 = v1 - __builtin_bit_cast(float, a1 & 0xffff0000u);
    const unsigned b0 = __builtin_bit_cast(unsigned, r0), b1 = __builtin_bit_cast(unsigned, r1);
    const float s0 = r0 - __builtin_bit_cast(float, b0 & 0xffff0000u), s1 = r1 - __builtin_bit_cast(float, b1 & 0xffff0000u);
    constexpr unsigned SEL = 0x07060302u;                          // {hi16(first operand), hi16(second operand)}
    p0 = __builtin_amdgcn_perm(a1, a0, SEL);
    p1 = __builtin_amdgcn_perm(b1, b0, SEL);
    p2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), SEL);
}

struct SgShape { int ng, kc; };
static inline SgShape sg_shape(int N) { return N <= 128 ? SgShape{8, 1} : SgShape{3, 4}; }
static inline int64_t sg_groups(int N, int ng) { return ((N + 15) / 16 + ng - 1) / ng; }
static inline int64_t sg_kchunks(int K, int kc) { return (K + 32 * kc - 1) / (32 * kc); }

// k of k-slot i (0..7) of lane group q in k-step s: the same permutation on both operands (cond_split_kernels.hip uses it as well)
__device__ __forceinline__ int sg_k(int s, int q, int i) { return 32 * s + 16 * (i >> 2) + 4 * q + (i & 3); }

// one thread per (chunk, tile, k-step, lane): the three pieces' fragments, 16 bytes each
__global__ void __launch_bounds__(256) sg_pack_kernel(const float* __restrict__ W, int64_t ws, int64_t wks, int N, int K, int NG, int KC, int n_kc, int64_t total,
                                                      unsigned char* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    int64_t rest = idx >> 6;
    const int s = (int)(rest % KC); rest /= KC;
    const int t = (int)(rest % NG); rest /= NG;
    const int64_t chunk = rest;
    const int kc = (int)(chunk % n_kc), ng = (int)(chunk / n_kc);
    const int m = lane & 15, q = lane >> 4;
    const int n = 16 * (ng * NG + t) + m;
    sg_bf16x8 f[SG_NP];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = sg_k(kc * KC + s, q, i);
        const float w = (n < N && k < K) ? W[(int64_t)n * ws + (int64_t)k * wks] : 0.0f;
        __bf16 p0, p1, p2;
        sg_split(w, p0, p1, p2);
        f[0][i] = p0; f[1][i] = p1; f[2][i] = p2;
    }
    unsigned char* base = out + (size_t)chunk * ((size_t)NG * KC * SG_NP * SG_FRAG);
#pragma unroll
    for (int p = 0; p < SG_NP; ++p)
        *reinterpret_cast<sg_bf16x8*>(base + (size_t)((t * KC + s) * SG_NP + p) * SG_FRAG + lane * 16) = f[p];
}

struct SgArgs {
    const float* X; int64_t xs;
    const unsigned char* packed;
    const float* bias;
    int64_t B;
    int K, N, n_groups, n_kc;
    float* out; int64_t os;
};

template <int NG, int KC, int TB, int RG> __global__ void __launch_bounds__(256, 2) split_gemm_kernel(const SgArgs a) {
    constexpr int CHUNK = NG * KC * SG_NP * SG_FRAG;
    constexpr int NB = NG / TB, NSTEP = KC * NB;
    static_assert(NG % TB == 0 && CHUNK % 4096 == 0, "chunk layout");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * (64 * RG);
    const int64_t last = a.B - 1;
    const int total = a.n_groups * a.n_kc;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.packed), 0, total * CHUNK, 0x00027000);
    const int lane_off = wave * 1024 + lane * 16;
    auto dma = [&](int chunk) {
        unsigned char* l = smem_raw + (chunk & 1) * CHUNK;
#pragma unroll
        for (int u = 0; u < CHUNK / 4096; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (sg_lptr)(l + (u * 4 + wave) * 1024), 16, lane_off, chunk * CHUNK + u * 4096, 0, 0);
    };
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    dma(0);
    // the bias, zero-padded to whole tiles, behind the two chunk buffers: the epilogue of a column group must not wait for a global load
    float* bs = reinterpret_cast<float*>(smem_raw + 2 * CHUNK);
    for (int i = tid; i < a.n_groups * NG * 16; i += 256) bs[i] = (a.bias && i < a.N) ? a.bias[i] : 0.f;

    int64_t row[RG]; const float* xrow[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        row[g] = row0 + (wave * RG + g) * 16 + li;
        xrow[g] = a.X + (row[g] <= last ? row[g] : last) * a.xs;
    }
    // raw X slice of one k-chunk: per k-step the two 16-byte pieces k = 32 s + 4 q .. and 32 s + 16 + 4 q ..
    f32x4 xraw[RG][KC][2];
    auto load_x = [&](int kc) {
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int s = 0; s < KC; ++s)
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {
                    const int k = 32 * (kc * KC + s) + 16 * hlf + 4 * lq;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (k < a.K) v = *reinterpret_cast<const f32x4*>(xrow[g] + k);       // K % 4 == 0: a piece is inside or outside as a whole
                    xraw[g][s][hlf] = v;
                }
    };
    sg_bf16x8 xB[RG][KC][SG_NP];
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    auto split_x = [&]() {
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                u32x4 q0, q1, q2;
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    unsigned p0, p1, p2;
                    sg_split2(xraw[g][s][i >> 2][i & 3], xraw[g][s][i >> 2][(i & 3) + 1], p0, p1, p2);
                    q0[i >> 1] = p0; q1[i >> 1] = p1; q2[i >> 1] = p2;
                }
                xB[g][s][0] = __builtin_bit_cast(sg_bf16x8, q0); xB[g][s][1] = __builtin_bit_cast(sg_bf16x8, q1); xB[g][s][2] = __builtin_bit_cast(sg_bf16x8, q2);
            }
    };
    load_x(0);
    landed();                                                      // chunk 0 in buffer 0, X slice 0 in registers
    int chunk = 0;
    for (int ng = 0; ng < a.n_groups; ++ng) {
        f32x4 acc[RG][NG];
#pragma unroll
        for (int t = 0; t < NG; ++t)
#pragma unroll
            for (int g = 0; g < RG; ++g) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < a.n_kc; ++kc, ++chunk) {
            if (chunk + 1 < total) dma(chunk + 1);                 // in flight while this chunk is multiplied
            if (a.n_kc > 1 || ng == 0) split_x();                  // one k-chunk in all: the pieces of group 0 serve every column group
            if (a.n_kc > 1) load_x(kc + 1 < a.n_kc ? kc + 1 : 0);  // the next slice's loads fly behind this chunk's MFMAs
            const unsigned char* Ws = smem_raw + (chunk & 1) * CHUNK;
            sg_bf16x8 A[2][TB][SG_NP];
            auto load_a = [&](int step, int buf) {
                const int s = step / NB, tb = step % NB;
#pragma unroll
                for (int t = 0; t < TB; ++t)
#pragma unroll
                    for (int p = 0; p < SG_NP; ++p)
                        A[buf][t][p] = *reinterpret_cast<const sg_bf16x8*>(Ws + (((tb * TB + t) * KC + s) * SG_NP + p) * SG_FRAG + lane * 16);
            };
            load_a(0, 0);
#pragma unroll
            for (int step = 0; step < NSTEP; ++step) {
                const int s = step / NB, tb = step % NB, b = step & 1;
                if (step + 1 < NSTEP) load_a(step + 1, b ^ 1);
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};       // piece indices pa + pb <= 2, smallest products first
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int t = 0; t < TB; ++t)
#pragma unroll
                        for (int g = 0; g < RG; ++g)
                            acc[g][tb * TB + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[b][t][PA[i]], xB[g][s][PB[i]], acc[g][tb * TB + t], 0, 0, 0);
            }
            landed();                                              // next chunk (and X slice) in place, every wave has read this one
        }
#pragma unroll
        for (int t = 0; t < NG; ++t) {                              // the bias joins at the end: one rounding, not one per accumulation step
            const int n0 = 16 * (ng * NG + t) + 4 * lq;
            const f32x4 b = *reinterpret_cast<const f32x4*>(bs + n0);
#pragma unroll
            for (int g = 0; g < RG; ++g)
                if (row[g] <= last && n0 < a.N) *reinterpret_cast<f32x4*>(a.out + row[g] * a.os + n0) = acc[g][t] + b;
        }
    }
}

static bool sg_supported(const void* X, int64_t xs, const float* bias, const void* out, int64_t os, int K, int N) {
    return K % 4 == 0 && N % 4 == 0 && xs % 4 == 0 && os % 4 == 0 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(out) |
            reinterpret_cast<uintptr_t>(bias)) & 15u) == 0;
}

static int64_t sg_packed_bytes(int N, int K) {
    if (!width_ok(N) || !width_ok(K)) return JF_ERR_BADARG;
    const SgShape sh = sg_shape(N);
    return sg_groups(N, sh.ng) * sg_kchunks(K, sh.kc) * (int64_t)sh.ng * sh.kc * SG_NP * SG_FRAG;
}

static int sg_pack(const float* W, int64_t ws, int64_t wks, int N, int K, void* packed, void* stream) {
    if (!W || !packed || !width_ok(N) || !width_ok(K)) return JF_ERR_BADARG;
    const SgShape sh = sg_shape(N);
    const int n_kc = (int)sg_kchunks(K, sh.kc);
    const int64_t total = sg_groups(N, sh.ng) * n_kc * sh.ng * sh.kc * 64;
    jf::launch(sg_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ws, wks, N, K, sh.ng, sh.kc, n_kc, total,
                       static_cast<unsigned char*>(packed));
    return check_launch();
}

template <int NG, int KC, int TB> static int sg_launch(const SgArgs& a, hipStream_t st) {
    constexpr int RG = 2;
    const size_t lds = (size_t)2 * NG * KC * SG_NP * SG_FRAG + (size_t)a.n_groups * NG * 16 * sizeof(float);
    if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;              // (the bias row of a very wide layer: N > ~20000)
    auto k = split_gemm_kernel<NG, KC, TB, RG>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    jf::launch(k, dim3((unsigned)((a.B + 64 * RG - 1) / (64 * RG))), dim3(256), lds, st, a);
    return check_launch();
}

static int sg_linear(const float* X, int64_t xs, const void* packed, const float* bias, int64_t B, int K, int N, float* out, int64_t os, void* stream) {
    if (!X || !packed || !out || !rows_ok(B) || !width_ok(K) || !width_ok(N)) return JF_ERR_BADARG;
    if (!sg_supported(X, xs, bias, out, os, K, N) || (reinterpret_cast<uintptr_t>(packed) & 15u)) return JF_ERR_UNSUPPORTED;
    if (B == 0) return JF_OK;
    const SgShape sh = sg_shape(N);
    SgArgs a{};
    a.X = X; a.xs = xs; a.packed = static_cast<const unsigned char*>(packed); a.bias = bias; a.B = B; a.K = K; a.N = N;
    a.n_groups = (int)sg_groups(N, sh.ng); a.n_kc = (int)sg_kchunks(K, sh.kc); a.out = out; a.os = os;
    if ((int64_t)a.n_groups * a.n_kc * sh.ng * sh.kc * SG_NP * SG_FRAG > 0x7fffffffLL) return JF_ERR_UNSUPPORTED;
    return sh.ng == 8 ? sg_launch<8, 1, 4>(a, (hipStream_t)stream) : sg_launch<3, 4, 3>(a, (hipStream_t)stream);
}

}  // namespace jf

extern "C" {
int64_t jf_linear_split_packed_bytes(int32_t N, int32_t K) { return (jf::width_ok(N) && jf::width_ok(K)) ? jf::sg_packed_bytes(N, K) : (int64_t)JF_ERR_BADARG; }
int jf_linear_split_pack_f32(const float* W, int64_t w_row_stride, int64_t w_col_stride, int32_t N, int32_t K, void* packed, void* stream) {
    return jf::sg_pack(W, w_row_stride, w_col_stride, N, K, packed, stream);
}
int jf_linear_split_f32(const float* X, int64_t x_stride, const void* packed, const float* bias, int64_t B, int32_t K, int32_t N, float* out,
                        int64_t out_stride, void* stream) {
    return jf::sg_linear(X, x_stride, packed, bias, B, K, N, out, out_stride, stream);
}
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// Weight gradient on the same arithmetic:  g_W (N, K) = sum_b g[b, :]^T in[b, :]  (+ column sums of g), K <= 128, B >> N.
// The reduction runs over the BATCH, so both MFMA operands need 8 batch rows per lane for one column -- a transposition of the row-major inputs.
// A workgroup (4 waves) owns 128 columns n of g and a range of rows; per step of 32 rows every thread loads one 4 x 4 patch (4 rows x 4
// columns, 16-byte loads) of g and one of `in`, splits its 16 + 16 values ONCE into bf16 pieces and writes them -- transposed in registers --
// as 8-byte runs of 4 rows into fragment images in LDS; the waves then read ready-made fragments (ds_read_b128) for 2 x 8 tiles each.
// Inside a fragment the 16-byte slot of lane (m, q) is the lane's own (16 q + m): the fragment reads (30 per wave and step, 1 KiB each) are
// conflict-free, the patch writes (12 KB per wave and step) take 2-way conflicts.  (Round 2 had the slots permuted for conflict-free writes,
// (m & 3) * 16 + (m >> 2) * 4 + q, which makes every fragment read a 4-way conflict -- 6.9e7 conflict cycles per launch in the counters; the
// kernel time is the same either way, 0.287 ms per 2^18 x 576 x 128: the two barriers per 32-row step and the split arithmetic bound it.)
// Partial slabs per row range, added by the caller.
// ------------------------------------------------------------------------------------------------------------------------------------------
namespace jf {

struct WsArgs {
    const float* g; int64_t gs;
    const float* in; int64_t is;
    int64_t B, rows_per_split;
    int K, N;
    float* pw; float* pb;
    const float* gmax;                          // NP = 2: device scalar, an upper bound of |g| (power-of-two scale of the f16 pieces)
    int in_exp;                                 // NP = 2: `in` is scaled by 2^in_exp (14 for activations in (-1, 1))
};

// NP = 2: f16 pairs instead of bf16 triples (three products instead of six, two fragments per tile instead of three), for the one caller
// whose operands have a known range: the packed gradient rows of the conditional block's adjoint (scaled by the power of two that brings
// their largest entry, handed over by that kernel, into [2^14, 2^15)) and its tanh activations (2^14).  An entry's absolute error is
// <= max(2^-22 |g|, 2^-39 max|g|): entries more than 2^-18 below the batch's largest gradient entry keep fewer than 22 bits -- their
// contribution to a batch sum is below the rounding of the f32 accumulation of the larger ones.
using ws_f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using ws_f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using ws_f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ void ws_split16(float v0, float v1, unsigned& hi, unsigned& lo) {
    const ws_f16x2 h = __builtin_convertvector(ws_f32x2{v0, v1}, ws_f16x2);
    const ws_f32x2 back = __builtin_convertvector(h, ws_f32x2);
    const ws_f16x2 l = __builtin_convertvector(ws_f32x2{v0 - back[0], v1 - back[1]}, ws_f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

constexpr int WS_NW = 128;                         // columns of g per workgroup
constexpr int WS_TILES = WS_NW / 16;               // = k-tiles of `in` (K <= 128)
constexpr int WS_FRAG = SG_FRAG + 16;              // fragment pitch in LDS: the 8 tiles a patch-write instruction touches start 12 banks apart

__device__ __forceinline__ int ws_slot(int m, int q) { return q * 16 + m; }    // = the lane: the 8 lanes a ds_read_b128 serves per cycle read 128 contiguous bytes

template <int NP> __global__ void __launch_bounds__(256, 2) wgrad_split_kernel(const WsArgs a) {
    __shared__ __align__(16) unsigned char Gs[WS_TILES * NP * WS_FRAG];        // 24 KiB (NP = 3): fragments of the g tile
    __shared__ __align__(16) unsigned char Is[WS_TILES * NP * WS_FRAG];        // 24 KiB: fragments of the `in` tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * WS_NW;
    const int64_t b0 = (int64_t)blockIdx.y * a.rows_per_split;
    const int64_t b1 = b0 + a.rows_per_split < a.B ? b0 + a.rows_per_split : a.B;
    const int steps = b1 > b0 ? (int)((b1 - b0 + 31) / 32) : 0;
    // staging patch of this thread: columns 4 pn .. 4 pn + 3, rows 4 b4 .. 4 b4 + 3 of the step
    const int pn = tid & 31, b4 = tid >> 5, pq = b4 >> 1, phi = b4 & 1;       // rows 16 phi + 4 pq ..: the two halves of a wave differ in phi (+ 8 bytes)
    const bool gcol = n0 + 4 * pn < a.N, icol = 4 * pn < a.K;
    const float* gp = a.g + (gcol ? n0 + 4 * pn : 0);
    const float* ip = a.in + (icol ? 4 * pn : 0);
    const int woff = ((pn >> 2) * NP) * WS_FRAG + (pq * 16 + (pn & 3) * 4) * 16 + phi * 8;   // + jj * 16 (slot m = 4 (pn & 3) + jj of lane group pq) + piece * WS_FRAG
    // two patch buffers: the loads of step i + 2 are issued when step i has been written to LDS (one step of MFMAs does not cover an HBM round trip)
    f32x4 gv[2][4], iv[2][4];
    auto load_patch = [&](int step, f32x4 (&gb)[4], f32x4 (&ib)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = b0 + 32 * (int64_t)step + 16 * phi + 4 * pq + j;
            const bool ok = r < b1;
            const int64_t rr = ok ? r : b1 - 1;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            gb[j] = (ok && gcol) ? *reinterpret_cast<const f32x4*>(gp + rr * a.gs) : z;
            ib[j] = (ok && icol) ? *reinterpret_cast<const f32x4*>(ip + rr * a.is) : z;
        }
    };
    using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
    float g_scale = 1.f, in_scale = 1.f, out_scale = 1.f;
    if constexpr (NP == 2) {
        const float gm = *a.gmax;
        int eg = 0;
        if (gm > 0.f && gm < INFINITY) { eg = 14 - ilogbf(gm); eg = eg < -60 ? -60 : (eg > 60 ? 60 : eg); }
        g_scale = ldexpf(1.f, eg); in_scale = ldexpf(1.f, a.in_exp); out_scale = ldexpf(1.f, -(eg + a.in_exp));
    }
    auto write_patch = [&](const f32x4 (&v)[4], unsigned char* base, float scale) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            unsigned char* w = base + woff + jj * 16;
            if constexpr (NP == 3) {
                unsigned a0, a1, a2, c0, c1, c2;
                sg_split2(v[0][jj], v[1][jj], a0, a1, a2);
                sg_split2(v[2][jj], v[3][jj], c0, c1, c2);
                const u32x2 p0 = {a0, c0}, p1 = {a1, c1}, p2 = {a2, c2};
                *reinterpret_cast<u32x2*>(w) = p0;
                *reinterpret_cast<u32x2*>(w + WS_FRAG) = p1;
                *reinterpret_cast<u32x2*>(w + 2 * WS_FRAG) = p2;
            } else {
                unsigned h0, l0, h1, l1;
                ws_split16(v[0][jj] * scale, v[1][jj] * scale, h0, l0);
                ws_split16(v[2][jj] * scale, v[3][jj] * scale, h1, l1);
                const u32x2 p0 = {h0, h1}, p1 = {l0, l1};
                *reinterpret_cast<u32x2*>(w) = p0;
                *reinterpret_cast<u32x2*>(w + WS_FRAG) = p1;
            }
        }
    };
    f32x4 acc[2][WS_TILES];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k = 0; k < WS_TILES; ++k) acc[t][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const int roff = ws_slot(li, lq) * 16;
    const bool wave_live = n0 + 16 * (2 * wave) < a.N;             // wave-uniform: some column of this wave's two tiles exists
    auto one_step = [&](int step, f32x4 (&gb)[4], f32x4 (&ib)[4]) {
        __syncthreads();                                           // every wave has read the previous step's fragments
        write_patch(gb, Gs, g_scale);
        write_patch(ib, Is, in_scale);
#pragma unroll
        for (int j = 0; j < 4; ++j) bsum += gb[j];
        __syncthreads();
        if (step + 2 < steps) load_patch(step + 2, gb, ib);        // in flight behind two steps of MFMAs
        if (!wave_live) return;
        sg_bf16x8 A[2][NP];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < NP; ++p) A[t][p] = *reinterpret_cast<const sg_bf16x8*>(Gs + ((2 * wave + t) * NP + p) * WS_FRAG + roff);
        constexpr int KB = 2;                                      // k-tiles of `in` held at a time
#pragma unroll
        for (int kh = 0; kh < WS_TILES / KB; ++kh) {
            sg_bf16x8 Bf[KB][NP];
#pragma unroll
            for (int k = 0; k < KB; ++k)
#pragma unroll
                for (int p = 0; p < NP; ++p) Bf[k][p] = *reinterpret_cast<const sg_bf16x8*>(Is + ((kh * KB + k) * NP + p) * WS_FRAG + roff);
            if constexpr (NP == 3) {
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int k = 0; k < KB; ++k)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            acc[t][kh * KB + k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[t][PA[i]], Bf[k][PB[i]], acc[t][kh * KB + k], 0, 0, 0);
            } else {
                constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};   // lo x hi, hi x lo, hi x hi
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = 0; k < KB; ++k)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            acc[t][kh * KB + k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ws_f16x8, A[t][PA[i]]), __builtin_bit_cast(ws_f16x8, Bf[k][PB[i]]),
                                                                                       acc[t][kh * KB + k], 0, 0, 0);
            }
        }
    };
    if (steps > 0) load_patch(0, gv[0], iv[0]);
    if (steps > 1) load_patch(1, gv[1], iv[1]);
    for (int step = 0; step < steps; step += 2) {
        one_step(step, gv[0], iv[0]);
        if (step + 1 < steps) one_step(step + 1, gv[1], iv[1]);
    }
    // acc[t][k][r] = g_W[n0 + 16 (2 wave + t) + 4 lq + r][16 k + li]
    float* slab = a.pw + (int64_t)blockIdx.y * a.N * a.K;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k = 0; k < WS_TILES; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * (2 * wave + t) + 4 * lq + r, kk = 16 * k + li;
                if (n < a.N && kk < a.K) slab[(int64_t)n * a.K + kk] = NP == 2 ? acc[t][k][r] * out_scale : acc[t][k][r];
            }
    if (a.pb != nullptr) {                                         // column sums of g: this thread's 4 columns over its rows, then over the 8 row groups
        __syncthreads();
        float* red = reinterpret_cast<float*>(Gs);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) red[b4 * WS_NW + 4 * pn + jj] = bsum[jj];
        __syncthreads();
        if (tid < WS_NW && n0 + tid < a.N) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += red[q * WS_NW + tid];
            a.pb[(int64_t)blockIdx.y * a.N + n0 + tid] = s;
        }
    }
}

static int64_t ws_splits(int64_t B, int N) {
    if (N < 1 || B < 1) return 1;                                  // (the entry points reject such shapes; the query must not divide by zero)
    const int64_t cols = (N + WS_NW - 1) / WS_NW;
    int64_t s = 512 / cols;                                        // ONE resident round (2 workgroups per CU): every slab is 4 N K bytes written and read
    if (s < 1) s = 1;                                              // again by jf_slab_sum (round 4: two rounds -> one, 205 -> 102 slabs of 295 KB for C3)
    const int64_t max_s = (B + 511) / 512;                         // at least 512 rows per split
    if (s > max_s) s = max_s;
    if (s > 65535) s = 65535;
    return s < 1 ? 1 : s;
}

static int ws_wgrad(const float* g, int64_t gs, const float* in, int64_t is, int64_t B, int K, int N, float* pw, float* pb, void* stream,
                    const float* gmax = nullptr, int in_exp = 0) {
    if (!g || !in || !pw || !rows_ok(B) || !width_ok(K) || !width_ok(N)) return JF_ERR_BADARG;
    if (K > 128 || K % 4 || N % 4 || gs % 4 || is % 4 || ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(in)) & 15u)) return JF_ERR_UNSUPPORTED;
    if (B == 0) return JF_OK;
    WsArgs a{};
    a.g = g; a.gs = gs; a.in = in; a.is = is; a.B = B; a.K = K; a.N = N; a.pw = pw; a.pb = pb; a.gmax = gmax; a.in_exp = in_exp;
    const int64_t S = ws_splits(B, N);
    a.rows_per_split = ((B + S - 1) / S + 31) / 32 * 32;
    const dim3 grid((unsigned)((N + WS_NW - 1) / WS_NW), (unsigned)S);
    if (gmax) jf::launch(wgrad_split_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else jf::launch(wgrad_split_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return check_launch();
}

}  // namespace jf

extern "C" {
int64_t jf_linear_wgrad_split_splits(int64_t B, int32_t N) { return (jf::width_ok(N) && jf::rows_ok(B)) ? jf::ws_splits(B, N) : (int64_t)JF_ERR_BADARG; }
int jf_linear_wgrad_split_f32(const float* g, int64_t g_stride, const float* in, int64_t in_stride, int64_t B, int32_t K, int32_t N, float* partial_w,
                              float* partial_b, void* stream) {
    return jf::ws_wgrad(g, g_stride, in, in_stride, B, K, N, partial_w, partial_b, stream);
}
int jf_linear_wgrad_split16_f32(const float* g, int64_t g_stride, const float* in, int64_t in_stride, int64_t B, int32_t K, int32_t N,
                                const float* g_absmax, int32_t in_exp, float* partial_w, float* partial_b, void* stream) {
    if (!g_absmax || in_exp < -60 || in_exp > 60) return JF_ERR_BADARG;
    return jf::ws_wgrad(g, g_stride, in, in_stride, B, K, N, partial_w, partial_b, stream, g_absmax, in_exp);
}
}
