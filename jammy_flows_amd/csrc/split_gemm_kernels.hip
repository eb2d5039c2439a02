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
    auto split_x = [&]() {
#pragma unroll
        for (int g = 0; g < RG; ++g)
#pragma unroll
            for (int s = 0; s < KC; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __bf16 p0, p1, p2;
                    sg_split(xraw[g][s][i >> 2][i & 3], p0, p1, p2);
                    xB[g][s][0][i] = p0; xB[g][s][1][i] = p1; xB[g][s][2][i] = p2;
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
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (a.bias && n0 < a.N) b = *reinterpret_cast<const f32x4*>(a.bias + n0);
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
    if (N < 1 || K < 1) return JF_ERR_BADARG;
    const SgShape sh = sg_shape(N);
    return sg_groups(N, sh.ng) * sg_kchunks(K, sh.kc) * (int64_t)sh.ng * sh.kc * SG_NP * SG_FRAG;
}

static int sg_pack(const float* W, int64_t ws, int64_t wks, int N, int K, void* packed, void* stream) {
    if (!W || !packed || N < 1 || K < 1) return JF_ERR_BADARG;
    const SgShape sh = sg_shape(N);
    const int n_kc = (int)sg_kchunks(K, sh.kc);
    const int64_t total = sg_groups(N, sh.ng) * n_kc * sh.ng * sh.kc * 64;
    hipLaunchKernelGGL(sg_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ws, wks, N, K, sh.ng, sh.kc, n_kc, total,
                       static_cast<unsigned char*>(packed));
    return check_launch();
}

template <int NG, int KC, int TB> static int sg_launch(const SgArgs& a, hipStream_t st) {
    constexpr int RG = 2;
    const size_t lds = (size_t)2 * NG * KC * SG_NP * SG_FRAG;
    auto k = split_gemm_kernel<NG, KC, TB, RG>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)((a.B + 64 * RG - 1) / (64 * RG))), dim3(256), lds, st, a);
    return check_launch();
}

static int sg_linear(const float* X, int64_t xs, const void* packed, const float* bias, int64_t B, int K, int N, float* out, int64_t os, void* stream) {
    if (!X || !packed || !out || B < 0 || K < 1 || N < 1) return JF_ERR_BADARG;
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
int64_t jf_linear_split_packed_bytes(int32_t N, int32_t K) { return jf::sg_packed_bytes(N, K); }
int jf_linear_split_pack_f32(const float* W, int64_t w_row_stride, int64_t w_col_stride, int32_t N, int32_t K, void* packed, void* stream) {
    return jf::sg_pack(W, w_row_stride, w_col_stride, N, K, packed, stream);
}
int jf_linear_split_f32(const float* X, int64_t x_stride, const void* packed, const float* bias, int64_t B, int32_t K, int32_t N, float* out,
                        int64_t out_stride, void* stream) {
    return jf::sg_linear(X, x_stride, packed, bias, B, K, N, out, out_stride, stream);
}
}
