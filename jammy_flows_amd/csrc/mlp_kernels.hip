// Dense layers of the parameter-emitting MLPs on the matrix cores:  out = act(in @ W^T + bias).
//
// Replaces torch.nn.Linear (+ tanh) of the default amortisation MLP (jammy_flows/main/default.py:656-670) and the
// U / V^T products of AmortizableMLP with permanent parameters (jammy_flows/amortizable_mlp.py:508-578).
// Shapes on the hot path: M = batch (2^20), K <= 128 (inputs / hidden / rank), N = 8 ... 1224 (parameter block width).
//
// f32: v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD); f64: v_mfma_f64_16x16x4_f64.
// Workgroup = 4 waves stacked along M: tile 128 (rows) x 64 (cols); each wave owns 32 x 64.  K is walked in chunks of 32
// staged through LDS with rows padded by one element (conflict-free ds_read_b32 for the A[i][k] / B[k][j] fragment reads).
// The epilogue adds the bias, applies tanh and writes 128-byte row segments per half wave.
#include "jf_common.h"
#include "jf_math.h"

namespace jf {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <typename T> struct Mfma;
template <> struct Mfma<float> {
    static constexpr int MT = 32, KS = 2, NREG = 16;
    using Acc = f32x16;
    static __device__ __forceinline__ Acc mma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
};
template <> struct Mfma<double> {
    static constexpr int MT = 16, KS = 4, NREG = 4;
    using Acc = f64x4;
    static __device__ __forceinline__ Acc mma(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int reg, int lane) { return (lane >> 4) + 4 * reg; }
};

constexpr int BM = 128, BN = 64, KC = 32, WM = 32, WN = 64, LDP = KC + 1;

// [ROWS x KC] chunk of a row-major matrix -> LDS rows padded to LDP; rows >= max_row and columns >= kc are zero filled
template <typename T, int ROWS>
__device__ __forceinline__ void stage_chunk(T* __restrict__ dst, const T* __restrict__ src, int64_t stride, int64_t row0, int64_t max_row, int k0,
                                            int kc, int tid, bool vec_ok) {
    constexpr int N = Vec16<T>::N;
    using V = typename Vec16<T>::type;
    if (vec_ok) {
        constexpr int PER_ROW = KC / N;
        constexpr int CNT = ROWS * PER_ROW / 256;
        V v[CNT];
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
            const int idx = u * 256 + tid;
            const int r = idx / PER_ROW, c = (idx % PER_ROW) * N;
            const int64_t gr = row0 + r;
            if (gr < max_row && c < kc) v[u] = *reinterpret_cast<const V*>(src + gr * stride + k0 + c);     // kc % N == 0 when vec_ok
            else if constexpr (N == 4) v[u] = V{T(0), T(0), T(0), T(0)};
            else v[u] = V{T(0), T(0)};
        }
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
            v[u] = (gr < max_row && c < kc) ? src[gr * stride + k0 + c] : T(0);
        }
#pragma unroll
        for (int u = 0; u < CNT; ++u) {
            const int idx = u * 256 + tid;
            dst[(idx / KC) * LDP + (idx % KC)] = v[u];
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) linear_kernel(const T* __restrict__ in, int64_t in_stride, const T* __restrict__ W, int64_t w_stride,
                                                     const T* __restrict__ bias, int64_t B, int K, int N, int act, T* __restrict__ out,
                                                     int64_t out_stride) {
    using MF = Mfma<T>;
    constexpr int MT = MF::MT, KS = MF::KS, NREG = MF::NREG;
    constexpr int TM = WM / MT, TN = WN / MT;   // mfma tiles per wave
    __shared__ T As[BM * LDP];
    __shared__ T Ws[BN * LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = (N + BN - 1) / BN;                 // 1-D grid, N tile is the fast index: the blocks that share an activation
    const int col0 = (int)(blockIdx.x % n_tiles) * BN;     // tile are dispatched back to back and find it in L2
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
        stage_chunk<T, BN>(Ws, W, w_stride, (int64_t)col0, (int64_t)N, k0, kc, tid, vec_w);
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

template <typename T>
static int linear(const T* in, int64_t in_stride, const T* W, int64_t w_stride, const T* bias, int64_t B, int32_t K, int32_t N, int32_t act, T* out,
                  int64_t out_stride, void* stream) {
    if (!in || !W || !out || K < 1 || N < 1 || B < 0 || (act != 0 && act != 1)) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    const int64_t blocks = ((B + BM - 1) / BM) * ((N + BN - 1) / BN);
    if (blocks > 0x7fffffffLL) return JF_ERR_UNSUPPORTED;
    dim3 grid((unsigned)blocks);
    hipLaunchKernelGGL(linear_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, in, in_stride, W, w_stride, bias, B, (int)K, (int)N, (int)act, out,
                       out_stride);
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
}
