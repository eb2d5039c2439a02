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
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int col0 = blockIdx.y * BN;

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
        // stage A chunk [BM x KC] and W chunk [BN x KC], zero padded
        for (int idx = tid; idx < BM * KC; idx += 256) {
            const int r = idx / KC, c = idx - r * KC;
            const int64_t gr = row0 + r;
            As[r * LDP + c] = (gr < B && c < kc) ? in[gr * in_stride + k0 + c] : T(0);
        }
        for (int idx = tid; idx < BN * KC; idx += 256) {
            const int r = idx / KC, c = idx - r * KC;
            const int gc = col0 + r;
            Ws[r * LDP + c] = (gc < N && c < kc) ? W[(int64_t)gc * w_stride + k0 + c] : T(0);
        }
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
    dim3 grid((unsigned)((B + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
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
