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

// ------------------------------------------------------------------------------------------------------------------
// Fused two-layer amortisation MLP:  out = (tanh(in @ W1^T + b1)) @ W2^T + b2      (main/default.py:656-670 with one hidden layer)
// The hidden activations of a row tile stay in LDS (never written to HBM); W2 is streamed through LDS in 64-column tiles with the
// next tile prefetched into registers while the current one feeds the MFMAs.  Requires K1 <= 64 and H in {32, 64, 96, 128}.
// ------------------------------------------------------------------------------------------------------------------
constexpr int HMAX = 128, K1MAX = 32, LDH = HMAX + 1, LDK1 = K1MAX + 1;
constexpr int BM2 = 64, WM2 = 32, WN2 = 32;     // 64-row block, 4 waves as 2 (rows) x 2 (cols): 66 KB (f32) of LDS => 2 workgroups per CU

template <typename T>
__global__ void __launch_bounds__(256) mlp2_kernel(const T* __restrict__ in, int64_t in_stride, const T* __restrict__ W1, int64_t w1_stride,
                                                   const T* __restrict__ b1, const T* __restrict__ W2, int64_t w2_stride, const T* __restrict__ b2,
                                                   int64_t B, int K1, int H, int N, T* __restrict__ out, int64_t out_stride) {
    using MF = Mfma<T>;
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    constexpr int MT = MF::MT, KS = MF::KS, NREG = MF::NREG;
    constexpr int TM = WM2 / MT, TN = WN2 / MT;              // mfma tiles per wave in phase 2 (f32: 1 x 1, f64: 2 x 2)
    constexpr int TH = (HMAX / 2) / MT;                      // phase 1: each wave owns 32 rows x 64 hidden units
    constexpr int WPT = BN * HMAX / VN / 256;                // 16-byte pieces of a W2 tile per thread
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* Hs = reinterpret_cast<T*>(smem_raw);                  // [BM2][LDH]   hidden activations
    T* Ws = Hs + BM2 * LDH;                                  // [BN][LDH]    W2 tile (phase 2)
    T* Xs = Ws;                                              // [BM2][LDK1]  input tile  } phase 1 only: overlays the W2 tile region
    T* W1s = Xs + BM2 * LDK1;                                // [HMAX][LDK1] W1          } (a barrier separates the phases)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane % MT, lk = lane / MT;
    const int64_t row0 = (int64_t)blockIdx.x * BM2;
    const int k1p = (K1 + KS - 1) / KS * KS;

    // ---- phase 1: h = tanh(x W1^T + b1)   (loads unconditional: clamped address + select)
    {
        const int64_t last = B - 1;
        for (int idx = tid; idx < BM2 * k1p; idx += 256) {
            const int r = idx / k1p, c = idx - r * k1p;
            const int64_t gr = row0 + r;
            const T t = in[(gr <= last ? gr : last) * in_stride + (c < K1 ? c : 0)];
            Xs[r * LDK1 + c] = (gr <= last && c < K1) ? t : T(0);
        }
        for (int idx = tid; idx < HMAX * k1p; idx += 256) {
            const int r = idx / k1p, c = idx - r * k1p;
            const T t = W1[(int64_t)(r < H ? r : H - 1) * w1_stride + (c < K1 ? c : 0)];
            W1s[r * LDK1 + c] = (r < H && c < K1) ? t : T(0);
        }
    }
    __syncthreads();
    {
        typename MF::Acc acc[TM][TH];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TH; ++j)
#pragma unroll
                for (int r = 0; r < NREG; ++r) acc[i][j][r] = T(0);
        for (int s = 0; s < k1p / KS; ++s) {
            const int kk = s * KS + lk;
            T a[TM], b[TH];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Xs[(wm * WM2 + i * MT + li) * LDK1 + kk];
#pragma unroll
            for (int j = 0; j < TH; ++j) b[j] = W1s[(wn * (HMAX / 2) + j * MT + li) * LDK1 + kk];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TH; ++j) acc[i][j] = MF::mma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();       // every wave is done reading Xs / W1s (the W2 tile region) -- Hs is a separate region
#pragma unroll
        for (int j = 0; j < TH; ++j) {
            const int hc = wn * (HMAX / 2) + j * MT + li;
            const T bv = b1[hc < H ? hc : 0];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    const int hr = wm * WM2 + i * MT + MF::row_of(r, lane);
                    Hs[hr * LDH + hc] = hc < H ? M<T>::tanh(acc[i][j][r] + bv) : T(0);
                }
        }
    }
    // ---- phase 2: out = h W2^T + b2, W2 streamed in BN-column tiles (prefetched into registers one tile ahead)
    const int n_tiles = (N + BN - 1) / BN;
    V wreg[WPT];
    auto fetch = [&](int t) {       // straight-line: clamped addresses + selects, so the WPT loads are issued back to back
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int idx = u * 256 + tid;
            const int r = idx / (HMAX / VN), c = (idx % (HMAX / VN)) * VN;
            const int gc = t * BN + r;
            const bool ok = (gc < N) && (c < H);
            const V v = *reinterpret_cast<const V*>(W2 + (int64_t)(gc < N ? gc : N - 1) * w2_stride + (c < H ? c : 0));
            wreg[u].x = ok ? v.x : T(0); wreg[u].y = ok ? v.y : T(0);
            if constexpr (VN == 4) { wreg[u].z = ok ? v.z : T(0); wreg[u].w = ok ? v.w : T(0); }
        }
    };
    fetch(0);
    const int ksteps = (H + KS - 1) / KS;
    for (int t = 0; t < n_tiles; ++t) {
        __syncthreads();                                     // previous tile's MFMAs are done with Ws; first pass: Hs is complete
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int idx = u * 256 + tid;
            const int r = idx / (HMAX / VN), c = (idx % (HMAX / VN)) * VN;
            T* d = Ws + r * LDH + c;
            d[0] = wreg[u].x; d[1] = wreg[u].y;
            if constexpr (VN == 4) { d[2] = wreg[u].z; d[3] = wreg[u].w; }
        }
        __syncthreads();
        if (t + 1 < n_tiles) fetch(t + 1);                   // in flight while the MFMAs below run
        typename MF::Acc acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < NREG; ++r) acc[i][j][r] = T(0);
        const T* ha = Hs + (wm * WM2 + li) * LDH + lk;
        const T* wb = Ws + (wn * WN2 + li) * LDH + lk;
#pragma unroll 2
        for (int s = 0; s < ksteps; ++s) {
            T a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = ha[i * MT * LDH + s * KS];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = wb[j * MT * LDH + s * KS];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(a[i], b[j], acc[i][j]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gc = t * BN + wn * WN2 + j * MT + li;
            const T bv = (b2 != nullptr) ? b2[gc < N ? gc : 0] : T(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < NREG; ++r) {
                    const int64_t gr = row0 + wm * WM2 + i * MT + MF::row_of(r, lane);
                    if (gr < B && gc < N) out[gr * out_stride + gc] = acc[i][j][r] + bv;
                }
        }
    }
}

template <typename T>
static int mlp2(const T* in, int64_t in_stride, const T* W1, int64_t w1_stride, const T* b1, const T* W2, int64_t w2_stride, const T* b2, int64_t B,
                int32_t K1, int32_t H, int32_t N, T* out, int64_t out_stride, void* stream) {
    if (!in || !W1 || !b1 || !W2 || !out || K1 < 1 || N < 1 || H < 1 || B < 0) return JF_ERR_BADARG;
    if (K1 > K1MAX || H > HMAX) return JF_ERR_UNSUPPORTED;
    if ((H % Vec16<T>::N) || (w2_stride % Vec16<T>::N) || (reinterpret_cast<uintptr_t>(W2) & 15u)) return JF_ERR_UNSUPPORTED;   // 16-byte W2 rows
    if (B == 0) return JF_OK;
    const size_t phase1 = (size_t)BM2 * LDK1 + (size_t)HMAX * LDK1, phase2 = (size_t)BN * LDH;
    const size_t lds = ((size_t)BM2 * LDH + (phase1 > phase2 ? phase1 : phase2)) * sizeof(T);
    auto k = mlp2_kernel<T>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)((B + BM2 - 1) / BM2)), dim3(256), lds, (hipStream_t)stream, in, in_stride, W1, w1_stride, b1, W2, w2_stride, b2, B,
                       (int)K1, (int)H, (int)N, out, out_stride);
    return check_launch();
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
int jf_mlp2_f32(const float* in, int64_t is, const float* W1, int64_t w1s, const float* b1, const float* W2, int64_t w2s, const float* b2, int64_t B,
                int32_t K1, int32_t H, int32_t N, float* out, int64_t os, void* s) {
    return jf::mlp2<float>(in, is, W1, w1s, b1, W2, w2s, b2, B, K1, H, N, out, os, s);
}
int jf_mlp2_f64(const double* in, int64_t is, const double* W1, int64_t w1s, const double* b1, const double* W2, int64_t w2s, const double* b2, int64_t B,
                int32_t K1, int32_t H, int32_t N, double* out, int64_t os, void* s) {
    return jf::mlp2<double>(in, is, W1, w1s, b1, W2, w2s, b2, B, K1, H, N, out, os, s);
}
}
