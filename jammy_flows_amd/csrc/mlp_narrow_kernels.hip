// jf_mlp2_f32 for NARROW amortisation MLPs: out = tanh(in W1^T + b1) W2^T + b2 with <= 4 inputs, <= 128 hidden units, <= 64 outputs
// (main/default.py:656-670) -- the MLP in front of an S1 / interval block: 1 -> 128 -> 8 for pdf("i1+s1", "r+o"), BASELINE configuration 4;
// 4 -> 128 -> 46 in front of the sphere block of pdf("e4+s2+e4", "gggg+f+gggg") with the spline options the docs recommend (c3b).
//
// mlp2_kernel (mlp_kernels.hip) runs both products of such an MLP on exact-f32 MFMA: v_mfma_f32_16x16x4_f32 issues at the VECTOR rate on CDNA4,
// the one input column is padded to a 4-wide k-step, the 8 output columns to a 32-column tile -- 0.119 ms per 2^20 rows for 1.1 kflop per row.
// Here a wave takes 16 rows at a time, lane = (row n = lane % 16, quad q = lane / 16):
//   * first layer on the vector unit: the lane's 32 hidden units 32 s + 8 q + i (s < 4, i < 8) are K1 <= 4 FMAs each -- cheaper than one
//     padded MFMA k-step per 16 units -- with W1 / b1 pre-multiplied by 2 log2(e), so that tanh(v) = 1 - 2 / (2^z + 1) is v_exp_f32, an add,
//     v_rcp_f32 and one FMA (saturates cleanly: 2^z = inf -> 1, 0 -> -1; NaN propagates);
//   * second layer on the f16 matrix pipe with the arithmetic of the fused blocks (jf_cond_split.h): every f32 operand as two f16 pieces
//     scaled into the normal range, three v_mfma_f32_16x16x32_f16 passes (lo hi, hi lo, hi hi), f32 accumulation -- error below a plain f32
//     product's own rounding.  The unit order above IS the B-operand layout of that instruction, so the hidden activations go from the
//     vector unit into the matrix pipe without a shuffle; W2's eight A fragments (4 k-steps x 2 pieces) stay in 32 registers for the whole
//     kernel; the result registers are 4 consecutive output columns of the lane's row: one 16-byte store.
// More than 16 outputs: NT <= 4 column tiles of 16 share the hidden activations of the k-step (the split of h is done once), NT x 3 MFMAs
// per k-step, W2's fragments for all tiles in registers (32 per tile).  4 -> 128 -> 46: 0.151 ms (mlp2_kernel) per 2^18 rows before.
// A resident set of workgroups walks the row tiles (weights staged once per workgroup).  What is left per row is 128 tanh = 256 quarter-rate
// transcendentals: ~0.03 ms per 2^20 rows of v_exp / v_rcp issue alone.
#include "jf_cond_split.h"
#include "jf_math.h"

namespace jf {

constexpr int MN_HMAX = 128, MN_K1MAX = 4, MN_NTMAX = 4, MN_NMAX = 16 * MN_NTMAX;

struct MnArgs {
    const float* in; int64_t is;
    const float* W1; int64_t w1s; const float* b1;
    const float* W2; int64_t w2s; const float* b2;
    int64_t B;
    int K1, H, N;
    float* out; int64_t os;
    int vec_out;                                                     // 4: 16-byte aligned output rows and N a multiple of 4: one store per lane and tile; 2: 8-byte; 1: scalar
};

template <int NT>
__global__ void __launch_bounds__(256) mlp2_narrow_kernel(const MnArgs a) {
    __shared__ __align__(16) float w1s[MN_K1MAX][MN_HMAX];           // [input][unit], times 2 log2(e)
    __shared__ __align__(16) float b1s[MN_HMAX];
    __shared__ __align__(16) unsigned char frag[NT * CS_KSTEPS * 2 * CS_FRAG];   // (tile, k-step, piece): 64 lanes x 8 f16
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    constexpr float TWO_LOG2E = 2.8853900817779268f;
    for (int i = tid; i < MN_K1MAX * MN_HMAX; i += 256) {
        const int k = i / MN_HMAX, u = i - k * MN_HMAX;
        w1s[k][u] = (k < a.K1 && u < a.H) ? a.W1[(int64_t)u * a.w1s + k] * TWO_LOG2E : 0.f;
    }
    for (int i = tid; i < MN_HMAX; i += 256) b1s[i] = i < a.H ? a.b1[i] * TWO_LOG2E : 0.f;
    // absmax of W2 -> the power of two that puts it into [2^14, 2^15) (f16 normal range for the low pieces as well: jf_cond_split.h)
    float amax = 0.f;
    for (int i = tid; i < a.N * a.H; i += 256) {
        const int r = i / a.H, c = i - r * a.H;
        amax = fmaxf(amax, fabsf(a.W2[(int64_t)r * a.w2s + c]));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const int e = (amax > 0.f && amax < INFINITY) ? 14 - ilogbf(amax) : 0;
    const float wscale = ldexpf(1.0f, e), w2_inv = ldexpf(1.0f, -(e + 14));
    // this thread's fragment lane: k-step s = wave, lane (m, q): output column 16 j + m, hidden units 32 s + 8 q + i
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int s = wave, col = 16 * j + n;
        f16x8 hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int u = 32 * s + 8 * q + i;
            const float w = (col < a.N && u < a.H) ? a.W2[(int64_t)col * a.w2s + u] * wscale : 0.f;
            const _Float16 h16 = (_Float16)w;
            hi[i] = h16; lo[i] = (_Float16)(w - (float)h16);
        }
        *reinterpret_cast<f16x8*>(frag + (size_t)((j * CS_KSTEPS + s) * 2 + 0) * CS_FRAG + lane * 16) = hi;
        *reinterpret_cast<f16x8*>(frag + (size_t)((j * CS_KSTEPS + s) * 2 + 1) * CS_FRAG + lane * 16) = lo;
    }
    __syncthreads();
    f16x8 aH[NT][CS_KSTEPS], aL[NT][CS_KSTEPS];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int s = 0; s < CS_KSTEPS; ++s) {
            aH[j][s] = *reinterpret_cast<const f16x8*>(frag + (size_t)((j * CS_KSTEPS + s) * 2 + 0) * CS_FRAG + lane * 16);
            aL[j][s] = *reinterpret_cast<const f16x8*>(frag + (size_t)((j * CS_KSTEPS + s) * 2 + 1) * CS_FRAG + lane * 16);
        }
    f32x4 bias[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int v = 0; v < 4; ++v) bias[j][v] = (a.b2 != nullptr && 16 * j + 4 * q + v < a.N) ? a.b2[16 * j + 4 * q + v] : 0.f;

    const int64_t last = a.B - 1, n_tiles = (a.B + 15) / 16;
    const int K1 = a.K1;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n_tiles; t += (int64_t)gridDim.x * 4) {
        const int64_t row = t * 16 + n;
        const int64_t rrow = row <= last ? row : last;
        float x[MN_K1MAX];
#pragma unroll
        for (int k = 0; k < MN_K1MAX; ++k) x[k] = k < K1 ? a.in[rrow * a.is + k] : 0.f;
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < CS_KSTEPS; ++s) {
            const int u0 = 32 * s + 8 * q;
            float z[8];
            {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(b1s + u0), b1v = *reinterpret_cast<const f32x4*>(b1s + u0 + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) { z[i] = b0[i]; z[4 + i] = b1v[i]; }
            }
#pragma unroll
            for (int k = 0; k < MN_K1MAX; ++k) {
                if (k < K1) {                                    // uniform
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(&w1s[k][u0]), w1v = *reinterpret_cast<const f32x4*>(&w1s[k][u0 + 4]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { z[i] = fmaf(w0[i], x[k], z[i]); z[4 + i] = fmaf(w1v[i], x[k], z[4 + i]); }
                }
            }
            using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
            u32x4 ph, pl;
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                const float h0 = fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z[i]) + 1.0f), 1.0f);
                const float h1 = fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z[i + 1]) + 1.0f), 1.0f);
                unsigned hh, ll;
                cs_split16(h0 * CS_H_SCALE, h1 * CS_H_SCALE, hh, ll);
                ph[i >> 1] = hh; pl[i >> 1] = ll;
            }
            const f16x8 hH = __builtin_bit_cast(f16x8, ph), hL = __builtin_bit_cast(f16x8, pl);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aL[j][s], hH, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aH[j][s], hL, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aH[j][s], hH, acc[j], 0, 0, 0);
            }
        }
        // acc[j][v] = output column 16 j + 4 q + v of row n, in units of 2^(e + 14)
        if (row <= last) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int c0 = 16 * j + 4 * q;
                float* o = a.out + row * a.os + c0;
                const f32x4 r = acc[j] * w2_inv + bias[j];
                if (a.vec_out == 4) {
                    if (c0 < a.N) *reinterpret_cast<f32x4*>(o) = r;
                } else if (a.vec_out == 2) {
                    using f32x2 = __attribute__((ext_vector_type(2))) float;
                    if (c0 < a.N) *reinterpret_cast<f32x2*>(o) = f32x2{r[0], r[1]};
                    if (c0 + 2 < a.N) *reinterpret_cast<f32x2*>(o + 2) = f32x2{r[2], r[3]};
                } else {
#pragma unroll
                    for (int v = 0; v < 4; ++v) if (c0 + v < a.N) o[v] = r[v];
                }
            }
        }
    }
}

// -> JF_ERR_UNSUPPORTED when the shape is not this kernel's (the caller then takes mlp2_kernel)
int mlp2_narrow_f32(const float* in, int64_t is, const float* W1, int64_t w1s, const float* b1, const float* W2, int64_t w2s, const float* b2, int64_t B,
                    int32_t K1, int32_t H, int32_t N, float* out, int64_t os, void* stream) {
    if (K1 < 1 || K1 > MN_K1MAX || H < 1 || H > MN_HMAX || N < 1 || N > MN_NMAX) return JF_ERR_UNSUPPORTED;
    if (B == 0) return JF_OK;
    MnArgs a{in, is, W1, w1s, b1, W2, w2s, b2, B, K1, H, N, out, os, 0};
    const bool al16 = (reinterpret_cast<uintptr_t>(out) & 15u) == 0, al8 = (reinterpret_cast<uintptr_t>(out) & 7u) == 0;
    a.vec_out = (N % 4 == 0 && os % 4 == 0 && al16) ? 4 : (N % 2 == 0 && os % 2 == 0 && al8) ? 2 : 1;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    auto go = [&](auto kernel) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        const int64_t wg_tiles = (B + 63) / 64, resident = (int64_t)cus * per_cu;
        jf::launch(kernel, dim3((unsigned)(wg_tiles < resident ? wg_tiles : resident)), dim3(256), 0, (hipStream_t)stream, a);
    };
    switch ((N + 15) / 16) {
        case 1: go(mlp2_narrow_kernel<1>); break;
        case 2: go(mlp2_narrow_kernel<2>); break;
        case 3: go(mlp2_narrow_kernel<3>); break;
        default: go(mlp2_narrow_kernel<4>); break;
    }
    return check_launch();
}

}  // namespace jf
