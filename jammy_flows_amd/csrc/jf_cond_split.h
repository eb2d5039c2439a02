// Shared by the split-bf16 conditional-block kernels (cond_split_kernels.hip: log-prob and sampling direction; cond_bwd_kernels.hip: the
// adjoint): chunk geometry of the packed W2 image, the reductions over the four coordinate lanes of a row, the supported layer options.
#pragma once
#include "jf_cond_in.h"
#include "jf_cond_regs.h"
#include "jf_mfma.h"

namespace jf {

constexpr int CS_TILES = 9;                        // 16-column MFMA tiles per layer (36 slots x 4 coordinates)
static_assert(CS_SLOTS == 4 * CS_TILES, "one register per slot, four registers per tile");
constexpr int CS_CT = 3;                           // tiles per chunk
constexpr int CS_CPL = CS_TILES / CS_CT;           // chunks per layer
constexpr int CS_KSTEPS = 4;                       // 128 hidden units = 4 x 32
constexpr int CS_NP = 3;                           // bf16 pieces per f32 operand
constexpr int CS_FRAG = 1024;                      // bytes of one A fragment (64 lanes x 8 bf16)
constexpr int CS_W_BYTES = CS_CT * CS_KSTEPS * CS_NP * CS_FRAG;       // 36864
constexpr int CS_B_BYTES = CS_CT * 16 * 4;                            // 192: the chunk's bias, permuted column order
constexpr int CS_CHUNK_BYTES = CS_W_BYTES + CS_B_BYTES;               // 37056 (16-byte multiple)
// second arithmetic (JF_SPLIT_F16X2): every operand = TWO f16 pieces (11 + 11 significant bits), three products (lo hi, hi lo, hi hi) instead of
// six, two fragments per (tile, k-step) instead of three.  Representation error <= 2^-22 per operand -- below the rounding of the f32
// accumulation either arithmetic ends in (scripts/probe/f16split.py: rms error of the 128-term products 2.4e-8 vs 6.8e-8 for a plain f32 matrix
// product, 1.8e-9 for the bf16 triple).  Range: W2 is scaled by a power of two so that its largest entry sits in [2^14, 2^15), h (in (-1, 1)) by
// 2^14, so a low piece leaves the normal f16 range only below 2^-28 of the largest weight / 2^-28 absolute in h -- whether the matrix core
// keeps or flushes such subnormals is then immaterial (<= 4e-9 |w| per term).  The chunk's bias tail carries the bias in scaled units (the
// accumulators start from it) and the inverse scale 2^-(e + 14) that brings the result back (exact).
constexpr int CS_NP16 = 2;
constexpr int CS_W16_BYTES = CS_CT * CS_KSTEPS * CS_NP16 * CS_FRAG;  // 24576
constexpr int CS_B16_BYTES = CS_B_BYTES + 16;                         // bias + {2^-(e + 14), pad}
constexpr int CS_CHUNK16_BYTES = CS_W16_BYTES + CS_B16_BYTES;         // 24784 (16-byte multiple)
constexpr float CS_H_SCALE = 16384.0f;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
// power-of-two scale of W2 for the f16 pieces: the largest entry lands in [2^14, 2^15) (exponent clamped: a matrix whose largest entry is
// below 2^-46 or above 2^74 keeps its pieces inside the f16 range only partly -- not a weight matrix)
__device__ __forceinline__ int cs_w_exponent(float wmax) {
    if (!(wmax > 0.f && wmax < INFINITY)) return 0;
    const int e = 14 - ilogbf(wmax);
    return e < -60 ? -60 : (e > 60 ? 60 : e);
}
// largest |W[n][k]| of an (N, H) matrix with row stride ws -> out[0] (out[1..3] = 0): ONE workgroup of 1024 threads walks the matrix (70 K
// elements for the C3 block: ~4 us).  The first version used one atomicMax per wave of a 274-block grid on a zeroed word: 14 us of
// serialised atomics per pack, and a memset node that broke HIP-graph capture on ROCm 7.2.
static __global__ void __launch_bounds__(1024) cs_absmax_kernel(const float* __restrict__ W, int64_t ws, int N, int H, float* __restrict__ out) {
    __shared__ float part[16];
    float m = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (H == 128 && ws % 4 == 0 && (reinterpret_cast<uintptr_t>(W) & 15u) == 0) {
        // 16-byte loads: half a wave covers a row, a wave takes 8 rows per turn (4 independent loads in flight), 16 waves 128 rows
        const int half = lane >> 5, l4 = (lane & 31) * 4;
        for (int n0 = 8 * wave; n0 < N; n0 += 128) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = n0 + 2 * u + half;
                v[u] = n < N ? *reinterpret_cast<const f32x4*>(W + (int64_t)n * ws + l4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u][0]), fabsf(v[u][1]))), fmaxf(fabsf(v[u][2]), fabsf(v[u][3])));
        }
    } else {
        for (int n0 = 4 * wave; n0 < N; n0 += 64) {                // a wave takes 4 rows per turn: independent loads, no index division
            for (int k = lane; k < H; k += 64) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = n0 + u < N ? W[(int64_t)(n0 + u) * ws + k] : 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) m = fmaxf(m, fabsf(v[u]));
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 4) {
        float t = 0.f;
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 0; w < 16; ++w) t = fmaxf(t, part[w]);
        }
        out[threadIdx.x] = t;
    }
}
template <int NP> struct CsGeom;
template <> struct CsGeom<3> { static constexpr int W = CS_W_BYTES, B = CS_B_BYTES, CHUNK = CS_CHUNK_BYTES; };
template <> struct CsGeom<2> { static constexpr int W = CS_W16_BYTES, B = CS_B16_BYTES, CHUNK = CS_CHUNK16_BYTES; };
// v (two values) -> packed f16 pairs {hi0, hi1} and {lo0, lo1}: hi = RN_f16(v), lo = RN_f16(v - hi)
__device__ __forceinline__ void cs_split16(float v0, float v1, unsigned& hi, unsigned& lo) {
    const f16x2 h = __builtin_convertvector(f32x2{v0, v1}, f16x2);
    const f32x2 back = __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(f32x2{v0 - back[0], v1 - back[1]}, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
constexpr int CS_ROWS1 = 64;                       // rows per workgroup and row group (4 waves x 16); a wave carries RG row groups
constexpr int CS_HMAX = 128, CS_K1MAX = 28;

// ---------------------------------------------------------------------------------------------------------- row-group reductions
// the 4 coordinate lanes of a row are l, l^16, l^32, l^48.  v_permlane16_swap(vdst, src) exchanges vdst's odd 16-lane rows with src's even
// rows, v_permlane32_swap the upper half of vdst with the lower half of src (scripts/probe/swapsem.hip), so with both operands = v the
// two results are "my pair's even member" and "my pair's odd member" in every lane.  Written as inline asm: hipcc (ROCm 7.2) miscompiles
// __builtin_amdgcn_permlane{16,32}_swap(v, v) followed by op(r[0], r[1]) into op(r[0], r[0]) (scripts/probe/layout16x32.hip caught it).
// s_nop 1 = the two wait states the swap needs after a VALU write of its operands.
template <typename Op> __device__ __forceinline__ float cs_rreduce(float v, Op op) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    float c = op(a, b), e = c;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(e));
    return op(c, e);
}
__device__ __forceinline__ float cs_rsum(float v) { return cs_rreduce(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ float cs_rmax(float v) { return cs_rreduce(v, [](float a, float b) { return fmaxf(a, b); }); }

static inline bool cs_layer_supported(const jf_gf_layer& h, int D) {
    return h.num_kde == CS_K && h.hh_iter >= 0 && h.hh_iter <= CS_HH && h.nonlinear_stretch_type == JF_GF_STRETCH_CLASSIC &&
           h.rotation_mode == JF_GF_ROT_HOUSEHOLDER && !h.center_mean && !h.add_skewness &&
           h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION && !h.clamp_widths && h.fit_normalization && h.regulate_normalization &&
           h.width_min > 0 && h.width_max > 0 && D >= 3 && D <= 4;
}

// one chunk of a packed image -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: 1 KiB per wave instruction, no register hop): wave w of the 4 moves
// KiB pieces w, w + 4, ...; the bias tail goes with wave 0.  A plain function with the sizes as arguments (constants after inlining): the same
// builtin inside a lambda of a kernel TEMPLATE whose sizes depend on a template parameter made hipcc's host pass drop the kernel's stub
// without a diagnostic (undefined __device_stub__ symbols when the library loads).
__device__ __forceinline__ void cs_dma_chunk(__amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int g, int lane_off, int wave, int lane, int w_bytes,
                                             int b_bytes) {
#pragma unroll
    for (int u = 0; u < CS_W_BYTES / 4096; ++u)
        if (u < w_bytes / 4096) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (cs_lptr)(dst + (u * 4 + wave) * 1024), 16, lane_off, g + u * 4096, 0, 0);
    if (wave == 0 && lane < b_bytes / 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (cs_lptr)(dst + w_bytes), 16, lane * 16, g + w_bytes, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------- phase 1
// h^T = tanh(W1 x^T + b1) for the wave's RG x 16 rows (exact f32 MFMA; rows past B replicate row B-1), returned as MFMA B operands: three
// bf16 pieces per value, k-slot i of lane group q in k-step s = hidden unit 16 (2 s + i / 4) + 4 q + i % 4.  Xs: LDS scratch of
// (CS_ROWS1 RG + CS_HMAX) (k1p + 1) + CS_HMAX floats.  STORE_H: the f32 activations also go to h_out (B, H) -- the adjoint's weight-gradient
// product reads them.  Ends with every wave past the barrier that follows the staging, NOT past one after the MFMA reads: the caller's next
// barrier covers those.
template <int RG, bool STORE_H, int NP = CS_NP>
__device__ __forceinline__ void cs_hidden(const float* __restrict__ in, int64_t in_stride, const float* __restrict__ W1, int64_t w1s,
                                          const float* __restrict__ b1, int K1, int H, int64_t row0, int64_t last, float* Xs,
                                          bf16x8 (&hB)[RG][CS_KSTEPS][NP], float* __restrict__ h_out, int64_t hs, const CondIn* cin = nullptr) {
    using MF = Mfma16<float>;
    constexpr int CS_ROWS = CS_ROWS1 * RG;
    constexpr int MT = 16, KS = 4, NREG = 4, JH = CS_HMAX / MT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int k1p = (K1 + KS - 1) / KS * KS, ldk = k1p + 1;
    float* W1s = Xs + CS_ROWS * ldk;
    float* b1s = W1s + CS_HMAX * ldk;
    {
        const int nx = CS_ROWS * k1p, nw = CS_HMAX * k1p;
        if (cin) {
            // segments (jf_cond_in.h): locate, then every load of the batch, then the embedding arithmetic -- straight-line, one wait
            const bool any_embed = cond_in_any_embed(*cin);
            for (int base = 0; base < nx; base += 4 * 256) {
                CondLoc<float> loc[4]; float va[4], vb[4]; int o[4]; bool keep[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + u * 256 + tid;
                    const int r = idx / k1p, c = idx - r * k1p;
                    const int64_t gr = row0 + r;
                    loc[u] = cond_in_locate<float>(*cin, gr <= last ? gr : last, c < K1 ? c : 0);
                    keep[u] = c < K1;
                    o[u] = idx < nx ? r * ldk + c : -1;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { va[u] = *loc[u].pa; vb[u] = *loc[u].pb; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float t = cond_in_finish<float, true>(loc[u], va[u], vb[u], any_embed);
                    if (o[u] >= 0) Xs[o[u]] = keep[u] ? t : 0.f;
                }
            }
        } else
        for (int base = 0; base < nx; base += 4 * 256) {
            float v[4]; int o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256 + tid;
                const int r = idx / k1p, c = idx - r * k1p;
                const int64_t gr = row0 + r;
                const float t = in[(gr <= last ? gr : last) * in_stride + (c < K1 ? c : 0)];
                v[u] = c < K1 ? t : 0.f;
                o[u] = idx < nx ? r * ldk + c : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (o[u] >= 0) Xs[o[u]] = v[u];
        }
        for (int base = 0; base < nw; base += 4 * 256) {
            float v[4]; int o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256 + tid;
                const int r = idx / k1p, c = idx - r * k1p;
                const float t = W1[(int64_t)(r < H ? r : H - 1) * w1s + (c < K1 ? c : 0)];
                v[u] = (r < H && c < K1) ? t : 0.f;
                o[u] = idx < nw ? r * ldk + c : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (o[u] >= 0) W1s[o[u]] = v[u];
        }
        if (tid < CS_HMAX) b1s[tid] = tid < H ? b1[tid < H ? tid : 0] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        typename MF::Acc acc[JH];
#pragma unroll
        for (int j = 0; j < JH; ++j)
#pragma unroll
            for (int r = 0; r < NREG; ++r) acc[j][r] = 0.f;
        for (int s = 0; s < k1p / KS; ++s) {
            const int kk = s * KS + lq;
            const float xb = Xs[((wave * RG + g) * MT + li) * ldk + kk];
#pragma unroll
            for (int j = 0; j < JH; ++j) acc[j] = MF::mma(W1s[(j * MT + li) * ldk + kk], xb, acc[j]);
        }
        // acc[j][r] = pre-activation of hidden unit 16 j + 4 lq + r for row li: k-slot i of k-step s <-> (j = 2 s + i / 4, r = i % 4)
#pragma unroll
        for (int s = 0; s < CS_KSTEPS; ++s) {
            // split by truncation, two values at a time (and / sub / and / sub + one v_perm_b32 per packed pair; exact as well: 24 significant
            // bits = 3 x 8): three v_cvt_pk_bf16_f32 per value were the expensive part of this loop
            using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
            u32x4 q0, q1, q2;
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                const int j = 2 * s + (i >> 2), r = i & 3;
                const float h0 = M<float>::tanh_fast(acc[j][r] + b1s[j * MT + 4 * lq + r]);
                const float h1 = M<float>::tanh_fast(acc[j][r + 1] + b1s[j * MT + 4 * lq + r + 1]);
                if constexpr (STORE_H) { acc[j][r] = h0; acc[j][r + 1] = h1; }
                if constexpr (NP == 2) {
                    unsigned ph, pl;
                    cs_split16(h0 * CS_H_SCALE, h1 * CS_H_SCALE, ph, pl);
                    q0[i >> 1] = ph; q1[i >> 1] = pl;
                    continue;
                }
                const unsigned a0 = __builtin_bit_cast(unsigned, h0), a1 = __builtin_bit_cast(unsigned, h1);
                const float r0 = h0 - __builtin_bit_cast(float, a0 & 0xffff0000u), r1 = h1 - __builtin_bit_cast(float, a1 & 0xffff0000u);
                const unsigned c0 = __builtin_bit_cast(unsigned, r0), c1 = __builtin_bit_cast(unsigned, r1);
                const float s0 = r0 - __builtin_bit_cast(float, c0 & 0xffff0000u), s1 = r1 - __builtin_bit_cast(float, c1 & 0xffff0000u);
                q0[i >> 1] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
                q1[i >> 1] = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
                q2[i >> 1] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
            }
            hB[g][s][0] = __builtin_bit_cast(bf16x8, q0); hB[g][s][1] = __builtin_bit_cast(bf16x8, q1);
            if constexpr (NP == 3) hB[g][s][2] = __builtin_bit_cast(bf16x8, q2);
        }
        if constexpr (STORE_H) {
            const int64_t r = row0 + (wave * RG + g) * MT + li;
            if (r <= last) {
#pragma unroll
                for (int j = 0; j < JH; ++j) {
                    const int c = j * MT + 4 * lq;                 // hidden units c .. c + 3 (H is a multiple of 4 on this path)
                    if (c < H) *reinterpret_cast<f32x4*>(h_out + r * hs + c) = acc[j];
                }
            }
        }
    }
}

}  // namespace jf
