// Conditional manifold block (amortisation MLP + chain of r / o / m / f layers, log-prob or sampling direction): kernel arguments and the body
// of cond_mchain_kernel as a device function -- shared by the stand-alone kernel (cond_manifold_kernels.hip) and the merged log-prob step
// (merged_kernels.hip).  main/default.py:656-670 (MLP), :998-1031 (layer loop).
#pragma once
#include <type_traits>

#include "jf_cond_split.h"
#include "jf_manifold.h"
#include "jf_mfma.h"

namespace jf {

constexpr int CM_HMAX = 128, CM_K1MAX = 28, CM_NMAX = 64;

// NL: capacity of the layer table (the merged step carries a one-layer copy: an `f` layer descriptor is 872 bytes and a launch has 4 KB of arguments)
template <typename T, typename CLayer, int NL = JF_MAX_MCHAIN> struct CmArgs {
    const T* in; int64_t in_stride;
    const T* W1; int64_t w1s; const T* b1;
    const T* W2; int64_t w2s; const T* b2;
    int K1, H, N;
    const T* x; int64_t xs;
    const T* ld_in;
    int64_t B;
    int n_layers, dim, tile_stride, scratch, tab;      // tab: lane-private knot-table elements (0 when no layer of the chain evaluates a spline)
    int col0[NL];
    CLayer L[NL];
    T* x_out; int64_t xos;
    T* ld_out;
    const T* blp_in; T* blp_out;
    int32_t* status;
};

// NT threads per workgroup (256 in float32; 128 in float64, whose lane-private knot tables are twice as large)
// FWD: the sampling direction (layers first to last, Fam::apply<T, true>: main/default.py:1482-1506); no base log-prob there
// (float32: a register budget for three waves per SIMD -- 140 + 32 AGPRs -> 112 VGPRs, the `f` block of C3 0.109 -> 0.105 ms per 2^20 rows)
// `block` / `n_blocks`: the workgroup's index among, and the number of, the block's workgroups (a resident set walks the row tiles)
template <typename T, class Fam, int NT, bool FWD, class Args>
__device__ __forceinline__ void cond_mchain_body(const Args& a, const int block, const int n_blocks, unsigned char* smem_raw) {
    using MF = Mfma16<T>;
    constexpr int MT = 16, KS = 4, NREG = 4, JH = CM_HMAX / MT;
    const int k1p = (a.K1 + KS - 1) / KS * KS, ldk = k1p + 1;
    const int np = (a.N + MT - 1) / MT * MT;                       // output columns padded to whole MFMA tiles
    constexpr int LDW = CM_HMAX + 1;
    T* W1s = reinterpret_cast<T*>(smem_raw);                       // [128][ldk]
    T* b1s = W1s + CM_HMAX * ldk;                                  // [128]
    T* W2s = b1s + CM_HMAX;                                        // [np][LDW]
    T* b2s = W2s + np * LDW;                                       // [np]
    T* Xs = b2s + np + 8;                                          // [NT][ldk]  (8: the absmax partials of the float32 path)
    T* tiles = Xs + NT * ldk;                                      // [NT][tile_stride]
    T* tabs = tiles + NT * a.tile_stride;                          // [NT][JF_SPLINE_TAB (+ scratch)]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int64_t last = a.B - 1;
    for (int i = tid; i < CM_HMAX * k1p; i += NT) {
        const int r = i / k1p, c = i - r * k1p;
        W1s[r * ldk + c] = (r < a.H && c < a.K1) ? a.W1[(int64_t)r * a.w1s + c] : T(0);
    }
    for (int i = tid; i < CM_HMAX; i += NT) b1s[i] = i < a.H ? a.b1[i] : T(0);
    float w2_inv = 1.f;                                              // float32: 2^-(e + 14), undoes the scales of W2 and h
    if constexpr (std::is_same<T, float>::value) {
        // absmax of W2 -> the power of two that puts it into [2^14, 2^15) (f16 normal range for the low pieces as well: jf_cond_split.h)
        float amax = 0.f;
        for (int i = tid; i < a.N * a.H; i += NT) {
            const int r = i / a.H, c = i - r * a.H;
            amax = fmaxf(amax, fabsf(a.W2[(int64_t)r * a.w2s + c]));
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
        float* red = reinterpret_cast<float*>(b2s + np);               // NT / 64 partial maxima (the host reserves them behind the bias)
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        amax = red[0];
        for (int w = 1; w < NT / 64; ++w) amax = fmaxf(amax, red[w]);
        const int e = (amax > 0.f && amax < INFINITY) ? 14 - ilogbf(amax) : 0;
        const float wscale = ldexpf(1.0f, e);
        w2_inv = ldexpf(1.0f, -(e + 14));
        // fragment (column tile ct, k-step s, piece p) = the A operand of one v_mfma_f32_16x16x32_f16: lane (m, q) carries output column 16 ct + m,
        // hidden units 16 (2 s + i / 4) + 4 q + i % 4, i = 0..7 (the k order phase 1 leaves the activations in)
        unsigned char* W2p = reinterpret_cast<unsigned char*>(W2s);
        for (int f = tid; f < (np / MT) * CS_KSTEPS * 64; f += NT) {
            const int fl = f & 63, s = (f >> 6) % CS_KSTEPS, ct = (f >> 6) / CS_KSTEPS;
            const int m = fl & 15, q = fl >> 4, col = ct * MT + m;
            f16x8 hi, lo;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 16 * (2 * s + (i >> 2)) + 4 * q + (i & 3);
                const float w = (col < a.N && k < a.H) ? a.W2[(int64_t)col * a.w2s + k] * wscale : 0.f;
                const _Float16 h16 = (_Float16)w;
                hi[i] = h16; lo[i] = (_Float16)(w - (float)h16);
            }
            *reinterpret_cast<f16x8*>(W2p + (size_t)((ct * CS_KSTEPS + s) * 2 + 0) * CS_FRAG + fl * 16) = hi;
            *reinterpret_cast<f16x8*>(W2p + (size_t)((ct * CS_KSTEPS + s) * 2 + 1) * CS_FRAG + fl * 16) = lo;
        }
    } else {
        for (int i = tid; i < np * CM_HMAX; i += NT) {
            const int r = i / CM_HMAX, c = i - r * CM_HMAX;
            W2s[r * LDW + c] = (r < a.N && c < a.H) ? a.W2[(int64_t)r * a.w2s + c] : T(0);
        }
    }
    for (int i = tid; i < np; i += NT) b2s[i] = (i < a.N && a.b2) ? a.b2[i] : T(0);
    // a resident set of workgroups walks the row tiles: the weights (and the float32 path's fragment image of W2) are staged once per workgroup,
    // not once per 256 rows (4096 workgroups each spent ~5 us on three dependent rounds of L2 loads before their first MFMA)
    for (int64_t row0 = (int64_t)block * NT; row0 < a.B; row0 += (int64_t)n_blocks * NT) {
    __syncthreads();                                               // the previous tile's readers of Xs / tiles are done
    for (int i = tid; i < NT * k1p; i += NT) {
        const int r = i / k1p, c = i - r * k1p;
        const int64_t gr = row0 + r;
        Xs[r * ldk + c] = c < a.K1 ? a.in[(gr <= last ? gr : last) * a.in_stride + c] : T(0);
    }
    __syncthreads();

    // ---- parameters of the wave's 64 rows -> its LDS tile, 16 rows at a time
    T* tile = tiles + wave * 64 * a.tile_stride;
    for (int rt = 0; rt < 4; ++rt) {
        T hreg[JH][NREG];
        {
            typename MF::Acc acc[JH];
#pragma unroll
            for (int j = 0; j < JH; ++j)
#pragma unroll
                for (int r = 0; r < NREG; ++r) acc[j][r] = T(0);
            for (int s = 0; s < k1p / KS; ++s) {
                const int kk = s * KS + lq;
                const T xb = Xs[(wave * 64 + rt * MT + li) * ldk + kk];
#pragma unroll
                for (int j = 0; j < JH; ++j) acc[j] = MF::mma(W1s[(j * MT + li) * ldk + kk], xb, acc[j]);
            }
#pragma unroll
            for (int j = 0; j < JH; ++j)
#pragma unroll
                for (int r = 0; r < NREG; ++r) hreg[j][r] = M<T>::tanh_fast(acc[j][r] + b1s[j * MT + MF::row_of(r, lane)]);
        }
        if constexpr (std::is_same<T, float>::value) {
            // activations -> f16 pairs in the B-operand layout (as cs_hidden), then three MFMA passes per (column tile, k-step): lo x hi, hi x lo, hi x hi
            f16x8 hH[CS_KSTEPS], hL[CS_KSTEPS];
#pragma unroll
            for (int s = 0; s < CS_KSTEPS; ++s) {
                using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
                u32x4 q0, q1;
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    const int j = 2 * s + (i >> 2), r = i & 3;
                    unsigned ph, pl;
                    cs_split16(hreg[j][r] * CS_H_SCALE, hreg[j][r + 1] * CS_H_SCALE, ph, pl);
                    q0[i >> 1] = ph; q1[i >> 1] = pl;
                }
                hH[s] = __builtin_bit_cast(f16x8, q0); hL[s] = __builtin_bit_cast(f16x8, q1);
            }
            const unsigned char* W2p = reinterpret_cast<const unsigned char*>(W2s);
            for (int ct = 0; ct < np / MT; ++ct) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < CS_KSTEPS; ++s) {
                    const f16x8 aH = *reinterpret_cast<const f16x8*>(W2p + (size_t)((ct * CS_KSTEPS + s) * 2 + 0) * CS_FRAG + lane * 16);
                    const f16x8 aL = *reinterpret_cast<const f16x8*>(W2p + (size_t)((ct * CS_KSTEPS + s) * 2 + 1) * CS_FRAG + lane * 16);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aL, hH[s], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aH, hL[s], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aH, hH[s], acc, 0, 0, 0);
                }
                // acc[v] = parameter (16 ct + 4 lq + v) of row (rt * 16 + li), in units of 2^(e + 14)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = ct * MT + 4 * lq + v;
                    tile[(rt * MT + li) * a.tile_stride + c] = acc[v] * w2_inv + b2s[c];
                }
            }
        } else
        for (int ct = 0; ct < np / MT; ++ct) {
            typename MF::Acc acc;
#pragma unroll
            for (int r = 0; r < NREG; ++r) acc[r] = T(0);
            const T* wrow = W2s + (ct * MT + li) * LDW;
#pragma unroll
            for (int j = 0; j < JH; ++j)
#pragma unroll
                for (int r = 0; r < NREG; ++r) acc = MF::mma(wrow[j * MT + MF::row_of(r, lane)], hreg[j][r], acc);
            // acc[v] = parameter (16 ct + row_of(v, lane)) of row (rt * 16 + li)
#pragma unroll
            for (int v = 0; v < NREG; ++v) {
                const int c = ct * MT + MF::row_of(v, lane);
                tile[(rt * MT + li) * a.tile_stride + c] = acc[v] + b2s[c];
            }
        }
    }
    __syncthreads();

    // ---- the layers, lane-per-row (as mchain_kernel)
    const int64_t row = row0 + tid;
    const bool active = row <= last;
    const int64_t rrow = active ? row : last;
    T x[3] = {T(0), T(0), T(0)};
#pragma unroll
    for (int d = 0; d < Fam::DIM; ++d) if (d < a.dim) x[d] = a.x[rrow * a.xs + d];
    T ld = a.ld_in ? a.ld_in[rrow] : T(0);
    LaneCtx<T> ctx;
    ctx.tab = tabs + tid * (a.tab + a.scratch);
    ctx.corr = ctx.tab + a.tab;
    ctx.bins = nullptr; ctx.bin_i = 0;
    ctx.oob = ctx.nonconv = ctx.nonfinite = false;
    ctx.lane_valid = active;
    const T* prow = tiles + tid * a.tile_stride;
    for (int i = 0; i < a.n_layers; ++i) {
        const int l = FWD ? i : a.n_layers - 1 - i;
        Fam::template apply<T, FWD>(a.L[l], prow + a.col0[l], x, ld, ctx);
    }
    bool bad = !M<T>::finite(ld);
#pragma unroll
    for (int d = 0; d < Fam::DIM; ++d) bad = bad || !M<T>::finite(x[d]);
    if (active) {
#pragma unroll
        for (int d = 0; d < Fam::DIM; ++d) if (d < a.dim) a.x_out[row * a.xos + d] = x[d];
        a.ld_out[row] = ld;
        if (a.blp_out) {
            T s = a.blp_in ? a.blp_in[row] : T(0);
#pragma unroll
            for (int d = 0; d < Fam::DIM; ++d) if (d < a.dim) s += T(-0.5) * x[d] * x[d] - M<T>::HALF_LN_2PI;
            a.blp_out[row] = s;
        }
    }
    status_add(a.status, JF_STATUS_NONFINITE, active && (bad || ctx.nonfinite));
    status_add(a.status, JF_STATUS_OUT_OF_RANGE, active && ctx.oob);
    status_add(a.status, JF_STATUS_NONCONVERGED, active && ctx.nonconv);
    }                                                              // row tiles
}

}  // namespace jf
