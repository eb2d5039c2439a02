// Training: the head of a two-stage low-rank AmortizableMLP -- everything in front of the last U product,
//     t1 = V1 c,   h = tanh(U1 t1 + b1),   t2 = V2 h          (amortizable_mlp.py:508-578; K1 <= 32, H <= 128 and a multiple of 16, ranks <= 8)
// -- forward in ONE launch (jf_lowrank_head_f64: the MFMA chain of jf_amlp_mfma.h, which also leaves t1 and h behind for the backward) and
// backward in ONE launch + a reduction (jf_lowrank_head_bwd_f64).  Round 3 / the first half of round 4 ran this part of a C5 training step
// as 3 + 8 dense / elementwise launches, each bound by HBM on a (B, 128) float64 activation (0.8 of the 2.83 ms at 2^17 rows).
//
// Backward, per wave of 16 rows in the layout of jf_amlp_mfma.h (lane = (row n, group q); a product's result layout is the next product's
// B operand), one hidden tile of 16 units at a time, nothing but h read from and (optionally) g_c written to HBM:
//     g_h^T tile  = V2^T g_t2^T                    2 products   (B = g_t2[q], g_t2[q + 4] of row n)
//     g_pre       = g_h (1 - h^2)                  registers: unit 16 t + 4 r + q of row n
//     g_t1^T     += U1^T g_pre^T                   4 products   (B = g_pre's registers)
//     g_U1 | g_b1 tile += g_pre^T [t1 | 1]         4 products over the wave's 16 rows (g_pre transposed through 2 KB of LDS scratch)
//     g_V2^T tile += h^T g_t2                      4 products over the rows (h transposed the same way)
// and after the eight tiles  g_c^T = V1^T g_t1^T (2 products per 16 inputs) and  g_V1 += g_t1^T c  (4 per 16 inputs).  The three weight
// gradients of U1, b1, V2 accumulate in MFMA result registers over all row tiles of the wave (128 VGPRs) and meet in LDS once per wave (the small
// g_V1 tile goes there per row tile); one partial image
// per workgroup, summed in a fixed order by lrm_reduce_kernel.
#pragma once

namespace jf {

constexpr int LRM_NW = 4;                          // waves of a backward workgroup (~235 VGPRs: two waves per SIMD, two workgroups per CU)
constexpr int LRM_MAX_WG = 512;
constexpr int LRM_HT = AG_HMAX / 16, LRM_KT = AG_K1MAX / 16;
__host__ __device__ constexpr int lrm_psz(int H, int KT) { return H * AM_R + H * (AM_R + 1) + AM_R * 16 * KT; }

struct LrmFwdArgs {
    const double* in; int64_t in_stride;
    const double* V1; const double* U1; const double* b1; const double* V2;
    int K1, H, r1, r2;
    int64_t B;
    double* t1; double* h; double* t2;               // (B, 8), (B, H), (B, 8)
};

struct LrmBwdArgs {
    const double* in; int64_t in_stride;
    const double* V1; const double* U1; const double* V2;
    int K1, H, r1, r2;
    int64_t B, n_row_tiles;
    const double* t1; const double* h;
    const double* g_t2; int64_t gs;
    double* g_c; int64_t gcs;                        // nullable
    double* partial;
};

// t2 of the wave's rows with the intermediate values stored: am_t2 (jf_amlp_mfma.h) restated with the two stores (a kernel of its own so that the
// inference kernels' register allocation does not move)
__global__ void __launch_bounds__(AM_THREADS) lrm_head_fwd_kernel(const LrmFwdArgs a) {
    using T = double;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const AmMlp<T> I = am_build_mlp(a, reinterpret_cast<T*>(smem_raw), tid);
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * am_rows((int)blockDim.x) + wave * 16 + n;
    const bool row_valid = row < a.B;
    const int64_t rrow = row_valid ? row : a.B - 1;
    const T* c = a.in + rrow * a.in_stride;
    const f64x4_t zero4 = {0.0, 0.0, 0.0, 0.0};
    f64x4_t acc = zero4;
    for (int s = 0; s < I.k1s; ++s) {
        const int k = 4 * s + q;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(I.fV1[s * 64 + lane], k < a.K1 ? c[k] : T(0), acc, 0, 0, 0);
    }
    const T t1a = acc[0], t1b = acc[1];
    if (row_valid) { a.t1[row * AM_R + q] = t1a; a.t1[row * AM_R + q + 4] = t1b; }
    f64x4_t t2 = zero4;
#pragma unroll
    for (int t = 0; t < AG_HMAX / 16; ++t) {
        if (t < I.HT) {
            f64x4_t p = {I.sb1[16 * t + q], I.sb1[16 * t + 4 + q], I.sb1[16 * t + 8 + q], I.sb1[16 * t + 12 + q]};
            p = __builtin_amdgcn_mfma_f64_16x16x4f64(I.fU1[(2 * t) * 64 + lane], t1a, p, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f64_16x16x4f64(I.fU1[(2 * t + 1) * 64 + lane], t1b, p, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const T hv = tanh_tab(I.ttab, p[r]);
                if (row_valid) a.h[row * a.H + 16 * t + 4 * r + q] = hv;
                t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(I.fV2[(4 * t + r) * 64 + lane], hv, t2, 0, 0, 0);
            }
        }
    }
    if (row_valid) { a.t2[row * AM_R + q] = t2[0]; a.t2[row * AM_R + q + 4] = t2[1]; }
}

// KTM: compiled bound of the 16-input tiles (1 for K1 <= 16, 2 up to 32)
template <int KTM> __global__ void __launch_bounds__(LRM_NW * 64) lrm_head_bwd_kernel(const LrmBwdArgs a) {
    using T = double;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int H = a.H, HT = H / 16, KT = (a.K1 + 15) / 16;
    T* fV2T = reinterpret_cast<T*>(smem_raw);            // HT tiles x 2 steps: lane (m, kq), step s: V2[4 s + kq][16 t + m]
    T* fU1T = fV2T + HT * 2 * 64;                        // H / 4 steps f = 4 t + r: lane (m = rank, kq): U1[16 t + 4 r + kq][m]
    T* fV1T = fU1T + (H / 4) * 64;                       // KT tiles x 2 steps: lane (m, kq), step s: V1[4 s + kq][16 i + m]
    T* acc = fV1T + KT * 2 * 64;                         // partial image: g_V2^T [unit][8] | g_U1|g_b1 [unit][9] | g_V1 [rank][16 KT]
    const int psz = lrm_psz(H, KT);
    T* scrA = acc + psz + wave * (2 * 16 * 17);
    T* scrB = scrA + 16 * 17;
    for (int e = tid; e < HT * 2 * 64; e += LRM_NW * 64) {
        const int f = e >> 6, l = e & 63, m = l & 15, k = 4 * (f & 1) + (l >> 4), t = f >> 1;
        fV2T[e] = k < a.r2 ? a.V2[(int64_t)k * H + 16 * t + m] : T(0);
    }
    for (int e = tid; e < (H / 4) * 64; e += LRM_NW * 64) {
        const int f = e >> 6, l = e & 63, m = l & 15, unit = 4 * f + (l >> 4);
        fU1T[e] = m < a.r1 ? a.U1[(int64_t)unit * a.r1 + m] : T(0);
    }
    for (int e = tid; e < KT * 2 * 64; e += LRM_NW * 64) {
        const int f = e >> 6, l = e & 63, m = l & 15, k = 4 * (f & 1) + (l >> 4), i = 16 * (f >> 1) + m;
        fV1T[e] = (k < a.r1 && i < a.K1) ? a.V1[(int64_t)k * a.K1 + i] : T(0);
    }
    for (int e = tid; e < psz; e += LRM_NW * 64) acc[e] = T(0);
    __syncthreads();
    const f64x4_t zero4 = {0.0, 0.0, 0.0, 0.0};
    f64x4_t aU[LRM_HT], aV[LRM_HT];
#pragma unroll
    for (int t = 0; t < LRM_HT; ++t) { aU[t] = zero4; aV[t] = zero4; }
    for (int64_t rt = (int64_t)blockIdx.x * LRM_NW + wave; rt < a.n_row_tiles; rt += (int64_t)gridDim.x * LRM_NW) {
        const int64_t row = rt * 16 + n;
        const bool row_valid = row < a.B;
        const int64_t rrow = row_valid ? row : a.B - 1;
        const T ga = (row_valid && q < a.r2) ? a.g_t2[rrow * a.gs + q] : T(0);
        const T gb = (row_valid && q + 4 < a.r2) ? a.g_t2[rrow * a.gs + q + 4] : T(0);
        T g2op[4], t1op[4];                             // B operands of the products over the rows: lane (j = n, k = q) of step s: row 4 s + k
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t rr = rt * 16 + 4 * s + q;
            const bool okr = rr < a.B;
            g2op[s] = (okr && n < a.r2) ? a.g_t2[rr * a.gs + n] : T(0);
            const T tv = (okr && n < a.r1) ? a.t1[rr * AM_R + n] : T(0);
            t1op[s] = (okr && n == a.r1) ? T(1) : tv;   // column r1: the bias
        }
        f64x4_t gt1 = zero4;
#pragma unroll
        for (int t = 0; t < LRM_HT; ++t) {
            if (t < HT) {
                f64x4_t gh = __builtin_amdgcn_mfma_f64_16x16x4f64(fV2T[(2 * t) * 64 + lane], ga, zero4, 0, 0, 0);
                gh = __builtin_amdgcn_mfma_f64_16x16x4f64(fV2T[(2 * t + 1) * 64 + lane], gb, gh, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const T hv = row_valid ? a.h[row * H + 16 * t + 4 * r + q] : T(0);
                    const T gp = gh[r] * (T(1) - hv * hv);
                    gt1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fU1T[(4 * t + r) * 64 + lane], gp, gt1, 0, 0, 0);
                    scrA[(q + 4 * r) * 17 + n] = gp;
                    scrB[(q + 4 * r) * 17 + n] = hv;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    aU[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(scrA[n * 17 + 4 * s + q], t1op[s], aU[t], 0, 0, 0);
                    aV[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(scrB[n * 17 + 4 * s + q], g2op[s], aV[t], 0, 0, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        // g_c^T = V1^T g_t1^T; g_V1 += g_t1^T c (g_t1 transposed through the scratch tile: ranks q, q + 4 of row n; rows 8..15 zero)
        scrA[q * 17 + n] = gt1[0]; scrA[(q + 4) * 17 + n] = gt1[1]; scrA[(q + 8) * 17 + n] = T(0); scrA[(q + 12) * 17 + n] = T(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < KTM; ++i) {
            if (i < KT) {
                if (a.g_c) {
                    f64x4_t gc = __builtin_amdgcn_mfma_f64_16x16x4f64(fV1T[(2 * i) * 64 + lane], gt1[0], zero4, 0, 0, 0);
                    gc = __builtin_amdgcn_mfma_f64_16x16x4f64(fV1T[(2 * i + 1) * 64 + lane], gt1[1], gc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = 16 * i + q + 4 * r;
                        if (row_valid && j < a.K1) a.g_c[row * a.gcs + j] = gc[r];
                    }
                }
                f64x4_t aw = zero4;                        // g_V1 tile of this row tile: straight into the workgroup's image (2 atomics per tile)
#pragma unroll
                for (int s = 0; s < 4; ++s) {              // (the inputs are loaded here, not ahead of the hidden tiles: 16 registers less across them)
                    const int64_t rr = rt * 16 + 4 * s + q;
                    const T cv = (rr < a.B && 16 * i + n < a.K1) ? a.in[rr * a.in_stride + 16 * i + n] : T(0);
                    aw = __builtin_amdgcn_mfma_f64_16x16x4f64(scrA[n * 17 + 4 * s + q], cv, aw, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) atomicAdd(acc + H * AM_R + H * (AM_R + 1) + (q + 4 * r) * (16 * KT) + 16 * i + n, aw[r]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // the wave's accumulators -> the workgroup's image (result layout: lane (n = column j, q), register r = row q + 4 r of the product)
    T* accV = acc; T* accU = acc + H * AM_R;
#pragma unroll
    for (int t = 0; t < LRM_HT; ++t) {
        if (t < HT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int unit = 16 * t + q + 4 * r;
                if (n < AM_R) atomicAdd(accV + unit * AM_R + n, aV[t][r]);
                if (n <= AM_R) atomicAdd(accU + unit * (AM_R + 1) + n, aU[t][r]);
            }
        }
    }
    __syncthreads();
    T* out = a.partial + (int64_t)blockIdx.x * psz;
    for (int e = tid; e < psz; e += LRM_NW * 64) out[e] = acc[e];
}

struct LrmReduceArgs {
    const double* partial; int n_wg, H, K1, r1, r2;
    double* g_V1; double* g_U1; double* g_b1; double* g_V2;
};
__global__ void __launch_bounds__(256) lrm_reduce_kernel(const LrmReduceArgs a) {
    __shared__ double part[4][64];
    const int KT = (a.K1 + 15) / 16, psz = lrm_psz(a.H, KT);
    const int le = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + le;
    const bool in = e < psz;
    const double* p = a.partial + (in ? e : 0);
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (in) {
        int g = sl;
        for (; g + 28 < a.n_wg; g += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += p[(int64_t)(g + 4 * u) * psz];
        }
        for (; g < a.n_wg; g += 4) s[0] += p[(int64_t)g * psz];
    }
    part[sl][le] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (sl != 0 || !in) return;
    const double sum = (part[0][le] + part[1][le]) + (part[2][le] + part[3][le]);
    const int nV = a.H * AM_R, nU = a.H * (AM_R + 1);
    if (e < nV) {
        const int unit = e / AM_R, k = e % AM_R;
        if (k < a.r2) a.g_V2[(int64_t)k * a.H + unit] = sum;
    } else if (e < nV + nU) {
        const int w = e - nV, unit = w / (AM_R + 1), k = w % (AM_R + 1);
        if (k < a.r1) a.g_U1[(int64_t)unit * a.r1 + k] = sum;
        else if (k == a.r1) a.g_b1[unit] = sum;
    } else {
        const int w = e - nV - nU, k = w / (16 * KT), i = w % (16 * KT);
        if (k < a.r1 && i < a.K1) a.g_V1[(int64_t)k * a.K1 + i] = sum;
    }
}

inline int lrm_n_wg(int64_t B) {
    const int64_t tiles = (B + 15) / 16, wgs = (tiles + LRM_NW - 1) / LRM_NW;
    return (int)(wgs < LRM_MAX_WG ? (wgs < 1 ? 1 : wgs) : LRM_MAX_WG);
}

}  // namespace jf
