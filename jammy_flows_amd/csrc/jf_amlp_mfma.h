// float64 low-rank AmortizableMLP + g layers on the f64 matrix cores (jf_amlp_gf_chain_inv_f64 when r1, r2 <= 8 and H % 16 == 0).
//
// Why a second kernel.  amlp_gf_kernel (amlp_gf_kernels.hip) regenerates every parameter as b2[j] + <U2[j, :], t2> with 8 scalar FMAs whose
// U2 operands come from LDS: 4 ds_read_b128 per parameter and lane, 612 per lane and layer loop, none of it shared between the rows of a wave.
// Measured by removing parts of that kernel (scripts/probe/amlp_parts.py, 2^19 rows of the C5 block): 0.84 of its 2.35 ms are these reads,
// another 0.45 ms the first two MLP stages, which read their weights from LDS the same way.  On MI355X the f64 MFMA rate equals the f64
// vector rate, so the matrix cores do not make the products faster -- but one v_mfma_f64_16x16x4 takes its A operand as ONE double per lane
// for 16 rows at a time: 16x less LDS traffic for the same arithmetic.
//
// Layout.  A wave owns 16 rows; lane = (row n = lane % 16, group q = lane / 16).  Everything is a chain of 16x16x4 products whose RESULT layout
// (lane (n, q), register r  <->  matrix row m = q + 4 r, column n) is at the same time the B-OPERAND layout of the next product
// (B[k = q][n]) -- no lane ever has to hand a value to another one:
//     t1^T  = V1 c^T                      (K1/4 steps; B = c[n][4 s + q] straight from HBM)        -> lane holds t1[q], t1[q + 4]
//     pre^T = U1 t1^T + b1                (8 unit tiles x 2 steps; B = t1[q] / t1[q + 4])           -> lane holds unit 16 t + 4 r + q
//     t2^T  = V2 tanh(pre)^T              (32 steps, step (t, r) covers units 16 t + 4 r + {0..3})  -> lane holds t2[q], t2[q + 4]
//     P^T   = U2' t2^T + b2'              (21 tiles per layer x 2 steps)
// U2' / b2' are U2 / b2 with their rows PERMUTED so that register r of tile tt of lane group q is exactly the parameter the flow arithmetic
// wants there: group q owns the coordinates q and q + 4 of its row; per layer 5 tiles carry the 8 reflections and the offset of both
// coordinates, 8 tiles the 30 mixture parameters of coordinate q, 8 tiles those of coordinate q + 4.  The permutation costs nothing: the
// workgroup gathers the weights into LDS in fragment order once (115 KB for C5, L2 hits), and every A operand is then one conflict-free
// ds_read_b64 (fragment f of a product = 64 consecutive doubles).  The flow arithmetic is jf_gf.h's, on registers; the three reductions over a
// row's coordinates (reflections, log-det) are two cross-group shuffles.
#pragma once
#include "jf_cond_regs.h"

namespace jf {

// threads per workgroup: the waves of a workgroup share ONE LDS copy of the weights (115 KB for C5: one workgroup per CU).  Log-prob direction and
// the MLP alone: 12 waves (165 VGPRs: three per SIMD; with 8 the kernel ran 1.43 instead of 1.34 ms per 2^19 rows).  Sampling direction:
// 8 waves (its two register-resident solver rows need 256 VGPRs).  The kernels take the count from blockDim.
constexpr int AM_THREADS = 768, AM_THREADS_FWD = 512;
__host__ __device__ constexpr int am_rows(int threads) { return threads / 64 * 16; }
constexpr int AM_TILES_R = 5, AM_TILES_M = 8, AM_TILES = AM_TILES_R + 2 * AM_TILES_M;     // per layer
constexpr int AM_R = 8;                          // rank bound (two K steps)

using f64x4_t = __attribute__((ext_vector_type(4))) double;

// parameter column (inside the chain's row) that register `reg` of tile `tt` of lane group `q` holds, -1 for padding / absent entries
template <typename L> __device__ __forceinline__ int am_col(const L& o, int D, int tt, int q, int reg) {
    const int slot = (tt < AM_TILES_R ? tt : (tt - AM_TILES_R) % AM_TILES_M) * 4 + reg;
    if (tt < AM_TILES_R) {                                              // reflections i of coordinates q (slots 0-7), q + 4 (8-15); offsets (16, 17)
        if (slot < 16) {
            const int i = slot & 7, d = q + 4 * (slot >> 3);
            return (i < o.hh && d < D) ? o.col0 + o.off_rot + i * D + d : -1;
        }
        if (slot < 18) {
            const int d = q + 4 * (slot - 16);
            return (o.model_offset && d < D) ? o.col0 + d : -1;
        }
        return -1;
    }
    const int d = q + 4 * ((tt - AM_TILES_R) / AM_TILES_M);
    if (slot >= 30 || d >= D) return -1;
    const int k = slot % 10, sec = slot / 10;
    return o.col0 + (sec == 0 ? o.off_mean : sec == 1 ? o.off_lw : o.off_ln) + k * D + d;
}

// reductions over the 4 lane groups of a row (lanes l, l^16, l^32, l^48) on the permlane swaps of gfx950, both dwords of the double (inline asm
// for the reason given at cs_rreduce, jf_cond_split.h).  Rounds 2-3 used __shfl_xor here: two ds_bpermute_b32 per step through the LDS pipe,
// ~70 of them per row tile and layer chain.
template <typename Op> __device__ __forceinline__ double am_xreduce(double v, Op op) {
    unsigned a0 = (unsigned)__double2loint(v), a1 = (unsigned)__double2hiint(v), b0 = a0, b1 = a1;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
    const double c = op(__hiloint2double((int)a1, (int)a0), __hiloint2double((int)b1, (int)b0));
    a0 = (unsigned)__double2loint(c); a1 = (unsigned)__double2hiint(c); b0 = a0; b1 = a1;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
    return op(__hiloint2double((int)a1, (int)a0), __hiloint2double((int)b1, (int)b0));
}
__device__ __forceinline__ double am_xsum(double v) { return am_xreduce(v, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ double am_xmax(double v) { return am_xreduce(v, [](double a, double b) { return fmax(a, b); }); }

// the MLP's own weights in fragment order + the rank-r2 vector t2 of the wave's 16 rows -- shared by the block kernel and the MLP-only kernel
template <typename T> struct AmMlp {
    T* fV1; T* fU1; T* sb1; T* fV2; T* ttab; T* next;      // ttab: tanh(k / 32) (jf_math.h: tanh_tab); next: first free LDS element behind the MLP image
    int k1s, HT; bool lowrank1;
};

template <typename Args> __device__ inline AmMlp<double> am_build_mlp(const Args& a, double* lds, int tid) {
    using T = double;
    AmMlp<T> I;
    const int H = a.H;
    I.HT = H / 16; I.k1s = (a.K1 + 3) / 4; I.lowrank1 = a.V1 != nullptr;
    I.fV1 = lds;                                                          // low-rank: k1s fragments; full: HT * k1s fragments of W1
    const int nV1 = (I.lowrank1 ? I.k1s : I.HT * I.k1s) * 64;
    I.fU1 = I.fV1 + nV1;                                                  // HT * 2 fragments (low-rank only)
    const int nU1 = I.lowrank1 ? I.HT * 2 * 64 : 0;
    I.sb1 = I.fU1 + nU1;                                                  // H
    I.fV2 = I.sb1 + H;                                                    // H / 4 fragments
    I.ttab = I.fV2 + (H / 4) * 64;
    I.next = I.ttab + JF_TANH_TAB_N;
    tanh_tab_load(I.ttab, tid, (int)blockDim.x);
    const T* W = I.lowrank1 ? a.V1 : a.U1;
    for (int e = tid; e < nV1; e += (int)blockDim.x) {
        const int f = e >> 6, l = e & 63, m = l & 15, k = 4 * (I.lowrank1 ? f : f % I.k1s) + (l >> 4);
        const int rowi = I.lowrank1 ? m : 16 * (f / I.k1s) + m;
        I.fV1[e] = (k < a.K1 && (I.lowrank1 ? m < a.r1 : true)) ? W[rowi * a.K1 + k] : T(0);
    }
    for (int e = tid; e < nU1; e += (int)blockDim.x) {
        const int f = e >> 6, l = e & 63, unit = 16 * (f >> 1) + (l & 15), k = 4 * (f & 1) + (l >> 4);
        I.fU1[e] = k < a.r1 ? a.U1[unit * a.r1 + k] : T(0);
    }
    for (int e = tid; e < H; e += (int)blockDim.x) I.sb1[e] = a.b1[e];
    for (int e = tid; e < (H / 4) * 64; e += (int)blockDim.x) {
        const int f = e >> 6, l = e & 63, m = l & 15, unit = 4 * f + (l >> 4);              // step f = 4 t + r covers units 16 t + 4 r + k
        I.fV2[e] = m < a.r2 ? a.V2[m * H + unit] : T(0);
    }
    return I;
}

// t2[q], t2[q + 4] of row n (lane = (n, q)); c = the row's conditioning inputs
template <typename Args> __device__ __forceinline__ void am_t2(const Args& a, const AmMlp<double>& I, int lane, const double* __restrict__ c, double& t2a,
                                                                double& t2b) {
    using T = double;
    const int q = lane >> 4;
    const f64x4_t zero4 = {0.0, 0.0, 0.0, 0.0};
    T hh[AG_HMAX / 16][4];                                                // unit 16 t + 4 r + q of row n
    if (I.lowrank1) {
        f64x4_t acc = zero4;
        for (int s = 0; s < I.k1s; ++s) {
            const int k = 4 * s + q;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(I.fV1[s * 64 + lane], k < a.K1 ? c[k] : T(0), acc, 0, 0, 0);
        }
        const T t1a = acc[0], t1b = acc[1];                               // t1[q], t1[q + 4]
#pragma unroll
        for (int t = 0; t < AG_HMAX / 16; ++t) {
            if (t < I.HT) {
                f64x4_t p = {I.sb1[16 * t + q], I.sb1[16 * t + 4 + q], I.sb1[16 * t + 8 + q], I.sb1[16 * t + 12 + q]};
                p = __builtin_amdgcn_mfma_f64_16x16x4f64(I.fU1[(2 * t) * 64 + lane], t1a, p, 0, 0, 0);
                p = __builtin_amdgcn_mfma_f64_16x16x4f64(I.fU1[(2 * t + 1) * 64 + lane], t1b, p, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) hh[t][r] = tanh_tab(I.ttab, p[r]);
            }
        }
    } else {
        T cin[AG_K1MAX / 4];
#pragma unroll
        for (int s = 0; s < AG_K1MAX / 4; ++s) cin[s] = (s < I.k1s && 4 * s + q < a.K1) ? c[4 * s + q] : T(0);
#pragma unroll
        for (int t = 0; t < AG_HMAX / 16; ++t) {
            if (t < I.HT) {
                f64x4_t p = {I.sb1[16 * t + q], I.sb1[16 * t + 4 + q], I.sb1[16 * t + 8 + q], I.sb1[16 * t + 12 + q]};
#pragma unroll
                for (int s = 0; s < AG_K1MAX / 4; ++s)
                    if (s < I.k1s) p = __builtin_amdgcn_mfma_f64_16x16x4f64(I.fV1[(t * I.k1s + s) * 64 + lane], cin[s], p, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) hh[t][r] = tanh_tab(I.ttab, p[r]);
            }
        }
    }
    f64x4_t acc = zero4;
#pragma unroll
    for (int t = 0; t < AG_HMAX / 16; ++t) {
        if (t < I.HT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(I.fV2[(4 * t + r) * 64 + lane], hh[t][r], acc, 0, 0, 0);
        }
    }
    t2a = acc[0]; t2b = acc[1];
}

// jf_amlp2_f64 on the matrix cores: out (B, N) = U2 t2 + b2, tiles of 16 output columns (no permutation)
template <typename Args>
__global__ void __launch_bounds__(AM_THREADS) amlp2_mfma_kernel(const Args a) {
    using T = double;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const AmMlp<T> I = am_build_mlp(a, reinterpret_cast<T*>(smem_raw), tid);
    const int NT = (a.N + 15) / 16;
    T* fU2 = I.next;
    T* sb2 = fU2 + NT * 2 * 64;
    for (int e = tid; e < NT * 2 * 64; e += (int)blockDim.x) {
        const int f = e >> 6, l = e & 63, j = 16 * (f >> 1) + (l & 15), k = 4 * (f & 1) + (l >> 4);
        fU2[e] = (j < a.N && k < a.r2) ? a.U2[(int64_t)j * a.r2 + k] : T(0);
    }
    for (int e = tid; e < NT * 16; e += (int)blockDim.x) sb2[e] = e < a.N ? a.b2[e] : T(0);
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * am_rows((int)blockDim.x) + wave * 16 + n;
    const bool row_valid = row < a.B;
    const int64_t rrow = row_valid ? row : a.B - 1;
    T t2a, t2b;
    am_t2(a, I, lane, a.in + rrow * a.in_stride, t2a, t2b);
    for (int t = 0; t < NT; ++t) {
        const T* b = sb2 + t * 16 + q;
        f64x4_t p = {b[0], b[4], b[8], b[12]};
        p = __builtin_amdgcn_mfma_f64_16x16x4f64(fU2[(2 * t) * 64 + lane], t2a, p, 0, 0, 0);
        p = __builtin_amdgcn_mfma_f64_16x16x4f64(fU2[(2 * t + 1) * 64 + lane], t2b, p, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + q + 4 * r;
            if (row_valid && j < a.N) a.params_out[row * a.pos + j] = p[r];
        }
    }
}

// FWD: the SAMPLING direction of the block (main/default.py:1420-1506 with gaussianization_flow.py:911-989, bisection_n_newton.py:11-135): layers
// first to last; per layer the two coordinates of a lane are solved one after the other on register-resident derived rows (cs_solve,
// jf_cond_regs.h: float32 bracket phase, float64 Newton phase), then the reflections in reverse and the offset.
template <typename Args, bool FWD = false>
__global__ void __launch_bounds__(FWD ? AM_THREADS_FWD : AM_THREADS) amlp_gf_mfma_kernel(const Args a) {
    using T = double;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int D = a.D;
    const AmMlp<T> I = am_build_mlp(a, reinterpret_cast<T*>(smem_raw), tid);
    T* fU2 = I.next;                                                      // n_layers * AM_TILES * 2 fragments
    T* sb2 = fU2 + a.n_layers * AM_TILES * 2 * 64;                        // n_layers * AM_TILES * 16
    {
        const int nU2 = a.n_layers * AM_TILES * 2 * 64;
        for (int e = tid; e < nU2; e += (int)blockDim.x) {
            const int f = e >> 6, l = e & 63, m = l & 15, k = 4 * (f & 1) + (l >> 4);
            const int tl = f >> 1, layer = tl / AM_TILES, tt = tl % AM_TILES;
            const int col = am_col(a.L[layer], D, tt, m & 3, m >> 2);
            fU2[e] = (col >= 0 && k < a.r2) ? a.U2[(int64_t)col * a.r2 + k] : T(0);
        }
        const int nb2 = a.n_layers * AM_TILES * 16;
        for (int e = tid; e < nb2; e += (int)blockDim.x) {
            const int tl = e >> 4, m = e & 15, layer = tl / AM_TILES, tt = tl % AM_TILES;
            const int col = am_col(a.L[layer], D, tt, m & 3, m >> 2);
            sb2[e] = col >= 0 ? a.b2[col] : T(0);
        }
    }
    __syncthreads();

    const int64_t row = (int64_t)blockIdx.x * am_rows((int)blockDim.x) + wave * 16 + n;
    const bool row_valid = row < a.B;
    const int64_t rrow = row_valid ? row : a.B - 1;
    T t2a, t2b;
    am_t2(a, I, lane, a.in + rrow * a.in_stride, t2a, t2b);

    // ---- flow: group q owns coordinates q and q + 4 of row n
    const bool v0 = q < D, v1 = q + 4 < D;
    T x0 = v0 ? a.x[rrow * a.xs + q] : T(0);
    T x1 = v1 ? a.x[rrow * a.xs + q + 4] : T(0);
    T ld = a.ld_in ? a.ld_in[rrow] : T(0);
    auto tile = [&](int layer, int tt) -> f64x4_t {                       // registers r = 0..3 of tile tt: parameters am_col(layer, tt, q, r)
        const int tl = layer * AM_TILES + tt;
        const T* b = sb2 + tl * 16 + q;
        f64x4_t p = {b[0], b[4], b[8], b[12]};
        p = __builtin_amdgcn_mfma_f64_16x16x4f64(fU2[(2 * tl) * 64 + lane], t2a, p, 0, 0, 0);
        return __builtin_amdgcn_mfma_f64_16x16x4f64(fU2[(2 * tl + 1) * 64 + lane], t2b, p, 0, 0, 0);
    };
    if constexpr (FWD) {
        for (int l = 0; l < a.n_layers; ++l) {
            const auto o = a.L[l];
            T logd = T(0);
            CsSolveInfo inf[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                T R[CS_SLOTS];
#pragma unroll
                for (int k = 0; k < CS_SLOTS; ++k) R[k] = T(0);
#pragma unroll
                for (int tt = 0; tt < AM_TILES_M; ++tt) {
                    const f64x4_t p = tile(l, AM_TILES_R + half * AM_TILES_M + tt);
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (4 * tt + r < 30) R[4 * tt + r] = p[r];
                }
                // regulate once: log-width slots -> 1 / width, log-weight slots -> normalised weight (ag_mixture's formulas)
                T Nn = T(0);
#pragma unroll
                for (int k = 0; k < CS_K; ++k) {
                    const T ae = o.inv_wmax + M<T>::exp_fast(-R[CS_SLOT_LW + k]);
                    R[CS_SLOT_LW + k] = ae * M<T>::rcp(o.wmin * ae + T(1));
                    const T w = o.nmin + o.nmax * M<T>::rcp(T(1) + M<T>::exp_fast(-R[CS_SLOT_LN + k]));
                    R[CS_SLOT_LN + k] = w;
                    Nn += w;
                }
                const T invN = M<T>::rcp(Nn);
#pragma unroll
                for (int k = 0; k < CS_K; ++k) R[CS_SLOT_LN + k] *= invN;
                const bool live = half == 0 ? v0 : v1;
                T slogd;
                const T xs = cs_solve<T>(R, o.inv_type, live, half == 0 ? x0 : x1, row_valid, q == 0, a.status, [](T v) { return am_xsum(v); },
                                        [](T v) { return am_xmax(v); }, &inf[half], &slogd);
                logd += live ? slogd : T(0);
                if (half == 0) x0 = xs; else x1 = xs;
            }
            ld -= am_xsum(logd);
            // the row's status: Newton row-steps = the longer of its two solves
            const int steps = inf[0].steps > inf[1].steps ? inf[0].steps : inf[1].steps;
            for (int i = 0; i < 20; ++i) status_add(a.status, JF_STATUS_NEWTON_STEPS, row_valid && q == 0 && i < steps);
            status_add(a.status, JF_STATUS_NONCONVERGED, row_valid && q == 0 && (inf[0].nonconv || inf[1].nonconv));
            status_add(a.status, JF_STATUS_NONFINITE, row_valid && q == 0 && (inf[0].nonfinite || inf[1].nonfinite));
            {
                T R[AM_TILES_R * 4];
#pragma unroll
                for (int tt = 0; tt < AM_TILES_R; ++tt) {
                    const f64x4_t p = tile(l, tt);
#pragma unroll
                    for (int r = 0; r < 4; ++r) R[4 * tt + r] = p[r];
                }
#pragma unroll
                for (int i = AG_HH - 1; i >= 0; --i) {
                    if (i < o.hh) {                                       // x <- Q x: the reflections of the log-prob direction in reverse
                        const T va = R[i], vb = R[8 + i];
                        const T n2 = am_xsum(va * va + vb * vb), dot = am_xsum(va * x0 + vb * x1);
                        const T f = T(2) * dot / n2;
                        x0 -= f * va; x1 -= f * vb;
                    }
                }
                x0 += R[16]; x1 += R[17];                                 // euclidean_base.py:63-68
            }
        }
    } else
    for (int l = a.n_layers - 1; l >= 0; --l) {
        const auto o = a.L[l];
        {
            T R[AM_TILES_R * 4];
#pragma unroll
            for (int tt = 0; tt < AM_TILES_R; ++tt) {
                const f64x4_t p = tile(l, tt);
#pragma unroll
                for (int r = 0; r < 4; ++r) R[4 * tt + r] = p[r];
            }
            x0 -= R[16]; x1 -= R[17];                                     // euclidean_base.py:40-45 (zero rows when the layer models no offset)
#pragma unroll
            for (int i = 0; i < AG_HH; ++i) {
                if (i < o.hh) {                                           // x <- Q^T x (gaussianization_flow.py:1038), H_i = I - 2 v v^T / |v|^2
                    const T va = R[i], vb = R[8 + i];
                    const T n2 = am_xsum(va * va + vb * vb), dot = am_xsum(va * x0 + vb * x1);
                    const T f = T(2) * dot / n2;
                    x0 -= f * va; x1 -= f * vb;
                }
            }
        }
        T logd = T(0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            T P[AG_SLOTS];
#pragma unroll
            for (int tt = 0; tt < AM_TILES_M; ++tt) {
                const f64x4_t p = tile(l, AM_TILES_R + half * AM_TILES_M + tt);
#pragma unroll
                for (int r = 0; r < 4; ++r) if (4 * tt + r < 30) P[4 * tt + r] = p[r];
            }
            const bool live = half == 0 ? v0 : v1;
            const MixQ<T> mq = ag_mixture<T>(P, o, half == 0 ? x0 : x1, live);
            const IcdfOut<T> s = gf_icdf<T>(o.inv_type, mq);
            if (half == 0) x0 = s.y; else x1 = s.y;
            logd += live ? s.logd : T(0);
        }
        ld += am_xsum(logd);
    }
    if (row_valid && v0) a.x_out[row * a.xos + q] = x0;
    if (row_valid && v1) a.x_out[row * a.xos + q + 4] = x1;
    T sb = T(0);
    if (!FWD && a.blp_out) sb = am_xsum((v0 ? T(-0.5) * x0 * x0 - M<T>::HALF_LN_2PI : T(0)) + (v1 ? T(-0.5) * x1 * x1 - M<T>::HALF_LN_2PI : T(0)));
    if (row_valid && q == 0) {
        a.ld_out[row] = ld;
        if (a.blp_out) a.blp_out[row] = sb + (a.blp_in ? a.blp_in[row] : T(0));
    }
    const bool badx = (v0 && !M<T>::finite(x0)) || (v1 && !M<T>::finite(x1));
    const T bad = am_xsum(badx ? T(1) : T(0));
    status_add(a.status, JF_STATUS_NONFINITE, row_valid && q == 0 && (bad > T(0) || !M<T>::finite(ld)));
}

// LDS doubles of the fragment image: the MLP's own part + the last stage (block kernel: permuted tiles per layer; MLP only: ceil(N / 16) tiles)
inline size_t am_lds_mlp(int K1, int H, bool lowrank1) {
    const int k1s = (K1 + 3) / 4, HT = H / 16;
    return (size_t)(lowrank1 ? k1s : HT * k1s) * 64 + (lowrank1 ? (size_t)HT * 2 * 64 : 0) + H + (size_t)(H / 4) * 64 + JF_TANH_TAB_N;
}
inline size_t am_lds_doubles(int K1, int H, bool lowrank1, int n_layers) {
    return am_lds_mlp(K1, H, lowrank1) + (size_t)n_layers * AM_TILES * 2 * 64 + (size_t)n_layers * AM_TILES * 16;
}
inline size_t am_lds_mlp_only(int K1, int H, bool lowrank1, int N) {
    return am_lds_mlp(K1, H, lowrank1) + (size_t)((N + 15) / 16) * (2 * 64 + 16);
}

}  // namespace jf
