// Small row-wise helpers of the pdf orchestrator.
//   jf_normal_logp_*: out[b] = (in ? in[b] : 0) + sum_d N(0,1).log_prob(z[b,d])      (jammy_flows/main/default.py:1110-1115, 1657, 1670)
//   jf_conditioning_rows_*: the input rows of the amortisation MLPs, cat[conditional_input, embed(x_0), embed(x_1), ...]
//                            (main/default.py:946-962; embed = identity / (cos, sin) / (x, y, z): sphere_base.py:305-332, 786-794), one launch
#include "jf_cond_in.h"

namespace jf {

struct CondSegs { int n; jf_cond_segment s[JF_MAX_SEGMENTS]; };

// one thread per OUTPUT element (coalesced stores; a thread-per-row version wrote 24 strided doubles per thread and ran at 0.3 TB/s).  A
// workgroup owns 64 whole rows and walks their 64 W elements with 32-bit index arithmetic (round 2 divided a 64-bit global element index by
// W per thread: ~60 vector instructions, 90 % VALU busy for a copy kernel).
constexpr int COND_ROWS = 64;
template <typename T>
__global__ void __launch_bounds__(256) conditioning_kernel(const CondSegs a, int64_t B, int W, T* __restrict__ out, int64_t os) {
    const int64_t row0 = (int64_t)blockIdx.x * COND_ROWS;
    const unsigned n_rows = (unsigned)(B - row0 < COND_ROWS ? B - row0 : COND_ROWS);
    const unsigned n = n_rows * (unsigned)W;
    for (unsigned e = threadIdx.x; e < n; e += 256) {
        const unsigned rl = e / (unsigned)W;
        int col = (int)(e - rl * (unsigned)W);
        const int out_col = col;
        const int64_t row = row0 + rl;
        T val = T(0);
        for (int i = 0; i < a.n; ++i) {
            const jf_cond_segment g = a.s[i];
            const int w = g.kind == 0 ? g.n_in : g.kind + 1;
            if (col < w) {
                val = cond_seg_value<T>(g, row, col);
                break;
            }
            col -= w;
        }
        out[row * os + out_col] = val;
    }
}

template <typename T> static int conditioning_rows(const jf_cond_segment* segs, int32_t n, int64_t B, T* out, int64_t os, void* stream) {
    if (!segs || !out || n < 1 || n > JF_MAX_SEGMENTS || B < 0) return JF_ERR_BADARG;
    CondSegs a{};
    a.n = n;
    for (int i = 0; i < n; ++i) {
        if (!segs[i].src || segs[i].kind < 0 || segs[i].kind > 2 || segs[i].n_in < 0) return JF_ERR_BADARG;
        a.s[i] = segs[i];
    }
    if (B == 0) return JF_OK;
    int W = 0;
    for (int i = 0; i < n; ++i) W += segs[i].kind == 0 ? segs[i].n_in : segs[i].kind + 1;
    if (W == 0) return JF_OK;
    jf::launch(conditioning_kernel<T>, dim3((unsigned)((B + COND_ROWS - 1) / COND_ROWS)), dim3(256), 0, (hipStream_t)stream, a, B, W, out, os);
    return check_launch();
}

template <typename T>
__global__ void __launch_bounds__(256) normal_logp_kernel(const T* __restrict__ z, int64_t zs, int64_t B, int D, const T* __restrict__ in,
                                                          T* __restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    T s = in ? in[row] : T(0);
    const T* r = z + row * zs;
    for (int d = 0; d < D; ++d) s += T(-0.5) * r[d] * r[d] - M<T>::HALF_LN_2PI;
    out[row] = s;
}

// ---- coverage: histogram of 2 (log N(0) - log_prob_base) over ascending chi^2 quantile thresholds (helper_fns/coverage.py:45-65)
constexpr int COV_MAX_T = 1024;
template <typename T>
__global__ void __launch_bounds__(256) coverage_hist_kernel(const T* __restrict__ lpb, int64_t B, T log_at_zero, const T* __restrict__ thr, int n,
                                                            unsigned long long* __restrict__ hist, T* __restrict__ twice_out) {
    __shared__ unsigned int h[COV_MAX_T + 1];
    __shared__ T th[COV_MAX_T];
    for (int i = threadIdx.x; i <= n; i += 256) h[i] = 0u;
    for (int i = threadIdx.x; i < n; i += 256) th[i] = thr[i];
    __syncthreads();
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < B; row += (int64_t)gridDim.x * 256) {
        const T tw = T(2) * (log_at_zero - lpb[row]);
        if (twice_out) twice_out[row] = tw;
        int lo = 0, hi = n;                                  // first index with tw < th[idx] (n: none; NaN lands there too)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (tw < th[mid]) hi = mid; else lo = mid + 1; }
        atomicAdd(&h[(tw == tw) ? lo : n], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= n; i += 256) if (h[i]) atomicAdd(hist + i, (unsigned long long)h[i]);
}
template <typename T> static int coverage_hist(const T* lpb, int64_t B, double log_at_zero, const T* thr, int32_t n, int64_t* hist, T* twice, void* stream) {
    if (!lpb || !thr || !hist || n < 1 || B < 0) return JF_ERR_BADARG;
    if (n > COV_MAX_T) return JF_ERR_UNSUPPORTED;
    if (B == 0) return JF_OK;
    int64_t blocks = (B + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    jf::launch(coverage_hist_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lpb, B, (T)log_at_zero, thr, (int)n,
                       reinterpret_cast<unsigned long long*>(hist), twice);
    return check_launch();
}

// ---- entropy reductions: out[g] = -mean_s in[g, s]  (mode 0)   or   log-mean-exp_s in[g, s]  (mode 1: logsumexp - log S)
template <typename T>
__global__ void __launch_bounds__(64) segment_reduce_kernel(const T* __restrict__ in, int64_t n_seg, int64_t seg_len, int mode, T* __restrict__ out) {
    const int64_t g = blockIdx.x;
    if (g >= n_seg) return;
    const T* p = in + g * seg_len;
    const int lane = threadIdx.x;
    if (mode == 0) {
        T s = T(0);
        for (int64_t i = lane; i < seg_len; i += 64) s += p[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) out[g] = -s / T(seg_len);
    } else {
        T m = -INFINITY;
        for (int64_t i = lane; i < seg_len; i += 64) m = M<T>::max(m, p[i]);
        for (int off = 32; off > 0; off >>= 1) m = M<T>::max(m, __shfl_xor(m, off, 64));
        T s = T(0);
        for (int64_t i = lane; i < seg_len; i += 64) s += M<T>::exp(p[i] - m);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) out[g] = m + M<T>::log(s) - M<T>::log(T(seg_len));
    }
}
template <typename T> static int segment_reduce(const T* in, int64_t n_seg, int64_t seg_len, int32_t mode, T* out, void* stream) {
    if (!in || !out || n_seg < 0 || seg_len < 1 || mode < 0 || mode > 1) return JF_ERR_BADARG;
    if (n_seg == 0) return JF_OK;
    jf::launch(segment_reduce_kernel<T>, dim3((unsigned)n_seg), dim3(64), 0, (hipStream_t)stream, in, n_seg, seg_len, (int)mode, out);
    return check_launch();
}

template <typename T> static int normal_logp(const T* z, int64_t zs, int64_t B, int32_t D, const T* in, T* out, void* stream) {
    if (!z || !out || D < 0 || B < 0) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    jf::launch(normal_logp_kernel<T>, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, zs, B, (int)D, in, out);
    return check_launch();
}

// g (1 - y^2), 16 bytes per lane and access
template <typename T>
__global__ void __launch_bounds__(256) tanh_bwd_kernel(const T* __restrict__ g, const T* __restrict__ y, int64_t n, T* __restrict__ out, bool vec) {
    constexpr int N = Vec16<T>::N;
    using V = typename Vec16<T>::type;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * N;
    if (i >= n) return;
    if (vec && i + N <= n) {
        const V gv = *reinterpret_cast<const V*>(g + i), yv = *reinterpret_cast<const V*>(y + i);
        V o;
        const T* gp = reinterpret_cast<const T*>(&gv); const T* yp = reinterpret_cast<const T*>(&yv); T* op = reinterpret_cast<T*>(&o);
#pragma unroll
        for (int j = 0; j < N; ++j) op[j] = gp[j] * (T(1) - yp[j] * yp[j]);
        *reinterpret_cast<V*>(out + i) = o;
    } else {
        for (int64_t j = i; j < n && j < i + N; ++j) out[j] = g[j] * (T(1) - y[j] * y[j]);
    }
}
// Sum of partial slabs: out_a[c * na + i] = sum over the slabs s of chunk c of a[s * na + i] (chunk c = slabs [c * chunk, (c + 1) * chunk)), the
// same for b in the same launch (weights and bias of one layer).  The batch-reducing backward kernels leave one slab per workgroup or batch
// split (up to 4096); a workgroup owns 32 consecutive elements x one chunk, its 8 slab lanes walk the chunk 8 slabs apart and meet in LDS in a
// FIXED order, so the result does not depend on scheduling (unlike atomics).  Two launches (chunked, then chunk = all) reduce thousands of slabs
// with every CU busy; few slabs take one.
// map (the launch that produces the totals only: one chunk): element i of a / element na + i of b goes to out_a[map[...]] (negative: dropped)
// -- the totals land in the layout their consumer wants (a weight gradient transposed, bias columns split off a slab, packed parameter rows
// back in natural order) instead of being re-laid by copy / gather launches of ~5 us each afterwards.
template <typename T>
__global__ void __launch_bounds__(256) slab_sum_kernel(const T* __restrict__ a, int64_t na, T* __restrict__ out_a, const T* __restrict__ b, int64_t nb,
                                                       T* __restrict__ out_b, int S, int chunk, const int32_t* __restrict__ map) {
    __shared__ T part[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + tx;
    const int c = blockIdx.y;
    const int s0 = c * chunk, s1 = s0 + chunk < S ? s0 + chunk : S;
    const T* src = nullptr; T* dst = nullptr; int64_t n = 0, j = 0;
    if (i < na) { src = a; dst = out_a; n = na; j = i; }
    else if (i < na + nb) { src = b; dst = out_b; n = nb; j = i - na; }
    T acc[4] = {T(0), T(0), T(0), T(0)};
    if (src) {
        int sl = s0 + ty;
        for (; sl + 24 < s1; sl += 32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += src[(int64_t)(sl + 8 * u) * n + j];
        }
        for (; sl < s1; sl += 8) acc[0] += src[(int64_t)sl * n + j];
    }
    part[ty][tx] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (ty == 0 && src) {
        T t = part[0][tx];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += part[u][tx];
        if (map) {
            const int32_t m = map[i];
            if (m >= 0) out_a[m] = t;
        } else {
            dst[(int64_t)c * n + j] = t;
        }
    }
}
template <typename T> static int slab_sum(const T* a, int64_t na, T* out_a, const T* b, int64_t nb, T* out_b, int32_t S, int32_t chunk, void* stream,
                                          const int32_t* map = nullptr) {
    if (!a || !out_a || na < 1 || nb < 0 || (nb > 0 && (!b || (!out_b && !map))) || S < 1 || chunk < 1 || na > JF_MAX_ROWS || nb > JF_MAX_ROWS) return JF_ERR_BADARG;
    const int64_t chunks = ((int64_t)S + chunk - 1) / chunk;
    if (chunks > 65535) return JF_ERR_UNSUPPORTED;
    if (map && chunks != 1) return JF_ERR_BADARG;
    jf::launch(slab_sum_kernel<T>, dim3((unsigned)((na + nb + 31) / 32), (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, a, na, out_a, b, nb,
                       out_b, (int)S, (int)chunk, map);
    return hipPeekAtLastError() == hipSuccess ? JF_OK : JF_ERR_LAUNCH;
}

template <typename T> static int tanh_bwd(const T* g, const T* y, int64_t n, T* out, void* stream) {
    if (!g || !y || !out || n < 0) return JF_ERR_BADARG;
    if (n == 0) return JF_OK;
    constexpr int N = Vec16<T>::N;
    const bool vec = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    jf::launch(tanh_bwd_kernel<T>, dim3((unsigned)((n + 256 * N - 1) / (256 * N))), dim3(256), 0, (hipStream_t)stream, g, y, n, out, vec);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------------- AmortizableMLP nonlinearities
// The activations of extra_functions.py:81-89 other than tanh (which is fused into the dense kernels): an elementwise pass on the layer's
// pre-activation z, and its backward g * act'(z).  code: JF_ACT_RELU .. JF_ACT_IDENTITY (include/jammy_hip.h).
template <typename T> __device__ __forceinline__ T act_value(int code, T z) {
    switch (code) {
        case JF_ACT_RELU: return z > T(0) ? z : T(0);
        case JF_ACT_SOFTPLUS: return z > T(20) ? z : M<T>::log1p(M<T>::exp(z));                 // torch.nn.Softplus(beta=1, threshold=20)
        case JF_ACT_ELU: return z > T(0) ? z : M<T>::expm1(z);                                  // alpha = 1
        case JF_ACT_SWISH: return z / (T(1) + M<T>::exp(-z));                                   // x sigmoid(beta x), beta = 1 (extra_functions.py:62-68)
        case JF_ACT_SQUARE: return z * z;
        default: return z;
    }
}
template <typename T> __device__ __forceinline__ T act_deriv(int code, T z) {
    switch (code) {
        case JF_ACT_RELU: return z > T(0) ? T(1) : T(0);
        case JF_ACT_SOFTPLUS: return z > T(20) ? T(1) : T(1) / (T(1) + M<T>::exp(-z));
        case JF_ACT_ELU: return z > T(0) ? T(1) : M<T>::exp(z);
        case JF_ACT_SWISH: { const T sg = T(1) / (T(1) + M<T>::exp(-z)); return sg * (T(1) + z * (T(1) - sg)); }
        case JF_ACT_SQUARE: return T(2) * z;
        default: return T(1);
    }
}
template <typename T, bool BWD> __global__ void __launch_bounds__(256) act_kernel(const T* __restrict__ g, const T* __restrict__ z, int64_t n, int code,
                                                                                  T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = BWD ? g[i] * act_deriv<T>(code, z[i]) : act_value<T>(code, z[i]);
}
template <typename T, bool BWD> static int activation(const T* g, const T* z, int64_t n, int code, T* out, void* stream) {
    if (!z || !out || (BWD && !g) || n < 0) return JF_ERR_BADARG;
    if (code < JF_ACT_RELU || code > JF_ACT_IDENTITY) return JF_ERR_BADARG;
    if (n == 0) return JF_OK;
    jf::launch((act_kernel<T, BWD>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, z, n, code, out);
    return check_launch();
}

// out = a + b: the last operation of pdf.forward, log_prob = log_prob_base + log_det (main/default.py:1110-1117), as a library launch so that a
// recorded step plan (plan.hip) holds the WHOLE step
template <typename T> __global__ void __launch_bounds__(256) add_rows_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t n, T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
template <typename T> static int add_rows(const T* a, const T* b, int64_t n, T* out, void* stream) {
    if (!a || !b || !out || n < 0) return JF_ERR_BADARG;
    if (n == 0) return JF_OK;
    jf::launch(add_rows_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, out);
    return check_launch();
}

// sums of the per-block log-dets / base log-probs (include/jammy_hip.h: jf_combine_rows), list order, one row per thread
struct RowLists { const void* ld[JF_MAX_ROW_LISTS]; const void* blp[JF_MAX_ROW_LISTS]; int n_ld, n_blp; };
template <typename T> __global__ void __launch_bounds__(256) combine_rows_kernel(const RowLists a, int64_t B, T* __restrict__ ld_out, T* __restrict__ blp_out,
                                                                                 T* __restrict__ total_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    T ld = T(0), blp = T(0);
    for (int k = 0; k < a.n_ld; ++k) { const T v = static_cast<const T*>(a.ld[k])[i]; ld = k == 0 ? v : ld + v; }
    for (int k = 0; k < a.n_blp; ++k) { const T v = static_cast<const T*>(a.blp[k])[i]; blp = k == 0 ? v : blp + v; }
    if (ld_out) ld_out[i] = ld;
    if (blp_out) blp_out[i] = blp;
    if (total_out) total_out[i] = blp + ld;
}
template <typename T> static int combine_rows(const jf_row_list* ld, const jf_row_list* blp, int64_t B, T* ld_out, T* blp_out, T* total_out, void* stream) {
    if (!rows_ok(B) || (!ld_out && !blp_out && !total_out)) return JF_ERR_BADARG;
    RowLists a{};
    a.n_ld = ld ? ld->n : 0; a.n_blp = blp ? blp->n : 0;
    if (a.n_ld < 0 || a.n_ld > JF_MAX_ROW_LISTS || a.n_blp < 0 || a.n_blp > JF_MAX_ROW_LISTS) return JF_ERR_BADARG;
    for (int k = 0; k < a.n_ld; ++k) { if (!ld->p[k]) return JF_ERR_BADARG; a.ld[k] = ld->p[k]; }
    for (int k = 0; k < a.n_blp; ++k) { if (!blp->p[k]) return JF_ERR_BADARG; a.blp[k] = blp->p[k]; }
    if (B == 0) return JF_OK;
    jf::launch(combine_rows_kernel<T>, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, B, ld_out, blp_out, total_out);
    return check_launch();
}

// Adam over ALL parameter tensors of a model in one launch.  The models of this path have < 1 MB of parameters in ~40 tensors; torch's foreach
// implementation walks them with 6-7 multi_tensor_apply launches of 10-20 us each (0.09 ms of a 1.85 ms C3 training step, profiles/r03_train.md).
// Same update as torch.optim.Adam (amsgrad off, no weight decay), same operation order:
//   m <- m + (g - m) (1 - b1);  v <- v b2 + (1 - b2) g g;  p <- p - (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
template <typename T> struct AdamTable {
    T* p[JF_ADAM_MAX_TENSORS]; const T* g[JF_ADAM_MAX_TENSORS]; T* m[JF_ADAM_MAX_TENSORS]; T* v[JF_ADAM_MAX_TENSORS];
    int first_block[JF_ADAM_MAX_TENSORS + 1];
    int64_t n[JF_ADAM_MAX_TENSORS];
    int count;
};
template <typename T> __global__ void __launch_bounds__(256) adam_kernel(const AdamTable<T> t, double step_size, double b1, double b2, double inv_sqrt_bc2,
                                                                          double eps) {
    int k = 0;
    while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;                 // block-uniform: <= 96 scalar compares
    const int64_t i = (int64_t)((int)blockIdx.x - t.first_block[k]) * 256 + threadIdx.x;
    if (i >= t.n[k]) return;
    const T g = t.g[k][i];
    T m = t.m[k][i], v = t.v[k][i];
    m = m + (g - m) * (T)(1.0 - b1);
    v = v * (T)b2 + (T)(1.0 - b2) * g * g;
    t.m[k][i] = m; t.v[k][i] = v;
    const T denom = M<T>::sqrt(v) * (T)inv_sqrt_bc2 + (T)eps;
    t.p[k][i] = t.p[k][i] - (T)step_size * (m / denom);
}
// the step count read from device memory (a launch recorded in a HIP graph replays with the count of the replay, not of the capture): the bias
// corrections are computed by the first thread of every block
template <typename T> __global__ void __launch_bounds__(256) adam_dev_kernel(const AdamTable<T> t, double lr, double b1, double b2, double eps,
                                                                              const int64_t* __restrict__ step_dev) {
    __shared__ double corr[2];
    if (threadIdx.x == 0) {
        const double step = (double)(*step_dev < 1 ? (int64_t)1 : *step_dev);
        corr[0] = lr / (1.0 - pow(b1, step));
        corr[1] = 1.0 / sqrt(1.0 - pow(b2, step));
    }
    __syncthreads();
    const double step_size = corr[0], inv_sqrt_bc2 = corr[1];
    int k = 0;
    while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;
    const int64_t i = (int64_t)((int)blockIdx.x - t.first_block[k]) * 256 + threadIdx.x;
    if (i >= t.n[k]) return;
    const T g = t.g[k][i];
    T m = t.m[k][i], v = t.v[k][i];
    m = m + (g - m) * (T)(1.0 - b1);
    v = v * (T)b2 + (T)(1.0 - b2) * g * g;
    t.m[k][i] = m; t.v[k][i] = v;
    const T denom = M<T>::sqrt(v) * (T)inv_sqrt_bc2 + (T)eps;
    t.p[k][i] = t.p[k][i] - (T)step_size * (m / denom);
}
template <typename T> static int adam_step(const jf_adam_tensor* tensors, int32_t n, double lr, double b1, double b2, double eps, int64_t step,
                                           const int64_t* step_dev, void* stream) {
    if (!tensors || n < 1 || n > JF_ADAM_MAX_TENSORS || (!step_dev && step < 1) || !(lr >= 0) || !(b1 >= 0 && b1 < 1) || !(b2 >= 0 && b2 < 1) || !(eps >= 0))
        return JF_ERR_BADARG;
    AdamTable<T> t{};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const jf_adam_tensor& e = tensors[i];
        if (!e.param || !e.grad || !e.exp_avg || !e.exp_avg_sq || e.n < 0 || e.n > ((int64_t)1 << 31)) return JF_ERR_BADARG;
        t.p[i] = static_cast<T*>(e.param); t.g[i] = static_cast<const T*>(e.grad); t.m[i] = static_cast<T*>(e.exp_avg); t.v[i] = static_cast<T*>(e.exp_avg_sq);
        t.n[i] = e.n; t.first_block[i] = blocks;
        blocks += (int)((e.n + 255) / 256);
    }
    t.first_block[n] = blocks; t.count = n;
    if (blocks == 0) return JF_OK;
    if (step_dev) {
        jf::launch(adam_dev_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t, lr, b1, b2, eps, step_dev);
        return check_launch();
    }
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    jf::launch(adam_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t, lr / bc1, b1, b2, 1.0 / sqrt(bc2), eps);
    return check_launch();
}

// the scalar math policy of the flow kernels (jf_math.h), elementwise: what tests/test_gpu_math.py measures against torch
template <typename T> __global__ void __launch_bounds__(256) math_kernel(const T* __restrict__ x, int64_t n, int fn, T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const T v = x[i];
    out[i] = fn == JF_MATH_EXP_FAST ? M<T>::exp_fast(v) : fn == JF_MATH_LOG_FAST ? M<T>::log_fast(v) : fn == JF_MATH_TANH_FAST ? M<T>::tanh_fast(v) : M<T>::rcp(v);
}
template <typename T> static int device_math(const T* x, int64_t n, int fn, T* out, void* stream) {
    if (!x || !out || n < 0 || fn < JF_MATH_EXP_FAST || fn > JF_MATH_RCP) return JF_ERR_BADARG;
    if (n == 0) return JF_OK;
    jf::launch((math_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, fn, out);
    return check_launch();
}

}  // namespace jf

extern "C" {
int jf_device_math_f32(const float* x, int64_t n, int32_t fn, float* out, void* s) { return jf::device_math<float>(x, n, fn, out, s); }
int jf_device_math_f64(const double* x, int64_t n, int32_t fn, double* out, void* s) { return jf::device_math<double>(x, n, fn, out, s); }
int jf_combine_rows_f32(const jf_row_list* ld, const jf_row_list* blp, int64_t B, float* lo, float* bo, float* to, void* s) { return jf::combine_rows<float>(ld, blp, B, lo, bo, to, s); }
int jf_combine_rows_f64(const jf_row_list* ld, const jf_row_list* blp, int64_t B, double* lo, double* bo, double* to, void* s) { return jf::combine_rows<double>(ld, blp, B, lo, bo, to, s); }
int jf_adam_step_f32(const jf_adam_tensor* t, int32_t n, double lr, double b1, double b2, double eps, int64_t step, void* s) { return jf::adam_step<float>(t, n, lr, b1, b2, eps, step, nullptr, s); }
int jf_adam_step_f64(const jf_adam_tensor* t, int32_t n, double lr, double b1, double b2, double eps, int64_t step, void* s) { return jf::adam_step<double>(t, n, lr, b1, b2, eps, step, nullptr, s); }
int jf_adam_step_dev_f32(const jf_adam_tensor* t, int32_t n, double lr, double b1, double b2, double eps, const int64_t* step_dev, void* s) {
    return step_dev ? jf::adam_step<float>(t, n, lr, b1, b2, eps, 0, step_dev, s) : JF_ERR_BADARG;
}
int jf_adam_step_dev_f64(const jf_adam_tensor* t, int32_t n, double lr, double b1, double b2, double eps, const int64_t* step_dev, void* s) {
    return step_dev ? jf::adam_step<double>(t, n, lr, b1, b2, eps, 0, step_dev, s) : JF_ERR_BADARG;
}
int jf_add_rows_f32(const float* a, const float* b, int64_t n, float* out, void* s) { return jf::add_rows<float>(a, b, n, out, s); }
int jf_add_rows_f64(const double* a, const double* b, int64_t n, double* out, void* s) { return jf::add_rows<double>(a, b, n, out, s); }
int jf_conditioning_rows_f32(const jf_cond_segment* g, int32_t n, int64_t B, float* out, int64_t os, void* s) {
    return jf::conditioning_rows<float>(g, n, B, out, os, s);
}
int jf_conditioning_rows_f64(const jf_cond_segment* g, int32_t n, int64_t B, double* out, int64_t os, void* s) {
    return jf::conditioning_rows<double>(g, n, B, out, os, s);
}
int jf_coverage_histogram_f32(const float* l, int64_t B, double z0, const float* thr, int32_t n, int64_t* hist, float* tw, void* s) {
    return jf::coverage_hist<float>(l, B, z0, thr, n, hist, tw, s);
}
int jf_coverage_histogram_f64(const double* l, int64_t B, double z0, const double* thr, int32_t n, int64_t* hist, double* tw, void* s) {
    return jf::coverage_hist<double>(l, B, z0, thr, n, hist, tw, s);
}
int jf_segment_reduce_f32(const float* in, int64_t n_seg, int64_t seg_len, int32_t mode, float* out, void* s) {
    return jf::segment_reduce<float>(in, n_seg, seg_len, mode, out, s);
}
int jf_segment_reduce_f64(const double* in, int64_t n_seg, int64_t seg_len, int32_t mode, double* out, void* s) {
    return jf::segment_reduce<double>(in, n_seg, seg_len, mode, out, s);
}
int jf_normal_logp_f32(const float* z, int64_t zs, int64_t B, int32_t D, const float* in, float* out, void* s) {
    return jf::normal_logp<float>(z, zs, B, D, in, out, s);
}
int jf_normal_logp_f64(const double* z, int64_t zs, int64_t B, int32_t D, const double* in, double* out, void* s) {
    return jf::normal_logp<double>(z, zs, B, D, in, out, s);
}
int jf_slab_sum_f32(const float* a, int64_t na, float* out_a, const float* b, int64_t nb, float* out_b, int32_t S, int32_t chunk, void* s) {
    return jf::slab_sum<float>(a, na, out_a, b, nb, out_b, S, chunk, s);
}
int jf_slab_sum_f64(const double* a, int64_t na, double* out_a, const double* b, int64_t nb, double* out_b, int32_t S, int32_t chunk, void* s) {
    return jf::slab_sum<double>(a, na, out_a, b, nb, out_b, S, chunk, s);
}
int jf_slab_sum_map_f32(const float* a, int64_t na, const float* b, int64_t nb, const int32_t* map, float* out, int32_t S, void* s) {
    return map ? jf::slab_sum<float>(a, na, out, b, nb, nullptr, S, S, s, map) : JF_ERR_BADARG;
}
int jf_slab_sum_map_f64(const double* a, int64_t na, const double* b, int64_t nb, const int32_t* map, double* out, int32_t S, void* s) {
    return map ? jf::slab_sum<double>(a, na, out, b, nb, nullptr, S, S, s, map) : JF_ERR_BADARG;
}
int jf_tanh_bwd_f32(const float* g, const float* y, int64_t n, float* out, void* s) { return jf::tanh_bwd<float>(g, y, n, out, s); }
int jf_tanh_bwd_f64(const double* g, const double* y, int64_t n, double* out, void* s) { return jf::tanh_bwd<double>(g, y, n, out, s); }
int jf_activation_f32(const float* z, int64_t n, int32_t code, float* out, void* s) { return jf::activation<float, false>(nullptr, z, n, code, out, s); }
int jf_activation_f64(const double* z, int64_t n, int32_t code, double* out, void* s) { return jf::activation<double, false>(nullptr, z, n, code, out, s); }
int jf_activation_bwd_f32(const float* g, const float* z, int64_t n, int32_t code, float* out, void* s) { return jf::activation<float, true>(g, z, n, code, out, s); }
int jf_activation_bwd_f64(const double* g, const double* z, int64_t n, int32_t code, double* out, void* s) { return jf::activation<double, true>(g, z, n, code, out, s); }
}
