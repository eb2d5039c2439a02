// AmortizableMLP with PER-SAMPLE weights (amortize_everything / fully_amortized_pdf): one dense stage
//     out[b] = act( W_b in[b] + bias_b ) (+ residual[b]),    W_b = U_b (n_out x n_in)   or   U_b (n_out x rank) V_b (rank x n_in)
// where every sample b brings its own [U | V | bias] segment inside its row of the hyper-network's output
// (_apply_amortized_mlp with extra_inputs, jammy_flows/amortizable_mlp.py:508-578; the segment layout: :272-375).
//   jf_amlp_stage_*       forward
//   jf_amlp_stage_bwd_*   backward: g_out -> (g_in, g_segment (B, n_u + n_v + n_b)); the outer products g (x) in are per sample too
// No weight is shared between rows, so there is nothing for the matrix cores to reuse: the stage is a streaming pass over the (B, P)
// parameter block (HBM bound: every weight is read exactly once).  One wave per sample; lane j owns outputs j, j + 64, ...; the sample's
// input vector and the rank-space intermediate sit in LDS (broadcast reads).
#include "jf_common.h"
#include "jf_math.h"

namespace jf {

constexpr int AMLP_MAX_IN = 1024, AMLP_MAX_RANK = 64;

template <typename T> struct AmlpArgs {
    const T* in; int64_t ins;
    const T* seg; int64_t segs;           // this stage's [U | V | b] segment of every row, row stride segs
    int64_t B;
    int n_in, n_out, rank, has_bias, act;
    const T* res; int64_t ress;
    T* out; int64_t outs;
    // backward
    const T* g_out; int64_t gos;
    const T* y; int64_t ys;               // the stage's output (needed for the tanh derivative when act = 1)
    T* g_in; int64_t gis;
    T* g_seg; int64_t gss;
};

template <typename T> __global__ void __launch_bounds__(64) amlp_stage_kernel(const AmlpArgs<T> a) {
    __shared__ T xs[AMLP_MAX_IN];
    __shared__ T ts[AMLP_MAX_RANK];
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x;
    const T* x = a.in + b * a.ins;
    const T* U = a.seg + b * a.segs;
    for (int i = lane; i < a.n_in; i += 64) xs[i] = x[i];
    __syncthreads();
    const T* bias;
    if (a.rank > 0) {
        const T* V = U + (int64_t)a.n_out * a.rank;
        bias = V + (int64_t)a.rank * a.n_in;
        for (int r = lane; r < a.rank; r += 64) {
            T acc = T(0);
            for (int i = 0; i < a.n_in; ++i) acc += V[(int64_t)r * a.n_in + i] * xs[i];
            ts[r] = acc;
        }
        __syncthreads();
    } else {
        bias = U + (int64_t)a.n_out * a.n_in;
    }
    for (int j = lane; j < a.n_out; j += 64) {
        T acc = a.has_bias ? bias[j] : T(0);
        if (a.rank > 0) { for (int r = 0; r < a.rank; ++r) acc += U[(int64_t)j * a.rank + r] * ts[r]; }
        else { for (int i = 0; i < a.n_in; ++i) acc += U[(int64_t)j * a.n_in + i] * xs[i]; }
        if (a.act) acc = M<T>::tanh(acc);
        if (a.res) acc += a.res[b * a.ress + j];
        a.out[b * a.outs + j] = acc;
    }
}

// backward of one stage.  g = g_out (* (1 - y^2) for tanh, y = the stage's activated output WITHOUT the residual)
template <typename T> __global__ void __launch_bounds__(64) amlp_stage_bwd_kernel(const AmlpArgs<T> a) {
    __shared__ T xs[AMLP_MAX_IN];
    __shared__ T gs[AMLP_MAX_IN];          // g (pre-activation gradient), n_out <= AMLP_MAX_IN
    __shared__ T ts[AMLP_MAX_RANK];
    __shared__ T gt[AMLP_MAX_RANK];
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x;
    const T* x = a.in + b * a.ins;
    const T* U = a.seg + b * a.segs;
    T* gU = a.g_seg + b * a.gss;
    for (int i = lane; i < a.n_in; i += 64) xs[i] = x[i];
    for (int j = lane; j < a.n_out; j += 64) {
        T g = a.g_out[b * a.gos + j];
        if (a.act) { const T y = a.y[b * a.ys + j]; g *= (T(1) - y * y); }
        gs[j] = g;
    }
    __syncthreads();
    if (a.rank > 0) {
        const T* V = U + (int64_t)a.n_out * a.rank;
        T* gV = gU + (int64_t)a.n_out * a.rank;
        T* gb = gV + (int64_t)a.rank * a.n_in;
        for (int r = lane; r < a.rank; r += 64) {              // t = V x,  g_t = U^T g
            T acc = T(0), gacc = T(0);
            for (int i = 0; i < a.n_in; ++i) acc += V[(int64_t)r * a.n_in + i] * xs[i];
            for (int j = 0; j < a.n_out; ++j) gacc += U[(int64_t)j * a.rank + r] * gs[j];
            ts[r] = acc; gt[r] = gacc;
        }
        __syncthreads();
        for (int e = lane; e < a.n_out * a.rank; e += 64) gU[e] = gs[e / a.rank] * ts[e % a.rank];
        for (int e = lane; e < a.rank * a.n_in; e += 64) gV[e] = gt[e / a.n_in] * xs[e % a.n_in];
        if (a.has_bias) for (int j = lane; j < a.n_out; j += 64) gb[j] = gs[j];
        if (a.g_in) for (int i = lane; i < a.n_in; i += 64) {
            T acc = T(0);
            for (int r = 0; r < a.rank; ++r) acc += V[(int64_t)r * a.n_in + i] * gt[r];
            a.g_in[b * a.gis + i] = acc;
        }
    } else {
        T* gb = gU + (int64_t)a.n_out * a.n_in;
        for (int e = lane; e < a.n_out * a.n_in; e += 64) gU[e] = gs[e / a.n_in] * xs[e % a.n_in];
        if (a.has_bias) for (int j = lane; j < a.n_out; j += 64) gb[j] = gs[j];
        if (a.g_in) for (int i = lane; i < a.n_in; i += 64) {
            T acc = T(0);
            for (int j = 0; j < a.n_out; ++j) acc += U[(int64_t)j * a.n_in + i] * gs[j];
            a.g_in[b * a.gis + i] = acc;
        }
    }
}

template <typename T> static int amlp_check(const AmlpArgs<T>& a) {
    if (!a.in || !a.seg || a.B < 0 || a.n_in < 1 || a.n_out < 1 || a.rank < 0) return JF_ERR_BADARG;
    if (a.n_in > AMLP_MAX_IN || a.n_out > AMLP_MAX_IN || a.rank > AMLP_MAX_RANK) return JF_ERR_UNSUPPORTED;
    return JF_OK;
}

template <typename T>
static int amlp_stage(const T* in, int64_t ins, const T* seg, int64_t segs, int64_t B, int32_t n_in, int32_t n_out, int32_t rank, int32_t has_bias,
                      int32_t act, const T* res, int64_t ress, T* out, int64_t outs, void* stream) {
    AmlpArgs<T> a{};
    a.in = in; a.ins = ins; a.seg = seg; a.segs = segs; a.B = B; a.n_in = n_in; a.n_out = n_out; a.rank = rank; a.has_bias = has_bias; a.act = act;
    a.res = res; a.ress = ress; a.out = out; a.outs = outs;
    const int rc = amlp_check<T>(a);
    if (rc != JF_OK) return rc;
    if (!out) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    jf::launch(amlp_stage_kernel<T>, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, a);
    return check_launch();
}

template <typename T>
static int amlp_stage_bwd(const T* in, int64_t ins, const T* seg, int64_t segs, int64_t B, int32_t n_in, int32_t n_out, int32_t rank, int32_t has_bias,
                          int32_t act, const T* y, int64_t ys, const T* g_out, int64_t gos, T* g_in, int64_t gis, T* g_seg, int64_t gss, void* stream) {
    AmlpArgs<T> a{};
    a.in = in; a.ins = ins; a.seg = seg; a.segs = segs; a.B = B; a.n_in = n_in; a.n_out = n_out; a.rank = rank; a.has_bias = has_bias; a.act = act;
    a.y = y; a.ys = ys; a.g_out = g_out; a.gos = gos; a.g_in = g_in; a.gis = gis; a.g_seg = g_seg; a.gss = gss;
    const int rc = amlp_check<T>(a);
    if (rc != JF_OK) return rc;
    if (!g_out || !g_seg || (act && !y)) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    jf::launch(amlp_stage_bwd_kernel<T>, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, a);
    return check_launch();
}

}  // namespace jf

extern "C" {
#define JF_AMLP_DEF(T, suffix)                                                                                                              \
    int jf_amlp_stage_##suffix(const T* in, int64_t ins, const T* seg, int64_t segs, int64_t B, int32_t n_in, int32_t n_out, int32_t rank,    \
                               int32_t has_bias, int32_t act, const T* res, int64_t ress, T* out, int64_t outs, void* s) {                   \
        return jf::amlp_stage<T>(in, ins, seg, segs, B, n_in, n_out, rank, has_bias, act, res, ress, out, outs, s);                          \
    }                                                                                                                                       \
    int jf_amlp_stage_bwd_##suffix(const T* in, int64_t ins, const T* seg, int64_t segs, int64_t B, int32_t n_in, int32_t n_out, int32_t rank, \
                                   int32_t has_bias, int32_t act, const T* y, int64_t ys, const T* g_out, int64_t gos, T* g_in, int64_t gis,  \
                                   T* g_seg, int64_t gss, void* s) {                                                                         \
        return jf::amlp_stage_bwd<T>(in, ins, seg, segs, B, n_in, n_out, rank, has_bias, act, y, ys, g_out, gos, g_in, gis, g_seg, gss, s);  \
    }
JF_AMLP_DEF(float, f32)
JF_AMLP_DEF(double, f64)
}
