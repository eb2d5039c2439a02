// Small row-wise helpers of the pdf orchestrator.
//   jf_normal_logp_*: out[b] = (in ? in[b] : 0) + sum_d N(0,1).log_prob(z[b,d])      (jammy_flows/main/default.py:1110-1115, 1657, 1670)
//   jf_conditioning_rows_*: the input rows of the amortisation MLPs, cat[conditional_input, embed(x_0), embed(x_1), ...]
//                            (main/default.py:946-962; embed = identity / (cos, sin) / (x, y, z): sphere_base.py:305-332, 786-794), one launch
#include "jf_common.h"
#include "jf_math.h"
#include "jf_sphere.h"

namespace jf {

struct CondSegs { int n; jf_cond_segment s[JF_MAX_SEGMENTS]; };

template <typename T>
__global__ void __launch_bounds__(256) conditioning_kernel(const CondSegs a, int64_t B, T* __restrict__ out, int64_t os) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    T* o = out + row * os;
    for (int i = 0; i < a.n; ++i) {
        const jf_cond_segment g = a.s[i];                       // uniform
        const T* r = static_cast<const T*>(g.src) + row * g.stride;
        if (g.kind == 0) {
            for (int c = 0; c < g.n_in; ++c) o[c] = r[c];
            o += g.n_in;
        } else if (g.kind == 1) {
            T e[3];
            s1_to_eucl<T>(r[0], e);
            o[0] = e[0]; o[1] = e[1];
            o += 2;
        } else {
            T e[3], ld = T(0);
            s2_to_eucl<T>(r[0], r[1], e, ld);
            o[0] = e[0]; o[1] = e[1]; o[2] = e[2];
            o += 3;
        }
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
    hipLaunchKernelGGL(conditioning_kernel<T>, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, B, out, os);
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

template <typename T> static int normal_logp(const T* z, int64_t zs, int64_t B, int32_t D, const T* in, T* out, void* stream) {
    if (!z || !out || D < 0 || B < 0) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    hipLaunchKernelGGL(normal_logp_kernel<T>, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, zs, B, (int)D, in, out);
    return check_launch();
}

}  // namespace jf

extern "C" {
int jf_conditioning_rows_f32(const jf_cond_segment* g, int32_t n, int64_t B, float* out, int64_t os, void* s) {
    return jf::conditioning_rows<float>(g, n, B, out, os, s);
}
int jf_conditioning_rows_f64(const jf_cond_segment* g, int32_t n, int64_t B, double* out, int64_t os, void* s) {
    return jf::conditioning_rows<double>(g, n, B, out, os, s);
}
int jf_normal_logp_f32(const float* z, int64_t zs, int64_t B, int32_t D, const float* in, float* out, void* s) {
    return jf::normal_logp<float>(z, zs, B, D, in, out, s);
}
int jf_normal_logp_f64(const double* z, int64_t zs, int64_t B, int32_t D, const double* in, double* out, void* s) {
    return jf::normal_logp<double>(z, zs, B, D, in, out, s);
}
}
