// Small row-wise helpers of the pdf orchestrator.
//   jf_normal_logp_*: out[b] = (in ? in[b] : 0) + sum_d N(0,1).log_prob(z[b,d])      (jammy_flows/main/default.py:1110-1115, 1657, 1670)
#include "jf_common.h"
#include "jf_math.h"

namespace jf {

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
int jf_normal_logp_f32(const float* z, int64_t zs, int64_t B, int32_t D, const float* in, float* out, void* s) {
    return jf::normal_logp<float>(z, zs, B, D, in, out, s);
}
int jf_normal_logp_f64(const double* z, int64_t zs, int64_t B, int32_t D, const double* in, double* out, void* s) {
    return jf::normal_logp<double>(z, zs, B, D, in, out, s);
}
}
