// Interval / sphere base-class arithmetic for lane-per-sample kernels:
//   jammy_flows/layers/intervals/interval_base.py:33-59      (R <-> [a,b] through the normal CDF)
//   jammy_flows/layers/spheres/sphere_base.py:8-38           (pole / seam clamps)
//                                            :242-335        (S1 / S2 <-> embedding space)
//                                            :456-598        (sphere <-> plane charts of the first layer)
//                                            :112-127,222-240 (Householder rotation in embedding space)
#pragma once
#include "jf_common.h"
#include "jf_math.h"

namespace jf {

constexpr double PI_D = 3.14159265358979323846;

// ---- clamps (margins are the reference's literal constants; evaluated in double, then rounded to T like torch does)
template <typename T> __device__ __forceinline__ T safe_angle_pi(T x) {         // return_safe_angle_within_pi(x, 1e-7)
    const T lo = T(1e-7), hi = T(PI_D - 1e-7);
    return x > hi ? hi : (x < lo ? lo : x);
}
template <typename T> __device__ __forceinline__ T safe_angle_2pi(T x) {        // spline_fns.return_safe_angle_within_2pi(x, 1e-7)
    const T lo = T(1e-7), hi = T(2.0 * PI_D - 1e-7);
    return x > hi ? hi : (x < lo ? lo : x);
}
template <typename T> __device__ __forceinline__ T safe_cos(T x, T margin) {    // return_safe_costheta
    const T lo = T(-1) + margin, hi = T(1) - margin;
    return x > hi ? hi : (x < lo ? lo : x);
}

// ---- intervals
template <typename T> __device__ __forceinline__ T real_line_to_interval(T x, T lo, T hi, T& ld) {      // interval_base.py:33-45
    const T w = hi - lo;
    ld += T(-0.5) * x * x - M<T>::HALF_LN_2PI + M<T>::log(w);
    return (T(0.5) + T(0.5) * M<T>::erf(x / M<T>::SQRT2)) * w + lo;
}
template <typename T> __device__ __forceinline__ T interval_to_real_line(T x, T lo, T hi, T& ld) {      // interval_base.py:47-59
    const T w = hi - lo;
    const T r = M<T>::erfinv(T(2) * ((x - lo) / w) - T(1)) * M<T>::SQRT2;
    ld -= T(-0.5) * r * r - M<T>::HALF_LN_2PI + M<T>::log(w);
    return r;
}

// ---- S1 <-> embedding
template <typename T> __device__ __forceinline__ void s1_to_eucl(T phi, T (&e)[3]) { e[0] = M<T>::cos(phi); e[1] = M<T>::sin(phi); }
template <typename T> __device__ __forceinline__ T eucl_to_s1(const T (&e)[3]) {                        // sphere_base.py:248-265
    const T r = M<T>::sqrt(e[0] * e[0] + e[1] * e[1]);
    const T a = M<T>::acos(e[0] / r);
    return e[1] < T(0) ? M<T>::TWO_PI - a : a;
}
// ---- S2 <-> embedding (log_det: +log sin(theta) to the embedding, -log sin(theta) back; sphere_base.py:266-282, 313-332)
template <typename T> __device__ __forceinline__ void s2_to_eucl(T theta, T phi, T (&e)[3], T& ld) {
    theta = safe_angle_pi(theta);
    const T st = M<T>::sin(theta);
    e[0] = st * M<T>::cos(phi);
    e[1] = st * M<T>::sin(phi);
    e[2] = M<T>::cos(theta);
    ld += M<T>::log(st);
}
template <typename T> __device__ __forceinline__ void eucl_to_s2(const T (&e)[3], T& theta, T& phi, T& ld) {
    theta = safe_angle_pi(M<T>::acos(e[2] / M<T>::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2])));
    ld -= M<T>::log(M<T>::sin(theta));
    T arg = e[0] / M<T>::sqrt(e[0] * e[0] + e[1] * e[1]);
    arg = arg > T(1) ? T(1) : (arg < T(-1) ? T(-1) : arg);
    const T a = M<T>::acos(arg);
    phi = e[1] < T(0) ? M<T>::TWO_PI - a : a;
}

// ---- charts of the first layer of a sphere block
template <typename T> __device__ __forceinline__ T s1_to_plane(T x, T& ld) {                           // sphere_base.py:460-480
    const bool flip = x > M<T>::PI;
    T nx = flip ? M<T>::TWO_PI - x : x;
    if (nx <= T(0)) nx = M<T>::EPS_S1;
    if (nx >= M<T>::TWO_PI) nx = M<T>::TWO_PI - M<T>::EPS_S1;
    const T y = M<T>::SQRT2 * M<T>::erfinv(T(1) - nx / M<T>::PI);
    ld += -M<T>::HALF_LN_2PI + T(0.5) * y * y;
    return flip ? -y : y;
}
template <typename T> __device__ __forceinline__ T plane_to_s1(T x, T& ld) {                           // sphere_base.py:529-539, 371-381
    const T r = M<T>::abs(x);
    ld += M<T>::HALF_LN_2PI - T(0.5) * r * r;
    const T a = M<T>::PI * (T(1) - M<T>::erf(r / M<T>::SQRT2));
    return x >= T(0) ? a : M<T>::TWO_PI - a;
}
template <typename T> __device__ __forceinline__ void s2_to_plane(T theta, T phi, T (&p)[3], T& ld) {   // sphere_base.py:496-513, 416-430
    const T st = safe_angle_pi(theta);
    const T c = safe_cos(M<T>::cos(st), T(1e-6));
    const T r = M<T>::sqrt(T(-2) * M<T>::log((T(1) - c) * T(0.5)));
    ld += -M<T>::log(T(1) - c) + M<T>::log(M<T>::sin(st));
    p[0] = r * M<T>::cos(phi);
    p[1] = r * M<T>::sin(phi);
}
template <typename T> __device__ __forceinline__ void plane_to_s2(const T (&p)[3], T& theta, T& phi, T& ld) {   // sphere_base.py:569-592, 371-399
    const T r = M<T>::sqrt(p[0] * p[0] + p[1] * p[1]);
    const T arg = r == T(0) ? T(1) : p[0] / r;
    const T a = M<T>::acos(arg);
    phi = p[1] < T(0) ? M<T>::TWO_PI - a : a;
    theta = safe_angle_pi(M<T>::acos(T(1) - T(2) * M<T>::exp(T(-0.5) * r * r)));
    ld += M<T>::log(T(1) - M<T>::cos(theta)) - M<T>::log(M<T>::sin(theta));
}

// ---- Householder rotation in embedding space R^E (E = 2 or 3): Q = H_0 H_1 ... ; reads raw v's (not pre-normalised)
template <typename T, int E> __device__ __forceinline__ void reflect_raw(const T* __restrict__ v, T (&x)[3]) {
    T n2 = T(0), dot = T(0);
#pragma unroll
    for (int i = 0; i < E; ++i) { n2 += v[i] * v[i]; dot += v[i] * x[i]; }
    const T f = T(2) * dot / n2;
#pragma unroll
    for (int i = 0; i < E; ++i) x[i] -= f * v[i];
}
template <typename T, int E> __device__ __forceinline__ void rotate_embed(const T* __restrict__ vs, int n_iter, T (&x)[3], bool transpose) {
    if (transpose) { for (int i = 0; i < n_iter; ++i) reflect_raw<T, E>(vs + i * E, x); }       // Q^T x : H_0 first
    else { for (int i = n_iter - 1; i >= 0; --i) reflect_raw<T, E>(vs + i * E, x); }            // Q x
}

}  // namespace jf
