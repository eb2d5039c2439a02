// Interval / sphere base-class arithmetic for lane-per-sample kernels:
//   jammy_flows/layers/intervals/interval_base.py:33-59      (R <-> [a,b] through the normal CDF)
//   jammy_flows/layers/spheres/sphere_base.py:8-38           (pole / seam clamps)
//                                            :242-335        (S1 / S2 <-> embedding space)
//                                            :456-598        (sphere <-> plane charts of the first layer)
//                                            :112-127,222-240 (Householder rotation in embedding space)
#pragma once
#include "jf_common.h"
#include "jf_math.h"
#include "jf_dual.h"

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

// ---- azimuth of (x, y) in [0, 2 pi): acos(x / rho), mirrored for y < 0 -- the reference's formula (sphere_base.py:260-262, 275-280), kept for
// value parity.  Its derivative, though, is taken from the geometry, d phi = (x dy - y dx) / rho^2, not from acos': within ~3e-4 (float32;
// 1.5e-8 float64) of phi = 0 or pi the argument rounds to exactly +-1, acos' is infinite there and one such row (2e-4 of uniformly drawn
// float32 rows) would turn every gradient of a training step into NaN although the map is perfectly smooth at those points.
template <typename T> __device__ __forceinline__ T azimuth(T x, T y) {
    const T rho = M<T>::sqrt(x * x + y * y);
    T arg = rho == T(0) ? T(1) : x / rho;
    arg = arg > T(1) ? T(1) : (arg < T(-1) ? T(-1) : arg);
    const T a = M<T>::acos(arg);
    return y < T(0) ? M<T>::TWO_PI - a : a;
}
template <typename T> __device__ __forceinline__ Dual<T> azimuth(Dual<T> x, Dual<T> y) {
    const T r2 = x.v * x.v + y.v * y.v;
    return Dual<T>(azimuth<T>(x.v, y.v), r2 > T(0) ? (x.v * y.d - y.v * x.d) / r2 : T(0));
}
template <typename T, int N> __device__ __forceinline__ DualN<T, N> azimuth(DualN<T, N> x, DualN<T, N> y) {
    const T r2 = x.v * x.v + y.v * y.v;
    DualN<T, N> r;
    r.v = azimuth<T>(x.v, y.v);
#pragma unroll
    for (int c = 0; c < N; ++c) r.d[c] = r2 > T(0) ? (x.v * y.d[c] - y.v * x.d[c]) / r2 : T(0);
    return r;
}

// ---- S1 <-> embedding
template <typename T> __device__ __forceinline__ void s1_to_eucl(T phi, T (&e)[3]) { e[0] = M<T>::cos(phi); e[1] = M<T>::sin(phi); }
template <typename T> __device__ __forceinline__ T eucl_to_s1(const T (&e)[3]) { return azimuth(e[0], e[1]); }      // sphere_base.py:248-265
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
    phi = azimuth(e[0], e[1]);
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
    phi = azimuth(p[0], p[1]);
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
// Rotation of the embedding vector.  `hh` encodes the reference's rotation_mode (sphere_base.py:112-240) in the layer descriptors' hh_iter field:
//   hh >= 0  "householder": Q = H_0 H_1 ... H_{hh-1}, hh * E raw reflection vectors
//   hh == -1 "angles":      product of Givens rotations over all index pairs (a < b) in lexicographic order, E (E - 1) / 2 angles (:132-160)
//   hh == -2 "xyz":         the rotation that takes e_z to the direction of the 3 parameters (:162-185), S2 only
//   hh == -3 "quaternion":  unnormalised quaternion (a, i, j, k), 4 parameters (:186-216), S2 only
// transpose = true applies Q^T (log-prob direction), false applies Q.
constexpr int JF_ROT_ANGLES = -1, JF_ROT_XYZ = -2, JF_ROT_QUATERNION = -3;
__host__ __device__ inline int rot_len(int hh, int E) { return hh >= 0 ? hh * E : (hh == JF_ROT_ANGLES ? E * (E - 1) / 2 : (hh == JF_ROT_XYZ ? 3 : 4)); }

template <typename T, int E> __device__ __forceinline__ void rotate_embed(const T* __restrict__ vs, int hh, T (&x)[3], bool transpose) {
    if (hh >= 0) {
        if (transpose) { for (int i = 0; i < hh; ++i) reflect_raw<T, E>(vs + i * E, x); }       // Q^T x : H_0 first
        else { for (int i = hh - 1; i >= 0; --i) reflect_raw<T, E>(vs + i * E, x); }            // Q x
        return;
    }
    T R[3][3] = {{T(1), T(0), T(0)}, {T(0), T(1), T(0)}, {T(0), T(0), T(1)}};
    if (hh == JF_ROT_ANGLES) {                       // R = G_last ... G_1 G_0 (prev = new @ prev)
        int ind = 0;
#pragma unroll
        for (int a = 0; a < E; ++a)
#pragma unroll
            for (int b = a + 1; b < E; ++b) {
                const T c = M<T>::cos(vs[ind]), s = M<T>::sin(vs[ind]);
                ++ind;
#pragma unroll
                for (int j = 0; j < E; ++j) {        // rows a, b of (G R): G[a][a] = c, G[a][b] = s, G[b][a] = -s, G[b][b] = c
                    const T ra = R[a][j], rb = R[b][j];
                    R[a][j] = c * ra + s * rb;
                    R[b][j] = -s * ra + c * rb;
                }
            }
    } else if (hh == JF_ROT_XYZ) {
        const T nrm = M<T>::sqrt(vs[0] * vs[0] + vs[1] * vs[1] + vs[2] * vs[2]);
        const T mx = vs[0] / nrm, my = vs[1] / nrm, mz = vs[2] / nrm;
        const T d = T(1) + mz;
        R[0][0] = T(1) - mx * mx / d; R[0][1] = -mx * my / d;      R[0][2] = mx;
        R[1][0] = -mx * my / d;      R[1][1] = T(1) - my * my / d; R[1][2] = my;
        R[2][0] = -mx;               R[2][1] = -my;               R[2][2] = mz;
    } else {
        const T qa = vs[0], qi = vs[1], qj = vs[2], qk = vs[3];
        const T n2 = qa * qa + qi * qi + qj * qj + qk * qk;
        R[0][0] = T(1) - T(2) * (qj * qj + qk * qk) / n2; R[0][1] = T(2) * (qi * qj - qa * qk) / n2;       R[0][2] = T(2) * (qi * qk + qj * qa) / n2;
        R[1][0] = T(2) * (qi * qj + qa * qk) / n2;       R[1][1] = T(1) - T(2) * (qi * qi + qk * qk) / n2; R[1][2] = T(2) * (qj * qk - qi * qa) / n2;
        R[2][0] = T(2) * (qi * qk - qj * qa) / n2;       R[2][1] = T(2) * (qj * qk + qi * qa) / n2;       R[2][2] = T(1) - T(2) * (qi * qi + qj * qj) / n2;
    }
    T y[3] = {T(0), T(0), T(0)};
#pragma unroll
    for (int i = 0; i < E; ++i)
#pragma unroll
        for (int j = 0; j < E; ++j) y[i] += (transpose ? R[j][i] : R[i][j]) * x[j];
#pragma unroll
    for (int i = 0; i < E; ++i) x[i] = y[i];
}

}  // namespace jf
