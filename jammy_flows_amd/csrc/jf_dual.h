// Forward-mode dual numbers for the manifold layers' backward pass.
//
// The manifold layers ('r', 'o', 'm', 'f', 'v', base-class charts / rotations) carry at most a few dozen parameters per sample, their
// device code (jf_spline.h, jf_sphere.h, jf_manifold.h, jf_expmap.h) is templated on the scalar type and written against the math policy
// M<T>.  Instantiating that very code on Dual<T> = (value, one directional derivative) gives exact derivatives of the functions the
// forward kernels evaluate -- including the bin-wise spline branches, the Newton / bisection iterations and every clamp -- without a
// second, hand-derived implementation that could drift from the first.  The backward kernel (manifold_bwd_kernels.hip) runs one pass
// per input direction (target coordinates + parameters of the row) and contracts the output tangents with the upstream gradients.
// Comparisons, bin searches and loop exits look at the value part only, so a dual pass takes exactly the branches of the forward pass.
#pragma once
#include "jf_math.h"

namespace jf {

template <typename T> struct Dual {
    T v, d;
    __host__ __device__ Dual() : v(T(0)), d(T(0)) {}
    __host__ __device__ Dual(T v_) : v(v_), d(T(0)) {}
    __host__ __device__ Dual(T v_, T d_) : v(v_), d(d_) {}
    template <typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value && !std::is_same<U, T>::value>::type>
    __host__ __device__ Dual(U u) : v((T)u), d(T(0)) {}
    __host__ __device__ explicit operator T() const { return v; }
    __host__ __device__ explicit operator int() const { return (int)v; }
};

#define JF_DUAL_BIN(op, VAL, DER)                                                                                                        \
    template <typename T> __host__ __device__ __forceinline__ Dual<T> operator op(const Dual<T>& a, const Dual<T>& b) { return Dual<T>(VAL, DER); } \
    template <typename T, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>                            \
    __host__ __device__ __forceinline__ Dual<T> operator op(const Dual<T>& a, U u) { const Dual<T> b((T)u); return Dual<T>(VAL, DER); }    \
    template <typename T, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>                            \
    __host__ __device__ __forceinline__ Dual<T> operator op(U u, const Dual<T>& b) { const Dual<T> a((T)u); return Dual<T>(VAL, DER); }
JF_DUAL_BIN(+, a.v + b.v, a.d + b.d)
JF_DUAL_BIN(-, a.v - b.v, a.d - b.d)
JF_DUAL_BIN(*, a.v * b.v, a.d * b.v + a.v * b.d)
JF_DUAL_BIN(/, a.v / b.v, (a.d - (a.v / b.v) * b.d) / b.v)
#undef JF_DUAL_BIN
template <typename T> __host__ __device__ __forceinline__ Dual<T> operator-(const Dual<T>& a) { return Dual<T>(-a.v, -a.d); }
template <typename T> __host__ __device__ __forceinline__ Dual<T> operator+(const Dual<T>& a) { return a; }
#define JF_DUAL_ASSIGN(op)                                                                                                   \
    template <typename T> __host__ __device__ __forceinline__ Dual<T>& operator op##=(Dual<T>& a, const Dual<T>& b) { a = a op b; return a; } \
    template <typename T, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>                \
    __host__ __device__ __forceinline__ Dual<T>& operator op##=(Dual<T>& a, U u) { a = a op Dual<T>((T)u); return a; }
JF_DUAL_ASSIGN(+)
JF_DUAL_ASSIGN(-)
JF_DUAL_ASSIGN(*)
JF_DUAL_ASSIGN(/)
#undef JF_DUAL_ASSIGN
#define JF_DUAL_CMP(op)                                                                                                                   \
    template <typename T> __host__ __device__ __forceinline__ bool operator op(const Dual<T>& a, const Dual<T>& b) { return a.v op b.v; }      \
    template <typename T, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>                             \
    __host__ __device__ __forceinline__ bool operator op(const Dual<T>& a, U u) { return a.v op (T)u; }                                     \
    template <typename T, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>                             \
    __host__ __device__ __forceinline__ bool operator op(U u, const Dual<T>& b) { return (T)u op b.v; }
JF_DUAL_CMP(<)
JF_DUAL_CMP(>)
JF_DUAL_CMP(<=)
JF_DUAL_CMP(>=)
JF_DUAL_CMP(==)
JF_DUAL_CMP(!=)
#undef JF_DUAL_CMP

template <typename T> struct M<Dual<T>> {
    using D = Dual<T>;
    using B = M<T>;
    static constexpr T PI = B::PI;
    static constexpr T TWO_PI = B::TWO_PI;
    static constexpr T HALF_LN_2PI = B::HALF_LN_2PI;
    static constexpr T SQRT2 = B::SQRT2;
    static constexpr T TINY = B::TINY;
    static constexpr T EPS_COS = B::EPS_COS;
    static constexpr T EPS_S1 = B::EPS_S1;
    static constexpr T KAPPA_ID = B::KAPPA_ID;
    // accurate functions throughout: a backward pass is not the place for the ~1e-6 hardware approximations
    static __device__ __forceinline__ D exp(D x) { const T e = B::exp(x.v); return D(e, e * x.d); }
    static __device__ __forceinline__ D exp_fast(D x) { return exp(x); }
    static __device__ __forceinline__ D log(D x) { return D(B::log(x.v), x.d / x.v); }
    static __device__ __forceinline__ D log_fast(D x) { return log(x); }
    static __device__ __forceinline__ D expm1(D x) { return D(B::expm1(x.v), x.d * B::exp(x.v)); }
    static __device__ __forceinline__ D log1p(D x) { return D(B::log1p(x.v), x.d / (T(1) + x.v)); }
    static __device__ __forceinline__ D sqrt(D x) { const T s = B::sqrt(x.v); return D(s, x.d * T(0.5) / s); }
    static __device__ __forceinline__ D sqrt_fast(D x) { return sqrt(x); }
    static __device__ __forceinline__ D rcp(D x) { const T r = T(1) / x.v; return D(r, -x.d * r * r); }
    static __device__ __forceinline__ D erf(D x) { return D(B::erf(x.v), x.d * T(1.1283791670955125739) * B::exp(-x.v * x.v)); }
    static __device__ __forceinline__ D erfinv(D x) { const T y = B::erfinv(x.v); return D(y, x.d * T(0.88622692545275801365) * B::exp(y * y)); }
    static __device__ __forceinline__ D erfcinv(D x) { const T y = B::erfcinv(x.v); return D(y, -x.d * T(0.88622692545275801365) * B::exp(y * y)); }
    static __device__ __forceinline__ D sin(D x) { return D(B::sin(x.v), x.d * B::cos(x.v)); }
    static __device__ __forceinline__ D cos(D x) { return D(B::cos(x.v), -x.d * B::sin(x.v)); }
    static __device__ __forceinline__ D acos(D x) { return D(B::acos(x.v), -x.d / B::sqrt(T(1) - x.v * x.v)); }
    static __device__ __forceinline__ D atan2(D y, D x) { return D(B::atan2(y.v, x.v), (x.v * y.d - y.v * x.d) / (x.v * x.v + y.v * y.v)); }
    static __device__ __forceinline__ D tanh(D x) { const T t = B::tanh(x.v); return D(t, x.d * (T(1) - t * t)); }
    static __device__ __forceinline__ D tanh_fast(D x) { return tanh(x); }
    static __device__ __forceinline__ D abs(D x) { return x.v < T(0) ? D(-x.v, -x.d) : x; }
    static __device__ __forceinline__ D max(D a, D b) { return a.v >= b.v ? a : b; }       // torch.max / clamp: the gradient follows the selected branch
    static __device__ __forceinline__ D min(D a, D b) { return a.v <= b.v ? a : b; }
    static __device__ __forceinline__ bool finite(D x) { return B::finite(x.v); }
};


// ---------------------------------------------------------------------------------------------------------- N tangents at once
// DualN<T, N> = (value, N directional derivatives).  One pass over a chain then serves N input directions: the value part -- the
// transcendentals, divisions, bin searches and Newton iterations -- is evaluated once instead of N times, each tangent costs the few
// multiply-adds of the chain rule.  Same interface as Dual<T> (the layer code is templated on the scalar type), same branches (comparisons
// look at the value).  Round 3: the 'f' layer's backward went from 12 single-tangent passes to 3 four-tangent passes.
template <typename T, int N> struct DualN {
    T v, d[N];
    __host__ __device__ DualN() : v(T(0)) { for (int c = 0; c < N; ++c) d[c] = T(0); }
    __host__ __device__ DualN(T v_) : v(v_) { for (int c = 0; c < N; ++c) d[c] = T(0); }
    template <typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value && !std::is_same<U, T>::value>::type>
    __host__ __device__ DualN(U u) : v((T)u) { for (int c = 0; c < N; ++c) d[c] = T(0); }
    __host__ __device__ explicit operator T() const { return v; }
    __host__ __device__ explicit operator int() const { return (int)v; }
};
#define JF_DN_LOOP _Pragma("unroll") for (int c = 0; c < N; ++c)
#define JF_DUALN_BIN(op, VAL, PRE, DER)                                                                                                      \
    template <typename T, int N> __host__ __device__ __forceinline__ DualN<T, N> operator op(const DualN<T, N>& a, const DualN<T, N>& b) {     \
        DualN<T, N> r; r.v = VAL; PRE; JF_DN_LOOP r.d[c] = DER; return r; }                                                                   \
    template <typename T, int N, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>                          \
    __host__ __device__ __forceinline__ DualN<T, N> operator op(const DualN<T, N>& a, U u) { return a op DualN<T, N>((T)u); }                  \
    template <typename T, int N, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>                          \
    __host__ __device__ __forceinline__ DualN<T, N> operator op(U u, const DualN<T, N>& b) { return DualN<T, N>((T)u) op b; }
JF_DUALN_BIN(+, a.v + b.v, (void)0, a.d[c] + b.d[c])
JF_DUALN_BIN(-, a.v - b.v, (void)0, a.d[c] - b.d[c])
JF_DUALN_BIN(*, a.v * b.v, (void)0, a.d[c] * b.v + a.v * b.d[c])
JF_DUALN_BIN(/, a.v / b.v, const T q = r.v, (a.d[c] - q * b.d[c]) / b.v)
#undef JF_DUALN_BIN
template <typename T, int N> __host__ __device__ __forceinline__ DualN<T, N> operator-(const DualN<T, N>& a) {
    DualN<T, N> r; r.v = -a.v; JF_DN_LOOP r.d[c] = -a.d[c]; return r;
}
template <typename T, int N> __host__ __device__ __forceinline__ DualN<T, N> operator+(const DualN<T, N>& a) { return a; }
#define JF_DUALN_ASSIGN(op)                                                                                                                   \
    template <typename T, int N> __host__ __device__ __forceinline__ DualN<T, N>& operator op##=(DualN<T, N>& a, const DualN<T, N>& b) { a = a op b; return a; } \
    template <typename T, int N, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>                          \
    __host__ __device__ __forceinline__ DualN<T, N>& operator op##=(DualN<T, N>& a, U u) { a = a op DualN<T, N>((T)u); return a; }
JF_DUALN_ASSIGN(+)
JF_DUALN_ASSIGN(-)
JF_DUALN_ASSIGN(*)
JF_DUALN_ASSIGN(/)
#undef JF_DUALN_ASSIGN
#define JF_DUALN_CMP(op)                                                                                                                      \
    template <typename T, int N> __host__ __device__ __forceinline__ bool operator op(const DualN<T, N>& a, const DualN<T, N>& b) { return a.v op b.v; } \
    template <typename T, int N, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>                          \
    __host__ __device__ __forceinline__ bool operator op(const DualN<T, N>& a, U u) { return a.v op (T)u; }                                    \
    template <typename T, int N, typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value>::type>                          \
    __host__ __device__ __forceinline__ bool operator op(U u, const DualN<T, N>& b) { return (T)u op b.v; }
JF_DUALN_CMP(<)
JF_DUALN_CMP(>)
JF_DUALN_CMP(<=)
JF_DUALN_CMP(>=)
JF_DUALN_CMP(==)
JF_DUALN_CMP(!=)
#undef JF_DUALN_CMP

template <typename T, int N> struct M<DualN<T, N>> {
    using D = DualN<T, N>;
    using B = M<T>;
    static constexpr T PI = B::PI;
    static constexpr T TWO_PI = B::TWO_PI;
    static constexpr T HALF_LN_2PI = B::HALF_LN_2PI;
    static constexpr T SQRT2 = B::SQRT2;
    static constexpr T TINY = B::TINY;
    static constexpr T EPS_COS = B::EPS_COS;
    static constexpr T EPS_S1 = B::EPS_S1;
    static constexpr T KAPPA_ID = B::KAPPA_ID;
    // value v, tangents x.d[c] * s
    static __device__ __forceinline__ D chain(T v, const D& x, T s) { D r; r.v = v; JF_DN_LOOP r.d[c] = x.d[c] * s; return r; }
    static __device__ __forceinline__ D exp(D x) { const T e = B::exp(x.v); return chain(e, x, e); }
    static __device__ __forceinline__ D exp_fast(D x) { return exp(x); }
    static __device__ __forceinline__ D log(D x) { return chain(B::log(x.v), x, T(1) / x.v); }
    static __device__ __forceinline__ D log_fast(D x) { return log(x); }
    static __device__ __forceinline__ D expm1(D x) { return chain(B::expm1(x.v), x, B::exp(x.v)); }
    static __device__ __forceinline__ D log1p(D x) { return chain(B::log1p(x.v), x, T(1) / (T(1) + x.v)); }
    static __device__ __forceinline__ D sqrt(D x) { const T s = B::sqrt(x.v); return chain(s, x, T(0.5) / s); }
    static __device__ __forceinline__ D sqrt_fast(D x) { return sqrt(x); }
    static __device__ __forceinline__ D rcp(D x) { const T r = T(1) / x.v; return chain(r, x, -r * r); }
    static __device__ __forceinline__ D erf(D x) { return chain(B::erf(x.v), x, T(1.1283791670955125739) * B::exp(-x.v * x.v)); }
    static __device__ __forceinline__ D erfinv(D x) { const T y = B::erfinv(x.v); return chain(y, x, T(0.88622692545275801365) * B::exp(y * y)); }
    static __device__ __forceinline__ D erfcinv(D x) { const T y = B::erfcinv(x.v); return chain(y, x, -T(0.88622692545275801365) * B::exp(y * y)); }
    static __device__ __forceinline__ D sin(D x) { return chain(B::sin(x.v), x, B::cos(x.v)); }
    static __device__ __forceinline__ D cos(D x) { return chain(B::cos(x.v), x, -B::sin(x.v)); }
    static __device__ __forceinline__ D acos(D x) { return chain(B::acos(x.v), x, -T(1) / B::sqrt(T(1) - x.v * x.v)); }
    static __device__ __forceinline__ D atan2(D y, D x) {
        const T r2 = x.v * x.v + y.v * y.v;
        D r; r.v = B::atan2(y.v, x.v);
        JF_DN_LOOP r.d[c] = (x.v * y.d[c] - y.v * x.d[c]) / r2;
        return r;
    }
    static __device__ __forceinline__ D tanh(D x) { const T t = B::tanh(x.v); return chain(t, x, T(1) - t * t); }
    static __device__ __forceinline__ D tanh_fast(D x) { return tanh(x); }
    static __device__ __forceinline__ D abs(D x) { return x.v < T(0) ? -x : x; }
    static __device__ __forceinline__ D max(D a, D b) { return a.v >= b.v ? a : b; }
    static __device__ __forceinline__ D min(D a, D b) { return a.v <= b.v ? a : b; }
    static __device__ __forceinline__ bool finite(D x) { return B::finite(x.v); }
};

// ---- values of a dual row / the implicit-function tangent (DualTraits, jf_math.h)
template <typename T> struct DualTraits<Dual<T>> {
    static constexpr bool is_dual = true;
    using value_type = T;
    static __host__ __device__ __forceinline__ T value(const Dual<T>& a) { return a.v; }
    // x = xv with the tangent that solves f(x, p) = z:  dx = (dz - df|_{x fixed}) / f'(x)
    static __host__ __device__ __forceinline__ Dual<T> implicit(T xv, const Dual<T>& z, const Dual<T>& f, T fprime) { return Dual<T>(xv, (z.d - f.d) / fprime); }
};
template <typename T, int N> struct DualTraits<DualN<T, N>> {
    static constexpr bool is_dual = true;
    using value_type = T;
    static __host__ __device__ __forceinline__ T value(const DualN<T, N>& a) { return a.v; }
    static __host__ __device__ __forceinline__ DualN<T, N> implicit(T xv, const DualN<T, N>& z, const DualN<T, N>& f, T fprime) {
        DualN<T, N> r(xv);
        const T inv = T(1) / fprime;
        JF_DN_LOOP r.d[c] = (z.d[c] - f.d[c]) * inv;
        return r;
    }
};
// a row of duals read as its values
template <typename D> struct DualValues {
    const D* p;
    __host__ __device__ __forceinline__ typename DualTraits<D>::value_type operator[](int i) const { return p[i].v; }
};

}  // namespace jf
