// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build or call this code.
// PARITY: the reference (andyyankai/psdr-jit) cannot be built or imported here (drjit/OptiX/CUDA absent, no
// tests upstream); pinned against the outputs its tutorial notebooks embed - see oracle/README.md.
//
// num.h — scalar number types for the CPU restatement.
//   Real<ad>  = float (ad=false, the reference's "C" arrays) or Dual (ad=true, the
//   reference's DiffArray "D" arrays restricted to ONE forward tangent direction).
//   The reference gets derivatives from drjit's tape (include/psdr/types.h:23-40); here the
//   same arithmetic is run on (value, tangent) pairs so that `detach(x)` == drop the tangent.
// Vector helpers follow drjit's fused forms (dot = fmadd chain, cross = fmsub) so that the
// HIP path and this file round the same way.
#pragma once
#include <cmath>
#include <cstdint>
#include <type_traits>

namespace orc {

struct Dual {
    float v = 0.f, d = 0.f;
    Dual() = default;
    Dual(float v_) : v(v_), d(0.f) {}
    Dual(float v_, float d_) : v(v_), d(d_) {}
};

inline float detach(float x) { return x; }
inline float detach(const Dual &x) { return x.v; }
inline float tangent(float) { return 0.f; }
inline float tangent(const Dual &x) { return x.d; }

inline Dual operator+(const Dual &a, const Dual &b) { return {a.v + b.v, a.d + b.d}; }
inline Dual operator-(const Dual &a, const Dual &b) { return {a.v - b.v, a.d - b.d}; }
inline Dual operator-(const Dual &a) { return {-a.v, -a.d}; }
inline Dual operator*(const Dual &a, const Dual &b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
inline Dual operator/(const Dual &a, const Dual &b) {
    float q = a.v / b.v;
    return {q, (a.d - q * b.d) / b.v};
}
inline Dual &operator+=(Dual &a, const Dual &b) { a = a + b; return a; }
inline Dual &operator-=(Dual &a, const Dual &b) { a = a - b; return a; }
inline Dual &operator*=(Dual &a, const Dual &b) { a = a * b; return a; }
inline Dual &operator/=(Dual &a, const Dual &b) { a = a / b; return a; }

inline float fma_(float a, float b, float c) { return std::fmaf(a, b, c); }
inline Dual fma_(const Dual &a, const Dual &b, const Dual &c) {
    return {std::fmaf(a.v, b.v, c.v), a.d * b.v + a.v * b.d + c.d};
}
inline float sqrt_(float a) { return std::sqrt(a); }
inline Dual sqrt_(const Dual &a) {
    float s = std::sqrt(a.v);
    return {s, a.d / (2.f * s)};
}
// drjit::safe_sqrt: sqrt(max(x, 0))
inline float safe_sqrt(float a) { return std::sqrt(a > 0.f ? a : 0.f); }
inline Dual safe_sqrt(const Dual &a) {
    if (!(a.v > 0.f)) return {0.f, 0.f};
    return sqrt_(a);
}
inline float abs_(float a) { return std::fabs(a); }
inline Dual abs_(const Dual &a) { return a.v < 0.f ? -a : a; }
inline float rcp(float a) { return 1.f / a; }
inline Dual rcp(const Dual &a) { return Dual(1.f) / a; }
inline float sqr(float a) { return a * a; }
inline Dual sqr(const Dual &a) { return a * a; }
inline bool isfinite_(float a) { return std::isfinite(a); }

template <bool ad> using Real = std::conditional_t<ad, Dual, float>;

// ---------------------------------------------------------------- vectors
template <typename T> struct V2 {
    T x{}, y{};
    V2() = default;
    V2(T x_, T y_) : x(x_), y(y_) {}
    template <typename U> explicit V2(const V2<U> &o) : x(o.x), y(o.y) {}
};
template <typename T> struct V3 {
    T x{}, y{}, z{};
    V3() = default;
    V3(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
    explicit V3(T s) : x(s), y(s), z(s) {}
    template <typename U, typename = std::enable_if_t<!std::is_same_v<U, T>>>
    V3(const V3<U> &o) : x(o.x), y(o.y), z(o.z) {}   // float -> Dual promotion
    T &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    const T &operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
using V2f = V2<float>; using V2d = V2<Dual>;
using V3f = V3<float>; using V3d = V3<Dual>;

inline V3f detach(const V3f &a) { return a; }
inline V3f detach(const V3d &a) { return {a.x.v, a.y.v, a.z.v}; }
inline V3f tangent(const V3d &a) { return {a.x.d, a.y.d, a.z.d}; }
inline V2f detach(const V2f &a) { return a; }
inline V2f detach(const V2d &a) { return {a.x.v, a.y.v}; }

#define ORC_V3_BIN(op) \
    template <typename T> V3<T> operator op(const V3<T> &a, const V3<T> &b) { return {a.x op b.x, a.y op b.y, a.z op b.z}; } \
    template <typename T> V3<T> operator op(const V3<T> &a, const T &b) { return {a.x op b, a.y op b, a.z op b}; } \
    template <typename T> V3<T> operator op(const T &a, const V3<T> &b) { return {a op b.x, a op b.y, a op b.z}; }
ORC_V3_BIN(+) ORC_V3_BIN(-) ORC_V3_BIN(*) ORC_V3_BIN(/)
#undef ORC_V3_BIN
template <typename T> V3<T> operator-(const V3<T> &a) { return {-a.x, -a.y, -a.z}; }
template <typename T> V3<T> &operator+=(V3<T> &a, const V3<T> &b) { a = a + b; return a; }
template <typename T> V3<T> &operator*=(V3<T> &a, const V3<T> &b) { a = a * b; return a; }
template <typename T> V3<T> &operator*=(V3<T> &a, const T &b) { a = a * b; return a; }
template <typename T> V3<T> &operator/=(V3<T> &a, const T &b) { a = a / b; return a; }
// mixed float-vector (x) Dual-scalar
inline V3d operator*(const V3f &a, const Dual &b) { return V3d(a) * b; }
inline V3d operator*(const V3d &a, float b) { return a * Dual(b); }
inline V3d operator/(const V3d &a, float b) { return a / Dual(b); }
inline V3d operator*(const V3d &a, const V3f &b) { return a * V3d(b); }
inline V3d operator-(const V3d &a, const V3f &b) { return a - V3d(b); }
inline V3d operator-(const V3f &a, const V3d &b) { return V3d(a) - b; }
inline V3d operator+(const V3d &a, const V3f &b) { return a + V3d(b); }

template <typename T> V2<T> operator+(const V2<T> &a, const V2<T> &b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> V2<T> operator-(const V2<T> &a, const V2<T> &b) { return {a.x - b.x, a.y - b.y}; }
template <typename T> V2<T> operator*(const V2<T> &a, const T &b) { return {a.x * b, a.y * b}; }
template <typename T> V2<T> operator/(const V2<T> &a, const T &b) { return {a.x / b, a.y / b}; }

// drjit dot(): a0*b0, then fmadd for the remaining entries.
template <typename T> T dot(const V3<T> &a, const V3<T> &b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
inline Dual dot(const V3d &a, const V3f &b) { return dot(a, V3d(b)); }
inline Dual dot(const V3f &a, const V3d &b) { return dot(V3d(a), b); }
template <typename T> T dot(const V2<T> &a, const V2<T> &b) { return fma_(a.y, b.y, a.x * b.x); }
inline Dual dot(const V2d &a, const V2f &b) { return fma_(a.y, Dual(b.y), a.x * Dual(b.x)); }
// drjit cross(): fmsub(a.yzx, b.zxy, a.zxy * b.yzx)
template <typename T> V3<T> cross(const V3<T> &a, const V3<T> &b) {
    return {fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x))};
}
inline V3d cross(const V3d &a, const V3f &b) { return cross(a, V3d(b)); }
inline V3d cross(const V3f &a, const V3d &b) { return cross(V3d(a), b); }
template <typename T> T squared_norm(const V3<T> &a) { return dot(a, a); }
template <typename T> T norm(const V3<T> &a) { return sqrt_(dot(a, a)); }
template <typename T> T norm(const V2<T> &a) { return sqrt_(dot(a, a)); }
// drjit normalize(): a * rsqrt(squared_norm(a)); rsqrt restated as IEEE 1/sqrt.
template <typename T> V3<T> normalize(const V3<T> &a) { return a * rcp(sqrt_(dot(a, a))); }

// ---------------------------------------------------------------- 4x4 matrices (row-major)
template <typename T> struct M4 {
    T m[4][4];
    M4() { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m[i][j] = T(i == j ? 1.f : 0.f); }
};
using M4f = M4<float>; using M4d = M4<Dual>;
inline M4f detach(const M4d &a) { M4f r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[i][j].v; return r; }
inline M4f detach(const M4f &a) { return a; }
inline M4d promote(const M4f &a) { M4d r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = Dual(a.m[i][j]); return r; }
inline M4d make_m4d(const float *v, const float *d) {
    M4d r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = Dual(v[i * 4 + j], d ? d[i * 4 + j] : 0.f); return r;
}
template <typename T> M4<T> operator*(const M4<T> &a, const M4<T> &b) {
    M4<T> r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            T s = a.m[i][0] * b.m[0][j];
            for (int k = 1; k < 4; ++k) s = fma_(a.m[i][k], b.m[k][j], s);
            r.m[i][j] = s;
        }
    return r;
}
// general inverse by cofactors (drjit::inverse for 4x4 is a closed-form cofactor expansion)
template <typename T> M4<T> inverse(const M4<T> &A) {
    const T (*a)[4] = A.m;
    T s0 = a[0][0] * a[1][1] - a[1][0] * a[0][1];
    T s1 = a[0][0] * a[1][2] - a[1][0] * a[0][2];
    T s2 = a[0][0] * a[1][3] - a[1][0] * a[0][3];
    T s3 = a[0][1] * a[1][2] - a[1][1] * a[0][2];
    T s4 = a[0][1] * a[1][3] - a[1][1] * a[0][3];
    T s5 = a[0][2] * a[1][3] - a[1][2] * a[0][3];
    T c5 = a[2][2] * a[3][3] - a[3][2] * a[2][3];
    T c4 = a[2][1] * a[3][3] - a[3][1] * a[2][3];
    T c3 = a[2][1] * a[3][2] - a[3][1] * a[2][2];
    T c2 = a[2][0] * a[3][3] - a[3][0] * a[2][3];
    T c1 = a[2][0] * a[3][2] - a[3][0] * a[2][2];
    T c0 = a[2][0] * a[3][1] - a[3][0] * a[2][1];
    T det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    T id = rcp(det);
    M4<T> r;
    r.m[0][0] = ( a[1][1] * c5 - a[1][2] * c4 + a[1][3] * c3) * id;
    r.m[0][1] = (-a[0][1] * c5 + a[0][2] * c4 - a[0][3] * c3) * id;
    r.m[0][2] = ( a[3][1] * s5 - a[3][2] * s4 + a[3][3] * s3) * id;
    r.m[0][3] = (-a[2][1] * s5 + a[2][2] * s4 - a[2][3] * s3) * id;
    r.m[1][0] = (-a[1][0] * c5 + a[1][2] * c2 - a[1][3] * c1) * id;
    r.m[1][1] = ( a[0][0] * c5 - a[0][2] * c2 + a[0][3] * c1) * id;
    r.m[1][2] = (-a[3][0] * s5 + a[3][2] * s2 - a[3][3] * s1) * id;
    r.m[1][3] = ( a[2][0] * s5 - a[2][2] * s2 + a[2][3] * s1) * id;
    r.m[2][0] = ( a[1][0] * c4 - a[1][1] * c2 + a[1][3] * c0) * id;
    r.m[2][1] = (-a[0][0] * c4 + a[0][1] * c2 - a[0][3] * c0) * id;
    r.m[2][2] = ( a[3][0] * s4 - a[3][1] * s2 + a[3][3] * s0) * id;
    r.m[2][3] = (-a[2][0] * s4 + a[2][1] * s2 - a[2][3] * s0) * id;
    r.m[3][0] = (-a[1][0] * c3 + a[1][1] * c1 - a[1][2] * c0) * id;
    r.m[3][1] = ( a[0][0] * c3 - a[0][1] * c1 + a[0][2] * c0) * id;
    r.m[3][2] = (-a[3][0] * s3 + a[3][1] * s1 - a[3][2] * s0) * id;
    r.m[3][3] = ( a[2][0] * s3 - a[2][1] * s1 + a[2][2] * s0) * id;
    return r;
}
template <typename T> T det3(const M4<T> &A) {
    const T (*a)[4] = A.m;
    return a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1])
         - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0])
         + a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
}
// reference include/psdr/core/transform.h:110-118
template <typename T> V3<T> transform_pos(const M4<T> &M, const V3<T> &p) {
    T r[4];
    for (int i = 0; i < 4; ++i) r[i] = fma_(M.m[i][2], p.z, fma_(M.m[i][1], p.y, M.m[i][0] * p.x)) + M.m[i][3];
    return V3<T>(r[0], r[1], r[2]) / r[3];
}
template <typename T> V3<T> transform_dir(const M4<T> &M, const V3<T> &p) {
    T r[3];
    for (int i = 0; i < 3; ++i) r[i] = fma_(M.m[i][2], p.z, fma_(M.m[i][1], p.y, M.m[i][0] * p.x));
    return V3<T>(r[0], r[1], r[2]);
}

} // namespace orc
