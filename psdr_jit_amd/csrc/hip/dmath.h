// dmath.h — device-side number types for the gfx950 kernels.
//
// The reference differentiates by running the whole path tracer on drjit DiffArray lanes and
// replaying a tape (reference include/psdr/types.h:23-40).  Here the derivative is carried in
// registers: Num<true> is a (value, tangent) pair for ONE forward direction, Num<false> is a bare
// float, and `detach()` (the reference's drjit::detach) just drops the tangent.  Vector helpers use
// the same fused forms drjit emits (dot = fma chain, cross = fma(a,b,-c*d)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PSDR_DEV __device__ __forceinline__
#define PSDR_HD __host__ __device__ __forceinline__

namespace psdr {

// WEIGHT ARITHMETIC (round 3).  An IEEE-rounded fp32 division is a ten-instruction, ~44-cycle sequence on gfx950 (v_div_scale x 2, v_rcp,
// four fma, v_div_fmas, v_div_fixup) and the primary-edge kernel of the README box carried 111 of them.  Two kinds of numbers flow
// through a path:
//   * GEOMETRY - hit points, directions, sample positions, everything a later ray, a comparison or a table index is computed from.
//     It stays correctly rounded (plain `/`, __builtin_sqrtf): one ulp in a direction moves the next hit, once in ~10^5 paths across an
//     edge, and the path - not its last bits - differs from the oracle's.  Measured with v_rcp / v_sqrt everywhere: renderD 7.58 -> 7.02 ms,
//     but 4 samples in a million take another path and the frame's relative L2 against the oracle goes from 1e-8 to 7e-4 (small test
//     frames: 1e-2).  Not taken.
//   * WEIGHTS - geometry terms, pdfs, MIS weights, throughput, radiance and EVERY TANGENT: they are multiplied into the result and
//     never decide anything.  They divide by multiplying with v_rcp_f32 (1 ulp): fdiv / frcp / div_ below, the tangent half of the Dual
//     operators.  x / 0 = +-inf, 0 / 0 = NaN, x / inf = 0 as in IEEE; NOT as in IEEE: v_rcp_f32 flushes a denormal divisor (x / 1e-40 = inf)
//     and returns 0 for a divisor above 2^126 - callers whose divisor can get there rescale first (shade.h::mis_weight).
// -DPSDR_EXACT_DIV restores the correctly rounded forms everywhere (measurement knob).
#ifdef PSDR_EXACT_DIV
__device__ __forceinline__ float frcp(float a) { return 1.f / a; }
__device__ __forceinline__ float fdiv(float a, float b) { return a / b; }
#else
__device__ __forceinline__ float frcp(float a) { return __builtin_amdgcn_rcpf(a); }
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#endif

struct Dual {
    float v, d;
    PSDR_DEV Dual() : v(0.f), d(0.f) {}
    PSDR_DEV Dual(float v_) : v(v_), d(0.f) {}
    PSDR_DEV Dual(float v_, float d_) : v(v_), d(d_) {}
};

template <bool AD> struct NumT { using type = float; };
template <> struct NumT<true> { using type = Dual; };
template <bool AD> using Num = typename NumT<AD>::type;

PSDR_DEV float detach(float x) { return x; }
PSDR_DEV float detach(const Dual &x) { return x.v; }
PSDR_DEV float tangent(float) { return 0.f; }
PSDR_DEV float tangent(const Dual &x) { return x.d; }

PSDR_DEV Dual operator+(const Dual &a, const Dual &b) { return Dual(a.v + b.v, a.d + b.d); }
PSDR_DEV Dual operator-(const Dual &a, const Dual &b) { return Dual(a.v - b.v, a.d - b.d); }
PSDR_DEV Dual operator-(const Dual &a) { return Dual(-a.v, -a.d); }
PSDR_DEV Dual operator*(const Dual &a, const Dual &b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
PSDR_DEV Dual operator/(const Dual &a, const Dual &b) {      // value: geometry (correctly rounded); tangent: weight arithmetic
    const float q = a.v / b.v;
    return Dual(q, (a.d - q * b.d) * frcp(b.v));
}
PSDR_DEV Dual operator*(const Dual &a, float b) { return Dual(a.v * b, a.d * b); }
PSDR_DEV Dual operator*(float a, const Dual &b) { return Dual(a * b.v, a * b.d); }
PSDR_DEV Dual operator+(const Dual &a, float b) { return Dual(a.v + b, a.d); }
PSDR_DEV Dual operator+(float a, const Dual &b) { return Dual(a + b.v, b.d); }
PSDR_DEV Dual operator-(const Dual &a, float b) { return Dual(a.v - b, a.d); }
PSDR_DEV Dual operator-(float a, const Dual &b) { return Dual(a - b.v, -b.d); }
PSDR_DEV Dual operator/(const Dual &a, float b) { return Dual(a.v / b, a.d * frcp(b)); }
PSDR_DEV Dual operator/(float a, const Dual &b) { return Dual(a) / b; }

PSDR_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PSDR_DEV Dual fma_(const Dual &a, const Dual &b, const Dual &c) {
    return Dual(__builtin_fmaf(a.v, b.v, c.v), a.d * b.v + a.v * b.d + c.d);
}
PSDR_DEV Dual fma_(const Dual &a, float b, const Dual &c) { return Dual(__builtin_fmaf(a.v, b, c.v), a.d * b + c.d); }
PSDR_DEV float sqrt_(float a) { return __builtin_sqrtf(a); }
PSDR_DEV Dual sqrt_(const Dual &a) {
    float s = __builtin_sqrtf(a.v);
    return Dual(s, a.d * frcp(2.f * s));
}
PSDR_DEV float safe_sqrt(float a) { return __builtin_sqrtf(a > 0.f ? a : 0.f); }
PSDR_DEV Dual safe_sqrt(const Dual &a) { return (a.v > 0.f) ? sqrt_(a) : Dual(0.f, 0.f); }
PSDR_DEV float abs_(float a) { return __builtin_fabsf(a); }
PSDR_DEV Dual abs_(const Dual &a) { return a.v < 0.f ? -a : a; }
PSDR_DEV float rcp_(float a) { return 1.f / a; }
// a / b of the WEIGHT arithmetic for either number type (plain `/` is the correctly rounded division of the geometry)
PSDR_DEV float div_(float a, float b) { return fdiv(a, b); }
PSDR_DEV Dual rcp_(const Dual &a) { return Dual(1.f) / a; }
PSDR_DEV Dual div_(const Dual &a, const Dual &b) { const float r = frcp(b.v), q = a.v * r; return Dual(q, (a.d - q * b.d) * r); }
PSDR_DEV Dual div_(const Dual &a, float b) { const float r = frcp(b); return Dual(a.v * r, a.d * r); }
PSDR_DEV Dual div_(float a, const Dual &b) { return div_(Dual(a), b); }
PSDR_DEV float sqr(float a) { return a * a; }
PSDR_DEV Dual sqr(const Dual &a) { return a * a; }
PSDR_DEV bool signbit_(float x) { return (__float_as_uint(x) >> 31) != 0u; }
PSDR_DEV float mulsign(float a, float b) { return signbit_(b) ? -a : a; }
PSDR_DEV Dual mulsign(const Dual &a, float b) { return signbit_(b) ? -a : a; }
PSDR_DEV bool finite_(float x) { return (__float_as_uint(x) & 0x7f800000u) != 0x7f800000u; }

// ------------------------------------------------------------------ 3-vectors
template <typename T> struct Vec3 {
    T x, y, z;
    PSDR_DEV Vec3() : x(0.f), y(0.f), z(0.f) {}
    PSDR_DEV Vec3(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
    PSDR_DEV explicit Vec3(T s) : x(s), y(s), z(s) {}
};
using Vec3f = Vec3<float>;
using Vec3d = Vec3<Dual>;
template <bool AD> using VecN = Vec3<Num<AD>>;

PSDR_DEV Vec3f detach(const Vec3f &a) { return a; }
PSDR_DEV Vec3f detach(const Vec3d &a) { return Vec3f(a.x.v, a.y.v, a.z.v); }
PSDR_DEV Vec3d make_dual(const Vec3f &v, const Vec3f &d) { return Vec3d(Dual(v.x, d.x), Dual(v.y, d.y), Dual(v.z, d.z)); }
PSDR_DEV Vec3d promote(const Vec3f &v) { return Vec3d(Dual(v.x), Dual(v.y), Dual(v.z)); }
PSDR_DEV Vec3f promote_f(const Vec3f &v) { return v; }

#define PSDR_V3OP(op) \
    template <typename T> PSDR_DEV Vec3<T> operator op(const Vec3<T> &a, const Vec3<T> &b) { return Vec3<T>(a.x op b.x, a.y op b.y, a.z op b.z); } \
    template <typename T> PSDR_DEV Vec3<T> operator op(const Vec3<T> &a, const T &b) { return Vec3<T>(a.x op b, a.y op b, a.z op b); }
PSDR_V3OP(+) PSDR_V3OP(-) PSDR_V3OP(*) PSDR_V3OP(/)
#undef PSDR_V3OP
template <typename T> PSDR_DEV Vec3<T> operator-(const Vec3<T> &a) { return Vec3<T>(-a.x, -a.y, -a.z); }
PSDR_DEV Vec3d operator*(const Vec3d &a, float b) { return Vec3d(a.x * b, a.y * b, a.z * b); }
PSDR_DEV Vec3d operator*(const Vec3f &a, const Dual &b) { return Vec3d(a.x * b, a.y * b, a.z * b); }
PSDR_DEV Vec3d operator-(const Vec3d &a, const Vec3f &b) { return Vec3d(a.x - b.x, a.y - b.y, a.z - b.z); }
PSDR_DEV Vec3d operator-(const Vec3f &a, const Vec3d &b) { return Vec3d(a.x - b.x, a.y - b.y, a.z - b.z); }

template <typename T> PSDR_DEV T dot(const Vec3<T> &a, const Vec3<T> &b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
PSDR_DEV Dual dot(const Vec3d &a, const Vec3f &b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
PSDR_DEV Dual dot(const Vec3f &a, const Vec3d &b) { return dot(b, a); }
template <typename T> PSDR_DEV Vec3<T> cross(const Vec3<T> &a, const Vec3<T> &b) {
    return Vec3<T>(fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)));
}
PSDR_DEV Vec3d cross(const Vec3d &a, const Vec3f &b) { return cross(a, promote(b)); }
template <typename T> PSDR_DEV T squared_norm(const Vec3<T> &a) { return dot(a, a); }
template <typename T> PSDR_DEV T norm(const Vec3<T> &a) { return sqrt_(dot(a, a)); }
template <typename T> PSDR_DEV Vec3<T> normalize(const Vec3<T> &a) { return a * rcp_(sqrt_(dot(a, a))); }
// v / s of the weight arithmetic
PSDR_DEV Vec3<float> vdiv_(const Vec3<float> &a, float b) { const float r = frcp(b); return Vec3<float>(a.x * r, a.y * r, a.z * r); }
PSDR_DEV Vec3<Dual> vdiv_(const Vec3<Dual> &a, const Dual &b) { const Dual r = div_(Dual(1.f), b); return Vec3<Dual>(a.x * r, a.y * r, a.z * r); }
PSDR_DEV Vec3<Dual> vdiv_(const Vec3<Dual> &a, float b) { const float r = frcp(b); return Vec3<Dual>(a.x * r, a.y * r, a.z * r); }
template <typename T> PSDR_DEV Vec3<T> madd3(const Vec3<T> &e1, const T &s, const Vec3<T> &e2, const T &t, const Vec3<T> &p0) {
    // reference utils.h:64-67 bilinear(): fmadd(e1, s, fmadd(e2, t, p0))
    return Vec3<T>(fma_(e1.x, s, fma_(e2.x, t, p0.x)), fma_(e1.y, s, fma_(e2.y, t, p0.y)), fma_(e1.z, s, fma_(e2.z, t, p0.z)));
}

// row-major 4x4 applied to points / directions (reference include/psdr/core/transform.h:110-118)
template <typename T> struct Mat4 { T m[16]; };
template <typename T> PSDR_DEV Vec3<T> xform_pos(const Mat4<T> &M, const Vec3<T> &p) {
    T r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fma_(M.m[4 * i + 2], p.z, fma_(M.m[4 * i + 1], p.y, M.m[4 * i] * p.x)) + M.m[4 * i + 3];
    return Vec3<T>(r[0], r[1], r[2]) / r[3];
}
template <typename T> PSDR_DEV Vec3<T> xform_dir(const Mat4<T> &M, const Vec3<T> &p) {
    T r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) r[i] = fma_(M.m[4 * i + 2], p.z, fma_(M.m[4 * i + 1], p.y, M.m[4 * i] * p.x));
    return Vec3<T>(r[0], r[1], r[2]);
}

} // namespace psdr
