// hnum.h — host-side (value, tangent) arithmetic used while configuring a scene.
//
// The reference keeps scene parameters as drjit DiffArrays and lets the tape differentiate
// Mesh::configure / PerspectiveCamera::configure (reference src/shape/mesh.cpp:317-382,
// src/sensor/perspective.cpp:10-152).  The host code here pushes ONE forward tangent through the same
// formulas so that the device snapshot carries d(TriangleInfo)/d(theta), d(edges)/d(theta) and
// d(camera)/d(theta) next to the values.
#pragma once
#include <array>
#include <cmath>

// The same arithmetic runs on the device (round 6: csrc/hip/scene_build.hip computes a moved mesh's rows there - process_mesh, reference src/shape/mesh.cpp:23-62 - and the
// rows must be the bits the host would have written): under hipcc every function below is __host__ __device__; both compilers run with -ffp-contract=off, every fused
// multiply-add is an explicit fmaf, and division and square root are correctly rounded on both sides.
#if defined(__HIPCC__)
#define PSDR_HNUM_HD __host__ __device__
#else
#define PSDR_HNUM_HD
#endif

namespace psdr_host {

struct DF {              // dual float
    float v = 0.f, d = 0.f;
    PSDR_HNUM_HD DF() {}
    PSDR_HNUM_HD DF(float v_) : v(v_) {}
    PSDR_HNUM_HD DF(float v_, float d_) : v(v_), d(d_) {}
};
PSDR_HNUM_HD inline DF operator+(DF a, DF b) { return {a.v + b.v, a.d + b.d}; }
PSDR_HNUM_HD inline DF operator-(DF a, DF b) { return {a.v - b.v, a.d - b.d}; }
PSDR_HNUM_HD inline DF operator-(DF a) { return {-a.v, -a.d}; }
PSDR_HNUM_HD inline DF operator*(DF a, DF b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
PSDR_HNUM_HD inline DF operator/(DF a, DF b) { float q = a.v / b.v; return {q, (a.d - q * b.d) / b.v}; }
PSDR_HNUM_HD inline DF dfma(DF a, DF b, DF c) { return {std::fmaf(a.v, b.v, c.v), a.d * b.v + a.v * b.d + c.d}; }
PSDR_HNUM_HD inline DF dsqrt(DF a) { float s = std::sqrt(a.v); return {s, a.d / (2.f * s)}; }
PSDR_HNUM_HD inline DF drcp(DF a) { return DF(1.f) / a; }

struct D3 { DF x, y, z; };
PSDR_HNUM_HD inline D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PSDR_HNUM_HD inline D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PSDR_HNUM_HD inline D3 operator*(D3 a, DF s) { return {a.x * s, a.y * s, a.z * s}; }
PSDR_HNUM_HD inline D3 operator/(D3 a, DF s) { return {a.x / s, a.y / s, a.z / s}; }
PSDR_HNUM_HD inline DF ddot(D3 a, D3 b) { return dfma(a.z, b.z, dfma(a.y, b.y, a.x * b.x)); }
PSDR_HNUM_HD inline D3 dcross(D3 a, D3 b) {
    return {dfma(a.y, b.z, -(a.z * b.y)), dfma(a.z, b.x, -(a.x * b.z)), dfma(a.x, b.y, -(a.y * b.x))};
}
PSDR_HNUM_HD inline DF dnorm(D3 a) { return dsqrt(ddot(a, a)); }
PSDR_HNUM_HD inline D3 dnormalize(D3 a) { return a * drcp(dsqrt(ddot(a, a))); }

struct DM4 {             // row-major 4x4 of duals
    DF m[4][4];
    PSDR_HNUM_HD static DM4 identity() { DM4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = DF(i == j ? 1.f : 0.f); return r; }
    static DM4 from(const std::array<float, 16> &v, const std::array<float, 16> &d) {
        DM4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = DF(v[4 * i + j], d[4 * i + j]); return r;
    }
    void split(float *v, float *d) const { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { v[4 * i + j] = m[i][j].v; d[4 * i + j] = m[i][j].d; } }
};
PSDR_HNUM_HD inline DM4 operator*(const DM4 &a, const DM4 &b) {
    DM4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            DF s = a.m[i][0] * b.m[0][j];
            for (int k = 1; k < 4; ++k) s = dfma(a.m[i][k], b.m[k][j], s);
            r.m[i][j] = s;
        }
    return r;
}
PSDR_HNUM_HD inline D3 xform_pos(const DM4 &M, D3 p) {
    DF r[4];
    for (int i = 0; i < 4; ++i) r[i] = dfma(M.m[i][2], p.z, dfma(M.m[i][1], p.y, M.m[i][0] * p.x)) + M.m[i][3];
    return D3{r[0], r[1], r[2]} / r[3];
}
PSDR_HNUM_HD inline D3 xform_dir(const DM4 &M, D3 p) {
    DF r[3];
    for (int i = 0; i < 3; ++i) r[i] = dfma(M.m[i][2], p.z, dfma(M.m[i][1], p.y, M.m[i][0] * p.x));
    return {r[0], r[1], r[2]};
}
// closed-form cofactor inverse
PSDR_HNUM_HD inline DM4 inverse(const DM4 &A) {
    const DF (*a)[4] = A.m;
    DF s0 = a[0][0] * a[1][1] - a[1][0] * a[0][1], s1 = a[0][0] * a[1][2] - a[1][0] * a[0][2], s2 = a[0][0] * a[1][3] - a[1][0] * a[0][3];
    DF s3 = a[0][1] * a[1][2] - a[1][1] * a[0][2], s4 = a[0][1] * a[1][3] - a[1][1] * a[0][3], s5 = a[0][2] * a[1][3] - a[1][2] * a[0][3];
    DF c5 = a[2][2] * a[3][3] - a[3][2] * a[2][3], c4 = a[2][1] * a[3][3] - a[3][1] * a[2][3], c3 = a[2][1] * a[3][2] - a[3][1] * a[2][2];
    DF c2 = a[2][0] * a[3][3] - a[3][0] * a[2][3], c1 = a[2][0] * a[3][2] - a[3][0] * a[2][2], c0 = a[2][0] * a[3][1] - a[3][0] * a[2][1];
    DF det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    DF id = drcp(det);
    DM4 r;
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
PSDR_HNUM_HD inline float det3(const DM4 &A) {
    auto a = [&](int i, int j) { return A.m[i][j].v; };
    return a(0, 0) * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) - a(0, 1) * (a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0)) + a(0, 2) * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0));
}

} // namespace psdr_host
