// ORACLE — TEST INFRASTRUCTURE ONLY (see num.h header).
//
// envmap.h — EnvironmentMap emitter (reference src/emitter/envmap.cpp:17-173), the lat-long Bitmap lookup it uses
// (src/core/bitmap.cpp:47-128, envmap_mode), HyperCubeDistribution<2> (src/core/cube_distrb.cpp:9-64) and
// ray_intersect_scene_aabb (include/psdr/utils.h:145-164).
//
// drjit's atan2 / acos / sincos are its own polynomial kernels and are not in /root/reference; the published Cephes
// single-precision algorithms are restated here with explicit fma (the HIP path states the same), derivatives are the
// analytic ones.
#pragma once
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>
#include "scene.h"

namespace orc {

void sincos_cephes(float xx, float &s_out, float &c_out);     // integrator.cpp

// ---------------------------------------------------------------- Cephes atanf / atan2f / asinf / acosf
inline float atan_cephes(float xx) {
    float x = std::fabs(xx), y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.f;
    const float z = x * x;
    const float p = fma_(fma_(fma_(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    y += fma_(p * z, x, x);
    return xx < 0.f ? -y : y;
}
inline float atan2_cephes(float y, float x) {
    const float PIF = 3.141592653589793f, PIO2F = 1.5707963267948966f;
    int code = 0;
    if (x < 0.f) code = 2;
    if (y < 0.f) code |= 1;
    if (x == 0.f) {
        if (code & 1) return -PIO2F;
        if (y == 0.f) return 0.f;
        return PIO2F;
    }
    if (y == 0.f) return (code & 2) ? PIF : 0.f;
    const float w = code == 2 ? PIF : (code == 3 ? -PIF : 0.f);
    return w + atan_cephes(y / x);
}
inline float asin_cephes(float xx) {
    float a = std::fabs(xx), x, z;
    bool flag = false;
    if (a > 1.0f) return 0.f;
    if (a < 1.0e-4f) return xx;
    if (a > 0.5f) { z = 0.5f * (1.0f - a); x = std::sqrt(z); flag = true; }
    else { x = a; z = x * x; }
    const float p = fma_(fma_(fma_(fma_(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z, 1.6666752422e-1f);
    z = fma_(p * z, x, x);
    if (flag) { z = z + z; z = 1.5707963267948966f - z; }
    return xx < 0.f ? -z : z;
}
inline float acos_cephes(float x) {
    if (x < -0.5f) return 3.141592653589793f - 2.0f * asin_cephes(std::sqrt(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin_cephes(std::sqrt(0.5f * (1.0f - x)));
    return 1.5707963267948966f - asin_cephes(x);
}
// drjit::safe_acos = acos(clamp(x, -1, 1)); d/dx = -1/sqrt(1 - x^2)
inline float safe_acos_(float x) { return acos_cephes(std::min(std::max(x, -1.f), 1.f)); }
inline Dual safe_acos_(const Dual &x) {
    const float c = std::min(std::max(x.v, -1.f), 1.f);
    return Dual(acos_cephes(c), -x.d / std::sqrt(fma_(-c, c, 1.f)));
}
inline float atan2_(float y, float x) { return atan2_cephes(y, x); }
inline Dual atan2_(const Dual &y, const Dual &x) {
    return Dual(atan2_cephes(y.v, x.v), fma_(x.v, y.d, -(y.v * x.d)) / fma_(x.v, x.v, y.v * y.v));
}
inline float floor_(float x) { return std::floor(x); }
inline Dual floor_(const Dual &x) { return Dual(std::floor(x.v), 0.f); }

// ---------------------------------------------------------------- Bitmap3fD (lat-long, envmap_mode) and the cell grid

// Bitmap::eval's uv transform, bitmap.cpp:64-74 (ad) = :76-86 (detached): m_rot, m_scale, m_trans (bitmap.h:37-39) are differentiable
// members; xf = {rotate, scale, translate.x, translate.y} as (value, tangent), R = float drops the tangents (detach)
template <typename R> inline R xf_part(const Dual &q) { if constexpr (std::is_same<R, Dual>::value) return q; else return q.v; }
template <typename R> inline void uv_transform_ref(const Dual xf[4], bool flip_v, R u, R v, R &x, R &y) {
    float sv, cv;
    sincos_cephes(xf[0].v, sv, cv);
    R sr, cr;
    if constexpr (std::is_same<R, Dual>::value) { sr = Dual(sv, cv * xf[0].d); cr = Dual(cv, -(sv * xf[0].d)); } else { sr = sv; cr = cv; }
    const R scale = xf_part<R>(xf[1]);
    x = (u - R(0.5f)) * cr + (v - R(0.5f)) * sr;
    y = -(u - R(0.5f)) * sr + (v - R(0.5f)) * cr;
    x = x + R(0.5f); y = y + R(0.5f);
    if (flip_v) y = -y;                                                           // flip the v coordinates to match common practices
    x = x * scale; y = y * scale;
    const R off = R(-.5f) + scale * R(0.5f);                                      // -.5f + m_scale / 2
    x = x - off; y = y + off;
    x = x + xf_part<R>(xf[2]); y = y + xf_part<R>(xf[3]);
}

// Bitmap<3>::eval<ad>(uv, flip_v = false, envmap_mode = true) (bitmap.cpp:47-128)
template <typename R> V3<R> envmap_bitmap_eval(const EnvmapC &E, R u, R v) {
    const int W = E.width, H = E.height;
    R x, y;
    uv_transform_ref<R>(E.uv_xf, false, u, v, x, y);
    x = x - R((float) (0.5 / W));                                                 // :83
    x = x - floor_(x); y = y - floor_(y);
    x = x * R((float) W); y = y * R((float) (H - 1));
    int px = (int) std::floor(detach(x)), py = (int) std::floor(detach(y));
    const R w1x = x - R((float) px), w1y = y - R((float) py), w0x = R(1.0f) - w1x, w0y = R(1.0f) - w1y;
    const int yw = std::min(py, H - 2) * W;
    const int xp1 = (px + 1) % W;
    const int last = W * H - 1;
    // texels carry the tangent d_data when one was given (m_radiance is a Bitmap3fD: differentiable)
    auto texel = [&](int i) -> V3<R> {
        i = std::min(std::max(i, 0), last);
        if constexpr (std::is_same<R, Dual>::value) {
            if (!E.d_data.empty())
                return V3<R>(Dual(E.data[3 * i], E.d_data[3 * i]), Dual(E.data[3 * i + 1], E.d_data[3 * i + 1]), Dual(E.data[3 * i + 2], E.d_data[3 * i + 2]));
        }
        return V3<R>(R(E.data[3 * i]), R(E.data[3 * i + 1]), R(E.data[3 * i + 2]));
    };
    const V3<R> v00 = texel(yw + px), v10 = texel(yw + xp1), v01 = texel(yw + px + W), v11 = texel(yw + xp1 + W);
    auto lerp3 = [](const R &a, const V3<R> &p, const R &b, const V3<R> &q) {
        return V3<R>(fma_(a, p.x, b * q.x), fma_(a, p.y, b * q.y), fma_(a, p.z, b * q.z));
    };
    const V3<R> v0 = lerp3(w0x, v00, w1x, v10), v1 = lerp3(w0x, v01, w1x, v11);
    return lerp3(w0y, v0, w1y, v1);
}

// Bitmap<CH>::eval<ad>(uv, flip_v = true, envmap_mode = false) for a resolution above 1x1 (bitmap.cpp:60-128): the texture
// lookup of Diffuse::m_reflectance (diffuse.cpp:38) and of Microfacet's three parameters (microfacet.cpp:38-45).
// Texels carry the tangent d_data.
template <bool ad> void tex_eval(const float *data, const float *d_data, int W, int H, int CH, const V2<Real<ad>> &uv, Real<ad> *out, const Dual xf[4]) {
    using R = Real<ad>;
    R x, y;
    uv_transform_ref<R>(xf, true, uv.x, uv.y, x, y);                              // flip_v
    x = x - floor_(x); y = y - floor_(y);
    x = x * R((float) (W - 1)); y = y * R((float) (H - 1));
    int px = (int) std::floor(detach(x)), py = (int) std::floor(detach(y));
    const R w1x = x - R((float) px), w1y = y - R((float) py), w0x = R(1.0f) - w1x, w0y = R(1.0f) - w1y;
    px = std::max(std::min(px, W - 2), 0); py = std::max(std::min(py, H - 2), 0);
    const int i00 = py * W + px, i10 = i00 + 1, i01 = i00 + W, i11 = i01 + 1;
    auto texel = [&](int i, int c) -> R {
        if constexpr (ad) return Dual(data[CH * i + c], d_data[CH * i + c]); else return data[CH * i + c];
    };
    for (int c = 0; c < CH; ++c) {
        const R v0 = fma_(w0x, texel(i00, c), w1x * texel(i10, c));
        const R v1 = fma_(w0x, texel(i01, c), w1x * texel(i11, c));
        out[c] = fma_(w0y, v0, w1y * v1);
    }
}
// 1x1 bitmaps return their value (bitmap.cpp:54-59)
template <bool ad> V3<Real<ad>> bsdf_reflectance(const BsdfC &b, const V2<Real<ad>> &uv) {
    using R = Real<ad>;
    if (b.tex_w == 0) {
        if constexpr (ad) return b.reflectance; else return detach(b.reflectance);
    }
    R out[3];
    tex_eval<ad>(b.tex.data(), b.d_tex.data(), b.tex_w, b.tex_h, 3, uv, out, b.uv_xf[0]);
    return V3<R>(out[0], out[1], out[2]);
}
// Microfacet: m_specularReflectance / m_roughness as bitmaps (value, tangent); the caller detaches in C mode
template <bool ad> V3d bsdf_specular(const BsdfC &b, const V2<Real<ad>> &uv) {
    if (b.spec_w == 0) return b.specular;
    Dual out[3];
    tex_eval<true>(b.spec_tex.data(), b.d_spec_tex.data(), b.spec_w, b.spec_h, 3, V2d(Dual(uv.x), Dual(uv.y)), out, b.uv_xf[1]);
    return V3d(out[0], out[1], out[2]);
}
template <bool ad> Dual bsdf_roughness(const BsdfC &b, const V2<Real<ad>> &uv) {
    if (b.rough_w == 0) return b.roughness;
    Dual out[1];
    tex_eval<true>(b.rough_tex.data(), b.d_rough_tex.data(), b.rough_w, b.rough_h, 1, V2d(Dual(uv.x), Dual(uv.y)), out, b.uv_xf[2]);
    return out[0];
}

// EnvironmentMap::configure (envmap.cpp:17-44)
inline void envmap_configure(EnvmapC &E) {
    if (!(E.width > 1 && E.height > 1)) throw std::runtime_error("EnvironmentMap: width > 1 && height > 1");
    const int w2 = (E.width - 1) << 1, h2 = (E.height - 1) << 1;
    E.reso[0] = w2; E.reso[1] = h2;
    E.num_cells = w2 * h2;
    E.unit[0] = 1.f / (float) w2; E.unit[1] = 1.f / (float) h2;
    std::vector<float> mass((size_t) E.num_cells);
    const float dtheta = Pi / (float) h2;
    for (int idx = 0; idx < E.num_cells; ++idx) {
        const int cx = idx / h2, cy = idx - cx * h2;                              // cube_distrb.cpp:22-29
        const float u = ((float) cx + .5f) * E.unit[0], v = ((float) cy + .5f) * E.unit[1];
        const V3f val = envmap_bitmap_eval<float>(E, u, v);
        const float theta = ((float) (idx % h2) + .5f) * dtheta;
        float s, c;
        sincos_cephes(theta, s, c);
        mass[idx] = (val.x * .2126f + val.y * .7152f + val.z * .0722f) * s;
    }
    E.cell_distrb.init(mass);
    E.from_world = inverse(E.to_world);
}

// EnvironmentMap::eval_direction (envmap.cpp:59-77)
template <bool ad> V3<Real<ad>> envmap_eval_direction(const EnvmapC &E, const V3<Real<ad>> &wi) {
    using R = Real<ad>;
    V3<R> v;
    if constexpr (ad) v = transform_dir(E.from_world, wi); else v = transform_dir(detach(E.from_world), wi);
    R u = atan2_(v.x, -v.z) * R(InvTwoPi), w = safe_acos_(v.y) * R(InvPi);
    u = u - floor_(u); w = w - floor_(w);
    if constexpr (ad) return envmap_bitmap_eval<R>(E, u, w) * E.scale; else return envmap_bitmap_eval<R>(E, u, w) * E.scale.v;
}

// HyperCubeDistribution<2>::sample_reuse / pdf (cube_distrb.cpp:42-64)
inline float envmap_cell_sample_reuse(const EnvmapC &E, float &sx, float &sy) {
    float pdf;
    const int idx = E.cell_distrb.sample_reuse(sy, pdf);
    const int cx = idx / E.reso[1], cy = idx - cx * E.reso[1];
    sx = (sx + (float) cx) * E.unit[0];
    sy = (sy + (float) cy) * E.unit[1];
    return pdf * (float) E.num_cells;
}
inline float envmap_cell_pdf(const EnvmapC &E, float u, float v) {
    const int ix = (int) std::floor(u * (float) E.reso[0]), iy = (int) std::floor(v * (float) E.reso[1]);
    if (!(ix >= 0 && ix < E.reso[0] && iy >= 0 && iy < E.reso[1])) return 0.f;
    const int idx = ix * E.reso[1] + iy;
    return (E.cell_distrb.pmf[idx] / E.cell_distrb.sum) * (float) E.num_cells;
}

// EnvironmentMap::sample_direction (envmap.cpp:118-132)
inline V3f envmap_sample_direction(const EnvmapC &E, float &sx, float &sy, float &pdf) {
    pdf = envmap_cell_sample_reuse(E, sx, sy);
    const float theta = sy * Pi, phi = sx * TwoPi;
    float st, ct, sp, cp;
    sincos_cephes(theta, st, ct);
    sincos_cephes(phi, sp, cp);
    const V3f d0(cp * st, sp * st, ct);                                           // sphdir, utils.h:56-61
    V3f d(d0.y, d0.z, -d0.x);
    const float inv_sin_theta = 1.f / std::sqrt(std::max(fma_(d.x, d.x, d.z * d.z), Epsilon * Epsilon));   // safe_rsqrt
    if (pdf > Epsilon) pdf *= inv_sin_theta * (.5f / (Pi * Pi));
    return transform_dir(detach(E.to_world), d);
}

// ray_intersect_scene_aabb<false> (utils.h:145-164)
inline void ray_intersect_scene_aabb(const V3f &o, const V3f &d, const V3f &lower, const V3f &upper, float &t, V3f &n, float &G) {
    const float t1[3] = {(lower.x - o.x) / d.x, (lower.y - o.y) / d.y, (lower.z - o.z) / d.z};
    const float t2[3] = {(upper.x - o.x) / d.x, (upper.y - o.y) / d.y, (upper.z - o.z) / d.z};
    const float dd[3] = {d.x, d.y, d.z};
    float t2p[3];
    for (int i = 0; i < 3; ++i) t2p[i] = std::fmax(t1[i], t2[i]);                 // drjit::maximum
    t = t2p[0];
    int idx = 0;
    for (int i = 1; i < 3; ++i) if (t2p[i] < t) { t = t2p[i]; idx = i; }          // argmin, utils.h:95-105
    float nn[3] = {0.f, 0.f, 0.f};
    const float sg = dd[idx] > 0.f ? 1.f : (dd[idx] < 0.f ? -1.f : dd[idx]);      // drjit::sign (copysign(1, x); +-0 irrelevant here)
    nn[idx] = -sg;
    n = V3f(nn[0], nn[1], nn[2]);
    G = dot(n, -d) * (1.f / (t * t));
}

} // namespace orc
