// Arithmetic shared by the host (Scene::configure: cell masses of the environment map) and the HIP kernels
// (EnvironmentMap::eval / sample_position / sample_position_pdf): Cephes single-precision sincos / atan2 / acos with
// every multiply-add written as an explicit fma, the lat-long bitmap lookup, and the scene-box exit point.
// Compiled by g++ (host) and hipcc (device); nothing here depends on either side's vector types.
//
// Reference: src/emitter/envmap.cpp:17-173, src/core/bitmap.cpp:47-128 (envmap_mode), include/psdr/utils.h:56-61,
// 145-164.  drjit's own polynomial kernels for these functions are not part of the reference tree.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define PSDR_HD __host__ __device__ inline
#else
#define PSDR_HD inline
#endif

namespace psdr {
namespace env {

constexpr float kPi = 3.14159265358979323846f, kInvPi = 0.31830988618379067154f;
constexpr float kTwoPi = 6.28318530717958647692f, kInvTwoPi = 0.15915494309189533577f;
constexpr float kEps = 1e-5f;

PSDR_HD void sincos_f(float xx, float &s_out, float &c_out) {
    const float FOPI = 1.27323954473516f, DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
    float x = fabsf(xx);
    int j = (int) (FOPI * x);
    float y = (float) j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    float sign_s = xx < 0.f ? -1.f : 1.f, sign_c = 1.f;
    if (j > 3) { sign_s = -sign_s; sign_c = -sign_c; j -= 4; }
    if (j > 1) sign_c = -sign_c;
    x = fmaf(-y, DP1, x); x = fmaf(-y, DP2, x); x = fmaf(-y, DP3, x);
    const float z = x * x;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, x, x);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
    const bool swap = (j == 1) || (j == 2);
    s_out = sign_s * (swap ? pc : ps);
    c_out = sign_c * (swap ? ps : pc);
}

PSDR_HD float atan_f(float xx) {
    float x = fabsf(xx), y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.f;
    const float z = x * x;
    const float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    y += fmaf(p * z, x, x);
    return xx < 0.f ? -y : y;
}
PSDR_HD float atan2_f(float y, float x) {
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
    return w + atan_f(y / x);
}
PSDR_HD float asin_f(float xx) {
    float a = fabsf(xx), x, z;
    bool flag = false;
    if (a > 1.0f) return 0.f;
    if (a < 1.0e-4f) return xx;
    if (a > 0.5f) { z = 0.5f * (1.0f - a); x = sqrtf(z); flag = true; }
    else { x = a; z = x * x; }
    const float p = fmaf(fmaf(fmaf(fmaf(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z, 1.6666752422e-1f);
    z = fmaf(p * z, x, x);
    if (flag) { z = z + z; z = 1.5707963267948966f - z; }
    return xx < 0.f ? -z : z;
}
PSDR_HD float acos_f(float x) {
    if (x < -0.5f) return 3.141592653589793f - 2.0f * asin_f(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin_f(sqrtf(0.5f * (1.0f - x)));
    return 1.5707963267948966f - asin_f(x);
}
PSDR_HD float safe_acos_f(float x) { return acos_f(fminf(fmaxf(x, -1.f), 1.f)); }

// scalar hooks the templated lookup needs; the device adds overloads for its (value, tangent) type
PSDR_HD float e_fma(float a, float b, float c) { return fmaf(a, b, c); }
PSDR_HD float e_floor(float a) { return floorf(a); }
PSDR_HD float e_value(float a) { return a; }
PSDR_HD void e_sincos(float a, float &s, float &c) { sincos_f(a, s, c); }

// Bitmap::m_rot, m_scale, m_trans (bitmap.h:37-39, bound as rotate / scale / translate, psdr.cpp:204-206, 217-219): the uv transform
// every lookup applies (bitmap.cpp:64-86).  R = float, or the device's (value, tangent) type - the three are differentiable members.
template <typename R> struct UvXf {
    R rot, scale, tx, ty;
    PSDR_HD UvXf() : rot(0.f), scale(1.f), tx(0.f), ty(0.f) {}
    PSDR_HD UvXf(R r, R s, R x, R y) : rot(r), scale(s), tx(x), ty(y) {}
    PSDR_HD explicit UvXf(const float *p) : rot(p[0]), scale(p[1]), tx(p[2]), ty(p[3]) {}
};
// bitmap.cpp:64-86 (ad) = :76-86 (detached): rotate about the centre, flip v, scale about the centre, translate
template <typename R> PSDR_HD void uv_transform(const UvXf<R> &xf, bool flip_v, R u, R v, R &x, R &y) {
    R sr, cr;
    e_sincos(xf.rot, sr, cr);                                                     // cos(m_rot), sin(m_rot)
    x = (u - R(0.5f)) * cr + (v - R(0.5f)) * sr;
    y = -(u - R(0.5f)) * sr + (v - R(0.5f)) * cr;
    x = x + R(0.5f); y = y + R(0.5f);
    if (flip_v) y = -y;
    x = x * xf.scale; y = y * xf.scale;
    const R off = R(-.5f) + xf.scale * R(0.5f);                                   // -.5f + m_scale / 2
    x = x - off; y = y + off;
    x = x + xf.tx; y = y + xf.ty;
}

// Bitmap<3>::eval<ad>(uv, flip_v = false, envmap_mode = true) (bitmap.cpp:47-128).  data: [H*W*3] row-major rgb.
// texel(i, c) returns channel c of texel i as an R (so that (value, tangent) texels can be supplied)
template <typename R, typename TexelFn>
PSDR_HD void bitmap_eval_fn(TexelFn texel, int W, int H, R u, R v, R out[3], const UvXf<R> &xf = UvXf<R>()) {
    R x, y;
    uv_transform<R>(xf, false, u, v, x, y);
    x = x - R((float) (0.5 / W));
    x = x - e_floor(x); y = y - e_floor(y);
    x = x * R((float) W); y = y * R((float) (H - 1));
    const int px = (int) floorf(e_value(x)), py = (int) floorf(e_value(y));
    const R w1x = x - R((float) px), w1y = y - R((float) py), w0x = R(1.0f) - w1x, w0y = R(1.0f) - w1y;
    const int yw = (py < H - 2 ? py : H - 2) * W;
    const int xp1 = (px + 1) % W;
    const int last = W * H - 1;
    int i00 = yw + px, i10 = yw + xp1, i01 = yw + px + W, i11 = yw + xp1 + W;
    i00 = i00 < 0 ? 0 : (i00 > last ? last : i00); i10 = i10 < 0 ? 0 : (i10 > last ? last : i10);
    i01 = i01 < 0 ? 0 : (i01 > last ? last : i01); i11 = i11 < 0 ? 0 : (i11 > last ? last : i11);
    for (int c = 0; c < 3; ++c) {
        const R v0 = e_fma(w0x, texel(i00, c), w1x * texel(i10, c));
        const R v1 = e_fma(w0x, texel(i01, c), w1x * texel(i11, c));
        out[c] = e_fma(w0y, v0, w1y * v1);
    }
}
template <typename R>
PSDR_HD void bitmap_eval(const float *data, int W, int H, R u, R v, R out[3], const UvXf<R> &xf = UvXf<R>()) {
    bitmap_eval_fn<R>([&](int i, int c) { return R(data[3 * i + c]); }, W, H, u, v, out, xf);
}
// the four texels and weights bitmap_eval reads at (u, v) (envmap mode): d out[c] / d texel[idx[k]][c] = w[k]
PSDR_HD void bitmap_footprint_env(int W, int H, float u, float v, int idx[4], float w[4], const UvXf<float> &xf = UvXf<float>()) {
    float x, y;
    uv_transform<float>(xf, false, u, v, x, y);
    x = x - (float) (0.5 / W);
    x = x - floorf(x); y = y - floorf(y);
    x = x * (float) W; y = y * (float) (H - 1);
    const int px = (int) floorf(x), py = (int) floorf(y);
    const float w1x = x - (float) px, w1y = y - (float) py, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    const int yw = (py < H - 2 ? py : H - 2) * W;
    const int xp1 = (px + 1) % W;
    const int last = W * H - 1;
    int i00 = yw + px, i10 = yw + xp1, i01 = yw + px + W, i11 = yw + xp1 + W;
    idx[0] = i00 < 0 ? 0 : (i00 > last ? last : i00); idx[1] = i10 < 0 ? 0 : (i10 > last ? last : i10);
    idx[2] = i01 < 0 ? 0 : (i01 > last ? last : i01); idx[3] = i11 < 0 ? 0 : (i11 > last ? last : i11);
    w[0] = w0y * w0x; w[1] = w0y * w1x; w[2] = w1y * w0x; w[3] = w1y * w1x;
}

// Bitmap<CH>::eval<ad>(uv, flip_v, envmap_mode = false) (bitmap.cpp:47-128): the texture lookup of Diffuse::m_reflectance
// (diffuse.cpp:38, flip_v = true).  texel(i, c) returns channel c of texel i as an R (so that a (value, tangent) texel can
// be supplied).
template <typename R, int CH = 3, typename TexelFn>
PSDR_HD void bitmap_eval_tex(TexelFn texel, int W, int H, R u, R v, bool flip_v, R *out, const UvXf<R> &xf = UvXf<R>()) {
    R x, y;
    uv_transform<R>(xf, flip_v, u, v, x, y);
    x = x - e_floor(x); y = y - e_floor(y);
    x = x * R((float) (W - 1)); y = y * R((float) (H - 1));
    int px = (int) floorf(e_value(x)), py = (int) floorf(e_value(y));
    const R w1x = x - R((float) px), w1y = y - R((float) py), w0x = R(1.0f) - w1x, w0y = R(1.0f) - w1y;
    px = px < W - 2 ? px : W - 2; py = py < H - 2 ? py : H - 2;
    px = px < 0 ? 0 : px; py = py < 0 ? 0 : py;
    const int i00 = py * W + px, i10 = i00 + 1, i01 = i00 + W, i11 = i01 + 1;
    for (int c = 0; c < CH; ++c) {
        const R v0 = e_fma(w0x, texel(i00, c), w1x * texel(i10, c));
        const R v1 = e_fma(w0x, texel(i01, c), w1x * texel(i11, c));
        out[c] = e_fma(w0y, v0, w1y * v1);
    }
}

// the four texels and bilinear weights bitmap_eval_tex reads at (u, v): d out[c] / d texel[idx[k]][c] = w[k]
PSDR_HD void bitmap_footprint(int W, int H, float u, float v, bool flip_v, int idx[4], float w[4], const UvXf<float> &xf = UvXf<float>()) {
    float x, y;
    uv_transform<float>(xf, flip_v, u, v, x, y);
    x = x - floorf(x); y = y - floorf(y);
    x = x * (float) (W - 1); y = y * (float) (H - 1);
    int px = (int) floorf(x), py = (int) floorf(y);
    const float w1x = x - (float) px, w1y = y - (float) py, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    px = px < W - 2 ? px : W - 2; py = py < H - 2 ? py : H - 2;
    px = px < 0 ? 0 : px; py = py < 0 ? 0 : py;
    idx[0] = py * W + px; idx[1] = idx[0] + 1; idx[2] = idx[0] + W; idx[3] = idx[2] + 1;
    w[0] = w0y * w0x; w[1] = w0y * w1x; w[2] = w1y * w0x; w[3] = w1y * w1x;
}

// mass of cell idx of HyperCubeDistribution2f (envmap.cpp:28-31, cube_distrb.cpp:22-29): luminance * sin(theta)
PSDR_HD float cell_mass(const float *data, int W, int H, int w2, int h2, int idx, const UvXf<float> &xf = UvXf<float>()) {
    const int cx = idx / h2, cy = idx - cx * h2;
    const float ux = 1.f / (float) w2, uy = 1.f / (float) h2;
    const float u = ((float) cx + .5f) * ux, v = ((float) cy + .5f) * uy;
    float val[3];
    bitmap_eval<float>(data, W, H, u, v, val, xf);
    const float theta = ((float) (idx % h2) + .5f) * (kPi / (float) h2);
    float s, c;
    sincos_f(theta, s, c);
    return (val[0] * .2126f + val[1] * .7152f + val[2] * .0722f) * s;
}

// ray_intersect_scene_aabb<false> (utils.h:145-164): exit point of a ray that starts inside the box
PSDR_HD void scene_aabb_exit(const float o[3], const float d[3], const float lower[3], const float upper[3], float &t, float n[3], float &G) {
    float t2p[3];
    for (int i = 0; i < 3; ++i) t2p[i] = fmaxf((lower[i] - o[i]) / d[i], (upper[i] - o[i]) / d[i]);
    t = t2p[0];
    int idx = 0;
    for (int i = 1; i < 3; ++i) if (t2p[i] < t) { t = t2p[i]; idx = i; }
    n[0] = n[1] = n[2] = 0.f;
    const float di = idx == 0 ? d[0] : (idx == 1 ? d[1] : d[2]);
    const float sg = di > 0.f ? 1.f : (di < 0.f ? -1.f : di);
    if (idx == 0) n[0] = -sg; else if (idx == 1) n[1] = -sg; else n[2] = -sg;
    const float ndd = fmaf(n[2], -d[2], fmaf(n[1], -d[1], n[0] * -d[0]));
    G = ndd * (1.f / (t * t));
}

} // namespace env
} // namespace psdr
