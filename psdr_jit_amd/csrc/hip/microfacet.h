// microfacet.h — Microfacet BSDF (reference src/bsdf/microfacet.cpp:18-134: Lambertian diffuse + GGX specular with the
// 2^(...) Schlick-style Fresnel) and GGXDistribution (src/bsdf/ggx.cpp:8-107, visible-normal sampling).
// drjit::pow(2, x) = the Cephes exp2f polynomial with explicit fma (as in the oracle); drjit::rsqrt = 1/sqrt.
// Only instantiated in the LDS=false kernels (shade.h).
#pragma once
#include "scene_dev.h"

namespace psdr {

PSDR_DEV float exp2_cephes(float x) {
    if (x > 127.f) return __builtin_inff();
    if (x < -127.f) return 0.f;
    float px = floorf(x);
    int i0 = (int) px;
    x = x - px;
    if (x > 0.5f) { i0 += 1; x = x - 1.f; }
    px = fma_(fma_(fma_(fma_(fma_(1.535336188319500e-4f, x, 1.339887440266574e-3f), x, 9.618437357674640e-3f), x, 5.550332471162809e-2f), x,
                   2.402264791363012e-1f), x, 6.931472028550421e-1f);
    px = fma_(px, x, 1.0f);
    return ldexpf(px, i0);
}
PSDR_DEV float exp2_(float x) { return exp2_cephes(x); }
PSDR_DEV Dual exp2_(const Dual &x) { const float v = exp2_cephes(x.v); return Dual(v, v * 0.6931471805599453f * x.d); }

template <typename R> struct GGX {
    R au, av;                                                                      // alpha_u, alpha_v
    PSDR_DEV R eval(const Vec3<R> &m) const {                                     // ggx.cpp:13-33
        const R alpha_uv = au * av;
        const R r = rcp_(R(kPi) * alpha_uv * sqr(sqr(m.x / au) + sqr(m.y / av) + sqr(m.z)));
        return (detach(r) * detach(m.z) > 1e-20f) ? r : R(0.f);
    }
    PSDR_DEV R smith_g1(const Vec3<R> &v, const Vec3<R> &m) const {                // ggx.cpp:84-97
        const R xy_alpha_2 = sqr(au * v.x) + sqr(av * v.y);
        const R tan_theta_alpha_2 = xy_alpha_2 / sqr(v.z);
        R result = R(2.f) / (R(1.f) + sqrt_(R(1.f) + tan_theta_alpha_2));
        if (detach(xy_alpha_2) == 0.f) result = R(1.f);
        if (detach(dot(v, m)) * detach(v.z) <= 0.f) result = R(0.f);
        return result;
    }
};

PSDR_DEV void sincos_cephes(float xx, float &s_out, float &c_out);                // shade.h

// ggx.cpp:99-107 on warp.h:16-52
PSDR_DEV void ggx_sample_visible_11(float cos_theta_i, float sx, float sy, float &slope_x, float &slope_y) {
    const float x0 = fma_(2.f, sx, -1.f), y0 = fma_(2.f, sy, -1.f);
    const bool is_zero = (x0 == 0.f) && (y0 == 0.f), q13 = fabsf(x0) < fabsf(y0);
    const float r = q13 ? y0 : x0, rp = q13 ? x0 : y0;
    float phi = .25f * kPi * rp / r;
    if (q13) phi = .5f * kPi - phi;
    if (is_zero) phi = 0.f;
    float sn, cs;
    sincos_cephes(phi, sn, cs);
    const float px = r * cs;
    float py = r * sn;
    const float s = .5f * (1.f + cos_theta_i);
    const float a0 = safe_sqrt(1.f - sqr(px));
    py = fma_(py, s, fma_(-a0, s, a0));                                           // drjit::lerp
    const float z = safe_sqrt(1.f - fma_(py, py, px * px));
    const float sin_theta_i = safe_sqrt(1.f - sqr(cos_theta_i));
    const float norm = 1.f / fma_(sin_theta_i, py, cos_theta_i * z);
    slope_x = fma_(cos_theta_i, py, -(sin_theta_i * z)) * norm;
    slope_y = px * norm;
}
// ggx.cpp:35-82 (detached)
PSDR_DEV Vec3f ggx_sample(float au, float av, const Vec3f &wi, float sx, float sy, float &pdf) {
    const Vec3f wi_p = normalize(Vec3f(au * wi.x, av * wi.y, wi.z));
    const float sin_theta_2 = fma_(wi_p.x, wi_p.x, sqr(wi_p.y));
    const float inv_sin_theta = 1.f / sqrtf(sin_theta_2);
    const bool deg = fabsf(sin_theta_2) <= 4.f * kEpsilon;
    const float sin_phi = deg ? 0.f : fminf(fmaxf(wi_p.y * inv_sin_theta, -1.f), 1.f);
    const float cos_phi = deg ? 1.f : fminf(fmaxf(wi_p.x * inv_sin_theta, -1.f), 1.f);
    float slx, sly;
    ggx_sample_visible_11(wi_p.z, sx, sy, slx, sly);
    const float s0 = fma_(cos_phi, slx, -(sin_phi * sly)) * au, s1 = fma_(sin_phi, slx, cos_phi * sly) * av;
    const Vec3f m = normalize(Vec3f(-s0, -s1, 1.f));
    GGX<float> g{au, av};
    pdf = g.smith_g1(wi, m) * fabsf(dot(wi, m)) * g.eval(m) / fabsf(wi.z);
    return m;
}

// microfacet.cpp:22-62
template <typename R>
PSDR_DEV Vec3<R> microfacet_eval(const Vec3<R> &spec, const Vec3<R> &diff, const R &roughness, bool two_sided, Vec3<R> wi, Vec3<R> wo, bool active) {
    using V = Vec3<R>;
    if (two_sided) { wo.z = mulsign(wo.z, detach(wi.z)); wi.z = abs_(wi.z); }
    const R cos_theta_nv = wi.z, cos_theta_nl = wo.z;
    if (!(active && detach(cos_theta_nv) > 0.f && detach(cos_theta_nl) > 0.f)) return V(R(0.f));
    const V diffuse = diff * R(kInvPi);
    const V H = normalize(wi + wo);
    const R cos_theta_vh = dot(H, wi);
    GGX<R> distr{sqr(roughness), sqr(roughness)};
    const R ggx = distr.eval(H);
    const R coeff = cos_theta_vh * (R(-5.55473f) * cos_theta_vh - R(6.8316f));
    const V fresnel = spec + (V(R(1.f)) - spec) * exp2_(coeff);
    const R smithG = distr.smith_g1(wi, H) * distr.smith_g1(wo, H);
    const V numerator = fresnel * (ggx * smithG);
    const R denominator = R(4.f) * cos_theta_nl * cos_theta_nv;
    const V specular = numerator / (denominator + R(1e-6f));
    return (diffuse + specular) * cos_theta_nl;
}
// microfacet.cpp:108-131
// (alpha_u, alpha_v) = (roughness^2, roughness^2) for Microfacet; RoughConductor::__pdf (roughconductor.cpp:70-90) is the same
PSDR_DEV float ggx_pdf(float au, float av, bool two_sided, Vec3f wi, Vec3f wo, bool active) {
    if (two_sided) { wo.z = mulsign(wo.z, wi.z); wi.z = fabsf(wi.z); }
    const Vec3f m = normalize(wo + wi);
    if (!(active && wi.z > 0.f && wo.z > 0.f && dot(wi, m) > 0.f && dot(wo, m) > 0.f)) return 0.f;
    GGX<float> distr{au, av};
    return distr.eval(m) * distr.smith_g1(wi, m) / (4.f * wi.z);
}
// microfacet.cpp:75-98: the first two of the three sample numbers; the direction stays in the upper hemisphere
// (also RoughConductor::__sample, roughconductor.cpp:92-116)
PSDR_DEV void ggx_reflect_sample(float au, float av, bool two_sided, Vec3f wi, float s0, float s1, bool active, Vec3f &wo, float &pdf, bool &valid) {
    if (two_sided) wi.z = fabsf(wi.z);
    float m_pdf;
    const Vec3f m = ggx_sample(au, av, wi, s0, s1, m_pdf);
    const float k = 2.f * dot(wi, m);
    wo = Vec3f(fma_(m.x, k, -wi.x), fma_(m.y, k, -wi.y), fma_(m.z, k, -wi.z));
    pdf = m_pdf / (4.f * dot(wo, m));
    valid = active && (wi.z > 0.f) && (pdf != 0.f) && (wo.z > 0.f);
}

// conductor Fresnel, reference include/psdr/utils.h:166-182 (per colour channel)
template <typename R> PSDR_DEV R fresnel_conductor(const R &eta_r, const R &eta_i, const R &cos_theta_i) {
    const R cos_theta_i_2 = sqr(cos_theta_i), sin_theta_i_2 = R(1.f) - cos_theta_i_2, sin_theta_i_4 = sqr(sin_theta_i_2);
    const R temp_1 = sqr(eta_r) - sqr(eta_i) - sin_theta_i_2;
    const R a_2_pb_2 = safe_sqrt(sqr(temp_1) + R(4.f) * sqr(eta_i * eta_r));
    const R a = safe_sqrt(R(.5f) * (a_2_pb_2 + temp_1));
    const R term_1 = a_2_pb_2 + cos_theta_i_2, term_2 = R(2.f) * cos_theta_i * a;
    const R r_s = (term_1 - term_2) / (term_1 + term_2);
    const R term_3 = a_2_pb_2 * cos_theta_i_2 + sin_theta_i_4, term_4 = term_2 * sin_theta_i_2;
    const R r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return R(.5f) * (r_s + r_p);
}

// RoughConductor::__eval, reference src/bsdf/roughconductor.cpp:30-68
template <typename R>
PSDR_DEV Vec3<R> conductor_eval(const R &au, const R &av, const Vec3<R> &eta, const Vec3<R> &k, const Vec3<R> &spec, bool two_sided,
                                Vec3<R> wi, Vec3<R> wo, bool active) {
    using V = Vec3<R>;
    if (two_sided) { wo.z = mulsign(wo.z, detach(wi.z)); wi.z = abs_(wi.z); }
    if (!(active && detach(wi.z) > 0.f && detach(wo.z) > 0.f)) return V(R(0.f));
    GGX<R> distr{au, av};
    const V H = normalize(wo + wi);
    const R D = distr.eval(H);
    if (detach(D) == 0.f) return V(R(0.f));
    const R G = distr.smith_g1(wi, H) * distr.smith_g1(wo, H);
    const R result = D * G / (R(4.f) * wi.z);
    const R c = dot(wi, H);
    const V F(fresnel_conductor<R>(eta.x, k.x, c), fresnel_conductor<R>(eta.y, k.y, c), fresnel_conductor<R>(eta.z, k.z, c));
    return F * result * spec;
}

// MicrofacetPerVertex::__eval, reference src/bsdf/microfacet_pv.cpp:21-66 (its own GGX / Smith-Schlick terms)
template <typename R>
PSDR_DEV Vec3<R> microfacet_pv_eval(const Vec3<R> &spec, const Vec3<R> &diff, const R &roughness, bool two_sided, Vec3<R> wi, Vec3<R> wo, bool active) {
    using V = Vec3<R>;
    if (two_sided) { wo.z = mulsign(wo.z, detach(wi.z)); wi.z = abs_(wi.z); }
    const R cos_theta_nv = wi.z, cos_theta_nl = wo.z;
    if (!(active && detach(cos_theta_nv) > 0.f && detach(cos_theta_nl) > 0.f)) return V(R(0.f));
    const V diffuse = diff * R(kInvPi);
    const V H = normalize(wi + wo);
    const R cos_theta_nh = H.z, cos_theta_vh = dot(H, wi);
    const R alpha = sqr(roughness);
    const R k = sqr(roughness + R(1.f)) / R(8.f);
    const R tmp = alpha / (cos_theta_nh * cos_theta_nh * (sqr(alpha) - R(1.f)) + R(1.f));
    const R ggx = tmp * tmp * R(kInvPi);
    const R coeff = cos_theta_vh * (R(-5.55473f) * cos_theta_vh - R(6.8316f));
    const V fresnel = spec + (V(R(1.f)) - spec) * exp2_(coeff);
    const R smithG1 = cos_theta_nv / (cos_theta_nv * (R(1.f) - k) + k);
    const R smithG2 = cos_theta_nl / (cos_theta_nl * (R(1.f) - k) + k);
    const R smithG = smithG1 * smithG2;
    const V numerator = fresnel * (ggx * smithG);
    const R denominator = R(4.f) * cos_theta_nl * cos_theta_nv;
    const V specular = numerator / (denominator + R(1e-6f));
    return (diffuse + specular) * cos_theta_nl;
}

// ---------------------------------------------------------------- RoughDielectric, reference src/bsdf/roughdielectric.cpp:35-237
// fresnel_dielectric, reference include/psdr/utils.h:184-215
template <typename R> PSDR_DEV void fresnel_dielectric(const R &eta, const R &cos_theta_i, R &F, R &cos_theta_t, R &eta_it, R &eta_ti) {
    const bool outside = detach(cos_theta_i) >= 0.f;
    const R rcp_eta = rcp_(eta);
    eta_it = outside ? eta : rcp_eta;
    eta_ti = outside ? rcp_eta : eta;
    const R cos_theta_t_sqr = fma_(-fma_(-cos_theta_i, cos_theta_i, R(1.f)), eta_ti * eta_ti, R(1.f));
    const R cos_theta_i_abs = abs_(cos_theta_i), cos_theta_t_abs = safe_sqrt(cos_theta_t_sqr);
    const bool index_matched = detach(eta) == 1.f, special = index_matched || detach(cos_theta_i_abs) == 0.f;
    const R a_s = fma_(-eta_it, cos_theta_t_abs, cos_theta_i_abs) / fma_(eta_it, cos_theta_t_abs, cos_theta_i_abs);
    const R a_p = fma_(-eta_it, cos_theta_i_abs, cos_theta_t_abs) / fma_(eta_it, cos_theta_i_abs, cos_theta_t_abs);
    F = R(.5f) * (sqr(a_s) + sqr(a_p));
    if (special) F = index_matched ? R(0.f) : R(1.f);
    cos_theta_t = signbit_(detach(cos_theta_i)) ? cos_theta_t_abs : -cos_theta_t_abs;      // mulsign_neg
}

// eta_p = intIOR / extIOR, inv_eta_p = extIOR / intIOR (RoughDielectric::m_eta, m_inv_eta)
template <typename R>
PSDR_DEV Vec3<R> dielectric_eval(const R &au, const R &av, const R &eta_p, const R &inv_eta_p, bool two_sided, Vec3<R> wi, Vec3<R> wo, bool active) {
    using V = Vec3<R>;
    if (two_sided) { wo.z = mulsign(wo.z, detach(wi.z)); wi.z = abs_(wi.z); }
    const R cos_theta_i = wi.z, cos_theta_o = wo.z;
    if (!(active && detach(cos_theta_i) != 0.f)) return V(R(0.f));
    const bool reflect = detach(cos_theta_i) * detach(cos_theta_o) > 0.f;
    const bool front = detach(cos_theta_i) > 0.f;
    const R eta = front ? eta_p : inv_eta_p, inv_eta = front ? inv_eta_p : eta_p;
    V m = normalize(wi + wo * (reflect ? R(1.f) : eta));
    if (signbit_(detach(m.z))) m = -m;
    GGX<R> distr{au, av};
    const R D = distr.eval(m);
    R F, ct, e_it, e_ti;
    fresnel_dielectric<R>(eta_p, dot(wi, m), F, ct, e_it, e_ti);
    const R G = distr.smith_g1(wi, m) * distr.smith_g1(wo, m);
    if (reflect) return V(F * D * G / (R(4.f) * abs_(cos_theta_i)));
    const R scale = sqr(inv_eta);
    const R value = abs_((scale * (R(1.f) - F) * D * G * eta * eta * dot(wi, m) * dot(wo, m)) / (cos_theta_i * sqr(dot(wi, m) + eta * dot(wo, m))));
    return V(value);
}
PSDR_DEV float dielectric_pdf(float au, float av, float eta_p, float inv_eta_p, bool two_sided, Vec3f wi, Vec3f wo, bool active) {
    if (two_sided) { wo.z = mulsign(wo.z, wi.z); wi.z = fabsf(wi.z); }
    const float cos_theta_i = wi.z, cos_theta_o = wo.z;
    active = active && cos_theta_i != 0.f;
    const bool reflect = cos_theta_i * cos_theta_o > 0.f;
    const float eta = cos_theta_i > 0.f ? eta_p : inv_eta_p;
    Vec3f m = normalize(wi + wo * (reflect ? 1.f : eta));
    if (signbit_(m.z)) m = -m;
    active = active && dot(wi, m) * wi.z > 0.f && dot(wo, m) * wo.z > 0.f;
    if (!active) return 0.f;
    const float dwh_dwo = reflect ? 1.f / (4.f * dot(wo, m)) : (eta * eta * dot(wo, m)) / sqr(dot(wi, m) + eta * dot(wo, m));
    GGX<float> distr{au, av};
    const Vec3f pwi = signbit_(wi.z) ? -wi : wi;
    float prob = distr.eval(m) * distr.smith_g1(pwi, m) / pwi.z;
    float F, ct, e_it, e_ti;
    fresnel_dielectric<float>(eta_p, dot(wi, m), F, ct, e_it, e_ti);
    prob *= reflect ? F : 1.f - F;
    return prob * fabsf(dwh_dwo);
}
PSDR_DEV void dielectric_sample(float au, float av, float eta_p, bool two_sided, Vec3f wi, float s0, float s1, float s2, bool active,
                                Vec3f &wo, float &pdf_out, bool &valid) {
    if (two_sided) wi.z = fabsf(wi.z);
    wo = Vec3f(0.f, 0.f, 0.f);
    const float cos_theta_i = wi.z;
    active = active && cos_theta_i != 0.f;
    const Vec3f pwi = signbit_(cos_theta_i) ? -wi : wi;
    float m_pdf;
    const Vec3f m = ggx_sample(au, av, pwi, s0, s1, m_pdf);
    active = active && m_pdf != 0.f;
    float F, cos_theta_t, eta_it, eta_ti;
    fresnel_dielectric<float>(eta_p, dot(wi, m), F, cos_theta_t, eta_it, eta_ti);
    const bool sel_r = (s2 <= F) && active, sel_t = !sel_r && active;
    float pdf = m_pdf * (sel_r ? F : 1.f - F);
    const float bs_eta = sel_r ? 1.f : eta_it;
    float dwh_dwo = 0.f;
    if (sel_r) {
        const float k = 2.f * dot(wi, m);
        wo = Vec3f(fma_(m.x, k, -wi.x), fma_(m.y, k, -wi.y), fma_(m.z, k, -wi.z));
        dwh_dwo = 1.f / (4.f * dot(wo, m));
    }
    if (sel_t) {
        const float k = fma_(dot(wi, m), eta_ti, cos_theta_t);
        wo = Vec3f(fma_(m.x, k, -(wi.x * eta_ti)), fma_(m.y, k, -(wi.y * eta_ti)), fma_(m.z, k, -(wi.z * eta_ti)));
        dwh_dwo = (sqr(bs_eta) * dot(wo, m)) / sqr(dot(wi, m) + bs_eta * dot(wo, m));
    }
    GGX<float> distr{au, av};
    pdf *= fabsf(dwh_dwo) * distr.smith_g1(wo, m);
    pdf_out = pdf;
    valid = active && (sel_t || sel_r);
}

} // namespace psdr
