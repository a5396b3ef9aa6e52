// ORACLE — TEST INFRASTRUCTURE ONLY (see num.h header).
//
// microfacet.h — Microfacet BSDF (reference src/bsdf/microfacet.cpp:18-134: Lambertian diffuse + GGX specular with the
// Schlick-style 2^(...) Fresnel) and its GGXDistribution (src/bsdf/ggx.cpp:8-107, visible-normal sampling).
// drjit::pow(2, x) is not in /root/reference; it is restated as the Cephes exp2f polynomial with explicit fma
// (the HIP path states the same); drjit::rsqrt is taken as 1/sqrt.
#pragma once
#include <cmath>
#include "scene.h"

namespace orc {

void sincos_cephes(float xx, float &s_out, float &c_out);     // integrator.cpp

inline float exp2_cephes(float x) {
    if (x > 127.f) return INFINITY;
    if (x < -127.f) return 0.f;
    float px = std::floor(x);
    int i0 = (int) px;
    x = x - px;
    if (x > 0.5f) { i0 += 1; x = x - 1.f; }
    px = fma_(fma_(fma_(fma_(fma_(1.535336188319500e-4f, x, 1.339887440266574e-3f), x, 9.618437357674640e-3f), x, 5.550332471162809e-2f), x,
                   2.402264791363012e-1f), x, 6.931472028550421e-1f);
    px = fma_(px, x, 1.0f);
    return std::ldexp(px, i0);
}
inline float exp2_(float x) { return exp2_cephes(x); }
inline Dual exp2_(const Dual &x) { const float v = exp2_cephes(x.v); return Dual(v, v * 0.6931471805599453f * x.d); }

struct MicrofacetParams { V3d specular, diffuse; Dual roughness; bool two_sided; };

template <typename R> struct GGX {
    R au, av;                                                      // alpha_u, alpha_v (Microfacet: both = roughness^2)
    // ggx.cpp:13-33
    R eval(const V3<R> &m) const {
        const R alpha_uv = au * av;
        const R cos_theta = m.z;
        const R r = rcp(R(Pi) * alpha_uv * sqr(sqr(m.x / au) + sqr(m.y / av) + sqr(m.z)));
        return (detach(r) * detach(cos_theta) > 1e-20f) ? r : R(0.f);
    }
    // ggx.cpp:84-97
    R smith_g1(const V3<R> &v, const V3<R> &m) const {
        const R xy_alpha_2 = sqr(au * v.x) + sqr(av * v.y);
        const R tan_theta_alpha_2 = xy_alpha_2 / sqr(v.z);
        R result = R(2.f) / (R(1.f) + sqrt_(R(1.f) + tan_theta_alpha_2));
        if (detach(xy_alpha_2) == 0.f) result = R(1.f);
        if (detach(dot(v, m)) * detach(v.z) <= 0.f) result = R(0.f);
        return result;
    }
};

// warp.h:16-52
inline V2f square_to_uniform_disk_concentric(float sx, float sy) {
    float x = fma_(2.f, sx, -1.f), y = fma_(2.f, sy, -1.f);
    bool is_zero = (x == 0.f) && (y == 0.f), q13 = std::fabs(x) < std::fabs(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = .25f * Pi * rp / r;
    if (q13) phi = .5f * Pi - phi;
    if (is_zero) phi = 0.f;
    float s, c;
    sincos_cephes(phi, s, c);
    return V2f(r * c, r * s);
}
// ggx.cpp:99-107
inline V2f ggx_sample_visible_11(float cos_theta_i, float sx, float sy) {
    V2f p = square_to_uniform_disk_concentric(sx, sy);
    const float s = .5f * (1.f + cos_theta_i);
    const float a0 = safe_sqrt(1.f - sqr(p.x));
    p.y = fma_(p.y, s, fma_(-a0, s, a0));                          // drjit::lerp(a, b, t) = fmadd(b, t, fnmadd(a, t, a))
    const float x = p.x, y = p.y, z = safe_sqrt(1.f - fma_(p.y, p.y, p.x * p.x));
    const float sin_theta_i = safe_sqrt(1.f - sqr(cos_theta_i));
    const float norm = 1.f / fma_(sin_theta_i, y, cos_theta_i * z);
    return V2f(fma_(cos_theta_i, y, -(sin_theta_i * z)) * norm, x * norm);
}
// ggx.cpp:35-82 (everything detached)
inline V3f ggx_sample(float au, float av, const V3f &wi, float sx, float sy, float &pdf) {
    const V3f wi_p = normalize(V3f(au * wi.x, av * wi.y, wi.z));
    const float sin_theta_2 = fma_(wi_p.x, wi_p.x, sqr(wi_p.y));                          // frame.h:81
    const float inv_sin_theta = 1.f / std::sqrt(sin_theta_2);
    const bool deg = std::fabs(sin_theta_2) <= 4.f * Epsilon;
    const float sin_phi = deg ? 0.f : std::min(std::max(wi_p.y * inv_sin_theta, -1.f), 1.f);
    const float cos_phi = deg ? 1.f : std::min(std::max(wi_p.x * inv_sin_theta, -1.f), 1.f);
    V2f slope = ggx_sample_visible_11(wi_p.z, sx, sy);
    slope = V2f(fma_(cos_phi, slope.x, -(sin_phi * slope.y)) * au, fma_(sin_phi, slope.x, cos_phi * slope.y) * av);
    const V3f m = normalize(V3f(-slope.x, -slope.y, 1.f));
    GGX<float> g{au, av};
    pdf = g.smith_g1(wi, m) * std::fabs(dot(wi, m)) * g.eval(m) / std::fabs(wi.z);
    return m;
}

// microfacet.cpp:22-62
template <bool ad> V3<Real<ad>> microfacet_eval(const MicrofacetParams &P, V3<Real<ad>> wi, V3<Real<ad>> wo, bool active) {
    using R = Real<ad>; using V = V3<R>;
    if (P.two_sided) { wo.z = mulsign(wo.z, detach(wi.z)); wi.z = abs_(wi.z); }
    const R cos_theta_nv = wi.z, cos_theta_nl = wo.z;
    active = active && (detach(cos_theta_nv) > 0.f && detach(cos_theta_nl) > 0.f);
    if (!active) return V(R(0.f));
    const V diffuse = pick<ad>(P.diffuse) * R(InvPi);
    const V H = normalize(wi + wo);
    const R cos_theta_vh = dot(H, wi);
    const V F0 = pick<ad>(P.specular);
    R roughness;
    if constexpr (ad) roughness = P.roughness; else roughness = P.roughness.v;
    GGX<R> distr{sqr(roughness), sqr(roughness)};
    const R ggx = distr.eval(H);
    const R coeff = cos_theta_vh * (R(-5.55473f) * cos_theta_vh - R(6.8316f));
    const V fresnel = F0 + (V(R(1.f)) - F0) * exp2_(coeff);
    const R smithG = distr.smith_g1(wi, H) * distr.smith_g1(wo, H);
    const V numerator = fresnel * (ggx * smithG);
    const R denominator = R(4.f) * cos_theta_nl * cos_theta_nv;
    const V specular = numerator / (denominator + R(1e-6f));
    return (diffuse + specular) * cos_theta_nl;
}
// microfacet.cpp:108-131 (detached)
inline float microfacet_pdf(const MicrofacetParams &P, V3f wi, V3f wo, bool active) {
    if (P.two_sided) { wo.z = mulsign(wo.z, wi.z); wi.z = std::fabs(wi.z); }
    const float cos_theta_i = wi.z, cos_theta_o = wo.z;
    const V3f m = normalize(wo + wi);
    active = active && cos_theta_i > 0.f && cos_theta_o > 0.f && dot(wi, m) > 0.f && dot(wo, m) > 0.f;
    if (!active) return 0.f;
    GGX<float> distr{sqr(P.roughness.v), sqr(P.roughness.v)};
    return distr.eval(m) * distr.smith_g1(wi, m) / (4.f * cos_theta_i);
}
// microfacet.cpp:75-98: uses sample.x, sample.y (not the tail); the sampled direction stays in the upper hemisphere
struct MicrofacetSample { V3f wo; float pdf; bool valid; };
inline MicrofacetSample microfacet_sample(const MicrofacetParams &P, V3f wi, const float s3[3], bool active) {
    if (P.two_sided) wi.z = std::fabs(wi.z);
    MicrofacetSample bs;
    const float cos_theta_i = wi.z;
    float m_pdf;
    const V3f m = ggx_sample(sqr(P.roughness.v), sqr(P.roughness.v), wi, s3[0], s3[1], m_pdf);
    const float k = 2.f * dot(wi, m);
    bs.wo = V3f(fma_(m.x, k, -wi.x), fma_(m.y, k, -wi.y), fma_(m.z, k, -wi.z));
    bs.pdf = m_pdf / (4.f * dot(bs.wo, m));
    bs.valid = active && (cos_theta_i > 0.f) && (bs.pdf != 0.f) && (bs.wo.z > 0.f);
    return bs;
}

// ---------------------------------------------------------------- RoughConductor (reference src/bsdf/roughconductor.cpp:30-118)
struct ConductorParams { Dual alpha_u, alpha_v; V3d eta, k, specular; bool two_sided; };

// conductor Fresnel, reference include/psdr/utils.h:166-182 (per colour channel)
template <typename R> R fresnel_conductor(const R &eta_r, const R &eta_i, const R &cos_theta_i) {
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

template <bool ad> V3<Real<ad>> conductor_eval(const ConductorParams &P, V3<Real<ad>> wi, V3<Real<ad>> wo, bool active) {
    using R = Real<ad>; using V = V3<R>;
    if (P.two_sided) { wo.z = mulsign(wo.z, detach(wi.z)); wi.z = abs_(wi.z); }
    active = active && (detach(wi.z) > 0.f && detach(wo.z) > 0.f);
    if (!active) return V(R(0.f));
    GGX<R> distr{pick<ad>(P.alpha_u), pick<ad>(P.alpha_v)};
    const V H = normalize(wo + wi);
    const R D = distr.eval(H);
    if (detach(D) == 0.f) return V(R(0.f));
    const R G = distr.smith_g1(wi, H) * distr.smith_g1(wo, H);
    const R result = D * G / (R(4.f) * wi.z);
    const V eta = pick<ad>(P.eta), k = pick<ad>(P.k);
    const R c = dot(wi, H);
    const V F(fresnel_conductor<R>(eta.x, k.x, c), fresnel_conductor<R>(eta.y, k.y, c), fresnel_conductor<R>(eta.z, k.z, c));
    return F * result * pick<ad>(P.specular);
}
inline float conductor_pdf(const ConductorParams &P, V3f wi, V3f wo, bool active) {
    if (P.two_sided) { wo.z = mulsign(wo.z, wi.z); wi.z = std::fabs(wi.z); }
    const V3f m = normalize(wo + wi);
    active = active && wi.z > 0.f && wo.z > 0.f && dot(wi, m) > 0.f && dot(wo, m) > 0.f;
    if (!active) return 0.f;
    GGX<float> distr{P.alpha_u.v, P.alpha_v.v};
    return distr.eval(m) * distr.smith_g1(wi, m) / (4.f * wi.z);
}
inline MicrofacetSample conductor_sample(const ConductorParams &P, V3f wi, const float s3[3], bool active) {
    if (P.two_sided) wi.z = std::fabs(wi.z);
    MicrofacetSample bs;
    float m_pdf;
    const V3f m = ggx_sample(P.alpha_u.v, P.alpha_v.v, wi, s3[0], s3[1], m_pdf);
    const float kk = 2.f * dot(wi, m);
    bs.wo = V3f(fma_(m.x, kk, -wi.x), fma_(m.y, kk, -wi.y), fma_(m.z, kk, -wi.z));
    bs.pdf = m_pdf / (4.f * dot(bs.wo, m));
    bs.valid = active && (wi.z > 0.f) && (bs.pdf != 0.f) && (bs.wo.z > 0.f);
    return bs;
}

// MicrofacetPerVertex::__eval (microfacet_pv.cpp:21-66): its own GGX / Smith-Schlick terms, not GGXDistribution's
template <bool ad> V3<Real<ad>> microfacet_pv_eval(const MicrofacetParams &P, V3<Real<ad>> wi, V3<Real<ad>> wo, bool active) {
    using R = Real<ad>; using V = V3<R>;
    if (P.two_sided) { wo.z = mulsign(wo.z, detach(wi.z)); wi.z = abs_(wi.z); }
    const R cos_theta_nv = wi.z, cos_theta_nl = wo.z;
    active = active && (detach(cos_theta_nv) > 0.f && detach(cos_theta_nl) > 0.f);
    if (!active) return V(R(0.f));
    const V diffuse = pick<ad>(P.diffuse) * R(InvPi);
    const V H = normalize(wi + wo);
    const R cos_theta_nh = H.z, cos_theta_vh = dot(H, wi);
    const V F0 = pick<ad>(P.specular);
    const R roughness = pick<ad>(P.roughness);
    const R alpha = sqr(roughness);
    const R k = sqr(roughness + R(1.f)) / R(8.f);
    const R tmp = alpha / (cos_theta_nh * cos_theta_nh * (sqr(alpha) - R(1.f)) + R(1.f));
    const R ggx = tmp * tmp * R(InvPi);
    const R coeff = cos_theta_vh * (R(-5.55473f) * cos_theta_vh - R(6.8316f));
    const V fresnel = F0 + (V(R(1.f)) - F0) * exp2_(coeff);
    const R smithG1 = cos_theta_nv / (cos_theta_nv * (R(1.f) - k) + k);
    const R smithG2 = cos_theta_nl / (cos_theta_nl * (R(1.f) - k) + k);
    const R smithG = smithG1 * smithG2;
    const V numerator = fresnel * (ggx * smithG);
    const R denominator = R(4.f) * cos_theta_nl * cos_theta_nv;
    const V specular = numerator / (denominator + R(1e-6f));
    return (diffuse + specular) * cos_theta_nl;
}

// ---------------------------------------------------------------- RoughDielectric (reference src/bsdf/roughdielectric.cpp:35-237)
struct DielectricParams { Dual alpha_u, alpha_v, eta, inv_eta; bool two_sided; };

// fresnel_dielectric, reference include/psdr/utils.h:184-215: returns F, cos_theta_t, eta_it, eta_ti
template <typename R> void fresnel_dielectric(const R &eta, const R &cos_theta_i, R &F, R &cos_theta_t, R &eta_it, R &eta_ti) {
    const bool outside = detach(cos_theta_i) >= 0.f;
    const R rcp_eta = rcp(eta);
    eta_it = outside ? eta : rcp_eta;
    eta_ti = outside ? rcp_eta : eta;
    const R cos_theta_t_sqr = fma_(-fma_(-cos_theta_i, cos_theta_i, R(1.f)), eta_ti * eta_ti, R(1.f));
    const R cos_theta_i_abs = abs_(cos_theta_i), cos_theta_t_abs = safe_sqrt(cos_theta_t_sqr);
    const bool index_matched = detach(eta) == 1.f, special = index_matched || detach(cos_theta_i_abs) == 0.f;
    const R a_s = fma_(-eta_it, cos_theta_t_abs, cos_theta_i_abs) / fma_(eta_it, cos_theta_t_abs, cos_theta_i_abs);
    const R a_p = fma_(-eta_it, cos_theta_i_abs, cos_theta_t_abs) / fma_(eta_it, cos_theta_i_abs, cos_theta_t_abs);
    F = R(.5f) * (sqr(a_s) + sqr(a_p));
    if (special) F = index_matched ? R(0.f) : R(1.f);
    cos_theta_t = std::signbit(detach(cos_theta_i)) ? cos_theta_t_abs : -cos_theta_t_abs;       // mulsign_neg
}

template <bool ad> V3<Real<ad>> dielectric_eval(const DielectricParams &P, V3<Real<ad>> wi, V3<Real<ad>> wo, bool active) {
    using R = Real<ad>; using V = V3<R>;
    if (P.two_sided) { wo.z = mulsign(wo.z, detach(wi.z)); wi.z = abs_(wi.z); }
    const R cos_theta_i = wi.z, cos_theta_o = wo.z;
    active = active && detach(cos_theta_i) != 0.f;
    if (!active) return V(R(0.f));
    const bool reflect = detach(cos_theta_i) * detach(cos_theta_o) > 0.f;
    const bool front = detach(cos_theta_i) > 0.f;
    const R eta = front ? pick<ad>(P.eta) : pick<ad>(P.inv_eta), inv_eta = front ? pick<ad>(P.inv_eta) : pick<ad>(P.eta);
    V m = normalize(wi + wo * (reflect ? R(1.f) : eta));
    if (std::signbit(detach(m.z))) m = -m;                                          // mulsign(m, cos_theta(m))
    GGX<R> distr{pick<ad>(P.alpha_u), pick<ad>(P.alpha_v)};
    const R D = distr.eval(m);
    R F, ct, e_it, e_ti;
    fresnel_dielectric<R>(pick<ad>(P.eta), dot(wi, m), F, ct, e_it, e_ti);
    const R G = distr.smith_g1(wi, m) * distr.smith_g1(wo, m);
    if (reflect) return V(F * D * G / (R(4.f) * abs_(cos_theta_i)));
    const R scale = sqr(inv_eta);
    const R value = abs_((scale * (R(1.f) - F) * D * G * eta * eta * dot(wi, m) * dot(wo, m)) / (cos_theta_i * sqr(dot(wi, m) + eta * dot(wo, m))));
    return V(value);
}
inline float dielectric_pdf(const DielectricParams &P, V3f wi, V3f wo, bool active) {
    if (P.two_sided) { wo.z = mulsign(wo.z, wi.z); wi.z = std::fabs(wi.z); }
    const float cos_theta_i = wi.z, cos_theta_o = wo.z;
    active = active && cos_theta_i != 0.f;
    const bool reflect = cos_theta_i * cos_theta_o > 0.f;
    const float eta = cos_theta_i > 0.f ? P.eta.v : P.inv_eta.v;
    V3f m = normalize(wi + wo * (reflect ? 1.f : eta));
    if (std::signbit(m.z)) m = -m;
    active = active && dot(wi, m) * wi.z > 0.f && dot(wo, m) * wo.z > 0.f;
    if (!active) return 0.f;
    const float dwh_dwo = reflect ? 1.f / (4.f * dot(wo, m)) : (eta * eta * dot(wo, m)) / sqr(dot(wi, m) + eta * dot(wo, m));
    GGX<float> distr{P.alpha_u.v, P.alpha_v.v};
    const V3f pwi = std::signbit(wi.z) ? -wi : wi;
    float prob = distr.eval(m) * distr.smith_g1(pwi, m) / pwi.z;
    float F, ct, e_it, e_ti;
    fresnel_dielectric<float>(P.eta.v, dot(wi, m), F, ct, e_it, e_ti);
    prob *= reflect ? F : 1.f - F;
    return prob * std::fabs(dwh_dwo);
}
inline MicrofacetSample dielectric_sample(const DielectricParams &P, V3f wi, const float s3[3], bool active) {
    if (P.two_sided) wi.z = std::fabs(wi.z);
    MicrofacetSample bs;
    bs.wo = V3f(0.f, 0.f, 0.f); bs.pdf = 0.f; bs.valid = false;
    const float cos_theta_i = wi.z;
    active = active && cos_theta_i != 0.f;
    const V3f pwi = std::signbit(cos_theta_i) ? -wi : wi;
    float m_pdf;
    const V3f m = ggx_sample(P.alpha_u.v, P.alpha_v.v, pwi, s3[0], s3[1], m_pdf);
    active = active && m_pdf != 0.f;
    float F, cos_theta_t, eta_it, eta_ti;
    fresnel_dielectric<float>(P.eta.v, dot(wi, m), F, cos_theta_t, eta_it, eta_ti);
    const bool sel_r = (s3[2] <= F) && active, sel_t = !sel_r && active;
    float pdf = m_pdf * (sel_r ? F : 1.f - F);
    const float bs_eta = sel_r ? 1.f : eta_it;
    float dwh_dwo = 0.f;
    if (sel_r) {
        const float k = 2.f * dot(wi, m);
        bs.wo = V3f(fma_(m.x, k, -wi.x), fma_(m.y, k, -wi.y), fma_(m.z, k, -wi.z));
        dwh_dwo = 1.f / (4.f * dot(bs.wo, m));
    }
    if (sel_t) {
        const float k = fma_(dot(wi, m), eta_ti, cos_theta_t);
        bs.wo = V3f(fma_(m.x, k, -(wi.x * eta_ti)), fma_(m.y, k, -(wi.y * eta_ti)), fma_(m.z, k, -(wi.z * eta_ti)));
        dwh_dwo = (sqr(bs_eta) * dot(bs.wo, m)) / sqr(dot(wi, m) + bs_eta * dot(bs.wo, m));
    }
    GGX<float> distr{P.alpha_u.v, P.alpha_v.v};
    pdf *= std::fabs(dwh_dwo) * distr.smith_g1(bs.wo, m);
    bs.pdf = pdf;
    bs.valid = active && (sel_t || sel_r);
    return bs;
}

} // namespace orc
