// ORACLE — TEST INFRASTRUCTURE ONLY (see num.h header).
//
// integrator.cpp — one-lane restatement of the reference hot path.  Every function names the
// reference file:line it follows.  `ad=false` is the reference's C instantiation, `ad=true` the
// D instantiation with drjit's tape replaced by one forward tangent (num.h).
#include "integrator.h"
#include "envmap.h"
#include <cmath>

namespace orc {

using std::fabs;

// ---------------------------------------------------------------- small helpers
static inline float mulsign(float a, float b) { return std::signbit(b) ? -a : a; }
static inline Dual mulsign(const Dual &a, float b) { return std::signbit(b) ? -a : a; }
static inline float sign1(float x) { return std::signbit(x) ? -1.f : 1.f; }         // drjit::sign
static inline int sign_eps(float x, float eps) { return x > eps ? 1 : (x < -eps ? -1 : 0); }   // utils.h:47-53
static inline float mis_weight(float p1, float p2) { float w1 = p1 * p1, w2 = p2 * p2; return w1 / (w1 + w2); }   // utils.h:277-281

template <typename T> static V3<T> bilinear(const V3<T> &p0, const V3<T> &e1, const V3<T> &e2, const T &s, const T &t) {   // utils.h:64-67
    return V3<T>(fma_(e1.x, s, fma_(e2.x, t, p0.x)), fma_(e1.y, s, fma_(e2.y, t, p0.y)), fma_(e1.z, s, fma_(e2.z, t, p0.z)));
}

// frame.h:9-28 coordinate_system (Duff et al.)
template <typename T> static void coordinate_system(const V3<T> &n, V3<T> &s, V3<T> &t) {
    float nz = detach(n.z);
    T sign = T(sign1(nz));
    T a = -rcp(sign + n.z);
    T b = n.x * n.y * a;
    s = V3<T>(mulsign(sqr(n.x) * a, nz) + T(1.f), mulsign(b, nz), -mulsign(n.x, nz));
    t = V3<T>(b, sign + sqr(n.y) * a, -n.y);
}
template <bool ad> static V3<Real<ad>> to_local(const Frame<ad> &f, const V3<Real<ad>> &v) { return {dot(v, f.s), dot(v, f.t), dot(v, f.n)}; }
template <bool ad> static V3<Real<ad>> to_world(const Frame<ad> &f, const V3<Real<ad>> &v) { return f.s * v.x + f.t * v.y + f.n * v.z; }

// drjit::sincos is not libm: drjit evaluates the Cephes single-precision kernels (S. Moshier, sinf.c /
// cosf.c: octant reduction with the 3-term extended-precision pi/4 split, degree-3 polynomials in z=x^2).
// drjit is absent from /root/reference, so the published Cephes algorithm is restated here with every
// multiply-add written as an explicit fma; the HIP path states the same algorithm, which makes the two
// agree bit-for-bit on this step.
void sincos_cephes(float xx, float &s_out, float &c_out) {
    const float FOPI = 1.27323954473516f, DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
    float x = fabs(xx);
    int j = (int) (FOPI * x);
    float y = (float) j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    float sign_s = xx < 0.f ? -1.f : 1.f, sign_c = 1.f;
    if (j > 3) { sign_s = -sign_s; sign_c = -sign_c; j -= 4; }
    if (j > 1) sign_c = -sign_c;
    x = fma_(-y, DP1, x); x = fma_(-y, DP2, x); x = fma_(-y, DP3, x);
    const float z = x * x;
    const float ps = fma_(fma_(fma_(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, x, x);
    const float pc = fma_(fma_(fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fma_(-0.5f, z, 1.0f));
    const bool swap = (j == 1) || (j == 2);
    s_out = sign_s * (swap ? pc : ps);
    c_out = sign_c * (swap ? ps : pc);
}

// warp.h:16-63
static V3f square_to_cosine_hemisphere(float sx, float sy) {
    float x = fma_(2.f, sx, -1.f), y = fma_(2.f, sy, -1.f);
    bool is_zero = (x == 0.f) && (y == 0.f), q13 = fabs(x) < fabs(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = .25f * Pi * rp / r;
    if (q13) phi = .5f * Pi - phi;
    if (is_zero) phi = 0.f;
    float s, c;
    sincos_cephes(phi, s, c);
    float px = r * c, py = r * s;
    float z = safe_sqrt(1.f - fma_(py, py, px * px));
    return {px, py, z};
}
// warp.h:79-82
static void square_to_uniform_triangle(float sx, float sy, float &a, float &b) {
    float t = safe_sqrt(1.f - sx);
    a = 1.f - t; b = t * sy;
}

// ---------------------------------------------------------------- camera
// perspective.cpp:160-178
template <bool ad> Ray<ad> sample_primary_ray(const CameraC &cam, const V2<Real<ad>> &s) {
    if (cam.orthographic) {          // orthographic.cpp:161-181: origin on the near plane, direction = the camera axis
        Ray<ad> r;
        const V3f near_p = transform_pos(cam.sample_to_camera, V3f(detach(s.x), detach(s.y), 0.f));
        if constexpr (ad) {
            r.o = transform_pos(cam.to_world, V3d(near_p));
            r.d = transform_dir(cam.to_world, V3d(Dual(0.f), Dual(0.f), Dual(1.f)));
        } else {
            M4f tw = detach(cam.to_world);
            r.o = transform_pos(tw, near_p);
            r.d = transform_dir(tw, V3f(0.f, 0.f, 1.f));
        }
        return r;
    }
    V3f d = normalize(transform_pos(cam.sample_to_camera, V3f(detach(s.x), detach(s.y), 0.f)));   // detached in D mode (:172)
    Ray<ad> r;
    if constexpr (ad) {
        r.o = transform_pos(cam.to_world, V3d(Dual(0.f)));
        r.d = transform_dir(cam.to_world, V3d(d));
    } else {
        M4f tw = detach(cam.to_world);
        r.o = transform_pos(tw, V3f(0.f));
        r.d = transform_dir(tw, d);
    }
    return r;
}
template RayC sample_primary_ray<false>(const CameraC &, const V2f &);
template RayD sample_primary_ray<true>(const CameraC &, const V2d &);

static inline int floor2int(float x) { return (int) std::floor(x); }

// perspective.cpp:181-197
SensorDirectSample sample_direct(const Scene &sc, const CameraC &cam, const V3f &p) {
    SensorDirectSample r;
    V3f q3 = transform_pos(detach(cam.world_to_sample), p);
    r.q = V2f(q3.x, q3.y);
    int ix = floor2int(r.q.x * (float) sc.width), iy = floor2int(r.q.y * (float) sc.height);
    r.valid = ix >= 0 && ix < sc.width && iy >= 0 && iy < sc.height;
    r.pixel_idx = r.valid ? iy * sc.width + ix : -1;
    V3f dir = p - detach(cam.pos);
    float dist2 = squared_norm(dir);
    dir = dir / safe_sqrt(dist2);
    float cosTheta = dot(detach(cam.dir), dir);
    float rc = rcp(cosTheta);
    r.sensor_val = rcp(dist2) * (rc * rc * rc) * cam.inv_area;     // pow(rcp(cos), 3)
    return r;
}

// perspective.cpp:200-226
PrimaryEdgeSample sample_primary_edge(const Scene &sc, const CameraC &cam, float sample1) {
    PrimaryEdgeSample r;
    float pdf;
    int ei = cam.edge_distrb.sample_reuse(sample1, pdf);
    const PrimEdge &e = cam.edges[ei];
    r.pdf = pdf / e.length;
    Dual s(sample1), oms(1.0f - sample1);
    V2d p_(fma_(e.p0.x, oms, e.p1.x * s), fma_(e.p0.y, oms, e.p1.y * s));
    V2f p = detach(p_);
    r.x_dot_n = dot(p_, e.normal);
    int ix = floor2int(p.x * (float) sc.width), iy = floor2int(p.y * (float) sc.height);
    bool valid = ix >= 0 && ix < sc.width && iy >= 0 && iy < sc.height;
    r.idx = valid ? iy * sc.width + ix : -1;
    r.ray_p = sample_primary_ray<false>(cam, V2f(p.x + EdgeEpsilon * e.normal.x, p.y + EdgeEpsilon * e.normal.y));
    r.ray_n = sample_primary_ray<false>(cam, V2f(p.x - EdgeEpsilon * e.normal.x, p.y - EdgeEpsilon * e.normal.y));
    return r;
}

// ---------------------------------------------------------------- Scene::ray_intersect
// utils.h:82-93 with differentiable triangle and ray
template <typename T> static void ray_intersect_triangle(const V3<T> &p0, const V3<T> &e1, const V3<T> &e2,
                                                         const V3<T> &o, const V3<T> &d, T &u, T &v, T &t) {
    V3<T> h = cross(d, e2);
    T a = dot(e1, h);
    T f = rcp(a);
    V3<T> s = o - p0;
    u = f * dot(s, h);
    V3<T> q = cross(s, e1);
    v = f * dot(d, q);
    t = f * dot(e2, q);
}

template <bool ad> static V3<Real<ad>> pick(const V3d &a) { if constexpr (ad) return a; else return detach(a); }
template <bool ad> static Real<ad> pick(const Dual &a) { if constexpr (ad) return a; else return a.v; }
} // namespace orc
#include "microfacet.h"
namespace orc {

// scene.cpp:612-806
template <bool ad, bool path_space>
Its<ad> ray_intersect(const Scene &sc, const Ray<ad> &ray, bool active, int *out_tri) {
    static_assert(ad || !path_space);
    using R = Real<ad>; using V = V3<R>;
    Its<ad> its;
    if (out_tri) *out_tri = -1;
    if (!active) return its;
    Hit h = trace_closest(sc, detach(ray.o), detach(ray.d), sc.use_bvh);
    if (h.tri < 0) return its;
    if (out_tri) *out_tri = h.tri;
    const Tri &T = sc.tris[h.tri];
    its.valid = true; its.tri = h.tri; its.mesh = T.mesh;
    V p0 = pick<ad>(T.p0), e1 = pick<ad>(T.e1), e2 = pick<ad>(T.e2);
    V n0 = pick<ad>(T.n0), n1 = pick<ad>(T.n1), n2 = pick<ad>(T.n2);
    its.n = pick<ad>(T.fn);
    if constexpr (ad && path_space) its.J = T.area / Dual(T.area.v); else its.J = R(1.f);

    // dp/du from the uv parameterisation (scene.cpp:724-734)
    V2f duv0 = T.uv[1] - T.uv[0], duv1 = T.uv[2] - T.uv[0];
    float det = fma_(duv0.x, duv1.y, -(duv0.y * duv1.x));
    float inv_det = rcp(det);
    bool valid_dp = det != 0.f;

    R u, v;
    V sh_n, dir_in;
    if constexpr (!ad || path_space) {
        // path-space formulation: barycentrics are the (detached) ones the tracer returned
        u = R(h.u); v = R(h.v);
        its.p = bilinear(p0, e1, e2, u, v);
        V dir = its.p - ray.o;
        its.t = norm(dir);
        dir /= its.t;
        dir_in = dir;
    } else {
        // solid-angle formulation: re-intersect differentiably (scene.cpp:774-784)
        R t;
        ray_intersect_triangle<R>(p0, e1, e2, ray.o, ray.d, u, v, t);
        its.p = V(fma_(ray.d.x, t, ray.o.x), fma_(ray.d.y, t, ray.o.y), fma_(ray.d.z, t, ray.o.z));
        its.t = t;
        dir_in = ray.d;
    }
    sh_n = normalize(bilinear(n0, n1 - n0, n2 - n0, u, v));
    if (T.flat) sh_n = its.n;
    its.uv = V2<R>(fma_(R(duv0.x), u, fma_(R(duv1.x), v, R(T.uv[0].x))), fma_(R(duv0.y), u, fma_(R(duv1.y), v, R(T.uv[0].y))));
    its.sh.n = sh_n;
    coordinate_system(sh_n, its.sh.s, its.sh.t);
    if (valid_dp) {
        // dp_du = fmsub(duv1.y, dp0, duv0.y*dp1) * inv_det   (scene.cpp:762)
        V dp_du = (e1 * R(duv1.y) - e2 * R(duv0.y)) * R(inv_det);
        its.dp_du = dp_du;
        its.sh.s = normalize(dp_du - sh_n * dot(sh_n, dp_du));
        its.sh.t = cross(sh_n, its.sh.s);
    }
    its.wi = to_local<ad>(its.sh, -dir_in);
    its.bc = V2<R>(u, v);
    return its;
}
template Its<false> ray_intersect<false, false>(const Scene &, const RayC &, bool, int *);
template Its<true> ray_intersect<true, false>(const Scene &, const RayD &, bool, int *);
template Its<true> ray_intersect<true, true>(const Scene &, const RayD &, bool, int *);

// ---------------------------------------------------------------- emitters
template <bool ad> static bool is_emitter(const Scene &sc, const Its<ad> &its) { return its.valid && sc.meshes[its.mesh].emitter >= 0; }

// intersection.h:35-42 -> area.cpp:17-26 (one-sided, zero when the shape has no emitter) / envmap.cpp:47-56
template <bool ad> static V3<Real<ad>> Le(const Scene &sc, const Its<ad> &its, bool active) {
    using V = V3<Real<ad>>;
    if (!its.valid || !active) return V(Real<ad>(0.f));
    int e = sc.meshes[its.mesh].emitter;
    if (e < 0) return V(Real<ad>(0.f));
    if (sc.emitters[e].type == 1) {
        const V wi_world = to_world<ad>(its.sh, its.wi);
        return envmap_eval_direction<ad>(sc.env, -wi_world);
    }
    if (!(detach(its.wi.z) > 0.f)) return V(Real<ad>(0.f));
    return pick<ad>(sc.emitters[e].radiance);
}

template <bool ad> struct PositionSample { V3<Real<ad>> p, n; Real<ad> J; float pdf; bool valid; };

// mesh.cpp:413-454
template <bool ad> static PositionSample<ad> mesh_sample_position(const Scene &sc, const MeshC &m, float sx, float sy) {
    PositionSample<ad> r;
    float pdf_face;
    int fi = m.face_distrb.sample_reuse(sx, pdf_face);
    float a, b;
    square_to_uniform_triangle(sx, sy, a, b);
    const Tri &T = sc.tris[m.face_offset + fi];
    using R = Real<ad>;
    r.p = bilinear(pick<ad>(T.p0), pick<ad>(T.e1), pick<ad>(T.e2), R(a), R(b));
    r.n = pick<ad>(T.fn);
    if constexpr (ad) r.J = T.area / Dual(T.area.v); else r.J = 1.f;
    r.pdf = m.inv_total_area;
    r.valid = true;
    return r;
}

// EnvironmentMap::__sample_position (envmap.cpp:91-116): a direction from the cell grid, carried to the scene's
// bounding box; everything detached
template <bool ad> static PositionSample<ad> envmap_sample_position(const Scene &sc, const V3f &ref_p, float sx, float sy) {
    using R = Real<ad>;
    PositionSample<ad> r;
    float pdf, t, G;
    const V3f d = envmap_sample_direction(sc.env, sx, sy, pdf);
    V3f n;
    ray_intersect_scene_aabb(ref_p, d, sc.env.lower, sc.env.upper, t, n, G);
    const V3f p(fma_(d.x, t, ref_p.x), fma_(d.y, t, ref_p.y), fma_(d.z, t, ref_p.z));
    r.p = V3<R>(R(p.x), R(p.y), R(p.z));
    r.n = V3<R>(R(n.x), R(n.y), R(n.z));
    r.pdf = pdf * G;
    r.J = R(1.f);
    r.valid = true;
    return r;
}

void kat_env_sample(const Scene &sc, const V3f &ref_p, float sx, float sy, V3f &p, V3f &n, float &pdf) {
    const PositionSample<false> r = envmap_sample_position<false>(sc, ref_p, sx, sy);
    p = r.p; n = r.n; pdf = r.pdf;
}

template <bool ad> static PositionSample<ad> emitter_sample_position(const Scene &sc, int ei, const V3f &ref_p, float sx, float sy) {
    if (sc.emitters[ei].type == 1) return envmap_sample_position<ad>(sc, ref_p, sx, sy);
    return mesh_sample_position<ad>(sc, sc.meshes[sc.emitters[ei].mesh], sx, sy);
}

// scene.cpp:987-1013
template <bool ad> static PositionSample<ad> sample_emitter_position(const Scene &sc, const V3f &ref_p, float sx, float sy) {
    if (sc.emitters.size() == 1) return emitter_sample_position<ad>(sc, 0, ref_p, sx, sy);
    float epdf;
    int ei = sc.emitters_distrb.sample_reuse(sy, epdf);
    PositionSample<ad> r = emitter_sample_position<ad>(sc, ei, ref_p, sx, sy);
    r.pdf *= epdf;
    return r;
}

float kat_env_pdf(const Scene &sc, const V3f &ref_p, const V3f &p, const V3f &n);
// scene.cpp:1016-1024 -> area.cpp:48-59 -> mesh.cpp:457-466, or envmap.cpp:146-166
template <bool ad> static float emitter_position_pdf(const Scene &sc, const V3f &ref_p, const Its<ad> &its) {
    if (!its.valid) return 0.f;
    const MeshC &m = sc.meshes[its.mesh];
    if (m.emitter < 0) return 0.f;
    if (sc.emitters[m.emitter].type == 1) return kat_env_pdf(sc, ref_p, detach(its.p), detach(its.n));
    return sc.emitters[m.emitter].sampling_weight * m.inv_total_area;
}
float kat_env_pdf(const Scene &sc, const V3f &ref_p, const V3f &p, const V3f &n) {
    {
        V3f d = p - ref_p;
        const float dist2 = squared_norm(d);
        d = d / safe_sqrt(dist2);
        const float G = std::fabs(dot(d, n)) / dist2;
        d = transform_dir(detach(sc.env.from_world), d);
        const float factor = G * (1.f / std::sqrt(std::max(fma_(d.x, d.x, d.z * d.z), Epsilon * Epsilon))) * (.5f / (Pi * Pi));
        float u = atan2_cephes(d.x, -d.z) * InvTwoPi;
        float v = safe_acos_(d.y) * InvPi;
        u -= std::floor(u); v -= std::floor(v);
        return envmap_cell_pdf(sc.env, u, v) * factor;
    }
}

// MicrofacetPerVertex::__interpolate (microfacet_pv.cpp:145-160): v0 + (v1 - v0) bc.x + (v2 - v0) bc.y over the hit triangle's
// (mesh-local) vertex indices; the per-vertex values carry tangents, the barycentrics do where the intersection does
template <bool ad> static Dual pv_interp1(const std::vector<float> &v, const std::vector<float> &d, const int fi[3], const V2<Real<ad>> &bc) {
    auto at = [&](int i) { return Dual(v[i], d.empty() ? 0.f : d[i]); };
    const Dual v0 = at(fi[0]), v1 = at(fi[1]), v2 = at(fi[2]);
    return fma_(v1 - v0, Dual(bc.x), fma_(v2 - v0, Dual(bc.y), v0));
}
template <bool ad> static V3d pv_interp3(const std::vector<float> &v, const std::vector<float> &d, const int fi[3], const V2<Real<ad>> &bc) {
    Dual o[3];
    for (int c = 0; c < 3; ++c) {
        auto at = [&](int i) { return Dual(v[3 * i + c], d.empty() ? 0.f : d[3 * i + c]); };
        const Dual v0 = at(fi[0]), v1 = at(fi[1]), v2 = at(fi[2]);
        o[c] = fma_(v1 - v0, Dual(bc.x), fma_(v2 - v0, Dual(bc.y), v0));
    }
    return V3d(o[0], o[1], o[2]);
}

// ---------------------------------------------------------------- Diffuse BSDF (diffuse.cpp:24-108)
// a mesh without BSDF (the envmap's bounding cube): drjit's vcall on a null pointer returns zeros
// RoughConductor / RoughDielectric parameters with their optional bitmaps (roughconductor.cpp:37-43, roughdielectric.cpp:75-78):
// tex = eta map, spec_tex = k map, rough_tex = alpha map (both axes, as the XML loader fills them)
template <bool ad> static ConductorParams conductor_params(const BsdfC &b, const V2<Real<ad>> &uv) {
    ConductorParams P{b.alpha_u, b.alpha_v, b.eta, b.k, b.specular, b.two_sided};
    if (b.tex_w != 0) { const V3<Real<ad>> t = bsdf_reflectance<ad>(b, uv); P.eta = V3d(Dual(t.x), Dual(t.y), Dual(t.z)); }
    if (b.spec_w != 0) P.k = bsdf_specular<ad>(b, uv);
    if (b.rough_w != 0) { P.alpha_u = bsdf_roughness<ad>(b, uv); P.alpha_v = P.alpha_u; }
    return P;
}
template <bool ad> static DielectricParams dielectric_params(const BsdfC &b, const V2<Real<ad>> &uv) {
    DielectricParams P{b.alpha_u, b.alpha_v, b.eta.x, b.eta.y, b.two_sided};
    if (b.rough_w != 0) { P.alpha_u = bsdf_roughness<ad>(b, uv); P.alpha_v = P.alpha_u; }
    return P;
}

// (bid, wi) are explicit so that NormalMap can evaluate its nested BSDF with a perturbed incident direction
template <bool ad> static V3<Real<ad>> bsdf_eval_id(const Scene &sc, int bid, const Its<ad> &its, const V3<Real<ad>> &wi_, V3<Real<ad>> wo, bool active) {
    using R = Real<ad>; using V = V3<R>;
    if (bid < 0) return V(R(0.f));
    const BsdfC &b = sc.bsdfs[bid];
    if (b.type == 1) {          // Microfacet (microfacet.cpp); its diffuse reflectance is b.reflectance
        V3d diff = b.reflectance;
        if (b.tex_w != 0) { const V3<R> t = bsdf_reflectance<ad>(b, its.uv); diff = V3d(Dual(t.x), Dual(t.y), Dual(t.z)); }
        MicrofacetParams P{bsdf_specular<ad>(b, its.uv), diff, bsdf_roughness<ad>(b, its.uv), b.two_sided};
        return microfacet_eval<ad>(P, wi_, wo, active);
    }
    if (b.type == 2) {          // RoughConductor (roughconductor.cpp)
        const ConductorParams P = conductor_params<ad>(b, its.uv);
        return conductor_eval<ad>(P, wi_, wo, active);
    }
    if (b.type == 4) {          // MicrofacetPerVertex (microfacet_pv.cpp): parameters interpolated over the hit triangle's vertices
        MicrofacetParams P{pv_interp3<ad>(b.pv_spec, b.d_pv_spec, sc.tris[its.tri].fi, its.bc), pv_interp3<ad>(b.pv_diff, b.d_pv_diff, sc.tris[its.tri].fi, its.bc),
                           pv_interp1<ad>(b.pv_rough, b.d_pv_rough, sc.tris[its.tri].fi, its.bc), b.two_sided};
        return microfacet_pv_eval<ad>(P, wi_, wo, active);
    }
    if (b.type == 3) {          // RoughDielectric (roughdielectric.cpp); eta.x = intIOR / extIOR, eta.y = extIOR / intIOR
        const DielectricParams P = dielectric_params<ad>(b, its.uv);
        return dielectric_eval<ad>(P, wi_, wo, active);
    }
    R wiz = wi_.z;
    if (b.two_sided) { wo.z = mulsign(wo.z, detach(wiz)); wiz = abs_(wiz); }
    active = active && (detach(wiz) > 0.f && detach(wo.z) > 0.f);
    if (!active) return V(R(0.f));
    return bsdf_reflectance<ad>(b, its.uv) * R(InvPi) * wo.z;
}
template <bool ad> static float bsdf_pdf_id(const Scene &sc, int bid, const Its<ad> &its, const V3<Real<ad>> &wi_, const V3<Real<ad>> &wo_, bool active) {
    if (bid < 0) return 0.f;
    const BsdfC &b = sc.bsdfs[bid];
    if (b.type == 1) {
        MicrofacetParams P{b.specular, b.reflectance, bsdf_roughness<ad>(b, its.uv), b.two_sided};
        return microfacet_pdf(P, detach(wi_), detach(wo_), active);
    }
    if (b.type == 2) {
        const ConductorParams P = conductor_params<ad>(b, its.uv);
        return conductor_pdf(P, detach(wi_), detach(wo_), active);
    }
    if (b.type == 4) {
        MicrofacetParams P{b.specular, b.reflectance, pv_interp1<ad>(b.pv_rough, b.d_pv_rough, sc.tris[its.tri].fi, its.bc), b.two_sided};
        return microfacet_pdf(P, detach(wi_), detach(wo_), active);
    }
    if (b.type == 3) {
        const DielectricParams P = dielectric_params<ad>(b, its.uv);
        return dielectric_pdf(P, detach(wi_), detach(wo_), active);
    }
    float wiz = detach(wi_.z), woz = detach(wo_.z);
    if (b.two_sided) { woz = mulsign(woz, wiz); wiz = fabs(wiz); }
    active = active && (wiz > 0.f && woz > 0.f);
    return active ? InvPi * woz : 0.f;
}
struct BSDFSample { V3f wo; float pdf; bool valid; };
template <bool ad> static BSDFSample bsdf_sample_id(const Scene &sc, int bid, const Its<ad> &its, const V3<Real<ad>> &wi_, const float s3[3], bool active) {
    if (bid < 0) { BSDFSample z; z.wo = V3f(0.f, 0.f, 0.f); z.pdf = 0.f; z.valid = false; return z; }
    const BsdfC &b = sc.bsdfs[bid];
    if (b.type == 1) {
        MicrofacetParams P{b.specular, b.reflectance, bsdf_roughness<ad>(b, its.uv), b.two_sided};
        const MicrofacetSample m = microfacet_sample(P, detach(wi_), s3, active);
        BSDFSample r; r.wo = m.wo; r.pdf = m.pdf; r.valid = m.valid;
        return r;
    }
    if (b.type == 4) {
        MicrofacetParams P{b.specular, b.reflectance, pv_interp1<ad>(b.pv_rough, b.d_pv_rough, sc.tris[its.tri].fi, its.bc), b.two_sided};
        const MicrofacetSample m = microfacet_sample(P, detach(wi_), s3, active);
        BSDFSample r; r.wo = m.wo; r.pdf = m.pdf; r.valid = m.valid;
        return r;
    }
    if (b.type == 2) {
        const ConductorParams P = conductor_params<ad>(b, its.uv);
        const MicrofacetSample m = conductor_sample(P, detach(wi_), s3, active);
        BSDFSample r; r.wo = m.wo; r.pdf = m.pdf; r.valid = m.valid;
        return r;
    }
    if (b.type == 3) {
        const DielectricParams P = dielectric_params<ad>(b, its.uv);
        const MicrofacetSample m = dielectric_sample(P, detach(wi_), s3, active);
        BSDFSample r; r.wo = m.wo; r.pdf = m.pdf; r.valid = m.valid;
        return r;
    }
    float wiz = detach(wi_.z);
    if (b.two_sided) wiz = fabs(wiz);
    BSDFSample bs;
    bs.wo = square_to_cosine_hemisphere(s3[1], s3[2]);      // tail<2>(sample), diffuse.cpp:63
    bs.pdf = InvPi * bs.wo.z;
    bs.valid = active && (wiz > 0.f);
    return bs;
}

// ---------------------------------------------------------------- NormalMap (normalmap.cpp:20-187)
// wp = the normal of the map in the shading frame; the "tangent facet" wt closes the microsurface (Schuessler et al. 2017, without
// the i -> p -> t -> o path, which the reference comments out)
template <typename R> static V3<R> nm_wt(const V3<R> &wp) { return normalize(V3<R>(-wp.x, -wp.y, R(0.f))); }
template <typename R> static R nm_pdot(const V3<R> &a, const V3<R> &b) { const R d = dot(a, b); return detach(d) > 0.f ? d : R(0.f); }
template <typename R> static R nm_sin_theta(const V3<R> &v) { return safe_sqrt(fma_(v.x, v.x, v.y * v.y)); }       // frame.h:81 sin_theta_2 = x^2 + y^2
template <typename R> static R nm_G1(const V3<R> &wp, const V3<R> &w) {
    const R cw = detach(w.z) > 0.f ? w.z : R(0.f), cp = detach(wp.z) > 0.f ? wp.z : R(0.f);
    const R g = cw * cp / (nm_pdot(w, wp) + nm_pdot(w, nm_wt(wp)) * nm_sin_theta(wp));
    return detach(g) < 1.f ? g : R(1.f);                          // minimum(1, .): a NaN passes through, as in drjit
}
template <typename R> static R nm_lambda_p(const V3<R> &wp, const V3<R> &wi) {
    const R i_dot_p = nm_pdot(wp, wi);
    return i_dot_p / (i_dot_p + nm_pdot(nm_wt(wp), wi) * nm_sin_theta(wp));
}
// Frame(n, s): t = normalize(n x s), s = normalize(t x n)  (frame.h:43-46)
template <typename R> struct NmFrame {
    V3<R> s, t, n;
    NmFrame(const V3<R> &n_, const V3<R> &s_) : n(n_) { t = normalize(cross(n_, s_)); s = normalize(cross(t, n_)); }
    V3<R> to_local(const V3<R> &v) const { return V3<R>(dot(v, s), dot(v, t), dot(v, n)); }
    V3<R> to_world(const V3<R> &v) const { return s * v.x + t * v.y + n * v.z; }
};
template <bool ad> static NmFrame<Real<ad>> nm_frame(const BsdfC &b, const Its<ad> &its, V3<Real<ad>> &wp) {
    using R = Real<ad>; using V = V3<R>;
    const V c = bsdf_reflectance<ad>(b, its.uv);                  // m_nmap.eval<ad>(its.uv)
    wp = normalize(V(fma_(c.x, R(2.f), R(-1.f)), fma_(c.y, R(2.f), R(-1.f)), fma_(c.z, R(2.f), R(-1.f))));
    const V s = normalize(its.dp_du - wp * dot(wp, its.dp_du));   // fnmadd(wp, dot(wp, dp_du), dp_du): mixes the local wp with the world dp_du (kept)
    return NmFrame<R>(wp, s);
}
template <bool ad> static V3<Real<ad>> normalmap_eval(const Scene &sc, const BsdfC &b, const Its<ad> &its, V3<Real<ad>> wo, bool active) {
    using R = Real<ad>; using V = V3<R>;
    V wi = its.wi;
    if (b.two_sided) { wo.z = mulsign(wo.z, detach(wi.z)); wi.z = abs_(wi.z); }
    active = active && detach(wi.z) > 0.f && detach(wo.z) > 0.f;
    V wp;
    const NmFrame<R> pf = nm_frame<ad>(b, its, wp);
    const V pwi = pf.to_local(wi), pwo = pf.to_local(wo);
    const R shadowing = nm_G1(wp, wo), lambda_p = nm_lambda_p(wp, wi);
    const V wt = nm_wt(wp);
    V value = bsdf_eval_id<ad>(sc, b.nested, its, pwi, pwo, active) * lambda_p * shadowing;              // i -> p -> o
    if (detach(dot(wi, wt)) > 0.f) {                                                                      // i -> t -> p -> o
        const V wi_r = normalize(wi - wt * (R(2.0f) * dot(wi, wt)));
        value = value + bsdf_eval_id<ad>(sc, b.nested, its, pf.to_local(wi_r), pwo, active) * (R(1.f) - lambda_p) * shadowing;
    }
    return active ? value : V(R(0.f));
}
template <bool ad> static float normalmap_pdf(const Scene &sc, const BsdfC &b, const Its<ad> &its, const V3<Real<ad>> &wo_, bool active) {
    using R = Real<ad>; using V = V3<R>;
    V wi = its.wi, wo = wo_;
    if (b.two_sided) { wo.z = mulsign(wo.z, detach(wi.z)); wi.z = abs_(wi.z); }
    active = active && detach(wi.z) > 0.f && detach(wo.z) > 0.f;
    V wp;
    const NmFrame<R> pf = nm_frame<ad>(b, its, wp);
    const V pwo = pf.to_local(wo);
    const float prob = detach(nm_lambda_p(wp, wi));
    const V wt = nm_wt(wp);
    const V wi_r = normalize(wi - wt * (R(2.0f) * dot(wi, wt)));
    const float value = prob * bsdf_pdf_id<ad>(sc, b.nested, its, pf.to_local(wi), pwo, active)
                        + (1.f - prob) * bsdf_pdf_id<ad>(sc, b.nested, its, pf.to_local(wi_r), pwo, active);
    return active ? value : 0.f;
}
template <bool ad> static BSDFSample normalmap_sample(const Scene &sc, const BsdfC &b, const Its<ad> &its, const float s3[3], bool active) {
    using R = Real<ad>; using V = V3<R>;
    V wi = its.wi;
    if (b.two_sided) wi.z = abs_(wi.z);
    V wp;
    const NmFrame<R> pf = nm_frame<ad>(b, its, wp);
    const V pwi = pf.to_local(wi);
    const float prob = detach(nm_lambda_p(wp, wi));
    const V wt = nm_wt(wp);
    const bool itpo = s3[2] >= prob;
    BSDFSample bs = bsdf_sample_id<ad>(sc, b.nested, its, pwi, s3, active && !itpo);                    // i -> p -> o
    const V wi_r = normalize(wi - wt * (R(2.0f) * dot(wi, wt)));
    const V rwi = pf.to_local(wi_r);
    const BSDFSample bs2 = bsdf_sample_id<ad>(sc, b.nested, its, rwi, s3, active && itpo);              // i -> t -> p -> o
    if (itpo) bs.wo = bs2.wo;
    const V wo_l(R(bs.wo.x), R(bs.wo.y), R(bs.wo.z));
    const float pdf1 = bsdf_pdf_id<ad>(sc, b.nested, its, pwi, wo_l, active), pdf2 = bsdf_pdf_id<ad>(sc, b.nested, its, rwi, wo_l, active);
    bs.pdf = prob * pdf1 + (1.f - prob) * pdf2;
    const V3f wo_w = detach(pf.to_world(wo_l));
    bs.wo = wo_w;
    bs.valid = active && (bs.valid || bs2.valid);
    return bs;
}

// BSDF of the mesh the intersection lies on (null vcall = zeros for the envmap's bounding cube)
template <bool ad> static V3<Real<ad>> bsdf_eval(const Scene &sc, const Its<ad> &its, V3<Real<ad>> wo, bool active) {
    const int bid = sc.meshes[its.mesh].bsdf;
    if (bid >= 0 && sc.bsdfs[bid].type == 5) return normalmap_eval<ad>(sc, sc.bsdfs[bid], its, wo, active);
    return bsdf_eval_id<ad>(sc, bid, its, its.wi, wo, active);
}
template <bool ad> static float bsdf_pdf(const Scene &sc, const Its<ad> &its, const V3<Real<ad>> &wo_, bool active) {
    const int bid = sc.meshes[its.mesh].bsdf;
    if (bid >= 0 && sc.bsdfs[bid].type == 5) return normalmap_pdf<ad>(sc, sc.bsdfs[bid], its, wo_, active);
    return bsdf_pdf_id<ad>(sc, bid, its, its.wi, wo_, active);
}
template <bool ad> static BSDFSample bsdf_sample(const Scene &sc, const Its<ad> &its, const float s3[3], bool active) {
    const int bid = sc.meshes[its.mesh].bsdf;
    if (bid >= 0 && sc.bsdfs[bid].type == 5) return normalmap_sample<ad>(sc, sc.bsdfs[bid], its, s3, active);
    return bsdf_sample_id<ad>(sc, bid, its, its.wi, s3, active);
}

// ---------------------------------------------------------------- PathTracer::__Li (path.cpp:35-127)
// Always consumes exactly 5*max_depth sampler draws (the reference draws for masked lanes too).
template <bool ad>
V3<Real<ad>> Li(const Scene &sc, LaneSampler &sampler, const Ray<ad> &ray_, bool active, int max_depth, bool hide_emitters) {
    using R = Real<ad>; using V = V3<R>;
    Ray<ad> curr = ray_;
    Its<ad> its = ray_intersect<ad, false>(sc, curr, active);
    active = active && its.valid;
    V throughput(R(1.f));
    if (sc.field >= 0) {
        // FieldExtractionIntegrator::__Li (field.cpp:49-121) / CollocatedIntegrator::__Li (collocated.cpp:24-55): first hit only
        bool ok = its.valid;
        if (sc.env_emitter >= 0 && sc.field != 8) ok = ok && its.valid && sc.meshes[its.mesh].bsdf >= 0;     // field.cpp:55-58
        if (sc.field_object >= 0) ok = ok && its.valid && its.mesh == sc.field_object;
        if (!ok) return V(R(0.f));
        switch (sc.field) {
            case 0: return V(R(1.f));
            case 1: return its.p;
            case 2: return V(its.t);
            case 3: return its.n;
            case 4: return its.sh.n;
            case 5: return V(its.uv.x, its.uv.y, R(0.f));
            case 6: return bsdf_eval<ad>(sc, its, its.wi, true);
            case 7: return V(R((float) (its.mesh + 1)));
            default: {
                V r = bsdf_eval<ad>(sc, its, its.wi, true) / sqr(its.t);
                if constexpr (ad) return r * sc.intensity; else return r * sc.intensity.v;
            }
        }
    }
    V result = hide_emitters ? V(R(0.f)) : Le<ad>(sc, its, active);
    // DirectIntegrator(mis) (reference src/integrator/direct.cpp:34-132) is the same body run once: mis = 0 draws and uses
    // only the emitter sample (weight 1), mis = 1 only the BSDF sample (weight 1), mis = 2 both with MIS; -1 = PathTracer
    const int mis = sc.direct_mis;
    const int nd = mis == 0 ? 2 : (mis == 1 ? 3 : 5);
    if (mis >= 0) max_depth = 1;
    for (int depth = 0; depth < max_depth; ++depth) {
        if (!active) { sampler.rng.advance((uint64_t) nd * (max_depth - depth)); break; }
        if (mis != 1) {   // next-event estimation
            float sx = sampler.next_1d(), sy = sampler.next_1d();
            PositionSample<ad> ps = sample_emitter_position<ad>(sc, detach(its.p), sx, sy);
            bool active_direct = active && ps.valid && !is_emitter<ad>(sc, its);
            V wod = ps.p - its.p;
            R dist_sqr = squared_norm(wod);
            R dist = safe_sqrt(dist_sqr);
            wod /= dist;
            Ray<ad> ray1{its.p, wod};
            Its<ad> its1 = ray_intersect<ad, ad>(sc, ray1, active_direct);
            active_direct = active_direct && its1.valid;
            active_direct = active_direct && (detach(its1.t) > detach(dist) - ShadowEpsilon) && is_emitter<ad>(sc, its1);
            if (active_direct) {
                R cos_val = dot(its1.n, -wod);
                R G_val = abs_(cos_val) / dist_sqr;
                V emitter_val = Le<ad>(sc, its1, active);
                V wo_local = to_local<ad>(its.sh, wod);
                V bsdf_val2 = bsdf_eval<ad>(sc, its, wo_local, active_direct);
                bsdf_val2 *= G_val * ps.J / R(ps.pdf);
                float pdf1 = bsdf_pdf<ad>(sc, its, wo_local, active_direct) * detach(G_val);
                active_direct = active_direct && (pdf1 != 0.f);
                float weight1 = mis == 0 ? 1.f : mis_weight(ps.pdf, pdf1);
                if (active_direct) result += throughput * emitter_val * bsdf_val2 * R(weight1);
            }
        }
        if (mis != 0) {   // BSDF sampling
            float s3[3];
            s3[0] = sampler.next_1d(); s3[1] = sampler.next_1d(); s3[2] = sampler.next_1d();
            BSDFSample bs = bsdf_sample<ad>(sc, its, s3, active);
            curr = Ray<ad>{its.p, to_world<ad>(its.sh, V(R(bs.wo.x), R(bs.wo.y), R(bs.wo.z)))};
            Its<ad> its1 = ray_intersect<ad, ad>(sc, curr, active);
            active = active && bs.valid && its1.valid;
            if (!active) continue;       // next iteration fast-forwards the sampler
            V bsdf_val;
            float pdf0;
            if constexpr (ad) {
                V wo = its1.p - its.p;
                wo /= its1.t;
                R cos_val = dot(its1.n, -wo);
                R G_val = abs_(cos_val) / sqr(its1.t);
                pdf0 = bs.pdf * G_val.v;
                if (its1.t.v < Epsilon) bsdf_val = V(R(0.f));
                else bsdf_val = bsdf_eval<ad>(sc, its, to_local<ad>(its.sh, wo), active) * G_val * its1.J / R(pdf0);
            } else {
                float cos_val = dot(its1.n, -curr.d);
                float G_val = fabs(cos_val) / sqr(its1.t);
                pdf0 = bs.pdf * G_val;
                if (its1.t < Epsilon) bsdf_val = V(0.f);
                else bsdf_val = bsdf_eval<ad>(sc, its, V(bs.wo.x, bs.wo.y, bs.wo.z), active) / bs.pdf;
            }
            float weight2 = mis == 1 ? 1.f : mis_weight(pdf0, emitter_position_pdf<ad>(sc, detach(its.p), its1));
            throughput *= bsdf_val;
            result += Le<ad>(sc, its1, active) * throughput * R(weight2);
            its = its1;
        }
    }
    return result;
}
template V3f Li<false>(const Scene &, LaneSampler &, const RayC &, bool, int, bool);
template V3d Li<true>(const Scene &, LaneSampler &, const RayD &, bool, int, bool);

// ---------------------------------------------------------------- secondary edges
// scene.cpp:1027-1068
BoundarySegSampleDirect sample_boundary_segment_direct(const Scene &sc, V3f sample3) {
    BoundarySegSampleDirect r;
    float sample1 = sample3.x, pdf0;
    int ei = sc.sec_edge_distrb.sample_reuse(sample1, pdf0);
    const SecEdge &info = sc.sec_edges[ei];
    Dual s1(sample1);
    r.p0 = V3d(fma_(info.e1.x, s1, info.p0.x), fma_(info.e1.y, s1, info.p0.y), fma_(info.e1.z, s1, info.p0.z));
    V3f e1 = detach(info.e1);
    r.edge = normalize(e1);
    r.edge2 = detach(info.p2) - detach(info.p0);
    V3f p0 = detach(r.p0);
    pdf0 /= norm(e1);
    PositionSample<false> ps2 = sample_emitter_position<false>(sc, p0, sample3.y, sample3.z);
    r.p2 = ps2.p; r.n = ps2.n;
    V3f e = r.p2 - p0;
    float distSqr = squared_norm(e);
    e = e / safe_sqrt(distSqr);
    float cosTheta = dot(r.n, -e);
    int sgn0 = sign_eps(dot(detach(info.n0), e), EdgeEpsilon), sgn1 = sign_eps(dot(detach(info.n1), e), EdgeEpsilon);
    r.valid = (cosTheta > Epsilon) && ((info.is_boundary && sgn0 != 0) || (!info.is_boundary && sgn0 * sgn1 < 0));
    r.pdf = r.valid ? pdf0 * ps2.pdf * (distSqr / cosTheta) : 0.f;
    return r;
}

// path.cpp:171-270
template <bool ad>
int eval_secondary_edge(const Scene &sc, const CameraC &cam, const V3f &sample3, V3<Real<ad>> &value) {
    using R = Real<ad>; using V = V3<R>;
    value = V(R(0.f));
    BoundarySegSampleDirect bss = sample_boundary_segment_direct(sc, sample3);
    bool valid = bss.valid;
    V3f _p0 = detach(bss.p0), _p2 = bss.p2, _dir = normalize(_p2 - _p0);

    int tri2 = -1;
    Its<false> _its2 = ray_intersect<false, false>(sc, RayC{_p0, _dir}, valid, &tri2);
    valid = valid && is_emitter<false>(sc, _its2) && _its2.valid && norm(_its2.p - _p2) < ShadowEpsilon;

    Its<false> _its1 = ray_intersect<false, false>(sc, RayC{_p0, -_dir}, valid);
    valid = valid && _its1.valid;
    if (!valid) return -1;
    V3f _p1 = _its1.p;

    SensorDirectSample sds = sample_direct(sc, cam, _p1);
    valid = valid && sds.valid;
    if (!valid) return -1;

    Ray<ad> camera_ray = sample_primary_ray<ad>(cam, V2<R>(R(sds.q.x), R(sds.q.y)));
    Its<ad> its1 = ray_intersect<ad, false>(sc, camera_ray, valid);
    valid = valid && its1.valid && norm(detach(its1.p) - _p1) < ShadowEpsilon;
    if (!valid) return -1;
    valid = valid && sc.meshes[its1.mesh].bsdf >= 0;
    if (!valid) return -1;

    float dist = norm(_p2 - _p1), cos2 = fabs(dot(bss.n, -_dir));
    V3f e = cross(bss.edge, _dir);
    float sinphi = norm(e);
    V3f proj = normalize(cross(e, bss.n));
    float sinphi2 = norm(cross(_dir, proj));
    float base_v = (_its1.t / dist) * (sinphi / sinphi2) * cos2;
    valid = valid && (sinphi > Epsilon) && (sinphi2 > Epsilon);
    if (!valid) return -1;

    V3f d0 = -detach(camera_ray.d);
    V3f d0_local = to_local<false>(_its1.sh, d0);
    V3f bsdf_val = bsdf_eval<false>(sc, _its1, d0_local, valid);
    float correction = fabs((_its1.wi.z * dot(d0, _its1.n)) / (d0_local.z * dot(_dir, _its1.n)));
    bsdf_val *= correction;
    V3f value0 = bsdf_val * Le<false>(sc, _its2, valid) * (base_v * sds.sensor_val / bss.pdf);

    if constexpr (ad) {
        V3f n = normalize(cross(bss.n, proj));
        value0 *= sign1(dot(e, bss.edge2)) * sign1(dot(e, n));
        const Tri &T = sc.tris[tri2];
        V3d sd = normalize(bss.p0 - its1.p);
        Dual u, v, t;
        ray_intersect_triangle<Dual>(T.p0, T.e1, T.e2, its1.p, sd, u, v, t);
        V3d u2 = bilinear(V3d(detach(T.p0)), V3d(detach(T.e1)), V3d(detach(T.e2)), u, v);
        V3d result = V3d(value0) * dot(V3d(n), u2);
        value = V3d(Dual(0.f, result.x.d), Dual(0.f, result.y.d), Dual(0.f, result.z.d));   // result - detach(result)
        return sds.pixel_idx;
    } else {
        value = value0;      // guiding: value without the normal velocity (path.cpp:267-268)
        return -1;
    }
}
template int eval_secondary_edge<false>(const Scene &, const CameraC &, const V3f &, V3f &);
template int eval_secondary_edge<true>(const Scene &, const CameraC &, const V3f &, V3d &);

// cube_distrb.cpp:34-40
float Guiding::sample_reuse(V3f &s) const {
    float pdf;
    int idx = distrb.sample_reuse(s.z, pdf);
    int c0 = idx / (reso[1] * reso[2]);
    int rem = idx - c0 * (reso[1] * reso[2]);
    int c1 = rem / reso[2];
    int c2 = rem - c1 * reso[2];
    s.x = (s.x + (float) c0) * unit[0];
    s.y = (s.y + (float) c1) * unit[1];
    s.z = (s.z + (float) c2) * unit[2];
    return pdf * (float) num_cells;
}

// exported for the known-answer tests
static MicrofacetParams kat_params(const float *p, int two_sided) {
    MicrofacetParams P;
    P.specular = V3d(Dual(p[0], p[7]), Dual(p[1], p[8]), Dual(p[2], p[9]));
    P.diffuse = V3d(Dual(p[3], p[10]), Dual(p[4], p[11]), Dual(p[5], p[12]));
    P.roughness = Dual(p[6], p[13]);
    P.two_sided = two_sided != 0;
    return P;
}
void kat_microfacet_eval(const float *params, int two_sided, const float *wi, const float *wo, float *out) {
    const V3d r = microfacet_eval<true>(kat_params(params, two_sided), V3d(V3f(wi[0], wi[1], wi[2])), V3d(V3f(wo[0], wo[1], wo[2])), true);
    out[0] = r.x.v; out[1] = r.y.v; out[2] = r.z.v; out[3] = r.x.d; out[4] = r.y.d; out[5] = r.z.d;
}
float kat_microfacet_pdf(float roughness, int two_sided, const float *wi, const float *wo) {
    MicrofacetParams P; P.roughness = Dual(roughness); P.two_sided = two_sided != 0;
    return microfacet_pdf(P, V3f(wi[0], wi[1], wi[2]), V3f(wo[0], wo[1], wo[2]), true);
}
int kat_microfacet_sample(float roughness, int two_sided, const float *wi, const float *s3, float *wo_out, float *pdf_out) {
    MicrofacetParams P; P.roughness = Dual(roughness); P.two_sided = two_sided != 0;
    const MicrofacetSample m = microfacet_sample(P, V3f(wi[0], wi[1], wi[2]), s3, true);
    wo_out[0] = m.wo.x; wo_out[1] = m.wo.y; wo_out[2] = m.wo.z; *pdf_out = m.pdf;
    return m.valid ? 1 : 0;
}
float kat_ggx_eval(float alpha, const float *m) { GGX<float> g{alpha, alpha}; return g.eval(V3f(m[0], m[1], m[2])); }
float kat_fresnel_conductor(float eta, float k, float c) { return fresnel_conductor<float>(eta, k, c); }
// RoughDielectric hooks: q = {alpha, eta}; eval out = value (scalar), d/d(alpha), d/d(eta)
static DielectricParams kat_dielectric_params(const float *q, float da, float de) {
    DielectricParams P; P.alpha_u = Dual(q[0], da); P.alpha_v = Dual(q[0], da); P.eta = Dual(q[1], de);
    P.inv_eta = Dual(1.f / q[1], -de / (q[1] * q[1])); P.two_sided = false;
    return P;
}
void kat_dielectric_eval(const float *q, const float *wi, const float *wo, float *out) {
    const V3f a(wi[0], wi[1], wi[2]), b(wo[0], wo[1], wo[2]);
    out[0] = dielectric_eval<false>(kat_dielectric_params(q, 0.f, 0.f), a, b, true).x;
    out[1] = dielectric_eval<true>(kat_dielectric_params(q, 1.f, 0.f), V3d(a), V3d(b), true).x.d;
    out[2] = dielectric_eval<true>(kat_dielectric_params(q, 0.f, 1.f), V3d(a), V3d(b), true).x.d;
}
float kat_dielectric_pdf(const float *q, const float *wi, const float *wo) {
    return dielectric_pdf(kat_dielectric_params(q, 0.f, 0.f), V3f(wi[0], wi[1], wi[2]), V3f(wo[0], wo[1], wo[2]), true);
}
int kat_dielectric_sample(const float *q, const float *wi, const float *s3, float *wo_out, float *pdf_out) {
    const MicrofacetSample m = dielectric_sample(kat_dielectric_params(q, 0.f, 0.f), V3f(wi[0], wi[1], wi[2]), s3, true);
    wo_out[0] = m.wo.x; wo_out[1] = m.wo.y; wo_out[2] = m.wo.z; *pdf_out = m.pdf;
    return m.valid ? 1 : 0;
}
void kat_fresnel_dielectric(float eta, float c, float *out) {
    float F, ct, it, ti; fresnel_dielectric<float>(eta, c, F, ct, it, ti);
    out[0] = F; out[1] = ct; out[2] = it; out[3] = ti;
}
void kat_cosine_hemisphere(float sx, float sy, float *o) { V3f v = square_to_cosine_hemisphere(sx, sy); o[0] = v.x; o[1] = v.y; o[2] = v.z; }
void kat_uniform_triangle(float sx, float sy, float *o) { square_to_uniform_triangle(sx, sy, o[0], o[1]); }
void kat_coordinate_system(const float *n, float *s, float *t) {
    V3f S, T; coordinate_system(V3f(n[0], n[1], n[2]), S, T);
    s[0] = S.x; s[1] = S.y; s[2] = S.z; t[0] = T.x; t[1] = T.y; t[2] = T.z;
}

} // namespace orc
