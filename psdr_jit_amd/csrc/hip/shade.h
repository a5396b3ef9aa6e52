// shade.h — per-lane shading code of the path tracer: Scene::ray_intersect post-processing,
// area-light sampling, the diffuse BSDF and PathTracer::__Li, all in registers.
// The reference materialises every quantity below as an N-lane device array between OptiX launches
// (SURVEY.md §2 kernel table); here one lane = one thread and nothing leaves the register file.
// AD=false is the reference's C instantiation, AD=true its D instantiation (one forward tangent).
#pragma once
#include "scene_dev.h"
#include "sampler.h"
#include "../common/envmath.h"
#include "microfacet.h"

namespace psdr {

template <bool AD> struct RayT { VecN<AD> o, d; };

// reference include/psdr/core/intersection.h:24-60 — only the members the diffuse path reads
template <bool AD> struct Its {
    bool valid;
    int slot, mesh;
    VecN<AD> p, n, wi, fs, ft, fn;     // position, geometric normal, local incident dir, shading frame
    Num<AD> t, J;
    VecN<AD> dp_du;                    // world-space dp/du (Intersection::dp_du), zeros without a uv parameterisation; scene class 0 only
    Num<AD> bu, bv;                    // barycentrics (Intersection::bc); only kept when a per-vertex BSDF exists (LDS=false kernels)
    Num<AD> tu, tv;                    // texture coordinates (Intersection::uv); only kept by the LDS=false kernels of textured scenes
};

template <bool AD> PSDR_DEV VecN<AD> to_local(const Its<AD> &its, const VecN<AD> &v) { return VecN<AD>(dot(v, its.fs), dot(v, its.ft), dot(v, its.fn)); }
template <bool AD> PSDR_DEV VecN<AD> to_world(const Its<AD> &its, const Vec3f &v) {
    return its.fs * Num<AD>(v.x) + its.ft * Num<AD>(v.y) + its.fn * Num<AD>(v.z);
}
PSDR_DEV Vec3d to_world_d(const Its<true> &its, const Vec3d &v) { return its.fs * v.x + its.ft * v.y + its.fn * v.z; }

// reference include/psdr/core/frame.h:9-28 (Duff et al. 2017)
template <typename T> PSDR_DEV void coordinate_system(const Vec3<T> &n, Vec3<T> &s, Vec3<T> &t) {
    const float nz = detach(n.z);
    const float sg = signbit_(nz) ? -1.f : 1.f;
    T a = -rcp_(T(sg) + n.z);
    T b = n.x * n.y * a;
    s = Vec3<T>(mulsign(sqr(n.x) * a, nz) + T(1.f), mulsign(b, nz), -mulsign(n.x, nz));
    t = Vec3<T>(b, T(sg) + sqr(n.y) * a, -n.y);
}

template <bool AD, int LDS> struct TriData { VecN<AD> p0, e1, e2; };

template <bool AD, int LDS> PSDR_DEV VecN<AD> pick3(const Vec3f &v, const Vec3f &d) {
    if constexpr (AD) return make_dual(v, d); else return v;
}

// load p0,e1,e2 (+tangents) of a triangle slot
template <bool AD, int LDS> PSDR_DEV void load_geom(const SceneView<LDS> &S, int slot, VecN<AD> &p0, VecN<AD> &e1, VecN<AD> &e2) {
    const SceneTables &T = *S.T;
    const int w = T.trav_off + 3 * slot;
    const float4 a = S.ld(w), b = S.ld(w + 1), c = S.ld(w + 2);
    Vec3f vp0(a.x, a.y, a.z), ve1(a.w, b.x, b.y), ve2(b.z, b.w, c.x);
    if constexpr (AD) {
        if (S.tan_on()) {
            const float4 ta = S.tanw(slot, 0), tb = S.tanw(slot, 1), tc = S.tanw(slot, 2);
            p0 = make_dual(vp0, Vec3f(ta.x, ta.y, ta.z)); e1 = make_dual(ve1, Vec3f(ta.w, tb.x, tb.y)); e2 = make_dual(ve2, Vec3f(tb.z, tb.w, tc.x));
        } else { p0 = promote(vp0); e1 = promote(ve1); e2 = promote(ve2); }
    } else { p0 = vp0; e1 = ve1; e2 = ve2; }
}

template <typename T> PSDR_DEV void ray_tri_uvt(const Vec3<T> &p0, const Vec3<T> &e1, const Vec3<T> &e2, const Vec3<T> &o, const Vec3<T> &d, T &u, T &v, T &t) {
    Vec3<T> h = cross(d, e2);
    T a = dot(e1, h);
    T f = rcp_(a);
    Vec3<T> s = o - p0;
    u = f * dot(s, h);
    Vec3<T> q = cross(s, e1);
    v = f * dot(d, q);
    t = f * dot(e2, q);
}

// Scene::ray_intersect<ad, path_space> after the trace, reference src/scene/scene.cpp:621-806.
// `path_space` is a run-time flag so that lanes at different path depths can share one instruction stream
// (it only matters in AD mode: first camera hit = solid-angle form, later hits = material form).
// FRAME=false skips the tangent frame (shadow rays only need n, wi.z, t, J).
template <bool AD, int LDS, bool FRAME>
PSDR_DEV Its<AD> make_its(const SceneView<LDS> &S, const Hit &h, const RayT<AD> &ray, bool path_space) {
    using R = Num<AD>; using V = VecN<AD>;
    Its<AD> its;
    its.valid = false; its.slot = -1; its.mesh = -1; its.t = R(0.f); its.J = R(1.f);
    if (h.slot < 0) return its;
    const SceneTables &T = *S.T;
    its.valid = true; its.slot = h.slot;
    V p0, e1, e2;
    load_geom<AD, LDS>(S, h.slot, p0, e1, e2);
    const int w = T.shade_off + 6 * h.slot;
    const float4 s0 = S.ld(w), s1 = S.ld(w + 1), s2 = S.ld(w + 2), s3 = S.ld(w + 3);
    its.mesh = __float_as_int(s1.w);
    const bool flat = (__float_as_int(s2.w) & 1) != 0;
    V n0, n1, n2;
    if constexpr (AD) {
        if (S.tan_on()) {
            const float4 tc = S.tanw(h.slot, 2), td = S.tanw(h.slot, 3), te = S.tanw(h.slot, 4), tf = S.tanw(h.slot, 5);
            n0 = make_dual(Vec3f(s0.x, s0.y, s0.z), Vec3f(tc.y, tc.z, tc.w));
            n1 = make_dual(Vec3f(s1.x, s1.y, s1.z), Vec3f(td.x, td.y, td.z));
            n2 = make_dual(Vec3f(s2.x, s2.y, s2.z), Vec3f(td.w, te.x, te.y));
            its.n = make_dual(Vec3f(s3.x, s3.y, s3.z), Vec3f(te.z, te.w, tf.x));
            if (path_space) its.J = Dual(s0.w, tf.y) / Dual(s0.w);      // area / detach(area), scene.cpp:680
        } else {
            n0 = promote(Vec3f(s0.x, s0.y, s0.z)); n1 = promote(Vec3f(s1.x, s1.y, s1.z)); n2 = promote(Vec3f(s2.x, s2.y, s2.z));
            its.n = promote(Vec3f(s3.x, s3.y, s3.z));
        }
    } else {
        n0 = Vec3f(s0.x, s0.y, s0.z); n1 = Vec3f(s1.x, s1.y, s1.z); n2 = Vec3f(s2.x, s2.y, s2.z);
        its.n = Vec3f(s3.x, s3.y, s3.z);
    }
    R u, v;
    V dir_in;
    if (!AD || path_space) {
        u = R(h.u); v = R(h.v);                         // detached barycentrics from the tracer
        its.p = madd3(e1, u, e2, v, p0);
        V dir = its.p - ray.o;
        its.t = norm(dir);
        dir_in = dir / its.t;
    } else {
        if constexpr (AD) {
            R t;
            ray_tri_uvt<R>(p0, e1, e2, ray.o, ray.d, u, v, t);     // differentiable re-intersection, scene.cpp:774
            its.p = V(fma_(ray.d.x, t, ray.o.x), fma_(ray.d.y, t, ray.o.y), fma_(ray.d.z, t, ray.o.z));
            its.t = t;
            dir_in = ray.d;
        }
    }
    V nb = madd3(n1 - n0, u, n2 - n0, v, n0);                       // the vertex normals enter only through this blend
    if constexpr (AD) {
        // reverse mode, probe kind 8: a unit tangent on one component of the blend of ONE traced hit (the record just consumed);
        // the kernel spreads the result over n0, n1, n2 with the barycentric weights - 3 probes per hit instead of 9 per triangle
        if (S.probe_kind == 8 && S.rec_i - 1 == S.probe_id) { Dual &q = S.probe_comp == 0 ? nb.x : (S.probe_comp == 1 ? nb.y : nb.z); q.d += 1.f; }
    }
    V sh_n = normalize(nb);
    if (flat) sh_n = its.n;
    its.fn = sh_n;
    if constexpr (FRAME) {
        coordinate_system(sh_n, its.fs, its.ft);
        // tangent frame from the uv parameterisation when it is non-degenerate (scene.cpp:724-766)
        const float4 s4 = S.ld(w + 4), s5 = S.ld(w + 5);
        const float du0x = s4.z - s4.x, du0y = s4.w - s4.y, du1x = s5.x - s4.x, du1y = s5.y - s4.y;
        if constexpr (has_mat(LDS)) {              // its.uv = bilinear2(uv0, uv1 - uv0, uv2 - uv0, barycentrics), scene.cpp:715/779
            if (T.tex != nullptr || S.field == 5) {
                its.tu = fma_(R(du0x), u, fma_(R(du1x), v, R(s4.x)));
                its.tv = fma_(R(du0y), u, fma_(R(du1y), v, R(s4.y)));
            }
        }
        if constexpr (has_mat(LDS)) if (T.pv != nullptr) { its.bu = u; its.bv = v; }
        if constexpr (has_mat(LDS)) its.dp_du = V(R(0.f));
        const float det = fma_(du0x, du1y, -(du0y * du1x));
        if (det != 0.f) {
            const float inv_det = 1.f / det;
            V dp_du = (e1 * R(du1y) - e2 * R(du0y)) * R(inv_det);
            if constexpr (has_mat(LDS)) its.dp_du = dp_du;
            its.fs = normalize(dp_du - sh_n * dot(sh_n, dp_du));
            its.ft = cross(sh_n, its.fs);
        }
        its.wi = to_local<AD>(its, -dir_in);
    } else {
        its.wi = V(R(0.f), R(0.f), dot(-dir_in, sh_n));
    }
    return its;
}

// Scene::ray_intersect<ad, path_space>, reference src/scene/scene.cpp:612-806
template <bool AD, bool PATH_SPACE, int LDS, bool COUNT, bool FRAME = true>
PSDR_DEV Its<AD> ray_intersect(SceneView<LDS> &S, const RayT<AD> &ray, bool active) {
    static_assert(AD || !PATH_SPACE, "path-space needs AD");
    Hit h; h.slot = -1; h.u = h.v = h.t = 0.f;
    if (active) h = trace<LDS, COUNT>(S, detach(ray.o), detach(ray.d));
    if (COUNT) { if (h.slot >= 0) S.c_hits++; }
    return make_its<AD, LDS, FRAME>(S, h, ray, PATH_SPACE);
}

// ---------------------------------------------------------------- records from the blob
struct MeshRec { int bsdf, emitter, face_offset, n_faces; float inv_total_area; int distrb_offset; float distrb_sum; };
template <int LDS> PSDR_DEV MeshRec load_mesh(const SceneView<LDS> &S, int mesh) {
    const int w = S.T->mesh_off + 2 * mesh;
    const float4 a = S.ld(w), b = S.ld(w + 1);
    MeshRec m;
    m.bsdf = __float_as_int(a.x); m.emitter = __float_as_int(a.y); m.face_offset = __float_as_int(a.z); m.n_faces = __float_as_int(a.w);
    m.inv_total_area = b.x; m.distrb_offset = __float_as_int(b.y); m.distrb_sum = b.z;
    return m;
}
template <int LDS> PSDR_DEV int mesh_emitter(const SceneView<LDS> &S, int mesh) { return __float_as_int(S.ld(S.T->mesh_off + 2 * mesh).y); }
template <int LDS> PSDR_DEV int mesh_bsdf(const SceneView<LDS> &S, int mesh) { return __float_as_int(S.ld(S.T->mesh_off + 2 * mesh).x); }

// DiscreteDistribution::sample_reuse, reference src/core/pmf.cpp:26-45 (size 1 leaves the sample untouched)
// EXACT: a correctly rounded pdf as well (the environment-map sampler, which psdr_hip_env_sample exposes for bit-exact comparison);
// everywhere else the pdf is weight arithmetic (dmath.h)
template <bool EXACT = false, typename PmfFn, typename CmfFn>
PSDR_DEV int sample_reuse(int size, float sum, PmfFn pmf, CmfFn cmf, float &s, float &pdf) {
    if (size == 1) { pdf = 1.f; return 0; }
    s *= sum;
    int lo = 0, hi = size - 1;               // first i in [0, size-1) with !(cmf[i] < s), else size-1
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cmf(mid) < s) lo = mid + 1; else hi = mid; }
    const int idx = lo;
    if (idx > 0) s -= cmf(idx - 1);
    const float p = pmf(idx);
    if (p > 0.f) s /= p;                                 // (the re-used sample places the next point: geometry)
    s = fminf(fmaxf(s, 0.f), 1.f);
    pdf = EXACT ? p / sum : fdiv(p, sum);
    return idx;
}

// The same function for the large tables (2 M cells of a 1024 x 512 environment map, the cells of a guiding grid): 21 dependent loads
// from a 8-16 MB array per sample, the lower levels of the search cold in L2 - measured on BASELINE config 5 as a quarter of all L2
// misses of the path kernels.  A GUIDE TABLE built with the distribution narrows the search before it starts: bucket k of guide_n
// (a power of two, so k / guide_n and floor(s guide_n) are exact) holds the answer for the smallest sample of the bucket, bucket k + 1
// the answer for the smallest sample of the next one; the answer is monotone in the sample, so it lies between the two, and a
// binary search between ANY bounds that enclose the answer returns that answer (the running sums are non-decreasing).  Same index,
// same re-used sample, same pdf - two loads from a table of a few hundred KB and ~4 from one or two cache lines instead of 21.
template <bool EXACT = false, typename PmfFn, typename CmfFn>
PSDR_DEV int sample_reuse_guided(const int *guide, int guide_n, int size, float sum, PmfFn pmf, CmfFn cmf, float &s, float &pdf) {
    if (size == 1) { pdf = 1.f; return 0; }
    int lo = 0, hi = size - 1;
    if (guide_n > 0) {
        int k = (int) (s * (float) guide_n);
        k = k < 0 ? 0 : (k > guide_n - 1 ? guide_n - 1 : k);
        lo = guide[k]; hi = guide[k + 1];
    }
    s *= sum;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cmf(mid) < s) lo = mid + 1; else hi = mid; }
    const int idx = lo;
    if (idx > 0) s -= cmf(idx - 1);
    const float p = pmf(idx);
    if (p > 0.f) s /= p;
    s = fminf(fmaxf(s, 0.f), 1.f);
    pdf = EXACT ? p / sum : fdiv(p, sum);
    return idx;
}

// ---------------------------------------------------------------- EnvironmentMap (reference src/emitter/envmap.cpp)
// hooks of csrc/common/envmath.h for the (value, tangent) type; derivatives of atan2 / acos are the analytic ones
PSDR_DEV Dual e_fma(const Dual &a, const Dual &b, const Dual &c) { return fma_(a, b, c); }
PSDR_DEV Dual e_floor(const Dual &a) { return Dual(floorf(a.v), 0.f); }
PSDR_DEV float e_value(const Dual &a) { return a.v; }
PSDR_DEV void e_sincos(const Dual &a, Dual &s, Dual &c) { float sv, cv; env::sincos_f(a.v, sv, cv); s = Dual(sv, cv * a.d); c = Dual(cv, -(sv * a.d)); }
// a bitmap's uv transform (bitmap.h:37-39) as (value, tangent): the forward tangents of rotate / scale / translate in a render; none in the
// replays and sweeps of reverse mode (their adjoints: tex_xf_adjoint / env_xf_adjoint below)
PSDR_DEV env::UvXf<Dual> uv_xf_d(const float *xf, const float *d_xf, bool tan) {
    return env::UvXf<Dual>(Dual(xf[0], tan ? d_xf[0] : 0.f), Dual(xf[1], tan ? d_xf[1] : 0.f), Dual(xf[2], tan ? d_xf[2] : 0.f), Dual(xf[3], tan ? d_xf[3] : 0.f));
}
// reverse mode of rotate / scale / translate (bitmap.cpp:64-86): the adjoint ob[CH] of ONE lookup's output taken to the four members by four forward
// evaluations of that lookup with unit tangents (the lookup is a handful of multiply-adds), added to acc4 = [rot, scale, tx, ty] (an LDS accumulator)
template <int CH> PSDR_DEV void tex_xf_adjoint(const TexDev &td, float tu, float tv, const float *ob, float *acc4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float e[4] = {j == 0 ? 1.f : 0.f, j == 1 ? 1.f : 0.f, j == 2 ? 1.f : 0.f, j == 3 ? 1.f : 0.f};
        Dual o[CH];
        env::bitmap_eval_tex<Dual, CH>([&](int i, int c) { return Dual(td.data[CH * i + c], 0.f); }, td.w, td.h, Dual(tu, 0.f), Dual(tv, 0.f), true, o, uv_xf_d(td.xf, e, true));
        float g = 0.f;
        for (int c = 0; c < CH; ++c) g += ob[c] * o[c].d;
        if (g != 0.f && finite_(g)) atomicAdd(&acc4[j], g);
    }
}
// the environment map's radiance lookup at (u, w) (envmap.cpp:47-56, envmap mode of the bitmap)
PSDR_DEV void env_xf_adjoint(const EnvDev &E, float u, float w, const float *ob, float *acc4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float e[4] = {j == 0 ? 1.f : 0.f, j == 1 ? 1.f : 0.f, j == 2 ? 1.f : 0.f, j == 3 ? 1.f : 0.f};
        Dual o[3];
        env::bitmap_eval_fn<Dual>([&](int i, int c) { return Dual(E.radiance[3 * i + c], 0.f); }, E.width, E.height, Dual(u, 0.f), Dual(w, 0.f), o, uv_xf_d(E.xf, e, true));
        const float g = ob[0] * o[0].d + ob[1] * o[1].d + ob[2] * o[2].d;
        if (g != 0.f && finite_(g)) atomicAdd(&acc4[j], g);
    }
}
template <int LDS> PSDR_DEV env::UvXf<Dual> tex_xf_d(const SceneView<LDS> &S, const TexDev &td) { return uv_xf_d(td.xf, td.d_xf, S.mode == 0 && S.probe_kind == 0); }
template <int LDS> PSDR_DEV env::UvXf<Dual> env_xf_d(const SceneView<LDS> &S, const EnvDev &E) { return uv_xf_d(E.xf, E.d_xf, S.mode == 0 && S.probe_kind == 0); }
PSDR_DEV float env_atan2(float y, float x) { return env::atan2_f(y, x); }
PSDR_DEV Dual env_atan2(const Dual &y, const Dual &x) { return Dual(env::atan2_f(y.v, x.v), fmaf(x.v, y.d, -(y.v * x.d)) / fmaf(x.v, x.v, y.v * y.v)); }
PSDR_DEV float env_safe_acos(float x) { return env::safe_acos_f(x); }
PSDR_DEV Dual env_safe_acos(const Dual &x) {
    const float c = fminf(fmaxf(x.v, -1.f), 1.f);
    return Dual(env::acos_f(c), -x.d / sqrtf(fmaf(-c, c, 1.f)));
}
PSDR_DEV Mat4<Dual> promote(const Mat4<float> &M) { Mat4<Dual> r; for (int i = 0; i < 16; ++i) r.m[i] = Dual(M.m[i]); return r; }
PSDR_DEV float env_floor(float a) { return floorf(a); }
PSDR_DEV Dual env_floor(const Dual &a) { return Dual(floorf(a.v), 0.f); }

// EnvironmentMap::eval_direction, envmap.cpp:59-77.  m_radiance (texels), m_scale and m_from_world are differentiable
// (envmap.h:40-45): forward mode carries their tangents; in reverse mode the lookup is noted / probed like a BSDF bitmap
// (scene_dev.h: id kEnvLookup), which yields the texel and scale adjoints.
// The environment-map branches exist only in the LDS=false instantiations: a scene with an environment map is never
// staged into LDS (api.hip), so the small-scene kernels (all Cornell boxes) carry none of this code or its registers.
template <bool AD, int LDS> PSDR_DEV VecN<AD> env_eval_direction(const SceneView<LDS> &S, const EnvDev &E, const VecN<AD> &wi) {
    using R = Num<AD>;
    VecN<AD> v;
    if constexpr (AD) {
        const bool tan = S.mode == 0;                              // probes put their own unit tangents (kind 7: one entry of from_world)
        Mat4<Dual> M;
#pragma unroll
        for (int i = 0; i < 16; ++i) M.m[i] = Dual(E.from_world.m[i], tan ? E.d_from_world.m[i] : ((S.probe_kind == 7 && S.probe_comp == i) ? 1.f : 0.f));
        v = xform_dir(M, wi);
    } else {
        v = xform_dir(E.from_world, wi);
    }
    R u = env_atan2(v.x, -v.z) * R(env::kInvTwoPi), w = env_safe_acos(v.y) * R(env::kInvPi);
    u = u - env_floor(u); w = w - env_floor(w);
    R rgb[3];
    if constexpr (AD) {
        const bool tt = S.mode == 0 && E.d_radiance != nullptr;
        env::bitmap_eval_fn<Dual>([&](int i, int c) { return Dual(E.radiance[3 * i + c], tt ? E.d_radiance[3 * i + c] : 0.f); }, E.width, E.height, u, w, rgb, env_xf_d(S, E));
        S.note_lookup(kEnvLookup, u.v, w.v);
        const int hot = S.lookup_hot(kEnvLookup, u.v, w.v, 0, 3);
        if (hot >= 0) rgb[hot].d += 1.f;
        return VecN<AD>(rgb[0], rgb[1], rgb[2]) * Dual(E.scale, S.mode == 0 ? E.d_scale : 0.f);
    } else {
        env::bitmap_eval<R>(E.radiance, E.width, E.height, u, w, rgb, env::UvXf<float>(E.xf));
        return VecN<AD>(rgb[0], rgb[1], rgb[2]) * R(E.scale);
    }
}

// EnvironmentMap::__sample_position_pdf, envmap.cpp:146-166
PSDR_DEV float env_position_pdf(const EnvDev &E, const Vec3f &ref_p, const Vec3f &p, const Vec3f &n) {
    Vec3f d = p - ref_p;
    const float dist2 = squared_norm(d);
    { const float len = __builtin_sqrtf(dist2 > 0.f ? dist2 : 0.f); d = Vec3f(d.x / len, d.y / len, d.z / len); }      // (correctly rounded: psdr_hip_env_pdf is compared bit for bit)
    const float G = fabsf(dot(d, n)) / dist2;
    d = xform_dir(E.from_world, d);
    const float factor = G * (1.f / sqrtf(fmaxf(fma_(d.x, d.x, d.z * d.z), kEpsilon * kEpsilon))) * (.5f / (kPi * kPi));
    float u = env::atan2_f(d.x, -d.z) * env::kInvTwoPi;
    float v = env::safe_acos_f(d.y) * env::kInvPi;
    u -= floorf(u); v -= floorf(v);
    const int ix = (int) floorf(u * (float) E.reso0), iy = (int) floorf(v * (float) E.reso1);
    if (!(ix >= 0 && ix < E.reso0 && iy >= 0 && iy < E.reso1)) return 0.f;
    return (E.cell_pmf[ix * E.reso1 + iy] / E.cell_sum) * (float) E.num_cells * factor;
}

// EnvironmentMap::__sample_position (envmap.cpp:91-132): a direction from the cell grid (HyperCubeDistribution2f,
// cube_distrb.cpp:42-49), carried to the scene box (utils.h:145-164); everything detached
PSDR_DEV void env_sample_position(const EnvDev &E, const Vec3f &ref_p, float sx, float sy, Vec3f &p_out, Vec3f &n_out, float &pdf_out) {
    float pdf;
    const int idx = sample_reuse_guided<true>(E.cell_guide, E.guide_n, E.num_cells, E.cell_sum, [&](int i) { return E.cell_pmf[i]; }, [&](int i) { return E.cell_cmf[i]; }, sy, pdf);
    const int cx = idx / E.reso1, cy = idx - cx * E.reso1;
    sx = (sx + (float) cx) * (1.f / (float) E.reso0);
    sy = (sy + (float) cy) * (1.f / (float) E.reso1);
    pdf *= (float) E.num_cells;
    const float theta = sy * kPi, phi = sx * env::kTwoPi;
    float st, ct, sp, cp;
    env::sincos_f(theta, st, ct);
    env::sincos_f(phi, sp, cp);
    const Vec3f d0(cp * st, sp * st, ct);                                      // sphdir, utils.h:56-61
    Vec3f d(d0.y, d0.z, -d0.x);
    const float inv_sin_theta = 1.f / sqrtf(fmaxf(fma_(d.x, d.x, d.z * d.z), kEpsilon * kEpsilon));
    if (pdf > kEpsilon) pdf *= inv_sin_theta * (.5f / (kPi * kPi));
    d = xform_dir(E.to_world, d);
    const float o3[3] = {ref_p.x, ref_p.y, ref_p.z}, d3[3] = {d.x, d.y, d.z};
    float t, G, n3[3];
    env::scene_aabb_exit(o3, d3, E.lower, E.upper, t, n3, G);
    p_out = Vec3f(fma_(d.x, t, ref_p.x), fma_(d.y, t, ref_p.y), fma_(d.z, t, ref_p.z));
    n_out = Vec3f(n3[0], n3[1], n3[2]);
    pdf_out = pdf * G;
}

// ---------------------------------------------------------------- emitters
// Intersection::Le -> AreaLight::eval, reference intersection.h:35-42, area.cpp:17-26
template <bool AD, int LDS> PSDR_DEV VecN<AD> eval_Le(const SceneView<LDS> &S, const Its<AD> &its, bool active) {
    using V = VecN<AD>;
    if (!active || !its.valid) return V(Num<AD>(0.f));
    const int e = mesh_emitter(S, its.mesh);
    if (e < 0) return V(Num<AD>(0.f));
    if constexpr (has_env(LDS)) {
        if (e == S.T->env_emitter) {           // EnvironmentMap::eval, envmap.cpp:47-56
            V wi_world;
            if constexpr (AD) wi_world = to_world_d(its, its.wi); else wi_world = to_world<false>(its, its.wi);
            return env_eval_direction<AD, LDS>(S, S.T->env, -wi_world);
        }
    }
    if (!(detach(its.wi.z) > 0.f)) return V(Num<AD>(0.f));
    const int w = S.T->emit_off + 2 * e;
    const float4 a = S.ld(w);
    if constexpr (AD) { const float4 b = S.rgb_tan(w + 1, 3, e); return make_dual(Vec3f(a.x, a.y, a.z), Vec3f(b.x, b.y, b.z)); }
    else return Vec3f(a.x, a.y, a.z);
}

template <bool AD> struct PositionSample { VecN<AD> p, n; Num<AD> J; float pdf; int slot; float ba, bb; };   // (ba, bb): barycentrics of p on triangle `slot`

// Scene::sample_emitter_position -> AreaLight::sample_position -> Mesh::__sample_position
// reference scene.cpp:987-1013, mesh.cpp:413-454, warp.h:79-82
// EXACT: correctly rounded pdfs - the secondary-edge term's values fill the guiding grid, whose running sums are SAMPLED from: there a
// weight becomes geometry (edges.h)
template <bool AD, int LDS, bool EXACT = false> PSDR_DEV PositionSample<AD> sample_emitter_position(const SceneView<LDS> &S, const Vec3f &ref_p, float sx, float sy) {
    const SceneTables &T = *S.T;
    float epdf = 1.f;
    int ei = 0;
    if (T.n_emitters > 1) {
        ei = sample_reuse<EXACT>(T.n_emitters, T.emitter_sum,
                          [&](int i) { return S.ldf(T.ecdf_off, i); },
                          [&](int i) { return S.ldf(T.ecdf_off, T.n_emitters + i); }, sy, epdf);
    }
    if (has_env(LDS) && ei == T.env_emitter) {
        Vec3f p, nn;
        float pdf_env;
        env_sample_position(T.env, ref_p, sx, sy, p, nn, pdf_env);
        PositionSample<AD> r;
        if constexpr (AD) { r.p = promote(p); r.n = promote(nn); }
        else { r.p = p; r.n = nn; }
        r.J = Num<AD>(1.f);
        r.pdf = pdf_env * epdf;
        r.slot = -1; r.ba = 0.f; r.bb = 0.f;
        return r;
    }
    const int mesh = __float_as_int(S.ld(T.emit_off + 2 * ei + 1).w);
    const MeshRec m = load_mesh(S, mesh);
    float fpdf;
    const int fi = sample_reuse<EXACT>(m.n_faces, m.distrb_sum,
                                [&](int i) { return S.ldf(T.fcdf_off, m.distrb_offset + i); },
                                [&](int i) { return S.ldf(T.fcdf_off, T.n_fcdf + m.distrb_offset + i); }, sx, fpdf);
    const float tt = safe_sqrt(1.f - sx);
    const float a = 1.f - tt, b = tt * sy;
    const int slot = S.ldi(T.map_off, m.face_offset + fi);
    PositionSample<AD> r;
    VecN<AD> p0, e1, e2;
    load_geom<AD, LDS>(S, slot, p0, e1, e2);
    r.p = madd3(e1, Num<AD>(a), e2, Num<AD>(b), p0);
    const float4 s0 = S.ld(T.shade_off + 6 * slot), s3 = S.ld(T.shade_off + 6 * slot + 3);
    r.J = Num<AD>(1.f);
    if constexpr (AD) {
        if (S.tan_on()) {
            const float4 te = S.tanw(slot, 4), tf = S.tanw(slot, 5);
            r.n = make_dual(Vec3f(s3.x, s3.y, s3.z), Vec3f(te.z, te.w, tf.x));
            r.J = Dual(s0.w, tf.y) / Dual(s0.w);
        } else r.n = promote(Vec3f(s3.x, s3.y, s3.z));
    } else r.n = Vec3f(s3.x, s3.y, s3.z);
    r.pdf = m.inv_total_area * epdf;
    r.slot = slot; r.ba = a; r.bb = b;
    return r;
}

// Scene::emitter_position_pdf, reference scene.cpp:1016-1024 -> area.cpp:48-59 -> mesh.cpp:457-466
template <bool AD, int LDS> PSDR_DEV float emitter_position_pdf(const SceneView<LDS> &S, const Vec3f &ref_p, const Its<AD> &its) {
    if (!its.valid) return 0.f;
    const MeshRec m = load_mesh(S, its.mesh);
    if (m.emitter < 0) return 0.f;
    if (has_env(LDS) && m.emitter == S.T->env_emitter) return env_position_pdf(S.T->env, ref_p, detach(its.p), detach(its.n));
    return S.ld(S.T->emit_off + 2 * m.emitter).w * m.inv_total_area;
}

// ---------------------------------------------------------------- Diffuse BSDF, reference src/bsdf/diffuse.cpp:24-108
// MicrofacetPerVertex::__interpolate<1> of the roughness, detached (microfacet_pv.cpp:89,127)
template <bool AD, int LDS> PSDR_DEV float pv_roughness(const SceneView<LDS> &S, int id, const Its<AD> &its) {
    const PvDev pv = S.T->pv[id];
    const int *fi = S.T->tri_fi + 3 * its.slot;
    const float v0 = pv.rough[fi[0]], v1 = pv.rough[fi[1]], v2 = pv.rough[fi[2]];
    return fma_(v1 - v0, detach(its.bu), fma_(v2 - v0, detach(its.bv), v0));
}
// Microfacet::m_roughness as a bitmap, detached (microfacet.cpp:88,117)
template <bool AD, int LDS> PSDR_DEV float roughness_lookup(const SceneView<LDS> &S, int id, const Its<AD> &its) {
    const TexDev td = S.T->tex[3 * id + 2];
    float o[1];
    env::bitmap_eval_tex<float, 1>([&](int i, int) { return td.data[i]; }, td.w, td.h, detach(its.tu), detach(its.tv), true, o, env::UvXf<float>(td.xf));
    return o[0];
}
// bitmap parameter `slot` (0..2) of BSDF `id` at its.uv as (value, tangent): forward texel tangents in a render, the probe's unit
// tangent in a reverse-mode replay (components 3*slot .. of the lookup record)
template <int CH, bool AD, int LDS> PSDR_DEV void param_lookup(const SceneView<LDS> &S, int id, int slot, const Its<AD> &its, Dual *out) {
    const TexDev td = S.T->tex[3 * id + slot];
    const bool tt = AD && S.mode == 0 && td.d_data != nullptr;
    const Dual tu = Dual(its.tu), tv = Dual(its.tv);
    env::bitmap_eval_tex<Dual, CH>([&](int i, int c) { return Dual(td.data[CH * i + c], tt ? td.d_data[CH * i + c] : 0.f); }, td.w, td.h, tu, tv, true, out, tex_xf_d(S, td));
    if constexpr (AD) {
        S.note_lookup(id, tu.v, tv.v);
        const int hot = S.lookup_hot(id, tu.v, tv.v, 3 * slot, CH);
        if (hot >= 0) out[hot].d += 1.f;
    }
}

// (bid_, wi_) are explicit so that NormalMap can evaluate its nested BSDF with a perturbed incident direction
template <bool AD, int LDS> PSDR_DEV VecN<AD> bsdf_eval_id(const SceneView<LDS> &S, int bid_, const Its<AD> &its, const VecN<AD> &wi_, VecN<AD> wo, bool active) {
    using R = Num<AD>; using V = VecN<AD>;
    if (bid_ < 0) return V(R(0.f));          // the envmap's bounding cube has no BSDF (null vcall = zeros)
    const int w = S.T->bsdf_off + 2 * bid_;
    const float4 a = S.ld(w);
    if constexpr (has_mat(LDS)) {
        if (__float_as_int(a.w) & 4) {         // Microfacet (microfacet.cpp); its diffuse reflectance is the record's colour
            const int id = bid_;
            const MatDev md = S.T->mat[id];
            const int fl = __float_as_int(a.w);
            // bitmap parameters (microfacet.cpp:38-45): looked up with (value, tangent) texels and uv, detached in C mode
            const bool tan = AD && S.mode == 0;            // (reverse mode returns no adjoint for the constant specular / roughness)
            const Dual tu = Dual(its.tu), tv = Dual(its.tv);
            // constants: g_mat row = [specular rgb, roughness]
            Vec3d spec(Dual(md.specular[0], AD ? S.mat_tan(id, 0, md.d_specular[0]) : 0.f), Dual(md.specular[1], AD ? S.mat_tan(id, 1, md.d_specular[1]) : 0.f),
                       Dual(md.specular[2], AD ? S.mat_tan(id, 2, md.d_specular[2]) : 0.f));
            Dual rough(md.roughness, AD ? S.mat_tan(id, 3, md.d_roughness) : 0.f);
            Vec3d diff;
            if constexpr (AD) { const float4 b = S.rgb_tan(w + 1, 2, id); diff = make_dual(Vec3f(a.x, a.y, a.z), Vec3f(b.x, b.y, b.z)); }
            else diff = make_dual(Vec3f(a.x, a.y, a.z), Vec3f(0.f, 0.f, 0.f));
            if (fl & (2 | 32 | 64)) {
                auto look = [&](int slot, auto ch, Dual *out) {
                    const TexDev td = S.T->tex[3 * id + slot];
                    constexpr int CH = decltype(ch)::value;
                    const bool tt = tan && td.d_data != nullptr;
                    env::bitmap_eval_tex<Dual, CH>([&](int i, int c) { return Dual(td.data[CH * i + c], tt ? td.d_data[CH * i + c] : 0.f); }, td.w, td.h, tu, tv, true, out, tex_xf_d(S, td));
                    if constexpr (AD) {                    // reverse mode: note the lookup / carry the probe's unit tangent (scene_dev.h)
                        S.note_lookup(id, tu.v, tv.v);
                        const int hot = S.lookup_hot(id, tu.v, tv.v, 3 * slot, CH);
                        if (hot >= 0) out[hot].d += 1.f;
                    }
                };
                Dual o[3];
                if (fl & 2) { look(0, std::integral_constant<int, 3>(), o); diff = Vec3d(o[0], o[1], o[2]); }
                if (fl & 32) { look(1, std::integral_constant<int, 3>(), o); spec = Vec3d(o[0], o[1], o[2]); }
                if (fl & 64) { look(2, std::integral_constant<int, 1>(), o); rough = o[0]; }
            }
            if constexpr (AD) return microfacet_eval<Dual>(spec, diff, rough, (fl & 1) != 0, wi_, wo, active);
            else return microfacet_eval<float>(detach(spec), detach(diff), rough.v, (fl & 1) != 0, wi_, wo, active);
        }
        if (__float_as_int(a.w) & 128) {       // MicrofacetPerVertex (microfacet_pv.cpp)
            const PvDev pv = S.T->pv[bid_];
            const int *fi = S.T->tri_fi + 3 * its.slot;
            const bool tan = AD && S.mode == 0;            // (reverse mode: the adjoints of the per-vertex values come from the lookup probes)
            const Dual bu = Dual(its.bu), bv = Dual(its.bv);
            auto lerp = [&](const float *val, const float *dval, int stride, int c) {
                auto at = [&](int i) { return Dual(val[stride * i + c], (tan && dval) ? dval[stride * i + c] : 0.f); };
                const Dual v0 = at(fi[0]), v1 = at(fi[1]), v2 = at(fi[2]);
                return fma_(v1 - v0, bu, fma_(v2 - v0, bv, v0));
            };
            Vec3d spec(lerp(pv.spec, pv.d_spec, 3, 0), lerp(pv.spec, pv.d_spec, 3, 1), lerp(pv.spec, pv.d_spec, 3, 2));
            Vec3d diff(lerp(pv.diff, pv.d_diff, 3, 0), lerp(pv.diff, pv.d_diff, 3, 1), lerp(pv.diff, pv.d_diff, 3, 2));
            Dual rough = lerp(pv.rough, pv.d_rough, 1, 0);
            if constexpr (AD) {                    // reverse mode: note the interpolation / carry the probe's unit tangent (components as Microfacet's maps)
                S.note_lookup(kPvLookup - its.slot, bu.v, bv.v);
                const int hot = S.lookup_hot(kPvLookup - its.slot, bu.v, bv.v, 0, 7);
                if (hot >= 0) {
                    Dual &q = hot == 0 ? diff.x : hot == 1 ? diff.y : hot == 2 ? diff.z : hot == 3 ? spec.x : hot == 4 ? spec.y : hot == 5 ? spec.z : rough;
                    q.d += 1.f;
                }
            }
            const bool two = (__float_as_int(a.w) & 1) != 0;
            if constexpr (AD) return microfacet_pv_eval<Dual>(spec, diff, rough, two, wi_, wo, active);
            else return microfacet_pv_eval<float>(detach(spec), detach(diff), rough.v, two, wi_, wo, active);
        }
        if (__float_as_int(a.w) & 8) {         // RoughConductor (roughconductor.cpp)
            const MatDev md = S.T->mat[bid_];
            const bool two = (__float_as_int(a.w) & 1) != 0;
            // g_mat row = [alpha_u, alpha_v, eta rgb, k rgb, specular_reflectance rgb]; bitmaps: slot 0 eta, 1 k, 2 alpha (both axes,
            // as the XML loader fills them, scene_loader.cpp:334-344)
            const int bid = bid_, fl = __float_as_int(a.w);
            auto mt = [&](int kk, float fwd) { return AD ? S.mat_tan(bid, kk, fwd) : 0.f; };
            Dual au(md.alpha_u, mt(0, md.d_alpha_u)), av(md.alpha_v, mt(1, md.d_alpha_v));
            Vec3d eta(Dual(md.eta[0], mt(2, md.d_eta[0])), Dual(md.eta[1], mt(3, md.d_eta[1])), Dual(md.eta[2], mt(4, md.d_eta[2])));
            Vec3d kk3(Dual(md.k[0], mt(5, md.d_k[0])), Dual(md.k[1], mt(6, md.d_k[1])), Dual(md.k[2], mt(7, md.d_k[2])));
            const Vec3d spec(Dual(md.specular[0], mt(8, md.d_specular[0])), Dual(md.specular[1], mt(9, md.d_specular[1])), Dual(md.specular[2], mt(10, md.d_specular[2])));
            if (fl & (2 | 32 | 64)) {
                Dual o[3];
                if (fl & 2) { param_lookup<3, AD, LDS>(S, bid, 0, its, o); eta = Vec3d(o[0], o[1], o[2]); }
                if (fl & 32) { param_lookup<3, AD, LDS>(S, bid, 1, its, o); kk3 = Vec3d(o[0], o[1], o[2]); }
                if (fl & 64) { param_lookup<1, AD, LDS>(S, bid, 2, its, o); au = o[0]; av = o[0]; }
            }
            if constexpr (AD) return conductor_eval<Dual>(au, av, eta, kk3, spec, two, wi_, wo, active);
            else return conductor_eval<float>(au.v, av.v, detach(eta), detach(kk3), detach(spec), two, wi_, wo, active);
        }
    }
    if constexpr (has_mat(LDS)) {
        if (__float_as_int(a.w) & 16) {        // RoughDielectric (roughdielectric.cpp): eta[0] = intIOR/extIOR, eta[1] = extIOR/intIOR
            const MatDev md = S.T->mat[bid_];
            const bool two = (__float_as_int(a.w) & 1) != 0;
            // g_mat row = [alpha_u, alpha_v, eta]; m_inv_eta = 1 / m_eta moves with eta; bitmap: slot 2 alpha (both axes)
            const int bid = bid_;
            const float de = AD ? S.mat_tan(bid, 2, md.d_eta[0]) : 0.f;
            const float dinv = (!AD || S.mode == 0) ? (AD ? md.d_eta[1] : 0.f) : -de / (md.eta[0] * md.eta[0]);
            Dual au(md.alpha_u, AD ? S.mat_tan(bid, 0, md.d_alpha_u) : 0.f), av(md.alpha_v, AD ? S.mat_tan(bid, 1, md.d_alpha_v) : 0.f);
            if (__float_as_int(a.w) & 64) { Dual o[1]; param_lookup<1, AD, LDS>(S, bid, 2, its, o); au = o[0]; av = o[0]; }
            if constexpr (AD) return dielectric_eval<Dual>(au, av, Dual(md.eta[0], de), Dual(md.eta[1], dinv), two, wi_, wo, active);
            else return dielectric_eval<float>(au.v, av.v, md.eta[0], md.eta[1], two, wi_, wo, active);
        }
    }
    R wiz = wi_.z;
    if (__float_as_int(a.w) & 1) { wo.z = mulsign(wo.z, detach(wiz)); wiz = abs_(wiz); }
    if (!(active && detach(wiz) > 0.f && detach(wo.z) > 0.f)) return V(R(0.f));
    V refl;
    bool textured = false;
    if constexpr (has_mat(LDS)) {
        if (__float_as_int(a.w) & 2) {         // Bitmap3fD reflectance, diffuse.cpp:38 -> bitmap.cpp:47-128 (flip_v)
            textured = true;
            const TexDev td = S.T->tex[3 * bid_];
            R rgb[3];
            if constexpr (AD) {
                const bool tan = td.d_data != nullptr && S.mode == 0;      // (reverse mode: the texel adjoints come from the lookup probes)
                env::bitmap_eval_tex<Dual>([&](int i, int c) { return Dual(td.data[3 * i + c], tan ? td.d_data[3 * i + c] : 0.f); },
                                           td.w, td.h, its.tu, its.tv, true, rgb, tex_xf_d(S, td));
                S.note_lookup(bid_, its.tu.v, its.tv.v);
                const int hot = S.lookup_hot(bid_, its.tu.v, its.tv.v, 0, 3);
                if (hot >= 0) rgb[hot].d += 1.f;
            } else {
                env::bitmap_eval_tex<float>([&](int i, int c) { return td.data[3 * i + c]; }, td.w, td.h, its.tu, its.tv, true, rgb, env::UvXf<float>(td.xf));
            }
            refl = V(rgb[0], rgb[1], rgb[2]);
        }
    }
    if (!textured) {
        if constexpr (AD) { const float4 b = S.rgb_tan(w + 1, 2, bid_); refl = make_dual(Vec3f(a.x, a.y, a.z), Vec3f(b.x, b.y, b.z)); }
        else refl = Vec3f(a.x, a.y, a.z);
    }
    return refl * R(kInvPi) * wo.z;
}
template <bool AD, int LDS> PSDR_DEV float bsdf_pdf_id(const SceneView<LDS> &S, int bid_, const Its<AD> &its, const VecN<AD> &wi_, const VecN<AD> &wo, bool active) {
    if (bid_ < 0) return 0.f;
    const float4 a = S.ld(S.T->bsdf_off + 2 * bid_);
    if constexpr (has_mat(LDS)) {
        if (__float_as_int(a.w) & 128) {
            const float r = pv_roughness(S, bid_, its);
            return ggx_pdf(sqr(r), sqr(r), (__float_as_int(a.w) & 1) != 0, detach(wi_), detach(wo), active);
        }
        if (__float_as_int(a.w) & 12) {
            const MatDev md = S.T->mat[bid_];
            const bool mf = (__float_as_int(a.w) & 4) != 0;
            const float rough = (__float_as_int(a.w) & 64) ? roughness_lookup(S, bid_, its) : md.roughness;      // (conductor: the alpha map)
            if (!mf && (__float_as_int(a.w) & 64)) return ggx_pdf(rough, rough, (__float_as_int(a.w) & 1) != 0, detach(wi_), detach(wo), active);
            return ggx_pdf(mf ? sqr(rough) : md.alpha_u, mf ? sqr(rough) : md.alpha_v, (__float_as_int(a.w) & 1) != 0, detach(wi_), detach(wo), active);
        }
        if (__float_as_int(a.w) & 16) {
            const MatDev md = S.T->mat[bid_];
            if (__float_as_int(a.w) & 64) { const float al = roughness_lookup(S, bid_, its); return dielectric_pdf(al, al, md.eta[0], md.eta[1], (__float_as_int(a.w) & 1) != 0, detach(wi_), detach(wo), active); }
            return dielectric_pdf(md.alpha_u, md.alpha_v, md.eta[0], md.eta[1], (__float_as_int(a.w) & 1) != 0, detach(wi_), detach(wo), active);
        }
    }
    float wiz = detach(wi_.z), woz = detach(wo.z);
    if (__float_as_int(a.w) & 1) { woz = mulsign(woz, wiz); wiz = fabsf(wiz); }
    return (active && wiz > 0.f && woz > 0.f) ? kInvPi * woz : 0.f;
}
// sin/cos as drjit computes them: the Cephes single-precision kernels (octant reduction with the three-term
// pi/4 split, cubic polynomials in x^2), every multiply-add an explicit fma.  ~20 VALU ops, no libm call,
// and bit-reproducible on any IEEE target.
PSDR_DEV void sincos_cephes(float xx, float &s_out, float &c_out) {
    float x = fabsf(xx);
    int j = (int) (1.27323954473516f * x);
    float y = (float) j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    float sign_s = xx < 0.f ? -1.f : 1.f, sign_c = 1.f;
    if (j > 3) { sign_s = -sign_s; sign_c = -sign_c; j -= 4; }
    if (j > 1) sign_c = -sign_c;
    x = fma_(-y, 0.78515625f, x); x = fma_(-y, 2.4187564849853515625e-4f, x); x = fma_(-y, 3.77489497744594108e-8f, x);
    const float z = x * x;
    const float ps = fma_(fma_(fma_(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, x, x);
    const float pc = fma_(fma_(fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fma_(-0.5f, z, 1.0f));
    const bool swap = (j == 1) || (j == 2);
    s_out = sign_s * (swap ? pc : ps);
    c_out = sign_c * (swap ? ps : pc);
}

// reference include/psdr/core/warp.h:16-63
PSDR_DEV Vec3f square_to_cosine_hemisphere(float sx, float sy) {
    const float x = fma_(2.f, sx, -1.f), y = fma_(2.f, sy, -1.f);
    const bool is_zero = (x == 0.f) && (y == 0.f), q13 = fabsf(x) < fabsf(y);
    const float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = .25f * kPi * rp / r;
    if (q13) phi = .5f * kPi - phi;
    if (is_zero) phi = 0.f;
    float s, c;
    sincos_cephes(phi, s, c);
    const float px = r * c, py = r * s;
    return Vec3f(px, py, safe_sqrt(1.f - fma_(py, py, px * px)));
}
struct BSDFSample { Vec3f wo; float pdf; bool valid; };
template <bool AD, int LDS> PSDR_DEV BSDFSample bsdf_sample_id(const SceneView<LDS> &S, int bid_, const Its<AD> &its, const VecN<AD> &wi_, float s0, float s1, float s2, bool active) {
    (void) s0;
    if (bid_ < 0) { BSDFSample z; z.wo = Vec3f(0.f, 0.f, 0.f); z.pdf = 0.f; z.valid = false; return z; }
    const float4 a = S.ld(S.T->bsdf_off + 2 * bid_);
    if constexpr (has_mat(LDS)) {
        if (__float_as_int(a.w) & 128) {       // MicrofacetPerVertex::__sample (microfacet_pv.cpp:80-103)
            BSDFSample m;
            const float r = pv_roughness(S, bid_, its);
            ggx_reflect_sample(sqr(r), sqr(r), (__float_as_int(a.w) & 1) != 0, detach(wi_), s0, s1, active, m.wo, m.pdf, m.valid);
            return m;
        }
        if (__float_as_int(a.w) & 12) {        // Microfacet / RoughConductor ::sample use the first two numbers (microfacet.cpp:88)
            BSDFSample m;
            const MatDev md = S.T->mat[bid_];
            const bool mf = (__float_as_int(a.w) & 4) != 0;
            const float rough = (__float_as_int(a.w) & 64) ? roughness_lookup(S, bid_, its) : md.roughness;
            const bool amap = !mf && (__float_as_int(a.w) & 64);
            ggx_reflect_sample(mf ? sqr(rough) : (amap ? rough : md.alpha_u), mf ? sqr(rough) : (amap ? rough : md.alpha_v), (__float_as_int(a.w) & 1) != 0, detach(wi_), s0, s1, active,
                               m.wo, m.pdf, m.valid);
            return m;
        }
        if (__float_as_int(a.w) & 16) {        // RoughDielectric::sample: the third number picks reflection or refraction
            BSDFSample m;
            const MatDev md = S.T->mat[bid_];
            const float al_u = (__float_as_int(a.w) & 64) ? roughness_lookup(S, bid_, its) : md.alpha_u, al_v = (__float_as_int(a.w) & 64) ? al_u : md.alpha_v;
            dielectric_sample(al_u, al_v, md.eta[0], (__float_as_int(a.w) & 1) != 0, detach(wi_), s0, s1, s2, active, m.wo, m.pdf, m.valid);
            return m;
        }
    }
    float wiz = detach(wi_.z);
    if (__float_as_int(a.w) & 1) wiz = fabsf(wiz);
    BSDFSample bs;
    bs.wo = square_to_cosine_hemisphere(s1, s2);
    bs.pdf = kInvPi * bs.wo.z;
    bs.valid = active && (wiz > 0.f);
    return bs;
}

// ---------------------------------------------------------------- NormalMap, reference src/bsdf/normalmap.cpp:20-187
// wp = the normal of the map in the shading frame; the "tangent facet" wt closes the microsurface (the i -> p -> t -> o path is
// commented out in the reference).  Scene class 0 only.
template <typename R> PSDR_DEV Vec3<R> nm_wt(const Vec3<R> &wp) { return normalize(Vec3<R>(-wp.x, -wp.y, R(0.f))); }
template <typename R> PSDR_DEV R nm_pdot(const Vec3<R> &a, const Vec3<R> &b) { const R d = dot(a, b); return detach(d) > 0.f ? d : R(0.f); }
template <typename R> PSDR_DEV R nm_sin_theta(const Vec3<R> &v) { return safe_sqrt(fma_(v.x, v.x, v.y * v.y)); }
template <typename R> PSDR_DEV R nm_G1(const Vec3<R> &wp, const Vec3<R> &w) {
    const R cw = detach(w.z) > 0.f ? w.z : R(0.f), cp = detach(wp.z) > 0.f ? wp.z : R(0.f);
    const R g = cw * cp / (nm_pdot(w, wp) + nm_pdot(w, nm_wt(wp)) * nm_sin_theta(wp));
    return detach(g) < 1.f ? g : R(1.f);
}
template <typename R> PSDR_DEV R nm_lambda_p(const Vec3<R> &wp, const Vec3<R> &wi) {
    const R i_dot_p = nm_pdot(wp, wi);
    return i_dot_p / (i_dot_p + nm_pdot(nm_wt(wp), wi) * nm_sin_theta(wp));
}
template <typename R> struct NmFrame {          // Frame(n, s): t = normalize(n x s), s = normalize(t x n)  (frame.h:43-46)
    Vec3<R> s, t, n;
    PSDR_DEV NmFrame(const Vec3<R> &n_, const Vec3<R> &s_) : n(n_) { t = normalize(cross(n_, s_)); s = normalize(cross(t, n_)); }
    PSDR_DEV Vec3<R> to_local(const Vec3<R> &v) const { return Vec3<R>(dot(v, s), dot(v, t), dot(v, n)); }
    PSDR_DEV Vec3<R> to_world(const Vec3<R> &v) const { return s * v.x + t * v.y + n * v.z; }
};
// m_nmap.eval(its.uv) (constant or bitmap, slot 0 of the NormalMap's own record) -> wp and the perturbed frame
template <bool AD, int LDS> PSDR_DEV NmFrame<Num<AD>> nm_frame(const SceneView<LDS> &S, int bid, const Its<AD> &its, VecN<AD> &wp) {
    using R = Num<AD>; using V = VecN<AD>;
    const int w = S.T->bsdf_off + 2 * bid;
    const float4 a = S.ld(w);
    V c;
    if (__float_as_int(a.w) & 2) {
        const TexDev td = S.T->tex[3 * bid];
        R rgb[3];
        if constexpr (AD) {
            const bool tan = td.d_data != nullptr && S.mode == 0;
            env::bitmap_eval_tex<Dual>([&](int i, int ch) { return Dual(td.data[3 * i + ch], tan ? td.d_data[3 * i + ch] : 0.f); }, td.w, td.h, its.tu, its.tv, true, rgb, tex_xf_d(S, td));
            S.note_lookup(bid, its.tu.v, its.tv.v);
            const int hot = S.lookup_hot(bid, its.tu.v, its.tv.v, 0, 3);
            if (hot >= 0) rgb[hot].d += 1.f;
        } else {
            env::bitmap_eval_tex<float>([&](int i, int ch) { return td.data[3 * i + ch]; }, td.w, td.h, its.tu, its.tv, true, rgb, env::UvXf<float>(td.xf));
        }
        c = V(rgb[0], rgb[1], rgb[2]);
    } else {
        if constexpr (AD) { const float4 b = S.rgb_tan(w + 1, 2, bid); c = make_dual(Vec3f(a.x, a.y, a.z), Vec3f(b.x, b.y, b.z)); }
        else c = Vec3f(a.x, a.y, a.z);
    }
    wp = normalize(V(fma_(c.x, R(2.f), R(-1.f)), fma_(c.y, R(2.f), R(-1.f)), fma_(c.z, R(2.f), R(-1.f))));
    const V s = normalize(its.dp_du - wp * dot(wp, its.dp_du));
    return NmFrame<R>(wp, s);
}
template <bool AD, int LDS> PSDR_DEV VecN<AD> normalmap_eval(const SceneView<LDS> &S, int bid, const Its<AD> &its, VecN<AD> wo, bool active) {
    using R = Num<AD>; using V = VecN<AD>;
    const float4 a = S.ld(S.T->bsdf_off + 2 * bid);
    const int nested = __float_as_int(S.ld(S.T->bsdf_off + 2 * bid + 1).w);
    V wi = its.wi;
    if (__float_as_int(a.w) & 1) { wo.z = mulsign(wo.z, detach(wi.z)); wi.z = abs_(wi.z); }
    active = active && detach(wi.z) > 0.f && detach(wo.z) > 0.f;
    V wp;
    const NmFrame<R> pf = nm_frame<AD, LDS>(S, bid, its, wp);
    const V pwi = pf.to_local(wi), pwo = pf.to_local(wo);
    const R shadowing = nm_G1(wp, wo), lambda_p = nm_lambda_p(wp, wi);
    const V wt = nm_wt(wp);
    V value = bsdf_eval_id<AD, LDS>(S, nested, its, pwi, pwo, active) * lambda_p * shadowing;
    if (detach(dot(wi, wt)) > 0.f) {
        const V wi_r = normalize(wi - wt * (R(2.0f) * dot(wi, wt)));
        value = value + bsdf_eval_id<AD, LDS>(S, nested, its, pf.to_local(wi_r), pwo, active) * (R(1.f) - lambda_p) * shadowing;
    }
    return active ? value : V(R(0.f));
}
template <bool AD, int LDS> PSDR_DEV float normalmap_pdf(const SceneView<LDS> &S, int bid, const Its<AD> &its, const VecN<AD> &wo_, bool active) {
    using R = Num<AD>; using V = VecN<AD>;
    const float4 a = S.ld(S.T->bsdf_off + 2 * bid);
    const int nested = __float_as_int(S.ld(S.T->bsdf_off + 2 * bid + 1).w);
    V wi = its.wi, wo = wo_;
    if (__float_as_int(a.w) & 1) { wo.z = mulsign(wo.z, detach(wi.z)); wi.z = abs_(wi.z); }
    active = active && detach(wi.z) > 0.f && detach(wo.z) > 0.f;
    V wp;
    const NmFrame<R> pf = nm_frame<AD, LDS>(S, bid, its, wp);
    const V pwo = pf.to_local(wo);
    const float prob = detach(nm_lambda_p(wp, wi));
    const V wt = nm_wt(wp);
    const V wi_r = normalize(wi - wt * (R(2.0f) * dot(wi, wt)));
    const float value = prob * bsdf_pdf_id<AD, LDS>(S, nested, its, pf.to_local(wi), pwo, active)
                        + (1.f - prob) * bsdf_pdf_id<AD, LDS>(S, nested, its, pf.to_local(wi_r), pwo, active);
    return active ? value : 0.f;
}
template <bool AD, int LDS> PSDR_DEV BSDFSample normalmap_sample(const SceneView<LDS> &S, int bid, const Its<AD> &its, float s0, float s1, float s2, bool active) {
    using R = Num<AD>; using V = VecN<AD>;
    const float4 a = S.ld(S.T->bsdf_off + 2 * bid);
    const int nested = __float_as_int(S.ld(S.T->bsdf_off + 2 * bid + 1).w);
    V wi = its.wi;
    if (__float_as_int(a.w) & 1) wi.z = abs_(wi.z);
    V wp;
    const NmFrame<R> pf = nm_frame<AD, LDS>(S, bid, its, wp);
    const V pwi = pf.to_local(wi);
    const float prob = detach(nm_lambda_p(wp, wi));
    const V wt = nm_wt(wp);
    const bool itpo = s2 >= prob;
    BSDFSample bs = bsdf_sample_id<AD, LDS>(S, nested, its, pwi, s0, s1, s2, active && !itpo);
    const V wi_r = normalize(wi - wt * (R(2.0f) * dot(wi, wt)));
    const V rwi = pf.to_local(wi_r);
    const BSDFSample bs2 = bsdf_sample_id<AD, LDS>(S, nested, its, rwi, s0, s1, s2, active && itpo);
    if (itpo) bs.wo = bs2.wo;
    const V wo_l(R(bs.wo.x), R(bs.wo.y), R(bs.wo.z));
    const float pdf1 = bsdf_pdf_id<AD, LDS>(S, nested, its, pwi, wo_l, active), pdf2 = bsdf_pdf_id<AD, LDS>(S, nested, its, rwi, wo_l, active);
    bs.pdf = prob * pdf1 + (1.f - prob) * pdf2;
    bs.wo = detach(pf.to_world(wo_l));
    bs.valid = active && (bs.valid || bs2.valid);
    return bs;
}

// BSDF of the mesh the intersection lies on (a mesh without BSDF - the envmap's bounding cube - gives zeros, as drjit's null vcall)
template <bool AD, int LDS> PSDR_DEV VecN<AD> bsdf_eval(const SceneView<LDS> &S, const Its<AD> &its, VecN<AD> wo, bool active) {
    const int bid = mesh_bsdf(S, its.mesh);
    if constexpr (has_mat(LDS)) if (bid >= 0 && (__float_as_int(S.ld(S.T->bsdf_off + 2 * bid).w) & 256)) return normalmap_eval<AD, LDS>(S, bid, its, wo, active);
    return bsdf_eval_id<AD, LDS>(S, bid, its, its.wi, wo, active);
}
template <bool AD, int LDS> PSDR_DEV float bsdf_pdf(const SceneView<LDS> &S, const Its<AD> &its, const VecN<AD> &wo, bool active) {
    const int bid = mesh_bsdf(S, its.mesh);
    if constexpr (has_mat(LDS)) if (bid >= 0 && (__float_as_int(S.ld(S.T->bsdf_off + 2 * bid).w) & 256)) return normalmap_pdf<AD, LDS>(S, bid, its, wo, active);
    return bsdf_pdf_id<AD, LDS>(S, bid, its, its.wi, wo, active);
}
template <bool AD, int LDS> PSDR_DEV BSDFSample bsdf_sample(const SceneView<LDS> &S, const Its<AD> &its, float s0, float s1, float s2, bool active) {
    const int bid = mesh_bsdf(S, its.mesh);
    if constexpr (has_mat(LDS)) if (bid >= 0 && (__float_as_int(S.ld(S.T->bsdf_off + 2 * bid).w) & 256)) return normalmap_sample<AD, LDS>(S, bid, its, s0, s1, s2, active);
    return bsdf_sample_id<AD, LDS>(S, bid, its, its.wi, s0, s1, s2, active);
}


// FieldExtractionIntegrator::__Li (reference src/integrator/field.cpp:49-121) and CollocatedIntegrator::__Li
// (src/integrator/collocated.cpp:24-55): a function of the first hit only.  LDS=false instantiations only.
template <bool AD, int LDS> PSDR_DEV VecN<AD> first_hit_value(const SceneView<LDS> &S, const Its<AD> &its) {
    using R = Num<AD>; using V = VecN<AD>;
    bool ok = its.valid;
    if (S.T->env_emitter >= 0 && S.field != 8) ok = ok && mesh_bsdf(S, its.mesh) >= 0;        // field.cpp:55-58
    if (S.field_object >= 0) ok = ok && its.mesh == S.field_object;
    if (!ok) return V(R(0.f));
    switch (S.field) {
        case 0: return V(R(1.f));
        case 1: return its.p;
        case 2: return V(its.t);
        case 3: return its.n;
        case 4: return its.fn;
        case 5: return V(its.tu, its.tv, R(0.f));
        case 6: return bsdf_eval<AD, LDS>(S, its, its.wi, true);
        case 7: return V(R((float) (its.mesh + 1)));
        default: {
            const V r = bsdf_eval<AD, LDS>(S, its, its.wi, true) / sqr(its.t);
            if constexpr (AD) return r * Dual(S.intensity, S.mode == 0 ? S.d_intensity : 0.f); else return r * S.intensity;
        }
    }
}

// reference utils.h:277-281.  The squares of two small pdfs can sum to a denormal, which v_rcp_f32 flushes (-> inf, and a weight that should be
// <= 1 turns into inf / NaN and the sample is dropped): such sums are scaled into the normal range first (2^64 on both sides, exact)
PSDR_DEV float mis_weight(float p1, float p2) {
    const float w1 = p1 * p1, w2 = p2 * p2, den = w1 + w2;
    const float s = den < 7.8886091e-31f ? 18446744073709551616.f : 1.f;          // 2^-100, 2^64
    return fdiv(w1 * s, den * s);
}

// ---------------------------------------------------------------- PathTracer::__Li, reference src/integrator/path.cpp:35-127
// Consumes exactly 5*max_depth draws of `rng` whatever the path does (the reference draws for masked lanes too).
template <bool AD, int LDS, bool COUNT>
PSDR_DEV VecN<AD> Li(SceneView<LDS> &S, LaneRng &rng, const RayT<AD> &ray_in, bool active, int max_depth, bool hide_emitters) {
    using R = Num<AD>; using V = VecN<AD>;
    Its<AD> its = ray_intersect<AD, false, LDS, COUNT>(S, ray_in, active);
    active = active && its.valid;
    V throughput(R(1.f));
    if (S.field >= 0) { if constexpr (has_mat(LDS)) return first_hit_value<AD, LDS>(S, its); else return V(R(0.f)); }
    V result = hide_emitters ? V(R(0.f)) : eval_Le<AD, LDS>(S, its, active);
    // DirectIntegrator(mis) (reference direct.cpp:34-132) is one pass of the same body: mis = 0 draws and uses only the emitter
    // sample (weight 1), mis = 1 only the BSDF sample (weight 1), mis = 2 both with MIS
    const int mis = S.mis;
    const int nd = mis == 0 ? 2 : (mis == 1 ? 3 : 5);
    if (mis >= 0) max_depth = 1;
    for (int depth = 0; depth < max_depth; ++depth) {
        if (!active) { rng.advance((uint64_t) (nd * (max_depth - depth))); break; }
        if (mis != 1) {   // next-event estimation (path.cpp:47-83)
            const float sx = rng.next_1d(), sy = rng.next_1d();
            const bool its_is_emitter = mesh_emitter(S, its.mesh) >= 0;
            if (!its_is_emitter) {
                PositionSample<AD> ps = sample_emitter_position<AD, LDS>(S, detach(its.p), sx, sy);
                S.note_slot(ps.slot);
                V wod = ps.p - its.p;
                const R dist_sqr = squared_norm(wod);
                const R dist = safe_sqrt(dist_sqr);
                wod = wod / dist;
                RayT<AD> ray1; ray1.o = its.p; ray1.d = wod;
                Its<AD> its1 = ray_intersect<AD, AD, LDS, COUNT>(S, ray1, true);
                bool active_direct = its1.valid && (detach(its1.t) > detach(dist) - kShadowEpsilon) && (mesh_emitter(S, its1.mesh) >= 0);
                if (active_direct) {
                    const R cos_val = dot(its1.n, -wod);
                    const R G_val = div_(abs_(cos_val), dist_sqr);
                    const V emitter_val = eval_Le<AD, LDS>(S, its1, true);
                    const V wo_local = to_local<AD>(its, wod);
                    V bsdf_val2 = bsdf_eval<AD, LDS>(S, its, wo_local, true);
                    bsdf_val2 = bsdf_val2 * div_(G_val * ps.J, R(ps.pdf));
                    const float pdf1 = bsdf_pdf<AD, LDS>(S, its, wo_local, true) * detach(G_val);
                    if (pdf1 != 0.f) {
                        const float weight1 = mis == 0 ? 1.f : mis_weight(ps.pdf, pdf1);
                        result = result + throughput * emitter_val * bsdf_val2 * R(weight1);
                    }
                }
            }
        }
        if (mis != 0) {   // BSDF sampling (path.cpp:86-123)
            const float s0 = rng.next_1d(), s1 = rng.next_1d(), s2 = rng.next_1d();
            (void) s0;
            const BSDFSample bs = bsdf_sample<AD, LDS>(S, its, s0, s1, s2, true);
            RayT<AD> curr; curr.o = its.p; curr.d = to_world<AD>(its, bs.wo);
            Its<AD> its1 = ray_intersect<AD, AD, LDS, COUNT>(S, curr, bs.valid);     // (an invalid sample ends the path)
            active = bs.valid && its1.valid;
            if (!active) continue;
            V bsdf_val;
            float pdf0;
            if constexpr (AD) {
                V wo = (its1.p - its.p) / its1.t;
                const R cos_val = dot(its1.n, -wo);
                const R G_val = div_(abs_(cos_val), sqr(its1.t));
                pdf0 = bs.pdf * G_val.v;
                if (its1.t.v < kEpsilon) bsdf_val = V(R(0.f));
                else bsdf_val = bsdf_eval<AD, LDS>(S, its, to_local<AD>(its, wo), true) * div_(G_val * its1.J, R(pdf0));
            } else {
                const float cos_val = dot(its1.n, -curr.d);
                const float G_val = fdiv(fabsf(cos_val), sqr(its1.t));
                pdf0 = bs.pdf * G_val;
                if (its1.t < kEpsilon) bsdf_val = V(0.f);
                else bsdf_val = vdiv_(bsdf_eval<AD, LDS>(S, its, bs.wo, true), bs.pdf);
            }
            const float weight2 = mis == 1 ? 1.f : mis_weight(pdf0, emitter_position_pdf<AD, LDS>(S, detach(its.p), its1));
            throughput = throughput * bsdf_val;
            result = result + eval_Le<AD, LDS>(S, its1, true) * throughput * R(weight2);
            its = its1;
        }
    }
    return result;
}

} // namespace psdr
