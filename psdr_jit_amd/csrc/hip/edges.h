// edges.h — camera sampling and the two boundary-integral estimators of renderD.
//   primary edges   : Integrator::render_primary_edges, reference src/integrator/integrator.cpp:179-198
//                     + PerspectiveCamera::sample_primary_edge, src/sensor/perspective.cpp:200-226
//   secondary edges : PathTracer::eval_secondary_edge / render_secondary_edges, src/integrator/path.cpp:171-294
//                     + Scene::sample_boundary_segment_direct, src/scene/scene.cpp:1027-1068
// Both have zero primal (value - detach(value)); only the tangent is accumulated.
#pragma once
#include "shade.h"

namespace psdr {

// PerspectiveCamera::sample_primary_ray, reference perspective.cpp:160-178 (direction detached in D mode)
template <bool AD> PSDR_DEV RayT<AD> sample_primary_ray(const SensorDev &cam, float sx, float sy) {
    if (cam.ortho) {      // OrthographicCamera::sample_primary_ray, orthographic.cpp:161-181
        const Vec3f near_p = xform_pos(cam.sample_to_camera, Vec3f(sx, sy, 0.f));
        RayT<AD> r;
        if constexpr (AD) {
            Mat4<Dual> M;
#pragma unroll
            for (int i = 0; i < 16; ++i) M.m[i] = Dual(cam.to_world.m[i], cam.d_to_world.m[i]);
            r.o = xform_pos(M, promote(near_p));
            r.d = xform_dir(M, Vec3d(Dual(0.f), Dual(0.f), Dual(1.f)));
        } else {
            r.o = xform_pos(cam.to_world, near_p);
            r.d = xform_dir(cam.to_world, Vec3f(0.f, 0.f, 1.f));
        }
        return r;
    }
    const Vec3f d = normalize(xform_pos(cam.sample_to_camera, Vec3f(sx, sy, 0.f)));
    RayT<AD> r;
    if constexpr (AD) {
        Mat4<Dual> M;
#pragma unroll
        for (int i = 0; i < 16; ++i) M.m[i] = Dual(cam.to_world.m[i], cam.d_to_world.m[i]);
        r.o = xform_pos(M, Vec3d(Dual(0.f)));
        r.d = xform_dir(M, promote(d));
    } else {
        r.o = xform_pos(cam.to_world, Vec3f(0.f));
        r.d = xform_dir(cam.to_world, d);
    }
    return r;
}

// reverse mode w.r.t. the camera pose: the tangent of the primary ray for a unit tangent on entry (r, c) = (comp / 4, comp % 4)
// of rows 0-2 of to_world (origin = to_world . o_cam, direction = to_world . d_cam; all other tangents of `ray` are cleared)
PSDR_DEV void primary_ray_pose_tangent(const SensorDev &cam, float sx, float sy, int comp, RayT<true> &ray) {
    const int r = comp >> 2, c = comp & 3;
    const Vec3f pc = xform_pos(cam.sample_to_camera, Vec3f(sx, sy, 0.f));
    const Vec3f o_cam = cam.ortho ? pc : Vec3f(0.f), d_cam = cam.ortho ? Vec3f(0.f, 0.f, 1.f) : normalize(pc);
    const float od = c == 0 ? o_cam.x : (c == 1 ? o_cam.y : (c == 2 ? o_cam.z : 1.f));
    const float dd = c == 0 ? d_cam.x : (c == 1 ? d_cam.y : (c == 2 ? d_cam.z : 0.f));
    ray.o.x.d = r == 0 ? od : 0.f; ray.o.y.d = r == 1 ? od : 0.f; ray.o.z.d = r == 2 ? od : 0.f;
    ray.d.x.d = r == 0 ? dd : 0.f; ray.d.y.d = r == 1 ? dd : 0.f; ray.d.z.d = r == 2 ? dd : 0.f;
}

struct SensorDirectSample { float qx, qy; int pixel_idx; float sensor_val; bool valid; };
// PerspectiveCamera::sample_direct, reference perspective.cpp:181-197
PSDR_DEV SensorDirectSample sample_direct(const SceneTables &T, const SensorDev &cam, const Vec3f &p) {
    SensorDirectSample r;
    const Vec3f q = xform_pos(cam.world_to_sample, p);
    r.qx = q.x; r.qy = q.y;
    const int ix = (int) floorf(q.x * (float) T.width), iy = (int) floorf(q.y * (float) T.height);
    r.valid = ix >= 0 && ix < T.width && iy >= 0 && iy < T.height;
    r.pixel_idx = r.valid ? iy * T.width + ix : -1;
    Vec3f dir = p - Vec3f(cam.cam_pos[0], cam.cam_pos[1], cam.cam_pos[2]);
    const float dist2 = squared_norm(dir);
    dir = dir / safe_sqrt(dist2);
    const float cosTheta = dot(Vec3f(cam.cam_dir[0], cam.cam_dir[1], cam.cam_dir[2]), dir);
    const float rc = 1.f / cosTheta;
    r.sensor_val = (1.f / dist2) * (rc * rc * rc) * cam.inv_area;
    return r;
}

struct BoundarySegSampleDirect { bool valid; float pdf; Vec3d p0; Vec3f edge, edge2, p2, n; int emitter_slot; int edge_id; float s1; };

PSDR_DEV int sign_eps(float x, float eps) { return x > eps ? 1 : (x < -eps ? -1 : 0); }     // reference utils.h:47-53
PSDR_DEV float sign1(float x) { return signbit_(x) ? -1.f : 1.f; }                          // drjit::sign

// Scene::sample_boundary_segment_direct, reference scene.cpp:1027-1068
template <int LDS> PSDR_DEV BoundarySegSampleDirect sample_boundary_segment_direct(const SceneView<LDS> &S, const SecEdgeTables &E, Vec3f s3) {
    BoundarySegSampleDirect r;
    float sample1 = s3.x, pdf0;
    const int ei = sample_reuse_guided<true>(E.guide, E.guide_n, E.n, E.sum, [&](int i) { return S.ldf(E.cdf_off, i); }, [&](int i) { return S.ldf(E.cdf_off, E.n + i); }, sample1, pdf0);
    const int w = E.off + 6 * ei;
    const float4 q0 = S.ld(w), q1 = S.ld(w + 1), q2 = S.ld(w + 2), q3 = S.ld(w + 3), q4 = S.ld(w + 4), q5 = S.ld(w + 5);
    const Vec3f p0(q0.x, q0.y, q0.z), e1(q0.w, q1.x, q1.y), n0(q1.z, q1.w, q2.x), n1(q2.y, q2.z, q2.w), p2(q3.x, q3.y, q3.z);
    const Vec3f dp0(q4.x, q4.y, q4.z), de1(q4.w, q5.x, q5.y);
    const Dual s1(sample1);
    r.p0 = Vec3d(fma_(Dual(e1.x, de1.x), s1, Dual(p0.x, dp0.x)), fma_(Dual(e1.y, de1.y), s1, Dual(p0.y, dp0.y)), fma_(Dual(e1.z, de1.z), s1, Dual(p0.z, dp0.z)));
    r.edge = normalize(e1);
    r.edge2 = p2 - p0;
    r.edge_id = ei; r.s1 = sample1;
    const Vec3f p0v = detach(r.p0);
    pdf0 /= norm(e1);
    const PositionSample<false> ps2 = sample_emitter_position<false, LDS, true>(S, p0v, s3.y, s3.z);      // (correctly rounded pdfs: these values fill the guiding grid)
    r.p2 = ps2.p; r.n = ps2.n; r.emitter_slot = ps2.slot;
    Vec3f e = r.p2 - p0v;
    const float distSqr = squared_norm(e);
    e = e / safe_sqrt(distSqr);
    const float cosTheta = dot(r.n, -e);
    const bool is_boundary = __float_as_int(q3.w) != 0;
    const int sgn0 = sign_eps(dot(n0, e), kEdgeEpsilon), sgn1 = sign_eps(dot(n1, e), kEdgeEpsilon);
    r.valid = (cosTheta > kEpsilon) && ((is_boundary && sgn0 != 0) || (!is_boundary && sgn0 * sgn1 < 0));
    r.pdf = r.valid ? pdf0 * ps2.pdf * (distSqr / cosTheta) : 0.f;
    return r;
}

// PathTracer::eval_secondary_edge<ad>, reference path.cpp:171-270.
// AD=true : returns the pixel index (or -1) and the tangent of the estimator in `value`.
// AD=false: guiding pass, `value` = value0 without the normal velocity (path.cpp:267-268), returns -1.
// what the closed-form reverse mode of the term needs from one evaluation (api.hip::k_secondary_edges<.., ADJ>): the detached factor, the
// normal the velocity is projected on, the two triangles and the camera ray
struct SecAdjInfo { Vec3f value0, n, x1, cam_o, cam_d, sd; int slot2, slot1; float qx, qy; };

template <bool AD, int LDS, bool COUNT>
PSDR_DEV int eval_boundary_segment(SceneView<LDS> &S, const SensorDev &cam, const BoundarySegSampleDirect &bss, Vec3f &value, int cam_comp = -1, SecAdjInfo *info = nullptr);

template <bool AD, int LDS, bool COUNT>
PSDR_DEV int eval_secondary_edge(SceneView<LDS> &S, const SecEdgeTables &E, const SensorDev &cam, const Vec3f &s3, Vec3f &value) {
    value = Vec3f(0.f);
    const BoundarySegSampleDirect bss = sample_boundary_segment_direct<LDS>(S, E, s3);
    if (!bss.valid) return -1;
    return eval_boundary_segment<AD, LDS, COUNT>(S, cam, bss, value);
}

// The traced part of eval_secondary_edge (path.cpp:176-270) for an already sampled, valid boundary segment, in the three pieces between its three rays
// (eval_boundary_segment below strings them together one ray at a time; api.hip::k_secondary_edges pipelines them - round 6):
//   sec_light_hit_ok   the segment reaches its emitter sample (closest hit of p0 -> p2 lies at p2, on an emitter)
//   sec_camera_sample  the opposite ray's hit p1 is seen by the sensor: sensor sample and camera ray through p1
//   sec_value          the estimator's value from the three hits
template <int LDS> PSDR_DEV bool sec_light_hit_ok(const SceneView<LDS> &S, const Its<false> &its2, const Vec3f &p2) {
    return its2.valid && mesh_emitter(S, its2.mesh) >= 0 && norm(its2.p - p2) < kShadowEpsilon;
}

template <bool AD> PSDR_DEV bool sec_camera_sample(const SceneTables &T, const SensorDev &cam, const Vec3f &p1, SensorDirectSample &sds, RayT<AD> &camera_ray, int cam_comp) {
    sds = sample_direct(T, cam, p1);
    if (!sds.valid) return false;
    camera_ray = sample_primary_ray<AD>(cam, sds.qx, sds.qy);
    if constexpr (AD) if (cam_comp >= 0) primary_ray_pose_tangent(cam, sds.qx, sds.qy, cam_comp, camera_ray);      // camera-pose probe
    return true;
}

template <bool AD, int LDS>
PSDR_DEV int sec_value(SceneView<LDS> &S, const BoundarySegSampleDirect &bss, const Its<false> &its2, const Its<false> &its1c, const Its<AD> &its1, const RayT<AD> &camera_ray,
                       const SensorDirectSample &sds, Vec3f &value, SecAdjInfo *info) {
    const Vec3f _p0 = detach(bss.p0), _p2 = bss.p2, _dir = normalize(_p2 - _p0);
    const Vec3f _p1 = its1c.p;
    if (!(its1.valid && norm(detach(its1.p) - _p1) < kShadowEpsilon)) return -1;
    if (mesh_bsdf(S, its1.mesh) < 0) return -1;

    const float dist = norm(_p2 - _p1), cos2 = fabsf(dot(bss.n, -_dir));
    const Vec3f e = cross(bss.edge, _dir);
    const float sinphi = norm(e);
    const Vec3f proj = normalize(cross(e, bss.n));
    const float sinphi2 = norm(cross(_dir, proj));
    const float base_v = (its1c.t / dist) * (sinphi / sinphi2) * cos2;
    if (!((sinphi > kEpsilon) && (sinphi2 > kEpsilon))) return -1;

    const Vec3f d0 = -detach(camera_ray.d);
    const Vec3f d0_local = to_local<false>(its1c, d0);
    Vec3f bsdf_val = bsdf_eval<false, LDS>(S, its1c, d0_local, true);
    const float correction = fabsf((its1c.wi.z * dot(d0, its1c.n)) / (d0_local.z * dot(_dir, its1c.n)));
    bsdf_val = bsdf_val * correction;
    Vec3f value0 = bsdf_val * eval_Le<false, LDS>(S, its2, true) * (base_v * sds.sensor_val / bss.pdf);

    if constexpr (AD) {
        const Vec3f n = normalize(cross(bss.n, proj));
        value0 = value0 * (sign1(dot(e, bss.edge2)) * sign1(dot(e, n)));
        Vec3d v0, e1, e2;
        load_geom<true, LDS>(S, its2.slot, v0, e1, e2);
        const Vec3d sd = normalize(bss.p0 - its1.p);
        if (info != nullptr) {
            info->value0 = value0; info->n = n; info->x1 = detach(its1.p); info->cam_o = detach(camera_ray.o); info->cam_d = detach(camera_ray.d);
            info->sd = detach(sd); info->slot2 = its2.slot; info->slot1 = its1.slot; info->qx = sds.qx; info->qy = sds.qy;
        }
        Dual u, v, t;
        ray_tri_uvt<Dual>(v0, e1, e2, its1.p, sd, u, v, t);
        // u2 = bilinear(detach(v0), detach(e1), detach(e2), uv): only the barycentrics carry a tangent
        const Vec3f v0f = detach(v0), e1f = detach(e1), e2f = detach(e2);
        const Vec3f du2(e1f.x * u.d + e2f.x * v.d, e1f.y * u.d + e2f.y * v.d, e1f.z * u.d + e2f.z * v.d);
        const float dn = dot(n, du2);
        value = value0 * dn;                                  // result - detach(result)
        return sds.pixel_idx;
    } else {
        value = value0;
        return -1;
    }
}

template <bool AD, int LDS, bool COUNT>
PSDR_DEV int eval_boundary_segment(SceneView<LDS> &S, const SensorDev &cam, const BoundarySegSampleDirect &bss, Vec3f &value, int cam_comp, SecAdjInfo *info) {
    value = Vec3f(0.f);
    const SceneTables &T = *S.T;
    const Vec3f _p0 = detach(bss.p0), _p2 = bss.p2, _dir = normalize(_p2 - _p0);

    RayT<false> r2; r2.o = _p0; r2.d = _dir;
    const Its<false> its2 = ray_intersect<false, false, LDS, COUNT>(S, r2, true);
    if (!sec_light_hit_ok(S, its2, _p2)) return -1;

    RayT<false> r1; r1.o = _p0; r1.d = -_dir;
    const Its<false> its1c = ray_intersect<false, false, LDS, COUNT>(S, r1, true);
    if (!its1c.valid) return -1;

    SensorDirectSample sds;
    RayT<AD> camera_ray;
    if (!sec_camera_sample<AD>(T, cam, its1c.p, sds, camera_ray, cam_comp)) return -1;
    const Its<AD> its1 = ray_intersect<AD, false, LDS, COUNT>(S, camera_ray, true);
    return sec_value<AD, LDS>(S, bss, its2, its1c, its1, camera_ray, sds, value, info);
}

} // namespace psdr
