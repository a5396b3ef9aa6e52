// ORACLE — TEST INFRASTRUCTURE ONLY (see num.h header).
//
// integrator.h — per-sample restatement of the hot path: Scene::ray_intersect, emitter/BSDF
// sampling, PathTracer::__Li, the primary- and secondary-edge estimators.
// One call = one lane of the reference's wavefront arrays.
#pragma once
#include "scene.h"
#include "sampler.h"

namespace orc {

template <bool ad> struct Ray { V3<Real<ad>> o, d; };
using RayC = Ray<false>; using RayD = Ray<true>;

template <bool ad> struct Frame { V3<Real<ad>> s, t, n; };

// reference include/psdr/core/intersection.h:24-60
template <bool ad> struct Its {
    using R = Real<ad>;
    bool valid = false;
    int tri = -1, mesh = -1;
    V3<R> wi, p, n;
    R t = R(0.f), J = R(1.f);
    Frame<ad> sh;
    V2<R> uv;
    V3<R> dp_du = V3<R>(R(0.f));   // zeros without a uv parameterisation (scene.cpp:760-762)
    V2<R> bc;                    // barycentrics (Intersection::bc): detached in C / path-space mode, differentiable otherwise
};

struct PrimaryEdgeSample { Dual x_dot_n; int idx; RayC ray_n, ray_p; float pdf; };
struct SensorDirectSample { V2f q; int pixel_idx; float sensor_val; bool valid; };
struct BoundarySegSampleDirect { bool valid; float pdf; V3d p0; V3f edge, edge2, p2, n; };

// cell grid of HyperCubeDistribution<3> (reference src/core/cube_distrb.cpp:10-64)
struct Guiding {
    int reso[3] = {0, 0, 0};
    int num_cells = 0;
    float unit[3] = {0, 0, 0};
    Distrb distrb;
    float sample_reuse(V3f &s) const;
};

template <bool ad> Ray<ad> sample_primary_ray(const CameraC &cam, const V2<Real<ad>> &s);
SensorDirectSample sample_direct(const Scene &sc, const CameraC &cam, const V3f &p);
PrimaryEdgeSample sample_primary_edge(const Scene &sc, const CameraC &cam, float sample1);

template <bool ad, bool path_space>
Its<ad> ray_intersect(const Scene &sc, const Ray<ad> &ray, bool active, int *out_tri = nullptr);

template <bool ad>
V3<Real<ad>> Li(const Scene &sc, LaneSampler &sampler, const Ray<ad> &ray, bool active, int max_depth, bool hide_emitters);

BoundarySegSampleDirect sample_boundary_segment_direct(const Scene &sc, V3f sample3);

// returns pixel index (or -1) and the value; ad=true: value = result - detach(result)
template <bool ad>
int eval_secondary_edge(const Scene &sc, const CameraC &cam, const V3f &sample3, V3<Real<ad>> &value);

void kat_env_sample(const Scene &sc, const V3f &ref_p, float sx, float sy, V3f &p, V3f &n, float &pdf);
float kat_env_pdf(const Scene &sc, const V3f &ref_p, const V3f &p, const V3f &n);
void kat_microfacet_eval(const float *params, int two_sided, const float *wi, const float *wo, float *out);
float kat_microfacet_pdf(float roughness, int two_sided, const float *wi, const float *wo);
int kat_microfacet_sample(float roughness, int two_sided, const float *wi, const float *s3, float *wo_out, float *pdf_out);
float kat_ggx_eval(float alpha, const float *m);
float kat_fresnel_conductor(float eta, float k, float c);
void kat_dielectric_eval(const float *q, const float *wi, const float *wo, float *out);
float kat_dielectric_pdf(const float *q, const float *wi, const float *wo);
int kat_dielectric_sample(const float *q, const float *wi, const float *s3, float *wo_out, float *pdf_out);
void kat_fresnel_dielectric(float eta, float c, float *out);
void kat_cosine_hemisphere(float sx, float sy, float *o);
void kat_uniform_triangle(float sx, float sy, float *o);
void kat_coordinate_system(const float *n, float *s, float *t);

} // namespace orc
