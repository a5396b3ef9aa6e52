/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain scalar C++17, one sample per loop iteration, OpenMP over samples) of the
 * psdr-jit PathTracer.renderC / renderD hot path, used ONLY as the checker in tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The shipped product
 * (psdr_jit_amd/, include/psdr_hip.h) never includes, links or calls anything in oracle/.
 *
 * PARITY: the reference cannot be compiled or imported in this environment (its drjit submodule is
 * empty and un-pinned, it needs CUDA+OptiX, and it ships no tests).  This restatement follows the
 * reference sources cited next to every function and is pinned against the reference's only held
 * outputs - the figures and log lines embedded in its tutorial notebooks, extracted into
 * tests/golden/notebooks/ (tests/test_oracle_notebooks.py) - and by analytic known-answer tests
 * (tests/test_oracle_kat.py).  One figure is met only loosely, see DESIGN.md section 2.
 *
 * C ABI so that tests can drive it through ctypes.  All matrices are row-major float[16].
 * "d_*" members are the forward-mode tangent of the member they shadow with respect to ONE scalar
 * scene parameter (the reference obtains the same quantity with drjit.forward_to, README.md:87-104).
 */
#ifndef ORC_ORACLE_H
#define ORC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_mesh {            /* reference Mesh, include/psdr/shape/mesh.h:13-146 */
    int n_vertices, n_faces, n_uvs;
    const float *vertices;           /* [n_vertices*3] object space (m_vertex_positions_raw) */
    const float *d_vertices;         /* optional tangent, may be NULL */
    const int *faces;                /* [n_faces*3] */
    const float *uvs;                /* [n_uvs*2] or NULL */
    const int *face_uvs;             /* [n_faces*3] or NULL */
    float to_world_left[16], to_world_raw[16], to_world_right[16];
    float d_to_world_left[16], d_to_world_raw[16], d_to_world_right[16];
    int bsdf_id;
    int emitter_id;                  /* index into orc_scene_desc.emitters, -1 = not a light */
    int use_face_normals;            /* m_use_face_normals (default 0) */
    int enable_edges;                /* m_enable_edges (default 1) */
} orc_mesh;

typedef struct orc_bsdf {            /* Diffuse, include/psdr/bsdf/diffuse.h */
    int type;                        /* 0 = Diffuse, 1 = Microfacet, 2 = RoughConductor, 3 = RoughDielectric (alpha_u/v; eta[0] = intIOR/extIOR, eta[1] = extIOR/intIOR), 4 = MicrofacetPerVertex, 5 = NormalMap */
    float reflectance[3], d_reflectance[3];
    int two_sided;
    /* textured reflectance (Bitmap3fD with resolution > 1x1, bitmap.cpp:47-128): tex_data != NULL overrides `reflectance` */
    int tex_width, tex_height;
    const float *tex_data;           /* [tex_height*tex_width*3] row-major rgb */
    const float *d_tex_data;         /* optional tangent of the texels */
    /* type 1 = Microfacet (src/bsdf/microfacet.cpp): reflectance = diffuse reflectance, plus */
    float specular[3], d_specular[3];
    float roughness, d_roughness;
    /* type 2 = RoughConductor (src/bsdf/roughconductor.cpp): specular = specular_reflectance, plus */
    float alpha_u, alpha_v, d_alpha_u, d_alpha_v;
    float eta[3], d_eta[3], k[3], d_k[3];
    /* Microfacet bitmap parameters above 1x1 (microfacet.cpp:38-45): override specular / roughness; tex_data is then the
     * diffuse reflectance map.  spec: rgb, rough: one channel */
    int spec_tex_width, spec_tex_height;
    const float *spec_tex_data, *d_spec_tex_data;
    int rough_tex_width, rough_tex_height;
    const float *rough_tex_data, *d_rough_tex_data;
    /* type 4 = MicrofacetPerVertex (src/bsdf/microfacet_pv.cpp): values per mesh-local vertex, interpolated with the barycentrics */
    int pv_count;
    const float *pv_specular, *pv_diffuse, *pv_roughness;          /* [n*3], [n*3], [n] */
    const float *d_pv_specular, *d_pv_diffuse, *d_pv_roughness;    /* optional tangents */
    /* type 5 = NormalMap (src/bsdf/normalmap.cpp): reflectance / tex_data = the normal map (rgb in [0,1]), nested_bsdf = index of the
     * BSDF it perturbs (any type but NormalMap) */
    int nested_bsdf;
    /* uv transform of the bitmap slots tex / spec_tex / rough_tex: Bitmap::m_rot, m_scale, m_trans.x, m_trans.y
     * (include/psdr/core/bitmap.h:37-39, src/core/bitmap.cpp:64-86) and their tangents; identity = {0, 1, 0, 0} */
    float tex_xf[3][4], d_tex_xf[3][4];
} orc_bsdf;

typedef struct orc_emitter {         /* AreaLight (include/psdr/emitter/area.h) or EnvironmentMap (emitter/envmap.h) */
    float radiance[3], d_radiance[3];
    int type;                        /* 0 = AreaLight, 1 = EnvironmentMap (at most one, Scene::add_EnvironmentMap) */
    int env_width, env_height;       /* m_radiance.m_resolution */
    const float *env_data;           /* [env_height*env_width*3] row-major rgb, lat-long */
    float env_scale;                 /* m_scale */
    float env_to_world_left[16], env_to_world_raw[16];
    /* tangents of the differentiable members (envmap.h:40-45): texels of m_radiance (may be NULL), m_scale, m_to_world_left */
    const float *d_env_data;
    float d_env_scale;
    float d_env_to_world_left[16];
    float env_uv_xf[4], d_env_uv_xf[4];   /* m_radiance's rotate, scale, translate.x, translate.y (as orc_bsdf.tex_xf) */
} orc_emitter;

typedef struct orc_camera {          /* PerspectiveCamera(fov_x, near, far) */
    float fov_x, near_clip, far_clip;
    float to_world_left[16], to_world_raw[16], to_world_right[16];
    float d_to_world_left[16], d_to_world_raw[16], d_to_world_right[16];
    int orthographic;                /* 1: OrthographicCamera(near, far) (src/sensor/orthographic.cpp), fov_x unused */
} orc_camera;

typedef struct orc_scene_desc {
    int n_meshes;   const orc_mesh *meshes;
    int n_bsdfs;    const orc_bsdf *bsdfs;
    int n_emitters; const orc_emitter *emitters;
    int n_cameras;  const orc_camera *cameras;
    int width, height, spp, sppe, sppse;   /* RenderOption, include/psdr/types.h:217-228 */
} orc_scene_desc;

/* One of the reference's three N-lane samplers (Scene::m_samplers[0..2]) in closed form:
 * lane i was seeded with seed_value = seed + (pix_ids ? pix_ids[i / spp] : i)   (integrator.cpp:24-28)
 * and has already produced `skip` numbers since then (m_samplers is mutable state that every
 * render call advances, scene.h:76). */
typedef struct orc_sampler {
    uint64_t seed;
    uint64_t skip;
} orc_sampler;

typedef struct orc_scene orc_scene;

/* = Scene::configure(active_sensor) (scene.cpp:311-601).  Primary-edge lists are kept only for the
 * sensors listed in active_sensors (scene.cpp:381-405).  Returns NULL on error. */
orc_scene *orc_scene_create(const orc_scene_desc *desc, const int *active_sensors, int n_active);
void orc_scene_destroy(orc_scene *s);
const char *orc_last_error(void);
/* integrator used by the render calls that follow: -1 = PathTracer(max_depth) (default), 0/1/2 = DirectIntegrator(mis)
 * (reference src/integrator/direct.cpp: emitter sampling only / BSDF sampling only / both with MIS; max_depth is ignored) */
void orc_set_direct_mis(orc_scene *s, int mis);
/* first-hit integrators for the render calls that follow (no secondary-edge term): field = -1 none, 0 silhouette, 1 position,
 * 2 depth, 3 geoNormal, 4 shNormal, 5 uv, 6 bsdf, 7 segmentation (FieldExtractionIntegrator, src/integrator/field.cpp), 8 =
 * CollocatedIntegrator(intensity) (src/integrator/collocated.cpp); object = mesh index filter or -1 */
void orc_set_field(orc_scene *s, int field, int object, float intensity, float d_intensity);
void orc_set_num_threads(int n);     /* 0 = OpenMP default */
int orc_get_num_threads(void);

/* --- introspection used by the parity tests on the configured snapshot --- */
int orc_num_triangles(const orc_scene *s);
int orc_num_sec_edges(const orc_scene *s);
int orc_num_primary_edges(const orc_scene *s, int sensor_id);
/* 25 floats per triangle: p0 e1 e2 n0 n1 n2 face_normal (3 each), face_area, then 3 ints bit-cast
 * (face_indices); `tangent` selects the d/dparam of the float fields. */
void orc_get_triangle_info(const orc_scene *s, int tangent, float *out);
/* 16 floats per edge: p0 e1 n0 n1 p2 (3 each), is_boundary */
void orc_get_sec_edges(const orc_scene *s, int tangent, float *out);
/* 7 floats per edge: p0.xy p1.xy normal.xy length */
void orc_get_primary_edges(const orc_scene *s, int sensor_id, int tangent, float *out);
/* mesh-local edge list of mesh m: 5 ints per edge (v0 v1 f0 f1 opp); returns count */
int orc_get_mesh_edges(const orc_scene *s, int mesh, int *out, int cap);
float orc_emitter_sampling_weight(const orc_scene *s, int emitter);
/* configured EnvironmentMap (0 if the scene has none): bounds = lower xyz, upper xyz; reso = cell grid; cell arrays have reso[0]*reso[1] entries */
/* Scene::m_lower / m_upper: lower xyz, upper xyz (all vertices and perspective-camera positions; with an environment map, + its 5 % margin) */
void orc_scene_aabb(const orc_scene *s, float bounds[6]);
int orc_envmap_info(const orc_scene *s, float bounds[6], int reso[2], float *cell_sum);
void orc_envmap_cells(const orc_scene *s, float *pmf, float *cmf);
/* EnvironmentMap::sample_position / sample_position_pdf alone (host arrays, same layout as psdr_hip_env_sample / _pdf) */
void orc_env_sample(const orc_scene *s, int n, const float *ref_p, const float *s2, float *out_p, float *out_n, float *out_pdf);
void orc_env_pdf(const orc_scene *s, int n, const float *ref_p, const float *p, const float *nrm, float *out_pdf);

/* closest-hit query (Scene_OptiX::ray_intersect restated, scene_optix.cpp:343-410):
 * out_tri = global triangle id or -1, out_uv barycentrics, out_t distance. use_bvh: 0 = brute force. */
void orc_trace(const orc_scene *s, int n, const float *o, const float *d, int use_bvh,
               int *out_tri, float *out_uv, float *out_t);

/* which terms of renderD to evaluate (bit mask) */
#define ORC_TERM_INTERIOR 1
#define ORC_TERM_PRIMARY  2
#define ORC_TERM_SECONDARY 4

/* = Integrator::renderC (integrator.cpp:12-48).  pix_ids == NULL renders the full frame
 * (out_rgb is [H*W*3], pixel-interleaved, pixel = y*W + x); otherwise the batch path
 * (__render_batch, integrator.cpp:139-176) with out_rgb [n_pix*3].
 * Only lanes in [lane_begin, lane_end) are evaluated (lane_end < 0 = all) so that a partition of
 * the lane range can be checked to sum to the full image (multi-GPU sharding). */
int orc_render_c(const orc_scene *s, int sensor_id, int max_depth, int hide_emitters,
                 orc_sampler sampler, const int *pix_ids, int n_pix,
                 int64_t lane_begin, int64_t lane_end, float *out_rgb);

/* = Integrator::renderD (integrator.cpp:51-100) followed by drjit.forward_to(img):
 * out_rgb = image, out_drgb = d image / d parameter.  samplers[0..2] = interior / primary edge /
 * secondary edge.  guiding: handle from orc_guiding_build or NULL. */
typedef struct orc_guiding orc_guiding;
int orc_render_d(const orc_scene *s, int sensor_id, int max_depth, int hide_emitters,
                 const orc_sampler samplers[3], const int *pix_ids, int n_pix,
                 const orc_guiding *guiding, int terms,
                 int shard_rank, int shard_count,   /* 256-lane chunks k with k % c == r of each sampler; c<=1 = all */
                 int shard_mode,                    /* 0: the interleaved chunks above; 1: contiguous runs (pixel-row tiles; psdr_render_args.shard_mode) */
                 float *out_rgb, float *out_drgb);

/* per-lane radiance of the interior term (debug/parity aid): out [n_lanes*3] */
int orc_li_lanes(const orc_scene *s, int sensor_id, int max_depth, int hide_emitters,
                 orc_sampler sampler, int64_t lane_begin, int64_t lane_end, float *out);

/* = PathTracer::preprocess_secondary_edges (path.cpp:130-168) */
orc_guiding *orc_guiding_build(const orc_scene *s, int sensor_id, int max_depth,
                               const int reso[4], int nrounds, int seed);
int orc_guiding_num_cells(const orc_guiding *g);
void orc_guiding_get_mass(const orc_guiding *g, float *out);
void orc_guiding_destroy(orc_guiding *g);

/* building blocks exposed for known-answer tests */
uint64_t orc_tea64(uint64_t v0, uint64_t v1);
void orc_pcg32_raw(uint64_t initstate, uint64_t initseq, int n, uint32_t *out);
void orc_sampler_floats(uint64_t seed_value, uint64_t lane, uint64_t skip, int n, float *out);
void orc_square_to_cosine_hemisphere(int n, const float *uv, float *out_xyz);
void orc_square_to_uniform_triangle(int n, const float *uv, float *out_ab);
void orc_coordinate_system(const float n[3], float s[3], float t[3]);
int orc_distrb_sample_reuse(int size, const float *pmf, float *sample_inout, float *pdf_out);
/* Microfacet BSDF building blocks in the local shading frame (microfacet.cpp / ggx.cpp): params = specular rgb, diffuse rgb,
 * roughness, then their tangents in the same order (14 floats); eval returns value rgb then tangent rgb (6 floats) */
void orc_microfacet_eval(const float params[14], int two_sided, const float wi[3], const float wo[3], float out[6]);
float orc_microfacet_pdf(float roughness, int two_sided, const float wi[3], const float wo[3]);
int orc_microfacet_sample(float roughness, int two_sided, const float wi[3], const float s3[3], float wo_out[3], float *pdf_out);
float orc_ggx_eval(float alpha, const float m[3]);
float orc_fresnel_conductor(float eta, float k, float cos_theta_i);      /* utils.h:166-182 */
/* RoughDielectric (roughdielectric.cpp): q = {alpha, intIOR/extIOR}; eval out = {value, d/d alpha, d/d eta} */
void orc_dielectric_eval(const float q[2], const float wi[3], const float wo[3], float out[3]);
float orc_dielectric_pdf(const float q[2], const float wi[3], const float wo[3]);
int orc_dielectric_sample(const float q[2], const float wi[3], const float s3[3], float wo_out[3], float *pdf_out);
void orc_fresnel_dielectric(float eta, float c, float out[4]);   /* F, cos_theta_t, eta_it, eta_ti (utils.h:184-215) */

#ifdef __cplusplus
}
#endif
#endif
