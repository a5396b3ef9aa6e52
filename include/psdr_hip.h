/* psdr_hip.h — C ABI of libpsdr_hip.so: the MI355X (gfx950) implementation of psdr-jit's
 * PathTracer.renderC / renderD hot path.
 *
 * The reference has no FFI seam for this path (Python -> pybind11 -> C++ virtuals -> drjit/OptiX,
 * reference src/psdr.cpp:100-441); the seam below is the one SURVEY.md §8(b) specifies: a POD
 * "configured-scene snapshot" (= what Scene::configure leaves in Scene's public SoA members,
 * reference include/psdr/scene/scene.h:56-90) plus render entry points that replace
 *   Integrator::renderC / renderD            reference src/integrator/integrator.cpp:12-100
 *   Integrator::__render / __render_batch    reference src/integrator/integrator.cpp:103-176
 *   Integrator::render_primary_edges         reference src/integrator/integrator.cpp:179-198
 *   PathTracer::__Li                         reference src/integrator/path.cpp:35-127
 *   PathTracer::render_secondary_edges       reference src/integrator/path.cpp:171-294
 *   PathTracer::preprocess_secondary_edges   reference src/integrator/path.cpp:130-168
 *   Scene::ray_intersect + Scene_OptiX       reference src/scene/scene.cpp:612-806, scene_optix.cpp:265-410
 *
 * Conventions: every function returns 0 on success, non-zero on error (message via
 * psdr_hip_last_error(), which mirrors psdr_jit::Exception's text, reference include/misc/Exception.h).
 * All pointers inside psdr_scene_snapshot are HOST pointers owned by the caller and are copied
 * during psdr_hip_scene_create.  Image buffers passed to the render calls are DEVICE pointers
 * owned by the caller (e.g. a torch tensor's data_ptr); `stream` is a hipStream_t (NULL = default
 * stream).  No call synchronises the device.  Float data is IEEE float32, indices int32, RNG state uint64.
 * Matrices are row-major float[16].  Tangent ("d_") arrays hold d(value)/d(theta) for ONE scalar scene
 * parameter theta (forward mode; the reference gets the same from drjit.forward_to) and may be NULL.
 */
#ifndef PSDR_HIP_H
#define PSDR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PSDR_HIP_ABI_VERSION 16

/* TriangleInfo SoA, reference include/psdr/types.h:162-175 (+ Scene::m_triangle_uv,
 * m_triangle_face_normals, scene.cpp:528-542).  Arrays of n_triangles rows. */
typedef struct psdr_triangles {
    int32_t n_triangles;
    const float *p0, *e1, *e2, *n0, *n1, *n2, *face_normal;   /* [n*3] each */
    const float *face_area;                                   /* [n]   */
    const float *uv;                                          /* [n*6] uv0 uv1 uv2, zeros if no uv */
    const int32_t *mesh_id;                                   /* [n]   */
    const uint8_t *use_face_normal;                           /* [n]   */
    const int32_t *face_indices;                              /* [n*3] mesh-local vertex ids (TriangleInfo::face_indices, types.h:172); may be NULL
                                                                 unless a MicrofacetPerVertex BSDF is present */
    /* forward tangents (NULL = none) */
    const float *d_p0, *d_e1, *d_e2, *d_n0, *d_n1, *d_n2, *d_face_normal, *d_face_area;
} psdr_triangles;

typedef struct psdr_mesh_rec {       /* what the kernels need of reference Mesh (mesh.h:82-141) */
    int32_t bsdf_id;                 /* -1 = none */
    int32_t emitter_id;              /* -1 = not an emitter */
    int32_t face_offset, n_faces;    /* rows of psdr_triangles */
    float inv_total_area;            /* Mesh::m_inv_total_area */
    /* Mesh::m_face_distrb (DiscreteDistribution over faces ~ area); used when emitter_id >= 0 */
    int32_t distrb_offset;           /* into psdr_scene_snapshot.face_pmf / face_cmf */
    float distrb_sum;
} psdr_mesh_rec;

typedef struct psdr_bsdf_rec {       /* Diffuse, reference src/bsdf/diffuse.cpp */
    int32_t type;                    /* 0 = Diffuse, 1 = Microfacet, 2 = RoughConductor, 3 = RoughDielectric, 4 = MicrofacetPerVertex, 5 = NormalMap */
    int32_t two_sided;
    float reflectance[3], d_reflectance[3];
    /* textured reflectance: Bitmap3fD with a resolution above 1x1 (bitmap.cpp:47-128, looked up at its.uv with flip_v);
     * tex_data != NULL overrides `reflectance`.  Host pointers, copied by psdr_hip_scene_create. */
    int32_t tex_width, tex_height;
    const float *tex_data;           /* [tex_height*tex_width*3] row-major rgb */
    const float *d_tex_data;         /* forward tangent of the texels, may be NULL */
    /* type 1 = Microfacet (src/bsdf/microfacet.cpp, ggx.cpp): `reflectance` is its diffuse reflectance, plus */
    float specular[3], d_specular[3];
    float roughness, d_roughness;
    /* type 2 = RoughConductor (src/bsdf/roughconductor.cpp): `specular` is its specular_reflectance, plus */
    float alpha_u, alpha_v, d_alpha_u, d_alpha_v;
    float eta[3], d_eta[3], k[3], d_k[3];
    /* type 3 = RoughDielectric (src/bsdf/roughdielectric.cpp): alpha_u / alpha_v as above, eta[0] = m_eta = intIOR / extIOR,
     * eta[1] = m_inv_eta = extIOR / intIOR (the reference stores both), d_eta[0..1] their tangents */
    /* Microfacet bitmap parameters with a resolution above 1x1 (microfacet.cpp:38-45, looked up at its.uv like tex_data, which
     * is then the diffuse reflectance map): override `specular` (rgb) / `roughness` (one channel).  Host pointers, copied. */
    int32_t spec_tex_width, spec_tex_height;
    const float *spec_tex_data, *d_spec_tex_data;
    int32_t rough_tex_width, rough_tex_height;
    const float *rough_tex_data, *d_rough_tex_data;
    /* type 4 = MicrofacetPerVertex (src/bsdf/microfacet_pv.cpp): parameters per mesh-local vertex, interpolated over the hit
     * triangle with the barycentrics (its.face_indices, its.bc).  Host pointers, copied. */
    int32_t pv_count;
    const float *pv_specular, *pv_diffuse, *pv_roughness;          /* [n*3], [n*3], [n] */
    const float *d_pv_specular, *d_pv_diffuse, *d_pv_roughness;    /* forward tangents, may be NULL */
    /* type 5 = NormalMap (src/bsdf/normalmap.cpp): `reflectance` / tex_data = the normal map (rgb in [0, 1], looked up at its.uv),
     * nested_bsdf = index in psdr_scene_snapshot.bsdfs of the BSDF it perturbs (any type but NormalMap; the host may append the
     * nested BSDF as an extra entry no mesh refers to) */
    int32_t nested_bsdf;
    /* uv transform of the three bitmap slots above (0 tex_data, 1 spec_tex_data, 2 rough_tex_data): Bitmap::m_rot, m_scale,
     * m_trans.x, m_trans.y (include/psdr/core/bitmap.h:37-39, applied by src/core/bitmap.cpp:64-86; the reference binds them as
     * rotate / scale / translate, src/psdr.cpp:204-206, 217-219) and their forward tangents.  Identity = {0, 1, 0, 0}.
     * Reverse mode: psdr_grads.g_uv_xf (ABI 12; until then four psdr_hip_render_d_fwd launches per bitmap). */
    float tex_xf[3][4], d_tex_xf[3][4];
} psdr_bsdf_rec;

typedef struct psdr_emitter_rec {    /* AreaLight (src/emitter/area.cpp) or EnvironmentMap (src/emitter/envmap.cpp) */
    int32_t mesh_id;                 /* the light's mesh; for the envmap the bounding cube scene.cpp:442-480 adds */
    float sampling_weight;           /* normalised, scene.cpp:511-514 */
    float radiance[3], d_radiance[3];
    int32_t type;                    /* 0 = AreaLight, 1 = EnvironmentMap (psdr_scene_snapshot.envmap) */
} psdr_emitter_rec;

/* EnvironmentMap after configure() (envmap.cpp:17-44): lat-long radiance, transform, the scene box it is carried to
 * (scene.cpp:436-440) and its HyperCubeDistribution2f over (2(W-1)) x (2(H-1)) cells, cell index = cx*reso[1] + cy */
typedef struct psdr_envmap_rec {
    int32_t width, height;
    const float *radiance;           /* [height*width*3] row-major rgb (Bitmap3fD m_radiance) */
    float scale;                     /* m_scale */
    float to_world[16], from_world[16];
    float lower[3], upper[3];
    int32_t reso[2];
    const float *cell_pmf, *cell_cmf; /* unnormalised masses and their running sums, [reso[0]*reso[1]] */
    float cell_sum;
    /* forward tangents of the differentiable members (envmap.h:40-45): texels of m_radiance (host pointer, may be NULL),
     * m_scale, m_to_world / m_from_world */
    const float *d_radiance;
    float d_scale;
    float d_to_world[16], d_from_world[16];
    /* uv transform of m_radiance (as psdr_bsdf_rec.tex_xf: rotate, scale, translate.x, translate.y) and its forward tangent;
     * cell_pmf / cell_cmf are the masses of the transformed map (envmap.cpp:28-31 evaluates m_radiance through Bitmap::eval) */
    float radiance_xf[4], d_radiance_xf[4];
} psdr_envmap_rec;

/* SecondaryEdgeInfo SoA, reference include/psdr/edge/edge.h:49-68 */
typedef struct psdr_sec_edges {
    int32_t n_edges;
    const float *p0, *e1, *n0, *n1, *p2;      /* [n*3] */
    const uint8_t *is_boundary;               /* [n] */
    const float *d_p0, *d_e1;                 /* tangents used by the estimator */
    const float *pmf, *cmf;                   /* Scene::m_sec_edge_distrb, [n] */
    float sum;
} psdr_sec_edges;

/* PerspectiveCamera after configure(), reference src/sensor/perspective.cpp:10-152 */
typedef struct psdr_sensor_rec {
    float sample_to_camera[16];
    float to_world[16], d_to_world[16];
    float world_to_sample[16], d_world_to_sample[16];
    float cam_pos[3], cam_dir[3];
    float inv_area;
    int32_t orthographic;            /* 1: OrthographicCamera (src/sensor/orthographic.cpp): rays start on the near plane along cam_dir */
    /* PrimaryEdgeInfo (edge.h:27-40) + Sensor::m_edge_distrb; n_edges == 0 disables the term */
    int32_t n_edges;
    const float *edge_p0, *edge_p1;           /* [n*2] sample-space end points */
    const float *d_edge_p0, *d_edge_p1;       /* [n*2] tangents */
    const float *edge_normal;                 /* [n*2] */
    const float *edge_length;                 /* [n] */
    const float *edge_pmf, *edge_cmf;         /* [n] */
    float edge_sum;
} psdr_sensor_rec;

/* What ONE mesh's rows of psdr_triangles / psdr_sec_edges are a function of (ABI 16): Mesh::configure + process_mesh, reference src/shape/mesh.cpp:23-62, 317-400, run on
 * drjit device arrays - the reference computes these rows ON THE GPU in every Scene::configure.  With psdr_scene_snapshot.geometry set, psdr_hip_scene_update computes the
 * rows of the meshes flagged `moved` on the device, in the host's own (value, tangent) arithmetic (csrc/host/hnum.h compiled for both sides: the same bits), from the raw vertices
 * and the composed transform: ~1 MB travels instead of the 30 MB of rows the host would write.  The snapshot's row arrays stay valid (they are what psdr_hip_scene_create,
 * a rebuild and the host-side consumers read); the edge CDFs and the sensors' primary edges still come from the host. */
typedef struct psdr_mesh_geometry {
    int32_t n_vertices, n_faces, n_edges;          /* n_edges: this mesh's rows of psdr_sec_edges (0 when its edges are disabled or sppse == 0) */
    const float *vertices_raw, *d_vertices_raw;    /* [n_vertices*3] Mesh::m_vertex_positions_raw and its forward tangent */
    float to_world[16], d_to_world[16];            /* m_to_world_left * m_to_world_raw * m_to_world_right, row-major, composed in (value, tangent) arithmetic by the host */
    const int32_t *faces;                          /* [n_faces*3] mesh-local vertex ids (Mesh::m_face_indices) */
    const int32_t *vf_begin, *vf_item;             /* [n_vertices+1], [n_faces*3]: per vertex its (face << 2 | corner) incidences in the order the reference's three scatter passes
                                                      visit them (mesh.cpp:34-41), so that the per-vertex normal sums add in that very order */
    const int32_t *edges;                          /* [n_edges*5] v0 v1 f0 f1 opp (Mesh::m_edge_indices; f1 = -1: boundary edge) */
    int32_t mesh_id, use_face_normals;
    uint64_t topology_version;                     /* changes whenever faces / vf lists / edges / the counts do: the device keeps the topology of the version it saw last */
    int32_t moved;                                 /* 1: vertices, transform or their tangents differ from what the previous create / update carried */
} psdr_mesh_geometry;

typedef struct psdr_scene_snapshot {
    int32_t abi_version;                      /* PSDR_HIP_ABI_VERSION */
    int32_t width, height, spp, sppe, sppse;  /* RenderOption, reference include/psdr/types.h:217-228 */
    psdr_triangles tris;
    int32_t n_meshes;   const psdr_mesh_rec *meshes;
    int32_t n_bsdfs;    const psdr_bsdf_rec *bsdfs;
    int32_t n_emitters; const psdr_emitter_rec *emitters;
    const float *emitter_pmf, *emitter_cmf;   /* Scene::m_emitters_distrb, [n_emitters] */
    float emitter_sum;
    int32_t n_face_distrb;                    /* total length of the concatenated face CDFs */
    const float *face_pmf, *face_cmf;
    psdr_sec_edges sec_edges;
    int32_t n_sensors;  const psdr_sensor_rec *sensors;
    const psdr_envmap_rec *envmap;            /* Scene::m_emitter_env, NULL = none */
    const psdr_mesh_geometry *geometry;       /* [n_meshes] or NULL: lets psdr_hip_scene_update compute moved meshes' rows on the device (BVH scenes; see psdr_mesh_geometry) */
    int32_t rows_valid;                       /* psdr_hip_scene_update only.  1: the row arrays of `tris` and `sec_edges` hold this state.  0: the caller has not computed the rows of
                                                 the moved meshes (their sizes, the CDFs and everything else are valid) and relies on `geometry`: the update computes them on the
                                                 device, or returns PSDR_HIP_NEED_ROWS with the scene unchanged - then call again with the rows */
} psdr_scene_snapshot;

/* One of Scene::m_samplers[0..2] in closed form: lane i was seeded with
 * seed_value = seed + (pix_ids ? pix_ids[i / spp] : i) (integrator.cpp:24-28, scene.cpp:330-344)
 * and has already produced `skip` numbers.  The kernels never store per-lane RNG state. */
typedef struct psdr_sampler {
    uint64_t seed;
    uint64_t skip;
} psdr_sampler;

#define PSDR_SHARD_CHUNK 256
#define PSDR_SHARD_INTERLEAVED 0
#define PSDR_SHARD_ROWS 1

#define PSDR_TERM_INTERIOR  1
#define PSDR_TERM_PRIMARY   2
#define PSDR_TERM_SECONDARY 4

typedef struct psdr_hip_scene psdr_hip_scene;      /* device-resident scene + BVH */
typedef struct psdr_hip_guiding psdr_hip_guiding;  /* device-resident HyperCubeDistribution3f */

typedef struct psdr_render_args {
    int32_t sensor_id;
    int32_t max_depth;            /* PathTracer(max_depth) */
    int32_t hide_emitters;        /* PathTracer::m_hide_emitters */
    psdr_sampler samplers[3];     /* interior / primary edge / secondary edge */
    const int32_t *pix_ids;       /* DEVICE pointer, batch_pix (integrator.cpp:139-176); NULL = full frame */
    int32_t n_pix;
    int32_t terms;                /* PSDR_TERM_* mask (renderD only) */
    int32_t shard_rank, shard_count;  /* multi-GPU: rank r of c evaluates the PSDR_SHARD_CHUNK-lane chunks k of each
                                       * sampler's lane range with k % c == r (interleaved for load balance); c<=1 = all */
    const psdr_hip_guiding *guiding;  /* NULL = unguided secondary edges */
    int32_t zero_output;          /* 1: the call clears out buffers first (hipMemsetAsync on `stream`) */
    int32_t direct_mode;          /* 0: PathTracer(max_depth); 1 + mis: DirectIntegrator(mis), mis = 0/1/2 (reference
                                     src/integrator/direct.cpp:34-132: emitter sample only / BSDF sample only / both with MIS) */
    int32_t field_mode;           /* 0: none; 1 + f: first-hit integrators (no secondary-edge term): FieldExtractionIntegrator (reference
                                     src/integrator/field.cpp) f = 0 silhouette, 1 position, 2 depth, 3 geoNormal, 4 shNormal, 5 uv, 6 bsdf,
                                     7 segmentation; f = 8: CollocatedIntegrator(intensity) (src/integrator/collocated.cpp) */
    int32_t field_object;         /* mesh index the field is restricted to (field.cpp:59-66), -1 = all; only read when field_mode > 0 */
    float intensity, d_intensity; /* CollocatedIntegrator::m_intensity and its forward tangent */
    int32_t skip_static_edges;    /* psdr_hip_render_d_fwd, primary-edge term: 1 = a sample whose edge point has a zero normal velocity under the installed
                                     tangents (an edge of a mesh that does not move, a camera that does not move) is not traced - its contribution to out_drgb is
                                     d(x.n) x (Ln - Lp) / pdf = exactly 0 (integrator.cpp:179-198), so the same numbers are added to the derivative image; 0 = every sample's
                                     two paths are traced, as the reference does */
    int32_t shard_mode;           /* multi-GPU partition of every sampler's lanes over shard_count ranks (ABI 15): PSDR_SHARD_INTERLEAVED (0) = the chunks k % c == r above;
                                     PSDR_SHARD_ROWS (1) = CONTIGUOUS ranges - the interior sampler by whole pixel rows (rank r renders rows [r R, (r + 1) R), R = ceil(H / c):
                                     one block of the image per rank, which an all-gather assembles; batch rendering: by whole pixels of the list), the two edge samplers by
                                     equal runs of whole PSDR_SHARD_CHUNK-lane chunks.  Every lane's random stream is a function of (seed, lane), so both partitions add up to
                                     the same frame; the reference's nearest hook is batch_pix (integrator.cpp:139-176) */
} psdr_render_args;

/* counters of the instrumented build: filled by psdr_hip_render_*_counted */
typedef struct psdr_counters {
    uint64_t rays, nodes_visited, tris_tested, shaded_hits;
} psdr_counters;

/* Threading / streams: every entry point that launches kernels takes the stream to launch on.  Calls on one stream are ordered by
 * the stream.  The launches of ONE scene share device scratch (work-queue ring, traversal-stack overflow, adjoint records), so calls
 * on the same scene from different streams or host threads are serialised by the library: a call first makes its stream wait for
 * the completion event of the scene's previous call (no host synchronisation).  Different scenes run concurrently. */
const char *psdr_hip_last_error(void);
int psdr_hip_abi_version(void);
int psdr_hip_device_count(void);
int psdr_hip_set_device(int device);

/* replaces Scene_OptiX::configure (GAS build) + the jit uploads of Scene::configure */
int psdr_hip_scene_create(const psdr_scene_snapshot *snapshot, psdr_hip_scene **out);
int psdr_hip_scene_destroy(psdr_hip_scene *scene);
/* Scene::configure() AGAIN on a scene that already has a device copy (the reference's per-step pattern, README.md:87-106: set_transform ->
 * configure -> renderD -> backward; the reference re-uploads every array and rebuilds its OptiX GAS each time, src/scene/scene.cpp:311-599,
 * src/scene/scene_optix.cpp:265-332).  The new snapshot replaces the old one inside the same handle (guiding grids built from the old state
 * are the caller's to rebuild, as PathTracer::preprocess_secondary_edges is in the reference):
 *   - same triangle count: the tree and the triangle order are kept.  Moved triangles -> the 4-wide tree is refitted on the device (bottom-up, the
 *     builder's own box padding and quantisation), and built again only when the refitted tree's SAH cost exceeds 1.4 x the cost it was built with;
 *   - another triangle count: the tree is built again (host threads).
 * `same`: PSDR_SAME_* bits by which the caller vouches that a part of the snapshot holds the values of the previous create / update of this handle -
 * such a part is neither rewritten nor sent.  0 is always correct (everything is rewritten; the tree is still kept and refitted).  The call waits for
 * the scene's pending render calls and returns when the device copy is complete; `info` (may be NULL) says what was done. */
#define PSDR_SAME_TRIANGLES     1u   /* every array of psdr_triangles except the d_ (tangent) arrays */
#define PSDR_SAME_TRI_TANGENTS  2u   /* the d_ arrays of psdr_triangles */
#define PSDR_SAME_SEC_EDGES     4u   /* psdr_sec_edges, tangents included */
#define PSDR_SAME_PRIM_EDGES    8u   /* the primary-edge arrays of every sensor, tangents included */
#define PSDR_SAME_ENV_TEXELS   16u   /* psdr_envmap_rec.radiance, cell_pmf, cell_cmf (not its transform, scale or tangents) */
#define PSDR_SAME_ENV_TANGENT  32u   /* psdr_envmap_rec.d_radiance */
#define PSDR_SAME_BITMAPS      64u   /* the texel / per-vertex arrays of every psdr_bsdf_rec and their tangents (not the constants or uv transforms) */
typedef struct psdr_update_info {
    int32_t tree;                    /* 0 kept as it was, 1 refitted on the device, 2 built */
    int32_t reallocated;             /* device allocations whose size changed */
    int64_t bytes_uploaded;
    double sah_cost, sah_cost_built; /* SAH cost of the tree now / when it was built (sum over child boxes of half-area x triangles-or-1, over the root's half-area) */
    double ms_tree, ms_fill, ms_upload, ms_total;   /* host wall clock: tree build / refit, writing the host copy, copies issued, whole call */
} psdr_update_info;
#define PSDR_HIP_NEED_ROWS 2          /* psdr_hip_scene_update: snapshot->rows_valid == 0 and this update cannot do without the rows (see psdr_scene_snapshot.rows_valid) */
int psdr_hip_scene_update(psdr_hip_scene *scene, const psdr_scene_snapshot *snapshot, uint32_t same, psdr_update_info *info);
/* what the last create / update of this handle did */
int psdr_hip_scene_last_update(const psdr_hip_scene *scene, psdr_update_info *info);
/* test aid (synchronises, downloads the tree): number of places where the device tree is NOT a bounding hierarchy of the device triangles - a child
 * box that does not contain the padded box of a triangle below it, a triangle slot under no or several leaves.  0 for brute-force scenes. */
int psdr_hip_scene_check_tree(const psdr_hip_scene *scene, int64_t *violations);
/* test aid (synchronises, downloads the sections): 32-bit words of the device's triangle rows (traversal, shading, tangent) and secondary-edge rows that differ from the rows
 * the host path would write from `snapshot` - 0 when the kernels that compute a moved mesh's rows on the device (psdr_mesh_geometry) produced the host's bits */
int psdr_hip_scene_check_rows(const psdr_hip_scene *scene, const psdr_scene_snapshot *snapshot, int64_t *mismatches);
/* BVH statistics for DESIGN/bench: nodes, leaves, max depth, bytes resident in LDS per workgroup */
int psdr_hip_scene_stats(const psdr_hip_scene *scene, int32_t *n_nodes, int32_t *n_leaves, int32_t *max_depth, int32_t *lds_bytes);
/* The live-pixel mask of a sensor (HOST bits[(width*height + 31) / 32], bit y*width + x; either pointer may be NULL): a pixel is dead when
 * no ray through its square can reach a triangle - the conservative screen-space coverage of the scene, built at scene creation.  The
 * interior term of every sample of a dead pixel is exactly zero (Integrator::__render, src/integrator/integrator.cpp:103-131: the
 * camera ray misses), and psdr_hip_render_c / _d_fwd skip such samples before seeding them.  All ones: no mask (environment-lit
 * scenes, a triangle across the camera plane, 2^31 lanes or more, less than a quarter of the frame dead, PSDR_NO_LIVE_MASK=1 in the
 * environment). */
int psdr_hip_scene_live_pixels(const psdr_hip_scene *scene, int32_t sensor_id, uint32_t *bits, int64_t *n_live);
/* bytes of one node of the 4-wide BVH this build walks (64: quantised child boxes, csrc/hip/bvh.h) - the unit of the algorithmic
 * bytes bench.py prices a node visit at */
int psdr_hip_bvh_node_bytes(void);

/* Scene::ray_intersect<false> for a batch of rays (reference src/scene/scene.cpp:612-806; bound to Python as
 * Scene.unit_ray_intersect, src/psdr.cpp:404).  Device arrays o[n*3], d[n*3] -> out[n*24]:
 * {valid, mesh id (-1 = miss), t, J, p.xyz, n.xyz (geometric), sh_frame.s.xyz, .t.xyz, .n.xyz, wi.xyz (local), uv.xy} */
#define PSDR_ITS_STRIDE 24
int psdr_hip_ray_intersect(const psdr_hip_scene *scene, int32_t n, const float *o, const float *d, float *out, void *stream);
/* closest hit for a batch of rays (device arrays o[n*3], d[n*3] -> tri[n], uv[n*2], t[n]); parity aid */
int psdr_hip_trace(const psdr_hip_scene *scene, int32_t n, const float *o, const float *d,
                   int32_t *out_tri, float *out_uv, float *out_t, void *stream);
/* same query, two rays per lane through the two-ray tracer the path kernels use (rays 2i, 2i+1 share a lane) */
int psdr_hip_trace_pairs(const psdr_hip_scene *scene, int32_t n, const float *o, const float *d,
                         int32_t *out_tri, float *out_uv, float *out_t, void *stream);

/* EnvironmentMap::sample_position / sample_position_pdf alone (envmap.cpp:91-166; device arrays; parity aids):
 * ref_p[n*3], s2[n*2] -> p[n*3] on the scene box, its normal n[n*3], pdf[n];  ref_p, p, nrm -> pdf[n] */
int psdr_hip_env_sample(const psdr_hip_scene *scene, int32_t n, const float *ref_p, const float *s2,
                        float *out_p, float *out_n, float *out_pdf, void *stream);
int psdr_hip_env_pdf(const psdr_hip_scene *scene, int32_t n, const float *ref_p, const float *p, const float *nrm,
                     float *out_pdf, void *stream);

/* EnvironmentMap::configure, the cell masses of its HyperCubeDistribution2f (reference src/emitter/envmap.cpp:17-44,
 * src/core/cube_distrb.cpp:22-29: luminance x sin(theta) at the centre of each of the 2(W-1) x 2(H-1) cells, which the
 * reference evaluates on the device in every Scene::configure).  HOST arrays: texels[height*width*3] ->
 * mass[2(width-1) * 2(height-1)], cell index = cx * 2(height-1) + cy.  The prefix sums stay with the caller. */
int psdr_hip_env_cell_masses(const float *texels, int32_t width, int32_t height, float *mass);
/* the same with m_radiance's uv transform uv_xf[4] = rotate, scale, translate.x, translate.y (psdr_envmap_rec.radiance_xf;
 * NULL = identity): envmap.cpp:28-31 looks the cell centres up through Bitmap::eval, bitmap.cpp:64-86 */
int psdr_hip_env_cell_masses_xf(const float *texels, int32_t width, int32_t height, const float *uv_xf, float *mass);

/* Integrator::renderC: out_rgb is [n_pixels*3] float32, pixel-interleaved, pixel = y*W + x */
int psdr_hip_render_c(const psdr_hip_scene *scene, const psdr_render_args *args, float *out_rgb, void *stream);
/* Integrator::renderD + forward derivative: out_rgb = image, out_drgb = d image / d theta */
int psdr_hip_render_d_fwd(const psdr_hip_scene *scene, const psdr_render_args *args,
                          float *out_rgb, float *out_drgb, void *stream);
/* Reverse mode of renderD (the reference's drjit.backward through Integrator::renderD, README.md:102-106):
 * given d_rgb = d loss / d image ([n_pixels*3], device), accumulate the adjoints of the snapshot quantities
 * the image depends on.  All buffers are DEVICE pointers owned by the caller; rows follow the snapshot order.
 * The host chains them to vertices / transforms / colours / camera pose.  With args->pix_ids (batch rendering) d_rgb is
 * [n_pix*3] and only the interior term exists, as in the forward path.  g_bsdf / g_mat / psdr_hip_scene_tex_layout rows cover
 * every entry of psdr_scene_snapshot.bsdfs, including the records nested in normal maps. */
typedef struct psdr_grads {
    float *g_triangles;    /* [n_triangles*22] rows [p0 e1 e2 n0 n1 n2 face_normal face_area] */
    float *g_bsdf;         /* [n_bsdfs*3] reflectance */
    float *g_emitter;      /* [n_emitters*3] radiance */
    float *g_sec_edges;    /* [n_sec_edges*6] (p0, e1); required when sppse > 0 and the term is requested */
    float *g_prim_edges;   /* [n_primary_edges(sensor)*4] sample-space (p0.xy, p1.xy); required when sppe > 0 */
    /* optional filter (autograd's requires_grad at the level of the snapshot): the interior adjoint skips what is not wanted */
    const uint8_t *mesh_filter;   /* DEVICE [n_meshes]: 1 = triangle rows of this mesh are wanted; NULL = all meshes */
    int32_t skip_bsdf, skip_emitter;   /* 1 = g_bsdf / g_emitter are not wanted (left untouched by the interior term) */
    /* texel adjoints of the bitmap parameters (drjit.backward into Bitmap3fD / Bitmap1fD data): DEVICE [total] floats laid out
     * as psdr_hip_scene_tex_layout reports, or NULL = not wanted */
    float *g_tex;
    /* adjoint of the sensor's to_world through the camera rays of the interior and secondary-edge terms (DEVICE [16], row major,
     * rows 0-2 filled), or NULL = not wanted.  The camera's share of the primary-edge term arrives through g_prim_edges (sample-space edge
     * endpoints = world_to_sample . vertex: the host chains both). */
    float *g_camera;
    /* adjoints of the environment map: its texels (DEVICE [height*width*3]) and its scale (DEVICE [1]); NULL = not wanted */
    float *g_env, *g_env_scale;
    /* adjoints of the constant parameters of the GGX BSDFs, DEVICE [n_bsdfs*16], one row per BSDF, or NULL = not wanted:
     *   Microfacet      [specular rgb, roughness]
     *   RoughConductor  [alpha_u, alpha_v, eta rgb, k rgb, specular_reflectance rgb]
     *   RoughDielectric [alpha_u, alpha_v, eta (= intIOR / extIOR)]                     (the diffuse colour stays in g_bsdf) */
    float *g_mat;
    /* adjoint of the environment map's from_world (DEVICE [16], row major; the 3x3 block that maps world directions), or NULL */
    float *g_env_from_world;
    /* adjoints of the bitmaps' uv transforms (psdr_bsdf_rec.tex_xf, psdr_emitter_rec's radiance transform): DEVICE
     * [(3*n_bsdfs + 1)*4], row 3*b + k = [rotate, scale, translate.x, translate.y] of bitmap k of BSDF b, last row = the environment
     * map's; or NULL = not wanted.  Every lookup of the interior term adds its share in the same pass that fills g_tex / g_env:
     * give g_tex beside it for the rows of BSDF bitmaps and g_env for the environment map's row (the record-and-probe form visits a
     * lookup only for a buffer that is wanted).  The edge terms do not see the transforms (their values are
     * detached, src/integrator/integrator.cpp:179-198, path.cpp:171-270). */
    float *g_uv_xf;
    /* primary-edge term: DEVICE [n_primary_edges(sensor)], 1 = the row of g_prim_edges is wanted; an edge sample of an unwanted row is not traced (its
     * whole contribution is that row).  NULL = all rows.  (A caller that differentiates one mesh and not the camera wants the rows of that mesh's edges.) */
    const uint8_t *prim_edge_filter;
} psdr_grads;
/* offsets[3*n_bsdfs] (HOST): float offset of the texel block of BSDF b's bitmap k (0 reflectance / diffuse reflectance rgb,
 * 1 specular reflectance rgb, 2 roughness; RoughConductor: eta, k, alpha; NormalMap: the map; MicrofacetPerVertex: the per-vertex
 * diffuse / specular / roughness arrays) inside psdr_grads.g_tex, same row-major layout as the bitmap; -1 = constant.
 * *total = floats to allocate (0 = the scene has no bitmap parameter).  Either pointer may be NULL. */
int psdr_hip_scene_tex_layout(const psdr_hip_scene *scene, int64_t *offsets, int64_t *total);
int psdr_hip_render_d_bwd(const psdr_hip_scene *scene, const psdr_render_args *args, const float *d_rgb,
                          const psdr_grads *grads, void *stream);
/* same kernels with traversal counters enabled (slower; counters is a HOST struct, call synchronises) */
int psdr_hip_render_c_counted(const psdr_hip_scene *scene, const psdr_render_args *args, float *out_rgb,
                              psdr_counters *counters, void *stream);
int psdr_hip_render_d_fwd_counted(const psdr_hip_scene *scene, const psdr_render_args *args, float *out_rgb,
                                  float *out_drgb, psdr_counters *counters, void *stream);
/* per-lane radiance of the interior term, out [n_lanes*3] (device); parity aid */
int psdr_hip_li_lanes(const psdr_hip_scene *scene, const psdr_render_args *args, int64_t lane_begin, int64_t lane_end,
                      float *out, void *stream);

/* PathTracer::preprocess_secondary_edges: reso = {rx, ry, rz, samples per cell} */
int psdr_hip_guiding_build(const psdr_hip_scene *scene, int32_t sensor_id, int32_t max_depth, const int32_t reso[4],
                           int32_t nrounds, int32_t seed, psdr_hip_guiding **out, void *stream);
int psdr_hip_guiding_mass(const psdr_hip_guiding *g, float *out_host, int32_t cap);   /* returns n_cells via cap check */
int psdr_hip_guiding_num_cells(const psdr_hip_guiding *g);
int psdr_hip_guiding_destroy(psdr_hip_guiding *g);

/* sampler building blocks (host side, bit-exact with the kernels): known-answer tests */
uint64_t psdr_hip_tea64(uint64_t v0, uint64_t v1);
int psdr_hip_sampler_floats(uint64_t seed_value, uint64_t lane, uint64_t skip, int32_t n, float *out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PSDR_HIP_H */
