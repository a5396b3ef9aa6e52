// scene_host.h — host-side scene model behind the psdr_jit Python surface.
//
// Mirrors the reference classes that the hot path needs (same member names / semantics):
//   RenderOption (include/psdr/types.h:217-228), Object (include/psdr/object.h), Mesh (shape/mesh.h),
//   Diffuse (bsdf/diffuse.h), AreaLight (emitter/area.h), PerspectiveCamera (sensor/perspective.h),
//   Scene (scene/scene.h), Integrator / PathTracer (integrator/integrator.h, path.h).
// drjit arrays become plain host vectors; every differentiable member carries one forward tangent.
// Scene::configure() produces the psdr_scene_snapshot of include/psdr_hip.h and hands it to
// libpsdr_hip.so; rendering is entirely behind that C ABI.
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/psdr_hip.h"
#include "hnum.h"

namespace psdr_host {

using M16 = std::array<float, 16>;
inline M16 identity16() { return {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; }
inline M16 zeros16() { M16 z{}; return z; }

// reference include/misc/Exception.h:7-40 (text carries file/line); surfaces as RuntimeError in Python
struct Exception : std::runtime_error { using std::runtime_error::runtime_error; };
[[noreturn]] void throw_assert(const char *cond, const char *file, int line, const std::string &msg);
#define PSDR_ASSERT_MSG(cond, msg) do { if (!(cond)) ::psdr_host::throw_assert(#cond, __FILE__, __LINE__, (msg)); } while (0)
#define PSDR_ASSERT(cond) PSDR_ASSERT_MSG(cond, "")

struct Object {
    virtual ~Object() {}
    virtual std::string type_name() const = 0;
    virtual std::string to_string() const { return type_name(); }
    void log(const std::string &msg) const;
    std::string m_id;
};

struct RenderOption {
    RenderOption() : width(128), height(128), spp(1), sppe(0), sppse(0), log_level(1) {}
    RenderOption(int w, int h, int s) : width(w), height(h), spp(s), sppe(s), sppse(s), log_level(1) {}
    RenderOption(int w, int h, int s1, int s2) : width(w), height(h), spp(s1), sppe(s2), sppse(s2), log_level(1) {}
    RenderOption(int w, int h, int s1, int s2, int s3) : width(w), height(h), spp(s1), sppe(s2), sppse(s3), log_level(1) {}
    int width, height, spp, sppe, sppse, log_level;
};

// DiscreteDistribution (host part), reference include/psdr/core/pmf.h:12-38, src/core/pmf.cpp:6-15
struct Distrb {
    int size = 0;
    float sum = 0.f;
    std::vector<float> pmf, cmf;
    void init(const std::vector<float> &p);
};

struct BSDF : Object {
    bool m_twoSide = false;
    virtual bool anisotropic() const = 0;
    // uv transform of the BSDF's (up to three) bitmap parameters, slot order of psdr_bsdf_rec (0 reflectance / diffuse / eta / normal map,
    // 1 specular / k, 2 roughness / alpha): Bitmap::m_rot, m_scale, m_trans.x, m_trans.y (reference bitmap.h:37-39) + forward tangents
    float uv_xf[3][4] = {{0, 1, 0, 0}, {0, 1, 0, 0}, {0, 1, 0, 0}}, d_uv_xf[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
};
struct Diffuse : BSDF {
    Diffuse() : reflectance{0.5f, 0.5f, 0.5f}, d_reflectance{0, 0, 0} {}
    explicit Diffuse(const std::array<float, 3> &r) : reflectance(r), d_reflectance{0, 0, 0} {}
    std::string type_name() const override { return "Diffuse"; }
    std::string to_string() const override { return std::string("Diffuse[id=") + m_id + "]"; }
    bool anisotropic() const override { return false; }
    std::array<float, 3> reflectance, d_reflectance;   // Bitmap3fD with 1x1 resolution (bitmap.cpp:54-59)
    // Bitmap3fD with a larger resolution: [tex_h*tex_w*3] row-major rgb texels (+ their forward tangent); tex_w == 0: none
    int tex_w = 0, tex_h = 0;
    std::vector<float> tex, d_tex;
};

// a bitmap parameter with a resolution above 1x1: [h*w*ch] row-major texels (+ their forward tangent); w == 0: constant
struct ParamTex { int w = 0, h = 0; std::vector<float> v, d; };

// Microfacet, reference include/psdr/bsdf/microfacet.h: three Bitmap parameters, each a constant (1x1) or a texture
struct Microfacet : BSDF {
    Microfacet() {}
    Microfacet(const std::array<float, 3> &spec, const std::array<float, 3> &diff, float rough) : specular(spec), diffuse(diff), roughness(rough) {}
    std::string type_name() const override { return "Microfacet"; }
    std::string to_string() const override { return std::string("Microfacet[id=") + m_id + "]"; }
    bool anisotropic() const override { return false; }
    std::array<float, 3> specular{0.04f, 0.04f, 0.04f}, diffuse{0.5f, 0.5f, 0.5f}, d_specular{0, 0, 0}, d_diffuse{0, 0, 0};
    float roughness = 0.8f, d_roughness = 0.f;
    ParamTex specular_tex, diffuse_tex, roughness_tex;      // rgb, rgb, one channel
};

// RoughConductor, reference include/psdr/bsdf/roughconductor.h (constant parameters)
struct RoughConductor : BSDF {
    RoughConductor() {}
    std::string type_name() const override { return "RoughConductor"; }
    std::string to_string() const override { return std::string("RoughConductor[id=") + m_id + "]"; }
    bool anisotropic() const override { return alpha_u != alpha_v; }
    float alpha_u = 0.1f, alpha_v = 0.1f, d_alpha_u = 0.f, d_alpha_v = 0.f;
    std::array<float, 3> eta{0, 0, 0}, k{1, 1, 1}, specular{1, 1, 1}, d_eta{0, 0, 0}, d_k{0, 0, 0}, d_specular{0, 0, 0};
    ParamTex eta_tex, k_tex, alpha_tex;     // bitmaps above 1x1: rgb, rgb, one channel (the alpha map serves both axes, as in the XML loader)
};

// MicrofacetPerVertex, reference include/psdr/bsdf/microfacet_pv.h: one value per mesh-local vertex
struct MicrofacetPerVertex : BSDF {
    MicrofacetPerVertex() {}
    MicrofacetPerVertex(const std::vector<float> &s, const std::vector<float> &d, const std::vector<float> &r) : specular(s), diffuse(d), roughness(r) {}
    std::string type_name() const override { return "MicrofacetPerVertex"; }
    std::string to_string() const override { return std::string("MicrofacetPerVertex[id=") + m_id + "]"; }
    bool anisotropic() const override { return false; }
    std::vector<float> specular, diffuse, roughness, d_specular, d_diffuse, d_roughness;      // [n*3], [n*3], [n] (+ forward tangents)
};

// NormalMap, reference include/psdr/bsdf/normalmap.h: a normal map (Bitmap3fD: constant or texture) over a nested BSDF it owns
struct NormalMap : BSDF {
    NormalMap() {}
    explicit NormalMap(const std::array<float, 3> &n) : normal(n) {}
    NormalMap(const NormalMap &o) : BSDF(o), normal(o.normal), d_normal(o.d_normal), tex_w(o.tex_w), tex_h(o.tex_h), tex(o.tex), d_tex(o.d_tex), m_bsdf(nullptr) {}
    NormalMap &operator=(const NormalMap &) = delete;
    ~NormalMap() override;
    void set_nested(const BSDF *b);                    // takes a copy
    std::string type_name() const override { return "NormalMap"; }
    std::string to_string() const override { return std::string("NormalMap[id=") + m_id + "]"; }
    bool anisotropic() const override { return false; }
    std::array<float, 3> normal{0.5f, 0.5f, 1.f}, d_normal{0, 0, 0};       // Bitmap3fD with 1x1 resolution
    int tex_w = 0, tex_h = 0;
    std::vector<float> tex, d_tex;
    BSDF *m_bsdf = nullptr;
};

// RoughDielectric, reference include/psdr/bsdf/roughdielectric.h (constant alpha; m_eta = intIOR / extIOR, m_inv_eta = extIOR / intIOR)
struct RoughDielectric : BSDF {
    RoughDielectric() { eta = 1.5f / 1.0f; inv_eta = 1.f / eta; }
    RoughDielectric(float intIOR, float extIOR) : eta(intIOR / extIOR), inv_eta(extIOR / intIOR) {}
    std::string type_name() const override { return "RoughDielectric"; }
    std::string to_string() const override { return std::string("RoughDielectric[id=") + m_id + "]"; }
    bool anisotropic() const override { return false; }
    float alpha_u = 0.1f, alpha_v = 0.1f, d_alpha_u = 0.f, d_alpha_v = 0.f;
    float eta, inv_eta, d_eta = 0.f, d_inv_eta = 0.f;
    ParamTex alpha_tex;                     // alpha bitmap above 1x1 (both axes)
};

struct Mesh;
struct Emitter : Object { float m_sampling_weight = 1.f; bool m_ready = false; };
struct AreaLight : Emitter {
    explicit AreaLight(const std::array<float, 3> &r) : radiance(r), d_radiance{0, 0, 0} {}
    std::string type_name() const override { return "AreaLight"; }
    std::array<float, 3> radiance, d_radiance;
    const Mesh *m_mesh = nullptr;
};

// EnvironmentMap, reference include/psdr/emitter/envmap.h + src/emitter/envmap.cpp.  The radiance is a lat-long
// Bitmap3fD given as an array (the reference's constructor reads an OpenEXR file; the Python layer does the file I/O).
struct EnvironmentMap : Emitter {
    EnvironmentMap() {}
    EnvironmentMap(int w, int h, const std::vector<float> &rgb) : width(w), height(h), data(rgb) {}
    std::string type_name() const override { return "EnvironmentMap"; }
    std::string to_string() const override;
    void set_transform(const M16 &mat) { to_world_left = mat; m_ready = false; }      // envmap.h:19-22
    void configure(bool on_device = false);                                            // envmap.cpp:17-44 (on_device: cell masses by psdr_hip_env_cell_masses)
    int width = 0, height = 0;
    std::vector<float> data, d_data;                                                   // [height*width*3] row-major rgb (+ forward tangent, may be empty)
    float scale = 1.f, d_scale = 0.f;
    float uv_xf[4] = {0, 1, 0, 0}, d_uv_xf[4] = {0, 0, 0, 0};                          // m_radiance's m_rot, m_scale, m_trans (bitmap.h:37-39) + forward tangent
    M16 to_world_raw = identity16(), to_world_left = identity16(), d_to_world_left = zeros16();
    // configured state
    float to_world[16], from_world[16], d_to_world[16], d_from_world[16];
    float lower[3] = {0, 0, 0}, upper[3] = {0, 0, 0};
    int reso[2] = {0, 0};
    Distrb cell_distrb;
    bool m_cells_dirty = true;                                                         // texels changed since cell_distrb was built
    int m_bound_mesh_id = -1;
};

struct Transformable {
    M16 to_world_raw = identity16(), to_world_left = identity16(), to_world_right = identity16();
    M16 d_to_world_raw = zeros16(), d_to_world_left = zeros16(), d_to_world_right = zeros16();
    // Mesh::set_transform / append_transform, reference mesh.h:26-42 (same on Sensor, sensor.h:34-48)
    void set_transform(const M16 &mat, const M16 &dmat, bool set_left);
    void append_transform(const M16 &mat, const M16 &dmat, bool append_left);
    DM4 to_world() const;      // left * raw * right
};

struct MeshEdge { int v0, v1, f0, f1, opp; };

struct Mesh : Object, Transformable {
    std::string type_name() const override { return "Mesh"; }
    std::string to_string() const override;
    void load(const std::string &fname, bool verbose = false);     // OBJ (triangulates polygons)
    void load_raw(const std::vector<float> &v, const std::vector<int> &f, const std::vector<float> &uv, const std::vector<int> &fuv, bool verbose = false);
    void configure();
    void dump(const std::string &fname, bool raw) const;

    int m_mesh_id = -1;
    bool m_ready = false, m_use_face_normals = false, m_has_uv = false, m_enable_edges = true;
    int m_bsdf_id = -1, m_emitter_id = -1;
    const BSDF *m_bsdf = nullptr;
    const Emitter *m_emitter = nullptr;
    int m_num_vertices = 0, m_num_faces = 0;
    std::vector<float> vertex_positions_raw, d_vertex_positions_raw;   // [nv*3]
    std::vector<float> vertex_uv;                                       // [nuv*2]
    std::vector<int> face_indices, face_uv_indices;                     // [nf*3]
    std::vector<MeshEdge> edges;
    // results of configure()
    std::vector<float> vertex_positions, d_vertex_positions;            // world space
    std::vector<float> vertex_normals_raw;
    float m_total_area = 0.f, m_inv_total_area = 0.f;
    Distrb face_distrb;
    // per-face TriangleInfo rows (value, tangent), 22 floats each: p0 e1 e2 n0 n1 n2 fn area
    std::vector<float> tri, d_tri;
    // LEAN configure (round 6): when the device computes this mesh's rows itself (psdr_mesh_geometry) the host only needs, per face, the first vertex and the unit face
    // normal (primary- and secondary-edge selection) and the area (face distribution) - face_p0n [nf*6], values only, the value parts of the same dual-number formulas -
    // and leaves tri / d_tri for ensure_rows() to compute when a consumer asks (a device scene to create or rebuild, Scene::_snapshot, psdr_hip_scene_check_rows)
    bool m_lean = false, rows_valid = false;
    std::vector<float> face_p0n;
    void ensure_rows();
    // configure() compares its inputs with the ones of its previous run and does nothing when they are the same (the reference recomputes and
    // re-uploads every mesh in every Scene::configure, src/scene/scene.cpp:346-371).  What it found: m_geo_version counts the runs in which a value
    // input (raw vertices, a to_world factor, the topology) differed, m_tan_version the runs in which a value or a tangent input differed -
    // Scene::configure_host rewrites the snapshot rows of a mesh, and tells the device library what is unchanged, from these.
    uint64_t m_topo_version = 1;               // bumped by load / load_raw (faces, uv indices, vertex count)
    uint64_t m_geo_version = 0, m_tan_version = 0;
    float m_lower[3] = {0, 0, 0}, m_upper[3] = {0, 0, 0};       // box of the world-space vertices
    float m_tw[16] = {0}, m_d_tw[16] = {0};                     // to_world() of the last configure(), value and tangent (psdr_mesh_geometry: the device computes a moved mesh's rows from it)
private:
    void build_edges();
public:
    void build_vertex_faces();
    const std::vector<int> &vertex_face_begin() const { return vf_begin; }
    const std::vector<int> &vertex_face_item() const { return vf_item; }
    bool vertex_faces_current() const { return vf_topo == m_topo_version && vf_begin.size() == (size_t) m_num_vertices + 1; }
private:
    std::vector<float> cfg_raw, cfg_d_raw;     // inputs of the previous configure()
    M16 cfg_m[3], cfg_dm[3];
    uint64_t cfg_topo = 0;
    bool cfg_valid = false, raw_normals_valid = false;
    std::vector<int> vf_begin, vf_item;        // per vertex: its (corner, face) incidences as corner * num_faces + face, ascending = the order process_mesh sums them in
    uint64_t vf_topo = 0;
};

struct PrimaryEdges {
    std::vector<float> p0, p1, d_p0, d_p1, normal, length;
    std::vector<int> ids;          // (mesh id, v0, v1) per kept edge: lets the host chain gradients back to vertices
    Distrb distrb;
};

struct Scene;
struct Sensor : Object, Transformable {
    virtual void configure(const Scene &scene, bool keep_edges) = 0;
    bool m_enable_edges = false;
    PrimaryEdges m_edges;
    // inputs of the previous configure(): the sensor's own members, the render options and the versions of the scene's meshes it was run against
    // (Scene::configure_host skips a sensor whose inputs are the same; m_edges_version counts the runs that rebuilt the primary edges)
    std::vector<float> cfg_key;
    uint64_t m_edges_version = 0;
};
struct PerspectiveCamera : Sensor {
    PerspectiveCamera(float fov_x, float near_, float far_) : m_fov_x(fov_x), m_near_clip(near_), m_far_clip(far_) {}
    std::string type_name() const override { return "PerspectiveCamera"; }
    void configure(const Scene &scene, bool keep_edges) override;
    float m_fov_x, m_near_clip, m_far_clip;
    int m_width = 0, m_height = 0;
    bool m_orthographic = false;   // OrthographicCamera(near, far), reference src/sensor/orthographic.cpp (same class here: it differs from
                                   // the perspective camera only in the projection matrix and in sample_primary_ray)
    psdr_sensor_rec rec{};     // filled by configure (edge pointers are patched when the snapshot is assembled)
};

// closed form of one of Scene::m_samplers[3] (reference scene.h:76, sampler.h:8-40)
struct SamplerState {
    bool ready = false;
    int64_t sample_count = 0;
    uint64_t seed = 0, skip = 0;
};

struct Scene : Object {
    Scene();
    ~Scene() override;
    std::string type_name() const override { return "Scene"; }

    void add_Sensor(const Sensor *sensor);
    void add_BSDF(const BSDF *bsdf, const std::string &bsdf_id, bool twoSide = false);
    void add_normalmap_BSDF(const NormalMap *bsdf, const Microfacet *micro, const std::string &bsdf_id, bool twoSide = false);
    void add_Mesh(const std::string &fname, const M16 &transform, const std::string &bsdf_id, const Emitter *emitter);
    void add_Mesh(const Mesh *mesh, const std::string &bsdf_id, const Emitter *emitter);
    void add_EnvironmentMap(const EnvironmentMap *emitter);             // scene.cpp:85-105 (one per scene)
    void configure(const std::vector<int> &active_sensor = {});
    void configure_host(const std::vector<int> &active_sensor = {});   // host half (no device needed)
    void upload();                                                      // BVH build + device upload
    void fill_snapshot(psdr_scene_snapshot &sn, bool full = true);
    void ensure_full_snapshot();                                        // computes what a lean configure_host left out (rows of moved meshes, secondary-edge rows)
    void fill_sec_rows();
    int64_t check_device_rows();
    bool is_ready() const;
    size_t get_num_emitters() const { return m_emitters.size(); }
    // Reverse-mode chain rule of the differentiable part of configure() (Mesh::configure / process_mesh, the secondary-edge rows, the primary-edge projection:
    // reference src/shape/mesh.cpp:23-62,317-369, src/sensor/perspective.cpp:130-143 - what drjit.backward walks between the configured arrays and the user's
    // parameters): from the adjoints of the snapshot rows psdr_hip_render_d_bwd returns to the adjoints of each wanted mesh's combined to_world (4 x 4) and raw
    // vertices and of the sensor's world_to_sample.  double arithmetic, host threads.  g_tri [n_triangles * 22], g_sec [n_sec_edges * 6], g_prim [n_primary_edges * 4].
    struct MeshAdjoint { int mesh = -1; double g_to_world[16]; std::vector<double> g_vertices; };
    struct GeometryAdjoint { std::vector<MeshAdjoint> meshes; double g_world_to_sample[16]; };
    GeometryAdjoint chain_geometry(int sensor_id, const float *g_tri, const float *g_sec, const float *g_prim, const std::vector<uint8_t> &want_mesh, bool want_camera,
                                   const double *world_to_sample = nullptr) const;      // world_to_sample: the sensor's matrix in double (default: the configured float one)

    int seed = 0;
    RenderOption m_opts;
    int m_num_sensors = 0, m_num_meshes = 0;
    std::vector<Sensor *> m_sensors;
    std::vector<Emitter *> m_emitters;
    std::vector<BSDF *> m_bsdfs;
    std::vector<Mesh *> m_meshes;
    std::unordered_map<std::string, Object *> m_param_map;
    mutable SamplerState m_samplers[3];
    float m_lower[3], m_upper[3];
    EnvironmentMap *m_emitter_env = nullptr;
    bool m_has_bound_mesh = false;

    // configured snapshot (host arrays the psdr_scene_snapshot points into)
    struct Snapshot {
        std::vector<float> p0, e1, e2, n0, n1, n2, fn, area, uv, d_p0, d_e1, d_e2, d_n0, d_n1, d_n2, d_fn, d_area;
        std::vector<int32_t> mesh_id, face_indices;
        std::vector<uint8_t> flat;
        std::vector<psdr_mesh_rec> meshes;
        std::vector<psdr_bsdf_rec> bsdfs;
        std::vector<psdr_emitter_rec> emitters;
        Distrb emitters_distrb, sec_edge_distrb;
        std::vector<float> face_pmf, face_cmf;
        std::vector<float> se_p0, se_e1, se_n0, se_n1, se_p2, se_d_p0, se_d_e1;
        std::vector<uint8_t> se_boundary;
        std::vector<psdr_sensor_rec> sensors;
        int n_sec_edges = 0;
        psdr_envmap_rec envmap{};
        bool has_envmap = false;
    } snap;
    psdr_hip_scene *m_hip = nullptr;
    bool m_configured = false, m_host_ready = false;
    bool m_device_config = false;      // configure_host() runs inside configure(): its device-side steps are allowed
    // Incremental configure.  configure_host() rewrites only the snapshot rows whose inputs changed and accumulates in m_same the PSDR_SAME_* bits
    // (include/psdr_hip.h) that still hold relative to the snapshot the device has; upload() hands them to psdr_hip_scene_update and resets them.
    uint32_t m_same = 0;
    psdr_update_info m_last_update{};
    double m_ms_host = 0.0;            // wall clock of the host half (configure_host) of the last configure()
    bool m_always_rebuild = false;     // test / measurement aid: destroy and create the device scene in every configure() (what rounds 1-4 did)
private:
    void rebuild_param_map();
    void release_device();
    struct MeshKey { const Mesh *mesh; uint64_t topo; int nf, bsdf, emitter; bool uv, flat, edges; bool operator==(const MeshKey &o) const { return mesh == o.mesh && topo == o.topo && nf == o.nf && bsdf == o.bsdf && emitter == o.emitter && uv == o.uv && flat == o.flat && edges == o.edges; } };
    std::vector<MeshKey> m_snap_keys;                      // the meshes the snapshot's rows were laid out for
    std::vector<uint64_t> m_seen_geo, m_seen_tan;          // per mesh: the versions its snapshot rows hold
    std::vector<uint64_t> m_bits_geo, m_bits_tan;          // per mesh: the versions the PSDR_SAME_* bits were last derived from
    bool m_lean = false;                                   // the configure_host in progress / last run was lean (see scene_host.cpp)
    std::vector<uint64_t> m_up_geo, m_up_tan;              // per mesh: the versions the device scene holds (psdr_mesh_geometry.moved)
    bool m_sec_rows_stale = false;                         // lean configure: the secondary-edge row arrays of the snapshot are behind (their CDF is not)
    bool m_device_rows_ok = false;                         // the last upload went through with the device computing the moved meshes' rows: the next configure_host may be lean
    std::vector<psdr_mesh_geometry> m_geometry;            // upload(): the per-mesh inputs of the device's row computation
    uint64_t m_layout_version = 0;                         // counts the configure_host() runs that laid the triangle rows out anew (another mesh list, topology or flag)
    uint64_t m_sec_layout = ~0ull;
    uint64_t m_sec_geo = ~0ull, m_sec_tan = ~0ull;         // sum of the mesh versions the secondary-edge arrays were built from
    int m_sec_sppse = -1;
    uint64_t m_bitmap_hash = 0, m_env_tan_hash = 0;
    std::vector<uint64_t> m_seen_edges;                    // per sensor: Sensor::m_edges_version at the last upload-relevant configure
};

struct Integrator : Object {
    // out / dout / pix_ids are DEVICE pointers (uintptr_t from the caller's tensors), stream a hipStream_t
    void renderC(const Scene &scene, int sensor_id, int seed, uintptr_t pix_ids, int n_pix, uintptr_t out, uintptr_t stream,
                 int shard_rank, int shard_count) const;
    void renderD(const Scene &scene, int sensor_id, int seed, uintptr_t pix_ids, int n_pix, uintptr_t out, uintptr_t dout, uintptr_t stream,
                 int shard_rank, int shard_count, int terms) const;
    virtual int max_depth() const = 0;
    virtual bool hide_emitters() const = 0;
    virtual const psdr_hip_guiding *guiding(int sensor_id) const { (void) sensor_id; return nullptr; }
    virtual int direct_mis() const { return -1; }                 // >= 0: DirectIntegrator(mis)
    // first-hit integrators (psdr_render_args.field_mode): -1 = none, 0..7 FieldExtractionIntegrator fields, 8 CollocatedIntegrator
    virtual int field() const { return -1; }
    virtual std::string field_object() const { return ""; }
    virtual float intensity(bool tangent) const { return tangent ? 0.f : 1.f; }
    int draws_per_level() const { const int m = direct_mis(); return m == 0 ? 2 : (m == 1 ? 3 : 5); }
    // renderD, primary-edge term: false (default) = an edge sample whose edge point does not move under the installed tangents is not traced - it adds exactly
    // zero to the derivative image (psdr_render_args.skip_static_edges); true = every sample's two paths are traced, as the reference does
    bool m_trace_static_edges = false;
    int m_shard_mode = 0;                // psdr_render_args.shard_mode of the launches (multi-GPU: 0 interleaved 256-lane chunks, 1 contiguous runs = pixel-row tiles)
};

struct PathTracer : Integrator {
    explicit PathTracer(int max_depth = 1);
    ~PathTracer() override;
    std::string type_name() const override { return "PathTracer"; }
    int max_depth() const override { return m_max_depth; }
    bool hide_emitters() const override { return m_hide_emitters; }
    const psdr_hip_guiding *guiding(int sensor_id) const override;
    void preprocess_secondary_edges(const Scene &scene, int sensor_id, const std::array<int, 4> &reso, int nrounds = 1, int seed = 0);
    std::vector<float> guiding_mass(int sensor_id) const;
    bool m_hide_emitters = false;
    int m_max_depth;
    std::vector<psdr_hip_guiding *> m_warpper;
};

// DirectIntegrator(mis), reference src/integrator/direct.cpp: one bounce; mis = 0 emitter sampling only, 1 BSDF sampling
// only, 2 both with MIS (= PathTracer(1)).  The edge terms and the guiding pass are those of PathTracer (direct.cpp:135-277).
struct DirectIntegrator : PathTracer {
    explicit DirectIntegrator(int mis = 1) : PathTracer(1), m_mis(mis) { PSDR_ASSERT(mis >= 0 && mis <= 2); }
    std::string type_name() const override { return "DirectIntegrator"; }
    int direct_mis() const override { return m_mis; }
    int m_mis;
};

// FieldExtractionIntegrator("field [object]"), reference src/integrator/field.cpp:8-31: the first hit's silhouette, position, depth,
// geoNormal, shNormal, uv, bsdf value or segmentation id; optionally restricted to one mesh (by id string or index).
struct FieldExtractionIntegrator : Integrator {
    explicit FieldExtractionIntegrator(const std::string &spec);
    std::string type_name() const override { return "FieldExtractionIntegrator"; }
    int max_depth() const override { return 0; }
    bool hide_emitters() const override { return false; }
    int field() const override { return m_field; }
    std::string field_object() const override { return m_object; }
    std::string m_field_name, m_object;
    int m_field = 0;
};

// CollocatedIntegrator(intensity), reference src/integrator/collocated.cpp: bsdf(wi, wi) / t^2 * intensity at the first hit
struct CollocatedIntegrator : Integrator {
    explicit CollocatedIntegrator(float intensity_) : m_intensity(intensity_) {}
    std::string type_name() const override { return "CollocatedIntegrator"; }
    int max_depth() const override { return 0; }
    bool hide_emitters() const override { return false; }
    int field() const override { return 8; }
    float intensity(bool tangent) const override { return tangent ? d_intensity : m_intensity; }
    float m_intensity, d_intensity = 0.f;
};

int field_object_index(const Scene &scene, const Integrator &it);      // scene_host.cpp: FieldExtractionIntegrator's object filter as a mesh index

} // namespace psdr_host
