// ORACLE — TEST INFRASTRUCTURE ONLY (see num.h header).
//
// scene.h — the configured scene: what Scene::configure() (reference src/scene/scene.cpp:311-601)
// leaves behind, restated as plain host arrays of (value, tangent) pairs.
#pragma once
#include <algorithm>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "num.h"
#include "../oracle.h"

namespace orc {

// reference include/psdr/constants.h:12-17
constexpr float Epsilon = 1e-5f, RayEpsilon = 1e-3f, ShadowEpsilon = 1e-3f, EdgeEpsilon = 1e-5f;
constexpr float Pi = 3.14159265358979323846f, InvPi = 0.31830988618379067154f;
constexpr float TwoPi = 6.28318530717958647692f, InvTwoPi = 0.15915494309189533577f;   // constants.h:21-23
constexpr float TraceTMax = 100000000.f;   // scene_optix.cpp:376

// sequential fp32 sum (drjit::sum is an fp32 device reduction whose order is unspecified)
inline float sum_f32(const std::vector<float> &v) { float s = 0.f; for (float x : v) s += x; return s; }

// ---------------------------------------------------------------- DiscreteDistribution
// reference include/psdr/core/pmf.h:12-38 (compute_cdf), src/core/pmf.cpp:6-51
struct Distrb {
    int size = 0;
    float sum = 0.f;
    std::vector<float> pmf, cmf;
    void init(const std::vector<float> &p) {
        if (p.empty()) throw std::runtime_error("DiscreteDistribution: empty distribution!");
        size = (int) p.size();
        pmf = p;
        sum = sum_f32(p);
        cmf.resize(size);
        double acc = 0.0;                       // host double accumulation, pmf.h:21-25
        for (int i = 0; i < size; ++i) {
            if (p[i] < 0.f) throw std::runtime_error("DiscreteDistribution: entries must be non-negative!");
            acc += (double) p[i];
            cmf[i] = (float) acc;
        }
    }
    // pmf.cpp:26-45.  size==1 returns without touching the sample (pmf.cpp:32-34).
    int sample_reuse(float &s, float &pdf) const {
        if (size == 1) { pdf = 1.f; return 0; }
        s *= sum;
        // binary_search over [0, size-1): first i with !(cmf[i] < s), else size-1
        int idx = (int) (std::lower_bound(cmf.begin(), cmf.begin() + (size - 1), s) - cmf.begin());
        if (idx > 0) s -= cmf[idx - 1];
        float p = pmf[idx];
        if (p > 0.f) s /= p;
        s = std::min(std::max(s, 0.f), 1.f);
        pdf = p / sum;
        return idx;
    }
};

// ---------------------------------------------------------------- records
struct Tri {                       // TriangleInfo row, include/psdr/types.h:162-175
    V3d p0, e1, e2, n0, n1, n2, fn;
    Dual area;
    int fi[3];
    V2f uv[3];                     // Scene::m_triangle_uv (zeros when the mesh has no uv, scene.cpp:530)
    bool flat;                     // Scene::m_triangle_face_normals
    int mesh;
};
struct MeshEdge { int v0, v1, f0, f1, opp; };            // Mesh::m_edge_indices, mesh.h:112-115
struct SecEdge { V3d p0, e1, n0, n1, p2; bool is_boundary; };   // include/psdr/edge/edge.h:49-68
struct PrimEdge { V2d p0, p1; V2f normal; float length; };      // edge.h:27-40

struct MeshC {
    int face_offset = 0, n_faces = 0, n_vertices = 0;
    int bsdf = -1, emitter = -1;
    bool use_face_normals = false, enable_edges = true, has_uv = false;
    std::vector<V3d> verts;                 // world space, m_vertex_positions
    std::vector<int> faces, face_uvs;
    std::vector<MeshEdge> edges;
    float total_area = 0.f, inv_total_area = 0.f;
    Distrb face_distrb;
};
struct BsdfC { int type; V3d reflectance; bool two_sided; int tex_w = 0, tex_h = 0; std::vector<float> tex, d_tex;
               V3d specular; Dual roughness;       // type 1 = Microfacet: reflectance is its diffuse reflectance
               Dual alpha_u, alpha_v; V3d eta, k;
               int spec_w = 0, spec_h = 0, rough_w = 0, rough_h = 0;       // Microfacet bitmap parameters (rgb / 1 channel)
               std::vector<float> spec_tex, d_spec_tex, rough_tex, d_rough_tex;
               // type 4 = MicrofacetPerVertex: [n*3], [n*3], [n] values per (mesh-local) vertex, with optional tangents
               std::vector<float> pv_spec, pv_diff, pv_rough, d_pv_spec, d_pv_diff, d_pv_rough;
               // uv transform of the three bitmap slots (tex, spec_tex, rough_tex): Bitmap::m_rot, m_scale, m_trans.x, m_trans.y (bitmap.h:37-39)
               Dual uv_xf[3][4] = {{Dual(0.f), Dual(1.f), Dual(0.f), Dual(0.f)}, {Dual(0.f), Dual(1.f), Dual(0.f), Dual(0.f)}, {Dual(0.f), Dual(1.f), Dual(0.f), Dual(0.f)}};
               int nested = -1; };                // type 5 = NormalMap: index of the nested BSDF; reflectance / tex = the normal map   // type 2 = RoughConductor (specular = specular_reflectance)   // tex: Bitmap3fD texels when textured
// type 0 = AreaLight (area.h), 1 = EnvironmentMap (envmap.h); an envmap's mesh is the bounding cube scene.cpp:442-480 adds
struct EmitterC { V3d radiance; int mesh = -1; float sampling_weight = 1.f; int type = 0; };

struct EnvmapC {
    int width = 0, height = 0;               // m_radiance.m_resolution
    std::vector<float> data, d_data;         // [height*width*3], row-major rgb (+ optional tangent of the texels)
    Dual scale = Dual(1.f);                  // m_scale (FloatD)
    Dual uv_xf[4] = {Dual(0.f), Dual(1.f), Dual(0.f), Dual(0.f)};   // m_radiance.m_rot, m_scale, m_trans (bitmap.h:37-39)
    M4d to_world, from_world;                // envmap.cpp:41-42
    V3f lower, upper;                        // scene AABB + margin (scene.cpp:436-440)
    // HyperCubeDistribution2f m_cell_distrb
    int reso[2] = {0, 0};
    int num_cells = 0;
    float unit[2] = {0.f, 0.f};
    Distrb cell_distrb;
};


struct CameraC {                   // PerspectiveCamera, src/sensor/perspective.cpp:10-152
    M4d to_world, world_to_sample, sample_to_world;
    M4f camera_to_sample, sample_to_camera;
    V3d pos, dir;
    float inv_area = 0.f;
    bool enable_edges = false;
    bool orthographic = false;     // OrthographicCamera (orthographic.cpp): parallel rays from the near plane
    std::vector<PrimEdge> edges;
    Distrb edge_distrb;
};

struct BvhNode { float lo[3], hi[3]; int left, right, first, count; };   // oracle-private BVH

struct Scene {
    int width = 0, height = 0, spp = 0, sppe = 0, sppse = 0;
    std::vector<MeshC> meshes;
    std::vector<BsdfC> bsdfs;
    std::vector<EmitterC> emitters;
    int env_emitter = -1;              // index of the EnvironmentMap in emitters (Scene::m_emitter_env), -1 = none
    EnvmapC env;
    V3f lower, upper;                  // Scene::m_lower / m_upper
    std::vector<CameraC> cameras;
    std::vector<Tri> tris;
    std::vector<SecEdge> sec_edges;
    Distrb sec_edge_distrb, emitters_distrb;
    std::vector<BvhNode> bvh;
    std::vector<int> bvh_tris;
    bool use_bvh = false;
    int direct_mis = -1;               // >= 0: Li is DirectIntegrator(mis) (direct.cpp), -1: PathTracer
    // first-hit integrators: FieldExtractionIntegrator (field.cpp) 0 silhouette 1 position 2 depth 3 geoNormal 4 shNormal 5 uv
    // 6 bsdf 7 segmentation, CollocatedIntegrator (collocated.cpp) 8; -1 = none.  field_object: mesh index filter or -1
    int field = -1, field_object = -1;
    Dual intensity = Dual(1.f);
};

Scene *configure_scene(const orc_scene_desc &d, const int *active, int n_active);
void build_bvh(Scene &s);

struct Hit { int tri = -1; float u = 0.f, v = 0.f, t = 0.f; };
Hit trace_closest(const Scene &s, const V3f &o, const V3f &d, bool use_bvh);

} // namespace orc
