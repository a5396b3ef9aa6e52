// ORACLE — TEST INFRASTRUCTURE ONLY (see num.h header).
//
// scene.cpp — Scene::configure restated (reference src/scene/scene.cpp:311-601,
// src/shape/mesh.cpp:23-62,102-150,317-400, src/sensor/perspective.cpp:10-152,
// src/emitter/area.cpp:9-14) plus a closest-hit query that stands in for OptiX
// (src/scene/scene_optix.cpp:343-410).
#include "scene.h"
#include "envmap.h"
#include <limits>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>

namespace orc {

// ---------------------------------------------------------------- mesh preprocessing
// mesh.cpp:23-62 process_mesh<true>
static void process_mesh(const std::vector<V3d> &V, const std::vector<int> &F, int nf,
                         std::vector<Tri> &out) {
    const int nv = (int) V.size();
    out.resize(nf);
    std::vector<V3d> vn(nv);
    std::vector<Dual> vw(nv);
    std::vector<V3d> fnrm(nf);
    std::vector<Dual> farea(nf);
    for (int f = 0; f < nf; ++f) {
        Tri &t = out[f];
        for (int k = 0; k < 3; ++k) t.fi[k] = F[3 * f + k];
        t.p0 = V[t.fi[0]];
        t.e1 = V[t.fi[1]] - t.p0;
        t.e2 = V[t.fi[2]] - t.p0;
        fnrm[f] = cross(t.e1, t.e2);
        farea[f] = norm(fnrm[f]);
    }
    // area-weighted vertex normals (scatter_reduce order: corner-major, then face order)
    for (int i = 0; i < 3; ++i)
        for (int f = 0; f < nf; ++f) {
            int v = F[3 * f + i];
            vn[v] += fnrm[f];
            vw[v] += farea[f];
        }
    for (int v = 0; v < nv; ++v) vn[v] = normalize(vn[v] / vw[v]);
    for (int f = 0; f < nf; ++f) {
        Tri &t = out[f];
        t.n0 = vn[t.fi[0]]; t.n1 = vn[t.fi[1]]; t.n2 = vn[t.fi[2]];
        t.fn = fnrm[f] / farea[f];
        t.area = farea[f] * Dual(0.5f);
    }
}

// mesh.cpp:102-150 / 244-305: std::map keyed by (min, max) vertex id; value = [opposite vertex of
// the first face that introduced the edge, face ids...]; emitted in key order.
static void build_edges(const std::vector<int> &F, int nf, std::vector<MeshEdge> &out) {
    std::map<std::pair<int, int>, std::vector<int>> edge_map;
    for (int f = 0; f < nf; ++f)
        for (int i = 0; i < 3; ++i) {
            int a = F[3 * f + i], b = F[3 * f + (i + 1) % 3], c = F[3 * f + (i + 2) % 3];
            auto key = a < b ? std::make_pair(a, b) : std::make_pair(b, a);
            auto it = edge_map.find(key);
            if (it == edge_map.end()) it = edge_map.insert({key, std::vector<int>{c}}).first;
            it->second.push_back(f);
        }
    out.clear();
    for (auto &kv : edge_map) {
        MeshEdge e;
        e.v0 = kv.first.first; e.v1 = kv.first.second;
        e.opp = kv.second[0];
        e.f0 = kv.second[1];
        e.f1 = kv.second.size() >= 3 ? kv.second[2] : -1;
        out.push_back(e);
    }
}

// transform.h:48-61
static M4f perspective(float fov, float near_, float far_) {
    float recip = 1.f / (far_ - near_);
    float t = std::tan((fov * .5f) * (Pi / 180.f)), cot = 1.f / t;
    M4f m;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.m[i][j] = 0.f;
    m.m[0][0] = cot; m.m[1][1] = cot; m.m[2][2] = far_ * recip;
    m.m[2][3] = -near_ * far_ * recip;
    m.m[3][2] = 1.f;
    return m;
}
static M4f scale_m(float x, float y, float z) { M4f m; m.m[0][0] = x; m.m[1][1] = y; m.m[2][2] = z; return m; }
static M4f translate_m(float x, float y, float z) { M4f m; m.m[0][3] = x; m.m[1][3] = y; m.m[2][3] = z; return m; }

static void configure_camera(Scene &sc, CameraC &cam, const orc_camera &d, bool keep_edges) {
    // sensor.cpp:7-14
    float aspect = (float) sc.width / (float) sc.height;
    cam.to_world = make_m4d(d.to_world_left, d.d_to_world_left) * make_m4d(d.to_world_raw, d.d_to_world_raw)
                   * make_m4d(d.to_world_right, d.d_to_world_right);
    if (!(std::fabs(det3(detach(cam.to_world)) - 1.f) < Epsilon))
        throw std::runtime_error("Sensor transformation should not involve scaling!");
    // perspective.cpp:22-46
    cam.orthographic = d.orthographic != 0;
    // orthographic.cpp:12-16 / transform.h:75-78: orthographic(near, far) = scale(1, 1, 1/(far-near)) * translate(0, 0, -near)
    const M4f proj = cam.orthographic ? scale_m(1.f, 1.f, 1.f / (d.far_clip - d.near_clip)) * translate_m(0.f, 0.f, -d.near_clip)
                                      : perspective(d.fov_x, d.near_clip, d.far_clip);
    cam.camera_to_sample = scale_m(-0.5f, -0.5f * aspect, 1.f) * translate_m(-1.f, -1.f / aspect, 0.f) * proj;
    cam.sample_to_camera = inverse(cam.camera_to_sample);
    cam.world_to_sample = promote(cam.camera_to_sample) * inverse(cam.to_world);
    cam.sample_to_world = cam.to_world * promote(cam.sample_to_camera);
    cam.pos = transform_pos(cam.to_world, V3d(Dual(0.f)));
    cam.dir = transform_dir(cam.to_world, V3d(Dual(0.f), Dual(0.f), Dual(1.f)));
    V3f v00 = transform_pos(cam.sample_to_camera, V3f(0.f, 0.f, 0.f)),
        v10 = transform_pos(cam.sample_to_camera, V3f(1.f, 0.f, 0.f)),
        v11 = transform_pos(cam.sample_to_camera, V3f(1.f, 1.f, 0.f)),
        vc  = transform_pos(cam.sample_to_camera, V3f(.5f, .5f, 0.f));
    cam.inv_area = rcp(norm(v00 - v10) * norm(v11 - v10)) * squared_norm(vc);

    cam.edges.clear();
    cam.enable_edges = false;
    if (sc.sppe <= 0) return;
    // perspective.cpp:52-151 — silhouette test per mesh edge
    V3f cpos = detach(cam.pos);
    std::vector<PrimEdge> edges;
    for (size_t mi = 0; mi < sc.meshes.size(); ++mi) {
        const MeshC &m = sc.meshes[mi];
        if (!m.enable_edges) continue;
        int kept = 0;
        for (const MeshEdge &e : m.edges) {
            bool valid = e.f1 >= 0;
            const Tri &t0 = sc.tris[m.face_offset + e.f0];
            V3f e0 = normalize(cpos - detach(t0.p0)), n0 = detach(t0.fn);
            V3f e1(0.f), n1(0.f);
            if (valid) {
                const Tri &t1 = sc.tris[m.face_offset + e.f1];
                e1 = normalize(cpos - detach(t1.p0)); n1 = detach(t1.fn);
            } else {
                e1 = normalize(cpos);       // masked gather returns 0 for the triangle point
            }
            bool uv_mask = false;
            if (m.has_uv) {
                int a[3], b[3] = {0, 0, 0};
                for (int k = 0; k < 3; ++k) a[k] = m.face_uvs[3 * e.f0 + k];
                if (valid) for (int k = 0; k < 3; ++k) b[k] = m.face_uvs[3 * e.f1 + k];
                int uv_cut = 0;
                for (int k = 0; k < 3; ++k) if (a[k] == b[0] || a[k] == b[1] || a[k] == b[2]) uv_cut++;
                uv_mask = (uv_cut != 2);
            }
            bool keep;
            if (m.use_face_normals) {
                bool skip = valid && ((dot(e0, n0) < Epsilon && dot(e1, n1) < Epsilon) || (dot(n0, n1) > 1.f - Epsilon));
                keep = !skip;
            } else {
                bool act = !valid || ((dot(e0, n0) > Epsilon) != (dot(e1, n1) > Epsilon));
                keep = act;
            }
            if (m.has_uv) keep = keep || uv_mask;
            if (!keep) continue;
            ++kept;
            V3d q0 = transform_pos(cam.world_to_sample, m.verts[e.v0]),
                q1 = transform_pos(cam.world_to_sample, m.verts[e.v1]);
            PrimEdge pe;
            pe.p0 = V2d(q0.x, q0.y); pe.p1 = V2d(q1.x, q1.y);
            V2f ev(q1.x.v - q0.x.v, q1.y.v - q0.y.v);
            float len = norm(ev);
            ev = ev / len;
            pe.normal = V2f(-ev.y, ev.x);
            pe.length = len;
            edges.push_back(pe);
        }
        if (kept == 0) throw std::runtime_error("PerspectiveCamera::configure: mesh without primary edges (slices(info) > 0)");
    }
    if (!edges.empty() && keep_edges) {
        cam.edges = edges;
        std::vector<float> len;
        for (auto &e : edges) len.push_back(e.length);
        cam.edge_distrb.init(len);
        cam.enable_edges = true;
    }
}

Scene *configure_scene(const orc_scene_desc &d, const int *active, int n_active) {
    auto sc = std::make_unique<Scene>();
    sc->width = d.width; sc->height = d.height; sc->spp = d.spp; sc->sppe = d.sppe; sc->sppse = d.sppse;
    if (d.n_meshes <= 0) throw std::runtime_error("Missing meshes!");
    if (d.n_cameras <= 0) throw std::runtime_error("Missing sensor!");

    for (int i = 0; i < d.n_bsdfs; ++i) {
        const orc_bsdf &b = d.bsdfs[i];
        BsdfC bc;
        bc.type = b.type; bc.two_sided = b.two_sided != 0;
        bc.reflectance = V3d(Dual(b.reflectance[0], b.d_reflectance[0]), Dual(b.reflectance[1], b.d_reflectance[1]),
                             Dual(b.reflectance[2], b.d_reflectance[2]));
        bc.specular = V3d(Dual(b.specular[0], b.d_specular[0]), Dual(b.specular[1], b.d_specular[1]), Dual(b.specular[2], b.d_specular[2]));
        bc.roughness = Dual(b.roughness, b.d_roughness);
        bc.alpha_u = Dual(b.alpha_u, b.d_alpha_u); bc.alpha_v = Dual(b.alpha_v, b.d_alpha_v);
        bc.eta = V3d(Dual(b.eta[0], b.d_eta[0]), Dual(b.eta[1], b.d_eta[1]), Dual(b.eta[2], b.d_eta[2]));
        bc.k = V3d(Dual(b.k[0], b.d_k[0]), Dual(b.k[1], b.d_k[1]), Dual(b.k[2], b.d_k[2]));
        if (b.type < 0 || b.type > 5) throw std::runtime_error("Unknown BSDF type!");
        if (b.type == 5) {
            if (b.nested_bsdf < 0 || b.nested_bsdf >= d.n_bsdfs || d.bsdfs[b.nested_bsdf].type == 5) throw std::runtime_error("NormalMap: invalid nested BSDF");
            bc.nested = b.nested_bsdf;
        }
        if (b.tex_data != nullptr) {
            if (b.tex_width < 2 || b.tex_height < 2) throw std::runtime_error("Bitmap: invalid resolution!");
            bc.tex_w = b.tex_width; bc.tex_h = b.tex_height;
            const size_t n = (size_t) 3 * b.tex_width * b.tex_height;
            bc.tex.assign(b.tex_data, b.tex_data + n);
            if (b.d_tex_data) bc.d_tex.assign(b.d_tex_data, b.d_tex_data + n); else bc.d_tex.assign(n, 0.f);
        }
        auto take = [](const float *src, const float *dsrc, int w, int h, int ch, int &ow, int &oh, std::vector<float> &dst, std::vector<float> &ddst) {
            if (src == nullptr) return;
            if (w < 2 || h < 2) throw std::runtime_error("Bitmap: invalid resolution!");
            ow = w; oh = h;
            const size_t n = (size_t) ch * w * h;
            dst.assign(src, src + n);
            if (dsrc) ddst.assign(dsrc, dsrc + n); else ddst.assign(n, 0.f);
        };
        if (b.type == 4) {
            if (b.pv_count <= 0 || !b.pv_specular || !b.pv_diffuse || !b.pv_roughness) throw std::runtime_error("MicrofacetPerVertex: missing per-vertex data");
            const size_t n = (size_t) b.pv_count;
            bc.pv_spec.assign(b.pv_specular, b.pv_specular + 3 * n); bc.pv_diff.assign(b.pv_diffuse, b.pv_diffuse + 3 * n); bc.pv_rough.assign(b.pv_roughness, b.pv_roughness + n);
            if (b.d_pv_specular) bc.d_pv_spec.assign(b.d_pv_specular, b.d_pv_specular + 3 * n);
            if (b.d_pv_diffuse) bc.d_pv_diff.assign(b.d_pv_diffuse, b.d_pv_diffuse + 3 * n);
            if (b.d_pv_roughness) bc.d_pv_rough.assign(b.d_pv_roughness, b.d_pv_roughness + n);
        }
        take(b.spec_tex_data, b.d_spec_tex_data, b.spec_tex_width, b.spec_tex_height, 3, bc.spec_w, bc.spec_h, bc.spec_tex, bc.d_spec_tex);
        take(b.rough_tex_data, b.d_rough_tex_data, b.rough_tex_width, b.rough_tex_height, 1, bc.rough_w, bc.rough_h, bc.rough_tex, bc.d_rough_tex);
        for (int k = 0; k < 3; ++k) for (int q = 0; q < 4; ++q) bc.uv_xf[k][q] = Dual(b.tex_xf[k][q], b.d_tex_xf[k][q]);
        sc->bsdfs.push_back(bc);
    }
    for (int i = 0; i < d.n_emitters; ++i) {
        const orc_emitter &e = d.emitters[i];
        EmitterC ec;
        ec.radiance = V3d(Dual(e.radiance[0], e.d_radiance[0]), Dual(e.radiance[1], e.d_radiance[1]),
                          Dual(e.radiance[2], e.d_radiance[2]));
        ec.type = e.type;
        if (e.type == 1) {
            if (sc->env_emitter >= 0) throw std::runtime_error("A scene is only allowed to have one envmap!");
            sc->env_emitter = i;
            EnvmapC &E = sc->env;
            E.width = e.env_width; E.height = e.env_height;
            if (!e.env_data || E.width < 1 || E.height < 1) throw std::runtime_error("EnvironmentMap: missing radiance data");
            E.data.assign(e.env_data, e.env_data + (size_t) 3 * E.width * E.height);
            if (e.d_env_data) E.d_data.assign(e.d_env_data, e.d_env_data + (size_t) 3 * E.width * E.height);
            E.scale = Dual(e.env_scale, e.d_env_scale);
            for (int q = 0; q < 4; ++q) E.uv_xf[q] = Dual(e.env_uv_xf[q], e.d_env_uv_xf[q]);
            const float zero16[16] = {0};
            E.to_world = make_m4d(e.env_to_world_left, e.d_env_to_world_left) * make_m4d(e.env_to_world_raw, zero16);      // envmap.cpp:41
        }
        sc->emitters.push_back(ec);
    }

    // --- meshes: Mesh::configure (mesh.cpp:317-382) + concatenation (scene.cpp:528-571)
    int face_offset = 0;
    for (int mi = 0; mi < d.n_meshes; ++mi) {
        const orc_mesh &m = d.meshes[mi];
        MeshC mc;
        mc.face_offset = face_offset; mc.n_faces = m.n_faces; mc.n_vertices = m.n_vertices;
        mc.bsdf = m.bsdf_id; mc.emitter = m.emitter_id;
        mc.use_face_normals = m.use_face_normals != 0; mc.enable_edges = m.enable_edges != 0;
        mc.has_uv = (m.uvs != nullptr && m.n_uvs > 0);
        mc.faces.assign(m.faces, m.faces + 3 * m.n_faces);
        if (mc.has_uv) mc.face_uvs.assign(m.face_uvs, m.face_uvs + 3 * m.n_faces);
        M4d to_world = make_m4d(m.to_world_left, m.d_to_world_left) * make_m4d(m.to_world_raw, m.d_to_world_raw)
                       * make_m4d(m.to_world_right, m.d_to_world_right);            // mesh.cpp:325
        mc.verts.resize(m.n_vertices);
        for (int v = 0; v < m.n_vertices; ++v) {
            V3d raw(Dual(m.vertices[3 * v], m.d_vertices ? m.d_vertices[3 * v] : 0.f),
                    Dual(m.vertices[3 * v + 1], m.d_vertices ? m.d_vertices[3 * v + 1] : 0.f),
                    Dual(m.vertices[3 * v + 2], m.d_vertices ? m.d_vertices[3 * v + 2] : 0.f));
            mc.verts[v] = transform_pos(to_world, raw);                              // mesh.cpp:329
        }
        std::vector<Tri> tris;
        process_mesh(mc.verts, mc.faces, m.n_faces, tris);                           // mesh.cpp:339
        std::vector<float> areas(m.n_faces);
        for (int f = 0; f < m.n_faces; ++f) {
            Tri &t = tris[f];
            t.flat = mc.use_face_normals; t.mesh = mi;
            for (int k = 0; k < 3; ++k) t.uv[k] = V2f(0.f, 0.f);
            if (mc.has_uv)
                for (int k = 0; k < 3; ++k) { int ui = m.face_uvs[3 * f + k]; t.uv[k] = V2f(m.uvs[2 * ui], m.uvs[2 * ui + 1]); }
            areas[f] = t.area.v;
        }
        mc.total_area = sum_f32(areas);                                              // mesh.cpp:342-343
        mc.inv_total_area = 1.f / mc.total_area;
        mc.face_distrb.init(areas);                                                  // mesh.cpp:352
        if (mc.enable_edges) build_edges(mc.faces, m.n_faces, mc.edges);
        sc->tris.insert(sc->tris.end(), tris.begin(), tris.end());
        face_offset += m.n_faces;
        if (mc.emitter >= 0) {
            if (mc.emitter >= (int) sc->emitters.size()) throw std::runtime_error("bad emitter id");
            sc->emitters[mc.emitter].mesh = mi;
        }
        sc->meshes.push_back(std::move(mc));
    }

    // --- secondary edges (mesh.cpp:355-369, scene.cpp:547-568); every mesh edge is kept
    if (sc->sppse > 0) {
        std::vector<float> pmf;
        for (const MeshC &m : sc->meshes) {
            if (!m.enable_edges) continue;
            for (const MeshEdge &e : m.edges) {
                SecEdge se;
                se.is_boundary = e.f1 < 0;
                se.p0 = m.verts[e.v0];
                se.e1 = m.verts[e.v1] - se.p0;
                se.n0 = sc->tris[m.face_offset + e.f0].fn;
                se.n1 = se.is_boundary ? V3d(Dual(0.f)) : sc->tris[m.face_offset + e.f1].fn;
                se.p2 = m.verts[e.opp];
                sc->sec_edges.push_back(se);
                pmf.push_back(norm(detach(se.e1)));
            }
        }
        if (!pmf.empty()) sc->sec_edge_distrb.init(pmf);
    }

    // --- sensors (scene.cpp:376-416)
    sc->cameras.resize(d.n_cameras);
    for (int ci = 0; ci < d.n_cameras; ++ci) {
        bool keep = false;
        for (int k = 0; k < n_active; ++k) keep |= (active[k] == ci);
        configure_camera(*sc, sc->cameras[ci], d.cameras[ci], keep);
    }

    // --- scene bounds (scene.cpp:357-371, 379-416): all mesh vertices and all camera positions.
    // m_upper starts at numeric_limits<float>::min() (the smallest positive float, not the lowest), as in the reference.
    {
        float lo[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
        float hi[3] = {std::numeric_limits<float>::min(), std::numeric_limits<float>::min(), std::numeric_limits<float>::min()};
        auto grow = [&](const V3f &p) {
            const float c[3] = {p.x, p.y, p.z};
            for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], c[k]); hi[k] = std::max(hi[k], c[k]); }
        };
        for (const MeshC &m : sc->meshes) for (const V3d &v : m.verts) grow(detach(v));
        for (const CameraC &c : sc->cameras) if (!c.orthographic) grow(detach(c.pos));      // only PerspectiveCamera (scene.cpp:383-387)
        sc->lower = V3f(lo[0], lo[1], lo[2]); sc->upper = V3f(hi[0], hi[1], hi[2]);
    }

    // --- environment lighting: margin + bounding cube mesh (scene.cpp:434-485)
    if (sc->env_emitter >= 0) {
        const float ex[3] = {sc->upper.x - sc->lower.x, sc->upper.y - sc->lower.y, sc->upper.z - sc->lower.z};
        const float margin = std::min(ex[0], std::min(ex[1], ex[2])) * 0.05f;
        sc->lower = V3f(sc->lower.x - margin, sc->lower.y - margin, sc->lower.z - margin);
        sc->upper = V3f(sc->upper.x + margin, sc->upper.y + margin, sc->upper.z + margin);
        sc->env.lower = sc->lower; sc->env.upper = sc->upper;
        const float lo[3] = {sc->lower.x, sc->lower.y, sc->lower.z}, hi[3] = {sc->upper.x, sc->upper.y, sc->upper.z};
        static const int face_data[3][12] = {{0, 0, 1, 1, 2, 2, 0, 0, 0, 0, 4, 4}, {1, 3, 5, 7, 3, 7, 5, 4, 2, 6, 7, 6}, {3, 2, 7, 3, 7, 6, 1, 5, 6, 4, 5, 7}};
        MeshC mc;
        const int mi = (int) sc->meshes.size();
        mc.face_offset = (int) sc->tris.size(); mc.n_faces = 12; mc.n_vertices = 8;
        mc.bsdf = -1; mc.emitter = sc->env_emitter;
        mc.use_face_normals = true; mc.enable_edges = false; mc.has_uv = false;
        mc.verts.resize(8);
        for (int i = 0; i < 8; ++i)
            mc.verts[i] = V3d(Dual((i & 1) ? hi[0] : lo[0]), Dual((i & 2) ? hi[1] : lo[1]), Dual((i & 4) ? hi[2] : lo[2]));
        mc.faces.resize(36);
        for (int f = 0; f < 12; ++f) for (int k = 0; k < 3; ++k) mc.faces[3 * f + k] = face_data[k][f];
        std::vector<Tri> tris;
        process_mesh(mc.verts, mc.faces, 12, tris);
        std::vector<float> areas(12);
        for (int f = 0; f < 12; ++f) {
            Tri &t = tris[f];
            t.flat = true; t.mesh = mi;
            for (int k = 0; k < 3; ++k) t.uv[k] = V2f(0.f, 0.f);
            areas[f] = t.area.v;
        }
        mc.total_area = sum_f32(areas);
        mc.inv_total_area = 1.f / mc.total_area;
        mc.face_distrb.init(areas);
        sc->tris.insert(sc->tris.end(), tris.begin(), tris.end());
        sc->emitters[sc->env_emitter].mesh = mi;
        sc->meshes.push_back(std::move(mc));
    }

    // --- emitters (area.cpp:9-14, envmap.cpp:17-44, scene.cpp:488-515)
    if (!sc->emitters.empty()) {
        std::vector<float> w;
        double total_weight = 0.0;
        for (EmitterC &e : sc->emitters) {
            if (e.mesh < 0) throw std::runtime_error("emitter without mesh");
            if (e.type == 1) {
                envmap_configure(sc->env);
                e.sampling_weight = 0.f;
            } else {
                V3f r = detach(e.radiance);
                float lum = r.x * .2126f + r.y * .7152f + r.z * .0722f;              // utils.h:76-79
                e.sampling_weight = sc->meshes[e.mesh].total_area * lum;
            }
            total_weight += e.sampling_weight;
        }
        for (EmitterC &e : sc->emitters) if (e.type == 1) e.sampling_weight = (float) total_weight;   // scene.cpp:500-504
        for (EmitterC &e : sc->emitters) w.push_back(e.sampling_weight);
        sc->emitters_distrb.init(w);
        float inv_total = 1.f / sc->emitters_distrb.sum;
        for (EmitterC &e : sc->emitters) e.sampling_weight *= inv_total;
    }

    build_bvh(*sc);
    sc->use_bvh = sc->tris.size() > 64;
    return sc.release();
}

// ---------------------------------------------------------------- closest hit
// Triangle test = the reference's own ray_intersect_triangle (include/psdr/utils.h:82-93).  Hit
// selection: smallest t in (RayEpsilon, 1e8) (scene_optix.cpp:376), ties -> smallest triangle id
// (OptiX's tie-break is unspecified; a total order makes the result traversal-independent).
static inline bool tri_test(const Tri &T, const V3f &o, const V3f &d, float &u, float &v, float &t) {
    V3f p0 = detach(T.p0), e1 = detach(T.e1), e2 = detach(T.e2);
    V3f h = cross(d, e2);
    float a = dot(e1, h);
    float f = 1.f / a;
    V3f s = o - p0;
    u = f * dot(s, h);
    V3f q = cross(s, e1);
    v = f * dot(d, q);
    t = f * dot(e2, q);
    return (u >= 0.f) && (v >= 0.f) && (u + v <= 1.f) && (t > RayEpsilon) && (t < TraceTMax);
}

void build_bvh(Scene &s) {
    const int n = (int) s.tris.size();
    s.bvh.clear();
    s.bvh_tris.resize(n);
    std::vector<V3f> lo(n), hi(n), ctr(n);
    for (int i = 0; i < n; ++i) {
        s.bvh_tris[i] = i;
        V3f a = detach(s.tris[i].p0), b = a + detach(s.tris[i].e1), c = a + detach(s.tris[i].e2);
        for (int k = 0; k < 3; ++k) {
            lo[i][k] = std::min(a[k], std::min(b[k], c[k]));
            hi[i][k] = std::max(a[k], std::max(b[k], c[k]));
            ctr[i][k] = 0.5f * (lo[i][k] + hi[i][k]);
        }
    }
    struct Job { int node, first, count; };
    std::vector<Job> stack;
    s.bvh.push_back(BvhNode{});
    stack.push_back({0, 0, n});
    while (!stack.empty()) {
        Job j = stack.back(); stack.pop_back();
        BvhNode nd;
        for (int k = 0; k < 3; ++k) { nd.lo[k] = std::numeric_limits<float>::max(); nd.hi[k] = -std::numeric_limits<float>::max(); }
        float clo[3], chi[3];
        for (int k = 0; k < 3; ++k) { clo[k] = nd.lo[k]; chi[k] = nd.hi[k]; }
        for (int i = j.first; i < j.first + j.count; ++i) {
            int t = s.bvh_tris[i];
            for (int k = 0; k < 3; ++k) {
                nd.lo[k] = std::min(nd.lo[k], lo[t][k]); nd.hi[k] = std::max(nd.hi[k], hi[t][k]);
                clo[k] = std::min(clo[k], ctr[t][k]); chi[k] = std::max(chi[k], ctr[t][k]);
            }
        }
        // pad so that the slab test can never reject a ray the triangle test accepts
        for (int k = 0; k < 3; ++k) {
            float pad = 1e-4f * std::max(1.f, std::max(std::fabs(nd.lo[k]), std::fabs(nd.hi[k])));
            nd.lo[k] -= pad; nd.hi[k] += pad;
        }
        nd.left = nd.right = -1; nd.first = j.first; nd.count = j.count;
        int axis = 0;
        for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
        if (j.count > 4 && chi[axis] > clo[axis]) {
            int mid = j.first + j.count / 2;
            std::nth_element(s.bvh_tris.begin() + j.first, s.bvh_tris.begin() + mid, s.bvh_tris.begin() + j.first + j.count,
                             [&](int a, int b) { return ctr[a][axis] < ctr[b][axis] || (ctr[a][axis] == ctr[b][axis] && a < b); });
            nd.left = (int) s.bvh.size(); s.bvh.push_back(BvhNode{});
            nd.right = (int) s.bvh.size(); s.bvh.push_back(BvhNode{});
            nd.count = 0;
            stack.push_back({nd.left, j.first, mid - j.first});
            stack.push_back({nd.right, mid, j.first + j.count - mid});
        }
        s.bvh[j.node] = nd;
    }
}

Hit trace_closest(const Scene &s, const V3f &o, const V3f &d, bool use_bvh) {
    Hit best;
    // scene_optix.cpp:348-353: rays with a NaN component are masked out
    if (std::isnan(o.x) || std::isnan(o.y) || std::isnan(o.z) || std::isnan(d.x) || std::isnan(d.y) || std::isnan(d.z)) return best;
    float best_t = std::numeric_limits<float>::infinity();
    auto test = [&](int ti) {
        float u, v, t;
        if (tri_test(s.tris[ti], o, d, u, v, t)) {
            if (t < best_t || (t == best_t && ti < best.tri)) { best_t = t; best.tri = ti; best.u = u; best.v = v; best.t = t; }
        }
    };
    if (!use_bvh) {
        for (int i = 0; i < (int) s.tris.size(); ++i) test(i);
        return best;
    }
    float inv[3] = {1.f / d.x, 1.f / d.y, 1.f / d.z};
    int stack[64], sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const BvhNode &nd = s.bvh[stack[--sp]];
        float tn = 0.f, tf = best_t;
        bool ok = true;
        for (int k = 0; k < 3 && ok; ++k) {
            float t0 = (nd.lo[k] - o[k]) * inv[k], t1 = (nd.hi[k] - o[k]) * inv[k];
            if (std::isnan(t0) || std::isnan(t1)) { ok = (o[k] >= nd.lo[k] && o[k] <= nd.hi[k]); continue; }
            if (t0 > t1) std::swap(t0, t1);
            t1 *= 1.0000004f;   // Ize's robust slab correction
            tn = std::max(tn, t0); tf = std::min(tf, t1);
            ok = tn <= tf;
        }
        if (!ok) continue;
        if (nd.left < 0) { for (int i = nd.first; i < nd.first + nd.count; ++i) test(s.bvh_tris[i]); }
        else { stack[sp++] = nd.left; stack[sp++] = nd.right; }
    }
    return best;
}

} // namespace orc
