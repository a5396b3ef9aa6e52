// scene_host.cpp — host scene model: OBJ loading, Mesh/Camera/Scene::configure, snapshot assembly and
// the render drivers that call libpsdr_hip.so.  Reference sites are cited per function.
#include "scene_host.h"
#include "../common/envmath.h"
#include "../common/threads.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>

namespace psdr_host {

static constexpr float Epsilon = 1e-5f;      // reference include/psdr/constants.h:12
static constexpr float kPi = 3.14159265358979323846f;

void throw_assert(const char *cond, const char *file, int line, const std::string &msg) {
    std::ostringstream oss;
    oss << "Assertion failed in " << file << ":" << line << " : " << cond;
    if (!msg.empty()) oss << " (" << msg << ")";
    throw Exception(oss.str());
}

void Object::log(const std::string &msg) const { std::cout << "[" << type_name() << "] " << msg << std::endl; }

static void hip_check(int rc) { if (rc) throw Exception(std::string("libpsdr_hip: ") + psdr_hip_last_error()); }

static float sum_f32(const std::vector<float> &v) { float s = 0.f; for (float x : v) s += x; return s; }

// reference include/psdr/core/pmf.h:12-38 + src/core/pmf.cpp:6-15
void Distrb::init(const std::vector<float> &p) {
    PSDR_ASSERT_MSG(!p.empty(), "DiscreteDistribution: empty distribution!");
    size = (int) p.size();
    pmf = p;
    sum = sum_f32(p);
    cmf.resize(p.size());
    double acc = 0.0;
    for (size_t i = 0; i < p.size(); ++i) {
        PSDR_ASSERT_MSG(!(p[i] < 0.f), "DiscreteDistribution: entries must be non-negative!");
        acc += (double) p[i];
        cmf[i] = (float) acc;
    }
}

// ------------------------------------------------------------------------------------------------
// EnvironmentMap::configure, reference src/emitter/envmap.cpp:17-44
void EnvironmentMap::configure(bool on_device) {
    m_sampling_weight = 0.0f;
    PSDR_ASSERT(width > 1 && height > 1);
    PSDR_ASSERT_MSG(data.size() == (size_t) 3 * width * height, "Bitmap: invalid data size!");
    const int w2 = (width - 1) << 1, h2 = (height - 1) << 1;
    reso[0] = w2; reso[1] = h2;
    // the cell distribution depends on the texels only: rebuilt when they changed (2 M cells for a 1024 x 512 map; the reference
    // recomputes it on the device in every Scene::configure).  Scene::configure evaluates the masses on the device
    // (psdr_hip_env_cell_masses, one thread per cell); the host loop serves configure_host(), the device-less half the host tests
    // drive - the same envmath.h::cell_mass, bit-equal results.  The double-precision prefix sums stay on the host.
    if (m_cells_dirty || (int) cell_distrb.pmf.size() != w2 * h2) {
        std::vector<float> mass((size_t) w2 * h2);
        const int n_cells = w2 * h2;
        if (on_device) {
            hip_check(psdr_hip_env_cell_masses_xf(data.data(), width, height, uv_xf, mass.data()));
        } else {
            psdr::parallel_for((size_t) n_cells, 16384, [&](size_t b, size_t e) {
                for (size_t idx = b; idx < e; ++idx) mass[idx] = psdr::env::cell_mass(data.data(), width, height, w2, h2, (int) idx, psdr::env::UvXf<float>(uv_xf));
            });
        }
        cell_distrb.init(mass);
        m_cells_dirty = false;
    }
    const M16 z = zeros16();
    const DM4 tw = DM4::from(to_world_left, d_to_world_left) * DM4::from(to_world_raw, z);
    const DM4 fw = inverse(tw);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        to_world[4 * i + j] = tw.m[i][j].v; from_world[4 * i + j] = fw.m[i][j].v;
        d_to_world[4 * i + j] = tw.m[i][j].d; d_from_world[4 * i + j] = fw.m[i][j].d;
    }
    m_ready = true;
}
std::string EnvironmentMap::to_string() const {
    std::ostringstream oss;
    oss << "EnvironmentMap[sampling_weight = " << m_sampling_weight << "]";
    return oss.str();
}

static DM4 mul3(const M16 &l, const M16 &dl, const M16 &r, const M16 &dr, const M16 &rt, const M16 &drt) {
    return DM4::from(l, dl) * DM4::from(r, dr) * DM4::from(rt, drt);
}
DM4 Transformable::to_world() const { return mul3(to_world_left, d_to_world_left, to_world_raw, d_to_world_raw, to_world_right, d_to_world_right); }

void Transformable::set_transform(const M16 &mat, const M16 &dmat, bool set_left) {
    if (set_left) { to_world_left = mat; d_to_world_left = dmat; } else { to_world_right = mat; d_to_world_right = dmat; }
}
void Transformable::append_transform(const M16 &mat, const M16 &dmat, bool append_left) {
    DM4 m = DM4::from(mat, dmat);
    if (append_left) { DM4 r = m * DM4::from(to_world_left, d_to_world_left); r.split(to_world_left.data(), d_to_world_left.data()); }
    else { DM4 r = DM4::from(to_world_right, d_to_world_right) * m; r.split(to_world_right.data(), d_to_world_right.data()); }
}

// ------------------------------------------------------------------------------------------------
// OBJ reader.  Polygons are triangulated by ear clipping in the plane of the dominant normal axis
// (the published tinyobjloader algorithm the reference relies on, include/tiny_obj_loader; a convex
// n-gon becomes the fan (v0,v1,v2), (v0,v2,v3), ...).
namespace {
struct Corner { int v, vt; };

bool point_in_tri(const float *x, const float *y, float tx, float ty) {
    bool c = false;
    for (int i = 0, j = 2; i < 3; j = i++)
        if (((y[i] > ty) != (y[j] > ty)) && (tx < (x[j] - x[i]) * (ty - y[i]) / (y[j] - y[i]) + x[i])) c = !c;
    return c;
}

void triangulate(const std::vector<Corner> &poly, const std::vector<float> &V, std::vector<Corner> &out) {
    const size_t n = poly.size();
    if (n < 3) return;
    if (n == 3) { out.insert(out.end(), poly.begin(), poly.end()); return; }
    int ax0 = 1, ax1 = 2;
    for (size_t k = 0; k < n; ++k) {
        const float *a = &V[3 * poly[k].v], *b = &V[3 * poly[(k + 1) % n].v], *c = &V[3 * poly[(k + 2) % n].v];
        float e0[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e1[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
        float cx = std::fabs(e0[1] * e1[2] - e0[2] * e1[1]), cy = std::fabs(e0[2] * e1[0] - e0[0] * e1[2]), cz = std::fabs(e0[0] * e1[1] - e0[1] * e1[0]);
        const float eps = std::numeric_limits<float>::epsilon();
        if (cx > eps || cy > eps || cz > eps) {
            if (!(cx > cy && cx > cz)) { ax0 = 0; if (cz > cx && cz > cy) ax1 = 1; }
            break;
        }
    }
    float area = 0.f;
    for (size_t k = 0; k < n; ++k) {
        const float *a = &V[3 * poly[k].v], *b = &V[3 * poly[(k + 1) % n].v];
        area += (a[ax0] * b[ax1] - a[ax1] * b[ax0]) * 0.5f;
    }
    std::vector<Corner> rem = poly;
    size_t guess = 0, budget = n, prev = n;
    while (rem.size() > 3 && budget > 0) {
        const size_t m = rem.size();
        if (guess >= m) guess -= m;
        if (prev != m) { prev = m; budget = m; } else --budget;
        Corner tri[3]; float x[3], y[3];
        for (int k = 0; k < 3; ++k) { tri[k] = rem[(guess + k) % m]; x[k] = V[3 * tri[k].v + ax0]; y[k] = V[3 * tri[k].v + ax1]; }
        const float cr = (x[1] - x[0]) * (y[2] - y[1]) - (y[1] - y[0]) * (x[2] - x[1]);
        if (cr * area < 0.f) { ++guess; continue; }
        bool overlap = false;
        for (size_t o = 3; o < m && !overlap; ++o) {
            const Corner &c = rem[(guess + o) % m];
            overlap = point_in_tri(x, y, V[3 * c.v + ax0], V[3 * c.v + ax1]);
        }
        if (overlap) { ++guess; continue; }
        out.push_back(tri[0]); out.push_back(tri[1]); out.push_back(tri[2]);
        rem.erase(rem.begin() + (guess + 1) % m);
    }
    if (rem.size() == 3) out.insert(out.end(), rem.begin(), rem.end());
}
} // namespace

void Mesh::load(const std::string &fname, bool verbose) {
    std::ifstream in(fname);
    PSDR_ASSERT_MSG(in.good(), std::string("Failed to load OBJ from: ") + fname);
    std::vector<float> V, VT;
    std::vector<Corner> corners;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag)) continue;
        if (tag == "v") { float x, y, z; ss >> x >> y >> z; V.push_back(x); V.push_back(y); V.push_back(z); }
        else if (tag == "vt") { float u = 0.f, v = 0.f; ss >> u >> v; VT.push_back(u); VT.push_back(v); }
        else if (tag == "f") {
            std::vector<Corner> poly;
            std::string tok;
            while (ss >> tok) {
                Corner c{0, -1};
                int vi = 0, ti = 0;
                size_t s1 = tok.find('/');
                vi = std::stoi(tok.substr(0, s1));
                if (s1 != std::string::npos) {
                    size_t s2 = tok.find('/', s1 + 1);
                    std::string t = tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                    if (!t.empty()) ti = std::stoi(t);
                }
                c.v = vi > 0 ? vi - 1 : (int) V.size() / 3 + vi;
                c.vt = ti > 0 ? ti - 1 : (ti < 0 ? (int) VT.size() / 2 + ti : -1);
                poly.push_back(c);
            }
            for (const Corner &c : poly) PSDR_ASSERT_MSG(c.v >= 0 && 3 * c.v + 2 < (int) V.size(), std::string("Bad vertex index in: ") + fname);
            triangulate(poly, V, corners);
        }
    }
    std::vector<int> f(corners.size()), fuv;
    for (size_t i = 0; i < corners.size(); ++i) f[i] = corners[i].v;
    if (!VT.empty()) { fuv.resize(corners.size()); for (size_t i = 0; i < corners.size(); ++i) fuv[i] = corners[i].vt; }
    load_raw(V, f, VT, fuv, verbose);
}

void Mesh::load_raw(const std::vector<float> &v, const std::vector<int> &f, const std::vector<float> &uv, const std::vector<int> &fuv, bool verbose) {
    PSDR_ASSERT(v.size() % 3 == 0 && f.size() % 3 == 0);
    m_num_vertices = (int) v.size() / 3;
    m_num_faces = (int) f.size() / 3;
    vertex_positions_raw = v;
    d_vertex_positions_raw.assign(v.size(), 0.f);
    m_has_uv = !uv.empty();
    vertex_uv = uv;
    face_uv_indices = m_has_uv ? fuv : std::vector<int>();
    if (m_has_uv) PSDR_ASSERT(fuv.size() == f.size());
    face_indices = f;
    for (int idx : f) PSDR_ASSERT(idx >= 0 && idx < m_num_vertices);
    edges.clear();
    ++m_topo_version;
    if (m_enable_edges) build_edges();
    if (verbose) std::cout << "Loaded " << m_num_vertices << " vertices, " << m_num_faces << " faces, " << edges.size() << " edges. " << std::endl;
    m_ready = false;
}

// reference src/shape/mesh.cpp:102-150: std::map over (min,max) vertex ids; first entry of the value is the
// vertex opposite to the edge in the first face seen, then the face ids; emitted in key order.
void Mesh::build_edges() {
    std::map<std::pair<int, int>, std::vector<int>> em;
    for (int f = 0; f < m_num_faces; ++f)
        for (int i = 0; i < 3; ++i) {
            const int a = face_indices[3 * f + i], b = face_indices[3 * f + (i + 1) % 3], c = face_indices[3 * f + (i + 2) % 3];
            const auto key = a < b ? std::make_pair(a, b) : std::make_pair(b, a);
            auto it = em.find(key);
            if (it == em.end()) it = em.emplace(key, std::vector<int>{c}).first;
            it->second.push_back(f);
        }
    edges.clear();
    edges.reserve(em.size());
    for (const auto &kv : em) edges.push_back({kv.first.first, kv.first.second, kv.second[1], kv.second.size() >= 3 ? kv.second[2] : -1, kv.second[0]});
}

// per vertex, its (corner i, face f) incidences in the order the reference's three scatter passes visit them (mesh.cpp:34-41: corner 0 of every face,
// then corner 1, then corner 2), so that the per-vertex sums below add in that very order whatever the thread count
void Mesh::build_vertex_faces() {
    const int nv = m_num_vertices, nf = m_num_faces;
    vf_begin.assign((size_t) nv + 1, 0);
    for (int i = 0; i < 3 * nf; ++i) vf_begin[(size_t) face_indices[i] + 1]++;
    for (int v = 0; v < nv; ++v) vf_begin[(size_t) v + 1] += vf_begin[(size_t) v];
    vf_item.resize((size_t) 3 * nf);
    std::vector<int> at(vf_begin.begin(), vf_begin.end() - 1);
    for (int i = 0; i < 3; ++i)
        for (int f = 0; f < nf; ++f) vf_item[(size_t) at[(size_t) face_indices[3 * f + i]]++] = (f << 2) | i;      // (face, corner); ascending in i * nf + f by construction
    vf_topo = m_topo_version;
}

// process_mesh<true>, reference src/shape/mesh.cpp:23-62; rows of 22 floats: p0 e1 e2 n0 n1 n2 fn area
// (vf_begin / vf_item: Mesh::build_vertex_faces; the loops are independent per face / per vertex and run on the OpenMP team)
static void process_mesh(const std::vector<D3> &V, const std::vector<int> &F, int nf, const std::vector<int> &vf_begin, const std::vector<int> &vf_item,
                         std::vector<float> &tri, std::vector<float> &d_tri, std::vector<float> *vertex_normals_out) {
    const size_t nv = V.size();
    std::vector<D3> vn(nv), fnrm(nf);
    std::vector<DF> farea(nf);
    psdr::parallel_for((size_t) nf, 4096, [&](size_t b, size_t e) {
        for (size_t f = b; f < e; ++f) {
            const D3 p0 = V[F[3 * f]], e1 = V[F[3 * f + 1]] - p0, e2 = V[F[3 * f + 2]] - p0;
            fnrm[f] = dcross(e1, e2);
            farea[f] = dnorm(fnrm[f]);
        }
    });
    psdr::parallel_for(nv, 4096, [&](size_t b, size_t e) {
        for (size_t v = b; v < e; ++v) {
            D3 acc; DF w;
            for (int k = vf_begin[v]; k < vf_begin[v + 1]; ++k) { const int f = vf_item[(size_t) k] >> 2; acc = acc + fnrm[f]; w = w + farea[f]; }
            vn[v] = dnormalize(acc / w);
        }
    });
    if (vertex_normals_out) {
        vertex_normals_out->resize(3 * nv);
        for (size_t v = 0; v < nv; ++v) { (*vertex_normals_out)[3 * v] = vn[v].x.v; (*vertex_normals_out)[3 * v + 1] = vn[v].y.v; (*vertex_normals_out)[3 * v + 2] = vn[v].z.v; }
        return;
    }
    tri.resize(22 * (size_t) nf); d_tri.resize(22 * (size_t) nf);
    auto put = [&](size_t row, int col, const D3 &a) {
        tri[22 * row + col] = a.x.v; tri[22 * row + col + 1] = a.y.v; tri[22 * row + col + 2] = a.z.v;
        d_tri[22 * row + col] = a.x.d; d_tri[22 * row + col + 1] = a.y.d; d_tri[22 * row + col + 2] = a.z.d;
    };
    psdr::parallel_for((size_t) nf, 4096, [&](size_t b, size_t e) {
        for (size_t f = b; f < e; ++f) {
            const D3 p0 = V[F[3 * f]], e1 = V[F[3 * f + 1]] - p0, e2 = V[F[3 * f + 2]] - p0;
            put(f, 0, p0); put(f, 3, e1); put(f, 6, e2);
            put(f, 9, vn[F[3 * f]]); put(f, 12, vn[F[3 * f + 1]]); put(f, 15, vn[F[3 * f + 2]]);
            put(f, 18, fnrm[f] / farea[f]);
            const DF a = farea[f] * DF(0.5f);
            tri[22 * f + 21] = a.v; d_tri[22 * f + 21] = a.d;
        }
    });
}

// Mesh::configure, reference src/shape/mesh.cpp:317-382
void Mesh::configure() {
    if (m_bsdf != nullptr) PSDR_ASSERT(!m_bsdf->anisotropic() || !m_use_face_normals);
    PSDR_ASSERT_MSG(m_num_faces > 0, "Mesh has no faces!");
    if (d_vertex_positions_raw.size() != vertex_positions_raw.size()) d_vertex_positions_raw.assign(vertex_positions_raw.size(), 0.f);
    // the same inputs as last time: nothing to do (values: raw vertices, the three to_world factors, the topology; tangents: theirs)
    const M16 *mv[3] = {&to_world_left, &to_world_raw, &to_world_right}, *md[3] = {&d_to_world_left, &d_to_world_raw, &d_to_world_right};
    bool same_values = cfg_valid && cfg_topo == m_topo_version && cfg_raw == vertex_positions_raw, same_tangents = cfg_valid && cfg_d_raw == d_vertex_positions_raw;
    const bool same_raw = same_values, same_d_raw = same_tangents;
    for (int k = 0; k < 3; ++k) { same_values = same_values && cfg_m[k] == *mv[k]; same_tangents = same_tangents && cfg_dm[k] == *md[k]; }
    if (same_values && same_tangents) {
        // (a mesh loaded with enable_edges = False has no edge list; switched on between two configure() calls it needs one although no value changed)
        if (m_enable_edges && edges.empty()) build_edges();
        m_ready = true;
        return;
    }
    if (vf_topo != m_topo_version || vf_begin.size() != (size_t) m_num_vertices + 1) build_vertex_faces();
    std::vector<D3> raw(m_num_vertices), world(m_num_vertices);
    for (int v = 0; v < m_num_vertices; ++v)
        raw[v] = {DF(vertex_positions_raw[3 * v], d_vertex_positions_raw[3 * v]), DF(vertex_positions_raw[3 * v + 1], d_vertex_positions_raw[3 * v + 1]),
                  DF(vertex_positions_raw[3 * v + 2], d_vertex_positions_raw[3 * v + 2])};
    std::vector<float> dummy1, dummy2;
    if (!raw_normals_valid || !same_raw || !same_d_raw) {       // (a function of the raw vertices alone)
        process_mesh(raw, face_indices, m_num_faces, vf_begin, vf_item, dummy1, dummy2, &vertex_normals_raw);
        raw_normals_valid = true;
    }
    const DM4 tw = to_world();
    tw.split(m_tw, m_d_tw);
    vertex_positions.resize(3 * (size_t) m_num_vertices); d_vertex_positions.resize(3 * (size_t) m_num_vertices);
    psdr::parallel_for((size_t) m_num_vertices, 4096, [&](size_t b, size_t e) {
        for (size_t v = b; v < e; ++v) {
            world[v] = xform_pos(tw, raw[v]);
            vertex_positions[3 * v] = world[v].x.v; vertex_positions[3 * v + 1] = world[v].y.v; vertex_positions[3 * v + 2] = world[v].z.v;
            d_vertex_positions[3 * v] = world[v].x.d; d_vertex_positions[3 * v + 1] = world[v].y.d; d_vertex_positions[3 * v + 2] = world[v].z.d;
        }
    });
    for (int k = 0; k < 3; ++k) { m_lower[k] = std::numeric_limits<float>::max(); m_upper[k] = -std::numeric_limits<float>::max(); }
    for (int v = 0; v < m_num_vertices; ++v)
        for (int k = 0; k < 3; ++k) { m_lower[k] = std::min(m_lower[k], vertex_positions[3 * v + k]); m_upper[k] = std::max(m_upper[k], vertex_positions[3 * v + k]); }
    std::vector<float> areas(m_num_faces);
    face_p0n.resize(6 * (size_t) m_num_faces);
    if (m_lean) {
        // per face: first vertex, unit normal and area as VALUES - the value parts of process_mesh's dual-number formulas, operation for operation (dcross / dnorm of hnum.h)
        rows_valid = false;
        const float *P = vertex_positions.data();
        psdr::parallel_for((size_t) m_num_faces, 4096, [&](size_t b, size_t e) {
            for (size_t f = b; f < e; ++f) {
                const float *a = P + 3 * (size_t) face_indices[3 * f], *q1 = P + 3 * (size_t) face_indices[3 * f + 1], *q2 = P + 3 * (size_t) face_indices[3 * f + 2];
                const float e1[3] = {q1[0] - a[0], q1[1] - a[1], q1[2] - a[2]}, e2[3] = {q2[0] - a[0], q2[1] - a[1], q2[2] - a[2]};
                const float n[3] = {std::fmaf(e1[1], e2[2], -(e1[2] * e2[1])), std::fmaf(e1[2], e2[0], -(e1[0] * e2[2])), std::fmaf(e1[0], e2[1], -(e1[1] * e2[0]))};
                const float a2 = std::sqrt(std::fmaf(n[2], n[2], std::fmaf(n[1], n[1], n[0] * n[0])));
                float *o = &face_p0n[6 * f];
                o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = n[0] / a2; o[4] = n[1] / a2; o[5] = n[2] / a2;
                areas[f] = a2 * 0.5f;
            }
        });
    } else {
        process_mesh(world, face_indices, m_num_faces, vf_begin, vf_item, tri, d_tri, nullptr);
        rows_valid = true;
        for (int f = 0; f < m_num_faces; ++f) {
            areas[f] = tri[22 * (size_t) f + 21];
            for (int k = 0; k < 3; ++k) { face_p0n[6 * (size_t) f + k] = tri[22 * (size_t) f + k]; face_p0n[6 * (size_t) f + 3 + k] = tri[22 * (size_t) f + 18 + k]; }
        }
    }
    if (!same_values) {           // (areas are values: the face distribution follows them)
        m_total_area = sum_f32(areas);
        m_inv_total_area = 1.f / m_total_area;
        face_distrb.init(areas);
    }
    if (m_enable_edges && edges.empty()) build_edges();
    if (!same_values) ++m_geo_version;
    ++m_tan_version;
    cfg_raw = vertex_positions_raw; cfg_d_raw = d_vertex_positions_raw; cfg_topo = m_topo_version;
    for (int k = 0; k < 3; ++k) { cfg_m[k] = *mv[k]; cfg_dm[k] = *md[k]; }
    cfg_valid = true;
    m_ready = true;
}

// the TriangleInfo rows a lean configure() left out: the same process_mesh on the same world vertices (stored as floats: the very values)
void Mesh::ensure_rows() {
    if (rows_valid) return;
    std::vector<D3> world((size_t) m_num_vertices);
    for (size_t v = 0; v < (size_t) m_num_vertices; ++v)
        world[v] = {DF(vertex_positions[3 * v], d_vertex_positions[3 * v]), DF(vertex_positions[3 * v + 1], d_vertex_positions[3 * v + 1]), DF(vertex_positions[3 * v + 2], d_vertex_positions[3 * v + 2])};
    if (vf_topo != m_topo_version || vf_begin.size() != (size_t) m_num_vertices + 1) build_vertex_faces();
    process_mesh(world, face_indices, m_num_faces, vf_begin, vf_item, tri, d_tri, nullptr);
    rows_valid = true;
}

std::string Mesh::to_string() const {
    std::ostringstream oss;
    oss << "Mesh[nv=" << m_num_vertices << ", nf=" << m_num_faces;
    if (!m_id.empty()) oss << ", id=" << m_id;
    if (m_bsdf) oss << ", bsdf=" << m_bsdf->to_string();
    oss << "]";
    return oss.str();
}

// Mesh::dump, reference src/shape/mesh.cpp:469-541 (raw=true writes the world-space positions there)
void Mesh::dump(const std::string &fname, bool raw) const {
    const std::vector<float> &P = raw ? vertex_positions : vertex_positions_raw;
    FILE *fout = std::fopen(fname.c_str(), "wt");
    PSDR_ASSERT_MSG(fout != nullptr, std::string("Cannot write: ") + fname);
    for (int i = 0; i < m_num_vertices; ++i) std::fprintf(fout, "v %.6e %.6e %.6e\n", P[3 * i], P[3 * i + 1], P[3 * i + 2]);
    if (m_has_uv) for (size_t i = 0; i + 1 < vertex_uv.size(); i += 2) std::fprintf(fout, "vt %.6e %.6e\n", vertex_uv[i], vertex_uv[i + 1]);
    for (int f = 0; f < m_num_faces; ++f) {
        const int *v = &face_indices[3 * f];
        if (m_has_uv) { const int *t = &face_uv_indices[3 * f]; std::fprintf(fout, "f %d/%d %d/%d %d/%d\n", v[0] + 1, t[0] + 1, v[1] + 1, t[1] + 1, v[2] + 1, t[2] + 1); }
        else std::fprintf(fout, "f %d %d %d\n", v[0] + 1, v[1] + 1, v[2] + 1);
    }
    std::fclose(fout);
}

// ------------------------------------------------------------------------------------------------
// PerspectiveCamera::configure, reference src/sensor/perspective.cpp:10-152 + sensor.cpp:7-14
static DM4 persp_matrix(float fov, float near_, float far_) {       // reference include/psdr/core/transform.h:48-61
    const float recip = 1.f / (far_ - near_);
    const float cot = 1.f / std::tan((fov * .5f) * (kPi / 180.f));
    DM4 m;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.m[i][j] = DF(0.f);
    m.m[0][0] = DF(cot); m.m[1][1] = DF(cot); m.m[2][2] = DF(far_ * recip); m.m[2][3] = DF(-near_ * far_ * recip); m.m[3][2] = DF(1.f);
    return m;
}
static DM4 scale_matrix(float x, float y, float z) { DM4 m = DM4::identity(); m.m[0][0] = DF(x); m.m[1][1] = DF(y); m.m[2][2] = DF(z); return m; }
static DM4 translate_matrix(float x, float y, float z) { DM4 m = DM4::identity(); m.m[0][3] = DF(x); m.m[1][3] = DF(y); m.m[2][3] = DF(z); return m; }
static D3 fvec(const float *p) { return {DF(p[0]), DF(p[1]), DF(p[2])}; }
static D3 dvec(const float *p, const float *d) { return {DF(p[0], d[0]), DF(p[1], d[1]), DF(p[2], d[2])}; }
static float fdot(D3 a, D3 b) { return ddot(a, b).v; }

void PerspectiveCamera::configure(const Scene &scene, bool keep_edges) {
    const RenderOption &o = scene.m_opts;
    m_width = o.width; m_height = o.height;                    // Sensor::m_resolution (sensor.cpp:16-22)
    const float aspect = (float) o.width / (float) o.height;
    const DM4 tw = to_world();
    PSDR_ASSERT_MSG(std::fabs(det3(tw) - 1.f) < Epsilon, "Sensor transformation should not involve scaling!");
    const DM4 proj = m_orthographic ? scale_matrix(1.f, 1.f, 1.f / (m_far_clip - m_near_clip)) * translate_matrix(0.f, 0.f, -m_near_clip)   // transform.h:75-78
                                    : persp_matrix(m_fov_x, m_near_clip, m_far_clip);
    const DM4 c2s = scale_matrix(-0.5f, -0.5f * aspect, 1.f) * translate_matrix(-1.f, -1.f / aspect, 0.f) * proj;
    rec.orthographic = m_orthographic ? 1 : 0;
    const DM4 s2c = inverse(c2s);
    const DM4 w2s = c2s * inverse(tw);
    float dummy[16];
    s2c.split(rec.sample_to_camera, dummy);
    tw.split(rec.to_world, rec.d_to_world);
    w2s.split(rec.world_to_sample, rec.d_world_to_sample);
    const D3 pos = xform_pos(tw, {DF(0.f), DF(0.f), DF(0.f)}), dir = xform_dir(tw, {DF(0.f), DF(0.f), DF(1.f)});
    rec.cam_pos[0] = pos.x.v; rec.cam_pos[1] = pos.y.v; rec.cam_pos[2] = pos.z.v;
    rec.cam_dir[0] = dir.x.v; rec.cam_dir[1] = dir.y.v; rec.cam_dir[2] = dir.z.v;
    const D3 v00 = xform_pos(s2c, {DF(0.f), DF(0.f), DF(0.f)}), v10 = xform_pos(s2c, {DF(1.f), DF(0.f), DF(0.f)}),
             v11 = xform_pos(s2c, {DF(1.f), DF(1.f), DF(0.f)}), vc = xform_pos(s2c, {DF(.5f), DF(.5f), DF(0.f)});
    rec.inv_area = (1.f / (dnorm(v00 - v10).v * dnorm(v11 - v10).v)) * ddot(vc, vc).v;

    m_edges = PrimaryEdges();
    m_enable_edges = false;
    if (o.sppe <= 0) return;
    PrimaryEdges pe;
    const D3 cpos = {DF(pos.x.v), DF(pos.y.v), DF(pos.z.v)};
    for (const Mesh *mesh : scene.m_meshes) {
        if (!mesh->m_enable_edges) continue;
        // pass 1 (parallel): which edges are kept; pass 2 (parallel): their rows at the positions an in-order walk would give them
        const int ne = (int) mesh->edges.size();
        std::vector<uint8_t> keep_flag((size_t) ne, 0);
        psdr::parallel_for((size_t) ne, 4096, [&](size_t eb, size_t ee) {
          for (size_t i = eb; i < ee; ++i) {
            const MeshEdge &e = mesh->edges[i];
            const bool valid = e.f1 >= 0;
            const float *t0 = &mesh->face_p0n[6 * (size_t) e.f0];          // (first vertex and unit normal of the face: Mesh::configure keeps them whether or not it wrote the full rows)
            const D3 e0 = dnormalize(cpos - fvec(t0)), n0 = fvec(t0 + 3);
            D3 e1 = dnormalize(cpos), n1 = {DF(0.f), DF(0.f), DF(0.f)};        // masked gathers return zeros
            if (valid) { const float *t1 = &mesh->face_p0n[6 * (size_t) e.f1]; e1 = dnormalize(cpos - fvec(t1)); n1 = fvec(t1 + 3); }
            bool uv_mask = false;
            if (mesh->m_has_uv) {
                int b[3] = {0, 0, 0}, cut = 0;
                if (valid) for (int k = 0; k < 3; ++k) b[k] = mesh->face_uv_indices[3 * e.f1 + k];
                for (int k = 0; k < 3; ++k) { const int a = mesh->face_uv_indices[3 * e.f0 + k]; if (a == b[0] || a == b[1] || a == b[2]) ++cut; }
                uv_mask = cut != 2;
            }
            bool keep;
            if (mesh->m_use_face_normals) keep = !(valid && ((fdot(e0, n0) < Epsilon && fdot(e1, n1) < Epsilon) || fdot(n0, n1) > 1.f - Epsilon));
            else keep = !valid || ((fdot(e0, n0) > Epsilon) != (fdot(e1, n1) > Epsilon));
            if (mesh->m_has_uv) keep = keep || uv_mask;
            keep_flag[i] = keep ? 1 : 0;
          }
        });
        std::vector<int> kept_ids;
        for (int i = 0; i < ne; ++i) if (keep_flag[(size_t) i]) kept_ids.push_back(i);
        const int kept = (int) kept_ids.size();
        PSDR_ASSERT_MSG(kept > 0, "slices(info) > 0");
        const size_t base = pe.length.size();
        for (std::vector<float> *v : {&pe.p0, &pe.p1, &pe.d_p0, &pe.d_p1, &pe.normal}) v->resize(2 * (base + (size_t) kept));
        pe.length.resize(base + (size_t) kept); pe.ids.resize(3 * (base + (size_t) kept));
        psdr::parallel_for((size_t) kept, 4096, [&](size_t jb, size_t je) {
          for (size_t j = jb; j < je; ++j) {
            const MeshEdge &e = mesh->edges[(size_t) kept_ids[j]];
            const size_t r = base + j;
            const D3 q0 = xform_pos(w2s, dvec(&mesh->vertex_positions[3 * e.v0], &mesh->d_vertex_positions[3 * e.v0])),
                     q1 = xform_pos(w2s, dvec(&mesh->vertex_positions[3 * e.v1], &mesh->d_vertex_positions[3 * e.v1]));
            float ex = q1.x.v - q0.x.v, ey = q1.y.v - q0.y.v;
            const float len = std::sqrt(std::fmaf(ey, ey, ex * ex));
            ex /= len; ey /= len;
            pe.p0[2 * r] = q0.x.v; pe.p0[2 * r + 1] = q0.y.v; pe.p1[2 * r] = q1.x.v; pe.p1[2 * r + 1] = q1.y.v;
            pe.d_p0[2 * r] = q0.x.d; pe.d_p0[2 * r + 1] = q0.y.d; pe.d_p1[2 * r] = q1.x.d; pe.d_p1[2 * r + 1] = q1.y.d;
            pe.normal[2 * r] = -ey; pe.normal[2 * r + 1] = ex;
            pe.length[r] = len;
            pe.ids[3 * r] = mesh->m_mesh_id; pe.ids[3 * r + 1] = e.v0; pe.ids[3 * r + 2] = e.v1;
          }
        });
    }
    if (!pe.length.empty() && keep_edges) {
        pe.distrb.init(pe.length);
        m_edges = std::move(pe);
        m_enable_edges = true;
    }
}

// ------------------------------------------------------------------------------------------------
Scene::Scene() { for (int k = 0; k < 3; ++k) { m_lower[k] = 0.f; m_upper[k] = 0.f; } }
Scene::~Scene() {
    release_device();
    for (Sensor *s : m_sensors) delete s;
    for (Emitter *e : m_emitters) delete e;
    for (BSDF *b : m_bsdfs) delete b;
    for (Mesh *m : m_meshes) delete m;
}
void Scene::release_device() { if (m_hip) { psdr_hip_scene_destroy(m_hip); m_hip = nullptr; } }

// build_param_map, reference src/scene/scene.cpp:29-47
template <typename T> static void add_params(std::unordered_map<std::string, Object *> &pm, const std::vector<T *> &arr, const char *name) {
    for (size_t i = 0; i < arr.size(); ++i) {
        std::ostringstream k1; k1 << name << "[" << i << "]";
        pm[k1.str()] = arr[i];
        if (!arr[i]->m_id.empty()) { std::ostringstream k2; k2 << name << "[id=" << arr[i]->m_id << "]"; pm[k2.str()] = arr[i]; }
    }
}
void Scene::rebuild_param_map() {
    m_param_map.clear();
    add_params(m_param_map, m_meshes, "Mesh"); add_params(m_param_map, m_bsdfs, "BSDF");
    add_params(m_param_map, m_emitters, "Emitter"); add_params(m_param_map, m_sensors, "Sensor");
}

void Scene::add_Sensor(const Sensor *sensor) {       // scene.cpp:107-126 (the scene keeps its own copy)
    if (m_opts.log_level > 0) std::cout << "add_Sensor: " << sensor->to_string() << std::endl;
    const PerspectiveCamera *pc = dynamic_cast<const PerspectiveCamera *>(sensor);
    PSDR_ASSERT_MSG(pc != nullptr, "Unknown sensor type!");
    m_sensors.push_back(new PerspectiveCamera(*pc));
    m_num_sensors = (int) m_sensors.size();
    rebuild_param_map();
}
// a heap copy of one of the plain BSDFs (nullptr in -> a default Microfacet, as the reference's add_BSDF gives a NormalMap)
static BSDF *clone_bsdf(const BSDF *b) {
    if (const Diffuse *d = dynamic_cast<const Diffuse *>(b)) return new Diffuse(*d);
    if (const Microfacet *m = dynamic_cast<const Microfacet *>(b)) return new Microfacet(*m);
    if (const RoughConductor *r = dynamic_cast<const RoughConductor *>(b)) return new RoughConductor(*r);
    if (const RoughDielectric *r = dynamic_cast<const RoughDielectric *>(b)) return new RoughDielectric(*r);
    if (const MicrofacetPerVertex *r = dynamic_cast<const MicrofacetPerVertex *>(b)) return new MicrofacetPerVertex(*r);
    PSDR_ASSERT_MSG(b == nullptr, "Unsupported normal map nested BSDF");
    return new Microfacet();
}
NormalMap::~NormalMap() { delete m_bsdf; }
void NormalMap::set_nested(const BSDF *b) { BSDF *c = clone_bsdf(b); delete m_bsdf; m_bsdf = c; }
// Scene::add_normalmap_BSDF, reference scene.cpp:128-145
void Scene::add_normalmap_BSDF(const NormalMap *bsdf, const Microfacet *micro, const std::string &bsdf_id, bool twoSide) {
    PSDR_ASSERT_MSG(bsdf != nullptr && micro != nullptr, "add_normalmap_BSDF: null argument");
    NormalMap tmp(*bsdf);
    tmp.m_bsdf = const_cast<Microfacet *>(micro);          // cloned by add_BSDF
    try { add_BSDF(&tmp, bsdf_id, twoSide); } catch (...) { tmp.m_bsdf = nullptr; throw; }
    tmp.m_bsdf = nullptr;
}
void Scene::add_BSDF(const BSDF *bsdf, const std::string &bsdf_id, bool twoSide) {      // scene.cpp:148-247
    const Diffuse *d = dynamic_cast<const Diffuse *>(bsdf);
    const Microfacet *mf = dynamic_cast<const Microfacet *>(bsdf);
    const RoughConductor *rc = dynamic_cast<const RoughConductor *>(bsdf);
    const RoughDielectric *rd = dynamic_cast<const RoughDielectric *>(bsdf);     // (the reference only reaches it through the XML loader)
    const MicrofacetPerVertex *pvb = dynamic_cast<const MicrofacetPerVertex *>(bsdf);
    if (const NormalMap *nm = dynamic_cast<const NormalMap *>(bsdf)) {
        // (the reference's add_BSDF replaces a NormalMap by a flat one over a default Microfacet, scene.cpp:219-228; its
        //  add_normalmap_BSDF and its XML loader keep the map and the nested BSDF - as done here)
        if (m_opts.log_level > 0) std::cout << "add_BSDF: NormalMapBSDF " << bsdf_id << std::endl;
        PSDR_ASSERT_MSG(m_param_map.find("BSDF[id=" + bsdf_id + "]") == m_param_map.end(), std::string("Duplicate BSDF id: ") + bsdf_id);
        NormalMap *c = new NormalMap(*nm);
        c->m_bsdf = clone_bsdf(nm->m_bsdf);
        c->m_twoSide = twoSide; c->m_id = bsdf_id;
        m_bsdfs.push_back(c);
        rebuild_param_map();
        return;
    }
    PSDR_ASSERT_MSG(d != nullptr || mf != nullptr || rc != nullptr || rd != nullptr || pvb != nullptr, "Unknown BSDF type!");
    if (m_opts.log_level > 0) std::cout << "add_BSDF: " << bsdf->type_name() << " " << bsdf_id << std::endl;
    PSDR_ASSERT_MSG(m_param_map.find("BSDF[id=" + bsdf_id + "]") == m_param_map.end(), std::string("Duplicate BSDF id: ") + bsdf_id);
    BSDF *c = d ? static_cast<BSDF *>(new Diffuse(*d)) : (mf ? static_cast<BSDF *>(new Microfacet(*mf)) : (rc ? static_cast<BSDF *>(new RoughConductor(*rc)) : (rd ? static_cast<BSDF *>(new RoughDielectric(*rd)) : static_cast<BSDF *>(new MicrofacetPerVertex(*pvb)))));
    c->m_twoSide = twoSide; c->m_id = bsdf_id;
    m_bsdfs.push_back(c);
    rebuild_param_map();
}
static int find_bsdf(const Scene &s, const std::string &id) {
    for (size_t i = 0; i < s.m_bsdfs.size(); ++i) if (s.m_bsdfs[i]->m_id == id) return (int) i;
    return -1;
}
void Scene::add_Mesh(const std::string &fname, const M16 &transform, const std::string &bsdf_id, const Emitter *emitter) {   // scene.cpp:249-279
    Mesh m;
    m.load(fname, false);
    m.to_world_raw = transform;
    add_Mesh(&m, bsdf_id, emitter);
}
void Scene::add_Mesh(const Mesh *mesh_, const std::string &bsdf_id, const Emitter *emitter) {                              // scene.cpp:281-309
    if (m_opts.log_level > 0) std::cout << "add_Mesh: " << m_meshes.size() << std::endl;
    const int b = find_bsdf(*this, bsdf_id);
    PSDR_ASSERT_MSG(b >= 0, std::string("Unknown BSDF id: ") + bsdf_id);
    Mesh *mesh = new Mesh(*mesh_);
    mesh->m_mesh_id = (int) m_meshes.size();
    mesh->m_bsdf_id = b; mesh->m_bsdf = m_bsdfs[b];
    mesh->m_emitter = nullptr; mesh->m_emitter_id = -1;
    if (emitter) {
        const AreaLight *al = dynamic_cast<const AreaLight *>(emitter);
        PSDR_ASSERT_MSG(al != nullptr, "Unknown emitter type!");
        if (m_opts.log_level > 0) std::cout << "add Area light" << std::endl;
        AreaLight *c = new AreaLight(*al);
        c->m_mesh = mesh;
        mesh->m_emitter = c; mesh->m_emitter_id = (int) m_emitters.size();
        m_emitters.push_back(c);
    }
    m_meshes.push_back(mesh);
    m_num_meshes = (int) m_meshes.size();
    rebuild_param_map();
}

void Scene::add_EnvironmentMap(const EnvironmentMap *emitter_) {                                                           // scene.cpp:97-105
    PSDR_ASSERT_MSG(m_emitter_env == nullptr, "A scene is only allowed to have one envmap!");
    PSDR_ASSERT_MSG(emitter_ != nullptr, "Unknown emitter type!");
    EnvironmentMap *emitter = new EnvironmentMap(*emitter_);
    m_emitters.push_back(emitter);
    m_emitter_env = emitter;
    rebuild_param_map();
}

bool Scene::is_ready() const {
    return m_configured && m_hip != nullptr && (m_opts.spp == 0 || m_samplers[0].ready) && (m_opts.sppe == 0 || m_samplers[1].ready) &&
           (m_opts.sppse == 0 || m_samplers[2].ready);
}

// Scene::configure, reference src/scene/scene.cpp:311-601
void Scene::configure(const std::vector<int> &active_sensor) {
    using namespace std::chrono;
    const auto start_time = high_resolution_clock::now();
    m_device_config = true;           // (device-side steps of the host half: the environment map's cell masses)
    try { configure_host(active_sensor); } catch (...) { m_device_config = false; throw; }
    m_device_config = false;
    m_ms_host = duration_cast<duration<double, std::milli>>(high_resolution_clock::now() - start_time).count();
    upload();
    if (m_opts.log_level > 0) {
        std::ostringstream oss;
        oss << "Configured in " << duration_cast<duration<double>>(high_resolution_clock::now() - start_time).count() << " seconds.";
        log(oss.str());
    }
}

// host half of configure: everything except the device upload / BVH build
void Scene::configure_host(const std::vector<int> &active_sensor) {
    m_configured = false;
    if (m_opts.log_level > 0) std::cout << "[Scene] resolution: " << m_opts.height << " " << m_opts.width << std::endl;
    PSDR_ASSERT(m_num_sensors == (int) m_sensors.size());
    PSDR_ASSERT(m_num_meshes == (int) m_meshes.size());

    // seed samplers when the lane count changed (scene.cpp:329-344)
    const int spps[3] = {m_opts.spp, m_opts.sppe, m_opts.sppse};
    for (int k = 0; k < 3; ++k)
        if (spps[k] > 0) {
            const int64_t count = (int64_t) m_opts.height * m_opts.width * spps[k];
            if (!m_samplers[k].ready || m_samplers[k].sample_count != count) m_samplers[k] = SamplerState{true, count, (uint64_t) (int64_t) seed, 0};
        }

    PSDR_ASSERT_MSG(!m_meshes.empty(), "Missing meshes!");
    PSDR_ASSERT_MSG(!m_sensors.empty(), "Missing sensor!");
    for (int id : active_sensor) PSDR_ASSERT_MSG(id >= 0 && id < m_num_sensors, "Invalid sensor id!");

    Snapshot &S = snap;
    // the small per-object records are rebuilt every time; the per-triangle and per-edge arrays keep their rows unless their mesh changed
    S.meshes.clear(); S.bsdfs.clear(); S.emitters.clear(); S.face_pmf.clear(); S.face_cmf.clear(); S.envmap = psdr_envmap_rec{}; S.has_envmap = false;
    S.emitters_distrb = Distrb();
    uint32_t same = PSDR_SAME_TRIANGLES | PSDR_SAME_TRI_TANGENTS | PSDR_SAME_SEC_EDGES | PSDR_SAME_PRIM_EDGES | PSDR_SAME_ENV_TEXELS | PSDR_SAME_ENV_TANGENT | PSDR_SAME_BITMAPS;
    // m_upper starts at numeric_limits<float>::min(), the smallest positive float, as in the reference (scene.cpp:357-358)
    for (int k = 0; k < 3; ++k) { m_lower[k] = std::numeric_limits<float>::max(); m_upper[k] = std::numeric_limits<float>::min(); }
    // LEAN (round 6): when the device scene exists and computed the moved meshes' rows itself in the last upload (psdr_mesh_geometry), this configure leaves the
    // per-triangle rows, their snapshot copies and the secondary-edge row arrays for ensure_full_snapshot() - nobody reads them unless the device scene has to be created or
    // rebuilt, or a test asks - and computes what the host itself consumes: world vertices, face normals and areas, the edge selections and their CDFs.
    // (environment-lit scenes only: without a map the device rebuilds its live-pixel masks from the rows whenever a triangle moves)
    bool has_env = false;
    for (const Emitter *e : m_emitters) has_env = has_env || dynamic_cast<const EnvironmentMap *>(e) != nullptr;
    {
        std::vector<MeshKey> keys_now;
        for (const Mesh *mesh : m_meshes) keys_now.push_back(MeshKey{mesh, mesh->m_topo_version, mesh->m_num_faces, mesh->m_bsdf_id, mesh->m_emitter_id, mesh->m_has_uv, mesh->m_use_face_normals, mesh->m_enable_edges});
        size_t total_faces = 0;
        for (const Mesh *mesh : m_meshes) total_faces += (size_t) mesh->m_num_faces;
        // (scenes of at most 64 triangles are traced by brute force: their device scene always takes the host's rows)
        m_lean = m_hip != nullptr && !m_always_rebuild && m_device_rows_ok && has_env && total_faces > 64 && keys_now == m_snap_keys && std::getenv("PSDR_HOST_GEOMETRY") == nullptr;
    }
    for (Mesh *mesh : m_meshes) { mesh->m_lean = m_lean; mesh->configure(); }          // (does nothing for a mesh whose inputs are those of its previous run)
    auto key_of = [](const Mesh *m) { return MeshKey{m, m->m_topo_version, m->m_num_faces, m->m_bsdf_id, m->m_emitter_id, m->m_has_uv, m->m_use_face_normals, m->m_enable_edges}; };
    // rows [face_offset, face_offset + n_faces) of the triangle arrays for one mesh: the values and / or the tangents
    auto write_rows = [&](const Mesh *mesh, size_t face_offset, bool values, bool tangents) {
        const int nf = mesh->m_num_faces;
        auto put3 = [](std::vector<float> &dst, size_t row, const float *src) { dst[3 * row] = src[0]; dst[3 * row + 1] = src[1]; dst[3 * row + 2] = src[2]; };
        psdr::parallel_for((size_t) nf, 4096, [&](size_t fb, size_t fe) {
          for (size_t f = fb; f < fe; ++f) {
            const size_t row = face_offset + f;
            const float *t = &mesh->tri[22 * (size_t) f], *d = &mesh->d_tri[22 * (size_t) f];
            if (values) {
                put3(S.p0, row, t); put3(S.e1, row, t + 3); put3(S.e2, row, t + 6); put3(S.n0, row, t + 9); put3(S.n1, row, t + 12); put3(S.n2, row, t + 15); put3(S.fn, row, t + 18); S.area[row] = t[21];
                for (int k = 0; k < 3; ++k) {
                    if (mesh->m_has_uv) { const int ui = mesh->face_uv_indices[3 * f + k]; S.uv[6 * row + 2 * k] = mesh->vertex_uv[2 * ui]; S.uv[6 * row + 2 * k + 1] = mesh->vertex_uv[2 * ui + 1]; }
                    else { S.uv[6 * row + 2 * k] = 0.f; S.uv[6 * row + 2 * k + 1] = 0.f; }
                    S.face_indices[3 * row + k] = mesh->face_indices[3 * f + k];
                }
                S.mesh_id[row] = mesh->m_mesh_id;
                S.flat[row] = mesh->m_use_face_normals ? 1 : 0;
            }
            if (tangents) { put3(S.d_p0, row, d); put3(S.d_e1, row, d + 3); put3(S.d_e2, row, d + 6); put3(S.d_n0, row, d + 9); put3(S.d_n1, row, d + 12); put3(S.d_n2, row, d + 15); put3(S.d_fn, row, d + 18); S.d_area[row] = d[21]; }
          }
        });
    };
    auto resize_rows = [&](size_t n) {
        for (std::vector<float> *v : {&S.p0, &S.e1, &S.e2, &S.n0, &S.n1, &S.n2, &S.fn, &S.d_p0, &S.d_e1, &S.d_e2, &S.d_n0, &S.d_n1, &S.d_n2, &S.d_fn}) v->resize(3 * n);
        S.area.resize(n); S.d_area.resize(n); S.uv.resize(6 * n); S.mesh_id.resize(n); S.face_indices.resize(3 * n); S.flat.resize(n);
    };
    // the mesh records, the face distributions of the emitters' meshes and the scene box (cheap: per mesh, not per triangle)
    int face_offset = 0;
    auto append_record = [&](const Mesh *mesh) {
        psdr_mesh_rec r{};
        r.bsdf_id = mesh->m_bsdf_id; r.emitter_id = mesh->m_emitter_id; r.face_offset = face_offset; r.n_faces = mesh->m_num_faces;
        r.inv_total_area = mesh->m_inv_total_area; r.distrb_offset = 0; r.distrb_sum = mesh->face_distrb.sum;
        if (mesh->m_emitter_id >= 0) {
            r.distrb_offset = (int) S.face_pmf.size();
            S.face_pmf.insert(S.face_pmf.end(), mesh->face_distrb.pmf.begin(), mesh->face_distrb.pmf.end());
            S.face_cmf.insert(S.face_cmf.end(), mesh->face_distrb.cmf.begin(), mesh->face_distrb.cmf.end());
        }
        S.meshes.push_back(r);
        face_offset += mesh->m_num_faces;
        for (int k = 0; k < 3; ++k) { m_lower[k] = std::min(m_lower[k], mesh->m_lower[k]); m_upper[k] = std::max(m_upper[k], mesh->m_upper[k]); }
    };
    {
        std::vector<MeshKey> keys;
        size_t total = 0;
        for (const Mesh *mesh : m_meshes) { keys.push_back(key_of(mesh)); total += (size_t) mesh->m_num_faces; }
        const bool same_layout = keys == m_snap_keys && S.area.size() == total && m_seen_geo.size() == m_meshes.size();
        if (!same_layout) {
            resize_rows(total);
            m_seen_geo.assign(m_meshes.size(), ~0ull); m_seen_tan.assign(m_meshes.size(), ~0ull);
            m_bits_geo.assign(m_meshes.size(), ~0ull); m_bits_tan.assign(m_meshes.size(), ~0ull);
            m_snap_keys = keys;
            ++m_layout_version;
        }
        for (size_t i = 0; i < m_meshes.size(); ++i) {
            const Mesh *mesh = m_meshes[i];
            // (m_bits_*: the versions the PSDR_SAME_* bits were last derived from; m_seen_*: the versions the snapshot's rows hold - behind after a lean configure)
            if (m_bits_geo.size() != m_meshes.size()) { m_bits_geo.assign(m_meshes.size(), ~0ull); m_bits_tan.assign(m_meshes.size(), ~0ull); }
            if (m_bits_geo[i] != mesh->m_geo_version) same &= ~PSDR_SAME_TRIANGLES;
            if (m_bits_tan[i] != mesh->m_tan_version) same &= ~PSDR_SAME_TRI_TANGENTS;
            m_bits_geo[i] = mesh->m_geo_version; m_bits_tan[i] = mesh->m_tan_version;
            const bool values = m_seen_geo[i] != mesh->m_geo_version, tangents = m_seen_tan[i] != mesh->m_tan_version;
            if ((values || tangents) && !m_lean) {
                const_cast<Mesh *>(mesh)->ensure_rows();
                write_rows(mesh, (size_t) face_offset, values, tangents);
                m_seen_geo[i] = mesh->m_geo_version; m_seen_tan[i] = mesh->m_tan_version;
            }
            append_record(mesh);
        }
    }
    uint64_t geo_sum = 0, tan_sum = 0;           // (versions only grow: equal sums = no mesh changed)
    for (const Mesh *mesh : m_meshes) { geo_sum += mesh->m_geo_version; tan_sum += mesh->m_tan_version; }

    // sensors: only the active ones keep their primary-edge list (scene.cpp:381-416)
    std::vector<size_t> num_edges;
    for (int sid = 0; sid < m_num_sensors; ++sid) {
        const bool active = std::find(active_sensor.begin(), active_sensor.end(), sid) != active_sensor.end();
        PerspectiveCamera *cam = static_cast<PerspectiveCamera *>(m_sensors[sid]);
        {
            // the sensor's own members, the render options and the state of the meshes it projects: the same as in its previous run -> nothing to do
            std::vector<float> key;
            for (const M16 *m : {&cam->to_world_left, &cam->to_world_raw, &cam->to_world_right, &cam->d_to_world_left, &cam->d_to_world_raw, &cam->d_to_world_right}) key.insert(key.end(), m->begin(), m->end());
            const double extra[] = {(double) cam->m_fov_x, (double) cam->m_near_clip, (double) cam->m_far_clip, cam->m_orthographic ? 1.0 : 0.0, (double) m_opts.width, (double) m_opts.height,
                                    (double) m_opts.sppe, active ? 1.0 : 0.0};
            for (double x : extra) key.push_back((float) x);
            // the version sums in 16-bit pieces: each is exact as a float (a 32-bit piece is not beyond 2^24, and two sums that round to the same float would keep stale edges)
            for (uint64_t v : {geo_sum, tan_sum, (uint64_t) m_layout_version})
                for (int k = 0; k < 4; ++k) key.push_back((float) ((v >> (16 * k)) & 0xffffull));
            if (key != cam->cfg_key || m_opts.sppe <= 0) {
                cam->configure(*this, active);
                cam->cfg_key = key;
                cam->m_edges_version++;
            }
        }
        if (!cam->m_orthographic)      // only PerspectiveCamera positions extend the scene box (scene.cpp:383-387, 410-414)
            for (int k = 0; k < 3; ++k) { m_lower[k] = std::min(m_lower[k], cam->rec.cam_pos[k]); m_upper[k] = std::max(m_upper[k], cam->rec.cam_pos[k]); }
        if (m_opts.sppe > 0 && (active || active_sensor.empty())) num_edges.push_back(cam->m_enable_edges ? cam->m_edges.length.size() : 1);
    }
    if (m_opts.log_level > 0) {
        std::ostringstream oss;
        oss << "AABB: [lower = [[" << m_lower[0] << ", " << m_lower[1] << ", " << m_lower[2] << "]], upper = [[" << m_upper[0] << ", " << m_upper[1] << ", " << m_upper[2] << "]]]";
        log(oss.str());
        if (m_opts.sppe > 0 && !num_edges.empty()) {
            std::ostringstream o2; o2 << "(" << num_edges[0];
            for (size_t i = 1; i < num_edges.size(); ++i) o2 << ", " << num_edges[i];
            o2 << ") primary edges initialized.";
            log(o2.str());
        }
    }

    // environment lighting: margin + bounding cube, added once (scene.cpp:434-485)
    if (m_emitter_env != nullptr && !m_has_bound_mesh) {
        const float margin = std::min(m_upper[0] - m_lower[0], std::min(m_upper[1] - m_lower[1], m_upper[2] - m_lower[2])) * 0.05f;
        for (int k = 0; k < 3; ++k) { m_lower[k] -= margin; m_upper[k] += margin; m_emitter_env->lower[k] = m_lower[k]; m_emitter_env->upper[k] = m_upper[k]; }
        static const int face_data[3][12] = {{0, 0, 1, 1, 2, 2, 0, 0, 0, 0, 4, 4}, {1, 3, 5, 7, 3, 7, 5, 4, 2, 6, 7, 6}, {3, 2, 7, 3, 7, 6, 1, 5, 6, 4, 5, 7}};
        Mesh *bound = new Mesh();
        bound->m_num_vertices = 8; bound->m_num_faces = 12;
        bound->m_use_face_normals = true; bound->m_enable_edges = false;
        bound->m_bsdf = nullptr; bound->m_bsdf_id = -1;
        bound->m_emitter = m_emitter_env;
        bound->m_emitter_id = (int) (std::find(m_emitters.begin(), m_emitters.end(), (Emitter *) m_emitter_env) - m_emitters.begin());
        bound->m_mesh_id = (int) m_meshes.size();
        bound->vertex_positions_raw.resize(24);
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 3; ++j) bound->vertex_positions_raw[3 * i + j] = (i & (1 << j)) ? m_upper[j] : m_lower[j];
        bound->face_indices.resize(36);
        for (int f = 0; f < 12; ++f) for (int k = 0; k < 3; ++k) bound->face_indices[3 * f + k] = face_data[k][f];
        m_emitter_env->m_bound_mesh_id = bound->m_mesh_id;
        m_meshes.push_back(bound);
        ++m_num_meshes;
        m_has_bound_mesh = true;
        bound->configure();
        resize_rows(S.area.size() + (size_t) bound->m_num_faces);
        write_rows(bound, (size_t) face_offset, true, true);
        append_record(bound);
        m_snap_keys.push_back(key_of(bound)); m_seen_geo.push_back(bound->m_geo_version); m_seen_tan.push_back(bound->m_tan_version);
        m_bits_geo.push_back(bound->m_geo_version); m_bits_tan.push_back(bound->m_tan_version);
        geo_sum += bound->m_geo_version; tan_sum += bound->m_tan_version;
        ++m_layout_version;
        same &= ~(PSDR_SAME_TRIANGLES | PSDR_SAME_TRI_TANGENTS);
        if (m_opts.log_level > 0) log("Bounding mesh added for environmental lighting.");
    }

    // emitters (area.cpp:9-14, envmap.cpp:17-44, scene.cpp:488-515)
    bool env_cells_rebuilt = false;
    if (!m_emitters.empty()) {
        std::vector<float> weights;
        double total_weight = 0.0;
        for (Emitter *e : m_emitters) {
            if (EnvironmentMap *env = dynamic_cast<EnvironmentMap *>(e)) {
                env_cells_rebuilt = env_cells_rebuilt || env->m_cells_dirty || env->cell_distrb.pmf.empty();
                env->configure(m_device_config);
            } else {
                AreaLight *al = static_cast<AreaLight *>(e);
                PSDR_ASSERT(al->m_mesh != nullptr && al->m_mesh->m_ready);
                al->m_sampling_weight = al->m_mesh->m_total_area * (al->radiance[0] * .2126f + al->radiance[1] * .7152f + al->radiance[2] * .0722f);
                al->m_ready = true;
            }
            total_weight += e->m_sampling_weight;
        }
        for (Emitter *e : m_emitters) if (dynamic_cast<EnvironmentMap *>(e)) e->m_sampling_weight = (float) total_weight;
        for (Emitter *e : m_emitters) weights.push_back(e->m_sampling_weight);
        S.emitters_distrb.init(weights);
        const float inv_total = 1.f / S.emitters_distrb.sum;
        for (Emitter *e : m_emitters) {
            e->m_sampling_weight *= inv_total;
            psdr_emitter_rec r{};
            r.sampling_weight = e->m_sampling_weight;
            if (EnvironmentMap *env = dynamic_cast<EnvironmentMap *>(e)) {
                r.type = 1; r.mesh_id = env->m_bound_mesh_id;
                psdr_envmap_rec &er = S.envmap;
                er.width = env->width; er.height = env->height; er.radiance = env->data.data(); er.scale = env->scale;
                std::memcpy(er.to_world, env->to_world, 64); std::memcpy(er.from_world, env->from_world, 64);
                std::memcpy(er.d_to_world, env->d_to_world, 64); std::memcpy(er.d_from_world, env->d_from_world, 64);
                er.d_scale = env->d_scale;
                er.d_radiance = env->d_data.size() == env->data.size() ? env->d_data.data() : nullptr;
                for (int k = 0; k < 3; ++k) { er.lower[k] = env->lower[k]; er.upper[k] = env->upper[k]; }
                er.reso[0] = env->reso[0]; er.reso[1] = env->reso[1];
                er.cell_pmf = env->cell_distrb.pmf.data(); er.cell_cmf = env->cell_distrb.cmf.data(); er.cell_sum = env->cell_distrb.sum;
                for (int k = 0; k < 4; ++k) { er.radiance_xf[k] = env->uv_xf[k]; er.d_radiance_xf[k] = env->d_uv_xf[k]; }
                S.has_envmap = true;
            } else {
                AreaLight *al = static_cast<AreaLight *>(e);
                r.type = 0; r.mesh_id = al->m_mesh->m_mesh_id;
                for (int k = 0; k < 3; ++k) { r.radiance[k] = al->radiance[k]; r.d_radiance[k] = al->d_radiance[k]; }
            }
            S.emitters.push_back(r);
        }
    }
    auto rec_of_type = [&](const BSDF *b, std::vector<psdr_bsdf_rec> &dst) {
        if (const NormalMap *nm = dynamic_cast<const NormalMap *>(b)) {
            psdr_bsdf_rec r{};
            r.type = 5; r.two_sided = nm->m_twoSide ? 1 : 0; r.nested_bsdf = -1;       // patched below
            for (int k = 0; k < 3; ++k) { r.reflectance[k] = nm->normal[k]; r.d_reflectance[k] = nm->d_normal[k]; }
            if (nm->tex_w > 0) {
                PSDR_ASSERT_MSG(nm->tex_w >= 2 && nm->tex_h >= 2, "Bitmap: invalid resolution!");
                PSDR_ASSERT_MSG(nm->tex.size() == (size_t) 3 * nm->tex_w * nm->tex_h, "Bitmap: invalid data size!");
                r.tex_width = nm->tex_w; r.tex_height = nm->tex_h; r.tex_data = nm->tex.data();
                r.d_tex_data = nm->d_tex.size() == nm->tex.size() ? nm->d_tex.data() : nullptr;
            }
            dst.push_back(r);
            return;
        }
        if (const Microfacet *mf = dynamic_cast<const Microfacet *>(b)) {
            psdr_bsdf_rec r{};
            r.type = 1; r.two_sided = mf->m_twoSide ? 1 : 0;
            for (int k = 0; k < 3; ++k) {
                r.reflectance[k] = mf->diffuse[k]; r.d_reflectance[k] = mf->d_diffuse[k];
                r.specular[k] = mf->specular[k]; r.d_specular[k] = mf->d_specular[k];
            }
            r.roughness = mf->roughness; r.d_roughness = mf->d_roughness;
            auto chk = [](const ParamTex &t, int ch) {
                if (t.w == 0) return false;
                PSDR_ASSERT_MSG(t.w >= 2 && t.h >= 2, "Bitmap: invalid resolution!");
                PSDR_ASSERT_MSG(t.v.size() == (size_t) ch * t.w * t.h, "Bitmap: invalid data size!");
                return true;
            };
            if (chk(mf->diffuse_tex, 3)) {
                r.tex_width = mf->diffuse_tex.w; r.tex_height = mf->diffuse_tex.h; r.tex_data = mf->diffuse_tex.v.data();
                r.d_tex_data = mf->diffuse_tex.d.size() == mf->diffuse_tex.v.size() ? mf->diffuse_tex.d.data() : nullptr;
            }
            if (chk(mf->specular_tex, 3)) {
                r.spec_tex_width = mf->specular_tex.w; r.spec_tex_height = mf->specular_tex.h; r.spec_tex_data = mf->specular_tex.v.data();
                r.d_spec_tex_data = mf->specular_tex.d.size() == mf->specular_tex.v.size() ? mf->specular_tex.d.data() : nullptr;
            }
            if (chk(mf->roughness_tex, 1)) {
                r.rough_tex_width = mf->roughness_tex.w; r.rough_tex_height = mf->roughness_tex.h; r.rough_tex_data = mf->roughness_tex.v.data();
                r.d_rough_tex_data = mf->roughness_tex.d.size() == mf->roughness_tex.v.size() ? mf->roughness_tex.d.data() : nullptr;
            }
            dst.push_back(r);
            return;
        }
        if (const RoughConductor *rc = dynamic_cast<const RoughConductor *>(b)) {
            psdr_bsdf_rec r{};
            r.type = 2; r.two_sided = rc->m_twoSide ? 1 : 0;
            r.alpha_u = rc->alpha_u; r.alpha_v = rc->alpha_v; r.d_alpha_u = rc->d_alpha_u; r.d_alpha_v = rc->d_alpha_v;
            for (int k = 0; k < 3; ++k) {
                r.eta[k] = rc->eta[k]; r.d_eta[k] = rc->d_eta[k]; r.k[k] = rc->k[k]; r.d_k[k] = rc->d_k[k];
                r.specular[k] = rc->specular[k]; r.d_specular[k] = rc->d_specular[k];
            }
            auto put_tex = [](const ParamTex &t, int ch, int32_t &w, int32_t &h, const float *&data, const float *&ddata) {
                if (t.w == 0) return;
                PSDR_ASSERT_MSG(t.w >= 2 && t.h >= 2, "Bitmap: invalid resolution!");
                PSDR_ASSERT_MSG(t.v.size() == (size_t) ch * t.w * t.h, "Bitmap: invalid data size!");
                w = t.w; h = t.h; data = t.v.data(); ddata = t.d.size() == t.v.size() ? t.d.data() : nullptr;
            };
            put_tex(rc->eta_tex, 3, r.tex_width, r.tex_height, r.tex_data, r.d_tex_data);                         // slot 0 = eta
            put_tex(rc->k_tex, 3, r.spec_tex_width, r.spec_tex_height, r.spec_tex_data, r.d_spec_tex_data);       // slot 1 = k
            put_tex(rc->alpha_tex, 1, r.rough_tex_width, r.rough_tex_height, r.rough_tex_data, r.d_rough_tex_data);   // slot 2 = alpha
            dst.push_back(r);
            return;
        }
        if (const MicrofacetPerVertex *pv = dynamic_cast<const MicrofacetPerVertex *>(b)) {
            psdr_bsdf_rec r{};
            r.type = 4; r.two_sided = pv->m_twoSide ? 1 : 0;
            const size_t nv = pv->roughness.size();
            PSDR_ASSERT_MSG(nv > 0 && pv->specular.size() == 3 * nv && pv->diffuse.size() == 3 * nv, "MicrofacetPerVertex: specular / diffuse need 3 values and roughness 1 value per vertex");
            r.pv_count = (int) nv;
            r.pv_specular = pv->specular.data(); r.pv_diffuse = pv->diffuse.data(); r.pv_roughness = pv->roughness.data();
            r.d_pv_specular = pv->d_specular.size() == 3 * nv ? pv->d_specular.data() : nullptr;
            r.d_pv_diffuse = pv->d_diffuse.size() == 3 * nv ? pv->d_diffuse.data() : nullptr;
            r.d_pv_roughness = pv->d_roughness.size() == nv ? pv->d_roughness.data() : nullptr;
            dst.push_back(r);
            return;
        }
        if (const RoughDielectric *rd = dynamic_cast<const RoughDielectric *>(b)) {
            psdr_bsdf_rec r{};
            r.type = 3; r.two_sided = rd->m_twoSide ? 1 : 0;
            r.alpha_u = rd->alpha_u; r.alpha_v = rd->alpha_v; r.d_alpha_u = rd->d_alpha_u; r.d_alpha_v = rd->d_alpha_v;
            r.eta[0] = rd->eta; r.eta[1] = rd->inv_eta; r.d_eta[0] = rd->d_eta; r.d_eta[1] = rd->d_inv_eta;
            if (rd->alpha_tex.w > 0) {
                const ParamTex &t = rd->alpha_tex;
                PSDR_ASSERT_MSG(t.w >= 2 && t.h >= 2 && t.v.size() == (size_t) t.w * t.h, "Bitmap: invalid resolution!");
                r.rough_tex_width = t.w; r.rough_tex_height = t.h; r.rough_tex_data = t.v.data(); r.d_rough_tex_data = t.d.size() == t.v.size() ? t.d.data() : nullptr;
            }
            dst.push_back(r);
            return;
        }
        const Diffuse *d = static_cast<const Diffuse *>(b);
        psdr_bsdf_rec r{};
        r.type = 0; r.two_sided = d->m_twoSide ? 1 : 0;
        for (int k = 0; k < 3; ++k) { r.reflectance[k] = d->reflectance[k]; r.d_reflectance[k] = d->d_reflectance[k]; }
        if (d->tex_w > 0) {
            PSDR_ASSERT_MSG(d->tex_w >= 2 && d->tex_h >= 2, "Bitmap: invalid resolution!");
            PSDR_ASSERT_MSG(d->tex.size() == (size_t) 3 * d->tex_w * d->tex_h, "Bitmap: invalid data size!");
            r.tex_width = d->tex_w; r.tex_height = d->tex_h; r.tex_data = d->tex.data();
            r.d_tex_data = d->d_tex.size() == d->tex.size() ? d->d_tex.data() : nullptr;
        }
        dst.push_back(r);
    };
    auto rec_of = [&](const BSDF *b, std::vector<psdr_bsdf_rec> &dst) {
        rec_of_type(b, dst);
        std::memcpy(dst.back().tex_xf, b->uv_xf, sizeof(b->uv_xf));
        std::memcpy(dst.back().d_tex_xf, b->d_uv_xf, sizeof(b->d_uv_xf));
    };
    for (BSDF *b : m_bsdfs) rec_of(b, S.bsdfs);
    // the BSDF a NormalMap perturbs travels as an extra entry behind the scene's own (no mesh refers to it)
    for (size_t i = 0; i < m_bsdfs.size(); ++i)
        if (const NormalMap *nm = dynamic_cast<const NormalMap *>(m_bsdfs[i])) {
            PSDR_ASSERT_MSG(nm->m_bsdf != nullptr && dynamic_cast<const NormalMap *>(nm->m_bsdf) == nullptr, "NormalMap: missing or unsupported nested BSDF");
            S.bsdfs[i].nested_bsdf = (int) S.bsdfs.size();
            rec_of(nm->m_bsdf, S.bsdfs);
        }

    // secondary edges (mesh.cpp:355-369, scene.cpp:546-571): every mesh edge is kept
    const bool sec_same = m_opts.sppse > 0 && m_sec_sppse > 0 && m_sec_geo == geo_sum && m_sec_tan == tan_sum && m_sec_layout == m_layout_version;
    if (!sec_same) {
        same &= ~PSDR_SAME_SEC_EDGES;
        for (std::vector<float> *v : {&S.se_p0, &S.se_e1, &S.se_n0, &S.se_n1, &S.se_p2, &S.se_d_p0, &S.se_d_e1}) v->clear();
        S.se_boundary.clear(); S.n_sec_edges = 0; S.sec_edge_distrb = Distrb();
        m_sec_geo = geo_sum; m_sec_tan = tan_sum; m_sec_sppse = m_opts.sppse; m_sec_layout = m_layout_version;
    }
    if (m_opts.sppse > 0 && !sec_same) {
        std::vector<float> pmf;
        size_t total_edges = 0;
        for (const Mesh *mesh : m_meshes) if (mesh->m_enable_edges) total_edges += mesh->edges.size();
        pmf.resize(total_edges);
        size_t base = 0;
        for (const Mesh *mesh : m_meshes) {          // the distribution over the edges (~ their lengths): always
            if (!mesh->m_enable_edges) continue;
            const int ne = (int) mesh->edges.size();
            const float *P = mesh->vertex_positions.data();
            psdr::parallel_for((size_t) ne, 4096, [&](size_t eb, size_t ee) {
              for (size_t i = eb; i < ee; ++i) {
                const MeshEdge &e = mesh->edges[i];
                float e1[3];
                for (int k = 0; k < 3; ++k) e1[k] = P[3 * e.v1 + k] - P[3 * e.v0 + k];
                pmf[base + i] = std::sqrt(std::fmaf(e1[2], e1[2], std::fmaf(e1[1], e1[1], e1[0] * e1[0])));
              }
            });
            base += (size_t) ne;
        }
        S.n_sec_edges = (int) pmf.size();
        if (!pmf.empty()) S.sec_edge_distrb.init(pmf);
        m_sec_rows_stale = true;
        if (!m_lean) fill_sec_rows();                 // (a lean configure leaves the row arrays to ensure_full_snapshot: the device computes its rows itself)
        if (m_opts.log_level > 0) { std::ostringstream oss; oss << S.n_sec_edges << " secondary edges initialized."; log(oss.str()); }
    }

    // primary edges: unchanged when no sensor ran its configure
    if (m_seen_edges.size() != m_sensors.size()) { m_seen_edges.assign(m_sensors.size(), ~0ull); }
    for (size_t i = 0; i < m_sensors.size(); ++i) {
        if (m_seen_edges[i] != m_sensors[i]->m_edges_version) same &= ~PSDR_SAME_PRIM_EDGES;
        m_seen_edges[i] = m_sensors[i]->m_edges_version;
    }
    // bitmap parameters and the environment map's tangent: by content (FNV-style hash of the arrays the records point to)
    {
        auto hash = [](uint64_t h, const float *p, size_t n) {
            const size_t n8 = n / 2;
            const uint64_t *q = reinterpret_cast<const uint64_t *>(p);
            uint64_t a = h ^ (n * 0x9e3779b97f4a7c15ull), b = 0x2545f4914f6cdd1dull, c = 0x9e3779b97f4a7c15ull, d = 0xc2b2ae3d27d4eb4full;
            size_t i = 0;
            for (; i + 4 <= n8; i += 4) {         // four independent lanes: the loop runs at memory speed
                uint64_t w0, w1, w2, w3;
                std::memcpy(&w0, q + i, 8); std::memcpy(&w1, q + i + 1, 8); std::memcpy(&w2, q + i + 2, 8); std::memcpy(&w3, q + i + 3, 8);
                a = (a ^ w0) * 0x100000001b3ull; b = (b ^ w1) * 0x100000001b3ull; c = (c ^ w2) * 0x100000001b3ull; d = (d ^ w3) * 0x100000001b3ull;
            }
            for (; i < n8; ++i) { uint64_t w0; std::memcpy(&w0, q + i, 8); a = (a ^ w0) * 0x100000001b3ull; }
            if (n & 1) { uint32_t w0; std::memcpy(&w0, p + n - 1, 4); a = (a ^ w0) * 0x100000001b3ull; }
            uint64_t r = a;
            r = (r ^ (b >> 7)) * 0x100000001b3ull; r = (r ^ (c >> 11)) * 0x100000001b3ull; r = (r ^ (d >> 13)) * 0x100000001b3ull;
            return r;
        };
        uint64_t hb = 0xcbf29ce484222325ull;
        for (const psdr_bsdf_rec &b : S.bsdfs) {
            const float *ptrs[6] = {b.tex_data, b.d_tex_data, b.spec_tex_data, b.d_spec_tex_data, b.rough_tex_data, b.d_rough_tex_data};
            const size_t lens[6] = {(size_t) 3 * b.tex_width * b.tex_height, (size_t) 3 * b.tex_width * b.tex_height, (size_t) 3 * b.spec_tex_width * b.spec_tex_height,
                                    (size_t) 3 * b.spec_tex_width * b.spec_tex_height, (size_t) b.rough_tex_width * b.rough_tex_height, (size_t) b.rough_tex_width * b.rough_tex_height};
            for (int k = 0; k < 6; ++k) hb = ptrs[k] ? hash(hb, ptrs[k], lens[k]) : hb * 31 + 7;
            const float *pv[6] = {b.pv_specular, b.d_pv_specular, b.pv_diffuse, b.d_pv_diffuse, b.pv_roughness, b.d_pv_roughness};
            const size_t pl[6] = {(size_t) 3 * b.pv_count, (size_t) 3 * b.pv_count, (size_t) 3 * b.pv_count, (size_t) 3 * b.pv_count, (size_t) b.pv_count, (size_t) b.pv_count};
            for (int k = 0; k < 6; ++k) hb = pv[k] ? hash(hb, pv[k], pl[k]) : hb * 31 + 11;
        }
        if (hb != m_bitmap_hash) same &= ~PSDR_SAME_BITMAPS;
        m_bitmap_hash = hb;
        uint64_t he = 0xcbf29ce484222325ull;
        if (S.has_envmap && S.envmap.d_radiance) he = hash(he, S.envmap.d_radiance, (size_t) 3 * S.envmap.width * S.envmap.height);
        if (he != m_env_tan_hash) same &= ~PSDR_SAME_ENV_TANGENT;
        m_env_tan_hash = he;
    }
    if (env_cells_rebuilt) same &= ~PSDR_SAME_ENV_TEXELS;
    m_same &= same;              // (several configure_host() calls may pass before the next upload)
    m_host_ready = true;
}

// SecondaryEdgeInfo rows of the snapshot (mesh.cpp:355-369, scene.cpp:546-571): every edge of every mesh with edges, in mesh order
void Scene::fill_sec_rows() {
    Snapshot &S = snap;
    size_t total_edges = 0;
    for (const Mesh *mesh : m_meshes) if (mesh->m_enable_edges) total_edges += mesh->edges.size();
    if (m_opts.sppse <= 0) total_edges = 0;
    for (std::vector<float> *v : {&S.se_p0, &S.se_e1, &S.se_n0, &S.se_n1, &S.se_p2, &S.se_d_p0, &S.se_d_e1}) v->resize(3 * total_edges);
    S.se_boundary.resize(total_edges);
    size_t base = 0;
    if (total_edges > 0)
    for (const Mesh *mesh : m_meshes) {
        if (!mesh->m_enable_edges) continue;
        const int ne = (int) mesh->edges.size();
        const float *P = mesh->vertex_positions.data(), *dP = mesh->d_vertex_positions.data();
        psdr::parallel_for((size_t) ne, 4096, [&](size_t eb, size_t ee) {
          for (size_t i = eb; i < ee; ++i) {
            const MeshEdge &e = mesh->edges[i];
            const size_t r = base + i;
            for (int k = 0; k < 3; ++k) {
                S.se_e1[3 * r + k] = P[3 * e.v1 + k] - P[3 * e.v0 + k]; S.se_d_e1[3 * r + k] = dP[3 * e.v1 + k] - dP[3 * e.v0 + k];
                S.se_p0[3 * r + k] = P[3 * e.v0 + k]; S.se_d_p0[3 * r + k] = dP[3 * e.v0 + k];
                S.se_p2[3 * r + k] = P[3 * e.opp + k];
                S.se_n0[3 * r + k] = mesh->face_p0n[6 * (size_t) e.f0 + 3 + k];
                S.se_n1[3 * r + k] = e.f1 >= 0 ? mesh->face_p0n[6 * (size_t) e.f1 + 3 + k] : 0.f;
            }
            S.se_boundary[r] = e.f1 < 0 ? 1 : 0;
          }
        });
        base += (size_t) ne;
    }
    m_sec_rows_stale = false;
}

// what a lean configure_host left out: the TriangleInfo rows of the meshes that changed since the snapshot last held them, and the secondary-edge rows
void Scene::ensure_full_snapshot() {
    PSDR_ASSERT_MSG(m_host_ready, "configure_host() first");
    Snapshot &S = snap;
    auto put3 = [](std::vector<float> &dst, size_t row, const float *src) { dst[3 * row] = src[0]; dst[3 * row + 1] = src[1]; dst[3 * row + 2] = src[2]; };
    size_t face_offset = 0;
    for (size_t i = 0; i < m_meshes.size() && i < m_seen_geo.size(); ++i) {
        Mesh *mesh = m_meshes[i];
        const bool values = m_seen_geo[i] != mesh->m_geo_version, tangents = m_seen_tan[i] != mesh->m_tan_version;
        if (values || tangents) {
            mesh->ensure_rows();
            const size_t nf = (size_t) mesh->m_num_faces;
            psdr::parallel_for(nf, 4096, [&](size_t fb, size_t fe) {
              for (size_t f = fb; f < fe; ++f) {
                const size_t row = face_offset + f;
                const float *t = &mesh->tri[22 * f], *d = &mesh->d_tri[22 * f];
                if (values) { put3(S.p0, row, t); put3(S.e1, row, t + 3); put3(S.e2, row, t + 6); put3(S.n0, row, t + 9); put3(S.n1, row, t + 12); put3(S.n2, row, t + 15); put3(S.fn, row, t + 18); S.area[row] = t[21]; }
                if (tangents) { put3(S.d_p0, row, d); put3(S.d_e1, row, d + 3); put3(S.d_e2, row, d + 6); put3(S.d_n0, row, d + 9); put3(S.d_n1, row, d + 12); put3(S.d_n2, row, d + 15); put3(S.d_fn, row, d + 18); S.d_area[row] = d[21]; }
              }
            });
            m_seen_geo[i] = mesh->m_geo_version; m_seen_tan[i] = mesh->m_tan_version;
        }
        face_offset += (size_t) mesh->m_num_faces;
    }
    if (m_sec_rows_stale) fill_sec_rows();
}

// ------------------------------------------------------------------------------------------------
// Scene::chain_geometry (scene_host.h): the transpose of the value part of Mesh::configure / the edge assembly above, in double.
namespace {
struct V3d { double x, y, z; };
inline V3d operator+(V3d a, V3d b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3d operator-(V3d a, V3d b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3d operator*(V3d a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot3(V3d a, V3d b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3d cross3(V3d a, V3d b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3d ld3f(const float *p) { return {(double) p[0], (double) p[1], (double) p[2]}; }
// q = (M p)_xyz / (M p)_w and its transpose: g_p, and g_M accumulated
inline V3d xform_d(const double *M, V3d p, double *w_out) {
    const double h[3] = {M[0] * p.x + M[1] * p.y + M[2] * p.z + M[3], M[4] * p.x + M[5] * p.y + M[6] * p.z + M[7], M[8] * p.x + M[9] * p.y + M[10] * p.z + M[11]};
    const double w = M[12] * p.x + M[13] * p.y + M[14] * p.z + M[15];
    *w_out = w;
    return {h[0] / w, h[1] / w, h[2] / w};
}
inline V3d xform_d_T(const double *M, V3d p, V3d q, double w, V3d g_q, double *g_M) {
    const V3d g_h = g_q * (1.0 / w);
    const double g_w = -dot3(g_q, q) / w;
    const double gh[3] = {g_h.x, g_h.y, g_h.z}, pp[3] = {p.x, p.y, p.z};
    if (g_M) {
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) g_M[4 * r + c] += gh[r] * pp[c]; g_M[4 * r + 3] += gh[r]; }
        for (int c = 0; c < 3; ++c) g_M[12 + c] += g_w * pp[c];
        g_M[15] += g_w;
    }
    return {M[0] * gh[0] + M[4] * gh[1] + M[8] * gh[2] + M[12] * g_w, M[1] * gh[0] + M[5] * gh[1] + M[9] * gh[2] + M[13] * g_w, M[2] * gh[0] + M[6] * gh[1] + M[10] * gh[2] + M[14] * g_w};
}
void mul44(const double *a, const double *b, double *c) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double x = 0.0; for (int k = 0; k < 4; ++k) x += a[4 * i + k] * b[4 * k + j]; c[4 * i + j] = x; } }
} // namespace

Scene::GeometryAdjoint Scene::chain_geometry(int sensor_id, const float *g_tri, const float *g_sec, const float *g_prim, const std::vector<uint8_t> &want_mesh, bool want_camera,
                                             const double *world_to_sample) const {
    PSDR_ASSERT_MSG(m_host_ready, "configure() first");
    PSDR_ASSERT_MSG(sensor_id >= 0 && sensor_id < m_num_sensors, "Invalid sensor id!");
    GeometryAdjoint out;
    for (double &x : out.g_world_to_sample) x = 0.0;
    const PerspectiveCamera *cam = static_cast<const PerspectiveCamera *>(m_sensors[(size_t) sensor_id]);
    double w2s[16];
    for (int i = 0; i < 16; ++i) w2s[i] = world_to_sample ? world_to_sample[i] : (double) cam->rec.world_to_sample[i];
    const std::vector<int> &ids = cam->m_enable_edges ? cam->m_edges.ids : std::vector<int>();
    size_t face_offset = 0, sec_offset = 0;
    for (size_t mi = 0; mi < m_meshes.size(); ++mi) {
        Mesh *mesh = m_meshes[mi];
        const size_t nf = (size_t) mesh->m_num_faces, nv = (size_t) mesh->m_num_vertices;
        const size_t n_edges = (m_opts.sppse > 0 && mesh->m_enable_edges) ? mesh->edges.size() : 0;
        const bool wanted = mi < want_mesh.size() && want_mesh[mi] != 0;
        if (wanted || want_camera) {
            // world-space vertices in double from the leaf values, as the forward chain rule would
            double M[16];
            {
                double L[16], R0[16], Rt[16], t[16];
                for (int i = 0; i < 16; ++i) { L[i] = mesh->to_world_left[(size_t) i]; R0[i] = mesh->to_world_raw[(size_t) i]; Rt[i] = mesh->to_world_right[(size_t) i]; }
                mul44(L, R0, t); mul44(t, Rt, M);
            }
            std::vector<V3d> Vw(nv);
            std::vector<double> Ww(nv);
            for (size_t v = 0; v < nv; ++v) Vw[v] = xform_d(M, ld3f(&mesh->vertex_positions_raw[3 * v]), &Ww[v]);
            std::vector<V3d> gVw(nv, V3d{0.0, 0.0, 0.0});
            if (wanted) {
                if (!mesh->vertex_faces_current()) mesh->build_vertex_faces();
                const std::vector<int> &F = mesh->face_indices;
                // per face: what its rows send to its three corners and to the face normal sum; per vertex: the normalised normal's adjoint
                std::vector<V3d> gN(nf), gc0(nf), gc1(nf), gc2(nf), vn(nv, V3d{0.0, 0.0, 0.0}), g_nv(nv, V3d{0.0, 0.0, 0.0});
                std::vector<V3d> N(nf);
                for (size_t f = 0; f < nf; ++f) {
                    const V3d p0 = Vw[(size_t) F[3 * f]], e1 = Vw[(size_t) F[3 * f + 1]] - p0, e2 = Vw[(size_t) F[3 * f + 2]] - p0;
                    N[f] = cross3(e1, e2);
                }
                const std::vector<int> &vb = mesh->vertex_face_begin(), &vi = mesh->vertex_face_item();
                psdr::parallel_for(nv, 4096, [&](size_t b, size_t e) {
                    for (size_t v = b; v < e; ++v) {
                        V3d acc{0.0, 0.0, 0.0}, g{0.0, 0.0, 0.0};
                        for (int k = vb[v]; k < vb[v + 1]; ++k) {
                            const size_t f = (size_t) (vi[(size_t) k] >> 2);
                            const int corner = vi[(size_t) k] & 3;
                            acc = acc + N[f];
                            g = g + ld3f(g_tri + 22 * (face_offset + f) + 9 + 3 * corner);       // the row block n0 / n1 / n2 of this corner
                        }
                        vn[v] = acc; g_nv[v] = g;
                    }
                });
                std::vector<V3d> g_vn(nv);
                for (size_t v = 0; v < nv; ++v) {
                    const double len = std::sqrt(dot3(vn[v], vn[v]));
                    if (!(len > 0.0)) { g_vn[v] = V3d{0.0, 0.0, 0.0}; continue; }
                    const V3d n = vn[v] * (1.0 / len);
                    g_vn[v] = (g_nv[v] - n * dot3(n, g_nv[v])) * (1.0 / len);
                }
                psdr::parallel_for(nf, 4096, [&](size_t b, size_t e) {
                    for (size_t f = b; f < e; ++f) {
                        const float *row = g_tri + 22 * (face_offset + f);
                        const size_t i0 = (size_t) F[3 * f], i1 = (size_t) F[3 * f + 1], i2 = (size_t) F[3 * f + 2];
                        const V3d p0 = Vw[i0], e1 = Vw[i1] - p0, e2 = Vw[i2] - p0;
                        const double a = std::sqrt(dot3(N[f], N[f]));
                        V3d gn = g_vn[i0] + g_vn[i1] + g_vn[i2];
                        if (a > 0.0) {
                            const V3d fn = N[f] * (1.0 / a), g_fn = ld3f(row + 18);
                            gn = gn + (g_fn - fn * dot3(fn, g_fn)) * (1.0 / a) + fn * (0.5 * (double) row[21]);
                        }
                        const V3d g_e1 = ld3f(row + 3) + cross3(e2, gn), g_e2 = ld3f(row + 6) + cross3(gn, e1);
                        gc0[f] = ld3f(row) - g_e1 - g_e2; gc1[f] = g_e1; gc2[f] = g_e2;
                    }
                });
                for (size_t f = 0; f < nf; ++f) {
                    gVw[(size_t) F[3 * f]] = gVw[(size_t) F[3 * f]] + gc0[f];
                    gVw[(size_t) F[3 * f + 1]] = gVw[(size_t) F[3 * f + 1]] + gc1[f];
                    gVw[(size_t) F[3 * f + 2]] = gVw[(size_t) F[3 * f + 2]] + gc2[f];
                }
                // secondary-edge rows: p0 = Vw[v0], e1 = Vw[v1] - Vw[v0]
                for (size_t i = 0; i < n_edges; ++i) {
                    const MeshEdge &ed = mesh->edges[i];
                    const float *row = g_sec + 6 * (sec_offset + i);
                    const V3d gp = ld3f(row), ge = ld3f(row + 3);
                    gVw[(size_t) ed.v0] = gVw[(size_t) ed.v0] + gp - ge;
                    gVw[(size_t) ed.v1] = gVw[(size_t) ed.v1] + ge;
                }
            }
            // primary-edge rows of this mesh: (q0.xy, q1.xy) = world_to_sample . Vw[v0 / v1]
            for (size_t k = 0; 3 * k + 2 < ids.size(); ++k) {
                if ((size_t) ids[3 * k] != mi) continue;
                for (int side = 0; side < 2; ++side) {
                    const size_t v = (size_t) ids[3 * k + 1 + side];
                    const V3d gq{(double) g_prim[4 * k + 2 * side], (double) g_prim[4 * k + 2 * side + 1], 0.0};
                    double w;
                    const V3d q = xform_d(w2s, Vw[v], &w);
                    const V3d gp = xform_d_T(w2s, Vw[v], q, w, gq, want_camera ? out.g_world_to_sample : nullptr);
                    if (wanted) gVw[v] = gVw[v] + gp;
                }
            }
            if (wanted) {
                MeshAdjoint ma;
                ma.mesh = (int) mi;
                for (double &x : ma.g_to_world) x = 0.0;
                ma.g_vertices.assign(3 * nv, 0.0);
                for (size_t v = 0; v < nv; ++v) {
                    const V3d gp = xform_d_T(M, ld3f(&mesh->vertex_positions_raw[3 * v]), Vw[v], Ww[v], gVw[v], ma.g_to_world);
                    ma.g_vertices[3 * v] = gp.x; ma.g_vertices[3 * v + 1] = gp.y; ma.g_vertices[3 * v + 2] = gp.z;
                }
                out.meshes.push_back(std::move(ma));
            }
        }
        face_offset += nf;
        sec_offset += n_edges;
    }
    return out;
}

// assemble the C-ABI snapshot and hand it to the HIP library (replaces Scene_OptiX::configure)
// the configured snapshot as the C ABI's struct (pointers into `snap`, the sensors' edge lists and the meshes)
void Scene::fill_snapshot(psdr_scene_snapshot &sn, bool full) {
    PSDR_ASSERT_MSG(m_host_ready, "configure_host() first");
    if (full) ensure_full_snapshot();
    Snapshot &S = snap;
    S.sensors.clear();
    sn = psdr_scene_snapshot{};
    sn.abi_version = PSDR_HIP_ABI_VERSION;
    sn.width = m_opts.width; sn.height = m_opts.height; sn.spp = m_opts.spp; sn.sppe = m_opts.sppe; sn.sppse = m_opts.sppse;
    psdr_triangles &t = sn.tris;
    t.n_triangles = (int) S.area.size();
    t.p0 = S.p0.data(); t.e1 = S.e1.data(); t.e2 = S.e2.data(); t.n0 = S.n0.data(); t.n1 = S.n1.data(); t.n2 = S.n2.data();
    t.face_normal = S.fn.data(); t.face_area = S.area.data(); t.uv = S.uv.data(); t.mesh_id = S.mesh_id.data(); t.use_face_normal = S.flat.data();
    t.face_indices = S.face_indices.data();
    t.d_p0 = S.d_p0.data(); t.d_e1 = S.d_e1.data(); t.d_e2 = S.d_e2.data(); t.d_n0 = S.d_n0.data(); t.d_n1 = S.d_n1.data(); t.d_n2 = S.d_n2.data();
    t.d_face_normal = S.d_fn.data(); t.d_face_area = S.d_area.data();
    sn.n_meshes = (int) S.meshes.size(); sn.meshes = S.meshes.data();
    sn.n_bsdfs = (int) S.bsdfs.size(); sn.bsdfs = S.bsdfs.data();
    sn.n_emitters = (int) S.emitters.size(); sn.emitters = S.emitters.data();
    sn.envmap = S.has_envmap ? &S.envmap : nullptr;
    sn.emitter_pmf = S.emitters_distrb.pmf.data(); sn.emitter_cmf = S.emitters_distrb.cmf.data(); sn.emitter_sum = S.emitters_distrb.sum;
    sn.n_face_distrb = (int) S.face_pmf.size(); sn.face_pmf = S.face_pmf.data(); sn.face_cmf = S.face_cmf.data();
    psdr_sec_edges &se = sn.sec_edges;
    se.n_edges = S.n_sec_edges;
    if (S.n_sec_edges > 0) {
        se.p0 = S.se_p0.data(); se.e1 = S.se_e1.data(); se.n0 = S.se_n0.data(); se.n1 = S.se_n1.data(); se.p2 = S.se_p2.data();
        se.is_boundary = S.se_boundary.data(); se.d_p0 = S.se_d_p0.data(); se.d_e1 = S.se_d_e1.data();
        se.pmf = S.sec_edge_distrb.pmf.data(); se.cmf = S.sec_edge_distrb.cmf.data(); se.sum = S.sec_edge_distrb.sum;
    }
    for (Sensor *s : m_sensors) {
        PerspectiveCamera *cam = static_cast<PerspectiveCamera *>(s);
        psdr_sensor_rec r = cam->rec;
        r.n_edges = cam->m_enable_edges ? (int) cam->m_edges.length.size() : 0;
        if (r.n_edges > 0) {
            const PrimaryEdges &pe = cam->m_edges;
            r.edge_p0 = pe.p0.data(); r.edge_p1 = pe.p1.data(); r.d_edge_p0 = pe.d_p0.data(); r.d_edge_p1 = pe.d_p1.data();
            r.edge_normal = pe.normal.data(); r.edge_length = pe.length.data(); r.edge_pmf = pe.distrb.pmf.data(); r.edge_cmf = pe.distrb.cmf.data();
            r.edge_sum = pe.distrb.sum;
        }
        S.sensors.push_back(r);
    }
    sn.n_sensors = (int) S.sensors.size(); sn.sensors = S.sensors.data();
    // what the rows of every mesh are a function of: with it the device computes a moved mesh's triangle and secondary-edge rows itself (psdr_mesh_geometry, include/psdr_hip.h)
    {
        const size_t nm = m_meshes.size();
        if (m_up_geo.size() != nm) { m_up_geo.assign(nm, ~0ull); m_up_tan.assign(nm, ~0ull); }
        m_geometry.assign(nm, psdr_mesh_geometry{});
        bool complete = true;
        for (size_t i = 0; i < nm; ++i) {
            Mesh *mesh = m_meshes[i];
            psdr_mesh_geometry &g = m_geometry[i];
            if (!mesh->vertex_faces_current() || mesh->d_vertex_positions_raw.size() != mesh->vertex_positions_raw.size()) { complete = false; break; }
            const bool has_edges = m_opts.sppse > 0 && mesh->m_enable_edges;
            g.n_vertices = mesh->m_num_vertices; g.n_faces = mesh->m_num_faces; g.n_edges = has_edges ? (int) mesh->edges.size() : 0;
            g.vertices_raw = mesh->vertex_positions_raw.data(); g.d_vertices_raw = mesh->d_vertex_positions_raw.data();
            std::memcpy(g.to_world, mesh->m_tw, 64); std::memcpy(g.d_to_world, mesh->m_d_tw, 64);
            g.faces = mesh->face_indices.data();
            g.vf_begin = mesh->vertex_face_begin().data(); g.vf_item = mesh->vertex_face_item().data();
            static_assert(sizeof(MeshEdge) == 5 * sizeof(int32_t), "psdr_mesh_geometry.edges: v0 v1 f0 f1 opp");
            g.edges = has_edges && !mesh->edges.empty() ? reinterpret_cast<const int32_t *>(mesh->edges.data()) : nullptr;
            g.mesh_id = mesh->m_mesh_id; g.use_face_normals = mesh->m_use_face_normals ? 1 : 0;
            g.topology_version = (mesh->m_topo_version << 2) | (has_edges ? 2u : 0u) | 1u;
            g.moved = (m_up_geo[i] != mesh->m_geo_version || m_up_tan[i] != mesh->m_tan_version) ? 1 : 0;
        }
        sn.geometry = complete ? m_geometry.data() : nullptr;
    }
    // rows_valid = 0: the row arrays of psdr_triangles / psdr_sec_edges are behind (a lean configure) - psdr_hip_scene_update either computes the rows on the device or
    // answers PSDR_HIP_NEED_ROWS without having changed anything
    bool stale = m_sec_rows_stale && S.n_sec_edges > 0;
    for (size_t i = 0; i < m_meshes.size() && i < m_seen_geo.size(); ++i) stale = stale || m_seen_geo[i] != m_meshes[i]->m_geo_version || m_seen_tan[i] != m_meshes[i]->m_tan_version;
    sn.rows_valid = stale ? 0 : 1;
    if (stale) PSDR_ASSERT_MSG(sn.geometry != nullptr, "lean snapshot without geometry");
}

// test aid: words of the device's triangle / secondary-edge rows that differ from the rows the host path would write (psdr_hip_scene_check_rows)
int64_t Scene::check_device_rows() {
    PSDR_ASSERT_MSG(m_hip != nullptr && m_configured, "configure() first");
    psdr_scene_snapshot sn;
    fill_snapshot(sn);
    int64_t bad = -1;
    hip_check(psdr_hip_scene_check_rows(m_hip, &sn, &bad));
    return bad;
}

void Scene::upload() {
    psdr_scene_snapshot sn;
    // a lean configure_host left the rows out: the device is asked first whether it can do without them
    const bool try_lean = m_lean && m_hip != nullptr && !m_always_rebuild;
    fill_snapshot(sn, !try_lean);
    // the device copy: created once, then updated in place - only what changed since the previous upload is rewritten and sent, the tree is kept
    // (refitted on the device when triangles moved); psdr_hip_scene_update, include/psdr_hip.h
    if (m_hip != nullptr && m_always_rebuild) release_device();
    if (m_hip == nullptr) {
        hip_check(psdr_hip_scene_create(&sn, &m_hip));
        hip_check(psdr_hip_scene_last_update(m_hip, &m_last_update));
    } else {
        // a failed update leaves the device scene half-written (and poisoned, scene_build.hip): the PSDR_SAME_* bits of the NEXT configure() would be relative to a
        // snapshot the device never received, so the handle is dropped and the next upload creates the scene again
        int rc = psdr_hip_scene_update(m_hip, &sn, m_same, &m_last_update);
        if (rc == PSDR_HIP_NEED_ROWS) {            // (the device scene is as it was: compute the rows after all and send the complete snapshot)
            fill_snapshot(sn, true);
            rc = psdr_hip_scene_update(m_hip, &sn, m_same, &m_last_update);
        }
        if (rc) {
            const std::string why = psdr_hip_last_error();
            release_device();
            m_same = 0;
            m_up_geo.clear(); m_up_tan.clear(); m_device_rows_ok = false;
            throw Exception("libpsdr_hip: " + why);
        }
    }
    m_same = PSDR_SAME_TRIANGLES | PSDR_SAME_TRI_TANGENTS | PSDR_SAME_SEC_EDGES | PSDR_SAME_PRIM_EDGES | PSDR_SAME_ENV_TEXELS | PSDR_SAME_ENV_TANGENT | PSDR_SAME_BITMAPS;
    for (size_t i = 0; i < m_meshes.size() && i < m_up_geo.size(); ++i) { m_up_geo[i] = m_meshes[i]->m_geo_version; m_up_tan[i] = m_meshes[i]->m_tan_version; }
    m_device_rows_ok = sn.geometry != nullptr;        // (the next configure_host may be lean: the device has the topology or will take it with the next update)
    m_configured = true;
}

// ------------------------------------------------------------------------------------------------
// Integrator::renderC / renderD, reference src/integrator/integrator.cpp:12-100
// FieldExtractionIntegrator's object filter as a mesh index (-1 = none).  Mesh::get_obj_mask, reference mesh.h:49-63: a mesh with an id is matched by
// name, the others by index
int field_object_index(const Scene &scene, const Integrator &it) {
    const std::string obj = it.field_object();
    if (obj.empty()) return -1;
    int by_index = -1;
    try { by_index = std::stoi(obj); } catch (...) { by_index = -1; }
    for (const Mesh *m : scene.m_meshes)
        if (!m->m_id.empty() ? (m->m_id == obj) : (m->m_mesh_id == by_index)) return m->m_mesh_id;
    return 1 << 30;        // nothing matches: an empty field
}

static void fill_args(psdr_render_args &a, const Scene &scene, const Integrator &it, int sensor_id, uintptr_t pix_ids, int n_pix, int rank, int count) {
    std::memset(&a, 0, sizeof(a));
    a.sensor_id = sensor_id; a.max_depth = it.max_depth(); a.hide_emitters = it.hide_emitters() ? 1 : 0;
    for (int k = 0; k < 3; ++k) { a.samplers[k].seed = scene.m_samplers[k].seed; a.samplers[k].skip = scene.m_samplers[k].skip; }
    a.pix_ids = reinterpret_cast<const int32_t *>(pix_ids); a.n_pix = n_pix;
    a.shard_rank = rank; a.shard_count = count; a.zero_output = 1;
    a.guiding = it.guiding(sensor_id);
    a.direct_mode = it.direct_mis() + 1;
    a.field_mode = it.field() + 1;
    a.field_object = field_object_index(scene, it);
    a.intensity = it.intensity(false); a.d_intensity = it.intensity(true);
    a.skip_static_edges = it.m_trace_static_edges ? 0 : 1;
    a.shard_mode = it.m_shard_mode;
}

void Integrator::renderC(const Scene &scene, int sensor_id, int seed, uintptr_t pix_ids, int n_pix, uintptr_t out, uintptr_t stream, int rank, int count) const {
    using namespace std::chrono;
    const auto start_time = high_resolution_clock::now();
    const RenderOption &opts = scene.m_opts;
    PSDR_ASSERT_MSG(!(pix_ids != 0 && seed == -1), "While using batch rendering, seed must be set!");
    PSDR_ASSERT_MSG(scene.is_ready(), "Input scene must be configured!");
    PSDR_ASSERT_MSG(sensor_id >= 0 && sensor_id < scene.m_num_sensors, "Invalid sensor id!");
    const int64_t npx = pix_ids ? n_pix : (int64_t) opts.width * opts.height;
    PSDR_ASSERT(npx * opts.spp <= std::numeric_limits<int>::max());
    if (seed != -1) scene.m_samplers[0] = SamplerState{true, npx * opts.spp, (uint64_t) (int64_t) seed, 0};
    psdr_render_args a;
    fill_args(a, scene, *this, sensor_id, pix_ids, n_pix, rank, count);
    hip_check(psdr_hip_render_c(scene.m_hip, &a, reinterpret_cast<float *>(out), reinterpret_cast<void *>(stream)));
    if (opts.spp > 0) scene.m_samplers[0].skip += 2 + (uint64_t) draws_per_level() * (uint64_t) max_depth();
    if (opts.log_level) {
        std::ostringstream oss;
        oss << "Rendered in " << duration_cast<duration<double>>(high_resolution_clock::now() - start_time).count() << " seconds.";
        log(oss.str());
    }
}

void Integrator::renderD(const Scene &scene, int sensor_id, int seed, uintptr_t pix_ids, int n_pix, uintptr_t out, uintptr_t dout, uintptr_t stream,
                         int rank, int count, int terms) const {
    using namespace std::chrono;
    const auto start_time = high_resolution_clock::now();
    const RenderOption &opts = scene.m_opts;
    PSDR_ASSERT_MSG(!(pix_ids != 0 && seed == -1), "While using batch rendering, seed must be set!");
    PSDR_ASSERT_MSG(scene.is_ready(), "Input scene must be configured!");
    PSDR_ASSERT_MSG(sensor_id >= 0 && sensor_id < scene.m_num_sensors, "Invalid sensor id!");
    const int64_t num_pixels = (int64_t) opts.height * opts.width;
    const int64_t npx = pix_ids ? n_pix : num_pixels;
    if (seed != -1) {
        if (opts.spp > 0) scene.m_samplers[0] = SamplerState{true, npx * opts.spp, (uint64_t) (int64_t) seed, 0};
        if (opts.sppe > 0) scene.m_samplers[1] = SamplerState{true, num_pixels * opts.sppe, (uint64_t) (int64_t) seed, 0};
        if (opts.sppse > 0) scene.m_samplers[2] = SamplerState{true, num_pixels * opts.sppse, (uint64_t) (int64_t) seed, 0};
    }
    psdr_render_args a;
    fill_args(a, scene, *this, sensor_id, pix_ids, n_pix, rank, count);
    // terms: bits 0-2 = the terms to launch; bits 4-6 (optional) = the terms whose samplers advance, as in a full renderD.
    // The Python layer renders the primal image with the interior term alone (both edge terms have zero primal,
    // integrator.cpp:192, path.cpp:265) and computes derivatives later from the recorded sampler state.
    const int launch = terms & 7;
    terms = ((terms >> 4) & 7) ? ((terms >> 4) & 7) : launch;
    if (field() >= 0) terms &= ~PSDR_TERM_SECONDARY;         // Integrator::render_secondary_edges is a no-op for the first-hit integrators
    a.terms = launch & terms;
    if (a.terms)
        hip_check(psdr_hip_render_d_fwd(scene.m_hip, &a, reinterpret_cast<float *>(out), reinterpret_cast<float *>(dout), reinterpret_cast<void *>(stream)));
    const PerspectiveCamera *cam = static_cast<const PerspectiveCamera *>(scene.m_sensors[sensor_id]);
    if (opts.spp > 0 && (terms & PSDR_TERM_INTERIOR)) scene.m_samplers[0].skip += 2 + (uint64_t) draws_per_level() * (uint64_t) max_depth();
    if (opts.sppe > 0 && cam->m_enable_edges && (terms & PSDR_TERM_PRIMARY) && !pix_ids) scene.m_samplers[1].skip += 1 + 2 * (uint64_t) draws_per_level() * (uint64_t) max_depth();
    if (opts.sppse > 0 && (terms & PSDR_TERM_SECONDARY) && !pix_ids) scene.m_samplers[2].skip += 3;
    if (opts.log_level) {
        std::ostringstream oss;
        oss << "Rendered in " << duration_cast<duration<double>>(high_resolution_clock::now() - start_time).count() << " seconds.";
        log(oss.str());
    }
}

FieldExtractionIntegrator::FieldExtractionIntegrator(const std::string &spec) {
    std::istringstream iss(spec);
    iss >> m_field_name;
    iss >> m_object;
    static const char *names[8] = {"silhouette", "position", "depth", "geoNormal", "shNormal", "uv", "bsdf", "segmentation"};
    m_field = -1;
    for (int i = 0; i < 8; ++i) if (m_field_name == names[i]) m_field = i;
    PSDR_ASSERT_MSG(m_field >= 0, std::string("Unsupported field: ") + m_field_name);
}

PathTracer::PathTracer(int max_depth) : m_max_depth(max_depth) { PSDR_ASSERT(max_depth >= 0); }
PathTracer::~PathTracer() { for (psdr_hip_guiding *g : m_warpper) if (g) psdr_hip_guiding_destroy(g); }
const psdr_hip_guiding *PathTracer::guiding(int sensor_id) const {
    return (sensor_id >= 0 && sensor_id < (int) m_warpper.size()) ? m_warpper[sensor_id] : nullptr;
}
// PathTracer::preprocess_secondary_edges, reference src/integrator/path.cpp:130-168
void PathTracer::preprocess_secondary_edges(const Scene &scene, int sensor_id, const std::array<int, 4> &reso, int nrounds, int seed) {
    PSDR_ASSERT(nrounds > 0);
    PSDR_ASSERT_MSG(scene.is_ready(), "Scene needs to be configured!");
    PSDR_ASSERT_MSG(sensor_id >= 0 && sensor_id < scene.m_num_sensors, "Invalid sensor id!");
    if ((int) m_warpper.size() != scene.m_num_sensors) m_warpper.resize(scene.m_num_sensors, nullptr);
    if (m_warpper[sensor_id]) { psdr_hip_guiding_destroy(m_warpper[sensor_id]); m_warpper[sensor_id] = nullptr; }
    hip_check(psdr_hip_guiding_build(scene.m_hip, sensor_id, m_max_depth, reso.data(), nrounds, seed, &m_warpper[sensor_id], nullptr));
}
std::vector<float> PathTracer::guiding_mass(int sensor_id) const {
    const psdr_hip_guiding *g = guiding(sensor_id);
    PSDR_ASSERT_MSG(g != nullptr, "no guiding distribution for this sensor");
    std::vector<float> out(psdr_hip_guiding_num_cells(g));
    hip_check(psdr_hip_guiding_mass(g, out.data(), (int) out.size()));
    return out;
}

} // namespace psdr_host
