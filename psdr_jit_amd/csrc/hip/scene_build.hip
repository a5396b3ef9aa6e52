// scene_build.hip — psdr_hip_scene_create / psdr_hip_scene_update: from a configured-scene snapshot to the device-resident scene.
//
// Replaces what the reference does at the end of every Scene::configure() (src/scene/scene.cpp:311-599): the jit uploads of the
// concatenated scene arrays and Scene_OptiX::configure (src/scene/scene_optix.cpp:265-332), which rebuilds the OptiX geometry
// acceleration structure on the device (optixAccelBuild, :310).  The reference's usage pattern is set_transform -> configure ->
// renderD -> backward once per optimisation step (README.md:87-106), so this is on the path of every step:
//
//   * nothing geometric changed (a colour, a texel, the camera, a forward tangent): the tree and the triangle order stay, only the
//     sections of the scene blob the change touches are rewritten in the pinned host copy and sent;
//   * vertices moved, topology unchanged: the triangle sections are rewritten and the 4-wide tree is REFITTED on the device, bottom
//     up, one small kernel launch per level of the tree (k_refit_level), with the builder's own box and quantisation code (bvh.h), so
//     a refitted node holds the bytes the builder would produce for the same topology.  The refit also evaluates the tree's SAH cost;
//   * the cost exceeds kRebuildFactor x the cost at build time, or the triangle count changed: the tree is built again (bvh.h::
//     build_bvh, host threads) and every section is rewritten.
// The hit of a ray is defined by the exact triangle test alone (trav4.h), so none of this can change a result - only how many nodes a
// ray visits (tests/test_gpu_configure.py: hits after refits against the oracle's brute-force definition, bit-equal).
#include <chrono>
#include <cstdio>

#include "scene_obj.h"
#include "bvh.h"
#include "filter.h"
#include "../host/hnum.h"

using namespace psdr;

static int fail(const std::string &msg) { return psdr::api_fail(msg); }

constexpr double kRebuildFactor = 1.4;          // a refitted tree whose SAH cost exceeds this multiple of the cost it was built with is built again

static inline void put4(float *b, size_t word, float x, float y, float z, float w) { float *q = b + 4 * word; q[0] = x; q[1] = y; q[2] = z; q[3] = w; }
static inline float ibits(int32_t v) { float f; std::memcpy(&f, &v, 4); return f; }
static inline size_t words_for_floats(size_t n) { return (n + 3) / 4; }
static inline double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

// Conservative screen-space coverage of the scene for one sensor: bit (y * width + x) = some triangle's projection (world_to_sample, a projective map: the
// image of a triangle in front of the camera is the triangle of its projected vertices) comes within a quarter of a pixel of the pixel's square.  Row by row:
// the x-extent of the triangle inside the (padded) row slab.  false = no mask (a triangle crosses the camera plane: its image is not a triangle).
static bool build_live_mask(const psdr_triangles &tr, const float *w2s, int W, int H, std::vector<unsigned> &mask) {
    const double pad = 0.25;
    mask.assign(((size_t) W * H + 31) / 32, 0u);
    for (int t = 0; t < tr.n_triangles; ++t) {
        double X[3], Y[3];
        int behind = 0;
        for (int v = 0; v < 3; ++v) {
            double p[3];
            for (int c = 0; c < 3; ++c) p[c] = (double) tr.p0[3 * t + c] + (v == 1 ? (double) tr.e1[3 * t + c] : (v == 2 ? (double) tr.e2[3 * t + c] : 0.0));
            if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) return false;
            const double x = w2s[0] * p[0] + w2s[1] * p[1] + w2s[2] * p[2] + w2s[3], y = w2s[4] * p[0] + w2s[5] * p[1] + w2s[6] * p[2] + w2s[7];
            const double w = w2s[12] * p[0] + w2s[13] * p[1] + w2s[14] * p[2] + w2s[15];
            if (!(w > 1e-9)) { ++behind; continue; }
            X[v] = x / w * W; Y[v] = y / w * H;
        }
        if (behind == 3) continue;              // behind the camera: no forward ray reaches it
        if (behind != 0) return false;
        const double ymin = std::min(Y[0], std::min(Y[1], Y[2])) - pad, ymax = std::max(Y[0], std::max(Y[1], Y[2])) + pad;
        if (!(ymax >= 0.0 && ymin < (double) H)) continue;
        const int r0 = (int) std::max(0.0, std::floor(ymin)), r1 = (int) std::min((double) (H - 1), std::floor(ymax));
        for (int row = r0; row <= r1; ++row) {
            const double lo = row - pad, hi = row + 1 + pad;
            double xmin = 1e300, xmax = -1e300;
            for (int v = 0; v < 3; ++v) {
                if (Y[v] >= lo && Y[v] <= hi) { xmin = std::min(xmin, X[v]); xmax = std::max(xmax, X[v]); }
                const int u = (v + 1) % 3;
                const double dy = Y[u] - Y[v];
                if (dy != 0.0)
                    for (double yc : {lo, hi}) {
                        const double s = (yc - Y[v]) / dy;
                        if (s >= 0.0 && s <= 1.0) { const double xc = X[v] + s * (X[u] - X[v]); xmin = std::min(xmin, xc); xmax = std::max(xmax, xc); }
                    }
            }
            if (xmin > xmax) continue;
            xmin -= pad; xmax += pad;
            if (!(xmax >= 0.0 && xmin < (double) W)) continue;
            const int c0 = (int) std::max(0.0, std::floor(xmin)), c1 = (int) std::min((double) (W - 1), std::floor(xmax));
            // bits [b0, b1] of the mask, a word at a time (a wall of the Cornell box at 2048 x 2048 is four million pixels per triangle)
            const size_t b0 = (size_t) row * W + c0, b1 = (size_t) row * W + c1;
            for (size_t wi = b0 >> 5; wi <= (b1 >> 5); ++wi) {
                const unsigned lo_bit = wi == (b0 >> 5) ? (unsigned) (b0 & 31) : 0u, hi_bit = wi == (b1 >> 5) ? (unsigned) (b1 & 31) : 31u;
                const unsigned m_hi = hi_bit == 31u ? 0xffffffffu : ((1u << (hi_bit + 1u)) - 1u);
                mask[wi] |= m_hi & ~((1u << lo_bit) - 1u);
            }
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// refit of the 4-wide tree (bvh.h::Bvh4Result layout) after the triangles moved
//
// One thread per node of one level (`ids`: node indices of equal height above the leaves, children always in an earlier launch): the
// float box of every child - the union of a leaf's padded triangle boxes (from the traversal rows the update has just sent), or the box
// an earlier level stored for an inner child - then the node's ten box words by bvh.h::bvh_quantise, the child codes untouched.
// `cost` accumulates sum half_area(child box) x (triangles of a leaf | 1): the tree's SAH cost up to the division by the root's area.
__global__ void k_refit_level(float4 *__restrict__ blob, int nodes_off, int trav_off, const int *__restrict__ ids, int count, unsigned leaf_bit,
                              float *__restrict__ fbox, double *__restrict__ cost) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    double c = 0.0;
    if (tid < count) {
        const int node = ids[tid];
        float *q = reinterpret_cast<float *>(blob + nodes_off + (kNodeFloats / 4) * (size_t) node);
        const uint32_t *codes = reinterpret_cast<const uint32_t *>(q) + kNodeCodeOff;
        float los[kBvhW][3], his[kBvhW][3];
        float ulo[3] = {3e38f, 3e38f, 3e38f}, uhi[3] = {-3e38f, -3e38f, -3e38f};
        int nc = 0;
        for (int k = 0; k < kBvhW; ++k) {
            const uint32_t code = codes[k];
            if (code == 0xffffffffu) break;
            ++nc;
            float lo[3], hi[3];
            float weight = 1.f;
            if (code & leaf_bit) {
                const int payload = (int) (code & (leaf_bit - 1u)), first = payload >> 2, cnt = (payload & 3) + 1;
                weight = (float) cnt;
                for (int a = 0; a < 3; ++a) { lo[a] = 3e38f; hi[a] = -3e38f; }
                for (int t = 0; t < cnt; ++t) {
                    const float4 w0 = blob[trav_off + 3 * (size_t) (first + t)], w1 = blob[trav_off + 3 * (size_t) (first + t) + 1], w2 = blob[trav_off + 3 * (size_t) (first + t) + 2];
                    const float p0[3] = {w0.x, w0.y, w0.z}, e1[3] = {w0.w, w1.x, w1.y}, e2[3] = {w1.z, w1.w, w2.x};
                    float tl[3], th[3];
                    bvh_tri_box(p0, e1, e2, tl, th);
                    for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], tl[a]); hi[a] = fmaxf(hi[a], th[a]); }
                }
            } else {
                const float *b = fbox + 6 * (size_t) code;
                for (int a = 0; a < 3; ++a) { lo[a] = b[a]; hi[a] = b[3 + a]; }
            }
            for (int a = 0; a < 3; ++a) { los[k][a] = lo[a]; his[k][a] = hi[a]; ulo[a] = fminf(ulo[a], lo[a]); uhi[a] = fmaxf(uhi[a], hi[a]); }
            c += (double) bvh_half_area(lo, hi) * (double) weight;
        }
        if (nc > 0) {
            bvh_quantise(los, his, nc, q);
            float *b = fbox + 6 * (size_t) node;
            for (int a = 0; a < 3; ++a) { b[a] = ulo[a]; b[3 + a] = uhi[a]; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c != 0.0) atomicAdd(cost, c);
}

// ------------------------------------------------------------------------------------------------
// the tree: built on the host (bvh.h), its topology kept for refits
static int tree_build(psdr_hip_scene *sc, const psdr_triangles &tr, std::vector<float> &nodes_out) {
    const int n = tr.n_triangles;
    BvhResult bvh;
    build_bvh(tr.p0, tr.e1, tr.e2, n, bvh);
    Bvh4Result bvh4;
    build_bvh4(bvh, n, bvh4);
    sc->order = bvh.order;
    sc->orig2slot.assign((size_t) n, 0);
    for (int slot = 0; slot < n; ++slot) sc->orig2slot[(size_t) bvh.order[(size_t) slot]] = slot;
    sc->tree_tris = n;
    sc->tree_max_stack = bvh4.max_stack; sc->tree_ref_bits = bvh4.ref_bits;
    sc->n_leaves = bvh.n_leaves; sc->max_depth = bvh4.max_depth;
    sc->T.n_nodes = bvh4.n_nodes; sc->T.ref_bits = bvh4.ref_bits;
    sc->cost_built = bvh4.cost;
    // refit schedule: node ids by height above the leaves (counting sort), one launch per height
    int levels = 0;
    for (int h : bvh4.height) levels = std::max(levels, h + 1);
    sc->refit_level_begin.assign((size_t) levels + 1, 0);
    for (int h : bvh4.height) sc->refit_level_begin[(size_t) h + 1]++;
    for (int l = 0; l < levels; ++l) sc->refit_level_begin[(size_t) l + 1] += sc->refit_level_begin[(size_t) l];
    std::vector<int> ids((size_t) std::max(1, bvh4.n_nodes)), at(sc->refit_level_begin.begin(), sc->refit_level_begin.end());
    for (int i = 0; i < bvh4.n_nodes; ++i) ids[(size_t) at[(size_t) bvh4.height[(size_t) i]]++] = i;
    if (sc->refit_order.upload(ids.data(), ids.size() * sizeof(int))) return 1;
    if (sc->refit_box.ensure(sizeof(float) * 6 * (size_t) std::max(1, bvh4.n_nodes))) return 1;
    if (sc->refit_cost.ensure(sizeof(double))) return 1;
    nodes_out.swap(bvh4.nodes);
    return 0;
}

// refit the device tree from the traversal rows in the device blob; -> *cost = the refitted tree's SAH cost (normalised like Bvh4Result::cost)
static int tree_refit(psdr_hip_scene *sc, double *cost) {
    const SceneTables &T = sc->T;
    hipStream_t st = nullptr;
    HIPCHK(hipMemsetAsync(sc->refit_cost.p, 0, sizeof(double), st));
    const unsigned leaf_bit = 1u << (T.ref_bits - 1);
    for (size_t l = 0; l + 1 < sc->refit_level_begin.size(); ++l) {
        const int begin = sc->refit_level_begin[l], count = sc->refit_level_begin[l + 1] - begin;
        if (count <= 0) continue;
        hipLaunchKernelGGL(k_refit_level, dim3((unsigned) ((count + 127) / 128)), dim3(128), 0, st, (float4 *) sc->blob.p, T.nodes_off, T.trav_off,
                           sc->refit_order.as<int>() + begin, count, leaf_bit, (float *) sc->refit_box.p, (double *) sc->refit_cost.p);
    }
    HIPCHK(hipGetLastError());
    double sum = 0.0;
    float root[6];
    HIPCHK(hipMemcpyAsync(&sum, sc->refit_cost.p, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(root, sc->refit_box.p, sizeof(root), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const double area = std::max((double) bvh_half_area(root, root + 3), 1e-300);
    *cost = sum / area;
    return 0;
}

// a device array outside the blob: (re)allocated when its size changed, copied unless the caller vouches for the content
template <typename P> static int sync_named(psdr_hip_scene *sc, const std::string &key, const void *src, size_t bytes, bool same, const P *&out, psdr_update_info &info) {
    out = nullptr;
    if (!src) return 0;
    DevBuf &b = sc->buf(key);
    bool moved = false;
    if (b.ensure(bytes, &moved)) return 1;
    if (moved) info.reallocated++;
    if (moved || !same) {
        HIPCHK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));          // (synchronous: the sources are the caller's arrays or temporaries of scene_sync)
        info.bytes_uploaded += (int64_t) bytes;
    }
    out = reinterpret_cast<const P *>(b.p);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// GEOMETRY ON THE DEVICE (round 6; psdr_mesh_geometry, include/psdr_hip.h).  The reference's Mesh::configure / process_mesh run on drjit device arrays
// (src/shape/mesh.cpp:23-62, 317-400); until this round a moved vertex made the HOST recompute every row of its mesh and psdr_hip_scene_update rewrite and send
// them (34 MB for config 5).  Here the rows of the moved meshes are computed by five small kernels from the raw vertices and the composed transform, in the
// host's own dual-number code (csrc/host/hnum.h, __host__ __device__; -ffp-contract=off and correctly rounded division / square root on both sides), so the
// sections hold the bits the host would have written (psdr_hip_scene_check_rows compares them):
//   k_geo_world    world[v]  = xform_pos(to_world, raw[v])                              Mesh::configure, mesh.cpp:330-340
//   k_geo_face     fn[f]     = cross(e1, e2), area2[f] = |fn|                           process_mesh, mesh.cpp:26-33
//   k_geo_vnormal  vn[v]     = normalize(sum fn / sum area2) over the vertex' faces in the reference's scatter order     mesh.cpp:34-44
//   k_geo_rows     traversal / shading / tangent rows of the blob, at the slot of the triangle (leaf order)
//   k_geo_sec      secondary-edge rows (mesh.cpp:355-369, scene.cpp:546-571); their CDF stays with the host (a sequential float prefix sum the parity tests pin)
struct GeoMeshDev { float tw[16], d_tw[16]; int v_off, n_v, f_off, n_f, e_off, n_e, mesh_id, flat, moved, pad[3]; };
using psdr_host::DF; using psdr_host::D3; using psdr_host::DM4;

__device__ inline D3 geo_ld(const float *a, size_t i) { const float *q = a + 6 * i; return D3{DF(q[0], q[1]), DF(q[2], q[3]), DF(q[4], q[5])}; }
__device__ inline void geo_st(float *a, size_t i, const D3 &v) { float *q = a + 6 * i; q[0] = v.x.v; q[1] = v.x.d; q[2] = v.y.v; q[3] = v.y.d; q[4] = v.z.v; q[5] = v.z.d; }

__global__ void k_geo_world(const GeoMeshDev *__restrict__ M, const int *__restrict__ vmesh, const float *__restrict__ raw, const float *__restrict__ d_raw,
                            float *__restrict__ world, int nv) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const GeoMeshDev &m = M[vmesh[v]];
    if (!m.moved) return;
    DM4 tw;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) tw.m[i][j] = DF(m.tw[4 * i + j], m.d_tw[4 * i + j]);
    const D3 p{DF(raw[3 * (size_t) v], d_raw[3 * (size_t) v]), DF(raw[3 * (size_t) v + 1], d_raw[3 * (size_t) v + 1]), DF(raw[3 * (size_t) v + 2], d_raw[3 * (size_t) v + 2])};
    geo_st(world, (size_t) v, psdr_host::xform_pos(tw, p));
}

__global__ void k_geo_face(const GeoMeshDev *__restrict__ M, const int *__restrict__ fmesh, const int *__restrict__ faces, const float *__restrict__ world,
                           float *__restrict__ fnrm, float *__restrict__ farea, int nf) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    if (!M[fmesh[f]].moved) return;
    const D3 p0 = geo_ld(world, (size_t) faces[3 * (size_t) f]), e1 = geo_ld(world, (size_t) faces[3 * (size_t) f + 1]) - p0, e2 = geo_ld(world, (size_t) faces[3 * (size_t) f + 2]) - p0;
    const D3 n = psdr_host::dcross(e1, e2);
    const DF a = psdr_host::dnorm(n);
    geo_st(fnrm, (size_t) f, n);
    farea[2 * (size_t) f] = a.v; farea[2 * (size_t) f + 1] = a.d;
}

__global__ void k_geo_vnormal(const GeoMeshDev *__restrict__ M, const int *__restrict__ vmesh, const int *__restrict__ vf_begin, const int *__restrict__ vf_item,
                              const float *__restrict__ fnrm, const float *__restrict__ farea, float *__restrict__ vn, int nv) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    if (!M[vmesh[v]].moved) return;
    D3 acc; DF w;
    for (int k = vf_begin[v]; k < vf_begin[v + 1]; ++k) {
        const size_t f = (size_t) (vf_item[k] >> 2);
        acc = acc + geo_ld(fnrm, f);
        w = w + DF(farea[2 * f], farea[2 * f + 1]);
    }
    geo_st(vn, (size_t) v, psdr_host::dnormalize(acc / w));
}

__device__ inline float geo_ibits(int v) { return __int_as_float(v); }

__global__ void k_geo_rows(float4 *__restrict__ blob, SceneTables T, const GeoMeshDev *__restrict__ M, const int *__restrict__ fmesh, const int *__restrict__ faces,
                           const float *__restrict__ world, const float *__restrict__ fnrm, const float *__restrict__ farea, const float *__restrict__ vn, int nf, int values, int tangents) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const GeoMeshDev &m = M[fmesh[f]];
    if (!m.moved) return;
    const int slot = reinterpret_cast<const int *>(blob + T.map_off)[f];
    const int i0 = faces[3 * (size_t) f], i1 = faces[3 * (size_t) f + 1], i2 = faces[3 * (size_t) f + 2];
    const D3 p0 = geo_ld(world, (size_t) i0), e1 = geo_ld(world, (size_t) i1) - p0, e2 = geo_ld(world, (size_t) i2) - p0;
    const D3 n0 = geo_ld(vn, (size_t) i0), n1 = geo_ld(vn, (size_t) i1), n2 = geo_ld(vn, (size_t) i2);
    const DF a2 = DF(farea[2 * (size_t) f], farea[2 * (size_t) f + 1]);
    const D3 fn = geo_ld(fnrm, (size_t) f) / a2;
    const DF area = a2 * DF(0.5f);
    if (values) {
        float4 *t = blob + T.trav_off + 3 * (size_t) slot;
        t[0] = make_float4(p0.x.v, p0.y.v, p0.z.v, e1.x.v);
        t[1] = make_float4(e1.y.v, e1.z.v, e2.x.v, e2.y.v);
        t[2] = make_float4(e2.z.v, geo_ibits(f), 0.f, 0.f);
        float4 *w = blob + T.shade_off + 6 * (size_t) slot;          // (words 4 and 5 - the uv of the three corners - do not depend on the vertices)
        w[0] = make_float4(n0.x.v, n0.y.v, n0.z.v, area.v);
        w[1] = make_float4(n1.x.v, n1.y.v, n1.z.v, geo_ibits(m.mesh_id));
        w[2] = make_float4(n2.x.v, n2.y.v, n2.z.v, geo_ibits(m.flat ? 1 : 0));
        w[3] = make_float4(fn.x.v, fn.y.v, fn.z.v, geo_ibits(f));
    }
    if (tangents && T.has_tangent) {
        float4 *w = blob + T.tan_off + 6 * (size_t) slot;
        w[0] = make_float4(p0.x.d, p0.y.d, p0.z.d, e1.x.d);
        w[1] = make_float4(e1.y.d, e1.z.d, e2.x.d, e2.y.d);
        w[2] = make_float4(e2.z.d, n0.x.d, n0.y.d, n0.z.d);
        w[3] = make_float4(n1.x.d, n1.y.d, n1.z.d, n2.x.d);
        w[4] = make_float4(n2.y.d, n2.z.d, fn.x.d, fn.y.d);
        w[5] = make_float4(fn.z.d, area.d, 0.f, 0.f);
    }
}

// edges: [ne][6] = v0 v1 opp (global vertex ids), f0 f1 (global face ids, f1 = -1: boundary), mesh index
__global__ void k_geo_sec(float4 *__restrict__ blob, int sec_off, const GeoMeshDev *__restrict__ M, const int *__restrict__ edges, const float *__restrict__ world,
                          const float *__restrict__ fnrm, const float *__restrict__ farea, int ne) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ne) return;
    const int *e = edges + 6 * (size_t) r;
    if (!M[e[5]].moved) return;
    const D3 a = geo_ld(world, (size_t) e[0]), b = geo_ld(world, (size_t) e[1]), c = geo_ld(world, (size_t) e[2]);
    const float e1[3] = {b.x.v - a.x.v, b.y.v - a.y.v, b.z.v - a.z.v}, de1[3] = {b.x.d - a.x.d, b.y.d - a.y.d, b.z.d - a.z.d};
    const D3 n0 = geo_ld(fnrm, (size_t) e[3]) / DF(farea[2 * (size_t) e[3]], farea[2 * (size_t) e[3] + 1]);
    float n1[3] = {0.f, 0.f, 0.f};
    if (e[4] >= 0) { const D3 q = geo_ld(fnrm, (size_t) e[4]) / DF(farea[2 * (size_t) e[4]], farea[2 * (size_t) e[4] + 1]); n1[0] = q.x.v; n1[1] = q.y.v; n1[2] = q.z.v; }
    float4 *w = blob + sec_off + 6 * (size_t) r;
    w[0] = make_float4(a.x.v, a.y.v, a.z.v, e1[0]);
    w[1] = make_float4(e1[1], e1[2], n0.x.v, n0.y.v);
    w[2] = make_float4(n0.z.v, n1[0], n1[1], n1[2]);
    w[3] = make_float4(c.x.v, c.y.v, c.z.v, geo_ibits(e[4] < 0 ? 1 : 0));
    w[4] = make_float4(a.x.d, a.y.d, a.z.d, de1[0]);
    w[5] = make_float4(de1[1], de1[2], 0.f, 0.f);
}

// uploads what the kernels need (topology when a mesh's version changed, raw vertices and transforms of the moved meshes) and runs them; -> 1 on error.
// `usable` = false (nothing done) when the snapshot's geometry does not describe the scene's triangles and edges one to one.
static int geometry_on_device(psdr_hip_scene *sc, const psdr_scene_snapshot *s, bool values, bool tangents, bool sec_rows, psdr_update_info &info, bool &usable) {
    usable = false;
    const psdr_mesh_geometry *G = s->geometry;
    if (!G) return 0;
    const int nm = s->n_meshes;
    size_t NV = 0, NF = 0, NE = 0;
    for (int i = 0; i < nm; ++i) {
        if (G[i].n_faces != s->meshes[i].n_faces || (size_t) s->meshes[i].face_offset != NF || !G[i].vertices_raw || !G[i].d_vertices_raw || !G[i].faces || !G[i].vf_begin || !G[i].vf_item) return 0;
        if (G[i].n_edges > 0 && !G[i].edges) return 0;
        NV += (size_t) G[i].n_vertices; NF += (size_t) G[i].n_faces; NE += (size_t) G[i].n_edges;
    }
    if (NF != (size_t) s->tris.n_triangles || NE != (size_t) std::max(0, s->sec_edges.n_edges) || NV == 0) return 0;
    // ---- topology: kept while every mesh reports the version (and the counts) the device saw last
    bool same_topo = sc->geo_versions.size() == (size_t) nm && sc->geo_counts.size() == 3 * (size_t) nm;
    for (int i = 0; same_topo && i < nm; ++i)
        same_topo = sc->geo_versions[(size_t) i] == G[i].topology_version && sc->geo_counts[3 * (size_t) i] == G[i].n_vertices && sc->geo_counts[3 * (size_t) i + 1] == G[i].n_faces &&
                    sc->geo_counts[3 * (size_t) i + 2] == G[i].n_edges;
    std::vector<GeoMeshDev> mt((size_t) nm);
    {
        size_t vo = 0, fo = 0, eo = 0;
        for (int i = 0; i < nm; ++i) {
            GeoMeshDev &m = mt[(size_t) i];
            std::memcpy(m.tw, G[i].to_world, 64); std::memcpy(m.d_tw, G[i].d_to_world, 64);
            m.v_off = (int) vo; m.n_v = G[i].n_vertices; m.f_off = (int) fo; m.n_f = G[i].n_faces; m.e_off = (int) eo; m.n_e = G[i].n_edges;
            m.mesh_id = G[i].mesh_id; m.flat = G[i].use_face_normals; m.moved = (G[i].moved || !same_topo) ? 1 : 0; m.pad[0] = m.pad[1] = m.pad[2] = 0;
            vo += (size_t) G[i].n_vertices; fo += (size_t) G[i].n_faces; eo += (size_t) G[i].n_edges;
        }
    }
    auto up = [&](const char *key, const void *src, size_t bytes) -> int {
        DevBuf &b = sc->buf(key);
        if (b.ensure(bytes)) return 1;
        if (src) { HIPCHK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice)); info.bytes_uploaded += (int64_t) bytes; }         // (synchronous: the sources are locals of this function)
        return 0;
    };
    if (!same_topo) {
        std::vector<int> faces(3 * NF), vmesh(NV), fmesh(NF), vfb(NV + 1), vfi(3 * NF), edges(6 * std::max<size_t>(1, NE));
        size_t k = 0;
        vfb[0] = 0;
        for (int i = 0; i < nm; ++i) {
            const GeoMeshDev &m = mt[(size_t) i];
            for (int v = 0; v < m.n_v; ++v) vmesh[(size_t) m.v_off + (size_t) v] = i;
            for (int f = 0; f < m.n_f; ++f) {
                fmesh[(size_t) m.f_off + (size_t) f] = i;
                for (int c = 0; c < 3; ++c) {
                    const int vi = G[i].faces[3 * (size_t) f + c];
                    if (vi < 0 || vi >= m.n_v) return fail("psdr_mesh_geometry: face index outside the mesh's vertices");
                    faces[3 * ((size_t) m.f_off + (size_t) f) + c] = m.v_off + vi;
                }
            }
            // (vf_begin of a mesh counts from 0; items are (local face << 2 | corner): both move to the global numbering)
            for (int v = 0; v < m.n_v; ++v) {
                for (int q = G[i].vf_begin[v]; q < G[i].vf_begin[v + 1]; ++q) { const int it = G[i].vf_item[q]; vfi[k++] = (((it >> 2) + m.f_off) << 2) | (it & 3); }
                vfb[(size_t) m.v_off + (size_t) v + 1] = (int) k;
            }
            for (int e = 0; e < m.n_e; ++e) {
                const int *q = G[i].edges + 5 * (size_t) e;
                int *d = &edges[6 * ((size_t) m.e_off + (size_t) e)];
                d[0] = m.v_off + q[0]; d[1] = m.v_off + q[1]; d[2] = m.v_off + q[4]; d[3] = m.f_off + q[2]; d[4] = q[3] >= 0 ? m.f_off + q[3] : -1; d[5] = i;
            }
        }
        if (k != 3 * NF) return fail("psdr_mesh_geometry: vf lists do not cover the faces");
        if (up("geo.faces", faces.data(), faces.size() * 4) || up("geo.vmesh", vmesh.data(), vmesh.size() * 4) || up("geo.fmesh", fmesh.data(), fmesh.size() * 4) ||
            up("geo.vf_begin", vfb.data(), vfb.size() * 4) || up("geo.vf_item", vfi.data(), vfi.size() * 4) || up("geo.edges", edges.data(), edges.size() * 4)) return 1;
        if (up("geo.raw", nullptr, 12 * NV) || up("geo.d_raw", nullptr, 12 * NV) || up("geo.world", nullptr, 24 * NV) || up("geo.vn", nullptr, 24 * NV) ||
            up("geo.fnrm", nullptr, 24 * NF) || up("geo.farea", nullptr, 8 * NF)) return 1;
        HIPCHK(hipStreamSynchronize(nullptr));                     // (the staging vectors go out of scope)
        sc->geo_versions.resize((size_t) nm); sc->geo_counts.resize(3 * (size_t) nm);
        for (int i = 0; i < nm; ++i) {
            sc->geo_versions[(size_t) i] = G[i].topology_version;
            sc->geo_counts[3 * (size_t) i] = G[i].n_vertices; sc->geo_counts[3 * (size_t) i + 1] = G[i].n_faces; sc->geo_counts[3 * (size_t) i + 2] = G[i].n_edges;
        }
    }
    // ---- this update's inputs: the mesh table, raw vertices and their tangents of the moved meshes
    if (up("geo.meshes", mt.data(), mt.size() * sizeof(GeoMeshDev))) return 1;
    float *raw = (float *) sc->buf("geo.raw").p, *d_raw = (float *) sc->buf("geo.d_raw").p;
    bool any = false;
    for (int i = 0; i < nm; ++i) {
        const GeoMeshDev &m = mt[(size_t) i];
        if (!m.moved || m.n_v == 0) continue;
        any = true;
        HIPCHK(hipMemcpyAsync(raw + 3 * (size_t) m.v_off, G[i].vertices_raw, 12 * (size_t) m.n_v, hipMemcpyHostToDevice, nullptr));
        HIPCHK(hipMemcpyAsync(d_raw + 3 * (size_t) m.v_off, G[i].d_vertices_raw, 12 * (size_t) m.n_v, hipMemcpyHostToDevice, nullptr));
        info.bytes_uploaded += (int64_t) (24 * (size_t) m.n_v);
    }
    usable = true;
    if (!any) return 0;
    const GeoMeshDev *Md = sc->buf("geo.meshes").as<GeoMeshDev>();
    const int *vmesh = sc->buf("geo.vmesh").as<int>(), *fmesh = sc->buf("geo.fmesh").as<int>(), *faces = sc->buf("geo.faces").as<int>();
    float *world = (float *) sc->buf("geo.world").p, *vn = (float *) sc->buf("geo.vn").p, *fnrm = (float *) sc->buf("geo.fnrm").p, *farea = (float *) sc->buf("geo.farea").p;
    const unsigned bv = (unsigned) ((NV + 255) / 256), bf = (unsigned) ((NF + 255) / 256), be = (unsigned) ((NE + 255) / 256);
    hipLaunchKernelGGL(k_geo_world, dim3(bv), dim3(256), 0, nullptr, Md, vmesh, raw, d_raw, world, (int) NV);
    hipLaunchKernelGGL(k_geo_face, dim3(bf), dim3(256), 0, nullptr, Md, fmesh, faces, world, fnrm, farea, (int) NF);
    hipLaunchKernelGGL(k_geo_vnormal, dim3(bv), dim3(256), 0, nullptr, Md, vmesh, sc->buf("geo.vf_begin").as<int>(), sc->buf("geo.vf_item").as<int>(), fnrm, farea, vn, (int) NV);
    if (values || tangents)
        hipLaunchKernelGGL(k_geo_rows, dim3(bf), dim3(256), 0, nullptr, (float4 *) sc->blob.p, sc->T, Md, fmesh, faces, world, fnrm, farea, vn, (int) NF, values ? 1 : 0, tangents ? 1 : 0);
    if (sec_rows && NE > 0)
        hipLaunchKernelGGL(k_geo_sec, dim3(be), dim3(256), 0, nullptr, (float4 *) sc->blob.p, sc->E.off, Md, sc->buf("geo.edges").as<int>(), world, fnrm, farea, (int) NE);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));                         // (the mesh table is a local; the kernels take ~0.1 ms)
    return 0;
}

struct WordRange { size_t b, e; };

// Everything between a snapshot and a renderable device scene.  fresh: the handle is new.  same: PSDR_SAME_* bits the caller vouches for
// (relative to the snapshot of the previous create / update of this handle); force_build: build the tree even if the triangle count fits.
static int scene_sync(psdr_hip_scene *sc, const psdr_scene_snapshot *s, unsigned same, bool fresh, bool force_build, psdr_update_info *info_out) {
    const auto t_start = std::chrono::steady_clock::now();
    psdr_update_info info{};
    const psdr_triangles &tr = s->tris;
    const int n = tr.n_triangles;
    if (n <= 0) return fail("Missing meshes!");
    if (s->n_sensors <= 0) return fail("Missing sensor!");
    for (int i = 0; i < s->n_bsdfs; ++i) {
        const psdr_bsdf_rec &b = s->bsdfs[i];
        if (b.type < 0 || b.type > 5) return fail("Unknown BSDF type!");
        if (b.type == 5 && (b.nested_bsdf < 0 || b.nested_bsdf >= s->n_bsdfs || s->bsdfs[b.nested_bsdf].type == 5)) return fail("NormalMap: invalid nested BSDF");
    }
    const bool build = fresh || force_build || n != sc->tree_tris;
    // rows_valid = 0 (updates only; psdr_scene_snapshot): the caller relies on the device computing the moved meshes' rows.  Whatever needs the rows themselves - a tree to
    // build, a section that moves, a live-pixel mask to rebuild - answers PSDR_HIP_NEED_ROWS before anything has been changed
    const bool rows_valid = fresh || s->rows_valid != 0;
    if (!rows_valid && (build || s->geometry == nullptr || n <= kBruteForceMax)) return PSDR_HIP_NEED_ROWS;
    if (build) same = 0;
    if (!fresh) {
        // the previous calls on this scene read what is about to be overwritten: wait for the last of them (configure() is a synchronisation point in the reference as well)
        std::lock_guard<std::mutex> lk(sc->mu);
        if (sc->ev && sc->have_last) HIPCHK(hipEventSynchronize(sc->ev));
    }
    const bool same_tris = (same & PSDR_SAME_TRIANGLES) != 0, same_tan = (same & PSDR_SAME_TRI_TANGENTS) != 0, same_sec = (same & PSDR_SAME_SEC_EDGES) != 0,
               same_prim = (same & PSDR_SAME_PRIM_EDGES) != 0, same_env = (same & PSDR_SAME_ENV_TEXELS) != 0, same_env_tan = (same & PSDR_SAME_ENV_TANGENT) != 0,
               same_bitmaps = (same & PSDR_SAME_BITMAPS) != 0;

    std::vector<float> new_nodes;
    auto t_tree = std::chrono::steady_clock::now();
    if (build) {
        if (tree_build(sc, tr, new_nodes)) return 1;
        info.tree = 2;
    }
    info.ms_tree = ms_since(t_tree);
    const auto t_fill = std::chrono::steady_clock::now();
    const bool uses_bvh = n > kBruteForceMax;
    const bool has_tan = tr.d_p0 != nullptr;
    const std::vector<int32_t> &order = sc->order;

    // ---- filter primitives of the brute-force tracer (their number depends on the geometry: coplanar neighbours become quads)
    std::vector<FilterPrim> filt;
    // (a snapshot that gains or loses its tangent arrays shifts every section behind them: treated like moved triangles)
    const bool geo = !same_tris || (has_tan ? 1 : 0) != sc->T.has_tangent;
    if (geo) build_filter_prims(tr.p0, tr.e1, tr.e2, order.data(), n, filt);

    // ---- layout (float4-word offsets); a section keeps its content only when the caller vouches for it AND it stays where it was
    const SceneTables Told = sc->T;
    const SecEdgeTables Eold = sc->E;
    SceneTables &T = sc->T;
    SecEdgeTables &E = sc->E;
    const psdr_sec_edges &se = s->sec_edges;
    size_t w = 0;
    T.nodes_off = (int) w; w += (size_t) (kNodeFloats / 4) * (size_t) T.n_nodes;      // nodes of the 4-wide tree (bvh.h: 64 bytes each)
    T.trav_off = (int) w;  w += 3 * (size_t) n;
    T.shade_off = (int) w; w += 6 * (size_t) n;
    T.tan_off = (int) w;   w += has_tan ? 6 * (size_t) n : 0;
    T.map_off = (int) w;   w += words_for_floats((size_t) n);
    T.filt_off = (int) w;  T.n_filt = geo ? (int) filt.size() : Told.n_filt; w += 6 * (size_t) T.n_filt;
    const size_t small_begin = w;
    T.mesh_off = (int) w;  w += 2 * (size_t) s->n_meshes;
    T.bsdf_off = (int) w;  w += 2 * (size_t) std::max(1, s->n_bsdfs);
    T.emit_off = (int) w;  w += 2 * (size_t) std::max(1, s->n_emitters);
    T.ecdf_off = (int) w;  w += words_for_floats(2 * (size_t) std::max(1, s->n_emitters));
    T.fcdf_off = (int) w;  w += words_for_floats(2 * (size_t) std::max(1, s->n_face_distrb));
    const size_t small_end = w;
    E.n = se.n_edges; E.sum = se.sum;
    E.off = (int) w;       w += 6 * (size_t) std::max(0, se.n_edges);
    E.cdf_off = (int) w;   w += words_for_floats(2 * (size_t) std::max(1, se.n_edges));
    const size_t sec_end = w;
    std::vector<std::pair<int, int>> pe_offs;
    for (int i = 0; i < s->n_sensors; ++i) {
        const int ne = std::max(0, s->sensors[i].n_edges);
        const int o1 = (int) w; w += 3 * (size_t) ne;
        const int o2 = (int) w; w += words_for_floats(2 * (size_t) std::max(1, ne));
        pe_offs.emplace_back(o1, o2);
    }
    if (w > 0x7fffffffull / 4) return fail("scene too large for 32-bit blob offsets");
    T.blob_words = (int) w;
    T.n_tris = n; T.n_meshes = s->n_meshes; T.n_bsdfs = s->n_bsdfs; T.n_emitters = s->n_emitters;
    T.n_fcdf = s->n_face_distrb; T.has_tangent = has_tan ? 1 : 0;
    T.emitter_sum = s->emitter_sum;
    T.width = s->width; T.height = s->height; T.spp = s->spp; T.sppe = s->sppe; T.sppse = s->sppse;
    T.env_emitter = -1;
    for (int i = 0; i < s->n_emitters; ++i) if (s->emitters[i].type == 1) T.env_emitter = i;
    // traversal stack: the first kStackLds entries of a lane in LDS, deeper ones in a per-lane global array (trav4.h);
    // scenes that are traced by brute force (<= kBruteForceMax triangles) need neither
    // kStackLdsMax + kTravRows = 40 KB per workgroup: four workgroups per CU (round 6: 12 stack rows + 2 rows of tree top, until then 8 + 6 - trav4.h).  Round 5 measured both sides of that choice on config 5 (LABNOTES): FEWER workgroups cost a lot
    // (3 per CU: +17 %, 2: +55 %), a FIFTH brings nothing (a 30 KB layout - no hit rows, 4 stack rows - lost exactly what its shorter stack costs at equal
    // occupancy), and stack rows going to the global tail cost 1.2 % (6 rows), 6.8 % (4), 9.5 % (2).
    // PSDR_STACK_LDS = 2 ... kStackLdsMax (test knob, read when a scene is created or rebuilt): fewer rows, so that small test scenes reach the global tail too
    // (tests/test_gpu_configs.py: the forked terms of a renderD against the serial call)
    int kStackLds = kStackLdsMax;
    if (const char *e = std::getenv("PSDR_STACK_LDS")) kStackLds = std::max(2, std::min(kStackLdsMax, std::atoi(e)));
    T.stack_lds = uses_bvh ? std::min(kStackLds, sc->tree_max_stack) : 0;
    T.stack_depth = uses_bvh ? T.stack_lds + kTravRows : kColdRows;   // BVH: + parked rays, best hits and the pair ring of the traversal (trav4.h); brute force: cold path state (paths.h)

    if (!rows_valid) {
        const bool layout_kept = sc->blob.p && sc->blob.bytes >= 16 * w && T.trav_off == Told.trav_off && T.shade_off == Told.shade_off && T.tan_off == Told.tan_off && T.map_off == Told.map_off &&
                                 (has_tan ? 1 : 0) == Told.has_tangent && E.off == Eold.off && E.cdf_off == Eold.cdf_off && E.n == Eold.n && T.env_emitter >= 0 && Told.env_emitter >= 0 &&
                                 std::getenv("PSDR_HOST_GEOMETRY") == nullptr;
        if (!layout_kept) { sc->T = Told; sc->E = Eold; return PSDR_HIP_NEED_ROWS; }
    }
    // ---- the blob: device allocation with head room, pinned host copy
    bool blob_moved = false;
    if (!sc->blob.p || sc->blob.bytes < 16 * w) {
        const size_t cap = 16 * (w + w / 4 + 64);
        void *fresh_p = nullptr;
        HIPCHK(hipMalloc(&fresh_p, cap));
        if (sc->blob.p) {
            // the nodes a refit has written exist on the device only: they move with the allocation
            if (!build && uses_bvh) HIPCHK(hipMemcpy((char *) fresh_p + 16 * (size_t) T.nodes_off, (const char *) sc->blob.p + 16 * (size_t) Told.nodes_off, 16 * (size_t) (kNodeFloats / 4) * (size_t) T.n_nodes, hipMemcpyDeviceToDevice));
            HIPCHK(hipFree(sc->blob.p));
            info.reallocated++;
        }
        sc->blob.p = fresh_p; sc->blob.bytes = cap;
        blob_moved = true;
    }
    if (sc->hblob.ensure(sc->blob.bytes)) return 1;
    float *H = (float *) sc->hblob.p;
    std::vector<WordRange> dirty;
    auto mark = [&](size_t b, size_t e) { if (e > b) dirty.push_back({b, e}); };

    if (build) {
        std::memcpy(H + 4 * (size_t) T.nodes_off, new_nodes.data(), sizeof(float) * new_nodes.size());
        mark((size_t) T.nodes_off, (size_t) T.nodes_off + new_nodes.size() / 4);
    }
    if (build || blob_moved || T.map_off != Told.map_off) {
        for (int i = 0; i < n; ++i) H[4 * (size_t) T.map_off + (size_t) i] = ibits(sc->orig2slot[(size_t) i]);
        mark((size_t) T.map_off, (size_t) T.map_off + words_for_floats((size_t) n));
    }
    // ---- moved meshes: their rows computed on the device (geometry_on_device above) when nothing else about the layout changed; the host then writes none of them
    bool dev_geo = false;
    {
        const bool layout_same = !fresh && !build && !blob_moved && uses_bvh && T.trav_off == Told.trav_off && T.shade_off == Told.shade_off && T.tan_off == Told.tan_off &&
                                 T.map_off == Told.map_off && (has_tan ? 1 : 0) == Told.has_tangent && E.off == Eold.off && E.cdf_off == Eold.cdf_off && E.n == Eold.n;
        const bool forced_host = std::getenv("PSDR_HOST_GEOMETRY") != nullptr;         // test knob, read per call: the host writes the rows (what psdr_hip_scene_check_rows compares with)
        if (layout_same && !forced_host && s->geometry != nullptr && (geo || !same_tan || !same_sec)) {
            if (geometry_on_device(sc, s, geo, has_tan && !same_tan, !same_sec, info, dev_geo)) return 1;
            if (!dev_geo && !rows_valid) { sc->T = Told; sc->E = Eold; return PSDR_HIP_NEED_ROWS; }
        }
    }
    // (the pinned host copy of a section the device wrote is behind the device's; it is only ever sent after the host has rewritten the whole section from the snapshot -
    //  a change of the section itself, or a moved allocation, both of which write it first)
    const bool write_geo = !dev_geo && (geo || blob_moved || T.trav_off != Told.trav_off || T.shade_off != Told.shade_off);
    if (write_geo) {
        parallel_for((size_t) n, 8192, [&](size_t b0, size_t e0) {
            for (size_t slot = b0; slot < e0; ++slot) {
                const size_t o = (size_t) order[slot];
                const float *p0 = tr.p0 + 3 * o, *e1 = tr.e1 + 3 * o, *e2 = tr.e2 + 3 * o;
                put4(H, T.trav_off + 3 * slot, p0[0], p0[1], p0[2], e1[0]);
                put4(H, T.trav_off + 3 * slot + 1, e1[1], e1[2], e2[0], e2[1]);
                put4(H, T.trav_off + 3 * slot + 2, e2[2], ibits((int32_t) o), 0.f, 0.f);
                const float *n0 = tr.n0 + 3 * o, *n1 = tr.n1 + 3 * o, *n2 = tr.n2 + 3 * o, *fn = tr.face_normal + 3 * o;
                const size_t sw = T.shade_off + 6 * slot;
                put4(H, sw, n0[0], n0[1], n0[2], tr.face_area[o]);
                put4(H, sw + 1, n1[0], n1[1], n1[2], ibits(tr.mesh_id[o]));
                put4(H, sw + 2, n2[0], n2[1], n2[2], ibits(tr.use_face_normal && tr.use_face_normal[o] ? 1 : 0));
                put4(H, sw + 3, fn[0], fn[1], fn[2], ibits((int32_t) o));
                if (tr.uv) {
                    const float *uv = tr.uv + 6 * o;
                    put4(H, sw + 4, uv[0], uv[1], uv[2], uv[3]);
                    put4(H, sw + 5, uv[4], uv[5], 0.f, 0.f);
                } else { put4(H, sw + 4, 0.f, 0.f, 0.f, 0.f); put4(H, sw + 5, 0.f, 0.f, 0.f, 0.f); }
            }
        });
        mark((size_t) T.trav_off, (size_t) T.trav_off + 9 * (size_t) n);
    }
    if (has_tan && !dev_geo && (!same_tan || blob_moved || T.tan_off != Told.tan_off || !Told.has_tangent)) {
        parallel_for((size_t) n, 8192, [&](size_t b0, size_t e0) {
            for (size_t slot = b0; slot < e0; ++slot) {
                const size_t o = (size_t) order[slot];
                const float *a = tr.d_p0 + 3 * o, *b = tr.d_e1 + 3 * o, *c = tr.d_e2 + 3 * o, *d0 = tr.d_n0 + 3 * o, *d1 = tr.d_n1 + 3 * o,
                            *d2 = tr.d_n2 + 3 * o, *df = tr.d_face_normal + 3 * o;
                const size_t tw = T.tan_off + 6 * slot;
                put4(H, tw, a[0], a[1], a[2], b[0]);
                put4(H, tw + 1, b[1], b[2], c[0], c[1]);
                put4(H, tw + 2, c[2], d0[0], d0[1], d0[2]);
                put4(H, tw + 3, d1[0], d1[1], d1[2], d2[0]);
                put4(H, tw + 4, d2[1], d2[2], df[0], df[1]);
                put4(H, tw + 5, df[2], tr.d_face_area[o], 0.f, 0.f);
            }
        });
        mark((size_t) T.tan_off, (size_t) T.tan_off + 6 * (size_t) n);
    }
    if (geo && uses_bvh) {
        // (the bounding sphere and the filter primitives belong to the brute-force tracer, scene_dev.h::trace2: a BVH scene has neither - and the serial pass over its
        //  triangles' vertices was 0.5 ms of every moved-vertex update of config 5)
        T.center[0] = T.center[1] = T.center[2] = 0.f; T.radius = 0.f; T.filt_kmax = 0.f; T.filt_hasb[0] = T.filt_hasb[1] = 0u;
    } else if (geo) {
        // bounding sphere of the scene (for the absolute slack of the quad filter)
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int i = 0; i < n; ++i)
            for (int v = 0; v < 3; ++v)
                for (int k = 0; k < 3; ++k) {
                    const double x = (double) tr.p0[3 * (size_t) i + k] + (v == 1 ? (double) tr.e1[3 * (size_t) i + k] : v == 2 ? (double) tr.e2[3 * (size_t) i + k] : 0.0);
                    lo[k] = std::min(lo[k], x); hi[k] = std::max(hi[k], x);
                }
        double r2 = 0.0;
        for (int k = 0; k < 3; ++k) { T.center[k] = (float) (0.5 * (lo[k] + hi[k])); r2 += 0.25 * (hi[k] - lo[k]) * (hi[k] - lo[k]); }
        T.radius = (float) (std::sqrt(r2) * 1.0001);
        T.filt_kmax = 0.f;
        T.filt_hasb[0] = T.filt_hasb[1] = 0u;
        for (size_t i = 0; i < filt.size(); ++i) {
            // dot-product form of the filter (scene_dev.h::trace2): with oc = o - centre, m = oc x d and p = p0 - centre
            //   u-numerator = m.e2 + d.(p x e2)   v-numerator = d.(e1 x p) - m.e1   -det = d.(e1 x e2)   t-numerator = oc.(e1 x e2) - p.(e1 x e2)
            const FilterPrim &f = filt[i];
            const size_t fw = T.filt_off + 6 * i;
            {
                const size_t base = i & ~(size_t) 31, cnt = std::min<size_t>(32, filt.size() - base);
                if (f.slot_b >= 0) T.filt_hasb[i >> 5] |= 1u << (cnt - 1 - (i - base));
            }
            const double p[3] = {(double) f.p0[0] - (double) T.center[0], (double) f.p0[1] - (double) T.center[1], (double) f.p0[2] - (double) T.center[2]};
            const double e1[3] = {f.e1[0], f.e1[1], f.e1[2]}, e2[3] = {f.e2[0], f.e2[1], f.e2[2]};
            auto crs = [](const double *a, const double *b, double *c) { c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0]; };
            double A[3], B[3], N[3];
            crs(p, e2, A); crs(e1, p, B); crs(e1, e2, N);
            const double npn = -(p[0] * N[0] + p[1] * N[1] + p[2] * N[2]);
            const double K = std::max({std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]), (double) f.k16 * 32768.0});
            T.filt_kmax = std::max(T.filt_kmax, (float) (K * 1.0001));
            put4(H, fw, f.e2[0], f.e2[1], f.e2[2], (float) A[0]);
            put4(H, fw + 1, (float) A[1], (float) A[2], f.e1[0], f.e1[1]);
            put4(H, fw + 2, f.e1[2], (float) B[0], (float) B[1], (float) B[2]);
            put4(H, fw + 3, (float) N[0], (float) N[1], (float) N[2], (float) npn);
            put4(H, fw + 4, f.umax, f.vmax, f.smax, f.da);
            put4(H, fw + 5, f.db, f.da + f.db, (float) (K * (1.0001 / 32768.0)), ibits(f.slot_a | ((f.slot_b < 0 ? 0xff : f.slot_b) << 8)));
        }
        mark((size_t) T.filt_off, (size_t) T.filt_off + 6 * filt.size());
    } else if (blob_moved) mark((size_t) T.filt_off, (size_t) T.filt_off + 6 * (size_t) T.n_filt);       // (same triangles: same offset, the host copy is current)

    // ---- the small tables: always written
    std::memset(H + 4 * small_begin, 0, 16 * (small_end - small_begin));
    for (int i = 0; i < s->n_meshes; ++i) {
        const psdr_mesh_rec &m = s->meshes[i];
        put4(H, T.mesh_off + 2 * (size_t) i, ibits(m.bsdf_id), ibits(m.emitter_id), ibits(m.face_offset), ibits(m.n_faces));
        put4(H, T.mesh_off + 2 * (size_t) i + 1, m.inv_total_area, ibits(m.distrb_offset), m.distrb_sum, 0.f);
    }
    sc->simple_mats = true; sc->has_nmap = false;
    for (int i = 0; i < s->n_bsdfs; ++i) {
        const psdr_bsdf_rec &b = s->bsdfs[i];
        // (a NormalMap, type 5, is its nested BSDF seen through the map: the nested record is an entry of its own and decides)
        if (b.type == 5) sc->has_nmap = true;
        put4(H, T.bsdf_off + 2 * (size_t) i, b.reflectance[0], b.reflectance[1], b.reflectance[2], ibits((b.two_sided ? 1 : 0) | (b.tex_data ? 2 : 0) | (b.type == 1 ? 4 : 0) | (b.type == 2 ? 8 : 0) | (b.type == 3 ? 16 : 0) | (b.spec_tex_data ? 32 : 0) | (b.rough_tex_data ? 64 : 0) | (b.type == 4 ? 128 : 0) | (b.type == 5 ? 256 : 0)));
        put4(H, T.bsdf_off + 2 * (size_t) i + 1, b.d_reflectance[0], b.d_reflectance[1], b.d_reflectance[2], ibits(b.type == 5 ? b.nested_bsdf : -1));
    }
    for (int i = 0; i < s->n_emitters; ++i) {
        const psdr_emitter_rec &e = s->emitters[i];
        put4(H, T.emit_off + 2 * (size_t) i, e.radiance[0], e.radiance[1], e.radiance[2], e.sampling_weight);
        put4(H, T.emit_off + 2 * (size_t) i + 1, e.d_radiance[0], e.d_radiance[1], e.d_radiance[2], ibits(e.mesh_id));
        H[4 * (size_t) T.ecdf_off + (size_t) i] = s->emitter_pmf ? s->emitter_pmf[i] : 1.f;
        H[4 * (size_t) T.ecdf_off + (size_t) s->n_emitters + (size_t) i] = s->emitter_cmf ? s->emitter_cmf[i] : 1.f;
    }
    for (int i = 0; i < s->n_face_distrb; ++i) {
        H[4 * (size_t) T.fcdf_off + (size_t) i] = s->face_pmf[i];
        H[4 * (size_t) T.fcdf_off + (size_t) s->n_face_distrb + (size_t) i] = s->face_cmf[i];
    }
    mark(small_begin, small_end);

    // ---- secondary edges
    const bool write_sec = !same_sec || blob_moved || E.off != Eold.off || E.cdf_off != Eold.cdf_off || E.n != Eold.n;
    if (write_sec && se.n_edges > 0 && dev_geo) {
        // (the rows are the device's; the distribution over the edges - a sequential float prefix sum - comes from the host)
        std::memcpy(H + 4 * (size_t) E.cdf_off, se.pmf, sizeof(float) * (size_t) se.n_edges);
        std::memcpy(H + 4 * (size_t) E.cdf_off + (size_t) se.n_edges, se.cmf, sizeof(float) * (size_t) se.n_edges);
        mark((size_t) E.cdf_off, sec_end);
    } else if (write_sec && se.n_edges > 0) {
        parallel_for((size_t) se.n_edges, 8192, [&](size_t b0, size_t e0) {
            const float z3[3] = {0.f, 0.f, 0.f};
            for (size_t i = b0; i < e0; ++i) {
                const float *p0 = se.p0 + 3 * i, *e1 = se.e1 + 3 * i, *n0 = se.n0 + 3 * i, *n1 = se.n1 + 3 * i, *p2 = se.p2 + 3 * i;
                const float *dp0 = se.d_p0 ? se.d_p0 + 3 * i : z3, *de1 = se.d_e1 ? se.d_e1 + 3 * i : z3;
                const size_t ew = E.off + 6 * i;
                put4(H, ew, p0[0], p0[1], p0[2], e1[0]);
                put4(H, ew + 1, e1[1], e1[2], n0[0], n0[1]);
                put4(H, ew + 2, n0[2], n1[0], n1[1], n1[2]);
                put4(H, ew + 3, p2[0], p2[1], p2[2], ibits(se.is_boundary[i] ? 1 : 0));
                put4(H, ew + 4, dp0[0], dp0[1], dp0[2], de1[0]);
                put4(H, ew + 5, de1[1], de1[2], 0.f, 0.f);
                H[4 * (size_t) E.cdf_off + i] = se.pmf[i];
                H[4 * (size_t) E.cdf_off + (size_t) se.n_edges + i] = se.cmf[i];
            }
        });
        mark((size_t) E.off, sec_end);
    }
    if (write_sec) {
        E.guide = nullptr; E.guide_n = 0;
        if (se.n_edges > 0 && se.cmf) {              // every sample of the secondary-edge term starts with this search (17 dependent loads for config 5's 122 885 edges)
            std::vector<int> guide;
            build_cdf_guide(se.cmf, se.n_edges, se.sum, guide, 4);
            if (!guide.empty()) {
                if (sync_named(sc, "sec.guide", guide.data(), guide.size() * sizeof(int), false, E.guide, info)) return 1;
                E.guide_n = (int) guide.size() - 1;
            }
        }
    } else { E.guide = Eold.guide; E.guide_n = Eold.guide_n; }

    // ---- sensors: matrices, primary edges, live-pixel masks
    const std::vector<SensorDev> sensors_old = sc->sensors;
    const std::vector<float> w2s_old = sc->sensor_w2s;
    sc->sensors.assign((size_t) s->n_sensors, SensorDev{});
    sc->sensor_w2s.assign(16 * (size_t) s->n_sensors, 0.f);
    sc->live_host.resize((size_t) s->n_sensors);
    for (int k = 0; k < s->n_sensors; ++k) {
        const psdr_sensor_rec &r = s->sensors[k];
        SensorDev &d = sc->sensors[(size_t) k];
        const SensorDev *od = (size_t) k < sensors_old.size() ? &sensors_old[(size_t) k] : nullptr;
        std::memcpy(d.sample_to_camera.m, r.sample_to_camera, 64); std::memcpy(d.to_world.m, r.to_world, 64);
        std::memcpy(d.d_to_world.m, r.d_to_world, 64); std::memcpy(d.world_to_sample.m, r.world_to_sample, 64);
        std::memcpy(d.d_world_to_sample.m, r.d_world_to_sample, 64);
        for (int q = 0; q < 3; ++q) { d.cam_pos[q] = r.cam_pos[q]; d.cam_dir[q] = r.cam_dir[q]; }
        d.inv_area = r.inv_area; d.n_edges = r.n_edges; d.edge_sum = r.edge_sum; d.ortho = r.orthographic;
        d.pe_off = pe_offs[(size_t) k].first; d.pecdf_off = pe_offs[(size_t) k].second;
        const bool write_pe = !same_prim || blob_moved || !od || od->pe_off != d.pe_off || od->pecdf_off != d.pecdf_off || od->n_edges != d.n_edges;
        if (write_pe) {
            parallel_for((size_t) std::max(0, r.n_edges), 8192, [&](size_t ib, size_t ie) {
              for (size_t i = ib; i < ie; ++i) {
                const size_t pw = (size_t) d.pe_off + 3 * (size_t) i;
                put4(H, pw, r.edge_p0[2 * i], r.edge_p0[2 * i + 1], r.edge_p1[2 * i], r.edge_p1[2 * i + 1]);
                put4(H, pw + 1, r.d_edge_p0 ? r.d_edge_p0[2 * i] : 0.f, r.d_edge_p0 ? r.d_edge_p0[2 * i + 1] : 0.f,
                     r.d_edge_p1 ? r.d_edge_p1[2 * i] : 0.f, r.d_edge_p1 ? r.d_edge_p1[2 * i + 1] : 0.f);
                put4(H, pw + 2, r.edge_normal[2 * i], r.edge_normal[2 * i + 1], r.edge_length[i], 0.f);
                H[4 * (size_t) d.pecdf_off + (size_t) i] = r.edge_pmf[i];
                H[4 * (size_t) d.pecdf_off + (size_t) r.n_edges + (size_t) i] = r.edge_cmf[i];
              }
            });
            mark((size_t) d.pe_off, (size_t) d.pecdf_off + words_for_floats(2 * (size_t) std::max(1, r.n_edges)));
            d.pe_guide = nullptr; d.pe_guide_n = 0;
            if (r.n_edges > 0 && r.edge_cmf) {       // a sample of the primary-edge term starts with this search (15 dependent loads for config 5's 26 592 edges)
                std::vector<int> guide;
                build_cdf_guide(r.edge_cmf, r.n_edges, r.edge_sum, guide, 4);
                if (!guide.empty()) {
                    if (sync_named(sc, "sensor." + std::to_string(k) + ".guide", guide.data(), guide.size() * sizeof(int), false, d.pe_guide, info)) return 1;
                    d.pe_guide_n = (int) guide.size() - 1;
                }
            }
        } else { d.pe_guide = od->pe_guide; d.pe_guide_n = od->pe_guide_n; }
        // live-pixel mask: a function of the triangles, this sensor's world_to_sample and the frame size
        std::memcpy(&sc->sensor_w2s[16 * (size_t) k], r.world_to_sample, 64);
        const bool same_view = od && !geo && 16 * (size_t) k + 16 <= w2s_old.size() && std::memcmp(&w2s_old[16 * (size_t) k], r.world_to_sample, 64) == 0 &&
                               Told.width == T.width && Told.height == T.height && (Told.env_emitter >= 0) == (T.env_emitter >= 0) &&
                               ((long long) Told.width * Told.height * std::max(1, Told.spp) < (1ll << 31)) == ((long long) T.width * T.height * std::max(1, T.spp) < (1ll << 31));
        if (same_view) d.live = od->live;
        else {
            d.live = nullptr;
            std::vector<unsigned> live;
            static const bool no_live = std::getenv("PSDR_NO_LIVE_MASK") != nullptr;       // documented switch (include/psdr_hip.h): render the provably-zero samples as well
            bool use = !no_live && T.env_emitter < 0 && (long long) s->width * s->height * std::max(1, s->spp) < (1ll << 31) && build_live_mask(s->tris, r.world_to_sample, s->width, s->height, live);
            if (use) {       // worth a window of bit tests per regeneration only when a good part of the frame is dead (the sphere box, all of it live: +1.7 % with the mask)
                long long n_set = 0;
                for (unsigned x : live) n_set += __builtin_popcount(x);
                use = n_set * 4 <= (long long) s->width * s->height * 3;
            }
            if (use) { if (sync_named(sc, "sensor." + std::to_string(k) + ".live", live.data(), live.size() * sizeof(unsigned), false, d.live, info)) return 1; }
            else live.clear();
            sc->live_host[(size_t) k].swap(live);
        }
    }

    // ---- environment map (global memory, outside the blob)
    if (T.env_emitter >= 0) {
        const psdr_envmap_rec *er = s->envmap;
        if (!er || !er->radiance || !er->cell_pmf || !er->cell_cmf || er->width < 2 || er->height < 2) return fail("EnvironmentMap emitter without a configured psdr_envmap_rec");
        EnvDev &ED = T.env;
        const size_t cells = (size_t) er->reso[0] * er->reso[1], texels = (size_t) 3 * er->width * er->height;
        const bool keep = same_env && Told.env_emitter >= 0 && Told.env.width == er->width && Told.env.height == er->height && Told.env.num_cells == (int) cells;
        const bool had_d = Told.env_emitter >= 0 && Told.env.d_radiance != nullptr;
        const int *old_guide = Told.env.cell_guide; const int old_guide_n = Told.env.guide_n;
        if (sync_named(sc, "env.radiance", er->radiance, texels * sizeof(float), keep, ED.radiance, info)) return 1;
        if (sync_named(sc, "env.cell_pmf", er->cell_pmf, cells * sizeof(float), keep, ED.cell_pmf, info)) return 1;
        if (sync_named(sc, "env.cell_cmf", er->cell_cmf, cells * sizeof(float), keep, ED.cell_cmf, info)) return 1;
        if (sync_named(sc, "env.d_radiance", er->d_radiance, texels * sizeof(float), keep && same_env_tan && had_d, ED.d_radiance, info)) return 1;
        if (keep) { ED.cell_guide = old_guide; ED.guide_n = old_guide_n; }
        else {
            ED.cell_guide = nullptr; ED.guide_n = 0;
            std::vector<int> guide;
            build_cdf_guide(er->cell_cmf, (int) cells, er->cell_sum, guide);
            if (!guide.empty()) {
                if (sync_named(sc, "env.guide", guide.data(), guide.size() * sizeof(int), false, ED.cell_guide, info)) return 1;
                ED.guide_n = (int) guide.size() - 1;
            }
        }
        ED.width = er->width; ED.height = er->height; ED.reso0 = er->reso[0]; ED.reso1 = er->reso[1]; ED.num_cells = (int) cells;
        ED.scale = er->scale; ED.cell_sum = er->cell_sum;
        std::memcpy(ED.to_world.m, er->to_world, 64); std::memcpy(ED.from_world.m, er->from_world, 64);
        std::memcpy(ED.d_from_world.m, er->d_from_world, 64); ED.d_scale = er->d_scale;
        for (int k = 0; k < 3; ++k) { ED.lower[k] = er->lower[k]; ED.upper[k] = er->upper[k]; }
        for (int k = 0; k < 4; ++k) { ED.xf[k] = er->radiance_xf[k]; ED.d_xf[k] = er->d_radiance_xf[k]; }
    } else T.env = EnvDev{};

    // ---- bitmap parameters: three slots per BSDF - [0] reflectance / diffuse reflectance (rgb), [1] specular (rgb), [2] roughness (1 channel)
    T.tex = nullptr;
    sc->tex_total = 0;
    sc->tex_layout.clear();
    {
        bool any_tex = false;
        for (int i = 0; i < s->n_bsdfs; ++i) any_tex |= s->bsdfs[i].tex_data != nullptr || s->bsdfs[i].spec_tex_data != nullptr || s->bsdfs[i].rough_tex_data != nullptr;
        if (any_tex) {
            std::vector<TexDev> td((size_t) 3 * s->n_bsdfs, TexDev{nullptr, nullptr, 0, 0, -1, {0.f, 1.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}});
            for (int i = 0; i < s->n_bsdfs; ++i) {
                const psdr_bsdf_rec &b = s->bsdfs[i];
                const float *src[3] = {b.tex_data, b.spec_tex_data, b.rough_tex_data}, *dsrc[3] = {b.d_tex_data, b.d_spec_tex_data, b.d_rough_tex_data};
                const int tw[3] = {b.tex_width, b.spec_tex_width, b.rough_tex_width}, th[3] = {b.tex_height, b.spec_tex_height, b.rough_tex_height};
                for (int k = 0; k < 3; ++k) {
                    if (!src[k]) continue;
                    if (k > 0 && b.type != 1 && b.type != 2 && !(b.type == 3 && k == 2)) return fail("this BSDF type has no second / third bitmap parameter");
                    if (tw[k] < 2 || th[k] < 2) return fail("Bitmap: invalid resolution!");
                    const size_t nt = (size_t) (k == 2 ? 1 : 3) * tw[k] * th[k];
                    const std::string key = "tex." + std::to_string(i) + "." + std::to_string(k);
                    TexDev &t = td[3 * (size_t) i + k];
                    const bool had = sc->named.count(key) != 0, had_d = sc->named.count(key + ".d") != 0;
                    if (sync_named(sc, key, src[k], nt * sizeof(float), same_bitmaps && had, t.data, info)) return 1;
                    if (sync_named(sc, key + ".d", dsrc[k], nt * sizeof(float), same_bitmaps && had_d, t.d_data, info)) return 1;
                    t.w = tw[k]; t.h = th[k];
                    t.g_off = sc->tex_total; sc->tex_total += (long long) nt;
                    for (int q = 0; q < 4; ++q) { t.xf[q] = b.tex_xf[k][q]; t.d_xf[q] = b.d_tex_xf[k][q]; }
                }
            }
            if (sync_named(sc, "tex.table", td.data(), td.size() * sizeof(TexDev), false, T.tex, info)) return 1;
            sc->tex_layout.resize(td.size());
            for (size_t i = 0; i < td.size(); ++i) sc->tex_layout[i] = td[i].g_off;
        }
    }
    T.mat = nullptr;
    {
        bool any = false;
        for (int i = 0; i < s->n_bsdfs; ++i) any |= s->bsdfs[i].type != 0;
        if (any) {
            std::vector<MatDev> md((size_t) s->n_bsdfs);
            for (int i = 0; i < s->n_bsdfs; ++i) {
                const psdr_bsdf_rec &b = s->bsdfs[i];
                MatDev &m = md[(size_t) i];
                for (int k = 0; k < 3; ++k) { m.specular[k] = b.specular[k]; m.d_specular[k] = b.d_specular[k]; }
                m.roughness = b.roughness; m.d_roughness = b.d_roughness;
                m.alpha_u = b.alpha_u; m.alpha_v = b.alpha_v; m.d_alpha_u = b.d_alpha_u; m.d_alpha_v = b.d_alpha_v;
                for (int k = 0; k < 3; ++k) { m.eta[k] = b.eta[k]; m.d_eta[k] = b.d_eta[k]; m.k[k] = b.k[k]; m.d_k[k] = b.d_k[k]; }
            }
            if (sync_named(sc, "mat.table", md.data(), md.size() * sizeof(MatDev), false, T.mat, info)) return 1;
        }
    }
    T.pv = nullptr; T.tri_fi = nullptr;
    {   // MicrofacetPerVertex: parameter arrays per BSDF + the mesh-local vertex ids of every triangle slot
        bool any_pv = false;
        for (int i = 0; i < s->n_bsdfs; ++i) any_pv |= s->bsdfs[i].type == 4;
        if (any_pv) {
            if (!tr.face_indices) return fail("MicrofacetPerVertex needs psdr_triangles.face_indices");
            std::vector<PvDev> pd((size_t) s->n_bsdfs, PvDev{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, {-1, -1, -1}});
            if (sc->tex_layout.size() < (size_t) 3 * s->n_bsdfs) sc->tex_layout.resize((size_t) 3 * s->n_bsdfs, -1);
            for (int i = 0; i < s->n_bsdfs; ++i) {
                const psdr_bsdf_rec &b = s->bsdfs[i];
                if (b.type != 4) continue;
                if (b.pv_count <= 0 || !b.pv_specular || !b.pv_diffuse || !b.pv_roughness) return fail("MicrofacetPerVertex: missing per-vertex data");
                const size_t nv = (size_t) b.pv_count;
                const std::string key = "pv." + std::to_string(i) + ".";
                PvDev &p = pd[(size_t) i];
                const float *srcs[6] = {b.pv_specular, b.d_pv_specular, b.pv_diffuse, b.d_pv_diffuse, b.pv_roughness, b.d_pv_roughness};
                const float **dsts[6] = {&p.spec, &p.d_spec, &p.diff, &p.d_diff, &p.rough, &p.d_rough};
                const size_t lens[6] = {3 * nv, 3 * nv, 3 * nv, 3 * nv, nv, nv};
                for (int q = 0; q < 6; ++q) {
                    const std::string kq = key + std::to_string(q);
                    const bool had = sc->named.count(kq) != 0;
                    if (sync_named(sc, kq, srcs[q], lens[q] * sizeof(float), same_bitmaps && had, *dsts[q], info)) return 1;
                }
                p.n = b.pv_count;
                // adjoint blocks in psdr_grads.g_tex, numbered like Microfacet's maps: 0 diffuse, 1 specular, 2 roughness
                const size_t sizes[3] = {3 * nv, 3 * nv, nv};
                for (int k = 0; k < 3; ++k) { p.g_off[k] = sc->tex_total; sc->tex_layout[3 * (size_t) i + k] = sc->tex_total; sc->tex_total += (long long) sizes[k]; }
            }
            // every mesh that uses a per-vertex BSDF must index inside its arrays
            for (int i = 0; i < n; ++i) {
                const int bid = s->meshes[tr.mesh_id[i]].bsdf_id;
                if (bid >= 0 && s->bsdfs[bid].type == 4)
                    for (int k = 0; k < 3; ++k)
                        if (tr.face_indices[3 * (size_t) i + k] < 0 || tr.face_indices[3 * (size_t) i + k] >= s->bsdfs[bid].pv_count) return fail("MicrofacetPerVertex: fewer values than mesh vertices");
            }
            const bool had_fi = sc->named.count("tri_fi") != 0;
            if (geo || !had_fi) {
                std::vector<int> fi((size_t) 3 * n);
                for (int slot = 0; slot < n; ++slot)
                    for (int k = 0; k < 3; ++k) fi[3 * (size_t) slot + k] = tr.face_indices[3 * (size_t) order[(size_t) slot] + k];
                if (sync_named(sc, "tri_fi", fi.data(), fi.size() * sizeof(int), false, T.tri_fi, info)) return 1;
            } else T.tri_fi = sc->buf("tri_fi").as<int>();
            if (sync_named(sc, "pv.table", pd.data(), pd.size() * sizeof(PvDev), false, T.pv, info)) return 1;
        }
    }
    if (build) {   // hot triangles of the reverse-mode accumulators: emitter meshes first, then by area, at most kHotMax (an update keeps the choice of the build: it only decides which rows accumulate in LDS)
        constexpr int kHotMax = 720;                                   // 720 x 22 floats = 62 KB of LDS
        std::vector<int> ord((size_t) n);
        std::vector<float> key((size_t) n);
        for (int i = 0; i < n; ++i) {
            ord[(size_t) i] = i;
            const float *a1 = tr.e1 + 3 * (size_t) i, *a2 = tr.e2 + 3 * (size_t) i;
            const float cx = a1[1] * a2[2] - a1[2] * a2[1], cy = a1[2] * a2[0] - a1[0] * a2[2], cz = a1[0] * a2[1] - a1[1] * a2[0];
            const bool emit = s->meshes[tr.mesh_id[i]].emitter_id >= 0;
            key[(size_t) i] = std::sqrt(cx * cx + cy * cy + cz * cz) * (emit ? 1e30f : 1.f);
        }
        sc->n_hot = std::min(n, kHotMax);
        // the n_hot largest keys, ties to the smaller index (what a stable sort of all keys puts first)
        auto before = [&](int x, int y) { return key[(size_t) x] > key[(size_t) y] || (key[(size_t) x] == key[(size_t) y] && x < y); };
        std::partial_sort(ord.begin(), ord.begin() + sc->n_hot, ord.end(), before);
        std::vector<int> hmap((size_t) std::max(1, n), -1), hinv((size_t) std::max(1, sc->n_hot), 0);
        for (int h = 0; h < sc->n_hot; ++h) { hmap[(size_t) ord[(size_t) h]] = h; hinv[(size_t) h] = ord[(size_t) h]; }
        if (sc->hot_map.upload(hmap.data(), hmap.size() * sizeof(int)) || sc->hot_inv.upload(hinv.data(), hinv.size() * sizeof(int))) return 1;
    }
    info.ms_fill = ms_since(t_fill);

    // ---- send the sections that were written (adjacent ones as one copy)
    const auto t_up = std::chrono::steady_clock::now();
    if (blob_moved) { dirty.clear(); dirty.push_back({0, w}); }
    std::sort(dirty.begin(), dirty.end(), [](const WordRange &a, const WordRange &b) { return a.b < b.b; });
    auto send = [&](size_t b, size_t e) -> int {
        if (e <= b) return 0;
        HIPCHK(hipMemcpyAsync((char *) sc->blob.p + 16 * b, (const char *) H + 16 * b, 16 * (e - b), hipMemcpyHostToDevice, nullptr));
        info.bytes_uploaded += (int64_t) (16 * (e - b));
        return 0;
    };
    for (size_t i = 0; i < dirty.size();) {
        size_t b = dirty[i].b, e = dirty[i].e, j = i + 1;
        while (j < dirty.size() && dirty[j].b <= e) { e = std::max(e, dirty[j].e); ++j; }
        if (blob_moved && !build && uses_bvh) {
            // (after a move of the allocation the nodes a refit wrote were carried over on the device: the host copy of a refitted tree is stale)
            const size_t nb = (size_t) T.nodes_off, ne = nb + (size_t) (kNodeFloats / 4) * (size_t) T.n_nodes;
            if (send(b, std::min(e, nb)) || send(std::max(b, ne), e)) return 1;
        } else if (send(b, e)) return 1;
        i = j;
    }
    info.ms_upload = ms_since(t_up);

    // ---- the tree follows the triangles
    info.sah_cost_built = sc->cost_built;
    info.sah_cost = sc->cost_built;
    if (!build && geo && uses_bvh) {
        const auto t_refit = std::chrono::steady_clock::now();
        double cost = 0.0;
        if (tree_refit(sc, &cost)) return 1;
        info.tree = 1;
        info.sah_cost = cost;
        info.ms_tree += ms_since(t_refit);
        // the topology no longer fits the geometry: build again (everything that depends on the triangle order is rewritten)
        if (!(cost <= kRebuildFactor * sc->cost_built)) {
            if (!rows_valid) return PSDR_HIP_NEED_ROWS;         // (the rows on the device are this state's, the tables are complete: the caller comes back with its rows and the tree is built then)
            return scene_sync(sc, s, 0, false, true, info_out);
        }
    }

    // ---- scene class, launch geometry
    const size_t stack_bytes = (size_t) T.stack_depth * kBlock * sizeof(int);
    const size_t blob_bytes = (size_t) T.blob_words * 16;
    // keeps >= 4 workgroups per CU (160 KiB LDS); the environment-map and texture code lives in the LDS=false kernels only (shade.h)
    bool no_lds = false;
#ifdef PSDR_DEV_KNOBS
    no_lds = std::getenv("PSDR_NO_LDS") != nullptr;      // measurement knob: run small scenes through the global-memory classes
#endif
    sc->lds = !no_lds && !uses_bvh && blob_bytes + stack_bytes <= 40 * 1024 && T.env_emitter < 0 && T.tex == nullptr && T.mat == nullptr && T.pv == nullptr;      // (LDS class = brute-force scenes)
    sc->lean = !sc->lds && T.tex == nullptr && T.mat == nullptr && T.pv == nullptr;
    // class 3: the same staging for small scenes WITH materials / bitmap parameters (the material and texture tables stay in global
    // memory; the triangle, BSDF, emitter and edge tables are what every path vertex reads)
    sc->lds_mat = !no_lds && !sc->lds && !uses_bvh && blob_bytes + stack_bytes <= 40 * 1024 && T.env_emitter < 0 && T.pv == nullptr;
    sc->smem_bytes = ((sc->lds || sc->lds_mat) ? blob_bytes : 0) + stack_bytes;
    if (sc->smem_bytes > 64 * 1024) return fail("BVH too deep for the LDS traversal stack");
    if (fresh) {
        if (sc->counters.upload(nullptr, sizeof(Counters))) return 1;
        if (sc->queues.upload(nullptr, sizeof(unsigned long long) * kQueueRing)) return 1;
    }
    if (build) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount; }
        // workgroups per launch: a multiple of what fits on the device (persistent workgroups pull work until the launch's queue is empty; the ones that start late find it empty)
        // (measured, round 4: brute-force scenes 4 per CU - what is resident at most - C3 forward -0.5 %, its backward pass 8.28 -> 8.06 ms; BVH scenes 8: config 5 218.7 ms, with 4 220.1)
        int per_cu = uses_bvh ? 8 : 4;
#ifdef PSDR_DEV_KNOBS
        if (const char *e = std::getenv("PSDR_GRID_PER_CU")) per_cu = std::max(1, std::atoi(e));
#endif
        sc->grid = cus * per_cu;
        T.gstack = nullptr; T.gstack_stride = 0;
        if (uses_bvh && sc->tree_max_stack > T.stack_lds) {
            // stack entries beyond the LDS part: one int per entry and lane of the largest grid any kernel is launched with - kTermSlices of them, because the
            // three terms of a renderD run as concurrent launches on forked streams (api.hip::render_impl) and a slice is indexed by workgroup and thread only
            const size_t stride = (size_t) sc->grid * kBlock;
            sc->gstack_slice = stride * (size_t) (sc->tree_max_stack - T.stack_lds);
            if (sc->gstack.ensure(sizeof(int) * sc->gstack_slice * (size_t) kTermSlices)) return 1;
            T.gstack = (int *) sc->gstack.p; T.gstack_stride = (int) stride;
        }
    }
    HIPCHK(hipStreamSynchronize(nullptr));
    info.ms_total = ms_since(t_start);
    sc->last_info = info;
    if (info_out) *info_out = info;
    return 0;
}

extern "C" {

int psdr_hip_scene_create(const psdr_scene_snapshot *s, psdr_hip_scene **out) {
    if (!s || !out) return fail("psdr_hip_scene_create: null argument");
    if (s->abi_version != PSDR_HIP_ABI_VERSION) return fail("psdr_hip_scene_create: ABI version mismatch");
    auto sc = std::make_unique<psdr_hip_scene>();
    if (scene_sync(sc.get(), s, 0, true, true, nullptr)) return 1;
    *out = sc.release();
    return 0;
}

int psdr_hip_scene_update(psdr_hip_scene *scene, const psdr_scene_snapshot *s, uint32_t same, psdr_update_info *info) {
    if (!scene || !s) return fail("psdr_hip_scene_update: null argument");
    if (s->abi_version != PSDR_HIP_ABI_VERSION) return fail("psdr_hip_scene_update: ABI version mismatch");
    // scene_sync rewrites the handle in place (tables, offsets, named buffers, sensors) while it validates and uploads: a failure midway - a validation message, out of
    // memory, an LDS budget - leaves old and new sections mixed.  Such a handle is POISONED: the render entry points refuse it, and the next update ignores what the
    // caller vouches for (its `same` bits are relative to a snapshot the device never fully received) and builds everything again.
    const bool was_poisoned = scene->poisoned;
    const int rc = scene_sync(scene, s, was_poisoned ? 0u : same, false, was_poisoned, info);
    if (rc == PSDR_HIP_NEED_ROWS) return rc;                  // (nothing was changed, or only rows that are this state's: see psdr_scene_snapshot.rows_valid)
    scene->poisoned = rc != 0;
    if (rc) scene->tree_tris = -1;
    return rc;
}

int psdr_hip_scene_destroy(psdr_hip_scene *scene) { delete scene; return 0; }
int psdr_hip_bvh_node_bytes(void) { return kNodeFloats * 4; }

int psdr_hip_scene_stats(const psdr_hip_scene *sc, int32_t *n_nodes, int32_t *n_leaves, int32_t *max_depth, int32_t *lds_bytes) {
    if (!sc) return fail("null scene");
    if (n_nodes) *n_nodes = sc->T.n_nodes;
    if (n_leaves) *n_leaves = sc->n_leaves;
    if (max_depth) *max_depth = sc->max_depth;
    if (lds_bytes) *lds_bytes = (int32_t) sc->smem_bytes * (sc->lds ? 1 : -1);
    return 0;
}

int psdr_hip_scene_last_update(const psdr_hip_scene *sc, psdr_update_info *info) {
    if (!sc || !info) return fail("null argument");
    *info = sc->last_info;
    return 0;
}

// Test aid (synchronises, downloads the sections): the triangle rows (traversal, shading, tangent) and the secondary-edge rows of the device blob against the rows the host
// path would write from `snapshot` - the check that the kernels of geometry_on_device produce the host's bits.  -> number of 32-bit words that differ.
int psdr_hip_scene_check_rows(const psdr_hip_scene *sc, const psdr_scene_snapshot *s, int64_t *mismatches) {
    if (!sc || !s || !mismatches) return fail("null argument");
    *mismatches = 0;
    const SceneTables &T = sc->T;
    const psdr_triangles &tr = s->tris;
    const int n = tr.n_triangles;
    if (n != T.n_tris) return fail("psdr_hip_scene_check_rows: another triangle count than the device scene's");
    HIPCHK(hipDeviceSynchronize());
    auto fetch = [&](int off_words, size_t words, std::vector<float> &dst) -> int {
        dst.resize(4 * words);
        if (words) HIPCHK(hipMemcpy(dst.data(), (const char *) sc->blob.p + 16 * (size_t) off_words, 16 * words, hipMemcpyDeviceToHost));
        return 0;
    };
    std::vector<float> trav, shade, tan, sec;
    if (fetch(T.trav_off, 3 * (size_t) n, trav) || fetch(T.shade_off, 6 * (size_t) n, shade)) return 1;
    if (T.has_tangent && fetch(T.tan_off, 6 * (size_t) n, tan)) return 1;
    long long bad = 0;
    auto cmp = [&](const float *dev, float x, float y, float z, float w) { const float h[4] = {x, y, z, w}; for (int k = 0; k < 4; ++k) bad += std::memcmp(dev + k, h + k, 4) != 0 ? 1 : 0; };
    for (int slot = 0; slot < n; ++slot) {
        const size_t o = (size_t) sc->order[(size_t) slot];
        const float *p0 = tr.p0 + 3 * o, *e1 = tr.e1 + 3 * o, *e2 = tr.e2 + 3 * o;
        const float *t = &trav[12 * (size_t) slot];
        cmp(t, p0[0], p0[1], p0[2], e1[0]); cmp(t + 4, e1[1], e1[2], e2[0], e2[1]); cmp(t + 8, e2[2], ibits((int32_t) o), 0.f, 0.f);
        const float *n0 = tr.n0 + 3 * o, *n1 = tr.n1 + 3 * o, *n2 = tr.n2 + 3 * o, *fn = tr.face_normal + 3 * o;
        const float *w = &shade[24 * (size_t) slot];
        cmp(w, n0[0], n0[1], n0[2], tr.face_area[o]); cmp(w + 4, n1[0], n1[1], n1[2], ibits(tr.mesh_id[o]));
        cmp(w + 8, n2[0], n2[1], n2[2], ibits(tr.use_face_normal && tr.use_face_normal[o] ? 1 : 0)); cmp(w + 12, fn[0], fn[1], fn[2], ibits((int32_t) o));
        if (T.has_tangent && tr.d_p0) {
            const float *a = tr.d_p0 + 3 * o, *b = tr.d_e1 + 3 * o, *c = tr.d_e2 + 3 * o, *d0 = tr.d_n0 + 3 * o, *d1 = tr.d_n1 + 3 * o, *d2 = tr.d_n2 + 3 * o, *df = tr.d_face_normal + 3 * o;
            const float *q = &tan[24 * (size_t) slot];
            cmp(q, a[0], a[1], a[2], b[0]); cmp(q + 4, b[1], b[2], c[0], c[1]); cmp(q + 8, c[2], d0[0], d0[1], d0[2]); cmp(q + 12, d1[0], d1[1], d1[2], d2[0]);
            cmp(q + 16, d2[1], d2[2], df[0], df[1]); cmp(q + 20, df[2], tr.d_face_area[o], 0.f, 0.f);
        }
    }
    const psdr_sec_edges &se = s->sec_edges;
    if (se.n_edges > 0 && se.n_edges == sc->E.n) {
        if (fetch(sc->E.off, 6 * (size_t) se.n_edges, sec)) return 1;
        const float z3[3] = {0.f, 0.f, 0.f};
        for (size_t i = 0; i < (size_t) se.n_edges; ++i) {
            const float *p0 = se.p0 + 3 * i, *e1 = se.e1 + 3 * i, *n0 = se.n0 + 3 * i, *n1 = se.n1 + 3 * i, *p2 = se.p2 + 3 * i;
            const float *dp0 = se.d_p0 ? se.d_p0 + 3 * i : z3, *de1 = se.d_e1 ? se.d_e1 + 3 * i : z3;
            const float *q = &sec[24 * i];
            cmp(q, p0[0], p0[1], p0[2], e1[0]); cmp(q + 4, e1[1], e1[2], n0[0], n0[1]); cmp(q + 8, n0[2], n1[0], n1[1], n1[2]);
            cmp(q + 12, p2[0], p2[1], p2[2], ibits(se.is_boundary[i] ? 1 : 0)); cmp(q + 16, dp0[0], dp0[1], dp0[2], de1[0]); cmp(q + 20, de1[1], de1[2], 0.f, 0.f);
        }
    }
    *mismatches = bad;
    return 0;
}

// Checks the device tree against the device triangles (test aid, synchronises): every node's quantised child boxes must contain the padded
// boxes of all triangles below the child, and every triangle slot must hang under exactly one leaf.
int psdr_hip_scene_check_tree(const psdr_hip_scene *sc, int64_t *violations) {
    if (!sc || !violations) return fail("null argument");
    *violations = 0;
    const SceneTables &T = sc->T;
    if (T.n_tris <= kBruteForceMax) return 0;
    HIPCHK(hipDeviceSynchronize());
    std::vector<float> nodes((size_t) kNodeFloats * (size_t) T.n_nodes), trav(12 * (size_t) T.n_tris);
    HIPCHK(hipMemcpy(nodes.data(), (const char *) sc->blob.p + 16 * (size_t) T.nodes_off, nodes.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(trav.data(), (const char *) sc->blob.p + 16 * (size_t) T.trav_off, trav.size() * sizeof(float), hipMemcpyDeviceToHost));
    const unsigned leaf_bit = 1u << (T.ref_bits - 1);
    // float box of every node; children come after their parent in memory, so a reverse sweep sees the children first
    std::vector<float> box(6 * (size_t) T.n_nodes);
    long long bad = 0;
    std::vector<int> seen((size_t) T.n_tris, 0);
    for (int i = T.n_nodes - 1; i >= 0; --i) {
        const float *q = &nodes[(size_t) kNodeFloats * (size_t) i];
        const uint32_t *u = reinterpret_cast<const uint32_t *>(q);
        float ulo[3] = {3e38f, 3e38f, 3e38f}, uhi[3] = {-3e38f, -3e38f, -3e38f};
        for (int k = 0; k < kBvhW; ++k) {
            const uint32_t code = u[kNodeCodeOff + k];
            if (code == 0xffffffffu) continue;
            float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
            if (code & leaf_bit) {
                const int payload = (int) (code & (leaf_bit - 1u)), first = payload >> 2, cnt = (payload & 3) + 1;
                for (int t = first; t < first + cnt; ++t) {
                    if (t < 0 || t >= T.n_tris) { ++bad; continue; }
                    seen[(size_t) t]++;
                    const float *r = &trav[12 * (size_t) t];
                    const float p0[3] = {r[0], r[1], r[2]}, e1[3] = {r[3], r[4], r[5]}, e2[3] = {r[6], r[7], r[8]};
                    float tl[3], th[3];
                    bvh_tri_box(p0, e1, e2, tl, th);
                    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], tl[a]); hi[a] = std::max(hi[a], th[a]); }
                }
            } else {
                if ((int) code <= i || (int) code >= T.n_nodes) { ++bad; continue; }
                for (int a = 0; a < 3; ++a) { lo[a] = box[6 * (size_t) code + a]; hi[a] = box[6 * (size_t) code + 3 + a]; }
            }
            // the quantised box of child k as the traversal decodes it
            for (int a = 0; a < 3; ++a) {
                const double step = std::ldexp(1.0, (int) ((u[3] >> (8 * a)) & 0xffu) - 127);
                constexpr int WW = kBvhW / 4;
                const uint32_t wl = u[4 + a * WW + (k >> 2)], wh = u[4 + (3 + a) * WW + (k >> 2)];
                const double ql = (double) q[a] + step * (double) ((wl >> (8 * (k & 3))) & 0xffu), qh = (double) q[a] + step * (double) ((wh >> (8 * (k & 3))) & 0xffu);
                if (!(ql <= (double) lo[a] && qh >= (double) hi[a])) ++bad;
            }
            for (int a = 0; a < 3; ++a) { ulo[a] = std::min(ulo[a], lo[a]); uhi[a] = std::max(uhi[a], hi[a]); }
        }
        for (int a = 0; a < 3; ++a) { box[6 * (size_t) i + a] = ulo[a]; box[6 * (size_t) i + 3 + a] = uhi[a]; }
    }
    for (int t = 0; t < T.n_tris; ++t) if (seen[(size_t) t] != 1) ++bad;
    *violations = bad;
    return 0;
}

int psdr_hip_scene_live_pixels(const psdr_hip_scene *sc, int32_t sensor_id, uint32_t *bits, int64_t *n_live) {
    if (!sc) return fail("null scene");
    if (sensor_id < 0 || sensor_id >= (int) sc->live_host.size()) return fail("Invalid sensor id!");
    const std::vector<unsigned> &m = sc->live_host[(size_t) sensor_id];
    const long long npx = (long long) sc->T.width * sc->T.height;
    long long count = 0;
    for (long long i = 0; i < (npx + 31) / 32; ++i) {
        unsigned w = m.empty() ? 0xffffffffu : m[(size_t) i];
        if (i == (npx + 31) / 32 - 1 && (npx & 31)) w &= (1u << (npx & 31)) - 1u;
        if (bits) bits[i] = w;
        count += __builtin_popcount(w);
    }
    if (n_live) *n_live = count;
    return 0;
}

} // extern "C"
