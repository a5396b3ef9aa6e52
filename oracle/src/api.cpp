// ORACLE — TEST INFRASTRUCTURE ONLY (see num.h header).
//
// api.cpp — C ABI of the oracle (oracle.h): the render drivers
// (reference src/integrator/integrator.cpp:12-198, src/integrator/path.cpp:130-168,274-294)
// looping over lanes with OpenMP, plus small accessors for the tests.
#include <omp.h>
#include <cstring>
#include <memory>
#include <string>
#include "integrator.h"

using namespace orc;

struct orc_scene { std::unique_ptr<Scene> sc; };
struct orc_guiding { Guiding g; };

static thread_local std::string g_err;
static int g_threads = 0;

extern "C" {

const char *orc_last_error(void) { return g_err.c_str(); }
void orc_set_num_threads(int n) { g_threads = n; }
int orc_get_num_threads(void) { return g_threads > 0 ? g_threads : omp_get_max_threads(); }

orc_scene *orc_scene_create(const orc_scene_desc *desc, const int *active, int n_active) {
    try {
        auto *s = new orc_scene;
        s->sc.reset(configure_scene(*desc, active, n_active));
        return s;
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
void orc_scene_destroy(orc_scene *s) { delete s; }

int orc_num_triangles(const orc_scene *s) { return (int) s->sc->tris.size(); }
int orc_num_sec_edges(const orc_scene *s) { return (int) s->sc->sec_edges.size(); }
int orc_num_primary_edges(const orc_scene *s, int c) { return (int) s->sc->cameras[c].edges.size(); }

static inline float sel(const Dual &x, int tangent) { return tangent ? x.d : x.v; }
static inline void put3(float *&o, const V3d &v, int tg) { *o++ = sel(v.x, tg); *o++ = sel(v.y, tg); *o++ = sel(v.z, tg); }

void orc_get_triangle_info(const orc_scene *s, int tg, float *out) {
    for (const Tri &t : s->sc->tris) {
        put3(out, t.p0, tg); put3(out, t.e1, tg); put3(out, t.e2, tg);
        put3(out, t.n0, tg); put3(out, t.n1, tg); put3(out, t.n2, tg); put3(out, t.fn, tg);
        *out++ = sel(t.area, tg);
        for (int k = 0; k < 3; ++k) { float f; std::memcpy(&f, &t.fi[k], 4); *out++ = f; }
    }
}
void orc_get_sec_edges(const orc_scene *s, int tg, float *out) {
    for (const SecEdge &e : s->sc->sec_edges) {
        put3(out, e.p0, tg); put3(out, e.e1, tg); put3(out, e.n0, tg); put3(out, e.n1, tg); put3(out, e.p2, tg);
        *out++ = e.is_boundary ? 1.f : 0.f;
    }
}
void orc_get_primary_edges(const orc_scene *s, int c, int tg, float *out) {
    for (const PrimEdge &e : s->sc->cameras[c].edges) {
        *out++ = sel(e.p0.x, tg); *out++ = sel(e.p0.y, tg); *out++ = sel(e.p1.x, tg); *out++ = sel(e.p1.y, tg);
        *out++ = tg ? 0.f : e.normal.x; *out++ = tg ? 0.f : e.normal.y; *out++ = tg ? 0.f : e.length;
    }
}
int orc_get_mesh_edges(const orc_scene *s, int mesh, int *out, int cap) {
    const auto &E = s->sc->meshes[mesh].edges;
    int n = (int) E.size();
    for (int i = 0; i < n && i < cap; ++i) { out[5 * i] = E[i].v0; out[5 * i + 1] = E[i].v1; out[5 * i + 2] = E[i].f0; out[5 * i + 3] = E[i].f1; out[5 * i + 4] = E[i].opp; }
    return n;
}
void orc_microfacet_eval(const float params[14], int two_sided, const float wi[3], const float wo[3], float out[6]) { orc::kat_microfacet_eval(params, two_sided, wi, wo, out); }
float orc_microfacet_pdf(float roughness, int two_sided, const float wi[3], const float wo[3]) { return orc::kat_microfacet_pdf(roughness, two_sided, wi, wo); }
int orc_microfacet_sample(float roughness, int two_sided, const float wi[3], const float s3[3], float wo_out[3], float *pdf_out) { return orc::kat_microfacet_sample(roughness, two_sided, wi, s3, wo_out, pdf_out); }
float orc_ggx_eval(float alpha, const float m[3]) { return orc::kat_ggx_eval(alpha, m); }
float orc_fresnel_conductor(float eta, float k, float c) { return orc::kat_fresnel_conductor(eta, k, c); }
void orc_dielectric_eval(const float q[2], const float wi[3], const float wo[3], float out[3]) { orc::kat_dielectric_eval(q, wi, wo, out); }
float orc_dielectric_pdf(const float q[2], const float wi[3], const float wo[3]) { return orc::kat_dielectric_pdf(q, wi, wo); }
int orc_dielectric_sample(const float q[2], const float wi[3], const float s3[3], float wo_out[3], float *pdf_out) { return orc::kat_dielectric_sample(q, wi, s3, wo_out, pdf_out); }
void orc_fresnel_dielectric(float eta, float c, float out[4]) { orc::kat_fresnel_dielectric(eta, c, out); }
void orc_set_direct_mis(orc_scene *s, int mis) { s->sc->direct_mis = mis; }
void orc_set_field(orc_scene *s, int field, int object, float intensity, float d_intensity) {
    s->sc->field = field; s->sc->field_object = object; s->sc->intensity = orc::Dual(intensity, d_intensity);
}
float orc_emitter_sampling_weight(const orc_scene *s, int e) { return s->sc->emitters[e].sampling_weight; }
/* Scene::m_lower / m_upper (scene.cpp:355-370, 383-387): what the reference logs as "[Scene] AABB" */
void orc_scene_aabb(const orc_scene *s, float bounds[6]) {
    const orc::Scene &sc = *s->sc;
    bounds[0] = sc.lower.x; bounds[1] = sc.lower.y; bounds[2] = sc.lower.z; bounds[3] = sc.upper.x; bounds[4] = sc.upper.y; bounds[5] = sc.upper.z;
}

int orc_envmap_info(const orc_scene *s, float bounds[6], int reso[2], float *cell_sum) {
    if (s->sc->env_emitter < 0) return 0;
    const orc::EnvmapC &E = s->sc->env;
    bounds[0] = E.lower.x; bounds[1] = E.lower.y; bounds[2] = E.lower.z; bounds[3] = E.upper.x; bounds[4] = E.upper.y; bounds[5] = E.upper.z;
    reso[0] = E.reso[0]; reso[1] = E.reso[1];
    *cell_sum = E.cell_distrb.sum;
    return 1;
}
void orc_env_sample(const orc_scene *s, int n, const float *ref_p, const float *s2, float *out_p, float *out_n, float *out_pdf) {
    for (int i = 0; i < n; ++i) {
        orc::V3f p, nn; float pdf;
        orc::kat_env_sample(*s->sc, orc::V3f(ref_p[3 * i], ref_p[3 * i + 1], ref_p[3 * i + 2]), s2[2 * i], s2[2 * i + 1], p, nn, pdf);
        out_p[3 * i] = p.x; out_p[3 * i + 1] = p.y; out_p[3 * i + 2] = p.z; out_n[3 * i] = nn.x; out_n[3 * i + 1] = nn.y; out_n[3 * i + 2] = nn.z; out_pdf[i] = pdf;
    }
}
void orc_env_pdf(const orc_scene *s, int n, const float *ref_p, const float *p, const float *nrm, float *out_pdf) {
    for (int i = 0; i < n; ++i)
        out_pdf[i] = orc::kat_env_pdf(*s->sc, orc::V3f(ref_p[3 * i], ref_p[3 * i + 1], ref_p[3 * i + 2]), orc::V3f(p[3 * i], p[3 * i + 1], p[3 * i + 2]),
                                      orc::V3f(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]));
}
void orc_envmap_cells(const orc_scene *s, float *pmf, float *cmf) {
    const orc::EnvmapC &E = s->sc->env;
    for (int i = 0; i < E.num_cells; ++i) { pmf[i] = E.cell_distrb.pmf[i]; cmf[i] = E.cell_distrb.cmf[i]; }
}

void orc_trace(const orc_scene *s, int n, const float *o, const float *d, int use_bvh, int *out_tri, float *out_uv, float *out_t) {
#pragma omp parallel for schedule(static) num_threads(orc_get_num_threads())
    for (int i = 0; i < n; ++i) {
        Hit h = trace_closest(*s->sc, V3f(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3f(d[3 * i], d[3 * i + 1], d[3 * i + 2]), use_bvh != 0);
        out_tri[i] = h.tri; out_uv[2 * i] = h.u; out_uv[2 * i + 1] = h.v; out_t[i] = h.t;
    }
}

} // extern "C"

// ---------------------------------------------------------------- interior term
static inline void seed_lane(LaneSampler &sm, const orc_sampler &spec, const int *pix_ids, int spp, int64_t lane) {
    uint64_t sv = spec.seed + (pix_ids ? (uint64_t) (int64_t) pix_ids[lane / spp] : (uint64_t) lane);
    sm.seed(sv, (uint64_t) lane);
    if (spec.skip) sm.rng.advance(spec.skip);
}
static inline void scrub(float &x) { if (!std::isfinite(x)) x = 0.f; }   // integrator.cpp:126

// Integrator::__render / __render_batch (integrator.cpp:103-176), one lane
template <bool ad>
static V3<Real<ad>> interior_lane(const Scene &sc, const CameraC &cam, int max_depth, bool hide, const orc_sampler &spec,
                                  const int *pix_ids, int64_t lane) {
    using R = Real<ad>;
    LaneSampler sm;
    seed_lane(sm, spec, pix_ids, sc.spp, lane);
    int64_t k = sc.spp > 1 ? lane / sc.spp : lane;
    int pix = pix_ids ? pix_ids[k] : (int) k;
    float bx = (float) (pix % sc.width), by = (float) (pix / sc.width);
    float jx = sm.next_1d(), jy = sm.next_1d();
    V2<R> samples(R((bx + jx) / (float) sc.width), R((by + jy) / (float) sc.height));
    Ray<ad> ray = sample_primary_ray<ad>(cam, samples);
    return Li<ad>(sc, sm, ray, true, max_depth, hide);
}

extern "C" {

int orc_render_c(const orc_scene *s, int sensor_id, int max_depth, int hide, orc_sampler spec, const int *pix_ids, int n_pix,
                 int64_t lane_begin, int64_t lane_end, float *out) {
    const Scene &sc = *s->sc;
    if (sensor_id < 0 || sensor_id >= (int) sc.cameras.size()) { g_err = "Invalid sensor id!"; return 1; }
    const CameraC &cam = sc.cameras[sensor_id];
    const int64_t npx = pix_ids ? n_pix : (int64_t) sc.width * sc.height;
    const int64_t N = npx * sc.spp;
    if (lane_end < 0 || lane_end > N) lane_end = N;
    if (lane_begin < 0) lane_begin = 0;
    std::memset(out, 0, sizeof(float) * 3 * npx);
    if (sc.spp <= 0) return 0;
    const int64_t px0 = lane_begin / sc.spp, px1 = (lane_end + sc.spp - 1) / sc.spp;
#pragma omp parallel for schedule(dynamic, 64) num_threads(orc_get_num_threads())
    for (int64_t px = px0; px < px1; ++px) {
        float acc[3] = {0, 0, 0};
        for (int64_t lane = std::max(px * sc.spp, lane_begin); lane < std::min((px + 1) * sc.spp, lane_end); ++lane) {
            V3f v = interior_lane<false>(sc, cam, max_depth, hide != 0, spec, pix_ids, lane);
            scrub(v.x); scrub(v.y); scrub(v.z);
            acc[0] += v.x; acc[1] += v.y; acc[2] += v.z;
        }
        for (int c = 0; c < 3; ++c) out[3 * px + c] = sc.spp > 1 ? acc[c] / (float) sc.spp : acc[c];
    }
    return 0;
}

int orc_li_lanes(const orc_scene *s, int sensor_id, int max_depth, int hide, orc_sampler spec, int64_t lane_begin, int64_t lane_end, float *out) {
    const Scene &sc = *s->sc;
    const CameraC &cam = sc.cameras[sensor_id];
#pragma omp parallel for schedule(dynamic, 256) num_threads(orc_get_num_threads())
    for (int64_t lane = lane_begin; lane < lane_end; ++lane) {
        V3f v = interior_lane<false>(sc, cam, max_depth, hide != 0, spec, nullptr, lane);
        float *o = out + 3 * (lane - lane_begin);
        o[0] = v.x; o[1] = v.y; o[2] = v.z;
    }
    return 0;
}

// ---------------------------------------------------------------- renderD
// multi-GPU sharding of the product (include/psdr_hip.h, psdr_render_args.shard_mode): 0 = rank r of c owns the 256-lane chunks k with k % c == r; 1 = contiguous runs of whole
// `unit` lanes (a pixel row of the interior sampler, a pixel of a batch list, a 256-lane chunk of an edge sampler): rank r owns units [r U, (r + 1) U), U = ceil(units / c)
struct Shard {
    int rank, count, mode; int64_t per;
    Shard(int r, int c, int m, int64_t n_lanes, int64_t unit) : rank(r), count(c), mode(m), per(0) {
        if (count > 1 && mode == 1) { const int64_t n_units = (n_lanes + unit - 1) / unit; per = ((n_units + count - 1) / count) * unit; }
    }
    bool has(int64_t lane) const {
        if (count <= 1) return true;
        return mode == 1 ? (lane >= rank * per && lane < (rank + 1) * per) : ((lane / 256) % count) == rank;
    }
};

int orc_render_d(const orc_scene *s, int sensor_id, int max_depth, int hide, const orc_sampler samplers[3], const int *pix_ids, int n_pix,
                 const orc_guiding *guiding, int terms, int shard_rank, int shard_count, int shard_mode, float *out, float *dout) {
    const Scene &sc = *s->sc;
    if (sensor_id < 0 || sensor_id >= (int) sc.cameras.size()) { g_err = "Invalid sensor id!"; return 1; }
    const CameraC &cam = sc.cameras[sensor_id];
    const int64_t npx = pix_ids ? n_pix : (int64_t) sc.width * sc.height;
    std::memset(out, 0, sizeof(float) * 3 * npx);
    std::memset(dout, 0, sizeof(float) * 3 * npx);
    const int nthreads = orc_get_num_threads();

    // interior integral (integrator.cpp:75-81)
    if ((terms & ORC_TERM_INTERIOR) && sc.spp > 0) {
        const int64_t lb = 0, le = npx * sc.spp;
        const int64_t px0 = 0, px1 = npx;
        const Shard shard(shard_rank, shard_count, shard_mode, le, pix_ids ? (int64_t) sc.spp : (int64_t) sc.width * sc.spp);
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
        for (int64_t px = px0; px < px1; ++px) {
            float acc[3] = {0, 0, 0}, dacc[3] = {0, 0, 0};
            for (int64_t lane = std::max(px * sc.spp, lb); lane < std::min((px + 1) * sc.spp, le); ++lane) {
                if (!shard.has(lane)) continue;
                V3d v = interior_lane<true>(sc, cam, max_depth, hide != 0, samplers[0], pix_ids, lane);
                for (int c = 0; c < 3; ++c) {
                    // the reference masks value where the primal is non-finite (integrator.cpp:126)
                    float pv = v[c].v, dv = v[c].d;
                    if (!std::isfinite(pv)) { pv = 0.f; dv = 0.f; }
                    if (!std::isfinite(dv)) dv = 0.f;
                    acc[c] += pv; dacc[c] += dv;
                }
            }
            for (int c = 0; c < 3; ++c) {
                out[3 * px + c] = sc.spp > 1 ? acc[c] / (float) sc.spp : acc[c];
                dout[3 * px + c] = sc.spp > 1 ? dacc[c] / (float) sc.spp : dacc[c];
            }
        }
    }
    if (pix_ids) return 0;   // edge terms scatter by full-frame pixel index; the batch path is interior-only here

    // boundary integrals scatter to arbitrary pixels: evaluate lanes in parallel, add serially in lane order
    const int64_t CH = 1 << 20;
    std::vector<int> idx(CH);
    std::vector<float> val(3 * CH);

    // primary edges (integrator.cpp:179-198)
    if ((terms & ORC_TERM_PRIMARY) && sc.sppe > 0 && cam.enable_edges) {
        const int64_t lb = 0, le = npx * sc.sppe;
        const Shard shard(shard_rank, shard_count, shard_mode, le, 256);
        for (int64_t c0 = lb; c0 < le; c0 += CH) {
            int64_t c1 = std::min(c0 + CH, le);
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
            for (int64_t lane = c0; lane < c1; ++lane) {
                idx[lane - c0] = -1;
                if (!shard.has(lane)) continue;
                LaneSampler sm;
                seed_lane(sm, samplers[1], nullptr, 1, lane);
                PrimaryEdgeSample es = sample_primary_edge(sc, cam, sm.next_1d());
                bool valid = es.idx >= 0;
                // Li(ray_n) is evaluated first, then Li(ray_p); both advance sampler 1
                V3f Ln = Li<false>(sc, sm, es.ray_n, valid, max_depth, hide != 0);
                V3f Lp = Li<false>(sc, sm, es.ray_p, valid, max_depth, hide != 0);
                V3f dL = (Ln - Lp) / es.pdf;
                int64_t o = lane - c0;
                idx[o] = es.idx;
                for (int c = 0; c < 3; ++c) {
                    float pv = es.x_dot_n.v * dL[c], dv = es.x_dot_n.d * dL[c];
                    if (!std::isfinite(pv) || !std::isfinite(dv)) dv = 0.f;     // integrator.cpp:188
                    if (sc.sppe > 1) dv /= (float) sc.sppe;
                    val[3 * o + c] = dv;                                         // value -= detach(value)
                }
            }
            for (int64_t o = 0; o < c1 - c0; ++o)
                if (idx[o] >= 0) for (int c = 0; c < 3; ++c) dout[3 * (int64_t) idx[o] + c] += val[3 * o + c];
        }
    }

    // secondary edges (path.cpp:274-294)
    if ((terms & ORC_TERM_SECONDARY) && sc.field < 0 && sc.sppse > 0 && !sc.sec_edges.empty()) {
        const int64_t lb = 0, le = npx * sc.sppse;
        const Shard shard(shard_rank, shard_count, shard_mode, le, 256);
        for (int64_t c0 = lb; c0 < le; c0 += CH) {
            int64_t c1 = std::min(c0 + CH, le);
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
            for (int64_t lane = c0; lane < c1; ++lane) {
                idx[lane - c0] = -1;
                if (!shard.has(lane)) continue;
                LaneSampler sm;
                seed_lane(sm, samplers[2], nullptr, 1, lane);
                V3f s3;
                s3.x = sm.next_1d(); s3.y = sm.next_1d(); s3.z = sm.next_1d();
                float pdf0 = guiding ? guiding->g.sample_reuse(s3) : 1.f;
                V3d v;
                int pi = eval_secondary_edge<true>(sc, cam, s3, v);
                int64_t o = lane - c0;
                idx[o] = pi;
                for (int c = 0; c < 3; ++c) {
                    float dv = v[c].d;
                    if (pdf0 > Epsilon) dv /= pdf0;
                    if (sc.sppse > 1) dv /= (float) sc.sppse;
                    if (!std::isfinite(dv)) dv = 0.f;
                    val[3 * o + c] = dv;
                }
            }
            for (int64_t o = 0; o < c1 - c0; ++o)
                if (idx[o] >= 0) for (int c = 0; c < 3; ++c) dout[3 * (int64_t) idx[o] + c] += val[3 * o + c];
        }
    }
    return 0;
}

// ---------------------------------------------------------------- guiding (path.cpp:130-168)
orc_guiding *orc_guiding_build(const orc_scene *s, int sensor_id, int max_depth, const int reso[4], int nrounds, int seed) {
    (void) max_depth;
    const Scene &sc = *s->sc;
    if (nrounds <= 0) { g_err = "nrounds > 0"; return nullptr; }
    const CameraC &cam = sc.cameras[sensor_id];
    auto *g = new orc_guiding;
    Guiding &G = g->g;
    for (int k = 0; k < 3; ++k) { G.reso[k] = reso[k]; G.unit[k] = 1.f / (float) reso[k]; }
    G.num_cells = reso[0] * reso[1] * reso[2];
    const int per = reso[3];
    const int64_t N = (int64_t) G.num_cells * per;
    std::vector<float> mass(G.num_cells, 0.f);
    for (int round = 0; round < nrounds; ++round) {
#pragma omp parallel for schedule(dynamic, 64) num_threads(orc_get_num_threads())
        for (int cell = 0; cell < G.num_cells; ++cell) {
            int c0 = cell / (reso[1] * reso[2]), rem = cell - c0 * (reso[1] * reso[2]);
            int c1 = rem / reso[2], c2 = rem - c1 * reso[2];
            float acc = 0.f;
            for (int j = 0; j < per; ++j) {
                int64_t lane = (int64_t) cell * per + j;
                LaneSampler sm;
                sm.seed((uint64_t) lane + (uint64_t) (int64_t) seed, (uint64_t) lane);
                sm.rng.advance((uint64_t) 3 * round);
                V3f s3;
                s3.x = sm.next_1d(); s3.y = sm.next_1d(); s3.z = sm.next_1d();
                s3 = V3f((s3.x + (float) c0) * G.unit[0], (s3.y + (float) c1) * G.unit[1], (s3.z + (float) c2) * G.unit[2]);
                V3f v;
                eval_secondary_edge<false>(sc, cam, s3, v);
                for (int c = 0; c < 3; ++c) { if (!std::isfinite(v[c])) v[c] = 0.f; if (per > 1) v[c] /= (float) per; }
                acc += std::max(v.x, std::max(v.y, v.z));
            }
            mass[cell] += acc;
        }
    }
    (void) N;
    if (nrounds > 1) for (float &m : mass) m /= (float) nrounds;
    try { G.distrb.init(mass); } catch (const std::exception &e) { g_err = e.what(); delete g; return nullptr; }
    return g;
}
int orc_guiding_num_cells(const orc_guiding *g) { return g->g.num_cells; }
void orc_guiding_get_mass(const orc_guiding *g, float *out) { std::memcpy(out, g->g.distrb.pmf.data(), sizeof(float) * g->g.num_cells); }
void orc_guiding_destroy(orc_guiding *g) { delete g; }

// ---------------------------------------------------------------- KAT building blocks
uint64_t orc_tea64(uint64_t v0, uint64_t v1) { return sample_tea_64(v0, v1); }
void orc_pcg32_raw(uint64_t initstate, uint64_t initseq, int n, uint32_t *out) {
    PCG32 r; r.seed(initstate, initseq);
    for (int i = 0; i < n; ++i) out[i] = r.next_uint32();
}
void orc_sampler_floats(uint64_t seed_value, uint64_t lane, uint64_t skip, int n, float *out) {
    LaneSampler s; s.seed(seed_value, lane);
    if (skip) s.rng.advance(skip);
    for (int i = 0; i < n; ++i) out[i] = s.next_1d();
}
void orc_square_to_cosine_hemisphere(int n, const float *uv, float *o) { for (int i = 0; i < n; ++i) kat_cosine_hemisphere(uv[2 * i], uv[2 * i + 1], o + 3 * i); }
void orc_square_to_uniform_triangle(int n, const float *uv, float *o) { for (int i = 0; i < n; ++i) kat_uniform_triangle(uv[2 * i], uv[2 * i + 1], o + 2 * i); }
void orc_coordinate_system(const float n[3], float s[3], float t[3]) { kat_coordinate_system(n, s, t); }
int orc_distrb_sample_reuse(int size, const float *pmf, float *sample, float *pdf) {
    Distrb d; d.init(std::vector<float>(pmf, pmf + size));
    return d.sample_reuse(*sample, *pdf);
}

} // extern "C"
