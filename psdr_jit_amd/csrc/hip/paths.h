// paths.h — persistent-lane path tracer with wave-level path regeneration.
//
// Stage r01a ran one lane = one sample from start to finish; the PMC pass showed only 33-50 % of the
// lanes active per VALU instruction because paths end at different depths (70 % of the camera rays of
// the README scene miss the box) and BVH walks differ in length.  Here every lane of a wave64 is a
// persistent worker: each loop iteration performs, for all lanes together,
//     [A] one shadow-ray trace (next-event estimation of the lane's current vertex) and
//     [B] one extension-ray trace (the camera ray of a freshly fetched sample, or the BSDF-sampled ray),
// and a lane whose path has ended fetches the next sample index before the next iteration
// (wave-aggregated: one atomicAdd per 256 samples, handed out with ballot/popcount prefix ranks).
// The per-lane arithmetic is exactly PathTracer::__Li (reference src/integrator/path.cpp:35-127) as in
// shade.h::Li — only the order in which samples are executed changes, so results are unchanged.
//
// MODE 0: interior term (Integrator::__render / __render_batch, reference integrator.cpp:103-176)
// MODE 1: primary-edge term (Integrator::render_primary_edges, integrator.cpp:179-198): one work item =
//         one edge sample = two consecutive paths (ray_n then ray_p) sharing the lane's RNG stream.
#pragma once
#include "edges.h"

namespace psdr {

struct PathParams {
    int max_depth, hide_emitters;
    unsigned long long seed;
    SkipAhead skip;                  // the sampler's draws so far, as pcg32's skip-ahead map (sampler.h)
    const int *pix_ids;
    long long begin, end;            // lane range of the sampler
    int shard_rank, shard_count;     // 256-lane chunks k with k % count == rank
    long long n_local;               // number of local work items (multiple of 256)
    unsigned long long *counter;     // work queue head (zeroed before launch)
    float *out, *dout, *lanes_out;
    // reverse mode of the edge terms (NULL = forward): w = d loss / d image; outputs are adjoints of the edge tables
    const float *adj_w;
    float *g_prim;                   // [n_primary_edges * 4]  (p0.xy, p1.xy in sample space)
    float *g_sec;                    // [n_sec_edges * 6]      (p0, e1)
    float *g_tri;                    // [n_tris * 22]
    float *g_cam;                    // [16] adjoint of the sensor's to_world through the camera ray of the secondary-edge term, or NULL
    int lds_acc, n_prim, n_sec;      // 1: the kernel accumulates the adjoint tables in LDS first (same-address atomics)
    int mis;                         // -1: PathTracer; 0/1/2: DirectIntegrator(mis), reference direct.cpp:34-132 (max_depth = 1)
    int sec_closed;                  // reverse mode of the secondary-edge term: 1 = closed form (two adjoint Moeller-Trumbore solves), 0 = record and probe
    // closed form on scenes whose tables do not fit LDS (lds_acc == 0): the rows of the scene's hot triangles (emitter meshes first - EVERY sample adds to the
    // emitter triangle its boundary ray ends on, under an environment map one of the 12 of the scene box -, then by area; adjoint.h) accumulate per workgroup
    const int *hot_map, *hot_inv;    // [n_tris] original triangle id -> hot index or -1; [n_hot] back
    int n_hot;                       // rows kept in LDS (9 floats each: p0, e1, e2), 0 = none
    int field, field_object;         // >= 0: first-hit integrator (shade.h first_hit_value), max_depth = 0
    float intensity, d_intensity;
    // primary-edge samples that cannot contribute are not traced: forward mode - the edge point's normal velocity is zero (psdr_render_args.skip_static_edges);
    // reverse mode - the row of g_prim the sample would add to is not wanted (psdr_grads.prim_edge_filter)
    int skip_static;
    const unsigned char *prim_filter;
};

// (x_dot_n.d == 0: the sample's d_out is 0 x (Ln - Lp) / pdf - zero, or a NaN that the accumulation drops)
PSDR_DEV bool edge_sample_idle(const PathParams &P, int ei, float xdn_d) {
    return P.adj_w == nullptr ? (P.skip_static != 0 && xdn_d == 0.f) : (P.prim_filter != nullptr && P.prim_filter[ei] == 0);
}

#ifndef PSDR_FETCH_BATCH
#define PSDR_FETCH_BATCH 256
#endif
constexpr int kFetchBatch = PSDR_FETCH_BATCH;      // work items a wave takes from the launch's queue per atomic
static_assert(kFetchBatch % 64 == 0, "PSDR_FETCH_BATCH");
#ifndef PSDR_REGEN_MIN
#define PSDR_REGEN_MIN 1
#endif
constexpr int kRegenMin = PSDR_REGEN_MIN;

// position of the r-th (0-based) set bit of m; r < popcount(m)
PSDR_DEV int nth_set_bit(unsigned long long m, int r) {
    unsigned w = (unsigned) m;
    int pos = 0;
    const int c0 = __popc(w);
    if (r >= c0) { r -= c0; w = (unsigned) (m >> 32); pos = 32; }
#pragma unroll
    for (int half = 16; half >= 1; half >>= 1) {
        const int cl = __popc(w & ((1u << half) - 1u));
        if (r >= cl) { r -= cl; w >>= half; pos += half; }
    }
    return pos;
}

// LIVE PIXELS ONLY (round 4).  64 % of the README frame is background: a sample there is seeded, its camera ray traced, and nothing comes of it.  Pixels no ray
// can leave towards a triangle are known to the host (SensorDev::live); the wave looks at the next 64 queue positions at once - every lane one position -, and the
// lanes that want a work item take the live ones in order.  Dead positions cost a bit test.  -> this lane's item (q_end: none), n_taken = queue positions used up
template <typename Params>
PSDR_DEV long long take_live_items(const SceneTables &T, const SensorDev &cam, const Params &P, bool wants, long long q_next, long long q_end, int lane_id,
                                   unsigned long long lt_mask, int &n_taken) {
    long long item = q_end;
    n_taken = 0;
#pragma nounroll
    for (int win = 0; win < 3; ++win) {
        const unsigned long long want = __ballot(wants && item >= q_end);
        const long long w0 = q_next + n_taken;
        if (want == 0ull || w0 >= q_end) break;
        const int wlen = q_end - w0 < 64 ? (int) (q_end - w0) : 64;
        bool live = false;
        if (lane_id < wlen) {
            const long long it = w0 + lane_id;
            const long long chunk = (it >> 8) * P.shard_count + P.shard_rank;
            const long long lane = P.begin + (chunk << 8) + (it & 255);
            if (lane < P.end) {
                const unsigned k = T.spp > 1 ? (unsigned) lane / (unsigned) T.spp : (unsigned) lane;      // (fewer than 2^31 lanes: the callers check P.end)
                const int pix = P.pix_ids ? P.pix_ids[k] : (int) k;
                live = (unsigned) pix >= (unsigned) (T.width * T.height) || ((cam.live[pix >> 5] >> (pix & 31)) & 1u) != 0u;      // (a pixel id outside the frame - batch rendering - is left to the path code)
            }
        }
        const unsigned long long lv = __ballot(live);
        const int n_live = __popcll(lv), n_want = __popcll(want);
        if (wants && item >= q_end) { const int r = __popcll(want & lt_mask); if (r < n_live) item = w0 + nth_set_bit(lv, r); }
        // the window is used up to its last taken position (all of it when every live position found a lane)
        n_taken += n_live <= n_want ? wlen : nth_set_bit(lv, n_want - 1) + 1;
        if (n_live >= n_want) break;
    }
    return item;
}

template <bool AD, int LDS, bool COUNT, int MODE>
PSDR_DEV void run_paths(SceneView<LDS> &S, const SensorDev &cam, const PathParams &P) {
    using R = Num<AD>; using V = VecN<AD>;
    const SceneTables &T = *S.T;
    const int lane_id = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane_id) - 1ull;
    const float inv_spp = T.spp > 1 ? 1.f / (float) T.spp : 1.f;

    // wave-uniform local queue
    long long q_next = 0, q_end = 0;
    bool exhausted = false;

    // per-lane path state
    bool busy = false;
    int depth = -1;                       // -1: the extension ray is a camera ray
    LaneRng rng; rng.state = 0; rng.inc = 1;
    Its<AD> its;                          // current vertex (valid when depth >= 0)
    its.valid = false; its.slot = -1; its.mesh = -1;
    V thr(R(1.f)), res(R(0.f));
    RayT<AD> ext;                         // pending extension ray
    int side = 0;                         // MODE 1: which of the two paths of the edge sample
    // What a lane only touches when it takes a work item and when its path ends - the pixel, the sample index, the edge sample
    // and the first path's radiance - waits in the lane's LDS column (kColdRows rows behind the blob): registers are what this
    // kernel runs out of, and a value parked here costs two LDS instructions per path instead of scratch traffic per vertex.
    lds_float_t *cold = (lds_float_t *) (reinterpret_cast<float *>(S.stack) + (T.stack_depth - kColdRows) * kBlock);
    enum { kLnX = 0, kLnY, kLnZ, kXdnV, kXdnD, kPdf, kEdgeS, kNx, kNy, kEdgeI, kPix, kHpSlot, kHpU, kHpV, kHpT, kLaneLo = 0, kLaneHi = 1 };
    auto cold_f = [&](int r) -> float { return cold[r * kBlock]; };
    auto cold_i = [&](int r) -> int { return __float_as_int(cold[r * kBlock]); };
    auto park_f = [&](int r, float v) { cold[r * kBlock] = v; };
    auto park_i = [&](int r, int v) { cold[r * kBlock] = __int_as_float(v); };
    static_assert(kColdRows >= 15, "scene_dev.h::kColdRows");
    static_assert(MODE == 0 || !AD, "the primary-edge paths are traced in C mode");

#if PSDR_DIAG == 6
    // phase timers (wave cycles, every lane adds the same delta): c_rays = fetch + regeneration, c_nodes = drawing the vertex's
    // two rays, c_tris = trace2, c_hits = consuming the hits + path end
    unsigned long long t_ph = __builtin_readcyclecounter();
#define PSDR_PHASE(field) do { if (COUNT) { const unsigned long long t_now = __builtin_readcyclecounter(); S.field += (unsigned) (t_now - t_ph); t_ph = t_now; } } while (0)
#else
#define PSDR_PHASE(field) do { } while (0)
#endif
    for (;;) {
        // MODE 1: the camera ray of an edge sample's SECOND path rides in the first path's first trace (whose next-event slot is idle: no vertex yet); its hit waits in
        // the lane's cold rows until the first path has ended.  Same rays, same sampler draws - one trace2 pass per edge sample less (8 -> 7 at depth 3).
        RayT<AD> cam_p; cam_p.o = V(R(0.f)); cam_p.d = V(R(0.f));
        bool trace_p = false;
        // ------------------------------------------------------------------ fetch work for idle lanes
        if (q_next >= q_end && !exhausted) {
            unsigned long long base = 0;
            if (lane_id == 0) base = atomicAdd(P.counter, (unsigned long long) kFetchBatch);
            base = __shfl(base, 0);
            if ((long long) base >= P.n_local) exhausted = true;
            else { q_next = (long long) base; q_end = q_next + kFetchBatch < P.n_local ? q_next + kFetchBatch : P.n_local; }
        }
        const unsigned long long need = __ballot(!busy);
        // regenerate only when enough lanes are idle: seeding a lane (two 64-bit TEA hashes + pcg32 seed, ~600
        // instructions) runs under the mask of the fetching lanes, so doing it every iteration for a handful of
        // lanes costs as much as doing it for a quarter of the wave
        if (need != 0ull && q_next < q_end && (__popcll(need) >= kRegenMin || __ballot(busy) == 0ull)) {
            const int n_need = __popcll(need);
            long long item = q_end;                       // the work item this lane starts (q_end: none)
            int n_taken;                                  // queue positions consumed
            if (MODE == 0 && !COUNT && cam.live != nullptr && P.lanes_out == nullptr && P.end < (1ll << 31)) {      // (the counted builds and the per-lane output see every sample)
                // LIVE PIXELS ONLY (round 4): take_live_items above
                item = take_live_items(T, cam, P, !busy, q_next, q_end, lane_id, lt_mask, n_taken);
            } else {
                item = q_next + __popcll(need & lt_mask);
                n_taken = n_need < (int) (q_end - q_next) ? n_need : (int) (q_end - q_next);
            }
            if (!busy && item < q_end) {
                const long long chunk = (item >> 8) * P.shard_count + P.shard_rank;
                const long long lane = P.begin + (chunk << 8) + (item & 255);
                if (lane < P.end) {
                    busy = true; depth = -1; thr = V(R(1.f)); res = V(R(0.f));
                    if (MODE == 0) {
                        const long long k = T.spp > 1 ? lane / T.spp : lane;
                        const int pix = P.pix_ids ? P.pix_ids[k] : (int) k;
                        park_i(kPix, (int) k);
                        if (P.lanes_out) { park_i(kLaneLo, (int) (unsigned) ((lane - P.begin) & 0xffffffffll)); park_i(kLaneHi, (int) ((lane - P.begin) >> 32)); }
                        rng.seed(P.seed + (P.pix_ids ? (unsigned long long) (long long) pix : (unsigned long long) lane), (unsigned long long) lane, P.skip);
                        const float bx = (float) (pix % T.width), by = (float) (pix / T.width);
                        const float jx = rng.next_1d(), jy = rng.next_1d();
                        ext = sample_primary_ray<AD>(cam, (bx + jx) / (float) T.width, (by + jy) / (float) T.height);
                    } else {
                        // PerspectiveCamera::sample_primary_edge, reference perspective.cpp:200-226
                        rng.seed(P.seed + (unsigned long long) lane, (unsigned long long) lane, P.skip);
                        float s = rng.next_1d(), pdf;
                        const int ei = sample_reuse_guided(cam.pe_guide, cam.pe_guide_n, cam.n_edges, cam.edge_sum, [&](int i) { return S.ldf(cam.pecdf_off, i); },
                                                    [&](int i) { return S.ldf(cam.pecdf_off, cam.n_edges + i); }, s, pdf);
                        const float4 r0 = S.ld(cam.pe_off + 3 * ei), r1 = S.ld(cam.pe_off + 3 * ei + 1), r2 = S.ld(cam.pe_off + 3 * ei + 2);
                        pdf = fdiv(pdf, r2.z);
                        const float nx = r2.x, ny = r2.y;
                        const float oms = 1.0f - s;
                        const Dual p0x(r0.x, r1.x), p0y(r0.y, r1.y), p1x(r0.z, r1.z), p1y(r0.w, r1.w);
                        const Dual px = fma_(p0x, oms, p1x * s), py = fma_(p0y, oms, p1y * s);
                        const Dual x_dot_n = fma_(py, ny, px * nx);
                        const int ix = (int) floorf(px.v * (float) T.width), iy = (int) floorf(py.v * (float) T.height);
                        const bool edge_valid = ix >= 0 && ix < T.width && iy >= 0 && iy < T.height && !edge_sample_idle(P, ei, x_dot_n.d);
                        park_i(kPix, edge_valid ? iy * T.width + ix : -1);
                        const RayT<false> ray_p = sample_primary_ray<false>(cam, px.v + kEdgeEpsilon * nx, py.v + kEdgeEpsilon * ny);
                        const RayT<false> ray_n = sample_primary_ray<false>(cam, px.v - kEdgeEpsilon * nx, py.v - kEdgeEpsilon * ny);
                        if constexpr (!AD) { ext = ray_n; cam_p = ray_p; trace_p = edge_valid; }      // (ray_p is rebuilt from (edge_i, edge_s) when the first path has ended: make_its wants it)
                        side = 0;
                        park_f(kXdnV, x_dot_n.v); park_f(kXdnD, x_dot_n.d); park_f(kPdf, pdf); park_f(kEdgeS, s); park_f(kNx, nx); park_f(kNy, ny); park_i(kEdgeI, ei);
                        if (!edge_valid) busy = false;       // Li(..., valid=false) contributes nothing
                    }
                }
            }
            q_next += n_taken;
        }
        PSDR_PHASE(c_rays);
        if (__ballot(busy) == 0ull) { if (exhausted && q_next >= q_end) break; continue; }

        // ------------------------------------------------------------------ [A] next-event estimation (path.cpp:47-83)
        const bool at_vertex = busy && depth >= 0;
        bool do_nee = false;
        PositionSample<AD> ps;
        V wod(R(0.f)); R dist_sqr(0.f), dist(0.f);
        RayT<AD> ray1; ray1.o = V(R(0.f)); ray1.d = V(R(0.f));
        if (at_vertex && P.mis != 1) {           // (DirectIntegrator(1) neither draws nor uses the emitter sample)
            const float sx = rng.next_1d(), sy = rng.next_1d();
            do_nee = mesh_emitter(S, its.mesh) < 0;
            if (do_nee) {
                ps = sample_emitter_position<AD, LDS>(S, detach(its.p), sx, sy);
                wod = ps.p - its.p;
                dist_sqr = squared_norm(wod);
                dist = safe_sqrt(dist_sqr);
                wod = wod / dist;
                ray1.o = its.p; ray1.d = wod;
            }
        }
        // ------------------------------------------------------------------ [B] extension ray (drawn before the traces
        // so that both rays of this vertex share one pass over the triangles; the draw order is unchanged)
        BSDFSample bs; bs.wo = Vec3f(0.f, 0.f, 1.f); bs.pdf = 1.f; bs.valid = true;
        const bool do_bsdf = at_vertex && P.mis != 0;      // (DirectIntegrator(0) stops after the emitter sample)
        if (at_vertex && !do_bsdf) bs.valid = false;
        if (do_bsdf) {
            const float s0 = rng.next_1d(), s1 = rng.next_1d(), s2 = rng.next_1d();
            (void) s0;
            bs = bsdf_sample<AD, LDS>(S, its, s0, s1, s2, true);
            ext.o = its.p; ext.d = to_world<AD>(its, bs.wo);
        }
        Hit h, hx;
        PSDR_PHASE(c_nodes);
        // an invalid BSDF sample (e.g. on the BSDF-less bounding cube of the environment map: wo = 0) ends the path whatever
        // its ray hits; a zero-direction ray would wander through every BVH node that contains its origin
        if (MODE == 1) { if (trace_p) ray1 = cam_p; }
        trace2<LDS, COUNT>(S, detach(ray1.o), detach(ray1.d), do_nee || (MODE == 1 && trace_p), detach(ext.o), detach(ext.d), busy && bs.valid, h, hx);
        PSDR_PHASE(c_tris);
        if (MODE == 1) { if (trace_p) { park_i(kHpSlot, h.slot); park_f(kHpU, h.u); park_f(kHpV, h.v); park_f(kHpT, h.t); } }
        {
            if (do_nee && h.slot >= 0) {
                if (COUNT) S.c_hits++;
                // shadow hits only need n, wi.z, t, J - except on the environment map, whose radiance is looked up along
                // the direction rebuilt from the shading frame (envmap.cpp:47-56)
                Its<AD> its1 = make_its<AD, LDS, false>(S, h, ray1, true);
                if constexpr (has_env(LDS)) { if (T.env_emitter >= 0 && mesh_emitter(S, its1.mesh) == T.env_emitter) its1 = make_its<AD, LDS, true>(S, h, ray1, true); }
                if ((detach(its1.t) > detach(dist) - kShadowEpsilon) && (mesh_emitter(S, its1.mesh) >= 0)) {
                    const R cos_val = dot(its1.n, -wod);
                    const R G_val = div_(abs_(cos_val), dist_sqr);
                    const V emitter_val = eval_Le<AD, LDS>(S, its1, true);
                    const V wo_local = to_local<AD>(its, wod);
                    V bsdf_val2 = bsdf_eval<AD, LDS>(S, its, wo_local, true);
                    bsdf_val2 = bsdf_val2 * div_(G_val * ps.J, R(ps.pdf));
                    const float pdf1 = bsdf_pdf<AD, LDS>(S, its, wo_local, true) * detach(G_val);
                    if (pdf1 != 0.f) res = res + thr * emitter_val * bsdf_val2 * R(P.mis == 0 ? 1.f : mis_weight(ps.pdf, pdf1));
                }
            }
        }

        bool finished = false;
        if (busy) {
            if (COUNT) { if (hx.slot >= 0) S.c_hits++; }
            const Its<AD> itx = make_its<AD, LDS, true>(S, hx, ext, depth >= 0);
            if (depth < 0) {
                // first hit: result = Le (path.cpp:38-43)
                its = itx;
                if (S.field >= 0) { if constexpr (has_mat(LDS)) res = first_hit_value<AD, LDS>(S, itx); }
                else if (!P.hide_emitters) res = eval_Le<AD, LDS>(S, itx, itx.valid);
                depth = 0;
                finished = !itx.valid || P.max_depth == 0;
            } else {
                // BSDF-sampled vertex (path.cpp:86-123)
                if (bs.valid && itx.valid) {
                    V bsdf_val;
                    float pdf0;
                    if constexpr (AD) {
                        V wo = (itx.p - its.p) / itx.t;
                        const R cos_val = dot(itx.n, -wo);
                        const R G_val = div_(abs_(cos_val), sqr(itx.t));
                        pdf0 = bs.pdf * G_val.v;
                        if (itx.t.v < kEpsilon) bsdf_val = V(R(0.f));
                        else bsdf_val = bsdf_eval<AD, LDS>(S, its, to_local<AD>(its, wo), true) * div_(G_val * itx.J, R(pdf0));
                    } else {
                        const float cos_val = dot(itx.n, -ext.d);
                        const float G_val = fdiv(fabsf(cos_val), sqr(itx.t));
                        pdf0 = bs.pdf * G_val;
                        if (itx.t < kEpsilon) bsdf_val = V(0.f);
                        else bsdf_val = vdiv_(bsdf_eval<AD, LDS>(S, its, bs.wo, true), bs.pdf);
                    }
                    const float weight2 = P.mis == 1 ? 1.f : mis_weight(pdf0, emitter_position_pdf<AD, LDS>(S, detach(its.p), itx));
                    thr = thr * bsdf_val;
                    res = res + eval_Le<AD, LDS>(S, itx, true) * thr * R(weight2);
                    its = itx;
                    depth += 1;
                    finished = depth >= P.max_depth;
                } else {
                    depth += 1;
                    finished = true;
                }
            }
        }

        // ------------------------------------------------------------------ path end: accumulate, or start the second edge path
        if (busy && finished) {
            if (MODE == 0) {
                const float pv[3] = {detach(res.x), detach(res.y), detach(res.z)};
                const float tv[3] = {tangent(res.x), tangent(res.y), tangent(res.z)};
                const int pix_slot = cold_i(kPix);
                if (P.lanes_out) {
                    const long long o = 3 * (((long long) cold_i(kLaneHi) << 32) | (long long) (unsigned) cold_i(kLaneLo));
                    P.lanes_out[o] = pv[0]; P.lanes_out[o + 1] = pv[1]; P.lanes_out[o + 2] = pv[2];
                }
                if (P.out) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {             // NaN/Inf scrub, integrator.cpp:126
                        const bool okp = finite_(pv[c]);
                        const float v = okp ? pv[c] : 0.f;
                        if (v != 0.f) atomicAdd(&P.out[3 * (long long) pix_slot + c], v * inv_spp);
                        if (AD) {
                            const float d = (okp && finite_(tv[c])) ? tv[c] : 0.f;
                            if (d != 0.f) atomicAdd(&P.dout[3 * (long long) pix_slot + c], d * inv_spp);
                        }
                    }
                }
                busy = false;
            } else {
                bool sample_done = side == 1;
                if (side == 0) {
                    { const Vec3f Ln = detach(res); park_f(kLnX, Ln.x); park_f(kLnY, Ln.y); park_f(kLnZ, Ln.z); }
                    // the reference's Li always draws 5 numbers per depth level; skip what this path left
                    if (depth < P.max_depth) rng.advance((unsigned long long) ((P.mis == 0 ? 2 : (P.mis == 1 ? 3 : 5)) * (P.max_depth - depth)));
                    side = 1; depth = -1; thr = V(R(1.f)); res = V(R(0.f));
                    if constexpr (!AD) {
                        // ray_p = sample_primary_ray(p + EdgeEpsilon * n): the same arithmetic as at the start of the work item
                        const int edge_i = cold_i(kEdgeI);
                        const float edge_s = cold_f(kEdgeS), edge_nx = cold_f(kNx), edge_ny = cold_f(kNy);
                        const float4 r0 = S.ld(cam.pe_off + 3 * edge_i);
                        const float oms = 1.0f - edge_s;
                        const float pxv = fmaf(r0.x, oms, r0.z * edge_s), pyv = fmaf(r0.y, oms, r0.w * edge_s);
                        ext = sample_primary_ray<false>(cam, pxv + kEdgeEpsilon * edge_nx, pyv + kEdgeEpsilon * edge_ny);
                        // ... and its hit was found in the first path's first trace: the second path starts AT its first vertex (path.cpp:38-43)
                        Hit hp; hp.slot = cold_i(kHpSlot); hp.u = cold_f(kHpU); hp.v = cold_f(kHpV); hp.t = cold_f(kHpT);
                        const Its<AD> itp = make_its<AD, LDS, true>(S, hp, ext, false);
                        its = itp;
                        if (S.field >= 0) { if constexpr (has_mat(LDS)) res = first_hit_value<AD, LDS>(S, itp); }
                        else if (!P.hide_emitters) res = eval_Le<AD, LDS>(S, itp, itp.valid);
                        depth = 0;
                        sample_done = !itp.valid || P.max_depth == 0;
                    }
                }
                if (sample_done) {
                    // value = x_dot_n * (Ln - Lp) / pdf, scrub, / sppe; only the tangent survives (integrator.cpp:187-192)
                    const Vec3f Lp = detach(res), Ln(cold_f(kLnX), cold_f(kLnY), cold_f(kLnZ));
                    const float edge_xdn_v = cold_f(kXdnV), edge_xdn_d = cold_f(kXdnD), edge_pdf = cold_f(kPdf);
                    const int pix_slot = cold_i(kPix);
                    const Vec3f dL = vdiv_(Ln - Lp, edge_pdf);
                    const float o3[3] = {dL.x, dL.y, dL.z};
                    if (P.adj_w == nullptr) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float pv = edge_xdn_v * o3[c];
                            float dv = edge_xdn_d * o3[c];
                            if (!finite_(pv) || !finite_(dv)) dv = 0.f;
                            if (T.sppe > 1) dv /= (float) T.sppe;
                            if (dv != 0.f) atomicAdd(&P.dout[3 * (long long) pix_slot + c], dv);
                        }
                    } else {
                        // adjoint: the tangent is d(x_dot_n) * o3 / sppe with d(x_dot_n) = n . ((1-s) d p0 + s d p1)
                        float kw = 0.f;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            float k = o3[c];
                            if (!finite_(edge_xdn_v * k) || !finite_(k)) k = 0.f;
                            if (T.sppe > 1) k /= (float) T.sppe;
                            kw += P.adj_w[3 * (long long) pix_slot + c] * k;
                        }
                        if (kw != 0.f) {
                            const int edge_i = cold_i(kEdgeI);
                            const float edge_s = cold_f(kEdgeS), edge_nx = cold_f(kNx), edge_ny = cold_f(kNy);
                            const float a = (1.0f - edge_s) * kw, b = edge_s * kw;
                            atomicAdd(&P.g_prim[4 * edge_i], edge_nx * a); atomicAdd(&P.g_prim[4 * edge_i + 1], edge_ny * a);
                            atomicAdd(&P.g_prim[4 * edge_i + 2], edge_nx * b); atomicAdd(&P.g_prim[4 * edge_i + 3], edge_ny * b);
                        }
                    }
                    busy = false;
                }
            }
        }
        PSDR_PHASE(c_hits);
    }
#undef PSDR_PHASE
}

// ---------------------------------------------------------------------------------------------------------------------------
// BVH scenes (more than kBruteForceMax triangles): the same per-lane arithmetic with the traversal DECOUPLED from the shading.
//
// In run_paths every iteration traces the two rays of all 64 lanes to completion before any lane shades: a wave waits for
// its slowest ray at every path vertex, and with rays of 10 to 200 traversal steps most lanes idle (measured on the 82 k-
// triangle scene of BASELINE config 5: 13 % of the lanes active per VALU instruction, 25 % on the 652-triangle tutorial box).
// Here a lane is in one of two conditions:
//     in flight - its vertex's rays are in the wave's ray queue / being walked (trav4.h), or
//     ready     - it has hits to consume, a new sample to fetch, and rays to generate.
// The wave alternates between a TRAVERSAL phase that runs until kShadeMin lanes have their rays back and a SHADING phase
// in which only the ready lanes consume their hits, regenerate and post their next two rays.  In the traversal phase every lane
// is a WORKER that pulls rays from the wave's queue - its own or anybody's (round 3; round 2 traced a lane's two rays in that
// lane, one after the other, and a lane whose rays were finished idled until the shading phase) - and a walk in progress stays in
// the worker's registers across shading phases.  The order of a lane's operations - and of its sampler draws - is unchanged:
// [consume hits of vertex k] [path end -> next sample] [draw + post the rays of vertex k+1].
#ifndef PSDR_SHADE_MIN
#define PSDR_SHADE_MIN 44
#endif
constexpr int kShadeMin = PSDR_SHADE_MIN;

// (round 5, measured and not kept here: posting the second edge path's camera ray beside the first's - what run_paths does - leaves config 5's primary-edge kernel at
//  163.6 ms (163.3 before): the four registers of the parked hit and the second make_its cost what the saved queue rounds give)
template <bool AD, int LDS, bool COUNT, int MODE>
PSDR_DEV void run_paths_async(SceneView<LDS> &S, const SensorDev &cam, const PathParams &P) {
    using R = Num<AD>; using V = VecN<AD>;
    const SceneTables &T = *S.T;
    const int lane_id = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane_id) - 1ull;
    const float inv_spp = T.spp > 1 ? 1.f / (float) T.spp : 1.f;

    long long q_next = 0, q_end = 0;
    bool exhausted = false;

    // per-lane path state (as in run_paths)
    bool busy = false;
    int depth = -1;
    LaneRng rng; rng.state = 0; rng.inc = 1;
    Its<AD> its;
    its.valid = false; its.slot = -1; its.mesh = -1;
    V thr(R(1.f)), res(R(0.f));
    RayT<AD> ext;
    long long lane = 0;
    int pix_slot = -1;
    int side = 0;
    Vec3f Ln(0.f);
    float edge_xdn_v = 0.f, edge_xdn_d = 0.f, edge_pdf = 1.f, edge_s = 0.f, edge_nx = 0.f, edge_ny = 0.f;
    int edge_i = 0;
    bool edge_valid = false;
    static_assert(MODE == 0 || !AD, "the primary-edge paths are traced in C mode");

    // what a vertex keeps between posting its rays and consuming their hits
    bool inflight = false, has_hits = false, do_nee = false, ext_traced = false;
    PositionSample<AD> ps;
    V wod(R(0.f)); R dist_sqr(0.f), dist(0.f);
    BSDFSample bs; bs.wo = Vec3f(0.f, 0.f, 1.f); bs.pdf = 1.f; bs.valid = true;
    Trav4 tr;                             // this lane as a traversal WORKER: the ray it walks may belong to any lane of the wave (trav4.h)
    tr.reset();
    int posted = 0;                       // this lane as an OWNER: which of its vertex's two rays are in the wave's queue

#if PSDR_DIAG == 8
    unsigned long long t_sh = 0ull;
#endif
    for (;;) {
#if PSDR_DIAG == 8
        if (COUNT) t_sh = __builtin_readcyclecounter();
#endif
        const unsigned long long m_fly = __ballot(inflight);
        const int n_can = __popcll(__ballot(!inflight && (has_hits || busy || (q_next < q_end) || !exhausted)));
        if (m_fly == 0ull || n_can >= kShadeMin) {
            const bool ready = !inflight;
#if PSDR_DIAG == 4
            if (COUNT) S.c_hits++;
#elif PSDR_DIAG == 5
            if (COUNT) { if (ready && (has_hits || busy)) S.c_hits++; }
#endif
            // ---------------------------------------------------------------- consume the hits of the lane's current vertex
            bool finished = false;
            if (ready && has_hits) {
                has_hits = false;
                Hit h, hx;
                h.slot = -1; h.u = h.v = h.t = 0.f; hx = h;
                if (posted & 1) h = t4_result(S, 0);
                if (posted & 2) hx = t4_result(S, 1);
                posted = 0;
                RayT<AD> ray1; ray1.o = its.p; ray1.d = wod;
                if (do_nee && h.slot >= 0) {
#if !PSDR_DIAG
                    if (COUNT) S.c_hits++;
#endif
                    Its<AD> its1 = make_its<AD, LDS, false>(S, h, ray1, true);
                    if constexpr (has_env(LDS)) { if (T.env_emitter >= 0 && mesh_emitter(S, its1.mesh) == T.env_emitter) its1 = make_its<AD, LDS, true>(S, h, ray1, true); }
                    if ((detach(its1.t) > detach(dist) - kShadowEpsilon) && (mesh_emitter(S, its1.mesh) >= 0)) {
                        const R cos_val = dot(its1.n, -wod);
                        const R G_val = div_(abs_(cos_val), dist_sqr);
                        const V emitter_val = eval_Le<AD, LDS>(S, its1, true);
                        const V wo_local = to_local<AD>(its, wod);
                        V bsdf_val2 = bsdf_eval<AD, LDS>(S, its, wo_local, true);
                        bsdf_val2 = bsdf_val2 * div_(G_val * ps.J, R(ps.pdf));
                        const float pdf1 = bsdf_pdf<AD, LDS>(S, its, wo_local, true) * detach(G_val);
                        if (pdf1 != 0.f) res = res + thr * emitter_val * bsdf_val2 * R(P.mis == 0 ? 1.f : mis_weight(ps.pdf, pdf1));
                    }
                }
#if !PSDR_DIAG
                if (COUNT) { if (hx.slot >= 0) S.c_hits++; }
#endif
                const Its<AD> itx = make_its<AD, LDS, true>(S, hx, ext, depth >= 0);
                if (depth < 0) {
                    its = itx;
                    if (S.field >= 0) { if constexpr (has_mat(LDS)) res = first_hit_value<AD, LDS>(S, itx); }
                    else if (!P.hide_emitters) res = eval_Le<AD, LDS>(S, itx, itx.valid);
                    depth = 0;
                    finished = !itx.valid || P.max_depth == 0;
                } else {
                    if (bs.valid && itx.valid) {
                        V bsdf_val;
                        float pdf0;
                        if constexpr (AD) {
                            V wo = (itx.p - its.p) / itx.t;
                            const R cos_val = dot(itx.n, -wo);
                            const R G_val = div_(abs_(cos_val), sqr(itx.t));
                            pdf0 = bs.pdf * G_val.v;
                            if (itx.t.v < kEpsilon) bsdf_val = V(R(0.f));
                            else bsdf_val = bsdf_eval<AD, LDS>(S, its, to_local<AD>(its, wo), true) * div_(G_val * itx.J, R(pdf0));
                        } else {
                            const float cos_val = dot(itx.n, -ext.d);
                            const float G_val = fdiv(fabsf(cos_val), sqr(itx.t));
                            pdf0 = bs.pdf * G_val;
                            if (itx.t < kEpsilon) bsdf_val = V(0.f);
                            else bsdf_val = vdiv_(bsdf_eval<AD, LDS>(S, its, bs.wo, true), bs.pdf);
                        }
                        const float weight2 = P.mis == 1 ? 1.f : mis_weight(pdf0, emitter_position_pdf<AD, LDS>(S, detach(its.p), itx));
                        thr = thr * bsdf_val;
                        res = res + eval_Le<AD, LDS>(S, itx, true) * thr * R(weight2);
                        its = itx;
                        depth += 1;
                        finished = depth >= P.max_depth;
                    } else {
                        depth += 1;
                        finished = true;
                    }
                }
            }
            // ---------------------------------------------------------------- path end: accumulate, or start the second edge path
            if (ready && busy && finished) {
                if (MODE == 0) {
                    const float pv[3] = {detach(res.x), detach(res.y), detach(res.z)};
                    const float tv[3] = {tangent(res.x), tangent(res.y), tangent(res.z)};
                    if (P.lanes_out) {
                        const long long o = 3 * (lane - P.begin);
                        P.lanes_out[o] = pv[0]; P.lanes_out[o + 1] = pv[1]; P.lanes_out[o + 2] = pv[2];
                    }
                    if (P.out) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {             // NaN/Inf scrub, integrator.cpp:126
                            const bool okp = finite_(pv[c]);
                            const float v = okp ? pv[c] : 0.f;
                            if (v != 0.f) atomicAdd(&P.out[3 * (long long) pix_slot + c], v * inv_spp);
                            if (AD) {
                                const float d = (okp && finite_(tv[c])) ? tv[c] : 0.f;
                                if (d != 0.f) atomicAdd(&P.dout[3 * (long long) pix_slot + c], d * inv_spp);
                            }
                        }
                    }
                    busy = false;
                } else {
                    if (side == 0) {
                        Ln = detach(res);
                        if (depth < P.max_depth) rng.advance((unsigned long long) ((P.mis == 0 ? 2 : (P.mis == 1 ? 3 : 5)) * (P.max_depth - depth)));
                        side = 1; depth = -1; thr = V(R(1.f)); res = V(R(0.f));
                        if constexpr (!AD) {
                            const float4 r0 = S.ld(cam.pe_off + 3 * edge_i);
                            const float oms = 1.0f - edge_s;
                            const float pxv = fmaf(r0.x, oms, r0.z * edge_s), pyv = fmaf(r0.y, oms, r0.w * edge_s);
                            ext = sample_primary_ray<false>(cam, pxv + kEdgeEpsilon * edge_nx, pyv + kEdgeEpsilon * edge_ny);
                        }
                    } else {
                        const Vec3f Lp = detach(res);
                        const Vec3f dL = vdiv_(Ln - Lp, edge_pdf);
                        const float o3[3] = {dL.x, dL.y, dL.z};
                        if (P.adj_w == nullptr) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const float pv = edge_xdn_v * o3[c];
                                float dv = edge_xdn_d * o3[c];
                                if (!finite_(pv) || !finite_(dv)) dv = 0.f;
                                if (T.sppe > 1) dv /= (float) T.sppe;
                                if (dv != 0.f) atomicAdd(&P.dout[3 * (long long) pix_slot + c], dv);
                            }
                        } else {
                            float kw = 0.f;
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                float k = o3[c];
                                if (!finite_(edge_xdn_v * k) || !finite_(k)) k = 0.f;
                                if (T.sppe > 1) k /= (float) T.sppe;
                                kw += P.adj_w[3 * (long long) pix_slot + c] * k;
                            }
                            if (kw != 0.f) {
                                const float a = (1.0f - edge_s) * kw, b = edge_s * kw;
                                atomicAdd(&P.g_prim[4 * edge_i], edge_nx * a); atomicAdd(&P.g_prim[4 * edge_i + 1], edge_ny * a);
                                atomicAdd(&P.g_prim[4 * edge_i + 2], edge_nx * b); atomicAdd(&P.g_prim[4 * edge_i + 3], edge_ny * b);
                            }
                        }
                        busy = false;
                    }
                }
            }
            // ---------------------------------------------------------------- fetch work for the ready lanes without a path
            if (q_next >= q_end && !exhausted) {
                unsigned long long base = 0;
                if (lane_id == 0) base = atomicAdd(P.counter, (unsigned long long) kFetchBatch);
                base = __shfl(base, 0);
                if ((long long) base >= P.n_local) exhausted = true;
                else { q_next = (long long) base; q_end = q_next + kFetchBatch < P.n_local ? q_next + kFetchBatch : P.n_local; }
            }
            const unsigned long long need = __ballot(ready && !busy);
            if (need != 0ull && q_next < q_end) {
                const int n_need = __popcll(need);
                long long item;
                int n_taken;
                if (MODE == 0 && !COUNT && cam.live != nullptr && P.lanes_out == nullptr && P.end < (1ll << 31)) item = take_live_items(T, cam, P, ready && !busy, q_next, q_end, lane_id, lt_mask, n_taken);
                else { item = q_next + __popcll(need & lt_mask); n_taken = n_need < (int) (q_end - q_next) ? n_need : (int) (q_end - q_next); }
                if (ready && !busy && item < q_end) {
                    const long long chunk = (item >> 8) * P.shard_count + P.shard_rank;
                    lane = P.begin + (chunk << 8) + (item & 255);
                    if (lane < P.end) {
                        busy = true; depth = -1; thr = V(R(1.f)); res = V(R(0.f));
                        if (MODE == 0) {
                            const long long k = T.spp > 1 ? lane / T.spp : lane;
                            const int pix = P.pix_ids ? P.pix_ids[k] : (int) k;
                            pix_slot = (int) k;
                            rng.seed(P.seed + (P.pix_ids ? (unsigned long long) (long long) pix : (unsigned long long) lane), (unsigned long long) lane, P.skip);
                            const float bx = (float) (pix % T.width), by = (float) (pix / T.width);
                            const float jx = rng.next_1d(), jy = rng.next_1d();
                            ext = sample_primary_ray<AD>(cam, (bx + jx) / (float) T.width, (by + jy) / (float) T.height);
                        } else {
                            rng.seed(P.seed + (unsigned long long) lane, (unsigned long long) lane, P.skip);
                            float s = rng.next_1d(), pdf;
                            const int ei = sample_reuse_guided(cam.pe_guide, cam.pe_guide_n, cam.n_edges, cam.edge_sum, [&](int i) { return S.ldf(cam.pecdf_off, i); },
                                                        [&](int i) { return S.ldf(cam.pecdf_off, cam.n_edges + i); }, s, pdf);
                            const float4 r0 = S.ld(cam.pe_off + 3 * ei), r1 = S.ld(cam.pe_off + 3 * ei + 1), r2 = S.ld(cam.pe_off + 3 * ei + 2);
                            pdf = fdiv(pdf, r2.z);
                            const float nx = r2.x, ny = r2.y;
                            const float oms = 1.0f - s;
                            const Dual p0x(r0.x, r1.x), p0y(r0.y, r1.y), p1x(r0.z, r1.z), p1y(r0.w, r1.w);
                            const Dual px = fma_(p0x, oms, p1x * s), py = fma_(p0y, oms, p1y * s);
                            const Dual x_dot_n = fma_(py, ny, px * nx);
                            const int ix = (int) floorf(px.v * (float) T.width), iy = (int) floorf(py.v * (float) T.height);
                            edge_valid = ix >= 0 && ix < T.width && iy >= 0 && iy < T.height && !edge_sample_idle(P, ei, x_dot_n.d);
                            pix_slot = edge_valid ? iy * T.width + ix : -1;
                            const RayT<false> ray_n = sample_primary_ray<false>(cam, px.v - kEdgeEpsilon * nx, py.v - kEdgeEpsilon * ny);
                            if constexpr (!AD) ext = ray_n;
                            side = 0;
                            edge_xdn_v = x_dot_n.v; edge_xdn_d = x_dot_n.d; edge_pdf = pdf; edge_s = s; edge_nx = nx; edge_ny = ny; edge_i = ei;
                            if (!edge_valid) busy = false;
                        }
                    }
                }
                q_next += n_taken;
            }
            // ---------------------------------------------------------------- draw and post the rays of the lane's next vertex
            if (ready && busy) {
                const bool at_vertex = depth >= 0;
                do_nee = false;
                wod = V(R(0.f)); dist_sqr = R(0.f); dist = R(0.f);
                if (at_vertex && P.mis != 1) {
                    const float sx = rng.next_1d(), sy = rng.next_1d();
                    do_nee = mesh_emitter(S, its.mesh) < 0;
                    if (do_nee) {
                        ps = sample_emitter_position<AD, LDS>(S, detach(its.p), sx, sy);
                        wod = ps.p - its.p;
                        dist_sqr = squared_norm(wod);
                        dist = safe_sqrt(dist_sqr);
                        wod = wod / dist;
                    }
                }
                bs.wo = Vec3f(0.f, 0.f, 1.f); bs.pdf = 1.f; bs.valid = true;
                const bool do_bsdf = at_vertex && P.mis != 0;
                if (at_vertex && !do_bsdf) bs.valid = false;
                if (do_bsdf) {
                    const float s0 = rng.next_1d(), s1 = rng.next_1d(), s2 = rng.next_1d();
                    bs = bsdf_sample<AD, LDS>(S, its, s0, s1, s2, true);
                    ext.o = its.p; ext.d = to_world<AD>(its, bs.wo);
                }
                ext_traced = bs.valid;
                // the emitter sample only counts when the shadow ray's closest hit lies at the sample (t > dist - ShadowEpsilon): any hit
                // clearly in front of it settles that, so the shadow ray stops at the first such hit instead of looking for the closest
                posted = t4_post(S, detach(its.p), detach(wod), do_nee, detach(ext.o), detach(ext.d), ext_traced, do_nee ? (detach(dist) - kShadowEpsilon) * 0.9999f : -__builtin_inff());
                inflight = true;
            }
            if (__ballot(busy || inflight) == 0ull && exhausted && q_next >= q_end) break;
        }
#if PSDR_DIAG == 8
        if (COUNT) S.c_rays += (unsigned) (__builtin_readcyclecounter() - t_sh);
#endif
        // -------------------------------------------------------------------- traversal: until enough lanes have finished
        {
            const int n_fly = __popcll(__ballot(inflight));
            if (n_fly > 0) {
                const int want_new = n_fly < 2 * kShadeMin ? (n_fly + 1) / 2 : kShadeMin;
                const bool mine_done = trav4_run<LDS, COUNT, true>(S, tr, inflight ? posted : 0, n_fly - want_new);
                if (inflight && mine_done) { inflight = false; has_hits = true; }
            }
        }
    }
}

} // namespace psdr
