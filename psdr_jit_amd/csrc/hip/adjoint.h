// adjoint.h — reverse-mode ("radiative backprop" style) derivative of the interior term.
//
// Given w = d(loss)/d(image), accumulate  g[q] = sum over samples of  w_pixel . d(Li)/d(q) / spp  for every
// scene quantity q the forward tangent of renderD can enter through:
//     triangle rows  [p0 e1 e2 n0 n1 n2 face_normal area]  (22 floats per triangle, the reference's TriangleInfo),
//     diffuse reflectance (3 per BSDF), area-light radiance (3 per emitter).
// The host chains these to vertices / transforms (psdr_jit_amd/__init__.py).
//
// Method: per contributing path, (1) run the D-mode path tracer once while RECORDING the hit of every ray,
// (2) for each scene quantity the path touched, re-run the same D-mode shading with a ONE-HOT tangent on that
// quantity, REPLAYING the recorded hits instead of traversing (scene_dev.h: SceneView::mode / probe_*).  The
// forward tangent is linear in the tangent data, so each probe returns one column of the path's Jacobian; no
// separate adjoint derivation exists that could disagree with the forward-mode code the parity tests pin.
// Cost: ~(22 x touched triangles + 3 x touched BSDFs + 3) shading replays per path.
#pragma once
#include "paths.h"

namespace psdr {

// per-lane LDS records of one path, sized from the path depth D by the launch: 1 + 2 D hits of 4 words, 2 D light-sample slots,
// 4 D + 4 lookups of 3 words (per vertex: the BSDF's bitmaps and, under a normal map, the nested BSDF's; per bounce: two environment lookups)
inline __host__ __device__ int adj_hit_words(int depth) { return 4 * (1 + 2 * depth); }
inline __host__ __device__ int adj_ext_words(int depth) { return 2 * depth > 2 ? 2 * depth : 2; }
inline __host__ __device__ int adj_lk_entries(int depth) { return 4 * depth + 4; }
// LDS words per lane of the interior adjoint kernel: scenes without bitmap / per-vertex parameters and without an environment map
// make no lookups, carry no lookup record and keep more workgroups per CU (AdjointParams::lk_words)
// the reverse sweep keeps, per lane: 3 words per vertex (slot, u, v) and 11 per bounce (light sample, shadow hit, constants, throughput)
inline __host__ __device__ int adj_sweep_words(int depth) { return 3 * (depth + 1) + 11 * (depth > 0 ? depth : 1); }
inline __host__ __device__ int adj_lane_words(int depth, bool with_lookups) { return adj_hit_words(depth) + adj_ext_words(depth) + (with_lookups ? 3 * adj_lk_entries(depth) : 0); }
// the secondary-edge adjoint records three hits per lane, followed by 16 floats of camera-pose accumulators
constexpr int kSecAdjLaneWords = 12;
constexpr int kSecAdjScratch = kSecAdjLaneWords * kBlock + 16;

struct AdjointParams {
    int max_depth, hide_emitters;
    unsigned long long seed;
    SkipAhead skip;                  // the sampler's draws so far, as pcg32's skip-ahead map (sampler.h)
    const int *pix_ids;
    long long begin, end;
    int shard_rank, shard_count;
    long long n_local;
    unsigned long long *counter;
    const float *w;                 // d loss / d image, [n_pixels * 3]
    float *g_tri;                   // [n_tris * 22], ORIGINAL triangle order
    float *g_bsdf, *g_emitter;      // [n_bsdfs * 3], [n_emitters * 3]
    // Adjoint rows accumulate per workgroup in LDS for the scene's "hot" triangles - emitter meshes (every path's light samples
    // land there) and the largest ones (hit most often) - and go to global memory with plain atomics for the rest.  Millions of paths
    // adding to the 44 floats of a two-triangle luminaire in global memory serialise: 830 ms instead of 40 on the sphere scene.
    const int *hot_map;             // [n_tris] original triangle id -> hot index, -1 = not hot
    const int *hot_inv;             // [n_hot] hot index -> original triangle id
    int n_hot;
    int mis;                        // -1: PathTracer; 0/1/2: DirectIntegrator(mis)
    int field, field_object;        // >= 0: first-hit integrator
    float intensity, d_intensity;
    const unsigned char *mesh_filter;   // [n_meshes] or NULL: only these meshes' triangle rows are probed
    int skip_bsdf, skip_emitter;
    float *g_tex;                   // texel adjoints of the bitmap parameters (TexDev::g_off), or NULL
    float *g_cam;                   // [16] adjoint of the sensor's to_world (row major, rows 0-2 filled), or NULL
    float *g_env, *g_env_scale;     // texel adjoints [H*W*3] and scale adjoint [1] of the environment map, or NULL
    float *g_env_xf;                // [16] adjoint of the environment map's from_world (rows 0-2, columns 0-2 filled), or NULL
    float *g_uv_xf;                 // [(3 * n_bsdfs + 1) * 4] adjoints of the bitmaps' uv transforms (psdr_grads.g_uv_xf), or NULL
    int hit_words, ext_words, lk_words;   // sizes of the three per-lane records (lk_words = 0 when the scene cannot make lookups)
    float *g_mat;                   // [n_bsdfs*16] adjoints of the constant parameters of the GGX BSDFs (psdr_grads.g_mat), or NULL
    int sweep;                      // 1: run_interior_adjoint_sweep (Diffuse BSDFs + area lights), 0: record and probe
    int env_lds;                    // > 0: the sweeps accumulate the environment map's texel adjoints in LDS first (= 3 * width * height floats: a small map)
    float *rec_global;              // NULL: the per-lane records live in LDS; else [workgroup][word][lane of the workgroup] in global memory
                                    //   (paths too deep for 160 KB of LDS: any depth works, at global-memory latency)
};

constexpr int kMatOut = 16;         // a BSDF's row of psdr_grads.g_mat
constexpr int kMatRow = 28;         // ... and of its LDS accumulator when uv transforms are differentiated: the g_mat row, then [rot, scale, tx, ty] of its three bitmaps (g_uv_xf);
                                    // without g_uv_xf the accumulator's row is the g_mat row alone (`mrow` in the kernels, api.hip::adj LDS sizes)
constexpr int kAdjMisc = 32;        // LDS accumulators every path adds to: camera pose, environment scale and transform
// number of constant material parameters a BSDF record's flags announce (Microfacet 4 - fewer with maps -, RoughConductor 11, RoughDielectric 3)
PSDR_DEV int mat_param_count(int fl) { return (fl & 4) ? 4 : ((fl & 8) ? 11 : ((fl & 16) ? 3 : 0)); }

template <int LDS>
PSDR_DEV void adj_add(float *lds_g, float *glob, int idx, float v, bool use_lds) {
    if (v == 0.f || !finite_(v)) return;
    if (use_lds) atomicAdd(&lds_g[idx], v); else atomicAdd(&glob[idx], v);
}

// scratch: per-block LDS region behind the blob/stack: [hits: hit_words x 256][ext: ext_words x 256][lookups: lk_words x 256][accumulators]
template <int LDS>
PSDR_DEV void run_interior_adjoint(SceneView<LDS> &S, const SensorDev &cam, const AdjointParams &P, float *scratch) {
    const SceneTables &T = *S.T;
    const int mrow = S.uv_adj ? kMatRow : kMatOut;                     // LDS row of a BSDF's material adjoints (kMatRow)
    const int lane_id = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane_id) - 1ull;
    const float inv_spp = T.spp > 1 ? 1.f / (float) T.spp : 1.f;
    float *recs = P.rec_global ? P.rec_global + (size_t) blockIdx.x * (size_t) (P.hit_words + P.ext_words + P.lk_words) * kBlock : scratch;
    float *rec = recs + threadIdx.x;
    int *ext = reinterpret_cast<int *>(recs + P.hit_words * kBlock) + threadIdx.x;
    float *lk = recs + (P.hit_words + P.ext_words) * kBlock + threadIdx.x;
    float *acc_cam = P.rec_global ? scratch : scratch + (P.hit_words + P.ext_words + P.lk_words) * kBlock;        // 16 floats, always in LDS: every path adds to the same 12 entries
    float *acc_mat = acc_cam + kAdjMisc;                               // [n_bsdfs * mrow], always in LDS like the camera block
    float *acc = acc_mat + T.n_bsdfs * mrow;
    const int n_acc = P.n_hot * 22 + T.n_bsdfs * 3 + T.n_emitters * 3;
    const bool use_lds = true;                                         // colours and emitters always accumulate in LDS
    for (int i = threadIdx.x; i < n_acc; i += kBlock) acc[i] = 0.f;
    float *acc_bsdf = acc + P.n_hot * 22, *acc_emit = acc_bsdf + T.n_bsdfs * 3;
    if (threadIdx.x < kAdjMisc) acc_cam[threadIdx.x] = 0.f;    // [0..11] camera pose, [12] environment-map scale, [16..26] environment from_world, [28..31] the map's uv transform
    for (int i = threadIdx.x; i < T.n_bsdfs * mrow; i += kBlock) acc_mat[i] = 0.f;
    __syncthreads();
    S.rec = rec; S.ext = ext; S.lk = lk; S.ext_max = P.ext_words; S.lk_max = P.lk_words / 3;

    long long q_next = 0, q_end = 0;
    bool exhausted = false;
    bool have = false;
    long long lane = 0;
    for (;;) {
        // ---- phase 1: find samples whose camera ray hits the scene (70 % of the README frame is background)
        for (int round = 0; round < 8; ++round) {
            const unsigned long long need = __ballot(!have);
            if (__popcll(need) <= 6) break;
            if (q_next >= q_end && !exhausted) {
                unsigned long long base = 0;
                if (lane_id == 0) base = atomicAdd(P.counter, (unsigned long long) kFetchBatch);
                base = __shfl(base, 0);
                if ((long long) base >= P.n_local) exhausted = true;
                else { q_next = (long long) base; q_end = q_next + kFetchBatch < P.n_local ? q_next + kFetchBatch : P.n_local; }
            }
            if (q_next >= q_end) break;
            const int rank = __popcll(need & lt_mask);
            const long long item = q_next + rank;
            Vec3f o(0.f), d(0.f);
            bool cand = false;
            if (!have && item < q_end) {
                const long long chunk = (item >> 8) * P.shard_count + P.shard_rank;
                lane = P.begin + (chunk << 8) + (item & 255);
                if (lane < P.end) {
                    const long long k = T.spp > 1 ? lane / T.spp : lane;
                    const int pix = P.pix_ids ? P.pix_ids[k] : (int) k;
                    LaneRng rng;
                    rng.seed(P.seed + (P.pix_ids ? (unsigned long long) (long long) pix : (unsigned long long) lane), (unsigned long long) lane, P.skip);
                    const float bx = (float) (pix % T.width), by = (float) (pix / T.width);
                    const float jx = rng.next_1d(), jy = rng.next_1d();
                    const RayT<false> r = sample_primary_ray<false>(cam, (bx + jx) / (float) T.width, (by + jy) / (float) T.height);
                    o = r.o; d = r.d; cand = true;
                }
            }
            S.mode = 0;
            Hit h; h.slot = -1;
            if (cand) h = trace<LDS, false>(S, o, d);
            if (cand && h.slot >= 0) have = true;
            const int n_need = __popcll(need);
            q_next += n_need < (int) (q_end - q_next) ? n_need : (q_end - q_next);
        }
        if (__ballot(have) == 0ull) { if (exhausted && q_next >= q_end) break; continue; }

        // ---- phase 2: record the path, then probe
        if (have) {
            const long long k = T.spp > 1 ? lane / T.spp : lane;
            const int pix = P.pix_ids ? P.pix_ids[k] : (int) k;
            LaneRng rng;
            rng.seed(P.seed + (P.pix_ids ? (unsigned long long) (long long) pix : (unsigned long long) lane), (unsigned long long) lane, P.skip);
            const float bx = (float) (pix % T.width), by = (float) (pix / T.width);
            const float jx = rng.next_1d(), jy = rng.next_1d();
            const float sx = (bx + jx) / (float) T.width, sy = (by + jy) / (float) T.height;
            const RayT<true> ray = sample_primary_ray<true>(cam, sx, sy);
            RayT<true> ray_p = ray;                             // the ray the probes shade: `ray`, or `ray` with a camera-pose tangent
            const LaneRng rng0 = rng;
            S.mode = 1; S.rec_n = 0; S.rec_i = 0; S.ext_n = 0; S.lk_n = 0; S.probe_kind = 0;
            const Vec3d L0 = Li<true, LDS, false>(S, rng, ray, true, P.max_depth, P.hide_emitters != 0);
            const int n_hits = S.rec_n, n_ext = S.ext_n, n_lk = (has_env(LDS) && (P.g_tex != nullptr || P.g_env != nullptr || P.g_env_xf != nullptr)) ? S.lk_n : 0;
            float w[3];
            {
                const float pv[3] = {L0.x.v, L0.y.v, L0.z.v};
#pragma unroll
                for (int c = 0; c < 3; ++c) w[c] = finite_(pv[c]) ? P.w[3 * k + c] * inv_spp : 0.f;       // integrator.cpp:126 scrub
            }
            S.mode = 2;
            auto probe = [&]() -> float {
                S.rec_i = 0;
                LaneRng r = rng0;
                const Vec3d L = Li<true, LDS, false>(S, r, ray_p, true, P.max_depth, P.hide_emitters != 0);
                const float t[3] = {L.x.d, L.y.d, L.z.d};
                float g = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) if (finite_(t[c])) g += w[c] * t[c];
                return g;
            };
            auto slot_at = [&](int i) -> int { return i < n_hits ? __float_as_int(rec[4 * i * kBlock]) : ext[(i - n_hits) * kBlock]; };
            {
                // One probe loop with ONE call site of the replay: every lane walks its own list of (kind, id, component)
                // probes - triangle rows of the distinct triangles it touched (22 each), then the colours of the distinct
                // BSDFs and emitters (3 each) - and all lanes that still have a probe run it together.  (Three nested
                // loops with a replay call each inlined Li<true> four times; the kernel spent its time fetching code.)
                const int n_all = n_hits + n_ext;
                const bool wactive = (w[0] != 0.f || w[1] != 0.f || w[2] != 0.f);
                int st_stage = 0, st_i = 0, st_comp = 0, st_id = -1, st_orig = 0;
                auto advance = [&]() -> bool {
                    while (st_stage < 3) {
                        if (st_i >= n_all) { ++st_stage; st_i = 0; st_comp = 0; continue; }
                        const int slot = slot_at(st_i);
                        bool valid = slot >= 0;
                        if (valid && st_stage == 0) {
                            for (int q = 0; q < st_i; ++q) valid = valid && (slot_at(q) != slot);
                            if (valid && P.mesh_filter != nullptr) valid = P.mesh_filter[__float_as_int(S.ld(T.shade_off + 6 * slot + 1).w)] != 0;
                            if (valid) { st_id = slot; st_orig = __float_as_int(S.ld(T.shade_off + 6 * slot + 3).w); }
                            if (valid && st_comp == 9) st_comp = 18;           // the vertex normals (9-17) are covered per hit by stage 6
                        } else if (valid) {
                            const int mesh = __float_as_int(S.ld(T.shade_off + 6 * slot + 1).w);
                            const int id = st_stage == 1 ? mesh_bsdf(S, mesh) : mesh_emitter(S, mesh);
                            valid = id >= 0 && !(st_stage == 1 ? (P.skip_bsdf && P.g_mat == nullptr) : P.skip_emitter);
                            for (int q = 0; q < st_i && valid; ++q) {
                                const int sq = slot_at(q);
                                if (sq < 0) continue;
                                const int mq = __float_as_int(S.ld(T.shade_off + 6 * sq + 1).w);
                                valid = (st_stage == 1 ? mesh_bsdf(S, mq) : mesh_emitter(S, mq)) != id;
                            }
                            st_id = id;
                            if (valid && st_stage == 1) {
                                // components 0-2 the colour (g_bsdf), 3.. the constant parameters of a GGX BSDF (g_mat); a normal map is
                                // followed by the BSDF nested in it (components 32.. = the same list for the nested record)
                                const int fl = __float_as_int(S.ld(T.bsdf_off + 2 * id).w);
                                const int nested = (fl & 256) ? __float_as_int(S.ld(T.bsdf_off + 2 * id + 1).w) : -1;
                                for (;;) {
                                    const bool inner = st_comp >= 32;
                                    const int c = inner ? st_comp - 32 : st_comp, bid = inner ? nested : id;
                                    const int nm = P.g_mat != nullptr ? mat_param_count(__float_as_int(S.ld(T.bsdf_off + 2 * bid).w)) : 0;
                                    if (P.skip_bsdf && c < 3) { st_comp += 3 - c; continue; }
                                    if (c < 3 + nm) { st_id = bid; break; }
                                    if (!inner && nested >= 0) { st_comp = 32; continue; }
                                    valid = false; break;
                                }
                            }
                        }
                        if (!valid) { ++st_i; st_comp = 0; continue; }
                        return true;
                    }
                    // stage 3: the bitmap lookups the path made - components 0..2 the diffuse / reflectance map, 3..5 the
                    // specular map, 6 the roughness map of the BSDF; only the maps that exist are probed
                    // (only IN stage 3: stage 6 counts hits with the same st_i, and without the guard its second call came back from
                    // here with a lookup's id - the y / z components and every hit but the camera's lost their normal adjoints whenever
                    // the path had made lookups, i.e. with g_env / g_env_from_world / g_tex requested; found by the sweep-vs-probe
                    // cross-check of tools/sweep_check.py in round 4)
                    while (st_stage == 3 && st_i < n_lk) {
                        const int id = __float_as_int(lk[3 * st_i * kBlock]);
                        // an environment-map lookup has the three radiance components; a BSDF the maps it owns
                        const int fl = id == kEnvLookup ? (P.g_env != nullptr ? 2 : 0) : (id <= kPvLookup ? (P.g_tex != nullptr ? (2 | 32 | 64) : 0)
                                                                                                  : (P.g_tex != nullptr ? __float_as_int(S.ld(T.bsdf_off + 2 * id).w) : 0));
                        while (st_comp < 7 && !(fl & (st_comp < 3 ? 2 : (st_comp < 6 ? 32 : 64)))) st_comp = st_comp < 3 ? 3 : (st_comp < 6 ? 6 : 7);
                        if (st_comp >= 7) { ++st_i; st_comp = 0; continue; }
                        st_id = id;
                        return true;
                    }
                    // stage 4: the camera pose - entry (r, c) of rows 0-2 of to_world moves the primary ray (origin = to_world . o_cam,
                    // direction = to_world . d_cam, perspective.cpp:160-178 / orthographic.cpp:161-181)
                    if (st_stage == 3) { st_stage = 4; st_comp = P.g_cam != nullptr ? 0 : 12; }
                    if (st_stage == 4 && st_comp < 12) return true;
                    // stage 5: the environment map's from_world (the 3x3 block that turns world directions into map directions);
                    // only paths that looked the map up can depend on it
                    if (st_stage == 4) {
                        bool env_seen = false;
                        if (P.g_env_xf != nullptr) for (int q = 0; q < n_lk; ++q) env_seen = env_seen || __float_as_int(lk[3 * q * kBlock]) == kEnvLookup;
                        st_stage = 5; st_comp = env_seen ? 0 : 11;
                    }
                    if (st_stage == 5) {
                        while (st_comp < 11 && (st_comp & 3) == 3) ++st_comp;      // entries 0,1,2, 4,5,6, 8,9,10
                        if (st_comp < 11) return true;
                        st_stage = 6; st_i = 0; st_comp = 0;
                    }
                    // stage 6: the blended shading normal of every traced hit on a wanted, smooth-shaded mesh (3 components per hit)
                    while (st_i < n_hits) {
                        const int slot = __float_as_int(rec[4 * st_i * kBlock]);
                        bool valid = slot >= 0;
                        if (valid) {
                            const float4 s1 = S.ld(T.shade_off + 6 * slot + 1), s2 = S.ld(T.shade_off + 6 * slot + 2);
                            valid = (__float_as_int(s2.w) & 1) == 0;                                              // flat meshes ignore vertex normals
                            if (valid && P.mesh_filter != nullptr) valid = P.mesh_filter[__float_as_int(s1.w)] != 0;
                        }
                        if (!valid) { ++st_i; continue; }
                        st_id = st_i; st_orig = __float_as_int(S.ld(T.shade_off + 6 * slot + 3).w);
                        return true;
                    }
                    return false;
                };
                bool more = wactive && advance();
                while (__ballot(more) != 0ull) {
                    if (more) {
                        S.probe_kind = st_stage < 3 ? st_stage + 1 : (st_stage == 3 ? 5 : (st_stage == 4 ? 4 : (st_stage == 5 ? 7 : 8))); S.probe_id = st_id; S.probe_comp = st_comp;
                        const int bc = (st_stage == 1 && st_comp >= 32) ? st_comp - 32 : st_comp;      // component within the (own or nested) BSDF
                        if (st_stage == 1) { S.probe_kind = bc >= 3 ? 6 : 2; S.probe_comp = bc >= 3 ? bc - 3 : bc; }
                        if (st_stage == 3) { S.probe_u = lk[(3 * st_i + 1) * kBlock]; S.probe_v = lk[(3 * st_i + 2) * kBlock]; }
                        if (st_stage == 4) { ray_p = ray; primary_ray_pose_tangent(cam, sx, sy, st_comp, ray_p); }
                        const float gval = probe();
                        if (st_stage == 0) {
                            const int hot = P.hot_map[st_orig];
                            if (hot >= 0 && hot < P.n_hot) adj_add<LDS>(acc, acc, hot * 22 + st_comp, gval, true);      // (the launch may use fewer hot slots than the scene has)
                            else adj_add<LDS>(acc, P.g_tri, st_orig * 22 + st_comp, gval, false);
                        }
                        else if (st_stage == 1 && bc >= 3) adj_add<LDS>(acc_mat, acc_mat, st_id * mrow + bc - 3, gval, true);
                        else if (st_stage == 1) adj_add<LDS>(acc_bsdf, P.g_bsdf, st_id * 3 + bc, gval, use_lds);
                        else if (st_stage == 2) adj_add<LDS>(acc_emit, P.g_emitter, st_id * 3 + st_comp, gval, use_lds);
                        else if (st_stage == 4) { adj_add<LDS>(acc_cam, acc_cam, st_comp, gval, true); ray_p = ray; }
                        else if (st_stage == 5) adj_add<LDS>(acc_cam, acc_cam, 16 + st_comp, gval, true);
                        else if (st_stage == 6) {
                            // transpose of the blend n0 (1 - u - v) + n1 u + n2 v at this hit's barycentrics
                            const float bu = rec[(4 * st_i + 1) * kBlock], bv = rec[(4 * st_i + 2) * kBlock];
                            const float wv[3] = {1.f - bu - bv, bu, bv};
                            const int hot = P.hot_map[st_orig];
                            for (int k = 0; k < 3; ++k) {
                                const int comp = 9 + 3 * k + st_comp;
                                if (hot >= 0 && hot < P.n_hot) adj_add<LDS>(acc, acc, hot * 22 + comp, gval * wv[k], true);
                                else adj_add<LDS>(acc, P.g_tri, st_orig * 22 + comp, gval * wv[k], false);
                            }
                        }
                        else if constexpr (has_env(LDS)) {
                            // scatter over the footprint of the lookup (the transpose of the bilinear interpolation)
                            int idx[4]; float wt[4];
                            if (st_id == kEnvLookup) {
                                // gval = d L / d rgb[c] of this lookup (before the scale): texels by the footprint, and
                                // d L / d scale = sum_c gval_c . rgb[c] / scale
                                const EnvDev &E = T.env;
                                env::bitmap_footprint_env(E.width, E.height, S.probe_u, S.probe_v, idx, wt, env::UvXf<float>(E.xf));
                                if (gval != 0.f && finite_(gval)) {
                                    float rgb_c = 0.f;
                                    for (int k = 0; k < 4; ++k) {
                                        atomicAdd(&P.g_env[3ll * idx[k] + st_comp], gval * wt[k]);
                                        rgb_c += wt[k] * E.radiance[3ll * idx[k] + st_comp];
                                    }
                                    if (P.g_env_scale != nullptr && E.scale != 0.f) atomicAdd(&acc_cam[12], gval * rgb_c / E.scale);
                                    if (P.g_uv_xf != nullptr) { float ob[3] = {0.f, 0.f, 0.f}; ob[st_comp] = gval; env_xf_adjoint(E, S.probe_u, S.probe_v, ob, &acc_cam[28]); }
                                }
                            } else if (!has_mat(LDS)) {
                            } else if (st_id <= kPvLookup) {
                                // per-vertex values: the transpose of the barycentric interpolation over the triangle's three vertices
                                const int slot = kPvLookup - st_id;
                                const int bid = mesh_bsdf(S, __float_as_int(S.ld(T.shade_off + 6 * slot + 1).w));
                                const PvDev pv = T.pv[bid];
                                const int *fi = T.tri_fi + 3 * slot;
                                const int tslot = st_comp < 3 ? 0 : (st_comp < 6 ? 1 : 2), ch = tslot == 2 ? 1 : 3, c = st_comp - 3 * tslot;
                                const float wv[3] = {1.f - S.probe_u - S.probe_v, S.probe_u, S.probe_v};
                                if (gval != 0.f && finite_(gval))
                                    for (int k = 0; k < 3; ++k) atomicAdd(&P.g_tex[pv.g_off[tslot] + (long long) ch * fi[k] + c], gval * wv[k]);
                            } else {
                                const int tslot = st_comp < 3 ? 0 : (st_comp < 6 ? 1 : 2), ch = tslot == 2 ? 1 : 3, c = st_comp - 3 * tslot;
                                const TexDev td = T.tex[3 * st_id + tslot];
                                env::bitmap_footprint(td.w, td.h, S.probe_u, S.probe_v, true, idx, wt, env::UvXf<float>(td.xf));
                                if (gval != 0.f && finite_(gval)) {
                                    for (int k = 0; k < 4; ++k) atomicAdd(&P.g_tex[td.g_off + (long long) ch * idx[k] + c], gval * wt[k]);
                                    if (P.g_uv_xf != nullptr) {
                                        float ob[3] = {0.f, 0.f, 0.f}; ob[c] = gval;
                                        if (ch == 3) tex_xf_adjoint<3>(td, S.probe_u, S.probe_v, ob, &acc_mat[st_id * mrow + kMatOut + 4 * tslot]);
                                        else tex_xf_adjoint<1>(td, S.probe_u, S.probe_v, ob, &acc_mat[st_id * mrow + kMatOut + 4 * tslot]);
                                    }
                                }
                            }
                        }
                        ++st_comp;
                        if (st_stage < 4 && st_stage != 1 && st_comp >= (st_stage == 0 ? 22 : (st_stage == 3 ? 7 : 3))) { st_comp = 0; ++st_i; }      // (stage 1 ends in advance())
                        if (st_stage == 6 && st_comp >= 3) { st_comp = 0; ++st_i; }
                        more = advance();
                    }
                }
            }
            S.mode = 0; S.probe_kind = 0;
            have = false;
        }
    }
    __syncthreads();
    if (P.g_cam != nullptr && threadIdx.x < 12 && acc_cam[threadIdx.x] != 0.f) atomicAdd(&P.g_cam[threadIdx.x], acc_cam[threadIdx.x]);
    if (P.g_env_scale != nullptr && threadIdx.x == 12 && acc_cam[12] != 0.f) atomicAdd(P.g_env_scale, acc_cam[12]);
    if (P.g_env_xf != nullptr && threadIdx.x >= 16 && threadIdx.x < 27 && acc_cam[threadIdx.x] != 0.f) atomicAdd(&P.g_env_xf[threadIdx.x - 16], acc_cam[threadIdx.x]);
    if (P.g_mat != nullptr)
        for (int i = threadIdx.x; i < T.n_bsdfs * kMatOut; i += kBlock) { const float v = acc_mat[(i / kMatOut) * mrow + i % kMatOut]; if (v != 0.f) atomicAdd(&P.g_mat[i], v); }
    if (P.g_uv_xf != nullptr) {          // uv transforms: three bitmaps per BSDF, then the environment map's
        for (int i = threadIdx.x; i < T.n_bsdfs * 12; i += kBlock) { const float v = acc_mat[(i / 12) * mrow + kMatOut + i % 12]; if (v != 0.f) atomicAdd(&P.g_uv_xf[i], v); }
        if (threadIdx.x >= 28 && threadIdx.x < 32 && acc_cam[threadIdx.x] != 0.f) atomicAdd(&P.g_uv_xf[12 * T.n_bsdfs + threadIdx.x - 28], acc_cam[threadIdx.x]);
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < P.n_hot * 22; i += kBlock) if (acc[i] != 0.f) atomicAdd(&P.g_tri[P.hot_inv[i / 22] * 22 + i % 22], acc[i]);
        for (int i = threadIdx.x; i < T.n_bsdfs * 3; i += kBlock) if (acc_bsdf[i] != 0.f) atomicAdd(&P.g_bsdf[i], acc_bsdf[i]);
        for (int i = threadIdx.x; i < T.n_emitters * 3; i += kBlock) if (acc_emit[i] != 0.f) atomicAdd(&P.g_emitter[i], acc_emit[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Reverse sweep of the interior term for scenes of Diffuse BSDFs and area lights (scene classes 1 and 2 without an environment
// map, PathTracer) - what `drjit.backward` does through integrator.cpp:51-100 + path.cpp:35-127 - in ONE pass back over the
// path instead of one replay per touched scene quantity.
//
// In D mode the radiance of a path is
//     L = Le_0 + sum_k [ thr_k rho_k Le_hk cN_k s_Nk  +  thr_{k+1} Le_{k+1} w2_k ],     thr_{k+1} = thr_k rho_k cf_k s_fk,  thr_0 = 1
// with rho_k the reflectance at vertex k, Le the radiance of the emitter that was hit, cN / cf / w2 DETACHED pdfs and MIS
// weights, and both s_Nk (next-event term) and s_fk (BSDF-sampling term) of one geometric form
//     s(x, ns; z, nz, A_z) = (w . ns) |nz . w| / r^2 . (A_z / detach(A_z)),     w = (z - x) / r,  r = |z - x|
// x the shading point with shading normal ns; z the light sample (nz = geometric normal of the triangle the shadow ray hits)
// resp. the next vertex.  Every vertex is glued to its triangle at fixed barycentrics (x = p0 + u e1 + v e2) except the camera
// hit, which slides along the fixed ray (x_0 = o + t d, (u, v, t) the differentiable Moeller-Trumbore solution).
// Pass 1 walks the path forward (the primal arithmetic of D mode) and keeps (slot, u, v) per vertex and, per bounce, the light
// sample, the constants and thr_k; pass 2 walks back with A = d(w . L) / d thr_{k+1}, and emits per bounce the adjoints of
// x_k, ns_k, z, nz, A_z in closed form (`seg_eval`), of rho_k and of the emitter radiance, scattered into the 22-float
// triangle rows [p0 e1 e2 n0 n1 n2 fn area] and the colour rows that the host chains to its leaves (chain.py).
// Cost: two shading passes and one traversal per path, whatever the number of leaves - the probe form replays the path once
// per (touched triangle x 13 + 3) components.  The forward-mode kernels stay the checker: <w, J v> = <J^T w, v>
// (tests/test_gpu_adjoint.py).
struct SegGrad { Vec3f dx, dns, dz, dnz; float dA; };      // d s / d (x, ns, z, nz, A_z)

// s and its gradient; sgn = -1 when a two-sided BSDF is seen from its back
PSDR_DEV float seg_eval(const Vec3f &x, const Vec3f &ns, float sgn, const Vec3f &z, const Vec3f &nz, float area_z, SegGrad &g) {
    const Vec3f v = z - x;
    const float r2 = dot(v, v);
    g.dx = g.dns = g.dz = g.dnz = Vec3f(0.f); g.dA = 0.f;
    if (!(r2 > 0.f)) return 0.f;
    const float ir2 = 1.f / r2, k = ir2 * ir2;
    const float Pn = dot(v, ns) * sgn, Qs = dot(nz, v), sq = Qs < 0.f ? -1.f : 1.f, Q = Qs * sq;
    const float s = Pn * Q * k;                                          // (w.ns) |nz.w| / r^2 = (v.ns)(|nz.v|) / r^4
    g.dns = v * (sgn * Q * k);
    g.dnz = v * (sq * Pn * k);
    g.dz = (ns * (sgn * Q) + nz * (sq * Pn)) * k - v * (4.f * s * ir2);
    g.dx = -g.dz;
    g.dA = area_z > 0.f ? s / area_z : 0.f;
    return s;
}

// adjoint of the Moeller-Trumbore solve (ray_tri_uvt): (ub, vb, tb) -> the triangle and the ray
PSDR_DEV void mt_adjoint(const Vec3f &p0, const Vec3f &e1, const Vec3f &e2, const Vec3f &o, const Vec3f &d, float ub, float vb, float tb,
                         Vec3f &p0b, Vec3f &e1b, Vec3f &e2b, Vec3f &ob, Vec3f &db) {
    const Vec3f h = cross(d, e2);
    const float a = dot(e1, h), f = 1.f / a;
    const Vec3f s = o - p0, q = cross(s, e1);
    const float U = dot(s, h), V = dot(d, q), Wq = dot(e2, q);
    const float fb = ub * U + vb * V + tb * Wq, Ub = f * ub, Vb = f * vb, Wb = f * tb, ab = -f * f * fb;
    Vec3f sb = h * Ub, hb = s * Ub;
    db = q * Vb;
    Vec3f qb = d * Vb + e2 * Wb;
    e2b = q * Wb;
    e1b = h * ab; hb = hb + e1 * ab;
    sb = sb + cross(e1, qb); e1b = e1b + cross(qb, s);                   // q = s x e1
    db = db + cross(e2, hb); e2b = e2b + cross(hb, d);                   // h = d x e2
    ob = sb; p0b = -sb;
}

// geometry of one triangle slot at barycentrics (u, v)
struct VtxGeom {
    Vec3f p0, e1, e2, n0, n1, n2, fn, nb, ns, x;
    float area, nbl, u, v;
    int orig, mesh;
    bool flat;
};
template <int LDS> PSDR_DEV VtxGeom load_vertex(const SceneView<LDS> &S, int slot, float u, float v) {
    VtxGeom g;
    load_geom<false, LDS>(S, slot, g.p0, g.e1, g.e2);
    const int w = S.T->shade_off + 6 * slot;
    const float4 s0 = S.ld(w), s1 = S.ld(w + 1), s2 = S.ld(w + 2), s3 = S.ld(w + 3);
    g.n0 = Vec3f(s0.x, s0.y, s0.z); g.n1 = Vec3f(s1.x, s1.y, s1.z); g.n2 = Vec3f(s2.x, s2.y, s2.z); g.fn = Vec3f(s3.x, s3.y, s3.z);
    g.area = s0.w; g.mesh = __float_as_int(s1.w); g.flat = (__float_as_int(s2.w) & 1) != 0; g.orig = __float_as_int(s3.w);
    g.u = u; g.v = v;
    g.x = madd3(g.e1, u, g.e2, v, g.p0);
    g.nb = madd3(g.n1 - g.n0, u, g.n2 - g.n0, v, g.n0);
    g.nbl = norm(g.nb);
    g.ns = g.flat ? g.fn : g.nb / g.nbl;
    return g;
}

template <int LDS>
PSDR_DEV void run_interior_adjoint_sweep(SceneView<LDS> &S, const SensorDev &cam, const AdjointParams &P, float *scratch) {
    const SceneTables &T = *S.T;
    const int mrow = S.uv_adj ? kMatRow : kMatOut;                     // LDS row of a BSDF's material adjoints (kMatRow)
    const int lane_id = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane_id) - 1ull;
    const float inv_spp = T.spp > 1 ? 1.f / (float) T.spp : 1.f;
    const int D = P.max_depth;
    const int lane_words = P.hit_words + P.ext_words + P.lk_words;        // sized by the launch: >= adj_sweep_words(D)
    float *vrec = (P.rec_global ? P.rec_global + (size_t) blockIdx.x * (size_t) lane_words * kBlock : scratch) + threadIdx.x;    // [3 * (D + 1)] slot, u, v per vertex, stride kBlock
    float *brec = vrec + 3 * (D + 1) * kBlock;                            // [11 * D] per bounce: 0 light slot, 1-2 its barycentrics, 3 shadow-hit slot,
                                                                          //   4 cN, 5 cf, 6 w2, 7 flags, 8-10 thr_k
    float *acc_cam = P.rec_global ? scratch : scratch + lane_words * kBlock;      // same accumulator layout as run_interior_adjoint
    float *acc_mat = acc_cam + kAdjMisc;
    float *acc = acc_mat + T.n_bsdfs * mrow;
    const int n_acc = P.n_hot * 22 + T.n_bsdfs * 3 + T.n_emitters * 3 + P.env_lds;
#ifdef PSDR_LDS_POISON
    // (diagnostic) the per-lane records start as garbage: a record word that is read before this path wrote it shows up in the result
    if (!P.rec_global) { for (int i = threadIdx.x; i < lane_words * kBlock; i += kBlock) scratch[i] = PSDR_LDS_POISON; __syncthreads(); }
#endif
    for (int i = threadIdx.x; i < n_acc; i += kBlock) acc[i] = 0.f;
    float *acc_bsdf = acc + P.n_hot * 22, *acc_emit = acc_bsdf + T.n_bsdfs * 3;
    float *acc_env = acc_emit + T.n_emitters * 3;        // [env_lds] texel adjoints of a small environment map (every sample of a wave hits the same few texels)
    if (threadIdx.x < kAdjMisc) acc_cam[threadIdx.x] = 0.f;
    __syncthreads();
    S.mode = 0; S.probe_kind = 0;

    auto add_row = [&](const VtxGeom &g, int comp, float val) {
        if (val == 0.f || !finite_(val)) return;
        const int hot = P.hot_map[g.orig];
        if (hot >= 0 && hot < P.n_hot) atomicAdd(&acc[hot * 22 + comp], val); else atomicAdd(&P.g_tri[g.orig * 22 + comp], val);
    };
    auto add_vec = [&](const VtxGeom &g, int comp, const Vec3f &val) { add_row(g, comp, val.x); add_row(g, comp + 1, val.y); add_row(g, comp + 2, val.z); };
    auto wanted = [&](const VtxGeom &g) { return P.mesh_filter == nullptr || P.mesh_filter[g.mesh] != 0; };
    auto add_rgb = [&](float *tab, int id, const Vec3f &val) {
        if (val.x != 0.f && finite_(val.x)) atomicAdd(&tab[3 * id], val.x);
        if (val.y != 0.f && finite_(val.y)) atomicAdd(&tab[3 * id + 1], val.y);
        if (val.z != 0.f && finite_(val.z)) atomicAdd(&tab[3 * id + 2], val.z);
    };
    // Environment map (class 2): radiance along a world direction, and what the adjoint Lb = d (w.L) / d Le of one lookup gives -
    // the four texels of its footprint, the scale, from_world - and, returned, d (w.L) / d dir.  The Jacobian with respect to the
    // local direction comes from three forward evaluations of the lookup with unit tangents (atan2 / acos / bilinear: shade.h).
    auto env_radiance = [&](const Vec3f &dir) -> Vec3f {
        if constexpr (has_env(LDS)) return env_eval_direction<false, LDS>(S, T.env, dir);
        else return Vec3f(0.f);
    };
    // the direction an environment lookup uses for a ray from x that ended on the bounding cube at (slot, u, v): the hit point's local
    // incident direction taken back to the world through the cube face's frame (intersection.h / envmap.cpp:47-56) - bit for bit what
    // the forward pass looks up, so that both passes land in the same texel cell of the (piecewise bilinear) map
    auto env_dir_at = [&](int slot, float u, float v, const Vec3f &x) -> Vec3f {
        Hit h; h.slot = slot; h.u = u; h.v = v; h.t = 0.f;
        RayT<false> r; r.o = x; r.d = Vec3f(0.f, 0.f, 1.f);
        const Its<false> i1 = make_its<false, LDS, true>(S, h, r, true);
        return -to_world<false>(i1, i1.wi);
    };
    // the camera ray's own lookup (every background pixel makes one, and the samples of a wave share texels) is scattered by the whole
    // wave after the path work: one atomic per distinct texel and wave instead of one per lane
    bool pend_env = false;
    int pe_idx[4] = {0, 0, 0, 0};
    float pe_val[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto env_adjoint = [&](const Vec3f &dir, const Vec3f &Lb, bool defer = false) -> Vec3f {
        Vec3f dirb(0.f);
        if constexpr (has_env(LDS)) {
            const EnvDev &E = T.env;
            if (!(finite_(Lb.x) && finite_(Lb.y) && finite_(Lb.z)) || (Lb.x == 0.f && Lb.y == 0.f && Lb.z == 0.f)) return dirb;
            const Vec3f v = xform_dir(E.from_world, dir);
            float vb[3], uu = 0.f, ww = 0.f, rgb0[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const VecN<true> vd(Dual(v.x, j == 0 ? 1.f : 0.f), Dual(v.y, j == 1 ? 1.f : 0.f), Dual(v.z, j == 2 ? 1.f : 0.f));
                Dual u = env_atan2(vd.x, -vd.z) * Dual(env::kInvTwoPi), w = env_safe_acos(vd.y) * Dual(env::kInvPi);
                u = u - env_floor(u); w = w - env_floor(w);
                Dual rgb[3];
                env::bitmap_eval_fn<Dual>([&](int i, int c) { return Dual(E.radiance[3 * i + c], 0.f); }, E.width, E.height, u, w, rgb, uv_xf_d(E.xf, E.xf, false));
                vb[j] = E.scale * (Lb.x * rgb[0].d + Lb.y * rgb[1].d + Lb.z * rgb[2].d);
                if (j == 0) { uu = u.v; ww = w.v; rgb0[0] = rgb[0].v; rgb0[1] = rgb[1].v; rgb0[2] = rgb[2].v; }
            }
            const float lb[3] = {Lb.x, Lb.y, Lb.z}, dv[3] = {dir.x, dir.y, dir.z};
            if (P.g_env != nullptr) {
                int idx[4]; float wt[4];
                env::bitmap_footprint_env(E.width, E.height, uu, ww, idx, wt, env::UvXf<float>(E.xf));
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (lb[c] != 0.f) for (int k = 0; k < 4; ++k) {
                        if (P.env_lds) atomicAdd(&acc_env[3 * idx[k] + c], lb[c] * E.scale * wt[k]);
                        else if (defer) { pend_env = true; pe_idx[k] = idx[k]; pe_val[3 * k + c] = lb[c] * E.scale * wt[k]; }
                        else atomicAdd(&P.g_env[3ll * idx[k] + c], lb[c] * E.scale * wt[k]);
                    }
            }
            if (P.g_env_scale != nullptr) { const float sb = lb[0] * rgb0[0] + lb[1] * rgb0[1] + lb[2] * rgb0[2]; if (sb != 0.f && finite_(sb)) atomicAdd(&acc_cam[12], sb); }
            if (P.g_uv_xf != nullptr) { const float ob[3] = {lb[0] * E.scale, lb[1] * E.scale, lb[2] * E.scale}; env_xf_adjoint(E, uu, ww, ob, &acc_cam[28]); }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (!finite_(vb[r])) vb[r] = 0.f;
                if (P.g_env_xf != nullptr)
                    for (int c = 0; c < 3; ++c) { const float val = vb[r] * dv[c]; if (val != 0.f) atomicAdd(&acc_cam[16 + 4 * r + c], val); }
            }
            dirb = Vec3f(E.from_world.m[0] * vb[0] + E.from_world.m[4] * vb[1] + E.from_world.m[8] * vb[2],
                         E.from_world.m[1] * vb[0] + E.from_world.m[5] * vb[1] + E.from_world.m[9] * vb[2],
                         E.from_world.m[2] * vb[0] + E.from_world.m[6] * vb[1] + E.from_world.m[10] * vb[2]);
        }
        return dirb;
    };
    const int env_id = has_env(LDS) ? T.env_emitter : -1;
    // adjoint of x through dir = (z - x) / |z - x| (z fixed)
    auto dir_to_x = [&](const Vec3f &x, const Vec3f &z, const Vec3f &dirb) -> Vec3f {
        const Vec3f v = z - x;
        const float r = norm(v);
        if (!(r > 0.f)) return Vec3f(0.f);
        const Vec3f d = v / r;
        return (d * dot(d, dirb) - dirb) / r;
    };
    // the normal blend n0 (1 - u - v) + n1 u + n2 v behind a shading normal: its adjoint from the adjoint of ns
    auto blend_adjoint = [&](const VtxGeom &g, const Vec3f &nsb) { return (nsb - g.ns * dot(g.ns, nsb)) / g.nbl; };
    // adjoints of a vertex glued to its triangle: position, shading normal, geometric normal, area
    auto emit_glued = [&](const VtxGeom &g, const Vec3f &xb, const Vec3f &nsb, const Vec3f &ngb, float ab) {
        if (!wanted(g)) return;
        add_vec(g, 0, xb); add_vec(g, 3, xb * g.u); add_vec(g, 6, xb * g.v);
        Vec3f fnb = ngb;
        if (g.flat) fnb = fnb + nsb;
        else {
            const Vec3f nbb = blend_adjoint(g, nsb);
            add_vec(g, 9, nbb * (1.f - g.u - g.v)); add_vec(g, 12, nbb * g.u); add_vec(g, 15, nbb * g.v);
        }
        add_vec(g, 18, fnb);
        add_row(g, 21, ab);
    };

    long long q_next = 0, q_end = 0;
    bool exhausted = false;
    bool have = false;
    long long lane = 0;
    for (;;) {
        // ---- phase 1: find samples whose camera ray hits the scene (as in run_interior_adjoint)
        for (int round = 0; round < 8; ++round) {
            const unsigned long long need = __ballot(!have);
            if (__popcll(need) <= 6) break;
            if (q_next >= q_end && !exhausted) {
                unsigned long long base = 0;
                if (lane_id == 0) base = atomicAdd(P.counter, (unsigned long long) kFetchBatch);
                base = __shfl(base, 0);
                if ((long long) base >= P.n_local) exhausted = true;
                else { q_next = (long long) base; q_end = q_next + kFetchBatch < P.n_local ? q_next + kFetchBatch : P.n_local; }
            }
            if (q_next >= q_end) break;
            const int rank = __popcll(need & lt_mask);
            const long long item = q_next + rank;
            Vec3f o(0.f), d(0.f);
            bool cand = false;
            if (!have && item < q_end) {
                const long long chunk = (item >> 8) * P.shard_count + P.shard_rank;
                lane = P.begin + (chunk << 8) + (item & 255);
                if (lane < P.end) {
                    const long long k = T.spp > 1 ? lane / T.spp : lane;
                    const int pix = P.pix_ids ? P.pix_ids[k] : (int) k;
                    LaneRng rng;
                    rng.seed(P.seed + (P.pix_ids ? (unsigned long long) (long long) pix : (unsigned long long) lane), (unsigned long long) lane, P.skip);
                    const float bx = (float) (pix % T.width), by = (float) (pix / T.width);
                    const float jx = rng.next_1d(), jy = rng.next_1d();
                    const RayT<false> r = sample_primary_ray<false>(cam, (bx + jx) / (float) T.width, (by + jy) / (float) T.height);
                    o = r.o; d = r.d; cand = true;
                }
            }
            Hit h; h.slot = -1;
            if (cand) h = trace<LDS, false>(S, o, d);
            if (cand && h.slot >= 0) { have = true; vrec[0] = __int_as_float(h.slot); }
            const int n_need = __popcll(need);
            q_next += n_need < (int) (q_end - q_next) ? n_need : (q_end - q_next);
        }
        if (__ballot(have) == 0ull) { if (exhausted && q_next >= q_end) break; continue; }

        if (have) {
            const long long kpix = T.spp > 1 ? lane / T.spp : lane;
            const int pix = P.pix_ids ? P.pix_ids[kpix] : (int) kpix;
            LaneRng rng;
            rng.seed(P.seed + (P.pix_ids ? (unsigned long long) (long long) pix : (unsigned long long) lane), (unsigned long long) lane, P.skip);
            const float bx = (float) (pix % T.width), by = (float) (pix / T.width);
            const float jx = rng.next_1d(), jy = rng.next_1d();
            const float sx = (bx + jx) / (float) T.width, sy = (by + jy) / (float) T.height;
            const RayT<false> ray = sample_primary_ray<false>(cam, sx, sy);
            float wgt[3] = {P.w[3 * kpix] * inv_spp, P.w[3 * kpix + 1] * inv_spp, P.w[3 * kpix + 2] * inv_spp};

            // ------------------------------------------------------------ pass 1: forward, the primal arithmetic of D mode
            const int slot0 = __float_as_int(vrec[0]);
            float u0, v0, t0;
            {
                Vec3f a0, b0, c0;
                load_geom<false, LDS>(S, slot0, a0, b0, c0);
                ray_tri_uvt<float>(a0, b0, c0, ray.o, ray.d, u0, v0, t0);
            }
            vrec[kBlock] = u0; vrec[2 * kBlock] = v0;
            const Vec3f x0(fmaf(ray.d.x, t0, ray.o.x), fmaf(ray.d.y, t0, ray.o.y), fmaf(ray.d.z, t0, ray.o.z));     // the camera hit slides along the ray
            Hit hh; hh.slot = slot0; hh.u = u0; hh.v = v0; hh.t = t0;
            Its<false> its = make_its<false, LDS, true>(S, hh, ray, false);
            its.p = x0; its.t = t0;
            its.wi = to_local<false>(its, -ray.d);
            Vec3f thr(1.f), Lsum(0.f);
            const int e0 = mesh_emitter(S, its.mesh);
            const bool le0 = !P.hide_emitters && e0 >= 0 && (e0 == env_id || its.wi.z > 0.f);
            const Vec3f dir0 = -to_world<false>(its, its.wi);          // (= ray.d through the frame of the first hit, as eval_Le rebuilds it)
            if (le0) { if (e0 == env_id) Lsum = env_radiance(dir0); else { const float4 a = S.ld(T.emit_off + 2 * e0); Lsum = Vec3f(a.x, a.y, a.z); } }
            int nb = 0;                                                   // bounces recorded
            bool active = true;
            for (int depth = 0; depth < D && active; ++depth) {
                float *br = brec + 11 * depth * kBlock;
                br[8 * kBlock] = thr.x; br[9 * kBlock] = thr.y; br[10 * kBlock] = thr.z;
                nb = depth + 1;
                int flags = 0;              // 1 next-event term, 2 BSDF term, 4 back side of a two-sided BSDF, 8 the next vertex is a visible emitter
                const int bid = mesh_bsdf(S, its.mesh);
                const float4 ba = bid >= 0 ? S.ld(T.bsdf_off + 2 * bid) : make_float4(0.f, 0.f, 0.f, 0.f);
                const Vec3f rho(ba.x, ba.y, ba.z);
                const bool two = bid >= 0 && (__float_as_int(ba.w) & 1) != 0;
                const float sgn = (two && its.wi.z < 0.f) ? -1.f : 1.f;
                if (sgn < 0.f) flags |= 4;
                const bool front = bid >= 0 && (two ? fabsf(its.wi.z) : its.wi.z) > 0.f;
                // The vertex's two rays - the shadow ray of the emitter sample (path.cpp:47-83) and the extension ray of the BSDF sample (path.cpp:86-123) - are drawn
                // first (the sampler order is the forward pass's: two numbers, then three) and traced by ONE trace2 call, as run_paths does: on a BVH scene one post + one
                // run of the wave's queue with twice the rays in flight instead of two runs to completion (round 4: config 5's interior adjoint 51.5 -> 45.5 ms, C3's 2.2 -> 2.1 ms)
                // (DirectIntegrator(1) neither draws nor uses the emitter sample, DirectIntegrator(0) stops after it: direct.cpp:34-132)
                const float sn1 = P.mis != 1 ? rng.next_1d() : 0.f, sn2 = P.mis != 1 ? rng.next_1d() : 0.f;
                const bool do_nee = P.mis != 1 && mesh_emitter(S, its.mesh) < 0;
                PositionSample<false> ps;
                ps.p = Vec3f(0.f); ps.n = Vec3f(0.f); ps.J = 1.f; ps.pdf = 1.f; ps.slot = -1; ps.ba = ps.bb = 0.f;
                Vec3f wod(0.f);
                float dist_sqr = 0.f, dist = 0.f;
                if (do_nee) {
                    ps = sample_emitter_position<false, LDS>(S, its.p, sn1, sn2);
                    wod = ps.p - its.p;
                    dist_sqr = squared_norm(wod); dist = safe_sqrt(dist_sqr);
                    wod = wod / dist;
                }
                const bool do_bsdf = P.mis != 0;
                const float sb0 = do_bsdf ? rng.next_1d() : 0.f, sb1 = do_bsdf ? rng.next_1d() : 0.f, sb2 = do_bsdf ? rng.next_1d() : 0.f;
                BSDFSample bs = bsdf_sample<false, LDS>(S, its, sb0, sb1, sb2, do_bsdf);
                if (!do_bsdf) bs.valid = false;
                RayT<false> curr; curr.o = its.p; curr.d = to_world<false>(its, bs.wo);
                Hit h1, hx;
                trace2<LDS, false>(S, its.p, wod, do_nee, curr.o, curr.d, bs.valid, h1, hx, do_nee ? (dist - kShadowEpsilon) * 0.9999f : -__builtin_inff());     // (any occluder settles the shadow test)
                if (do_nee && h1.slot >= 0) {   // next-event estimation
                    RayT<false> ray1; ray1.o = its.p; ray1.d = wod;
                    const Its<false> its1 = make_its<false, LDS, false>(S, h1, ray1, true);
                    const int eh = mesh_emitter(S, its1.mesh);
                    if (its1.t > dist - kShadowEpsilon && eh >= 0) {
                        const float G = fabsf(dot(its1.n, -wod)) / dist_sqr;
                        const float woz = dot(wod, its.fn) * sgn;
                        const float pdf1 = ((front && woz > 0.f) ? kInvPi * woz : 0.f) * G;
                        if (front && woz > 0.f && pdf1 != 0.f) {
                            const float cN = kInvPi * (P.mis == 0 ? 1.f : mis_weight(ps.pdf, pdf1)) / ps.pdf;
                            if (eh == env_id) {
                                // the sample lies on the scene box (a fixed point: the record keeps it instead of a triangle),
                                // the radiance is looked up along the shadow ray
                                Lsum = Lsum + thr * rho * env_radiance(env_dir_at(h1.slot, h1.u, h1.v, its.p)) * (woz * G * cN);
                                br[0] = h1.u; br[kBlock] = h1.v; br[2 * kBlock] = 0.f; br[3 * kBlock] = __int_as_float(h1.slot);
                                br[4 * kBlock] = cN;
                                flags |= 1 | 16;
                            } else if (its1.wi.z > 0.f) {
                                const float4 ea = S.ld(T.emit_off + 2 * eh);
                                Lsum = Lsum + thr * rho * Vec3f(ea.x, ea.y, ea.z) * (woz * G * cN);
                                br[0] = __int_as_float(ps.slot); br[kBlock] = ps.ba; br[2 * kBlock] = ps.bb; br[3 * kBlock] = __int_as_float(h1.slot);
                                br[4 * kBlock] = cN;
                                flags |= 1;
                            }
                        }
                    }
                }
                {   // BSDF sampling
                    active = bs.valid && hx.slot >= 0;
                    if (active) {
                        const Its<false> itx = make_its<false, LDS, true>(S, hx, curr, true);
                        float *vr = vrec + 3 * (depth + 1) * kBlock;
                        vr[0] = __int_as_float(hx.slot); vr[kBlock] = hx.u; vr[2 * kBlock] = hx.v;
                        const Vec3f wo = (itx.p - its.p) / itx.t;
                        const float G = fabsf(dot(itx.n, -wo)) / sqr(itx.t);
                        const float pdf0 = bs.pdf * G;
                        const float woz = dot(wo, its.fn) * sgn;
                        const bool ok = !(itx.t < kEpsilon) && front && woz > 0.f;
                        const float cf = ok ? kInvPi / pdf0 : 0.f;
                        const float w2 = P.mis == 1 ? 1.f : mis_weight(pdf0, emitter_position_pdf<false, LDS>(S, its.p, itx));
                        thr = ok ? thr * rho * (woz * G * cf) : Vec3f(0.f);
                        const int ex = mesh_emitter(S, itx.mesh);
                        if (ex >= 0 && ex == env_id) { Lsum = Lsum + env_radiance(env_dir_at(hx.slot, hx.u, hx.v, its.p)) * thr * w2; flags |= 8; }
                        else if (ex >= 0 && itx.wi.z > 0.f) { const float4 ea = S.ld(T.emit_off + 2 * ex); Lsum = Lsum + Vec3f(ea.x, ea.y, ea.z) * thr * w2; flags |= 8; }
                        br[5 * kBlock] = cf; br[6 * kBlock] = w2;
                        flags |= 2;
                        its = itx;
                    }
                }
                br[7 * kBlock] = __int_as_float(flags);
            }
            {   // integrator.cpp:126: a non-finite channel contributes nothing
                const float pv[3] = {Lsum.x, Lsum.y, Lsum.z};
#pragma unroll
                for (int c = 0; c < 3; ++c) if (!finite_(pv[c])) wgt[c] = 0.f;
            }
            const Vec3f W(wgt[0], wgt[1], wgt[2]);
#if PSDR_SWEEP_DUMP == 1
            // (diagnostic) per-lane record of the sweep in the buffer behind P.g_tex: 64 floats per sample
            float *dbg = P.g_tex ? P.g_tex + 64 * lane : nullptr;
            if (dbg) {
                dbg[0] = (float) nb; dbg[1] = Lsum.x; dbg[2] = Lsum.y; dbg[3] = Lsum.z; dbg[4] = W.x; dbg[5] = W.y; dbg[6] = W.z; dbg[7] = le0 ? 1.f : 0.f;
                for (int k = 0; k < nb && k < 3; ++k) {
                    const float *br = brec + 11 * k * kBlock;
                    dbg[8 + 8 * k] = (float) __float_as_int(br[7 * kBlock]); dbg[9 + 8 * k] = br[8 * kBlock]; dbg[10 + 8 * k] = br[9 * kBlock]; dbg[11 + 8 * k] = br[10 * kBlock];
                    dbg[12 + 8 * k] = br[4 * kBlock]; dbg[13 + 8 * k] = br[5 * kBlock]; dbg[14 + 8 * k] = br[6 * kBlock]; dbg[15 + 8 * k] = (float) __float_as_int(vrec[3 * k * kBlock]);
                }
            }
#endif

            // ------------------------------------------------------------ pass 2: back over the bounces
            if (W.x != 0.f || W.y != 0.f || W.z != 0.f) {
                Vec3f cam_dirb(0.f);                    // adjoint of the camera ray's direction from an environment lookup along it
                if (le0 && e0 == env_id) cam_dirb = env_adjoint(dir0, W, true);
                else if (le0 && !P.skip_emitter) add_rgb(acc_emit, e0, W);          // the emitter seen by the camera
                Vec3f Abar(0.f);                       // d (w.L) / d thr_{k+1} from the bounces behind k
                Vec3f xb_next(0.f), nsb_next(0.f);     // what bounce k+1 gave vertex k+1 as ITS shading point
                Vec3f xb0(0.f), nsb0(0.f);             // the camera hit's totals
                for (int k = nb - 1; k >= 0; --k) {
                    const float *br = brec + 11 * k * kBlock;
                    const int flags = __float_as_int(br[7 * kBlock]);
                    const Vec3f thr_k(br[8 * kBlock], br[9 * kBlock], br[10 * kBlock]);
                    const float sgn = (flags & 4) ? -1.f : 1.f;
                    const float *vr = vrec + 3 * k * kBlock;
                    VtxGeom gk = load_vertex(S, __float_as_int(vr[0]), vr[kBlock], vr[2 * kBlock]);
                    if (k == 0) gk.x = x0;
                    const int bid = mesh_bsdf(S, gk.mesh);
                    const float4 ba = bid >= 0 ? S.ld(T.bsdf_off + 2 * bid) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const Vec3f rho(ba.x, ba.y, ba.z);
                    Vec3f xb(0.f), nsb(0.f), rhob(0.f), A_k(0.f);
                    SegGrad sg;
                    if (flags & 2) {
                        // thr_{k+1} = thr_k rho cf s_f,  L += thr_{k+1} Le_{k+1} w2
                        const float *vn = vrec + 3 * (k + 1) * kBlock;
                        const VtxGeom gz = load_vertex(S, __float_as_int(vn[0]), vn[kBlock], vn[2 * kBlock]);
                        const float cf = br[5 * kBlock], w2 = br[6 * kBlock];
                        const float sf = seg_eval(gk.x, gk.ns, sgn, gz.x, gz.fn, gz.area, sg) * cf;
                        Vec3f At = Abar;                                             // total adjoint of thr_{k+1}
                        if (flags & 8) {
                            const int ex = mesh_emitter(S, gz.mesh);
                            if (ex == env_id) {
                                const Vec3f dir = env_dir_at(__float_as_int(vn[0]), vn[kBlock], vn[2 * kBlock], gk.x);
                                At = At + W * env_radiance(dir) * w2;
                                xb = xb + dir_to_x(gk.x, gz.x, env_adjoint(dir, W * thr_k * rho * (sf * w2)));
                            } else {
                                const float4 ea = S.ld(T.emit_off + 2 * ex);
                                At = At + W * Vec3f(ea.x, ea.y, ea.z) * w2;
                                if (!P.skip_emitter) add_rgb(acc_emit, ex, W * thr_k * rho * (sf * w2));
                            }
                        }
                        const float sb = cf * (thr_k.x * rho.x * At.x + thr_k.y * rho.y * At.y + thr_k.z * rho.z * At.z);
                        rhob = rhob + thr_k * At * sf;
                        A_k = A_k + rho * At * sf;
                        xb = xb + sg.dx * sb; nsb = nsb + sg.dns * sb;
                        // vertex k+1 is complete: end point of this segment + shading point of bounce k+1
                        emit_glued(gz, xb_next + sg.dz * sb, nsb_next, sg.dnz * sb, sg.dA * sb);
                    }
                    if ((flags & 1) && (flags & 16)) {
                        // next-event sample on the environment map: L += thr_k rho Le(dir) cN s_N with a fixed sample point y on the scene box
                        const VtxGeom gh = load_vertex(S, __float_as_int(br[3 * kBlock]), br[0], br[kBlock]);       // the shadow ray's hit on the cube
                        const Vec3f y = gh.x;
                        const float cN = br[4 * kBlock];
                        const float sN = seg_eval(gk.x, gk.ns, sgn, y, gh.fn, 1.f, sg) * cN;
                        const Vec3f dir = env_dir_at(__float_as_int(br[3 * kBlock]), br[0], br[kBlock], gk.x);
                        const Vec3f Le = env_radiance(dir);
                        const Vec3f al = W * thr_k * rho * Le;
                        const float sb = cN * (al.x + al.y + al.z);
                        rhob = rhob + W * thr_k * Le * sN;
                        A_k = A_k + W * rho * Le * sN;
                        xb = xb + sg.dx * sb + dir_to_x(gk.x, y, env_adjoint(dir, W * thr_k * rho * sN));
                        nsb = nsb + sg.dns * sb;
#if PSDR_SWEEP_DUMP == 2
                        // (diagnostic, round 4) the environment lookup of this bounce's next-event sample: the build that kept the compiler defect visible -
                        // Le.x came back as the sky's 0.6 instead of the sun's 14.3 for the same direction
                        if (P.g_tex && k < 4) { float *q = P.g_tex + 64 * lane + 8 * k; q[0] = Le.x; q[1] = sN; q[2] = dir.x; q[3] = dir.y; q[4] = dir.z; q[5] = cN; q[6] = gk.x.x; q[7] = y.x; }
#elif PSDR_SWEEP_DUMP == 3
                        if (P.g_tex && k < 4) { float *q = P.g_tex + 64 * lane + 8 * k; q[0] = Le.x; q[1] = sN; }
#endif
                    } else if (flags & 1) {
                        // L += thr_k rho Le_h cN s_N
                        const VtxGeom gy = load_vertex(S, __float_as_int(br[0]), br[kBlock], br[2 * kBlock]);
                        const VtxGeom gh = load_vertex(S, __float_as_int(br[3 * kBlock]), 0.f, 0.f);
                        const float cN = br[4 * kBlock];
                        const float sN = seg_eval(gk.x, gk.ns, sgn, gy.x, gh.fn, gy.area, sg) * cN;
                        const int eh = mesh_emitter(S, gh.mesh);
                        const float4 ea = S.ld(T.emit_off + 2 * eh);
                        const Vec3f Le(ea.x, ea.y, ea.z);
                        const Vec3f al = W * thr_k * rho * Le;
                        const float sb = cN * (al.x + al.y + al.z);
                        rhob = rhob + W * thr_k * Le * sN;
                        A_k = A_k + W * rho * Le * sN;
                        if (!P.skip_emitter) add_rgb(acc_emit, eh, W * thr_k * rho * sN);
                        xb = xb + sg.dx * sb; nsb = nsb + sg.dns * sb;
                        emit_glued(gy, sg.dz * sb, Vec3f(0.f), Vec3f(0.f), sg.dA * sb);       // the light sample: position and area of ITS triangle
                        emit_glued(gh, Vec3f(0.f), Vec3f(0.f), sg.dnz * sb, 0.f);             // the normal of the triangle the shadow ray hit
                    }
                    if (bid >= 0 && !P.skip_bsdf) add_rgb(acc_bsdf, bid, rhob);
#if PSDR_SWEEP_DUMP == 1
                    if (dbg && k < 3) { dbg[32 + 8 * k] = rhob.x; dbg[33 + 8 * k] = rhob.y; dbg[34 + 8 * k] = rhob.z; dbg[35 + 8 * k] = A_k.x; dbg[36 + 8 * k] = A_k.y; dbg[37 + 8 * k] = A_k.z; dbg[38 + 8 * k] = (float) bid; dbg[39 + 8 * k] = (float) flags; }
#endif
                    xb_next = xb; nsb_next = nsb;
                    Abar = A_k;
                    if (k == 0) { xb0 = xb; nsb0 = nsb; }
                }
                // the camera hit: x_0 = o + t d and ns_0 = normalize(blend(u, v)) with (u, v, t) = Moeller-Trumbore(p0, e1, e2; o, d)
                if (nb > 0 || le0) {
                    const VtxGeom g0 = load_vertex(S, slot0, u0, v0);
                    float ub = 0.f, vb = 0.f;
                    const float tb = dot(ray.d, xb0);
                    Vec3f ob = xb0, db = xb0 * t0;
                    const bool want0 = wanted(g0);
                    if (g0.flat) { if (want0) add_vec(g0, 18, nsb0); }
                    else {
                        const Vec3f nbb = blend_adjoint(g0, nsb0);
                        ub = dot(g0.n1 - g0.n0, nbb); vb = dot(g0.n2 - g0.n0, nbb);
                        if (want0) { add_vec(g0, 9, nbb * (1.f - u0 - v0)); add_vec(g0, 12, nbb * u0); add_vec(g0, 15, nbb * v0); }
                    }
                    Vec3f p0b, e1b, e2b, ob2, db2;
                    mt_adjoint(g0.p0, g0.e1, g0.e2, ray.o, ray.d, ub, vb, tb, p0b, e1b, e2b, ob2, db2);
                    if (want0) { add_vec(g0, 0, p0b); add_vec(g0, 3, e1b); add_vec(g0, 6, e2b); }
                    if (P.g_cam != nullptr) {
                        // o = to_world . (o_cam, 1), d = to_world . (d_cam, 0)  (primary_ray_pose_tangent)
                        ob = ob + ob2; db = db + db2 + cam_dirb;
                        const Vec3f pc = xform_pos(cam.sample_to_camera, Vec3f(sx, sy, 0.f));
                        const Vec3f o_cam = cam.ortho ? pc : Vec3f(0.f), d_cam = cam.ortho ? Vec3f(0.f, 0.f, 1.f) : normalize(pc);
                        const float oc[4] = {o_cam.x, o_cam.y, o_cam.z, 1.f}, dc[4] = {d_cam.x, d_cam.y, d_cam.z, 0.f};
                        const float obv[3] = {ob.x, ob.y, ob.z}, dbv[3] = {db.x, db.y, db.z};
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float val = obv[r] * oc[c] + dbv[r] * dc[c];
                                if (val != 0.f && finite_(val)) atomicAdd(&acc_cam[4 * r + c], val);
                            }
                    }
                }
            }
            have = false;
        }
        // every lane of the wave is here: the pending camera lookups, one atomic per distinct texel
        if (__ballot(pend_env) != 0ull) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int key = pend_env ? pe_idx[k] : -1;
                for (int it = 0; it < 8; ++it) {
                    const unsigned long long m = __ballot(key >= 0);
                    if (m == 0ull) break;
                    const int leader = (int) __builtin_ctzll(m);
                    const int lk = __shfl(key, leader);
                    const bool same = key == lk;
                    float s0 = same ? pe_val[3 * k] : 0.f, s1 = same ? pe_val[3 * k + 1] : 0.f, s2 = same ? pe_val[3 * k + 2] : 0.f;
                    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
                    if (lane_id == leader) {
                        if (s0 != 0.f) atomicAdd(&P.g_env[3ll * lk], s0);
                        if (s1 != 0.f) atomicAdd(&P.g_env[3ll * lk + 1], s1);
                        if (s2 != 0.f) atomicAdd(&P.g_env[3ll * lk + 2], s2);
                    }
                    if (same) key = -1;
                }
                if (key >= 0) {           // more than eight distinct texels in the wave: the rest go one by one
                    if (pe_val[3 * k] != 0.f) atomicAdd(&P.g_env[3ll * key], pe_val[3 * k]);
                    if (pe_val[3 * k + 1] != 0.f) atomicAdd(&P.g_env[3ll * key + 1], pe_val[3 * k + 1]);
                    if (pe_val[3 * k + 2] != 0.f) atomicAdd(&P.g_env[3ll * key + 2], pe_val[3 * k + 2]);
                }
            }
            pend_env = false;
#pragma unroll
            for (int q = 0; q < 12; ++q) pe_val[q] = 0.f;
        }
    }
    __syncthreads();
    if (P.g_cam != nullptr && threadIdx.x < 12 && acc_cam[threadIdx.x] != 0.f) atomicAdd(&P.g_cam[threadIdx.x], acc_cam[threadIdx.x]);
    if (P.g_env_scale != nullptr && threadIdx.x == 12 && acc_cam[12] != 0.f) atomicAdd(P.g_env_scale, acc_cam[12]);
    if (P.g_env_xf != nullptr && threadIdx.x >= 16 && threadIdx.x < 27 && acc_cam[threadIdx.x] != 0.f) atomicAdd(&P.g_env_xf[threadIdx.x - 16], acc_cam[threadIdx.x]);
    if (P.g_uv_xf != nullptr && threadIdx.x >= 28 && threadIdx.x < 32 && acc_cam[threadIdx.x] != 0.f) atomicAdd(&P.g_uv_xf[12 * T.n_bsdfs + threadIdx.x - 28], acc_cam[threadIdx.x]);
    for (int i = threadIdx.x; i < P.n_hot * 22; i += kBlock) if (acc[i] != 0.f) atomicAdd(&P.g_tri[P.hot_inv[i / 22] * 22 + i % 22], acc[i]);
    for (int i = threadIdx.x; i < T.n_bsdfs * 3; i += kBlock) if (acc_bsdf[i] != 0.f) atomicAdd(&P.g_bsdf[i], acc_bsdf[i]);
    for (int i = threadIdx.x; i < T.n_emitters * 3; i += kBlock) if (acc_emit[i] != 0.f) atomicAdd(&P.g_emitter[i], acc_emit[i]);
    for (int i = threadIdx.x; i < P.env_lds; i += kBlock) if (acc_env[i] != 0.f) atomicAdd(&P.g_env[i], acc_env[i]);
}

} // namespace psdr
