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

constexpr int kAdjMaxDepth = 4;
constexpr int kAdjHitWords = 4 * (1 + 2 * kAdjMaxDepth);      // recorded hits per lane
constexpr int kAdjExtWords = 8;                                // light-sample slots per lane

struct AdjointParams {
    int max_depth, hide_emitters;
    unsigned long long seed, skip;
    const int *pix_ids;
    long long begin, end;
    int shard_rank, shard_count;
    long long n_local;
    unsigned long long *counter;
    const float *w;                 // d loss / d image, [n_pixels * 3]
    float *g_tri;                   // [n_tris * 22], ORIGINAL triangle order
    float *g_bsdf, *g_emitter;      // [n_bsdfs * 3], [n_emitters * 3]
    int lds_accum;                  // 1: accumulate in LDS first (small scenes), 0: global atomics
    int mis;                        // -1: PathTracer; 0/1/2: DirectIntegrator(mis)
    int field, field_object;        // >= 0: first-hit integrator
    float intensity, d_intensity;
};

template <bool LDS>
PSDR_DEV void adj_add(float *lds_g, float *glob, int idx, float v, bool use_lds) {
    if (v == 0.f || !finite_(v)) return;
    if (use_lds) atomicAdd(&lds_g[idx], v); else atomicAdd(&glob[idx], v);
}

// scratch: per-block LDS region behind the blob/stack: [hits: kAdjHitWords x 256][ext: kAdjExtWords x 256][accumulators]
template <bool LDS>
PSDR_DEV void run_interior_adjoint(SceneView<LDS> &S, const SensorDev &cam, const AdjointParams &P, float *scratch) {
    const SceneTables &T = *S.T;
    const int lane_id = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane_id) - 1ull;
    const float inv_spp = T.spp > 1 ? 1.f / (float) T.spp : 1.f;
    float *rec = scratch + threadIdx.x;
    int *ext = reinterpret_cast<int *>(scratch + kAdjHitWords * kBlock) + threadIdx.x;
    float *acc = scratch + (kAdjHitWords + kAdjExtWords) * kBlock;
    const int n_acc = T.n_tris * 22 + T.n_bsdfs * 3 + T.n_emitters * 3;
    const bool use_lds = P.lds_accum != 0;
    if (use_lds) {
        for (int i = threadIdx.x; i < n_acc; i += kBlock) acc[i] = 0.f;
        __syncthreads();
    }
    float *acc_bsdf = acc + T.n_tris * 22, *acc_emit = acc_bsdf + T.n_bsdfs * 3;
    S.rec = rec; S.ext = ext;

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
            const RayT<true> ray = sample_primary_ray<true>(cam, (bx + jx) / (float) T.width, (by + jy) / (float) T.height);
            const LaneRng rng0 = rng;
            S.mode = 1; S.rec_n = 0; S.rec_i = 0; S.ext_n = 0; S.probe_kind = 0;
            const Vec3d L0 = Li<true, LDS, false>(S, rng, ray, true, P.max_depth, P.hide_emitters != 0);
            const int n_hits = S.rec_n, n_ext = S.ext_n;
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
                const Vec3d L = Li<true, LDS, false>(S, r, ray, true, P.max_depth, P.hide_emitters != 0);
                const float t[3] = {L.x.d, L.y.d, L.z.d};
                float g = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) if (finite_(t[c])) g += w[c] * t[c];
                return g;
            };
            auto slot_at = [&](int i) -> int { return i < n_hits ? __float_as_int(rec[4 * i * kBlock]) : ext[(i - n_hits) * kBlock]; };
            if (w[0] != 0.f || w[1] != 0.f || w[2] != 0.f) {
                const int n_all = n_hits + n_ext;
                // triangles
                for (int i = 0; i < n_all; ++i) {
                    const int slot = slot_at(i);
                    if (slot < 0) continue;
                    bool dup = false;
                    for (int j = 0; j < i; ++j) dup = dup || (slot_at(j) == slot);
                    if (dup) continue;
                    const int orig = __float_as_int(S.ld(T.shade_off + 6 * slot + 3).w);
                    S.probe_kind = 1; S.probe_id = slot;
                    for (int comp = 0; comp < 22; ++comp) {
                        S.probe_comp = comp;
                        adj_add<LDS>(acc, P.g_tri, orig * 22 + comp, probe(), use_lds);
                    }
                }
                // reflectances and radiances of the meshes met along the path
                for (int i = 0; i < n_all; ++i) {
                    const int slot = slot_at(i);
                    if (slot < 0) continue;
                    const int mesh = __float_as_int(S.ld(T.shade_off + 6 * slot + 1).w);
                    const int bs = mesh_bsdf(S, mesh), em = mesh_emitter(S, mesh);
                    bool dup_b = false, dup_e = false;
                    for (int j = 0; j < i; ++j) {
                        const int sj = slot_at(j);
                        if (sj < 0) continue;
                        const int mj = __float_as_int(S.ld(T.shade_off + 6 * sj + 1).w);
                        dup_b = dup_b || (mesh_bsdf(S, mj) == bs);
                        dup_e = dup_e || (mesh_emitter(S, mj) == em);
                    }
                    if (bs >= 0 && !dup_b) {
                        S.probe_kind = 2; S.probe_id = bs;
                        for (int comp = 0; comp < 3; ++comp) { S.probe_comp = comp; adj_add<LDS>(acc_bsdf, P.g_bsdf, bs * 3 + comp, probe(), use_lds); }
                    }
                    if (em >= 0 && !dup_e) {
                        S.probe_kind = 3; S.probe_id = em;
                        for (int comp = 0; comp < 3; ++comp) { S.probe_comp = comp; adj_add<LDS>(acc_emit, P.g_emitter, em * 3 + comp, probe(), use_lds); }
                    }
                }
            }
            S.mode = 0; S.probe_kind = 0;
            have = false;
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < T.n_tris * 22; i += kBlock) if (acc[i] != 0.f) atomicAdd(&P.g_tri[i], acc[i]);
        for (int i = threadIdx.x; i < T.n_bsdfs * 3; i += kBlock) if (acc_bsdf[i] != 0.f) atomicAdd(&P.g_bsdf[i], acc_bsdf[i]);
        for (int i = threadIdx.x; i < T.n_emitters * 3; i += kBlock) if (acc_emit[i] != 0.f) atomicAdd(&P.g_emitter[i], acc_emit[i]);
    }
}

} // namespace psdr
