// api.hip — gfx950 kernels and the C ABI of libpsdr_hip.so (include/psdr_hip.h).
//
// Kernel structure (one thread = one sample lane of the reference's wavefront arrays):
//   * persistent workgroups of 256 threads (4 wave64): every wave pulls batches of 256 work items from a global queue and hands
//     them to whichever of its lanes have finished their path (ballot / popcount regeneration, paths.h), so the lanes of a wave
//     belong to different pixels at different bounces;
//   * at start each workgroup stages the scene blob into LDS (small scenes) with 16-byte loads;
//   * each lane re-derives its RNG state (sampler.h) and traces / shades its whole path in registers (brute-force scenes) or as a
//     traversal worker on the wave's shared ray queue (BVH scenes, trav4.h);
//   * a finished path adds its value (and tangent) to its pixel with one float atomic per channel, as the reference does
//     (scatter_reduce, integrator.cpp:127-129).  Round 1 combined the lanes of a pixel with a segmented scan first; with persistent
//     regeneration the finishing lanes of a wave rarely share a pixel, and the atomics are not what the kernels wait for: the C3
//     interior kernel takes 1.698 ms with them and 1.685 ms with the adds compiled out (round 4, -0.8 %), so nothing is aggregated.
// There is no host synchronisation inside a render call (the reference syncs before each of its 7+
// OptiX launches, scene_optix.cpp:345).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/psdr_hip.h"
#include "scene_obj.h"
#include "edges.h"
#include "paths.h"
#include "adjoint.h"
#include "adjoint_mat.h"

using namespace psdr;

// ------------------------------------------------------------------------------------------------
// error plumbing
// (psdr::api_fail and HIPCHK: scene_obj.h; the message store itself is in the host part of this file)
[[maybe_unused]] static int fail(const std::string &msg) { return psdr::api_fail(msg); }

// ------------------------------------------------------------------------------------------------
// device helpers shared by the kernels
struct LaneRange { long long begin, end; };

struct RenderParams {
    int max_depth, hide_emitters;
    unsigned long long seed, skip;
    const int *pix_ids;
    int n_pix;
    LaneRange range;
    int shard_rank, shard_count;   // rank r of c evaluates the 256-lane chunks k with k % c == r
};

template <int LDS>
PSDR_DEV SceneView<LDS> make_view(const float4 *blob, const SceneTables &T, float4 *smem) {
    const float4 *B = blob;
    if (in_lds(LDS)) {
        for (int i = threadIdx.x; i < T.blob_words; i += kBlock) smem[i] = blob[i];
        __syncthreads();
        B = smem;
    }
    SceneView<LDS> S;
    S.B = B; S.G = blob; S.T = &T;
    S.stack = reinterpret_cast<int *>(smem + (in_lds(LDS) ? T.blob_words : 0)) + threadIdx.x;
    S.c_nodes = S.c_tris = S.c_rays = S.c_hits = 0u;
    S.mis = -1; S.field = -1; S.field_object = -1; S.intensity = 1.f; S.d_intensity = 0.f; S.mode = 0; S.rec = nullptr; S.rec_i = 0; S.rec_n = 0; S.ext = nullptr; S.ext_n = 0; S.probe_kind = 0; S.probe_id = 0; S.probe_comp = 0;
    S.lk = nullptr; S.lk_n = 0; S.lk_max = 0; S.ext_max = 0; S.probe_u = 0.f; S.probe_v = 0.f;
    t4_init_lds(S);
    return S;
}

template <int LDS> PSDR_DEV void flush_counters(const SceneView<LDS> &S, Counters *ctr) {
    unsigned long long v[4] = {S.c_rays, S.c_nodes, S.c_tris, S.c_hits};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned long long x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long *>(ctr) + k, x);
    }
}

template <int LDS> PSDR_DEV float *scratch_base(float4 *smem, const SceneTables &T) {
    return reinterpret_cast<float *>(smem + (in_lds(LDS) ? T.blob_words : 0)) + T.stack_depth * kBlock;
}

// BVH scenes of the global-memory classes take the decoupled form (traversal refilled per ray, paths.h); brute-force scenes and the
// LDS class keep the lock-step form
template <bool AD, int LDS, bool COUNT, int MODE>
PSDR_DEV void run_paths_any(SceneView<LDS> &S, const SensorDev &cam, const PathParams &P) {
    if constexpr (!in_lds(LDS)) {
#ifndef PSDR_NO_ASYNC        // measurement knob: the lock-step form on BVH scenes too
        if (S.T->n_tris > kBruteForceMax) { run_paths_async<AD, LDS, COUNT, MODE>(S, cam, P); return; }
#endif
    }
    run_paths<AD, LDS, COUNT, MODE>(S, cam, P);
}

// ------------------------------------------------------------------------------------------------
// interior term (MODE 0) and primary-edge term (MODE 1): persistent lanes with path regeneration, paths.h
template <bool AD, int LDS, bool COUNT, int MODE>
#ifndef PSDR_GLOBAL_C_WAVES
#define PSDR_GLOBAL_C_WAVES 4
#endif
#ifndef PSDR_GLOBAL_AD_WAVES     // class 0: 2 / 3 waves measured on the envmap notebook's glossy bunny (512², 32 spp): interior 4.93 / 5.53 ms
#define PSDR_GLOBAL_AD_WAVES 2
#endif
#ifndef PSDR_LDS_AD_WAVES       // classes 1 / 3 (scene in LDS), AD kernel: 2 / 3 / 4 waves per SIMD measured on C3 1.67 / 1.79 / 2.11 ms (the tangents spill at 168 registers)
#define PSDR_LDS_AD_WAVES 2
#endif
#ifndef PSDR_LDS_C_WAVES
#define PSDR_LDS_C_WAVES 4
#endif
#ifndef PSDR_LEAN_AD_WAVES       // class 2 (BVH scenes): 1 / 2 / 3 / 4 waves per SIMD measured on config 5's interior kernel 47.9 / 31.6 / 34.5 / 37.1 ms, sphere box 13.2 / 7.6 / 8.2 / 8.8 ms -
#define PSDR_LEAN_AD_WAVES 2     // the (value, tangent) path state spills less at 256 registers than it gains from a third wave; the C-mode kernels want their four (3: +16 %, 2: +60 %)
#endif
__global__ __launch_bounds__(kBlock, (AD ? (in_lds(LDS) ? PSDR_LDS_AD_WAVES : (LDS == 2 ? PSDR_LEAN_AD_WAVES : PSDR_GLOBAL_AD_WAVES)) : (in_lds(LDS) ? PSDR_LDS_C_WAVES : PSDR_GLOBAL_C_WAVES))) void k_paths(const float4 *__restrict__ blob, const SceneTables T, const SensorDev cam,
                                                  const PathParams P, Counters *ctr) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneView<LDS> S = make_view<LDS>(blob, T, smem);
    S.mis = P.mis; S.field = P.field; S.field_object = P.field_object; S.intensity = P.intensity; S.d_intensity = P.d_intensity;
    if (MODE == 1 && P.adj_w != nullptr && P.lds_acc) {
        // reverse mode of the primary-edge term: 8.4 M samples add into a 42 x 4 table - accumulate per workgroup in LDS
        float *acc = scratch_base<LDS>(smem, T);
        for (int i = threadIdx.x; i < 4 * P.n_prim; i += kBlock) acc[i] = 0.f;
        __syncthreads();
        PathParams Q = P;
        Q.g_prim = acc;
        run_paths_any<AD, LDS, COUNT, MODE>(S, cam, Q);
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * P.n_prim; i += kBlock) if (acc[i] != 0.f) atomicAdd(&P.g_prim[i], acc[i]);
    } else {
        run_paths_any<AD, LDS, COUNT, MODE>(S, cam, P);
    }
    if (COUNT) flush_counters(S, ctr);
}

// reverse mode of the interior term (adjoint.h)
#ifndef PSDR_ADJ_REC_LDS        // measurement knob: 1 = the per-lane records stay in LDS whenever they fit 160 KB (rounds 2-4)
#define PSDR_ADJ_REC_LDS 0
#endif
// waves per SIMD of the interior adjoint kernels.  Class 2 (BVH scenes): 2 - with the per-lane records in global memory two workgroups fit a CU (40 KB of traversal
// rows + the hot accumulators <= 80 KB), and at <= 256 registers both run: config 5's interior adjoint 46.0 -> 29.3 ms.  The other classes keep the compiler's choice.
#ifndef PSDR_ADJ_WAVES
#define PSDR_ADJ_WAVES(cls) ((cls) == 2 ? 2 : 1)
#endif
#ifndef PSDR_SEC_ADJ_WAVES
#define PSDR_SEC_ADJ_WAVES 1
#endif
#ifndef PSDR_SEC_HOT_MAX        // triangle rows the secondary-edge adjoint keeps in LDS on large scenes (9 floats each)
#define PSDR_SEC_HOT_MAX 256
#endif
template <int LDS>
__global__ __launch_bounds__(kBlock, PSDR_ADJ_WAVES(LDS)) void k_interior_adjoint(const float4 *__restrict__ blob, const SceneTables T, const SensorDev cam,
                                                             const AdjointParams P) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneView<LDS> S = make_view<LDS>(blob, T, smem);
    S.mis = P.mis; S.field = P.field; S.field_object = P.field_object; S.intensity = P.intensity; S.d_intensity = P.d_intensity;
    if constexpr (!has_mat(LDS)) { if (P.sweep) { run_interior_adjoint_sweep<LDS>(S, cam, P, scratch_base<LDS>(smem, T)); return; } }
    run_interior_adjoint<LDS>(S, cam, P, scratch_base<LDS>(smem, T));
}

// the material sweep (adjoint_mat.h) in a kernel of its own: its registers are not shared with the record-and-probe form
template <int LDS>
__global__ __launch_bounds__(kBlock) void k_interior_adjoint_mat(const float4 *__restrict__ blob, const SceneTables T, const SensorDev cam,
                                                                 const AdjointParams P) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneView<LDS> S = make_view<LDS>(blob, T, smem);
    S.mis = P.mis; S.field = P.field; S.field_object = P.field_object; S.intensity = P.intensity; S.d_intensity = P.d_intensity; S.uv_adj = P.g_uv_xf != nullptr;
    run_interior_adjoint_sweep_mat<LDS>(S, cam, P, scratch_base<LDS>(smem, T));
}

struct GuidingDev {          // HyperCubeDistribution<3>, reference src/core/cube_distrb.cpp:10-64
    const float *pmf, *cmf;
    const int *guide;        // search bounds per sample bucket (shade.h::sample_reuse_guided), [guide_n + 1]; guide_n = 0: none
    int guide_n;
    float sum;
    int reso[3], num_cells;
    float unit[3];
};

PSDR_DEV float guiding_sample_reuse(const GuidingDev &G, Vec3f &s) {
    float pdf;
    const int idx = sample_reuse_guided<true>(G.guide, G.guide_n, G.num_cells, G.sum, [&](int i) { return G.pmf[i]; }, [&](int i) { return G.cmf[i]; }, s.z, pdf);
    const int c0 = idx / (G.reso[1] * G.reso[2]);
    const int rem = idx - c0 * (G.reso[1] * G.reso[2]);
    const int c1 = rem / G.reso[2], c2 = rem - c1 * G.reso[2];
    s.x = (s.x + (float) c0) * G.unit[0];
    s.y = (s.y + (float) c1) * G.unit[1];
    s.z = (s.z + (float) c2) * G.unit[2];
    return pdf * (float) G.num_cells;
}

// secondary-edge term, reference path.cpp:274-294.
// Only ~1 in 6 boundary-segment samples of the README scene passes the cheap validity test of
// sample_boundary_segment_direct (silhouette condition + light facing), and only those trace rays.  Each lane
// therefore keeps drawing candidates (RNG seed + three draws + the validity test, no ray) until the wave holds
// enough valid ones, and the traced part (3 rays) runs with nearly all lanes active (stage r01a: 17 %).
// waves per SIMD of the secondary-edge kernel (forward): its candidate rounds are chains of dependent loads, so it wants occupancy - measured on
// config 5 (class 2) 2 / 3 / 4 waves: 48.6 / 38.3 / 33.2 ms (the compiler's own choice was 2), on C3 (class 1) 3 / 4 / 5: 0.79 / 0.72 / 0.76 ms, on the
// Microfacet box (class 3) 0.83 -> 0.75 ms, on the glossy bunny under the ballroom map (class 0) 1.85 -> 1.52 ms; the reverse-mode instantiation keeps
// the compiler's choice (1 = no constraint; 2-4 measured: +1-3 %)
#ifndef PSDR_SEC_WAVES
#define PSDR_SEC_WAVES 4
#endif
template <int LDS, bool COUNT, bool ADJ>
__global__ __launch_bounds__(kBlock, (ADJ ? PSDR_SEC_ADJ_WAVES : PSDR_SEC_WAVES)) void k_secondary_edges(
                                                            const float4 *__restrict__ blob, const SceneTables T, const SecEdgeTables E,
                                                            const SensorDev cam, const PathParams P, const GuidingDev G, const int use_guiding,
                                                            Counters *ctr) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneView<LDS> S = make_view<LDS>(blob, T, smem);
    const int lane_id = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane_id) - 1ull;
    long long q_next = 0, q_end = 0;
    bool exhausted = false;
    bool have = false;
    BoundarySegSampleDirect bss;
    bss.valid = false; bss.pdf = 1.f; bss.p0 = Vec3d(Dual(0.f)); bss.edge = bss.edge2 = bss.p2 = bss.n = Vec3f(0.f); bss.emitter_slot = -1; bss.edge_id = 0; bss.s1 = 0.f;
    float pdf0 = 1.f;
    if constexpr (ADJ) if (P.lds_acc || P.n_hot > 0) {
        float *acc = scratch_base<LDS>(smem, T) + kSecAdjScratch;
        const int n_acc = P.lds_acc ? 6 * P.n_sec + 22 * T.n_tris : 9 * P.n_hot;
        for (int i = threadIdx.x; i < n_acc; i += kBlock) acc[i] = 0.f;
        __syncthreads();
    }
    // camera-pose adjoint: 12 entries every sample adds to - kept in LDS (behind the 3 recorded hits of this kernel)
    float *acc_cam = scratch_base<LDS>(smem, T) + kSecAdjLaneWords * kBlock;
    if constexpr (ADJ) if (P.g_cam != nullptr) {
        if (threadIdx.x < 16) acc_cam[threadIdx.x] = 0.f;
        __syncthreads();
    }
    // Candidates: one in six passes the silhouette / light-facing test.
    auto draw = [&](long long item, BoundarySegSampleDirect &out, float &pdf_out) -> bool {
        const long long chunk = (item >> 8) * P.shard_count + P.shard_rank;
        const long long lane = P.begin + (chunk << 8) + (item & 255);
        if (lane >= P.end) return false;
        LaneRng rng;
        rng.seed(P.seed + (unsigned long long) lane, (unsigned long long) lane, P.skip);
        Vec3f s3;
        s3.x = rng.next_1d(); s3.y = rng.next_1d(); s3.z = rng.next_1d();
        pdf_out = use_guiding ? guiding_sample_reuse(G, s3) : 1.f;
        out = sample_boundary_segment_direct<LDS>(S, E, s3);
        return out.valid;
    };
    auto refill_queue = [&]() {
        if (q_next >= q_end && !exhausted) {
            unsigned long long base = 0;
            if (lane_id == 0) base = atomicAdd(P.counter, (unsigned long long) kFetchBatch);
            base = __shfl(base, 0);
            if ((long long) base >= P.n_local) exhausted = true;
            else { q_next = (long long) base; q_end = q_next + kFetchBatch < P.n_local ? q_next + kFetchBatch : P.n_local; }
        }
    };
    // reverse mode, closed form: the tangent is value0 . n.(e1 du + e2 dv) with (u, v) = Moeller-Trumbore(emitter triangle; x1, sd), sd = normalize(p0 - x1)
    // and x1 = the camera ray's hit sliding along that ray - two adjoint solves instead of 21-33 replays
    auto scatter_closed = [&](int idx, const SecAdjInfo &I, const BoundarySegSampleDirect &seg, float pdf_seg) {
        if constexpr (ADJ) {
            float *g_sec = P.lds_acc ? scratch_base<LDS>(smem, T) + kSecAdjScratch : P.g_sec;
            float *g_tri = P.lds_acc ? g_sec + 6 * P.n_sec : P.g_tri;
            const float v0c[3] = {I.value0.x, I.value0.y, I.value0.z};
            float gsum = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float k = P.adj_w[3 * (long long) idx + c];
                if (pdf_seg > kEpsilon) k /= pdf_seg;
                if (T.sppse > 1) k /= (float) T.sppse;
                if (finite_(v0c[c])) gsum += k * v0c[c];
            }
            if (gsum != 0.f && finite_(gsum)) {
                auto add3 = [&](float *tab, int row, const Vec3f &val) {
                    if (val.x != 0.f && finite_(val.x)) atomicAdd(&tab[row], val.x);
                    if (val.y != 0.f && finite_(val.y)) atomicAdd(&tab[row + 1], val.y);
                    if (val.z != 0.f && finite_(val.z)) atomicAdd(&tab[row + 2], val.z);
                };
                // triangle rows: hot ones per workgroup in LDS (config 5: 93 -> 42 ms; 51 of the 93 were the scatter, most of it the 108 floats of the scene box)
                float *hot_acc = scratch_base<LDS>(smem, T) + kSecAdjScratch;
                auto add_tri = [&](int orig, int comp, const Vec3f &val) {
                    const int hot = (!P.lds_acc && P.n_hot > 0) ? P.hot_map[orig] : -1;
                    if (hot >= 0 && hot < P.n_hot) add3(hot_acc, 9 * hot + comp, val); else add3(g_tri, 22 * orig + comp, val);
                };
                Vec3f a2, b2, c2, a1, b1, c1;
                load_geom<false, LDS>(S, I.slot2, a2, b2, c2);
                load_geom<false, LDS>(S, I.slot1, a1, b1, c1);
                Vec3f p0b, e1b, e2b, ob, db;
                mt_adjoint(a2, b2, c2, I.x1, I.sd, gsum * dot(I.n, b2), gsum * dot(I.n, c2), 0.f, p0b, e1b, e2b, ob, db);
                const int orig2 = __float_as_int(S.ld(T.shade_off + 6 * I.slot2 + 3).w), orig1 = __float_as_int(S.ld(T.shade_off + 6 * I.slot1 + 3).w);
                add_tri(orig2, 0, p0b); add_tri(orig2, 3, e1b); add_tri(orig2, 6, e2b);
                const Vec3f q = detach(seg.p0) - I.x1;
                const Vec3f qb = (db - I.sd * dot(I.sd, db)) / norm(q);          // through sd = normalize(p0 - x1)
                add3(g_sec, 6 * seg.edge_id, qb); add3(g_sec, 6 * seg.edge_id + 3, qb * seg.s1);
                const Vec3f xb = ob - qb;                                         // the camera hit x1 = o + t d
                Vec3f p0c, e1c, e2c, oc2, dc2;
                mt_adjoint(a1, b1, c1, I.cam_o, I.cam_d, 0.f, 0.f, dot(I.cam_d, xb), p0c, e1c, e2c, oc2, dc2);
                add_tri(orig1, 0, p0c); add_tri(orig1, 3, e1c); add_tri(orig1, 6, e2c);
                if (P.g_cam != nullptr) {
                    const float t1 = dot(I.x1 - I.cam_o, I.cam_d);
                    const Vec3f obt = xb + oc2, dbt = xb * t1 + dc2;
                    const Vec3f pc = xform_pos(cam.sample_to_camera, Vec3f(I.qx, I.qy, 0.f));
                    const Vec3f o_cam = cam.ortho ? pc : Vec3f(0.f), d_cam = cam.ortho ? Vec3f(0.f, 0.f, 1.f) : normalize(pc);
                    const float occ[4] = {o_cam.x, o_cam.y, o_cam.z, 1.f}, dcc[4] = {d_cam.x, d_cam.y, d_cam.z, 0.f};
                    const float obv[3] = {obt.x, obt.y, obt.z}, dbv[3] = {dbt.x, dbt.y, dbt.z};
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float val = obv[r] * occ[c] + dbv[r] * dcc[c];
                            if (val != 0.f && finite_(val)) atomicAdd(&acc_cam[4 * r + c], val);
                        }
                }
            }
        }
    };
    // THE PIPELINED FORM (round 6; forward mode and the closed-form reverse mode).  A valid segment costs three rays, each of which can end it: the ray to its emitter
    // sample, the opposite ray to the surface point p1 the sensor sees it on, the camera ray through p1.  Traced one ray at a time by the lanes that still hold a segment
    // (eval_boundary_segment), the three traversals of a wave run at 58, ~40 and ~25 of 64 lanes, each with the tail of its slowest walk: lane utilisation 0.29 on the
    // 82 k-triangle scene of config 5.  Here a lane's segment is in one of two stages - FIRST RAYS (the emitter ray, which may stop at any occluder in front of the sample,
    // and the opposite ray, posted TOGETHER: the second one is speculative, and wasted when the first fails) and CAMERA RAY - and every call of trace2 carries the rays of
    // ALL lanes, whatever their stage: a lane whose segment ended takes the next candidate in the same iteration.  Candidates: every lane draws in every round, the indices
    // of the valid ones (one in six) wait in a per-wave pool in LDS (a segment is a function of its item's index); the pool sits in rows the traversal uses - stack rows of
    // BVH scenes, the cold path-state rows of brute-force scenes - so what is left of it rides in a register across a trace (fewer than 64 entries by construction).
    constexpr int kPoolCap = 192;            // < 64 needed + <= 64 new per round
    const bool pooled = T.stack_depth >= kPoolCap / 64 && P.n_local < (1ll << 31);
    // (BVH scenes.  Brute-force scenes keep round 5's form - the pool, then the three rays one after the other: their trace2 pays for a speculative second ray in full,
    //  C3's kernel 0.60 -> 0.62 ms pipelined)
    const bool pipelined = pooled && T.n_tris > kBruteForceMax && (!ADJ || P.sec_closed);
    lds_uint_t *pool_base = (lds_uint_t *) (S.stack - threadIdx.x) + (threadIdx.x & ~63);
    // entry i of this wave's pool: row i / 64 of the area, in the wave's OWN 64 columns - the rows are per-lane rows of all four waves (traversal stacks, parked rays), and
    // another wave may be in the middle of a trace while this one collects candidates
    auto pool_at = [&](int i) -> lds_uint_t & { return pool_base[(i >> 6) * kBlock + (i & 63)]; };
    int pool_n = 0;
    if (pipelined) {
        int stage = 0;
        unsigned spare = 0u;
        Hit h2, h1c;
        h2.slot = -1; h2.u = h2.v = h2.t = 0.f; h1c = h2;
        Vec3f p1(0.f);
        if constexpr (ADJ) { S.mode = 0; S.probe_kind = 99; S.probe_id = -1; }          // (zero tangents everywhere: only the primal factors are wanted)
        for (;;) {
            // ---- candidates for the lanes without a segment
            const int n_need = __popcll(__ballot(stage == 0));
            while (pool_n < n_need) {
                refill_queue();
                if (q_next >= q_end) break;
                const long long item = q_next + lane_id;
                bool ok = false;
                if (item < q_end) { BoundarySegSampleDirect tmp; float tpdf; ok = draw(item, tmp, tpdf); }
                const unsigned long long m_ok = __ballot(ok);
                if (ok) pool_at(pool_n + __popcll(m_ok & lt_mask)) = (unsigned) item;
                pool_n += __popcll(m_ok);
                const long long left = q_end - q_next;
                q_next += left < 64 ? left : 64;
            }
            wave_sync();
            {
                const unsigned long long m_need = __ballot(stage == 0);
                const int n_take = pool_n < n_need ? pool_n : n_need, rank = __popcll(m_need & lt_mask);
                if (stage == 0 && rank < n_take) {
                    if (draw((long long) pool_at(pool_n - n_take + rank), bss, pdf0)) stage = 1;
                    if constexpr (ADJ) bss.p0 = promote(detach(bss.p0));
                }
                pool_n -= n_take;
                if (lane_id < pool_n) spare = pool_at(lane_id);
            }
            wave_sync();
            if (__ballot(stage != 0) == 0ull) { if (exhausted && q_next >= q_end && pool_n == 0) break; continue; }
            // ---- the rays of every lane's stage in one call
            const Vec3f p0v = detach(bss.p0), dirv = normalize(bss.p2 - p0v);
            Vec3f oB = p0v, dB = -dirv;
            SensorDirectSample sds; sds.valid = false; sds.qx = sds.qy = 0.f; sds.pixel_idx = -1; sds.sensor_val = 0.f;
            RayT<true> camera_ray; camera_ray.o = Vec3d(Dual(0.f)); camera_ray.d = Vec3d(Dual(0.f));
            if (stage == 2) {
                sec_camera_sample<true>(T, cam, p1, sds, camera_ray, -1);
                oB = detach(camera_ray.o); dB = detach(camera_ray.d);
            }
            Hit hA, hB;
            // (the emitter ray only has to know whether its closest hit lies at the sample: any hit clearly in front of it settles that - as the next-event rays of the paths)
            trace2<LDS, COUNT>(S, p0v, dirv, stage == 1, oB, dB, stage != 0, hA, hB, (norm(bss.p2 - p0v) - kShadowEpsilon) * 0.9999f);
            if (lane_id < pool_n) pool_at(lane_id) = spare;
            wave_sync();
            if (stage == 1) {
                stage = 0;
                RayT<false> r2; r2.o = p0v; r2.d = dirv;
                if (COUNT) { if (hA.slot >= 0) S.c_hits++; if (hB.slot >= 0) S.c_hits++; }
                const Its<false> its2 = make_its<false, LDS, true>(S, hA, r2, false);
                if (sec_light_hit_ok(S, its2, bss.p2) && hB.slot >= 0) {
                    RayT<false> r1; r1.o = p0v; r1.d = -dirv;
                    const Its<false> its1c = make_its<false, LDS, true>(S, hB, r1, false);
                    SensorDirectSample s1; RayT<true> c1;
                    if (its1c.valid && sec_camera_sample<true>(T, cam, its1c.p, s1, c1, -1)) { stage = 2; h2 = hA; h1c = hB; p1 = its1c.p; }
                }
            } else if (stage == 2) {
                stage = 0;
                RayT<false> r2; r2.o = p0v; r2.d = dirv;
                RayT<false> r1; r1.o = p0v; r1.d = -dirv;
                if (COUNT) { if (hB.slot >= 0) S.c_hits++; }
                const Its<false> its2 = make_its<false, LDS, true>(S, h2, r2, false);
                const Its<false> its1c = make_its<false, LDS, true>(S, h1c, r1, false);
                const Its<true> its1 = make_its<true, LDS, true>(S, hB, camera_ray, false);
                Vec3f v;
                SecAdjInfo I;
                const int idx = sec_value<true, LDS>(S, bss, its2, its1c, its1, camera_ray, sds, v, ADJ ? &I : nullptr);
                if (idx >= 0) {
                    if constexpr (ADJ) scatter_closed(idx, I, bss, pdf0);
                    else {
                        float o[3] = {v.x, v.y, v.z};
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            if (pdf0 > kEpsilon) o[c] /= pdf0;
                            if (T.sppse > 1) o[c] /= (float) T.sppse;
                            if (finite_(o[c]) && o[c] != 0.f) atomicAdd(&P.dout[3 * (long long) idx + c], o[c]);
                        }
                    }
                }
            }
        }
        if constexpr (ADJ) { S.mode = 0; S.probe_kind = 0; }
    } else
    for (;;) {
        if (pooled && T.n_tris <= kBruteForceMax) {
            // brute-force scenes (round 5): every lane draws in every round, the valid items' indices wait in the pool, a wave's worth is taken at a time
            while (pool_n < 64) {
                refill_queue();
                if (q_next >= q_end) break;
                const long long item = q_next + lane_id;
                bool ok = false;
                if (item < q_end) { BoundarySegSampleDirect tmp; float tpdf; ok = draw(item, tmp, tpdf); }
                const unsigned long long m_ok = __ballot(ok);
                if (ok) pool_at(pool_n + __popcll(m_ok & lt_mask)) = (unsigned) item;
                pool_n += __popcll(m_ok);
                const long long left = q_end - q_next;
                q_next += left < 64 ? left : 64;
            }
            wave_sync();
            const int n_take = pool_n < 64 ? pool_n : 64;
            have = false;
            if (lane_id < n_take) have = draw((long long) pool_at(pool_n - n_take + lane_id), bss, pdf0);
            pool_n -= n_take;
            wave_sync();
            if (n_take == 0) { if (exhausted && q_next >= q_end) break; continue; }
        } else {
        for (int round = 0; round < 16; ++round) {
            const unsigned long long need = __ballot(!have);
            if (__popcll(need) <= 6) break;
            refill_queue();
            if (q_next >= q_end) break;
            const int rank = __popcll(need & lt_mask);
            const long long item = q_next + rank;
            if (!have && item < q_end) have = draw(item, bss, pdf0);
            const int n_need = __popcll(need);
            q_next += n_need < (int) (q_end - q_next) ? n_need : (q_end - q_next);
        }
        if (__ballot(have) == 0ull) { if (exhausted && q_next >= q_end) break; continue; }
        }
        if constexpr (ADJ) if (have) {
            // reverse mode: record the three rays once, then probe the quantities the tangent is linear in
            float *rec = scratch_base<LDS>(smem, T) + threadIdx.x;
            float *g_sec = P.lds_acc ? scratch_base<LDS>(smem, T) + kSecAdjScratch : P.g_sec;
            float *g_tri = P.lds_acc ? g_sec + 6 * P.n_sec : P.g_tri;
            S.rec = rec; S.mode = 1; S.rec_n = 0; S.rec_i = 0; S.probe_kind = 0;
            BoundarySegSampleDirect b0 = bss;
            b0.p0 = promote(detach(bss.p0));
            Vec3f v;
            if (P.sec_closed) {
                S.mode = 0; S.probe_kind = 99; S.probe_id = -1;         // (zero tangents everywhere: only the primal factors are wanted)
                SecAdjInfo I;
                const int idx = eval_boundary_segment<true, LDS, false>(S, cam, b0, v, -1, &I);
                if (idx >= 0) scatter_closed(idx, I, bss, pdf0);
                S.mode = 0; S.probe_kind = 0;
                have = false;
                continue;
            }
            const int idx = eval_boundary_segment<true, LDS, false>(S, cam, b0, v);
            if (idx >= 0) {
                float w3[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float k = P.adj_w[3 * (long long) idx + c];
                    if (pdf0 > kEpsilon) k /= pdf0;
                    if (T.sppse > 1) k /= (float) T.sppse;
                    w3[c] = k;
                }
                S.mode = 2;
                // (probe kind 99 matches nothing: the replays see zero tangents except the one the probe sets - with kind 0 the
                // snapshot's FORWARD tangents of the triangles would leak into the adjoint of the edge point)
                S.probe_kind = 99; S.probe_id = -1;
                auto probe = [&](const BoundarySegSampleDirect &b) -> float {
                    S.rec_i = 0;
                    Vec3f t;
                    const int id = eval_boundary_segment<true, LDS, false>(S, cam, b, t);
                    if (id < 0) return 0.f;
                    float g = 0.f;
                    if (finite_(t.x)) g += w3[0] * t.x;
                    if (finite_(t.y)) g += w3[1] * t.y;
                    if (finite_(t.z)) g += w3[2] * t.z;
                    return g;
                };
                for (int c = 0; c < 3; ++c) {
                    BoundarySegSampleDirect bp = b0;
                    if (c == 0) bp.p0.x.d = 1.f; else if (c == 1) bp.p0.y.d = 1.f; else bp.p0.z.d = 1.f;
                    const float g = probe(bp);
                    if (g != 0.f) { atomicAdd(&g_sec[6 * bss.edge_id + c], g); atomicAdd(&g_sec[6 * bss.edge_id + 3 + c], bss.s1 * g); }
                }
                for (int which = 0; which < 3; which += 2) {       // hit 0: emitter triangle, hit 2: camera-ray triangle
                    const int slot = __float_as_int(rec[4 * which * kBlock]);
                    if (slot < 0) continue;
                    const int orig = __float_as_int(S.ld(T.shade_off + 6 * slot + 3).w);
                    S.probe_kind = 1; S.probe_id = slot;
                    for (int comp = 0; comp < 9; ++comp) {
                        S.probe_comp = comp;
                        const float g = probe(b0);
                        if (g != 0.f) atomicAdd(&g_tri[22 * orig + comp], g);
                    }
                    S.probe_kind = 99;
                }
                if (P.g_cam != nullptr)                            // the camera ray through p1 moves with the pose (path.cpp:214)
                    for (int comp = 0; comp < 12; ++comp) {
                        S.rec_i = 0;
                        Vec3f t;
                        const int id = eval_boundary_segment<true, LDS, false>(S, cam, b0, t, comp);
                        if (id < 0) continue;
                        float g = 0.f;
                        if (finite_(t.x)) g += w3[0] * t.x;
                        if (finite_(t.y)) g += w3[1] * t.y;
                        if (finite_(t.z)) g += w3[2] * t.z;
                        if (g != 0.f) atomicAdd(&acc_cam[comp], g);
                    }
            }
            S.mode = 0; S.probe_kind = 0;
            have = false;
        }
        if (have) {
            Vec3f v;
            const int idx = eval_boundary_segment<true, LDS, COUNT>(S, cam, bss, v);
            if (idx >= 0) {
                float o[3] = {v.x, v.y, v.z};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (pdf0 > kEpsilon) o[c] /= pdf0;
                    if (T.sppse > 1) o[c] /= (float) T.sppse;
                    if (finite_(o[c]) && o[c] != 0.f) atomicAdd(&P.dout[3 * (long long) idx + c], o[c]);
                }
            }
            have = false;
        }
    }
    if constexpr (ADJ) if (P.g_cam != nullptr) {
        __syncthreads();
        if (threadIdx.x < 12 && acc_cam[threadIdx.x] != 0.f) atomicAdd(&P.g_cam[threadIdx.x], acc_cam[threadIdx.x]);
    }
    if constexpr (ADJ) if (P.lds_acc) {
        float *acc = scratch_base<LDS>(smem, T) + kSecAdjScratch;
        __syncthreads();
        for (int i = threadIdx.x; i < 6 * P.n_sec; i += kBlock) if (acc[i] != 0.f) atomicAdd(&P.g_sec[i], acc[i]);
        for (int i = threadIdx.x; i < 22 * T.n_tris; i += kBlock) if (acc[6 * P.n_sec + i] != 0.f) atomicAdd(&P.g_tri[i], acc[6 * P.n_sec + i]);
    } else if (P.n_hot > 0) {
        float *acc = scratch_base<LDS>(smem, T) + kSecAdjScratch;
        __syncthreads();
        for (int i = threadIdx.x; i < 9 * P.n_hot; i += kBlock) if (acc[i] != 0.f) atomicAdd(&P.g_tri[22 * P.hot_inv[i / 9] + i % 9], acc[i]);
    }
    if (COUNT) flush_counters(S, ctr);
}

// guiding grid: PathTracer::preprocess_secondary_edges, reference path.cpp:130-168 (one round per launch)
template <int LDS>
__global__ __launch_bounds__(kBlock) void k_guiding_round(const float4 *__restrict__ blob, const SceneTables T, const SecEdgeTables E,
                                                          const SensorDev cam, const GuidingDev G, const int per_cell, const int seed,
                                                          const int round, float *__restrict__ mass) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneView<LDS> S = make_view<LDS>(blob, T, smem);
    const long long n = (long long) G.num_cells * per_cell;
    const long long n_chunks = (n + kBlock - 1) / kBlock;
    for (long long chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const long long lane = chunk * kBlock + threadIdx.x;
        if (lane < n) {
            const int cell = (int) (lane / per_cell);
            const int c0 = cell / (G.reso[1] * G.reso[2]);
            const int rem = cell - c0 * (G.reso[1] * G.reso[2]);
            const int c1 = rem / G.reso[2], c2 = rem - c1 * G.reso[2];
            LaneRng rng;
            rng.seed((unsigned long long) lane + (unsigned long long) (long long) seed, (unsigned long long) lane, (unsigned long long) (3 * round));
            Vec3f s3;
            s3.x = rng.next_1d(); s3.y = rng.next_1d(); s3.z = rng.next_1d();
            s3 = Vec3f((s3.x + (float) c0) * G.unit[0], (s3.y + (float) c1) * G.unit[1], (s3.z + (float) c2) * G.unit[2]);
            Vec3f v;
            eval_secondary_edge<false, LDS, false>(S, E, cam, s3, v);
            float o[3] = {v.x, v.y, v.z};
#pragma unroll
            for (int c = 0; c < 3; ++c) { if (!finite_(o[c])) o[c] = 0.f; if (per_cell > 1) o[c] /= (float) per_cell; }
            const float m = fmaxf(o[0], fmaxf(o[1], o[2]));
            if (m != 0.f) atomicAdd(&mass[cell], m);
        }
    }
}

// batch closest-hit query (parity aid for the traversal alone)
template <int LDS>
__global__ __launch_bounds__(kBlock) void k_trace(const float4 *__restrict__ blob, const SceneTables T, int n, const float *__restrict__ o,
                                                  const float *__restrict__ d, int *__restrict__ out_tri, float *__restrict__ out_uv, float *__restrict__ out_t,
                                                  int pairs) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneView<LDS> S = make_view<LDS>(blob, T, smem);
    if (pairs) {
        // lane i carries rays 2i and 2i+1 through the two-ray path (trace2) that the path kernels use
        const long long np = ((long long) n + 1) / 2;
        for (long long i = (long long) blockIdx.x * kBlock + threadIdx.x; i < ((np + kBlock - 1) / kBlock) * kBlock; i += (long long) gridDim.x * kBlock) {
            const long long ia = 2 * i, ib = 2 * i + 1;
            const bool actA = ia < n, actB = ib < n;
            const long long ja = actA ? ia : 0, jb = actB ? ib : 0;
            Hit hA, hB;
            trace2<LDS, false>(S, Vec3f(o[3 * ja], o[3 * ja + 1], o[3 * ja + 2]), Vec3f(d[3 * ja], d[3 * ja + 1], d[3 * ja + 2]), actA,
                               Vec3f(o[3 * jb], o[3 * jb + 1], o[3 * jb + 2]), Vec3f(d[3 * jb], d[3 * jb + 1], d[3 * jb + 2]), actB, hA, hB);
            if (actA) { out_tri[ia] = hA.slot >= 0 ? __float_as_int(S.ld(T.trav_off + 3 * hA.slot + 2).y) : -1; out_uv[2 * ia] = hA.u; out_uv[2 * ia + 1] = hA.v; out_t[ia] = hA.t; }
            if (actB) { out_tri[ib] = hB.slot >= 0 ? __float_as_int(S.ld(T.trav_off + 3 * hB.slot + 2).y) : -1; out_uv[2 * ib] = hB.u; out_uv[2 * ib + 1] = hB.v; out_t[ib] = hB.t; }
        }
        return;
    }
    for (long long i = (long long) blockIdx.x * kBlock + threadIdx.x; i < (long long) ((n + kBlock - 1) / kBlock) * kBlock; i += (long long) gridDim.x * kBlock) {
        if (i < n) {
            const Hit h = trace<LDS, false>(S, Vec3f(o[3 * i], o[3 * i + 1], o[3 * i + 2]), Vec3f(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
            int id = -1;
            if (h.slot >= 0) id = __float_as_int(S.ld(T.trav_off + 3 * h.slot + 2).y);
            out_tri[i] = id; out_uv[2 * i] = h.u; out_uv[2 * i + 1] = h.v; out_t[i] = h.t;
        }
    }
}

// Scene::ray_intersect<false> for a batch of rays (the reference exposes it as Scene.unit_ray_intersect, psdr.cpp:404):
// 24 floats per ray - valid, mesh id, t, J, p, n (geometric), sh_frame.s/t/n, wi (local), uv
template <int LDS>
__global__ __launch_bounds__(kBlock) void k_intersect(const float4 *__restrict__ blob, const SceneTables T, int n, const float *__restrict__ o,
                                                      const float *__restrict__ d, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneView<LDS> S = make_view<LDS>(blob, T, smem);
    for (long long i = (long long) blockIdx.x * kBlock + threadIdx.x; i < (long long) ((n + kBlock - 1) / kBlock) * kBlock; i += (long long) gridDim.x * kBlock) {
        if (i < n) {
            RayT<false> r;
            r.o = Vec3f(o[3 * i], o[3 * i + 1], o[3 * i + 2]); r.d = Vec3f(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
            const Hit h = trace<LDS, false>(S, r.o, r.d);
            const Its<false> its = make_its<false, LDS, true>(S, h, r, false);
            float *q = out + 24 * i;
            if (!its.valid) { for (int k = 0; k < 24; ++k) q[k] = 0.f; q[1] = -1.f; continue; }
            // its.uv = bilinear2(uv0, uv1 - uv0, uv2 - uv0, barycentrics), scene.cpp:756-759 (the path kernels keep it only when a texture needs it)
            const float4 c = S.ld(T.shade_off + 6 * its.slot + 4), e = S.ld(T.shade_off + 6 * its.slot + 5);
            const float tu = fma_(c.z - c.x, h.u, fma_(e.x - c.x, h.v, c.x)), tv = fma_(c.w - c.y, h.u, fma_(e.y - c.y, h.v, c.y));
            const float v[24] = {1.f, (float) its.mesh, its.t, its.J, its.p.x, its.p.y, its.p.z, its.n.x, its.n.y, its.n.z,
                                 its.fs.x, its.fs.y, its.fs.z, its.ft.x, its.ft.y, its.ft.z, its.fn.x, its.fn.y, its.fn.z,
                                 its.wi.x, its.wi.y, its.wi.z, tu, tv};
            for (int k = 0; k < 24; ++k) q[k] = v[k];
        }
    }
}

// EnvironmentMap::sample_position / sample_position_pdf alone (parity aids)
#ifndef PSDR_TU        // (plain kernels: the main unit only)
__global__ void k_env_sample(const SceneTables T, int n, const float *__restrict__ ref_p, const float *__restrict__ s2,
                             float *__restrict__ out_p, float *__restrict__ out_n, float *__restrict__ out_pdf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Vec3f p, nn; float pdf;
    env_sample_position(T.env, Vec3f(ref_p[3 * i], ref_p[3 * i + 1], ref_p[3 * i + 2]), s2[2 * i], s2[2 * i + 1], p, nn, pdf);
    out_p[3 * i] = p.x; out_p[3 * i + 1] = p.y; out_p[3 * i + 2] = p.z;
    out_n[3 * i] = nn.x; out_n[3 * i + 1] = nn.y; out_n[3 * i + 2] = nn.z;
    out_pdf[i] = pdf;
}
__global__ void k_env_pdf(const SceneTables T, int n, const float *__restrict__ ref_p, const float *__restrict__ p, const float *__restrict__ nrm,
                          float *__restrict__ out_pdf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_pdf[i] = env_position_pdf(T.env, Vec3f(ref_p[3 * i], ref_p[3 * i + 1], ref_p[3 * i + 2]), Vec3f(p[3 * i], p[3 * i + 1], p[3 * i + 2]),
                                  Vec3f(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]));
}

__global__ void k_sampler_floats(unsigned long long seed_value, unsigned long long lane, unsigned long long skip, int n, float *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        LaneRng r; r.seed(seed_value, lane, skip_ahead(skip));        // (the kernels' form of the skip-ahead: sampler.h)
        for (int i = 0; i < n; ++i) out[i] = r.next_1d();
    }
}
#endif

// ------------------------------------------------------------------------------------------------
// Split build (psdr_jit_amd/build.py): the heavy kernel templates of each scene class are instantiated in translation units of
// their own - this file compiled with -DPSDR_TU=1..8, kernels only - and compiled in parallel; the main unit (-DPSDR_SPLIT: host
// code + the small kernels) declares those instantiations extern.  Without either macro the file is one self-contained unit
// (development builds, -DPSDR_CLS_MASK).
#define PSDR_INST_PATHS(PFX, AD_, C_, CNT_, M_) PFX template __global__ void k_paths<AD_, C_, CNT_, M_>(const float4 *, const SceneTables, const SensorDev, const PathParams, Counters *);
#define PSDR_INST_ADJ(PFX, C_) PFX template __global__ void k_interior_adjoint<C_>(const float4 *, const SceneTables, const SensorDev, const AdjointParams);
#define PSDR_INST_ADJM(PFX, C_) PFX template __global__ void k_interior_adjoint_mat<C_>(const float4 *, const SceneTables, const SensorDev, const AdjointParams);
#define PSDR_INST_SEC(PFX, C_, CNT_, ADJ_) PFX template __global__ void k_secondary_edges<C_, CNT_, ADJ_>(const float4 *, const SceneTables, const SecEdgeTables, const SensorDev, const PathParams, const GuidingDev, const int, Counters *);
#define PSDR_INST_PATHS6(PFX, C_) PSDR_INST_PATHS(PFX, true, C_, false, 0) PSDR_INST_PATHS(PFX, false, C_, false, 0) PSDR_INST_PATHS(PFX, false, C_, false, 1) \
                                  PSDR_INST_PATHS(PFX, true, C_, true, 0) PSDR_INST_PATHS(PFX, false, C_, true, 0) PSDR_INST_PATHS(PFX, false, C_, true, 1)
#define PSDR_TU1(PFX) PSDR_INST_PATHS(PFX, true, 0, false, 0) PSDR_INST_PATHS(PFX, false, 0, false, 0) PSDR_INST_PATHS(PFX, false, 0, false, 1)
#define PSDR_TU6(PFX) PSDR_INST_PATHS(PFX, true, 0, true, 0) PSDR_INST_PATHS(PFX, false, 0, true, 0) PSDR_INST_PATHS(PFX, false, 0, true, 1)
#define PSDR_TU2(PFX) PSDR_INST_ADJ(PFX, 0) PSDR_INST_SEC(PFX, 0, false, false) PSDR_INST_SEC(PFX, 0, true, false) PSDR_INST_SEC(PFX, 0, false, true)
#define PSDR_TU8(PFX) PSDR_INST_ADJM(PFX, 0)         // the material sweep, a unit of its own for the same reason as PSDR_TU7: its not-inlined bsdf_back lambda shows the allocator defect
#define PSDR_TU3(PFX) PSDR_INST_PATHS6(PFX, 1) PSDR_INST_ADJ(PFX, 1) PSDR_INST_SEC(PFX, 1, false, false) PSDR_INST_SEC(PFX, 1, true, false) PSDR_INST_SEC(PFX, 1, false, true)
#define PSDR_TU4(PFX) PSDR_INST_PATHS6(PFX, 2) PSDR_INST_SEC(PFX, 2, false, false) PSDR_INST_SEC(PFX, 2, true, false) PSDR_INST_SEC(PFX, 2, false, true)
#define PSDR_TU7(PFX) PSDR_INST_ADJ(PFX, 2)          // a unit of its own: when the ISA lint sends it to the second allocator (build.py), the class-2 path kernels do not pay for it
#define PSDR_TU5(PFX) PSDR_INST_PATHS(PFX, true, 3, false, 0) PSDR_INST_PATHS(PFX, false, 3, false, 0) PSDR_INST_PATHS(PFX, false, 3, false, 1) PSDR_INST_SEC(PFX, 3, false, false)
#if defined(PSDR_TU)
#if PSDR_TU == 1
PSDR_TU1()
#elif PSDR_TU == 2
PSDR_TU2()
#elif PSDR_TU == 3
PSDR_TU3()
#elif PSDR_TU == 4
PSDR_TU4()
#elif PSDR_TU == 5
PSDR_TU5()
#elif PSDR_TU == 6
PSDR_TU6()
#elif PSDR_TU == 7
PSDR_TU7()
#elif PSDR_TU == 8
PSDR_TU8()
#endif
#else
#if defined(PSDR_SPLIT)
PSDR_TU1(extern) PSDR_TU2(extern) PSDR_TU3(extern) PSDR_TU4(extern) PSDR_TU5(extern) PSDR_TU6(extern) PSDR_TU7(extern) PSDR_TU8(extern)
#endif

// ------------------------------------------------------------------------------------------------
// host side (the scene object, its creation and its updates: scene_obj.h, scene_build.hip)
static thread_local std::string g_err;
namespace psdr { int api_fail(const std::string &msg) { g_err = msg; return 1; } }

struct psdr_hip_guiding {
    GuidingDev G{};
    DevBuf pmf, cmf, guide;
    std::vector<float> mass;
};


extern "C" {

const char *psdr_hip_last_error(void) { return g_err.c_str(); }
int psdr_hip_abi_version(void) { return PSDR_HIP_ABI_VERSION; }
int psdr_hip_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
int psdr_hip_set_device(int device) { HIPCHK(hipSetDevice(device)); return 0; }

} // extern "C"

// number of 256-lane chunks of [0, n) that belong to shard `rank` of `count` (chunk k -> rank k % count)
static inline long long local_lanes(long long n, int rank, int count) {
    const long long chunks = (n + kBlock - 1) / kBlock;
    const long long mine = chunks > rank ? (chunks - rank + count - 1) / count : 0;
    return mine * kBlock;
}
// psdr_render_args.shard_mode == PSDR_SHARD_ROWS: rank `rank` of `count` owns one CONTIGUOUS run of a sampler's lanes, a whole number of `unit` lanes (a pixel row of the
// interior sampler - the rank's pixels are then one block of the image -, a pixel of a batch list, a 256-lane chunk of an edge sampler); the kernels see it as the lane range
// [begin, end) of a single rank, the form the per-lane entry points use
static inline void shard_run(long long n_lanes, long long unit, int rank, int count, long long &begin, long long &end) {
    const long long n_units = (n_lanes + unit - 1) / unit, per = (n_units + count - 1) / count;
    begin = std::min(n_lanes, (long long) rank * per * unit);
    end = std::min(n_lanes, ((long long) rank + 1) * per * unit);
}
template <typename Params> static inline void set_shard(Params &P, const psdr_render_args *a, long long n_lanes, long long unit, int rank, int count) {
    P.begin = 0; P.end = n_lanes; P.shard_rank = rank; P.shard_count = count;
    if (count > 1 && a->shard_mode == PSDR_SHARD_ROWS) { shard_run(n_lanes, unit, rank, count, P.begin, P.end); P.shard_rank = 0; P.shard_count = 1; }
    P.n_local = local_lanes(P.end - P.begin, P.shard_rank, P.shard_count);
}
static inline int grid_for(const psdr_hip_scene *sc, long long n) {
    long long chunks = (n + kBlock - 1) / kBlock;
    if (chunks < 1) chunks = 1;
    return (int) std::min<long long>(chunks, sc->grid);
}

// Development builds can leave scene classes out (hipcc -DPSDR_CLS_MASK=4 compiles the class-2 kernels only: a third of the build
// time); bit c = scene class c (scene_dev.h).  A scene of a class that is compiled out fails loudly.
#ifndef PSDR_CLS_MASK
#define PSDR_CLS_MASK 15
#endif
#if (PSDR_CLS_MASK) & 1
#define ON_CLS0(...) __VA_ARGS__
#else
#define ON_CLS0(...) return fail("scene class 0 is compiled out of this development build")
#endif
#if (PSDR_CLS_MASK) & 2
#define ON_CLS1(...) __VA_ARGS__
#else
#define ON_CLS1(...) return fail("scene class 1 is compiled out of this development build")
#endif
#if (PSDR_CLS_MASK) & 4
#define ON_CLS2(...) __VA_ARGS__
#else
#define ON_CLS2(...) return fail("scene class 2 is compiled out of this development build")
#endif

#if (PSDR_CLS_MASK) & 8
#define ON_CLS3(...) __VA_ARGS__
#else
#define ON_CLS3(...) return fail("scene class 3 is compiled out of this development build")
#endif

#if (PSDR_CLS_MASK) & 1
#define IF_CLS0(...) __VA_ARGS__
#else
#define IF_CLS0(...) (void) 0
#endif
#if (PSDR_CLS_MASK) & 2
#define IF_CLS1(...) __VA_ARGS__
#else
#define IF_CLS1(...) (void) 0
#endif
#if (PSDR_CLS_MASK) & 4
#define IF_CLS2(...) __VA_ARGS__
#else
#define IF_CLS2(...) (void) 0
#endif

// dynamic LDS of a kernel of scene class `cls` (0 global tables, 1 LDS blob, 2 lean BVH, 3 LDS blob with materials): the traversal stack / cold rows,
// plus the blob only for the classes that stage it - a class-0 kernel launched on a scene that ALSO has an LDS class (reverse mode,
// counted runs, ray batches, field integrators on Microfacet / bitmap boxes) must not reserve the blob's bytes it never fills
static size_t smem_for(const psdr_hip_scene *sc, int cls) {
    const size_t blob = (sc->lds || sc->lds_mat) ? (size_t) sc->T.blob_words * 16 : 0;
#ifdef PSDR_DEV_KNOBS
    // (measurement: PSDR_PAD_LDS=bytes asks for more LDS per workgroup than the kernels use - fewer workgroups per CU, to see what occupancy is worth)
    static const size_t pad = std::getenv("PSDR_PAD_LDS") ? (size_t) std::atoll(std::getenv("PSDR_PAD_LDS")) : 0;
    return ((cls == 1 || cls == 3) ? sc->smem_bytes : sc->smem_bytes - blob) + pad;
#else
    return (cls == 1 || cls == 3) ? sc->smem_bytes : sc->smem_bytes - blob;
#endif
}
#define LAUNCH(cls_, kernel, sc, n_lanes, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid_for(sc, n_lanes)), dim3(kBlock), smem_for(sc, cls_), (hipStream_t) (stream), __VA_ARGS__)

static int check_args(const psdr_hip_scene *sc, const psdr_render_args *a) {
    if (!sc || !a) return fail("null argument");
    if (a->sensor_id < 0 || a->sensor_id >= (int) sc->sensors.size()) return fail("Invalid sensor id!");
    if (a->max_depth < 0) return fail("max_depth >= 0");
    if (a->pix_ids && a->n_pix <= 0) return fail("batch rendering needs n_pix > 0");
    // Scene::sample_emitter_position asserts "No Emitter!" (reference scene.cpp:989); the first-hit integrators and
    // PathTracer(0) never sample an emitter, so emitter-less scenes (silhouette / depth rendering) are fine there
    if (sc->T.n_emitters == 0 && a->field_mode == 0 && (a->direct_mode > 0 || a->max_depth > 0)) return fail("No Emitter!");
    return 0;
}

template <bool COUNT>
static int render_impl(const psdr_hip_scene *sc, const psdr_render_args *a, bool ad, float *out, float *dout, float *lanes_out,
                       long long lane_b, long long lane_e, psdr_counters *counters, void *stream) {
    if (check_args(sc, a)) return 1;
    SCRATCH_GUARD(sc, stream);
    const SceneTables &T = sc->T;
    hipStream_t st = (hipStream_t) stream;
    const long long npx = a->pix_ids ? a->n_pix : (long long) T.width * T.height;
    if (a->zero_output) {
        if (out) HIPCHK(hipMemsetAsync(out, 0, sizeof(float) * 3 * npx, st));
        if (dout) HIPCHK(hipMemsetAsync(dout, 0, sizeof(float) * 3 * npx, st));
    }
    Counters *ctr = (Counters *) sc->counters.p;
    if (COUNT) HIPCHK(hipMemsetAsync(ctr, 0, sizeof(Counters), st));
    const SensorDev &cam = sc->sensors[a->sensor_id];
    const int terms = (ad ? (a->terms ? a->terms : 7) : PSDR_TERM_INTERIOR) & ((a->field_mode > 0 || sc->T.n_emitters == 0) ? ~PSDR_TERM_SECONDARY : ~0);   // first-hit integrators have no secondary-edge term
    const int count = a->shard_count > 1 ? a->shard_count : 1;
    const int rank = count > 1 ? a->shard_rank : 0;
    if (rank < 0 || rank >= count) return fail("bad shard rank");
    if (a->shard_mode != PSDR_SHARD_INTERLEAVED && a->shard_mode != PSDR_SHARD_ROWS) return fail("bad shard mode");
    // the first-hit integrators (field_mode) live in the LDS=false instantiations only
    const bool use_lds = sc->lds && a->field_mode == 0;
    // scene class of the kernels (scene_dev.h); class 3 has the plain forward instantiations only (the counted ones run class 0)
    const int cls = use_lds ? 1 : ((sc->lean && a->field_mode == 0) ? 2 : ((sc->lds_mat && a->field_mode == 0 && !COUNT) ? 3 : 0));
    const int fh_field = a->field_mode - 1;
    if (a->field_mode < 0 || a->field_mode > 9) return fail("bad field_mode");

    auto next_queue = [&](unsigned long long *&q) -> int {
        q = (unsigned long long *) sc->queues.p + (sc->queue_slot++ % kQueueRing);
        HIPCHK(hipMemsetAsync(q, 0, sizeof(unsigned long long), st));
        return 0;
    };
    // The three terms of a renderD are independent launches that add into the same two images: each is a grid of persistent workgroups that drains its own queue,
    // and the last workgroups of one launch leave most of the device idle while they finish.  With the edge terms on two side streams of the scene (forked from the
    // caller's stream after the clears, joined before the call returns) the next term's workgroups take the slots as they free up.
    hipStream_t s_prim = st, s_sec = st;
    // (BVH scenes only: config 5 219.3 -> 217.9 ms; the brute-force classes lose - C3 7.09 -> 7.34 ms - because a second kernel's waves beside a VALU-bound one only take issue slots)
    // (PSDR_NO_FORK: measurement / test knob, read per call - the three terms one after the other on the caller's stream, so that a profiler sees each kernel's own
    //  duration (tools/profile.sh) and a test can compare the forked call with the serial one)
    const bool no_fork = std::getenv("PSDR_NO_FORK") != nullptr;
    const bool fork = !no_fork && ad && !a->pix_ids && !lanes_out && !COUNT && T.n_tris > kBruteForceMax && ((terms & (PSDR_TERM_PRIMARY | PSDR_TERM_SECONDARY)) != 0) && (terms & (terms - 1)) != 0;
    unsigned long long *q_int = nullptr, *q_prim = nullptr, *q_sec = nullptr;
    // concurrent launches must not share the global tail of the traversal stack (trav4.h indexes it by workgroup and thread only): the edge terms get slices of their own
    SceneTables T_prim = T, T_sec = T;
    if (fork && T.gstack != nullptr) { T_prim.gstack = T.gstack + sc->gstack_slice; T_sec.gstack = T.gstack + 2 * sc->gstack_slice; }
    if (fork) {
        if (sc->make_term_streams()) return fail("term streams: cannot create");
        if (next_queue(q_int) || next_queue(q_prim) || next_queue(q_sec)) return 1;
        HIPCHK(hipEventRecord(sc->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(sc->aux[0], sc->ev_fork, 0));
        HIPCHK(hipStreamWaitEvent(sc->aux[1], sc->ev_fork, 0));
        s_prim = sc->aux[0]; s_sec = sc->aux[1];
    }
    if ((terms & PSDR_TERM_INTERIOR) && T.spp > 0) {
        PathParams P{};
        P.max_depth = fh_field >= 0 ? 0 : (a->direct_mode > 0 ? 1 : a->max_depth); P.mis = a->direct_mode - 1; P.field = fh_field; P.field_object = a->field_object; P.intensity = a->intensity; P.d_intensity = a->d_intensity; P.hide_emitters = a->hide_emitters; P.seed = a->samplers[0].seed; P.skip = skip_ahead(a->samplers[0].skip);
        P.pix_ids = a->pix_ids;
        set_shard(P, a, npx * T.spp, a->pix_ids ? (long long) T.spp : (long long) T.width * T.spp, rank, count);
        P.out = out; P.dout = dout; P.lanes_out = lanes_out;
        if (lanes_out) { P.begin = lane_b; P.end = lane_e; P.shard_rank = 0; P.shard_count = 1; P.n_local = local_lanes(P.end - P.begin, 0, 1); }
        if (P.n_local > 0) {
            if (fork) P.counter = q_int; else if (next_queue(P.counter)) return 1;
            if (ad) {
                if (cls == 1) ON_CLS1(LAUNCH(1, (k_paths<true, 1, COUNT, 0>), sc, P.n_local, st, sc->blob.as<float4>(), T, cam, P, ctr));
                else if (cls == 2) ON_CLS2(LAUNCH(2, (k_paths<true, 2, COUNT, 0>), sc, P.n_local, st, sc->blob.as<float4>(), T, cam, P, ctr));
                else if (cls == 3) ON_CLS3(LAUNCH(3, (k_paths<true, 3, false, 0>), sc, P.n_local, st, sc->blob.as<float4>(), T, cam, P, ctr));
                else ON_CLS0(LAUNCH(0, (k_paths<true, 0, COUNT, 0>), sc, P.n_local, st, sc->blob.as<float4>(), T, cam, P, ctr));
            } else {
                if (cls == 1) ON_CLS1(LAUNCH(1, (k_paths<false, 1, COUNT, 0>), sc, P.n_local, st, sc->blob.as<float4>(), T, cam, P, ctr));
                else if (cls == 2) ON_CLS2(LAUNCH(2, (k_paths<false, 2, COUNT, 0>), sc, P.n_local, st, sc->blob.as<float4>(), T, cam, P, ctr));
                else if (cls == 3) ON_CLS3(LAUNCH(3, (k_paths<false, 3, false, 0>), sc, P.n_local, st, sc->blob.as<float4>(), T, cam, P, ctr));
                else ON_CLS0(LAUNCH(0, (k_paths<false, 0, COUNT, 0>), sc, P.n_local, st, sc->blob.as<float4>(), T, cam, P, ctr));
            }
        }
    }
    if (ad && !a->pix_ids && !lanes_out) {
        if ((terms & PSDR_TERM_PRIMARY) && T.sppe > 0 && cam.n_edges > 0) {
            PathParams P{};
            P.max_depth = fh_field >= 0 ? 0 : (a->direct_mode > 0 ? 1 : a->max_depth); P.mis = a->direct_mode - 1; P.field = fh_field; P.field_object = a->field_object; P.intensity = a->intensity; P.d_intensity = a->d_intensity; P.hide_emitters = a->hide_emitters; P.seed = a->samplers[1].seed; P.skip = skip_ahead(a->samplers[1].skip);
            set_shard(P, a, npx * T.sppe, kBlock, rank, count); P.dout = dout;
            P.skip_static = a->skip_static_edges;
            if (P.n_local > 0) {
                if (fork) P.counter = q_prim; else if (next_queue(P.counter)) return 1;
                if (cls == 1) ON_CLS1(LAUNCH(1, (k_paths<false, 1, COUNT, 1>), sc, P.n_local, s_prim, sc->blob.as<float4>(), T_prim, cam, P, ctr));
                else if (cls == 2) ON_CLS2(LAUNCH(2, (k_paths<false, 2, COUNT, 1>), sc, P.n_local, s_prim, sc->blob.as<float4>(), T_prim, cam, P, ctr));
                else if (cls == 3) ON_CLS3(LAUNCH(3, (k_paths<false, 3, false, 1>), sc, P.n_local, s_prim, sc->blob.as<float4>(), T_prim, cam, P, ctr));
                else ON_CLS0(LAUNCH(0, (k_paths<false, 0, COUNT, 1>), sc, P.n_local, s_prim, sc->blob.as<float4>(), T_prim, cam, P, ctr));
            }
        }
        if ((terms & PSDR_TERM_SECONDARY) && T.sppse > 0 && sc->E.n > 0) {
            PathParams P{};
            P.max_depth = fh_field >= 0 ? 0 : (a->direct_mode > 0 ? 1 : a->max_depth); P.mis = a->direct_mode - 1; P.field = fh_field; P.field_object = a->field_object; P.intensity = a->intensity; P.d_intensity = a->d_intensity; P.hide_emitters = a->hide_emitters; P.seed = a->samplers[2].seed; P.skip = skip_ahead(a->samplers[2].skip);
            set_shard(P, a, npx * T.sppse, kBlock, rank, count); P.dout = dout;
            GuidingDev G{};
            const int use_g = a->guiding ? 1 : 0;
            if (a->guiding) G = a->guiding->G;
            if (P.n_local > 0) {
                if (fork) P.counter = q_sec; else if (next_queue(P.counter)) return 1;
                if (sc->lds) ON_CLS1(LAUNCH(1, (k_secondary_edges<1, COUNT, false>), sc, P.n_local, s_sec, sc->blob.as<float4>(), T_sec, sc->E, cam, P, G, use_g, ctr));
                else if (sc->lean) ON_CLS2(LAUNCH(2, (k_secondary_edges<2, COUNT, false>), sc, P.n_local, s_sec, sc->blob.as<float4>(), T_sec, sc->E, cam, P, G, use_g, ctr));
                else if (cls == 3) ON_CLS3(LAUNCH(3, (k_secondary_edges<3, false, false>), sc, P.n_local, s_sec, sc->blob.as<float4>(), T_sec, sc->E, cam, P, G, use_g, ctr));
                else ON_CLS0(LAUNCH(0, (k_secondary_edges<0, COUNT, false>), sc, P.n_local, s_sec, sc->blob.as<float4>(), T_sec, sc->E, cam, P, G, use_g, ctr));
            }
        }
    }
    if (fork) {
        HIPCHK(hipEventRecord(sc->ev_join[0], sc->aux[0]));
        HIPCHK(hipEventRecord(sc->ev_join[1], sc->aux[1]));
        HIPCHK(hipStreamWaitEvent(st, sc->ev_join[0], 0));
        HIPCHK(hipStreamWaitEvent(st, sc->ev_join[1], 0));
    }
    HIPCHK(hipGetLastError());
    if (COUNT && counters) {
        Counters h;
        HIPCHK(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        counters->rays = h.rays; counters->nodes_visited = h.nodes; counters->tris_tested = h.tris; counters->shaded_hits = h.hits;
    }
    return 0;
}

extern "C" {

int psdr_hip_render_c(const psdr_hip_scene *sc, const psdr_render_args *a, float *out, void *stream) {
    if (!out) return fail("null output");
    return render_impl<false>(sc, a, false, out, nullptr, nullptr, 0, 0, nullptr, stream);
}
int psdr_hip_render_d_fwd(const psdr_hip_scene *sc, const psdr_render_args *a, float *out, float *dout, void *stream) {
    if (!out || !dout) return fail("null output");
    return render_impl<false>(sc, a, true, out, dout, nullptr, 0, 0, nullptr, stream);
}
int psdr_hip_render_c_counted(const psdr_hip_scene *sc, const psdr_render_args *a, float *out, psdr_counters *c, void *stream) {
    if (!out) return fail("null output");
    return render_impl<true>(sc, a, false, out, nullptr, nullptr, 0, 0, c, stream);
}
int psdr_hip_render_d_fwd_counted(const psdr_hip_scene *sc, const psdr_render_args *a, float *out, float *dout, psdr_counters *c, void *stream) {
    if (!out || !dout) return fail("null output");
    return render_impl<true>(sc, a, true, out, dout, nullptr, 0, 0, c, stream);
}
int psdr_hip_li_lanes(const psdr_hip_scene *sc, const psdr_render_args *a, int64_t lane_begin, int64_t lane_end, float *out, void *stream) {
    if (!out || lane_end <= lane_begin) return fail("bad lane range");
    return render_impl<false>(sc, a, false, nullptr, nullptr, out, lane_begin, lane_end, nullptr, stream);
}

int psdr_hip_scene_tex_layout(const psdr_hip_scene *sc, int64_t *offsets, int64_t *total) {
    if (!sc) return fail("null scene");
    if (total) *total = sc->tex_total;
    if (offsets)
        for (int i = 0; i < 3 * sc->T.n_bsdfs; ++i) offsets[i] = (size_t) i < sc->tex_layout.size() ? sc->tex_layout[i] : -1;
    return 0;
}

int psdr_hip_render_d_bwd(const psdr_hip_scene *sc, const psdr_render_args *a, const float *d_rgb, const psdr_grads *g, void *stream) {
    if (check_args(sc, a)) return 1;
    if (!d_rgb || !g || !g->g_triangles || !g->g_bsdf || !g->g_emitter) return fail("null gradient buffer");
    SCRATCH_GUARD(sc, stream);
    // batch rendering (integrator.cpp:139-176): d_rgb is [n_pix*3]; only the interior term exists for a pixel list (as in the forward path)
    const SceneTables &T = sc->T;
    hipStream_t st = (hipStream_t) stream;
    const long long npx_full = (long long) T.width * T.height;
    const long long npx = a->pix_ids ? a->n_pix : npx_full;
    const int count = a->shard_count > 1 ? a->shard_count : 1;
    const int rank = count > 1 ? a->shard_rank : 0;
    if (rank < 0 || rank >= count) return fail("bad shard rank");
    if (a->shard_mode != PSDR_SHARD_INTERLEAVED && a->shard_mode != PSDR_SHARD_ROWS) return fail("bad shard mode");
    // the first-hit integrators (field_mode) live in the LDS=false instantiations only
    const bool use_lds = sc->lds && a->field_mode == 0;
    const int cls = use_lds ? 1 : ((sc->lean && a->field_mode == 0) ? 2 : 0);        // scene class of the kernels (scene_dev.h); the reverse mode has no class 3
    const int fh_field = a->field_mode - 1;
    if (a->field_mode < 0 || a->field_mode > 9) return fail("bad field_mode");
    SensorDev cam = sc->sensors[a->sensor_id];
    for (int i = 0; i < 16; ++i) { cam.d_to_world.m[i] = 0.f; cam.d_world_to_sample.m[i] = 0.f; }     // probes only
    const int terms = (a->terms ? a->terms : 7) & ((a->field_mode > 0 || sc->T.n_emitters == 0) ? ~PSDR_TERM_SECONDARY : ~0);
    if (a->zero_output) {
        HIPCHK(hipMemsetAsync(g->g_triangles, 0, sizeof(float) * 22 * (size_t) T.n_tris, st));
        HIPCHK(hipMemsetAsync(g->g_bsdf, 0, sizeof(float) * 3 * (size_t) std::max(1, T.n_bsdfs), st));
        HIPCHK(hipMemsetAsync(g->g_emitter, 0, sizeof(float) * 3 * (size_t) std::max(1, T.n_emitters), st));
        if (g->g_tex && sc->tex_total > 0) HIPCHK(hipMemsetAsync(g->g_tex, 0, sizeof(float) * (size_t) sc->tex_total, st));
        if (g->g_camera) HIPCHK(hipMemsetAsync(g->g_camera, 0, sizeof(float) * 16, st));
        if (g->g_env && T.env_emitter >= 0) HIPCHK(hipMemsetAsync(g->g_env, 0, sizeof(float) * 3 * (size_t) T.env.width * T.env.height, st));
        if (g->g_env_scale) HIPCHK(hipMemsetAsync(g->g_env_scale, 0, sizeof(float), st));
        if (g->g_env_from_world) HIPCHK(hipMemsetAsync(g->g_env_from_world, 0, sizeof(float) * 16, st));
        if (g->g_mat) HIPCHK(hipMemsetAsync(g->g_mat, 0, sizeof(float) * kMatOut * (size_t) std::max(1, T.n_bsdfs), st));
        if (g->g_uv_xf) HIPCHK(hipMemsetAsync(g->g_uv_xf, 0, sizeof(float) * 4 * (3 * (size_t) T.n_bsdfs + 1), st));
        if (g->g_sec_edges && sc->E.n > 0) HIPCHK(hipMemsetAsync(g->g_sec_edges, 0, sizeof(float) * 6 * (size_t) sc->E.n, st));
        if (g->g_prim_edges && cam.n_edges > 0) HIPCHK(hipMemsetAsync(g->g_prim_edges, 0, sizeof(float) * 4 * (size_t) cam.n_edges, st));
    }
    auto next_queue = [&](unsigned long long *&q) -> int {
        q = (unsigned long long *) sc->queues.p + (sc->queue_slot++ % kQueueRing);
        HIPCHK(hipMemsetAsync(q, 0, sizeof(unsigned long long), st));
        return 0;
    };
    const bool lds_acc = true;
    const bool with_lookups = T.tex != nullptr || T.pv != nullptr || T.env_emitter >= 0;
    // PSDR_ADJ_GLOBAL=1: run the interior adjoint of an LDS-class scene from global memory (no blob copy in LDS: more workgroups per CU)
#ifdef PSDR_DEV_KNOBS
    static const bool adj_global = std::getenv("PSDR_ADJ_GLOBAL") != nullptr;
#else
    constexpr bool adj_global = false;
#endif
    const int adj_cls = (use_lds && !adj_global) ? 1 : ((sc->lean || use_lds) && a->field_mode == 0 ? 2 : 0);
    // LDS: [blob (class 1)] [stacks] [per-lane records] [camera / env / material accumulators] [hot triangle rows, colours, emitters];
    // the number of hot triangle rows is what is left of the 160 KB
    // (the cold path-state rows of brute-force scenes belong to run_paths, i.e. to k_paths: the adjoint kernels get tables without them)
#ifdef PSDR_NO_ASYNC
    const int cold_rows = kColdRows;
#else
    const int cold_rows = T.n_tris > kBruteForceMax ? 0 : kColdRows;
#endif
    const size_t cold_bytes = (size_t) cold_rows * kBlock * sizeof(int);
    SceneTables Ta = T;
    Ta.stack_depth -= cold_rows;
    const size_t smem_base = (adj_cls == 1 ? sc->smem_bytes : sc->smem_bytes - ((sc->lds || sc->lds_mat) ? (size_t) T.blob_words * 16 : 0)) - cold_bytes;
    const int adj_depth = fh_field >= 0 ? 0 : (a->direct_mode > 0 ? 1 : a->max_depth);
    // Diffuse BSDFs + area lights / an environment map under PathTracer: the reverse sweep (adjoint.h); everything else: record and probe
    const bool no_sweep = std::getenv("PSDR_ADJ_PROBE") != nullptr;                // measurement / test knob, read per call: force the probe form
    const bool sweep = !no_sweep && adj_cls != 0 && a->field_mode == 0 && T.tex == nullptr && T.pv == nullptr && (T.env_emitter < 0 || adj_cls == 2);
    // GGX scenes (class 0): the material sweep, when every BSDF is Diffuse or a constant-parameter Microfacet
    // ... and the first-hit integrators on such scenes (the sweep's camera-hit block with the integrator's own adjoint; field 0 and 7 are constants)
    // the record-and-probe form fills g_uv_xf only while it visits a bitmap / environment lookup for g_tex / g_env: a caller that wants the uv-transform adjoints
    // without those buffers would get zeros from this form and numbers from the sweeps - refuse instead
    if (g->g_uv_xf && (no_sweep || a->field_mode == 0) && !sweep) {
        const bool sweep_mat_ = !no_sweep && adj_cls == 0 && sc->simple_mats;
        if (!sweep_mat_ && ((sc->tex_total > 0 && !g->g_tex) || (T.env_emitter >= 0 && !g->g_env)))
            return fail("g_uv_xf in the record-and-probe form needs g_tex (BSDF bitmaps) / g_env (environment map) beside it");
    }
    const bool sweep_mat = !no_sweep && !sweep && adj_cls == 0 && sc->simple_mats && (a->field_mode > 0 || T.mat != nullptr || T.tex != nullptr || T.pv != nullptr || sc->has_nmap);
    const int lane_words = (sweep || sweep_mat) ? adj_sweep_words(adj_depth) : adj_lane_words(adj_depth, with_lookups);
    // the per-lane records (hits, light samples, lookups of one path: 14 D + 3 words for the sweep) live in LDS when they fit beside
    // the accumulators, else in a global array of the scene (any depth works, at global-memory latency)
    // a small environment map (<= 32 KB of texel adjoints, e.g. 64 x 32) accumulates in LDS: all samples of a wave look up the same few
    // texels, and same-address atomics in global memory serialise (sweeps only)
    size_t env_lds = 0;
    if ((sweep || sweep_mat) && T.env_emitter >= 0 && g->g_env != nullptr && (size_t) T.env.width * T.env.height * 3 * sizeof(float) <= 32 * 1024)
        env_lds = (size_t) T.env.width * T.env.height * 3;
    const size_t mat_row = g->g_uv_xf ? kMatRow : kMatOut;      // (the uv transforms' twelve floats per BSDF only when they are wanted)
    const size_t acc_fixed = sizeof(float) * ((size_t) kAdjMisc + (size_t) T.n_bsdfs * mat_row + (size_t) T.n_bsdfs * 3 + (size_t) T.n_emitters * 3 + env_lds);
    // ... unless they are what keeps a SECOND workgroup off the CU: the kernels need <= 256 registers (two waves per SIMD fit), and one wave per SIMD cannot hide the
    // latency of the sweep's traces (config 5, depth 3: 40 KB of traversal rows + 45 KB of records; round 4 measured this with a 314-register kernel, where it could not help)
    const size_t rec_bytes = sizeof(float) * (size_t) lane_words * kBlock, min_hot = 64 * 22 * sizeof(float);
    const bool two_wg_with_rec = smem_base + rec_bytes + acc_fixed + min_hot <= 80 * 1024, two_wg_without = smem_base + acc_fixed + min_hot <= 80 * 1024;
    const bool rec_in_lds = (two_wg_with_rec || !two_wg_without || PSDR_ADJ_REC_LDS) && smem_base + rec_bytes + acc_fixed + min_hot <= 160 * 1024;
    const size_t fixed_bytes = acc_fixed + (rec_in_lds ? sizeof(float) * (size_t) lane_words * kBlock : 0);
    if (smem_base + fixed_bytes > 160 * 1024) return fail("scene too large for the adjoint kernel's LDS accumulators");
    // two workgroups per CU (80 KB each) when the fixed part allows it - one wave per SIMD cannot hide the global-memory latency of
    // the replays -, with at least 64 hot rows (emitters + the largest triangles)
    const size_t budget = (smem_base + fixed_bytes + 64 * 22 * sizeof(float) <= 80 * 1024) ? 80 * 1024 : 160 * 1024;
    const int n_hot_used = (int) std::min<size_t>((size_t) sc->n_hot, (budget - smem_base - fixed_bytes) / (22 * sizeof(float)));
    const size_t n_acc = (size_t) n_hot_used * 22 + (size_t) T.n_bsdfs * 3 + (size_t) T.n_emitters * 3 + env_lds;
    const size_t adj_bytes = sizeof(float) * ((rec_in_lds ? (size_t) lane_words * kBlock : 0) + kAdjMisc + (size_t) T.n_bsdfs * mat_row + (lds_acc ? n_acc : 0));
    const size_t smem = smem_base + adj_bytes;
    if (smem > 160 * 1024) return fail("scene too large for the adjoint kernel's LDS records");
    if (!sc->adj_attr_set) {         // (per scene = per device and context; a process-wide flag would skip the second device)
        IF_CLS1(HIPCHK(hipFuncSetAttribute((const void *) k_interior_adjoint<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        IF_CLS0(HIPCHK(hipFuncSetAttribute((const void *) k_interior_adjoint<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        IF_CLS0(HIPCHK(hipFuncSetAttribute((const void *) k_interior_adjoint_mat<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        IF_CLS2(HIPCHK(hipFuncSetAttribute((const void *) k_interior_adjoint<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        IF_CLS1(HIPCHK(hipFuncSetAttribute((const void *) k_secondary_edges<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        IF_CLS0(HIPCHK(hipFuncSetAttribute((const void *) k_secondary_edges<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        IF_CLS2(HIPCHK(hipFuncSetAttribute((const void *) k_secondary_edges<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        sc->adj_attr_set = true;
    }
    if ((terms & PSDR_TERM_INTERIOR) && T.spp > 0) {
        AdjointParams P{};
        P.max_depth = fh_field >= 0 ? 0 : (a->direct_mode > 0 ? 1 : a->max_depth); P.mis = a->direct_mode - 1; P.field = fh_field; P.field_object = a->field_object; P.intensity = a->intensity; P.d_intensity = a->d_intensity; P.hide_emitters = a->hide_emitters; P.seed = a->samplers[0].seed; P.skip = skip_ahead(a->samplers[0].skip);
        P.pix_ids = a->pix_ids; set_shard(P, a, npx * T.spp, a->pix_ids ? (long long) T.spp : (long long) T.width * T.spp, rank, count);
        P.w = d_rgb; P.g_tri = g->g_triangles; P.g_bsdf = g->g_bsdf; P.g_emitter = g->g_emitter; P.hot_map = sc->hot_map.as<int>(); P.hot_inv = sc->hot_inv.as<int>(); P.n_hot = n_hot_used;
        P.mesh_filter = g->mesh_filter; P.skip_bsdf = g->skip_bsdf; P.skip_emitter = g->skip_emitter;
        P.g_tex = sc->tex_total > 0 ? g->g_tex : nullptr;
#ifdef PSDR_SWEEP_DUMP
        P.g_tex = g->g_tex;        // (diagnostic build: the sweep's per-lane dump goes there)
#endif
        P.g_cam = g->g_camera;
        P.g_mat = sc->T.mat != nullptr ? g->g_mat : nullptr;
        P.g_env_xf = sc->T.env_emitter >= 0 ? g->g_env_from_world : nullptr;
        P.g_uv_xf = g->g_uv_xf;
        P.g_env = sc->T.env_emitter >= 0 ? g->g_env : nullptr; P.g_env_scale = sc->T.env_emitter >= 0 ? g->g_env_scale : nullptr;
        if (P.n_local > 0) {
            if (next_queue(P.counter)) return 1;
            const int grid = grid_for(sc, P.n_local);
            P.hit_words = adj_hit_words(adj_depth); P.ext_words = adj_ext_words(adj_depth); P.lk_words = with_lookups ? 3 * adj_lk_entries(adj_depth) : 0;
            P.sweep = sweep ? 1 : (sweep_mat ? 2 : 0);
            if (sweep || sweep_mat) { P.hit_words = lane_words; P.ext_words = 0; P.lk_words = 0; }
            P.env_lds = (int) env_lds;
            P.rec_global = nullptr;
            if (!rec_in_lds) {
                const size_t need = sizeof(float) * (size_t) grid * (size_t) lane_words * kBlock;
                if (need > sc->adj_rec_bytes) {
                    // (every earlier user of the records was ordered before this stream by the guard: its completion is this stream's)
                    if (sc->adj_rec.p) { HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipFree(sc->adj_rec.p)); sc->adj_rec.p = nullptr; sc->adj_rec_bytes = 0; }
                    if (sc->adj_rec.upload(nullptr, need)) return 1;
                    sc->adj_rec_bytes = need;
                }
                P.rec_global = (float *) sc->adj_rec.p;
            }
            if (adj_cls == 1) ON_CLS1(hipLaunchKernelGGL((k_interior_adjoint<1>), dim3(grid), dim3(kBlock), smem, st, sc->blob.as<float4>(), Ta, cam, P));
            else if (adj_cls == 2) ON_CLS2(hipLaunchKernelGGL((k_interior_adjoint<2>), dim3(grid), dim3(kBlock), smem, st, sc->blob.as<float4>(), Ta, cam, P));
            else if (sweep_mat) ON_CLS0(hipLaunchKernelGGL((k_interior_adjoint_mat<0>), dim3(grid), dim3(kBlock), smem, st, sc->blob.as<float4>(), Ta, cam, P));
            else ON_CLS0(hipLaunchKernelGGL((k_interior_adjoint<0>), dim3(grid), dim3(kBlock), smem, st, sc->blob.as<float4>(), Ta, cam, P));
        }
    }
    if (!a->pix_ids && (terms & PSDR_TERM_PRIMARY) && T.sppe > 0 && cam.n_edges > 0) {
        if (!g->g_prim_edges) return fail("g_prim_edges is required when the primary-edge term is requested");
        PathParams P{};
        P.max_depth = fh_field >= 0 ? 0 : (a->direct_mode > 0 ? 1 : a->max_depth); P.mis = a->direct_mode - 1; P.field = fh_field; P.field_object = a->field_object; P.intensity = a->intensity; P.d_intensity = a->d_intensity; P.hide_emitters = a->hide_emitters; P.seed = a->samplers[1].seed; P.skip = skip_ahead(a->samplers[1].skip);
        set_shard(P, a, npx_full * T.sppe, kBlock, rank, count);
        P.adj_w = d_rgb; P.g_prim = g->g_prim_edges; P.n_prim = cam.n_edges; P.lds_acc = (cam.n_edges <= 2048) ? 1 : 0;
        P.prim_filter = g->prim_edge_filter;
        if (P.n_local > 0) {
            if (next_queue(P.counter)) return 1;
            const int grid = grid_for(sc, P.n_local);
            const size_t sm = smem_for(sc, cls) + (P.lds_acc ? sizeof(float) * 4 * (size_t) cam.n_edges : 0);
            if (cls == 1) ON_CLS1(hipLaunchKernelGGL((k_paths<false, 1, false, 1>), dim3(grid), dim3(kBlock), sm, st, sc->blob.as<float4>(), T, cam, P, (Counters *) nullptr));
            else if (cls == 2) ON_CLS2(hipLaunchKernelGGL((k_paths<false, 2, false, 1>), dim3(grid), dim3(kBlock), sm, st, sc->blob.as<float4>(), T, cam, P, (Counters *) nullptr));
            else ON_CLS0(hipLaunchKernelGGL((k_paths<false, 0, false, 1>), dim3(grid), dim3(kBlock), sm, st, sc->blob.as<float4>(), T, cam, P, (Counters *) nullptr));
        }
    }
    if (!a->pix_ids && (terms & PSDR_TERM_SECONDARY) && T.sppse > 0 && sc->E.n > 0) {
        if (!g->g_sec_edges) return fail("g_sec_edges is required when the secondary-edge term is requested");
        PathParams P{};
        P.max_depth = fh_field >= 0 ? 0 : (a->direct_mode > 0 ? 1 : a->max_depth); P.mis = a->direct_mode - 1; P.field = fh_field; P.field_object = a->field_object; P.intensity = a->intensity; P.d_intensity = a->d_intensity; P.hide_emitters = a->hide_emitters; P.seed = a->samplers[2].seed; P.skip = skip_ahead(a->samplers[2].skip);
        set_shard(P, a, npx_full * T.sppse, kBlock, rank, count);
        P.adj_w = d_rgb; P.g_sec = g->g_sec_edges; P.g_tri = g->g_triangles; P.n_sec = sc->E.n;
        P.g_cam = g->g_camera;
        P.sec_closed = no_sweep ? 0 : 1;
        const size_t sec_acc = sizeof(float) * (6 * (size_t) sc->E.n + 22 * (size_t) T.n_tris);
        P.lds_acc = (sec_acc <= 48 * 1024) ? 1 : 0;
        // tables too large for LDS: the hot triangle rows still accumulate there (closed form only: it writes p0, e1, e2 = 9 floats per row)
        constexpr int kSecHotMax = PSDR_SEC_HOT_MAX;
        P.n_hot = (!P.lds_acc && P.sec_closed) ? std::min(sc->n_hot, kSecHotMax) : 0;
        P.hot_map = sc->hot_map.as<int>(); P.hot_inv = sc->hot_inv.as<int>();
        const size_t smem_sec = smem_for(sc, sc->lds ? 1 : 0) - cold_bytes + sizeof(float) * (size_t) kSecAdjScratch + (P.lds_acc ? sec_acc : sizeof(float) * 9 * (size_t) P.n_hot);
        GuidingDev G{};
        const int use_g = a->guiding ? 1 : 0;
        if (a->guiding) G = a->guiding->G;
        if (P.n_local > 0) {
            if (next_queue(P.counter)) return 1;
            const int grid = grid_for(sc, P.n_local);
            if (sc->lds) ON_CLS1(hipLaunchKernelGGL((k_secondary_edges<true, false, true>), dim3(grid), dim3(kBlock), smem_sec, st, sc->blob.as<float4>(), Ta, sc->E, cam, P, G, use_g, (Counters *) nullptr));
            else if (sc->lean && a->field_mode == 0) ON_CLS2(hipLaunchKernelGGL((k_secondary_edges<2, false, true>), dim3(grid), dim3(kBlock), smem_sec, st, sc->blob.as<float4>(), Ta, sc->E, cam, P, G, use_g, (Counters *) nullptr));      // (the lean instantiation, as the forward pass)
            else ON_CLS0(hipLaunchKernelGGL((k_secondary_edges<false, false, true>), dim3(grid), dim3(kBlock), smem_sec, st, sc->blob.as<float4>(), Ta, sc->E, cam, P, G, use_g, (Counters *) nullptr));
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static int trace_impl(const psdr_hip_scene *sc, int32_t n, const float *o, const float *d, int32_t *out_tri, float *out_uv, float *out_t, void *stream, int pairs) {
    if (!sc) return fail("null scene");
    if (n <= 0) return 0;
    SCRATCH_GUARD(sc, stream);
    if (sc->lds) ON_CLS1(LAUNCH(1, (k_trace<true>), sc, (long long) n, stream, sc->blob.as<float4>(), sc->T, n, o, d, out_tri, out_uv, out_t, pairs));
    else LAUNCH(0, (k_trace<false>), sc, (long long) n, stream, sc->blob.as<float4>(), sc->T, n, o, d, out_tri, out_uv, out_t, pairs);
    HIPCHK(hipGetLastError());
    return 0;
}
int psdr_hip_env_sample(const psdr_hip_scene *sc, int32_t n, const float *ref_p, const float *s2, float *out_p, float *out_n, float *out_pdf, void *stream) {
    if (!sc) return fail("null scene");
    if (sc->T.env_emitter < 0) return fail("the scene has no EnvironmentMap");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_env_sample, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t) stream, sc->T, n, ref_p, s2, out_p, out_n, out_pdf);
    HIPCHK(hipGetLastError());
    return 0;
}
int psdr_hip_env_pdf(const psdr_hip_scene *sc, int32_t n, const float *ref_p, const float *p, const float *nrm, float *out_pdf, void *stream) {
    if (!sc) return fail("null scene");
    if (sc->T.env_emitter < 0) return fail("the scene has no EnvironmentMap");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_env_pdf, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t) stream, sc->T, n, ref_p, p, nrm, out_pdf);
    HIPCHK(hipGetLastError());
    return 0;
}
// cell masses of the environment map's sampling distribution: one thread per cell, the same envmath.h::cell_mass the host half's
// test hook evaluates (bit-equal); 2 M cells of a 1024 x 512 map take ~0.1 ms instead of 80 ms on the host cores
__global__ void k_env_cell_mass(const float *__restrict__ texels, int W, int H, int w2, int h2, int n, float *__restrict__ mass, env::UvXf<float> xf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mass[i] = env::cell_mass(texels, W, H, w2, h2, i, xf);
}
int psdr_hip_env_cell_masses(const float *texels, int32_t width, int32_t height, float *mass) {
    return psdr_hip_env_cell_masses_xf(texels, width, height, nullptr, mass);
}
int psdr_hip_env_cell_masses_xf(const float *texels, int32_t width, int32_t height, const float *uv_xf, float *mass) {
    if (!texels || !mass) return fail("null texel / mass buffer");
    if (width < 2 || height < 2) return fail("EnvironmentMap: the bitmap needs at least 2 x 2 texels");
    const int w2 = (width - 1) << 1, h2 = (height - 1) << 1;
    const long long n = (long long) w2 * h2;
    if (n > 0x7fffffffll) return fail("EnvironmentMap: too many cells");
    DevBuf tex, out;
    if (tex.upload(texels, sizeof(float) * 3 * (size_t) width * height) || out.upload(nullptr, sizeof(float) * (size_t) n)) return 1;
    hipLaunchKernelGGL(k_env_cell_mass, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) nullptr, tex.as<float>(), width, height, w2, h2, (int) n, (float *) out.p, uv_xf ? env::UvXf<float>(uv_xf) : env::UvXf<float>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(mass, out.p, sizeof(float) * (size_t) n, hipMemcpyDeviceToHost));
    return 0;
}
int psdr_hip_ray_intersect(const psdr_hip_scene *sc, int32_t n, const float *o, const float *d, float *out, void *stream) {
    if (!sc) return fail("null scene");
    if (n <= 0) return 0;
    if (!o || !d || !out) return fail("null ray / output buffer");
    SCRATCH_GUARD(sc, stream);
    if (sc->lds) LAUNCH(1, (k_intersect<true>), sc, (long long) n, stream, sc->blob.as<float4>(), sc->T, n, o, d, out);
    else LAUNCH(0, (k_intersect<false>), sc, (long long) n, stream, sc->blob.as<float4>(), sc->T, n, o, d, out);
    HIPCHK(hipGetLastError());
    return 0;
}
int psdr_hip_trace(const psdr_hip_scene *sc, int32_t n, const float *o, const float *d, int32_t *out_tri, float *out_uv, float *out_t, void *stream) {
    return trace_impl(sc, n, o, d, out_tri, out_uv, out_t, stream, 0);
}
int psdr_hip_trace_pairs(const psdr_hip_scene *sc, int32_t n, const float *o, const float *d, int32_t *out_tri, float *out_uv, float *out_t, void *stream) {
    return trace_impl(sc, n, o, d, out_tri, out_uv, out_t, stream, 1);
}

int psdr_hip_guiding_build(const psdr_hip_scene *sc, int32_t sensor_id, int32_t max_depth, const int32_t reso[4], int32_t nrounds, int32_t seed,
                           psdr_hip_guiding **out, void *stream) {
    (void) max_depth;
    if (!sc || !out || !reso) return fail("null argument");
    if (nrounds <= 0) return fail("nrounds > 0");
    if (sensor_id < 0 || sensor_id >= (int) sc->sensors.size()) return fail("Invalid sensor id!");
    if (sc->E.n <= 0) return fail("Scene needs to be configured with sppse > 0!");
    const long long cells = (long long) reso[0] * reso[1] * reso[2];
    if (cells <= 0 || reso[3] <= 0 || cells * reso[3] > 2147483647LL) return fail("bad guiding resolution");
    auto g = std::make_unique<psdr_hip_guiding>();
    GuidingDev &G = g->G;
    for (int k = 0; k < 3; ++k) { G.reso[k] = reso[k]; G.unit[k] = 1.f / (float) reso[k]; }
    G.num_cells = (int) cells;
    SCRATCH_GUARD(sc, stream);
    hipStream_t st = (hipStream_t) stream;
    DevBuf mass;
    if (mass.upload(nullptr, sizeof(float) * cells)) return 1;
    const long long nl = cells * reso[3];
    for (int r = 0; r < nrounds; ++r) {
        if (sc->lds) ON_CLS1(LAUNCH(1, (k_guiding_round<true>), sc, nl, st, sc->blob.as<float4>(), sc->T, sc->E, sc->sensors[sensor_id], G, reso[3], seed, r, (float *) mass.p));
        else LAUNCH(0, (k_guiding_round<false>), sc, nl, st, sc->blob.as<float4>(), sc->T, sc->E, sc->sensors[sensor_id], G, reso[3], seed, r, (float *) mass.p);      // (a plain kernel of the main unit: present in every development build)
    }
    HIPCHK(hipGetLastError());
    g->mass.resize(cells);
    HIPCHK(hipMemcpyAsync(g->mass.data(), mass.p, sizeof(float) * cells, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // DiscreteDistribution::init on the host (double-precision CDF, reference pmf.h:12-38)
    std::vector<float> cmf(cells);
    float sum = 0.f; double acc = 0.0;
    for (long long i = 0; i < cells; ++i) {
        if (nrounds > 1) g->mass[i] /= (float) nrounds;
        sum += g->mass[i]; acc += (double) g->mass[i]; cmf[i] = (float) acc;
    }
    G.sum = sum;
    if (g->pmf.upload(g->mass.data(), sizeof(float) * cells) || g->cmf.upload(cmf.data(), sizeof(float) * cells)) return 1;
    G.pmf = g->pmf.as<float>(); G.cmf = g->cmf.as<float>();
    {
        std::vector<int> guide;
        build_cdf_guide(cmf.data(), (int) cells, sum, guide);
        G.guide = nullptr; G.guide_n = 0;
        if (!guide.empty()) {
            if (g->guide.upload(guide.data(), guide.size() * sizeof(int))) return 1;
            G.guide = g->guide.as<int>(); G.guide_n = (int) guide.size() - 1;
        }
    }
    *out = g.release();
    return 0;
}
int psdr_hip_guiding_num_cells(const psdr_hip_guiding *g) { return g ? g->G.num_cells : 0; }
int psdr_hip_guiding_mass(const psdr_hip_guiding *g, float *out_host, int32_t cap) {
    if (!g || !out_host) return fail("null argument");
    if (cap < g->G.num_cells) return fail("buffer too small");
    std::memcpy(out_host, g->mass.data(), sizeof(float) * g->G.num_cells);
    return 0;
}
int psdr_hip_guiding_destroy(psdr_hip_guiding *g) { delete g; return 0; }

uint64_t psdr_hip_tea64(uint64_t v0, uint64_t v1) { return tea64(v0, v1); }
int psdr_hip_sampler_floats(uint64_t seed_value, uint64_t lane, uint64_t skip, int32_t n, float *out_dev, void *stream) {
    hipLaunchKernelGGL(k_sampler_floats, dim3(1), dim3(64), 0, (hipStream_t) stream, seed_value, lane, skip, n, out_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

} // extern "C"

#endif   // !PSDR_TU
