// trav4.h — closest-hit traversal of the 4-wide BVH (bvh.h::build_bvh4) for scenes with more than kBruteForceMax
// triangles; replaces jit_optix_ray_trace (reference scene_optix.cpp:343-410).
//
// The hit is defined independently of the tree: the reference's own ray_intersect_triangle (tri_test) on every
// triangle, smallest t in (RayEpsilon, 1e8), ties to the smallest original triangle id.  The boxes are padded and
// the slab test is conservative, so the tree only decides which triangles are looked at.
//
// One step = one 128-byte node: four slab tests, then the (up to four) children that are hit are ordered with a
// five-comparator sorting network on ONE 32-bit key per child,
//        key = (bits of the entry distance, low `ref_bits` bits cleared) | child code,
// (entry distances are >= 0, so their float bits order like unsigned integers; a missed child is 0xffffffff).  The
// nearest child is visited next, the others go on the per-lane stack with the far ones below, and a popped key is
// dropped when its (rounded-down) entry distance lies behind the closest hit found meanwhile.
//
// The traversal is RESUMABLE: its state lives in `Trav4`, and `trav4_run` returns when a wanted number of lanes of
// the wave has finished its rays.  The path kernels (paths.h, run_paths_async) use this to shade the finished lanes and
// hand them new rays while the other lanes keep their place in the tree - finished lanes no longer wait for the
// slowest ray of the wave (config 5 ran with 13 % of the lanes active per VALU instruction when they did).
// Each lane owns a queue of up to two rays (the next-event ray and the extension ray of one path vertex).
//
// Stack: the first T.stack_lds entries of a lane are in LDS (stride kBlock, conflict-free), deeper entries - rare -
// in a per-lane global array (T.gstack), so the LDS footprint does not grow with the depth of the tree.
#pragma once
#include "scene_dev.h"

namespace psdr {

constexpr unsigned kT4Done = 0xffffffffu;      // Trav4::code: no node in hand, the current ray is finished (or there is none)
constexpr unsigned kT4Miss = 0xffffffffu;      // sort key of a child the ray does not enter

// ---- wave-level triangle queue -------------------------------------------------------------------------------------------
// A ray visits ~3 leaves of up to four triangles, and the exact triangle test (a division, five comparisons) is more than half
// of the traversal's arithmetic.  Tested inside the per-lane loop, the lanes that hold a leaf make every other lane wait
// (and wait themselves while the others walk inner nodes).  So a lane that reaches a leaf only ENQUEUES (triangle, lane)
// pairs in a per-wave LDS ring and goes on to its next node; whenever the ring holds a wave's worth of pairs all lanes
// test one pair each - any lane tests any lane's triangle: the ray comes from the owner's parked ray in LDS, the result goes
// back with one 64-bit LDS atomic min on (bits of t, original triangle id), which is exactly the (t, id) order that defines
// the hit.  The owner's node tests read the t half for culling.  When a ray's stack is empty and its last pair has been
// tested, the owner rebuilds (slot, u, v) of the winning triangle with one more exact test.
//
// LDS rows (of kBlock words) behind the traversal stack, per workgroup (scene_dev.h::kTravRows):
//   kParkWords rows  parked rays   [word][lane]           oA dA oB dB of every lane (written by whoever posts the rays)
//   2 rows           best          [wave][lane] x u64     (t bits << 32 | original id) of the ray the lane is tracing
//   kQueueRows rows  pair ring     [wave][kQueueCap]      (slot << 7 | ray << 6 | owner lane)
//   1 row            ring heads    [wave][2]              pairs enqueued / pairs tested so far
//   3 rows           hit           [word][lane]           slot, u, v of the pair that currently holds `best`
constexpr int kQueueCap = 512;                 // power of two >= 63 left over + 64 lanes x 4 triangles
constexpr int kQueueRows = (4 * kQueueCap + kBlock - 1) / kBlock;
constexpr int kHitRows = 3;                    // (slot, u, v) of the best hit so far, written by the lane that found it
static_assert(kTravRows == kParkWords + 2 + kQueueRows + 1 + kHitRows, "scene_dev.h::kTravRows");

typedef __attribute__((address_space(3))) int lds_int_t;
typedef __attribute__((address_space(3))) unsigned lds_uint_t;
typedef __attribute__((address_space(3))) float lds_float_t;
typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
typedef __attribute__((address_space(1))) int glb_int_t;

struct Trav4 {
    Vec3f o, d, inv;            // the ray being traced
    unsigned code;              // node / leaf in hand, kT4Done = none
    int sp;                     // stack pointer
    int cur;                    // 0 / 1: which of the lane's two rays is being traced, -1 none
    int pending;                // bit k: ray k still waits
    unsigned last_pair;         // sequence number (+1) of the last pair this ray enqueued
    float anyhit;               // ray 0 only: a hit closer than this ends the ray (shadow rays: any occluder will do), -inf = closest hit wanted
    Hit hA, hB;                 // results
    PSDR_DEV void reset() { code = kT4Done; sp = 0; cur = -1; pending = 0; last_pair = 0u; anyhit = -__builtin_inff(); hA.slot = -1; hA.u = hA.v = hA.t = 0.f; hB = hA; o = Vec3f(0.f); d = Vec3f(0.f); inv = Vec3f(0.f); }
    PSDR_DEV bool idle() const { return cur < 0 && pending == 0; }
};

// per-lane / per-wave views of the LDS rows above
template <int LDS> struct T4Lds {
    lds_int_t *stack;           // this lane's stack, stride kBlock
    lds_float_t *park;          // this lane's parked rays, stride kBlock
    lds_float_t *park0;         // lane 0 of this WAVE (owner lane l: park0 + l)
    lds_u64_t *best;            // this wave's 64 entries
    lds_uint_t *ring;           // this wave's kQueueCap pairs
    lds_uint_t *heads;          // this wave's {enqueued, tested}
    lds_float_t *hit0;          // (slot, u, v) rows, lane 0 of this wave
    PSDR_DEV explicit T4Lds(const SceneView<LDS> &S) {
        const int rows = S.T->stack_lds;
        lds_int_t *base = (lds_int_t *) (S.stack - threadIdx.x);          // row 0 of the workgroup's stack area
        const int wave = threadIdx.x >> 6;
        stack = base + threadIdx.x;
        park = (lds_float_t *) (base + rows * kBlock + threadIdx.x);
        park0 = (lds_float_t *) (base + rows * kBlock + (wave << 6));
        best = (lds_u64_t *) (base + (rows + kParkWords) * kBlock) + (wave << 6);
        ring = (lds_uint_t *) (base + (rows + kParkWords + 2) * kBlock) + wave * kQueueCap;
        heads = (lds_uint_t *) (base + (rows + kParkWords + 2 + kQueueRows) * kBlock) + 2 * wave;
        hit0 = (lds_float_t *) (base + (rows + kParkWords + 2 + kQueueRows + 1) * kBlock + (wave << 6));
    }
};

template <int LDS> PSDR_DEV void t4_push(const SceneView<LDS> &S, const T4Lds<LDS> &L, int &sp, unsigned key) {
    const SceneTables &T = *S.T;
    if (sp < T.stack_lds) L.stack[sp * kBlock] = (int) key;
    else ((glb_int_t *) T.gstack)[(size_t) (sp - T.stack_lds) * T.gstack_stride + (size_t) blockIdx.x * kBlock + threadIdx.x] = (int) key;
    ++sp;
}
template <int LDS> PSDR_DEV unsigned t4_pop(const SceneView<LDS> &S, const T4Lds<LDS> &L, int &sp) {
    const SceneTables &T = *S.T;
    --sp;
    unsigned key;
    if (sp < T.stack_lds) key = (unsigned) L.stack[sp * kBlock];
    else key = (unsigned) ((glb_int_t *) T.gstack)[(size_t) (sp - T.stack_lds) * T.gstack_stride + (size_t) blockIdx.x * kBlock + threadIdx.x];
    return key;
}

// next node of this lane's ray: pop until an entry is not behind the closest hit; kT4Done when the stack is empty
template <int LDS> PSDR_DEV unsigned t4_next(const SceneView<LDS> &S, const T4Lds<LDS> &L, Trav4 &tr, unsigned cmask, float best_t) {
    while (tr.sp > 0) {
        const unsigned key = t4_pop(S, L, tr.sp);
        if (__uint_as_float(key & ~cmask) <= best_t) return key & cmask;
    }
    return kT4Done;
}

constexpr unsigned long long kT4NoHit = (0x7f800000ull << 32) | 0x7fffffffull;      // t = +inf, id = INT_MAX

// start ray `k` of the lane from its parked copy: NaN rays miss (reference scene_optix.cpp:348-353)
template <int LDS, bool COUNT> PSDR_DEV void t4_start(SceneView<LDS> &S, const T4Lds<LDS> &L, Trav4 &tr, int k) {
    const lds_float_t *q = L.park + (k == 0 ? 0 : 6 * kBlock);
    const Vec3f o(q[0], q[kBlock], q[2 * kBlock]), d(q[3 * kBlock], q[4 * kBlock], q[5 * kBlock]);
    tr.cur = k;
    tr.o = o; tr.d = d;
    tr.inv = Vec3f(1.f / d.x, 1.f / d.y, 1.f / d.z);
    tr.sp = 0;
    tr.last_pair = 0u;
    L.best[threadIdx.x & 63] = kT4NoHit;
    const bool ok = (o.x == o.x && o.y == o.y && o.z == o.z && d.x == d.x && d.y == d.y && d.z == d.z);
    tr.code = ok ? 0u : kT4Done;       // node 0 = root
    if (COUNT) { if (ok) S.c_rays++; }
}

// one inner node for the lanes whose code is an inner node
template <int LDS, bool COUNT> PSDR_DEV void t4_node(SceneView<LDS> &S, const T4Lds<LDS> &L, Trav4 &tr, unsigned cmask) {
    const SceneTables &T = *S.T;
    const int w = T.nodes_off + 8 * (int) tr.code;
    const float4 lx = S.ld(w), ly = S.ld(w + 1), lz = S.ld(w + 2), hx = S.ld(w + 3), hy = S.ld(w + 4), hz = S.ld(w + 5), cd = S.ld(w + 6);
    if (COUNT) S.c_nodes++;
    const float bt = __uint_as_float((unsigned) (L.best[threadIdx.x & 63] >> 32));      // closest hit so far (tested pairs only)
    if (tr.cur == 0 && bt < tr.anyhit) { tr.sp = 0; tr.code = kT4Done; return; }        // shadow ray: an occluder has been found
    const float ox = tr.o.x, oy = tr.o.y, oz = tr.o.z, ix = tr.inv.x, iy = tr.inv.y, iz = tr.inv.z;
    unsigned key[4];
    const float lox[4] = {lx.x, lx.y, lx.z, lx.w}, loy[4] = {ly.x, ly.y, ly.z, ly.w}, loz[4] = {lz.x, lz.y, lz.z, lz.w};
    const float hix[4] = {hx.x, hx.y, hx.z, hx.w}, hiy[4] = {hy.x, hy.y, hy.z, hy.w}, hiz[4] = {hz.x, hz.y, hz.z, hz.w};
    const unsigned cds[4] = {__float_as_uint(cd.x), __float_as_uint(cd.y), __float_as_uint(cd.z), __float_as_uint(cd.w)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // slab test; fminf / fmaxf drop NaNs (0 * inf), which keeps the test conservative; the far side gets one ulp-scale of slack
        const float ax = (lox[k] - ox) * ix, bx = (hix[k] - ox) * ix;
        const float ay = (loy[k] - oy) * iy, by = (hiy[k] - oy) * iy;
        const float az = (loz[k] - oz) * iz, bz = (hiz[k] - oz) * iz;
        const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
        const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.0000004f;
        const bool hit = tn <= fminf(tf, bt);
        key[k] = hit ? ((__float_as_uint(tn) & ~cmask) | cds[k]) : kT4Miss;
    }
    // sorting network on four keys: (0,1) (2,3) (0,2) (1,3) (1,2)
    unsigned a = min(key[0], key[1]), b = max(key[0], key[1]), c = min(key[2], key[3]), e = max(key[2], key[3]);
    const unsigned k0 = min(a, c), m1 = max(a, c), m2 = min(b, e), k3 = max(b, e);
    const unsigned k1 = min(m1, m2), k2 = max(m1, m2);
    if (k3 != kT4Miss) t4_push(S, L, tr.sp, k3);
    if (k2 != kT4Miss) t4_push(S, L, tr.sp, k2);
    if (k1 != kT4Miss) t4_push(S, L, tr.sp, k1);
    tr.code = k0 != kT4Miss ? (k0 & cmask) : t4_next(S, L, tr, cmask, bt);
}

// the leaf in hand (called by the lanes that hold one, together): its triangles join the wave's pair ring
template <int LDS> PSDR_DEV void t4_enqueue(const SceneView<LDS> &S, const T4Lds<LDS> &L, Trav4 &tr, unsigned cmask, unsigned leaf_bit) {
    const int lane_id = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane_id) - 1ull;
    const int payload = (int) (tr.code & (leaf_bit - 1u)), first = payload >> 2, cnt = (payload & 3) + 1;
    // exclusive prefix sum of cnt (1..4) over the participating lanes
    const unsigned long long b1 = __ballot(true), b2 = __ballot(cnt > 1), b3 = __ballot(cnt > 2), b4 = __ballot(cnt > 3);
    const int before = __popcll(b1 & lt_mask) + __popcll(b2 & lt_mask) + __popcll(b3 & lt_mask) + __popcll(b4 & lt_mask);
    const int total = __popcll(b1) + __popcll(b2) + __popcll(b3) + __popcll(b4);
    const unsigned base = L.heads[0];                 // wave-synchronous: every participating lane reads the old head ...
    __builtin_amdgcn_wave_barrier();
    if (lane_id == (int) __builtin_ctzll(b1)) L.heads[0] = base + (unsigned) total;      // ... before the first of them advances it
    const unsigned mine = base + (unsigned) before;
    for (int k = 0; k < cnt; ++k) L.ring[(mine + k) & (kQueueCap - 1)] = ((unsigned) (first + k) << 7) | ((unsigned) tr.cur << 6) | (unsigned) lane_id;
    tr.last_pair = mine + (unsigned) cnt;
    const float bt = __uint_as_float((unsigned) (L.best[lane_id] >> 32));
    tr.code = t4_next(S, L, tr, cmask, bt);
}

// all participating lanes test one pair each, `n` pairs starting at sequence number `from`
template <int LDS, bool COUNT> PSDR_DEV void t4_test_pairs(SceneView<LDS> &S, const T4Lds<LDS> &L, unsigned from, int n) {
    const SceneTables &T = *S.T;
    const int rank = __popcll(__ballot(true) & ((1ull << (threadIdx.x & 63)) - 1ull));
#if PSDR_DIAG == 2
    if (COUNT) S.c_hits++;
#endif
    if (rank < n) {
        const unsigned e = L.ring[(from + (unsigned) rank) & (kQueueCap - 1)];
        const int owner = (int) (e & 63u), slot = (int) (e >> 7);
        const lds_float_t *q = L.park0 + owner + ((e & 64u) ? 6 * kBlock : 0);
        const Vec3f o(q[0], q[kBlock], q[2 * kBlock]), d(q[3 * kBlock], q[4 * kBlock], q[5 * kBlock]);
        const int w = T.trav_off + 3 * slot;
        const float4 a = S.ld(w), b = S.ld(w + 1), c = S.ld(w + 2);
        float u, v, t;
        if (COUNT) S.c_tris++;
        if (tri_test(a, b, c, o, d, u, v, t)) {
            const unsigned long long key = ((unsigned long long) __float_as_uint(t) << 32) | (unsigned long long) (unsigned) __float_as_int(c.y);
            __hip_atomic_fetch_min(&L.best[owner], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // whoever holds the minimum after this batch's atomics describes the hit (keys are unique: one pair per triangle and ray)
            if (L.best[owner] == key) { L.hit0[owner] = __int_as_float(slot); L.hit0[kBlock + owner] = u; L.hit0[2 * kBlock + owner] = v; }
        }
    }
}

// the finished ray's hit
template <int LDS> PSDR_DEV Hit t4_result(const SceneView<LDS> &S, const T4Lds<LDS> &L, const Trav4 &tr) {
    Hit h; h.slot = -1; h.u = h.v = h.t = 0.f;
    const int lane_id = threadIdx.x & 63;
    const unsigned long long key = L.best[lane_id];
    if ((unsigned) (key & 0xffffffffull) != 0x7fffffffu) {
        h.slot = __float_as_int(L.hit0[lane_id]); h.u = L.hit0[kBlock + lane_id]; h.v = L.hit0[2 * kBlock + lane_id];
        h.t = __uint_as_float((unsigned) (key >> 32));
    }
    return h;
}

// Runs the lanes' ray queues (rays parked in LDS by the caller, tr.pending says which).  Returns as soon as at most `max_busy`
// lanes still have rays to trace (0: run to completion).  May be called under a partial exec mask: only the active lanes take
// part (and are counted).  Results: tr.hA / tr.hB.
template <int LDS, bool COUNT>
PSDR_DEV void trav4_run(SceneView<LDS> &S, Trav4 &tr, int max_busy) {
    const SceneTables &T = *S.T;
    const T4Lds<LDS> L(S);
    const unsigned cmask = (1u << T.ref_bits) - 1u, leaf_bit = 1u << (T.ref_bits - 1);
    const int n_lanes = __popcll(__ballot(true));
    for (;;) {
        const unsigned tested = L.heads[1];
#if PSDR_DIAG == 3
        if (COUNT) S.c_hits++;
#endif
        if (tr.code == kT4Done) {
            if (tr.cur >= 0 && tr.last_pair <= tested) {       // the ray's stack is empty and its last pair has been tested
                const Hit h = t4_result(S, L, tr);
                if (tr.cur == 0) tr.hA = h; else tr.hB = h;
                tr.cur = -1;
            }
            if (tr.cur < 0 && tr.pending != 0) {               // next ray of this lane's queue
                const int k = (tr.pending & 1) ? 0 : 1;
                tr.pending &= ~(1 << k);
                t4_start<LDS, COUNT>(S, L, tr, k);
            }
        }
        if (__popcll(__ballot(!tr.idle())) <= max_busy) break;
        // traversal burst: until the ring holds a wave's worth of pairs, or nobody has a node in hand
        for (;;) {
            if (tr.code < leaf_bit) t4_node<LDS, COUNT>(S, L, tr, cmask);
            else if (tr.code != kT4Done) t4_enqueue(S, L, tr, cmask, leaf_bit);
#if PSDR_DIAG == 1
            if (COUNT) S.c_hits++;
#endif
            const int waiting = (int) (L.heads[0] - tested);
            if (waiting >= n_lanes || __ballot(tr.code != kT4Done) == 0ull) break;
        }
        // test the waiting pairs, a wave's worth at a time; the partial last batch only when a finished ray is waiting for it
        {
            unsigned from = tested;
            const unsigned head = L.heads[0];
            while ((int) (head - from) >= n_lanes) { t4_test_pairs<LDS, COUNT>(S, L, from, n_lanes); from += (unsigned) n_lanes; }
            if (head != from && __ballot(tr.code == kT4Done && tr.cur >= 0 && (int) (tr.last_pair - from) > 0) != 0ull) {
                t4_test_pairs<LDS, COUNT>(S, L, from, (int) (head - from));
                from = head;
            }
            __builtin_amdgcn_wave_barrier();
            if ((threadIdx.x & 63) == (int) __builtin_ctzll(__ballot(true))) L.heads[1] = from;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// called once per kernel by every thread of the workgroup (make_view): the ring heads start at zero
template <int LDS> PSDR_DEV void t4_init_lds(const SceneView<LDS> &S) {
    if (S.T->stack_lds > 0) {
        lds_int_t *base = (lds_int_t *) (S.stack - threadIdx.x);
        base[(S.T->stack_lds + kParkWords + 2 + kQueueRows) * kBlock + threadIdx.x] = 0;
        __syncthreads();
    }
}

// posts the two rays of this lane (run by the lane itself: the parked copy is what the testers of the wave read)
template <int LDS> PSDR_DEV void t4_post(const SceneView<LDS> &S, Trav4 &tr, const Vec3f &oA, const Vec3f &dA, bool actA, const Vec3f &oB, const Vec3f &dB, bool actB, float anyhit_a = -__builtin_inff()) {
    const T4Lds<LDS> L(S);
    lds_float_t *p = L.park;
    p[0] = oA.x; p[kBlock] = oA.y; p[2 * kBlock] = oA.z; p[3 * kBlock] = dA.x; p[4 * kBlock] = dA.y; p[5 * kBlock] = dA.z;
    p[6 * kBlock] = oB.x; p[7 * kBlock] = oB.y; p[8 * kBlock] = oB.z; p[9 * kBlock] = dB.x; p[10 * kBlock] = dB.y; p[11 * kBlock] = dB.z;
    tr.reset();
    tr.pending = (actA ? 1 : 0) | (actB ? 2 : 0);
    tr.anyhit = anyhit_a;
}

// two rays per lane, run to completion: the synchronous form behind trace() / trace2() (secondary-edge, guiding, adjoint
// recording and ray-batch kernels)
template <int LDS, bool COUNT>
PSDR_DEV void bvh4_trace2(SceneView<LDS> &S, const Vec3f &oA, const Vec3f &dA, bool actA, const Vec3f &oB, const Vec3f &dB, bool actB, Hit &hA, Hit &hB, float anyhit_a) {
    Trav4 tr;
    t4_post(S, tr, oA, dA, actA, oB, dB, actB, anyhit_a);
    trav4_run<LDS, COUNT>(S, tr, 0);
    hA = tr.hA;
    hB = tr.hB;
}

} // namespace psdr
