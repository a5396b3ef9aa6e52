// trav4.h — closest-hit traversal of the 4-wide BVH (bvh.h::build_bvh4) for scenes with more than kBruteForceMax
// triangles; replaces jit_optix_ray_trace (reference scene_optix.cpp:343-410).
//
// The hit is defined independently of the tree: the reference's own ray_intersect_triangle (tri_test) on every
// triangle, smallest t in (RayEpsilon, 1e8), ties to the smallest original triangle id.  The boxes are padded and
// the slab test is conservative, so the tree only decides which triangles are looked at.
//
// LANES ARE TRAVERSAL WORKERS, RAYS ARE WORK ITEMS (round 3).  A lane that owns a path vertex POSTS its (up to two) rays - the
// next-event ray and the extension ray - into LDS: the ray itself, an empty best-hit record and an entry in the wave's RAY
// QUEUE.  Any lane of the wave without a ray in hand pulls the next ray from that queue, walks it, and when the walk ends
// publishes the ray as finished and pulls another one - it neither waits for its own path's other ray nor for the shading of
// the paths that are already complete (round 2 traced a lane's two rays one after the other IN that lane: a third of the lanes
// of the node loop sat on finished rays, waiting for the shading threshold).  An owner's vertex is complete when both of its rays
// are; `trav4_run` returns when at most a wanted number of owners is still incomplete, and the path kernels (paths.h,
// run_paths_async) shade the complete ones while the workers keep their place in the tree.
//
// One step = one node (64 bytes, bvh.h: the children's boxes quantised to 8 bits inside the node's own box): four 16-byte loads,
// four slab tests, then the (up to four) children that are hit are ordered with a five-comparator sorting network on ONE 32-bit
// key per child,
//        key = (bits of the entry distance, low `ref_bits` bits cleared) | child code,
// (entry distances are >= 0, so their float bits order like unsigned integers; a missed child is 0xffffffff).  The nearest child
// is visited next, the others go on the worker's stack with the far ones below, and a popped key is dropped when its
// (rounded-down) entry distance lies behind the closest hit found meanwhile.
//
// THE WAVE TESTS THE TRIANGLES.  A worker that reaches a leaf only ENQUEUES (triangle, ray) pairs in a per-wave LDS ring and goes
// on to its next node; whenever the ring holds a wave's worth of pairs every lane tests ONE pair - anybody's: the ray comes from
// its parked copy, the result goes back with one 64-bit LDS atomic min on (bits of t << 32 | original triangle id), which IS the
// (t, id) order that defines the hit.  A ray is finished when its walk has ended and its last pair has been tested; its owner then turns the winning id back
// into the triangle (the blob's id -> slot map) and recomputes (u, v) with the same test on the parked ray - the same bits (round 5: until then the winning lane
// of every batch wrote (slot, u, v) to six LDS rows per workgroup; of those rows two now hold the TOP OF THE TREE, see kTopNodes, and four are stack rows - round 6).
//
// Stack: the first T.stack_lds (12) entries of a worker are in LDS (stride kBlock, conflict-free), deeper entries - a lane gets within three of them in 7 % of the burst iterations of config 5 -
// in a per-lane global array (T.gstack), so the LDS footprint does not grow with the depth of the tree.
#pragma once
#include "scene_dev.h"

namespace psdr {

// measurement knobs (tools/variants.py): leaf-enqueue rounds per burst iteration; a burst ends early when 1 / PSDR_BURST_DIV of the workers idle and as many rays wait
#ifndef PSDR_ENQ_ROUNDS
#define PSDR_ENQ_ROUNDS 1
#endif
#ifndef PSDR_BURST_DIV
#define PSDR_BURST_DIV 4
#endif
#ifndef PSDR_PAIR_MIN          // a burst ends when the ring holds this many pairs, and that many are worth a (partial) test round; 64 = a wave's worth
#define PSDR_PAIR_MIN 48         // (config 5: 32: 260.5, 48: 258.9, 64: 264.4 ms - earlier hits cull more; 128 - two rounds back to back, half the outer iterations - 285 vs 264 ms on config 5: the hits arrive later and cull less)
#endif
#ifndef PSDR_TOP_LDS           // 1: the first kTopNodes nodes of the tree are read from the workgroup's LDS copy
#define PSDR_TOP_LDS 1
#endif
#ifndef PSDR_FAST_STACK        // 1: pushes and pops without the global tail's branch in the burst iterations in which no lane is near it (wave-uniform test)
#define PSDR_FAST_STACK 1
#endif
#ifndef PSDR_STEAL             // 1: workers without a ray take over the bottom stack entry (the far subtree) of a walk in progress once the wave's ray queue is empty
#define PSDR_STEAL 1
#endif
#ifndef PSDR_STEAL_MIN         // a walk gives an entry away when its stack holds at least this many
#define PSDR_STEAL_MIN 1
#endif
#ifndef PSDR_STEAL_IDLE        // a steal round is worth its instructions when at least this many workers idle
#define PSDR_STEAL_IDLE 8
#endif
#ifndef PSDR_STEAL_TOP         // ray kinds (bit 0: next-event, bit 1: extension) that give their TOP stack entry (the next-nearest subtree) instead of the bottom one
#define PSDR_STEAL_TOP 3
#endif
#ifndef PSDR_STEAL_BREAK       // 0 = off.  Measurement knob, NOT validated: with 2, config 5's secondary-edge kernel runs 26.5 -> 24.5 ms and the traced hits stay bit-equal, but the
#define PSDR_STEAL_BREAK 0       // reverse sweeps of environment-lit scenes then miss their dot-product tests by 0.5-2 % (tests/test_gpu_adjoint.py::test_interior_sweep_environment_map): left off
#endif
#ifndef PSDR_STEAL_ROUNDS      // steal rounds per hand-over (a walk gives one entry per round)
#define PSDR_STEAL_ROUNDS 1
#endif
#ifndef PSDR_STEAL_ROUNDS_ASYNC
#define PSDR_STEAL_ROUNDS_ASYNC 1
#endif
#ifndef PSDR_STEAL_KINDS       // 3: any ray, 2: extension rays only (closest hit wanted), 1: next-event rays only
#define PSDR_STEAL_KINDS 3
#endif
// PSDR_DIAG (instrumented development builds, counted launches only): 8 = phase timers; 11 = census of the burst iterations' lane slots (c_nodes on an inner node, c_tris holding
// a leaf - sitting the node step out -, c_rays without a ray, c_hits walk ended and waiting for the hand-over); 12 = how often the stack's global tail is in reach (c_tris lanes
// on a node with fewer than three LDS rows left, c_rays burst iterations with at least one such lane, c_hits all lane slots)
#define PSDR_DIAG_PLAIN (PSDR_DIAG != 8 && PSDR_DIAG != 11 && PSDR_DIAG != 12)
constexpr unsigned kFinBusy = 0x80000000u, kFinShared = 0x40000000u, kFinCount = 0x3fffffffu;      // `fin` while a ray is posted / walked: busy flag | walked by more than one worker at some point | walkers
// BOUND: once a ray is finished the same word holds 1 + the wave's pair sequence number (heads[kHdPairEnq], zeroed per kernel launch), and bit 31 of THAT would read as
// "busy" for ever: a persistent wave may enqueue fewer than 2^31 (triangle, ray) pairs per launch.  Config 5 at full size reaches ~1e6 per wave (8.9 pairs per ray,
// ~1.2e5 rays per wave); a launch that could come near the bound has to be split by the caller (shard_count), nothing here wraps the counter.
constexpr unsigned kT4Done = 0xffffffffu;      // Trav4::code: no node in hand (walk finished, or no ray)
constexpr unsigned kT4Miss = 0xffffffffu;      // sort key of a child the ray does not enter / code of an unused child slot

// LDS rows (of kBlock words) behind the traversal stack, per workgroup of four waves (scene_dev.h::kTravRows):
//   kParkWords rows  parked rays   [word][lane]            oA dA oB dB + the any-hit distance of ray A, written by the owner
//   4 rows           best          [wave][kRayCap] x u64   (t bits << 32 | original id) per ray
//   kTopRows rows    top           [kTopNodes][16]         the first kTopNodes nodes of the tree (bvh.h numbers the top levels first): every walk starts there, and an
//                                                             LDS read costs the wave 8 clocks where 64 lanes on 64 different cache lines cost the L1 a tag lookup each
//   2 rows           fin           [wave][kRayCap]         0 = walk not finished; else 1 + sequence number after the ray's last pair
//   kPairRows rows   pair ring     [wave][kPairCap]        (slot << 7 | ray)
//   2 rows           ray queue     [wave][kRayCap]         rays posted and not yet taken by a worker
//   1 row            heads         [wave][4]               pairs enqueued, pairs tested, rays posted, rays taken
constexpr int kRayCap = 128;                   // rays of a wave: two per lane; ray id = k * 64 + owner lane (k = 0: next-event ray, 1: extension ray)
constexpr int kPairCap = 256;                  // power of two >= 63 left over + 64 workers x 2 triangles per leaf (bvh.h: leaves of <= 2)
constexpr int kPairRows = (4 * kPairCap + kBlock - 1) / kBlock;
constexpr int kRowBest = kParkWords, kRowTop = kRowBest + 4, kRowFin = kRowTop + kTopRows, kRowPair = kRowFin + 2, kRowRq = kRowPair + kPairRows, kRowHeads = kRowRq + 2;
constexpr int kTopNodes = kTopRows * kBlock / (4 * kNodeW4);       // 16 nodes of 64 bytes per row (8 of 128 in the 8-wide measurement build)
static_assert(kTravRows == kRowHeads + 1, "scene_dev.h::kTravRows");
static_assert(kParkWords == 13, "scene_dev.h::kParkWords");
enum { kHdPairEnq = 0, kHdPairTested = 1, kHdRayTail = 2, kHdRayHead = 3 };

typedef __attribute__((address_space(3))) int lds_int_t;
typedef __attribute__((address_space(3))) unsigned lds_uint_t;
typedef __attribute__((address_space(3))) float lds_float_t;
typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
typedef __attribute__((address_space(1))) int glb_int_t;
typedef float t4_v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) t4_v4f lds_v4f_t;

// Lanes of a wave talk to each other through LDS (owner -> worker -> tester -> owner).  The hardware executes a wave's LDS
// instructions in order; this keeps the COMPILER from moving or forwarding accesses across the points where another lane's
// write has to be seen.
PSDR_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// a worker's traversal state (registers)
struct Trav4 {
    Vec3f o, d, inv;            // the ray in hand
    unsigned code;              // node / leaf in hand, kT4Done = none
    int sp;                     // stack pointer
    int sb;                     // stack bottom: the entries below were given to other workers (PSDR_STEAL); the stack is [sb, sp)
    int rid;                    // ray in hand, -1 none
    unsigned last_pair;         // sequence number after the last pair this ray enqueued
    float anyhit;               // a hit closer than this ends the walk (shadow rays: any occluder will do), -inf = closest hit wanted
    PSDR_DEV void reset() { code = kT4Done; sp = 0; sb = 0; rid = -1; last_pair = 0u; anyhit = -__builtin_inff(); o = Vec3f(0.f); d = Vec3f(0.f); inv = Vec3f(0.f); }
};

// per-lane / per-wave views of the LDS rows above
template <int LDS> struct T4Lds {
    lds_int_t *stack0;          // row 0 of the workgroup's stack area (lane t: stack0 + t, stride kBlock; wave-uniform)
    lds_float_t *park;          // this lane's parked rays, stride kBlock
    lds_float_t *park0;         // lane 0 of this WAVE (owner lane l: park0 + l)
    lds_u64_t *best;            // this wave's kRayCap entries
    const lds_v4f_t *top;       // the workgroup's copy of the first kTopNodes nodes
    lds_uint_t *fin;            // this wave's kRayCap entries
    lds_uint_t *ring;           // this wave's kPairCap pairs
    lds_uint_t *rq;             // this wave's kRayCap queue slots
    lds_uint_t *heads;          // this wave's four counters
    PSDR_DEV explicit T4Lds(const SceneView<LDS> &S) {
        const int rows = S.T->stack_lds;
        lds_int_t *base = (lds_int_t *) (S.stack - threadIdx.x);          // row 0 of the workgroup's stack area
        const int wave = threadIdx.x >> 6;
        stack0 = base;
        park = (lds_float_t *) (base + rows * kBlock + threadIdx.x);
        park0 = (lds_float_t *) (base + rows * kBlock + (wave << 6));
        best = (lds_u64_t *) (base + (rows + kRowBest) * kBlock) + wave * kRayCap;
        top = (const lds_v4f_t *) (base + (rows + kRowTop) * kBlock);
        fin = (lds_uint_t *) (base + (rows + kRowFin) * kBlock) + wave * kRayCap;
        ring = (lds_uint_t *) (base + (rows + kRowPair) * kBlock) + wave * kPairCap;
        rq = (lds_uint_t *) (base + (rows + kRowRq) * kBlock) + wave * kRayCap;
        heads = (lds_uint_t *) (base + (rows + kRowHeads) * kBlock) + 4 * wave;
    }
};

// This lane's stack base, recomputed from the thread index where it is used: kept in a register across the whole kernel it is one of the
// values the allocator spills, and a push then starts with a scratch load and a full s_waitcnt vmcnt(0) (three per node step, seen in the ISA)
template <int LDS> PSDR_DEV lds_int_t *t4_stack_base(const T4Lds<LDS> &L) {
    unsigned tid;
    asm volatile("v_mov_b32 %0, %1" : "=v"(tid) : "v"(threadIdx.x));
    return L.stack0 + tid;
}
template <int LDS> PSDR_DEV void t4_push(const SceneView<LDS> &S, const T4Lds<LDS> &L, int &sp, unsigned key) {
    const SceneTables &T = *S.T;
    if (sp < T.stack_lds) t4_stack_base(L)[sp * kBlock] = (int) key;
    else ((glb_int_t *) T.gstack)[(size_t) (sp - T.stack_lds) * T.gstack_stride + (size_t) blockIdx.x * kBlock + threadIdx.x] = (int) key;
    ++sp;
}
template <int LDS> PSDR_DEV unsigned t4_pop(const SceneView<LDS> &S, const T4Lds<LDS> &L, int &sp) {
    const SceneTables &T = *S.T;
    --sp;
    unsigned key;
    if (sp < T.stack_lds) key = (unsigned) t4_stack_base(L)[sp * kBlock];
    else key = (unsigned) ((glb_int_t *) T.gstack)[(size_t) (sp - T.stack_lds) * T.gstack_stride + (size_t) blockIdx.x * kBlock + threadIdx.x];
    return key;
}

// next node of this worker's ray: pop until an entry is not behind the closest hit; kT4Done when the stack is empty
template <int LDS, bool FAST = false> PSDR_DEV unsigned t4_next(const SceneView<LDS> &S, const T4Lds<LDS> &L, Trav4 &tr, unsigned cmask, float best_t) {
#if PSDR_FAST_STACK
    if constexpr (FAST)
    // (round 6) every stack of the lanes that pop now lies in LDS - with 12 rows that is 93 % of the burst iterations of config 5 -: the loop without the global tail's branch
    if (__ballot(tr.sp > S.T->stack_lds) == 0ull) {
        lds_int_t *b = t4_stack_base(L);
        while (tr.sp > tr.sb) {
            --tr.sp;
            const unsigned key = (unsigned) b[tr.sp * kBlock];
            if (__uint_as_float(key & ~cmask) <= best_t) return key & cmask;
        }
        return kT4Done;
    }
#endif
    while (tr.sp > tr.sb) {
        const unsigned key = t4_pop(S, L, tr.sp);
        if (__uint_as_float(key & ~cmask) <= best_t) return key & cmask;
    }
    return kT4Done;
}

constexpr unsigned long long kT4NoHit = (0x7f800000ull << 32) | 0x7fffffffull;      // t = +inf, id = INT_MAX

PSDR_DEV float t4_best_t(const lds_u64_t *best, int rid) {
    return __uint_as_float((unsigned) (__hip_atomic_load(&best[rid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 32));
}

// 1 / d for the slab tests.  The walk only has to be conservative - the boxes are padded by 1e-4 of the coordinate magnitude (bvh.h) and the hit itself comes from
// tri_test on the parked ray - so the hardware reciprocal (1 ulp, one instruction) does what the IEEE division (a ten-instruction sequence, three per ray) did:
// +-0 still gives +-inf, NaN stays NaN
PSDR_DEV Vec3f t4_inv_dir(const Vec3f &d) {
#ifdef PSDR_T4_IEEE_INV
    return Vec3f(1.f / d.x, 1.f / d.y, 1.f / d.z);
#else
    return Vec3f(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
#endif
}

// a worker takes ray `rid` from its parked copy: NaN rays miss (reference scene_optix.cpp:348-353)
template <int LDS, bool COUNT> PSDR_DEV void t4_start(SceneView<LDS> &S, const T4Lds<LDS> &L, Trav4 &tr, int rid) {
    const int owner = rid & 63;
    const lds_float_t *q = L.park0 + owner + ((rid & 64) ? 6 * kBlock : 0);
    const Vec3f o(q[0], q[kBlock], q[2 * kBlock]), d(q[3 * kBlock], q[4 * kBlock], q[5 * kBlock]);
    tr.rid = rid;
    tr.o = o; tr.d = d;
    tr.inv = t4_inv_dir(d);
    tr.sp = 0; tr.sb = 0;
    tr.last_pair = 0u;
    L.fin[rid] = kFinBusy | 1u;            // one walker
    tr.anyhit = (rid & 64) ? -__builtin_inff() : L.park0[owner + 12 * kBlock];
    const bool ok = (o.x == o.x && o.y == o.y && o.z == o.z && d.x == d.x && d.y == d.y && d.z == d.z);
    tr.code = ok ? 0u : kT4Done;       // node 0 = root
#if PSDR_DIAG_PLAIN
    if (COUNT) { if (ok) S.c_rays++; }
#endif
}

// one inner node for the workers whose code is an inner node
template <int LDS, bool COUNT, bool FAST = false> PSDR_DEV void t4_node(SceneView<LDS> &S, const T4Lds<LDS> &L, Trav4 &tr, unsigned cmask) {
    const SceneTables &T = *S.T;
#if PSDR_DIAG != 8
    if (COUNT) S.c_nodes++;
#endif
    const float bt = t4_best_t(L.best, tr.rid);                               // closest hit so far (tested pairs only)
    if (bt < tr.anyhit) { tr.sp = tr.sb; tr.code = kT4Done; return; }              // shadow ray: an occluder has been found
    const float ox = tr.o.x, oy = tr.o.y, oz = tr.o.z, ix = tr.inv.x, iy = tr.inv.y, iz = tr.inv.z;
    unsigned key[4];
    // 64-byte node (bvh.h): {origin.xyz, exponents} {lo.x[4] lo.y[4] lo.z[4] hi.x[4]} {hi.y[4] hi.z[4] code0 code1} {code2 code3 - -};
    // child k's bounds are bytes k of the six words: lo = origin + 2^e q_lo (rounded down), hi = origin + 2^e q_hi (rounded up).
    // Along one axis  (origin + s q - o) * inv = q * (s inv) + (origin - o) inv: one convert and one fma per bound, as many
    // VALU instructions as the subtract and multiply of the float node, for four loads instead of seven.  The roundings of the
    // factored form (a few ulps of |origin - o| |inv|) stay far inside the builder's padding of every box (1e-4 of the coordinate
    // magnitude, bvh.h).
#if PSDR_BVH_WIDTH == 8
    // 8-wide measurement build (bvh.h, PSDR_BVH_WIDTH == 8): a node is one 128-byte line - w0 origin + exponents; w1 lo.x[8] lo.y[8]; w2 lo.z[8] hi.x[8]; w3 hi.y[8] hi.z[8];
    // w4 codes 0-3; w5 codes 4-7 - eight slab tests, a 19-comparator network (Batcher's odd-even merge sort), up to seven pushes
    float4 n0, n1, n2, n3, n4, n5;
#if PSDR_TOP_LDS
    if (tr.code < (unsigned) kTopNodes) {
        const lds_v4f_t *q = L.top + kNodeW4 * (int) tr.code;
        const t4_v4f q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5];
        n0 = make_float4(q0.x, q0.y, q0.z, q0.w); n1 = make_float4(q1.x, q1.y, q1.z, q1.w); n2 = make_float4(q2.x, q2.y, q2.z, q2.w);
        n3 = make_float4(q3.x, q3.y, q3.z, q3.w); n4 = make_float4(q4.x, q4.y, q4.z, q4.w); n5 = make_float4(q5.x, q5.y, q5.z, q5.w);
    } else
#endif
    {
        const int w = T.nodes_off + kNodeW4 * (int) tr.code;
        n0 = S.ld(w); n1 = S.ld(w + 1); n2 = S.ld(w + 2); n3 = S.ld(w + 3); n4 = S.ld(w + 4); n5 = S.ld(w + 5);
    }
    const unsigned ex = __float_as_uint(n0.w);
    const float sx = __uint_as_float((ex & 0xffu) << 23), sy = __uint_as_float(((ex >> 8) & 0xffu) << 23), sz = __uint_as_float(((ex >> 16) & 0xffu) << 23);
    const float ax0 = (n0.x - ox) * ix, ay0 = (n0.y - oy) * iy, az0 = (n0.z - oz) * iz;
    const float bxs = sx * ix, bys = sy * iy, bzs = sz * iz;
    const bool ngx = ix < 0.f, ngy = iy < 0.f, ngz = iz < 0.f;
    const unsigned wlx[2] = {__float_as_uint(n1.x), __float_as_uint(n1.y)}, wly[2] = {__float_as_uint(n1.z), __float_as_uint(n1.w)}, wlz[2] = {__float_as_uint(n2.x), __float_as_uint(n2.y)};
    const unsigned whx[2] = {__float_as_uint(n2.z), __float_as_uint(n2.w)}, why[2] = {__float_as_uint(n3.x), __float_as_uint(n3.y)}, whz[2] = {__float_as_uint(n3.z), __float_as_uint(n3.w)};
    const unsigned cds[8] = {__float_as_uint(n4.x), __float_as_uint(n4.y), __float_as_uint(n4.z), __float_as_uint(n4.w), __float_as_uint(n5.x), __float_as_uint(n5.y), __float_as_uint(n5.z), __float_as_uint(n5.w)};
    unsigned key8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int j = k >> 2, sh = 8 * (k & 3);
        const unsigned qnx = ngx ? whx[j] : wlx[j], qny = ngy ? why[j] : wly[j], qnz = ngz ? whz[j] : wlz[j];
        const unsigned qfx = ngx ? wlx[j] : whx[j], qfy = ngy ? wly[j] : why[j], qfz = ngz ? wlz[j] : whz[j];
        const float ax = fmaf((float) ((qnx >> sh) & 0xffu), bxs, ax0), bx = fmaf((float) ((qfx >> sh) & 0xffu), bxs, ax0);
        const float ay = fmaf((float) ((qny >> sh) & 0xffu), bys, ay0), by = fmaf((float) ((qfy >> sh) & 0xffu), bys, ay0);
        const float az = fmaf((float) ((qnz >> sh) & 0xffu), bzs, az0), bz = fmaf((float) ((qfz >> sh) & 0xffu), bzs, az0);
        const float tn = fmaxf(fmaxf(ax, ay), fmaxf(az, 0.f));
        const float tf = fminf(fminf(bx, by), bz) * 1.0000004f;
        const bool hit = (tn <= fminf(tf, bt)) & (cds[k] != kT4Miss);
        key8[k] = hit ? ((__float_as_uint(tn) & ~cmask) | cds[k]) : kT4Miss;
    }
#define PSDR_CX(i, j) do { const unsigned lo_ = min(key8[i], key8[j]), hi_ = max(key8[i], key8[j]); key8[i] = lo_; key8[j] = hi_; } while (0)
    PSDR_CX(0, 1); PSDR_CX(2, 3); PSDR_CX(4, 5); PSDR_CX(6, 7);
    PSDR_CX(0, 2); PSDR_CX(1, 3); PSDR_CX(4, 6); PSDR_CX(5, 7);
    PSDR_CX(1, 2); PSDR_CX(5, 6);
    PSDR_CX(0, 4); PSDR_CX(1, 5); PSDR_CX(2, 6); PSDR_CX(3, 7);
    PSDR_CX(2, 4); PSDR_CX(3, 5);
    PSDR_CX(1, 2); PSDR_CX(3, 4); PSDR_CX(5, 6);
#undef PSDR_CX
#pragma unroll
    for (int k = 7; k >= 1; --k) if (key8[k] != kT4Miss) t4_push(S, L, tr.sp, key8[k]);
    tr.code = key8[0] != kT4Miss ? (key8[0] & cmask) : t4_next<LDS, FAST>(S, L, tr, cmask, bt);
}
#else
    float4 n0, n1, n2, n3;
#if PSDR_TOP_LDS
    if (tr.code < (unsigned) kTopNodes) {             // (the top of the tree: the workgroup's LDS copy, t4_init_lds)
        const lds_v4f_t *q = L.top + kNodeW4 * (int) tr.code;
        const t4_v4f q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        n0 = make_float4(q0.x, q0.y, q0.z, q0.w); n1 = make_float4(q1.x, q1.y, q1.z, q1.w); n2 = make_float4(q2.x, q2.y, q2.z, q2.w); n3 = make_float4(q3.x, q3.y, q3.z, q3.w);
    } else
#endif
    {
        const int w = T.nodes_off + kNodeW4 * (int) tr.code;
        n0 = S.ld(w); n1 = S.ld(w + 1); n2 = S.ld(w + 2); n3 = S.ld(w + 3);
#ifdef PSDR_T_NODE_DUP
        // measurement build (round 6): the node's four loads issued a second time - same lines, so L1 hits: what one more set of lane-loads per node step costs the
        // kernel bounds what a cheaper fetch (a quad per node: one instruction per whole node) could give back
        { const float4 *q = &S.B[w];
          t4_v4f m0, m1, m2, m3;
          asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\tglobal_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3) : "v"(q) : "memory"); }
#endif
    }
    const unsigned ex = __float_as_uint(n0.w);
    const float sx = __uint_as_float((ex & 0xffu) << 23), sy = __uint_as_float(((ex >> 8) & 0xffu) << 23), sz = __uint_as_float(((ex >> 16) & 0xffu) << 23);
    const float ax0 = (n0.x - ox) * ix, ay0 = (n0.y - oy) * iy, az0 = (n0.z - oz) * iz;
    const float bxs = sx * ix, bys = sy * iy, bzs = sz * iz;
    // the planes a ray enters / leaves a box through are known from the signs of its direction: pick the byte words once per node
    // (lo or hi per axis) instead of ordering every child's two plane distances with a min and a max (six instructions per child)
    const bool ngx = ix < 0.f, ngy = iy < 0.f, ngz = iz < 0.f;
    const unsigned wlx = __float_as_uint(n1.x), wly = __float_as_uint(n1.y), wlz = __float_as_uint(n1.z);
    const unsigned whx = __float_as_uint(n1.w), why = __float_as_uint(n2.x), whz = __float_as_uint(n2.y);
    const unsigned qnx = ngx ? whx : wlx, qny = ngy ? why : wly, qnz = ngz ? whz : wlz;
    const unsigned qfx = ngx ? wlx : whx, qfy = ngy ? wly : why, qfz = ngz ? wlz : whz;
    const unsigned cds[4] = {__float_as_uint(n2.z), __float_as_uint(n2.w), __float_as_uint(n3.x), __float_as_uint(n3.y)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ax = fmaf((float) ((qnx >> (8 * k)) & 0xffu), bxs, ax0), bx = fmaf((float) ((qfx >> (8 * k)) & 0xffu), bxs, ax0);
        const float ay = fmaf((float) ((qny >> (8 * k)) & 0xffu), bys, ay0), by = fmaf((float) ((qfy >> (8 * k)) & 0xffu), bys, ay0);
        const float az = fmaf((float) ((qnz >> (8 * k)) & 0xffu), bzs, az0), bz = fmaf((float) ((qfz >> (8 * k)) & 0xffu), bzs, az0);
        // slab test; fminf / fmaxf drop NaNs (0 * inf, inf - inf for a direction component of zero), which keeps the test conservative; the far
        // side gets one ulp-scale of slack
        const float tn = fmaxf(fmaxf(ax, ay), fmaxf(az, 0.f));
        const float tf = fminf(fminf(bx, by), bz) * 1.0000004f;
        // (the code test stays although an unused slot carries an inverted box: for a node that is tiny against its distance from the ray's origin the fma absorbs
        //  255 steps and near == far on every axis - the box alone does not reject then)
        const bool hit = (tn <= fminf(tf, bt)) & (cds[k] != kT4Miss);      // `&`: with `&&` the compiler sinks the load of the child's code into a branch - a second memory round trip per node
        key[k] = hit ? ((__float_as_uint(tn) & ~cmask) | cds[k]) : kT4Miss;
    }
    // sorting network on four keys: (0,1) (2,3) (0,2) (1,3) (1,2)
    unsigned a = min(key[0], key[1]), b = max(key[0], key[1]), c = min(key[2], key[3]), e = max(key[2], key[3]);
    const unsigned k0 = min(a, c), m1 = max(a, c), m2 = min(b, e), k3 = max(b, e);
    const unsigned k1 = min(m1, m2), k2 = max(m1, m2);
#if PSDR_FAST_STACK
    // (round 6) the pushes without a branch when no lane of this step comes within three entries of the global tail (wave-uniform test): the keys are sorted, so the far
    // ones are the misses - all three words are written, a miss onto the slot the next valid key overwrites (LDS executes a wave's writes in order) or above the new top
    if (FAST && __ballot(tr.sp + 3 > T.stack_lds) == 0ull) {
        const int n_push = (k1 != kT4Miss ? 1 : 0) + (k2 != kT4Miss ? 1 : 0) + (k3 != kT4Miss ? 1 : 0);
        lds_int_t *b = t4_stack_base(L) + tr.sp * kBlock;
        b[(n_push > 3 ? n_push - 3 : 0) * kBlock] = (int) k3;
        b[(n_push > 2 ? n_push - 2 : 0) * kBlock] = (int) k2;
        b[(n_push > 1 ? n_push - 1 : 0) * kBlock] = (int) k1;
        tr.sp += n_push;
    } else
#endif
    {
        if (k3 != kT4Miss) t4_push(S, L, tr.sp, k3);
        if (k2 != kT4Miss) t4_push(S, L, tr.sp, k2);
        if (k1 != kT4Miss) t4_push(S, L, tr.sp, k1);
    }
    tr.code = k0 != kT4Miss ? (k0 & cmask) : t4_next<LDS, FAST>(S, L, tr, cmask, bt);
}
#endif

// the leaf in hand (called by the workers that hold one, together): its triangles join the wave's pair ring
template <int LDS, bool FAST = false> PSDR_DEV void t4_enqueue(const SceneView<LDS> &S, const T4Lds<LDS> &L, Trav4 &tr, unsigned cmask, unsigned leaf_bit) {
    const int payload = (int) (tr.code & (leaf_bit - 1u)), first = payload >> 2, cnt = (payload & 3) + 1;        // a leaf holds one or two triangles (bvh.h::build_bvh, kLeafMax)
#ifdef PSDR_ENQ_BALLOT
    // (round 2/3 form: exclusive prefix sum of cnt over the participating lanes - two ballots, four popcounts, a head read, a fence and a head write for the ~8 lanes
    // of 64 that hold a leaf in a burst iteration)
    const int lane_id = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane_id) - 1ull;
    const unsigned long long b1 = __ballot(true), b2 = __ballot(cnt > 1);
    const int before = __popcll(b1 & lt_mask) + __popcll(b2 & lt_mask);
    const int total = __popcll(b1) + __popcll(b2);
    const unsigned base = L.heads[kHdPairEnq];        // every participating lane reads the old head ...
    wave_sync();
    if (lane_id == (int) __builtin_ctzll(b1)) L.heads[kHdPairEnq] = base + (unsigned) total;      // ... before the first of them advances it
    const unsigned mine = base + (unsigned) before;
#else
    // each leaf-holding lane reserves its ring slots with ONE LDS atomic on the wave's head (few lanes take part, so the same-address serialisation is short);
    // the order of the pairs in the ring is free: a ray is finished when the test counter has passed its last pair
    const unsigned mine = __hip_atomic_fetch_add(&L.heads[kHdPairEnq], (unsigned) cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    L.ring[mine & (kPairCap - 1)] = ((unsigned) first << 7) | (unsigned) tr.rid;
    if (cnt > 1) L.ring[(mine + 1u) & (kPairCap - 1)] = ((unsigned) (first + 1) << 7) | (unsigned) tr.rid;
    tr.last_pair = mine + (unsigned) cnt;
    tr.code = t4_next<LDS, FAST>(S, L, tr, cmask, t4_best_t(L.best, tr.rid));
}

// all participating lanes test one pair each, `n` pairs starting at sequence number `from`
template <int LDS, bool COUNT> PSDR_DEV void t4_test_pairs(SceneView<LDS> &S, const T4Lds<LDS> &L, unsigned from, int n) {
    const SceneTables &T = *S.T;
    const int rank = __popcll(__ballot(true) & ((1ull << (threadIdx.x & 63)) - 1ull));
#if PSDR_DIAG == 2
    if (COUNT) S.c_hits++;
#endif
    if (rank < n) {
        const unsigned e = L.ring[(from + (unsigned) rank) & (kPairCap - 1)];
        const int rid = (int) (e & 127u), slot = (int) (e >> 7);
        const lds_float_t *q = L.park0 + (rid & 63) + ((rid & 64) ? 6 * kBlock : 0);
        const Vec3f o(q[0], q[kBlock], q[2 * kBlock]), d(q[3 * kBlock], q[4 * kBlock], q[5 * kBlock]);
        const int w = T.trav_off + 3 * slot;
        const float4 a = S.ld(w), b = S.ld(w + 1), c = S.ld(w + 2);
        float u, v, t;
#if PSDR_DIAG_PLAIN
        if (COUNT) S.c_tris++;
#endif
        if (tri_test(a, b, c, o, d, u, v, t)) {
            const unsigned long long key = ((unsigned long long) __float_as_uint(t) << 32) | (unsigned long long) (unsigned) __float_as_int(c.y);
            __hip_atomic_fetch_min(&L.best[rid], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// the finished ray's hit (read by its owner)
template <int LDS> PSDR_DEV Hit t4_result(const SceneView<LDS> &S, int k) {
    const T4Lds<LDS> L(S);
    Hit h; h.slot = -1; h.u = h.v = h.t = 0.f;
    const int rid = (k << 6) | (threadIdx.x & 63);
    const unsigned long long key = __hip_atomic_load(&L.best[rid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((unsigned) (key & 0xffffffffull) != 0x7fffffffu) {
        // the winning pair again: id -> slot (the map the per-mesh samplers use, shade.h), the triangle's rows, the ray's parked copy, the same test
        const SceneTables &T = *S.T;
        h.slot = S.ldi(T.map_off, (int) (unsigned) (key & 0xffffffffull));
        const lds_float_t *q = L.park0 + (threadIdx.x & 63) + (k ? 6 * kBlock : 0);
        const Vec3f o(q[0], q[kBlock], q[2 * kBlock]), d(q[3 * kBlock], q[4 * kBlock], q[5 * kBlock]);
        const int w = T.trav_off + 3 * h.slot;
        const float4 a = S.ld(w), b = S.ld(w + 1), c = S.ld(w + 2);
        float t;
        (void) tri_test(a, b, c, o, d, h.u, h.v, t);
        h.t = __uint_as_float((unsigned) (key >> 32));
    }
    return h;
}

// Works on the wave's ray queue until at most `max_busy` OWNERS still wait for a posted ray (0: until every posted ray is
// finished).  `posted`: bit k set = this lane posted ray k with t4_post and has not consumed it yet.  May be called under a partial
// exec mask: only the active lanes work (and are counted).  Returns true for a lane whose posted rays are all finished (their hits:
// t4_result); the workers' walks in progress stay in `tr` for the next call.
// FAST (round 6): the branch-free stack paths of t4_node / t4_next in the burst iterations that allow them - the path kernels (run_paths_async) gain 1.4 % from them, the
// synchronous callers lose (config 5's secondary-edge kernel 20.2 -> 21.6 ms: its registers are what it runs out of), so they keep the plain form
template <int LDS, bool COUNT, bool FAST = false>
PSDR_DEV bool trav4_run(SceneView<LDS> &S, Trav4 &tr, int posted, int max_busy) {
    const SceneTables &T = *S.T;
    const T4Lds<LDS> L(S);
    const unsigned cmask = (1u << T.ref_bits) - 1u, leaf_bit = 1u << (T.ref_bits - 1);
    const int lane_id = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane_id) - 1ull, m_all = __ballot(true);
    const int n_lanes = __popcll(m_all), first = (int) __builtin_ctzll(m_all);
    bool done = true;
#if PSDR_DIAG == 8
    // phase timers (wave cycles): c_nodes = node bursts, c_tris = pair tests, c_hits = ray hand-over and bookkeeping (c_rays: the shading phase, paths.h)
    unsigned long long t_ph = __builtin_readcyclecounter();
#define PSDR_T4PHASE(field) do { if (COUNT) { const unsigned long long t_now = __builtin_readcyclecounter(); S.field += (unsigned) (t_now - t_ph); t_ph = t_now; } } while (0)
#else
#define PSDR_T4PHASE(field) do { } while (0)
#endif
    for (;;) {
        wave_sync();
        const unsigned tested = L.heads[kHdPairTested];
#if PSDR_DIAG == 3
        if (COUNT) S.c_hits++;
#endif
        // a worker whose walk has ended hands the ray back (finished once the pairs up to last_pair are tested) ...
        // (a ray may have several walkers, PSDR_STEAL: the last one to end publishes it; a shared ray waits for every pair enqueued so far, its own are among them)
        if (tr.rid >= 0 && tr.code == kT4Done) {
            const unsigned old = __hip_atomic_fetch_add(&L.fin[tr.rid], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old & kFinCount) == 1u) L.fin[tr.rid] = ((old & kFinShared) ? L.heads[kHdPairEnq] : tr.last_pair) + 1u;
            tr.rid = -1;
        }
        // ... and the workers without a ray take the next ones of the queue
        const unsigned long long m_idle = __ballot(tr.rid < 0);
        const unsigned rq_tail = L.heads[kHdRayTail], rq_head = L.heads[kHdRayHead];
        int avail = (int) (rq_tail - rq_head);            // rays waiting in the queue (after the refill: still waiting)
        if (m_idle != 0ull && avail > 0) {
            const int rank = __popcll(m_idle & lt_mask), n_idle = __popcll(m_idle), n_take = n_idle < avail ? n_idle : avail;
            if (tr.rid < 0 && rank < avail) t4_start<LDS, COUNT>(S, L, tr, (int) L.rq[(rq_head + (unsigned) rank) & (kRayCap - 1)]);
            wave_sync();
            if (lane_id == first) L.heads[kHdRayHead] = rq_head + (unsigned) n_take;
            avail -= n_take;
        }
#if PSDR_STEAL
        // ... and when the queue is empty, the workers still without a ray take over the BOTTOM stack entry - the far subtree - of a walk in progress: the wave
        // issues the node steps anyway, so an idle lane walks for free what its owner would reach last (or never, if a hit culls it: then the thief is culled
        // as well at its next step).  The hit is the minimum over all tested pairs of the ray, whoever enqueued them.  Mailbox = the (empty) ray queue.
        if (avail <= 0) {
#pragma unroll
          for (int round = 0; round < PSDR_STEAL_ROUNDS; ++round) {
            if (round >= PSDR_STEAL_ROUNDS_ASYNC && max_busy != 0) break;      // (the path kernels: one round)
            if (round > 0) wave_sync();
            const unsigned long long m_idle2 = __ballot(tr.rid < 0);
            if (__popcll(m_idle2) >= PSDR_STEAL_IDLE) {
                const bool give = tr.rid >= 0 && tr.code != kT4Done && tr.sp - tr.sb >= PSDR_STEAL_MIN && (((PSDR_STEAL_TOP >> (tr.rid >> 6)) & 1) || tr.sb < T.stack_lds) && ((PSDR_STEAL_KINDS >> (tr.rid >> 6)) & 1);
                const unsigned long long m_give = __ballot(give);
                if (m_give != 0ull) {
                    const int n_idle = __popcll(m_idle2), n_give = __popcll(m_give), n = n_idle < n_give ? n_idle : n_give;
                    if (give) {
                        const int r = __popcll(m_give & lt_mask);
                        if (r < n) {
                            unsigned key;
                            if ((PSDR_STEAL_TOP >> (tr.rid >> 6)) & 1) key = t4_pop(S, L, tr.sp);
                            else { key = (unsigned) t4_stack_base(L)[tr.sb * kBlock]; ++tr.sb; }
                            L.rq[(rq_tail + (unsigned) r) & (kRayCap - 1)] = key;
                            L.rq[(rq_tail + 64u + (unsigned) r) & (kRayCap - 1)] = (unsigned) tr.rid;
                            __hip_atomic_fetch_add(&L.fin[tr.rid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_or(&L.fin[tr.rid], kFinShared, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    wave_sync();
                    if (tr.rid < 0) {
                        const int r = __popcll(m_idle2 & lt_mask);
                        if (r < n) {
                            const unsigned key = L.rq[(rq_tail + (unsigned) r) & (kRayCap - 1)];
                            const int rid = (int) L.rq[(rq_tail + 64u + (unsigned) r) & (kRayCap - 1)];
                            const int owner = rid & 63;
                            const lds_float_t *q = L.park0 + owner + ((rid & 64) ? 6 * kBlock : 0);
                            const Vec3f o(q[0], q[kBlock], q[2 * kBlock]), d(q[3 * kBlock], q[4 * kBlock], q[5 * kBlock]);
                            tr.rid = rid;
                            tr.o = o; tr.d = d;
                            tr.inv = t4_inv_dir(d);
                            tr.sp = 0; tr.sb = 0;
                            tr.last_pair = 0u;
                            tr.anyhit = (rid & 64) ? -__builtin_inff() : L.park0[owner + 12 * kBlock];
                            tr.code = (__uint_as_float(key & ~cmask) <= t4_best_t(L.best, rid)) ? (key & cmask) : kT4Done;
                        }
                    }
                }
            }
          }
        }
#endif
        wave_sync();
        // owners: are my posted rays finished?  (fin = kFinBusy | ... while posted or walked, then 1 + the sequence number after the ray's last pair)
        bool wait_pairs = false;
        done = true;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (posted & (1 << k)) {
                const unsigned f = L.fin[(k << 6) | lane_id];
                if (f & kFinBusy) done = false;
                else if ((int) (f - 1u - tested) > 0) { done = false; wait_pairs = true; }
            }
        PSDR_T4PHASE(c_hits);
        if (__popcll(__ballot(!done)) <= max_busy) break;
        // traversal burst: until the ring holds a wave's worth of pairs, or nobody has a node in hand
        const unsigned long long m_walk = __ballot(tr.code != kT4Done);
        if (m_walk != 0ull) {
            for (;;) {
                // node step, THEN the leaves - also the ones this step arrived at: a worker whose nearest child is a leaf enqueues it and pops its next
                // node in the same iteration instead of sitting out the next node step (round 3; a ray meets ~6 leaves on ~19 nodes)
#if PSDR_DIAG == 11
                if (COUNT) { if (tr.code >= leaf_bit && tr.code != kT4Done) S.c_tris++; else if (tr.rid < 0) S.c_rays++; else if (tr.code == kT4Done) S.c_hits++; }
#elif PSDR_DIAG == 12
#ifndef PSDR_DEEP_AT
#define PSDR_DEEP_AT T.stack_lds
#endif
                if (COUNT) { const bool deep = tr.code < leaf_bit && tr.sp + 3 > PSDR_DEEP_AT; if (deep) S.c_tris++; if (__ballot(deep) != 0ull) S.c_rays++; S.c_hits++; }
#endif
                if (tr.code < leaf_bit) t4_node<LDS, COUNT, FAST>(S, L, tr, cmask);
#pragma unroll
                for (int er = 0; er < PSDR_ENQ_ROUNDS; ++er)
                    if (tr.code >= leaf_bit && tr.code != kT4Done) t4_enqueue<LDS, FAST>(S, L, tr, cmask, leaf_bit);
#if PSDR_DIAG == 1
                if (COUNT) S.c_hits++;
#endif
                wave_sync();
                const int waiting = (int) (L.heads[kHdPairEnq] - tested);
                if (waiting >= (PSDR_PAIR_MIN < n_lanes ? PSDR_PAIR_MIN : n_lanes) || __ballot(tr.code != kT4Done) == 0ull) break;
                // workers that ran out of nodes idle until the burst ends: end it early when many do and rays are waiting
                if (avail > n_lanes / PSDR_BURST_DIV && __popcll(__ballot(tr.code == kT4Done)) >= n_lanes / PSDR_BURST_DIV) break;
#if PSDR_STEAL && PSDR_STEAL_BREAK > 0
                // ... or - in the synchronous form (bvh4_trace2: no shading phase will bring new rays) - when the queue is empty, 1 / PSDR_STEAL_BREAK of the workers idle and
                // walks in progress have entries to give: the hand-over at the top of the outer loop then puts the idle workers on those (config 5's secondary-edge kernel
                // 26.5 -> 24.5 ms; in the path kernels, where finished owners are shaded in between, the same rule costs 1-3 %)
                if (max_busy == 0 && avail <= 0 && __popcll(__ballot(tr.code == kT4Done)) >= n_lanes / PSDR_STEAL_BREAK && __ballot(tr.code != kT4Done && tr.sp > tr.sb) != 0ull) break;
#endif
            }
        }
        PSDR_T4PHASE(c_nodes);
        // test the waiting pairs, a wave's worth at a time; the partial last batch only when a finished walk is waiting for it
        {
            unsigned from = tested;
            const unsigned head = L.heads[kHdPairEnq];
            while ((int) (head - from) >= n_lanes) { t4_test_pairs<LDS, COUNT>(S, L, from, n_lanes); from += (unsigned) n_lanes; }
            if (head != from && ((int) (head - from) >= PSDR_PAIR_MIN || __ballot(wait_pairs) != 0ull || (m_walk == 0ull && avail <= 0))) {
                t4_test_pairs<LDS, COUNT>(S, L, from, (int) (head - from));
                from = head;
            }
            wave_sync();
            if (lane_id == first) L.heads[kHdPairTested] = from;
        }
        PSDR_T4PHASE(c_tris);
    }
#undef PSDR_T4PHASE
    return done;
}

// called once per kernel by every thread of the workgroup (make_view): the ring heads start at zero
template <int LDS> PSDR_DEV void t4_init_lds(const SceneView<LDS> &S) {
    if (S.T->stack_lds > 0) {
        lds_int_t *base = (lds_int_t *) (S.stack - threadIdx.x);
        base[(S.T->stack_lds + kRowHeads) * kBlock + threadIdx.x] = 0;
#if PSDR_TOP_LDS
        // the first kTopNodes nodes, four 16-byte words each
        lds_v4f_t *top = (lds_v4f_t *) (base + (S.T->stack_lds + kRowTop) * kBlock);
        const int n_words = kNodeW4 * (S.T->n_nodes < kTopNodes ? S.T->n_nodes : kTopNodes);
        for (int i = threadIdx.x; i < n_words; i += kBlock) { const float4 v = S.ld(S.T->nodes_off + i); t4_v4f w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; top[i] = w; }
#endif
        __syncthreads();
    }
}

// The owner posts its two rays (called by the owners together): parked copies, empty hit records, entries in the wave's ray queue.
// -> bit mask of the rays posted, to be handed to trav4_run until the hits have been read with t4_result.
template <int LDS> PSDR_DEV int t4_post(const SceneView<LDS> &S, const Vec3f &oA, const Vec3f &dA, bool actA, const Vec3f &oB, const Vec3f &dB, bool actB, float anyhit_a = -__builtin_inff()) {
    const T4Lds<LDS> L(S);
    const int lane_id = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane_id) - 1ull;
    lds_float_t *p = L.park;
    p[0] = oA.x; p[kBlock] = oA.y; p[2 * kBlock] = oA.z; p[3 * kBlock] = dA.x; p[4 * kBlock] = dA.y; p[5 * kBlock] = dA.z;
    p[6 * kBlock] = oB.x; p[7 * kBlock] = oB.y; p[8 * kBlock] = oB.z; p[9 * kBlock] = dB.x; p[10 * kBlock] = dB.y; p[11 * kBlock] = dB.z;
    p[12 * kBlock] = anyhit_a;
    L.best[lane_id] = kT4NoHit; L.best[64 + lane_id] = kT4NoHit;
    L.fin[lane_id] = kFinBusy; L.fin[64 + lane_id] = kFinBusy;
    const unsigned long long mA = __ballot(actA), mB = __ballot(actB), m_all = __ballot(true);
    // queue order: all extension rays (closest hit wanted: the long walks), then the next-event rays (they stop at the first occluder) - the
    // workers take rays in this order, so the long ones start first and the short ones fill the end of the traversal phase
    const int nB = __popcll(mB), total = __popcll(mA) + nB;
    const unsigned tail = L.heads[kHdRayTail];
    wave_sync();
#ifdef PSDR_RQ_INTERLEAVED
    const int before = __popcll(mA & lt_mask) + __popcll(mB & lt_mask);
    if (actA) L.rq[(tail + (unsigned) before) & (kRayCap - 1)] = (unsigned) lane_id;
    if (actB) L.rq[(tail + (unsigned) before + (actA ? 1u : 0u)) & (kRayCap - 1)] = 64u + (unsigned) lane_id;
#else
    if (actB) L.rq[(tail + (unsigned) __popcll(mB & lt_mask)) & (kRayCap - 1)] = 64u + (unsigned) lane_id;
    if (actA) L.rq[(tail + (unsigned) (nB + __popcll(mA & lt_mask))) & (kRayCap - 1)] = (unsigned) lane_id;
#endif
    if (lane_id == (int) __builtin_ctzll(m_all)) L.heads[kHdRayTail] = tail + (unsigned) total;
    wave_sync();
    return (actA ? 1 : 0) | (actB ? 2 : 0);
}

// two rays per lane, run to completion: the synchronous form behind trace() / trace2() (secondary-edge, guiding, adjoint
// recording and ray-batch kernels)
template <int LDS, bool COUNT>
PSDR_DEV void bvh4_trace2(SceneView<LDS> &S, const Vec3f &oA, const Vec3f &dA, bool actA, const Vec3f &oB, const Vec3f &dB, bool actB, Hit &hA, Hit &hB, float anyhit_a) {
    Trav4 tr;
    tr.reset();
    const int posted = t4_post(S, oA, dA, actA, oB, dB, actB, anyhit_a);
    trav4_run<LDS, COUNT>(S, tr, posted, 0);
    hA.slot = -1; hA.u = hA.v = hA.t = 0.f; hB = hA;
    if (actA) hA = t4_result(S, 0);
    if (actB) hB = t4_result(S, 1);
}

} // namespace psdr
