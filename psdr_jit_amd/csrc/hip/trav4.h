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

struct Trav4 {
    Vec3f o, d, inv;            // the ray being traced
    unsigned code;              // node / leaf in hand, kT4Done = none
    int sp;                     // stack pointer
    int cur;                    // 0 / 1: which of the lane's two rays is being traced, -1 none
    int pending;                // bit k: ray k still waits
    float best_t; int best_id;  // closest hit so far of the ray being traced
    int best_slot; float best_u, best_v;
    Hit hA;                     // result of ray 0 once it has finished
    PSDR_DEV void reset() { code = kT4Done; sp = 0; cur = -1; pending = 0; best_t = 0.f; best_id = 0; best_slot = -1; best_u = best_v = 0.f; hA.slot = -1; hA.u = hA.v = hA.t = 0.f; o = Vec3f(0.f); d = Vec3f(0.f); inv = Vec3f(0.f); }
    PSDR_DEV bool idle() const { return cur < 0 && pending == 0; }
    PSDR_DEV Hit result() const { Hit h; h.slot = best_slot; h.u = best_u; h.v = best_v; h.t = best_slot >= 0 ? best_t : 0.f; return h; }
};

// (the two halves of the stack are addressed through address-space qualified pointers: a select between an LDS and a global
// generic pointer makes hipcc 7.2 emit an illegal v_cmp on src_shared_base)
typedef __attribute__((address_space(3))) int lds_int_t;
typedef __attribute__((address_space(1))) int glb_int_t;
template <int LDS> PSDR_DEV void t4_push(const SceneView<LDS> &S, int &sp, unsigned key) {
    const SceneTables &T = *S.T;
    if (sp < T.stack_lds) ((lds_int_t *) S.stack)[sp * kBlock] = (int) key;
    else ((glb_int_t *) T.gstack)[(size_t) (sp - T.stack_lds) * T.gstack_stride + (size_t) blockIdx.x * kBlock + threadIdx.x] = (int) key;
    ++sp;
}
template <int LDS> PSDR_DEV unsigned t4_pop(const SceneView<LDS> &S, int &sp) {
    const SceneTables &T = *S.T;
    --sp;
    unsigned key;
    if (sp < T.stack_lds) key = (unsigned) ((lds_int_t *) S.stack)[sp * kBlock];
    else key = (unsigned) ((glb_int_t *) T.gstack)[(size_t) (sp - T.stack_lds) * T.gstack_stride + (size_t) blockIdx.x * kBlock + threadIdx.x];
    return key;
}

// next node of this lane's ray: pop until an entry is not behind the closest hit; kT4Done when the stack is empty
template <int LDS> PSDR_DEV unsigned t4_next(const SceneView<LDS> &S, Trav4 &tr, unsigned cmask) {
    while (tr.sp > 0) {
        const unsigned key = t4_pop(S, tr.sp);
        if (__uint_as_float(key & ~cmask) <= tr.best_t) return key & cmask;
    }
    return kT4Done;
}

// start ray `k` of the lane (o, d given): NaN rays miss (reference scene_optix.cpp:348-353)
template <int LDS, bool COUNT> PSDR_DEV void t4_start(SceneView<LDS> &S, Trav4 &tr, int k, const Vec3f &o, const Vec3f &d) {
    tr.cur = k;
    tr.o = o; tr.d = d;
    tr.inv = Vec3f(1.f / d.x, 1.f / d.y, 1.f / d.z);
    tr.best_t = __builtin_inff(); tr.best_id = 0x7fffffff; tr.best_slot = -1; tr.best_u = tr.best_v = 0.f;
    tr.sp = 0;
    const bool ok = (o.x == o.x && o.y == o.y && o.z == o.z && d.x == d.x && d.y == d.y && d.z == d.z);
    tr.code = ok ? 0u : kT4Done;       // node 0 = root
    if (COUNT) { if (ok) S.c_rays++; }
}

// one inner node for the lanes whose code is an inner node
template <int LDS, bool COUNT> PSDR_DEV void t4_node(SceneView<LDS> &S, Trav4 &tr, unsigned cmask) {
    const SceneTables &T = *S.T;
    const int w = T.nodes_off + 8 * (int) tr.code;
    const float4 lx = S.ld(w), ly = S.ld(w + 1), lz = S.ld(w + 2), hx = S.ld(w + 3), hy = S.ld(w + 4), hz = S.ld(w + 5), cd = S.ld(w + 6);
    if (COUNT) S.c_nodes++;
    const float ox = tr.o.x, oy = tr.o.y, oz = tr.o.z, ix = tr.inv.x, iy = tr.inv.y, iz = tr.inv.z, bt = tr.best_t;
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
    if (k3 != kT4Miss) t4_push(S, tr.sp, k3);
    if (k2 != kT4Miss) t4_push(S, tr.sp, k2);
    if (k1 != kT4Miss) t4_push(S, tr.sp, k1);
    tr.code = k0 != kT4Miss ? (k0 & cmask) : t4_next(S, tr, cmask);
}

// the leaf in hand: exact tests on its 1..4 triangles, then the next node
template <int LDS, bool COUNT> PSDR_DEV void t4_leaf(SceneView<LDS> &S, Trav4 &tr, unsigned cmask, unsigned leaf_bit) {
    const SceneTables &T = *S.T;
    const int payload = (int) (tr.code & (leaf_bit - 1u)), first = payload >> 2, cnt = (payload & 3) + 1;
    for (int k = 0; k < cnt; ++k) {
        const int w = T.trav_off + 3 * (first + k);
        const float4 a = S.ld(w), b = S.ld(w + 1), c = S.ld(w + 2);
        float u, v, t;
        if (COUNT) S.c_tris++;
        if (tri_test(a, b, c, tr.o, tr.d, u, v, t)) {
            const int id = __float_as_int(c.y);
            if (t < tr.best_t || (t == tr.best_t && id < tr.best_id)) { tr.best_t = t; tr.best_id = id; tr.best_slot = first + k; tr.best_u = u; tr.best_v = v; }
        }
    }
    tr.code = t4_next(S, tr, cmask);
}

// Runs the lanes' ray queues.  `rays(k, o, d)` hands out ray k of this lane when its turn comes.  Returns as soon as at most
// `max_busy` lanes still have rays to trace (0: run to completion).  May be called under a partial exec mask (only the
// active lanes are counted).  When a lane's ray 0 finishes its hit is kept in tr.hA; the hit of ray 1 is tr.result() once the
// lane is idle.
template <int LDS, bool COUNT, typename Rays>
PSDR_DEV void trav4_run(SceneView<LDS> &S, Trav4 &tr, int max_busy, Rays rays) {
    const SceneTables &T = *S.T;
    const unsigned cmask = (1u << T.ref_bits) - 1u, leaf_bit = 1u << (T.ref_bits - 1);
    for (;;) {
        if (tr.code == kT4Done) {
            if (tr.cur >= 0) {                     // a ray has just finished: ray 0's hit moves to hA, ray 1's stays in best_*
                if (tr.cur == 0) { tr.hA = tr.result(); tr.best_slot = -1; tr.best_u = tr.best_v = 0.f; }
                tr.cur = -1;
            }
            if (tr.pending != 0) {                 // next ray of this lane's queue
                const int k = (tr.pending & 1) ? 0 : 1;
                tr.pending &= ~(1 << k);
                Vec3f o, d;
                rays(k, o, d);
                t4_start<LDS, COUNT>(S, tr, k, o, d);
            }
        }
        if (__popcll(__ballot(!tr.idle())) <= max_busy) break;
        while (tr.code < leaf_bit) t4_node<LDS, COUNT>(S, tr, cmask);           // (kT4Done is not below leaf_bit)
        if (tr.code != kT4Done) t4_leaf<LDS, COUNT>(S, tr, cmask, leaf_bit);
    }
}

// two rays per lane, run to completion: the synchronous form behind trace() / trace2() (secondary-edge, guiding, adjoint
// recording and ray-batch kernels)
template <int LDS, bool COUNT>
PSDR_DEV void bvh4_trace2(SceneView<LDS> &S, const Vec3f &oA, const Vec3f &dA, bool actA, const Vec3f &oB, const Vec3f &dB, bool actB, Hit &hA, Hit &hB) {
    Trav4 tr;
    tr.reset();
    tr.pending = (actA ? 1 : 0) | (actB ? 2 : 0);
    trav4_run<LDS, COUNT>(S, tr, 0, [&](int k, Vec3f &o, Vec3f &d) { o = k == 0 ? oA : oB; d = k == 0 ? dA : dB; });
    hA = tr.hA;
    hB = actB ? tr.result() : Hit{-1, 0.f, 0.f, 0.f};
}

} // namespace psdr
