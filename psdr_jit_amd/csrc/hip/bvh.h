// bvh.h — the software BVH that replaces the OptiX GAS (reference src/scene/scene_optix.cpp:265-332 builds a geometry acceleration
// structure with optixAccelBuild in every Scene::configure; MI355X has no ray-tracing unit, so traversal is a HIP loop over this tree,
// trav4.h).  Host side: the builder (binned SAH over all three axes, multi-threaded) and the collapse into 4-wide nodes.  Shared with
// the device: the box of a triangle and the quantisation of a node's child boxes, so that the REFIT kernels of scene_build.hip
// (vertices moved, topology unchanged: psdr_hip_scene_update) write the very bytes the builder would write for the same tree.
//
// Boxes are padded so the slab test can never reject a ray that the triangle test accepts (hit selection must not depend on the
// traversal order): the hit is defined by the exact triangle test alone (trav4.h), the tree only decides which triangles are looked at.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

#include "../common/threads.h"

#if defined(__HIPCC__)
#define PSDR_BVH_HD __host__ __device__
#else
#define PSDR_BVH_HD
#endif

namespace psdr {

// ---------------------------------------------------------------------------------------------------------------------------
// shared by the host builder and the device refit
//
// padded box of the triangle (p0, p0 + e1, p0 + e2): 1e-4 of the coordinate magnitude (at least 1e-4) on every side
PSDR_BVH_HD inline void bvh_tri_box(const float *p0, const float *e1, const float *e2, float *lo, float *hi) {
    for (int k = 0; k < 3; ++k) {
        const float a = p0[k], b = a + e1[k], c = a + e2[k];
        float l = a < b ? a : b; l = l < c ? l : c;
        float h = a > b ? a : b; h = h > c ? h : c;
        const float al = l < 0.f ? -l : l, ah = h < 0.f ? -h : h;
        float m = al > ah ? al : ah; m = m > 1.f ? m : 1.f;
        const float pad = 1e-4f * m;
        lo[k] = l - pad; hi[k] = h + pad;
    }
}

// Width of the tree (children per node).  4: the shipped form.  8 (-DPSDR_BVH_WIDTH=8, measurement build): a node of 128 bytes - one line - with two words per byte plane
// (lo.x[8] ... hi.z[8] at words 4-15) and eight codes at words 16-23; a walk then takes fewer, wider steps.
#ifndef PSDR_BVH_WIDTH
#define PSDR_BVH_WIDTH 4
#endif
constexpr int kBvhW = PSDR_BVH_WIDTH;
static_assert(kBvhW == 4 || kBvhW == 8, "PSDR_BVH_WIDTH");
constexpr int kNodeFloats = kBvhW == 4 ? 16 : 32;
constexpr int kNodeCodeOff = 4 + 6 * (kBvhW / 4);        // word of the first child code (10 / 16)
#ifndef PSDR_TOP_ROWS
#define PSDR_TOP_ROWS 2
#endif
constexpr int kBvhTopNodes = PSDR_TOP_ROWS * 256 / kNodeFloats;      // nodes numbered breadth first from the root (= trav4.h::kTopNodes, the part of the tree a workgroup copies to LDS: scene_dev.h::kTopRows rows)

// 4-wide node = 64 bytes = 4 float4 words (two nodes per 128-byte L2 line, four 16-byte loads per step):
//   w0  origin.xyz (the lower corner of the union of the children's boxes), bits(ex | ey << 8 | ez << 16): biased exponents,
//       the grid step along axis a is 2^(e_a - 127)
//   w1  lo.x[4] lo.y[4] lo.z[4] hi.x[4]      one byte per child: lo = origin + step * q rounded DOWN, hi rounded UP - the
//   w2  hi.y[4] hi.z[4] code[0] code[1]      quantised box contains the (padded) float box, so the traversal visits a superset
//   w3  code[2] code[3] 0 0                  and the hit, defined by the exact triangle test alone, is unchanged
// code of a child: inner node -> its index; leaf -> leaf_bit | (first_triangle << 2 | count - 1); an unused child slot has code
// 0xffffffff (and an inverted / unreachable box).  code < 2^ref_bits, so the traversal packs (coarse entry distance | code) into
// ONE 32-bit sort key (trav4.h).
//
// bvh_quantise: words 0..9 of the node from the float boxes of its nc (1..4) children; the codes (words 10..13) are the builder's.
// smallest power-of-two step with (hi - origin) / step <= 255 after rounding up, per axis; all in double, exact for float inputs.
PSDR_BVH_HD inline void bvh_quantise(const float los[kBvhW][3], const float his[kBvhW][3], int nc, float *q) {
    constexpr int WW = kBvhW / 4;          // words per byte plane
    uint32_t exps = 0, qlo[3][WW] = {}, qhi[3][WW] = {};
    float o3[3];
    for (int a = 0; a < 3; ++a) {
        double mn = los[0][a], mx = his[0][a];
        for (int k = 1; k < nc; ++k) { mn = mn < (double) los[k][a] ? mn : (double) los[k][a]; mx = mx > (double) his[k][a] ? mx : (double) his[k][a]; }
        const double org = mn, ext = mx - mn;
        int e = 0;
        if (ext > 0.0) {
            // ceil(log2(ext / 255)) without a logarithm: x = m 2^k, m in [0.5, 1)
            int k2 = 0;
            const double m = frexp(ext / 255.0, &k2);
            e = (m == 0.5) ? k2 - 1 : k2;
        }
        e = e < -126 ? -126 : (e > 127 ? 127 : e);
        for (;;) {
            const double step = ldexp(1.0, e);
            bool ok = true;
            for (int k = 0; k < nc && ok; ++k) ok = ceil(((double) his[k][a] - org) / step) <= 255.0;
            if (ok || e >= 127) break;
            ++e;
        }
        const double step = ldexp(1.0, e);
        exps |= (uint32_t) (e + 127) << (8 * a);
        for (int k = 0; k < kBvhW; ++k) {
            uint32_t l = 255u, h = 0u;          // unused child: inverted box (and code 0xffffffff, which the traversal checks)
            if (k < nc) {
                double fl = floor(((double) los[k][a] - org) / step), ch = ceil(((double) his[k][a] - org) / step);
                fl = fl < 0.0 ? 0.0 : (fl > 255.0 ? 255.0 : fl); ch = ch < 0.0 ? 0.0 : (ch > 255.0 ? 255.0 : ch);
                l = (uint32_t) fl; h = (uint32_t) ch;
            }
            qlo[a][k >> 2] |= l << (8 * (k & 3)); qhi[a][k >> 2] |= h << (8 * (k & 3));
        }
        o3[a] = (float) org;                   // exact: org is one of the float bounds
    }
    uint32_t *u = reinterpret_cast<uint32_t *>(q);
    q[0] = o3[0]; q[1] = o3[1]; q[2] = o3[2]; u[3] = exps;
    for (int a = 0; a < 3; ++a)
        for (int j = 0; j < WW; ++j) { u[4 + a * WW + j] = qlo[a][j]; u[4 + (3 + a) * WW + j] = qhi[a][j]; }
}

PSDR_BVH_HD inline float bvh_half_area(const float *lo, const float *hi) {
    const float d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
    return d0 * d1 + d1 * d2 + d2 * d0;
}


// ---------------------------------------------------------------------------------------------------------------------------
// host builder
struct BvhResult {
    std::vector<int32_t> order;    // device triangle slot -> original triangle id
    int32_t n_leaves = 0, max_depth = 0;
    // the binary tree (node 0 = root; left < 0: leaf of `count` (1 or 2) triangles starting at slot `first`), input of build_bvh4
    std::vector<int32_t> tmp_left, tmp_right, tmp_first, tmp_count;
    std::vector<float> tmp_box;    // lo.xyz, hi.xyz per node (padded)
};

namespace bvh_detail {
struct Box {
    float lo[3], hi[3];
    Box() { for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<float>::max(); hi[k] = -std::numeric_limits<float>::max(); } }
    void grow(const Box &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    void grow(const float *p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    float half_area() const {
        float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (d[0] < 0.f) return 0.f;
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
    }
};
struct TmpNode { Box box; int left = -1, right = -1, first = 0, count = 0; };
constexpr int kBins = 64;          // config 5: 16 / 32 / 64 bins: 237.9 / 236.7 / 233.8 ms per renderD, 18.34 / 18.23 / 17.94 nodes per ray (round 3)
constexpr int kLeafMax = 2;        // triangles per leaf (measured with the triangle ring: 1: +15 %, 2: -1.5 %, 4: 0; the pair ring of trav4.h is sized for leaves of <= 2)
} // namespace bvh_detail

// Binned SAH over all three axes (64 bins; the cheapest left area x count + right area x count split wins), leaves of at most two triangles.
// The tree is a function of the triangles alone: every split is decided from order-independent sums (box unions, counts) and carried out by a
// stable partition, so `threads` changes the time, not the result (the numbering of the binary nodes differs; build_bvh4 walks the structure).
// Work is handed out per subtree: a job splits its range and queues the two halves until a range is small enough to finish in place.
inline void build_bvh(const float *p0, const float *e1, const float *e2, int n, BvhResult &out, int threads = 0) {
    using namespace bvh_detail;
    if (threads <= 0) threads = host_threads();
    std::vector<Box> tb((size_t) n);
    std::vector<float> ctr(3 * (size_t) n);
    out.order.resize((size_t) n);
    parallel_for((size_t) n, 4096, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            out.order[i] = (int32_t) i;
            // centroid of the UNPADDED box, then the padded box (bvh_tri_box)
            float lo[3], hi[3];
            for (int k = 0; k < 3; ++k) {
                const float a = p0[3 * i + k], bb = a + e1[3 * i + k], c = a + e2[3 * i + k];
                lo[k] = std::min(a, std::min(bb, c)); hi[k] = std::max(a, std::max(bb, c));
                ctr[3 * i + k] = 0.5f * (lo[k] + hi[k]);
            }
            bvh_tri_box(p0 + 3 * i, e1 + 3 * i, e2 + 3 * i, tb[i].lo, tb[i].hi);
        }
    }, threads);
    const size_t cap = 2 * (size_t) std::max(n, 1) + 2;
    std::vector<TmpNode> tmp(cap);
    std::atomic<int> n_tmp{1};
    std::atomic<int> max_depth{1};
    struct Job { int node, first, count, depth; };

    // one node: its box, and either a leaf or a split of out.order[first, first + count) -> true: children allocated (left, right ranges in l / r)
    auto split = [&](const Job &j, Job &l, Job &r) -> bool {
        Box box, cb;
        for (int i = j.first; i < j.first + j.count; ++i) { box.grow(tb[out.order[i]]); cb.grow(&ctr[3 * (size_t) out.order[i]]); }
        TmpNode nd; nd.box = box; nd.first = j.first; nd.count = j.count;
        bool inner = false;
        if (j.count > kLeafMax || (j.node == 0 && j.count > 1)) {
            int mid = -1, best_axis = -1, best_split = -1;
            float best = std::numeric_limits<float>::max();
            auto bin_of = [&](int t, int axis) {
                const float ext = cb.hi[axis] - cb.lo[axis];
                int b = (int) ((ctr[3 * (size_t) t + axis] - cb.lo[axis]) * (kBins / ext));
                return std::min(std::max(b, 0), kBins - 1);
            };
            for (int axis = 0; axis < 3; ++axis) {
                if (!(cb.hi[axis] - cb.lo[axis] > 0.f)) continue;
                Box bb[kBins]; int bc[kBins] = {0};
                for (int i = j.first; i < j.first + j.count; ++i) { const int t = out.order[i]; const int b = bin_of(t, axis); bb[b].grow(tb[t]); bc[b]++; }
                Box rb[kBins]; int rc[kBins];
                Box acc; int cnt = 0;
                for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); cnt += bc[b]; rb[b] = acc; rc[b] = cnt; }
                acc = Box(); cnt = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bb[b]); cnt += bc[b];
                    if (cnt == 0 || rc[b + 1] == 0) continue;
                    const float cost = acc.half_area() * (float) cnt + rb[b + 1].half_area() * (float) rc[b + 1];
                    if (cost < best) { best = cost; best_split = b; best_axis = axis; }
                }
            }
            if (best_axis >= 0) {
                auto it = std::stable_partition(out.order.begin() + j.first, out.order.begin() + j.first + j.count,
                                                [&](int t) { return bin_of(t, best_axis) <= best_split; });
                mid = (int) (it - out.order.begin());
            }
            if (mid <= j.first || mid >= j.first + j.count) mid = j.first + j.count / 2;       // degenerate: split by index
            const int at = n_tmp.fetch_add(2);
            nd.left = at; nd.right = at + 1; nd.count = 0;
            l = {nd.left, j.first, mid - j.first, j.depth + 1};
            r = {nd.right, mid, j.first + j.count - mid, j.depth + 1};
            inner = true;
        }
        tmp[(size_t) j.node] = nd;
        int d = max_depth.load();
        while (j.depth > d && !max_depth.compare_exchange_weak(d, j.depth)) {}
        return inner;
    };
    // a whole subtree on the calling thread
    auto finish = [&](const Job &root) {
        std::vector<Job> st;
        st.push_back(root);
        while (!st.empty()) {
            const Job j = st.back(); st.pop_back();
            Job l, r;
            if (split(j, l, r)) { st.push_back(l); st.push_back(r); }
        }
    };
    if (threads <= 1 || n < 8192) {
        finish({0, 0, n, 1});
    } else {
        const int small = std::max(2048, n / (8 * threads));         // ranges of at most this many triangles are finished in place
        std::mutex mu;
        std::condition_variable cv;
        std::vector<Job> queue;
        int pending = 1;                                             // jobs queued or being worked on
        queue.push_back({0, 0, n, 1});
        auto worker = [&] {
            for (;;) {
                Job j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !queue.empty() || pending == 0; });
                    if (queue.empty()) return;
                    // largest range first: the long jobs start early
                    size_t best = 0;
                    for (size_t i = 1; i < queue.size(); ++i) if (queue[i].count > queue[best].count) best = i;
                    j = queue[best]; queue[best] = queue.back(); queue.pop_back();
                }
                if (j.count <= small) finish(j);
                else {
                    Job l, r;
                    if (split(j, l, r)) {
                        std::lock_guard<std::mutex> lk(mu);
                        queue.push_back(l); queue.push_back(r); pending += 2;
                    }
                }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    --pending;
                }
                cv.notify_all();
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < threads; ++t) th.emplace_back(worker);
        worker();
        for (std::thread &t : th) t.join();
    }
    const size_t nt = (size_t) n_tmp.load();
    out.max_depth = max_depth.load();
    out.n_leaves = 0;
    out.tmp_left.resize(nt); out.tmp_right.resize(nt); out.tmp_first.resize(nt); out.tmp_count.resize(nt); out.tmp_box.resize(6 * nt);
    for (size_t i = 0; i < nt; ++i) {
        out.tmp_left[i] = tmp[i].left; out.tmp_right[i] = tmp[i].right; out.tmp_first[i] = tmp[i].first; out.tmp_count[i] = tmp[i].left >= 0 ? 0 : tmp[i].count;
        if (tmp[i].left < 0) out.n_leaves++;
        for (int k = 0; k < 3; ++k) { out.tmp_box[6 * i + k] = tmp[i].box.lo[k]; out.tmp_box[6 * i + 3 + k] = tmp[i].box.hi[k]; }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// 4-wide BVH for the device traversal (trav4.h): the binary SAH tree above with up to two levels collapsed into one node
// (the child with the largest surface area is opened first).  max_stack bounds the traversal stack of any ray (sum over a
// root-to-leaf path of the siblings left behind).  `height[i]` = levels of inner nodes below node i (0: all children are leaves): a
// refit processes the nodes by increasing height.  `cost` = the tree's SAH cost, sum over all child boxes of
// half_area(child) x (triangles of a leaf | 1 for an inner node) / half_area(root box) - what a refit re-evaluates to decide whether
// the tree still fits the geometry (bvh4_cost_term).
struct Bvh4Result {
    std::vector<float> nodes;      // kNodeFloats per node
    std::vector<int32_t> height;   // per node
    int32_t n_nodes = 0, max_depth = 0, max_stack = 0, ref_bits = 0;
    uint32_t leaf_bit = 0;
    double cost = 0.0;
};

inline void build_bvh4(const BvhResult &b2, int n_tris, Bvh4Result &out) {
    const size_t nt = b2.tmp_left.size();
    auto is_leaf = [&](int i) { return b2.tmp_left[i] < 0; };
    auto half_area = [&](int i) { const float *b = &b2.tmp_box[6 * (size_t) i]; return bvh_half_area(b, b + 3); };
    // payload bits: the largest leaf payload is ((n_tris - 1) << 2) | 3, inner indices are smaller than the number of binary nodes
    uint32_t max_payload = (uint32_t) std::max<int64_t>(((int64_t) std::max(n_tris, 1) - 1) * 4 + 3, (int64_t) nt);
    int bits = 1;
    while ((1ull << bits) <= max_payload) ++bits;
    out.leaf_bit = 1u << bits;
    out.ref_bits = bits + 1;
    out.nodes.clear(); out.height.clear(); out.cost = 0.0;
    if (nt == 0) { out.n_nodes = 0; return; }
    struct Item { int tmp; int node4; int depth; };
    std::vector<Item> todo;
    std::vector<int32_t> kid;            // per node4: up to four inner children (-1 = none)
    std::vector<int32_t> n_children;
    out.nodes.reserve((size_t) kNodeFloats * (nt / 2 + 2));
    auto new_node = [&]() {
        out.nodes.resize(out.nodes.size() + kNodeFloats, 0.f);
        kid.resize(kid.size() + kBvhW, -1); n_children.push_back(0);
        return (int) (out.nodes.size() / kNodeFloats) - 1;
    };
    const int root = new_node();
    todo.push_back({0, root, 1});
    int max_depth = 1;
    const double root_area = std::max((double) half_area(0), 1e-300);
    // numbering: the first kBvhTopNodes nodes breadth first - the top levels, which every walk passes and trav4.h keeps in LDS -, the rest depth first (a
    // subtree's nodes next to each other)
    size_t head = 0;
    while (head < todo.size()) {
        Item it;
        if ((int) (out.nodes.size() / kNodeFloats) < kBvhTopNodes) it = todo[head++];
        else { it = todo.back(); todo.pop_back(); }
        max_depth = std::max(max_depth, it.depth);
        int list[kBvhW], nl = 0;
        if (is_leaf(it.tmp)) list[nl++] = it.tmp;          // a scene of one leaf: the root holds it as its only child
        else { list[nl++] = b2.tmp_left[it.tmp]; list[nl++] = b2.tmp_right[it.tmp]; }
        while (nl < kBvhW) {
            int best = -1; float ba = -1.f;
            for (int k = 0; k < nl; ++k) if (!is_leaf(list[k]) && half_area(list[k]) > ba) { ba = half_area(list[k]); best = k; }
            if (best < 0) break;
            const int t = list[best];
            list[best] = b2.tmp_left[t];
            list[nl++] = b2.tmp_right[t];
        }
        n_children[it.node4] = nl;
        uint32_t codes[kBvhW];
        float los[kBvhW][3], his[kBvhW][3];
        for (int k = 0; k < kBvhW; ++k) {
            uint32_t code = 0xffffffffu;
            for (int a = 0; a < 3; ++a) { los[k][a] = 3e38f; his[k][a] = 3e38f; }
            if (k < nl) {
                const int t = list[k];
                for (int a = 0; a < 3; ++a) { los[k][a] = b2.tmp_box[6 * (size_t) t + a]; his[k][a] = b2.tmp_box[6 * (size_t) t + 3 + a]; }
                if (is_leaf(t)) {
                    code = out.leaf_bit | (uint32_t) ((b2.tmp_first[t] << 2) | (b2.tmp_count[t] - 1));
                    out.cost += (double) half_area(t) * b2.tmp_count[t] / root_area;
                } else {
                    const int c = new_node();          // (grows out.nodes: no pointer into it is held across this call)
                    kid[kBvhW * (size_t) it.node4 + k] = c;
                    code = (uint32_t) c;
                    todo.push_back({t, c, it.depth + 1});
                    out.cost += (double) half_area(t) / root_area;
                }
            }
            codes[k] = code;
        }
        float *q = &out.nodes[(size_t) kNodeFloats * (size_t) it.node4];
        bvh_quantise(los, his, nl, q);
        std::memcpy(&q[kNodeCodeOff], &codes[0], 4 * kBvhW);
    }
    out.n_nodes = (int) (out.nodes.size() / kNodeFloats);
    out.max_depth = max_depth;
    // stack need and height: children are emitted after their parent, so a reverse sweep sees every child before its parent
    std::vector<int> need((size_t) out.n_nodes, 0);
    out.height.assign((size_t) out.n_nodes, 0);
    for (int i = out.n_nodes - 1; i >= 0; --i) {
        int below = 0, h = 0;
        for (int k = 0; k < kBvhW; ++k) {
            const int c = kid[kBvhW * (size_t) i + k];
            if (c >= 0) { below = std::max(below, need[c]); h = std::max(h, out.height[c] + 1); }
        }
        need[i] = (n_children[i] - 1) + below;
        out.height[i] = h;
    }
    out.max_stack = need.empty() ? 1 : need[0] + 1;
}

} // namespace psdr
