// bvh.h — host-side builder of the software BVH that replaces the OptiX GAS
// (reference src/scene/scene_optix.cpp:265-332 builds a geometry acceleration structure with
// optixAccelBuild; MI355X has no ray-tracing unit, so traversal is a HIP loop over this tree).
//
// Layout (one 64-byte node = 4 x float4, fetched with four ds_read_b128 / one 64 B global line):
//   q0 = left.lo.xyz , bits(left_ref)      q1 = left.hi.xyz , bits(right_ref)
//   q2 = right.lo.xyz, 0                   q3 = right.hi.xyz, 0
// ref >= 0 : index of an inner node;  ref < 0 : leaf, ~ref = (first_triangle << 2) | (count - 1),
// 1..4 triangles, triangles re-ordered so that a leaf's triangles are contiguous.
// Binned SAH (64 bins); boxes are padded so the slab test can never reject a ray that the
// triangle test accepts (hit selection must not depend on the traversal order).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace psdr {

struct BvhResult {
    std::vector<float> nodes;      // 16 floats per node
    std::vector<int32_t> order;    // device triangle slot -> original triangle id
    int32_t n_nodes = 0, n_leaves = 0, max_depth = 0;
    // the binary tree itself (node 0 = root; left < 0: leaf of `count` triangles starting at slot `first`), input of build_bvh4
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
inline int32_t float_bits(int32_t v) { return v; }
} // namespace bvh_detail

inline void build_bvh(const float *p0, const float *e1, const float *e2, int n, BvhResult &out) {
    using namespace bvh_detail;
    // measurement knobs (tools/bvh_knobs.sh): bins of the binned SAH, exact sweep SAH (every split of the centroid order of every axis) for ranges of
    // at most PSDR_BVH_SWEEP triangles, split cost in leaves (area x ceil(count / kLeafMax)) instead of triangles
    constexpr int kMaxBins = 64;
    static const int kBins = std::getenv("PSDR_BVH_BINS") ? std::max(4, std::min(kMaxBins, std::atoi(std::getenv("PSDR_BVH_BINS")))) : 64;       // config 5: 16 / 32 / 64 bins: 237.9 / 236.7 / 233.8 ms, 18.34 / 18.23 / 17.94 nodes per ray
    static const int kSweep = std::getenv("PSDR_BVH_SWEEP") ? std::atoi(std::getenv("PSDR_BVH_SWEEP")) : 0;
    static const bool kLeafCost = std::getenv("PSDR_BVH_LEAFCOST") != nullptr;
    // triangles per leaf: 1 or 2 (measured with the triangle ring: 1: +15 %, 2: -1.5 %, 4: 0; the pair ring of trav4.h is sized for leaves of <= 2)
    static const int kLeafMax = std::getenv("PSDR_BVH_LEAF") ? std::max(1, std::min(2, std::atoi(std::getenv("PSDR_BVH_LEAF")))) : 2;
    std::vector<Box> tb(n);
    std::vector<float> ctr(3 * (size_t) n);
    out.order.resize(n);
    for (int i = 0; i < n; ++i) {
        out.order[i] = i;
        float a[3], b[3], c[3];
        for (int k = 0; k < 3; ++k) { a[k] = p0[3 * i + k]; b[k] = a[k] + e1[3 * i + k]; c[k] = a[k] + e2[3 * i + k]; }
        tb[i].grow(a); tb[i].grow(b); tb[i].grow(c);
        for (int k = 0; k < 3; ++k) {
            ctr[3 * (size_t) i + k] = 0.5f * (tb[i].lo[k] + tb[i].hi[k]);
            float pad = 1e-4f * std::max(1.f, std::max(std::fabs(tb[i].lo[k]), std::fabs(tb[i].hi[k])));
            tb[i].lo[k] -= pad; tb[i].hi[k] += pad;
        }
    }
    std::vector<TmpNode> tmp;
    tmp.reserve(2 * (size_t) n + 2);
    struct Job { int node, first, count, depth; };
    std::vector<Job> stack;
    tmp.emplace_back();
    stack.push_back({0, 0, n, 1});
    int max_depth = 1;
    while (!stack.empty()) {
        Job j = stack.back(); stack.pop_back();
        max_depth = std::max(max_depth, j.depth);
        Box box, cb;
        for (int i = j.first; i < j.first + j.count; ++i) { box.grow(tb[out.order[i]]); cb.grow(&ctr[3 * (size_t) out.order[i]]); }
        TmpNode nd; nd.box = box; nd.first = j.first; nd.count = j.count;
        if (j.count > kLeafMax || (j.node == 0 && j.count > 1)) {
            // binned SAH over all three axes: the cheapest (left area x count + right area x count) split wins
            int mid = -1, best_axis = -1, best_split = -1;
            float best = std::numeric_limits<float>::max();
            auto bin_of = [&](int t, int axis) {
                const float ext = cb.hi[axis] - cb.lo[axis];
                int b = (int) ((ctr[3 * (size_t) t + axis] - cb.lo[axis]) * (kBins / ext));
                return std::min(std::max(b, 0), kBins - 1);
            };
            int widest = 0;
            for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[widest] - cb.lo[widest]) widest = k;
            static const bool all_axes = std::getenv("PSDR_BVH_WIDEST_AXIS") == nullptr;
            auto units = [&](int c) { return kLeafCost ? (float) ((c + kLeafMax - 1) / kLeafMax) : (float) c; };
            bool swept = false;
            if (j.count <= kSweep) {
                // exact sweep: the triangles in centroid order along each axis, every prefix / suffix split priced
                std::vector<int> ord(out.order.begin() + j.first, out.order.begin() + j.first + j.count);
                std::vector<Box> suf(j.count);
                int best_k = -1, sweep_axis = -1;
                auto sort_by = [&](int axis) { std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return ctr[3 * (size_t) a + axis] < ctr[3 * (size_t) b + axis]; }); };
                for (int axis = 0; axis < 3; ++axis) {
                    std::copy(out.order.begin() + j.first, out.order.begin() + j.first + j.count, ord.begin());
                    sort_by(axis);
                    Box acc;
                    for (int i = j.count - 1; i > 0; --i) { acc.grow(tb[ord[i]]); suf[i] = acc; }
                    acc = Box();
                    for (int i = 0; i < j.count - 1; ++i) {
                        acc.grow(tb[ord[i]]);
                        const float cost = acc.half_area() * units(i + 1) + suf[i + 1].half_area() * units(j.count - i - 1);
                        if (cost < best) { best = cost; best_k = i + 1; sweep_axis = axis; }
                    }
                }
                if (best_k > 0) {
                    std::copy(out.order.begin() + j.first, out.order.begin() + j.first + j.count, ord.begin());
                    sort_by(sweep_axis);
                    std::copy(ord.begin(), ord.end(), out.order.begin() + j.first); mid = j.first + best_k; swept = true;
                }
            }
            for (int axis = 0; axis < 3 && !swept; ++axis) {
                if (!all_axes && axis != widest) continue;
                if (!(cb.hi[axis] - cb.lo[axis] > 0.f)) continue;
                Box bb[kMaxBins]; int bc[kMaxBins] = {0};
                for (int i = j.first; i < j.first + j.count; ++i) { int t = out.order[i]; int b = bin_of(t, axis); bb[b].grow(tb[t]); bc[b]++; }
                Box r[kMaxBins]; int rc[kMaxBins];
                Box acc; int cnt = 0;
                for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); cnt += bc[b]; r[b] = acc; rc[b] = cnt; }
                acc = Box(); cnt = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bb[b]); cnt += bc[b];
                    if (cnt == 0 || rc[b + 1] == 0) continue;
                    float cost = acc.half_area() * units(cnt) + r[b + 1].half_area() * units(rc[b + 1]);
                    if (cost < best) { best = cost; best_split = b; best_axis = axis; }
                }
            }
            if (best_axis >= 0 && !swept) {
                auto it = std::stable_partition(out.order.begin() + j.first, out.order.begin() + j.first + j.count,
                                                [&](int t) { return bin_of(t, best_axis) <= best_split; });
                mid = (int) (it - out.order.begin());
            }
            if (mid <= j.first || mid >= j.first + j.count) {   // degenerate: split by index
                mid = j.first + j.count / 2;
            }
            nd.left = (int) tmp.size(); tmp.emplace_back();
            nd.right = (int) tmp.size(); tmp.emplace_back();
            nd.count = 0;
            stack.push_back({nd.left, j.first, mid - j.first, j.depth + 1});
            stack.push_back({nd.right, mid, j.first + j.count - mid, j.depth + 1});
        }
        tmp[j.node] = nd;
    }
    // emit two-box nodes; inner tmp nodes get device indices in DFS order
    std::vector<int> dev_index(tmp.size(), -1);
    int n_inner = 0, n_leaves = 0;
    for (size_t i = 0; i < tmp.size(); ++i) { if (tmp[i].left >= 0) dev_index[i] = n_inner++; else n_leaves++; }
    bool single_leaf_root = (tmp[0].left < 0);
    if (single_leaf_root) n_inner = 1;
    out.nodes.assign(16 * (size_t) n_inner, 0.f);
    auto ref_of = [&](int ti) -> int32_t {
        const TmpNode &t = tmp[ti];
        if (t.left >= 0) return dev_index[ti];
        return ~((t.first << 2) | (t.count - 1));
    };
    auto put = [&](float *q, const Box &b, int32_t ref_lo_slot, int32_t ref_hi_slot, bool with_refs) {
        q[0] = b.lo[0]; q[1] = b.lo[1]; q[2] = b.lo[2];
        q[4] = b.hi[0]; q[5] = b.hi[1]; q[6] = b.hi[2];
        if (with_refs) { std::memcpy(&q[3], &ref_lo_slot, 4); std::memcpy(&q[7], &ref_hi_slot, 4); }
    };
    if (single_leaf_root) {
        float *q = &out.nodes[0];
        Box empty;   // lo > hi: never hit
        int32_t lref = n > 0 ? ref_of(0) : ~0, rref = lref;
        put(q, n > 0 ? tmp[0].box : empty, lref, rref, true);
        put(q + 8, empty, 0, 0, false);
    } else {
        for (size_t i = 0; i < tmp.size(); ++i) {
            if (tmp[i].left < 0) continue;
            float *q = &out.nodes[16 * (size_t) dev_index[i]];
            put(q, tmp[tmp[i].left].box, ref_of(tmp[i].left), ref_of(tmp[i].right), true);
            put(q + 8, tmp[tmp[i].right].box, 0, 0, false);
        }
    }
    out.n_nodes = n_inner; out.n_leaves = n_leaves; out.max_depth = max_depth;
    out.tmp_left.resize(tmp.size()); out.tmp_right.resize(tmp.size()); out.tmp_first.resize(tmp.size()); out.tmp_count.resize(tmp.size()); out.tmp_box.resize(6 * tmp.size());
    for (size_t i = 0; i < tmp.size(); ++i) {
        out.tmp_left[i] = tmp[i].left; out.tmp_right[i] = tmp[i].right; out.tmp_first[i] = tmp[i].first; out.tmp_count[i] = tmp[i].left >= 0 ? 0 : tmp[i].count;
        for (int k = 0; k < 3; ++k) { out.tmp_box[6 * i + k] = tmp[i].box.lo[k]; out.tmp_box[6 * i + 3 + k] = tmp[i].box.hi[k]; }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// 4-wide BVH for the device traversal (trav4.h): the binary SAH tree above with up to two levels collapsed into one node
// (the child with the largest surface area is opened first).
//
// One node = 64 bytes = 4 float4 words (round 3; two nodes per 128-byte L2 line, four 16-byte loads per step):
//   w0  origin.xyz (the lower corner of the union of the children's boxes), bits(ex | ey << 8 | ez << 16): biased exponents,
//       the grid step along axis a is 2^(e_a - 127)
//   w1  lo.x[4] lo.y[4] lo.z[4] hi.x[4]      one byte per child: lo = origin + step * q rounded DOWN, hi rounded UP - the
//   w2  hi.y[4] hi.z[4] code[0] code[1]      quantised box contains the (padded) float box, so the traversal visits a superset
//   w3  code[2] code[3] 0 0                  and the hit, defined by the exact triangle test alone, is unchanged
// With -DPSDR_NODE128 (measurement knob) the node is round 2's 128 bytes = 8 words: lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4]
// code[4] unused, boxes as floats.
// code of a child: inner node -> its index; leaf -> leaf_bit | (first_triangle << 2 | count - 1); an unused child slot has code
// 0xffffffff (and an inverted / unreachable box).  code < 2^ref_bits, so the traversal packs (coarse entry distance | code) into
// ONE 32-bit sort key (trav4.h).  max_stack bounds the traversal stack of any ray (sum over a root-to-leaf path of the siblings
// left behind).
#ifdef PSDR_NODE128
constexpr int kNodeFloats = 32;
#else
constexpr int kNodeFloats = 16;
#endif
struct Bvh4Result {
    std::vector<float> nodes;      // kNodeFloats per node
    int32_t n_nodes = 0, max_depth = 0, max_stack = 0, ref_bits = 0;
    uint32_t leaf_bit = 0;
};

inline void build_bvh4(const BvhResult &b2, int n_tris, Bvh4Result &out) {
    const size_t nt = b2.tmp_left.size();
    auto is_leaf = [&](int i) { return b2.tmp_left[i] < 0; };
    auto half_area = [&](int i) {
        const float *b = &b2.tmp_box[6 * (size_t) i];
        const float d0 = b[3] - b[0], d1 = b[4] - b[1], d2 = b[5] - b[2];
        return d0 * d1 + d1 * d2 + d2 * d0;
    };
    // payload bits: the largest leaf payload is ((n_tris - 1) << 2) | 3, inner indices are smaller than the number of binary nodes
    uint32_t max_payload = (uint32_t) std::max<int64_t>(((int64_t) std::max(n_tris, 1) - 1) * 4 + 3, (int64_t) nt);
    int bits = 1;
    while ((1ull << bits) <= max_payload) ++bits;
    out.leaf_bit = 1u << bits;
    out.ref_bits = bits + 1;
    struct Item { int tmp; int node4; int depth; };
    std::vector<Item> todo;
    std::vector<int> need;          // stack need below each emitted node (filled bottom-up afterwards)
    std::vector<std::vector<int>> kids;   // per node4: child node4 indices (inner children only)
    out.nodes.clear();
    auto new_node = [&]() { out.nodes.resize(out.nodes.size() + kNodeFloats, 0.f); kids.emplace_back(); return (int) (out.nodes.size() / kNodeFloats) - 1; };
    std::vector<int> n_children;
    if (nt == 0) { out.n_nodes = 0; return; }
    const int root = new_node();
    n_children.push_back(0);
    todo.push_back({0, root, 1});
    int max_depth = 1;
    while (!todo.empty()) {
        const Item it = todo.back(); todo.pop_back();
        max_depth = std::max(max_depth, it.depth);
        std::vector<int> list;
        if (is_leaf(it.tmp)) list.push_back(it.tmp);          // a scene of one leaf: the root holds it as its only child
        else { list.push_back(b2.tmp_left[it.tmp]); list.push_back(b2.tmp_right[it.tmp]); }
        while (list.size() < 4) {
            int best = -1; float ba = -1.f;
            for (size_t k = 0; k < list.size(); ++k) if (!is_leaf(list[k]) && half_area(list[k]) > ba) { ba = half_area(list[k]); best = (int) k; }
            if (best < 0) break;
            const int t = list[best];
            list[best] = b2.tmp_left[t];
            list.push_back(b2.tmp_right[t]);
        }
        n_children[it.node4] = (int) list.size();
        uint32_t codes[4];
        float los[4][3], his[4][3];
        for (int k = 0; k < 4; ++k) {
            uint32_t code = 0xffffffffu;
            float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {3e38f, 3e38f, 3e38f};
            if (k < (int) list.size()) {
                const int t = list[k];
                for (int a = 0; a < 3; ++a) { lo[a] = b2.tmp_box[6 * (size_t) t + a]; hi[a] = b2.tmp_box[6 * (size_t) t + 3 + a]; }
                if (is_leaf(t)) code = out.leaf_bit | (uint32_t) ((b2.tmp_first[t] << 2) | (b2.tmp_count[t] - 1));
                else {
                    const int c = new_node();          // (grows out.nodes: no pointer into it is held across this call)
                    n_children.push_back(0);
                    kids[it.node4].push_back(c);
                    code = (uint32_t) c;
                    todo.push_back({t, c, it.depth + 1});
                }
            }
            codes[k] = code;
            for (int a = 0; a < 3; ++a) { los[k][a] = lo[a]; his[k][a] = hi[a]; }
        }
        float *q = &out.nodes[(size_t) kNodeFloats * (size_t) it.node4];
#ifdef PSDR_NODE128
        for (int k = 0; k < 4; ++k) {
            for (int a = 0; a < 3; ++a) { q[4 * a + k] = los[k][a]; q[12 + 4 * a + k] = his[k][a]; }
            std::memcpy(&q[24 + k], &codes[k], 4);
        }
#else
        {
            const int nc = (int) list.size();
            double org[3], ext[3];
            for (int a = 0; a < 3; ++a) {
                double mn = los[0][a], mx = his[0][a];
                for (int k = 1; k < nc; ++k) { mn = std::min(mn, (double) los[k][a]); mx = std::max(mx, (double) his[k][a]); }
                org[a] = mn; ext[a] = mx - mn;
            }
            uint32_t exps = 0, qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
            for (int a = 0; a < 3; ++a) {
                // smallest power of two with extent / step <= 255 (checked on the rounded-up upper bounds below)
                int e = ext[a] > 0.0 ? (int) std::ceil(std::log2(ext[a] / 255.0)) : 0;
                e = std::max(-126, std::min(127, e));
                for (;;) {
                    const double step = std::ldexp(1.0, e);
                    bool ok = true;
                    for (int k = 0; k < nc && ok; ++k) ok = std::ceil(((double) his[k][a] - org[a]) / step) <= 255.0;
                    if (ok || e >= 127) break;
                    ++e;
                }
                const double step = std::ldexp(1.0, e);
                exps |= (uint32_t) (e + 127) << (8 * a);
                for (int k = 0; k < 4; ++k) {
                    uint32_t l = 255u, h = 0u;          // unused child: inverted box (and code 0xffffffff, which the traversal checks)
                    if (k < nc) {
                        l = (uint32_t) std::max(0.0, std::min(255.0, std::floor(((double) los[k][a] - org[a]) / step)));
                        h = (uint32_t) std::max(0.0, std::min(255.0, std::ceil(((double) his[k][a] - org[a]) / step)));
                    }
                    qlo[a] |= l << (8 * k); qhi[a] |= h << (8 * k);
                }
            }
            const float o3[3] = {(float) org[0], (float) org[1], (float) org[2]};     // exact: org is one of the float bounds
            q[0] = o3[0]; q[1] = o3[1]; q[2] = o3[2]; std::memcpy(&q[3], &exps, 4);
            std::memcpy(&q[4], &qlo[0], 4); std::memcpy(&q[5], &qlo[1], 4); std::memcpy(&q[6], &qlo[2], 4); std::memcpy(&q[7], &qhi[0], 4);
            std::memcpy(&q[8], &qhi[1], 4); std::memcpy(&q[9], &qhi[2], 4); std::memcpy(&q[10], &codes[0], 4); std::memcpy(&q[11], &codes[1], 4);
            std::memcpy(&q[12], &codes[2], 4); std::memcpy(&q[13], &codes[3], 4);
        }
#endif
    }
    out.n_nodes = (int) (out.nodes.size() / kNodeFloats);
    out.max_depth = max_depth;
    // stack need: children are emitted after their parent, so a reverse sweep sees every child before its parent
    need.assign(out.n_nodes, 0);
    for (int i = out.n_nodes - 1; i >= 0; --i) {
        int below = 0;
        for (int c : kids[i]) below = std::max(below, need[c]);
        need[i] = (n_children[i] - 1) + below;
    }
    out.max_stack = need.empty() ? 1 : need[0] + 1;
}

} // namespace psdr
