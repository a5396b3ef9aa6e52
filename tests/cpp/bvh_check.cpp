// CPU check of the tree builder of the HIP library (psdr_jit_amd/csrc/hip/bvh.h is plain C++ on the host): compiled by tests/test_bvh_cpu.py with -DPSDR_BVH_WIDTH=4 and =8.
// Random triangle soup -> build_bvh + build_bvh4 ->
//   * every triangle slot hangs under exactly one leaf, every inner child id lies behind its parent's,
//   * the quantised box of every child, decoded the way trav4.h decodes it, contains the padded boxes of all triangles below it,
//   * the first kBvhTopNodes node ids are the top of the tree in breadth-first order (depth never decreases along them, nothing deeper is numbered before them),
//   * the tree does not depend on the number of builder threads.
// Prints "ok <nodes> <depth> <max_stack>" or "FAIL ...".
#include <cstdio>
#include <random>
#include "../../psdr_jit_amd/csrc/hip/bvh.h"

using namespace psdr;

static long long check(const std::vector<float> &p0, const std::vector<float> &e1, const std::vector<float> &e2, const BvhResult &b2, const Bvh4Result &b, std::vector<int> &depth_of) {
    const int n = (int) (p0.size() / 3), nn = b.n_nodes;
    long long bad = 0;
    std::vector<int> seen((size_t) n, 0);
    std::vector<float> box(6 * (size_t) nn);
    depth_of.assign((size_t) nn, 0);
    for (int i = 0; i < nn; ++i) {          // parents come first: depths top-down
        const uint32_t *u = reinterpret_cast<const uint32_t *>(&b.nodes[(size_t) kNodeFloats * i]);
        for (int k = 0; k < kBvhW; ++k) { const uint32_t c = u[kNodeCodeOff + k]; if (c != 0xffffffffu && !(c & b.leaf_bit)) { if ((int) c <= i || (int) c >= nn) ++bad; else depth_of[c] = depth_of[i] + 1; } }
    }
    for (int i = nn - 1; i >= 0; --i) {
        const float *q = &b.nodes[(size_t) kNodeFloats * i];
        const uint32_t *u = reinterpret_cast<const uint32_t *>(q);
        float ulo[3] = {3e38f, 3e38f, 3e38f}, uhi[3] = {-3e38f, -3e38f, -3e38f};
        for (int k = 0; k < kBvhW; ++k) {
            const uint32_t code = u[kNodeCodeOff + k];
            if (code == 0xffffffffu) continue;
            float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
            if (code & b.leaf_bit) {
                const int payload = (int) (code & (b.leaf_bit - 1u)), first = payload >> 2, cnt = (payload & 3) + 1;
                for (int t = first; t < first + cnt; ++t) {
                    if (t < 0 || t >= n) { ++bad; continue; }
                    seen[t]++;
                    const int o = b2.order[t];
                    float tl[3], th[3];
                    bvh_tri_box(&p0[3 * o], &e1[3 * o], &e2[3 * o], tl, th);
                    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], tl[a]); hi[a] = std::max(hi[a], th[a]); }
                }
            } else {
                if ((int) code <= i || (int) code >= nn) continue;
                for (int a = 0; a < 3; ++a) { lo[a] = box[6 * (size_t) code + a]; hi[a] = box[6 * (size_t) code + 3 + a]; }
            }
            for (int a = 0; a < 3; ++a) {
                constexpr int WW = kBvhW / 4;
                const double step = std::ldexp(1.0, (int) ((u[3] >> (8 * a)) & 0xffu) - 127);
                const uint32_t wl = u[4 + a * WW + (k >> 2)], wh = u[4 + (3 + a) * WW + (k >> 2)];
                const double ql = (double) q[a] + step * (double) ((wl >> (8 * (k & 3))) & 0xffu), qh = (double) q[a] + step * (double) ((wh >> (8 * (k & 3))) & 0xffu);
                if (!(ql <= (double) lo[a] && qh >= (double) hi[a])) ++bad;
                ulo[a] = std::min(ulo[a], lo[a]); uhi[a] = std::max(uhi[a], hi[a]);
            }
        }
        for (int a = 0; a < 3; ++a) { box[6 * (size_t) i + a] = ulo[a]; box[6 * (size_t) i + 3 + a] = uhi[a]; }
    }
    for (int t = 0; t < n; ++t) if (seen[t] != 1) ++bad;
    return bad;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20000;
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> p0(3 * (size_t) n), e1(3 * (size_t) n), e2(3 * (size_t) n);
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) { p0[3 * i + a] = 100.f * U(rng); e1[3 * i + a] = 2.f * U(rng); e2[3 * i + a] = 2.f * U(rng); }
    BvhResult r1, r4;
    build_bvh(p0.data(), e1.data(), e2.data(), n, r1, 1);
    build_bvh(p0.data(), e1.data(), e2.data(), n, r4, 4);
    Bvh4Result t1, t4;
    build_bvh4(r1, n, t1);
    build_bvh4(r4, n, t4);
    if (r1.order != r4.order || t1.nodes.size() != t4.nodes.size() || std::memcmp(t1.nodes.data(), t4.nodes.data(), t1.nodes.size() * sizeof(float)) != 0) { printf("FAIL the tree depends on the thread count\n"); return 1; }
    std::vector<int> depth;
    const long long bad = check(p0, e1, e2, r1, t1, depth);
    if (bad) { printf("FAIL %lld violations\n", bad); return 1; }
    // breadth-first head: depths never decrease along the first kBvhTopNodes ids, and no later node is shallower than the last of them minus one level
    const int top = std::min(kBvhTopNodes, t1.n_nodes);
    for (int i = 1; i < top; ++i) if (depth[i] < depth[i - 1]) { printf("FAIL node %d (depth %d) after depth %d\n", i, depth[i], depth[i - 1]); return 1; }
    for (int i = top; i < t1.n_nodes; ++i) if (depth[i] < depth[top - 1]) { printf("FAIL node %d of depth %d is numbered behind the breadth-first head (last depth %d)\n", i, depth[i], depth[top - 1]); return 1; }
    printf("ok %d %d %d\n", t1.n_nodes, t1.max_depth, t1.max_stack);
    return 0;
}
