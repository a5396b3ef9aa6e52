// Filter primitives of the brute-force tracer (scenes of <= 64 triangles), host side.
//
// Phase 1 of trace2 / trace_scene tests every ray against a list of FILTER PRIMITIVES and only collects
// candidates; the exact Moeller-Trumbore test (= the reference's ray_intersect_triangle, utils.h:82-93) then runs on
// the candidates alone.  A filter primitive is a window  0 <= u <= umax, 0 <= v <= vmax, u + v <= smax  in the
// coordinates of a planar basis (p0; e1, e2):
//   * a single triangle: its own (p0, e1, e2), [0,1] x [0,1], u + v <= 1   -> candidate = that triangle
//   * two coplanar triangles A, B that share an edge S0-S1 (the two halves of a quad, which is what
//     Cornell-box-like scenes are made of): corner S0, e1 / e2 = the edges to the two unshared vertices, the window
//     that covers S1 = (a, b); the side of the diagonal, sign(b u - a v), tells which of the two triangles the
//     point belongs to (both inside a small band around the diagonal)      -> candidate = A or B (or both)
// so a quad costs one filter test instead of two.  The window only has to be conservative.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace psdr {

struct FilterPrim {
    float p0[3], e1[3], e2[3];
    float umax, vmax, smax;
    float da, db;              // diagonal: sel = db * u - da * v  (>= 0: triangle A's side)
    int32_t slot_a, slot_b;    // device triangle slots; slot_b = -1 for a single triangle
    float k16;                 // 2^-15 * (largest extent of the quad); 0 for a single triangle (api.hip takes max(|e1|, |e2|) then)
};
// Device form (api.hip uploads it, scene_dev.h::trace2 / trace_scene evaluate it).  With c = the centre of the scene's
// bounding sphere, oc = o - c, m = oc x d (per ray) and p = p0 - c, N = e1 x e2 (per primitive) the Moeller-Trumbore
// numerators are plain dot products of ray constants with primitive constants:
//     s.(d x e2) = m.e2 + d.(p x e2)        d.(s x e1) = d.(e1 x p) - m.e1
//     e1.(d x e2) = -d.N                    e2.(s x e1) = oc.N - p.N
// six words per primitive: {e2, A.x} {A.yz, e1.xy} {e1.z, B} {N, -p.N} {umax, vmax, smax, da} {db, da + db, k15, slots},
// A = p x e2 and B = e1 x p rounded from double.
//
// Slack.  These sums cancel: the terms are of size |oc| K and |p| K (K = the primitive's largest extent) while the result
// is of size |s| K, so the ABSOLUTE rounding error of a numerator is <= ~11 * 2^-24 (|oc| + |p|) K (six products and
// sums, the rounding of m and of the stored constants), whatever |det| is; the exact test's own numerators carry
// <= ~4 * 2^-24 |s| K.  Quads add the error of working in another basis of the same plane, (6 * 2 + 1) * 8 * 2^-24 |s| K
// (window functions = convex combinations, weights <= 2, of the triangle's three barycentric numerators scaled by the
// area ratio <= ~2).  Every window test therefore gets the slack  k15 (S + Kmax),  k15 = 2^-15 K,  S = |oc| + R >= |oc| +
// |p| (R = the bounding sphere's radius), Kmax = the largest K of the scene - 2^-15 = 512 * 2^-24 covers all of the above
// four times over - plus 2^-18 |det| for the rounding of 1/det and of the products with it; the t numerator (terms of
// size (|oc| + |p|) K^2) gets Kmax times that.  The window only has to be conservative: a wider one costs exact tests,
// never hits (tests: test_trace_matches_oracle, 300 k rays incl. edge-to-edge and vertex-to-vertex ones, bit-exact).
// p0/e1/e2: [n,3] rows in ORIGINAL triangle order; order[slot] = original index of the triangle in device slot `slot`
inline void build_filter_prims(const float *p0, const float *e1, const float *e2, const int *order, int n, std::vector<FilterPrim> &out) {
    out.clear();
    if (n > 64) return;
    struct D3 { double x, y, z; };
    auto ld = [](const float *p) { return D3{(double) p[0], (double) p[1], (double) p[2]}; };
    auto add = [](D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; };
    auto sub = [](D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; };
    auto dot = [](D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
    auto cross = [](D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
    auto st = [](float *q, D3 v) { q[0] = (float) v.x; q[1] = (float) v.y; q[2] = (float) v.z; };
    const double pad = 1e-4;
    std::vector<char> used(n, 0);
    for (int i = 0; i < n; ++i) {
        if (used[i]) continue;
        used[i] = 1;
        const int oi = order[i];
        const D3 P = ld(p0 + 3 * oi), E1 = ld(e1 + 3 * oi), E2 = ld(e2 + 3 * oi);
        const D3 va[3] = {P, add(P, E1), add(P, E2)};
        const D3 N = cross(E1, E2);
        const double nn = std::sqrt(dot(N, N)), size = std::sqrt(std::max(dot(E1, E1), dot(E2, E2)));
        FilterPrim f{};
        std::memcpy(f.p0, p0 + 3 * oi, 12); std::memcpy(f.e1, e1 + 3 * oi, 12); std::memcpy(f.e2, e2 + 3 * oi, 12);
        f.umax = 1.f; f.vmax = 1.f; f.smax = 1.f; f.da = 0.f; f.db = 0.f; f.slot_a = i; f.slot_b = -1; f.k16 = 0.f;
        for (int j = i + 1; j < n && nn > 0.0; ++j) {
            if (used[j]) continue;
            const int oj = order[j];
            const D3 Q = ld(p0 + 3 * oj);
            const D3 vb[3] = {Q, add(Q, ld(e1 + 3 * oj)), add(Q, ld(e2 + 3 * oj))};
            // match B's vertices to A's
            int match_b[3] = {-1, -1, -1}, shared = 0;
            for (int b = 0; b < 3; ++b)
                for (int a = 0; a < 3; ++a) { const D3 d = sub(vb[b], va[a]); if (match_b[b] < 0 && std::sqrt(dot(d, d)) <= 1e-6 * size) { match_b[b] = a; ++shared; } }
            if (shared != 2) continue;
            int lone_b = -1, lone_a = 3;
            for (int b = 0; b < 3; ++b) { if (match_b[b] < 0) lone_b = b; else lone_a -= match_b[b]; }
            if (lone_b < 0 || lone_a < 0 || lone_a > 2) continue;
            const int s0 = (lone_a + 1) % 3, s1 = (lone_a + 2) % 3;
            const D3 S0 = va[s0], S1 = va[s1], LA = va[lone_a], LB = vb[lone_b];
            if (std::fabs(dot(N, sub(LB, P))) > 1e-9 * nn * size) continue;        // coplanar up to double rounding only
            const D3 F1 = sub(LA, S0), F2 = sub(LB, S0), r = sub(S1, S0);
            const double g11 = dot(F1, F1), g12 = dot(F1, F2), g22 = dot(F2, F2), r1 = dot(F1, r), r2 = dot(F2, r);
            const double det = g11 * g22 - g12 * g12;
            if (!(det > 1e-12 * g11 * g22)) continue;
            const double a = (r1 * g22 - r2 * g12) / det, b = (r2 * g11 - r1 * g12) / det;
            if (!(a > 0.05 && b > 0.05 && a < 16.0 && b < 16.0)) continue;          // S0, LA, S1, LB must be a convex quad
            st(f.p0, S0); st(f.e1, F1); st(f.e2, F2);
            f.umax = (float) (std::max(1.0, a) + pad); f.vmax = (float) (std::max(1.0, b) + pad);
            f.smax = (float) (std::max(1.0, a + b) + 2.0 * pad);
            f.da = (float) a; f.db = (float) b;
            f.slot_b = j;
            {
                const D3 qv[4] = {S0, LA, S1, LB};
                double ext = 0.0;
                for (int x = 0; x < 4; ++x) for (int y = x + 1; y < 4; ++y) { const D3 d = sub(qv[x], qv[y]); ext = std::max(ext, std::sqrt(dot(d, d))); }
                f.k16 = (float) (ext * (1.0 / 32768.0));
            }
            used[j] = 1;
            break;
        }
        out.push_back(f);
    }
}

} // namespace psdr
