// scene_dev.h — device view of the configured scene and the closest-hit traversal.
//
// All read-only scene tables live in ONE contiguous "blob" of float4 words
//   [ BVH nodes | traversal triangles | shading triangles | triangle tangents | orig->slot map |
//     mesh / bsdf / emitter records | emitter CDF | emitter-mesh face CDFs ]
// which psdr_hip_scene_create uploads once.  When the blob is small (Cornell box: ~11 KB) every
// workgroup copies it into LDS with coalesced 16-byte loads at kernel start and all later scene
// reads are ds_read_b128 (wave-uniform addresses broadcast); large scenes read it through L2.
// The kernels are templated on the memory space so that both forms compile to their own ISA.
#pragma once
#include "dmath.h"

namespace psdr {

constexpr float kEpsilon = 1e-5f, kRayEpsilon = 1e-3f, kShadowEpsilon = 1e-3f, kEdgeEpsilon = 1e-5f;   // reference constants.h:12-17
constexpr float kPi = 3.14159265358979323846f, kInvPi = 0.31830988618379067154f;
constexpr float kTraceTMax = 100000000.f;   // reference scene_optix.cpp:376
constexpr int kBlock = 256;
constexpr int kBruteForceMax = 64;          // scenes with at most this many triangles skip the BVH
constexpr int kParkWords = 12;              // LDS words per lane behind the traversal stack: the lane's two parked rays (trav4.h)
constexpr int kTravRows = 26;               // LDS rows (of kBlock words) behind the stack of a BVH scene: parked rays, best hits, pair ring, ring heads (trav4.h)

// EnvironmentMap after configure() (psdr_envmap_rec): too large for the LDS blob, read from global memory
struct EnvDev {
    const float *radiance;          // [height*width*3]
    const float *cell_pmf, *cell_cmf;
    int width, height, reso0, reso1, num_cells;
    float scale, cell_sum;
    const float *d_radiance;        // forward tangent of the texels, or NULL
    float d_scale;
    Mat4<float> to_world, from_world, d_from_world;
    float lower[3], upper[3];
};

// one bitmap parameter of a BSDF (global memory; w == 0: constant).  SceneTables::tex holds three per BSDF:
// [0] reflectance / diffuse reflectance (rgb), [1] specular reflectance (rgb), [2] roughness (one channel)
struct TexDev { const float *data, *d_data; int w, h; long long g_off; };    // g_off: offset of its texel adjoints in psdr_grads.g_tex

// MicrofacetPerVertex (microfacet_pv.cpp): per-vertex parameter arrays of one BSDF in global memory (n == 0: not per-vertex)
struct PvDev { const float *spec, *d_spec, *diff, *d_diff, *rough, *d_rough; int n; long long g_off[3]; };   // g_off: diffuse / specular / roughness adjoints in psdr_grads.g_tex

// Microfacet parameters beyond the diffuse reflectance (global memory table, one entry per BSDF; microfacet.h)
struct MatDev {
    float specular[3], d_specular[3], roughness, d_roughness;                  // Microfacet; RoughConductor: specular = specular_reflectance
    float alpha_u, alpha_v, d_alpha_u, d_alpha_v, eta[3], d_eta[3], k[3], d_k[3];   // RoughConductor
};

struct SceneTables {
    // float4-word offsets into the blob
    int nodes_off, trav_off, shade_off, tan_off, map_off, mesh_off, bsdf_off, emit_off, ecdf_off, fcdf_off;
    int n_nodes, n_tris, n_meshes, n_bsdfs, n_emitters, n_fcdf;
    int has_tangent, stack_depth;   // stack_depth: LDS rows (of kBlock words) reserved per workgroup behind the blob = traversal stack + parked rays
    int stack_lds, ref_bits;        // 4-wide BVH (trav4.h): stack entries kept in LDS, bits of a child code
    int *gstack;                    // deeper stack entries, [entry][lane of the grid] (NULL when stack_lds covers the tree)
    int gstack_stride;
    int filt_off, n_filt;      // filter primitives of the brute-force tracer (4 words each, see filter.h)
    float center[3], radius;   // bounding sphere of all vertices
    int env_emitter;           // index of the EnvironmentMap among the emitters, -1 = none
    const TexDev *tex;         // [n_bsdfs] or NULL when no BSDF is textured
    const MatDev *mat;         // [n_bsdfs] or NULL when every BSDF is Diffuse
    const PvDev *pv;           // [n_bsdfs] or NULL when no BSDF is per-vertex
    const int *tri_fi;         // [n_tris*3] mesh-local vertex ids per triangle SLOT (with pv only)
    EnvDev env;
    float emitter_sum;
    int blob_words;            // float4 count
    int width, height, spp, sppe, sppse;
};

// Secondary-edge table inside the blob: 6 words per edge
//   {p0.xyz, e1.x} {e1.yz, n0.xy} {n0.z, n1.xyz} {p2.xyz, bits(is_boundary)} {d_p0.xyz, d_e1.x} {d_e1.yz, 0, 0}
// followed (at cdf_off) by pmf[n], cmf[n].  (Stage r01a kept these in global arrays: the CDF binary search was a
// chain of 7 dependent global loads per candidate and dominated the secondary-edge kernel.)
struct SecEdgeTables {
    int off, cdf_off, n;
    float sum;
};

// Primary-edge table of one sensor inside the blob: 3 words per edge {p0.xy, p1.xy} {d_p0.xy, d_p1.xy} {n.xy, length, 0},
// then pmf[n], cmf[n] at pecdf_off.
struct SensorDev {
    Mat4<float> sample_to_camera, to_world, d_to_world, world_to_sample, d_world_to_sample;
    float cam_pos[3], cam_dir[3];
    float inv_area;
    int n_edges, pe_off, pecdf_off;
    float edge_sum;
    int ortho;                 // OrthographicCamera
};

struct Counters { unsigned long long rays, nodes, tris, hits; };

// per-lane view used by every device function
// Scene classes (the `LDS` template parameter of every kernel and device function):
//   0  global memory, every feature (GGX BSDFs, bitmap / per-vertex parameters, first-hit integrators, environment map)
//   1  small scene staged in LDS: Diffuse BSDFs and area lights only (all Cornell boxes)
//   2  global memory, lean: Diffuse BSDFs, area lights and the environment map - the BVH scenes of the tutorials and of
//      BASELINE config 5 do not pay registers for material code they do not use
constexpr bool in_lds(int cls) { return cls == 1; }
constexpr bool has_env(int cls) { return cls != 1; }
constexpr bool has_mat(int cls) { return cls == 0; }

constexpr int kEnvLookup = -1;         // id of an environment-map lookup in the lookup record (BSDF ids are >= 0)
// a per-vertex BSDF interpolation at triangle slot s is recorded as id = kPvLookup - s, with the barycentrics as (u, v)
constexpr int kPvLookup = -2;
// (the lookup record of a path holds 4 * max_depth + 4 entries, adjoint.h)

template <int LDS> struct SceneView {
    const float4 *B;           // blob base (LDS or global)
    const float4 *G;           // blob base in global memory (wave-uniform reads become scalar loads)
    const SceneTables *T;      // kernel-argument copy
    int *stack;                // this lane's LDS traversal stack, stride kBlock
    unsigned int c_nodes, c_tris, c_rays, c_hits;   // instrumented build only

    // ---- reverse mode (k_adjoint only; constant 0 everywhere else, so the compiler folds it away)
    // The adjoint of a path is obtained by re-running its D-mode shading with ONE-HOT tangents ("probes")
    // on the scene quantities the path touched.  The hits found by the first (recording) run are replayed
    // instead of traversing again, so a probe costs shading only.
    int mis;                   // Li variant: -1 PathTracer, 0/1/2 DirectIntegrator(mis) (set by the kernels from their parameters)
    int field, field_object;   // >= 0: first-hit integrator (FieldExtractionIntegrator / CollocatedIntegrator), shade.h first_hit_value
    float intensity, d_intensity;
    int mode;                  // 0 = trace, 1 = trace + record hits, 2 = replay recorded hits
    float *rec;                // this lane's LDS record, stride kBlock: 4 words per hit (slot, u, v, t)
    int rec_i, rec_n;          // replay cursor / number of recorded hits
    int *ext;                  // extra triangle slots read without a trace (light samples), stride kBlock
    int ext_n;
    int probe_kind;            // 0 none, 1 triangle field, 2 bsdf reflectance, 3 emitter radiance, 4 camera to_world, 5 bitmap lookup, 6 material constant, 7 environment map from_world, 8 blended normal of one hit
    int probe_id, probe_comp;
    // bitmap-parameter lookups (textured BSDFs): the recording run notes (bsdf id, u, v) of every lookup; a probe of kind 5
    // puts a unit tangent on one component of the looked-up value of ONE such lookup (matched by id and the bit-equal uv the
    // replay reproduces); the kernel scatters the result over the four texels of the lookup's footprint
    float *lk;                 // this lane's LDS lookup record, stride kBlock: 3 words per entry (id, u, v)
    mutable int lk_n;
    int lk_max, ext_max;       // capacities of the lookup / light-sample records (sized from max_depth by the launch)
    float probe_u, probe_v;
    PSDR_DEV void note_lookup(int id, float u, float v) const {
        if (mode != 1) return;
        for (int i = 0; i < lk_n; ++i)
            if (__float_as_int(lk[3 * i * kBlock]) == id && lk[(3 * i + 1) * kBlock] == u && lk[(3 * i + 2) * kBlock] == v) return;
        if (lk_n >= lk_max) return;
        lk[3 * lk_n * kBlock] = __int_as_float(id); lk[(3 * lk_n + 1) * kBlock] = u; lk[(3 * lk_n + 2) * kBlock] = v;
        ++lk_n;
    }
    // component c0..c0+n-1 of the probe belongs to this lookup: which one gets the unit tangent (-1 = none)
    PSDR_DEV int lookup_hot(int id, float u, float v, int c0, int n) const {
        if (probe_kind != 5 || probe_id != id || u != probe_u || v != probe_v) return -1;
        const int c = probe_comp - c0;
        return (c >= 0 && c < n) ? c : -1;
    }

    PSDR_DEV bool tan_on() const { return probe_kind != 0 || T->has_tangent != 0; }
    // word k (0..5) of the 22-float tangent row [p0 e1 e2 n0 n1 n2 fn area] of triangle slot `slot`
    PSDR_DEV float4 tanw(int slot, int k) const {
        if (probe_kind == 0) return B[T->tan_off + 6 * slot + k];
        const bool on = probe_kind == 1 && slot == probe_id && (probe_comp >> 2) == k;
        const int l = probe_comp & 3;
        return make_float4(on && l == 0 ? 1.f : 0.f, on && l == 1 ? 1.f : 0.f, on && l == 2 ? 1.f : 0.f, on && l == 3 ? 1.f : 0.f);
    }
    PSDR_DEV float4 rgb_tan(int word, int kind, int id) const {      // tangent of a bsdf / emitter colour record
        if (probe_kind == 0) return B[word];
        const bool on = probe_kind == kind && id == probe_id;
        return make_float4(on && probe_comp == 0 ? 1.f : 0.f, on && probe_comp == 1 ? 1.f : 0.f, on && probe_comp == 2 ? 1.f : 0.f, 0.f);
    }
    // tangent of constant material parameter k of BSDF id (psdr_grads.g_mat row layout): the forward tangent in a render,
    // a unit tangent when a probe of kind 6 asks for exactly this one, zero in every other replay
    PSDR_DEV float mat_tan(int id, int k, float fwd) const {
        if (mode == 0) return fwd;
        return (probe_kind == 6 && probe_id == id && probe_comp == k) ? 1.f : 0.f;
    }
    PSDR_DEV void note_slot(int slot) { if (mode == 1 && slot >= 0 && ext_n < ext_max) { ext[ext_n * kBlock] = slot; ++ext_n; } }

    PSDR_DEV float4 ld(int word) const { return B[word]; }
    PSDR_DEV float ldf(int word_off, int idx) const { return reinterpret_cast<const float *>(B + word_off)[idx]; }
    PSDR_DEV int ldi(int word_off, int idx) const { return reinterpret_cast<const int *>(B + word_off)[idx]; }
};

typedef float f2 __attribute__((ext_vector_type(2)));
PSDR_DEV f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

struct Hit { int slot; float u, v, t; };   // slot = device triangle slot (BVH leaf order), -1 = miss

// Möller–Trumbore exactly as the reference's own ray_intersect_triangle (include/psdr/utils.h:82-93)
PSDR_DEV bool tri_test(const float4 &a, const float4 &b, const float4 &c, const Vec3f &o, const Vec3f &d, float &u, float &v, float &t) {
#pragma clang fp contract(off)
    Vec3f p0(a.x, a.y, a.z), e1(a.w, b.x, b.y), e2(b.z, b.w, c.x);
    Vec3f h = cross(d, e2);
    float det = dot(e1, h);
    float f = 1.f / det;
    Vec3f s = o - p0;
    u = f * dot(s, h);
    Vec3f q = cross(s, e1);
    v = f * dot(d, q);
    t = f * dot(e2, q);
    return (u >= 0.f) && (v >= 0.f) && (u + v <= 1.f) && (t > kRayEpsilon) && (t < kTraceTMax);
}

// Closest hit in (RayEpsilon, 1e8), ties -> smallest original triangle id
// (replaces jit_optix_ray_trace, reference scene_optix.cpp:343-410; NaN rays miss, :348-353).
template <int LDS, bool COUNT> PSDR_DEV Hit trace_scene(SceneView<LDS> &S, const Vec3f &o, const Vec3f &d);

template <int LDS, bool COUNT>
PSDR_DEV Hit trace(SceneView<LDS> &S, const Vec3f &o, const Vec3f &d) {
    if (S.mode == 2) {                       // replay: pop the hit the recording run found for this ray
        Hit h; h.slot = -1; h.u = h.v = h.t = 0.f;
        if (S.rec_i < S.rec_n) {
            const float *r = S.rec + 4 * S.rec_i * kBlock;
            h.slot = __float_as_int(r[0]); h.u = r[kBlock]; h.v = r[2 * kBlock]; h.t = r[3 * kBlock];
        }
        ++S.rec_i;
        return h;
    }
    const Hit h = trace_scene<LDS, COUNT>(S, o, d);
    if (S.mode == 1) {
        float *r = S.rec + 4 * S.rec_n * kBlock;
        r[0] = __int_as_float(h.slot); r[kBlock] = h.u; r[2 * kBlock] = h.v; r[3 * kBlock] = h.t;
        ++S.rec_n;
    }
    return h;
}

// 4-wide BVH traversal of up to two rays per lane (scenes with more than kBruteForceMax triangles): trav4.h
template <int LDS, bool COUNT>
PSDR_DEV void bvh4_trace2(SceneView<LDS> &S, const Vec3f &oA, const Vec3f &dA, bool actA, const Vec3f &oB, const Vec3f &dB, bool actB, Hit &hA, Hit &hB);

template <int LDS, bool COUNT>
PSDR_DEV Hit trace_scene(SceneView<LDS> &S, const Vec3f &o, const Vec3f &d) {
#pragma clang fp contract(off)
    Hit best; best.slot = -1; best.u = best.v = 0.f; best.t = 0.f;
    if (!(o.x == o.x && o.y == o.y && o.z == o.z && d.x == d.x && d.y == d.y && d.z == d.z)) return best;
    const SceneTables &T = *S.T;
    float best_t = __builtin_inff();
    int best_id = 0x7fffffff;
    if (T.n_tris <= kBruteForceMax) {
        // Tiny scenes (README Cornell box: 36 triangles): no tree.  Every ray is tested against every filter
        // primitive in a wave-uniform loop (primitive words arrive by scalar loads in SGPRs, no stack, no LDS traffic,
        // no divergence - for incoherent rays this beats any per-lane tree walk on a 64-wide SIMD machine), and the
        // exact tri_test then runs on the few surviving triangles of each lane.  See trace2 below for the two phases;
        // this is its one-ray form.  Same tri_test, same (t, id) order => same hit as the BVH path.
        if (COUNT) { S.c_rays++; S.c_tris += (unsigned) T.n_tris; }
        unsigned m0 = 0u, m1 = 0u;
        const float s_ray = norm(o - Vec3f(T.center[0], T.center[1], T.center[2])) + T.radius;
        {
            const float4 *prim = S.G + T.filt_off;
            float4 a = prim[0], b = prim[1], c = prim[2], g = prim[3];
            for (int k = 0; k < T.n_filt; ++k) {
                const int kn = (k + 1 < T.n_filt) ? k + 1 : k;
                const float4 na = prim[4 * kn], nb = prim[4 * kn + 1], nc = prim[4 * kn + 2], ng = prim[4 * kn + 3];
                const Vec3f p0(a.x, a.y, a.z), e1(a.w, b.x, b.y), e2(b.z, b.w, c.x);
                const Vec3f h = cross(d, e2);
                const float det = dot(e1, h);
                const Vec3f s = o - p0;
                const float un = dot(s, h);
                const Vec3f q = cross(s, e1);
                const float vn = dot(d, q), tn = dot(e2, q);
                const unsigned db = __float_as_uint(det), sg = db & 0x80000000u;
                const float ad = __uint_as_float(db & 0x7fffffffu);
                const float ua = __uint_as_float(__float_as_uint(un) ^ sg), va = __uint_as_float(__float_as_uint(vn) ^ sg);
                const float ta = __uint_as_float(__float_as_uint(tn) ^ sg);
                const float hi_u = fmaf(c.y, ad, -ua), hi_v = fmaf(c.z, ad, -va), hi_s = fmaf(c.w, ad, -(ua + va));
                const float m = __builtin_fminf(__builtin_fminf(__builtin_fminf(ua, va), __builtin_fminf(hi_u, hi_v)), hi_s);
                const float kq = g.w * 32768.f, slack = g.w * (s_ray + kq);          // absolute slack of a quad (filter.h)
                const bool pass = __builtin_fminf(fmaf(ad, 3.8146973e-06f, m) + slack, fmaf(slack, kq, ta)) >= 0.f;
                const float sel = fmaf(g.y, ua, -(g.x * va)), band = fmaf(ad, 1e-3f, slack) * (g.x + g.y);
                const int sab = __float_as_int(g.z), sa = sab & 0xff, sb = (sab >> 8) & 0xff;
                const unsigned a_lo = sa < 32 ? 1u << sa : 0u, a_hi = sa >= 32 ? 1u << (sa - 32) : 0u;
                const unsigned b_lo = sb < 32 ? 1u << sb : 0u, b_hi = (sb >= 32 && sb < 64) ? 1u << (sb - 32) : 0u;
                const bool in_a = pass & (sel >= -band), in_b = pass & (sel <= band);
                m0 |= (in_a ? a_lo : 0u) | (in_b ? b_lo : 0u);
                m1 |= (in_a ? a_hi : 0u) | (in_b ? b_hi : 0u);
                a = na; b = nb; c = nc; g = ng;
            }
        }
        for (;;) {
            const bool more = (m0 | m1) != 0u;
            if (__ballot(more) == 0ull) break;
            int k = 0;
            if (m0 != 0u) { k = __builtin_ctz(m0); m0 &= m0 - 1u; } else if (m1 != 0u) { k = 32 + __builtin_ctz(m1); m1 &= m1 - 1u; }
            const float4 a = S.ld(T.trav_off + 3 * k), b = S.ld(T.trav_off + 3 * k + 1), c = S.ld(T.trav_off + 3 * k + 2);
            float u, v, t;
            const bool ok = more & tri_test(a, b, c, o, d, u, v, t);
            const int id = __float_as_int(c.y);
            if (ok & ((t < best_t) | ((t == best_t) & (id < best_id)))) { best_t = t; best_id = id; best.slot = k; best.u = u; best.v = v; best.t = t; }
        }
        return best;
    }
    if constexpr (!in_lds(LDS)) {          // (the LDS class holds brute-force scenes only: its kernels carry no tree code)
        Hit other;
        bvh4_trace2<LDS, COUNT>(S, o, d, true, o, d, false, best, other);
    }
    return best;
}

// Two rays per lane through ONE pass over the triangles (the NEE shadow ray and the BSDF extension ray of the same
// path vertex).  The scalar triangle loads, the loop control and the exec bookkeeping are shared and the two
// independent dependency chains fill each other's latency slots; every ray still sees exactly tri_test's
// arithmetic and the (t, id) order, so the hits equal two trace() calls.  Inactive rays are given a NaN origin,
// which fails every comparison.  BVH scenes and the record/replay modes trace the rays one after the other.
template <int LDS, bool COUNT>
PSDR_DEV void trace2(SceneView<LDS> &S, const Vec3f &oA_, const Vec3f &dA, bool actA, const Vec3f &oB_, const Vec3f &dB, bool actB,
                     Hit &hA, Hit &hB) {
#pragma clang fp contract(off)
    hA.slot = -1; hA.u = hA.v = hA.t = 0.f;
    hB.slot = -1; hB.u = hB.v = hB.t = 0.f;
    const SceneTables &T = *S.T;
    if (S.mode != 0) {
        if (actA) hA = trace<LDS, COUNT>(S, oA_, dA);
        if (actB) hB = trace<LDS, COUNT>(S, oB_, dB);
        return;
    }
    if constexpr (!in_lds(LDS)) { if (T.n_tris > kBruteForceMax) { bvh4_trace2<LDS, COUNT>(S, oA_, dA, actA, oB_, dB, actB, hA, hB); return; } }
    const float qnan = __builtin_nanf("");
    const Vec3f oA = actA ? oA_ : Vec3f(qnan), oB = actB ? oB_ : Vec3f(qnan);
    if (COUNT) { const unsigned n = (actA ? 1u : 0u) + (actB ? 1u : 0u); S.c_rays += n; S.c_tris += n * (unsigned) T.n_tris; }
    // Both rays ride in the two halves of packed-f32 registers (element-wise identical to the scalar instructions).
    //
    // Two phases.  The exact test spends more than half of its issue cycles on the IEEE division and on eight
    // compares (measured on gfx950: v_mul/v_add/v_fmac 2 cycles per wave64, v_fma 3, v_cmp 4.3, v_rcp 8.2, the
    // division sequence 44).  Phase 1 therefore only FILTERS: the Moeller-Trumbore numerators and the determinant,
    // then one sign-folded min3 test that is conservative (it accepts everything tri_test accepts, with a 2^-18
    // relative margin that covers the rounding of 1/det and of the three products) and a per-lane bit mask of the
    // survivors.  Phase 2 runs the exact tri_test - same arithmetic, same (t, id) order as ever - on the 1-5
    // surviving triangles of each lane.  The hits are bit-equal to the single-phase loop.
    const f2 ox = {oA.x, oB.x}, oy = {oA.y, oB.y}, oz = {oA.z, oB.z};
    const f2 dx = {dA.x, dB.x}, dy = {dA.y, dB.y}, dz = {dA.z, dB.z};
    unsigned mA0 = 0u, mA1 = 0u, mB0 = 0u, mB1 = 0u;
    const Vec3f ctr(T.center[0], T.center[1], T.center[2]);
    const f2 s_ray = {norm(oA - ctr) + T.radius, norm(oB - ctr) + T.radius};
    {
        // filter primitives (filter.h): {p0.xyz, e1.x} {e1.yz, e2.xy} {e2.z, umax, vmax, smax} {da, db, slot_a, slot_b}
        const float4 *prim = S.G + T.filt_off;
        float4 a = prim[0], b = prim[1], c = prim[2], g = prim[3];
        for (int k = 0; k < T.n_filt; ++k) {
            const int kn = (k + 1 < T.n_filt) ? k + 1 : k;
            const float4 na = prim[4 * kn], nb = prim[4 * kn + 1], nc = prim[4 * kn + 2], ng = prim[4 * kn + 3];
            const f2 e1x = a.w, e1y = b.x, e1z = b.y, e2x = b.z, e2y = b.w, e2z = c.x;
            const f2 hx = pk_fma(dy, e2z, -(dz * e2y)), hy = pk_fma(dz, e2x, -(dx * e2z)), hz = pk_fma(dx, e2y, -(dy * e2x));
            const f2 det = pk_fma(e1z, hz, pk_fma(e1y, hy, e1x * hx));
            const f2 sx = ox - (f2) a.x, sy = oy - (f2) a.y, sz = oz - (f2) a.z;
            const f2 un = pk_fma(sz, hz, pk_fma(sy, hy, sx * hx));
            const f2 qx = pk_fma(sy, e1z, -(sz * e1y)), qy = pk_fma(sz, e1x, -(sx * e1z)), qz = pk_fma(sx, e1y, -(sy * e1x));
            const f2 vn = pk_fma(dz, qz, pk_fma(dy, qy, dx * qx));
            const f2 tn = pk_fma(e2z, qz, pk_fma(e2y, qy, e2x * qx));
            // fold the sign of det into the numerators: u = ua/ad, v = va/ad, t = ta/ad with ad = |det|
            f2 ua, va, ta, ad;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned db = __float_as_uint(det[r]), sg = db & 0x80000000u;
                ad[r] = __uint_as_float(db & 0x7fffffffu);
                ua[r] = __uint_as_float(__float_as_uint(un[r]) ^ sg);
                va[r] = __uint_as_float(__float_as_uint(vn[r]) ^ sg);
                ta[r] = __uint_as_float(__float_as_uint(tn[r]) ^ sg);
            }
            const f2 hi_u = pk_fma((f2) c.y, ad, -ua), hi_v = pk_fma((f2) c.z, ad, -va), hi_s = pk_fma((f2) c.w, ad, -(ua + va));
            const float kq = g.w * 32768.f;
            const f2 slack = (f2) g.w * (s_ray + (f2) kq);                    // absolute slack of a quad (filter.h)
            const f2 margin = pk_fma(ad, (f2) 3.8146973e-06f, slack);         // 2^-18 |det| + slack
            const f2 tlo = pk_fma(slack, (f2) kq, ta);
            const f2 sel = pk_fma((f2) g.y, ua, -((f2) g.x * va));            // side of the quad's diagonal
            const f2 band = pk_fma(ad, (f2) 1e-3f, slack) * (f2) (g.x + g.y);
            const int sab = __float_as_int(g.z), sa = sab & 0xff, sb = (sab >> 8) & 0xff;
            const unsigned a_lo = sa < 32 ? 1u << sa : 0u, a_hi = sa >= 32 ? 1u << (sa - 32) : 0u;
            const unsigned b_lo = sb < 32 ? 1u << sb : 0u, b_hi = (sb >= 32 && sb < 64) ? 1u << (sb - 32) : 0u;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float m = __builtin_fminf(__builtin_fminf(__builtin_fminf(ua[r], va[r]), __builtin_fminf(hi_u[r], hi_v[r])), hi_s[r]);
                const bool pass = __builtin_fminf(m + margin[r], tlo[r]) >= 0.f;
                const bool in_a = pass & (sel[r] >= -band[r]), in_b = pass & (sel[r] <= band[r]);
                const unsigned lo = (in_a ? a_lo : 0u) | (in_b ? b_lo : 0u), hi = (in_a ? a_hi : 0u) | (in_b ? b_hi : 0u);
                if (r == 0) { mA0 |= lo; mA1 |= hi; } else { mB0 |= lo; mB1 |= hi; }
            }
            a = na; b = nb; c = nc; g = ng;
        }
    }
    float btA = __builtin_inff(), btB = __builtin_inff();
    int bidA = 0x7fffffff, bidB = 0x7fffffff;
    for (;;) {
        const bool moreA = (mA0 | mA1) != 0u, moreB = (mB0 | mB1) != 0u;
        if (__ballot(moreA || moreB) == 0ull) break;
        int kA = 0, kB = 0;
        if (mA0 != 0u) { kA = __builtin_ctz(mA0); mA0 &= mA0 - 1u; } else if (mA1 != 0u) { kA = 32 + __builtin_ctz(mA1); mA1 &= mA1 - 1u; }
        if (mB0 != 0u) { kB = __builtin_ctz(mB0); mB0 &= mB0 - 1u; } else if (mB1 != 0u) { kB = 32 + __builtin_ctz(mB1); mB1 &= mB1 - 1u; }
        const float4 aA = S.ld(T.trav_off + 3 * kA), bA = S.ld(T.trav_off + 3 * kA + 1), cA = S.ld(T.trav_off + 3 * kA + 2);
        const float4 aB = S.ld(T.trav_off + 3 * kB), bB = S.ld(T.trav_off + 3 * kB + 1), cB = S.ld(T.trav_off + 3 * kB + 2);
        const f2 p0x = {aA.x, aB.x}, p0y = {aA.y, aB.y}, p0z = {aA.z, aB.z};
        const f2 e1x = {aA.w, aB.w}, e1y = {bA.x, bB.x}, e1z = {bA.y, bB.y};
        const f2 e2x = {bA.z, bB.z}, e2y = {bA.w, bB.w}, e2z = {cA.x, cB.x};
        const f2 hx = pk_fma(dy, e2z, -(dz * e2y)), hy = pk_fma(dz, e2x, -(dx * e2z)), hz = pk_fma(dx, e2y, -(dy * e2x));
        const f2 det = pk_fma(e1z, hz, pk_fma(e1y, hy, e1x * hx));
        f2 f; f.x = 1.f / det.x; f.y = 1.f / det.y;
        const f2 sx = ox - p0x, sy = oy - p0y, sz = oz - p0z;
        const f2 u = f * pk_fma(sz, hz, pk_fma(sy, hy, sx * hx));
        const f2 qx = pk_fma(sy, e1z, -(sz * e1y)), qy = pk_fma(sz, e1x, -(sx * e1z)), qz = pk_fma(sx, e1y, -(sy * e1x));
        const f2 v = f * pk_fma(dz, qz, pk_fma(dy, qy, dx * qx));
        const f2 t = f * pk_fma(e2z, qz, pk_fma(e2y, qy, e2x * qx));
        const f2 uv = u + v;
        const int idA = __float_as_int(cA.y), idB = __float_as_int(cB.y);
        const bool okA = moreA & (u.x >= 0.f) & (v.x >= 0.f) & (uv.x <= 1.f) & (t.x > kRayEpsilon) & (t.x < kTraceTMax);
        const bool okB = moreB & (u.y >= 0.f) & (v.y >= 0.f) & (uv.y <= 1.f) & (t.y > kRayEpsilon) & (t.y < kTraceTMax);
        const bool betA = okA & ((t.x < btA) | ((t.x == btA) & (idA < bidA)));
        const bool betB = okB & ((t.y < btB) | ((t.y == btB) & (idB < bidB)));
        if (betA) { btA = t.x; bidA = idA; hA.slot = kA; hA.u = u.x; hA.v = v.x; hA.t = t.x; }
        if (betB) { btB = t.y; bidB = idB; hB.slot = kB; hB.u = u.y; hB.v = v.y; hB.t = t.y; }
    }
}

} // namespace psdr

#include "trav4.h"
