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
constexpr int kParkWords = 13;              // LDS words per lane behind the traversal stack: the lane's two parked rays + the any-hit distance of the first (trav4.h)
constexpr int kColdRows = 15;               // LDS rows of a brute-force scene: per-lane state that only the start and the end of a path touch (paths.h::run_paths)
#ifndef PSDR_BVH_WIDTH                      // children per node of the tree (bvh.h): 4 = the shipped form, 8 = measurement build
#define PSDR_BVH_WIDTH 4
#endif
constexpr int kNodeW4 = PSDR_BVH_WIDTH == 4 ? 4 : 8;       // float4 words per node
#ifndef PSDR_TOP_ROWS                       // LDS rows that hold the top of the tree (trav4.h::kTopNodes = 16 nodes per row; bvh.h numbers that many nodes breadth first)
#define PSDR_TOP_ROWS 2
#endif
constexpr int kTopRows = PSDR_TOP_ROWS;
constexpr int kTravRows = 26 + kTopRows;    // LDS rows (of kBlock words) behind the stack of a BVH scene: parked rays, per-ray best hits, the top of the tree, pair ring, ray queue, heads (trav4.h)
constexpr int kStackLdsMax = 40 - kTravRows;      // stack rows kept in LDS: what is left of 40 KB per workgroup (four workgroups per CU; scene_build.hip)

// EnvironmentMap after configure() (psdr_envmap_rec): too large for the LDS blob, read from global memory
struct EnvDev {
    const float *radiance;          // [height*width*3]
    const float *cell_pmf, *cell_cmf;
    const int *cell_guide;          // search bounds of the cell distribution per bucket of the sample (shade.h::sample_reuse_guided), [guide_n + 1]
    int guide_n;                    // number of buckets (a power of two), 0 = no table
    int width, height, reso0, reso1, num_cells;
    float scale, cell_sum;
    const float *d_radiance;        // forward tangent of the texels, or NULL
    float d_scale;
    Mat4<float> to_world, from_world, d_from_world;
    float lower[3], upper[3];
    float xf[4], d_xf[4];           // m_radiance's uv transform (rotate, scale, translate.x, translate.y; bitmap.h:37-39) and its forward tangent
};

// one bitmap parameter of a BSDF (global memory; w == 0: constant).  SceneTables::tex holds three per BSDF:
// [0] reflectance / diffuse reflectance (rgb), [1] specular reflectance (rgb), [2] roughness (one channel)
struct TexDev {
    const float *data, *d_data; int w, h; long long g_off;     // g_off: offset of its texel adjoints in psdr_grads.g_tex
    float xf[4], d_xf[4];                                      // the bitmap's uv transform (rotate, scale, translate.x, translate.y; bitmap.h:37-39) and its forward tangent
};

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
    int filt_off, n_filt;      // filter primitives of the brute-force tracer (6 words each, see filter.h)
    float center[3], radius;   // bounding sphere of all vertices
    float filt_kmax;           // largest extent of a filter primitive
    unsigned filt_hasb[2];     // per chunk of 32 filter primitives: bit (count - 1 - k) set when primitive k is a quad (has a second triangle)
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
    const int *guide;          // search bounds of the edge distribution per bucket of the sample (shade.h::sample_reuse_guided), [guide_n + 1]; guide_n = 0: none
    int guide_n;
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
    const int *pe_guide;       // search bounds of the primary-edge distribution (as SecEdgeTables::guide)
    int pe_guide_n;
    // one bit per pixel: 0 = no ray through the pixel's footprint can hit a triangle (the conservative screen-space coverage of the scene, api.hip::build_live_mask);
    // a sample of such a pixel contributes exactly zero to the interior term, so the kernels skip it before they seed it.  NULL = every pixel is live
    // (environment-lit scenes: the bounding cube fills the frame; a triangle that crosses the camera plane; more than 2^31 lanes)
    const unsigned *live;
};

struct Counters { unsigned long long rays, nodes, tris, hits; };

// per-lane view used by every device function
// Scene classes (the `LDS` template parameter of every kernel and device function):
//   0  global memory, every feature (GGX BSDFs, bitmap / per-vertex parameters, first-hit integrators, environment map)
//   1  small scene staged in LDS: Diffuse BSDFs and area lights only (all Cornell boxes)
//   2  global memory, lean: Diffuse BSDFs, area lights and the environment map - the BVH scenes of the tutorials and of
//      BASELINE config 5 do not pay registers for material code they do not use
constexpr bool in_lds(int cls) { return cls == 1 || cls == 3; }
constexpr bool has_env(int cls) { return cls == 0 || cls == 2; }
constexpr bool has_mat(int cls) { return cls == 0 || cls == 3; }

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
    bool uv_adj = false;       // reverse mode: the adjoints of the bitmaps' uv transforms are wanted (psdr_grads.g_uv_xf)
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

struct Hit { int slot; float u, v, t; };

// m = 2 m + bit: the lane mask of a comparison (one __builtin_amdgcn_ballot_w64 per v_cmp) enters as the carry-in of
// v_addc_co_u32 - one VALU instruction per mask update instead of v_cndmask + v_lshl_or
PSDR_DEV unsigned mask_shift_in(unsigned m, unsigned long long lanes) {
#if PSDR_MASK_SHIFT == 2
    return (m << 1) | (unsigned) ((lanes >> (threadIdx.x & 63)) & 1ull);
#elif PSDR_MASK_SHIFT == 1
    unsigned r; unsigned long long carry_out;
    asm volatile("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(m), "s"(lanes));
    return r;
#else
    unsigned r; unsigned long long carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(m), "s"(lanes));
    return r;
#endif
}   // slot = device triangle slot (BVH leaf order), -1 = miss

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
template <int LDS, bool COUNT> PSDR_DEV Hit trace_scene(SceneView<LDS> &S, const Vec3f &o, const Vec3f &d, float anyhit = -__builtin_inff());

template <int LDS, bool COUNT>
// anyhit: a hit closer than this ends the search (shadow rays of the reverse sweeps, BVH scenes only; -inf = closest hit)
PSDR_DEV Hit trace(SceneView<LDS> &S, const Vec3f &o, const Vec3f &d, float anyhit = -__builtin_inff()) {
    if (S.mode == 2) {                       // replay: pop the hit the recording run found for this ray
        Hit h; h.slot = -1; h.u = h.v = h.t = 0.f;
        if (S.rec_i < S.rec_n) {
            const float *r = S.rec + 4 * S.rec_i * kBlock;
            h.slot = __float_as_int(r[0]); h.u = r[kBlock]; h.v = r[2 * kBlock]; h.t = r[3 * kBlock];
        }
        ++S.rec_i;
        return h;
    }
    const Hit h = trace_scene<LDS, COUNT>(S, o, d, anyhit);
    if (S.mode == 1) {
        float *r = S.rec + 4 * S.rec_n * kBlock;
        r[0] = __int_as_float(h.slot); r[kBlock] = h.u; r[2 * kBlock] = h.v; r[3 * kBlock] = h.t;
        ++S.rec_n;
    }
    return h;
}

// 4-wide BVH traversal of up to two rays per lane (scenes with more than kBruteForceMax triangles): trav4.h
template <int LDS, bool COUNT>
PSDR_DEV void bvh4_trace2(SceneView<LDS> &S, const Vec3f &oA, const Vec3f &dA, bool actA, const Vec3f &oB, const Vec3f &dB, bool actB, Hit &hA, Hit &hB, float anyhit_a = -__builtin_inff());

template <int LDS, bool COUNT>
PSDR_DEV Hit trace_scene(SceneView<LDS> &S, const Vec3f &o, const Vec3f &d, float anyhit) {
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
        const Vec3f oc = o - Vec3f(T.center[0], T.center[1], T.center[2]);
        const Vec3f m = cross(oc, d);
        const float s_ray = norm(oc) + T.radius + T.filt_kmax, s_t = s_ray * T.filt_kmax;
        for (int base = 0; base < T.n_filt; base += 32) {
            const int cnt = T.n_filt - base < 32 ? T.n_filt - base : 32;
            unsigned ma = 0u, mb = 0u;
            const float4 *prim = S.G + T.filt_off + 6 * base;
            for (int k = 0; k < cnt; ++k) {
                const float4 w0 = prim[6 * k], w1 = prim[6 * k + 1], w2 = prim[6 * k + 2], w3 = prim[6 * k + 3], w4 = prim[6 * k + 4], w5 = prim[6 * k + 5];
                const float un = fmaf(d.z, w1.y, fmaf(d.y, w1.x, fmaf(d.x, w0.w, fmaf(m.z, w0.z, fmaf(m.y, w0.y, m.x * w0.x)))));
                const float me1 = fmaf(m.z, w2.x, fmaf(m.y, w1.w, m.x * w1.z));
                const float vn = fmaf(d.z, w2.w, fmaf(d.y, w2.z, fmaf(d.x, w2.y, -me1)));
                const float dn = fmaf(d.z, w3.z, fmaf(d.y, w3.y, d.x * w3.x));                   // = -det
                const float tn = fmaf(oc.z, w3.z, fmaf(oc.y, w3.y, fmaf(oc.x, w3.x, w3.w)));
                const float sn = __builtin_copysignf(1.f, dn);
                const float ad = dn * sn, ua = -(un * sn), va = -(vn * sn), ta = -(tn * sn);
                const float hi_u = fmaf(w4.x, ad, -ua), hi_v = fmaf(w4.y, ad, -va), hi_s = fmaf(w4.z, ad, -(ua + va));
                const float slack = w5.z * s_ray, margin = fmaf(ad, 3.8146973e-06f, slack), st = w5.z * s_t;
                const float sel = fmaf(w5.x, ua, -(w4.w * va)), band = fmaf(ad, 1e-3f, slack) * w5.y;
                const float mn = __builtin_fminf(__builtin_fminf(__builtin_fminf(ua, va), hi_u), __builtin_fminf(hi_v, hi_s));
                const unsigned long long pass = __builtin_amdgcn_ballot_w64(mn >= -margin) & __builtin_amdgcn_ballot_w64(ta >= -st);
                ma = mask_shift_in(ma, pass & __builtin_amdgcn_ballot_w64(sel >= -band));
                mb = mask_shift_in(mb, pass & __builtin_amdgcn_ballot_w64(sel <= band));
            }
            mb &= T.filt_hasb[base >> 5];
            for (;;) {
                const bool more = (ma | mb) != 0u;
                if (__ballot(more) == 0ull) break;
                int pos = 0, sh = 0;
                if (ma != 0u) { pos = __builtin_ctz(ma); ma &= ma - 1u; } else if (mb != 0u) { pos = __builtin_ctz(mb); mb &= mb - 1u; sh = 8; }
                const int k = (S.ldi(T.filt_off, 24 * (base + cnt - 1 - pos) + 23) >> sh) & 0xff;
                const float4 a = S.ld(T.trav_off + 3 * k), b = S.ld(T.trav_off + 3 * k + 1), c = S.ld(T.trav_off + 3 * k + 2);
                float u, v, t;
                const bool ok = more & tri_test(a, b, c, o, d, u, v, t);
                const int id = __float_as_int(c.y);
                if (ok & ((t < best_t) | ((t == best_t) & (id < best_id)))) { best_t = t; best_id = id; best.slot = k; best.u = u; best.v = v; best.t = t; }
            }
        }
        return best;
    }
    if constexpr (!in_lds(LDS)) {          // (the LDS class holds brute-force scenes only: its kernels carry no tree code)
        Hit other;
        bvh4_trace2<LDS, COUNT>(S, o, d, true, o, d, false, best, other, anyhit);
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
                     Hit &hA, Hit &hB, float anyhit_a = -__builtin_inff()) {        // anyhit_a: ray A may stop at a hit closer than this (BVH scenes only; trace())
#pragma clang fp contract(off)
    hA.slot = -1; hA.u = hA.v = hA.t = 0.f;
    hB.slot = -1; hB.u = hB.v = hB.t = 0.f;
    const SceneTables &T = *S.T;
    if (S.mode != 0) {
        if (actA) hA = trace<LDS, COUNT>(S, oA_, dA, anyhit_a);
        if (actB) hB = trace<LDS, COUNT>(S, oB_, dB);
        return;
    }
    if constexpr (!in_lds(LDS)) { if (T.n_tris > kBruteForceMax) { bvh4_trace2<LDS, COUNT>(S, oA_, dA, actA, oB_, dB, actB, hA, hB, anyhit_a); return; } }
    const float qnan = __builtin_nanf("");
    const Vec3f oA = actA ? oA_ : Vec3f(qnan), oB = actB ? oB_ : Vec3f(qnan);
#if PSDR_DIAG != 6 && PSDR_DIAG != 7 && PSDR_DIAG != 9
    if (COUNT) { const unsigned n = (actA ? 1u : 0u) + (actB ? 1u : 0u); S.c_rays += n; S.c_tris += n * (unsigned) T.n_tris; }
#endif
#if PSDR_DIAG == 7
    // phase timers inside the brute-force tracer (wave cycles): c_rays = set-up, c_nodes = filter loop, c_tris = exact rounds, c_hits = number of exact rounds
    unsigned long long t_ph = __builtin_readcyclecounter();
#define PSDR_TPHASE(field) do { if (COUNT) { const unsigned long long t_now = __builtin_readcyclecounter(); S.field += (unsigned) (t_now - t_ph); t_ph = t_now; } } while (0)
#else
#define PSDR_TPHASE(field) do { } while (0)
#endif
#if PSDR_DIAG == 9
    // candidate statistics: c_rays = candidates (triangles handed to the exact test), c_nodes = active rays, c_tris = exact rounds (x 64), c_hits = calls (x 64)
    if (COUNT) { S.c_nodes += (actA ? 1u : 0u) + (actB ? 1u : 0u); S.c_hits++; }
#endif
    // Both rays ride in the two halves of packed-f32 registers (element-wise identical to the scalar instructions).
    //
    // Two phases.  The exact test spends more than half of its issue cycles on the IEEE division and on eight
    // compares (measured on gfx950, tools/ubench/pkrate.hip: v_mul/v_add/v_fmac 2.6 cycles per wave64, v_fma 3.3,
    // v_cmp / v_min3 / anything reading an SGPR 4.3, a packed instruction 4.4-5.4, v_rcp 8.3, the division
    // sequence 44).  Phase 1 therefore only FILTERS, and in a form that is nothing but dot products of per-ray
    // constants with per-primitive constants the host precomputes (filter.h): with oc = o - centre, m = oc x d,
    //     u-numerator = m.e2 + d.(p x e2)      v-numerator = d.(e1 x p) - m.e1
    //     -det        = d.N                    t-numerator = oc.N - p.N            (p = p0 - centre, N = e1 x e2)
    // 18 packed multiply-adds for two rays instead of the 27 of the cross-product form.  The sign of det is folded
    // into the numerators and one min3 pair tests the window; every test carries an absolute slack
    // 2^-15 K (|oc| + R + K) that covers the roundings of both forms (filter.h), so the filter accepts everything
    // tri_test accepts.  The survivors are one bit per primitive and side of the quad's diagonal, shifted into a
    // per-lane mask by v_addc_co_u32 (the comparison's lane mask is the carry-in).  Phase 2 runs the exact tri_test -
    // same arithmetic, same (t, id) order as ever - on the 1-5 surviving triangles of each lane.
    const Vec3f ctr(T.center[0], T.center[1], T.center[2]);
    const Vec3f ocA = oA - ctr, ocB = oB - ctr;
    const f2 ox = {ocA.x, ocB.x}, oy = {ocA.y, ocB.y}, oz = {ocA.z, ocB.z};
    const f2 dx = {dA.x, dB.x}, dy = {dA.y, dB.y}, dz = {dA.z, dB.z};
    const f2 mx = pk_fma(oy, dz, -(oz * dy)), my = pk_fma(oz, dx, -(ox * dz)), mz = pk_fma(ox, dy, -(oy * dx));
    const float kmax = T.filt_kmax;
    const f2 s_ray = {norm(ocA) + T.radius + kmax, norm(ocB) + T.radius + kmax};
    float btA = __builtin_inff(), btB = __builtin_inff();
    int bidA = 0x7fffffff, bidB = 0x7fffffff;
    PSDR_TPHASE(c_rays);
    for (int base = 0; base < T.n_filt; base += 32) {
        const int cnt = T.n_filt - base < 32 ? T.n_filt - base : 32;
        unsigned aA = 0u, bA = 0u, aB = 0u, bB = 0u;
        {
            // filter primitives (filter.h): {e2, A.x} {A.yz, e1.xy} {e1.z, B} {N, -p.N} {umax, vmax, smax, da} {db, da + db, 2^-15 K, slots}
            const float4 *prim = S.G + T.filt_off + 6 * base;
            for (int k = 0; k < cnt; ++k) {
                const float4 w0 = prim[6 * k], w1 = prim[6 * k + 1], w2 = prim[6 * k + 2], w3 = prim[6 * k + 3], w4 = prim[6 * k + 4], w5 = prim[6 * k + 5];
                const f2 un = pk_fma(dz, (f2) w1.y, pk_fma(dy, (f2) w1.x, pk_fma(dx, (f2) w0.w, pk_fma(mz, (f2) w0.z, pk_fma(my, (f2) w0.y, mx * (f2) w0.x)))));
                const f2 me1 = pk_fma(mz, (f2) w2.x, pk_fma(my, (f2) w1.w, mx * (f2) w1.z));
                const f2 vn = pk_fma(dz, (f2) w2.w, pk_fma(dy, (f2) w2.z, pk_fma(dx, (f2) w2.y, -me1)));
                const f2 dn = pk_fma(dz, (f2) w3.z, pk_fma(dy, (f2) w3.y, dx * (f2) w3.x));          // = -det
                const f2 tn = pk_fma(oz, (f2) w3.z, pk_fma(oy, (f2) w3.y, pk_fma(ox, (f2) w3.x, (f2) w3.w)));
                // fold the sign of det into the numerators: u = ua/ad, v = va/ad, t = ta/ad with ad = |det|
                const f2 sn = {__builtin_copysignf(1.f, dn.x), __builtin_copysignf(1.f, dn.y)};
                const f2 ad = dn * sn, ua = -(un * sn), va = -(vn * sn), ta = -(tn * sn);
                const f2 hi_u = pk_fma((f2) w4.x, ad, -ua), hi_v = pk_fma((f2) w4.y, ad, -va), hi_s = pk_fma((f2) w4.z, ad, -(ua + va));
                const f2 slack = (f2) w5.z * s_ray;                                // absolute slack (filter.h)
                const f2 margin = pk_fma(ad, (f2) 3.8146973e-06f, slack);         // 2^-18 |det| + slack
                const f2 st = slack * (f2) kmax;
                const f2 sel = pk_fma((f2) w5.x, ua, -((f2) w4.w * va));          // side of the quad's diagonal
                const f2 band = pk_fma(ad, (f2) 1e-3f, slack) * (f2) w5.y;
                const float mnA = __builtin_fminf(__builtin_fminf(__builtin_fminf(ua.x, va.x), hi_u.x), __builtin_fminf(hi_v.x, hi_s.x));
                const float mnB = __builtin_fminf(__builtin_fminf(__builtin_fminf(ua.y, va.y), hi_u.y), __builtin_fminf(hi_v.y, hi_s.y));
                const unsigned long long passA = __builtin_amdgcn_ballot_w64(mnA >= -margin.x) & __builtin_amdgcn_ballot_w64(ta.x >= -st.x);
                const unsigned long long passB = __builtin_amdgcn_ballot_w64(mnB >= -margin.y) & __builtin_amdgcn_ballot_w64(ta.y >= -st.y);
                aA = mask_shift_in(aA, passA & __builtin_amdgcn_ballot_w64(sel.x >= -band.x));
                bA = mask_shift_in(bA, passA & __builtin_amdgcn_ballot_w64(sel.x <= band.x));
                aB = mask_shift_in(aB, passB & __builtin_amdgcn_ballot_w64(sel.y >= -band.y));
                bB = mask_shift_in(bB, passB & __builtin_amdgcn_ballot_w64(sel.y <= band.y));
            }
        }
        bA &= T.filt_hasb[base >> 5]; bB &= T.filt_hasb[base >> 5];
        PSDR_TPHASE(c_nodes);
#if PSDR_DIAG == 9
        if (COUNT) S.c_rays += (unsigned) (__builtin_popcount(aA) + __builtin_popcount(bA) + __builtin_popcount(aB) + __builtin_popcount(bB));
#endif
        for (;;) {
            const bool moreA = (aA | bA) != 0u, moreB = (aB | bB) != 0u;
            if (__ballot(moreA || moreB) == 0ull) break;
#if PSDR_DIAG == 7
            if (COUNT) S.c_hits++;
#elif PSDR_DIAG == 9
            if (COUNT) S.c_tris++;
#endif
            int pA = 0, pB = 0, shA = 0, shB = 0;
            if (aA != 0u) { pA = __builtin_ctz(aA); aA &= aA - 1u; } else if (bA != 0u) { pA = __builtin_ctz(bA); bA &= bA - 1u; shA = 8; }
            if (aB != 0u) { pB = __builtin_ctz(aB); aB &= aB - 1u; } else if (bB != 0u) { pB = __builtin_ctz(bB); bB &= bB - 1u; shB = 8; }
            const int kA = (S.ldi(T.filt_off, 24 * (base + cnt - 1 - pA) + 23) >> shA) & 0xff;
            const int kB = (S.ldi(T.filt_off, 24 * (base + cnt - 1 - pB) + 23) >> shB) & 0xff;
            const float4 aA4 = S.ld(T.trav_off + 3 * kA), bA4 = S.ld(T.trav_off + 3 * kA + 1), cA4 = S.ld(T.trav_off + 3 * kA + 2);
            const float4 aB4 = S.ld(T.trav_off + 3 * kB), bB4 = S.ld(T.trav_off + 3 * kB + 1), cB4 = S.ld(T.trav_off + 3 * kB + 2);
            const f2 rox = {oA.x, oB.x}, roy = {oA.y, oB.y}, roz = {oA.z, oB.z};
            const f2 p0x = {aA4.x, aB4.x}, p0y = {aA4.y, aB4.y}, p0z = {aA4.z, aB4.z};
            const f2 e1x = {aA4.w, aB4.w}, e1y = {bA4.x, bB4.x}, e1z = {bA4.y, bB4.y};
            const f2 e2x = {bA4.z, bB4.z}, e2y = {bA4.w, bB4.w}, e2z = {cA4.x, cB4.x};
            const f2 hx = pk_fma(dy, e2z, -(dz * e2y)), hy = pk_fma(dz, e2x, -(dx * e2z)), hz = pk_fma(dx, e2y, -(dy * e2x));
            const f2 det = pk_fma(e1z, hz, pk_fma(e1y, hy, e1x * hx));
            f2 f; f.x = 1.f / det.x; f.y = 1.f / det.y;
            const f2 sx = rox - p0x, sy = roy - p0y, sz = roz - p0z;
            const f2 u = f * pk_fma(sz, hz, pk_fma(sy, hy, sx * hx));
            const f2 qx = pk_fma(sy, e1z, -(sz * e1y)), qy = pk_fma(sz, e1x, -(sx * e1z)), qz = pk_fma(sx, e1y, -(sy * e1x));
            const f2 v = f * pk_fma(dz, qz, pk_fma(dy, qy, dx * qx));
            const f2 t = f * pk_fma(e2z, qz, pk_fma(e2y, qy, e2x * qx));
            const f2 uv = u + v;
            const int idA = __float_as_int(cA4.y), idB = __float_as_int(cB4.y);
            const bool okA = moreA & (u.x >= 0.f) & (v.x >= 0.f) & (uv.x <= 1.f) & (t.x > kRayEpsilon) & (t.x < kTraceTMax);
            const bool okB = moreB & (u.y >= 0.f) & (v.y >= 0.f) & (uv.y <= 1.f) & (t.y > kRayEpsilon) & (t.y < kTraceTMax);
            const bool betA = okA & ((t.x < btA) | ((t.x == btA) & (idA < bidA)));
            const bool betB = okB & ((t.y < btB) | ((t.y == btB) & (idB < bidB)));
            if (betA) { btA = t.x; bidA = idA; hA.slot = kA; hA.u = u.x; hA.v = v.x; hA.t = t.x; }
            if (betB) { btB = t.y; bidB = idB; hB.slot = kB; hB.u = u.y; hB.v = v.y; hB.t = t.y; }
        }
        PSDR_TPHASE(c_tris);
    }
#undef PSDR_TPHASE
}

} // namespace psdr

#include "trav4.h"
