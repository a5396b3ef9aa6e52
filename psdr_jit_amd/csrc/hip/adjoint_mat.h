// adjoint_mat.h - the reverse sweep of the interior term (adjoint.h::run_interior_adjoint_sweep) for scenes with isotropic GGX
// BSDFs (psdr.MicrofacetBSDF with constant parameters, beside Diffuse ones): scene class 0.
//
// Same structure - pass 1 walks the path forward with the primal arithmetic of D mode and records (slot, u, v) per vertex and the
// constants of each bounce, pass 2 walks back - but the BSDF is a function F(wi, wo) of BOTH directions in the shading frame
// (microfacet.cpp:22-62), so a segment factors into F and the geometry term g = |nz.w| / r^2 . A / detach(A):
//     L += thr_k . F(wi_k, w) . Le . g cN          thr_{k+1} = thr_k . F(wi_k, w) . g cf
// with w = (z - x_k) / r and wi_k = (x_{k-1} - x_k) / r' (the camera ray at the first vertex; scene.cpp:686-690 rebuilds it from the
// positions).  The adjoint of F comes from forward evaluations of microfacet_eval<Dual> with unit tangents (six direction components,
// specular, roughness, diffuse); it goes to w and wi as vectors - and through them to x_k, z and x_{k-1} - and to ns by the rotation
// of the frame (an isotropic lobe does not see the tangents), and to the BSDF's parameters (g_mat rows, g_bsdf).
#pragma once
#include "adjoint.h"

namespace psdr {

struct GeoGrad { Vec3f dx, dz, dnz; float dA; };      // d g / d (x, z, nz, A_z)
// g = |nz.w| / r^2 and its gradient
PSDR_DEV float geo_eval(const Vec3f &x, const Vec3f &z, const Vec3f &nz, float area_z, GeoGrad &g) {
    const Vec3f v = z - x;
    const float r2 = dot(v, v);
    g.dx = g.dz = g.dnz = Vec3f(0.f); g.dA = 0.f;
    if (!(r2 > 0.f)) return 0.f;
    const float ir = 1.f / sqrtf(r2), ir3 = ir * ir * ir;
    const float Qs = dot(nz, v), sq = Qs < 0.f ? -1.f : 1.f, Q = Qs * sq;
    const float G = Q * ir3;
    g.dnz = v * (sq * ir3);
    g.dz = nz * (sq * ir3) - v * (3.f * G * ir * ir);
    g.dx = -g.dz;
    g.dA = area_z > 0.f ? G / area_z : 0.f;
    return G;
}

// the tangent vectors of a vertex's shading frame as make_its builds them (scene.cpp:724-766): from the uv parameterisation
// (dp_du = (e1 dv1 - e2 dv0) / det, Gram-Schmidt against ns) when it is non-degenerate, else Duff et al.'s frame of ns
template <typename R>
PSDR_DEV void vertex_frame(const Vec3<R> &ns, const Vec3<R> &e1, const Vec3<R> &e2, float du0x, float du0y, float du1x, float du1y, Vec3<R> &fs, Vec3<R> &ft) {
    coordinate_system(ns, fs, ft);
    const float det = fma_(du0x, du1y, -(du0y * du1x));
    if (det != 0.f) {
        const float inv_det = 1.f / det;
        const Vec3<R> dp_du = (e1 * R(du1y) - e2 * R(du0y)) * R(inv_det);
        fs = normalize(dp_du - ns * dot(ns, dp_du));
        ft = cross(ns, fs);
    }
}

// F(wi, wo) of BSDF `bid` in local coordinates (Diffuse, Microfacet or RoughConductor; constant or bitmap parameters looked up at (tu, tv)) and, when
// `Fb` is given, the adjoints of the six direction components.  The parameter adjoints are added to acc_bsdf (constant colour /
// diffuse reflectance) and acc_mat (g_mat row: constant specular rgb, roughness) or, for a bitmap parameter, scattered over the four
// texels of the lookup (g_tex, TexDev::g_off) and - uvb given - chained to the texture coordinates (the camera vertex: its
// barycentrics are differentiable).
template <int LDS>
PSDR_DEV Vec3f bsdf_plain_value_and_adjoint(const SceneView<LDS> &S, int bid, const Vec3f &wi, const Vec3f &wo, float tu, float tv, const Vec3f *Fb,
                                            float *wib, float *wob, float *acc_bsdf, float *acc_mat, float *g_tex, float *uvb,
                                            int slot = -1, float bu = 0.f, float bv = 0.f, float *bcb = nullptr) {
    if (wib) { wib[0] = wib[1] = wib[2] = 0.f; wob[0] = wob[1] = wob[2] = 0.f; }
    const int mrow = S.uv_adj ? kMatRow : kMatOut;                     // LDS row of a BSDF's material adjoints (kMatRow)
    if (bid < 0) return Vec3f(0.f);
    const float4 a = S.ld(S.T->bsdf_off + 2 * bid);
    const int fl = __float_as_int(a.w);
    const bool two = (fl & 1) != 0;
    auto add = [](float *p, float v) { if (v != 0.f && finite_(v)) atomicAdd(p, v); };
    Vec3f diff(a.x, a.y, a.z);
    // adjoint pb[CH] of the value looked up in texture slot `slot`: texels by the footprint, (tu, tv) by two forward evaluations
    auto tex_value = [&](int slot, auto ch, float *out) {
        if constexpr (has_mat(LDS)) {
            constexpr int CH = decltype(ch)::value;
            const TexDev td = S.T->tex[3 * bid + slot];
            env::bitmap_eval_tex<float, CH>([&](int i, int c) { return td.data[CH * i + c]; }, td.w, td.h, tu, tv, true, out, env::UvXf<float>(td.xf));
        }
    };
    auto tex_back = [&](int slot, auto ch, const float *pb) {
        if constexpr (has_mat(LDS)) {
            constexpr int CH = decltype(ch)::value;
            const TexDev td = S.T->tex[3 * bid + slot];
            if (g_tex != nullptr) {
                int idx[4]; float wt[4];
                env::bitmap_footprint(td.w, td.h, tu, tv, true, idx, wt, env::UvXf<float>(td.xf));
                for (int c = 0; c < CH; ++c)
                    if (pb[c] != 0.f && finite_(pb[c])) for (int k = 0; k < 4; ++k) atomicAdd(&g_tex[td.g_off + (long long) CH * idx[k] + c], pb[c] * wt[k]);
            }
            if (S.uv_adj && acc_mat != nullptr) tex_xf_adjoint<CH>(td, tu, tv, pb, &acc_mat[bid * mrow + kMatOut + 4 * slot]);
            if (uvb != nullptr) {
                for (int ax = 0; ax < 2; ++ax) {
                    Dual o[CH];
                    env::bitmap_eval_tex<Dual, CH>([&](int i, int c) { return Dual(td.data[CH * i + c], 0.f); }, td.w, td.h, Dual(tu, ax == 0 ? 1.f : 0.f), Dual(tv, ax == 1 ? 1.f : 0.f), true, o, uv_xf_d(td.xf, td.xf, false));
                    float g = 0.f;
                    for (int c = 0; c < CH; ++c) g += pb[c] * o[c].d;
                    if (finite_(g)) uvb[ax] += g;
                }
            }
        }
    };
    if constexpr (has_mat(LDS)) {
        if (fl & 2) { float o[3]; tex_value(0, std::integral_constant<int, 3>(), o); diff = Vec3f(o[0], o[1], o[2]); }
        if (fl & 4) {
            const MatDev md = S.T->mat[bid];
            Vec3f spec(md.specular[0], md.specular[1], md.specular[2]);
            float rough = md.roughness;
            if (fl & 32) { float o[3]; tex_value(1, std::integral_constant<int, 3>(), o); spec = Vec3f(o[0], o[1], o[2]); }
            if (fl & 64) { float o[1]; tex_value(2, std::integral_constant<int, 1>(), o); rough = o[0]; }
            const Vec3f F = microfacet_eval<float>(spec, diff, rough, two, wi, wo, true);
            if (Fb == nullptr) return F;
            if (!(finite_(Fb->x) && finite_(Fb->y) && finite_(Fb->z)) || (Fb->x == 0.f && Fb->y == 0.f && Fb->z == 0.f)) return F;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const float one = 1.f;
                const Vec3d wiD(Dual(wi.x, j == 0 ? one : 0.f), Dual(wi.y, j == 1 ? one : 0.f), Dual(wi.z, j == 2 ? one : 0.f));
                const Vec3d woD(Dual(wo.x, j == 3 ? one : 0.f), Dual(wo.y, j == 4 ? one : 0.f), Dual(wo.z, j == 5 ? one : 0.f));
                const float ts = j == 6 ? one : 0.f, tr = j == 7 ? one : 0.f, td = j == 8 ? one : 0.f;
                const Vec3d specD(Dual(spec.x, ts), Dual(spec.y, ts), Dual(spec.z, ts));
                const Vec3d diffD(Dual(diff.x, td), Dual(diff.y, td), Dual(diff.z, td));
                const Vec3d r = microfacet_eval<Dual>(specD, diffD, Dual(rough, tr), two, wiD, woD, true);
                const float pb[3] = {Fb->x * r.x.d, Fb->y * r.y.d, Fb->z * r.z.d};
                if (j < 3) wib[j] = pb[0] + pb[1] + pb[2];
                else if (j < 6) wob[j - 3] = pb[0] + pb[1] + pb[2];
                else if (j == 6) {
                    if (fl & 32) tex_back(1, std::integral_constant<int, 3>(), pb);
                    else if (acc_mat) { add(&acc_mat[bid * mrow], pb[0]); add(&acc_mat[bid * mrow + 1], pb[1]); add(&acc_mat[bid * mrow + 2], pb[2]); }
                } else if (j == 7) {
                    const float rb[1] = {pb[0] + pb[1] + pb[2]};
                    if (fl & 64) tex_back(2, std::integral_constant<int, 1>(), rb);
                    else if (acc_mat) add(&acc_mat[bid * mrow + 3], rb[0]);
                } else {
                    if (fl & 2) tex_back(0, std::integral_constant<int, 3>(), pb);
                    else if (acc_bsdf) { add(&acc_bsdf[3 * bid], pb[0]); add(&acc_bsdf[3 * bid + 1], pb[1]); add(&acc_bsdf[3 * bid + 2], pb[2]); }
                }
            }
            for (int j = 0; j < 3; ++j) { if (!finite_(wib[j])) wib[j] = 0.f; if (!finite_(wob[j])) wob[j] = 0.f; }
            return F;
        }
    }
    if constexpr (has_mat(LDS)) {
        if (fl & 128) {        // MicrofacetPerVertex (microfacet_pv.cpp): the three parameters interpolated over the hit triangle's vertices
            const PvDev pv = S.T->pv[bid];
            const int *fi = S.T->tri_fi + 3 * slot;
            auto lerp = [&](const float *val, int stride, int c, float *du, float *dv) {
                const float v0 = val[stride * fi[0] + c], v1 = val[stride * fi[1] + c], v2 = val[stride * fi[2] + c];
                *du = v1 - v0; *dv = v2 - v0;
                return fma_(v1 - v0, bu, fma_(v2 - v0, bv, v0));
            };
            float dsu[3], dsv[3], ddu[3], ddv[3], dru, drv;
            const Vec3f spec(lerp(pv.spec, 3, 0, &dsu[0], &dsv[0]), lerp(pv.spec, 3, 1, &dsu[1], &dsv[1]), lerp(pv.spec, 3, 2, &dsu[2], &dsv[2]));
            const Vec3f dif(lerp(pv.diff, 3, 0, &ddu[0], &ddv[0]), lerp(pv.diff, 3, 1, &ddu[1], &ddv[1]), lerp(pv.diff, 3, 2, &ddu[2], &ddv[2]));
            const float rough = lerp(pv.rough, 1, 0, &dru, &drv);
            const Vec3f F = microfacet_pv_eval<float>(spec, dif, rough, two, wi, wo, true);
            if (Fb == nullptr) return F;
            if (!(finite_(Fb->x) && finite_(Fb->y) && finite_(Fb->z)) || (Fb->x == 0.f && Fb->y == 0.f && Fb->z == 0.f)) return F;
            const float wv[3] = {1.f - bu - bv, bu, bv};
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const float one = 1.f;
                const Vec3d wiD(Dual(wi.x, j == 0 ? one : 0.f), Dual(wi.y, j == 1 ? one : 0.f), Dual(wi.z, j == 2 ? one : 0.f));
                const Vec3d woD(Dual(wo.x, j == 3 ? one : 0.f), Dual(wo.y, j == 4 ? one : 0.f), Dual(wo.z, j == 5 ? one : 0.f));
                const float ts = j == 6 ? one : 0.f, tr = j == 7 ? one : 0.f, td = j == 8 ? one : 0.f;
                const Vec3d specD(Dual(spec.x, ts), Dual(spec.y, ts), Dual(spec.z, ts)), diffD(Dual(dif.x, td), Dual(dif.y, td), Dual(dif.z, td));
                const Vec3d r = microfacet_pv_eval<Dual>(specD, diffD, Dual(rough, tr), two, wiD, woD, true);
                const float pb[3] = {Fb->x * r.x.d, Fb->y * r.y.d, Fb->z * r.z.d};
                if (j < 3) wib[j] = pb[0] + pb[1] + pb[2];
                else if (j < 6) wob[j - 3] = pb[0] + pb[1] + pb[2];
                else {
                    // value adjoints: to the three vertices with the barycentric weights (g_tex blocks: diffuse, specular, roughness), and at the
                    // camera vertex to the barycentrics
                    const int tslot = j == 8 ? 0 : (j == 6 ? 1 : 2), ch = j == 7 ? 1 : 3;
                    const float rb = pb[0] + pb[1] + pb[2];
                    for (int c = 0; c < ch; ++c) {
                        const float val = ch == 1 ? rb : pb[c];
                        if (val == 0.f || !finite_(val)) continue;
                        if (g_tex != nullptr) for (int q = 0; q < 3; ++q) atomicAdd(&g_tex[pv.g_off[tslot] + (long long) ch * fi[q] + c], val * wv[q]);
                        if (bcb != nullptr) {
                            const float du = j == 8 ? ddu[c] : (j == 6 ? dsu[c] : dru), dv = j == 8 ? ddv[c] : (j == 6 ? dsv[c] : drv);
                            bcb[0] += val * du; bcb[1] += val * dv;
                        }
                    }
                }
            }
            for (int j = 0; j < 3; ++j) { if (!finite_(wib[j])) wib[j] = 0.f; if (!finite_(wob[j])) wob[j] = 0.f; }
            return F;
        }
        if (fl & 8) {          // RoughConductor with constant parameters (roughconductor.cpp:30-68): g_mat row = [alpha_u, alpha_v, eta rgb, k rgb, specular rgb]
            const MatDev md = S.T->mat[bid];
            Vec3f eta(md.eta[0], md.eta[1], md.eta[2]), kk(md.k[0], md.k[1], md.k[2]);
            const Vec3f spec(md.specular[0], md.specular[1], md.specular[2]);
            float cau = md.alpha_u, cav = md.alpha_v;
            // bitmap parameters (round 3; shade.h::bsdf_eval_id): slot 0 eta, slot 1 k, slot 2 alpha (one map for both axes)
            if (fl & 2) { float o[3]; tex_value(0, std::integral_constant<int, 3>(), o); eta = Vec3f(o[0], o[1], o[2]); }
            if (fl & 32) { float o[3]; tex_value(1, std::integral_constant<int, 3>(), o); kk = Vec3f(o[0], o[1], o[2]); }
            if (fl & 64) { float o[1]; tex_value(2, std::integral_constant<int, 1>(), o); cau = o[0]; cav = o[0]; }
            const Vec3f F = conductor_eval<float>(cau, cav, eta, kk, spec, two, wi, wo, true);
            if (Fb == nullptr) return F;
            if (!(finite_(Fb->x) && finite_(Fb->y) && finite_(Fb->z)) || (Fb->x == 0.f && Fb->y == 0.f && Fb->z == 0.f)) return F;
#pragma unroll
            for (int j = 0; j < 11; ++j) {
                const float one = 1.f;
                const Vec3d wiD(Dual(wi.x, j == 0 ? one : 0.f), Dual(wi.y, j == 1 ? one : 0.f), Dual(wi.z, j == 2 ? one : 0.f));
                const Vec3d woD(Dual(wo.x, j == 3 ? one : 0.f), Dual(wo.y, j == 4 ? one : 0.f), Dual(wo.z, j == 5 ? one : 0.f));
                const float te = j == 8 ? one : 0.f, tk = j == 9 ? one : 0.f, ts = j == 10 ? one : 0.f;
                const Vec3d etaD(Dual(eta.x, te), Dual(eta.y, te), Dual(eta.z, te)), kD(Dual(kk.x, tk), Dual(kk.y, tk), Dual(kk.z, tk));
                const Vec3d specD(Dual(spec.x, ts), Dual(spec.y, ts), Dual(spec.z, ts));
                const Vec3d r = conductor_eval<Dual>(Dual(cau, j == 6 ? one : 0.f), Dual(cav, j == 7 ? one : 0.f), etaD, kD, specD, two, wiD, woD, true);
                const float pb[3] = {Fb->x * r.x.d, Fb->y * r.y.d, Fb->z * r.z.d};
                if (j < 3) wib[j] = pb[0] + pb[1] + pb[2];
                else if (j < 6) wob[j - 3] = pb[0] + pb[1] + pb[2];
                else if (j < 8 && (fl & 64)) { const float rb[1] = {pb[0] + pb[1] + pb[2]}; tex_back(2, std::integral_constant<int, 1>(), rb); }      // the alpha map feeds both axes
                else if (j == 8 && (fl & 2)) tex_back(0, std::integral_constant<int, 3>(), pb);
                else if (j == 9 && (fl & 32)) tex_back(1, std::integral_constant<int, 3>(), pb);
                else if (acc_mat) {
                    if (j < 8) add(&acc_mat[bid * mrow + (j - 6)], pb[0] + pb[1] + pb[2]);
                    else { const int o = j == 8 ? 2 : (j == 9 ? 5 : 8); add(&acc_mat[bid * mrow + o], pb[0]); add(&acc_mat[bid * mrow + o + 1], pb[1]); add(&acc_mat[bid * mrow + o + 2], pb[2]); }
                }
            }
            for (int j = 0; j < 3; ++j) { if (!finite_(wib[j])) wib[j] = 0.f; if (!finite_(wob[j])) wob[j] = 0.f; }
            return F;
        }
    }
    if constexpr (has_mat(LDS)) {
        if (fl & 16) {         // RoughDielectric with constant parameters (roughdielectric.cpp): g_mat row = [alpha_u, alpha_v, eta]; 1 / eta moves with eta
            const MatDev md = S.T->mat[bid];
            float cau = md.alpha_u, cav = md.alpha_v;
            if (fl & 64) { float o[1]; tex_value(2, std::integral_constant<int, 1>(), o); cau = o[0]; cav = o[0]; }      // alpha bitmap (slot 2, both axes)
            const Vec3f F = dielectric_eval<float>(cau, cav, md.eta[0], md.eta[1], two, wi, wo, true);
            if (Fb == nullptr) return F;
            if (!(finite_(Fb->x) && finite_(Fb->y) && finite_(Fb->z)) || (Fb->x == 0.f && Fb->y == 0.f && Fb->z == 0.f)) return F;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const float one = 1.f;
                const Vec3d wiD(Dual(wi.x, j == 0 ? one : 0.f), Dual(wi.y, j == 1 ? one : 0.f), Dual(wi.z, j == 2 ? one : 0.f));
                const Vec3d woD(Dual(wo.x, j == 3 ? one : 0.f), Dual(wo.y, j == 4 ? one : 0.f), Dual(wo.z, j == 5 ? one : 0.f));
                const Vec3d r = dielectric_eval<Dual>(Dual(cau, j == 6 ? one : 0.f), Dual(cav, j == 7 ? one : 0.f), Dual(md.eta[0], j == 8 ? one : 0.f),
                                                      Dual(md.eta[1], j == 8 ? -1.f / (md.eta[0] * md.eta[0]) : 0.f), two, wiD, woD, true);
                const float pb = Fb->x * r.x.d + Fb->y * r.y.d + Fb->z * r.z.d;
                if (j < 3) wib[j] = pb;
                else if (j < 6) wob[j - 3] = pb;
                else if (j < 8 && (fl & 64)) { const float rb[1] = {pb}; tex_back(2, std::integral_constant<int, 1>(), rb); }
                else if (acc_mat) add(&acc_mat[bid * mrow + (j - 6)], pb);
            }
            for (int j = 0; j < 3; ++j) { if (!finite_(wib[j])) wib[j] = 0.f; if (!finite_(wob[j])) wob[j] = 0.f; }
            return F;
        }
    }
    // Diffuse (diffuse.cpp:30-41): rho / pi . wo.z on the lit side
    float wiz = wi.z, woz = wo.z, sg = 1.f;
    if (two) { sg = wiz < 0.f ? -1.f : 1.f; woz = woz * sg; wiz = fabsf(wiz); }
    if (!(wiz > 0.f && woz > 0.f)) return Vec3f(0.f);
    const Vec3f F = diff * (kInvPi * woz);
    if (Fb != nullptr && finite_(Fb->x) && finite_(Fb->y) && finite_(Fb->z)) {
        wob[2] = sg * kInvPi * (Fb->x * diff.x + Fb->y * diff.y + Fb->z * diff.z);
        const float pb[3] = {Fb->x * kInvPi * woz, Fb->y * kInvPi * woz, Fb->z * kInvPi * woz};
        if (fl & 2) tex_back(0, std::integral_constant<int, 3>(), pb);
        else if (acc_bsdf) { add(&acc_bsdf[3 * bid], pb[0]); add(&acc_bsdf[3 * bid + 1], pb[1]); add(&acc_bsdf[3 * bid + 2], pb[2]); }
    }
    return F;
}

// NormalMap (normalmap.cpp:45-83; shade.h::normalmap_eval): what the map does to the two directions, as a function of (wi, wo, map value c, dp_du) -
// the directions in the perturbed frame, the reflected incident direction in it, the two weights
template <typename R>
PSDR_DEV void nm_geometry(const Vec3<R> &wi, const Vec3<R> &wo, const Vec3<R> &c, const Vec3<R> &dpdu, Vec3<R> &pwi, Vec3<R> &pwo, Vec3<R> &rwi, R &lp, R &g1, bool &refl) {
    const Vec3<R> wp = normalize(Vec3<R>(fma_(c.x, R(2.f), R(-1.f)), fma_(c.y, R(2.f), R(-1.f)), fma_(c.z, R(2.f), R(-1.f))));
    const Vec3<R> s = normalize(dpdu - wp * dot(wp, dpdu));
    const NmFrame<R> pf(wp, s);
    pwi = pf.to_local(wi); pwo = pf.to_local(wo);
    g1 = nm_G1(wp, wo); lp = nm_lambda_p(wp, wi);
    const Vec3<R> wt = nm_wt(wp);
    refl = detach(dot(wi, wt)) > 0.f;
    const Vec3<R> wi_r = normalize(wi - wt * (R(2.0f) * dot(wi, wt)));
    rwi = pf.to_local(wi_r);
}

// F(wi, wo) and its adjoints for any BSDF of the material sweep.  A NormalMap (round 3) is its nested BSDF seen through the map:
//     F = N(pwi, pwo) lp g1 + [wi.wt > 0] N(rwi, pwo) (1 - lp) g1,
// so the nested BSDF's own adjoint routine gives the adjoints of its parameters and of (pwi, pwo, rwi); the 11 quantities nm_geometry returns
// are differentiated with respect to its 12 inputs by forward evaluations, which chains those - and the adjoints of the two weights - to wi, wo,
// the map value (constant: the NormalMap's g_bsdf row; bitmap: the four texels of the lookup and, at the camera vertex, the texture
// coordinates) and dp_du (`dpb`: the caller takes it to the triangle's edges).  dpdu / dpb: the vertex's dp_du (make_its), needed by NormalMaps only.
template <int LDS>
PSDR_DEV Vec3f bsdf_value_and_adjoint(const SceneView<LDS> &S, int bid, const Vec3f &wi_, const Vec3f &wo_, float tu, float tv, const Vec3f *Fb,
                                      float *wib, float *wob, float *acc_bsdf, float *acc_mat, float *g_tex, float *uvb,
                                      int slot = -1, float bu = 0.f, float bv = 0.f, float *bcb = nullptr, const Vec3f *dpdu = nullptr, float *dpb = nullptr) {
    const int mrow = S.uv_adj ? kMatRow : kMatOut;                     // LDS row of a BSDF's material adjoints (kMatRow)
    if constexpr (has_mat(LDS)) {
        if (bid >= 0 && (__float_as_int(S.ld(S.T->bsdf_off + 2 * bid).w) & 256)) {
            if (wib) { wib[0] = wib[1] = wib[2] = 0.f; wob[0] = wob[1] = wob[2] = 0.f; }
            if (dpb) { dpb[0] = dpb[1] = dpb[2] = 0.f; }
            const float4 a = S.ld(S.T->bsdf_off + 2 * bid);
            const int fl = __float_as_int(a.w);
            const int nested = __float_as_int(S.ld(S.T->bsdf_off + 2 * bid + 1).w);
            Vec3f wi = wi_, wo = wo_;
            float sgn = 1.f;
            if (fl & 1) { sgn = wi.z < 0.f ? -1.f : 1.f; wo.z = wo.z * sgn; wi.z = fabsf(wi.z); }       // (mulsign / abs of the two-sided form)
            if (!(wi.z > 0.f && wo.z > 0.f)) return Vec3f(0.f);
            Vec3f c(a.x, a.y, a.z);
            const TexDev td = (fl & 2) ? S.T->tex[3 * bid] : TexDev{};
            if (fl & 2) {
                float o[3];
                env::bitmap_eval_tex<float, 3>([&](int i, int ch) { return td.data[3 * i + ch]; }, td.w, td.h, tu, tv, true, o, env::UvXf<float>(td.xf));
                c = Vec3f(o[0], o[1], o[2]);
            }
            const Vec3f dp = dpdu ? *dpdu : Vec3f(0.f);
            Vec3f pwi, pwo, rwi;
            float lp, g1;
            bool refl;
            nm_geometry<float>(wi, wo, c, dp, pwi, pwo, rwi, lp, g1, refl);
            const Vec3f N1 = bsdf_plain_value_and_adjoint<LDS>(S, nested, pwi, pwo, tu, tv, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, slot, bu, bv, nullptr);
            Vec3f N2(0.f);
            if (refl) N2 = bsdf_plain_value_and_adjoint<LDS>(S, nested, rwi, pwo, tu, tv, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, slot, bu, bv, nullptr);
            const Vec3f F = N1 * (lp * g1) + N2 * ((1.f - lp) * g1);
            if (Fb == nullptr) return F;
            if (!(finite_(Fb->x) && finite_(Fb->y) && finite_(Fb->z)) || (Fb->x == 0.f && Fb->y == 0.f && Fb->z == 0.f)) return F;
            // the nested BSDF's adjoints (its parameters accumulate inside) for the two evaluations
            float pwib[3], pwob[3], rwib[3] = {0.f, 0.f, 0.f}, pwob2[3] = {0.f, 0.f, 0.f};
            const Vec3f Nb1 = *Fb * (lp * g1);
            bsdf_plain_value_and_adjoint<LDS>(S, nested, pwi, pwo, tu, tv, &Nb1, pwib, pwob, acc_bsdf, acc_mat, g_tex, uvb, slot, bu, bv, bcb);
            if (refl) {
                const Vec3f Nb2 = *Fb * ((1.f - lp) * g1);
                bsdf_plain_value_and_adjoint<LDS>(S, nested, rwi, pwo, tu, tv, &Nb2, rwib, pwob2, acc_bsdf, acc_mat, g_tex, uvb, slot, bu, bv, bcb);
            }
            const Vec3f dN = N1 - N2, mix = N1 * lp + N2 * (1.f - lp);
            const float lpb = g1 * (Fb->x * dN.x + Fb->y * dN.y + Fb->z * dN.z), g1b = Fb->x * mix.x + Fb->y * mix.y + Fb->z * mix.z;
            float inb[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const float one = 1.f;
                const Vec3d wiD(Dual(wi.x, j == 0 ? one : 0.f), Dual(wi.y, j == 1 ? one : 0.f), Dual(wi.z, j == 2 ? one : 0.f));
                const Vec3d woD(Dual(wo.x, j == 3 ? one : 0.f), Dual(wo.y, j == 4 ? one : 0.f), Dual(wo.z, j == 5 ? one : 0.f));
                const Vec3d cD(Dual(c.x, j == 6 ? one : 0.f), Dual(c.y, j == 7 ? one : 0.f), Dual(c.z, j == 8 ? one : 0.f));
                const Vec3d dD(Dual(dp.x, j == 9 ? one : 0.f), Dual(dp.y, j == 10 ? one : 0.f), Dual(dp.z, j == 11 ? one : 0.f));
                Vec3d pwiD, pwoD, rwiD;
                Dual lpD, g1D;
                bool r2;
                nm_geometry<Dual>(wiD, woD, cD, dD, pwiD, pwoD, rwiD, lpD, g1D, r2);
                float g = pwib[0] * pwiD.x.d + pwib[1] * pwiD.y.d + pwib[2] * pwiD.z.d
                        + (pwob[0] + pwob2[0]) * pwoD.x.d + (pwob[1] + pwob2[1]) * pwoD.y.d + (pwob[2] + pwob2[2]) * pwoD.z.d
                        + lpb * lpD.d + g1b * g1D.d;
                if (refl) g += rwib[0] * rwiD.x.d + rwib[1] * rwiD.y.d + rwib[2] * rwiD.z.d;
                inb[j] = finite_(g) ? g : 0.f;
            }
            if (wib) { wib[0] = inb[0]; wib[1] = inb[1]; wib[2] = inb[2] * sgn; wob[0] = inb[3]; wob[1] = inb[4]; wob[2] = inb[5] * sgn; }
            // the map value: a constant's adjoint is the NormalMap's g_bsdf row, a bitmap's goes to the four texels of the lookup and to the texture coordinates
            if (fl & 2) {
                if (g_tex != nullptr) {
                    int idx[4]; float wt4[4];
                    env::bitmap_footprint(td.w, td.h, tu, tv, true, idx, wt4, env::UvXf<float>(td.xf));
                    for (int ch = 0; ch < 3; ++ch)
                        if (inb[6 + ch] != 0.f) for (int k = 0; k < 4; ++k) atomicAdd(&g_tex[td.g_off + 3ll * idx[k] + ch], inb[6 + ch] * wt4[k]);
                }
                if (S.uv_adj && acc_mat != nullptr) tex_xf_adjoint<3>(td, tu, tv, &inb[6], &acc_mat[bid * mrow + kMatOut]);
                if (uvb != nullptr) {
                    for (int ax = 0; ax < 2; ++ax) {
                        Dual o[3];
                        env::bitmap_eval_tex<Dual, 3>([&](int i, int ch) { return Dual(td.data[3 * i + ch], 0.f); }, td.w, td.h, Dual(tu, ax == 0 ? 1.f : 0.f), Dual(tv, ax == 1 ? 1.f : 0.f), true, o, uv_xf_d(td.xf, td.xf, false));
                        const float g = inb[6] * o[0].d + inb[7] * o[1].d + inb[8] * o[2].d;
                        if (finite_(g)) uvb[ax] += g;
                    }
                }
            } else if (acc_bsdf) {
                for (int ch = 0; ch < 3; ++ch) if (inb[6 + ch] != 0.f) atomicAdd(&acc_bsdf[3 * bid + ch], inb[6 + ch]);
            }
            if (dpb) { dpb[0] = inb[9]; dpb[1] = inb[10]; dpb[2] = inb[11]; }
            return F;
        }
    }
    return bsdf_plain_value_and_adjoint<LDS>(S, bid, wi_, wo_, tu, tv, Fb, wib, wob, acc_bsdf, acc_mat, g_tex, uvb, slot, bu, bv, bcb);
}

template <int LDS>
PSDR_DEV void run_interior_adjoint_sweep_mat(SceneView<LDS> &S, const SensorDev &cam, const AdjointParams &P, float *scratch) {
    const SceneTables &T = *S.T;
    const int mrow = S.uv_adj ? kMatRow : kMatOut;                     // LDS row of a BSDF's material adjoints (kMatRow)
    const int lane_id = threadIdx.x & 63;
    const unsigned long long lt_mask = (1ull << lane_id) - 1ull;
    const float inv_spp = T.spp > 1 ? 1.f / (float) T.spp : 1.f;
    const int D = P.max_depth;
    const int lane_words = P.hit_words + P.ext_words + P.lk_words;        // sized by the launch: >= adj_sweep_words(D)
    float *vrec = (P.rec_global ? P.rec_global + (size_t) blockIdx.x * (size_t) lane_words * kBlock : scratch) + threadIdx.x;    // [3 * (D + 1)] slot, u, v per vertex, stride kBlock
    float *brec = vrec + 3 * (D + 1) * kBlock;                            // [11 * D] per bounce: 0 light slot, 1-2 its barycentrics, 3 shadow-hit slot,
                                                                          //   4 cN, 5 cf, 6 w2, 7 flags, 8-10 thr_k
    float *acc_cam = P.rec_global ? scratch : scratch + lane_words * kBlock;      // same accumulator layout as run_interior_adjoint
    float *acc_mat = acc_cam + kAdjMisc;
    float *acc = acc_mat + T.n_bsdfs * mrow;
    const int n_acc = P.n_hot * 22 + T.n_bsdfs * 3 + T.n_emitters * 3 + P.env_lds;
    for (int i = threadIdx.x; i < n_acc; i += kBlock) acc[i] = 0.f;
    float *acc_bsdf = acc + P.n_hot * 22, *acc_emit = acc_bsdf + T.n_bsdfs * 3;
    float *acc_env = acc_emit + T.n_emitters * 3;        // [env_lds] texel adjoints of a small environment map (every sample of a wave hits the same few texels)
    if (threadIdx.x < kAdjMisc) acc_cam[threadIdx.x] = 0.f;
    for (int i = threadIdx.x; i < T.n_bsdfs * mrow; i += kBlock) acc_mat[i] = 0.f;
    __syncthreads();
    S.mode = 0; S.probe_kind = 0;

    auto add_row = [&](const VtxGeom &g, int comp, float val) {
        if (val == 0.f || !finite_(val)) return;
        const int hot = P.hot_map[g.orig];
        if (hot >= 0 && hot < P.n_hot) atomicAdd(&acc[hot * 22 + comp], val); else atomicAdd(&P.g_tri[g.orig * 22 + comp], val);
    };
    auto add_vec = [&](const VtxGeom &g, int comp, const Vec3f &val) { add_row(g, comp, val.x); add_row(g, comp + 1, val.y); add_row(g, comp + 2, val.z); };
    auto wanted = [&](const VtxGeom &g) { return P.mesh_filter == nullptr || P.mesh_filter[g.mesh] != 0; };
    auto add_rgb = [&](float *tab, int id, const Vec3f &val) {
        if (val.x != 0.f && finite_(val.x)) atomicAdd(&tab[3 * id], val.x);
        if (val.y != 0.f && finite_(val.y)) atomicAdd(&tab[3 * id + 1], val.y);
        if (val.z != 0.f && finite_(val.z)) atomicAdd(&tab[3 * id + 2], val.z);
    };
    // Environment map (class 2): radiance along a world direction, and what the adjoint Lb = d (w.L) / d Le of one lookup gives -
    // the four texels of its footprint, the scale, from_world - and, returned, d (w.L) / d dir.  The Jacobian with respect to the
    // local direction comes from three forward evaluations of the lookup with unit tangents (atan2 / acos / bilinear: shade.h).
    auto env_radiance = [&](const Vec3f &dir) -> Vec3f {
        if constexpr (has_env(LDS)) return env_eval_direction<false, LDS>(S, T.env, dir);
        else return Vec3f(0.f);
    };
    // the direction an environment lookup uses for a ray from x that ended on the bounding cube at (slot, u, v): the hit point's local
    // incident direction taken back to the world through the cube face's frame (intersection.h / envmap.cpp:47-56) - bit for bit what
    // the forward pass looks up, so that both passes land in the same texel cell of the (piecewise bilinear) map
    auto env_dir_at = [&](int slot, float u, float v, const Vec3f &x) -> Vec3f {
        Hit h; h.slot = slot; h.u = u; h.v = v; h.t = 0.f;
        RayT<false> r; r.o = x; r.d = Vec3f(0.f, 0.f, 1.f);
        const Its<false> i1 = make_its<false, LDS, true>(S, h, r, true);
        return -to_world<false>(i1, i1.wi);
    };
    // the camera ray's own lookup (every background pixel makes one, and the samples of a wave share texels) is scattered by the whole
    // wave after the path work: one atomic per distinct texel and wave instead of one per lane
    bool pend_env = false;
    int pe_idx[4] = {0, 0, 0, 0};
    float pe_val[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto env_adjoint = [&](const Vec3f &dir, const Vec3f &Lb, bool defer = false) -> Vec3f {
        Vec3f dirb(0.f);
        if constexpr (has_env(LDS)) {
            const EnvDev &E = T.env;
            if (!(finite_(Lb.x) && finite_(Lb.y) && finite_(Lb.z)) || (Lb.x == 0.f && Lb.y == 0.f && Lb.z == 0.f)) return dirb;
            const Vec3f v = xform_dir(E.from_world, dir);
            float vb[3], uu = 0.f, ww = 0.f, rgb0[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const VecN<true> vd(Dual(v.x, j == 0 ? 1.f : 0.f), Dual(v.y, j == 1 ? 1.f : 0.f), Dual(v.z, j == 2 ? 1.f : 0.f));
                Dual u = env_atan2(vd.x, -vd.z) * Dual(env::kInvTwoPi), w = env_safe_acos(vd.y) * Dual(env::kInvPi);
                u = u - env_floor(u); w = w - env_floor(w);
                Dual rgb[3];
                env::bitmap_eval_fn<Dual>([&](int i, int c) { return Dual(E.radiance[3 * i + c], 0.f); }, E.width, E.height, u, w, rgb, uv_xf_d(E.xf, E.xf, false));
                vb[j] = E.scale * (Lb.x * rgb[0].d + Lb.y * rgb[1].d + Lb.z * rgb[2].d);
                if (j == 0) { uu = u.v; ww = w.v; rgb0[0] = rgb[0].v; rgb0[1] = rgb[1].v; rgb0[2] = rgb[2].v; }
            }
            const float lb[3] = {Lb.x, Lb.y, Lb.z}, dv[3] = {dir.x, dir.y, dir.z};
            if (P.g_env != nullptr) {
                int idx[4]; float wt[4];
                env::bitmap_footprint_env(E.width, E.height, uu, ww, idx, wt, env::UvXf<float>(E.xf));
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (lb[c] != 0.f) for (int k = 0; k < 4; ++k) {
                        if (P.env_lds) atomicAdd(&acc_env[3 * idx[k] + c], lb[c] * E.scale * wt[k]);
                        else if (defer) { pend_env = true; pe_idx[k] = idx[k]; pe_val[3 * k + c] = lb[c] * E.scale * wt[k]; }
                        else atomicAdd(&P.g_env[3ll * idx[k] + c], lb[c] * E.scale * wt[k]);
                    }
            }
            if (P.g_env_scale != nullptr) { const float sb = lb[0] * rgb0[0] + lb[1] * rgb0[1] + lb[2] * rgb0[2]; if (sb != 0.f && finite_(sb)) atomicAdd(&acc_cam[12], sb); }
            if (P.g_uv_xf != nullptr) { const float ob[3] = {lb[0] * E.scale, lb[1] * E.scale, lb[2] * E.scale}; env_xf_adjoint(E, uu, ww, ob, &acc_cam[28]); }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (!finite_(vb[r])) vb[r] = 0.f;
                if (P.g_env_xf != nullptr)
                    for (int c = 0; c < 3; ++c) { const float val = vb[r] * dv[c]; if (val != 0.f) atomicAdd(&acc_cam[16 + 4 * r + c], val); }
            }
            dirb = Vec3f(E.from_world.m[0] * vb[0] + E.from_world.m[4] * vb[1] + E.from_world.m[8] * vb[2],
                         E.from_world.m[1] * vb[0] + E.from_world.m[5] * vb[1] + E.from_world.m[9] * vb[2],
                         E.from_world.m[2] * vb[0] + E.from_world.m[6] * vb[1] + E.from_world.m[10] * vb[2]);
        }
        return dirb;
    };
    const int env_id = has_env(LDS) ? T.env_emitter : -1;
    // adjoint of x through dir = (z - x) / |z - x| (z fixed)
    auto dir_to_x = [&](const Vec3f &x, const Vec3f &z, const Vec3f &dirb) -> Vec3f {
        const Vec3f v = z - x;
        const float r = norm(v);
        if (!(r > 0.f)) return Vec3f(0.f);
        const Vec3f d = v / r;
        return (d * dot(d, dirb) - dirb) / r;
    };
    // the normal blend n0 (1 - u - v) + n1 u + n2 v behind a shading normal: its adjoint from the adjoint of ns
    auto blend_adjoint = [&](const VtxGeom &g, const Vec3f &nsb) { return (nsb - g.ns * dot(g.ns, nsb)) / g.nbl; };
    // adjoints of a vertex glued to its triangle: position, shading normal, geometric normal, area
    auto emit_glued = [&](const VtxGeom &g, const Vec3f &xb, const Vec3f &nsb, const Vec3f &ngb, float ab) {
        if (!wanted(g)) return;
        add_vec(g, 0, xb); add_vec(g, 3, xb * g.u); add_vec(g, 6, xb * g.v);
        Vec3f fnb = ngb;
        if (g.flat) fnb = fnb + nsb;
        else {
            const Vec3f nbb = blend_adjoint(g, nsb);
            add_vec(g, 9, nbb * (1.f - g.u - g.v)); add_vec(g, 12, nbb * g.u); add_vec(g, 15, nbb * g.v);
        }
        add_vec(g, 18, fnb);
        add_row(g, 21, ab);
    };

    long long q_next = 0, q_end = 0;
    bool exhausted = false;
    bool have = false;
    long long lane = 0;
    for (;;) {
        // ---- phase 1: find samples whose camera ray hits the scene (as in run_interior_adjoint)
        for (int round = 0; round < 8; ++round) {
            const unsigned long long need = __ballot(!have);
            if (__popcll(need) <= 6) break;
            if (q_next >= q_end && !exhausted) {
                unsigned long long base = 0;
                if (lane_id == 0) base = atomicAdd(P.counter, (unsigned long long) kFetchBatch);
                base = __shfl(base, 0);
                if ((long long) base >= P.n_local) exhausted = true;
                else { q_next = (long long) base; q_end = q_next + kFetchBatch < P.n_local ? q_next + kFetchBatch : P.n_local; }
            }
            if (q_next >= q_end) break;
            const int rank = __popcll(need & lt_mask);
            const long long item = q_next + rank;
            Vec3f o(0.f), d(0.f);
            bool cand = false;
            if (!have && item < q_end) {
                const long long chunk = (item >> 8) * P.shard_count + P.shard_rank;
                lane = P.begin + (chunk << 8) + (item & 255);
                if (lane < P.end) {
                    const long long k = T.spp > 1 ? lane / T.spp : lane;
                    const int pix = P.pix_ids ? P.pix_ids[k] : (int) k;
                    LaneRng rng;
                    rng.seed(P.seed + (P.pix_ids ? (unsigned long long) (long long) pix : (unsigned long long) lane), (unsigned long long) lane, P.skip);
                    const float bx = (float) (pix % T.width), by = (float) (pix / T.width);
                    const float jx = rng.next_1d(), jy = rng.next_1d();
                    const RayT<false> r = sample_primary_ray<false>(cam, (bx + jx) / (float) T.width, (by + jy) / (float) T.height);
                    o = r.o; d = r.d; cand = true;
                }
            }
            Hit h; h.slot = -1;
            if (cand) h = trace<LDS, false>(S, o, d);
            if (cand && h.slot >= 0) { have = true; vrec[0] = __int_as_float(h.slot); }
            const int n_need = __popcll(need);
            q_next += n_need < (int) (q_end - q_next) ? n_need : (q_end - q_next);
        }
        if (__ballot(have) == 0ull) { if (exhausted && q_next >= q_end) break; continue; }

        if (have) {
            const long long kpix = T.spp > 1 ? lane / T.spp : lane;
            const int pix = P.pix_ids ? P.pix_ids[kpix] : (int) kpix;
            LaneRng rng;
            rng.seed(P.seed + (P.pix_ids ? (unsigned long long) (long long) pix : (unsigned long long) lane), (unsigned long long) lane, P.skip);
            const float bx = (float) (pix % T.width), by = (float) (pix / T.width);
            const float jx = rng.next_1d(), jy = rng.next_1d();
            const float sx = (bx + jx) / (float) T.width, sy = (by + jy) / (float) T.height;
            const RayT<false> ray = sample_primary_ray<false>(cam, sx, sy);
            float wgt[3] = {P.w[3 * kpix] * inv_spp, P.w[3 * kpix + 1] * inv_spp, P.w[3 * kpix + 2] * inv_spp};

            // ------------------------------------------------------------ pass 1: forward, the primal arithmetic of D mode
            const int slot0 = __float_as_int(vrec[0]);
            float u0, v0, t0;
            {
                Vec3f a0, b0, c0;
                load_geom<false, LDS>(S, slot0, a0, b0, c0);
                ray_tri_uvt<float>(a0, b0, c0, ray.o, ray.d, u0, v0, t0);
            }
            vrec[kBlock] = u0; vrec[2 * kBlock] = v0;
            const Vec3f x0(fmaf(ray.d.x, t0, ray.o.x), fmaf(ray.d.y, t0, ray.o.y), fmaf(ray.d.z, t0, ray.o.z));     // the camera hit slides along the ray
            Hit hh; hh.slot = slot0; hh.u = u0; hh.v = v0; hh.t = t0;
            Its<false> its = make_its<false, LDS, true>(S, hh, ray, false);
            its.p = x0; its.t = t0;
            its.wi = to_local<false>(its, -ray.d);
            Vec3f thr(1.f), Lsum(0.f);
            const int e0 = mesh_emitter(S, its.mesh);
            // the first-hit integrators (field.cpp:49-121, collocated.cpp:24-55): the sample's value is a function of this vertex alone
            const bool fh = P.field >= 0, fh_bsdf = P.field == 6 || P.field >= 8;
            bool fh_ok = false;
            const bool le0 = !fh && !P.hide_emitters && e0 >= 0 && (e0 == env_id || its.wi.z > 0.f);
            const Vec3f dir0 = -to_world<false>(its, its.wi);          // (= ray.d through the frame of the first hit, as eval_Le rebuilds it)
            if (le0) { if (e0 == env_id) Lsum = env_radiance(dir0); else { const float4 a = S.ld(T.emit_off + 2 * e0); Lsum = Vec3f(a.x, a.y, a.z); } }
            if (fh) {
                if constexpr (has_mat(LDS)) {
                    fh_ok = !(T.env_emitter >= 0 && P.field != 8) || mesh_bsdf(S, its.mesh) >= 0;
                    if (P.field_object >= 0) fh_ok = fh_ok && its.mesh == P.field_object;
                    Lsum = first_hit_value<false, LDS>(S, its);
                }
            }
            int nb = 0;                                                   // bounces recorded
            bool active = true;
            auto nonzero = [](const Vec3f &v) { return v.x != 0.f || v.y != 0.f || v.z != 0.f; };
            for (int depth = 0; depth < D && active; ++depth) {
                float *br = brec + 11 * depth * kBlock;
                br[8 * kBlock] = thr.x; br[9 * kBlock] = thr.y; br[10 * kBlock] = thr.z;
                nb = depth + 1;
                int flags = 0;              // 1 next-event term, 2 BSDF term, 8 the next vertex is a visible emitter, 16 the light sample is on the environment map
                // the vertex's two rays are drawn first (sampler order of the forward pass: two numbers, then three) and traced by ONE trace2 call (adjoint.h, round 4)
                // (DirectIntegrator(1) neither draws nor uses the emitter sample, DirectIntegrator(0) stops after it: direct.cpp:34-132)
                const float sn1 = P.mis != 1 ? rng.next_1d() : 0.f, sn2 = P.mis != 1 ? rng.next_1d() : 0.f;
                const bool do_nee = P.mis != 1 && mesh_emitter(S, its.mesh) < 0;
                PositionSample<false> ps;
                ps.p = Vec3f(0.f); ps.n = Vec3f(0.f); ps.J = 1.f; ps.pdf = 1.f; ps.slot = -1; ps.ba = ps.bb = 0.f;
                Vec3f wod(0.f);
                float dist_sqr = 0.f, dist = 0.f;
                if (do_nee) {
                    ps = sample_emitter_position<false, LDS>(S, its.p, sn1, sn2);
                    wod = ps.p - its.p;
                    dist_sqr = squared_norm(wod); dist = safe_sqrt(dist_sqr);
                    wod = wod / dist;
                }
                const bool do_bsdf = P.mis != 0;
                const float sb0 = do_bsdf ? rng.next_1d() : 0.f, sb1 = do_bsdf ? rng.next_1d() : 0.f, sb2 = do_bsdf ? rng.next_1d() : 0.f;
                BSDFSample bs = bsdf_sample<false, LDS>(S, its, sb0, sb1, sb2, do_bsdf);
                if (!do_bsdf) bs.valid = false;
                RayT<false> curr; curr.o = its.p; curr.d = to_world<false>(its, bs.wo);
                Hit h1, hx;
                trace2<LDS, false>(S, its.p, wod, do_nee, curr.o, curr.d, bs.valid, h1, hx, do_nee ? (dist - kShadowEpsilon) * 0.9999f : -__builtin_inff());     // (any occluder settles the shadow test)
                {   // next-event estimation (path.cpp:47-83): L += thr . F(wi, w) . Le . g cN, g = |nz.w| / r^2 . A / detach(A), cN = mis / pdf
                    if (do_nee) {
                        if (h1.slot >= 0) {
                            RayT<false> ray1; ray1.o = its.p; ray1.d = wod;
                            const Its<false> its1 = make_its<false, LDS, false>(S, h1, ray1, true);
                            const int eh = mesh_emitter(S, its1.mesh);
                            if (its1.t > dist - kShadowEpsilon && eh >= 0) {
                                const float G = fabsf(dot(its1.n, -wod)) / dist_sqr;
                                const Vec3f wo_l = to_local<false>(its, wod);
                                const Vec3f F = bsdf_eval<false, LDS>(S, its, wo_l, true);
                                const float pdf1 = bsdf_pdf<false, LDS>(S, its, wo_l, true) * G;
                                Vec3f Le(0.f);
                                if (eh == env_id) Le = env_radiance(env_dir_at(h1.slot, h1.u, h1.v, its.p));
                                else if (its1.wi.z > 0.f) { const float4 ea = S.ld(T.emit_off + 2 * eh); Le = Vec3f(ea.x, ea.y, ea.z); }
                                // (path.cpp:76-82 adds the product whenever pdf1 != 0: a non-finite factor beside a zero one poisons the
                                // sample there, and integrator.cpp:126 then drops it - the same must happen here)
                                const float cN = (P.mis == 0 ? 1.f : mis_weight(ps.pdf, pdf1)) / ps.pdf;
                                if (pdf1 != 0.f) Lsum = Lsum + thr * F * Le * (G * cN);
                                if (pdf1 != 0.f && nonzero(Le) && nonzero(F)) {
                                    if (eh == env_id) { br[0] = h1.u; br[kBlock] = h1.v; br[2 * kBlock] = 0.f; flags |= 16; }
                                    else { br[0] = __int_as_float(ps.slot); br[kBlock] = ps.ba; br[2 * kBlock] = ps.bb; }
                                    br[3 * kBlock] = __int_as_float(h1.slot);
                                    br[4 * kBlock] = cN;
                                    flags |= 1;
                                }
                            }
                        }
                    }
                }
                {   // BSDF sampling (path.cpp:86-123): thr' = thr . F(wi, w) . g cf, cf = 1 / pdf0
                    active = bs.valid && hx.slot >= 0;
                    if (active) {
                        const Its<false> itx = make_its<false, LDS, true>(S, hx, curr, true);
                        float *vr = vrec + 3 * (depth + 1) * kBlock;
                        vr[0] = __int_as_float(hx.slot); vr[kBlock] = hx.u; vr[2 * kBlock] = hx.v;
                        const Vec3f wo = (itx.p - its.p) / itx.t;
                        const float G = fabsf(dot(itx.n, -wo)) / sqr(itx.t);
                        const float pdf0 = bs.pdf * G;
                        const Vec3f F = (itx.t < kEpsilon) ? Vec3f(0.f) : bsdf_eval<false, LDS>(S, its, to_local<false>(its, wo), true);
                        const float cf = 1.f / pdf0;
                        const float w2 = P.mis == 1 ? 1.f : mis_weight(pdf0, emitter_position_pdf<false, LDS>(S, its.p, itx));
                        thr = thr * F * (G * cf);
                        const int ex = mesh_emitter(S, itx.mesh);
                        Vec3f Le(0.f);
                        if (ex >= 0 && ex == env_id) Le = env_radiance(env_dir_at(hx.slot, hx.u, hx.v, its.p));
                        else if (ex >= 0 && itx.wi.z > 0.f) { const float4 ea = S.ld(T.emit_off + 2 * ex); Le = Vec3f(ea.x, ea.y, ea.z); }
                        Lsum = Lsum + Le * thr * w2;         // (always, as path.cpp:118 does: a zero-pdf sample makes thr and w2 non-finite and the sample is dropped)
                        if (nonzero(Le)) flags |= 8;
                        br[5 * kBlock] = cf; br[6 * kBlock] = w2;
                        flags |= 2;
                        its = itx;
                    }
                }
                br[7 * kBlock] = __int_as_float(flags);
            }
            {   // integrator.cpp:126: a non-finite channel contributes nothing
                const float pv[3] = {Lsum.x, Lsum.y, Lsum.z};
#pragma unroll
                for (int c = 0; c < 3; ++c) if (!finite_(pv[c])) wgt[c] = 0.f;
            }
            const Vec3f W(wgt[0], wgt[1], wgt[2]);

            // ------------------------------------------------------------ pass 2: back over the bounces
            if ((W.x != 0.f || W.y != 0.f || W.z != 0.f) && (!fh || fh_ok)) {
                Vec3f cam_dirb(0.f);                    // adjoint of the camera ray's direction from an environment lookup along it
                if (le0 && e0 == env_id) cam_dirb = env_adjoint(dir0, W, true);
                else if (le0 && !P.skip_emitter) add_rgb(acc_emit, e0, W);          // the emitter seen by the camera
                Vec3f Abar(0.f);                       // d (w.L) / d thr_{k+1} from the bounces behind k
                Vec3f xb_next(0.f), nsb_next(0.f);     // what bounce k+1 gave vertex k+1 as ITS shading point
                Vec3f pb_a(0.f), pb_b(0.f);            // what the incident directions of bounces k+1 / k+2 gave the vertex before them
                Vec3f xb0(0.f), nsb0(0.f);             // the camera hit's totals
                Vec3f dcam_b(0.f);                     // adjoint of the camera ray's direction as the incident direction of bounce 0
                float ub_tex = 0.f, vb_tex = 0.f;      // adjoints of the camera hit's barycentrics through its texture coordinates
                float tb_field = 0.f;                  // adjoint of the camera hit's distance as a first-hit integrator's own argument
                // (a first-hit integrator has no bounces: its BSDF forms run the vertex part of iteration 0, the others only the camera-hit block)
                for (int k = fh ? (fh_bsdf ? 0 : -1) : nb - 1; k >= 0; --k) {
                    const float *br = brec + 11 * k * kBlock;
                    const int flags = fh ? 0 : __float_as_int(br[7 * kBlock]);
                    const Vec3f thr_k = fh ? Vec3f(1.f) : Vec3f(br[8 * kBlock], br[9 * kBlock], br[10 * kBlock]);
                    const float *vr = vrec + 3 * k * kBlock;
                    VtxGeom gk = load_vertex(S, __float_as_int(vr[0]), vr[kBlock], vr[2 * kBlock]);
                    if (k == 0) gk.x = x0;
                    const int bid = mesh_bsdf(S, gk.mesh);
                    // the incident direction: the camera ray at the first vertex, (x_{k-1} - x_k) / r behind it (scene.cpp:686-690)
                    Vec3f wi_w = -ray.d;
                    float rin = 1.f;
                    if (k > 0) {
                        const float *vp = vrec + 3 * (k - 1) * kBlock;
                        VtxGeom gp = load_vertex(S, __float_as_int(vp[0]), vp[kBlock], vp[2 * kBlock]);
                        if (k == 1) gp.x = x0;
                        const Vec3f vin = gp.x - gk.x;
                        rin = norm(vin);
                        wi_w = vin / rin;
                    }
                    // the vertex's shading frame as the forward pass builds it (an anisotropic lobe sees the tangents)
                    Vec3f fs, ft;
                    float fu0x, fu0y, fu1x, fu1y;
                    {
                        const int wsh = T.shade_off + 6 * __float_as_int(vr[0]);
                        const float4 s4 = S.ld(wsh + 4), s5 = S.ld(wsh + 5);
                        fu0x = s4.z - s4.x; fu0y = s4.w - s4.y; fu1x = s5.x - s4.x; fu1y = s5.y - s4.y;
                    }
                    vertex_frame<float>(gk.ns, gk.e1, gk.e2, fu0x, fu0y, fu1x, fu1y, fs, ft);
                    const Vec3f wi_l(dot(wi_w, fs), dot(wi_w, ft), dot(wi_w, gk.ns));
                    Vec3f fsb(0.f), ftb(0.f);                    // adjoints of the two tangent vectors (anisotropic lobes only)
                    bool aniso = false;
                    if (bid >= 0) {
                        const int bfl = __float_as_int(S.ld(T.bsdf_off + 2 * bid).w);
                        if (bfl & (8 | 16)) { const MatDev md = T.mat[bid]; aniso = md.alpha_u != md.alpha_v; }
                        if (bfl & 256) aniso = true;             // a NormalMap sees the frame through dp_du (a world-space vector beside the local map normal, normalmap.cpp:62)
                    }
                    // its.dp_du of the vertex (make_its; scene.cpp:724-766) and its adjoint: NormalMaps only
                    Vec3f dpdu_k(0.f), dpdu_b(0.f);
                    const float det_uv = fma_(fu0x, fu1y, -(fu0y * fu1x));
                    const float inv_det_uv = det_uv != 0.f ? 1.f / det_uv : 0.f;
                    if (det_uv != 0.f) dpdu_k = (gk.e1 * fu1y - gk.e2 * fu0y) * inv_det_uv;
                    Vec3f xb(0.f), nsb(0.f), A_k(0.f), wib_w(0.f);
                    // F and, for its adjoint Fb, the adjoints of the outgoing direction (returned), of the incident direction and of ns
                    // (accumulated), and of the BSDF's parameters (accumulated in LDS)
                    // texture coordinates of the vertex (scene.cpp:715/779): uv0 + (uv1 - uv0) u + (uv2 - uv0) v
                    float tu = 0.f, tv = 0.f, du0x = 0.f, du0y = 0.f, du1x = 0.f, du1y = 0.f, uvb[2] = {0.f, 0.f}, bcb[2] = {0.f, 0.f};
                    const float bary_u = k == 0 ? u0 : vr[kBlock], bary_v = k == 0 ? v0 : vr[2 * kBlock];
                    if (T.tex != nullptr) {
                        const int wsh = T.shade_off + 6 * __float_as_int(vr[0]);
                        const float4 s4 = S.ld(wsh + 4), s5 = S.ld(wsh + 5);
                        du0x = s4.z - s4.x; du0y = s4.w - s4.y; du1x = s5.x - s4.x; du1y = s5.y - s4.y;
                        const float bu = k == 0 ? u0 : vr[kBlock], bv = k == 0 ? v0 : vr[2 * kBlock];
                        tu = fmaf(du0x, bu, fmaf(du1x, bv, s4.x)); tv = fmaf(du0y, bu, fmaf(du1y, bv, s4.y));
                    }
                    auto bsdf_primal = [&](const Vec3f &w) -> Vec3f {
                        const Vec3f wo_l(dot(w, fs), dot(w, ft), dot(w, gk.ns));
                        return bsdf_value_and_adjoint<LDS>(S, bid, wi_l, wo_l, tu, tv, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, __float_as_int(vr[0]), bary_u, bary_v, nullptr, &dpdu_k, nullptr);
                    };
                    auto bsdf_back = [&](const Vec3f &w, const Vec3f &Fb) -> Vec3f {
                        const Vec3f wo_l(dot(w, fs), dot(w, ft), dot(w, gk.ns));
                        float wib_l[3], wob_l[3], dpb_l[3] = {0.f, 0.f, 0.f};
                        bsdf_value_and_adjoint<LDS>(S, bid, wi_l, wo_l, tu, tv, &Fb, wib_l, wob_l, P.skip_bsdf ? nullptr : acc_bsdf, acc_mat, P.g_tex, (k == 0 && T.tex != nullptr) ? uvb : nullptr,
                                                    __float_as_int(vr[0]), bary_u, bary_v, k == 0 ? bcb : nullptr, &dpdu_k, dpb_l);
                        dpdu_b = dpdu_b + Vec3f(dpb_l[0], dpb_l[1], dpb_l[2]);
                        wib_w = wib_w + fs * wib_l[0] + ft * wib_l[1] + gk.ns * wib_l[2];
                        if (aniso) {
                            // local components = dot products with the frame vectors: their adjoints (the tangents' go through vertex_frame below)
                            fsb = fsb + wi_w * wib_l[0] + w * wob_l[0];
                            ftb = ftb + wi_w * wib_l[1] + w * wob_l[1];
                            nsb = nsb + wi_w * wib_l[2] + w * wob_l[2];
                        } else {
                            // an isotropic lobe does not see the tangents: turning ns (the tangents follow) changes the local components of both
                            // directions by d u_l = (-e_x u_z, -e_y u_z, e_x u_x + e_y u_y)
                            const float ex = -wi_l.z * wib_l[0] + wi_l.x * wib_l[2] - wo_l.z * wob_l[0] + wo_l.x * wob_l[2];
                            const float ey = -wi_l.z * wib_l[1] + wi_l.y * wib_l[2] - wo_l.z * wob_l[1] + wo_l.y * wob_l[2];
                            nsb = nsb + fs * ex + ft * ey;
                        }
                        return fs * wob_l[0] + ft * wob_l[1] + gk.ns * wob_l[2];
                    };
                    GeoGrad gg;
                    if (flags & 2) {
                        // thr_{k+1} = thr_k . F_f . g_f cf,  L += thr_{k+1} Le_{k+1} w2
                        const float *vn = vrec + 3 * (k + 1) * kBlock;
                        const VtxGeom gz = load_vertex(S, __float_as_int(vn[0]), vn[kBlock], vn[2 * kBlock]);
                        const float cf = br[5 * kBlock], w2 = br[6 * kBlock];
                        const float gf = geo_eval(gk.x, gz.x, gz.fn, gz.area, gg) * cf;
                        const Vec3f w = normalize(gz.x - gk.x);
                        const Vec3f Ff = bsdf_primal(w);
                        Vec3f At = Abar;                                             // total adjoint of thr_{k+1}
                        if (flags & 8) {
                            const int ex = mesh_emitter(S, gz.mesh);
                            if (ex == env_id) {
                                const Vec3f dir = env_dir_at(__float_as_int(vn[0]), vn[kBlock], vn[2 * kBlock], gk.x);
                                At = At + W * env_radiance(dir) * w2;
                                xb = xb + dir_to_x(gk.x, gz.x, env_adjoint(dir, W * thr_k * Ff * (gf * w2)));
                            } else {
                                const float4 ea = S.ld(T.emit_off + 2 * ex);
                                At = At + W * Vec3f(ea.x, ea.y, ea.z) * w2;
                                if (!P.skip_emitter) add_rgb(acc_emit, ex, W * thr_k * Ff * (gf * w2));
                            }
                        }
                        const Vec3f tf = thr_k * Ff * At;
                        const float gb = cf * (tf.x + tf.y + tf.z);
                        A_k = A_k + Ff * At * gf;
                        const Vec3f wob = bsdf_back(w, thr_k * At * gf);
                        const Vec3f wx = dir_to_x(gk.x, gz.x, wob);                  // through w = (z - x) / r
                        xb = xb + gg.dx * gb + wx; 
                        // vertex k+1 is complete: end point of this segment + shading point of bounce k+1 + origin of bounce k+2's incident direction
                        emit_glued(gz, xb_next + pb_b + gg.dz * gb - wx, nsb_next, gg.dnz * gb, gg.dA * gb);
                    }
                    if (flags & 1) {
                        // L += thr_k . F_N . Le . g_N cN
                        const bool on_env = (flags & 16) != 0;
                        const VtxGeom gh = load_vertex(S, __float_as_int(br[3 * kBlock]), on_env ? br[0] : 0.f, on_env ? br[kBlock] : 0.f);
                        VtxGeom gy = gh;
                        Vec3f y;
                        float area_y = 1.f;
                        if (on_env) y = gh.x;                  // the shadow ray's hit on the cube
                        else { gy = load_vertex(S, __float_as_int(br[0]), br[kBlock], br[2 * kBlock]); y = gy.x; area_y = gy.area; }
                        const float cN = br[4 * kBlock];
                        const float gN = geo_eval(gk.x, y, gh.fn, area_y, gg) * cN;
                        const Vec3f w = normalize(y - gk.x);
                        const Vec3f FN = bsdf_primal(w);
                        Vec3f Le;
                        const int eh = mesh_emitter(S, gh.mesh);
                        const Vec3f edir = on_env ? env_dir_at(__float_as_int(br[3 * kBlock]), br[0], br[kBlock], gk.x) : w;
                        if (on_env) Le = env_radiance(edir); else { const float4 ea = S.ld(T.emit_off + 2 * eh); Le = Vec3f(ea.x, ea.y, ea.z); }
                        const Vec3f al = W * thr_k * FN * Le;
                        const float gb = cN * (al.x + al.y + al.z);
                        A_k = A_k + W * FN * Le * gN;
                        if (on_env) xb = xb + dir_to_x(gk.x, y, env_adjoint(edir, W * thr_k * FN * gN));
                        else if (!P.skip_emitter) add_rgb(acc_emit, eh, W * thr_k * FN * gN);
                        const Vec3f wob = bsdf_back(w, W * thr_k * Le * gN);
                        const Vec3f wx = dir_to_x(gk.x, y, wob);
                        xb = xb + gg.dx * gb + wx;
                        if (!on_env) {
                            emit_glued(gy, gg.dz * gb - wx, Vec3f(0.f), Vec3f(0.f), gg.dA * gb);     // the light sample: position and area of ITS triangle
                            emit_glued(gh, Vec3f(0.f), Vec3f(0.f), gg.dnz * gb, 0.f);                // the normal of the triangle the shadow ray hit
                        }
                    }
                    if (fh && fh_bsdf && bid >= 0) {
                        // field 6: F(wi, wi); CollocatedIntegrator: F(wi, wi) . intensity / t^2 - both directions are the camera ray's
                        float scale = 1.f;
                        if (P.field >= 8) {
                            const Vec3f wf = W * bsdf_primal(wi_w);
                            scale = P.intensity / sqr(t0);
                            tb_field = -2.f * (wf.x + wf.y + wf.z) * scale / t0;
                        }
                        dcam_b = dcam_b - bsdf_back(wi_w, W * scale);
                    }
                    // the tangents are functions of ns and, with a uv parameterisation, of the triangle's edges: J^T (fsb, ftb) by forward
                    // evaluations of vertex_frame with unit tangents
                    Vec3f e1b_f(0.f), e2b_f(0.f);
                    if (fsb.x != 0.f || fsb.y != 0.f || fsb.z != 0.f || ftb.x != 0.f || ftb.y != 0.f || ftb.z != 0.f) {
                        const bool uvf = fma_(fu0x, fu1y, -(fu0y * fu1x)) != 0.f;
                        float gj[9];
                        for (int j = 0; j < 9; ++j) gj[j] = 0.f;
#pragma unroll
                        for (int j = 0; j < 9; ++j) {
                            if (j >= 3 && !uvf) continue;
                            const Vec3d nsD(Dual(gk.ns.x, j == 0 ? 1.f : 0.f), Dual(gk.ns.y, j == 1 ? 1.f : 0.f), Dual(gk.ns.z, j == 2 ? 1.f : 0.f));
                            const Vec3d e1D(Dual(gk.e1.x, j == 3 ? 1.f : 0.f), Dual(gk.e1.y, j == 4 ? 1.f : 0.f), Dual(gk.e1.z, j == 5 ? 1.f : 0.f));
                            const Vec3d e2D(Dual(gk.e2.x, j == 6 ? 1.f : 0.f), Dual(gk.e2.y, j == 7 ? 1.f : 0.f), Dual(gk.e2.z, j == 8 ? 1.f : 0.f));
                            Vec3d fsD, ftD;
                            vertex_frame<Dual>(nsD, e1D, e2D, fu0x, fu0y, fu1x, fu1y, fsD, ftD);
                            const float g = fsb.x * fsD.x.d + fsb.y * fsD.y.d + fsb.z * fsD.z.d + ftb.x * ftD.x.d + ftb.y * ftD.y.d + ftb.z * ftD.z.d;
                            gj[j] = finite_(g) ? g : 0.f;
                        }
                        nsb = nsb + Vec3f(gj[0], gj[1], gj[2]);
                        if (uvf) { e1b_f = Vec3f(gj[3], gj[4], gj[5]); e2b_f = Vec3f(gj[6], gj[7], gj[8]); }
                        if (wanted(gk) && uvf) { add_vec(gk, 3, e1b_f); add_vec(gk, 6, e2b_f); }
                    }
                    // dp_du = (e1 dv1 - e2 dv0) / det: a NormalMap's adjoint of it goes to the triangle's edges
                    if (det_uv != 0.f && wanted(gk) && (dpdu_b.x != 0.f || dpdu_b.y != 0.f || dpdu_b.z != 0.f)) {
                        add_vec(gk, 3, dpdu_b * (fu1y * inv_det_uv));
                        add_vec(gk, 6, dpdu_b * (-fu0y * inv_det_uv));
                    }
                    // the incident direction's adjoint: to the camera ray at the first vertex, else to x_{k-1} and x_k
                    Vec3f pb(0.f);
                    if (k == 0) dcam_b = dcam_b - wib_w;
                    else { pb = (wib_w - wi_w * dot(wi_w, wib_w)) / rin; xb = xb - pb; }
                    xb_next = xb; nsb_next = nsb;
                    Abar = A_k;
                    pb_b = pb_a; pb_a = pb;
                    if (k == 0) { xb0 = xb + pb_b; nsb0 = nsb; ub_tex = uvb[0] * du0x + uvb[1] * du0y + bcb[0]; vb_tex = uvb[0] * du1x + uvb[1] * du1y + bcb[1]; }
                }
                // the camera hit: x_0 = o + t d and ns_0 = normalize(blend(u, v)) with (u, v, t) = Moeller-Trumbore(p0, e1, e2; o, d)
                if (nb > 0 || le0 || fh) {
                    const VtxGeom g0 = load_vertex(S, slot0, u0, v0);
                    if (fh) {
                        // field.cpp:60-106: 1 position, 2 depth (three equal channels), 3 geometric normal, 4 shading normal, 5 texture coordinates
                        if (P.field == 1) xb0 = xb0 + W;
                        else if (P.field == 2) tb_field = W.x + W.y + W.z;
                        else if (P.field == 3) { if (wanted(g0)) add_vec(g0, 18, W); }
                        else if (P.field == 4) nsb0 = nsb0 + W;
                        else if (P.field == 5) {
                            const float4 s4 = S.ld(T.shade_off + 6 * slot0 + 4), s5 = S.ld(T.shade_off + 6 * slot0 + 5);
                            ub_tex += W.x * (s4.z - s4.x) + W.y * (s4.w - s4.y);
                            vb_tex += W.x * (s5.x - s4.x) + W.y * (s5.y - s4.y);
                        }
                    }
                    float ub = ub_tex, vb = vb_tex;
                    const float tb = dot(ray.d, xb0) + tb_field;
                    Vec3f ob = xb0, db = xb0 * t0 + dcam_b;
                    const bool want0 = wanted(g0);
                    if (g0.flat) { if (want0) add_vec(g0, 18, nsb0); }
                    else {
                        const Vec3f nbb = blend_adjoint(g0, nsb0);
                        ub += dot(g0.n1 - g0.n0, nbb); vb += dot(g0.n2 - g0.n0, nbb);
                        if (want0) { add_vec(g0, 9, nbb * (1.f - u0 - v0)); add_vec(g0, 12, nbb * u0); add_vec(g0, 15, nbb * v0); }
                    }
                    Vec3f p0b, e1b, e2b, ob2, db2;
                    mt_adjoint(g0.p0, g0.e1, g0.e2, ray.o, ray.d, ub, vb, tb, p0b, e1b, e2b, ob2, db2);
                    if (want0) { add_vec(g0, 0, p0b); add_vec(g0, 3, e1b); add_vec(g0, 6, e2b); }
                    if (P.g_cam != nullptr) {
                        // o = to_world . (o_cam, 1), d = to_world . (d_cam, 0)  (primary_ray_pose_tangent)
                        ob = ob + ob2; db = db + db2 + cam_dirb;
                        const Vec3f pc = xform_pos(cam.sample_to_camera, Vec3f(sx, sy, 0.f));
                        const Vec3f o_cam = cam.ortho ? pc : Vec3f(0.f), d_cam = cam.ortho ? Vec3f(0.f, 0.f, 1.f) : normalize(pc);
                        const float oc[4] = {o_cam.x, o_cam.y, o_cam.z, 1.f}, dc[4] = {d_cam.x, d_cam.y, d_cam.z, 0.f};
                        const float obv[3] = {ob.x, ob.y, ob.z}, dbv[3] = {db.x, db.y, db.z};
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float val = obv[r] * oc[c] + dbv[r] * dc[c];
                                if (val != 0.f && finite_(val)) atomicAdd(&acc_cam[4 * r + c], val);
                            }
                    }
                }
            }
            have = false;
        }
        // every lane of the wave is here: the pending camera lookups, one atomic per distinct texel
        if (__ballot(pend_env) != 0ull) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int key = pend_env ? pe_idx[k] : -1;
                for (int it = 0; it < 8; ++it) {
                    const unsigned long long m = __ballot(key >= 0);
                    if (m == 0ull) break;
                    const int leader = (int) __builtin_ctzll(m);
                    const int lk = __shfl(key, leader);
                    const bool same = key == lk;
                    float s0 = same ? pe_val[3 * k] : 0.f, s1 = same ? pe_val[3 * k + 1] : 0.f, s2 = same ? pe_val[3 * k + 2] : 0.f;
                    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
                    if (lane_id == leader) {
                        if (s0 != 0.f) atomicAdd(&P.g_env[3ll * lk], s0);
                        if (s1 != 0.f) atomicAdd(&P.g_env[3ll * lk + 1], s1);
                        if (s2 != 0.f) atomicAdd(&P.g_env[3ll * lk + 2], s2);
                    }
                    if (same) key = -1;
                }
                if (key >= 0) {           // more than eight distinct texels in the wave: the rest go one by one
                    if (pe_val[3 * k] != 0.f) atomicAdd(&P.g_env[3ll * key], pe_val[3 * k]);
                    if (pe_val[3 * k + 1] != 0.f) atomicAdd(&P.g_env[3ll * key + 1], pe_val[3 * k + 1]);
                    if (pe_val[3 * k + 2] != 0.f) atomicAdd(&P.g_env[3ll * key + 2], pe_val[3 * k + 2]);
                }
            }
            pend_env = false;
#pragma unroll
            for (int q = 0; q < 12; ++q) pe_val[q] = 0.f;
        }
    }
    __syncthreads();
    if (P.g_cam != nullptr && threadIdx.x < 12 && acc_cam[threadIdx.x] != 0.f) atomicAdd(&P.g_cam[threadIdx.x], acc_cam[threadIdx.x]);
    if (P.g_env_scale != nullptr && threadIdx.x == 12 && acc_cam[12] != 0.f) atomicAdd(P.g_env_scale, acc_cam[12]);
    if (P.g_env_xf != nullptr && threadIdx.x >= 16 && threadIdx.x < 27 && acc_cam[threadIdx.x] != 0.f) atomicAdd(&P.g_env_xf[threadIdx.x - 16], acc_cam[threadIdx.x]);
    for (int i = threadIdx.x; i < P.n_hot * 22; i += kBlock) if (acc[i] != 0.f) atomicAdd(&P.g_tri[P.hot_inv[i / 22] * 22 + i % 22], acc[i]);
    for (int i = threadIdx.x; i < T.n_bsdfs * 3; i += kBlock) if (acc_bsdf[i] != 0.f) atomicAdd(&P.g_bsdf[i], acc_bsdf[i]);
    if (P.g_mat != nullptr)
        for (int i = threadIdx.x; i < T.n_bsdfs * kMatOut; i += kBlock) { const float v = acc_mat[(i / kMatOut) * mrow + i % kMatOut]; if (v != 0.f) atomicAdd(&P.g_mat[i], v); }
    if (P.g_uv_xf != nullptr) {          // uv transforms: three bitmaps per BSDF, then the environment map's
        for (int i = threadIdx.x; i < T.n_bsdfs * 12; i += kBlock) { const float v = acc_mat[(i / 12) * mrow + kMatOut + i % 12]; if (v != 0.f) atomicAdd(&P.g_uv_xf[i], v); }
        if (threadIdx.x >= 28 && threadIdx.x < 32 && acc_cam[threadIdx.x] != 0.f) atomicAdd(&P.g_uv_xf[12 * T.n_bsdfs + threadIdx.x - 28], acc_cam[threadIdx.x]);
    }
    for (int i = threadIdx.x; i < T.n_emitters * 3; i += kBlock) if (acc_emit[i] != 0.f) atomicAdd(&P.g_emitter[i], acc_emit[i]);
    for (int i = threadIdx.x; i < P.env_lds; i += kBlock) if (acc_env[i] != 0.f) atomicAdd(&P.g_env[i], acc_env[i]);
}



} // namespace psdr
