"""Reverse mode (psdr_hip_render_d_bwd) against forward mode on the GPU: the adjoint dot-product test
        < w , J v >  ==  < J^T w , v >
where v is the forward tangent of the configured snapshot (triangle rows, edge tables, colours) induced by one
scene parameter, J v = d(image) from psdr_hip_render_d_fwd and J^T w the buffers of psdr_hip_render_d_bwd.
The forward side is itself pinned against the CPU oracle (test_gpu_parity.py)."""
import ctypes as C

import os

import numpy as np
import pytest

import product
import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    import __graft_entry__
    __graft_entry__.build()
    import psdr_jit_amd
    from psdr_jit_amd import cabi
    return torch, psdr_jit_amd, cabi


def _bwd(torch, cabi, sc, w, depth, seeds, terms, n_tris, n_bsdfs, n_emitters, n_sec, n_prim):
    dev = "cuda"
    g_tri = torch.zeros((n_tris, 22), dtype=torch.float32, device=dev)
    g_bsdf = torch.zeros((max(1, n_bsdfs), 3), dtype=torch.float32, device=dev)
    g_em = torch.zeros((max(1, n_emitters), 3), dtype=torch.float32, device=dev)
    g_sec = torch.zeros((max(1, n_sec), 6), dtype=torch.float32, device=dev)
    g_prim = torch.zeros((max(1, n_prim), 4), dtype=torch.float32, device=dev)
    g = cabi.Grads(g_tri.data_ptr(), g_bsdf.data_ptr(), g_em.data_ptr(), g_sec.data_ptr(), g_prim.data_ptr())
    a = cabi.make_args(max_depth=depth, seeds=seeds, terms=terms)
    cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
    torch.cuda.synchronize()
    return [t.cpu().numpy().astype(np.float64) for t in (g_tri, g_bsdf, g_em, g_sec, g_prim)]


@pytest.mark.parametrize("param", ["light_x", "box_x", "albedo", "radiance"])
@pytest.mark.parametrize("terms", [1, 2, 4, 7])
def test_adjoint_dot_product(env, param, terms):
    torch, psdr, cabi = env
    res, spp, depth = 48, 8, 2
    spec = scenes.cbox_scene(res, res, spp, spp, spp, param=param)
    sc = product.build_scene(spec)
    snap = sc._snapshot()
    cam = sc.param_map["Sensor[0]"]
    d_tri = np.asarray(snap["d_triangles"], np.float64)
    d_sec = np.asarray(snap["d_sec_edges"], np.float64)[:, :6]
    d_prim = np.asarray(cam._primary_edges(True), np.float64)[:, :4]
    d_bsdf = np.array([b.d_reflectance for b in spec.bsdfs], np.float64)
    d_em = np.array([e.d_radiance for e in spec.emitters], np.float64)
    n = res * res
    seeds = (7, 8, 9)
    buf = torch.empty((2, n, 3), dtype=torch.float32, device="cuda")
    a = cabi.make_args(max_depth=depth, seeds=seeds, terms=terms)
    cabi.check(cabi.lib().psdr_hip_render_d_fwd(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None))
    gen = torch.Generator(device="cpu").manual_seed(3)
    w = (torch.rand((n, 3), generator=gen) + 0.5).to("cuda")
    lhs = float((buf[1].double() * w.double()).sum())
    g_tri, g_bsdf, g_em, g_sec, g_prim = _bwd(torch, cabi, sc, w, depth, seeds, terms, d_tri.shape[0], len(spec.bsdfs), len(spec.emitters),
                                              d_sec.shape[0], d_prim.shape[0])
    rhs = (g_tri * d_tri).sum() + (g_bsdf[:len(spec.bsdfs)] * d_bsdf).sum() + (g_em[:len(spec.emitters)] * d_em).sum()
    rhs += (g_sec[:d_sec.shape[0]] * d_sec).sum() + (g_prim[:d_prim.shape[0]] * d_prim).sum()
    scale = float((buf[1].double().abs() * w.double()).sum()) + 1e-12
    assert abs(lhs - rhs) <= 2e-4 * scale, (param, terms, lhs, rhs, scale)
    if terms & 1 and param in ("albedo", "radiance"):
        assert abs(lhs) > 1e-6          # the test is not vacuous


_FIELD_NAMES = {0: "silhouette", 1: "position", 2: "depth", 3: "geoNormal", 4: "shNormal", 5: "uv", 6: "bsdf", 7: "segmentation", 8: "collocated"}


def _dot_product_case(env, spec, depth, terms, seeds=(7, 8, 9), with_camera=False, with_mat=False, direct_mis=-1, oracle=None, field=-1, field_object=-1, intensity=1.0):
    """<w, J v> against <J^T w, v> for the forward tangent the spec carries; J v from HIP forward mode or, oracle = the oracle module, from the CPU oracle's
    forward mode (then no HIP forward kernel takes part in the check)"""
    torch, psdr, cabi = env
    sc = product.build_scene(spec)
    snap = sc._snapshot()
    cam = sc.param_map["Sensor[0]"]
    d_tri = np.asarray(snap["d_triangles"], np.float64)
    d_sec = np.asarray(snap["d_sec_edges"], np.float64)[:, :6]
    d_prim = np.asarray(cam._primary_edges(True), np.float64)[:, :4]
    d_bsdf = np.array([b.d_reflectance for b in spec.bsdfs], np.float64)
    d_em = np.array([e.d_radiance for e in spec.emitters], np.float64)
    n = spec.width * spec.height
    buf = torch.empty((2, n, 3), dtype=torch.float32, device="cuda")
    a = cabi.make_args(max_depth=depth, seeds=seeds, terms=terms, direct_mis=direct_mis, field=field, field_object=field_object, intensity=intensity)
    cabi.check(cabi.lib().psdr_hip_render_d_fwd(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None))
    gen = torch.Generator(device="cpu").manual_seed(3)
    w = (torch.rand((n, 3), generator=gen) + 0.5).to("cuda")
    if oracle is not None:
        assert direct_mis == -1
        osc = oracle.OracleScene(spec, [0])
        if field >= 0:
            osc.set_field(_FIELD_NAMES[field], obj=field_object, intensity=intensity)
        _img, d_img = osc.render_d(max_depth=depth, seeds=seeds, terms=terms)
        buf[1].copy_(torch.from_numpy(d_img).to("cuda"))
    lhs = float((buf[1].double() * w.double()).sum())
    dev = "cuda"
    g_tri = torch.zeros((d_tri.shape[0], 22), dtype=torch.float32, device=dev)
    n_hidden = sum(1 for b in spec.bsdfs if getattr(b, "type", 0) == 5)          # the BSDF nested in a normal map is a row of its own behind the scene's
    g_bsdf = torch.zeros((max(1, len(spec.bsdfs) + n_hidden), 3), dtype=torch.float32, device=dev)
    g_em = torch.zeros((max(1, len(spec.emitters)), 3), dtype=torch.float32, device=dev)
    g_sec = torch.zeros((max(1, d_sec.shape[0]), 6), dtype=torch.float32, device=dev)
    g_prim = torch.zeros((max(1, d_prim.shape[0]), 4), dtype=torch.float32, device=dev)
    g_cam = torch.zeros(16, dtype=torch.float32, device=dev)
    g = cabi.Grads(g_tri.data_ptr(), g_bsdf.data_ptr(), g_em.data_ptr(), g_sec.data_ptr(), g_prim.data_ptr())
    if with_camera:
        g.g_camera = g_cam.data_ptr()
    g_mat = torch.zeros((max(1, len(spec.bsdfs) + n_hidden), 16), dtype=torch.float32, device=dev)
    if with_mat:
        g.g_mat = g_mat.data_ptr()
    # bitmap parameters: the texel adjoints, laid out as psdr_hip_scene_tex_layout reports
    n_b = len(snap["bsdf_rows"]) if "bsdf_rows" in snap else len(spec.bsdfs) + n_hidden      # (the BSDFs nested in normal maps are rows of their own)
    offs = (C.c_int64 * (3 * max(1, n_b)))(); total = C.c_int64(0)
    cabi.check(cabi.lib().psdr_hip_scene_tex_layout(C.c_void_p(sc._hip_handle()), offs, C.byref(total)))
    g_tex = torch.zeros(max(1, total.value), dtype=torch.float32, device=dev)
    if total.value > 0:
        g.g_tex = g_tex.data_ptr()
    cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
    torch.cuda.synchronize()
    rhs = (g_tri.cpu().numpy().astype(np.float64) * d_tri).sum() + (g_bsdf.cpu().numpy().astype(np.float64)[:len(spec.bsdfs)] * d_bsdf).sum()
    rhs += (g_em.cpu().numpy().astype(np.float64)[:len(spec.emitters)] * d_em).sum()
    rhs += (g_sec.cpu().numpy().astype(np.float64)[:d_sec.shape[0]] * d_sec).sum() + (g_prim.cpu().numpy().astype(np.float64)[:d_prim.shape[0]] * d_prim).sum()
    if with_camera:
        d_tw = np.asarray(cam._get("to_world_left", True), np.float64).reshape(4, 4) if hasattr(cam, "_get") else np.zeros((4, 4))
        rhs += (g_cam.cpu().numpy().astype(np.float64).reshape(4, 4)[:3] * d_tw[:3]).sum()
    if total.value > 0:    # d texel / dP of the spec's bitmaps (reflectance / diffuse, specular, roughness)
        gt = g_tex.cpu().numpy().astype(np.float64)
        for i, b in enumerate(spec.bsdfs):
            pv = getattr(b, "type", 0) == 4      # per-vertex arrays use the same three blocks: diffuse, specular, roughness
            for k, (name, dname) in enumerate((("pv_diffuse", "d_pv_diffuse"), ("pv_specular", "d_pv_specular"), ("pv_roughness", "d_pv_roughness")) if pv else
                                              (("texture", "d_texture"), ("spec_texture", "d_spec_texture"), ("rough_texture", "d_rough_texture"))):
                t, dt = getattr(b, name, None), getattr(b, dname, None)
                if t is not None and dt is not None and offs[3 * i + k] >= 0:
                    dt = np.asarray(dt, np.float64).ravel()
                    rhs += (gt[offs[3 * i + k]:offs[3 * i + k] + dt.size] * dt).sum()
    if with_mat:          # g_mat row of a Microfacet BSDF: specular rgb, roughness
        gm = g_mat.cpu().numpy().astype(np.float64)
        for i, b in enumerate(spec.bsdfs):
            if getattr(b, "type", 0) == 1:
                rhs += (gm[i, 0:3] * np.asarray(getattr(b, "d_specular", (0, 0, 0)), np.float64)).sum() + gm[i, 3] * float(getattr(b, "d_roughness", 0.0))
            if getattr(b, "type", 0) == 3:      # RoughDielectric: alpha_u, alpha_v, eta (1 / eta moves with it)
                rhs += gm[i, 0] * float(getattr(b, "d_alpha_u", 0.0)) + gm[i, 1] * float(getattr(b, "d_alpha_v", 0.0)) + gm[i, 2] * float(np.asarray(getattr(b, "d_eta", (0, 0, 0)))[0])
            if getattr(b, "type", 0) == 2:      # RoughConductor: alpha_u, alpha_v, eta rgb, k rgb, specular rgb
                rhs += gm[i, 0] * float(getattr(b, "d_alpha_u", 0.0)) + gm[i, 1] * float(getattr(b, "d_alpha_v", 0.0))
                rhs += (gm[i, 2:5] * np.asarray(getattr(b, "d_eta", (0, 0, 0)), np.float64)).sum() + (gm[i, 5:8] * np.asarray(getattr(b, "d_k", (0, 0, 0)), np.float64)).sum()
                rhs += (gm[i, 8:11] * np.asarray(getattr(b, "d_specular", (0, 0, 0)), np.float64)).sum()
    # parameters of the BSDFs nested in normal maps: their adjoints arrive in the hidden rows (the scene's own copy of that BSDF shades nothing)
    hidden = len(spec.bsdfs)
    for b in spec.bsdfs:
        if getattr(b, "type", 0) == 5:
            inner = spec.bsdfs[b.nested]
            rhs += (g_bsdf.cpu().numpy().astype(np.float64)[hidden] * np.asarray(inner.d_reflectance, np.float64)).sum()
            if with_mat and getattr(inner, "type", 0) == 1:
                gm = g_mat.cpu().numpy().astype(np.float64)
                rhs += (gm[hidden, 0:3] * np.asarray(getattr(inner, "d_specular", (0, 0, 0)), np.float64)).sum() + gm[hidden, 3] * float(getattr(inner, "d_roughness", 0.0))
            hidden += 1
    scale = float((buf[1].double().abs() * w.double()).sum()) + 1e-12
    return lhs, rhs, scale


@pytest.mark.parametrize("param", ["light_x", "box_x", "albedo"])
def test_interior_sweep_deep_paths(env, param):
    """the reverse sweep of the interior term (adjoint.h) at depth 5: every bounce's position, normal, area, colour adjoints"""
    lhs, rhs, scale = _dot_product_case(env, scenes.cbox_scene(40, 40, 8, 0, 0, param=param), depth=5, terms=1)
    assert abs(lhs - rhs) <= 2e-4 * scale and abs(lhs) > 1e-6, (param, lhs, rhs, scale)


def test_interior_sweep_smooth_meshes(env):
    """tutorial sphere box (class 2: BVH scene, interpolated vertex normals): the light and the small sphere move - positions,
    areas and the blended shading normals of a finely tessellated mesh"""
    lhs, rhs, scale = _dot_product_case(env, scenes.sphere_scene(48, 48, 8, 0, 0), depth=3, terms=1)
    assert abs(lhs - rhs) <= 3e-4 * scale and abs(lhs) > 1e-6, (lhs, rhs, scale)
    lhs, rhs, scale = _dot_product_case(env, scenes.sphere_scene(40, 40, 4, 4, 4), depth=2, terms=7)
    assert abs(lhs - rhs) <= 3e-4 * scale, (lhs, rhs, scale)


def test_interior_sweep_two_sided_and_flat(env):
    """two-sided Diffuse BSDFs seen from the back and flat-shaded meshes (the shading normal is the face normal)"""
    spec = scenes.cbox_scene(40, 40, 8, 0, 0, param="box_x")
    for b in spec.bsdfs:
        b.two_sided = True
    for m in spec.meshes[1:3]:
        m.use_face_normals = True
    lhs, rhs, scale = _dot_product_case(env, spec, depth=3, terms=1)
    assert abs(lhs - rhs) <= 2e-4 * scale and abs(lhs) > 1e-6, (lhs, rhs, scale)


def test_reverse_mode_beyond_the_lds_records(env):
    """paths too deep for per-lane records in LDS (160 KB per workgroup: ~10 bounces with the scene blob beside them) keep their
    records in global memory: depth 24 through the reverse sweep (Diffuse box) and depth 12 through record-and-probe (GGX box),
    against the forward tangents of the same kernels"""
    lhs, rhs, scale = _dot_product_case(env, scenes.cbox_scene(32, 32, 4, 0, 0, param="box_x"), depth=24, terms=1)
    assert abs(lhs - rhs) <= 3e-4 * scale and abs(lhs) > 1e-6, (lhs, rhs, scale)
    lhs, rhs, scale = _dot_product_case(env, scenes.microfacet_cbox_scene(24, 24, 2, 0, 0, param="box_x"), depth=12, terms=1)
    assert abs(lhs - rhs) <= 5e-4 * scale and abs(lhs) > 1e-6, (lhs, rhs, scale)


@pytest.mark.parametrize("param,balls", [("box_x", False), ("albedo", False), ("albedo", True)])
def test_interior_sweep_environment_map(env, param, balls):
    """Diffuse scenes lit by an environment map (class 2): next-event samples on the scene box and BSDF rays that leave the scene
    look the radiance up along their direction - the sweep carries that direction's adjoint back to the shading point"""
    lhs, rhs, scale = _dot_product_case(env, scenes.envmap_scene(40, 40, 8, 0, 0, param=param, area_light=True, balls=balls), depth=3, terms=1)
    assert abs(lhs - rhs) <= 3e-4 * scale and abs(lhs) > 1e-6, (param, lhs, rhs, scale)


@pytest.mark.parametrize("param,two_sided", [("box_x", False), ("diffuse", False), ("roughness", False), ("specular", False), ("box_x", True), ("roughness", True)])
def test_interior_sweep_microfacet(env, param, two_sided):
    """Cornell box with Microfacet (GGX) boxes: the material sweep (adjoint_mat.h) - the lobe depends on both directions, so a bounce
    also reaches the vertex before it; geometry, diffuse reflectance, specular and roughness adjoints against forward mode"""
    spec = scenes.microfacet_cbox_scene(40, 40, 8, 0, 0, param=param, two_sided=two_sided)
    lhs, rhs, scale = _dot_product_case(env, spec, depth=3, terms=1, with_mat=True)
    assert abs(lhs - rhs) <= 5e-4 * scale and abs(lhs) > 1e-6, (param, two_sided, lhs, rhs, scale)


@pytest.mark.parametrize("scene,param", [("diffuse", "box_x"), ("diffuse", "texture"), ("microfacet", "box_x"), ("microfacet", "diffuse"),
                                         ("microfacet", "specular"), ("microfacet", "roughness")])
def test_interior_sweep_bitmap_parameters(env, scene, param):
    """bitmap parameters in the material sweep: the adjoint of a looked-up value is scattered over the four texels of its footprint,
    and at the camera vertex - whose barycentrics are differentiable - chained to the texture coordinates"""
    spec = scenes.textured_scene(40, 40, 8, 0, 0, param=param) if scene == "diffuse" else scenes.textured_microfacet_scene(40, 40, 8, 0, 0, param=param)
    lhs, rhs, scale = _dot_product_case(env, spec, depth=3, terms=1, with_mat=True)
    assert abs(lhs - rhs) <= 5e-4 * scale and abs(lhs) > 1e-6, (scene, param, lhs, rhs, scale)


def test_sweeps_under_a_high_resolution_hdr_map(env):
    """ballroom_1k.exr (1024 x 512, values up to 140): the lookup is piecewise bilinear, so the reverse pass must land in the texel cell the
    forward pass used - it rebuilds the lookup direction through the cube face's frame exactly as Intersection / EnvironmentMap::eval do.
    Diffuse boxes (class 2 sweep) and the notebook's glossy bunny (material sweep)"""
    from psdr_jit_amd import exr
    hdr = np.ascontiguousarray(exr.read_rgb(os.path.join(scenes.DATA, "..", "envmap", "ballroom_1k.exr")))
    lhs, rhs, scale = _dot_product_case(env, scenes.envmap_scene(96, 96, 16, 0, 0, param="box_x", env=hdr), depth=3, terms=1, seeds=(3, 3, 3))
    assert abs(lhs - rhs) <= 1e-5 * scale and abs(lhs) > 1e-6, (lhs, rhs, scale)
    lhs, rhs, scale = _dot_product_case(env, scenes.envmap_tutorial_scene(96, 96, 16, 0, 0, param="bunny_x", env_stride=1), depth=2, terms=1, seeds=(3, 3, 3), with_mat=True)
    assert abs(lhs - rhs) <= 1e-5 * scale and abs(lhs) > 1e-6, (lhs, rhs, scale)


@pytest.mark.parametrize("param", ["box_x", "alpha", "eta", "k"])
def test_interior_sweep_roughconductor(env, param):
    """anisotropic GGX conductors in the material sweep: the lobe sees the tangent vectors of the shading frame, whose adjoints go
    back through the frame construction (uv parameterisation or Duff et al.) to the shading normal and the triangle's edges"""
    lhs, rhs, scale = _dot_product_case(env, scenes.conductor_cbox_scene(40, 40, 8, 0, 0, param=param), depth=3, terms=1, with_mat=True)
    assert abs(lhs - rhs) <= 5e-4 * scale and abs(lhs) > 1e-6, (param, lhs, rhs, scale)


@pytest.mark.parametrize("param", ["box_x", "alpha", "eta"])
def test_interior_sweep_roughdielectric(env, param):
    """rough glass in the material sweep: paths refract into and out of the closed small box; the lobe's adjoint covers reflection and
    transmission, eta carries 1 / eta with it"""
    lhs, rhs, scale = _dot_product_case(env, scenes.dielectric_cbox_scene(40, 40, 8, 0, 0, param=param), depth=4, terms=1, with_mat=True)
    assert abs(lhs - rhs) <= 5e-4 * scale and abs(lhs) > 1e-6, (param, lhs, rhs, scale)


@pytest.mark.parametrize("param", ["ball_x", "diffuse", "specular", "roughness"])
def test_interior_sweep_per_vertex_parameters(env, param):
    """MicrofacetBSDFPerVertex in the material sweep: the adjoint of an interpolated value goes to the triangle's three vertices with
    the barycentric weights, and at the camera vertex to the barycentrics themselves"""
    lhs, rhs, scale = _dot_product_case(env, scenes.pervertex_scene(40, 40, 8, 0, 0, param=param), depth=3, terms=1, with_mat=True)
    assert abs(lhs - rhs) <= 5e-4 * scale and abs(lhs) > 1e-6, (param, lhs, rhs, scale)


@pytest.mark.parametrize("mis", [0, 1, 2])
@pytest.mark.parametrize("scene", ["diffuse", "microfacet"])
def test_sweeps_under_the_direct_integrator(env, scene, mis):
    """psdr.Direct(mis) (direct.cpp:34-132) through the sweeps: emitter sampling only / BSDF sampling only / both with the power heuristic"""
    spec = scenes.cbox_scene(40, 40, 8, 0, 0, param="light_x") if scene == "diffuse" else scenes.microfacet_cbox_scene(40, 40, 8, 0, 0, param="roughness")
    lhs, rhs, scale = _dot_product_case(env, spec, depth=1, terms=1, with_mat=True, direct_mis=mis)
    # (Direct(1) - BSDF sampling only - has no interior derivative with respect to a material constant in the reference's formulation:
    # both sides are exactly zero there, as in the oracle)
    assert abs(lhs - rhs) <= 5e-4 * scale and (abs(lhs) > 1e-6 or (mis == 1 and scene == "microfacet")), (scene, mis, lhs, rhs, scale)


@pytest.mark.parametrize("nested,nmap,param", [("microfacet", "bumpy", "nmap"), ("microfacet", "bumpy", "nested"), ("microfacet", "bumpy", "box_x"),
                                               ("diffuse", "tilted", "nmap"), ("diffuse", "bumpy", "box_x"), ("diffuse", "flat", "nested"), ("microfacet", "tilted", "roughness")])
def test_interior_sweep_normalmap(env, nested, nmap, param):
    """NormalMap BSDFs in the material sweep (round 3; record-and-probe before): the nested BSDF's own adjoint routine for its parameters and the directions
    in the perturbed frame, forward evaluations of the map's geometry for the chain to wi, wo, the map value (constant: g_bsdf row, bitmap: texels) and dp_du
    (the triangle's edges); a box moves, the map's texels / constant move, the nested BSDF's colour / roughness moves"""
    spec = scenes.normalmap_scene(40, 40, 8, 0, 0, param=None if param == "roughness" else param, nested=nested, nmap=nmap)
    if param == "roughness":
        spec.bsdfs[-1].d_roughness = 1.0
    lhs, rhs, scale = _dot_product_case(env, spec, depth=3, terms=1, with_mat=True)
    assert abs(lhs - rhs) <= 5e-4 * scale and abs(lhs) > 1e-6, (nested, nmap, param, lhs, rhs, scale)


def _all_adjoint_buffers(env, spec, depth, monkeypatch, probe, field=-1, intensity=1.0):
    """every adjoint buffer of psdr_hip_render_d_bwd (interior term) from the reverse sweep or, probe=True, from record-and-probe (PSDR_ADJ_PROBE is read per call)"""
    torch, psdr, cabi = env
    if probe:
        monkeypatch.setenv("PSDR_ADJ_PROBE", "1")
    else:
        monkeypatch.delenv("PSDR_ADJ_PROBE", raising=False)
    sc = product.build_scene(spec)
    snap = sc._snapshot()
    n_tris = np.asarray(snap["d_triangles"]).shape[0]
    n_b = len(snap["bsdf_rows"]) if "bsdf_rows" in snap else len(spec.bsdfs)
    z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device="cuda")
    g_tri, g_bsdf, g_em, g_sec, g_prim, g_cam, g_mat = z(n_tris, 22), z(max(1, n_b), 3), z(max(1, len(spec.emitters)), 3), z(1, 6), z(1, 4), z(16), z(max(1, n_b), 16)
    g = cabi.Grads(g_tri.data_ptr(), g_bsdf.data_ptr(), g_em.data_ptr(), g_sec.data_ptr(), g_prim.data_ptr())
    g.g_camera, g.g_mat = g_cam.data_ptr(), g_mat.data_ptr()
    offs = (C.c_int64 * (3 * max(1, n_b)))(); total = C.c_int64(0)
    cabi.check(cabi.lib().psdr_hip_scene_tex_layout(C.c_void_p(sc._hip_handle()), offs, C.byref(total)))
    g_tex = z(max(1, total.value))
    if total.value > 0:
        g.g_tex = g_tex.data_ptr()
    out = {"tri": g_tri, "bsdf": g_bsdf, "emitter": g_em, "camera": g_cam, "mat": g_mat, "tex": g_tex}
    env_em = [e for e in spec.emitters if getattr(e, "type", 0) == 1]
    if env_em:
        H, W = env_em[0].env_data.shape[:2]
        out["env"], out["env_scale"], out["env_xf"] = z(H * W * 3), z(1), z(16)
        g.g_env, g.g_env_scale, g.g_env_from_world = out["env"].data_ptr(), out["env_scale"].data_ptr(), out["env_xf"].data_ptr()
    gen = torch.Generator(device="cpu").manual_seed(11)
    w = (torch.rand((spec.width * spec.height, 3), generator=gen) + 0.5).to("cuda")
    a = cabi.make_args(max_depth=depth, seeds=(7, 8, 9), terms=1, field=field, intensity=intensity)
    cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
    torch.cuda.synchronize()
    monkeypatch.delenv("PSDR_ADJ_PROBE", raising=False)
    return {k: v.cpu().numpy().astype(np.float64) for k, v in out.items()}


def _assert_same_buffers(sw, pr, tol, what):
    seen = 0
    for k in sw:
        ref = np.abs(pr[k]).sum()
        if ref == 0.0 and np.abs(sw[k]).sum() == 0.0:
            continue
        seen += 1
        err = np.abs(sw[k] - pr[k]).sum() / (ref + 1e-12)
        assert err <= tol, (what, k, err)
    assert seen >= 2, what          # the comparison is not vacuous


def test_normalmap_sweep_equals_the_probe_form(env, monkeypatch):
    """the material sweep (adjoint_mat.h) and record-and-probe are two derivations of the same adjoints: every buffer of one against the other on the
    same samples - triangle rows (positions, blended normals, face normal, area), colours, GGX constants, the normal map's texels, the camera pose"""
    spec = scenes.normalmap_scene(48, 48, 8, 0, 0, param="box_x", nested="microfacet", nmap="bumpy")
    sw = _all_adjoint_buffers(env, spec, 3, monkeypatch, probe=False)
    pr = _all_adjoint_buffers(env, spec, 3, monkeypatch, probe=True)
    _assert_same_buffers(sw, pr, 5e-4, "normalmap")


_FAMILIES = {
    "cbox": lambda: scenes.cbox_scene(40, 40, 8, 0, 0, param="box_x"),
    "cbox_camera": lambda: scenes.cbox_scene(40, 40, 8, 0, 0, param="camera_x"),
    "sphere": lambda: scenes.sphere_scene(40, 40, 8, 0, 0),
    "microfacet": lambda: scenes.microfacet_cbox_scene(40, 40, 8, 0, 0, param="roughness"),
    "microfacet_two_sided": lambda: scenes.microfacet_cbox_scene(40, 40, 8, 0, 0, param="box_x", two_sided=True),
    "conductor": lambda: scenes.conductor_cbox_scene(40, 40, 8, 0, 0, param="alpha"),
    "dielectric": lambda: scenes.dielectric_cbox_scene(40, 40, 8, 0, 0, param="alpha"),
    "textured_env": lambda: scenes.textured_scene(40, 40, 8, 0, 0, param="box_x"),
    "textured_microfacet": lambda: scenes.textured_microfacet_scene(40, 40, 8, 0, 0, param="roughness"),
    "pervertex": lambda: scenes.pervertex_scene(40, 40, 8, 0, 0, param="ball_x"),
    "normalmap_diffuse": lambda: scenes.normalmap_scene(40, 40, 8, 0, 0, param="box_x", nested="diffuse", nmap="bumpy"),
    "textured_conductor": lambda: scenes.textured_ggx_scene(40, 40, 8, 0, 0, kind="roughconductor", param="alpha"),
    "textured_dielectric": lambda: scenes.textured_ggx_scene(40, 40, 8, 0, 0, kind="roughdielectric", param="alpha"),
    "envmap_tutorial": lambda: scenes.envmap_tutorial_scene(40, 40, 8, 0, 0, param="bunny_x"),
    "ortho": lambda: scenes.ortho_cbox_scene(40, 40, 8, 0, 0, param="box_x"),
}


@pytest.mark.parametrize("family", list(_FAMILIES))
def test_every_reverse_form_against_the_probe_form(env, monkeypatch, family):
    """whatever reverse form the launch picks for a scene (class-1 / class-2 sweep, material sweep, or record-and-probe itself) against record-and-probe forced, every buffer
    the call can fill (triangle rows, colours, emitters, GGX constants, texels / per-vertex values, camera pose, environment texels / scale / rotation) - the cross-check that
    found the probe form's stage-3 defect in round 4, over every scene family of the suite"""
    spec = _FAMILIES[family]()
    sw = _all_adjoint_buffers(env, spec, 3, monkeypatch, probe=False)
    pr = _all_adjoint_buffers(env, spec, 3, monkeypatch, probe=True)
    if any(getattr(e, "type", 0) == 1 for e in spec.emitters):
        for q in (sw, pr):
            q["tri"][-12:] = 0.0          # rows of the environment map's bounding box (see test_environment_sweep_equals_the_probe_form)
    _assert_same_buffers(sw, pr, 5e-4, family)


@pytest.mark.parametrize("balls", [False, True])
def test_environment_sweep_equals_the_probe_form(env, monkeypatch, balls):
    """class-2 reverse sweep against record-and-probe with EVERY optional buffer requested (texels, scale and rotation of the map, camera pose): round 4 found
    the probe form losing the y / z components of the vertex-normal adjoints (and those of every hit behind the camera's) as soon as a path had recorded
    lookups - its stage-3 loop ran again in stage 6 (adjoint.h) -, which no dot-product test saw because translations leave the normals' tangents zero"""
    spec = scenes.envmap_scene(40, 40, 8, 0, 0, param="box_rot_x", area_light=True, balls=balls)
    sw = _all_adjoint_buffers(env, spec, 3, monkeypatch, probe=False)
    pr = _all_adjoint_buffers(env, spec, 3, monkeypatch, probe=True)
    for q in (sw, pr):
        q["tri"][-12:] = 0.0          # the rows of the map's bounding box, appended last by Scene::configure (scene.cpp:434-485): fixed geometry, read by nobody - the
                                      # sweep glues its vertices to their triangles, the probe form differentiates the box hit as a ray-plane solve
    _assert_same_buffers(sw, pr, 3e-4, "envmap balls=%s" % balls)


@pytest.mark.parametrize("family", ["cbox_camera", "cbox_radiance", "conductor", "dielectric", "textured_env", "textured_microfacet", "pervertex", "textured_conductor", "envmap_tutorial", "ortho", "sphere"])
def test_reverse_mode_against_the_oracle_more_families(env, orc, family):
    """the remaining scene families of the suite, reverse mode against the oracle's forward mode (interior term, depth 3): camera pose, emitter radiance, GGX conductors and
    dielectrics (constants and bitmaps), textures under an environment map, per-vertex parameters, the tutorial's glossy mesh under its map, the orthographic camera, the BVH box"""
    kw = {}
    if family == "cbox_camera":
        spec, kw = scenes.cbox_scene(40, 40, 8, 0, 0, param="camera_x"), {"with_camera": True}
    elif family == "cbox_radiance":
        spec = scenes.cbox_scene(40, 40, 8, 0, 0, param="radiance")
    elif family == "conductor":
        spec, kw = scenes.conductor_cbox_scene(40, 40, 8, 0, 0, param="eta"), {"with_mat": True}
    elif family == "dielectric":
        spec, kw = scenes.dielectric_cbox_scene(40, 40, 8, 0, 0, param="alpha"), {"with_mat": True}
    elif family == "textured_env":
        spec = scenes.textured_scene(40, 40, 8, 0, 0, param="texture")
    elif family == "textured_microfacet":
        spec, kw = scenes.textured_microfacet_scene(40, 40, 8, 0, 0, param="roughness"), {"with_mat": True}
    elif family == "pervertex":
        spec, kw = scenes.pervertex_scene(40, 40, 8, 0, 0, param="roughness"), {"with_mat": True}
    elif family == "textured_conductor":
        spec, kw = scenes.textured_ggx_scene(40, 40, 8, 0, 0, kind="roughconductor", param="alpha"), {"with_mat": True}
    elif family == "envmap_tutorial":
        spec, kw = scenes.envmap_tutorial_scene(40, 40, 8, 0, 0, param="bunny_x"), {"with_mat": True}
    elif family == "ortho":
        spec = scenes.ortho_cbox_scene(40, 40, 8, 0, 0, param="box_x")
    else:
        spec = scenes.sphere_scene(40, 40, 8, 0, 0)
    lhs, rhs, scale = _dot_product_case(env, spec, depth=3, terms=1, oracle=orc, **kw)
    assert abs(lhs - rhs) <= 1e-3 * scale and abs(lhs) > 1e-6, (family, lhs, rhs, scale)


@pytest.mark.parametrize("family", ["diffuse_env", "diffuse_env_rot", "microfacet", "normalmap", "cbox"])
def test_reverse_sweeps_against_the_oracle(env, orc, family):
    """every sweep family against the ORACLE's forward mode (no HIP forward kernel in the loop, no finite differences): class-2 sweep under an environment map (translation and
    rotation of a smooth mesh), the material sweep of a Microfacet box, the normal-map sweep, the class-1 sweep of the README box"""
    if family == "diffuse_env":
        spec, kw = scenes.envmap_scene(40, 40, 8, 0, 0, param="box_x", area_light=True, balls=True), {}
    elif family == "diffuse_env_rot":
        spec, kw = scenes.envmap_scene(40, 40, 8, 0, 0, param="box_rot_x", area_light=True, balls=True), {}
    elif family == "microfacet":
        spec, kw = scenes.microfacet_cbox_scene(40, 40, 8, 0, 0, param="roughness"), {"with_mat": True}
    elif family == "normalmap":
        spec, kw = scenes.normalmap_scene(40, 40, 8, 0, 0, param="box_x", nested="microfacet", nmap="bumpy"), {"with_mat": True}
    else:
        spec, kw = scenes.cbox_scene(40, 40, 8, 0, 0, param="box_x"), {}
    lhs, rhs, scale = _dot_product_case(env, spec, depth=3, terms=1, oracle=orc, **kw)
    assert abs(lhs - rhs) <= 1e-3 * scale and abs(lhs) > 1e-6, (family, lhs, rhs, scale)


@pytest.mark.parametrize("axis", ["box_rot", "box_rot_x", "box_rot_z"])
@pytest.mark.parametrize("probe", [False, True])
def test_interior_adjoint_rotation_of_a_smooth_mesh(env, monkeypatch, axis, probe):
    """a mesh with interpolated normals turning about each axis: positions AND vertex normals carry tangents, for the sweep and for the probe form"""
    if probe:
        monkeypatch.setenv("PSDR_ADJ_PROBE", "1")
    else:
        monkeypatch.delenv("PSDR_ADJ_PROBE", raising=False)
    spec = scenes.envmap_scene(40, 40, 8, 0, 0, param=axis, area_light=True, balls=True)
    d_tri = np.asarray(product.build_scene(spec)._snapshot()["d_triangles"])
    assert np.abs(d_tri[:, 9:18]).sum() > 100.0                  # the normals do move
    lhs, rhs, scale = _dot_product_case(env, spec, depth=3, terms=1)
    monkeypatch.delenv("PSDR_ADJ_PROBE", raising=False)
    assert abs(lhs - rhs) <= 3e-4 * scale and abs(lhs) > 1e-6, (axis, probe, lhs, rhs, scale)


@pytest.mark.parametrize("kind,param", [("roughconductor", "eta"), ("roughconductor", "k"), ("roughconductor", "alpha"), ("roughconductor", "box_x"),
                                        ("roughdielectric", "alpha"), ("roughdielectric", "box_x")])
def test_interior_sweep_ggx_bitmaps(env, kind, param):
    """eta / k / alpha bitmaps of RoughConductor and the alpha bitmap of RoughDielectric in the material sweep (round 3; record-and-probe before): the
    looked-up values' adjoints go to the four texels of each footprint (one alpha map feeds both axes), geometry through the lobe as with constants"""
    spec = scenes.textured_ggx_scene(40, 40, 8, 0, 0, kind=kind, param=param)
    lhs, rhs, scale = _dot_product_case(env, spec, depth=3 if kind == "roughconductor" else 4, terms=1, with_mat=True)
    assert abs(lhs - rhs) <= 5e-4 * scale and abs(lhs) > 1e-6, (kind, param, lhs, rhs, scale)


_FIRST_HIT_SCENES = {
    "cbox": lambda: scenes.cbox_scene(40, 40, 8, 0, 0, param="box_x"),
    "cbox_camera": lambda: scenes.cbox_scene(40, 40, 8, 0, 0, param="camera_x"),
    "sphere": lambda: scenes.sphere_scene(40, 40, 8, 0, 0),
    "microfacet": lambda: scenes.microfacet_cbox_scene(40, 40, 8, 0, 0, param="roughness"),
    "microfacet_two_sided": lambda: scenes.microfacet_cbox_scene(40, 40, 8, 0, 0, param="box_x", two_sided=True),
    "conductor": lambda: scenes.conductor_cbox_scene(40, 40, 8, 0, 0, param="alpha"),
    "textured_microfacet": lambda: scenes.textured_microfacet_scene(40, 40, 8, 0, 0, param="roughness"),
    "textured_env": lambda: scenes.textured_scene(40, 40, 8, 0, 0, param="box_x"),
    "pervertex": lambda: scenes.pervertex_scene(40, 40, 8, 0, 0, param="ball_x"),
    "normalmap": lambda: scenes.normalmap_scene(40, 40, 8, 0, 0, param="box_x", nested="microfacet", nmap="bumpy"),
    "ortho": lambda: scenes.ortho_cbox_scene(40, 40, 8, 0, 0, param="box_x"),
}


@pytest.mark.parametrize("family", list(_FIRST_HIT_SCENES))
def test_first_hit_integrators_sweep_equals_the_probe_form(env, monkeypatch, family):
    """FieldExtractionIntegrator (field.cpp:49-121: position, depth, geometric / shading normal, uv, bsdf) and CollocatedIntegrator (collocated.cpp:24-55) in reverse mode: the
    one-pass form (the material sweep's camera-hit block, round 4) against record-and-probe forced, every buffer the call can fill"""
    spec = _FIRST_HIT_SCENES[family]()
    seen = 0
    for field, intensity in ((1, 1.0), (2, 1.0), (3, 1.0), (4, 1.0), (5, 1.0), (6, 1.0), (8, 2e5)):
        sw = _all_adjoint_buffers(env, spec, 0, monkeypatch, probe=False, field=field, intensity=intensity)
        pr = _all_adjoint_buffers(env, spec, 0, monkeypatch, probe=True, field=field, intensity=intensity)
        live = [k for k in pr if np.abs(pr[k]).sum() > 0.0]
        if not live:
            assert all(np.abs(sw[k]).sum() == 0.0 for k in sw), (family, field)     # (e.g. the texture coordinates of a mesh without them)
            continue
        seen += 1
        for k in sw:
            ref = np.abs(pr[k]).sum()
            err = np.abs(sw[k] - pr[k]).sum() / (ref + 1e-12) if ref > 0.0 else np.abs(sw[k]).sum()
            assert err <= 5e-4, (family, field, k, err)
    assert seen >= 5, family


def test_first_hit_sweep_is_the_same_in_every_run(env, monkeypatch):
    """Round 5: the sweep of the normal-mapped scene returned different adjoints from run to run (4 of 12 800 samples read stale registers: the not-inlined
    bsdf_back lambda of adjoint_mat.h carried the allocator defect of LABNOTES section 4, and isa_lint.py looked at kernels only).  The persistent waves hand
    samples to lanes in a different order every launch, so a stale register shows as run-to-run variation: ten launches, each against the probe form"""
    spec = _FIRST_HIT_SCENES["normalmap"]()
    for field, intensity in ((6, 1.0), (8, 2e5)):
        pr = _all_adjoint_buffers(env, spec, 0, monkeypatch, probe=True, field=field, intensity=intensity)
        for run in range(10):
            sw = _all_adjoint_buffers(env, spec, 0, monkeypatch, probe=False, field=field, intensity=intensity)
            for k in sw:
                ref = np.abs(pr[k]).sum()
                err = np.abs(sw[k] - pr[k]).sum() / (ref + 1e-12) if ref > 0.0 else np.abs(sw[k]).sum()
                assert err <= 1e-5, (field, run, k, err)


_EVERY_RUN = {
    # scene, depth: one per reverse kernel family (the scene class picks the kernel, api.hip::psdr_hip_render_d_bwd)
    "class1_lds": (lambda: scenes.cbox_scene(40, 40, 8, 8, 8, param="box_x"), 3),                            # k_interior_adjoint<1>, k_paths<false,1,.,1> with weights, k_secondary_edges<1,.,true>
    "class2_bvh": (lambda: scenes.sphere_scene(40, 40, 8, 8, 8), 3),                                           # the class-2 unit (records in global memory) and the BVH edge kernels
    "class2_env": (lambda: scenes.envmap_scene(40, 40, 8, 8, 8, param="box_x", area_light=True, balls=True), 3),     # ... with environment lookups on record
    "material_sweep": (lambda: scenes.microfacet_cbox_scene(40, 40, 8, 8, 8, param="roughness"), 3),          # adjoint_mat.h (class 0 / 3)
    "normalmap": (lambda: scenes.normalmap_scene(40, 40, 8, 8, 8, param="box_x", nested="microfacet", nmap="bumpy"), 2),
}


@pytest.mark.parametrize("family", list(_EVERY_RUN))
def test_every_adjoint_kernel_is_the_same_in_every_run(env, family):
    """The allocator defect of LABNOTES section 4 (a vector instruction ahead of a join block's exec restore) leaves lanes with stale registers; persistent waves deal the
    samples to other lanes in every launch, so it shows as RUN-TO-RUN variation.  isa_lint.py looks for the pattern in the ISA; this is the runtime side of the same guard,
    for every reverse kernel family and all three terms: ten launches of psdr_hip_render_d_bwd on one scene, every buffer against the first launch (float atomics reorder
    sums: 1e-5 of a buffer's L1 norm; a stale register showed as 1e-3 and more)"""
    torch, psdr, cabi = env
    make, depth = _EVERY_RUN[family]
    spec = make()
    sc = product.build_scene(spec)
    snap = sc._snapshot()
    n_tris = np.asarray(snap["d_triangles"]).shape[0]
    n_sec = np.asarray(snap["d_sec_edges"]).shape[0]
    n_prim = np.asarray(sc.param_map["Sensor[0]"]._primary_edges(True)).shape[0]
    n_b = len(snap["bsdf_rows"]) if "bsdf_rows" in snap else len(spec.bsdfs)
    offs = (C.c_int64 * (3 * max(1, n_b)))(); total = C.c_int64(0)
    cabi.check(cabi.lib().psdr_hip_scene_tex_layout(C.c_void_p(sc._hip_handle()), offs, C.byref(total)))
    gen = torch.Generator(device="cpu").manual_seed(11)
    w = (torch.rand((spec.width * spec.height, 3), generator=gen) + 0.5).to("cuda")
    env_em = [e for e in spec.emitters if getattr(e, "type", 0) == 1]

    def launch(terms):
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device="cuda")
        out = {"tri": z(n_tris, 22), "bsdf": z(max(1, n_b), 3), "emitter": z(max(1, len(spec.emitters)), 3), "sec": z(max(1, n_sec), 6), "prim": z(max(1, n_prim), 4),
               "camera": z(16), "mat": z(max(1, n_b), 16), "tex": z(max(1, total.value))}
        g = cabi.Grads(out["tri"].data_ptr(), out["bsdf"].data_ptr(), out["emitter"].data_ptr(), out["sec"].data_ptr(), out["prim"].data_ptr())
        g.g_camera, g.g_mat = out["camera"].data_ptr(), out["mat"].data_ptr()
        if total.value > 0:
            g.g_tex = out["tex"].data_ptr()
        if env_em:
            H, W = env_em[0].env_data.shape[:2]
            out["env"], out["env_scale"], out["env_xf"] = z(H * W * 3), z(1), z(16)
            g.g_env, g.g_env_scale, g.g_env_from_world = out["env"].data_ptr(), out["env_scale"].data_ptr(), out["env_xf"].data_ptr()
        a = cabi.make_args(max_depth=depth, seeds=(7, 8, 9), terms=terms)
        cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
        torch.cuda.synchronize()
        return {k: v.cpu().numpy().astype(np.float64) for k, v in out.items()}

    for terms in (1, 2, 4):
        first = launch(terms)
        if terms == 1:
            assert any(np.abs(v).sum() > 0 for v in first.values()), (family, terms)
        for run in range(9):
            again = launch(terms)
            for k in first:
                ref = np.abs(first[k]).sum()
                err = np.abs(again[k] - first[k]).sum() / (ref + 1e-12) if ref > 0.0 else np.abs(again[k]).sum()
                assert err <= 1e-5, (family, terms, run, k, err)


@pytest.mark.parametrize("family,kw", [("cbox", {}), ("cbox_camera", {"with_camera": True}), ("microfacet", {"with_mat": True}), ("conductor", {"with_mat": True}),
                                       ("textured_microfacet", {"with_mat": True}), ("pervertex", {"with_mat": True}), ("normalmap", {"with_mat": True}), ("sphere", {})])
def test_first_hit_integrators_reverse_mode_against_the_oracle(env, orc, family, kw):
    """<w, d_img> from the ORACLE's forward mode of the first-hit integrators against <J^T w, v> from psdr_hip_render_d_bwd (interior + primary-edge terms)"""
    for field, intensity in ((1, 1.0), (2, 1.0), (4, 1.0), (6, 1.0), (8, 2e5)):
        spec = _FIRST_HIT_SCENES[family]()
        spec.sppe = 4
        lhs, rhs, scale = _dot_product_case(env, spec, depth=0, terms=3, oracle=orc, field=field, intensity=intensity, **kw)
        assert abs(lhs - rhs) <= 1e-3 * scale, (family, field, lhs, rhs, scale)


_UV_XF = [[0.4, 1.6, 0.21, -0.13], [-0.7, 0.8, 0.05, 0.3], [1.1, 2.3, -0.4, 0.15]]
_UV_CASES = {
    "diffuse_env": lambda: scenes.textured_scene(40, 40, 8, 0, 0, texture=scenes.checker_texture(33, 17, 5), param=None, env=True),
    "microfacet": lambda: scenes.textured_microfacet_scene(40, 40, 8, 0, 0),
    "roughconductor": lambda: scenes.textured_ggx_scene(40, 40, 8, 0, 0, kind="roughconductor"),
    "roughdielectric": lambda: scenes.textured_ggx_scene(40, 40, 8, 0, 0, kind="roughdielectric"),
    "normalmap": lambda: scenes.normalmap_scene(40, 40, 8, 0, 0),
    "envmap": lambda: scenes.envmap_scene(40, 40, 8, 0, 0, param=None),
    "envmap_balls": lambda: scenes.envmap_scene(40, 40, 8, 0, 0, param=None, area_light=True, balls=True),
}


@pytest.mark.parametrize("probe", [False, True])
@pytest.mark.parametrize("case", list(_UV_CASES))
def test_uv_transform_adjoints_in_one_pass(env, orc, monkeypatch, case, probe):
    """psdr_grads.g_uv_xf (ABI 12): the adjoints of rotate / scale / translate of every bitmap (bitmap.cpp:64-86) from the pass that fills g_tex / g_env - every
    reverse form, <J^T w, v> against <w, J v> with J v from the ORACLE's forward mode, all components of all bitmaps of the scene carrying a tangent at once"""
    torch, psdr, cabi = env
    spec = _UV_CASES[case]()
    rng = np.random.default_rng(5)
    d_tex = rng.uniform(-1.0, 1.0, size=(len(spec.bsdfs), 3, 4))
    for i, b in enumerate(spec.bsdfs):
        if any(getattr(b, k, None) is not None for k in ("texture", "spec_texture", "rough_texture")):
            b.tex_xf, b.d_tex_xf = _UV_XF, d_tex[i]
        else:
            d_tex[i] = 0.0
    d_env = np.zeros(4)
    for e in spec.emitters:
        if getattr(e, "type", 0) == 1:
            d_env = rng.uniform(-1.0, 1.0, size=4)
            e.env_uv_xf, e.d_env_uv_xf = (0.3, 1.2, 0.27, -0.1), tuple(float(q) for q in d_env)
    assert np.abs(d_tex).sum() + np.abs(d_env).sum() > 0
    _img, d_img = orc.OracleScene(spec, [0]).render_d(max_depth=3, seeds=(7, 8, 9), terms=1)
    n = spec.width * spec.height
    gen = torch.Generator(device="cpu").manual_seed(3)
    w = (torch.rand((n, 3), generator=gen) + 0.5).to("cuda")
    lhs = float((torch.from_numpy(d_img).to("cuda").double() * w.double()).sum())
    scale = float((torch.from_numpy(d_img).to("cuda").double().abs() * w.double()).sum()) + 1e-12
    if probe:
        monkeypatch.setenv("PSDR_ADJ_PROBE", "1")
    else:
        monkeypatch.delenv("PSDR_ADJ_PROBE", raising=False)
    sc = product.build_scene(spec)
    snap = sc._snapshot()
    n_tris = np.asarray(snap["d_triangles"]).shape[0]
    n_b = len(snap["bsdf_rows"]) if "bsdf_rows" in snap else len(spec.bsdfs)
    z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device="cuda")
    keep = [z(n_tris, 22), z(max(1, n_b), 3), z(max(1, len(spec.emitters)), 3), z(1, 6), z(1, 4)]
    g = cabi.Grads(*[t.data_ptr() for t in keep])
    offs = (C.c_int64 * (3 * max(1, n_b)))(); total = C.c_int64(0)
    cabi.check(cabi.lib().psdr_hip_scene_tex_layout(C.c_void_p(sc._hip_handle()), offs, C.byref(total)))
    g_tex, g_uv = z(max(1, total.value)), z(3 * n_b + 1, 4)
    if total.value > 0:
        g.g_tex = g_tex.data_ptr()
    g.g_uv_xf = g_uv.data_ptr()
    env_em = [e for e in spec.emitters if getattr(e, "type", 0) == 1]
    if env_em:
        H, W = env_em[0].env_data.shape[:2]
        g_env = z(H * W * 3)
        g.g_env = g_env.data_ptr()
    a = cabi.make_args(max_depth=3, seeds=(7, 8, 9), terms=1)
    cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
    torch.cuda.synchronize()
    monkeypatch.delenv("PSDR_ADJ_PROBE", raising=False)
    gu = g_uv.cpu().numpy().astype(np.float64)
    rhs = float((gu[:3 * len(spec.bsdfs)].reshape(len(spec.bsdfs), 3, 4) * d_tex).sum() + (gu[3 * n_b] * d_env).sum())
    assert abs(lhs) > 1e-3 * scale and abs(lhs - rhs) <= 1e-3 * scale, (case, probe, lhs, rhs, scale)
