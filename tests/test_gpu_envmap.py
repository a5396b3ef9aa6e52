"""GPU parity of the EnvironmentMap emitter path (reference src/emitter/envmap.cpp, scene.cpp:434-515)
against the CPU oracle, whose envmap path is pinned analytically in tests/test_oracle_envmap.py."""
import os

import numpy as np
import pytest

import product
import scenes

pytestmark = pytest.mark.gpu
TOL = 1e-3       # BASELINE north_star: gradient L2 error < 1e-3; observed ~1e-7


@pytest.fixture(scope="module")
def psdr():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    import __graft_entry__
    __graft_entry__.build()
    import psdr_jit_amd
    return psdr_jit_amd


@pytest.mark.parametrize("area_light", [False, True])
@pytest.mark.parametrize("depth", [0, 1, 3])
def test_envmap_render_c(psdr, orc, depth, area_light):
    spec = scenes.envmap_scene(64, 48, 16, 0, 0, param=None, area_light=area_light)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    img = psdr.PathTracer(depth).renderC(sc, 0, seed=7).cpu().numpy()
    want = ref.render_c(max_depth=depth, seed=7)
    assert np.isfinite(img).all() and img.shape == want.shape
    assert product.rel_l2(img, want) < TOL


@pytest.mark.parametrize("param,area_light", [("albedo", False), ("box_x", False), ("box_x", True)])
def test_envmap_render_d_terms(psdr, orc, param, area_light):
    spec = scenes.envmap_scene(48, 48, 8, 8, 8, param=param, area_light=area_light)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(2)
    for terms in (orc.TERM_INTERIOR, orc.TERM_PRIMARY, orc.TERM_SECONDARY, orc.TERM_ALL):
        img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=5, terms=terms)
        wimg, wdimg = ref.render_d(max_depth=2, seeds=(5, 5, 5), terms=terms)
        if terms & orc.TERM_INTERIOR:
            assert product.rel_l2(img.cpu().numpy(), wimg) < TOL
        if np.abs(wdimg).max() > 0:
            assert product.rel_l2(dimg.cpu().numpy(), wdimg) < TOL, (param, terms)
        else:
            assert float(dimg.abs().max()) == 0.0


def test_env_sampling_and_pdf_bit_exact(psdr, orc):
    """EnvironmentMap::sample_position / sample_position_pdf alone, through the C ABI"""
    import torch
    from psdr_jit_amd import cabi
    spec = scenes.envmap_scene(32, 32, 1, 0, 0, param=None)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    rng = np.random.default_rng(1)
    n = 100000
    ref_p = rng.uniform([0, 0, 0], [550, 300, 550], size=(n, 3)).astype(np.float32)
    s2 = rng.random((n, 2)).astype(np.float32)
    s2[:100, 1] = 0.0
    s2[100:200, 1] = np.float32(1.0 - 2 ** -24)
    tp, ts = torch.from_numpy(ref_p).cuda(), torch.from_numpy(s2).cuda()
    op, on, opdf = torch.empty((n, 3), device="cuda"), torch.empty((n, 3), device="cuda"), torch.empty(n, device="cuda")
    cabi.check(cabi.lib().psdr_hip_env_sample(sc._hip_handle(), n, tp.data_ptr(), ts.data_ptr(), op.data_ptr(), on.data_ptr(), opdf.data_ptr(), None))
    wp, wn, wpdf = ref.env_sample(ref_p, s2)
    assert np.array_equal(op.cpu().numpy(), wp) and np.array_equal(on.cpu().numpy(), wn) and np.array_equal(opdf.cpu().numpy(), wpdf)
    tpp, tnn = torch.from_numpy(wp).cuda(), torch.from_numpy(wn).cuda()
    cabi.check(cabi.lib().psdr_hip_env_pdf(sc._hip_handle(), n, tp.data_ptr(), tpp.data_ptr(), tnn.data_ptr(), opdf.data_ptr(), None))
    assert np.array_equal(opdf.cpu().numpy(), ref.env_pdf(ref_p, wp, wn))


def test_envmap_bvh_mesh_with_guiding(psdr, orc):
    """BASELINE config 5 in small: envmap-lit curved meshes (> 64 triangles: BVH traversal), albedo parameter,
    secondary-edge guiding"""
    spec = scenes.envmap_scene(48, 48, 8, 8, 8, param="albedo", balls=True)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(2)
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=3)
    wimg, wd = ref.render_d(max_depth=2, seeds=(3, 3, 3))
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL and product.rel_l2(dimg.cpu().numpy(), wd) < TOL
    # moving geometry + guiding grid
    spec = scenes.envmap_scene(48, 48, 4, 4, 8, param=None, balls=True)
    dT = np.zeros((4, 4), dtype=np.float32)
    dT[0, 3] = 100.0
    spec.meshes[0].d_to_world_left = dT
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    reso = [60, 4, 4, 16]
    integ.preprocess_secondary_edges(sc, 0, reso, 1, 2)
    g = ref.guiding_build(0, reso, nrounds=1, seed=2)
    assert product.rel_l2(np.asarray(integ._guiding_mass(0)).reshape(-1), g.mass()) < TOL
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=8)
    wimg, wd = ref.render_d(max_depth=2, seeds=(8, 8, 8), guiding=g)
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL and product.rel_l2(dimg.cpu().numpy(), wd) < TOL


def test_envmap_api_and_reverse_mode(psdr, orc):
    """add_EnvironmentMap through the reference-style API, albedo gradient by loss.backward()"""
    import torch
    D = scenes.DATA
    import os
    sc = psdr.Scene()
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = 8, 0, 0
    sc.opts.width = sc.opts.height = 32
    sc.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    cam.to_world = psdr.Matrix4fD((scenes.translate(278.0, 400.0, -700.0) @ scenes._rot_x(np.radians(25.0))).tolist())
    sc.add_Sensor(cam)
    refl = torch.tensor([0.5, 0.5, 0.5], requires_grad=True)
    sc.add_BSDF(psdr.DiffuseBSDF(refl), "cat")
    sc.add_BSDF(psdr.DiffuseBSDF([0.8, 0.8, 0.8]), "white")
    sc.add_EnvironmentMap(psdr.EnvironmentMap(scenes.synthetic_envmap()))
    I = np.eye(4, dtype=np.float32)
    for f, b in (("cbox_smallbox", "cat"), ("cbox_largebox", "cat"), ("cbox_floor", "white")):
        sc.add_Mesh(os.path.join(D, f + ".obj"), psdr.Matrix4fC(I.tolist()), b, None)
    sc.configure()
    sc.configure([0])
    assert sc.num_meshes == 4
    integ = psdr.PathTracer(2)
    img = integ.renderD(sc, 0, seed=9)
    spec = scenes.envmap_scene(32, 32, 8, 0, 0, param="albedo")
    spec.bsdfs = spec.bsdfs[:2]
    ref = orc.OracleScene(spec, [0])
    wimg, wd = ref.render_d(max_depth=2, seeds=(9, 9, 9))
    assert product.rel_l2(img.detach().cpu().numpy(), wimg) < TOL
    img.sum().backward()
    assert abs(float(refl.grad.sum()) - float(wd.sum())) < 2e-3 * abs(float(wd.sum()))


@pytest.mark.parametrize("env", [True, False])
def test_textured_reflectance_render_c_and_d(psdr, orc, env):
    """Bitmap3fD reflectance (bitmap.cpp:47-128): image, texel tangent and geometry tangent against the oracle"""
    for param, tex in (("texture", scenes.ramp_texture()), ("box_x", scenes.checker_texture()), (None, scenes.checker_texture(33, 17, 5))):
        spec = scenes.textured_scene(48, 48, 8, 8, 8, texture=tex, param=param, env=env)
        sc = product.build_scene(spec)
        ref = orc.OracleScene(spec, [0])
        integ = psdr.PathTracer(2)
        img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=4)
        wimg, wd = ref.render_d(max_depth=2, seeds=(4, 4, 4))
        assert product.rel_l2(img.cpu().numpy(), wimg) < TOL, (param, env)
        if np.abs(wd).max() > 0:
            assert product.rel_l2(dimg.cpu().numpy(), wd) < TOL, (param, env)
        c = psdr.PathTracer(3).renderC(sc, 0, seed=2).cpu().numpy()
        assert product.rel_l2(c, ref.render_c(max_depth=3, seed=2)) < TOL


def test_textured_api(psdr, orc):
    """DiffuseBSDF(Bitmap3fD) / bsdf.reflectance = array through the reference-style API"""
    import os
    tex = scenes.checker_texture()
    spec = scenes.textured_scene(32, 32, 4, 0, 0, texture=tex, env=False)
    sc = psdr.Scene()
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = 4, 0, 0
    sc.opts.width = sc.opts.height = 32
    sc.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    cam.to_world = psdr.Matrix4fD(np.asarray(spec.cameras[0].to_world_raw).tolist())
    sc.add_Sensor(cam)
    sc.add_BSDF(psdr.DiffuseBSDF(psdr.Bitmap3fD(tex.shape[1], tex.shape[0], tex.reshape(-1, 3))), "tex")
    sc.add_BSDF(psdr.DiffuseBSDF([0.5, 0.5, 0.5]), "cat")
    sc.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light")
    floor = psdr.Mesh()
    m = spec.meshes[0]
    floor.load_raw(m.vertices, m.faces, m.uvs, m.face_uvs)
    sc.add_Mesh(floor, "tex", None)
    I = np.eye(4, dtype=np.float32)
    sc.add_Mesh(os.path.join(scenes.DATA, "cbox_smallbox.obj"), psdr.Matrix4fC(I.tolist()), "cat", None)
    sc.add_Mesh(os.path.join(scenes.DATA, "cbox_luminaire.obj"), psdr.Matrix4fC(scenes.translate(0.0, -100.0, 0.0).tolist()), "light", psdr.AreaLight([20.0, 20.0, 8.0]))
    sc.configure()
    sc.configure([0])
    img = psdr.PathTracer(2).renderC(sc, 0, seed=6).cpu().numpy()
    want = orc.OracleScene(spec, [0]).render_c(max_depth=2, seed=6)
    assert product.rel_l2(img, want) < TOL
    # switching back to a constant colour
    sc.param_map["BSDF[id=tex]"].reflectance = [0.5, 0.5, 0.5]
    sc.configure([0])
    spec.bsdfs[0].texture = None
    img = psdr.PathTracer(2).renderC(sc, 0, seed=6).cpu().numpy()
    assert product.rel_l2(img, orc.OracleScene(spec, [0]).render_c(max_depth=2, seed=6)) < TOL


@pytest.mark.parametrize("param", ["diffuse", "specular", "roughness", "box_x"])
def test_microfacet_bitmap_parameters(psdr, orc, param):
    """MicrofacetBSDF with its three parameters as bitmaps (microfacet.cpp:38-45): image, texel tangents of each map and the
    geometry tangent against the oracle"""
    spec = scenes.textured_microfacet_scene(48, 48, 8, 8, 8, param=param)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(2)
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=4)
    wimg, wd = ref.render_d(max_depth=2, seeds=(4, 4, 4))
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL
    assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < TOL
    c = psdr.PathTracer(3).renderC(sc, 0, seed=2).cpu().numpy()
    assert product.rel_l2(c, ref.render_c(max_depth=3, seed=2)) < TOL


def test_microfacet_bitmap_api(psdr, orc):
    """MicrofacetBSDF(Bitmap3fD, Bitmap3fD, Bitmap1fD) and map / constant assignment through the reference-style API"""
    spec = scenes.textured_microfacet_scene(32, 32, 4, 0, 0)
    b = spec.bsdfs[0]
    mf = psdr.MicrofacetBSDF(psdr.Bitmap3fD(b.spec_texture.shape[1], b.spec_texture.shape[0], b.spec_texture.reshape(-1, 3)),
                             psdr.Bitmap3fD(b.texture), psdr.Bitmap1fD(b.rough_texture.shape[1], b.rough_texture.shape[0], b.rough_texture.reshape(-1)))
    assert tuple(mf.roughness.shape) == b.rough_texture.shape and tuple(mf.specularReflectance.shape) == b.spec_texture.shape
    sc2 = psdr.Scene()
    sc2.opts.spp, sc2.opts.sppe, sc2.opts.sppse = 4, 0, 0
    sc2.opts.width = sc2.opts.height = 32
    sc2.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    cam.to_world = psdr.Matrix4fD(np.asarray(spec.cameras[0].to_world_raw).tolist())
    sc2.add_Sensor(cam)
    sc2.add_BSDF(mf, "tex")
    sc2.add_BSDF(psdr.DiffuseBSDF([0.5, 0.5, 0.5]), "cat")
    sc2.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light")
    floor = psdr.Mesh()
    m = spec.meshes[0]
    floor.load_raw(m.vertices, m.faces, m.uvs, m.face_uvs)
    sc2.add_Mesh(floor, "tex", None)
    I = np.eye(4, dtype=np.float32)
    sc2.add_Mesh(os.path.join(scenes.DATA, "cbox_smallbox.obj"), psdr.Matrix4fC(I.tolist()), "cat", None)
    sc2.add_Mesh(os.path.join(scenes.DATA, "cbox_luminaire.obj"), psdr.Matrix4fC(scenes.translate(0.0, -100.0, 0.0).tolist()), "light", psdr.AreaLight([20.0, 20.0, 8.0]))
    sc2.configure()
    sc2.configure([0])
    img = psdr.PathTracer(2).renderC(sc2, 0, seed=6).cpu().numpy()
    assert product.rel_l2(img, orc.OracleScene(spec, [0]).render_c(max_depth=2, seed=6)) < TOL
    # roughness back to a constant, specular stays a map
    sc2.param_map["BSDF[id=tex]"].roughness = 0.3
    sc2.configure([0])
    b.rough_texture = None; b.roughness = 0.3
    img = psdr.PathTracer(2).renderC(sc2, 0, seed=6).cpu().numpy()
    assert product.rel_l2(img, orc.OracleScene(spec, [0]).render_c(max_depth=2, seed=6)) < TOL


@pytest.mark.parametrize("kind", ["diffuse_bsdf", "microfacet"])
def test_texel_adjoints_reverse_mode(psdr, orc, kind):
    """loss.backward() into bitmap parameters (the texture-optimisation use case): <w, J v> == <J^T w, v> with J v from forward
    mode (pinned against the oracle above) for every map, plus the C-ABI layout query"""
    import torch
    rng = np.random.default_rng(11)
    spec = scenes.textured_microfacet_scene(40, 40, 8, 0, 0)
    b = spec.bsdfs[0]
    sc = psdr.Scene()
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = 8, 0, 0
    sc.opts.width = sc.opts.height = 40
    sc.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    cam.to_world = psdr.Matrix4fD(np.asarray(spec.cameras[0].to_world_raw).tolist())
    sc.add_Sensor(cam)
    diff = torch.tensor(b.texture, requires_grad=True)
    leaves = {"diffuse": diff}
    if kind == "microfacet":
        spc = torch.tensor(b.spec_texture, requires_grad=True)
        rgh = torch.tensor(b.rough_texture, requires_grad=True)
        leaves.update(specular=spc, roughness=rgh)
        sc.add_BSDF(psdr.MicrofacetBSDF(spc, diff, rgh), "tex")
    else:
        sc.add_BSDF(psdr.DiffuseBSDF(diff), "tex")
    sc.add_BSDF(psdr.DiffuseBSDF([0.5, 0.5, 0.5]), "cat")
    sc.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light")
    floor = psdr.Mesh()
    m = spec.meshes[0]
    floor.load_raw(m.vertices, m.faces, m.uvs, m.face_uvs)
    sc.add_Mesh(floor, "tex", None)
    I = np.eye(4, dtype=np.float32)
    sc.add_Mesh(os.path.join(scenes.DATA, "cbox_smallbox.obj"), psdr.Matrix4fC(I.tolist()), "cat", None)
    sc.add_Mesh(os.path.join(scenes.DATA, "cbox_luminaire.obj"), psdr.Matrix4fC(scenes.translate(0.0, -100.0, 0.0).tolist()), "light", psdr.AreaLight([20.0, 20.0, 8.0]))
    sc.configure()
    sc.configure([0])
    img = psdr.PathTracer(3).renderD(sc, 0, seed=5)
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    fwd = {}
    for name, t in leaves.items():
        v = torch.tensor(rng.standard_normal(tuple(t.shape)).astype(np.float32))
        fwd[name] = (v, float((psdr.forward_grad(img, t, direction=v) * w).sum()))
    (img * w).sum().backward()
    for name, t in leaves.items():
        v, want = fwd[name]
        assert t.grad is not None and tuple(t.grad.shape) == tuple(t.shape)
        got = float((t.grad * v).sum())
        assert abs(want) > 1e-3 and abs(got - want) < 2e-3 * max(1.0, abs(want)), (name, got, want)
        assert float(t.grad.abs().sum()) > 0


@pytest.mark.parametrize("param", ["bunny_x", "roughness"])
def test_envmap_tutorial_scene(psdr, orc, param):
    """The reference's Forward_AD_envmap tutorial (bunny_low.obj, MicrofacetBSDF, PathTracer(1), the PIZ-compressed ballroom map
    read by psdr_jit_amd.exr): renderC and all three terms of renderD against the oracle"""
    spec = scenes.envmap_tutorial_scene(48, 48, 8, 8, 8, param=param)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(1)
    c = integ.renderC(sc, 0, seed=2).cpu().numpy()
    assert c.mean() > 0.05 and product.rel_l2(c, ref.render_c(max_depth=1, seed=2)) < TOL
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=4)
    wimg, wd = ref.render_d(max_depth=1, seeds=(4, 4, 4))
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL
    assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < TOL


@pytest.mark.parametrize("param", ["texels", "scale", "rotation"])
def test_envmap_parameter_tangents(psdr, orc, param):
    """forward tangents of the EnvironmentMap's differentiable members (texels of m_radiance, m_scale, m_to_world_left;
    envmap.h:40-45) against the oracle, with the area light as a second emitter"""
    env = scenes.synthetic_envmap(64, 32)
    spec = scenes.envmap_scene(48, 48, 8, 0, 0, param=None, env=env, area_light=True)
    e = spec.emitters[0]
    if param == "texels":
        rng = np.random.default_rng(2)
        e.d_env_data = rng.random(env.shape).astype(np.float32)
    elif param == "scale":
        e.env_scale, e.d_env_scale = 1.5, 1.0
    else:
        a = 0.4
        m = np.eye(4, dtype=np.float32); m[0, 0], m[0, 2], m[2, 0], m[2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
        dm = np.zeros((4, 4), np.float32); dm[0, 0], dm[0, 2], dm[2, 0], dm[2, 2] = -np.sin(a), np.cos(a), -np.cos(a), -np.sin(a)
        e.env_to_world_left, e.d_env_to_world_left = m, dm
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(2)
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=4)
    wimg, wd = ref.render_d(max_depth=2, seeds=(4, 4, 4))
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL
    assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < TOL


def test_envmap_reverse_mode(psdr, orc):
    """loss.backward() into the environment map's texels and scale (lighting optimisation): <w, J v> == <J^T w, v> against forward
    mode; the scene also holds a textured floor, so BSDF and environment lookups share the lookup record"""
    import torch
    rng = np.random.default_rng(3)
    spec = scenes.textured_scene(40, 40, 8, 0, 0, texture=scenes.checker_texture(8, 8), env=True)
    env0 = scenes.synthetic_envmap(64, 32)                    # with the sun: rotating a constant map would change nothing
    rad = torch.tensor(env0, requires_grad=True)
    scale = psdr.FloatD(1.25).requires_grad_()
    tex = torch.tensor(spec.bsdfs[0].texture, requires_grad=True)
    sc = psdr.Scene()
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = 8, 0, 0
    sc.opts.width = sc.opts.height = 40
    sc.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    cam.to_world = psdr.Matrix4fD(np.asarray(spec.cameras[0].to_world_raw).tolist())
    sc.add_Sensor(cam)
    sc.add_BSDF(psdr.DiffuseBSDF(tex), "tex")
    sc.add_BSDF(psdr.DiffuseBSDF([0.5, 0.5, 0.5]), "cat")
    floor = psdr.Mesh()
    m = spec.meshes[0]
    floor.load_raw(m.vertices, m.faces, m.uvs, m.face_uvs)
    sc.add_Mesh(floor, "tex", None)
    sc.add_Mesh(os.path.join(scenes.DATA, "cbox_smallbox.obj"), psdr.Matrix4fC(np.eye(4, dtype=np.float32).tolist()), "cat", None)
    ang = psdr.FloatD(0.3).requires_grad_()
    ca, sa = torch.cos(ang), torch.sin(ang)
    e = psdr.EnvironmentMap(rad)
    e.scale = scale
    e.set_transform(psdr.Matrix4fD([[ca, 0., sa, 0.], [0., 1., 0., 0.], [-sa, 0., ca, 0.], [0., 0., 0., 1.]]))
    sc.add_EnvironmentMap(e)
    sc.configure()
    sc.configure([0])
    img = psdr.PathTracer(2).renderD(sc, 0, seed=5)
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    v_rad = torch.tensor(rng.standard_normal(env0.shape).astype(np.float32))
    v_tex = torch.tensor(rng.standard_normal(tuple(tex.shape)).astype(np.float32))
    want_rad = float((psdr.forward_grad(img, rad, direction=v_rad) * w).sum())
    want_scale = float((psdr.forward_grad(img, scale) * w).sum())
    want_tex = float((psdr.forward_grad(img, tex, direction=v_tex) * w).sum())
    want_ang = float((psdr.forward_grad(img, ang) * w).sum())
    (img * w).sum().backward()
    assert abs(want_ang) > 1e-3 and abs(float(ang.grad) - want_ang) < 3e-3 * max(1.0, abs(want_ang)), (float(ang.grad), want_ang)
    got_rad, got_scale, got_tex = float((rad.grad * v_rad).sum()), float(scale.grad), float((tex.grad * v_tex).sum())
    for name, got, want in (("radiance", got_rad, want_rad), ("scale", got_scale, want_scale), ("texture", got_tex, want_tex)):
        assert abs(want) > 1e-3 and abs(got - want) < 2e-3 * max(1.0, abs(want)), (name, got, want)
    # the image is linear in the scale: d/d scale = image / scale
    assert abs(got_scale - float((img.detach() * w).sum()) / 1.25) < 2e-3 * abs(got_scale)


@pytest.mark.parametrize("kind,param", [("roughconductor", "eta"), ("roughconductor", "k"), ("roughconductor", "alpha"), ("roughconductor", "box_x"),
                                        ("roughdielectric", "alpha"), ("roughdielectric", None)])
def test_ggx_bitmap_parameters(psdr, orc, kind, param):
    """RoughConductor with eta / k / alpha bitmaps and RoughDielectric with an alpha bitmap (roughconductor.cpp:37-43,
    roughdielectric.cpp:75-78): image and texel tangents against the oracle"""
    spec = scenes.textured_ggx_scene(48, 48, 8, 8, 8, kind=kind, param=param)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(2)
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=4)
    wimg, wd = ref.render_d(max_depth=2, seeds=(4, 4, 4))
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL
    if param is not None:
        assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < TOL
    c = psdr.PathTracer(3).renderC(sc, 0, seed=2).cpu().numpy()
    assert product.rel_l2(c, ref.render_c(max_depth=3, seed=2)) < TOL


def test_env_cell_masses_on_the_device_equal_the_oracle(psdr, orc):
    """Scene::configure evaluates the cell masses of the environment map on the device (psdr_hip_env_cell_masses, envmap.cpp:17-44,
    cube_distrb.cpp:22-29): bit-equal to the oracle's and to the host half's (configure_host), through the C ABI and through Scene"""
    import ctypes as C
    from psdr_jit_amd import cabi
    spec = scenes.envmap_scene(32, 32, 1, 0, 0, param=None)
    ref = orc.OracleScene(spec, [0])
    _, reso, cell_sum, pmf, cmf = ref.envmap_info()
    dev = product.build_scene(spec)._snapshot()                       # Scene::configure: device masses
    host = product.build_scene(spec, host_only=True)._snapshot()      # configure_host: the host loop
    for snap in (dev, host):
        assert tuple(snap["env_reso"]) == reso
        assert np.array_equal(np.asarray(snap["env_cell_pmf"]).ravel(), pmf) and np.array_equal(np.asarray(snap["env_cell_cmf"]).ravel(), cmf)
        assert float(snap["env_cell_sum"]) == cell_sum
    # the entry point alone, on a larger and rougher map, against the oracle's configure of the same texels
    W, H = 256, 128
    tex = (scenes.synthetic_envmap(W, H) * (0.5 + np.random.default_rng(3).random((H, W, 1), dtype=np.float32))).astype(np.float32)
    _, reso2, _, pmf2, _ = orc.OracleScene(scenes.envmap_scene(16, 16, 1, 0, 0, param=None, env=tex), [0]).envmap_info()
    w2, h2 = 2 * (W - 1), 2 * (H - 1)
    assert reso2 == (w2, h2)
    mass = np.empty(w2 * h2, np.float32)
    cabi.check(cabi.lib().psdr_hip_env_cell_masses(tex.ctypes.data_as(C.c_void_p), C.c_int32(W), C.c_int32(H), mass.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(mass, pmf2)
    with pytest.raises(Exception):
        cabi.check(cabi.lib().psdr_hip_env_cell_masses(tex.ctypes.data_as(C.c_void_p), C.c_int32(1), C.c_int32(H), mass.ctypes.data_as(C.c_void_p)))


def test_envmap_reverse_sweep(psdr, orc):
    """Diffuse scene under an environment map (scene class 2): the interior term's reverse mode is the adjoint sweep (adjoint.h) -
    loss.backward() into the map's texels, scale and rotation, the box translation, the albedo and the camera pose equals forward mode"""
    import torch
    rng = np.random.default_rng(5)
    spec = scenes.envmap_scene(40, 40, 8, 0, 0, param=None)
    env0 = scenes.synthetic_envmap(64, 32)
    rad = torch.tensor(env0, requires_grad=True)
    scale = psdr.FloatD(1.25).requires_grad_()
    albedo = torch.tensor([0.5, 0.4, 0.6], requires_grad=True)
    P = psdr.FloatD(0.).requires_grad_()
    C = psdr.FloatD(0.).requires_grad_()
    sc = psdr.Scene()
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = 8, 0, 0
    sc.opts.width = sc.opts.height = 40
    sc.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    base = torch.tensor(np.asarray(spec.cameras[0].to_world_raw, np.float32))
    shift = torch.zeros(4, 4); shift[0, 3] = 1.0
    cam.to_world = psdr.Matrix4fD(base + shift * C * 50.)
    sc.add_Sensor(cam)
    sc.add_BSDF(psdr.DiffuseBSDF(albedo), "cat")
    sc.add_BSDF(psdr.DiffuseBSDF([0.8, 0.8, 0.8]), "white")
    eye = psdr.Matrix4fC(np.eye(4, dtype=np.float32).tolist())
    sc.add_Mesh(os.path.join(scenes.DATA, "cbox_smallbox.obj"), eye, "cat", None)
    sc.add_Mesh(os.path.join(scenes.DATA, "cbox_largebox.obj"), eye, "cat", None)
    sc.add_Mesh(os.path.join(scenes.DATA, "cbox_floor.obj"), eye, "white", None)
    ang = psdr.FloatD(0.3).requires_grad_()
    ca, sa = torch.cos(ang), torch.sin(ang)
    e = psdr.EnvironmentMap(rad)
    e.scale = scale
    e.set_transform(psdr.Matrix4fD([[ca, 0., sa, 0.], [0., 1., 0., 0.], [-sa, 0., ca, 0.], [0., 0., 0., 1.]]))
    sc.add_EnvironmentMap(e)
    sc.configure()
    sc.param_map["Mesh[0]"].set_transform(psdr.Matrix4fD([[1., 0., 0., P * 100.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    sc.configure([0])
    img = psdr.PathTracer(3).renderD(sc, 0, seed=5)
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    v_rad = torch.tensor(rng.standard_normal(env0.shape).astype(np.float32))
    want = {"radiance": float((psdr.forward_grad(img, rad, direction=v_rad) * w).sum()), "scale": float((psdr.forward_grad(img, scale) * w).sum()),
            "angle": float((psdr.forward_grad(img, ang) * w).sum()), "box": float((psdr.forward_grad(img, P) * w).sum()),
            "albedo": float((psdr.forward_grad(img, albedo, direction=torch.ones(3)) * w).sum()), "camera": float((psdr.forward_grad(img, C) * w).sum())}
    (img * w).sum().backward()
    got = {"radiance": float((rad.grad * v_rad).sum()), "scale": float(scale.grad), "angle": float(ang.grad), "box": float(P.grad),
           "albedo": float(albedo.grad.sum()), "camera": float(C.grad)}
    for name in want:
        assert abs(want[name]) > 1e-3 and abs(got[name] - want[name]) < 3e-3 * max(1.0, abs(want[name])), (name, got[name], want[name])


def test_primary_edge_closed_form_under_an_hdr_step_background(psdr, orc):
    """the analytic magnitude case of tests/test_oracle_envmap.py on the HIP path: against the closed form (1 %) and the oracle"""
    W = H = 64
    L_pos, L_neg = 100.0, 1.5
    spec, want = scenes.hdr_step_square_scene(W, H, 0, 128, L_pos=L_pos, L_neg=L_neg)
    sc = product.build_scene(spec)
    _, dimg = psdr.render_d_fwd(psdr.PathTracer(1), sc, 0, seed=5)
    dimg = dimg.cpu().numpy().astype(np.float64).reshape(H, W, 3)
    _, ref = orc.OracleScene(spec, [0]).render_d(max_depth=1, seeds=(5, 5, 5))
    assert product.rel_l2(dimg.reshape(-1, 3), ref) < TOL
    # the camera matrix mirrors x: world x > 0 (the bright side) is the image's left half
    bright, dim = dimg[:, : W // 2].sum(axis=(0, 1)), dimg[:, W // 2:].sum(axis=(0, 1))
    if abs(bright[0]) < abs(dim[0]):
        bright, dim = dim, bright
    assert np.allclose(bright, -want * L_pos, rtol=0.01), (bright, -want * L_pos)
    assert np.allclose(dim, want * L_neg, rtol=0.01), (dim, want * L_neg)


# ---------------------------------------------------------------- Bitmap::m_rot / m_scale / m_trans (bitmap.cpp:64-86, psdr.cpp:204-206, 217-219)
XF = [[0.4, 1.6, 0.21, -0.13], [-0.7, 0.8, 0.05, 0.3], [1.1, 2.3, -0.4, 0.15]]


@pytest.mark.parametrize("comp", [-1, 0, 1, 2, 3])
def test_bitmap_uv_transform_diffuse(psdr, orc, comp):
    """textured DiffuseBSDF under a uv transform: image and the forward tangent of rotate / scale / translate.x / translate.y against the
    oracle (whose tangents are checked against finite differences in tests/test_oracle_envmap.py); edge terms on (box_x moves too for comp -1)"""
    tex = scenes.checker_texture(33, 17, 5)
    spec = scenes.textured_scene(48, 48, 8, 8, 8, texture=tex, param="box_x" if comp < 0 else None, env=True)
    spec.bsdfs[0].tex_xf = [XF[0], [0, 1, 0, 0], [0, 1, 0, 0]]
    if comp >= 0:
        d = np.zeros((3, 4)); d[0, comp] = 1.0
        spec.bsdfs[0].d_tex_xf = d
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    img, dimg = psdr.render_d_fwd(psdr.PathTracer(2), sc, 0, seed=4)
    wimg, wd = ref.render_d(max_depth=2, seeds=(4, 4, 4))
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL
    assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < TOL
    c = psdr.PathTracer(3).renderC(sc, 0, seed=2).cpu().numpy()
    assert product.rel_l2(c, ref.render_c(max_depth=3, seed=2)) < TOL
    # the transform matters: the untransformed texture gives another image
    spec.bsdfs[0].tex_xf = None
    assert product.rel_l2(c, orc.OracleScene(spec, [0]).render_c(max_depth=3, seed=2)) > 0.02


@pytest.mark.parametrize("slot,comp", [(0, 0), (1, 1), (2, 2), (1, 3), (2, 0)])
def test_bitmap_uv_transform_microfacet_slots(psdr, orc, slot, comp):
    """MicrofacetBSDF with three bitmaps, each under its own transform (diffuse, specular, roughness): one component's tangent at a time"""
    spec = scenes.textured_microfacet_scene(48, 48, 8, 8, 8)
    spec.bsdfs[0].tex_xf = XF
    d = np.zeros((3, 4)); d[slot, comp] = 1.0
    spec.bsdfs[0].d_tex_xf = d
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    img, dimg = psdr.render_d_fwd(psdr.PathTracer(2), sc, 0, seed=4)
    wimg, wd = ref.render_d(max_depth=2, seeds=(4, 4, 4))
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL
    assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < TOL


@pytest.mark.parametrize("kind", ["roughconductor", "roughdielectric", "normalmap"])
def test_bitmap_uv_transform_other_bsdfs(psdr, orc, kind):
    """eta / k / alpha maps of RoughConductor, the alpha map of RoughDielectric and a normal map under uv transforms"""
    spec = scenes.normalmap_scene(48, 48, 8, 8, 8) if kind == "normalmap" else scenes.textured_ggx_scene(48, 48, 8, 8, 8, kind=kind)
    spec.bsdfs[0].tex_xf = XF
    d = np.zeros((3, 4))
    d[0 if kind != "roughdielectric" else 2, 0] = 1.0; d[2, 2] = 0.5
    spec.bsdfs[0].d_tex_xf = d
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    img, dimg = psdr.render_d_fwd(psdr.PathTracer(2), sc, 0, seed=4)
    wimg, wd = ref.render_d(max_depth=2, seeds=(4, 4, 4))
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL
    assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < TOL


@pytest.mark.parametrize("comp", [-1, 0, 2])
def test_envmap_radiance_uv_transform(psdr, orc, comp):
    """m_radiance.rotate / scale / translate of the EnvironmentMap: lookups, the cell distribution EnvironmentMap::configure builds from
    the transformed map (device cell masses == the oracle's), emitter sampling and pdf, and the forward tangent of the transform"""
    spec = scenes.envmap_scene(48, 48, 8, 8, 8, param="box_x" if comp < 0 else None)
    spec.emitters[0].env_uv_xf = (0.3, 1.0, 0.27, 0.0)
    if comp >= 0:
        d = [0.0] * 4; d[comp] = 1.0
        spec.emitters[0].d_env_uv_xf = tuple(d)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    img, dimg = psdr.render_d_fwd(psdr.PathTracer(2), sc, 0, seed=4)
    wimg, wd = ref.render_d(max_depth=2, seeds=(4, 4, 4))
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL
    assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < TOL
    spec.emitters[0].env_uv_xf = (0.0, 1.0, 0.0, 0.0)
    assert product.rel_l2(img.cpu().numpy(), orc.OracleScene(spec, [0]).render_d(max_depth=2, seeds=(4, 4, 4))[0]) > 0.02


def test_bitmap_transform_api_and_reverse_mode(psdr, orc):
    """obj.uv_transform(name).rotate / scale / translate (the reference's bitmap.rotate / scale / translate members): assignment reaches the
    renderer, tensors are leaves; backward() == the forward-mode derivative of each component dotted with the adjoint image"""
    import torch
    tex = scenes.checker_texture(33, 17, 5)
    spec = scenes.textured_scene(40, 40, 8, 0, 0, texture=tex, env=True)
    sc = product.build_scene(spec)
    bs = sc.param_map["BSDF[0]"]
    rot = torch.tensor(0.4, requires_grad=True)
    scl = torch.tensor(1.6, requires_grad=True)
    tr = torch.tensor([0.21, -0.13], requires_grad=True)
    t = bs.uv_transform("reflectance")
    t.rotate, t.scale, t.translate = rot, scl, tr
    assert bs.uv_transform("reflectance").rotate is rot
    sc.configure([0])
    integ = psdr.PathTracer(2)
    img = integ.renderD(sc, 0, seed=5)
    spec.bsdfs[0].tex_xf = [XF[0], [0, 1, 0, 0], [0, 1, 0, 0]]
    want = orc.OracleScene(spec, [0]).render_d(max_depth=2, seeds=(5, 5, 5))[0]
    assert product.rel_l2(img.detach().cpu().numpy(), want) < TOL
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    (img * w).sum().backward()
    got = [float(rot.grad), float(scl.grad), float(tr.grad[0]), float(tr.grad[1])]
    for comp in range(4):
        d = np.zeros((3, 4)); d[0, comp] = 1.0
        spec.bsdfs[0].d_tex_xf = d
        wd = orc.OracleScene(spec, [0]).render_d(max_depth=2, seeds=(5, 5, 5))[1]
        ref = float((wd.astype(np.float64) * w.cpu().numpy().reshape(wd.shape)).sum())
        assert abs(got[comp] - ref) < 2e-3 * max(1.0, abs(ref)), (comp, got[comp], ref)
    # plain numbers again: the leaf goes away, the values stay
    t.rotate, t.scale, t.translate = 0.0, 1.0, (0.0, 0.0)
    sc.configure([0])
    spec.bsdfs[0].tex_xf = None
    c = integ.renderC(sc, 0, seed=2).cpu().numpy()
    assert product.rel_l2(c, orc.OracleScene(spec, [0]).render_c(max_depth=2, seed=2)) < TOL
    # a Bitmap3fD handed to a constructor brings its transform along
    bm = psdr.Bitmap3fD(tex.shape[1], tex.shape[0], tex.reshape(-1, 3))
    bm.rotate, bm.translate = 0.25, (0.1, 0.2)
    b2 = psdr.DiffuseBSDF(bm)
    assert np.allclose(np.asarray(b2._get_uv_xf(0, False)), [0.25, 1.0, 0.1, 0.2])
