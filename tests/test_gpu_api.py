"""GPU tests of the drop-in Python surface: the README differentiable-rendering example
(reference README.md:44-107) written against psdr_jit_amd, checked against the CPU oracle."""
import os

import numpy as np
import pytest

import product
import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def psdr():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    import __graft_entry__
    __graft_entry__.build()
    import psdr_jit_amd
    return psdr_jit_amd


def _readme_scene(psdr, P, res=48, spp=8):
    from psdr_jit_amd import Matrix4fC, Matrix4fD
    D = scenes.DATA
    sc = psdr.Scene()
    sc.opts.spp = spp
    sc.opts.sppe = spp
    sc.opts.sppse = spp
    sc.opts.height = res
    sc.opts.width = res
    sc.opts.log_level = 0
    sensor = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    sensor.to_world = Matrix4fD([[1., 0., 0., 208.], [0., 1., 0., 273.], [0., 0., 1., -800.], [0., 0., 0., 1.]])
    sc.add_Sensor(sensor)
    sc.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light")
    sc.add_BSDF(psdr.DiffuseBSDF(), "cat")
    sc.add_BSDF(psdr.DiffuseBSDF([0.95, 0.95, 0.95]), "white")
    sc.add_BSDF(psdr.DiffuseBSDF([0.20, 0.90, 0.20]), "green")
    sc.add_BSDF(psdr.DiffuseBSDF([0.90, 0.20, 0.20]), "red")
    I = [[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]
    sc.add_Mesh(os.path.join(D, "cbox_luminaire.obj"), Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., -0.5], [0., 0., 1., 0.], [0., 0., 0., 1.]]),
                "light", psdr.AreaLight([20.0, 20.0, 8.0]))
    for f, b in (("cbox_smallbox", "cat"), ("cbox_largebox", "cat"), ("cbox_floor", "white"), ("cbox_ceiling", "white"),
                 ("cbox_back", "white"), ("cbox_greenwall", "green"), ("cbox_redwall", "red")):
        sc.add_Mesh(os.path.join(D, f + ".obj"), Matrix4fC(I), b, None)
    sc.param_map["Mesh[0]"].set_transform(Matrix4fD([[1., 0., 0., P * 100.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    sc.configure()
    sc.configure([0])
    return sc


def test_readme_example_forward_and_reverse(psdr, orc):
    import torch
    P = psdr.FloatD(0.).requires_grad_()
    sc = _readme_scene(psdr, P)
    integrator = psdr.PathTracer(2)
    img = integrator.renderD(sc, 0, seed=5)
    assert img.shape == (48 * 48, 3) and img.is_cuda and img.requires_grad
    spec = scenes.cbox_scene(48, 48, 8, 8, 8, param="light_x")
    ref = orc.OracleScene(spec, [0])
    wimg, wd = ref.render_d(max_depth=2, seeds=(5, 5, 5))
    assert product.rel_l2(img.detach().cpu().numpy(), wimg) < 1e-3
    # forward mode: drjit.set_grad(P, 1); drjit.forward_to(img); drjit.grad(img)
    dimg = psdr.forward_grad(img, P)
    assert product.rel_l2(dimg.cpu().numpy(), wd) < 1e-3
    # reverse mode: d(sum of w*img)/dP must equal <w, d img/dP>
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    (img * w).sum().backward()
    want = float((torch.from_numpy(wd).to(img.device) * w).sum())
    assert abs(float(P.grad) - want) < 1e-3 * max(1.0, abs(want))


def test_renderd_makes_the_forward_mode_launch_when_the_previous_image_went_to_forward_grad(psdr, orc):
    """An optimisation loop repeats renderD -> forward_grad(img, P): from the second iteration on renderD installs P's tangents and makes the one launch
    forward mode needs (image and derivative of the same pass); forward_grad then returns what renderD kept.  Same numbers as the replay, with fixed seeds
    and with continuing samplers (seed = -1); a reverse-mode call on such an image still works and switches the prediction off"""
    import torch
    P = psdr.FloatD(0.).requires_grad_()
    sc = _readme_scene(psdr, P)
    integrator = psdr.PathTracer(2)
    spec = scenes.cbox_scene(48, 48, 8, 8, 8, param="light_x")
    ref = orc.OracleScene(spec, [0])
    for it, seed in enumerate((5, 6, 7)):
        img = integrator.renderD(sc, 0, seed=seed)
        predicted = img.grad_fn.state.get("dimg") is not None
        assert predicted == (it > 0)
        dimg = psdr.forward_grad(img, P)
        wimg, wd = ref.render_d(max_depth=2, seeds=(seed, seed, seed))
        assert product.rel_l2(img.detach().cpu().numpy(), wimg) < 1e-3 and product.rel_l2(dimg.cpu().numpy(), wd) < 1e-3
    # continuing samplers: the predicted call advances the three streams exactly like the unpredicted one (one renderD each)
    before = [sc._sampler_state(k) for k in range(3)]
    img = integrator.renderD(sc, 0)
    assert img.grad_fn.state.get("dimg") is not None
    after = [sc._sampler_state(k) for k in range(3)]
    d1 = psdr.forward_grad(img, P)
    assert [sc._sampler_state(k) for k in range(3)] == after and after != before
    # another parameter than the predicted one: the replay path, same image
    refl = torch.tensor([0.90, 0.20, 0.20], requires_grad=True)
    sc.param_map["BSDF[id=red]"].reflectance = refl
    sc.configure([0])
    img = integrator.renderD(sc, 0, seed=9)
    d_r = psdr.forward_grad(img, refl, direction=torch.tensor([1.0, 1.0, 1.0]))
    d_p = psdr.forward_grad(img, P)
    spec_a = scenes.cbox_scene(48, 48, 8, 8, 8, param="albedo")
    _, wd_a = orc.OracleScene(spec_a, [0]).render_d(max_depth=2, seeds=(9, 9, 9))
    _, wd_p = ref.render_d(max_depth=2, seeds=(9, 9, 9))
    assert product.rel_l2(d_r.cpu().numpy(), wd_a) < 1e-3 and product.rel_l2(d_p.cpu().numpy(), wd_p) < 1e-3
    # reverse mode on a predicted image
    img = integrator.renderD(sc, 0, seed=11)
    assert img.grad_fn.state.get("dimg") is not None
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    P.grad = None
    (img * w).sum().backward()
    _, wd11 = ref.render_d(max_depth=2, seeds=(11, 11, 11))
    want = float((torch.from_numpy(wd11).to(img.device) * w).sum())
    assert abs(float(P.grad) - want) < 1e-3 * max(1.0, abs(want))
    img = integrator.renderD(sc, 0, seed=12)
    assert img.grad_fn.state.get("dimg") is None          # (the last image went to backward)


def test_renderc_and_sampler_continuation(psdr, orc):
    import torch
    sc = _readme_scene(psdr, torch.tensor(0.0))
    integrator = psdr.PathTracer(1)
    a = integrator.renderC(sc, 0)        # seed=-1: scene.seed (0) streams from configure()
    b = integrator.renderC(sc, 0)        # continues the streams: different noise
    spec = scenes.cbox_scene(48, 48, 8, 8, 8, param=None)
    ref = orc.OracleScene(spec, [0])
    assert product.rel_l2(a.cpu().numpy(), ref.render_c(max_depth=1, seed=0)) < 1e-3
    assert product.rel_l2(b.cpu().numpy(), ref.render_c(max_depth=1, seed=0, skip=7)) < 1e-3
    assert not torch.allclose(a, b)


def test_albedo_gradient_through_autograd(psdr, orc):
    import torch
    sc = _readme_scene(psdr, torch.tensor(0.0))
    refl = torch.tensor([0.90, 0.20, 0.20], requires_grad=True)
    sc.param_map["BSDF[id=red]"].reflectance = refl
    sc.configure([0])
    integrator = psdr.PathTracer(2)
    img = integrator.renderD(sc, 0, seed=3)
    img.sum().backward()
    spec = scenes.cbox_scene(48, 48, 8, 8, 8, param="albedo")
    ref = orc.OracleScene(spec, [0])
    _, wd = ref.render_d(max_depth=2, seeds=(3, 3, 3))       # d/dP with d albedo/dP = (1,1,1)
    assert abs(float(refl.grad.sum()) - float(wd.sum())) < 1e-3 * abs(float(wd.sum()))


def test_derivative_passes_without_a_geometry_leaf_skip_the_boundary_terms(psdr, monkeypatch):
    """Both edge terms are detach(radiance difference) x the edge point's normal velocity (integrator.cpp:179-198, path.cpp:171-294): a colour moves no edge, so
    forward_grad / backward w.r.t. it launch the interior term alone - and return what the full launch returns: derivative image bit for bit, the three sampler
    streams where a full renderD leaves them, the same gradient.  A mesh transform among the differentiated leaves brings the edge terms back."""
    import torch
    sc = _readme_scene(psdr, torch.tensor(0.0))
    refl = torch.tensor([0.90, 0.20, 0.20], requires_grad=True)
    sc.param_map["BSDF[id=red]"].reflectance = refl
    sc.configure([0])
    integ = psdr.PathTracer(2)
    launched = []
    raw = psdr._render_d_raw
    monkeypatch.setattr(psdr, "_render_d_raw", lambda integrator, scene, sensor_id, seed, batch_pix, terms, distributed=None:
                        (launched.append(terms), raw(integrator, scene, sensor_id, seed, batch_pix, terms, distributed))[1])
    start = [sc._sampler_state(k) for k in range(3)]
    img = integ.renderD(sc, 0)
    d = psdr.forward_grad(img, refl)
    assert launched == [0x71, 0x71], launched                 # primal pass, then the replay with the colour's tangent: interior only, all three streams advance
    after = [sc._sampler_state(k) for k in range(3)]
    # the full launch from the same stream positions with the same tangent
    for k, s in enumerate(start):
        sc._set_sampler_state(k, *s)
    img_full, d_full = psdr.render_d_fwd(integ, sc, 0, tangents={refl: np.ones(3, np.float32)})
    assert launched[-1] == 7
    assert [sc._sampler_state(k) for k in range(3)] == after
    assert torch.equal(img.detach(), img_full) and torch.equal(d, d_full) and float(d.abs().sum()) > 0
    psdr._sync_params(sc, None, integ)
    sc.configure([0])
    # ... a mesh translation does move edges
    P = torch.tensor(0.0, requires_grad=True)
    sc2 = _readme_scene(psdr, P)
    dP = psdr.forward_grad(integ.renderD(sc2, 0), P)
    assert launched[-1] == 7 and float(dP.abs().sum()) > 0
    # reverse mode: the colour's gradient with and without the edge kernels
    bwd = psdr._core._render_d_bwd
    terms_seen = []
    monkeypatch.setattr(psdr._core, "_render_d_bwd", lambda *a: (terms_seen.append(a[14]), bwd(*a))[1])
    grads = []
    for full in (False, True):
        if full:
            monkeypatch.setattr(psdr, "_derivative_terms", lambda terms, leaves, moving: terms)
        img = integ.renderD(sc, 0, seed=11)
        w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
        (g,) = torch.autograd.grad((img * w).sum(), refl)
        grads.append(g.double())
    assert terms_seen == [1, 7], terms_seen
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * float(grads[1].abs().max()) and float(grads[1].abs().max()) > 0
    monkeypatch.undo()
    terms_seen = []
    monkeypatch.setattr(psdr._core, "_render_d_bwd", lambda *a: (terms_seen.append(a[14]), bwd(*a))[1])
    integ.renderD(sc2, 0, seed=11).sum().backward()
    assert terms_seen == [7] and P.grad is not None


def test_vertex_and_transform_gradients_reverse_matches_forward(psdr):
    """loss.backward() through the adjoint kernels + host chain == <w, forward derivative> for vertex positions
    (a many-parameter leaf only reverse mode can handle) and for the mesh transform."""
    import torch
    P = psdr.FloatD(0.).requires_grad_()
    sc = _readme_scene(psdr, P)
    mesh = sc.param_map["Mesh[1]"]                      # the small box
    V = torch.tensor(np.asarray(mesh._get("vertex_positions", False)), requires_grad=True)
    mesh.vertex_positions = V
    sc.configure([0])
    integrator = psdr.PathTracer(2)
    img = integrator.renderD(sc, 0, seed=4)
    gen = torch.Generator().manual_seed(0)
    w = (torch.rand(img.shape, generator=gen) + 0.5).to(img.device)
    dV = torch.randn(V.shape, generator=gen)
    d_img_V = psdr.forward_grad(img, V, direction=dV)
    d_img_P = psdr.forward_grad(img, P)
    (img * w).sum().backward()
    lhs_V, rhs_V = float((V.grad * dV).sum()), float((d_img_V * w).sum())
    lhs_P, rhs_P = float(P.grad), float((d_img_P * w).sum())
    assert abs(lhs_V - rhs_V) < 2e-3 * max(1.0, abs(rhs_V)), (lhs_V, rhs_V)
    assert abs(lhs_P - rhs_P) < 2e-3 * max(1.0, abs(rhs_P)), (lhs_P, rhs_P)
    assert abs(rhs_V) > 1e-3 and abs(rhs_P) > 1e-3


def test_xml_scene_renders_like_the_scripted_scene(psdr, orc):
    """Scene.load_string (reference scene_loader.cpp) of the README Cornell box == the add_* version, against the oracle"""
    D = scenes.DATA
    shapes = "".join(
        '<shape type="obj"><string name="filename" value="%s/%s.obj"/><ref id="%s"/></shape>' % (D, f, b)
        for f, b in (("cbox_smallbox", "cat"), ("cbox_largebox", "cat"), ("cbox_floor", "white"), ("cbox_ceiling", "white"),
                     ("cbox_back", "white"), ("cbox_greenwall", "green"), ("cbox_redwall", "red")))
    xml = """<scene version="0.6.0">
  <sensor type="perspective"><float name="fov" value="60"/><float name="near_clip" value="0.000001"/><float name="far_clip" value="10000000"/>
    <transform name="to_world"><translate x="208" y="273" z="-800"/></transform>
    <sampler type="independent"><integer name="sample_count" value="8"/></sampler>
    <film type="hdrfilm"><integer name="width" value="48"/><integer name="height" value="48"/></film></sensor>
  <bsdf type="diffuse" id="light"><rgb name="reflectance" value="0, 0, 0"/></bsdf>
  <bsdf type="diffuse" id="cat"><rgb name="reflectance" value="0.5"/></bsdf>
  <bsdf type="diffuse" id="white"><rgb name="reflectance" value="0.95, 0.95, 0.95"/></bsdf>
  <bsdf type="diffuse" id="green"><rgb name="reflectance" value="0.20, 0.90, 0.20"/></bsdf>
  <bsdf type="diffuse" id="red"><rgb name="reflectance" value="0.90, 0.20, 0.20"/></bsdf>
  <shape type="obj"><string name="filename" value="%s/cbox_luminaire.obj"/><ref id="light"/>
    <transform name="to_world"><translate y="-0.5"/></transform><emitter type="area"><rgb name="radiance" value="20, 20, 8"/></emitter></shape>
  %s
</scene>""" % (D, shapes)
    sc = psdr.Scene()
    sc.opts.log_level = 0
    sc.load_string(xml)                       # auto_configure
    assert (sc.opts.spp, sc.opts.sppe, sc.opts.sppse) == (8, 0, 0)
    img = psdr.PathTracer(2).renderC(sc, 0, seed=4).cpu().numpy()
    ref = orc.OracleScene(scenes.cbox_scene(48, 48, 8, 0, 0, param=None), [0])
    assert product.rel_l2(img, ref.render_c(max_depth=2, seed=4)) < 1e-3


@pytest.mark.parametrize("mis", [0, 1, 2])
def test_direct_integrator(psdr, orc, mis):
    """psdr.Direct(mis) (reference direct.cpp:34-132): renderC, all three renderD terms, sampler continuation, reverse mode"""
    import torch
    spec = scenes.cbox_scene(48, 48, 8, 8, 8, param="light_x")
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    ref.set_direct_mis(mis)
    integ = psdr.Direct(mis)
    assert integ.mis == mis and integ.max_depth == 1
    a = integ.renderC(sc, 0, seed=5).cpu().numpy()
    assert product.rel_l2(a, ref.render_c(max_depth=1, seed=5)) < 1e-3
    nd = {0: 2, 1: 3, 2: 5}[mis]
    b = integ.renderC(sc, 0).cpu().numpy()                       # continues the stream: 2 + nd draws were used
    assert product.rel_l2(b, ref.render_c(max_depth=1, seed=5, skip=2 + nd)) < 1e-3
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=9)
    wimg, wd = ref.render_d(max_depth=1, seeds=(9, 9, 9))
    assert product.rel_l2(img.cpu().numpy(), wimg) < 1e-3 and product.rel_l2(dimg.cpu().numpy(), wd) < 1e-3
    if mis == 2:
        pt = psdr.PathTracer(1).renderC(sc, 0, seed=5).cpu().numpy()
        assert np.array_equal(a, pt)
    # reverse mode == forward mode
    P = psdr.FloatD(0.).requires_grad_()
    sc2 = _readme_scene(psdr, P)
    im = integ.renderD(sc2, 0, seed=2)
    w = torch.linspace(0.5, 1.5, im.numel(), device=im.device).reshape(im.shape)
    d = psdr.forward_grad(im, P)
    (im * w).sum().backward()
    want = float((d * w).sum())
    assert abs(float(P.grad) - want) < 2e-3 * max(1.0, abs(want))


@pytest.mark.parametrize("param", ["box_x", "camera_x"])
def test_orthographic_camera(psdr, orc, param):
    """psdr.OrthographicCamera(near, far) (reference orthographic.cpp): renderC and all terms of renderD against the oracle"""
    spec = scenes.ortho_cbox_scene(48, 48, 8, 8, 8, param=param)
    sc = product.build_scene(spec)
    assert sc.param_map["Sensor[0]"].orthographic
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(2)
    c = integ.renderC(sc, 0, seed=3).cpu().numpy()
    assert c.max() > 0 and product.rel_l2(c, ref.render_c(max_depth=2, seed=3)) < 1e-3
    for terms in (1, 2, 4, 7):
        img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=6, terms=terms)
        wimg, wd = ref.render_d(max_depth=2, seeds=(6, 6, 6), terms=terms)
        if terms & 1:
            assert product.rel_l2(img.cpu().numpy(), wimg) < 1e-3
        if np.abs(wd).max() > 0:
            assert product.rel_l2(dimg.cpu().numpy(), wd) < 1e-3, (param, terms)


@pytest.mark.parametrize("param,two_sided", [("roughness", False), ("specular", False), ("diffuse", True), ("box_x", False)])
def test_microfacet_bsdf(psdr, orc, param, two_sided):
    """psdr.MicrofacetBSDF (reference microfacet.cpp / ggx.cpp): renderC, renderD (all terms, parameter and geometry tangents)"""
    spec = scenes.microfacet_cbox_scene(48, 48, 8, 8, 8, param=param, two_sided=two_sided)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(3)
    c = integ.renderC(sc, 0, seed=3).cpu().numpy()
    assert product.rel_l2(c, ref.render_c(max_depth=3, seed=3)) < 1e-3
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=6)
    wimg, wd = ref.render_d(max_depth=3, seeds=(6, 6, 6))
    assert product.rel_l2(img.cpu().numpy(), wimg) < 1e-3
    assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < 1e-3


def test_microfacet_api_and_reverse_mode(psdr, orc):
    """MicrofacetBSDF through the reference-style API; loss.backward() reaches the diffuse reflectance and the geometry"""
    import torch
    P = psdr.FloatD(0.).requires_grad_()
    diff = torch.tensor([0.3, 0.4, 0.5], requires_grad=True)
    from psdr_jit_amd import Matrix4fC, Matrix4fD
    D = scenes.DATA
    sc = psdr.Scene()
    sc.opts.spp = sc.opts.sppe = sc.opts.sppse = 8
    sc.opts.width = sc.opts.height = 40
    sc.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    cam.to_world = Matrix4fD([[1., 0., 0., 208.], [0., 1., 0., 273.], [0., 0., 1., -800.], [0., 0., 0., 1.]])
    sc.add_Sensor(cam)
    sc.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light")
    sc.add_BSDF(psdr.MicrofacetBSDF([0.6, 0.5, 0.4], diff, 0.35), "cat")
    sc.add_BSDF(psdr.DiffuseBSDF([0.95, 0.95, 0.95]), "white")
    I = np.eye(4, dtype=np.float32).tolist()
    sc.add_Mesh(os.path.join(D, "cbox_luminaire.obj"), Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., -0.5], [0., 0., 1., 0.], [0., 0., 0., 1.]]), "light", psdr.AreaLight([20.0, 20.0, 8.0]))
    for f, b in (("cbox_smallbox", "cat"), ("cbox_largebox", "cat"), ("cbox_floor", "white"), ("cbox_back", "white")):
        sc.add_Mesh(os.path.join(D, f + ".obj"), Matrix4fC(I), b, None)
    sc.param_map["Mesh[1]"].set_transform(Matrix4fD([[1., 0., 0., P * 100.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    sc.configure()
    sc.configure([0])
    integ = psdr.PathTracer(2)
    img = integ.renderD(sc, 0, seed=4)
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    d_P = psdr.forward_grad(img, P)
    d_diff = psdr.forward_grad(img, diff, direction=torch.tensor([1.0, -0.5, 0.25]))
    (img * w).sum().backward()
    assert abs(float(P.grad) - float((d_P * w).sum())) < 2e-3 * max(1.0, abs(float((d_P * w).sum())))
    lhs = float((diff.grad * torch.tensor([1.0, -0.5, 0.25])).sum())
    assert abs(lhs - float((d_diff * w).sum())) < 2e-3 * max(1.0, abs(lhs))


@pytest.mark.parametrize("param", ["alpha", "eta", "k", "box_x"])
def test_roughconductor_bsdf(psdr, orc, param):
    """psdr.RoughConductorBSDF (reference roughconductor.cpp: anisotropic GGX + conductor Fresnel) against the oracle"""
    spec = scenes.conductor_cbox_scene(48, 48, 8, 8, 8, param=param)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(3)
    c = integ.renderC(sc, 0, seed=3).cpu().numpy()
    assert product.rel_l2(c, ref.render_c(max_depth=3, seed=3)) < 1e-3
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=6)
    wimg, wd = ref.render_d(max_depth=3, seeds=(6, 6, 6))
    assert product.rel_l2(img.cpu().numpy(), wimg) < 1e-3
    assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < 1e-3
    b = psdr.RoughConductorBSDF(0.2, [0.2, 0.9, 1.1], [3.9, 2.4, 2.1])
    assert float(np.asarray(b.alpha_u)[0]) == float(np.asarray(b.alpha_v)[0]) == np.float32(0.2) and np.allclose(np.asarray(b.k), [3.9, 2.4, 2.1])


@pytest.mark.parametrize("param", ["alpha", "eta", "box_x"])
def test_roughdielectric_bsdf(psdr, orc, param):
    """psdr.RoughDielectricBSDF (reference roughdielectric.cpp: GGX reflection + refraction, dielectric Fresnel) against the oracle"""
    spec = scenes.dielectric_cbox_scene(48, 48, 8, 8, 8, param=param)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(4)
    c = integ.renderC(sc, 0, seed=3).cpu().numpy()
    assert product.rel_l2(c, ref.render_c(max_depth=4, seed=3)) < 1e-3
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=6)
    wimg, wd = ref.render_d(max_depth=4, seeds=(6, 6, 6))
    assert product.rel_l2(img.cpu().numpy(), wimg) < 1e-3
    assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < 1e-3
    b = psdr.RoughDielectricBSDF(0.2, 1.5, 1.0)
    assert float(np.asarray(b.alpha_u)[0]) == float(np.asarray(b.alpha_v)[0]) == np.float32(0.2) and float(np.asarray(b.eta)[0]) == 1.5
    assert float(np.asarray(psdr.RoughDielectricBSDF().eta)[0]) == 1.5 and float(np.asarray(psdr.RoughDielectricBSDF(1.33, 1.0).alpha_u)[0]) == np.float32(0.1)


def test_field_extraction_and_collocated_integrators(psdr, orc):
    """psdr.FieldExtractionIntegrator / psdr.CollocatedIntegrator (reference field.cpp, collocated.cpp) against the oracle"""
    import torch
    spec = scenes.cbox_scene(40, 40, 4, 8, 8, param="box_x")
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    for name in ("silhouette", "position", "depth", "geoNormal", "shNormal", "uv", "bsdf", "segmentation"):
        # (the silhouette of the closed box does not move with the small box: restrict it to that mesh)
        integ = psdr.FieldExtractionIntegrator(name + (" 1" if name == "silhouette" else ""))
        ref.set_field(name, obj=1 if name == "silhouette" else -1)
        img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=3)
        wimg, wd = ref.render_d(max_depth=0, seeds=(3, 3, 3))
        assert product.rel_l2(img.cpu().numpy(), wimg) < 1e-3, name
        assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < 1e-3, name
    # object filter by mesh index, renderC, sampler continuation (2 draws per call)
    integ = psdr.FieldExtractionIntegrator("depth 2")
    ref.set_field("depth", obj=2)
    a = integ.renderC(sc, 0, seed=5).cpu().numpy()
    assert a.max() > 0 and product.rel_l2(a, ref.render_c(max_depth=0, seed=5)) < 1e-3
    b = integ.renderC(sc, 0).cpu().numpy()
    assert product.rel_l2(b, ref.render_c(max_depth=0, seed=5, skip=2)) < 1e-3
    with pytest.raises(RuntimeError, match="Unsupported field"):
        psdr.FieldExtractionIntegrator("albedo")
    col = psdr.CollocatedIntegrator(5e5)
    ref.set_field("collocated", intensity=5e5)
    img, dimg = psdr.render_d_fwd(col, sc, 0, seed=7)
    wimg, wd = ref.render_d(max_depth=0, seeds=(7, 7, 7))
    assert product.rel_l2(img.cpu().numpy(), wimg) < 1e-3 and product.rel_l2(dimg.cpu().numpy(), wd) < 1e-3
    # a microfacet scene through the collocated integrator (its usual companion) + reverse mode == forward mode
    spec2 = scenes.microfacet_cbox_scene(40, 40, 4, 4, 0, param="diffuse")
    sc2 = product.build_scene(spec2)
    ref2 = orc.OracleScene(spec2, [0])
    ref2.set_field("collocated", intensity=2e5)
    col2 = psdr.CollocatedIntegrator(2e5)
    img, dimg = psdr.render_d_fwd(col2, sc2, 0, seed=1)
    wimg, wd = ref2.render_d(max_depth=0, seeds=(1, 1, 1))
    assert product.rel_l2(img.cpu().numpy(), wimg) < 1e-3 and product.rel_l2(dimg.cpu().numpy(), wd) < 1e-3
    P = psdr.FloatD(0.).requires_grad_()
    sc3 = _readme_scene(psdr, P)
    im = psdr.FieldExtractionIntegrator("depth").renderD(sc3, 0, seed=2)
    w = torch.linspace(0.5, 1.5, im.numel(), device=im.device).reshape(im.shape)
    d = psdr.forward_grad(im, P)
    (im * w).sum().backward()
    want = float((d * w).sum())
    assert abs(want) > 1e-3 and abs(float(P.grad) - want) < 2e-3 * max(1.0, abs(want))
    # ... and with an object filter (Mesh::get_obj_mask, mesh.h:49-63): the small box alone, which is the mesh that moves
    P = psdr.FloatD(0.).requires_grad_()
    sc3 = _readme_scene(psdr, P)
    im = psdr.FieldExtractionIntegrator("position 0").renderD(sc3, 0, seed=2)
    d = psdr.forward_grad(im, P)
    (im * w).sum().backward()
    want = float((d * w).sum())
    assert abs(want) > 1e-3 and abs(float(P.grad) - want) < 2e-3 * max(1.0, abs(want))
    whole = psdr.FieldExtractionIntegrator("position").renderD(sc3, 0, seed=2)
    assert float((whole.detach() - im.detach()).abs().sum()) > 1.0          # the filter removed the rest of the room


def test_scene_without_emitters(psdr):
    """silhouette / depth rendering needs no light: the first-hit integrators and PathTracer(0) work, anything that samples
    an emitter raises the reference's "No Emitter!" (scene.cpp:989)"""
    import torch
    D = scenes.DATA
    sc = psdr.Scene()
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = 4, 8, 0
    sc.opts.width = sc.opts.height = 32
    sc.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    cam.to_world = psdr.Matrix4fD([[1., 0., 0., 208.], [0., 1., 0., 273.], [0., 0., 1., -800.], [0., 0., 0., 1.]])
    sc.add_Sensor(cam)
    sc.add_BSDF(psdr.DiffuseBSDF([0.5, 0.5, 0.5]), "cat")
    P = psdr.FloatD(0.).requires_grad_()
    sc.add_Mesh(os.path.join(D, "cbox_smallbox.obj"), psdr.Matrix4fC(np.eye(4, dtype=np.float32).tolist()), "cat", None)
    sc.param_map["Mesh[0]"].set_transform(psdr.Matrix4fD([[1., 0., 0., P * 100.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    sc.configure()
    sc.configure([0])
    sil = psdr.FieldExtractionIntegrator("silhouette").renderD(sc, 0, seed=1)
    assert 0.02 < float(sil.detach().mean()) < 0.5
    d = psdr.forward_grad(sil, P)
    assert float(d.abs().sum()) > 0                       # the silhouette moves with the box (primary-edge term)
    sil.sum().backward()
    assert abs(float(P.grad) - float(d.sum())) < 1e-3 * max(1.0, abs(float(d.sum())))
    assert float(psdr.PathTracer(0).renderC(sc, 0, seed=1).abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="No Emitter"):
        psdr.PathTracer(1).renderC(sc, 0, seed=1)


def test_camera_pose_reverse_mode(psdr, orc):
    """loss.backward() w.r.t. the camera pose (a translation and a rotation angle inside sensor.to_world): equal to
    <w, forward_grad> for both parameters - the interior term through psdr_grads.g_camera, the primary-edge term through the
    projected edge endpoints"""
    import torch
    from psdr_jit_amd import Matrix4fC, Matrix4fD
    D = scenes.DATA
    tx = psdr.FloatD(0.).requires_grad_()
    ang = psdr.FloatD(0.).requires_grad_()
    sc = psdr.Scene()
    sc.opts.spp = sc.opts.sppe = sc.opts.sppse = 8
    sc.opts.width = sc.opts.height = 40
    sc.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    c, s = torch.cos(ang * 0.1), torch.sin(ang * 0.1)
    cam.to_world = Matrix4fD([[c, 0., s, 208. + tx * 50.], [0., 1., 0., 273.], [-s, 0., c, -800.], [0., 0., 0., 1.]])
    sc.add_Sensor(cam)
    sc.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light")
    sc.add_BSDF(psdr.DiffuseBSDF(), "cat")
    sc.add_BSDF(psdr.DiffuseBSDF([0.95, 0.95, 0.95]), "white")
    I = np.eye(4, dtype=np.float32).tolist()
    sc.add_Mesh(os.path.join(D, "cbox_luminaire.obj"), Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., -0.5], [0., 0., 1., 0.], [0., 0., 0., 1.]]), "light", psdr.AreaLight([20.0, 20.0, 8.0]))
    for f, b in (("cbox_smallbox", "cat"), ("cbox_largebox", "cat"), ("cbox_floor", "white"), ("cbox_back", "white")):
        sc.add_Mesh(os.path.join(D, f + ".obj"), Matrix4fC(I), b, None)
    sc.configure()
    sc.configure([0])
    img = psdr.PathTracer(2).renderD(sc, 0, seed=7)
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    want_t = float((psdr.forward_grad(img, tx) * w).sum())
    want_a = float((psdr.forward_grad(img, ang) * w).sum())
    (img * w).sum().backward()
    assert abs(want_t) > 1e-2 and abs(want_a) > 1e-2
    assert abs(float(tx.grad) - want_t) < 2e-3 * max(1.0, abs(want_t)), (float(tx.grad), want_t)
    assert abs(float(ang.grad) - want_a) < 2e-3 * max(1.0, abs(want_a)), (float(ang.grad), want_a)


def test_orthographic_camera_reverse_mode(psdr, orc):
    """loss.backward() through an OrthographicCamera with primary edges (the host chain rule projects the edge end points with the
    orthographic camera_to_sample, chain.py): a box translation and a camera translation equal <w, forward_grad>"""
    import torch
    from psdr_jit_amd import Matrix4fC, Matrix4fD
    D = scenes.DATA
    tx = psdr.FloatD(0.).requires_grad_()
    bx = psdr.FloatD(0.).requires_grad_()
    sc = psdr.Scene()
    sc.opts.spp = sc.opts.sppe = sc.opts.sppse = 8
    sc.opts.width = sc.opts.height = 40
    sc.opts.log_level = 0
    cam = psdr.OrthographicCamera(0.1, 100.0)
    cam.to_world = Matrix4fD([[1., 0., 0., tx * 0.2], [0., 1., 0., 0.], [0., 0., 1., -5.], [0., 0., 0., 1.]])
    sc.add_Sensor(cam)
    sc.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light")
    sc.add_BSDF(psdr.DiffuseBSDF(), "cat")
    sc.add_BSDF(psdr.DiffuseBSDF([0.95, 0.95, 0.95]), "white")
    S = (scenes._scale_m(1.0 / 300.0) @ scenes.translate(-278.0, -273.0, -280.0)).astype(np.float32)
    sc.add_Mesh(os.path.join(D, "cbox_luminaire.obj"), Matrix4fC((S @ scenes.translate(0.0, -0.5, 0.0)).tolist()), "light", psdr.AreaLight([20.0, 20.0, 8.0]))
    for f, b in (("cbox_smallbox", "cat"), ("cbox_largebox", "cat"), ("cbox_floor", "white"), ("cbox_back", "white")):
        sc.add_Mesh(os.path.join(D, f + ".obj"), Matrix4fC(S.tolist()), b, None)
    sc.param_map["Mesh[1]"].set_transform(Matrix4fD([[1., 0., 0., bx * 0.3], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    sc.configure()
    sc.configure([0])
    assert sc.param_map["Sensor[0]"].orthographic
    img = psdr.PathTracer(2).renderD(sc, 0, seed=7)
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    want_t = float((psdr.forward_grad(img, tx) * w).sum())
    want_b = float((psdr.forward_grad(img, bx) * w).sum())
    (img * w).sum().backward()
    assert abs(want_t) > 1e-3 and abs(want_b) > 1e-3
    assert abs(float(tx.grad) - want_t) < 2e-3 * max(1.0, abs(want_t)), (float(tx.grad), want_t)
    assert abs(float(bx.grad) - want_b) < 2e-3 * max(1.0, abs(want_b)), (float(bx.grad), want_b)


def test_unit_ray_intersect(psdr, orc):
    """Scene.unit_ray_intersect (reference psdr.cpp:404): the Intersection record of a batch of rays against the oracle's closest
    hits (same triangle, t) and the snapshot's triangle rows (p, geometric normal, shading frame, uv, wi)"""
    import torch
    for spec in (scenes.cbox_scene(16, 16, 1, 0, 0, param=None), scenes.textured_scene(16, 16, 1, 0, 0, env=False)):
        sc = product.build_scene(spec)
        ref = orc.OracleScene(spec, [0])
        rng = np.random.default_rng(4)
        n = 4096
        o = np.tile(np.array([[278.0, 273.0, -300.0]], np.float32), (n, 1)) + rng.normal(0, 20, (n, 3)).astype(np.float32)
        d = rng.normal(0, 1, (n, 3)).astype(np.float32); d[:, 2] = np.abs(d[:, 2]) + 0.3
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        its = sc.unit_ray_intersect(psdr.RayC(torch.from_numpy(o), torch.from_numpy(d)))
        tri, uv, t = ref.trace(o, d)
        valid = its.is_valid().cpu().numpy()
        assert np.array_equal(valid, tri >= 0) and valid.sum() > 200
        rows = np.asarray(sc._snapshot()["triangles"], np.float64)
        tri_mesh = np.concatenate([np.full(len(m.faces), i) for i, m in enumerate(spec.meshes)])
        k = tri[valid]
        p0, e1, e2, fn = rows[k, 0:3], rows[k, 3:6], rows[k, 6:9], rows[k, 18:21]
        p = p0 + uv[valid, :1] * e1 + uv[valid, 1:] * e2
        assert np.allclose(its.p.cpu().numpy()[valid], p, rtol=1e-5, atol=1e-3)
        assert np.allclose(its.t.cpu().numpy()[valid], t[valid], rtol=1e-5, atol=1e-3)
        assert np.allclose(its.n.cpu().numpy()[valid], fn, atol=1e-6)
        assert np.array_equal(its.shape.cpu().numpy()[valid], tri_mesh[k]) and np.all(its.shape.cpu().numpy()[~valid] == -1)
        s_, t_, n_ = (x.cpu().numpy()[valid].astype(np.float64) for x in (its.sh_frame.s, its.sh_frame.t, its.sh_frame.n))
        assert np.allclose((s_ * t_).sum(1), 0, atol=1e-5) and np.allclose((s_ * n_).sum(1), 0, atol=1e-5) and np.allclose((n_ * n_).sum(1), 1, atol=1e-5)
        wi = its.wi.cpu().numpy()[valid]
        want_wi = np.stack([(-d[valid] * s_).sum(1), (-d[valid] * t_).sum(1), (-d[valid] * n_).sum(1)], axis=1)
        assert np.allclose(wi, want_wi, atol=1e-5)
        assert np.all(its.J.cpu().numpy()[valid] == 1.0)
        # texture coordinates: the uv-mapped floor of the textured scene is uv = (x, z) / 560
        if spec.meshes[0].uvs is not None and len(spec.meshes[0].uvs) == 4:
            on_floor = valid & (its.shape.cpu().numpy() == 0)
            if on_floor.any():
                pf = its.p.cpu().numpy()[on_floor]
                assert np.allclose(its.uv.cpu().numpy()[on_floor], pf[:, [0, 2]] / 560.0, atol=1e-4)


def test_sample_direct_projects_first_hits_back_to_their_pixels(psdr, orc):
    """PerspectiveCamera.sample_direct (perspective.cpp:181-197) on the `position` field image: every first-hit point lands in the
    pixel it was seen through; sensor_val = 1 / (dist^2 cos^3 film area)"""
    import torch
    spec = scenes.cbox_scene(32, 32, 1, 0, 0, param=None)
    sc = product.build_scene(spec)
    pos = psdr.FieldExtractionIntegrator("position").renderC(sc, 0, seed=1)
    mask = psdr.FieldExtractionIntegrator("silhouette").renderC(sc, 0, seed=1)[:, 0] > 0
    cam = sc.param_map["Sensor[0]"]
    sds = cam.sample_direct(pos[mask])
    want = torch.arange(32 * 32, device=pos.device)[mask]
    assert bool(sds.is_valid.all()) and torch.equal(sds.pixel_idx.to(torch.int64), want)
    d = pos[mask] - torch.tensor([208.0, 273.0, -800.0], device=pos.device)
    dist2 = (d * d).sum(1)
    cos_t = d[:, 2] / dist2.sqrt()
    a = 2 * np.tan(np.radians(30.0))              # film width at unit distance, aspect 1
    assert torch.allclose(sds.sensor_val, 1.0 / (dist2 * cos_t ** 3 * a * a), rtol=1e-4)
    assert not bool(cam.sample_direct(torch.tensor([[5000.0, 273.0, 0.0]])).is_valid.any())        # far outside the field of view


@pytest.mark.parametrize("param", ["diffuse", "specular", "roughness", "ball_x"])
def test_microfacet_per_vertex_bsdf(psdr, orc, param):
    """psdr.MicrofacetBSDFPerVertex (reference microfacet_pv.cpp: parameters interpolated over the hit triangle's vertices with the
    barycentrics, which are differentiable at the first hit) against the oracle"""
    spec = scenes.pervertex_scene(48, 48, 8, 8, 8, param=param)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(3)
    c = integ.renderC(sc, 0, seed=3).cpu().numpy()
    assert product.rel_l2(c, ref.render_c(max_depth=3, seed=3)) < 1e-3
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=6)
    wimg, wd = ref.render_d(max_depth=3, seeds=(6, 6, 6))
    assert product.rel_l2(img.cpu().numpy(), wimg) < 1e-3
    assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < 1e-3
    b = sc.param_map["BSDF[id=pv]"]
    assert type(b).__name__ == "MicrofacetBSDFPerVertex" and tuple(b.roughness.shape) == (len(spec.meshes[1].vertices),)


def test_microfacet_per_vertex_reverse_mode(psdr, orc):
    """loss.backward() into per-vertex BSDF values (per-vertex material optimisation): <w, J v> == <J^T w, v> against forward mode"""
    import torch
    from psdr_jit_amd import Matrix4fC, Matrix4fD
    rng = np.random.default_rng(13)
    spec = scenes.pervertex_scene(40, 40, 8, 0, 0)
    b = spec.bsdfs[5]
    sp = torch.tensor(b.pv_specular, requires_grad=True)
    df = torch.tensor(b.pv_diffuse, requires_grad=True)
    rg = torch.tensor(b.pv_roughness, requires_grad=True)
    D = scenes.DATA
    sc = psdr.Scene()
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = 8, 0, 0
    sc.opts.width = sc.opts.height = 40
    sc.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    cam.to_world = Matrix4fD([[1., 0., 0., 208.], [0., 1., 0., 273.], [0., 0., 1., -800.], [0., 0., 0., 1.]])
    sc.add_Sensor(cam)
    sc.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light")
    sc.add_BSDF(psdr.MicrofacetBSDFPerVertex(sp, df, rg), "pv")
    sc.add_BSDF(psdr.DiffuseBSDF([0.95, 0.95, 0.95]), "white")
    I = np.eye(4, dtype=np.float32).tolist()
    sc.add_Mesh(os.path.join(D, "cbox_luminaire.obj"), Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., -0.5], [0., 0., 1., 0.], [0., 0., 0., 1.]]), "light", psdr.AreaLight([20.0, 20.0, 8.0]))
    sc.add_Mesh(os.path.join(D, "cbox_smallball.obj"), Matrix4fC(I), "pv", None)
    for f in ("cbox_floor", "cbox_back"):
        sc.add_Mesh(os.path.join(D, f + ".obj"), Matrix4fC(I), "white", None)
    sc.configure()
    sc.configure([0])
    img = psdr.PathTracer(2).renderD(sc, 0, seed=5)
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    fwd = {}
    for name, t in (("specular", sp), ("diffuse", df), ("roughness", rg)):
        v = torch.tensor(rng.standard_normal(tuple(t.shape)).astype(np.float32))
        fwd[name] = (t, v, float((psdr.forward_grad(img, t, direction=v) * w).sum()))
    (img * w).sum().backward()
    for name, (t, v, want) in fwd.items():
        got = float((t.grad * v).sum())
        assert abs(want) > 1e-4 and abs(got - want) < 2e-3 * max(1.0, abs(want)), (name, got, want)


@pytest.mark.parametrize("kind", ["microfacet", "roughconductor", "roughdielectric"])
def test_ggx_constant_parameters_reverse_mode(psdr, orc, kind):
    """loss.backward() into the constant parameters of the GGX BSDFs (roughness, specular colour, alpha, eta, k ...):
    <w, J v> == <J^T w, v> against forward mode for every parameter"""
    import torch
    from psdr_jit_amd import Matrix4fC, Matrix4fD
    D = scenes.DATA
    if kind == "microfacet":
        leaves = {"specularReflectance": torch.tensor([0.6, 0.5, 0.4], requires_grad=True), "diffuseReflectance": torch.tensor([0.3, 0.2, 0.1], requires_grad=True),
                  "roughness": psdr.FloatD(0.35).requires_grad_()}
        bsdf = psdr.MicrofacetBSDF(leaves["specularReflectance"], leaves["diffuseReflectance"], leaves["roughness"])
    elif kind == "roughconductor":
        leaves = {"alpha_u": psdr.FloatD(0.2).requires_grad_(), "alpha_v": psdr.FloatD(0.3).requires_grad_(), "eta": torch.tensor([0.2, 0.9, 1.1], requires_grad=True),
                  "k": torch.tensor([3.9, 2.4, 2.1], requires_grad=True), "specular_reflectance": torch.tensor([1.0, 0.9, 0.8], requires_grad=True)}
        bsdf = psdr.RoughConductorBSDF(leaves["alpha_u"], leaves["alpha_v"], leaves["eta"], leaves["k"], leaves["specular_reflectance"])
    else:
        leaves = {"alpha_u": psdr.FloatD(0.15).requires_grad_(), "alpha_v": psdr.FloatD(0.15).requires_grad_(), "eta": psdr.FloatD(1.5).requires_grad_()}
        bsdf = psdr.RoughDielectricBSDF(1.5, 1.0)
        bsdf.alpha_u, bsdf.alpha_v, bsdf.eta = leaves["alpha_u"], leaves["alpha_v"], leaves["eta"]
    sc = psdr.Scene()
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = 8, 0, 0
    sc.opts.width = sc.opts.height = 40
    sc.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    cam.to_world = Matrix4fD([[1., 0., 0., 208.], [0., 1., 0., 273.], [0., 0., 1., -800.], [0., 0., 0., 1.]])
    sc.add_Sensor(cam)
    sc.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light")
    sc.add_BSDF(bsdf, "m")
    sc.add_BSDF(psdr.DiffuseBSDF([0.95, 0.95, 0.95]), "white")
    I = np.eye(4, dtype=np.float32).tolist()
    sc.add_Mesh(os.path.join(D, "cbox_luminaire.obj"), Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., -0.5], [0., 0., 1., 0.], [0., 0., 0., 1.]]), "light", psdr.AreaLight([20.0, 20.0, 8.0]))
    for f, b in (("cbox_smallbox", "m"), ("cbox_largebox", "m"), ("cbox_floor", "white"), ("cbox_back", "white")):
        sc.add_Mesh(os.path.join(D, f + ".obj"), Matrix4fC(I), b, None)
    sc.configure()
    sc.configure([0])
    img = psdr.PathTracer(3).renderD(sc, 0, seed=4)
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    rng = np.random.default_rng(5)
    fwd = {}
    for name, t in leaves.items():
        v = torch.tensor(rng.standard_normal(tuple(t.shape)).astype(np.float32)) if t.dim() > 0 else torch.tensor(1.0)
        fwd[name] = (v, float((psdr.forward_grad(img, t, direction=v) * w).sum()))
    (img * w).sum().backward()
    for name, t in leaves.items():
        v, want = fwd[name]
        assert t.grad is not None, name
        got = float((t.grad * v).sum())
        assert abs(want) > 1e-4 and abs(got - want) < 3e-3 * max(1.0, abs(want)), (kind, name, got, want)


@pytest.mark.parametrize("nested,nmap,param", [("microfacet", "bumpy", "nmap"), ("microfacet", "bumpy", "box_x"), ("diffuse", "tilted", "nmap"),
                                               ("roughconductor", "bumpy", None), ("microfacet", "flat", "nested")])
def test_normalmap_bsdf(psdr, orc, nested, nmap, param):
    """psdr.NormalMapBSDF (reference normalmap.cpp: the map's normal and a tangent facet, the nested BSDF evaluated / sampled in the
    perturbed frame) against the oracle: image, texel / constant tangents of the map, of the nested BSDF and of the geometry"""
    spec = scenes.normalmap_scene(48, 48, 8, 8, 8, param=param, nested=nested, nmap=nmap)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(2)
    c = integ.renderC(sc, 0, seed=3).cpu().numpy()
    assert np.isfinite(c).all() and product.rel_l2(c, ref.render_c(max_depth=2, seed=3)) < 1e-3
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=6)
    wimg, wd = ref.render_d(max_depth=2, seeds=(6, 6, 6))
    assert product.rel_l2(img.cpu().numpy(), wimg) < 1e-3
    if param is not None:
        assert np.abs(wd).max() > 0 and product.rel_l2(dimg.cpu().numpy(), wd) < 1e-3
    live = sc.param_map["BSDF[id=tex]"]
    assert type(live).__name__ == "NormalMapBSDF" and type(live.nested_bsdf).__name__ in ("MicrofacetBSDF", "DiffuseBSDF", "RoughConductorBSDF")


def test_normalmap_api_and_xml(psdr, orc):
    """Scene.add_normalmap_BSDF(NormalMapBSDF, MicrofacetBSDF, name) and the XML form (scene_loader.cpp:373-430)"""
    nm = psdr.NormalMapBSDF([0.6, 0.45, 0.9])
    mf = psdr.MicrofacetBSDF([0.7, 0.6, 0.5], [0.3, 0.25, 0.2], 0.35)
    sc = psdr.Scene()
    sc.opts.log_level = 0
    sc.add_normalmap_BSDF(nm, mf, "n")
    live = sc.param_map["BSDF[id=n]"]
    assert np.allclose(np.asarray(live.normal_map), [0.6, 0.45, 0.9]) and abs(float(np.asarray(live.nested_bsdf.roughness)[0]) - 0.35) < 1e-7
    x = psdr.Scene()
    x.opts.log_level = 0
    x.load_string('<scene><bsdf type="normalmap" id="g"><rgb name="normalmap" value="0.5, 0.5, 1.0"/><bsdf type="roughconductor">'
                  '<float name="alpha" value="0.2"/><rgb name="eta" value="0.2, 0.9, 1.1"/><rgb name="k" value="3.9, 2.4, 2.1"/></bsdf></bsdf></scene>', False)
    g = x.param_map["BSDF[id=g]"]
    assert type(g).__name__ == "NormalMapBSDF" and type(g.nested_bsdf).__name__ == "RoughConductorBSDF"
    with pytest.raises(RuntimeError, match="Unsupported normal map nested BSDF"):
        psdr.Scene().load_string('<scene><bsdf type="normalmap" id="g"><rgb name="normalmap" value="0.5"/><bsdf type="normalmap"/></bsdf></scene>', False)


def test_normalmap_and_conductor_maps_reverse_mode(psdr, orc):
    """loss.backward() into a normal map's texels, into the parameters of the BSDF nested in it, and into the eta map of a RoughConductor:
    <w, J v> == <J^T w, v> against forward mode"""
    import torch
    rng = np.random.default_rng(17)
    spec = scenes.normalmap_scene(40, 40, 8, 0, 0)
    nmap = torch.tensor(spec.bsdfs[0].texture, requires_grad=True)
    rough = psdr.FloatD(0.35).requires_grad_()
    diff = torch.tensor([0.3, 0.25, 0.2], requires_grad=True)
    eta = torch.tensor((0.2 + rng.random((4, 5, 3))).astype(np.float32), requires_grad=True)
    sc = psdr.Scene()
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = 8, 0, 0
    sc.opts.width = sc.opts.height = 40
    sc.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    cam.to_world = psdr.Matrix4fD(np.asarray(spec.cameras[0].to_world_raw).tolist())
    sc.add_Sensor(cam)
    nm = psdr.NormalMapBSDF(nmap)
    nm.nested_bsdf = psdr.MicrofacetBSDF([0.7, 0.6, 0.5], diff, rough)
    sc.add_BSDF(nm, "tex")
    sc.add_BSDF(psdr.RoughConductorBSDF(0.25, eta, [3.9, 2.4, 2.1]), "cat")
    sc.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light")
    floor = psdr.Mesh()
    m = spec.meshes[0]
    floor.load_raw(m.vertices, m.faces, m.uvs, m.face_uvs)
    sc.add_Mesh(floor, "tex", None)
    # the box needs uv for its eta map: give it the floor's quad layout per face is overkill - use a second uv-mapped quad instead
    wall = psdr.Mesh()
    v = np.array([[100, 0, 400], [460, 0, 400], [460, 300, 400], [100, 300, 400]], np.float32)
    f = np.array([[0, 2, 1], [0, 3, 2]], np.int32)            # facing the camera (-z)
    wall.load_raw(v, f, m.uvs, f.copy())
    sc.add_Mesh(wall, "cat", None)
    sc.add_Mesh(os.path.join(scenes.DATA, "cbox_luminaire.obj"), psdr.Matrix4fC(scenes.translate(0.0, -100.0, 0.0).tolist()), "light", psdr.AreaLight([20.0, 20.0, 8.0]))
    sc.configure()
    sc.configure([0])
    img = psdr.PathTracer(2).renderD(sc, 0, seed=5)
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    fwd = {}
    for name, t in (("normal map", nmap), ("nested roughness", rough), ("nested diffuse", diff), ("eta map", eta)):
        v_ = torch.tensor(rng.standard_normal(tuple(t.shape)).astype(np.float32)) if t.dim() > 0 else torch.tensor(1.0)
        fwd[name] = (t, v_, float((psdr.forward_grad(img, t, direction=v_) * w).sum()))
    (img * w).sum().backward()
    for name, (t, v_, want) in fwd.items():
        assert t.grad is not None, name
        got = float((t.grad * v_).sum())
        assert abs(want) > 1e-4 and abs(got - want) < 3e-3 * max(1.0, abs(want)), (name, got, want)


def test_batch_render_reverse_mode(psdr, orc):
    """renderD(seed, batch_pix) followed by loss.backward() (crop-wise optimisation, reference integrator.cpp:139-176): the gradient
    of a crop equals the forward derivative of the same crop, and a crop of the full-frame primal is reproduced"""
    import torch
    P = psdr.FloatD(0.).requires_grad_()
    refl = torch.tensor([0.5, 0.5, 0.5], requires_grad=True)
    sc = _readme_scene(psdr, P, res=32, spp=8)
    sc.opts.sppe = sc.opts.sppse = 0
    sc.param_map["BSDF[id=cat]"].reflectance = refl
    sc.configure([0])
    pix = np.arange(32 * 32).reshape(32, 32)[8:24, 4:20].reshape(-1)
    integ = psdr.PathTracer(2)
    img = integ.renderD(sc, 0, seed=3, batch_pix=pix)
    assert tuple(img.shape) == (len(pix), 3)
    full = integ.renderC(sc, 0, seed=3).cpu().numpy()
    # (pixel k of the crop uses the seeds of pixel pix[k] of the full frame but its own lane index: equal in the mean)
    assert abs(img.detach().cpu().numpy().mean() - full[pix].mean()) < 0.1 * full[pix].mean()
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    want_P = float((psdr.forward_grad(img, P) * w).sum())
    want_r = float((psdr.forward_grad(img, refl, direction=torch.tensor([1.0, -0.5, 0.25])) * w).sum())
    (img * w).sum().backward()
    assert abs(want_P) > 1e-3 and abs(float(P.grad) - want_P) < 2e-3 * max(1.0, abs(want_P)), (float(P.grad), want_P)
    got_r = float((refl.grad * torch.tensor([1.0, -0.5, 0.25])).sum())
    assert abs(got_r - want_r) < 2e-3 * max(1.0, abs(want_r)), (got_r, want_r)


def test_collocated_intensity_is_differentiable(psdr, orc):
    """CollocatedIntegrator.m_intensity (FloatD in the reference, collocated.h): forward and reverse derivative = image / intensity"""
    import torch
    P = psdr.FloatD(0.).requires_grad_()
    sc = _readme_scene(psdr, P, res=32, spp=4)
    col = psdr.CollocatedIntegrator(2e5)
    inten = psdr.FloatD(2e5).requires_grad_()
    col.m_intensity = inten
    img = col.renderD(sc, 0, seed=2)
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    d = psdr.forward_grad(img, inten)
    assert float(img.detach().abs().sum()) > 0 and torch.allclose(d, img.detach() / 2e5, rtol=1e-4, atol=1e-12)
    (img * w).sum().backward()
    want = float((img.detach() * w).sum()) / 2e5
    assert abs(float(inten.grad) - want) < 1e-4 * abs(want)


def test_reverse_mode_deep_paths_and_many_lookups(psdr, orc):
    """loss.backward() at max_depth 6 (the per-lane records are sized from the depth) in a scene whose paths make many lookups
    (textured floor under an environment map): gradients of the geometry, the texels and the radiance equal forward mode"""
    import torch
    rng = np.random.default_rng(23)
    spec = scenes.textured_scene(32, 32, 8, 0, 0, texture=scenes.checker_texture(8, 8), env=True)
    env0 = scenes.synthetic_envmap(32, 16)
    rad = torch.tensor(env0, requires_grad=True)
    tex = torch.tensor(spec.bsdfs[0].texture, requires_grad=True)
    P = psdr.FloatD(0.).requires_grad_()
    sc = psdr.Scene()
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = 8, 0, 0
    sc.opts.width = sc.opts.height = 32
    sc.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    cam.to_world = psdr.Matrix4fD(np.asarray(spec.cameras[0].to_world_raw).tolist())
    sc.add_Sensor(cam)
    sc.add_BSDF(psdr.DiffuseBSDF(tex), "tex")
    sc.add_BSDF(psdr.DiffuseBSDF([0.8, 0.7, 0.6]), "cat")
    floor = psdr.Mesh()
    m = spec.meshes[0]
    floor.load_raw(m.vertices, m.faces, m.uvs, m.face_uvs)
    sc.add_Mesh(floor, "tex", None)
    sc.add_Mesh(os.path.join(scenes.DATA, "cbox_smallbox.obj"), psdr.Matrix4fC(np.eye(4, dtype=np.float32).tolist()), "cat", None)
    sc.add_EnvironmentMap(psdr.EnvironmentMap(rad))
    sc.param_map["Mesh[1]"].set_transform(psdr.Matrix4fD([[1., 0., 0., P * 100.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    sc.configure()
    sc.configure([0])
    img = psdr.PathTracer(6).renderD(sc, 0, seed=5)
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    v_rad = torch.tensor(rng.standard_normal(env0.shape).astype(np.float32))
    v_tex = torch.tensor(rng.standard_normal(tuple(tex.shape)).astype(np.float32))
    want = {"P": float((psdr.forward_grad(img, P) * w).sum()), "rad": float((psdr.forward_grad(img, rad, direction=v_rad) * w).sum()),
            "tex": float((psdr.forward_grad(img, tex, direction=v_tex) * w).sum())}
    (img * w).sum().backward()
    got = {"P": float(P.grad), "rad": float((rad.grad * v_rad).sum()), "tex": float((tex.grad * v_tex).sum())}
    for k in want:
        assert abs(want[k]) > 1e-4 and abs(got[k] - want[k]) < 3e-3 * max(1.0, abs(want[k])), (k, got[k], want[k])


def test_reverse_mode_through_vertex_normals(psdr, orc):
    """a rotation and a per-vertex displacement change the (smooth) vertex normals: the per-hit blended-normal probes of the
    interior adjoint must reproduce forward mode"""
    import torch
    from psdr_jit_amd import Matrix4fC, Matrix4fD
    D = scenes.DATA
    ang = psdr.FloatD(0.).requires_grad_()
    sc = psdr.Scene()
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = 8, 8, 8
    sc.opts.width = sc.opts.height = 40
    sc.opts.log_level = 0
    cam = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    cam.to_world = Matrix4fD([[1., 0., 0., 278.], [0., 1., 0., 273.], [0., 0., 1., -500.], [0., 0., 0., 1.]])
    sc.add_Sensor(cam)
    sc.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light")
    sc.add_BSDF(psdr.DiffuseBSDF([0.9, 0.6, 0.1]), "ball")
    sc.add_BSDF(psdr.DiffuseBSDF([0.95, 0.95, 0.95]), "white")
    I = np.eye(4, dtype=np.float32).tolist()
    sc.add_Mesh(os.path.join(D, "cbox_luminaire.obj"), Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., -0.5], [0., 0., 1., 0.], [0., 0., 0., 1.]]), "light", psdr.AreaLight([20.0, 20.0, 8.0]))
    ball = psdr.Mesh()
    ball.load(os.path.join(D, "cbox_smallball.obj"))
    V0 = torch.tensor(np.asarray(ball.vertex_positions, np.float32))
    disp = torch.zeros_like(V0, requires_grad=True)
    ball.vertex_positions = V0 + disp
    sc.add_Mesh(ball, "ball", None)
    for f in ("cbox_floor", "cbox_back"):
        sc.add_Mesh(os.path.join(D, f + ".obj"), Matrix4fC(I), "white", None)
    c, s = torch.cos(ang), torch.sin(ang)
    # rotate the (smooth-shaded) ball about a vertical axis through (185, *, 169): its vertex normals turn with it
    sc.param_map["Mesh[1]"].set_transform(Matrix4fD([[c, 0., s, 185. - 185. * c - 169. * s], [0., 1., 0., 0.], [-s, 0., c, 169. + 185. * s - 169. * c], [0., 0., 0., 1.]]))
    sc.configure()
    sc.configure([0])
    img = psdr.PathTracer(2).renderD(sc, 0, seed=9)
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    rng = np.random.default_rng(3)
    v = torch.tensor(rng.standard_normal(tuple(V0.shape)).astype(np.float32))
    want_d = float((psdr.forward_grad(img, disp, direction=v) * w).sum())
    want_a = float((psdr.forward_grad(img, ang) * w).sum())
    (img * w).sum().backward()
    got_d, got_a = float((disp.grad * v).sum()), float(ang.grad)
    assert abs(want_d) > 1e-3 and abs(got_d - want_d) < 3e-3 * max(1.0, abs(want_d)), (got_d, want_d)
    assert abs(want_a) > 1e-3 and abs(got_a - want_a) < 3e-3 * max(1.0, abs(want_a)), (got_a, want_a)
