"""CPU-side tests of the product's host logic (no GPU, no compute calls through the C ABI):
  * libpsdr_hip.so loads and exports every symbol include/psdr_hip.h declares;
  * the host C++ scene model (OBJ loader, Mesh/Camera/Scene::configure, edge lists, CDFs, forward
    tangents) reproduces the oracle's configured snapshot BIT-EXACTLY;
  * reference API semantics: param_map keys, error behaviour, sampler bookkeeping.
"""
import os
import re

import numpy as np
import pytest

import product
import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def psdr():
    import __graft_entry__
    __graft_entry__.build()
    import psdr_jit_amd
    return psdr_jit_amd


def test_cabi_exports_every_declared_symbol(psdr):
    from psdr_jit_amd import cabi
    hdr = open(os.path.join(ROOT, "include", "psdr_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(psdr_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 19
    L = cabi.lib()
    for name in declared:
        assert hasattr(L, name), "libpsdr_hip.so does not export %s" % name
    assert sorted(cabi.SYMBOLS) == declared
    assert L.psdr_hip_abi_version() == 16
    # host-side sampler building block is bit-exact with the oracle / golden table
    import json
    with open(os.path.join(ROOT, "tests", "golden", "tea64.json")) as fh:
        for a, b, want in json.load(fh):
            assert int(L.psdr_hip_tea64(int(a), int(b))) == int(want)


@pytest.mark.parametrize("name", ["cbox", "sphere"])
@pytest.mark.parametrize("param", ["light_x", "camera_x"])
def test_host_snapshot_matches_oracle(psdr, orc, name, param):
    if name == "cbox":
        spec = scenes.cbox_scene(96, 64, 4, 4, 4, param=param)
    else:
        spec = scenes.sphere_scene(96, 64, 1, 1, 1)
        if param == "camera_x":
            pytest.skip("one tangent configuration is enough for the sphere scene")
    ref = orc.OracleScene(spec, [0])
    sc = product.build_scene(spec, host_only=True)
    snap = sc._snapshot()
    assert np.array_equal(snap["triangles"], ref.triangle_info(False)[:, :22])
    assert np.array_equal(snap["d_triangles"], ref.triangle_info(True)[:, :22])
    assert np.array_equal(snap["sec_edges"], ref.sec_edges(False))
    assert np.array_equal(snap["d_sec_edges"][:, :6], ref.sec_edges(True)[:, :6])
    cam = sc.param_map["Sensor[0]"]
    assert np.array_equal(cam._primary_edges(False), ref.primary_edges(0, False))
    assert np.array_equal(cam._primary_edges(True), ref.primary_edges(0, True))
    for mi in range(sc.num_meshes):
        assert np.array_equal(sc.param_map["Mesh[%d]" % mi].edge_indices(), ref.mesh_edges(mi))
    assert abs(sc.param_map["Emitter[0]"].sampling_weight - ref.emitter_sampling_weight(0)) == 0.0
    if name == "sphere":
        assert snap["sec_edges"].shape[0] == 990 and cam._primary_edges(False).shape[0] == 79   # Forward_AD.ipynb:139-140


def _snap_equal(a, b):
    sa, sb = a._snapshot(), b._snapshot()
    assert sorted(sa.keys()) == sorted(sb.keys())
    for k in sa:
        if k in ("bsdf_rows", "env_reso"):
            assert list(sa[k]) == list(sb[k]), k
        elif k == "env_cell_sum":
            assert float(sa[k]) == float(sb[k])
        else:
            assert np.array_equal(np.asarray(sa[k]), np.asarray(sb[k])), k
    for sid in range(a.num_sensors):
        ca, cb = a.param_map["Sensor[%d]" % sid], b.param_map["Sensor[%d]" % sid]
        for tg in (False, True):
            assert np.array_equal(ca._primary_edges(tg), cb._primary_edges(tg)), ("primary edges", sid, tg)
        assert np.array_equal(np.asarray(ca._primary_edge_ids()), np.asarray(cb._primary_edge_ids()))


@pytest.mark.parametrize("name", ["cbox", "sphere", "envmap"])
def test_incremental_configure_equals_a_fresh_scene(psdr, name):
    """configure_host() on a scene that has been configured before rewrites only the snapshot rows whose inputs changed (Mesh::configure skips a mesh whose inputs are
    those of its previous run, sensors and edge lists follow the meshes' versions): after every kind of change - a tangent, a translation of one mesh, raw vertices,
    a colour, the camera, nothing - the snapshot equals, bit for bit, the one of a scene built from scratch in that state"""
    make = {"cbox": lambda: scenes.cbox_scene(24, 24, 4, 4, 4, param="light_x"), "sphere": lambda: scenes.sphere_scene(24, 24, 4, 4, 4),
            "envmap": lambda: scenes.envmap_scene(24, 24, 4, 4, 4, param="box_x", area_light=True)}[name]
    spec = make()
    sc = product.build_scene(spec, host_only=True)
    z44 = np.zeros((4, 4), np.float32)

    def fresh():
        f = product.build_scene(spec, host_only=True)
        _snap_equal(sc, f)

    fresh()
    sc._configure_host([0]); fresh()                                   # nothing changed
    m0 = spec.meshes[0]
    # 1. a tangent only
    d = z44.copy(); d[1, 3] = 2.5
    m0.d_to_world_left = d
    sc.param_map["Mesh[0]"]._set("to_world_left", np.asarray(m0.to_world_left, np.float32), d)
    sc._configure_host([0]); fresh()
    # 2. the mesh moves
    t = np.asarray(m0.to_world_left, np.float32).copy(); t[0, 3] += 7.0; t[2, 3] -= 3.0
    m0.to_world_left = t
    sc.param_map["Mesh[0]"]._set("to_world_left", t, d)
    sc._configure_host([0]); fresh()
    # 3. raw vertices of another mesh (and their tangents)
    m1 = spec.meshes[1]
    v = np.asarray(sc.param_map["Mesh[1]"]._get("vertex_positions", False), np.float32).copy()
    v[:, 0] = v[:, 0].mean() + 0.9 * (v[:, 0] - v[:, 0].mean())        # (inside the scene box of the first configure: the environment map's bounding cube is added once)
    dv = np.zeros_like(v); dv[:, 0] = 1.0
    sc.param_map["Mesh[1]"]._set("vertex_positions", v, dv)
    m1.vertices = v; m1.d_vertices = dv; m1.path = None               # (a spec with a path would re-read the file)
    sc._configure_host([0]); fresh()
    m1.d_vertices = None
    sc.param_map["Mesh[1]"]._set("vertex_positions", v, np.zeros_like(v))
    sc._configure_host([0]); fresh()
    # 4. a colour
    spec.bsdfs[1].reflectance = (0.25, 0.5, 0.75); spec.bsdfs[1].d_reflectance = (0.0, 1.0, 0.0)
    sc.param_map["BSDF[1]"]._set("reflectance", np.asarray([0.25, 0.5, 0.75], np.float32), np.asarray([0.0, 1.0, 0.0], np.float32))
    sc._configure_host([0]); fresh()
    # 5. the camera
    c = spec.cameras[0]
    tw = np.asarray(c.to_world_raw, np.float32).copy(); tw[0, 3] += 11.0
    c.to_world_raw = tw
    sc.param_map["Sensor[0]"]._set("to_world", tw, np.asarray(c.d_to_world_raw, np.float32))
    sc._configure_host([0]); fresh()
    # 6. back to no tangent on the first mesh
    m0.d_to_world_left = z44
    sc.param_map["Mesh[0]"]._set("to_world_left", t, z44)
    sc._configure_host([0]); fresh()


def test_enable_edges_switched_on_between_two_configures(psdr):
    """a mesh loaded with enable_edges = False has no edge list (Mesh::load_raw builds one only when the flag is set); switched on after a configure() - nothing else
    changed, so Mesh::configure takes its same-inputs return - it must still get its edges: the snapshot then equals that of a scene whose mesh had them from the start
    (round 5's early return sat in front of build_edges: the sensor's reconfigure then tripped on an empty edge list, or the mesh silently had no secondary edges)"""
    spec = scenes.cbox_scene(24, 24, 4, 4, 4, param="light_x")
    want = product.build_scene(spec, host_only=True)
    spec.meshes[1].enable_edges = False
    sc = product.build_scene(spec, host_only=True)
    assert np.asarray(sc._snapshot()["sec_edges"]).shape[0] < np.asarray(want._snapshot()["sec_edges"]).shape[0]
    sc.param_map["Mesh[1]"].enable_edges = True
    sc._configure_host([0])
    spec.meshes[1].enable_edges = True
    _snap_equal(sc, want)


@pytest.mark.parametrize("name", ["cbox", "sphere"])
def test_native_chain_rule_equals_the_autograd_restatement(psdr, name):
    """Scene::chain_geometry (host C++: from the adjoints of the snapshot's triangle / edge rows to mesh transforms, raw vertices and the camera pose) against
    psdr_jit_amd/chain.py::snapshot_tensors differentiated by torch.autograd, on random row adjoints"""
    import torch
    from psdr_jit_amd import chain
    spec = scenes.cbox_scene(24, 24, 4, 4, 4, param="light_x") if name == "cbox" else scenes.sphere_scene(24, 24, 4, 4, 4)
    sc = product.build_scene(spec, host_only=True)
    rng = np.random.default_rng(3)
    cam = sc.param_map["Sensor[0]"]
    n_tri, n_sec = (int(x) for x in sc._snapshot_counts())
    n_prim = np.asarray(cam._primary_edge_ids()).reshape(-1, 3).shape[0]
    assert n_prim > 0 and n_sec > 0
    g_tri, g_sec, g_prim = (rng.normal(size=s).astype(np.float32) for s in ((n_tri, 22), (n_sec, 6), (n_prim, 4)))
    g_cam = rng.normal(size=(4, 4)); g_cam[3] = 0.0
    meshes = [sc.param_map["Mesh[%d]" % i] for i in (0, 1, sc.num_meshes - 1)]
    wanted = [(m, n) for m in meshes for n in ("vertex_positions", "to_world_left", "to_world", "to_world_right")] + [(cam, n) for n in ("to_world_left", "to_world", "to_world_right")]
    # give the factors values other than the identity, so that their order matters
    for k, m in enumerate(meshes):
        r = np.eye(4, dtype=np.float32); r[0, 0], r[0, 2], r[2, 0], r[2, 2] = np.cos(0.1 * (k + 1)), np.sin(0.1 * (k + 1)), -np.sin(0.1 * (k + 1)), np.cos(0.1 * (k + 1))
        m._set("to_world_right", r, np.zeros((4, 4), np.float32))
        t = np.asarray(m._get("to_world_left", False), np.float32).copy(); t[1, 3] += 2.0 + k
        m._set("to_world_left", t, np.zeros((4, 4), np.float32))
    sc._configure_host([0])
    n_prim = np.asarray(cam._primary_edge_ids()).reshape(-1, 3).shape[0]
    g_prim = rng.normal(size=(n_prim, 4)).astype(np.float32)
    got = chain.native_geometry_grads(sc, 0, wanted, g_tri, g_sec, g_prim, g_cam)
    leaves = {}

    def leaf_of(obj, pname):
        key = (id(obj), pname)
        if key in {(id(o), n) for o, n in wanted}:
            if key not in leaves:
                leaves[key] = torch.as_tensor(np.asarray(obj._get(pname, False), dtype=np.float64)).clone().requires_grad_(True)
            return leaves[key]
        return torch.as_tensor(np.asarray(obj._get(pname, False), dtype=np.float64))
    tri, sec, prim, refl, rad, cam_tw = chain.snapshot_tensors(sc, 0, leaf_of)
    outs = [tri, sec, prim, cam_tw]
    gos = [torch.as_tensor(g_tri, dtype=torch.float64), torch.as_tensor(g_sec, dtype=torch.float64), torch.as_tensor(g_prim, dtype=torch.float64), torch.as_tensor(g_cam)]
    keys = list(leaves)
    res = torch.autograd.grad(outs, [leaves[k] for k in keys], gos, allow_unused=True)
    assert len(keys) == len(wanted)
    for k, r in zip(keys, res):
        want = np.zeros_like(got[k]) if r is None else r.numpy().reshape(got[k].shape)
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(got[k] - want).max() < 1e-9 * scale, (k, np.abs(got[k] - want).max(), scale)


def test_host_envmap_configure_matches_oracle(psdr, orc):
    """EnvironmentMap::configure, the bounding cube and the emitter weights (scene.cpp:434-515) on the host"""
    spec = scenes.envmap_scene(32, 32, 4, 4, 4, param="box_x", area_light=True)
    ref = orc.OracleScene(spec, [0])
    sc = product.build_scene(spec, host_only=True)
    snap = sc._snapshot()
    assert sc.num_meshes == len(spec.meshes) + 1                       # + bounding cube
    assert np.array_equal(snap["triangles"], ref.triangle_info(False)[:, :22])
    assert np.array_equal(snap["d_triangles"], ref.triangle_info(True)[:, :22])
    bounds, reso, cell_sum, pmf, cmf = ref.envmap_info()
    assert tuple(snap["env_reso"]) == reso == (126, 62)
    assert np.array_equal(np.asarray(snap["env_bounds"]).ravel(), bounds)
    assert np.array_equal(np.asarray(snap["env_cell_pmf"]).ravel(), pmf) and np.array_equal(np.asarray(snap["env_cell_cmf"]).ravel(), cmf)
    assert float(snap["env_cell_sum"]) == cell_sum
    w = np.asarray(snap["emitter_weights"]).ravel()
    assert w[0] == np.float32(ref.emitter_weight(0)) and w[1] == np.float32(ref.emitter_weight(1))
    assert abs(w[0] - 0.5) < 1e-6 and abs(w[1] - 0.5) < 1e-6          # the envmap takes the summed weight of the others
    # a second envmap is refused (scene.cpp:86)
    with pytest.raises(RuntimeError, match="only allowed to have one envmap"):
        sc.add_EnvironmentMap(psdr.EnvironmentMap(scenes.synthetic_envmap(8, 4)))


def _cbox_xml(data_dir, tex_file=None, env_file=None):
    refl = '<texture type="bitmap" name="reflectance"><string name="filename" value="%s"/></texture>' % tex_file if tex_file else '<rgb name="reflectance" value="0.95, 0.95, 0.95"/>'
    env = '<emitter type="envmap"><string name="filename" value="%s"/><float name="scale" value="2.0"/><transform name="to_world"><rotate y="1" angle="30"/></transform></emitter>' % env_file if env_file else ""
    return """<?xml version="1.0"?>
<scene version="0.6.0">
  <sensor type="perspective">
    <float name="fov" value="60"/><string name="fov_axis" value="x"/>
    <float name="near_clip" value="0.000001"/><float name="far_clip" value="10000000"/>
    <transform name="to_world"><look_at origin="208, 273, -800" target="208, 273, 0" up="0, 1, 0"/></transform>
    <sampler type="independent"><integer name="sample_count" value="4"/></sampler>
    <film type="hdrfilm"><integer name="width" value="40"/><integer name="height" value="30"/></film>
  </sensor>
  <bsdf type="diffuse" id="light"><rgb name="reflectance" value="0"/></bsdf>
  <bsdf type="diffuse" id="white">%s</bsdf>
  <bsdf type="diffuse" id="cat"><float name="reflectance" value="0.5"/></bsdf>
  %s
  <shape type="obj" id="lum"><string name="filename" value="%s/cbox_luminaire.obj"/><ref id="light"/>
    <transform name="to_world"><translate y="-0.5"/></transform>
    <emitter type="area"><rgb name="radiance" value="20, 20, 8"/></emitter></shape>
  <shape type="obj"><string name="filename" value="%s/cbox_floor.obj"/><ref id="white"/></shape>
  <shape type="obj"><string name="filename" value="%s/cbox_largebox.obj"/><ref id="cat"/><boolean name="face_normals" value="true"/>
    <transform name="to_world"><scale x="1.0" y="0.5" z="1.0"/><translate x="10"/></transform></shape>
</scene>""" % (refl, env, data_dir, data_dir, data_dir)


def test_xml_scene_loader(psdr, tmp_path):
    """Scene.load_string / load_file (scene_loader.cpp:174-510): same scene as the equivalent add_* calls"""
    from psdr_jit_amd import exr
    tex = scenes.checker_texture(8, 8, 2)
    exr.write_rgb(str(tmp_path / "tex.exr"), tex)
    envimg = scenes.synthetic_envmap(16, 8)
    exr.write_rgb(str(tmp_path / "env.exr"), envimg)
    xml = _cbox_xml(scenes.DATA, str(tmp_path / "tex.exr"), str(tmp_path / "env.exr"))
    sc = psdr.Scene()
    sc.opts.log_level = 0
    sc.load_string(xml, False)
    f = tmp_path / "scene.xml"
    f.write_text(xml)
    sc2 = psdr.Scene()
    sc2.opts.log_level = 0
    sc2.load_file(str(f), False)
    for s in (sc, sc2):
        assert (s.opts.width, s.opts.height, s.opts.spp, s.opts.sppe, s.opts.sppse) == (40, 30, 4, 0, 0)
        assert s.num_sensors == 1 and s.num_meshes == 3 and s.get_num_emitters() == 2
        assert np.array_equal(np.asarray(s.param_map["BSDF[id=white]"]._get("reflectance", False)), tex)
        assert np.allclose(np.asarray(s.param_map["BSDF[id=cat]"]._get("reflectance", False)), 0.5)
        env = s.param_map["Emitter[0]"]
        assert env.scale == 2.0 and np.array_equal(np.asarray(env.radiance), envimg)
        c, sn = np.cos(np.radians(30.0)), np.sin(np.radians(30.0))
        assert np.allclose(np.asarray(env.to_world), [[c, 0, sn, 0], [0, 1, 0, 0], [-sn, 0, c, 0], [0, 0, 0, 1]], atol=1e-6)
        assert np.allclose(np.asarray(s.param_map["Emitter[1]"]._get("radiance", False)), [20, 20, 8])
        assert "Mesh[id=lum]" in s.param_map
        cam = np.asarray(s.param_map["Sensor[0]"]._get("to_world", False))
        # look_at(origin, target=+z, up=+y): columns left = up x dir = +x, up, dir, origin (transform.h:83-104)
        assert np.allclose(cam, [[1, 0, 0, 208], [0, 1, 0, 273], [0, 0, 1, -800], [0, 0, 0, 1]], atol=1e-5)
        box = s.param_map["Mesh[2]"]
        assert box.use_face_normal is True
        assert np.allclose(np.asarray(box._get("to_world", False)), [[1, 0, 0, 10], [0, 0.5, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])   # scale, then translate
        s._configure_host([0])
        assert s.num_meshes == 4                            # + the environment map's bounding cube
    mf = psdr.Scene()
    mf.load_string('<scene><bsdf type="microfacet" id="m"><rgb name="specular_reflectance" value="0.6, 0.5, 0.4"/><rgb name="diffuseReflectance" value="0.1"/>'
                   '<float name="roughness" value="0.3"/></bsdf></scene>', False)
    b = mf.param_map["BSDF[id=m]"]
    assert np.allclose(np.asarray(b._get("specularReflectance", False)), [0.6, 0.5, 0.4]) and np.allclose(np.asarray(b._get("diffuseReflectance", False)), 0.1)
    assert abs(float(np.asarray(b._get("roughness", False))[0]) - 0.3) < 1e-7
    rd = psdr.Scene()
    rd.opts.log_level = 0
    rd.load_string('<scene><bsdf type="roughdielectric" id="g"><float name="alpha" value="0.2"/><float name="intIOR" value="1.5"/>'
                   '<float name="extIOR" value="1.2"/></bsdf></scene>', False)
    g = rd.param_map["BSDF[id=g]"]
    assert type(g).__name__ == "RoughDielectricBSDF" and float(np.asarray(g._get("eta", False))[0]) == np.float32(1.5) / np.float32(1.2)
    assert float(np.asarray(g._get("inv_eta", False))[0]) == np.float32(1.2) / np.float32(1.5) and float(np.asarray(g._get("alpha_v", False))[0]) == np.float32(0.2)
    with pytest.raises(RuntimeError, match="Unsupported normal map nested BSDF"):
        psdr.Scene().load_string('<scene><bsdf type="normalmap" id="a"/></scene>', False)
    with pytest.raises(RuntimeError, match="BSDF must have an id"):
        psdr.Scene().load_string('<scene><bsdf type="diffuse"><rgb name="reflectance" value="1"/></bsdf></scene>', False)
    with pytest.raises(RuntimeError, match="XML parsing failed"):
        psdr.Scene().load_string("<scene>", False)
    with pytest.raises(RuntimeError, match="Unsupported sensor"):
        psdr.Scene().load_string('<scene><sensor type="orthographic"><film><integer name="width" value="1"/><integer name="height" value="1"/></film><sampler><integer name="n" value="1"/></sampler></sensor></scene>', False)


def test_exr_round_trip(psdr, tmp_path):
    from psdr_jit_amd import exr
    img = (np.random.default_rng(0).random((19, 31, 3)) * 7).astype(np.float32)
    for comp in ("none", "zip"):
        f = str(tmp_path / ("a_%s.exr" % comp))
        exr.write_rgb(f, img, comp)
        assert np.array_equal(exr.read_rgb(f), img)
    e = psdr.EnvironmentMap(str(tmp_path / "a_zip.exr"))
    assert (e.width, e.height) == (31, 19) and np.array_equal(np.asarray(e.radiance), img)


def test_obj_loader_matches_test_loader(psdr):
    for f in sorted(os.listdir(scenes.DATA)):
        v, faces, uv, fuv = scenes.load_obj(os.path.join(scenes.DATA, f))
        m = psdr.Mesh()
        m.load(os.path.join(scenes.DATA, f))
        assert m.num_vertices == len(v) and m.num_faces == len(faces)
        assert np.array_equal(np.asarray(m._get("vertex_positions", False)), v)
        assert np.array_equal(np.asarray(m.face_indices), faces)
        if uv is not None:
            assert np.array_equal(np.asarray(m.vertex_uv), uv) and np.array_equal(np.asarray(m.face_uv_indices), fuv)


def test_obj_loader_concave_polygon(psdr, tmp_path):
    # an L-shaped hexagon: ear clipping must not create triangles outside the polygon
    p = tmp_path / "L.obj"
    p.write_text("v 0 0 0\nv 2 0 0\nv 2 1 0\nv 1 1 0\nv 1 2 0\nv 0 2 0\nf 1 2 3 4 5 6\n")
    m = psdr.Mesh()
    m.load(str(p))
    assert m.num_faces == 4
    v = np.asarray(m._get("vertex_positions", False))
    area = 0.0
    for a, b, c in np.asarray(m.face_indices):
        n = np.cross(v[b] - v[a], v[c] - v[a])
        assert n[2] > 0                      # consistent orientation
        area += 0.5 * n[2]
    assert abs(area - 3.0) < 1e-6


def test_api_semantics(psdr):
    sc = psdr.Scene()
    sc.opts.log_level = 0
    assert (sc.opts.width, sc.opts.height, sc.opts.spp, sc.opts.sppe) == (128, 128, 1, 0)
    ro = psdr.RenderOption(64, 32, 4)
    assert (ro.width, ro.height, ro.spp, ro.sppe, ro.sppse) == (64, 32, 4, 4, 4)
    sc.add_BSDF(psdr.DiffuseBSDF([0.2, 0.3, 0.4]), "a")
    with pytest.raises(RuntimeError, match="Duplicate BSDF id"):
        sc.add_BSDF(psdr.DiffuseBSDF(), "a")
    with pytest.raises(RuntimeError, match="Unknown BSDF id"):
        sc.add_Mesh(os.path.join(scenes.DATA, "cbox_floor.obj"), np.eye(4, dtype=np.float32), "nope", None)
    with pytest.raises(RuntimeError, match="Failed to load OBJ"):
        sc.add_Mesh("/nonexistent.obj", np.eye(4, dtype=np.float32), "a", None)
    with pytest.raises(RuntimeError, match="Missing meshes"):
        sc._configure_host([])
    sc.add_Mesh(os.path.join(scenes.DATA, "cbox_luminaire.obj"), psdr.Matrix4fC([[1, 0, 0, 0], [0, 1, 0, -0.5], [0, 0, 1, 0], [0, 0, 0, 1]]), "a",
                psdr.AreaLight([20.0, 20.0, 8.0]))
    with pytest.raises(RuntimeError, match="Missing sensor"):
        sc._configure_host([])
    cam = psdr.PerspectiveCamera(60, 1e-6, 1e7)
    cam.to_world = psdr.Matrix4fD([[1., 0., 0., 208.], [0., 1., 0., 273.], [0., 0., 1., -800.], [0., 0., 0., 1.]])
    sc.add_Sensor(cam)
    assert sc.num_meshes == 1 and sc.num_sensors == 1 and sc.get_num_emitters() == 1
    keys = set(sc.param_map.keys())
    assert {"Mesh[0]", "BSDF[0]", "BSDF[id=a]", "Emitter[0]", "Sensor[0]"} <= keys
    assert np.allclose(np.asarray(sc.param_map["Sensor[0]"]._get("to_world", False))[:3, 3], [208, 273, -800])
    assert np.allclose(np.asarray(sc.param_map["BSDF[id=a]"]._get("reflectance", False)), [0.2, 0.3, 0.4])
    # scaling in the sensor transform is rejected (sensor.cpp:11-13)
    sc.param_map["Sensor[0]"].set_transform(np.diag([2.0, 1, 1, 1]).astype(np.float32))
    with pytest.raises(RuntimeError, match="should not involve scaling"):
        sc._configure_host([0])
    sc.param_map["Sensor[0]"].set_transform(np.eye(4, dtype=np.float32))
    sc._configure_host([0])
    # samplers are seeded at configure with scene.seed, lane count = W*H*spp (scene.cpp:330-344)
    ready, count, seed, skip = sc._sampler_state(0)
    assert ready and count == 128 * 128 * 1 and seed == 0 and skip == 0
    assert not sc.is_ready()        # host half only: no device scene yet
    integ = psdr.PathTracer(3)
    assert integ.max_depth == 3 and integ.hide_emitters is False
    with pytest.raises(RuntimeError):
        psdr.PathTracer(-1)


def test_torch_leaves_follow_the_scene_copies(psdr):
    import torch
    sc = psdr.Scene()
    sc.opts.log_level = 0
    P = torch.tensor(0.25, requires_grad=True)
    refl = torch.tensor([0.9, 0.2, 0.2], requires_grad=True)
    sc.add_BSDF(psdr.DiffuseBSDF(refl), "red")
    sc.add_Mesh(os.path.join(scenes.DATA, "cbox_redwall.obj"), psdr.Matrix4fC(np.eye(4)), "red", None)
    sc.param_map["Mesh[0]"].set_transform(psdr.Matrix4fD([[1., 0., 0., P * 100.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    leaves = {(type(o).__name__, n) for (o, n, t) in psdr._leaves(sc)}
    assert ("Mesh", "to_world_left") in leaves and ("DiffuseBSDF", "reflectance") in leaves
    m = np.asarray(sc.param_map["Mesh[0]"]._get("to_world_left", False))
    assert m[0, 3] == np.float32(25.0)
    t = sc.param_map["Mesh[0]"].to_world_left
    (g,) = torch.autograd.grad(t[0, 3], P)
    assert float(g) == 100.0


def _piz_encode(planes):
    """Test-side PIZ encoder for one chunk (bitmap + LUT, forward wavelet, canonical Huffman with equal code lengths, 6-bit
    length table with zero-run escapes, run-length escape for repeats): the mirror of csrc/host/exr_piz.cpp, written
    separately so that encode -> decode is a real round trip.  planes: list of uint16 arrays [ny, nx, words]."""
    flat = np.concatenate([p.reshape(-1) for p in planes]).astype(np.uint16)
    present = np.zeros(65536, bool); present[flat] = True
    bitmap = np.zeros(8192, np.uint8)
    for i in np.nonzero(present)[0]:
        if i != 0:
            bitmap[i >> 3] |= 1 << (i & 7)
    nz = np.nonzero(bitmap)[0]
    min_nz, max_nz = (int(nz[0]), int(nz[-1])) if len(nz) else (8191, 0)
    vals = [i for i in range(65536) if i == 0 or present[i]]
    fwd = np.zeros(65536, np.int64); fwd[vals] = np.arange(len(vals))
    max_value = len(vals) - 1
    w14 = max_value < (1 << 14)

    def wenc(a, b):
        if w14:
            a = a - 65536 if a >= 32768 else a; b = b - 65536 if b >= 32768 else b
            m = (a + b) >> 1; d = a - b
            return m & 0xffff, d & 0xffff
        ao = (a + 32768) & 0xffff
        m = (ao + b) >> 1
        d = ao - b
        if d < 0:
            m = (m + 32768) & 0xffff
        return m, d & 0xffff

    coded = []
    for p in planes:
        ny, nx, wds = p.shape
        q = fwd[p.astype(np.int64)].astype(np.int64).copy()
        for j in range(wds):
            a = q[:, :, j]
            n = min(nx, ny); pp, p2 = 1, 2
            while p2 <= n:
                y = 0
                while y <= ny - p2:
                    x = 0
                    while x <= nx - p2:
                        i00, i01 = wenc(int(a[y, x]), int(a[y, x + pp])); i10, i11 = wenc(int(a[y + pp, x]), int(a[y + pp, x + pp]))
                        a[y, x], a[y + pp, x] = wenc(i00, i10); a[y, x + pp], a[y + pp, x + pp] = wenc(i01, i11)
                        x += p2
                    if nx & pp:
                        a[y, x], a[y + pp, x] = wenc(int(a[y, x]), int(a[y + pp, x]))
                    y += p2
                if ny & pp:
                    x = 0
                    while x <= nx - p2:
                        a[y, x], a[y, x + pp] = wenc(int(a[y, x]), int(a[y, x + pp]))
                        x += p2
                pp, p2 = p2, p2 << 1
        coded.append(q.reshape(-1))
    sym = np.concatenate(coded)
    used = sorted(set(int(s) for s in sym))
    im, iM = used[0], used[-1] + 1                       # iM = the run-length escape
    L = max(1, int(np.ceil(np.log2(len(used) + 1))))
    lens = {s: L for s in used}; lens[iM] = L
    count = [0] * 59
    for l in lens.values():
        count[l] += 1
    base = [0] * 59; c = 0
    for l in range(58, 0, -1):
        nc = (c + count[l]) >> 1; base[l] = c; c = nc
    code, nxt = {}, list(base)
    for s in sorted(lens):
        code[s] = nxt[lens[s]]; nxt[lens[s]] += 1
    bits = []

    def put(v, n):
        bits.extend((v >> (n - 1 - k)) & 1 for k in range(n))
    s = im
    while s <= iM:
        if s in lens:
            put(lens[s], 6); s += 1
            continue
        run = 0
        while s + run <= iM and (s + run) not in lens:
            run += 1
        if run >= 6:
            r = min(run, 255 + 6); put(63, 6); put(r - 6, 8)
        elif run >= 2:
            r = run; put(59 + r - 2, 6)
        else:
            r = 1; put(0, 6)
        s += r
    while len(bits) % 8:
        bits.append(0)
    table = np.packbits(np.array(bits, np.uint8)).tobytes()
    bits = []
    i = 0
    while i < len(sym):
        put(code[int(sym[i])], L)
        run = 0
        while i + 1 + run < len(sym) and sym[i + 1 + run] == sym[i] and run < 255:
            run += 1
        if run >= 3:
            put(code[iM], L); put(run, 8); i += run
        i += 1
    n_bits = len(bits)
    while len(bits) % 8:
        bits.append(0)
    data = np.packbits(np.array(bits, np.uint8)).tobytes()
    import struct
    huf = struct.pack("<5I", im, iM, len(table), n_bits, 0) + table + data
    head = struct.pack("<HH", min_nz, max_nz) + (bitmap[min_nz:max_nz + 1].tobytes() if min_nz <= max_nz else b"")
    return head + struct.pack("<I", len(huf)) + huf


@pytest.mark.parametrize("case", ["half_small_range", "float_full_range", "odd_sizes", "constant"])
def test_exr_piz_decoder_round_trip(psdr, case):
    """csrc/host/exr_piz.cpp against an independent test-side encoder: 14-bit and 16-bit wavelets, odd widths / heights,
    one- and two-word samples, run-length escapes, zero-run escapes of the code-length table"""
    from psdr_jit_amd import _psdr_core
    rng = np.random.default_rng({"half_small_range": 1, "float_full_range": 2, "odd_sizes": 3, "constant": 4}[case])
    if case == "half_small_range":
        nx, ny, words = 16, 8, [1, 1, 1]
        planes = [rng.integers(0, 300, (ny, nx, 1)).astype(np.uint16) * 7 for _ in words]
    elif case == "float_full_range":
        nx, ny, words = 12, 9, [2, 1]
        planes = [rng.integers(0, 65536, (ny, nx, w)).astype(np.uint16) for w in words]
        planes[0][0, :4, :] = 0            # make sure value 0 and a few repeats occur
    elif case == "odd_sizes":
        nx, ny, words = 13, 7, [1, 2, 1]
        planes = [np.sort(rng.integers(0, 20000, (ny, nx, w)).astype(np.uint16), axis=1) for w in words]
    else:
        nx, ny, words = 9, 5, [1]
        planes = [np.full((ny, nx, 1), 15360, np.uint16)]
    chunk = _piz_encode(planes)
    out = np.asarray(_psdr_core._piz_decode(chunk, nx, ny, words))
    want = np.concatenate([np.concatenate([p[y].reshape(-1) for p in planes]) for y in range(ny)])
    assert out.dtype == np.uint16 and np.array_equal(out, want)
    with pytest.raises(RuntimeError, match="PIZ"):
        _psdr_core._piz_decode(chunk[:len(chunk) // 2], nx, ny, words)


def test_exr_piz_tutorial_envmap(psdr):
    """the reference's own tutorial environment map (tutorials/data/envmap/ballroom_1k.exr, PIZ, HALF): decodes to a finite,
    smooth 1024x512 radiance image; statistics pinned as a regression (sum of all values is order independent enough in f64)"""
    from psdr_jit_amd import exr
    import psdr_jit_amd
    path = os.path.join(os.path.dirname(scenes.DATA), "envmap", "ballroom_1k.exr")
    img = exr.read_rgb(path)
    assert img.shape == (512, 1024, 3) and img.dtype == np.float32 and np.isfinite(img).all() and img.min() >= 0
    assert abs(float(img.max()) - 141.5) < 1e-3
    assert np.allclose(img.astype(np.float64).mean(axis=(0, 1)), [0.56065734, 0.45016956, 0.34719557], rtol=1e-6)
    rows = np.abs(np.diff(img, axis=0)).mean(axis=(1, 2))
    # no discontinuity at the 32-line chunk boundaries, and neighbouring pixels are correlated (a wrong wavelet / table is noise)
    assert rows[31::32].mean() < 1.2 * 0.5 * (rows[30::32].mean() + rows[32::32].mean())
    assert np.abs(np.diff(img, axis=1)).mean() < 0.2 * img.mean()
    e = psdr.EnvironmentMap(path)
    assert e.width == 1024 and e.height == 512


def test_python_sampler_and_distribution_match_the_fixtures(psdr, orc):
    """psdr.Sampler (numpy, host_utils.py) reproduces the frozen TEA-64 / PCG32 tables bit for bit; psdr.DiscreteDistribution and
    Bitmap.eval agree with the oracle's restatements"""
    import json
    from psdr_jit_amd import host_utils
    here = os.path.dirname(os.path.abspath(__file__))
    for v0, v1, want in json.load(open(os.path.join(here, "golden", "tea64.json"))):
        got = host_utils._tea64(np.array([int(v0)], np.uint64), np.array([int(v1)], np.uint64))
        assert int(got[0]) == int(want)
    for rec in json.load(open(os.path.join(here, "golden", "sampler_floats.json"))):
        lane, sv = int(rec["lane"]), int(rec["seed_value"])
        s = psdr.Sampler()
        s.seed(np.array([sv], np.uint64), lane_index=[lane])
        bits = [int(s.next_1d()[0].view(np.uint32)) for _ in rec["bits"]]
        assert bits == rec["bits"], (lane, sv)
    d = psdr.DiscreteDistribution()
    pmf = np.array([0.5, 0.0, 2.0, 1.5, 0.25], np.float32)
    d.init(pmf)
    assert abs(d.sum - 4.25) < 1e-6 and np.allclose(d.pmf(), pmf / 4.25)
    u = np.array([0.0, 0.1, 0.1177, 0.3, 0.6, 0.95, 0.999], np.float32)
    idx, p = d.sample(u)
    assert list(idx) == [0, 0, 2, 2, 3, 4, 4] and np.allclose(p, pmf[idx] / 4.25)
    # Bitmap::eval: texel centres of a ramp, both modes
    tex = scenes.ramp_texture()
    b = psdr.Bitmap3fD(tex)
    H, W = tex.shape[:2]
    uv = np.array([[0.0, 0.0], [1.0 - 1e-6, 0.0], [0.5, -0.5]], np.float32)
    got = b.eval(uv)
    assert np.allclose(got[0], tex[0, 0], atol=1e-5) and np.allclose(got[1], tex[0, W - 1], atol=1e-4)
    one = psdr.Bitmap3fD([0.2, 0.4, 0.6])
    assert np.allclose(one.eval(uv), [[0.2, 0.4, 0.6]] * 3)


def test_host_snapshot_of_the_ggx_family(psdr):
    """Scene::configure on the host for every BSDF type of the GGX family: record types, bitmap slots, per-vertex counts, the row of the
    BSDF nested in a normal map (appended behind the scene's own), mesh-local face indices"""
    cases = [(scenes.dielectric_cbox_scene(16, 16, 1, 1, 1), {5: (3, -1, 0, 0, 0, 0), 6: (3, -1, 0, 0, 0, 0)}),
             (scenes.textured_microfacet_scene(16, 16, 1, 1, 1), {0: (1, -1, 8, 5, 9, 0)}),
             (scenes.textured_ggx_scene(16, 16, 1, 1, 1, kind="roughconductor"), {0: (2, -1, 4, 6, 7, 0)}),
             (scenes.textured_ggx_scene(16, 16, 1, 1, 1, kind="roughdielectric"), {0: (3, -1, 0, 0, 7, 0)})]
    for spec, want in cases:
        sc = product.build_scene(spec, host_only=True)
        rows = sc._snapshot()["bsdf_rows"]
        assert len(rows) == len(spec.bsdfs)
        for i, w in want.items():
            assert tuple(rows[i]) == w, (i, rows[i], w)
    spec = scenes.pervertex_scene(16, 16, 1, 1, 1)
    snap = product.build_scene(spec, host_only=True)._snapshot()
    nv = len(spec.meshes[1].vertices)
    assert tuple(snap["bsdf_rows"][5]) == (4, -1, 0, 0, 0, nv)
    fi = np.asarray(snap["face_indices"]).reshape(-1, 3)
    off = sum(len(m.faces) for m in spec.meshes[:1])
    assert np.array_equal(fi[off:off + len(spec.meshes[1].faces)], spec.meshes[1].faces) and fi.max() < max(len(m.vertices) for m in spec.meshes)
    spec = scenes.normalmap_scene(16, 16, 1, 1, 1, nested="roughconductor")
    sc = product.build_scene(spec, host_only=True)
    rows = sc._snapshot()["bsdf_rows"]
    n_own = len([k for k in sc.param_map if k.startswith("BSDF[") and not k.startswith("BSDF[id=")])
    assert len(rows) == n_own + 1 and tuple(rows[0])[:3] == (5, n_own, 16) and rows[n_own][0] == 2
    assert type(sc.param_map["BSDF[id=tex]"].nested_bsdf).__name__ == "RoughConductorBSDF"


def test_scene_is_freed_when_dropped(psdr):
    """A populated Scene (sensor, BSDFs, meshes, normal map with a nested BSDF) must be collectable: the param_map wrappers the Python
    layer keeps in the scene's __dict__ reference the scene through a visible attribute, not through pybind11's hidden keep-alive."""
    import gc
    import weakref
    sc = product.build_scene(scenes.normalmap_scene(16, 16, 1, 0, 0), host_only=True)
    mesh = sc.param_map["Mesh[0]"]
    r = weakref.ref(sc)
    del sc
    gc.collect()
    assert r() is not None and mesh.num_vertices > 0       # a live wrapper keeps its scene (no dangling pointer)
    del mesh
    gc.collect()
    assert r() is None
    sc = product.build_scene(scenes.cbox_scene(16, 16, 1, 1, 1), host_only=True)
    r = weakref.ref(sc)
    del sc
    gc.collect()
    assert r() is None


def test_roughconductor_constructor_overloads(psdr):
    """(alpha, eta, k[, specular]) vs (alpha_u, alpha_v, eta, k[, specular]): told apart by count and Bitmap types, never guessed"""
    one = lambda b, n: np.asarray(b._get(n, False)).reshape(-1)
    b = psdr.RoughConductorBSDF(0.1, [1.5, 1.6, 1.7], [2.0, 2.1, 2.2], [1.0, 0.9, 0.8])
    assert np.allclose(one(b, "alpha_u"), 0.1) and np.allclose(one(b, "alpha_v"), 0.1) and np.allclose(one(b, "eta"), [1.5, 1.6, 1.7])
    assert np.allclose(one(b, "k"), [2.0, 2.1, 2.2]) and np.allclose(one(b, "specular_reflectance"), [1.0, 0.9, 0.8])
    b = psdr.RoughConductorBSDF(0.1, 0.2, [1.5, 1.6, 1.7], [2.0, 2.1, 2.2], [1.0, 0.9, 0.8])
    assert np.allclose(one(b, "alpha_u"), 0.1) and np.allclose(one(b, "alpha_v"), 0.2) and np.allclose(one(b, "k"), [2.0, 2.1, 2.2])
    b = psdr.RoughConductorBSDF(psdr.Bitmap1fD(0.1), psdr.Bitmap1fD(0.3), psdr.Bitmap3fD([1.5, 1.6, 1.7]), psdr.Bitmap3fD([2.0, 2.1, 2.2]))
    assert np.allclose(one(b, "alpha_v"), 0.3) and np.allclose(one(b, "eta"), [1.5, 1.6, 1.7])
    b = psdr.RoughConductorBSDF(psdr.Bitmap1fD(0.01), psdr.Bitmap3fD([0.155475, 0.116753, 0.138334]), psdr.Bitmap3fD([4.83181, 3.12296, 2.1486]))
    assert np.allclose(one(b, "alpha_v"), 0.01) and np.allclose(one(b, "k"), [4.83181, 3.12296, 2.1486])
    with pytest.raises(ValueError):
        psdr.RoughConductorBSDF(0.1, 1.5, 2.0, [1.0, 1.0, 1.0])          # scalar eta or alpha_v?


def test_alpha_v_transform_is_the_alpha_u_transform_and_says_so(psdr):
    """ONE roughness map serves both axes here (the reference keeps a Bitmap - and a transform - per axis, bitmap.h:37-39): a transform set through one
    name and then changed through the other warns; the same value again, or the same name again, does not"""
    import warnings
    b = psdr.RoughConductorBSDF(0.1, 0.2, [1.5, 1.6, 1.7], [2.0, 2.1, 2.2], [1.0, 0.9, 0.8])
    b.uv_transform("alpha_u").rotate = 0.3
    assert b.uv_transform("alpha_v").rotate == 0.3
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        b.uv_transform("alpha_u").rotate = 0.4          # the same name again
        b.uv_transform("alpha_v").rotate = 0.4          # the other name, the same value
    with pytest.warns(RuntimeWarning, match="share ONE roughness map"):
        b.uv_transform("alpha_v").rotate = 0.5
    assert b.uv_transform("alpha_u").rotate == 0.5 and np.allclose(np.asarray(b._get_uv_xf(2, False))[0], 0.5)


def test_cornell_box_from_coordinates_equals_the_tutorial_files():
    """bench.py and tests/scenes.py build the README scene from examples/synth.py's coordinate lists; they parse to the arrays of
    the Cornell-box files of the reference's tutorials (kept under examples/data for the tutorials' other scenes)"""
    import synth
    paths = synth.write_cornell_box()
    assert sorted(paths) == sorted(["luminaire", "smallbox", "largebox", "floor", "ceiling", "back", "greenwall", "redwall"])
    n_tri = 0
    for name, path in paths.items():
        a = scenes.load_obj(path)
        b = scenes.load_obj(os.path.join(ROOT, "examples", "data", "cbox", "cbox_%s.obj" % name))
        for x, y in zip(a, b):
            assert (x is None and y is None) or np.array_equal(x, y), name
        n_tri += a[1].shape[0]
    assert n_tri == 36


def test_mesh_sample_position_and_the_record_classes(psdr):
    """Mesh.sample_position (reference psdr.cpp:321-322, mesh.cpp:403-454) in its C (numpy) and D (torch) instantiation, Mesh.valid_edge_indices, the record classes"""
    import torch
    spec = scenes.cbox_scene(16, 16, 1, 0, 0, param=None)
    sc = product.build_scene(spec, host_only=True)
    mesh = sc.param_map["Mesh[0]"]
    rng = np.random.default_rng(3)
    s2 = rng.random((4096, 2), dtype=np.float32)
    ps = mesh.sample_position(s2)
    assert isinstance(ps, psdr.PositionSampleC) and isinstance(ps, psdr.SampleRecordC)
    V = np.asarray(mesh.vertex_positions_T, np.float64)
    F = np.asarray(mesh.face_indices)
    e1, e2 = V[F[:, 1]] - V[F[:, 0]], V[F[:, 2]] - V[F[:, 0]]
    area = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)
    assert ps.p.shape == (4096, 3) and ps.is_valid.all() and np.allclose(ps.J, 1.0) and np.allclose(ps.pdf, 1.0 / area.sum(), rtol=1e-6)
    # every point lies on a face of the mesh, and the faces are hit in proportion to their areas
    N = np.cross(e1, e2); N /= np.linalg.norm(N, axis=1, keepdims=True)
    d = np.abs(np.einsum("pfk,fk->pf", ps.p[:, None, :].astype(np.float64) - V[F[:, 0]][None], N)).min(axis=1)
    assert d.max() < 1e-3 * np.abs(V).max()
    cen = (area[:, None] * (V[F[:, 0]] + (e1 + e2) / 3.0)).sum(axis=0) / area.sum()
    assert np.abs(ps.p.mean(axis=0) - cen).max() < 0.03 * np.ptp(V, axis=0).max()
    # the first sample coordinate picks the face by the running sum of the areas: u -> 0 lands on face 0, u -> 1 on the last face
    lo, hi = mesh.sample_position(np.array([[1e-6, 0.5]], np.float32)), mesh.sample_position(np.array([[1 - 1e-6, 0.5]], np.float32))
    inside = lambda p, f: abs(np.dot(p - V[F[f, 0]], N[f])) < 1e-3
    assert inside(lo.p[0].astype(np.float64), 0) and inside(hi.p[0].astype(np.float64), len(F) - 1)
    # D: the graph reaches the sample and the mesh's leaves
    P = torch.tensor(0.0, requires_grad=True)
    mesh.set_transform(psdr.Matrix4fD([[1., 0., 0., P * 10.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    st = torch.tensor(s2[:8], requires_grad=True)
    pd = mesh.sample_position(st)
    assert isinstance(pd, psdr.PositionSampleD) and pd.p.dtype == torch.float64 and torch.allclose(pd.J, torch.ones_like(pd.J))
    g_P, g_s = torch.autograd.grad(pd.p[:, 0].sum(), [P, st])
    assert abs(float(g_P) - 80.0) < 1e-6 and g_s.abs().sum() > 0
    # the member the reference binds read-write and never uses
    assert mesh.valid_edge_indices.shape == (0, 2)
    mesh.valid_edge_indices = [[0, 1], [2, 3]]
    assert mesh.valid_edge_indices.tolist() == [[0, 1], [2, 3]]
    its = psdr.InteractionC(np.zeros((1, 3)), np.zeros((1, 3)), np.ones(1), np.array([True]))
    assert its.is_valid()[0] and psdr.InteractionD is psdr.InteractionC
