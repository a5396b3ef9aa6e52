"""Scene.configure() on a scene that already has a device copy (psdr_hip_scene_update): the tree is kept, refitted on the device when vertices
move, built again only when the refitted tree no longer fits; only what changed is rewritten and sent.  The reference rebuilds its OptiX GAS and
re-uploads every array in every Scene::configure (src/scene/scene_optix.cpp:265-332, src/scene/scene.cpp:311-599).

The hit of a ray is defined by the exact triangle test alone - every triangle, smallest (t, original id) - so whatever the tree went through, the hits
must stay BIT-EQUAL to the oracle's brute-force definition, and the images must stay those of a freshly created scene."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import product
import scenes

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    import psdr_jit_amd as psdr
    from psdr_jit_amd import cabi
    return torch, psdr, cabi


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def _trace(torch, cabi, sc, o, d):
    n = len(o)
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    tri = torch.empty(n, dtype=torch.int32, device="cuda")
    uv = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    t = torch.empty(n, dtype=torch.float32, device="cuda")
    cabi.check(cabi.lib().psdr_hip_trace(sc._hip_handle(), n, to.data_ptr(), td.data_ptr(), tri.data_ptr(), uv.data_ptr(), t.data_ptr(), None))
    return tri.cpu().numpy(), uv.cpu().numpy(), t.cpu().numpy()


def _violations(cabi, sc):
    import ctypes as C
    v = C.c_int64(-1)
    cabi.check(cabi.lib().psdr_hip_scene_check_tree(C.c_void_p(sc._hip_handle()), C.byref(v)))
    return v.value


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform([-100, 0, -200], [650, 500, 650], size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


def _check_hits(torch, cabi, orc, sc, spec, n=20000, seed=5, brute=4000):
    o, d = _rays(n, seed)
    tri, uv, t = _trace(torch, cabi, sc, o, d)
    ref = orc.OracleScene(spec, [0])
    wtri, wuv, wt = ref.trace(o, d, use_bvh=True)                         # (the oracle's own, independent tree)
    btri, buv, bt = ref.trace(o[:brute], d[:brute], use_bvh=False)        # the definition: every triangle, smallest (t, id)
    assert np.array_equal(wtri[:brute], btri) and np.array_equal(wt[:brute], bt)
    assert np.array_equal(tri, wtri)
    hit = wtri >= 0
    assert hit.mean() > 0.3
    assert np.array_equal(uv[hit], wuv[hit]) and np.array_equal(t[hit], wt[hit])


@pytest.mark.parametrize("level", [4, 6])
def test_refit_keeps_the_hits_bit_equal(env, orc, level):
    """config 5's mesh (81 920 triangles at level 6) translated, squashed and twisted with the topology unchanged: every configure() refits the device tree; the
    tree stays a bounding hierarchy of the triangles and 20 000 random rays hit what the oracle's brute-force loop says, bit for bit"""
    torch, psdr, cabi = env
    spec = scenes.config5_scene(64, 64, 2, 2, 2, level=level, env_res=(256, 128))
    sc = product.build_scene(spec)
    assert sc._last_update()["tree"] in ("built", "kept")
    assert _violations(cabi, sc) == 0
    mesh = sc.param_map["Mesh[0]"]
    v0 = np.asarray(spec.meshes[0].vertices, np.float32).copy()
    n_refit = 0
    # (the deformations stay inside the scene box of the first configure: the environment map's bounding cube is added once, reference scene.cpp:434-485,
    #  while the oracle scene below is made from scratch in every state)
    for step, (shift, squash, twist) in enumerate([((7.0, -3.0, 11.0), 1.0, 0.0), ((0.0, 0.0, 0.0), 0.8, 0.0), ((-20.0, 5.0, 0.0), 0.95, 0.6), ((0.0, 10.0, 0.0), 1.0, 0.0)]):
        v = v0.copy()
        v[:, 1] *= squash
        a = twist * v[:, 1] / 150.0
        x, z = v[:, 0].copy(), v[:, 2].copy()
        v[:, 0], v[:, 2] = np.cos(a) * x - np.sin(a) * z, np.sin(a) * x + np.cos(a) * z
        v += np.asarray(shift, np.float32)
        v = v.astype(np.float32)
        mesh._set("vertex_positions", v, np.zeros_like(v))
        sc.configure([0])
        info = sc._last_update()
        assert info["tree"] in ("refitted", "built"), info
        n_refit += info["tree"] == "refitted"
        assert _violations(cabi, sc) == 0
        spec.meshes[0].vertices = v
        _check_hits(torch, cabi, orc, sc, spec, seed=5 + step)
    assert n_refit >= 3            # (moderate deformations keep the topology useful: no build)
    # the image of the refitted scene = the image of a scene created from scratch in the same state
    integ = psdr.PathTracer(2)
    img = integ.renderC(sc, 0, seed=3).cpu().numpy()
    fresh = product.build_scene(spec)
    want = integ.renderC(fresh, 0, seed=3).cpu().numpy()
    assert product.rel_l2(img, want) < 1e-6
    ref = orc.OracleScene(spec, [0])
    assert product.rel_l2(img, ref.render_c(max_depth=2, seed=3)) < TOL


def test_moved_meshes_get_their_rows_from_the_device(env, orc, monkeypatch):
    """psdr_mesh_geometry (ABI 16): after a vertex move psdr_hip_scene_update computes the moved meshes' triangle rows (traversal, shading, tangent) and secondary-edge rows ON
    THE DEVICE from the raw vertices and the composed transform - the reference's Mesh::configure / process_mesh run on the GPU too (src/shape/mesh.cpp:23-62, 317-400) - in the
    host's own dual-number code compiled for both sides.  The rows must be the host's bits: psdr_hip_scene_check_rows compares every word with what the host path would write;
    far fewer bytes travel than the rows weigh; and the same update forced through the host path (PSDR_HOST_GEOMETRY) gives the same sections"""
    torch, psdr, cabi = env
    spec = scenes.config5_scene(48, 48, 2, 2, 2, level=5, env_res=(64, 32), param="blob_x")
    sc = product.build_scene(spec)
    assert sc._check_device_rows() == 0                       # (after the create: the host's rows)
    mesh = sc.param_map["Mesh[0]"]
    n_tris = sum(len(m.faces) for m in spec.meshes) + 12
    dT = np.zeros((4, 4), np.float32); dT[0, 3] = 100.0
    sent = []
    for step in range(4):
        if step % 2 == 0:                                      # a translation with a tangent (the README's parameter) ...
            t = np.eye(4, dtype=np.float32); t[0, 3] = 3.0 * (step + 1); t[1, 3] = -2.0 * step
            mesh._set("to_world_left", t, dT)
        else:                                                  # ... and a deformation of the raw vertices with a tangent of their own
            v = np.asarray(spec.meshes[0].vertices, np.float32).copy()
            v[:, 1] *= 1.0 - 0.03 * step
            dv = np.zeros_like(v); dv[:, 2] = 0.5 * v[:, 0]
            mesh._set("vertex_positions", v.astype(np.float32), dv)
        sc.configure([0])
        info = sc._last_update()
        assert info["tree"] in ("refitted", "built"), info
        assert sc._check_device_rows() == 0, step
        assert _violations(cabi, sc) == 0
        sent.append(info["bytes_uploaded"])
    # the rows of this scene weigh 240 B per triangle + 96 B per edge; what travels after the first update (which also carries the topology) is the vertices, the edge CDF, the sensor's edges
    assert sent[-1] < 0.25 * 240 * n_tris, sent
    img = psdr.PathTracer(2).renderC(sc, 0, seed=3).cpu().numpy()
    # the same state through the host path: every row rewritten and sent by the host
    monkeypatch.setenv("PSDR_HOST_GEOMETRY", "1")
    v = np.asarray(mesh._get("vertex_positions", False), np.float32).copy()
    mesh._set("vertex_positions", v, np.zeros_like(v))         # (a tangent change: the rows are written again)
    sc.configure([0])
    assert sc._check_device_rows() == 0
    assert sc._last_update()["bytes_uploaded"] > 96 * n_tris
    monkeypatch.delenv("PSDR_HOST_GEOMETRY")
    img_host = psdr.PathTracer(2).renderC(sc, 0, seed=3).cpu().numpy()
    assert product.rel_l2(img, img_host) < 1e-6
    # the lean configures in between left the host's own rows to whoever asks: asked now, they are those of a scene built from scratch in this state, bit for bit
    m0 = spec.meshes[0]
    m0.vertices = v; m0.d_vertices = None; m0.path = None
    t_last = np.eye(4, dtype=np.float32); t_last[0, 3] = 9.0; t_last[1, 3] = -4.0
    m0.to_world_left = t_last; m0.d_to_world_left = dT
    fresh = product.build_scene(spec, host_only=True)
    a, b = sc._snapshot(), fresh._snapshot()
    for k in ("triangles", "d_triangles", "sec_edges", "d_sec_edges"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_a_lean_configure_falls_back_to_the_rows_when_the_device_needs_them(env, orc):
    """psdr_scene_snapshot.rows_valid = 0 (what a lean Scene.configure sends once the device computes the rows itself): an update that cannot do without the rows - here a
    refit whose SAH cost calls for a new tree - answers PSDR_HIP_NEED_ROWS, the host computes the rows after all and the tree is built; hits stay bit-equal"""
    torch, psdr, cabi = env
    spec = scenes.config5_scene(32, 32, 1, 1, 1, level=4, env_res=(64, 32))
    sc = product.build_scene(spec)
    mesh = sc.param_map["Mesh[0]"]
    v = np.asarray(spec.meshes[0].vertices, np.float32).copy()
    mesh._set("vertex_positions", (v * 0.99).astype(np.float32), np.zeros_like(v))       # a first, gentle move: the device path, after which configures are lean
    sc.configure([0])
    assert sc._last_update()["tree"] == "refitted" and sc._check_device_rows() == 0
    rng = np.random.default_rng(1)
    v2 = (v * rng.uniform(0.2, 1.0, size=(len(v), 1))).astype(np.float32)
    mesh._set("vertex_positions", v2, np.zeros_like(v2))
    sc.configure([0])
    assert sc._last_update()["tree"] == "built", sc._last_update()
    assert _violations(cabi, sc) == 0 and sc._check_device_rows() == 0
    spec.meshes[0].vertices = v2
    _check_hits(torch, cabi, orc, sc, spec, n=8000, brute=2000)


def test_a_scrambled_mesh_is_built_again(env, orc):
    """vertices thrown far from where the topology was built for: the refitted tree's SAH cost exceeds 1.4 x the built one and configure() builds a new tree;
    hits stay bit-equal"""
    torch, psdr, cabi = env
    spec = scenes.config5_scene(32, 32, 1, 0, 0, level=4, env_res=(64, 32))
    sc = product.build_scene(spec)
    mesh = sc.param_map["Mesh[0]"]
    v = np.asarray(spec.meshes[0].vertices, np.float32).copy()
    rng = np.random.default_rng(1)
    v = (v * rng.uniform(0.2, 1.0, size=(len(v), 1))).astype(np.float32)       # radial scramble: neighbours in the tree are no longer neighbours in space
    mesh._set("vertex_positions", v, np.zeros_like(v))
    sc.configure([0])
    info = sc._last_update()
    assert info["tree"] == "built", info
    assert _violations(cabi, sc) == 0
    spec.meshes[0].vertices = v
    _check_hits(torch, cabi, orc, sc, spec, n=8000, brute=2000)


def test_parameter_changes_keep_the_tree_and_send_little(env, orc):
    """albedo, environment scale, camera pose, a forward tangent: the tree is kept, the triangle sections are not sent again"""
    torch, psdr, cabi = env
    spec = scenes.config5_scene(64, 64, 4, 4, 4, level=5, env_res=(256, 128))
    sc = product.build_scene(spec)
    sc.configure([0])
    first = sc._last_update()
    tri_bytes = 20480 * 15 * 16            # traversal + shading + tangent rows of the 20 480-triangle mesh
    integ = psdr.PathTracer(2)

    def reconfigure():
        sc.configure([0])
        info = sc._last_update()
        assert info["tree"] == "kept", info
        return info

    info = reconfigure()                                                   # nothing changed
    assert info["bytes_uploaded"] < 8192, info
    bs = sc.param_map["BSDF[0]"]
    bs._set("reflectance", np.asarray([0.3, 0.6, 0.2], np.float32), np.asarray([1, 1, 1], np.float32))
    info = reconfigure()
    assert info["bytes_uploaded"] < 8192, info
    spec.bsdfs[0].reflectance = (0.3, 0.6, 0.2)
    ref = orc.OracleScene(spec, [0])
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=2)
    wimg, wd = ref.render_d(max_depth=2, seeds=(2, 2, 2))
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL and product.rel_l2(dimg.cpu().numpy(), wd) < TOL
    # a tangent on the mesh transform: the tangent rows travel, the value rows and the tree do not
    mesh = sc.param_map["Mesh[0]"]
    d = np.zeros((4, 4), np.float32); d[0, 3] = 1.0
    mesh._set("to_world_left", np.eye(4, dtype=np.float32), d)
    bs._set("reflectance", np.asarray([0.3, 0.6, 0.2], np.float32), np.zeros(3, np.float32))
    info = reconfigure()
    # (rounds 1-5: the tangent rows of the triangles, the secondary-edge rows - 30 720 edges x 7 words - and the primary edges travelled: ~5.4 MB.  Round 6: the device computes the
    #  moved mesh's rows itself - the raw vertices, the edge CDF and the primary edges travel, and once the topology: ~2 MB, then well under 1 MB)
    assert info["bytes_uploaded"] < 3000000 and sc._check_device_rows() == 0, info
    d2 = d.copy(); d2[1, 3] = 0.5
    mesh._set("to_world_left", np.eye(4, dtype=np.float32), d2)
    info = reconfigure()
    assert info["bytes_uploaded"] < 1200000 and sc._check_device_rows() == 0, info
    mesh._set("to_world_left", np.eye(4, dtype=np.float32), d)
    reconfigure()
    spec.bsdfs[0].d_reflectance = (0.0, 0.0, 0.0)
    spec.meshes[0].d_to_world_left = d
    ref = orc.OracleScene(spec, [0])
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=4)
    wimg, wd = ref.render_d(max_depth=2, seeds=(4, 4, 4))
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL and product.rel_l2(dimg.cpu().numpy(), wd) < TOL
    # the camera moves: primary edges and matrices change, the triangles do not
    cam = sc.param_map["Sensor[0]"]
    tw = np.asarray(spec.cameras[0].to_world_raw, np.float32).copy()
    tw[0, 3] += 15.0
    cam._set("to_world", tw, np.zeros((4, 4), np.float32))
    info = reconfigure()
    assert info["bytes_uploaded"] < tri_bytes // 2, info
    spec.cameras[0].to_world_raw = tw
    ref = orc.OracleScene(spec, [0])
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=6)
    wimg, wd = ref.render_d(max_depth=2, seeds=(6, 6, 6))
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL and product.rel_l2(dimg.cpu().numpy(), wd) < TOL
    assert first["tree"] in ("built", "kept")


def test_cornell_box_updates_in_place(env, orc):
    """the README box (36 triangles, LDS class): a moved mesh rewrites its rows and the filter primitives in place; images against the oracle after every change"""
    torch, psdr, cabi = env
    spec = scenes.cbox_scene(64, 64, 8, 8, 8, param="light_x")
    sc = product.build_scene(spec)
    integ = psdr.PathTracer(2)
    for k, P in enumerate([0.0, 0.2, -0.15, 0.4]):
        scenes.set_param_value(spec, "light_x", P)
        m = spec.meshes[0]
        sc.param_map["Mesh[0]"]._set("to_world_left", np.asarray(m.to_world_left, np.float32), np.asarray(m.d_to_world_left, np.float32))
        sc.configure([0])
        assert sc._last_update()["tree"] in ("kept", "built", "refitted")
        ref = orc.OracleScene(spec, [0])
        img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=10 + k)
        wimg, wd = ref.render_d(max_depth=2, seeds=(10 + k,) * 3)
        assert product.rel_l2(img.cpu().numpy(), wimg) < TOL and product.rel_l2(dimg.cpu().numpy(), wd) < TOL


def test_update_equals_rebuild_on_every_scene_family(env, orc):
    """_always_rebuild (destroy + create in every configure, what rounds 1-4 did) against the in-place update on scenes with bitmaps, materials, per-vertex values,
    normal maps and an environment map: the same images after a parameter of each kind changed"""
    torch, psdr, cabi = env
    cases = [(scenes.textured_scene, {}), (scenes.microfacet_cbox_scene, {}), (scenes.pervertex_scene, {}), (scenes.normalmap_scene, {}),
             (scenes.envmap_scene, {"width": 48, "height": 48}), (scenes.textured_ggx_scene, {})]
    for make, kw in cases:
        spec = make(**kw)
        a, b = product.build_scene(spec), product.build_scene(spec)
        b._always_rebuild = True
        integ = psdr.PathTracer(2)
        for step in range(3):
            for sc in (a, b):
                if step == 1:           # a forward tangent on the first mesh (values unchanged)
                    d = np.zeros((4, 4), np.float32); d[2, 3] = 1.0
                    sc.param_map["Mesh[0]"]._set("to_world_left", np.asarray(sc.param_map["Mesh[0]"]._get("to_world_left", False), np.float32), d)
                if step == 2:           # the mesh moves
                    m = np.asarray(sc.param_map["Mesh[0]"]._get("to_world_left", False), np.float32).copy(); m[1, 3] += 3.0
                    sc.param_map["Mesh[0]"]._set("to_world_left", m, np.zeros((4, 4), np.float32))
                sc.configure([0])
            ia, da = psdr.render_d_fwd(integ, a, 0, seed=21 + step)
            ib, db = psdr.render_d_fwd(integ, b, 0, seed=21 + step)
            assert product.rel_l2(ia.cpu().numpy(), ib.cpu().numpy()) < 1e-6, (make.__name__, step)
            assert product.rel_l2(da.cpu().numpy(), db.cpu().numpy() + 1e-30) < 1e-5 or float(db.abs().max()) == 0.0, (make.__name__, step)
