"""BASELINE.json's configurations AS TIMED, HIP path (through the C ABI) vs the oracle:

  config 1  Cornell box 128 x 128, spp 4, PathTracer(1) renderC              - the whole frame
  config 2  Cornell box 512 x 512, spp 32, PathTracer(3) renderC           - full size, two bands of pixel rows + a lane range
  config 3  + sppe = sppse = 32, renderD w.r.t. Mesh[0] x-translation       - small at depth 3 (per term), and full size on a 1/64 shard
            (the kernels bench.py times: the LDS-class AD interior kernel and both edge kernels at depth 3)
  config 4  2048 x 2048, spp = sppe = sppse = 64 (268 M lanes per sampler)  - rank 3 of 8's lane arithmetic on 1/128 of its chunks
  config 5  82 k-triangle mesh under a 1024 x 512 environment map, albedo parameter, guiding grid [2000, 5, 5, 32], PathTracer(3)
            - the full mesh and map at 128 x 128 x 4 spp, and at full size (1024 x 1024 x 64) on a 1/1021 shard + shards add up to the frame

A shard = the 256-lane chunks k with k % count == rank of each of the three samplers (what a rank of the multi-GPU path
renders); oracle and kernels enumerate the same lanes, so the comparison is exact up to float summation order.
Tolerance: relative L2 < 1e-3 (BASELINE north_star), measured ~1e-7.
"""
import ctypes as C

import numpy as np
import pytest

import product
import scenes

pytestmark = pytest.mark.gpu
TOL = 1e-3


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


def _render_d(env, sc, n_pix, depth, seeds, rank=0, count=1, terms=7, guiding=None, skip_static_edges=False):
    torch, _, cabi = env
    buf = torch.empty((2, n_pix, 3), dtype=torch.float32, device="cuda")
    a = cabi.make_args(max_depth=depth, seeds=seeds, terms=terms, shard_rank=rank, shard_count=count, guiding=guiding, skip_static_edges=skip_static_edges)
    cabi.check(cabi.lib().psdr_hip_render_d_fwd(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None))
    torch.cuda.synchronize()
    return buf.cpu().numpy()


@pytest.mark.parametrize("param", ["light_x", "albedo"])
def test_config3_depth3_small_per_term(env, orc, param):
    """the timed kernels at the timed depth: class-1 (LDS) AD interior kernel + primary + secondary edges, PathTracer(3)"""
    spec = scenes.cbox_scene(64, 64, 8, 8, 8, param=param)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    for terms in (orc.TERM_INTERIOR, orc.TERM_PRIMARY, orc.TERM_SECONDARY, orc.TERM_ALL):
        got = _render_d(env, sc, 64 * 64, 3, (31, 32, 33), terms=terms)
        wimg, wd = ref.render_d(max_depth=3, seeds=(31, 32, 33), terms=terms)
        if terms & orc.TERM_INTERIOR:
            assert product.rel_l2(got[0], wimg) < TOL
        if np.abs(wd).max() > 0:
            assert product.rel_l2(got[1], wd) < TOL, (param, terms)
        else:
            assert np.abs(got[1]).max() == 0.0


@pytest.mark.parametrize("scene", ["cbox_lds", "sphere_bvh"])
def test_primary_edge_samples_that_cannot_contribute_are_not_traced(env, scene):
    """psdr_render_args.skip_static_edges (forward mode) and psdr_grads.prim_edge_filter (reverse mode), ABI 14: a primary-edge sample adds d(x.n) (Ln - Lp) / pdf to
    the derivative image and to ONE row of g_prim_edges (integrator.cpp:179-198).  With a zero normal velocity of the edge point the first is exactly zero, with an
    unwanted row nobody reads the second - such a sample's two paths are not traced: the same numbers are added to the derivative image by fewer rays, the wanted rows
    are unchanged"""
    torch, _, cabi = env
    spec = scenes.cbox_scene(64, 64, 8, 8, 8, param="light_x") if scene == "cbox_lds" else scenes.sphere_scene(64, 64, 8, 8, 8)
    sc = product.build_scene(spec)
    cam = sc.param_map["Sensor[0]"]
    d_prim = np.asarray(cam._primary_edges(True), np.float64)[:, :4]
    moving = np.abs(d_prim).sum(axis=1) > 0
    assert 0 < moving.sum() < moving.size, (moving.sum(), moving.size)        # some edges move, most do not
    n, depth, seeds = 64 * 64, 3, (41, 42, 43)
    imgs, rays = [], []
    for skip in (False, True):
        buf = torch.empty((2, n, 3), dtype=torch.float32, device="cuda")
        a = cabi.make_args(max_depth=depth, seeds=seeds, terms=2, skip_static_edges=skip)
        c = cabi.Counters()
        cabi.check(cabi.lib().psdr_hip_render_d_fwd_counted(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), C.byref(c), None))
        imgs.append(buf.cpu().numpy()); rays.append(int(c.rays))
    # (the same samples add the same numbers to a pixel; only the order of the float atomics differs from launch to launch)
    assert np.abs(imgs[0][1]).max() > 0 and np.abs(imgs[0][1] - imgs[1][1]).max() <= 2e-6 * np.abs(imgs[0][1]).max() and np.abs(imgs[1][0]).max() == 0.0
    assert rays[1] < 0.8 * rays[0], rays
    # reverse mode: only the rows of the moving edges are wanted
    gen = torch.Generator(device="cpu").manual_seed(5)
    w = (torch.rand((n, 3), generator=gen) + 0.5).to("cuda")
    snap = sc._snapshot()
    n_tris, n_sec = np.asarray(snap["d_triangles"]).shape[0], np.asarray(snap["d_sec_edges"]).shape[0]
    rows = []
    for filt in (None, torch.from_numpy(moving.astype(np.uint8)).to("cuda")):
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device="cuda")
        g_tri, g_bsdf, g_em, g_sec, g_prim = z(n_tris, 22), z(max(1, len(spec.bsdfs)), 3), z(max(1, len(spec.emitters)), 3), z(max(1, n_sec), 6), z(moving.size, 4)
        g = cabi.Grads(g_tri.data_ptr(), g_bsdf.data_ptr(), g_em.data_ptr(), g_sec.data_ptr(), g_prim.data_ptr())
        if filt is not None:
            g.prim_edge_filter = filt.data_ptr()
        a = cabi.make_args(max_depth=depth, seeds=seeds, terms=2)
        cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
        torch.cuda.synchronize()
        rows.append(g_prim.cpu().numpy().astype(np.float64))
    assert np.abs(rows[0][~moving]).max() > 0 and np.abs(rows[1][~moving]).max() == 0.0
    assert np.abs(rows[1][moving] - rows[0][moving]).max() <= 1e-5 * np.abs(rows[0][moving]).max()
    # <w, J v> = <J^T w, v> with the filtered rows: the unwanted rows meet zero tangents
    lhs = float((imgs[1][1].astype(np.float64) * w.cpu().numpy().astype(np.float64)).sum())
    rhs = float((rows[1] * d_prim).sum())
    assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), 1e-12), (lhs, rhs)


def _oracle_pinned_backward(env, orc_mod, sc, spec, ref, depth, seeds, rank, count, guiding=None, guiding_ref=None, tol=TOL, prim_filter=False):
    """reverse mode against the ORACLE on one shard, per term:  <w, d_img>  with d_img from the oracle's forward-mode render_d  ==  <J^T w, v>  with J^T w the buffers of
    psdr_hip_render_d_bwd on the same lanes and seeds and v the configured snapshot's tangent rows - no HIP forward kernel takes part"""
    torch, _, cabi = env
    snap = sc._snapshot()
    cam = sc.param_map["Sensor[0]"]
    d_tri = np.asarray(snap["d_triangles"], np.float64)
    d_sec = np.asarray(snap["d_sec_edges"], np.float64)[:, :6]
    d_prim = np.asarray(cam._primary_edges(True), np.float64)[:, :4]
    d_bsdf = np.array([b.d_reflectance for b in spec.bsdfs], np.float64)
    d_em = np.array([getattr(e, "d_radiance", (0.0, 0.0, 0.0)) for e in spec.emitters], np.float64)
    n = spec.width * spec.height
    gen = torch.Generator(device="cpu").manual_seed(3)
    w = torch.rand((n, 3), generator=gen) + 0.5
    wd = w.numpy().astype(np.float64)
    wg = w.to("cuda")
    res = {}
    for terms in (orc_mod.TERM_INTERIOR, orc_mod.TERM_PRIMARY, orc_mod.TERM_SECONDARY):
        _img, d_img = ref.render_d(max_depth=depth, seeds=seeds, terms=terms, shard_rank=rank, shard_count=count, guiding=guiding_ref)
        lhs = float((d_img.astype(np.float64) * wd).sum())
        scale = float((np.abs(d_img).astype(np.float64) * wd).sum()) + 1e-12
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device="cuda")
        g_tri, g_bsdf, g_em = z(d_tri.shape[0], 22), z(max(1, len(spec.bsdfs)), 3), z(max(1, len(spec.emitters)), 3)
        g_sec, g_prim = z(max(1, d_sec.shape[0]), 6), z(max(1, d_prim.shape[0]), 4)
        g = cabi.Grads(g_tri.data_ptr(), g_bsdf.data_ptr(), g_em.data_ptr(), g_sec.data_ptr(), g_prim.data_ptr())
        if prim_filter:
            # psdr_grads.prim_edge_filter: only the rows of the edges that move are wanted (what the host core asks for); the oracle still traces every sample
            moving = torch.from_numpy((np.abs(d_prim).sum(axis=1) > 0).astype(np.uint8)).to("cuda")
            g.prim_edge_filter = moving.data_ptr()
        a = cabi.make_args(max_depth=depth, seeds=seeds, terms=terms, shard_rank=rank, shard_count=count, guiding=guiding)
        cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), wg.data_ptr(), C.byref(g), None))
        torch.cuda.synchronize()
        f = lambda t: t.cpu().numpy().astype(np.float64)
        rhs = (f(g_tri) * d_tri).sum() + (f(g_bsdf)[:len(spec.bsdfs)] * d_bsdf).sum() + (f(g_em)[:len(spec.emitters)] * d_em).sum()
        rhs += (f(g_sec)[:d_sec.shape[0]] * d_sec).sum() + (f(g_prim)[:d_prim.shape[0]] * d_prim).sum()
        res[terms] = (lhs, float(rhs), scale)
        assert abs(lhs - rhs) <= tol * scale, (terms, lhs, rhs, scale)
    return res


def test_config3_backward_against_the_oracle(env, orc):
    """config 3 as timed (512 x 512, 32 / 32 / 32, PathTracer(3), Mesh[0] x-translation), shard 5 of 64: reverse mode per term against the oracle's forward mode"""
    spec = scenes.cbox_scene(512, 512, 32, 32, 32, param="light_x")
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    res = _oracle_pinned_backward(env, orc, sc, spec, ref, 3, (7, 7, 7), 5, 64)
    # the same with the work-elimination switch of reverse mode (rows of static edges not wanted -> their samples not traced): the oracle's number does not change
    res_f = _oracle_pinned_backward(env, orc, sc, spec, ref, 3, (7, 7, 7), 5, 64, prim_filter=True)
    assert abs(res_f[orc.TERM_PRIMARY][1] - res[orc.TERM_PRIMARY][1]) <= 1e-5 * res[orc.TERM_PRIMARY][2], (res, res_f)
    # (the interior term of a translated luminaire is zero in this estimator - positions on emitters are detached, path.cpp:47-83 -, on both sides; the edges carry the motion)
    assert abs(res[orc.TERM_PRIMARY][0]) > 1e-6 and abs(res[orc.TERM_SECONDARY][0]) > 1e-6 and res[orc.TERM_INTERIOR][2] < 1e-6, res


def test_config5_backward_against_the_oracle(env, orc):
    """config 5 as timed (1024 x 1024, 64 / 64 / 64, full mesh, map and guiding grid, albedo parameter), shard 777 of 1021: the class-2 reverse sweep, the primary-edge adjoint and
    the secondary-edge adjoint against the oracle's forward mode on the same lanes - the check round 3 lacked for the kernels behind `config5.backward`"""
    torch, psdr, cabi = env
    spec = scenes.config5_scene(1024, 1024, 64, 64, 64, level=6, env_res=(1024, 512))
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(3)
    reso = [2000, 5, 5, 32]
    integ.preprocess_secondary_edges(sc, 0, reso, 1, 0)
    g = ref.guiding_build(0, reso, nrounds=1, seed=0, max_depth=3)
    res = _oracle_pinned_backward(env, orc, sc, spec, ref, 3, (21, 22, 23), 777, 1021, guiding=integ._guiding_handle(0), guiding_ref=g)
    assert abs(res[orc.TERM_INTERIOR][0]) > 1e-6, res              # the albedo moves the interior term (the edge terms carry it through their path tails)


def test_config3_full_size_shard(env, orc):
    """512 x 512, 32 / 32 / 32, depth 3: shard 5 of 64 of every sampler (131 072 lanes each) - bench.py's workload and seeds layout"""
    spec = scenes.cbox_scene(512, 512, 32, 32, 32, param="light_x")
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    got = _render_d(env, sc, 512 * 512, 3, (7, 7, 7), rank=5, count=64)
    wimg, wd = ref.render_d(max_depth=3, seeds=(7, 7, 7), shard_rank=5, shard_count=64)
    assert np.isfinite(got).all()
    assert product.rel_l2(got[0], wimg) < TOL and product.rel_l2(got[1], wd) < TOL
    # the work-elimination switch of forward mode against the ORACLE (which traces every sample): primary-edge samples on edges that do not move add exactly zero
    skipped = _render_d(env, sc, 512 * 512, 3, (7, 7, 7), rank=5, count=64, skip_static_edges=True)
    assert product.rel_l2(skipped[0], wimg) < TOL and product.rel_l2(skipped[1], wd) < TOL
    # and the shards of a frame add up to the frame (what the all-reduce relies on), at full size
    full = _render_d(env, sc, 512 * 512, 3, (7, 7, 7))
    parts = sum(_render_d(env, sc, 512 * 512, 3, (7, 7, 7), rank=r, count=4) for r in range(4))
    assert product.rel_l2(parts[0], full[0]) < 1e-5 and product.rel_l2(parts[1], full[1]) < 1e-4


def test_forked_terms_do_not_share_the_stack_tail(env, monkeypatch):
    """The three terms of a renderD on a BVH scene run as concurrent launches on forked streams (api.hip::render_impl).  The global tail of the traversal stack is
    indexed by workgroup and thread only, so each forked term needs a slice of its own (round 6; until then workgroups with the same index in two launches wrote each
    other's entries whenever a walk went deeper than the LDS rows).  PSDR_STACK_LDS=2 sends nearly every walk of the 82 k-triangle mesh to the tail; the forked call
    must return what the three terms return one after the other (PSDR_NO_FORK), sample for sample - 4 samples per pixel, so one lost subtree is ~1e-4 of the frame"""
    torch, psdr, cabi = env
    monkeypatch.setenv("PSDR_STACK_LDS", "2")
    spec = scenes.config5_scene(96, 96, 4, 4, 4, level=6, env_res=(256, 128), param="blob_x")
    sc = product.build_scene(spec)
    monkeypatch.delenv("PSDR_STACK_LDS")
    nn, nl, md, lb = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    cabi.check(cabi.lib().psdr_hip_scene_stats(C.c_void_p(sc._hip_handle()), C.byref(nn), C.byref(nl), C.byref(md), C.byref(lb)))
    assert md.value > 4
    monkeypatch.setenv("PSDR_NO_FORK", "1")
    serial = _render_d(env, sc, 96 * 96, 3, (5, 6, 7))
    monkeypatch.delenv("PSDR_NO_FORK")
    for run in range(5):
        forked = _render_d(env, sc, 96 * 96, 3, (5, 6, 7))
        assert np.abs(serial[1]).max() > 0
        assert product.rel_l2(forked[0], serial[0]) < 2e-6 and product.rel_l2(forked[1], serial[1]) < 2e-5, run


def test_config1_exact_size(env, orc):
    """config 1 as BASELINE.json writes it - Cornell box 128 x 128, spp = 4, PathTracer(1), renderC ("plumbing"; the reference runs it on drjit's LLVM CPU backend): the
    whole frame against the oracle, through the Python surface the reference's README uses"""
    torch, psdr, cabi = env
    spec = scenes.cbox_scene(128, 128, 4, 0, 0, param=None)
    sc = product.build_scene(spec)
    img = psdr.PathTracer(1).renderC(sc, 0, seed=0).cpu().numpy()
    want = orc.OracleScene(spec, [0]).render_c(max_depth=1, seed=0)
    assert img.shape == (128 * 128, 3) and np.isfinite(img).all() and np.abs(want).max() > 0
    assert product.rel_l2(img, want) < TOL


def test_config2_full_size_rows(env, orc):
    """512 x 512, spp 32, PathTracer(3) renderC at full size; the oracle renders the lanes of pixel rows 200-207 (131 072 lanes,
    a contiguous lane range of its C-mode renderer) and a second band lower in the frame"""
    torch, _, cabi = env
    spec = scenes.cbox_scene(512, 512, 32, 0, 0, param=None)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    out = torch.empty((512 * 512, 3), dtype=torch.float32, device="cuda")
    a = cabi.make_args(max_depth=3, seeds=(3, 0, 0))
    cabi.check(cabi.lib().psdr_hip_render_c(sc._hip_handle(), C.byref(a), out.data_ptr(), None))
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    for r0, r1 in ((200, 208), (300, 306)):
        want = ref.render_c(max_depth=3, seed=3, lane_begin=r0 * 512 * 32, lane_end=r1 * 512 * 32)
        assert np.abs(want[r0 * 512:r1 * 512]).max() > 0
        assert product.rel_l2(got[r0 * 512:r1 * 512], want[r0 * 512:r1 * 512]) < TOL
    # per-lane radiance of the same kernel on a lane range deep in the frame
    lanes = torch.empty((65536, 3), dtype=torch.float32, device="cuda")
    b0 = 300 * 512 * 32
    cabi.check(cabi.lib().psdr_hip_li_lanes(sc._hip_handle(), C.byref(a), b0, b0 + 65536, lanes.data_ptr(), None))
    want = ref.li_lanes(b0, b0 + 65536, max_depth=3, seed=3)
    gl = lanes.cpu().numpy()
    bad = np.abs(gl - want).max(axis=1) > 1e-4 * np.abs(want).max()
    assert bad.mean() < 2e-4 and product.rel_l2(gl[~bad], want[~bad]) < 1e-5


def test_config4_lane_arithmetic_shard(env, orc):
    """2048 x 2048, spp = sppe = sppse = 64: 268 435 456 lanes per sampler (beyond 2^28, lane * 3 beyond 2^31).  Rank 3 of 8 owns
    the chunks k % 8 == 3; every 128-th of those (k % 1024 == 3 + 8 * 77) is rendered by both sides."""
    spec = scenes.cbox_scene(2048, 2048, 64, 64, 64, param="light_x")
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    rank, count = 3 + 8 * 77, 1024
    got = _render_d(env, sc, 2048 * 2048, 3, (11, 12, 13), rank=rank, count=count)
    wimg, wd = ref.render_d(max_depth=3, seeds=(11, 12, 13), shard_rank=rank, shard_count=count)
    assert np.isfinite(got).all() and np.abs(wimg).max() > 0
    assert product.rel_l2(got[0], wimg) < TOL and product.rel_l2(got[1], wd) < TOL
    # the last chunk of the frame lands on the last pixels
    last = (2048 * 2048 * 64) // 256 - 1
    tail = _render_d(env, sc, 2048 * 2048, 3, (11, 12, 13), rank=last % count, count=count, terms=orc.TERM_INTERIOR)
    wtail, _ = ref.render_d(max_depth=3, seeds=(11, 12, 13), terms=orc.TERM_INTERIOR, shard_rank=last % count, shard_count=count)
    assert np.array_equal(np.nonzero(tail[0].sum(axis=1))[0][-1:], np.nonzero(wtail.sum(axis=1))[0][-1:])
    assert product.rel_l2(tail[0], wtail) < TOL


def test_config5_full_mesh(env, orc):
    """the 81 920-triangle noisy icosphere + floor under the 1024 x 512 sun map (tests/scenes.py::config5_scene), DiffuseBSDF albedo
    parameter, guiding grid [2000, 5, 5, 32], PathTracer(3), all three terms: the BVH path on the full-size scene"""
    torch, psdr, cabi = env
    spec = scenes.config5_scene(128, 128, 4, 4, 4, level=6, env_res=(1024, 512))
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    assert ref.num_triangles == 81920 + 2 + 12
    integ = psdr.PathTracer(3)
    reso = [2000, 5, 5, 32]
    integ.preprocess_secondary_edges(sc, 0, reso, 1, 0)
    g = ref.guiding_build(0, reso, nrounds=1, seed=0, max_depth=3)
    mass = np.asarray(integ._guiding_mass(0)).reshape(-1)
    assert product.rel_l2(mass, g.mass()) < TOL
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=1)
    wimg, wd = ref.render_d(max_depth=3, seeds=(1, 1, 1), guiding=g)
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL and product.rel_l2(dimg.cpu().numpy(), wd) < TOL
    # closest hits on the full mesh: bit-exact against the oracle's brute-force definition on random rays
    rng = np.random.default_rng(5)
    n = 20000
    o = rng.uniform([-100, 0, -200], [650, 500, 650], size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    tri = torch.empty(n, dtype=torch.int32, device="cuda")
    uv = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    t = torch.empty(n, dtype=torch.float32, device="cuda")
    cabi.check(cabi.lib().psdr_hip_trace(sc._hip_handle(), n, to.data_ptr(), td.data_ptr(), tri.data_ptr(), uv.data_ptr(), t.data_ptr(), None))
    wtri, wuv, wt = ref.trace(o, d, use_bvh=True)
    btri, buv, bt = ref.trace(o[:3000], d[:3000], use_bvh=False)       # the definition: every triangle, smallest (t, id)
    assert np.array_equal(wtri[:3000], btri) and np.array_equal(wt[:3000], bt)
    assert np.array_equal(tri.cpu().numpy(), wtri)
    hit = wtri >= 0
    assert hit.mean() > 0.5
    assert np.array_equal(uv.cpu().numpy()[hit], wuv[hit]) and np.array_equal(t.cpu().numpy()[hit], wt[hit])


def test_config5_full_size_shard(env, orc):
    """config 5 AS TIMED (bench.py --config 5): 1024 x 1024, spp = sppe = sppse = 64, the full mesh, map and guiding grid.  Shard 777 of
    1021 of every sampler (~65 700 lanes each; a prime count, so the 256-lane chunks = groups of four pixels scatter over the whole frame) against the oracle, and the four quarter
    shards of the frame against the frame (size-independent: what the multi-GPU all-reduce relies on, on the BVH path)."""
    torch, psdr, cabi = env
    spec = scenes.config5_scene(1024, 1024, 64, 64, 64, level=6, env_res=(1024, 512))
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(3)
    reso = [2000, 5, 5, 32]
    integ.preprocess_secondary_edges(sc, 0, reso, 1, 0)
    g = ref.guiding_build(0, reso, nrounds=1, seed=0, max_depth=3)
    gh = integ._guiding_handle(0)
    n_pix = 1024 * 1024
    got = _render_d(env, sc, n_pix, 3, (21, 22, 23), rank=777, count=1021, guiding=gh)
    wimg, wd = ref.render_d(max_depth=3, seeds=(21, 22, 23), guiding=g, shard_rank=777, shard_count=1021)
    assert np.isfinite(got).all() and np.abs(wimg).max() > 0 and np.abs(wd).max() > 0
    assert product.rel_l2(got[0], wimg) < TOL and product.rel_l2(got[1], wd) < TOL
    full = _render_d(env, sc, n_pix, 3, (21, 22, 23), guiding=gh)
    assert np.isfinite(full).all()
    parts = sum(_render_d(env, sc, n_pix, 3, (21, 22, 23), rank=r, count=4, guiding=gh) for r in range(4))
    assert product.rel_l2(parts[0], full[0]) < 1e-5 and product.rel_l2(parts[1], full[1]) < 1e-4
    # every pixel of the full-size frame received radiance (sky or surface) and the albedo derivative is confined to the blob and its surroundings
    assert (full[0].sum(axis=1) > 0).mean() > 0.999
