"""GPU parity tests (run with -m gpu on an MI355X): the HIP path behind the C ABI vs the CPU oracle
on the same seeded inputs.

Tolerances.  Arithmetic is float32 on both sides with the same operation order where it matters
(explicit fma, no contraction), so individual sample lanes agree to ~1e-6 relative; the device
libm's sincosf and the wave-level summation order differ from the host's, and a handful of lanes per
million take a different branch at a geometric tie.  Image-level bounds used here:
   relative L2 over the image < 1e-3 for the primal image and for d(image)/d(theta)
   (BASELINE.json north_star: "gradient L2 error < 1e-3").
Integer work (RNG, indices) is bit-exact.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import product
import scenes

pytestmark = pytest.mark.gpu
TOL = 1e-3
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    return torch


@pytest.fixture(scope="module")
def psdr():
    import __graft_entry__
    __graft_entry__.build()
    import psdr_jit_amd
    return psdr_jit_amd


def test_sampler_bit_exact(torch_cuda, psdr, orc):
    from psdr_jit_amd import cabi
    L = cabi.lib()
    with open(os.path.join(GOLDEN, "tea64.json")) as fh:
        for a, b, want in json.load(fh):
            assert int(L.psdr_hip_tea64(int(a), int(b))) == int(want)
    with open(os.path.join(GOLDEN, "sampler_floats.json")) as fh:
        rows = json.load(fh)
    for row in rows:
        buf = torch_cuda.zeros(16, dtype=torch_cuda.float32, device="cuda")
        cabi.check(L.psdr_hip_sampler_floats(int(row["seed_value"]), int(row["lane"]), 0, 16, buf.data_ptr(), None))
        got = buf.cpu().numpy().view(np.uint32)
        assert [int(x) for x in got] == row["bits"]
    # closed-form skip-ahead == the oracle's sequential stream
    buf = torch_cuda.zeros(20, dtype=torch_cuda.float32, device="cuda")
    cabi.check(L.psdr_hip_sampler_floats(999, 123456, 17, 20, buf.data_ptr(), None))
    assert np.array_equal(buf.cpu().numpy(), orc.sampler_floats(999, 123456, 20, skip=17))
    # the kernels apply the skip-ahead as a precomputed affine map (sampler.h::skip_ahead): far into a run, lanes beyond 2^32, a skip beyond 2^40
    for seed_value, lane, skip in ((5, 7, 31 * 1000), (2 ** 33 + 11, 2 ** 32 + 99, 31 * 10 ** 6), (12345678901234, 268435455, 2 ** 40 + 12345), (0, 0, 2 ** 63 + 5)):
        cabi.check(L.psdr_hip_sampler_floats(seed_value, lane, skip, 20, buf.data_ptr(), None))
        assert np.array_equal(buf.cpu().numpy(), orc.sampler_floats(seed_value, lane, 20, skip=skip)), (seed_value, lane, skip)


@pytest.mark.parametrize("scene_name", ["cbox", "sphere"])
def test_trace_matches_oracle(torch_cuda, psdr, orc, scene_name):
    from psdr_jit_amd import cabi
    spec = scenes.cbox_scene(32, 32, 1, 0, 0) if scene_name == "cbox" else scenes.sphere_scene(32, 32, 1, 0, 0)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    rng = np.random.default_rng(3)
    n = 200000
    o = rng.uniform([20, 20, 20], [530, 530, 540], size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # rays that start on the camera and a few degenerate ones (axis aligned, NaN)
    o[:1000] = [208.0, 273.0, -800.0]
    d[1000:1010] = [0.0, 0.0, 1.0]
    d[1010] = [np.nan, 0.0, 1.0]
    # boundary-segment-like rays: from a point on a triangle edge (or a vertex) towards a point on another triangle's
    # edge / vertex - grazing, edge-on and in-plane rays, which is what the secondary-edge sampler produces and what
    # stresses the conservative candidate filter of the brute-force tracer
    ti = np.asarray(ref.triangle_info())[:, :9].astype(np.float64)
    m = 100000
    def edge_points(k):
        tr = ti[rng.integers(0, len(ti), size=k)]
        p0, e1, e2 = tr[:, 0:3], tr[:, 3:6], tr[:, 6:9]
        s = rng.random(k)[:, None]
        s[rng.random(k) < 0.2] = 0.0                     # vertices
        which = rng.integers(0, 3, size=k)[:, None]
        return np.where(which == 0, p0 + s * e1, np.where(which == 1, p0 + s * e2, p0 + e1 + s * (e2 - e1)))
    so, st = edge_points(m), edge_points(m)
    sd = st - so
    ln = np.linalg.norm(sd, axis=1, keepdims=True)
    keep = ln[:, 0] > 1e-3
    o[2000:2000 + m][keep] = so[keep].astype(np.float32)
    d[2000:2000 + m][keep] = (sd[keep] / ln[keep]).astype(np.float32)
    to, td = torch_cuda.from_numpy(o).cuda(), torch_cuda.from_numpy(d).cuda()
    tri = torch_cuda.empty(n, dtype=torch_cuda.int32, device="cuda")
    uv = torch_cuda.empty((n, 2), dtype=torch_cuda.float32, device="cuda")
    t = torch_cuda.empty(n, dtype=torch_cuda.float32, device="cuda")
    cabi.check(cabi.lib().psdr_hip_trace(sc._hip_handle(), n, to.data_ptr(), td.data_ptr(), tri.data_ptr(), uv.data_ptr(), t.data_ptr(), None))
    wtri, wuv, wt = ref.trace(o, d, use_bvh=False)
    assert np.array_equal(tri.cpu().numpy(), wtri)           # indices: bit exact
    hit = wtri >= 0
    assert np.array_equal(uv.cpu().numpy()[hit], wuv[hit]) and np.array_equal(t.cpu().numpy()[hit], wt[hit])
    # the two-rays-per-lane tracer of the path kernels (odd count: the last lane carries one ray)
    n2 = n - 1
    tri.fill_(-7); uv.zero_(); t.zero_()
    cabi.check(cabi.lib().psdr_hip_trace_pairs(sc._hip_handle(), n2, to.data_ptr(), td.data_ptr(), tri.data_ptr(), uv.data_ptr(), t.data_ptr(), None))
    assert np.array_equal(tri.cpu().numpy()[:n2], wtri[:n2]) and int(tri[n2]) == -7
    hit = wtri[:n2] >= 0
    assert np.array_equal(uv.cpu().numpy()[:n2][hit], wuv[:n2][hit]) and np.array_equal(t.cpu().numpy()[:n2][hit], wt[:n2][hit])


def _li_lanes(torch, sc, n, max_depth, seed, skip=0):
    from psdr_jit_amd import cabi
    out = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    a = cabi.make_args(max_depth=max_depth, seeds=(seed, 0, 0), skips=(skip, 0, 0))
    cabi.check(cabi.lib().psdr_hip_li_lanes(sc._hip_handle(), C.byref(a), 0, n, out.data_ptr(), None))
    return out.cpu().numpy()


@pytest.mark.parametrize("max_depth", [0, 1, 3])
def test_lane_radiance_matches_oracle(torch_cuda, psdr, orc, max_depth):
    spec = scenes.cbox_scene(64, 64, 8, 0, 0)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    n = 64 * 64 * 8
    got = _li_lanes(torch_cuda, sc, n, max_depth, seed=5)
    want = ref.li_lanes(0, n, max_depth=max_depth, seed=5)
    scale = np.abs(want).max()
    bad = np.abs(got - want).max(axis=1) > 1e-4 * scale
    assert bad.mean() < 2e-4, "fraction of lanes off by >1e-4: %g" % bad.mean()
    assert product.rel_l2(got[~bad], want[~bad]) < 1e-5


@pytest.mark.parametrize("scene_name,depth", [("cbox", 1), ("cbox", 3), ("sphere", 2)])
def test_render_c_matches_oracle(torch_cuda, psdr, orc, scene_name, depth):
    spec = scenes.cbox_scene(96, 64, 16, 0, 0) if scene_name == "cbox" else scenes.sphere_scene(64, 64, 16, 0, 0)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(depth)
    img = integ.renderC(sc, 0, seed=11).cpu().numpy()
    want = ref.render_c(max_depth=depth, seed=11)
    assert img.shape == want.shape and np.isfinite(img).all()
    assert product.rel_l2(img, want) < TOL
    # seed=-1 continues the sampler streams (scene.h:76): a second call differs and matches skip = 2+5*depth
    img2 = integ.renderC(sc, 0).cpu().numpy()
    want2 = ref.render_c(max_depth=depth, seed=11, skip=2 + 5 * depth)
    assert product.rel_l2(img2, want2) < TOL and product.rel_l2(img2, img) > 1e-2
    # hide_emitters
    integ.hide_emitters = True
    img3 = integ.renderC(sc, 0, seed=11).cpu().numpy()
    assert product.rel_l2(img3, ref.render_c(max_depth=depth, seed=11, hide_emitters=True)) < TOL


def test_render_c_batch_pixels(torch_cuda, psdr, orc):
    spec = scenes.cbox_scene(64, 48, 8, 0, 0)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(2)
    pix = np.array([0, 5, 64 * 10 + 3, 64 * 47 + 63, 1000, 1001, 1002], dtype=np.int32)
    img = integ.renderC(sc, 0, seed=3, batch_pix=torch_cuda.from_numpy(pix)).cpu().numpy()
    want = ref.render_c(max_depth=2, seed=3, pix_ids=pix)
    assert img.shape == (len(pix), 3) and product.rel_l2(img, want) < TOL
    with pytest.raises(RuntimeError):           # "While using batch rendering, seed must be set!"
        integ.renderC(sc, 0, seed=-1, batch_pix=torch_cuda.from_numpy(pix))
    with pytest.raises(RuntimeError):           # "Invalid sensor id!"
        integ.renderC(sc, 3, seed=1)


@pytest.mark.parametrize("param", ["light_x", "box_x", "albedo", "radiance", "camera_x"])
def test_render_d_terms_match_oracle(torch_cuda, psdr, orc, param):
    spec = scenes.cbox_scene(64, 64, 8, 8, 8, param=param)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(2)
    for terms in (orc.TERM_INTERIOR, orc.TERM_PRIMARY, orc.TERM_SECONDARY, orc.TERM_ALL):
        img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=21, terms=terms)
        wimg, wdimg = ref.render_d(max_depth=2, seeds=(21, 21, 21), terms=terms)
        if terms & orc.TERM_INTERIOR:
            assert product.rel_l2(img.cpu().numpy(), wimg) < TOL
        else:
            assert float(img.abs().max()) == 0.0          # edge terms have zero primal
        if np.abs(wdimg).max() > 0:
            assert product.rel_l2(dimg.cpu().numpy(), wdimg) < TOL, (param, terms)
        else:
            assert float(dimg.abs().max()) == 0.0


def test_render_d_sphere_scene(torch_cuda, psdr, orc):
    spec = scenes.sphere_scene(64, 64, 8, 8, 8)       # 652 triangles: BVH deeper than the box scene
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(1)
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=2)
    wimg, wdimg = ref.render_d(max_depth=1, seeds=(2, 2, 2))
    assert product.rel_l2(img.cpu().numpy(), wimg) < TOL and product.rel_l2(dimg.cpu().numpy(), wdimg) < TOL


def test_shards_sum_to_full_frame(torch_cuda, psdr, orc):
    """What the multi-GPU path relies on: disjoint lane ranges of the three samplers add up to the frame."""
    from psdr_jit_amd import cabi
    spec = scenes.cbox_scene(48, 48, 8, 8, 8)
    sc = product.build_scene(spec)
    n = 48 * 48
    def run(rank, count):
        buf = torch_cuda.empty((2, n, 3), dtype=torch_cuda.float32, device="cuda")
        a = cabi.make_args(max_depth=2, seeds=(4, 4, 4), shard_rank=rank, shard_count=count)
        cabi.check(cabi.lib().psdr_hip_render_d_fwd(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None))
        return buf.cpu().numpy()
    full = run(0, 1)
    parts = sum(run(r, 3) for r in range(3))
    assert product.rel_l2(parts[0], full[0]) < 1e-6 and product.rel_l2(parts[1], full[1]) < 1e-5
    ref = orc.OracleScene(spec, [0])
    w0, w1 = ref.render_d(max_depth=2, seeds=(4, 4, 4), shard_rank=1, shard_count=3)
    one = run(1, 3)
    assert product.rel_l2(one[0], w0) < TOL and product.rel_l2(one[1], w1) < TOL


def test_guiding_matches_oracle(torch_cuda, psdr, orc):
    spec = scenes.cbox_scene(48, 48, 4, 0, 8)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    integ = psdr.PathTracer(1)
    reso = [40, 4, 4, 16]
    integ.preprocess_secondary_edges(sc, 0, reso, 2, 5)
    mass = np.asarray(integ._guiding_mass(0)).reshape(-1)
    g = ref.guiding_build(0, reso, nrounds=2, seed=5)
    want = g.mass()
    assert mass.shape == want.shape and product.rel_l2(mass, want) < TOL
    _, dimg = psdr.render_d_fwd(integ, sc, 0, seed=6, terms=orc.TERM_SECONDARY)
    _, wd = ref.render_d(max_depth=1, seeds=(6, 6, 6), terms=orc.TERM_SECONDARY, guiding=g)
    assert product.rel_l2(dimg.cpu().numpy(), wd) < TOL


def test_counters_match_oracle_scale(torch_cuda, psdr):
    from psdr_jit_amd import cabi
    spec = scenes.cbox_scene(64, 64, 8, 0, 0)
    sc = product.build_scene(spec)
    out = torch_cuda.empty((64 * 64, 3), dtype=torch_cuda.float32, device="cuda")
    a = cabi.make_args(max_depth=3, seeds=(1, 0, 0))
    c = cabi.Counters()
    cabi.check(cabi.lib().psdr_hip_render_c_counted(sc._hip_handle(), C.byref(a), out.data_ptr(), C.byref(c), None))
    n = 64 * 64 * 8
    assert n <= c.rays <= 7 * n                      # 1 + 2*depth rays per lane at most
    # 36 triangles <= kBruteForceMax: every ray tests every triangle and visits no BVH node
    assert c.nodes_visited == 0 and c.tris_tested == 36 * c.rays and c.tris_tested >= c.shaded_hits
    plain = torch_cuda.empty_like(out)
    cabi.check(cabi.lib().psdr_hip_render_c(sc._hip_handle(), C.byref(a), plain.data_ptr(), None))
    assert torch_cuda.allclose(out, plain, rtol=1e-5, atol=1e-6)


def test_live_pixel_mask_is_conservative_and_exact(psdr, orc):
    """psdr_hip_scene_live_pixels: the interior term skips the samples of pixels no ray can leave towards a triangle (SensorDev::live, api.hip::build_live_mask).
    Conservative: every pixel in which some sample of a first-hit image lands on the scene is live - cameras and scenes of several kinds; not trivial: the README
    frame is mostly dead; exact: the frame equals the oracle's (which knows no mask) - and environment-lit scenes carry no mask."""
    import ctypes as C
    import torch
    from psdr_jit_amd import cabi
    L = cabi.lib()
    cases = [("cbox", scenes.cbox_scene(96, 80, 8, 0, 0, param="box_x"), 3), ("sphere", scenes.sphere_scene(64, 64, 8, 0, 0), 2),
             ("ortho", scenes.ortho_cbox_scene(72, 72, 8, 0, 0, param="box_x"), 2), ("microfacet", scenes.microfacet_cbox_scene(64, 48, 8, 0, 0, param="box_x"), 2),
             ("pervertex", scenes.pervertex_scene(64, 64, 8, 0, 0, param="ball_x"), 2)]
    fractions = {}
    for name, spec, depth in cases:
        sc = product.build_scene(spec)
        n = spec.width * spec.height
        bits = np.zeros((n + 31) // 32, np.uint32)
        n_live = C.c_int64(0)
        cabi.check(L.psdr_hip_scene_live_pixels(C.c_void_p(sc._hip_handle()), 0, bits.ctypes.data, C.byref(n_live)))
        live = ((bits[np.arange(n) >> 5] >> (np.arange(n) & 31).astype(np.uint32)) & 1).astype(bool)
        assert int(live.sum()) == n_live.value
        fractions[name] = n_live.value / n
        # the silhouette field (1 where a camera ray hits anything), 8 samples per pixel: no hit outside the mask
        a = cabi.make_args(max_depth=0, seeds=(3, 3, 3), terms=1, field=0)
        img = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        cabi.check(L.psdr_hip_render_c(C.c_void_p(sc._hip_handle()), C.byref(a), img.data_ptr(), None))
        hit = img.cpu().numpy()[:, 0] > 0
        assert hit.sum() > 0 and not np.any(hit & ~live), (name, int((hit & ~live).sum()))
        assert live.sum() <= 2.5 * hit.sum() + 4 * (spec.width + spec.height), (name, int(live.sum()), int(hit.sum()))        # ... and not much more than the hits
        # PathTracer frames equal the oracle's
        a = cabi.make_args(max_depth=depth, seeds=(5, 5, 5), terms=1)
        buf = torch.zeros((2, n, 3), dtype=torch.float32, device="cuda")
        cabi.check(L.psdr_hip_render_d_fwd(C.c_void_p(sc._hip_handle()), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None))
        wimg, wd = orc.OracleScene(spec, [0]).render_d(max_depth=depth, seeds=(5, 5, 5), terms=1)
        assert product.rel_l2(buf[0].cpu().numpy(), wimg) < TOL and product.rel_l2(buf[1].cpu().numpy(), wd) < TOL, name
    assert fractions["cbox"] < 0.9 or fractions["sphere"] < 0.9, fractions
    # a camera INSIDE the room: walls, floor and ceiling cross the camera plane (their images are not triangles) - no mask, and the frame still equals the oracle's;
    # a camera far away and off to the side: the room is a small part of the frame
    for name, move in (("inside", scenes.translate(278.0, 273.0, 150.0)), ("far", scenes.translate(-900.0, 273.0, -2500.0))):
        spec = scenes.cbox_scene(64, 64, 4, 0, 0, param="box_x")
        spec.cameras[0].to_world_raw = np.asarray(move, np.float32)
        sc = product.build_scene(spec)
        n = 64 * 64
        n_live = C.c_int64(0)
        cabi.check(L.psdr_hip_scene_live_pixels(C.c_void_p(sc._hip_handle()), 0, None, C.byref(n_live)))
        assert (n_live.value == n) if name == "inside" else (0 < n_live.value < n // 2), (name, n_live.value)
        a = cabi.make_args(max_depth=2, seeds=(9, 9, 9), terms=1)
        buf = torch.zeros((2, n, 3), dtype=torch.float32, device="cuda")
        cabi.check(L.psdr_hip_render_d_fwd(C.c_void_p(sc._hip_handle()), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None))
        wimg, wd = orc.OracleScene(spec, [0]).render_d(max_depth=2, seeds=(9, 9, 9), terms=1)
        assert np.abs(wimg).sum() > 0 and product.rel_l2(buf[0].cpu().numpy(), wimg) < TOL and product.rel_l2(buf[1].cpu().numpy(), wd) < TOL, name
    spec = scenes.envmap_scene(48, 48, 4, 0, 0)
    sc = product.build_scene(spec)
    n_live = C.c_int64(0)
    cabi.check(L.psdr_hip_scene_live_pixels(C.c_void_p(sc._hip_handle()), 0, None, C.byref(n_live)))
    assert n_live.value == 48 * 48
