"""Known-answer tests of the oracle's EnvironmentMap emitter (reference src/emitter/envmap.cpp, scene.cpp:434-515):
the oracle is unpinned against the reference (it cannot be built here), so its envmap path is pinned analytically."""
import numpy as np
import pytest

import scenes


def _latlong_lookup(img, d):
    """independent float64 restatement of EnvironmentMap::eval_direction + Bitmap::eval(envmap_mode) for unit vectors d [n,3]"""
    H, W, _ = img.shape
    u = np.arctan2(d[:, 0], -d[:, 2]) / (2 * np.pi)
    v = np.arccos(np.clip(d[:, 1], -1, 1)) / np.pi
    u -= np.floor(u); v -= np.floor(v)
    u -= 0.5 / W
    u -= np.floor(u)
    x, y = u * W, v * (H - 1)
    px, py = np.floor(x).astype(int), np.floor(y).astype(int)
    wx, wy = x - px, y - py
    yw = np.minimum(py, H - 2)
    x0, x1 = np.clip(px, 0, W - 1), (px + 1) % W
    v00, v10, v01, v11 = img[yw, x0], img[yw, x1], img[yw + 1, x0], img[yw + 1, x1]
    return ((1 - wx)[:, None] * v00 + wx[:, None] * v10) * (1 - wy)[:, None] + ((1 - wx)[:, None] * v01 + wx[:, None] * v11) * wy[:, None]


def test_constant_envmap_furnace(orc):
    L = np.array([0.6, 0.7, 0.9])
    spec = scenes.envmap_scene(32, 32, 256, 0, 0, param=None, env=scenes.synthetic_envmap(64, 32, sun=False), floor_only=True)
    sc = orc.OracleScene(spec, [0])
    img = sc.render_c(max_depth=1, seed=1).reshape(32, 32, 3)
    assert np.allclose(img[1, 16], L, rtol=1e-5)                      # camera ray that leaves the scene sees the map
    floor = img[14:17, 12:20].reshape(-1, 3).mean(axis=0)             # unoccluded diffuse floor: rho * L
    assert np.allclose(floor, 0.8 * L, rtol=0.02), floor
    # deeper paths add nothing on a lone floor (every bounce leaves the scene)
    img3 = sc.render_c(max_depth=3, seed=1).reshape(32, 32, 3)
    assert np.allclose(img3[14:17, 12:20].reshape(-1, 3).mean(axis=0), 0.8 * L, rtol=0.02)


def test_sun_envmap_irradiance_matches_quadrature(orc):
    env = scenes.synthetic_envmap(128, 64, sun=True)
    spec = scenes.envmap_scene(16, 16, 4096, 0, 0, param=None, env=env, floor_only=True)
    sc = orc.OracleScene(spec, [0])
    img = sc.render_c(max_depth=1, seed=3).reshape(16, 16, 3)
    # E = int L(w) cos(theta) dw over the upper hemisphere (floor normal +y), midpoint rule on a fine grid
    n_t, n_p = 1024, 2048
    th = (np.arange(n_t) + 0.5) * (0.5 * np.pi / n_t)
    ph = (np.arange(n_p) + 0.5) * (2 * np.pi / n_p)
    T, P = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], axis=-1).reshape(-1, 3)
    Lw = _latlong_lookup(env.astype(np.float64), d).reshape(n_t, n_p, 3)
    E = (Lw * (np.cos(T) * np.sin(T))[..., None]).sum(axis=(0, 1)) * (0.5 * np.pi / n_t) * (2 * np.pi / n_p)
    want = 0.8 / np.pi * E
    got = img[7:9, 6:10].reshape(-1, 3).mean(axis=0)
    assert np.allclose(got, want, rtol=0.03), (got, want)


def test_envmap_albedo_derivative_and_sampler_budget(orc):
    spec = scenes.envmap_scene(32, 32, 64, 0, 0, param="albedo")
    sc = orc.OracleScene(spec, [0])
    img, dimg = sc.render_d(max_depth=1, seeds=(2, 2, 2))
    # the image is exactly linear in the box albedo at depth 1 (same seeds = same paths): FD == derivative
    spec2 = scenes.envmap_scene(32, 32, 64, 0, 0, param="albedo")
    spec2.bsdfs[0].reflectance = (0.75, 0.75, 0.75)
    img2, _ = orc.OracleScene(spec2, [0]).render_d(max_depth=1, seeds=(2, 2, 2))
    fd = (img2 - img) / 0.25
    assert np.abs(dimg).sum() > 1.0
    assert np.allclose(dimg, fd, rtol=1e-4, atol=1e-5)
    assert np.all(np.isfinite(img)) and np.all(np.isfinite(dimg))
    # at depth 2 the floor picks up light from the boxes: the derivative spreads to floor pixels, still matches FD to O(h)
    img_a, d_a = sc.render_d(max_depth=2, seeds=(2, 2, 2))
    spec3 = scenes.envmap_scene(32, 32, 64, 0, 0, param="albedo")
    spec3.bsdfs[0].reflectance = (0.5 + 1e-2,) * 3
    img_b, _ = orc.OracleScene(spec3, [0]).render_d(max_depth=2, seeds=(2, 2, 2))
    assert np.allclose(d_a, (img_b - img_a) / 1e-2, rtol=2e-2, atol=2e-3)


def test_envmap_plus_area_light_runs_and_is_deterministic(orc):
    spec = scenes.envmap_scene(24, 24, 16, 8, 8, param="box_x", area_light=True)
    sc = orc.OracleScene(spec, [0])
    a, da = sc.render_d(max_depth=2, seeds=(4, 5, 6))
    b, db = sc.render_d(max_depth=2, seeds=(4, 5, 6))
    assert np.array_equal(a, b) and np.array_equal(da, db)
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(da)) and np.abs(da).sum() > 0


def test_textured_reflectance_matches_analytic_lookup(orc):
    """Bitmap3fD reflectance (bitmap.cpp:47-128, diffuse.cpp:38): a radially symmetric texture on a uv-mapped floor,
    seen from straight above under a constant map - every floor pixel must show rho(uv) * L."""
    W = H = 257
    u = np.arange(W) / (W - 1.0)
    v = np.arange(H) / (H - 1.0)
    uu, vv = np.meshgrid(u, v)
    r2 = (uu - 0.5) ** 2 + (vv - 0.5) ** 2
    tex = np.stack([0.2 + 1.2 * r2, 0.9 - 1.4 * r2, 0.5 + 0.0 * r2], axis=-1).astype(np.float32)
    res = 24
    spec = scenes.textured_scene(res, res, 256, 0, 0, texture=tex, box=False)
    spec.cameras[0].to_world_raw = scenes.translate(280.0, 700.0, 280.0) @ scenes._rot_x(np.radians(90.0))
    sc = orc.OracleScene(spec, [0])
    img = sc.render_c(max_depth=1, seed=5).reshape(res, res, 3)
    L = np.array([0.6, 0.7, 0.9])
    half = 700.0 * np.tan(np.radians(30.0))
    c = (np.arange(res) + 0.5 - res / 2) / (res / 2) * half
    dx, dz = np.meshgrid(c, c)
    inside = (np.abs(dx) < 270) & (np.abs(dz) < 270)
    rr = (dx / 560.0) ** 2 + (dz / 560.0) ** 2
    want = np.stack([0.2 + 1.2 * rr, 0.9 - 1.4 * rr, 0.5 + 0.0 * rr], axis=-1) * L
    assert inside.sum() > 100
    assert np.allclose(img[inside], want[inside], rtol=0.12, atol=0.0)            # per pixel: Monte Carlo noise at 256 spp
    ratio = (img[inside] / want[inside]).mean(axis=0)
    assert np.all(np.abs(ratio - 1.0) < 6e-3), ratio                                # per channel over ~250 pixels
    # a texture of equal texels is the constant reflectance
    spec_c = scenes.textured_scene(res, res, 16, 0, 0, texture=np.full((4, 4, 3), 0.5, np.float32), box=True)
    spec_k = scenes.textured_scene(res, res, 16, 0, 0, box=True)
    spec_k.bsdfs[0].texture = None
    a = orc.OracleScene(spec_c, [0]).render_c(max_depth=2, seed=2)
    b = orc.OracleScene(spec_k, [0]).render_c(max_depth=2, seed=2)
    assert np.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_texture_derivative_is_exact_at_depth_one(orc):
    """d image / d(all texels) with d texel = 1: the floor's reflectance moves by 1 everywhere -> FD in closed form"""
    spec = scenes.textured_scene(24, 24, 32, 0, 0, param="texture", box=True)
    img, dimg = orc.OracleScene(spec, [0]).render_d(max_depth=1, seeds=(3, 3, 3))
    spec2 = scenes.textured_scene(24, 24, 32, 0, 0, param="texture", box=True)
    spec2.bsdfs[0].texture = spec2.bsdfs[0].texture + np.float32(0.125)
    img2, _ = orc.OracleScene(spec2, [0]).render_d(max_depth=1, seeds=(3, 3, 3))
    assert np.abs(dimg).sum() > 1.0
    assert np.allclose(dimg, (img2 - img) / 0.125, rtol=1e-4, atol=1e-5)


def test_microfacet_bitmap_parameters(orc):
    """Microfacet with bitmap parameters (microfacet.cpp:38-45): maps of equal texels are the constants; the (value, tangent)
    arithmetic through the roughness / specular / diffuse maps equals finite differences of the maps"""
    spec_t = scenes.textured_microfacet_scene(24, 24, 16, 0, 0)
    b = spec_t.bsdfs[0]
    b.texture = np.full((4, 4, 3), 0.3, np.float32); b.spec_texture = np.full((3, 5, 3), 0.6, np.float32); b.rough_texture = np.full((5, 3), 0.4, np.float32)
    spec_k = scenes.textured_microfacet_scene(24, 24, 16, 0, 0)
    k = spec_k.bsdfs[0]
    k.texture = k.spec_texture = k.rough_texture = None
    k.reflectance, k.specular, k.roughness = (0.3, 0.3, 0.3), (0.6, 0.6, 0.6), 0.4
    a = orc.OracleScene(spec_t, [0]).render_c(max_depth=2, seed=2)
    c = orc.OracleScene(spec_k, [0]).render_c(max_depth=2, seed=2)
    assert a.max() > 0 and np.allclose(a, c, rtol=2e-5, atol=1e-6)
    for param, attr, h in (("roughness", "rough_texture", 1e-2), ("specular", "spec_texture", 0.0625), ("diffuse", "texture", 0.0625)):
        # emitter sampling only (Direct(0)): the sample placement must not depend on the roughness for FD to apply
        def scene_of(spec):
            s = orc.OracleScene(spec, [0])
            s.set_direct_mis(0)
            return s
        spec = scenes.textured_microfacet_scene(24, 24, 32, 0, 0, param=param)
        img, dimg = scene_of(spec).render_d(max_depth=1, seeds=(3, 3, 3))
        assert np.abs(dimg).sum() > 0.1, param
        up = scenes.textured_microfacet_scene(24, 24, 32, 0, 0)
        dn = scenes.textured_microfacet_scene(24, 24, 32, 0, 0)
        setattr(up.bsdfs[0], attr, getattr(up.bsdfs[0], attr) + np.float32(h))
        setattr(dn.bsdfs[0], attr, getattr(dn.bsdfs[0], attr) - np.float32(h))
        iu = scene_of(up).render_c(max_depth=1, seed=3)
        idn = scene_of(dn).render_c(max_depth=1, seed=3)
        fd = (iu - idn) / (2 * h)
        assert product_rel(dimg, fd) < (3e-2 if param == "roughness" else 2e-3), (param, product_rel(dimg, fd))


def product_rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-20))


def test_envmap_parameters_are_differentiable(orc):
    """m_radiance, m_scale, m_to_world_left of the EnvironmentMap (envmap.h:40-45).  The image is linear in the radiance: with
    d texel = texel, or d scale = 1 at scale 1, the derivative IS the image.  The rotation derivative is checked against central
    differences with BSDF sampling only (Direct(1)): the sample placement then does not depend on the map."""
    env = scenes.synthetic_envmap(32, 16)
    spec = scenes.envmap_scene(24, 24, 16, 0, 0, param=None, env=env)
    spec.emitters[0].d_env_data = env.copy()
    img, dimg = orc.OracleScene(spec, [0]).render_d(max_depth=2, seeds=(3, 3, 3))
    assert img.max() > 0 and np.allclose(dimg, img, rtol=2e-5, atol=1e-7)
    spec = scenes.envmap_scene(24, 24, 16, 0, 0, param=None, env=env)
    spec.emitters[0].d_env_scale = 1.0
    img, dimg = orc.OracleScene(spec, [0]).render_d(max_depth=2, seeds=(3, 3, 3))
    assert np.allclose(dimg, img, rtol=2e-5, atol=1e-7)
    # one texel alone: only pixels that can see / be lit by that direction change
    spec = scenes.envmap_scene(24, 24, 16, 0, 0, param=None, env=env)
    one = np.zeros_like(env); one[3, 5, 1] = 1.0
    spec.emitters[0].d_env_data = one
    _, d1 = orc.OracleScene(spec, [0]).render_d(max_depth=2, seeds=(3, 3, 3))
    assert d1[:, 1].max() > 0 and np.all(d1[:, 0] == 0) and np.all(d1[:, 2] == 0)       # a green texel moves the green channel only

    def rot_y(a):
        m = np.eye(4, dtype=np.float32)
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
        return m

    def render(angle, d=False):
        sp = scenes.envmap_scene(16, 16, 64, 0, 0, param=None, env=scenes.synthetic_envmap(64, 32))
        sp.emitters[0].env_to_world_left = rot_y(angle)
        if d:
            dm = np.zeros((4, 4), np.float32)
            dm[0, 0], dm[0, 2], dm[2, 0], dm[2, 2] = -np.sin(angle), np.cos(angle), -np.cos(angle), -np.sin(angle)
            sp.emitters[0].d_env_to_world_left = dm
        s = orc.OracleScene(sp, [0])
        s.set_direct_mis(1)
        return s.render_d(max_depth=1, seeds=(5, 5, 5))
    h = 2e-2
    _, dimg = render(0.3, d=True)
    up, _ = render(0.3 + h)
    dn, _ = render(0.3 - h)
    fd = (up - dn) / (2 * h)
    assert np.abs(fd).max() > 1e-3 and product_rel(dimg, fd) < 0.08, product_rel(dimg, fd)


def test_primary_edge_term_against_the_closed_form_under_an_hdr_step_background(orc):
    """Analytic pin of the primary-edge term's MAGNITUDE (integrator.cpp:179-198, perspective.cpp:200-226) under an HDR map.
    A black, flat-shaded square (half-size a, depth d in front of the camera, facing it) moves along x by 100 P in front of a
    map that is L1 = 100 on one side of the view axis and L2 = 1.5 on the other.  Only the two vertical edges move along their
    normal; behind each the map is constant, so with q = world_to_sample p (a pixel is 1/W x 1/H of the unit square):
        sum of d image / dP over the pixels of one edge = -/+ W H * (edge length in q) * (dq_x/dP) * L_behind
                                                        = -/+ W H * (2 a k) * (100 k) * L_behind,   k = 1 / (2 d tan(fov/2)),
    negative at the edge the square moves towards.  128 samples per pixel: 1 %."""
    W = H = 64
    a, d, fov, L_pos, L_neg = 20.0, 100.0, 60.0, 100.0, 1.5
    spec, want = scenes.hdr_step_square_scene(W, H, 0, 128, a, d, fov, L_pos, L_neg)
    k = 1.0 / (2.0 * d * np.tan(np.radians(fov) / 2.0))
    S = orc.OracleScene(spec, [0])
    assert S.num_primary_edges(0) == 4                      # the diagonal is coplanar (perspective.cpp:96-100)
    _, dimg = S.render_d(max_depth=1, seeds=(1, 2, 3))
    dimg = np.asarray(dimg, np.float64).reshape(H, W, 3)
    # which image half shows world x > 0: the primal behind the square's sides
    spec.spp, spec.sppe = 4, 0
    img = np.asarray(orc.OracleScene(spec, [0]).render_c(max_depth=1, seed=1)).reshape(H, W, 3)
    left_is_pos = img[H // 2, 2, 0] > 10.0
    assert abs(img[H // 2, 2, 0] - (L_pos if left_is_pos else L_neg)) < 1e-3 and abs(img[H // 2, W - 3, 0] - (L_neg if left_is_pos else L_pos)) < 1e-3
    halves = {True: dimg[:, : W // 2].sum(axis=(0, 1)), False: dimg[:, W // 2:].sum(axis=(0, 1))}
    got_pos, got_neg = halves[left_is_pos], halves[not left_is_pos]
    # the square moves towards +x: it covers the bright side (negative), uncovers the dim side (positive)
    assert np.allclose(got_pos, -want * L_pos, rtol=0.01), (got_pos, -want * L_pos)
    assert np.allclose(got_neg, +want * L_neg, rtol=0.01), (got_neg, want * L_neg)
    # the horizontal edges slide along themselves: rows above / below the square's vertical extent hold nothing else
    rows = np.abs(dimg).sum(axis=(1, 2))
    q_lo, q_hi = 0.5 - a * k, 0.5 + a * k
    inside = (np.arange(H) + 1.0) / H > q_lo - 1e-6
    inside &= (np.arange(H) + 0.0) / H < q_hi + 1e-6
    assert rows[~inside].sum() == 0.0


# ---------------------------------------------------------------- Bitmap::m_rot / m_scale / m_trans (bitmap.cpp:64-86)
def periodic_texture(n=33):
    """smooth texture whose last row / column repeats the first: the wrap `uv - floor(uv)` of a transformed lookup is continuous"""
    t = np.arange(n) / (n - 1.0)
    uu, vv = np.meshgrid(t, t)
    two_pi = 2.0 * np.pi
    return np.stack([0.5 + 0.3 * np.sin(two_pi * uu) * np.cos(two_pi * vv), 0.5 + 0.35 * np.cos(two_pi * (uu + vv)),
                     0.4 + 0.25 * np.sin(two_pi * 2 * uu) + 0.1 * np.cos(two_pi * vv)], axis=-1).astype(np.float32)


def test_bitmap_uv_transform_matches_the_analytic_lookup(orc):
    """A floor seen from straight above under a constant map shows rho(T(uv)) L, T = rotate about the centre, flip v, scale about
    the centre, translate (bitmap.cpp:64-74), for a texture given in closed form; the identity transform is the old lookup."""
    n = 129
    tex = periodic_texture(n)
    res = 24
    rot, scl, tx, ty = 0.4, 1.6, 0.21, -0.13
    spec = scenes.textured_scene(res, res, 256, 0, 0, texture=tex, box=False)
    spec.bsdfs[0].tex_xf = [[rot, scl, tx, ty], [0, 1, 0, 0], [0, 1, 0, 0]]
    spec.cameras[0].to_world_raw = scenes.translate(280.0, 700.0, 280.0) @ scenes._rot_x(np.radians(90.0))
    img = orc.OracleScene(spec, [0]).render_c(max_depth=1, seed=5).reshape(res, res, 3)
    spec_u = scenes.textured_scene(res, res, 256, 0, 0, texture=np.full((4, 4, 3), 1.0, np.float32), box=False)
    spec_u.cameras[0].to_world_raw = spec.cameras[0].to_world_raw
    white = orc.OracleScene(spec_u, [0]).render_c(max_depth=1, seed=5).reshape(res, res, 3)          # the same paths on a white floor
    # the uv of every pixel centre is not known in closed form here (camera orientation): recover it from a ramp texture
    spec_r = scenes.textured_scene(res, res, 256, 0, 0, texture=scenes.ramp_texture(), box=False)
    spec_r.cameras[0].to_world_raw = spec.cameras[0].to_world_raw
    ramp = orc.OracleScene(spec_r, [0]).render_c(max_depth=1, seed=5).reshape(res, res, 3)
    ok = white[..., 0] > 1e-3
    u = (ramp[..., 0] / np.where(ok, white[..., 0], 1.0) - 0.2) / 0.6          # ramp: r = 0.2 + 0.6 u, g = 0.8 - 0.5 v'  (v' = the flipped, wrapped v)
    vf = (0.8 - ramp[..., 1] / np.where(ok, white[..., 1], 1.0)) / 0.5
    v = np.where(vf > 0, 1.0 - vf, 0.0)                                         # flip_v: v' = -v - floor(-v) = 1 - v on (0, 1)
    x = (u - 0.5) * np.cos(rot) + (v - 0.5) * np.sin(rot) + 0.5
    y = -(-(u - 0.5) * np.sin(rot) + (v - 0.5) * np.cos(rot) + 0.5)
    off = -0.5 + scl / 2
    x, y = x * scl - off + tx, y * scl + off + ty
    two_pi = 2.0 * np.pi
    want = np.stack([0.5 + 0.3 * np.sin(two_pi * x) * np.cos(two_pi * y), 0.5 + 0.35 * np.cos(two_pi * (x + y)),
                     0.4 + 0.25 * np.sin(two_pi * 2 * x) + 0.1 * np.cos(two_pi * y)], axis=-1) * white
    inner = ok & (u > 0.05) & (u < 0.95) & (v > 0.05) & (v < 0.95)
    assert inner.sum() > 100
    # the pixel footprint averages the texture over ~1/24 of the floor: compare the means and the per-pixel values loosely
    err = np.abs(img[inner] - want[inner]).mean() / np.abs(want[inner]).mean()
    assert err < 0.06, err
    # identity transform given explicitly == no transform given
    spec_i = scenes.textured_scene(res, res, 8, 0, 0, texture=tex, box=True)
    a = orc.OracleScene(spec_i, [0]).render_c(max_depth=2, seed=2)
    spec_i.bsdfs[0].tex_xf = [[0, 1, 0, 0]] * 3
    b = orc.OracleScene(spec_i, [0]).render_c(max_depth=2, seed=2)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("comp", [0, 1, 2, 3])
def test_bitmap_uv_transform_tangents_match_finite_differences(orc, comp):
    """forward tangents of rotate / scale / translate.x / translate.y (the differentiable members m_rot, m_scale, m_trans,
    bitmap.h:37-39) against central differences of renderC with common random numbers; depth 2 with the box: the lookups of
    the second bounce carry the tangent too"""
    tex = periodic_texture(33)
    base = np.array([0.4, 1.6, 0.21, -0.13])
    h = 2e-3

    def make(delta, d):
        spec = scenes.textured_scene(24, 24, 64, 0, 0, texture=tex, box=True)
        xf = base.copy(); xf[comp] += delta
        spec.bsdfs[0].tex_xf = [xf.tolist(), [0, 1, 0, 0], [0, 1, 0, 0]]
        dx = np.zeros((3, 4)); dx[0, comp] = d
        spec.bsdfs[0].d_tex_xf = dx
        return orc.OracleScene(spec, [0])

    img, dimg = make(0.0, 1.0).render_d(max_depth=2, seeds=(3, 3, 3))
    fd = (make(+h, 0.0).render_c(max_depth=2, seed=3) - make(-h, 0.0).render_c(max_depth=2, seed=3)) / (2 * h)
    assert np.abs(fd).max() > 0.05
    rel = np.linalg.norm(dimg - fd) / np.linalg.norm(fd)
    assert rel < 0.03, rel            # bilinear interpolation: the tangent is piecewise constant, the difference quotient crosses texel borders


def test_envmap_radiance_translate_is_a_roll_of_the_texels(orc):
    """m_radiance.translate.x = k / W shifts the lat-long lookup (bitmap.cpp:74, envmap_mode :83-87) by k texels - the image, and the
    cell masses EnvironmentMap::configure builds through the same Bitmap::eval (envmap.cpp:28-31), are those of the rolled map"""
    env = scenes.synthetic_envmap(64, 32)
    k = 8
    spec_t = scenes.envmap_scene(24, 24, 16, 0, 0, param=None, env=env)
    spec_t.emitters[0].env_uv_xf = (0.0, 1.0, k / 64.0, 0.0)
    spec_r = scenes.envmap_scene(24, 24, 16, 0, 0, param=None, env=np.ascontiguousarray(np.roll(env, -k, axis=1)))
    a = orc.OracleScene(spec_t, [0]).render_c(max_depth=2, seed=3)
    b = orc.OracleScene(spec_r, [0]).render_c(max_depth=2, seed=3)
    assert np.linalg.norm(a - b) / np.linalg.norm(b) < 2e-3           # (u + k/W rounds differently from an index shift: a few samples change cells)
    # the translated map is not the original one
    c = orc.OracleScene(scenes.envmap_scene(24, 24, 16, 0, 0, param=None, env=env), [0]).render_c(max_depth=2, seed=3)
    assert np.linalg.norm(a - c) / np.linalg.norm(c) > 0.05
