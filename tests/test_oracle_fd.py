"""finite-difference validation of the oracle's full renderD
(interior + primary-edge + secondary-edge terms) against its own renderC with common random numbers.

Setup notes (measured, see LABNOTES.md "What the estimator does and does not differentiate"):
  * depth 1: the reference's secondary-edge term only covers DIRECT boundary segments (path.cpp:172-270
    never calls Li), so at depth >= 2 renderD is biased by design against a finite difference;
  * flat-shaded meshes: with interpolated shading normals the integrand n_s.w does not vanish at the
    geometric horizon of the shaded surface, which adds a discontinuity no edge term samples.
With both removed, renderD agrees with FD to within the FD's own Monte-Carlo noise.
"""
import numpy as np
import pytest

import scenes


@pytest.mark.slow
@pytest.mark.parametrize("param", ["light_x", "box_x"])
def test_render_d_matches_finite_differences(orc, param):
    W, spp, D, h = 24, 2048, 1, 0.01
    spec = scenes.cbox_scene(W, W, spp, spp, spp, param=param)
    for m in spec.meshes:
        m.use_face_normals = True
    sc = orc.OracleScene(spec, [0])
    _, d_int = sc.render_d(max_depth=D, seeds=(1, 2, 3), terms=orc.TERM_INTERIOR)
    _, d_pri = sc.render_d(max_depth=D, seeds=(1, 2, 3), terms=orc.TERM_PRIMARY)
    _, d_sec = sc.render_d(max_depth=D, seeds=(1, 2, 3), terms=orc.TERM_SECONDARY)
    fds = []
    for seed in (1, 7):
        ip = orc.OracleScene(scenes.set_param_value(spec, param, +h), [0]).render_c(max_depth=D, seed=seed)
        im = orc.OracleScene(scenes.set_param_value(spec, param, -h), [0]).render_c(max_depth=D, seed=seed)
        fds.append((ip - im) / (2 * h))
    fd = 0.5 * (fds[0] + fds[1])
    smooth = np.abs(d_pri).max(1) == 0          # pixels without a moving primary edge
    A, B = (d_int + d_sec)[smooth], fd[smooth]
    noise = np.linalg.norm(fds[0][smooth] - fds[1][smooth]) / np.linalg.norm(B)
    rel = np.linalg.norm(A - B) / np.linalg.norm(B)
    assert rel < max(0.12, 1.2 * noise), (rel, noise)
    # without the secondary-edge term the derivative is visibly wrong: the term is doing real work
    rel_int = np.linalg.norm(d_int[smooth] - B) / np.linalg.norm(B)
    assert rel_int > 2 * rel
    # primary-edge pixels: FD smears the jump over 2h, agreement is looser
    E = ~smooth
    assert E.sum() > 0
    tot = (d_int + d_pri + d_sec)[E]
    assert np.linalg.norm(tot - fd[E]) / np.linalg.norm(fd[E]) < 0.25


@pytest.mark.slow
def test_primary_edge_term_matches_finite_differences_under_a_uniform_environment(orc):
    """The primary-edge term ALONE against ground truth: the tutorial bunny (flat shaded, Diffuse) in front of a uniform
    environment map, translated in x.  The interior derivative of this scene vanishes up to shading noise, no emitter geometry moves,
    so the finite difference of renderC is the silhouette integral the primary-edge estimator samples: positive and negative sides
    within 5 % (measured 0.98 / 0.98).  (With interpolated normals the reference's estimator is 6-12 % above the finite
    difference - LABNOTES.md section 7 -; the figure Forward_AD_envmap.ipynb cell 10 shows is another 1.2 x above this
    restatement, which is why that figure's band in test_oracle_notebooks.py is wide.)"""
    from oracle.oracle import BsdfSpec, EmitterSpec
    res, h = 96, 5e-4

    def make(spp, sppe, param, P=0.0):
        spec = scenes.envmap_tutorial_scene(res, res, spp, sppe, 0, param=param, env_stride=8)
        spec.bsdfs = [BsdfSpec((0.5, 0.5, 0.5), name="bunny")]
        spec.emitters = [EmitterSpec(type=1, env_data=np.ones((16, 32, 3), np.float32), env_scale=1.0)]
        spec.meshes[0].use_face_normals = True
        if P != 0.0:
            spec.meshes[0].to_world_left = scenes.translate(100.0 * P, 0, 0)
        return spec
    a = orc.OracleScene(make(768, 0, None, +h), [0]).render_c(max_depth=1, seed=5)
    b = orc.OracleScene(make(768, 0, None, -h), [0]).render_c(max_depth=1, seed=5)
    fd = ((a.astype(np.float64) - b) / (2 * h)).reshape(res, res, 3).mean(axis=2)
    _, d = orc.OracleScene(make(0, 192, "bunny_x"), [0]).render_d(max_depth=1, seeds=(1, 2, 3))
    prim = d.reshape(res, res, 3).mean(axis=2).astype(np.float64)
    g = 12
    blk = lambda x: x.reshape(g, res // g, g, res // g).mean(axis=(1, 3))
    F, Pm = blk(fd), blk(prim)
    m = np.abs(F) > 0.1 * np.abs(F).max()
    scale = float((Pm[m] * F[m]).sum() / (F[m] ** 2).sum())
    corr = float(np.corrcoef(Pm[m], F[m])[0, 1])
    assert m.sum() >= 10 and corr > 0.97 and abs(scale - 1.0) < 0.06, (scale, corr, int(m.sum()))
