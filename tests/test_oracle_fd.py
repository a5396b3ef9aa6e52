"""finite-difference validation of the oracle's full renderD
(interior + primary-edge + secondary-edge terms) against its own renderC with common random numbers.

Setup notes (measured, see DESIGN.md "What the estimator does and does not differentiate"):
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
