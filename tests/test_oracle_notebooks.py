"""PIN of the oracle against the reference's own outputs: the figures and log lines the reference's five tutorial notebooks
embed (tests/golden/notebooks/, extracted from /root/reference/tutorials/*.ipynb by tests/golden/make_notebook_refs.py).
They are the only hot-path outputs the reference holds (it has no tests and cannot be built here).

The oracle renders each notebook's scene with the notebook's integrator and sample counts (at the notebook's resolution
or a reduced one - block means do not care) and is compared with what the figure DISPLAYS: sRGB pixel values for the
primal images, derivative values clipped to the colour bar's range for the viridis maps, both as means over a grid of
blocks because the noise realisations differ (different RNG streams).  Stated metrics, with what the oracle measures today:

  The bands asserted are tests/notebook_refs.py::BANDS - three sigma of the seed-to-seed spread of each statistic over 8 oracle renders plus the same again for
  the figure's own noise (tools/notebook_bands.py), the measured values beside them: +-0.004 on the scale of Forward_AD cell 5 (+-0.008 per colour channel), +-0.010 /
  +-0.012 for cell 6 on a 32 x 32 / 64 x 64 grid, +-0.018 / +-0.012 for the two batch_render figures (+-0.020 / +-0.015 per channel), +-0.03 for the guided
  secondary-edge figure, +-0.07 for the unguided one (4 samples per pixel: its noise is the band), +-0.012 on the mass of the one-pixel silhouette outline.
  Forward_AD_envmap: see test_envmap_figures (the figure was rendered with another geometry term than the reference's source has today).

ncc = normalised cross-correlation of the block means, scale = least-squares factor ours ~ scale * reference.
"""
import numpy as np
import pytest

import notebook_refs as nr
import scenes


def _check(img, name, grid, ncc, scale_tol, size=None):
    ref, spec = nr.figure(name)
    if size is not None:
        spec["width"], spec["height"] = size            # rendered smaller than the notebook: same field of view
    m = nr.compare(img, name, grid)
    assert m["ncc"] > ncc and abs(m["scale"] - 1.0) < scale_tol, (name, m)
    return m


def test_logged_integers_and_aabb(orc):
    """Forward_AD.ipynb cell 4/5 stdout: '[Scene] AABB: [lower = [[0, -9.1e-05, -500]], upper = [[556, 548.8, 559.2]]]',
    '(79) primary edges initialized', '990 secondary edges initialized'"""
    log = nr.logs()
    S = orc.OracleScene(scenes.sphere_scene(64, 64, 1, 1, 1), [0])
    assert S.num_primary_edges(0) == log["primary_edges"][-1] == 79
    assert S.num_sec_edges == log["secondary_edges"][-1] == 990
    lo, hi = S.aabb()
    # the log prints 6 significant digits
    assert np.allclose(lo, log["aabb_lower"], rtol=2e-6, atol=1e-6) and np.allclose(hi, log["aabb_upper"], rtol=2e-6, atol=1e-6), (lo, hi)


def test_forward_ad_figures(orc):
    """Forward_AD.ipynb cells 5-6: PathTracer(1).renderD, spp = sppe = sppse = 32, light + small sphere moving in x"""
    S = orc.OracleScene(scenes.sphere_scene(256, 256, 32, 32, 32), [0])
    img, d = S.render_d(max_depth=1, seeds=(1, 2, 3))
    for name in ("Forward_AD_cell5", "Forward_AD_cell6"):
        _, spec = nr.figure(name)
        spec["width"], spec["height"] = 256, 256        # rendered smaller than the notebook: same field of view
    nr.check_band(img, "Forward_AD_cell5")
    nr.check_band(d, "Forward_AD_cell6")
    nr.check_band(d, "Forward_AD_cell6@64")


def test_secondary_edge_guiding_figures(orc):
    """secondary_edge_guiding.ipynb cells 5-6: sppse = 4 alone, without and with preprocess_secondary_edges([2000,5,5,32], 1)"""
    S = orc.OracleScene(scenes.sphere_scene(256, 256, 0, 0, 4), [0])
    for name in ("secondary_edge_guiding_cell5", "secondary_edge_guiding_cell6"):
        _, spec = nr.figure(name)
        spec["width"], spec["height"] = 256, 256
    _, d = S.render_d(max_depth=1, seeds=(1, 2, 3))
    nr.check_band(d, "secondary_edge_guiding_cell5")
    g = S.guiding_build(0, [2000, 5, 5, 32], 1, seed=0, max_depth=1)
    _, d = S.render_d(max_depth=1, seeds=(1, 2, 3), guiding=g)
    nr.check_band(d, "secondary_edge_guiding_cell6")


def test_different_integrator_figure(orc):
    """different_integrator.ipynb cell 6: FieldExtractionIntegrator("silhouette 1") - the moving small sphere's mask; its
    derivative is the primary-edge term alone, a one-pixel outline saturated at the colour bar's +-0.1.  Rendered at the
    notebook's 512 x 512 because the outline's width in pixels sets the block means.  A one-pixel outline that sits half a pixel
    off the reference's shares most blocks only partly, which dilutes the least-squares scale (0.915 at ncc 0.97), so the
    magnitude is compared by MASS - the sum of |displayed value| over the frame, ours / reference = 0.997."""
    S = orc.OracleScene(scenes.sphere_scene(512, 512, 4, 32, 0), [0])
    S.set_field("silhouette", 1)
    _, d = S.render_d(max_depth=0, seeds=(1, 2, 3))
    nr.check_band(d, "different_integrator_cell6")


def conductor_sphere_scene(width=400, height=300, spp=32):
    """batch_render.ipynb cells 1-4: the sphere box with RoughConductorBSDF(alpha 0.01, eta, k) on the large sphere"""
    spec = scenes.sphere_scene(width, height, spp, 0, 0)
    b = spec.bsdfs[0]
    b.type, b.alpha_u, b.alpha_v = 2, 0.01, 0.01
    b.eta, b.k, b.specular = (0.155475, 0.116753, 0.138334), (4.83181, 3.12296, 2.1486), (1.0, 1.0, 1.0)
    for m in spec.meshes:
        m.d_to_world_left = np.zeros((4, 4), np.float32)
    return spec


def test_batch_render_figures(orc):
    """batch_render.ipynb cells 5-6: PathTracer(2).renderC, 400 x 300, spp 32; then renderC(seed=0, batch_pix=crop)"""
    S = orc.OracleScene(conductor_sphere_scene(), [0])
    img = S.render_c(max_depth=2, seed=5)
    nr.check_band(img, "batch_render_cell5")
    pix = np.arange(300 * 400).reshape(300, 400)[150:250, 100:200].reshape(-1).astype(np.int32)
    part = S.render_c(max_depth=2, seed=0, pix_ids=pix)
    nr.check_band(part, "batch_render_cell6")


def envelope(d, name, grid):
    """ncc / scale of the block means of |displayed value| (values below one colour step count as zero on both sides)"""
    ref, spec = nr.figure(name)
    ours = np.abs(nr.displayed(d, spec))
    step = (spec["vmax"] - spec["vmin"]) / 256.0
    refa = np.abs(ref)
    refa[refa < step] = 0.0
    ours[ours < step] = 0.0
    a, b = nr.block_means(ours, refa, spec, grid)
    return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum())), float((a * b).sum() / (b * b).sum())


def bunny_ratio(img, inner):
    """mean displayed value of our frame over the reference's inside the bunny (mask from the caller), frame 128 x 128"""
    ref, spec = nr.figure("Forward_AD_envmap_cell6")
    a, b = nr.block_means(nr.displayed(img, spec), ref, spec, 128)
    return float(a[inner].mean() / b[inner].mean())


@pytest.mark.slow
def test_envmap_figures(orc):
    """Forward_AD_envmap.ipynb: bunny_low.obj with MicrofacetBSDF([0.2, 0.9, 0.9], [0.01, 0.01, 0.01], 0.3) under ballroom_1k.exr,
    128 x 128, 128 samples per term.  cell 6 primal; cells 8 / 10 / 12 the interior / primary-edge / secondary-edge derivative
    alone, colour bar +-50.

    ROOT CAUSE of round 2's open item (the primary-edge outline read 0.80 x the figure, stably): the notebook's figures were
    rendered with the Schlick-k geometry term G1(c) = c / (c (1 - k) + k), k = (roughness + 1)^2 / 8 - the term the reference
    still carries in MicrofacetPerVertex::__eval (src/bsdf/microfacet_pv.cpp:49-63) - and not with the Smith term
    `distr.smith_g1(wi, H) * distr.smith_g1(wo, H)` that Microfacet::__eval has today (src/bsdf/microfacet.cpp:50).  The two
    classes' eval differ in nothing else (same GGX D, same 2^(...) Fresnel, same sampler and pdf), so the tutorial scene with its
    parameters on every vertex of a MicrofacetPerVertex IS the older Microfacet.  Evidence, all asserted below:
      * primal (cell 6): inside the bunny ours / figure = 1.053 with the Smith term, uniform over the colour channels (so not the
        Fresnel term) and growing towards grazing view angles (ref / ours 0.93 at cos 0.95 ... 0.53 at cos 0.25) - the signature of a
        masking term; 1.011 with the Schlick-k term; whole-frame ncc 0.9932 -> 0.9966;
      * primary-edge term (cell 10), signed 8 x 8 block means: scale 0.80 / ncc 0.978 with the Smith term, 0.96 / 0.993 with
        Schlick-k (a perfectly black bunny reads 1.006: the term is  background - rim radiance, and the rim is where the two
        geometry terms differ most);
      * the interior term's envelope (cell 8) stays inside its band with either (HDR texels in a glossy lobe at 128 samples).
    The product follows the reference's SOURCE (Smith); the figure pins everything around the BSDF through the per-vertex class."""
    from scipy import ndimage
    # coverage mask of the bunny (first hit on the mesh) from a black-body render
    spec = scenes.envmap_tutorial_scene(128, 128, 8, 0, 0, param=None, env_stride=8)
    spec.bsdfs[0].type, spec.bsdfs[0].reflectance = 0, (0.0, 0.0, 0.0)
    black = orc.OracleScene(spec, [0]).render_c(max_depth=1, seed=1)
    inner = ndimage.binary_erosion(np.asarray(black).reshape(128, 128, 3).sum(2) == 0, iterations=5)
    got = {}
    for schlick in (False, True):
        full = {}
        for cell, n in ((8, (128, 0, 0)), (10, (0, 128, 0)), (12, (0, 0, 128))):
            if cell == 12 and schlick:
                continue
            S = orc.OracleScene(scenes.envmap_tutorial_scene(128, 128, *n, param="bunny_x", env_stride=1, schlick_g=schlick), [0])
            img, d = S.render_d(max_depth=1, seeds=(1, 2, 3))
            full[cell] = d
            if cell == 8:
                full[6] = img
        got[schlick] = full
    # --- today's Microfacet::__eval (Smith): the frame as a whole agrees, the bunny is 5 % brighter (sRGB), the outline 0.80 x
    smith = got[False]
    _check(smith[6], "Forward_AD_envmap_cell6", 32, 0.985, 0.04)
    r = bunny_ratio(smith[6], inner)
    assert 1.03 < r < 1.08, r
    ncc, scale = envelope(smith[8], "Forward_AD_envmap_cell8", 16)
    assert ncc > 0.97 and 0.85 < scale < 1.1, (ncc, scale)
    m = nr.compare(smith[10], "Forward_AD_envmap_cell10", 8)
    assert m["ncc"] > 0.96 and 0.74 < m["scale"] < 0.86, m
    # --- the figure's own geometry term (MicrofacetPerVertex, Schlick-k): everything within the other figures' bands
    pv = got[True]
    m = _check(pv[6], "Forward_AD_envmap_cell6", 32, 0.995, 0.02)
    r = bunny_ratio(pv[6], inner)
    assert 0.98 < r < 1.03, r
    ncc, scale = envelope(pv[8], "Forward_AD_envmap_cell8", 16)
    assert ncc > 0.97 and 0.8 < scale < 1.1, (ncc, scale)
    m = nr.compare(pv[10], "Forward_AD_envmap_cell10", 8)
    assert m["ncc"] > 0.985 and abs(m["scale"] - 1.0) < 0.1, m
    m = nr.compare(pv[10], "Forward_AD_envmap_cell10", 32)
    assert m["ncc"] > 0.97 and abs(m["scale"] - 1.0) < 0.1, m
    # secondary term: isolated spikes on the bunny, nothing elsewhere
    ref, spec = nr.figure("Forward_AD_envmap_cell12")
    ours = nr.displayed(smith[12], spec)
    n_ours = int((np.abs(ours) > 1.0).sum())
    n_ref = (np.abs(ref) > 1.0).sum() * (128.0 / ref.shape[0]) * (128.0 / ref.shape[1])
    assert 0.5 * n_ref < n_ours < 2.0 * n_ref, (n_ours, n_ref)
    yy, xx = np.nonzero(np.abs(ours) > 1.0)
    assert yy.min() >= 30 and yy.max() <= 110 and xx.min() >= 20 and xx.max() <= 105


def test_product_host_matches_the_logged_scene_box_and_edge_counts():
    """the product's host model (no GPU needed) against the same Forward_AD.ipynb log lines"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import __graft_entry__
    __graft_entry__.build()
    import tutorials as tut
    sc = tut._scene(512, 512, 32, 32, 32)
    tut._camera(sc)
    tut._sphere_box(sc)
    sc._configure_host([])
    sc._configure_host([0])
    log = nr.logs()
    box = np.asarray(sc.aabb)
    assert np.allclose(box[0], log["aabb_lower"], rtol=2e-6, atol=1e-6) and np.allclose(box[1], log["aabb_upper"], rtol=2e-6, atol=1e-6), box
    assert sc._snapshot()["sec_edges"].shape[0] == log["secondary_edges"][-1] == 990
    assert sc.param_map["Sensor[0]"]._primary_edges(False).shape[0] == log["primary_edges"][-1] == 79
