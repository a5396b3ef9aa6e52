"""Comparison of a rendered image with the figure a reference notebook embeds (tests/golden/notebooks, extracted by
tests/golden/make_notebook_refs.py).  The figures are the only outputs of the reference that exist for this path;
their noise realisation differs from ours (other RNG streams, other hardware), so the comparison is on BLOCK MEANS of
what the figure displays: sRGB values for `plt.imshow(to_srgb(img))`, values clipped to [vmin, vmax] for the viridis maps.
"""
import json
import os

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "notebooks")
_cache = {}


def _load():
    if not _cache:
        _cache["meta"] = json.load(open(os.path.join(_DIR, "notebook_refs.json")))
        _cache["arr"] = np.load(os.path.join(_DIR, "notebook_refs.npz"))
    return _cache["meta"], _cache["arr"]


def logs():
    return _load()[0]["logs"]


def figure(name):
    meta, arr = _load()
    return np.asarray(arr[name]), meta["figures"][name]


def linear_to_srgb(l):
    """tutorials/image_util.py to_srgb (the notebooks display to_srgb(img)); restated: the IEC 61966-2-1 curve, clipped"""
    l = np.asarray(l, np.float64)
    s = np.where(l <= 0.00313066844250063, l * 12.92, 1.055 * np.maximum(l, 1e-30) ** (1.0 / 2.4) - 0.055)
    return np.clip(s, 0.0, 1.0)


def displayed(img, spec):
    """what the notebook's imshow call shows of `img` ([H*W, 3] or [H, W, 3] linear radiance / derivative)"""
    H, W = spec["height"], spec["width"]
    a = np.asarray(img, np.float64).reshape(H, W, -1)
    if spec["kind"] == "srgb":
        return linear_to_srgb(a)
    return np.clip(a.mean(axis=2), spec["vmin"], spec["vmax"])


def block_means(ours, ref, spec, grid):
    """Means over a grid x grid partition of the axes area: ours [H, W(, 3)] in data pixels, ref in display pixels."""
    x0, x1, y0, y1 = spec["axes_rect"]
    ox, oy = spec["inner_origin"]
    H, W = ours.shape[:2]
    h, w = ref.shape[:2]
    # data coordinate of each display pixel centre: the spine centres sit at -0.5 and W - 0.5 (H - 0.5)
    cx = -0.5 + (ox + np.arange(w) - x0) / float(x1 - x0) * W
    cy = -0.5 + (oy + np.arange(h) - y0) / float(y1 - y0) * H
    bx_ref = np.clip(((cx + 0.5) / W * grid).astype(int), 0, grid - 1)
    by_ref = np.clip(((cy + 0.5) / H * grid).astype(int), 0, grid - 1)
    bx_our = np.clip(((np.arange(W) + 0.5) / W * grid).astype(int), 0, grid - 1)
    by_our = np.clip(((np.arange(H) + 0.5) / H * grid).astype(int), 0, grid - 1)

    def reduce(a, by, bx):
        a = a.reshape(a.shape[0], a.shape[1], -1)
        out = np.zeros((grid, grid, a.shape[2]))
        cnt = np.zeros((grid, grid, 1))
        np.add.at(out, (by[:, None], bx[None, :]), a)
        np.add.at(cnt, (by[:, None], bx[None, :]), 1.0)
        return out / np.maximum(cnt, 1.0)

    return reduce(np.asarray(ours, np.float64), by_our, bx_our), reduce(np.asarray(ref, np.float64), by_ref, bx_ref)


def compare(img, name, grid=32):
    """-> dict(mean_abs, max_abs, ncc, scale): block-mean differences in displayed units, the normalised cross-correlation
    of the block means (after removing each side's mean for sRGB images; about zero for derivative maps) and the
    least-squares scale ours ~ scale * ref."""
    ref, spec = figure(name)
    ours = displayed(img, spec)
    a, b = block_means(ours, ref, spec, grid)
    d = np.abs(a - b)
    if spec["kind"] == "srgb":
        a0, b0 = a - a.mean(), b - b.mean()
    else:
        a0, b0 = a, b
    ncc = float((a0 * b0).sum() / max(np.sqrt((a0 * a0).sum() * (b0 * b0).sum()), 1e-30))
    scale = float((a0 * b0).sum() / max((b0 * b0).sum(), 1e-30))
    return dict(mean_abs=float(d.mean()), max_abs=float(d.max()), ncc=ncc, scale=scale, range=float(spec.get("vmax", 1.0)))


def mass_ratio(img, name):
    """sum of |displayed value| over the frame, ours / reference (the reference's sum is taken over its display pixels and scaled
    by the number of data pixels per display pixel) - the comparison for one-pixel outlines, where a least-squares scale on
    block means is diluted by every block the two outlines share only partly"""
    ref, spec = figure(name)
    ours = displayed(img, spec)
    x0, x1, y0, y1 = spec["axes_rect"]
    per_px = (spec["width"] / float(x1 - x0)) * (spec["height"] / float(y1 - y0))       # data pixels per display pixel
    return float(np.abs(ours).sum() / (np.abs(ref).sum() * per_px))


def channel_scales(img, name, grid=32, sat=0.8):
    """least-squares scale per colour channel of an sRGB figure, ours ~ scale_c * ref per channel (block means, each side's mean removed), over the blocks
    that are not saturated in the figure and do not touch a saturated block: a light source that clips to 1.0 turns a sub-pixel registration offset of the figure's
    axes into +-0.07 on its two rows of blocks, which is all the blue channel of a dim scene has to fit"""
    ref, spec = figure(name)
    a, b = block_means(displayed(img, spec), ref, spec, grid)
    hot = b.max(axis=2) > sat
    near = hot.copy()
    near[1:, :] |= hot[:-1, :]; near[:-1, :] |= hot[1:, :]; near[:, 1:] |= hot[:, :-1]; near[:, :-1] |= hot[:, 1:]
    keep = ~near
    out = []
    for c in range(3):
        a0, b0 = a[..., c][keep], b[..., c][keep]
        a0, b0 = a0 - a0.mean(), b0 - b0.mean()
        out.append(float((a0 * b0).sum() / max((b0 * b0).sum(), 1e-30)))
    return out


# The bands the notebook pins assert (tests/test_oracle_notebooks.py on the oracle, tests/test_gpu_notebooks.py on the HIP path), from the seed-to-seed spread of each
# statistic over 8 oracle renders (tools/notebook_bands.py): |scale - 1| < |mean - 1| + 3 sqrt(2) sigma - the figure is one noisy realisation too - rounded up by about a
# half for other resolutions and seeds.  Measured (mean +- sigma over seeds) -> 3-sigma band -> asserted:
#   figure                          ncc                 scale               3-sigma band      per-channel scale (r, g, b; blocks next to a saturated light left out)
#   Forward_AD_cell5 @32            0.9977 +- 0.00003   0.9987 +- 0.0002    0.0020            1.0035 1.0025 1.0038 +- 0.0002 -> 0.0048
#   Forward_AD_cell6 @32            0.9981 +- 0.0001    0.9966 +- 0.0007    0.0063
#   Forward_AD_cell6 @64            0.9954 +- 0.0001    0.9954 +- 0.0007    0.0076
#   secondary_edge_guiding_cell5    0.9879 +- 0.0014    1.0061 +- 0.0127    0.0599            (4 samples per pixel, unguided: the figure's own noise is the band)
#   secondary_edge_guiding_cell6    0.9936 +- 0.0002    1.0092 +- 0.0027    0.0207
#   batch_render_cell5 @32          0.9947 +- 0.0001    0.9897 +- 0.0005    0.0124            1.0082 1.0043 1.0041 +- 0.0011 -> 0.0127
#   batch_render_cell6 @25          0.9959 +- 0.0005    1.0021 +- 0.0013    0.0075            0.9987 1.0021 1.0016 +- 0.0017 -> 0.0088
#   different_integrator_cell6      0.9701 +- 0.0003    mass 0.9972 +- 0.0011 -> 0.0074
# (batch_render_cell5's whole-frame scale of 0.990 is the luminaire: its two rows of blocks read -0.07 / +0.08 in all three channels, a registration offset of a fraction of a
#  display pixel between the figure's axes and the data grid; with those blocks left out the three channels read 1.004-1.008.)
BANDS = {
    "Forward_AD_cell5": dict(grid=32, ncc=0.9965, scale=0.004, channel=0.008),
    "Forward_AD_cell6": dict(grid=32, ncc=0.9965, scale=0.010),
    "Forward_AD_cell6@64": dict(grid=64, ncc=0.993, scale=0.012),
    "secondary_edge_guiding_cell5": dict(grid=32, ncc=0.975, scale=0.07),
    "secondary_edge_guiding_cell6": dict(grid=32, ncc=0.990, scale=0.03),
    "batch_render_cell5": dict(grid=32, ncc=0.9925, scale=0.018, channel=0.020),
    "batch_render_cell6": dict(grid=25, ncc=0.991, scale=0.012, channel=0.015),
    "different_integrator_cell6": dict(grid=32, ncc=0.965, mass=0.012),
}


def check_band(img, key):
    """asserts BANDS[key] for `img` against the figure key names (key may carry an @grid suffix); -> the measured statistics"""
    name = key.split("@")[0]
    band = BANDS[key]
    img = np.asarray(img.detach().cpu().numpy() if hasattr(img, "detach") else img)
    m = compare(img, name, band["grid"])
    assert m["ncc"] > band["ncc"], (key, m, band)
    if "scale" in band:
        assert abs(m["scale"] - 1.0) < band["scale"], (key, m, band)
    if "channel" in band:
        ch = channel_scales(img, name, band["grid"])
        m["channel_scales"] = ch
        assert max(abs(c - 1.0) for c in ch) < band["channel"], (key, ch, band)
    if "mass" in band:
        m["mass"] = mass_ratio(img, name)
        assert abs(m["mass"] - 1.0) < band["mass"], (key, m, band)
    return m
