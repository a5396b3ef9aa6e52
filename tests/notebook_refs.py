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
