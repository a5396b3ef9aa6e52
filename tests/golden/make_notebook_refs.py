"""Extracts the ONLY outputs the reference holds for the hot path - the figures and log lines embedded in
`/root/reference/tutorials/*.ipynb` - into data fixtures under tests/golden/notebooks/:

  <notebook>_cell<N>.png   the embedded matplotlib figure, byte for byte (provenance)
  notebook_refs.npz        per figure, the DATA the figure shows, decoded from the axes area:
                             kind "srgb":    float32 [h, w, 3] sRGB values in [0, 1]           (plt.imshow(to_srgb(img)))
                             kind "viridis": float32 [h, w]    scalar in [vmin, vmax]          (plt.imshow(d, vmin, vmax, cmap=viridis))
                           at the figure's own display resolution (the 512-pixel images are shown 370 pixels wide)
  notebook_refs.json       per figure: notebook, cell, kind, vmin/vmax, data extent (the axis limits), plus the logged
                           integers / AABB / timings of Forward_AD.ipynb cell 4-5

Run in the build container (needs /root/reference, PIL and matplotlib - the fixtures themselves need neither):

    python tests/golden/make_notebook_refs.py

Nothing of the reference's SOURCE is copied: the notebooks' code cells are not stored, only their rendered outputs.
"""
import base64
import io
import json
import os
import re

import numpy as np

REF = "/root/reference/tutorials"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "notebooks")

# (notebook, cell) -> what the cell's imshow call displays (read off the notebook's code cell)
FIGURES = {
    ("Forward_AD", 5): dict(kind="srgb", width=512, height=512),
    ("Forward_AD", 6): dict(kind="viridis", vmin=-0.1, vmax=0.1, width=512, height=512),
    ("Forward_AD_envmap", 6): dict(kind="srgb", width=128, height=128),
    ("Forward_AD_envmap", 8): dict(kind="viridis", vmin=-50.0, vmax=50.0, width=128, height=128),
    ("Forward_AD_envmap", 10): dict(kind="viridis", vmin=-50.0, vmax=50.0, width=128, height=128),
    ("Forward_AD_envmap", 12): dict(kind="viridis", vmin=-50.0, vmax=50.0, width=128, height=128),
    ("batch_render", 5): dict(kind="srgb", width=400, height=300),
    ("batch_render", 6): dict(kind="srgb", width=100, height=100),
    ("different_integrator", 6): dict(kind="viridis", vmin=-0.1, vmax=0.1, width=512, height=512),
    ("secondary_edge_guiding", 5): dict(kind="viridis", vmin=-0.2, vmax=0.2, width=512, height=512),
    ("secondary_edge_guiding", 6): dict(kind="viridis", vmin=-0.2, vmax=0.2, width=512, height=512),
}


def axes_rect(rgba):
    """Bounding box (x0, x1, y0, y1; inclusive spine positions) of the image axes: the leftmost pair of long vertical
    black spines and the horizontal spines joining them."""
    dark = (rgba[..., :3].astype(np.int32).sum(axis=2) < 130) & (rgba[..., 3] > 200)

    def first_run(line, min_len):
        """(start, end) of the first run of True values at least min_len long, or None"""
        start = None
        for i, v in enumerate(list(line) + [False]):
            if v and start is None:
                start = i
            elif not v and start is not None:
                if i - start >= min_len:
                    return start, i - 1
                start = None
        return None

    # the top spine: the first row that holds a long dark run.  Its square caps overshoot the corner by a pixel or two,
    # so the left / right spines are the outermost columns near the run's ends that hold a long vertical run.
    for y0 in range(dark.shape[0]):
        run = first_run(dark[y0], 100)
        if run is not None:
            break
    xa, xb = run
    x0 = next(x for x in range(xa, xa + 6) if first_run(dark[:, x], 100) is not None)
    x1 = next(x for x in range(xb, xb - 6, -1) if first_run(dark[:, x], 100) is not None)
    ya, yb = first_run(dark[:, x0], 100)
    y0 = next(y for y in range(ya, ya + 6) if first_run(dark[y], 100) is not None)
    y1 = next(y for y in range(yb, yb - 6, -1) if first_run(dark[y], 100) is not None)
    return x0, x1, y0, y1


def viridis_lut():
    import matplotlib
    # matplotlib writes the colours as (rgba * 255).astype(uint8): truncated, not rounded
    return (np.asarray(matplotlib.colormaps["viridis"](np.arange(256))[:, :3]) * 255.0).astype(np.int32)


def decode_viridis(rgb, vmin, vmax, lut):
    flat = rgb.reshape(-1, 3).astype(np.int32)
    idx = np.empty(len(flat), dtype=np.int32)
    for s in range(0, len(flat), 65536):
        d = ((flat[s:s + 65536, None, :] - lut[None, :, :]) ** 2).sum(axis=2)
        idx[s:s + 65536] = d.argmin(axis=1)
    # colour i covers [i, i+1) / 256 of the range; the two end colours also hold everything clipped beyond
    val = vmin + (idx.astype(np.float64) + 0.5) / 256.0 * (vmax - vmin)
    val[idx == 0] = vmin
    val[idx == 255] = vmax
    # colour 128 holds [0, step): the maps' background is exactly zero, so it is decoded as 0 rather than step / 2
    if vmin == -vmax:
        val[idx == 128] = 0.0
    return val.reshape(rgb.shape[:2]).astype(np.float32)


def main():
    from PIL import Image
    os.makedirs(OUT, exist_ok=True)
    lut = viridis_lut()
    arrays, meta = {}, {"figures": {}, "logs": {}}
    for nbname in sorted({k[0] for k in FIGURES}):
        nb = json.load(open(os.path.join(REF, nbname + ".ipynb")))
        for ci, cell in enumerate(nb["cells"]):
            text = "".join("".join(o.get("text", [])) for o in cell.get("outputs", []) if o.get("output_type") == "stream")
            if nbname == "Forward_AD" and text:
                log = meta["logs"]
                m = re.search(r"AABB: \[lower = \[\[([^\]]*)\]\], upper = \[\[([^\]]*)\]\]\]", text)
                if m:
                    log["aabb_lower"] = [float(x) for x in m.group(1).split(",")]
                    log["aabb_upper"] = [float(x) for x in m.group(2).split(",")]
                for key, pat in (("primary_edges", r"\((\d+)\) primary edges"), ("secondary_edges", r"(\d+) secondary edges"),
                                 ("render_seconds", r"Rendered in ([0-9.]+) seconds"), ("configure_seconds", r"Configured in ([0-9.]+) seconds")):
                    for v in re.findall(pat, text):
                        log.setdefault(key, []).append(float(v) if "." in v else int(v))
            for o in cell.get("outputs", []):
                png = o.get("data", {}).get("image/png")
                if png is None or (nbname, ci) not in FIGURES:
                    continue
                raw = base64.b64decode(png)
                name = "%s_cell%d" % (nbname, ci)
                with open(os.path.join(OUT, name + ".png"), "wb") as fh:
                    fh.write(raw)
                rgba = np.asarray(Image.open(io.BytesIO(raw)).convert("RGBA"))
                x0, x1, y0, y1 = axes_rect(rgba)
                inner = rgba[y0 + 2:y1 - 1, x0 + 2:x1 - 1, :3]      # the pixels next to a spine are blended with it
                spec = dict(FIGURES[(nbname, ci)])
                if spec["kind"] == "srgb":
                    arrays[name] = (inner.astype(np.float32) / 255.0)
                else:
                    arrays[name] = decode_viridis(inner, spec["vmin"], spec["vmax"], lut)
                spec.update(notebook=nbname + ".ipynb", cell=ci, axes_rect=[x0, x1, y0, y1], inner_origin=[x0 + 2, y0 + 2], shape=list(arrays[name].shape))
                meta["figures"][name] = spec
                print(name, "axes", (x0, x1, y0, y1), "->", arrays[name].shape)
    np.savez_compressed(os.path.join(OUT, "notebook_refs.npz"), **arrays)
    with open(os.path.join(OUT, "notebook_refs.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print(json.dumps(meta["logs"]))


if __name__ == "__main__":
    main()
