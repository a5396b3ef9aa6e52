"""How tightly can the oracle be pinned against the reference's notebook figures?  Each figure is ONE noisy realisation; the oracle
renders the same scene with N seeds and the seed-to-seed spread of every statistic (ncc, least-squares scale, per-colour-channel scale)
says what a realisation's noise contributes.  The figure's own noise contributes as much again, so the band a test may assert is
    |statistic - 1|  <  |mean - 1| + 3 * sqrt(2) * sigma_seed          (and ncc > mean - 3 * sqrt(2) * sigma_seed)
Prints one row per figure; tests/test_oracle_notebooks.py and tests/test_gpu_notebooks.py carry the bands this run produced (BANDS).

    python tools/notebook_bands.py [--seeds 8]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import notebook_refs as nr          # noqa: E402
import scenes                       # noqa: E402


channel_scales = nr.channel_scales


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--only-srgb", action="store_true", help="only the sRGB figures (per-channel scales)")
    args = ap.parse_args()
    from oracle import oracle as orc
    orc.build()
    import test_oracle_notebooks as T
    rows = {}

    def add(name, grid, img, channels=False):
        m = nr.compare(img, name, grid)
        r = rows.setdefault("%s@%d" % (name, grid), {"ncc": [], "scale": [], "ch": []})
        r["ncc"].append(m["ncc"]); r["scale"].append(m["scale"])
        if channels:
            r["ch"].append(channel_scales(img, name, grid))

    S1 = orc.OracleScene(scenes.sphere_scene(256, 256, 32, 32, 32), [0])
    S2 = orc.OracleScene(scenes.sphere_scene(256, 256, 0, 0, 4), [0])
    g = S2.guiding_build(0, [2000, 5, 5, 32], 1, seed=0, max_depth=1)
    S3 = orc.OracleScene(T.conductor_sphere_scene(), [0])
    pix = np.arange(300 * 400).reshape(300, 400)[150:250, 100:200].reshape(-1).astype(np.int32)
    S4 = orc.OracleScene(scenes.sphere_scene(512, 512, 4, 32, 0), [0])
    S4.set_field("silhouette", 1)
    masses = []
    for k in range(args.seeds):
        s = (101 + 3 * k, 202 + 3 * k, 303 + 3 * k)
        img, d = S1.render_d(max_depth=1, seeds=s)
        _, spec = nr.figure("Forward_AD_cell5"); spec["width"], spec["height"] = 256, 256
        _, spec = nr.figure("Forward_AD_cell6"); spec["width"], spec["height"] = 256, 256
        add("Forward_AD_cell5", 32, img, channels=True)
        if not args.only_srgb:
            add("Forward_AD_cell6", 32, d)
            add("Forward_AD_cell6", 64, d)
        for nm in ("secondary_edge_guiding_cell5", "secondary_edge_guiding_cell6"):
            _, spec = nr.figure(nm); spec["width"], spec["height"] = 256, 256
        if not args.only_srgb:
            _, d5 = S2.render_d(max_depth=1, seeds=s)
            add("secondary_edge_guiding_cell5", 32, d5)
            _, d6 = S2.render_d(max_depth=1, seeds=s, guiding=g)
            add("secondary_edge_guiding_cell6", 32, d6)
        add("batch_render_cell5", 32, S3.render_c(max_depth=2, seed=s[0]), channels=True)
        add("batch_render_cell6", 25, S3.render_c(max_depth=2, seed=s[1], pix_ids=pix), channels=True)
        if not args.only_srgb:
            _, dd = S4.render_d(max_depth=0, seeds=s)
            add("different_integrator_cell6", 32, dd)
            masses.append(nr.mass_ratio(dd, "different_integrator_cell6"))
        print("seed set %d done" % k, flush=True)
    out = {}
    k3 = 3.0 * np.sqrt(2.0)
    print("%-38s %9s %9s %9s %9s   %s" % ("figure@grid", "ncc mean", "ncc sd", "scale", "scale sd", "band: ncc > / |scale - 1| <"))
    for name, r in rows.items():
        ncc, sc = np.asarray(r["ncc"]), np.asarray(r["scale"])
        band_ncc = float(ncc.mean() - k3 * ncc.std(ddof=1))
        band_sc = float(abs(sc.mean() - 1.0) + k3 * sc.std(ddof=1))
        out[name] = {"ncc_mean": float(ncc.mean()), "ncc_sd": float(ncc.std(ddof=1)), "scale_mean": float(sc.mean()), "scale_sd": float(sc.std(ddof=1)),
                     "ncc_min_band": band_ncc, "scale_band": band_sc}
        line = "%-38s %9.4f %9.4f %9.4f %9.4f   %.4f / %.4f" % (name, ncc.mean(), ncc.std(ddof=1), sc.mean(), sc.std(ddof=1), band_ncc, band_sc)
        if r["ch"]:
            ch = np.asarray(r["ch"])
            out[name]["channel_scale_mean"] = ch.mean(axis=0).tolist(); out[name]["channel_scale_sd"] = ch.std(axis=0, ddof=1).tolist()
            out[name]["channel_band"] = float(np.max(np.abs(ch.mean(axis=0) - 1.0) + k3 * ch.std(axis=0, ddof=1)))
            line += "   rgb scale %s sd %s band %.4f" % (np.round(ch.mean(axis=0), 4).tolist(), np.round(ch.std(axis=0, ddof=1), 4).tolist(), out[name]["channel_band"])
        print(line)
    m = np.asarray(masses if masses else [1.0, 1.0])
    out["different_integrator_cell6@mass"] = {"mass_mean": float(m.mean()), "mass_sd": float(m.std(ddof=1)), "mass_band": float(abs(m.mean() - 1) + k3 * m.std(ddof=1))}
    print("different_integrator_cell6 mass %.4f sd %.4f band %.4f" % (m.mean(), m.std(ddof=1), out["different_integrator_cell6@mass"]["mass_band"]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
