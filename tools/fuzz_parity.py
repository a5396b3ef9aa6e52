"""Randomised parity sweep: renderD (image + forward derivative, all three terms) of the HIP path against the CPU oracle over scene families, path depths,
frame sizes, sample counts and seeds nobody picked by hand.   python tools/fuzz_parity.py [cases=120] [seed=1]
Prints one line per failing case (relative L2 above 1e-3) and a summary; exit code 1 if any case fails."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__; __graft_entry__.build()
from psdr_jit_amd import cabi
from oracle import oracle as orc
import product, scenes

FAMILIES = [
    ("cbox", lambda w, h, s, rng: scenes.cbox_scene(w, h, s, s, s, param=rng.choice(["light_x", "box_x", "albedo", "radiance", "camera_x"]))),
    ("sphere", lambda w, h, s, rng: scenes.sphere_scene(w, h, s, s, s)),
    ("envmap", lambda w, h, s, rng: scenes.envmap_scene(w, h, s, s, s, param=rng.choice(["albedo", "box_x", "box_rot_x"]), area_light=bool(rng.integers(2)), balls=bool(rng.integers(2)))),
    ("microfacet", lambda w, h, s, rng: scenes.microfacet_cbox_scene(w, h, s, s, s, param=rng.choice(["roughness", "specular", "diffuse", "box_x"]), two_sided=bool(rng.integers(2)))),
    ("conductor", lambda w, h, s, rng: scenes.conductor_cbox_scene(w, h, s, s, s, param=rng.choice(["alpha", "eta", "k", "box_x"]))),
    ("dielectric", lambda w, h, s, rng: scenes.dielectric_cbox_scene(w, h, s, s, s, param=rng.choice(["alpha", "eta", "box_x"]))),
    ("textured", lambda w, h, s, rng: scenes.textured_scene(w, h, s, s, s, param=rng.choice(["texture", "box_x"]))),
    ("pervertex", lambda w, h, s, rng: scenes.pervertex_scene(w, h, s, s, s, param=rng.choice(["diffuse", "specular", "roughness", "ball_x"]))),
    ("normalmap", lambda w, h, s, rng: scenes.normalmap_scene(w, h, s, s, s, param=rng.choice(["nmap", "nested", "box_x"]), nested=rng.choice(["microfacet", "diffuse"]))),
    ("ortho", lambda w, h, s, rng: scenes.ortho_cbox_scene(w, h, s, s, s, param="box_x")),
]
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
worst = {}
for case in range(n_cases):
    name, make = FAMILIES[case % len(FAMILIES)]
    w, h = int(rng.integers(17, 57)), int(rng.integers(17, 57))
    spp = int(rng.choice([1, 2, 3, 5, 8]))
    depth = int(rng.integers(0, 5))
    seeds = tuple(int(x) for x in rng.integers(0, 2 ** 31, size=3))
    spec = make(w, h, spp, rng)
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    buf = torch.empty((2, w * h, 3), dtype=torch.float32, device="cuda")
    a = cabi.make_args(max_depth=depth, seeds=seeds, terms=7, skip_static_edges=bool(rng.integers(2)))      # (ABI 14: with or without the zero-velocity edge samples - the same image)
    cabi.check(cabi.lib().psdr_hip_render_d_fwd(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None))
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    wimg, wd = ref.render_d(max_depth=depth, seeds=seeds)
    e0 = product.rel_l2(got[0], wimg) if np.abs(wimg).max() > 0 else float(np.abs(got[0]).max())
    e1 = product.rel_l2(got[1], wd) if np.abs(wd).max() > 0 else float(np.abs(got[1]).max())
    ok = np.isfinite(got).all() and e0 < 1e-3 and e1 < 1e-3
    worst[name] = max(worst.get(name, 0.0), e0, e1)
    if not ok:
        bad += 1
        print("FAIL", name, "%dx%d spp %d depth %d seeds %s: image %.3g derivative %.3g finite %s" % (w, h, spp, depth, seeds, e0, e1, bool(np.isfinite(got).all())))
print("%d cases, %d failed; worst relative L2 per family: %s" % (n_cases, bad, {k: float("%.2g" % v) for k, v in worst.items()}))
sys.exit(1 if bad else 0)
