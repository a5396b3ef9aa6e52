"""Randomised reverse-mode sweep: <w, d_img> with d_img from the CPU oracle's forward mode against <J^T w, v> from psdr_hip_render_d_bwd, over scene families,
parameters, terms, depths, frame sizes and seeds nobody picked by hand.   python tools/fuzz_reverse.py [cases=100] [seed=1]      (exit code 1 if a case is off by more than 1e-3 of its scale)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__; __graft_entry__.build()
import psdr_jit_amd
from psdr_jit_amd import cabi
from oracle import oracle as orc
import scenes
import test_gpu_adjoint as t

FAMILIES = [
    ("cbox", lambda w, h, s, rng: (scenes.cbox_scene(w, h, s, s, s, param=rng.choice(["light_x", "box_x", "albedo", "radiance"])), {})),
    ("cbox_camera", lambda w, h, s, rng: (scenes.cbox_scene(w, h, s, s, s, param="camera_x"), {"with_camera": True})),
    ("sphere", lambda w, h, s, rng: (scenes.sphere_scene(w, h, s, s, s), {})),
    ("envmap", lambda w, h, s, rng: (scenes.envmap_scene(w, h, s, s, s, param=rng.choice(["albedo", "box_x", "box_rot", "box_rot_x", "box_rot_z"]), area_light=bool(rng.integers(2)), balls=bool(rng.integers(2))), {})),
    ("microfacet", lambda w, h, s, rng: (scenes.microfacet_cbox_scene(w, h, s, s, s, param=rng.choice(["roughness", "specular", "diffuse", "box_x"]), two_sided=bool(rng.integers(2))), {"with_mat": True})),
    ("conductor", lambda w, h, s, rng: (scenes.conductor_cbox_scene(w, h, s, s, s, param=rng.choice(["alpha", "eta", "k", "box_x"])), {"with_mat": True})),
    ("dielectric", lambda w, h, s, rng: (scenes.dielectric_cbox_scene(w, h, s, s, s, param=rng.choice(["alpha", "eta", "box_x"])), {"with_mat": True})),
    ("textured", lambda w, h, s, rng: (scenes.textured_scene(w, h, s, s, s, param=rng.choice(["texture", "box_x"])), {})),
    ("pervertex", lambda w, h, s, rng: (scenes.pervertex_scene(w, h, s, s, s, param=rng.choice(["diffuse", "specular", "roughness", "ball_x"])), {"with_mat": True})),
    ("normalmap", lambda w, h, s, rng: (scenes.normalmap_scene(w, h, s, s, s, param=rng.choice(["nmap", "nested", "box_x"]), nested=rng.choice(["microfacet", "diffuse"])), {"with_mat": True})),
    ("ortho", lambda w, h, s, rng: (scenes.ortho_cbox_scene(w, h, s, s, s, param="box_x"), {})),
]
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
env = (torch, psdr_jit_amd, cabi)
bad, worst = 0, {}
for case in range(n_cases):
    name, make = FAMILIES[case % len(FAMILIES)]
    w, h = int(rng.integers(17, 49)), int(rng.integers(17, 49))
    spp = int(rng.choice([2, 4, 8]))
    depth = int(rng.integers(1, 5))
    terms = int(rng.choice([1, 2, 4, 7]))
    seeds = tuple(int(x) for x in rng.integers(0, 2 ** 31, size=3))
    spec, kw = make(w, h, spp, rng)
    lhs, rhs, scale = t._dot_product_case(env, spec, depth=depth, terms=terms, seeds=seeds, oracle=orc, **kw)
    err = abs(lhs - rhs) / scale
    worst[name] = max(worst.get(name, 0.0), err)
    if not (err <= 1e-3):
        bad += 1
        print("FAIL", name, "%dx%d spp %d depth %d terms %d seeds %s: lhs %.6g rhs %.6g scale %.6g" % (w, h, spp, depth, terms, seeds, lhs, rhs, scale))
print("%d cases, %d failed; worst |lhs - rhs| / scale per family: %s" % (n_cases, bad, {k: float("%.2g" % v) for k, v in worst.items()}))
sys.exit(1 if bad else 0)
