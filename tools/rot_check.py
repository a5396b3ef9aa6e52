"""Diagnostic: dot-product identity with a ROTATION parameter (vertex normals carry a tangent) for the sweep and for the probe form.   python tools/rot_check.py [--pkg DIR]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--pkg" in sys.argv:
    sys.path.insert(0, os.path.abspath(sys.argv[sys.argv.index("--pkg") + 1]))
sys.path += [os.path.join(ROOT, "tests"), ROOT]
import numpy as np, torch
import psdr_jit_amd
from psdr_jit_amd import cabi
import product, scenes
import test_gpu_adjoint as t
env = (torch, psdr_jit_amd, cabi)
for balls, rot in ((False, "box_rot"), (True, "box_rot"), (False, "box_rot_x"), (True, "box_rot_x"), (False, "box_rot_z"), (True, "box_rot_z")):
    for probe in (False, True):
        if probe: os.environ["PSDR_ADJ_PROBE"] = "1"
        else: os.environ.pop("PSDR_ADJ_PROBE", None)
        spec = scenes.envmap_scene(40, 40, 8, 0, 0, param=rot, area_light=True, balls=balls)
        sc = product.build_scene(spec)
        d_tri = np.asarray(sc._snapshot()["d_triangles"])
        lhs, rhs, scale = t._dot_product_case(env, spec, depth=3, terms=1)
        print(rot, "balls", balls, "probe" if probe else "sweep", "lhs %.6f rhs %.6f err %.2e" % (lhs, rhs, abs(lhs - rhs) / scale), "| |d normals| %.3g |d positions| %.3g" % (np.abs(d_tri[:, 9:18]).sum(), np.abs(d_tri[:, :9]).sum()))
for probe in (False, True):
    if probe: os.environ["PSDR_ADJ_PROBE"] = "1"
    else: os.environ.pop("PSDR_ADJ_PROBE", None)
    spec = scenes.sphere_scene(40, 40, 8, 0, 0)
    lhs, rhs, scale = t._dot_product_case(env, spec, depth=3, terms=1)
    print("sphere box", "probe" if probe else "sweep", "lhs %.6f rhs %.6f err %.2e" % (lhs, rhs, abs(lhs - rhs) / scale))
