"""Cell masses of a 1024 x 512 environment map: psdr_hip_env_cell_masses (upload + kernel + download) vs the host loop behind
configure_host().   python tools/env_mass_timing.py"""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__; __graft_entry__.build()
import scenes, product
from psdr_jit_amd import cabi
W, H = 1024, 512
tex = np.random.default_rng(0).random((H, W, 3), dtype=np.float32)
mass = np.empty(4 * (W - 1) * (H - 1), np.float32)
L = cabi.lib()
for rep in range(6):
    t0 = time.perf_counter()
    cabi.check(L.psdr_hip_env_cell_masses(tex.ctypes.data_as(C.c_void_p), C.c_int32(W), C.c_int32(H), mass.ctypes.data_as(C.c_void_p)))
    dt = time.perf_counter() - t0
    if rep: print("device masses %dx%d: %.2f ms" % (W, H, dt * 1e3))
spec = scenes.config5_scene(64, 64, 1, 0, 0, level=2, env_res=(W, H))
for host_only in (True, False):
    t0 = time.perf_counter()
    product.build_scene(spec, host_only=host_only)
    print("scene build + %s: %.1f ms" % ("configure_host (host loop)" if host_only else "configure (device masses, upload, BVH)", (time.perf_counter() - t0) * 1e3))
