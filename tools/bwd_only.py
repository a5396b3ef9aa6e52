"""Reverse mode alone, for the profiler: N calls of psdr_hip_render_d_bwd (all three terms, every adjoint, nothing filtered) on the
bench workload - `c3` (README box, 512^2, 32/32/32, depth 3) or `c5` (BASELINE config 5's scene with its guiding grid [2000, 5, 5, 32] at [res] x [spp], default
the timed size 1024^2 x 64: what bench.py's `config5.backward` times) - so
that every kernel in a rocprofv3 trace of this command belongs to the backward pass.

    python tools/bwd_only.py c3 [calls]          python tools/bwd_only.py c5 [calls] [res] [spp]
"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import __graft_entry__; __graft_entry__.build()
from psdr_jit_amd import cabi
import product, scenes

what = sys.argv[1] if len(sys.argv) > 1 else "c3"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if what == "c3":
    res, spp = 512, 32
    spec = scenes.cbox_scene(res, res, spp, spp, spp, param="light_x")
else:
    res = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    spec = scenes.config5_scene(res, res, spp, spp, spp, level=6, env_res=(1024, 512))
sc = product.build_scene(spec)
guiding = None
if what != "c3":
    import psdr_jit_amd as psdr
    integ = psdr.PathTracer(3)
    integ.preprocess_secondary_edges(sc, 0, [2000, 5, 5, 32], 1, 0)
    guiding = integ._guiding_handle(0) or None
snap = sc._snapshot(); cam = sc.param_map["Sensor[0]"]
n = res * res
z = lambda *s: torch.zeros(s, device="cuda")
n_tri = np.asarray(snap["d_triangles"]).shape[0]
n_sec = max(1, np.asarray(snap["d_sec_edges"]).shape[0]); n_prim = max(1, np.asarray(cam._primary_edges(True)).shape[0])
g_tri, g_b, g_e, g_s, g_p = z(n_tri, 22), z(max(1, len(spec.bsdfs)), 3), z(max(1, len(spec.emitters)), 3), z(n_sec, 6), z(n_prim, 4)
g = cabi.Grads(g_tri.data_ptr(), g_b.data_ptr(), g_e.data_ptr(), g_s.data_ptr(), g_p.data_ptr())
keep = [g_tri, g_b, g_e, g_s, g_p]
envs = [e for e in spec.emitters if getattr(e, "env_data", None) is not None]
if envs:
    env = envs[0].env_data
    g_env, g_scale, g_xf, g_cam = z(env.shape[0] * env.shape[1], 3), z(1), z(16), z(16)
    g.g_env = g_env.data_ptr(); g.g_env_scale = g_scale.data_ptr(); g.g_env_from_world = g_xf.data_ptr(); g.g_camera = g_cam.data_ptr()
    keep += [g_env, g_scale, g_xf, g_cam]
w = torch.ones((n, 3), device="cuda")
L = cabi.lib()
a = cabi.make_args(max_depth=3, seeds=(1, 2, 3), terms=7, guiding=guiding)
cabi.check(L.psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None)); torch.cuda.synchronize()
t = time.perf_counter()
for i in range(calls):
    a = cabi.make_args(max_depth=3, seeds=(i, i, i), terms=7, guiding=guiding)
    cabi.check(L.psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
torch.cuda.synchronize()
print("%s backward (%d x %d, %d samples per pixel and term, depth 3, %d triangles): %.2f ms per call" % (what, res, res, spp, n_tri, (time.perf_counter() - t) / calls * 1e3))
