"""Backward pass (psdr_hip_render_d_bwd, all leaves) of environment-lit scenes: BASELINE config 5 at 512^2 x 16 and the small
envmap box.   python tools/time_bwd_env.py"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import __graft_entry__; __graft_entry__.build()
import psdr_jit_amd as psdr
from psdr_jit_amd import cabi
import product, scenes

def run(name, spec, depth):
    sc = product.build_scene(spec)
    snap = sc._snapshot(); cam = sc.param_map["Sensor[0]"]
    n = spec.width * spec.height
    n_tri = np.asarray(snap["d_triangles"]).shape[0]
    n_sec = max(1, np.asarray(snap["d_sec_edges"]).shape[0]); n_prim = max(1, np.asarray(cam._primary_edges(True)).shape[0])
    env = [e for e in spec.emitters if getattr(e, "env_data", None) is not None][0].env_data
    z = lambda *s: torch.zeros(s, device="cuda")
    g_tri, g_b, g_e, g_s, g_p = z(n_tri, 22), z(max(1, len(spec.bsdfs)), 3), z(max(1, len(spec.emitters)), 3), z(n_sec, 6), z(n_prim, 4)
    g_env, g_scale, g_xf, g_cam = z(env.shape[0] * env.shape[1], 3), z(1), z(16), z(16)
    w = torch.ones((n, 3), device="cuda")
    g = cabi.Grads(g_tri.data_ptr(), g_b.data_ptr(), g_e.data_ptr(), g_s.data_ptr(), g_p.data_ptr())
    g.g_env = g_env.data_ptr(); g.g_env_scale = g_scale.data_ptr(); g.g_env_from_world = g_xf.data_ptr(); g.g_camera = g_cam.data_ptr()
    buf = torch.empty((2, n, 3), device="cuda")
    for terms in (1, 7):
        a = cabi.make_args(max_depth=depth, seeds=(1, 2, 3), terms=terms)
        cabi.check(cabi.lib().psdr_hip_render_d_fwd(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None))
        cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None)); torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(3):
            cabi.check(cabi.lib().psdr_hip_render_d_fwd(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None))
        torch.cuda.synchronize(); tf = (time.perf_counter() - t) / 3
        t = time.perf_counter()
        for i in range(3):
            cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
        torch.cuda.synchronize(); tb = (time.perf_counter() - t) / 3
        print("%-12s terms %d: forward %.2f ms, backward %.2f ms" % (name, terms, tf * 1e3, tb * 1e3), flush=True)

run("envbox", scenes.envmap_scene(256, 256, 16, 16, 16, param="box_x"), 3)
run("envballs", scenes.envmap_scene(256, 256, 16, 16, 16, param="albedo", balls=True), 3)
run("config5", scenes.config5_scene(512, 512, 16, 16, 16, level=6, env_res=(1024, 512)), 3)
