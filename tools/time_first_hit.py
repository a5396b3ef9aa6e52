"""Forward / backward of the first-hit integrators (FieldExtractionIntegrator, CollocatedIntegrator) on the README box with Microfacet boxes, 512 x 512, 32 spp.   python tools/time_first_hit.py"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import __graft_entry__; __graft_entry__.build()
from psdr_jit_amd import cabi
import product, scenes
spec = scenes.microfacet_cbox_scene(512, 512, 32, 32, 0, param="box_x")
sc = product.build_scene(spec); snap = sc._snapshot(); cam = sc.param_map["Sensor[0]"]
n = 512 * 512
z = lambda *s: torch.zeros(s, device="cuda")
n_tri = np.asarray(snap["d_triangles"]).shape[0]
g_tri, g_b, g_e, g_s, g_p, g_mat = z(n_tri, 22), z(8, 3), z(2, 3), z(max(1, np.asarray(snap["d_sec_edges"]).shape[0]), 6), z(max(1, np.asarray(cam._primary_edges(True)).shape[0]), 4), z(8, 16)
g = cabi.Grads(g_tri.data_ptr(), g_b.data_ptr(), g_e.data_ptr(), g_s.data_ptr(), g_p.data_ptr()); g.g_mat = g_mat.data_ptr()
w = torch.ones((n, 3), device="cuda"); buf = torch.empty((2, n, 3), device="cuda")
for field, name in ((1, "position"), (2, "depth"), (4, "shading normal"), (6, "albedo"), (8, "collocated")):
    a = cabi.make_args(max_depth=0, seeds=(1, 2, 3), terms=3, field=field, intensity=2e5)
    for fn, lbl in ((lambda: cabi.lib().psdr_hip_render_d_fwd(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None), "fwd"), (lambda: cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None), "bwd")):
        cabi.check(fn()); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): cabi.check(fn())
        torch.cuda.synchronize(); print("%-15s %s %.2f ms" % (name, lbl, (time.perf_counter() - t) / 5 * 1e3))
