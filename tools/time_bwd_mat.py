"""Forward and backward passes (psdr_hip_render_d_fwd / _bwd, every leaf) of the README box with Microfacet / RoughConductor /
RoughDielectric boxes and of the plain box, 512 x 512, 32 spp, depth 3.   python tools/time_bwd_mat.py"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import psdr_jit_amd as psdr
from psdr_jit_amd import cabi
import product, scenes
for name, spec in (("microfacet", scenes.microfacet_cbox_scene(512, 512, 32, 32, 32, param="roughness")), ("conductor", scenes.conductor_cbox_scene(512, 512, 32, 32, 32, param="alpha")),
                   ("dielectric", scenes.dielectric_cbox_scene(512, 512, 32, 32, 32, param="alpha")), ("cbox", scenes.cbox_scene(512, 512, 32, 32, 32, param="light_x")),
                   # a normal-mapped Microfacet floor (PSDR_ADJ_PROBE=1 times the record-and-probe form the sweep replaced)
                   ("normalmap", scenes.normalmap_scene(512, 512, 32, 32, 32, param="box_x", nested="microfacet", nmap="bumpy"))):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    sc = product.build_scene(spec); snap = sc._snapshot(); cam = sc.param_map["Sensor[0]"]
    n = 512 * 512
    z = lambda *s: torch.zeros(s, device="cuda")
    n_tri = np.asarray(snap["d_triangles"]).shape[0]
    g_tri, g_b, g_e, g_s, g_p, g_mat = z(n_tri, 22), z(8, 3), z(2, 3), z(max(1, np.asarray(snap["d_sec_edges"]).shape[0]), 6), z(max(1, np.asarray(cam._primary_edges(True)).shape[0]), 4), z(8, 16)
    w = torch.ones((n, 3), device="cuda")
    g = cabi.Grads(g_tri.data_ptr(), g_b.data_ptr(), g_e.data_ptr(), g_s.data_ptr(), g_p.data_ptr()); g.g_mat = g_mat.data_ptr()
    offs = (C.c_int64 * (3 * 8))(); total = C.c_int64(0)
    cabi.check(cabi.lib().psdr_hip_scene_tex_layout(C.c_void_p(sc._hip_handle()), offs, C.byref(total)))
    g_tex = z(max(1, total.value))
    if total.value > 0: g.g_tex = g_tex.data_ptr()
    buf = torch.empty((2, n, 3), device="cuda")
    for terms in (1, 7):
        a = cabi.make_args(max_depth=3, seeds=(1, 2, 3), terms=terms)
        for fn, lbl in ((lambda: cabi.lib().psdr_hip_render_d_fwd(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None), "fwd"), (lambda: cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None), "bwd")):
            cabi.check(fn()); torch.cuda.synchronize(); t = time.perf_counter()
            for i in range(3): cabi.check(fn())
            torch.cuda.synchronize(); print(name, "terms", terms, lbl, "%.2f ms" % ((time.perf_counter() - t) / 3 * 1e3), flush=True)
