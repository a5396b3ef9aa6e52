import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import psdr_jit_amd as psdr
from psdr_jit_amd import cabi
import product, scenes
W = H = 512; spp = 32; D = 3
spec = scenes.cbox_scene(W, H, spp, spp, spp, param="light_x")
sc = product.build_scene(spec)
n = W * H
snap = sc._snapshot(); cam = sc.param_map["Sensor[0]"]
g_tri = torch.zeros((36, 22), device="cuda"); g_b = torch.zeros((5, 3), device="cuda"); g_e = torch.zeros((1, 3), device="cuda")
g_s = torch.zeros((66, 6), device="cuda"); g_p = torch.zeros((np.asarray(cam._primary_edge_ids()).shape[0], 4), device="cuda")
w = torch.ones((n, 3), device="cuda")
g = cabi.Grads(g_tri.data_ptr(), g_b.data_ptr(), g_e.data_ptr(), g_s.data_ptr(), g_p.data_ptr())
for terms in (1, 2, 4, 7):
    a = cabi.make_args(max_depth=D, seeds=(1, 2, 3), terms=terms)
    cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None)); torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(5):
        a = cabi.make_args(max_depth=D, seeds=(i, i, i), terms=terms)
        cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
    torch.cuda.synchronize()
    print('bwd terms', terms, 'ms', (time.perf_counter() - t) / 5 * 1e3)
# with the requires_grad filter: only the luminaire's triangles (the README's translation parameter), no colours
filt = torch.zeros(sc.num_meshes, dtype=torch.uint8, device="cuda"); filt[0] = 1
for label, gg in (("mesh 0 only, no colours", cabi.Grads(g_tri.data_ptr(), g_b.data_ptr(), g_e.data_ptr(), g_s.data_ptr(), g_p.data_ptr(), filt.data_ptr(), 1, 1)),
                  ("colours only", cabi.Grads(g_tri.data_ptr(), g_b.data_ptr(), g_e.data_ptr(), g_s.data_ptr(), g_p.data_ptr(), torch.zeros(sc.num_meshes, dtype=torch.uint8, device="cuda").data_ptr(), 0, 0)),
                  ("small box (mesh 1) only", cabi.Grads(g_tri.data_ptr(), g_b.data_ptr(), g_e.data_ptr(), g_s.data_ptr(), g_p.data_ptr(), torch.tensor([0, 1, 0, 0, 0, 0, 0, 0], dtype=torch.uint8, device="cuda").data_ptr(), 1, 1))):
    for terms in (1, 7):
        a = cabi.make_args(max_depth=D, seeds=(1, 2, 3), terms=terms)
        cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(gg), None)); torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(5):
            a = cabi.make_args(max_depth=D, seeds=(i, i, i), terms=terms)
            cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(gg), None))
        torch.cuda.synchronize()
        print('bwd [%s] terms %d ms %.2f' % (label, terms, (time.perf_counter() - t) / 5 * 1e3))
gg = cabi.Grads(g_tri.data_ptr(), g_b.data_ptr(), g_e.data_ptr(), g_s.data_ptr(), g_p.data_ptr(), torch.zeros(sc.num_meshes, dtype=torch.uint8, device="cuda").data_ptr(), 1, 1)
a = cabi.make_args(max_depth=D, seeds=(1, 2, 3), terms=1)
cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(gg), None)); torch.cuda.synchronize()
t = time.perf_counter()
for i in range(5):
    a = cabi.make_args(max_depth=D, seeds=(i, i, i), terms=1)
    cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(gg), None))
torch.cuda.synchronize()
print('bwd [nothing wanted: search + record only] terms 1 ms %.2f' % ((time.perf_counter() - t) / 5 * 1e3))
