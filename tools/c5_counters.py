"""nodes / triangles per ray of BASELINE config 5 (82 k triangles) per term: python tools/c5_counters.py [res] [spp]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__; __graft_entry__.build()
import scenes, product
from psdr_jit_amd import cabi
res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
spec = scenes.config5_scene(res, res, spp, spp, spp, level=6, env_res=(1024, 512))
sc = product.build_scene(spec)
buf = torch.empty((2, res * res, 3), dtype=torch.float32, device="cuda")
for terms in (1, 2, 4):
    c = cabi.Counters()
    a = cabi.make_args(max_depth=3, seeds=(0, 0, 0), terms=terms)
    cabi.check(cabi.lib().psdr_hip_render_d_fwd_counted(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), C.byref(c), None))
    print("terms", terms, "rays", c.rays, "nodes/ray %.1f tris/ray %.1f hits/ray %.2f" % (c.nodes_visited / c.rays, c.tris_tested / c.rays, c.shaded_hits / c.rays))
nn, nl, md, lb = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
cabi.check(cabi.lib().psdr_hip_scene_stats(C.c_void_p(sc._hip_handle()), C.byref(nn), C.byref(nl), C.byref(md), C.byref(lb)))
print("nodes", nn.value, "leaves", nl.value, "depth", md.value, "lds", lb.value)
