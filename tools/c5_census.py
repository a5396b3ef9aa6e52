"""raw counters of a counted launch on BASELINE config 5, per term (instrumented variant libraries, -DPSDR_DIAG=n: what the four counters mean is trav4.h's / paths.h's
comment for that n):    python tools/c5_census.py [res=512] [spp=16] [terms=2]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__; __graft_entry__.build()
import scenes, product
from psdr_jit_amd import cabi
res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
terms_list = [int(sys.argv[3])] if len(sys.argv) > 3 else [1, 2, 4]
spec = scenes.config5_scene(res, res, spp, spp, spp, level=6, env_res=(1024, 512))
sc = product.build_scene(spec)
buf = torch.empty((2, res * res, 3), dtype=torch.float32, device="cuda")
for terms in terms_list:
    c = cabi.Counters()
    a = cabi.make_args(max_depth=3, seeds=(0, 0, 0), terms=terms)
    cabi.check(cabi.lib().psdr_hip_render_d_fwd_counted(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), C.byref(c), None))
    tot = c.rays + c.nodes_visited + c.tris_tested + c.shaded_hits
    print("terms %d  c_rays %d  c_nodes %d  c_tris %d  c_hits %d   (shares of their sum: %.3f %.3f %.3f %.3f)" % (terms, c.rays, c.nodes_visited, c.tris_tested, c.shaded_hits,
          c.rays / tot, c.nodes_visited / tot, c.tris_tested / tot, c.shaded_hits / tot))
