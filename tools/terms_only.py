"""Forward mode of ONE term alone on the C3 workload (README box, 512^2, 32 / 32 / 32, depth 3), for the profiler and for variant libraries:

    python tools/terms_only.py [--pkg DIR] [terms=1] [calls=10]        # terms: 1 interior (k_paths<true,...,0>), 2 primary edges, 4 secondary edges

--pkg DIR = a staged package copy (tools/variants.py stage NAME): the variant library runs, nothing is rebuilt."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:]]
if "--pkg" in args:
    i = args.index("--pkg"); sys.path.insert(0, os.path.abspath(args[i + 1])); del args[i:i + 2]
else:
    sys.path.insert(0, ROOT)
    import __graft_entry__; __graft_entry__.build()
sys.path += [os.path.join(ROOT, "tests"), ROOT]
import torch
from psdr_jit_amd import cabi
import product, scenes
terms = int(args[0]) if args else 1
calls = int(args[1]) if len(args) > 1 else 10
res, spp = 512, 32
sc = product.build_scene(scenes.cbox_scene(res, res, spp, spp, spp, param="light_x"))
buf = torch.empty((2, res * res, 3), dtype=torch.float32, device="cuda")
L = cabi.lib()
def go(seed):
    a = cabi.make_args(max_depth=3, seeds=(seed, seed, seed), terms=terms)
    cabi.check(L.psdr_hip_render_d_fwd(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None))
go(99); torch.cuda.synchronize()
t = time.perf_counter()
for i in range(calls):
    go(i)
torch.cuda.synchronize()
print("terms %d: %.3f ms per call" % (terms, (time.perf_counter() - t) / calls * 1e3))
