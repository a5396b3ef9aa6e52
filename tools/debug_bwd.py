import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import psdr_jit_amd as psdr
import test_gpu_api as tg
P = psdr.FloatD(0.).requires_grad_()
sc = tg._readme_scene(psdr, P)
integ = psdr.PathTracer(2)
img = integ.renderD(sc, 0, seed=5)
print('img grad_fn', img.grad_fn, img.requires_grad)
w = torch.ones_like(img)
try:
    (img * w).sum().backward()
except Exception as e:
    import traceback; traceback.print_exc()
print('P.grad', P.grad)
