"""BASELINE config 4 on ONE GPU (the driver shards it over 8): Cornell box 2048x2048, spp=sppe=sppse=64, PathTracer(3)
renderD; checks the 268 M-lane index arithmetic and reports the single-GPU time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__; __graft_entry__.build()
import scenes, product
import psdr_jit_amd as psdr
res, spp = 2048, 64
spec = scenes.cbox_scene(res, res, spp, spp, spp, param="light_x")
sc = product.build_scene(spec)
integ = psdr.PathTracer(3)
img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=1); torch.cuda.synchronize()
t0 = time.time()
img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=2); torch.cuda.synchronize()
dt = time.time() - t0
print("config 4 on one GPU: %.1f ms per renderD, %.1f Msamples/s; finite %s; mean %.4f dmean %.5f" %
      (dt * 1e3, res * res * spp / dt / 1e6, bool(torch.isfinite(img).all() and torch.isfinite(dimg).all()), float(img.mean()), float(dimg.mean())))
# the same frame at 512x512 has the same per-pixel expectation: compare the means
spec2 = scenes.cbox_scene(512, 512, 32, 32, 32, param="light_x")
sc2 = product.build_scene(spec2)
i2, d2 = psdr.render_d_fwd(integ, sc2, 0, seed=2)
print("512x512x32 mean %.4f dmean %.5f" % (float(i2.mean()), float(d2.mean())))
