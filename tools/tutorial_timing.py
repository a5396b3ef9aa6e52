"""The one timing the reference publishes (tutorials/Forward_AD.ipynb:136-176: sphere Cornell box, 512x512, spp = sppe = sppse = 32,
PathTracer(1).renderD in 1.258 s, configure() 0.057 s, on an unnamed NVIDIA GPU), measured here on the same calls through
examples/tutorials.py.  Context only: other hardware, and the notebook's timer includes drjit tracing.
    python tools/tutorial_timing.py [depth]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import __graft_entry__; __graft_entry__.build()
import tutorials as tut
import psdr_jit_amd as psdr

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sc = tut._scene(512, 512, 32, 32, 32)
integrator = psdr.PathTracer(depth)
tut._camera(sc)
tut._sphere_box(sc)
t0 = time.time(); sc.configure(); t1 = time.time(); sc.configure([0]); t2 = time.time()
P = tut._move(sc, ["Mesh[0]", "Mesh[1]"])
t3 = time.time()
print("configure() %.4f s, configure([0]) %.4f s, configure([0]) after set_transform %.4f s   (notebook: 0.0573 / 0.00707 / 0.0132)" % (t1 - t0, t2 - t1, t3 - t2))
img = integrator.renderD(sc, 0); g = psdr.forward_grad(img, P); torch.cuda.synchronize()
ts = []
for i in range(5):
    torch.cuda.synchronize(); t = time.time()
    img = integrator.renderD(sc, 0)
    torch.cuda.synchronize(); ta = time.time()
    g = psdr.forward_grad(img, P)
    torch.cuda.synchronize(); tb = time.time()
    ts.append((ta - t, tb - ta))
a = sorted(x for x, _ in ts)[2]; b = sorted(y for _, y in ts)[2]
n = 512 * 512 * 32
print("PathTracer(%d) sphere scene 512x512 32/32/32: renderD (primal image) %.2f ms, forward_grad (image + derivative, all three terms) %.2f ms"
      % (depth, a * 1e3, b * 1e3))
print("  -> %.1f Msamples/s for image + derivative   (notebook, depth 1: 1258.44 ms = 6.67 Msamples/s)" % (n / b / 1e6))
