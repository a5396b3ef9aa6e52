"""One optimisation step as a psdr-jit user writes it (README.md:87-106): move a parameter, configure, renderD, loss, backward.
Wall-clock per stage on the README Cornell box at the BASELINE config-3 settings (512x512, spp = sppe = sppse = 32, depth 3).
    python tools/opt_step_timing.py [scene]        scene: cbox (default) | sphere | envbunny"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__; __graft_entry__.build()
import psdr_jit_amd as psdr
from psdr_jit_amd import Matrix4fC, Matrix4fD
import tutorials as tut

which = sys.argv[1] if len(sys.argv) > 1 else "cbox"
sc = tut._scene(512, 512, 32, 32, 32)
if which == "sphere":
    tut._camera(sc); tut._sphere_box(sc)
elif which == "envbunny":           # the Forward_AD_envmap notebook scene at 512x512 / 32 spp, depth 1
    sensor = psdr.PerspectiveCamera(80, 0.000001, 10000000.)
    sensor.to_world = Matrix4fD([[-1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., -1., 0.], [0., 0., 0., 1.]])
    sc.add_Sensor(sensor)
    sc.add_BSDF(psdr.MicrofacetBSDF([0.2, 0.9, 0.9], [0.01, 0.01, 0.01], 0.3), "bunny")
    sc.add_Mesh(os.path.join(tut.DATA, "mesh", "bunny_low.obj"), Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., -100.], [0., 0., 0., 1.]]), "bunny", None)
    sc.add_EnvironmentMap(os.path.join(tut.DATA, "envmap", "ballroom_1k.exr"), tut.I4, 1.0)
else:
    tut._camera(sc, 208., 273., -800.)
    sc.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light"); sc.add_BSDF(psdr.DiffuseBSDF(), "cat"); sc.add_BSDF(psdr.DiffuseBSDF([0.95, 0.95, 0.95]), "white")
    sc.add_BSDF(psdr.DiffuseBSDF([0.20, 0.90, 0.20]), "green"); sc.add_BSDF(psdr.DiffuseBSDF([0.90, 0.20, 0.20]), "red")
    cb = os.path.join(tut.DATA, "cbox")
    sc.add_Mesh(os.path.join(cb, "cbox_luminaire.obj"), Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., -0.5], [0., 0., 1., 0.], [0., 0., 0., 1.]]), "light", psdr.AreaLight([20.0, 20.0, 8.0]))
    for f, b in (("smallbox", "cat"), ("largebox", "cat"), ("floor", "white"), ("ceiling", "white"), ("back", "white"), ("greenwall", "green"), ("redwall", "red")):
        sc.add_Mesh(os.path.join(cb, "cbox_%s.obj" % f), Matrix4fC(tut.I4), b, None)
sc.configure(); sc.configure([0])
integ = psdr.PathTracer(1 if which == "envbunny" else 3)
P = psdr.FloatD(0.).requires_grad_()
target = None
T = {k: [] for k in ("set_transform + configure", "renderD", "loss", "backward", "total")}
for it in range(6):
    torch.cuda.synchronize(); t0 = time.time()
    sc.param_map["Mesh[0]"].set_transform(Matrix4fD([[1., 0., 0., P * 100], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    sc.configure([0]); t1 = time.time()
    img = integ.renderD(sc, 0, seed=it); torch.cuda.synchronize(); t2 = time.time()
    if target is None: target = img.detach() * 0.9
    loss = ((img - target) ** 2).mean(); torch.cuda.synchronize(); t3 = time.time()
    P.grad = None
    loss.backward(); torch.cuda.synchronize(); t4 = time.time()
    if it > 0:
        for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)): T[k].append(v * 1e3)
for k, v in T.items(): print("%-28s %8.2f ms" % (k, float(np.median(v))))
print("gradient", float(P.grad))
