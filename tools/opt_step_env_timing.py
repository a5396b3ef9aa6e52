"""One optimisation step on the Forward_AD_envmap notebook scene (bunny_low.obj + MicrofacetBSDF under ballroom_1k.exr): renderD,
loss, loss.backward() into the mesh translation, the BSDF's parameters and the environment radiance.
    python tools/opt_step_env_timing.py            (PSDR_ADJ_PROBE=1 in the environment: round 1's record-and-probe reverse mode)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import numpy as np, torch
import __graft_entry__; __graft_entry__.build()
import psdr_jit_amd as psdr
import tutorials as tut
from psdr_jit_amd import exr

res, spp, depth = 256, 32, 2
sc = tut._scene(res, res, spp, 0, 0)
sensor = psdr.PerspectiveCamera(80, 0.000001, 10000000.)
sensor.to_world = psdr.Matrix4fD([[-1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., -1., 0.], [0., 0., 0., 1.]])
sc.add_Sensor(sensor)
diffuse = torch.tensor([0.2, 0.9, 0.9], requires_grad=True)
rough = psdr.FloatD(0.3).requires_grad_()
sc.add_BSDF(psdr.MicrofacetBSDF([0.04, 0.04, 0.04], diffuse, rough), "bunny")
sc.add_Mesh(os.path.join(tut.DATA, "mesh", "bunny_low.obj"), psdr.Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., -100.], [0., 0., 0., 1.]]), "bunny", None)
rad = torch.tensor(np.ascontiguousarray(exr.read_rgb(os.path.join(tut.DATA, "envmap", "ballroom_1k.exr"))), requires_grad=True)
sc.add_EnvironmentMap(psdr.EnvironmentMap(rad))
sc.configure()
P = psdr.FloatD(0.).requires_grad_()
integ = psdr.PathTracer(depth)
def step(seed):
    for t in (diffuse, rough, rad, P):
        t.grad = None
    sc.param_map["Mesh[0]"].set_transform(psdr.Matrix4fD([[1., 0., 0., P * 10.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    sc.configure([0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img = integ.renderD(sc, 0, seed=seed)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss = (img ** 2).mean()
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return t1 - t0, t2 - t1
step(0)
ts = [step(i + 1) for i in range(3)]
print("envmap tutorial scene %dx%d, %d spp, PathTracer(%d), reverse mode %s: renderD %.1f ms, loss + backward %.1f ms" %
      (res, res, spp, depth, "record-and-probe" if os.environ.get("PSDR_ADJ_PROBE") else "sweep", 1e3 * np.median([a for a, b in ts]), 1e3 * np.median([b for a, b in ts])))
print("grads: P %.4g rough %.4g diffuse %s |radiance| %.4g" % (float(P.grad), float(rough.grad), diffuse.grad.tolist(), float(rad.grad.abs().sum())))
