"""BASELINE config 5 on one GPU: envmap-lit ~82k-triangle mesh, renderD w.r.t. the DiffuseBSDF albedo with
secondary-edge guiding.  Times the product and (optionally, small size) checks it against the oracle.
    python tools/config5.py [--res 512] [--spp 16] [--level 6] [--check]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__; __graft_entry__.build()
import scenes, product
import psdr_jit_amd as psdr

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=512); ap.add_argument("--spp", type=int, default=16); ap.add_argument("--level", type=int, default=6)
ap.add_argument("--depth", type=int, default=3); ap.add_argument("--check", action="store_true"); ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--guiding", type=int, nargs=4, default=[2000, 5, 5, 32])
ap.add_argument("--env-res", type=int, nargs=2, default=[1024, 512])
a = ap.parse_args()
spec = scenes.config5_scene(a.res, a.res, a.spp, a.spp, a.spp, level=a.level, env_res=tuple(a.env_res))
t0 = time.time()
sc = product.build_scene(spec)
t_cfg = time.time() - t0
import ctypes as C
from psdr_jit_amd import cabi
nn, nl, md, lb = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
cabi.check(cabi.lib().psdr_hip_scene_stats(C.c_void_p(sc._hip_handle()), C.byref(nn), C.byref(nl), C.byref(md), C.byref(lb)))
print("triangles", sum(len(m.faces) for m in spec.meshes) + 12, "bvh nodes", nn.value, "depth", md.value, "configure %.2f s" % t_cfg)
integ = psdr.PathTracer(a.depth)
integ.trace_static_edges = True            # every primary-edge sample traced (the parameter is a colour: the public surface would drop them all - bench_scene.py)
t0 = time.time(); integ.preprocess_secondary_edges(sc, 0, a.guiding, 1, 0); torch.cuda.synchronize(); print("guiding build %.3f s" % (time.time() - t0))
for terms, name in ((1, "interior"), (2, "primary"), (4, "secondary"), (7, "all")):
    psdr.render_d_fwd(integ, sc, 0, seed=1, terms=terms); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps): img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=1, terms=terms)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.steps
    print("renderD %-9s %8.2f ms   %.1f Msamples/s" % (name, dt * 1e3, a.res * a.res * a.spp / dt / 1e6))
print("finite", bool(torch.isfinite(img).all() and torch.isfinite(dimg).all()), "mean", float(img.mean()), "d mean", float(dimg.mean()))
if a.check:
    from oracle import oracle as orc
    ref = orc.OracleScene(spec, [0])
    g = ref.guiding_build(0, a.guiding, nrounds=1, seed=0)
    t0 = time.time(); wimg, wd = ref.render_d(max_depth=a.depth, seeds=(1, 1, 1), guiding=g); print("oracle %.1f s" % (time.time() - t0))
    print("rel_l2 image", product.rel_l2(img.cpu().numpy(), wimg), "derivative", product.rel_l2(dimg.cpu().numpy(), wd))
