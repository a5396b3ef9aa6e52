"""What Scene.configure() costs, per kind of change, through the public Python surface - the call a psdr-jit user makes once per
optimisation step (reference README.md:87-106: set_transform -> configure -> renderD -> backward; the reference rebuilds its OptiX
GAS in every configure, src/scene/scene_optix.cpp:265-332, called from src/scene/scene.cpp:575).

    python tools/configure_timing.py [c3] [c5] [--res N --spp K] [--reps R] [--json]

Per scene: wall time (synchronised) of configure() after
    nothing        no parameter touched since the last configure
    albedo         a DiffuseBSDF reflectance changed
    tangent        only a forward tangent changed (what forward_grad installs and removes)
    vertices       Mesh[0] translated (every triangle of the mesh moves, topology unchanged)
    first          the first configure of a fresh scene (tree build included)
and of one whole optimisation step w.r.t. the albedo / the translation: configure + renderD + loss + backward.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "examples")):
    if p not in sys.path:
        sys.path.insert(0, p)


def sync():
    import torch
    torch.cuda.synchronize()


def timed(fn, reps, prep=None):
    ts = []
    for i in range(reps + 1):
        if prep is not None:
            prep(i)
        sync()
        t = time.perf_counter()
        fn()
        sync()
        ts.append((time.perf_counter() - t) * 1e3)
    return round(statistics.median(ts[1:]), 3)


def measure(psdr, which, res, spp, reps, depth=3):
    import numpy as np
    import torch
    import bench
    import synth
    from psdr_jit_amd import FloatD, Matrix4fD
    t = time.perf_counter()
    if which == "c5":
        sc, _ = synth.config5_scene(psdr, res, spp)
        albedo_key = "BSDF[0]"
    else:
        sc, _ = bench.readme_scene(psdr, res, spp)
        albedo_key = "BSDF[1]"
    sync()
    first = (time.perf_counter() - t) * 1e3          # load + two configure() calls of the scene function (the second with the primary edges)
    out = {"scene": which, "res": res, "spp": spp, "build_scene_ms": round(first, 2)}
    integ = psdr.PathTracer(depth)
    bs = sc.param_map[albedo_key]
    mesh = sc.param_map["Mesh[0]"]
    out["nothing"] = timed(lambda: sc.configure([0]), reps)

    def set_albedo(i):
        bs.reflectance = torch.tensor([0.5 + 0.01 * (i % 7), 0.5, 0.5])
    out["albedo"] = timed(lambda: sc.configure([0]), reps, set_albedo)
    out["albedo_info"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in sc._last_update().items()}

    def set_x(i):
        mesh.set_transform(Matrix4fD([[1., 0., 0., 0.25 * (i % 5)], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    out["vertices"] = timed(lambda: sc.configure([0]), reps, set_x)
    out["vertices_info"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in sc._last_update().items()}

    # whole steps through autograd
    A = torch.tensor([0.5, 0.5, 0.5], requires_grad=True)
    bs.reflectance = A
    sc.configure([0])

    def step_albedo():
        bs.reflectance = A
        sc.configure([0])
        img = integ.renderD(sc, 0)
        loss = (img ** 2).mean()
        A.grad = None
        loss.backward()
    out["step_albedo"] = timed(step_albedo, max(2, reps // 2))
    bs.reflectance = torch.tensor([0.5, 0.5, 0.5])
    P = FloatD(0.).requires_grad_()

    def step_x():
        mesh.set_transform(Matrix4fD([[1., 0., 0., P * 100], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
        sc.configure([0])
        img = integ.renderD(sc, 0)
        loss = (img ** 2).mean()
        P.grad = None
        loss.backward()
    out["step_vertices"] = timed(step_x, max(2, reps // 2))

    def fwd():
        mesh.set_transform(Matrix4fD([[1., 0., 0., P * 100], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
        sc.configure([0])
        img = integ.renderD(sc, 0)
        psdr.forward_grad(img, P)
    out["step_forward_grad"] = timed(fwd, max(2, reps // 2))
    # the render alone (same public calls, nothing to configure): what the steps above add to it
    mesh.set_transform(Matrix4fD([[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    sc.configure([0])
    out["renderC"] = timed(lambda: integ.renderC(sc, 0), max(2, reps // 2))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scenes", nargs="*", default=["c3", "c5"])
    ap.add_argument("--res", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--reps", type=int, default=7)
    args = ap.parse_args()
    import __graft_entry__
    __graft_entry__.build()
    import psdr_jit_amd as psdr
    for which in args.scenes:
        res = args.res or (1024 if which == "c5" else 512)
        spp = args.spp or (64 if which == "c5" else 32)
        print(json.dumps(measure(psdr, which, res, spp, args.reps)), flush=True)


if __name__ == "__main__":
    main()
