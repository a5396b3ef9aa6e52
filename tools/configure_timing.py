"""What one optimisation step costs through the public Python surface - the calls a psdr-jit user makes (reference README.md:87-106:
set_transform -> configure -> renderD -> forward_to / backward; the reference re-uploads every array and rebuilds its OptiX GAS in every
configure, src/scene/scene.cpp:311-599, src/scene/scene_optix.cpp:265-332).

    python tools/configure_timing.py [c3] [c5] [--res N --spp K] [--reps R]

Per scene, wall time (device synchronised before and after), median of `reps` after one warm-up:
    configure_ms   unchanged   no parameter touched since the last configure()
                   parameter   a DiffuseBSDF reflectance changed
                   vertices    Mesh[0] translated: every triangle of the mesh moves, topology unchanged (`vertices_update` = what the device library did)
                   rebuild     the same change with the device scene destroyed and created anew, as rounds 1-4 did in every configure()
    step_ms        reverse_parameter / reverse_vertices   configure + renderD + loss + loss.backward() w.r.t. the albedo / the translation
                   forward_vertices                       configure + renderD + forward_grad(img, P)
    api_call_ms    renderD(sc, 0) + forward_grad(img, P) on a configured scene ("wall seconds of one renderD call")
bench.py imports measure() for its `api` object.
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


def _sync():
    import torch
    torch.cuda.synchronize()


def timed(fn, reps, prep=None):
    ts = []
    for i in range(reps + 1):
        if prep is not None:
            prep(i)
        _sync()
        t = time.perf_counter()
        fn()
        _sync()
        ts.append((time.perf_counter() - t) * 1e3)
    return round(statistics.median(ts[1:]), 3)


def measure(psdr, sc, albedo_key, reps=7, depth=3, heavy_reps=None):
    """sc: a configured scene whose Mesh[0] has an identity to_world_left; albedo_key: param_map key of a DiffuseBSDF"""
    import torch
    from psdr_jit_amd import FloatD, Matrix4fD
    heavy = heavy_reps or max(2, reps // 2)
    integ = psdr.PathTracer(depth)
    bs = sc.param_map[albedo_key]
    mesh = sc.param_map["Mesh[0]"]
    refl0 = torch.as_tensor(bs.reflectance).detach().clone().reshape(-1)

    def info():
        u = sc._last_update()
        return {"tree": u["tree"], "bytes_uploaded": int(u["bytes_uploaded"]), "host_ms": round(u["ms_host"], 3), "device_ms": round(u["ms_total"], 3),
                "tree_ms": round(u["ms_tree"], 3), "sah_cost": round(u["sah_cost"], 3), "sah_cost_built": round(u["sah_cost_built"], 3)}

    def translate(x):
        return Matrix4fD([[1., 0., 0., x], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]])

    out = {"configure_ms": {}, "step_ms": {}}
    cfg = out["configure_ms"]
    sc.configure([0])
    cfg["unchanged"] = timed(lambda: sc.configure([0]), reps)

    def set_albedo(i):
        bs.reflectance = torch.tensor([0.5 + 0.01 * (i % 7), 0.5, 0.5])
    cfg["parameter"] = timed(lambda: sc.configure([0]), reps, set_albedo)
    cfg["parameter_update"] = info()

    def set_x(i):
        mesh.set_transform(translate(0.25 * (1 + i % 5)))
    cfg["vertices"] = timed(lambda: sc.configure([0]), reps, set_x)
    cfg["vertices_update"] = info()
    sc._always_rebuild = True
    cfg["rebuild"] = timed(lambda: sc.configure([0]), max(2, reps // 2), set_x)
    cfg["rebuild_update"] = info()
    sc._always_rebuild = False
    mesh.set_transform(translate(0.))
    sc.configure([0])

    # whole steps through autograd
    A = refl0.clone().requires_grad_()

    def step_albedo():
        bs.reflectance = A
        sc.configure([0])
        img = integ.renderD(sc, 0)
        loss = (img ** 2).mean()
        A.grad = None
        loss.backward()
    out["step_ms"]["reverse_parameter"] = timed(step_albedo, heavy)
    bs.reflectance = refl0.clone()
    P = FloatD(0.).requires_grad_()

    def step_x():
        mesh.set_transform(translate(P * 100))
        sc.configure([0])
        img = integ.renderD(sc, 0)
        loss = (img ** 2).mean()
        P.grad = None
        loss.backward()
    out["step_ms"]["reverse_vertices"] = timed(step_x, heavy)

    def fwd():
        mesh.set_transform(translate(P * 100))
        sc.configure([0])
        img = integ.renderD(sc, 0)
        psdr.forward_grad(img, P)
    out["step_ms"]["forward_vertices"] = timed(fwd, heavy)

    def api_call():
        img = integ.renderD(sc, 0)
        psdr.forward_grad(img, P)
    api_call()
    out["api_call_ms"] = timed(api_call, max(heavy, 20 if reps >= 7 else heavy))
    out["renderC_ms"] = timed(lambda: integ.renderC(sc, 0), heavy)
    mesh.set_transform(translate(0.))
    sc.configure([0])
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
    import bench
    import synth
    import torch
    for which in args.scenes:
        res = args.res or (1024 if which == "c5" else 512)
        spp = args.spp or (64 if which == "c5" else 32)
        t = time.perf_counter()
        if which == "c5":
            sc, _ = synth.config5_scene(psdr, res, spp)
            key = "BSDF[0]"
        else:
            sc, _ = bench.readme_scene(psdr, res, spp)
            sc.param_map["Mesh[0]"].set_transform(psdr.Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
            key = "BSDF[1]"
        torch.cuda.synchronize()
        first = (time.perf_counter() - t) * 1e3
        out = {"scene": which, "res": res, "spp": spp, "build_scene_ms": round(first, 2)}
        out.update(measure(psdr, sc, key, args.reps, heavy_reps=(3 if which == "c5" else None)))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
