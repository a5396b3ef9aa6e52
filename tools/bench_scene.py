"""Per-term renderD timings of a named test scene (tests/scenes.py) at the C3 settings: 512x512, spp = sppe = sppse = 32, depth 3.
    python tools/bench_scene.py sphere|cbox|envballs|microfacet|conductor|bunny [--skip] [--terms-only]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__; __graft_entry__.build()
import scenes, product
import psdr_jit_amd as psdr
name = sys.argv[1] if len(sys.argv) > 1 else "sphere"
res, spp = 512, 32
spec = {"sphere": lambda: scenes.sphere_scene(res, res, spp, spp, spp),
        "cbox": lambda: scenes.cbox_scene(res, res, spp, spp, spp),
        "envballs": lambda: scenes.envmap_scene(res, res, spp, spp, spp, param="albedo", balls=True),
        "microfacet": lambda: scenes.microfacet_cbox_scene(res, res, spp, spp, spp),
        "conductor": lambda: scenes.conductor_cbox_scene(res, res, spp, spp, spp),
        # scene class 0 (global memory, materials + environment map): the envmap notebook's glossy bunny under the full-size ballroom map
        "bunny": lambda: scenes.envmap_tutorial_scene(res, res, spp, spp, spp, param="bunny_x", env_stride=1)}[name]()
sc = product.build_scene(spec)
integ = psdr.PathTracer(3)
# every primary-edge sample traced, as the reference does and as rounds 1-4 measured (Integrator.trace_static_edges; "--skip": the public surface's default, which
# drops the samples on edges that do not move under the scene's tangent - psdr_render_args.skip_static_edges)
integ.trace_static_edges = "--skip" not in sys.argv
# "--terms-only": the three terms alone, no combined call (the profiler passes: one kind of launch per kernel)
for terms, label in ((1, "interior"), (2, "primary"), (4, "secondary")) + ((() if "--terms-only" in sys.argv else ((7, "all"),))):
    psdr.render_d_fwd(integ, sc, 0, seed=1, terms=terms); torch.cuda.synchronize()
    t0 = time.time()
    for i in range(5): psdr.render_d_fwd(integ, sc, 0, seed=i, terms=terms)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    print("%-8s renderD %-9s %8.2f ms  %8.1f Msamples/s" % (name, label, dt * 1e3, res * res * spp / dt / 1e6))
