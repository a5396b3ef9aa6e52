"""Checks the interior reverse sweep of environment-lit scenes (scene class 2, adjoint.h::run_interior_adjoint_sweep<2>) of ONE build of
libpsdr_hip.so, in a process of its own: the codegen-matrix test (tests/test_gpu_codegen_matrix.py) runs it once per variant library.

    python tools/sweep_check.py [--pkg DIR]       # DIR: a staged package copy (tools/variants.py stage NAME); default: the in-tree package

For the three scenes of tests/test_gpu_adjoint.py::test_interior_sweep_environment_map it prints one JSON line each with
  * the dot-product pair  <w, J v> (HIP forward mode)  vs  <J^T w, v> (the sweep),
  * the sweep's adjoint buffers against the record-and-probe form of the same library on the same samples (PSDR_ADJ_PROBE, read per call): colours,
    emitter radiance, the triangle rows of every mesh but the environment map's bounding box, the map's texels / scale / rotation, the camera pose -
    two derivations of the same quantity that share no arithmetic beyond the path itself.
Exit code 1 when a pair differs by more than 3e-4 of its scale."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--pkg" in sys.argv:
    sys.path.insert(0, os.path.abspath(sys.argv[sys.argv.index("--pkg") + 1]))
sys.path += [os.path.join(ROOT, "tests"), ROOT]

import numpy as np
import torch

import psdr_jit_amd
from psdr_jit_amd import build as pbuild, cabi

TOL = 3e-4


def both_forms(sc, spec, depth, n_tris, cube_rows):
    """adjoint buffers of the sweep and of the probe form for one weight image; every optional buffer is requested"""
    n = spec.width * spec.height
    gen = torch.Generator(device="cpu").manual_seed(11)
    w = (torch.rand((n, 3), generator=gen) + 0.5).to("cuda")
    H, W = spec.emitters[0].env_data.shape[:2]
    out = []
    for probe in (False, True):
        if probe:
            os.environ["PSDR_ADJ_PROBE"] = "1"
        else:
            os.environ.pop("PSDR_ADJ_PROBE", None)
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device="cuda")
        g_tri, g_bsdf, g_em, g_sec, g_prim = z(n_tris, 22), z(len(spec.bsdfs), 3), z(len(spec.emitters), 3), z(1, 6), z(1, 4)
        g_cam, g_env, g_scale, g_xf = z(16), z(H * W * 3), z(1), z(16)
        g = cabi.Grads(g_tri.data_ptr(), g_bsdf.data_ptr(), g_em.data_ptr(), g_sec.data_ptr(), g_prim.data_ptr())
        g.g_camera, g.g_env, g.g_env_scale, g.g_env_from_world = g_cam.data_ptr(), g_env.data_ptr(), g_scale.data_ptr(), g_xf.data_ptr()
        a = cabi.make_args(max_depth=depth, seeds=(7, 8, 9), terms=1)
        cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
        torch.cuda.synchronize()
        tri = g_tri.cpu().numpy().astype(np.float64)
        tri[cube_rows] = 0.0          # the sweep also fills the rows of the map's bounding box (fixed geometry: nothing reads them), the probe form leaves them out
        out.append({"tri": tri, "bsdf": g_bsdf.cpu().numpy().astype(np.float64), "emitter": g_em.cpu().numpy().astype(np.float64),
                    "camera": g_cam.cpu().numpy().astype(np.float64)[:12], "env": g_env.cpu().numpy().astype(np.float64),
                    "env_scale": g_scale.cpu().numpy().astype(np.float64), "env_xf": g_xf.cpu().numpy().astype(np.float64)[:11]})
    os.environ.pop("PSDR_ADJ_PROBE", None)
    return out


def main():
    import product
    import scenes
    import test_gpu_adjoint as t
    env = (torch, psdr_jit_amd, cabi)
    got, want = int(cabi.lib().psdr_hip_abi_version()), pbuild.header_abi_version() if os.path.exists(os.path.join(pbuild.ROOT, "include", "psdr_hip.h")) else None
    ok = True
    print(json.dumps({"library": pbuild.HIP_LIB, "abi": got}))
    for param, balls in (("box_x", False), ("albedo", False), ("albedo", True)):
        spec = scenes.envmap_scene(40, 40, 8, 0, 0, param=param, area_light=True, balls=balls)
        lhs, rhs, scale = t._dot_product_case(env, spec, depth=3, terms=1)
        rec = {"case": "%s%s" % (param, "+balls" if balls else ""), "lhs": lhs, "rhs": float(rhs), "scale": scale, "dot_err": abs(lhs - float(rhs)) / scale}
        sc = product.build_scene(spec)
        snap = sc._snapshot()
        d_tri = np.asarray(snap["d_triangles"])
        # rows of the environment map's bounding box: the last 12 triangles the host appends (Scene::configure, scene.cpp:330-380)
        cube_rows = np.arange(d_tri.shape[0] - 12, d_tri.shape[0])
        sw, pr = both_forms(sc, spec, 3, d_tri.shape[0], cube_rows)
        worst = 0.0
        for k in sw:
            err = float(np.abs(sw[k] - pr[k]).sum() / (np.abs(pr[k]).sum() + 1e-12))
            rec["sweep_vs_probe_" + k] = err
            if np.abs(pr[k]).sum() > 0:
                worst = max(worst, err)
        rec["ok"] = bool(rec["dot_err"] <= TOL and worst <= TOL)
        ok = ok and rec["ok"]
        print(json.dumps(rec))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
