"""The 8-wide tree (-DPSDR_BVH_WIDTH=8: bvh.h collapses to eight children per node, trav4.h::t4_node tests eight boxes and sorts eight keys) on ONE build of libpsdr_hip.so,
in a process of its own (tests/test_gpu_codegen_matrix.py::test_eight_wide_tree runs it on a staged copy of the package):

    python tools/w8_check.py [--pkg DIR]

  * the node size the library reports, and psdr_hip_scene_check_tree (every triangle inside every ancestor's quantised box) on the 81 920-triangle mesh of config 5 and on the sphere box,
    before and after a refit;
  * closest hits of 20 000 random rays against the ORACLE's brute force (every triangle, smallest (t, id)): bit-equal triangle ids and distances;
  * a small renderD of config 5 (all three terms) against the oracle, rel L2 < 1e-3.
One JSON line per check; exit code 1 on a failure."""
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

import psdr_jit_amd as psdr
from psdr_jit_amd import build as pbuild, cabi
import product
import scenes
from oracle import oracle as orc


def trace(sc, o, d):
    n = len(o)
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    tri = torch.empty(n, dtype=torch.int32, device="cuda")
    uv = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    t = torch.empty(n, dtype=torch.float32, device="cuda")
    cabi.check(cabi.lib().psdr_hip_trace(sc._hip_handle(), n, to.data_ptr(), td.data_ptr(), tri.data_ptr(), uv.data_ptr(), t.data_ptr(), None))
    return tri.cpu().numpy(), t.cpu().numpy()


def violations(sc):
    v = C.c_int64(-1)
    cabi.check(cabi.lib().psdr_hip_scene_check_tree(C.c_void_p(sc._hip_handle()), C.byref(v)))
    return int(v.value)


def main():
    ok = True
    orc.build()
    print(json.dumps({"library": pbuild.HIP_LIB, "node_bytes": int(cabi.lib().psdr_hip_bvh_node_bytes())}))
    rng = np.random.default_rng(3)
    for name, spec in (("sphere", scenes.sphere_scene(32, 32, 2, 2, 2)), ("config5", scenes.config5_scene(48, 48, 2, 2, 2, level=6, env_res=(64, 32)))):
        sc = product.build_scene(spec)
        ref = orc.OracleScene(spec, [0])
        n = 20000
        o = rng.uniform([-100, 0, -200], [650, 500, 650], size=(n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        tri, t = trace(sc, o, d)
        wtri, _wuv, wt = ref.trace(o, d)
        same = bool(np.array_equal(tri, wtri) and np.array_equal(t[wtri >= 0], wt[wtri >= 0]))
        bad = violations(sc)
        rec = {"case": name + " hits", "hits": int((wtri >= 0).sum()), "bit_equal": same, "tree_violations": bad, "ok": same and bad == 0 and int((wtri >= 0).sum()) > n // 4}
        ok = ok and rec["ok"]
        print(json.dumps(rec))
    spec = scenes.config5_scene(48, 48, 4, 4, 4, level=6, env_res=(64, 32), param="blob_x")
    sc = product.build_scene(spec)
    ref = orc.OracleScene(spec, [0])
    buf = torch.empty((2, 48 * 48, 3), dtype=torch.float32, device="cuda")
    a = cabi.make_args(max_depth=3, seeds=(5, 6, 7), terms=7)
    cabi.check(cabi.lib().psdr_hip_render_d_fwd(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None))
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    wimg, wd = ref.render_d(max_depth=3, seeds=(5, 6, 7))
    e0, e1 = product.rel_l2(got[0], wimg), product.rel_l2(got[1], wd)
    rec = {"case": "config5 renderD", "rel_l2_image": e0, "rel_l2_derivative": e1, "ok": bool(e0 < 1e-3 and e1 < 1e-3 and np.abs(wd).max() > 0)}
    ok = ok and rec["ok"]
    print(json.dumps(rec))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
