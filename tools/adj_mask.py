"""Diagnostic: on a pixel list where the reverse sweep and the probe form disagree, which list POSITIONS carry the difference?  The weight image masks
all other positions (their lanes still run pass 1 of the sweep: the wave composition stays).   python tools/adj_mask.py depth first_pixel count"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import __graft_entry__
__graft_entry__.build()
from psdr_jit_amd import cabi
import product, scenes

depth, first, count = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
res, spp = 40, 8
spec = scenes.envmap_scene(res, res, spp, 0, 0, param="box_x", area_light=True, balls=False)
sc = product.build_scene(spec)
snap = sc._snapshot()
nt, nb, ne = np.asarray(snap["d_triangles"]).shape[0], len(spec.bsdfs), len(spec.emitters)
gen = torch.Generator(device="cpu").manual_seed(3)
w_full = (torch.rand((res * res, 3), generator=gen) + 0.5)


def bwd(pix, probe=False, keep=None, d=depth):
    if probe:
        os.environ["PSDR_ADJ_PROBE"] = "1"
    else:
        os.environ.pop("PSDR_ADJ_PROBE", None)
    g_tri = torch.zeros((nt, 22), dtype=torch.float32, device="cuda")
    g_bsdf = torch.zeros((nb, 3), dtype=torch.float32, device="cuda")
    g_em = torch.zeros((ne, 3), dtype=torch.float32, device="cuda")
    g_sec = torch.zeros((1, 6), dtype=torch.float32, device="cuda")
    g_prim = torch.zeros((1, 4), dtype=torch.float32, device="cuda")
    g = cabi.Grads(g_tri.data_ptr(), g_bsdf.data_ptr(), g_em.data_ptr(), g_sec.data_ptr(), g_prim.data_ptr())
    ids = torch.tensor(pix, dtype=torch.int32, device="cuda")
    w = w_full[torch.tensor(pix, dtype=torch.long)].contiguous()
    if keep is not None:
        m = torch.zeros(len(pix), 1); m[keep] = 1.0
        w = w * m
    w = w.to("cuda")
    a = cabi.make_args(max_depth=d, seeds=(7, 8, 9), terms=1, pix_ids_ptr=ids.data_ptr(), n_pix=len(pix))
    cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
    torch.cuda.synchronize()
    return np.concatenate([g_bsdf.cpu().numpy().astype(np.float64).ravel(), g_em.cpu().numpy().astype(np.float64).ravel()])


np.set_printoptions(precision=6, linewidth=220, suppress=True)
S = list(range(first, first + count))
a, b = bwd(S), bwd(S, True)
print("list", S, "\n sweep", a, "\n probe", b)
for j in range(len(S)):
    a, b = bwd(S, False, [j]), bwd(S, True, [j])
    flag = np.abs(a - b).max() > 1e-4 * (np.abs(b).max() + 1e-9)
    print("position", j, "pixel", S[j], "DIFFERS" if flag else "ok", "" if not flag else "\n   sweep %s\n   probe %s" % (a, b))
    if flag:
        # is it this position's own paths, whatever runs beside them?  the same position with every other pixel replaced by a copy of pixel S[j]
        T = [S[j]] * len(S)
        a2, b2 = bwd(T, False, [j]), bwd(T, True, [j])
        print("   among copies of itself:", "DIFFERS" if np.abs(a2 - b2).max() > 1e-4 * (np.abs(b2).max() + 1e-9) else "ok", a2[6:9], b2[6:9])
        # the list cut after this position, and with the positions before it replaced by a background pixel
        T = S[:j + 1]
        a3, b3 = bwd(T, False, [j]), bwd(T, True, [j])
        print("   list cut after it:", "DIFFERS" if np.abs(a3 - b3).max() > 1e-4 * (np.abs(b3).max() + 1e-9) else "ok")
        for dd in (1, 2, 3):
            a4, b4 = bwd(S, False, [j], dd), bwd(S, True, [j], dd)
            print("   depth", dd, "sweep", a4[:9], "probe", b4[:9])
