"""Diagnostic (builds with -DPSDR_SWEEP_DUMP): per-lane record of the reverse sweep for one list position.   python tools/adj_dump.py depth pixel copies position"""
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

depth, pixel, copies, pos = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
res, spp = 40, 8
spec = scenes.envmap_scene(res, res, spp, 0, 0, param="box_x", area_light=True, balls=False)
sc = product.build_scene(spec)
snap = sc._snapshot()
nt, nb, ne = np.asarray(snap["d_triangles"]).shape[0], len(spec.bsdfs), len(spec.emitters)
gen = torch.Generator(device="cpu").manual_seed(3)
w_full = (torch.rand((res * res, 3), generator=gen) + 0.5)
pix = [pixel] * copies


def bwd(probe):
    if probe:
        os.environ["PSDR_ADJ_PROBE"] = "1"
    else:
        os.environ.pop("PSDR_ADJ_PROBE", None)
    g_tri = torch.zeros((nt, 22), dtype=torch.float32, device="cuda")
    g_bsdf = torch.zeros((nb, 3), dtype=torch.float32, device="cuda")
    g_em = torch.zeros((ne, 3), dtype=torch.float32, device="cuda")
    g_sec = torch.zeros((1, 6), dtype=torch.float32, device="cuda")
    g_prim = torch.zeros((1, 4), dtype=torch.float32, device="cuda")
    dbg = torch.full((len(pix) * spp, 64), -777.0, dtype=torch.float32, device="cuda")
    g = cabi.Grads(g_tri.data_ptr(), g_bsdf.data_ptr(), g_em.data_ptr(), g_sec.data_ptr(), g_prim.data_ptr())
    g.g_tex = dbg.data_ptr()
    ids = torch.tensor(pix, dtype=torch.int32, device="cuda")
    w = w_full[torch.tensor(pix, dtype=torch.long)].contiguous()
    m = torch.zeros(len(pix), 1); m[pos] = 1.0
    w = (w * m).to("cuda")
    a = cabi.make_args(max_depth=depth, seeds=(7, 8, 9), terms=1, pix_ids_ptr=ids.data_ptr(), n_pix=len(pix))
    cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
    torch.cuda.synchronize()
    return g_bsdf.cpu().numpy().astype(np.float64).ravel(), dbg.cpu().numpy()


np.set_printoptions(precision=6, linewidth=250, suppress=False)
gs, d = bwd(False)
gp, _ = bwd(True)
print("sweep g_bsdf", gs, "\nprobe g_bsdf", gp, "\nDIFFERS" if np.abs(gs - gp).max() > 1e-4 * np.abs(gp).max() else "\nagree")
for l in range(pos * spp, (pos + 1) * spp):
    r = d[l]
    print("lane %d: nb %g Lsum %s W %s le0 %g" % (l, r[0], r[1:4], r[4:7], r[7]))
    for k in range(int(min(max(r[0], 0), 3))):
        print("   bounce %d: flags %g thr_k %s cN %g cf %g w2 %g slot %g | pass 2: rhob %s A_k %s bid %g flags %g" % (k, r[8 + 8 * k], r[9 + 8 * k:12 + 8 * k], r[12 + 8 * k], r[13 + 8 * k], r[14 + 8 * k], r[15 + 8 * k],
                                                                                                            r[32 + 8 * k:35 + 8 * k], r[35 + 8 * k:38 + 8 * k], r[38 + 8 * k], r[39 + 8 * k]))
np.save(sys.argv[5] if len(sys.argv) > 5 else "/tmp/dump.npy", d)
