"""Diagnostic: smallest list of pixels (pix_ids) on which the reverse sweep and the record-and-probe form (PSDR_ADJ_PROBE, read per call) of the
SAME library disagree.  (A pixel's samples depend on its position in the list - the lane index seeds the stream - so every comparison runs both
forms on the same list.)   python tools/adj_subset.py [depth] [row]"""
import os
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import __graft_entry__
__graft_entry__.build()
from psdr_jit_amd import cabi
import product, scenes

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
row = int(sys.argv[2]) if len(sys.argv) > 2 else 2
res, spp = 40, 8
spec = scenes.envmap_scene(res, res, spp, 0, 0, param="box_x", area_light=True, balls=False)
sc = product.build_scene(spec)
snap = sc._snapshot()
nt, nb, ne = np.asarray(snap["d_triangles"]).shape[0], len(spec.bsdfs), len(spec.emitters)
gen = torch.Generator(device="cpu").manual_seed(3)
w_full = (torch.rand((res * res, 3), generator=gen) + 0.5)


def bwd(pix, probe=False):
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
    w = w_full[torch.tensor(pix, dtype=torch.long)].contiguous().to("cuda")
    a = cabi.make_args(max_depth=depth, seeds=(7, 8, 9), terms=1, pix_ids_ptr=ids.data_ptr(), n_pix=len(pix))
    cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
    torch.cuda.synchronize()
    return np.concatenate([g_bsdf.cpu().numpy().astype(np.float64).ravel(), g_em.cpu().numpy().astype(np.float64).ravel(), g_tri.cpu().numpy().astype(np.float64).ravel()])


pixels = list(range(row * res, (row + 1) * res))


def bad(S):
    got = bwd(S)
    want = bwd(S, probe=True)
    k = 3 * nb + 3 * ne          # colours and emitters (the triangle rows of the scene box differ by design: the probe form leaves them out)
    return np.abs(got[:k] - want[:k]).max() > 1e-4 * (np.abs(want[:k]).max() + 1e-9), got, want


b, got, want = bad(pixels)
print("row", row, "depth", depth, "joint differs from singles:", b)
print(got[:3 * nb + 3 * ne], "\n", want[:3 * nb + 3 * ne])
S = list(pixels)
n = 2
while len(S) >= 2 and b:
    chunk = max(1, len(S) // n)
    reduced = False
    for i in range(0, len(S), chunk):
        T = S[:i] + S[i + chunk:]
        if len(T) >= 1 and bad(T)[0]:
            S = T; n = max(n - 1, 2); reduced = True
            break
    if not reduced:
        if chunk == 1:
            break
        n = min(len(S), n * 2)
print("minimal failing set:", S)
b, got, want = bad(S)
np.set_printoptions(precision=6, linewidth=220, suppress=True)
k = 3 * nb + 3 * ne
print("sweep g_bsdf | g_em\n", got[:k], "\nprobe\n", want[:k])
gt, wt = got[k:].reshape(nt, 22), want[k:].reshape(nt, 22)
for r in np.argsort(-np.abs(gt - wt).sum(axis=1))[:10]:
    print(" tri row", int(r), "\n   sweep", gt[r], "\n   probe", wt[r])
# order sensitivity and repeatability
print("reversed order fails:", bad(S[::-1])[0], " repeat:", [bool(bad(S)[0]) for _ in range(3)])
for k in range(1, len(S)):
    print("prefix", S[:k], bad(S[:k])[0])
