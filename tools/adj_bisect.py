"""Diagnostic: where do two forms of the interior reverse pass (the sweep and record-and-probe, PSDR_ADJ_PROBE=1) of ONE library
disagree?  Dumps the adjoint buffers of an environment-lit test scene for the whole frame, per pixel row and per pixel (pix_ids), at
depths 1-3, so that two dumps of the same library can be compared down to the first differing pixel and bounce.

    python tools/adj_bisect.py dump OUT.npz [box_x|albedo] [balls]      # honours PSDR_ADJ_PROBE
    python tools/adj_bisect.py cmp A.npz B.npz
"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, "tests"); sys.path.insert(0, ".")


def dump(out, param="box_x", balls=False):
    import torch
    import __graft_entry__
    __graft_entry__.build()
    from psdr_jit_amd import cabi
    import product, scenes
    res, spp = 40, 8
    spec = scenes.envmap_scene(res, res, spp, 0, 0, param=param, area_light=True, balls=balls)
    sc = product.build_scene(spec)
    snap = sc._snapshot()
    d_tri = np.asarray(snap["d_triangles"], np.float64)
    d_bsdf = np.array([b.d_reflectance for b in spec.bsdfs], np.float64)
    n = res * res
    gen = torch.Generator(device="cpu").manual_seed(3)
    w_full = (torch.rand((n, 3), generator=gen) + 0.5)
    nt, nb, ne = d_tri.shape[0], len(spec.bsdfs), len(spec.emitters)

    def bwd(depth, pix=None):
        g_tri = torch.zeros((nt, 22), dtype=torch.float32, device="cuda")
        g_bsdf = torch.zeros((nb, 3), dtype=torch.float32, device="cuda")
        g_em = torch.zeros((ne, 3), dtype=torch.float32, device="cuda")
        g_sec = torch.zeros((1, 6), dtype=torch.float32, device="cuda")
        g_prim = torch.zeros((1, 4), dtype=torch.float32, device="cuda")
        g = cabi.Grads(g_tri.data_ptr(), g_bsdf.data_ptr(), g_em.data_ptr(), g_sec.data_ptr(), g_prim.data_ptr())
        if pix is None:
            w = w_full.to("cuda")
            a = cabi.make_args(max_depth=depth, seeds=(7, 8, 9), terms=1)
        else:
            ids = torch.tensor(pix, dtype=torch.int32, device="cuda")
            w = w_full[torch.tensor(pix, dtype=torch.long)].contiguous().to("cuda")
            a = cabi.make_args(max_depth=depth, seeds=(7, 8, 9), terms=1, pix_ids_ptr=ids.data_ptr(), n_pix=len(pix))
        cabi.check(cabi.lib().psdr_hip_render_d_bwd(sc._hip_handle(), C.byref(a), w.data_ptr(), C.byref(g), None))
        torch.cuda.synchronize()
        return g_tri.cpu().numpy().astype(np.float64), g_bsdf.cpu().numpy().astype(np.float64), g_em.cpu().numpy().astype(np.float64)

    o = {"d_tri": d_tri, "d_bsdf": d_bsdf}
    for depth in (1, 2, 3):
        gt, gb, ge = bwd(depth)
        o["full_tri_%d" % depth], o["full_bsdf_%d" % depth], o["full_em_%d" % depth] = gt, gb, ge
        rows_t, rows_b = [], []
        for r in range(res):
            gt, gb, ge = bwd(depth, list(range(r * res, (r + 1) * res)))
            rows_t.append(gt); rows_b.append(gb)
        o["row_tri_%d" % depth], o["row_bsdf_%d" % depth] = np.stack(rows_t), np.stack(rows_b)
        px_t, px_b = [], []
        for p in range(n):
            gt, gb, ge = bwd(depth, [p])
            px_t.append((gt * d_tri).sum() if param == "box_x" else np.abs(gt).sum()); px_b.append(gb)
        o["px_tri_%d" % depth], o["px_bsdf_%d" % depth] = np.array(px_t), np.stack(px_b)
    np.savez(out, **o)
    print("wrote", out)


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    d_tri, d_bsdf = A["d_tri"], A["d_bsdf"]
    for depth in (1, 2, 3):
        print("=== depth", depth)
        for lvl in ("full", "row"):
            ta, tb = A["%s_tri_%d" % (lvl, depth)], B["%s_tri_%d" % (lvl, depth)]
            ba, bb = A["%s_bsdf_%d" % (lvl, depth)], B["%s_bsdf_%d" % (lvl, depth)]
            if lvl == "full":
                print("  full: <g_tri, d_tri> %.6f vs %.6f   g_bsdf sum %.6f vs %.6f" % ((ta * d_tri).sum(), (tb * d_tri).sum(), ba.sum(), bb.sum()))
                err = np.abs(ta - tb).sum(axis=1)
                bad = np.argsort(-err)[:6]
                print("  triangle rows with the largest |diff|:", [(int(i), float("%.3g" % err[i]), float("%.3g" % np.abs(ta[i]).sum())) for i in bad])
                comp = np.abs(ta - tb).sum(axis=0)
                print("  per component:", np.array2string(comp, precision=3))
                print("  g_bsdf\n", ba, "\n", bb)
            else:
                st, sb = (ta * d_tri).sum(axis=(1, 2)), (tb * d_tri).sum(axis=(1, 2))
                bad = [(r, float("%.5g" % st[r]), float("%.5g" % sb[r])) for r in range(len(st)) if abs(st[r] - sb[r]) > 1e-4 * (abs(st).max() + 1e-12)]
                print("  pixel rows that differ (tri):", bad)
                sa, sbb = ba.sum(axis=(1, 2)), bb.sum(axis=(1, 2))
                bad = [(r, float("%.6g" % sa[r]), float("%.6g" % sbb[r])) for r in range(len(sa)) if abs(sa[r] - sbb[r]) > 1e-4 * (abs(sa).max() + 1e-12)]
                print("  pixel rows that differ (bsdf):", bad)
        pa, pb = A["px_tri_%d" % depth], B["px_tri_%d" % depth]
        bad = [(p, float("%.5g" % pa[p]), float("%.5g" % pb[p])) for p in range(len(pa)) if abs(pa[p] - pb[p]) > 1e-3 * (abs(pa[p]) + abs(pb[p])) + 1e-6]
        print("  pixels that differ (tri): %d of %d" % (len(bad), len(pa)), bad[:40])
        qa, qb = A["px_bsdf_%d" % depth].sum(axis=(1, 2)), B["px_bsdf_%d" % depth].sum(axis=(1, 2))
        bad = [(p, float("%.6g" % qa[p]), float("%.6g" % qb[p])) for p in range(len(qa)) if abs(qa[p] - qb[p]) > 1e-3 * (abs(qa[p]) + abs(qb[p])) + 1e-6]
        print("  pixels that differ (bsdf): %d of %d" % (len(bad), len(qa)), bad[:40])
        print("  sums over pixels: tri %.6f vs %.6f, bsdf %.6f vs %.6f" % (pa.sum(), pb.sum(), qa.sum(), qb.sum()))


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "box_x", len(sys.argv) > 4 and sys.argv[4] == "balls")
    else:
        cmp(sys.argv[2], sys.argv[3])
