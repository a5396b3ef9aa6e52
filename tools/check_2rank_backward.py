"""Lane-sharded renderD + loss.backward() on 2 ranks (one GPU, gloo) against the single-process result.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/check_2rank_backward.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch.distributed as dist
import psdr_jit_amd as psdr
from psdr_jit_amd import Matrix4fC, Matrix4fD
import tutorials as tut


def grad_of(distributed):
    sc = tut._scene(64, 64, 8, 8, 8)
    tut._camera(sc, 208., 273., -800.)
    refl = torch.tensor([0.5, 0.4, 0.3], requires_grad=True)
    sc.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light"); sc.add_BSDF(psdr.DiffuseBSDF(refl), "cat"); sc.add_BSDF(psdr.DiffuseBSDF([0.95, 0.95, 0.95]), "white")
    cb = os.path.join(tut.DATA, "cbox")
    sc.add_Mesh(os.path.join(cb, "cbox_luminaire.obj"), Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., -0.5], [0., 0., 1., 0.], [0., 0., 0., 1.]]), "light", psdr.AreaLight([20.0, 20.0, 8.0]))
    for f, b in (("smallbox", "cat"), ("largebox", "cat"), ("floor", "white"), ("back", "white")):
        sc.add_Mesh(os.path.join(cb, "cbox_%s.obj" % f), Matrix4fC(tut.I4), b, None)
    P = psdr.FloatD(0.).requires_grad_()
    sc.param_map["Mesh[0]"].set_transform(Matrix4fD([[1., 0., 0., P * 100], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    sc.configure(); sc.configure([0])
    img = psdr.PathTracer(2).renderD(sc, 0, seed=3)
    w = torch.linspace(0.5, 1.5, img.numel(), device=img.device).reshape(img.shape)
    (img * w).sum().backward()
    return img.detach().cpu().numpy(), float(P.grad), refl.grad.numpy().copy()


single = grad_of(False) if "RANK" not in os.environ else None
if "RANK" in os.environ:
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    img, gp, gr = grad_of(True)                 # sharded over the 2 ranks, all-reduced
    # forward mode over the two ranks in both collective forms (psdr_jit_amd._render_terms): the interior term's all_reduce started ahead of the edge terms /
    # one all_reduce of everything - the same image and derivative
    def fwd_of():
        sc = tut._scene(64, 64, 8, 8, 8)
        tut._camera(sc, 208., 273., -800.)
        sc.add_BSDF(psdr.DiffuseBSDF([0.0, 0.0, 0.0]), "light"); sc.add_BSDF(psdr.DiffuseBSDF([0.5, 0.4, 0.3]), "cat")
        cb = os.path.join(tut.DATA, "cbox")
        sc.add_Mesh(os.path.join(cb, "cbox_luminaire.obj"), Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., -0.5], [0., 0., 1., 0.], [0., 0., 0., 1.]]), "light", psdr.AreaLight([20.0, 20.0, 8.0]))
        for f in ("smallbox", "largebox", "floor", "back"):
            sc.add_Mesh(os.path.join(cb, "cbox_%s.obj" % f), Matrix4fC(tut.I4), "cat", None)
        Pf = psdr.FloatD(0.).requires_grad_()
        sc.param_map["Mesh[0]"].set_transform(Matrix4fD([[1., 0., 0., Pf * 100], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
        sc.configure(); sc.configure([0])
        leaf = sc.param_map["Mesh[0]"].to_world_left
        d = np.zeros((4, 4), np.float32); d[0, 3] = 100.0
        return [np.stack([t.cpu().numpy() for t in psdr.render_d_fwd(psdr.PathTracer(2), sc, 0, seed=5 + k, tangents={leaf: d})]) for k in range(2)]
    os.environ["PSDR_SINGLE_COLLECTIVE"] = "0"
    f_split = fwd_of()
    os.environ["PSDR_SINGLE_COLLECTIVE"] = "1"
    f_single = fwd_of()
    os.environ.pop("PSDR_SINGLE_COLLECTIVE")
    # PSDR_SHARD=rows: contiguous pixel-row tiles (psdr_render_args.shard_mode 1), the interior term assembled by all_gather_into_tensor - forward and backward
    os.environ["PSDR_SHARD"] = "rows"
    f_rows = fwd_of()
    img_r, gp_r, gr_r = grad_of(True)
    os.environ.pop("PSDR_SHARD")
    dist.destroy_process_group()
    # reference: the same in one process (after the group is gone psdr shards over 1 rank)
    img1, gp1, gr1 = grad_of(False)
    f_one = fwd_of()
    for a, b, r, c in zip(f_split, f_single, f_rows, f_one):
        for x, nm in ((a, "split"), (b, "single"), (r, "rows")):
            e0, e1 = np.linalg.norm(x[0] - c[0]) / np.linalg.norm(c[0]), np.linalg.norm(x[1] - c[1]) / np.linalg.norm(c[1])
            assert e0 < 1e-5 and e1 < 1e-4, (nm, e0, e1)
    if int(os.environ["RANK"]) == 0:
        print("image rel L2 %.2e   dP %.6f vs %.6f   d refl %s vs %s" % (np.linalg.norm(img - img1) / np.linalg.norm(img1), gp, gp1, gr, gr1))
        assert np.linalg.norm(img - img1) / np.linalg.norm(img1) < 1e-5 and abs(gp - gp1) < 1e-4 * max(1.0, abs(gp1)) and np.allclose(gr, gr1, rtol=1e-4)
        assert np.linalg.norm(img_r - img1) / np.linalg.norm(img1) < 1e-5 and abs(gp_r - gp1) < 1e-4 * max(1.0, abs(gp1)) and np.allclose(gr_r, gr1, rtol=1e-4), (gp_r, gp1, gr_r, gr1)
        print("2-rank forward (split / single collective / row tiles) and backward OK")
else:
    print(single[1], single[2])
