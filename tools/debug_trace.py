"""Debug aid: edge-to-edge stress rays through psdr_hip_trace vs the oracle's brute-force trace."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__; __graft_entry__.build()
import scenes, product
from oracle import oracle as orc
from psdr_jit_amd import cabi

spec = scenes.cbox_scene(32, 32, 1, 0, 0)
sc = product.build_scene(spec)
ref = orc.OracleScene(spec, [0])
rng = np.random.default_rng(3)
ti = np.asarray(ref.triangle_info())[:, :9].astype(np.float64)
m = 400000
def edge_points(k):
    tr = ti[rng.integers(0, len(ti), size=k)]
    p0, e1, e2 = tr[:, 0:3], tr[:, 3:6], tr[:, 6:9]
    s = rng.random(k)[:, None]
    s[rng.random(k) < 0.2] = 0.0
    which = rng.integers(0, 3, size=k)[:, None]
    return np.where(which == 0, p0 + s * e1, np.where(which == 1, p0 + s * e2, p0 + e1 + s * (e2 - e1)))
so, st = edge_points(m), edge_points(m)
sd = st - so
ln = np.linalg.norm(sd, axis=1, keepdims=True)
keep = ln[:, 0] > 1e-3
o = so[keep].astype(np.float32); d = (sd[keep] / ln[keep]).astype(np.float32)
n = len(o)
to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
tri = torch.empty(n, dtype=torch.int32, device="cuda"); uv = torch.empty((n, 2), dtype=torch.float32, device="cuda"); t = torch.empty(n, dtype=torch.float32, device="cuda")
cabi.check(cabi.lib().psdr_hip_trace(sc._hip_handle(), n, to.data_ptr(), td.data_ptr(), tri.data_ptr(), uv.data_ptr(), t.data_ptr(), None))
wtri, wuv, wt = ref.trace(o, d, use_bvh=False)
g = tri.cpu().numpy(); gt = t.cpu().numpy(); guv = uv.cpu().numpy()
bad = np.nonzero(g != wtri)[0]
print("rays", n, "mismatching tri", len(bad))
for i in bad[:12]:
    print(i, "o", o[i], "d", d[i], "gpu", g[i], gt[i], guv[i], "oracle", wtri[i], wt[i], wuv[i])
hit = (wtri >= 0) & (g == wtri)
print("uv/t mismatches among equal tris:", int((guv[hit] != wuv[hit]).any(axis=1).sum()), int((gt[hit] != wt[hit]).sum()))
