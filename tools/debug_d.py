import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import psdr_jit_amd as psdr, product, scenes
from oracle import oracle as orc
param = sys.argv[1] if len(sys.argv) > 1 else 'light_x'
spec = scenes.cbox_scene(64, 64, 8, 8, 8, param=param)
sc = product.build_scene(spec); ref = orc.OracleScene(spec, [0])
integ = psdr.PathTracer(2)
for terms in (1, 2, 4):
    img, dimg = psdr.render_d_fwd(integ, sc, 0, seed=21, terms=terms)
    wimg, wd = ref.render_d(max_depth=2, seeds=(21, 21, 21), terms=terms)
    d = dimg.cpu().numpy(); diff = np.abs(d - wd).max(1)
    print('terms', terms, 'relL2', product.rel_l2(d, wd), 'img', product.rel_l2(img.cpu().numpy(), wimg) if terms == 1 else 0, 'norm', np.linalg.norm(wd), 'maxabs', np.abs(wd).max())
    idx = np.argsort(-diff)[:8]
    for i in idx: print('   pix', i, 'got', d[i], 'want', wd[i], 'primal got', img.cpu().numpy()[i], 'want', wimg[i])
    print('   n pix diff > 1e-4*max:', (diff > 1e-4 * np.abs(wd).max()).sum())
