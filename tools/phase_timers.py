"""Where the wave cycles of the path kernels go: needs a diagnostic build whose counted kernels time their phases with the shader clock
(PSDR_HIP_FLAGS="-DPSDR_DIAG=6|7|8", see paths.h / scene_dev.h / trav4.h) and prints the four buckets per term.

    PSDR_HIP_FLAGS="-DPSDR_CLS_MASK=2 -DPSDR_DIAG=6" python tools/phase_timers.py c3
    PSDR_HIP_FLAGS="-DPSDR_CLS_MASK=4 -DPSDR_DIAG=8" python tools/phase_timers.py c5 [res] [spp]
"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__; __graft_entry__.build()
import scenes, product
from psdr_jit_amd import cabi

LABELS = {
    "6": ("fetch + regeneration", "drawing the two rays", "trace2", "consume hits + path end"),
    "7": ("trace2 set-up", "filter loop", "exact rounds", "(exact rounds run: a count, not cycles)"),
    "9": ("candidates", "active rays", "exact rounds x 64", "trace2 calls x 64"),
    "8": ("shading phase", "node bursts", "pair tests", "ray hand-over + bookkeeping"),
}
what = sys.argv[1] if len(sys.argv) > 1 else "c3"
diag = [f.split("=")[1] for f in os.environ.get("PSDR_HIP_FLAGS", "").split() if f.startswith("-DPSDR_DIAG=")]
labels = LABELS.get(diag[0] if diag else "", ("rays", "nodes", "tris", "hits"))
if what == "c3":
    res, spp = 512, 32
    spec = scenes.cbox_scene(res, res, spp, spp, spp, param="light_x")
elif what == "sphere":
    res, spp = 512, 32
    spec = scenes.sphere_scene(res, res, spp, spp, spp)
else:
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    spec = scenes.config5_scene(res, res, spp, spp, spp, level=6, env_res=(1024, 512))
sc = product.build_scene(spec)
buf = torch.empty((2, res * res, 3), dtype=torch.float32, device="cuda")
for terms, name in ((1, "interior"), (2, "primary edges"), (4, "secondary edges")):
    c = cabi.Counters()
    a = cabi.make_args(max_depth=3, seeds=(0, 0, 0), terms=terms)
    cabi.check(cabi.lib().psdr_hip_render_d_fwd_counted(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), C.byref(c), None))
    v = (c.rays, c.nodes_visited, c.tris_tested, c.shaded_hits)
    tot = float(sum(v[:3]) + (v[3] if diag and diag[0] != "7" else 0)) or 1.0
    print("%-16s" % name, "  ".join("%s %.3g (%.1f %%)" % (l, x, 100.0 * x / tot) for l, x in zip(labels, v)))
