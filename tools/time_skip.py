"""What continuing samplers cost: the C3 frame with the samplers' skip counters at 0 and far into a run (every work item skips its stream ahead: sampler.h::advance).   python tools/time_skip.py"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__; __graft_entry__.build()
import psdr_jit_amd
from psdr_jit_amd import cabi
import scenes, product
spec = scenes.cbox_scene(512, 512, 32, 32, 32, param="box_x")
sc = product.build_scene(spec)
n = 512 * 512
buf = torch.empty((2, n, 3), dtype=torch.float32, device="cuda")
for skip in (0, 31, 31 * 1000, 31 * 10 ** 6, 2 ** 40 + 12345):
    a = cabi.make_args(max_depth=3, seeds=(1, 2, 3), skips=(skip, skip, skip), terms=7)
    for _ in range(3):
        cabi.check(cabi.lib().psdr_hip_render_d_fwd(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        cabi.check(cabi.lib().psdr_hip_render_d_fwd(sc._hip_handle(), C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), None))
    e1.record(); torch.cuda.synchronize()
    print("skip %-16d %.3f ms   checksum %.6f %.6f" % (skip, e0.elapsed_time(e1) / 20, float(buf[0].double().sum()), float(buf[1].double().sum())))
