"""Timing of the reference's Forward_AD_envmap notebook configuration (bunny_low.obj + MicrofacetBSDF under ballroom_1k.exr,
128x128, 128 samples, PathTracer(1); one derivative term per cell) through examples/tutorials.py.
    python tools/tutorial_env_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch
import __graft_entry__; __graft_entry__.build()
import tutorials as tut
for term in ("interior", "primary", "secondary"):
    tut.forward_ad_envmap(128, 128, 128, term); torch.cuda.synchronize()
    t = time.time()
    img, d = tut.forward_ad_envmap(128, 128, 128, term)
    torch.cuda.synchronize()
    print("%-9s scene build + configure + renderD + forward_grad: %.1f ms   (|d| sum %.3g)" % (term, (time.time() - t) * 1e3, float(d.abs().sum())))
