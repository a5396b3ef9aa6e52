"""Diagnostic: prints (lhs, rhs, scale) of the reverse-sweep dot-product cases of environment-lit scenes (tests/test_gpu_adjoint.py::test_interior_sweep_environment_map);
lhs = forward-mode tangents . w, rhs = adjoints . tangents.  LABNOTES.md section 4, BVH third pass, open item.   python tools/dp_env_case.py"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import pytest, scenes
import test_gpu_adjoint as t
import torch, __graft_entry__
__graft_entry__.build()
import psdr_jit_amd
from psdr_jit_amd import cabi
env = (torch, psdr_jit_amd, cabi)
for param, balls in (("box_x", False), ("albedo", False), ("albedo", True)):
    print(param, balls, t._dot_product_case(env, scenes.envmap_scene(40, 40, 8, 0, 0, param=param, area_light=True, balls=balls), depth=3, terms=1))
