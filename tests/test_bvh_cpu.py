"""The tree builder of the HIP library on the CPU (psdr_jit_amd/csrc/hip/bvh.h is plain C++ on the host; replaces the GAS build of reference
src/scene/scene_optix.cpp:265-332): containment of every triangle in every ancestor's quantised box, one leaf per triangle, the breadth-first numbering of the
top of the tree (what trav4.h copies to LDS), independence of the thread count - for the shipped 4-wide tree and the 8-wide measurement build."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("width", [4, 8])
def test_tree_builder(tmp_path, width):
    exe = str(tmp_path / ("bvh_check_%d" % width))
    src = os.path.join(ROOT, "tests", "cpp", "bvh_check.cpp")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-DPSDR_BVH_WIDTH=%d" % width, src, "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    for n in (1, 2, 7, 20000):
        r = subprocess.run([exe, str(n)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.startswith("ok "), (n, r.stdout[-500:])
    nodes, depth, stack = (int(x) for x in r.stdout.split()[1:4])
    assert nodes > 1000 and depth >= (5 if width == 8 else 7) and stack > depth
