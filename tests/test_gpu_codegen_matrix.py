"""Codegen matrix: the class-2 interior reverse sweep gives the same gradients whatever the code generated around it.

Round 3 ended with `k_interior_adjoint<2>` returning -8.4128 instead of -8.2052 when an unrelated knob of trav4.h (PSDR_STEAL_BREAK) changed
the register allocation of the kernel.  Round 4 found the cause in the compiler (vector constants re-materialised in front of a join block's
`s_or_b64 exec`; LABNOTES.md section 4): psdr_jit_amd/isa_lint.py finds the pattern in the ISA, and the build recipe compiles a unit that shows it a
second time with the scalar allocator that does not produce it (and fails if it still does).  This test keeps the property measurable: libpsdr_hip.so is built in several variants that
perturb the sweep kernel's code generation - among them the two that returned wrong numbers before the fix - and each variant, in a process of
its own (tools/sweep_check.py on a staged package copy), has to
  * pass the ISA lint,
  * satisfy the dot-product identity against HIP forward mode on the three environment-lit scenes (3e-4),
  * and return the same adjoint buffers from the sweep as from the record-and-probe form of the same library (3e-4)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# class-2 kernels only (-DPSDR_CLS_MASK=4: main unit + one kernel unit, about a minute per variant)
VARIANTS = {
    "m_default": ["-DPSDR_CLS_MASK=4"],
    "m_break": ["-DPSDR_CLS_MASK=4", "-DPSDR_STEAL_BREAK=2"],                       # returned -8.4128 / 384.59 / 303.26 before the fix
    "m_break_dump": ["-DPSDR_CLS_MASK=4", "-DPSDR_STEAL_BREAK=2", "-DPSDR_SWEEP_DUMP=3"],   # the diagnostic build that kept the defect visible
    "m_waves1": ["-DPSDR_CLS_MASK=4", "-DPSDR_ADJ_WAVES(cls)=1"],                   # the class-2 adjoint kernel without its two-waves bound (rounds 2-4): 512 registers to allocate in
    "m_o2": ["-DPSDR_CLS_MASK=4", "-O2"],
    "m_nosteal": ["-DPSDR_CLS_MASK=4", "-DPSDR_STEAL=0"],                           # showed the pattern at the commit that closed the item (an AGPR spill store in front of an exec
                                                                                    # restore; profiles/r04_sweep_defect_evidence.txt section 6) - which build shows it moves with every
                                                                                    # edit of the kernels; a variant's obj/lint.txt says whether the recipe compiled a unit twice
}


# the 8-wide tree of round 5 (a measurement build: twelve instead of nineteen node visits per ray at the same time) is kept alive by the driver's run, not only by the CPU
# builder test: tools/w8_check.py on the variant library
W8 = ("m_w8", ["-DPSDR_CLS_MASK=4", "-DPSDR_BVH_WIDTH=8"])


@pytest.fixture(scope="module")
def built():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    import __graft_entry__
    __graft_entry__.build()
    import variants
    procs = []
    for name, flags in list(VARIANTS.items()) + [W8]:
        if not variants.variant_is_current(name, flags):
            procs.append((name, subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "variants.py"), "build", name, " ".join(flags)],
                                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT)))
    for name, p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, "building variant %s failed:\n%s" % (name, out[-4000:])
    return variants


@pytest.mark.parametrize("name", list(VARIANTS))
def test_sweep_is_independent_of_the_code_around_it(built, name):
    isa_lint = built.load_build()._load_lint()
    lib = os.path.join(built.VDIR, name, "libpsdr_hip.so")
    findings = isa_lint.lint(lib)
    assert not findings, "variant %s: %s" % (name, findings[:3])
    top = built.stage(name)
    env = dict(os.environ)
    env.pop("PSDR_ADJ_PROBE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_check.py"), "--pkg", top], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env,
                       timeout=900)
    recs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert recs and recs[0]["library"].startswith(top), (r.stdout[-2000:], r.stderr[-2000:])      # the variant library is the one that ran
    cases = [x for x in recs if "case" in x]
    assert len(cases) == 3, (r.stdout[-2000:], r.stderr[-2000:])
    for c in cases:
        assert c["ok"], (name, c)
    assert r.returncode == 0


def test_eight_wide_tree(built):
    """-DPSDR_BVH_WIDTH=8 (bvh.h, trav4.h::t4_node, the refit and psdr_hip_scene_check_tree follow the width): 128-byte nodes, closest hits bit-equal to the oracle's brute force on
    the sphere box and on config 5's mesh, no containment violation, a config-5 renderD within the usual tolerance of the oracle"""
    name = W8[0]
    top = built.stage(name)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "w8_check.py"), "--pkg", top], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, timeout=900)
    recs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert recs and recs[0]["library"].startswith(top) and recs[0]["node_bytes"] == 128, (r.stdout[-2000:], r.stderr[-2000:])
    cases = [x for x in recs if "case" in x]
    assert len(cases) == 3, (r.stdout[-2000:], r.stderr[-2000:])
    for c in cases:
        assert c["ok"], c
    assert r.returncode == 0
