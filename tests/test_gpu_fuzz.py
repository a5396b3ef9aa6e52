"""Randomised parity: scene family, parameter, frame size, spp, depth, terms and seeds drawn by a generator instead of picked by hand.

tools/fuzz_parity.py compares renderD of the HIP path with the CPU oracle, tools/fuzz_reverse.py the reverse mode (<J^T w, v>) with the oracle's
forward mode (<w, J v>).  Round 4 ran 1650 + 660 cases of them without a failure (LABNOTES.md section 2); the suite keeps a short fixed-seed slice of each."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("tool,cases,seed", [("fuzz_parity.py", 40, 11), ("fuzz_reverse.py", 44, 12)])
def test_randomised_cases_against_the_oracle(tool, cases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(cases), str(seed)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0, tail
    assert "%d cases, 0 failed" % cases in r.stdout, tail
