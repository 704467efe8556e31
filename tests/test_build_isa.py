"""Build-time guard (CPU, needs hipcc): kernels that keep values in hand-assigned registers must not share those
registers with compiler-allocated values. See tools/check_isa.py."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_mla_rows128_registers_are_hand_owned():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa

    problems = check_isa.check(verbose=False)
    assert problems == [], "\n".join(problems)
