"""Build-time guard (CPU, needs hipcc): kernels that keep values in hand-assigned registers or count their waits by
hand must not share those registers with compiler-allocated values / must not spill. The check itself lives in
sgl-kernel-xpu_amd/build.py (check_isa) and fails the build; this test runs it on the assembly of the last build."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_hand_managed_registers_are_safe():
    spec = importlib.util.spec_from_file_location("sglk_build", os.path.join(ROOT, "sgl-kernel-xpu_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod._asm_path("mla_decode.hip")):
        mod.build(with_torch=False, verbose=False)
    problems = mod.check_isa(verbose=False)
    assert problems == [], "\n".join(problems)
