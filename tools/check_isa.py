#!/usr/bin/env python3
"""ISA guard for kernels that manage registers by hand.

mla_rows128_kernel keeps its O accumulator in the fixed registers a0..a255 from inline asm; the compiler does not
know about those values. That is only sound while the compiler itself never touches an AGPR in that kernel (under
VGPR pressure it would use them as overflow space) and never spills. This script compiles csrc/mla_decode.hip to
assembly with the build's flags and checks, for every shipped instantiation (PROBE = 0):
  * no AGPR operand outside the ;;#ASMSTART ... ;;#ASMEND blocks,
  * no scratch (spill) instructions,
  * the kernel reports 256 AGPRs (so that the hardware allocation covers a0..a255).
Exit code 0 = all good. Used by tests/test_build_isa.py and by build.py --check.
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-xpu_amd"))


def kernels(asm_text, pattern):
    """yield (mangled name, lines) of every function whose label matches pattern"""
    lines = asm_text.splitlines()
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m and re.search(pattern, m.group(1)):
            j = i
            while j < len(lines) and "s_endpgm" not in lines[j]:
                j += 1
            yield m.group(1), lines[i:j + 1]
            i = j
        i += 1


def check(verbose=True):
    import build as sglk_build  # the build script: same compiler and flags

    src = os.path.join(sglk_build.CSRC, "mla_decode.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "mla.s")
        cmd = [sglk_build.HIPCC] + sglk_build.HIP_FLAGS + ["--cuda-device-only", "-S", "-o", out, src]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + p.stdout)
        text = open(out).read()
    agpr = re.compile(r"(?<![\w.])a(\d+|\[\d+:\d+\])(?![\w])|accvgpr")
    problems = []
    found = 0
    # shipped instantiations: template arguments <T, 0> mangle as ...ELi0EEE
    for name, body in kernels(text, r"mla_rows128_kernelI"):
        found += 1
        in_asm = False
        for ln in body:
            if "#ASMSTART" in ln:
                in_asm = True
                continue
            if "#ASMEND" in ln:
                in_asm = False
                continue
            code = ln.split(";")[0]
            if not code.strip() or code.lstrip().startswith("."):
                continue
            if not in_asm and agpr.search(code):
                problems.append("%s: compiler-generated AGPR use: %s" % (name, code.strip()))
            if "scratch_" in code:
                problems.append("%s: spill: %s" % (name, code.strip()))
        meta = re.search(r"\.amdhsa_kernel %s\b(.*?)\.end_amdhsa_kernel" % re.escape(name), text, re.S)
        if meta:
            nv = re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta.group(1))
            ao = re.search(r"\.amdhsa_accum_offset (\d+)", meta.group(1))
            if not (nv and ao and int(nv.group(1)) - int(ao.group(1)) >= 256):
                problems.append("%s: fewer than 256 AGPRs allocated (next_free_vgpr %s, accum_offset %s)"
                                % (name, nv and nv.group(1), ao and ao.group(1)))
        else:
            problems.append("%s: no kernel descriptor found" % name)
    if found == 0:
        problems.append("no mla_rows128_kernel instantiation found")
    if verbose:
        print("[check_isa] %d kernels checked, %d problems" % (found, len(problems)))
        for pr in problems[:20]:
            print("  " + pr)
    return problems


if __name__ == "__main__":
    sys.exit(1 if check() else 0)
