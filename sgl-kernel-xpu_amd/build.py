#!/usr/bin/env python3
"""hipcc build of the gfx950 kernels and the torch operator registry.

Replaces the reference's cmake/SYCL.cmake + src/BuildOnLinux.cmake (icpx, one
.so per SYCL TU, AOT for `bmg`) with a direct hipcc build for gfx950:

  csrc/*.hip                -> build/obj/*.o  -> python/sgl_kernel/libsglk.so        (C-ABI, torch-free)
  csrc/torch_extension_hip.cc                 -> python/sgl_kernel/common_ops.abi3.so (TORCH_LIBRARY)

Both outputs are in-tree (git-ignored) so that they travel to the GPU box.
Incremental: a source is recompiled when it, a header or this script is newer
than its object.

After compiling, the ISA of the kernels that manage registers or wait counters by hand
(mla_decode.hip, gemm_8bit.hip) is checked (check_isa below): a compiler that spills or parks
values in the accumulator registers those kernels own would give silently wrong results, so a
failed check fails the build. Validated with ROCm 7.2.0 (AMD clang 20, /opt/rocm/bin/hipcc).

--probes additionally builds the DIAGNOSTIC library build/libsglk_probes.so (-DSGLK_PROBES: main-loop
variants and timing probes with garbage results behind sglk_debug_* switches) and tools/kbench against
it. The release libsglk.so contains neither the switches nor the probe code.

Usage: python build.py [--jobs N] [--force] [--no-torch] [--probes] [--check]
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys
import sysconfig
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build", "obj")
PKG = os.path.join(HERE, "python", "sgl_kernel")
INCLUDE = os.path.join(ROOT, "include")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HIP_FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
    "-fno-gpu-rdc", f"-I{INCLUDE}", f"-I{CSRC}", "-Wall", "-Wno-unused-function",
    "-Wno-implicit-fallthrough", "-Wno-unused-variable", "-Wno-shift-negative-value",
    "-Wno-unused-local-typedef",
]


# Per-file flags. -fno-slp-vectorize: a plain -O3 build packs adjacent scalar f32 adds / multiplies into v_pk_*_f32, which
# issue SLOWER next to MFMAs than the two scalar instructions they replace (MI355X_MICROARCH.md, "price of one filler
# beside MFMAs"; kbench issue: 4 v_fma_f32 per MFMA 2.88 PFLOP/s, 2 v_pk_fma_f32 2.40), and whose register pairs create
# false dependencies on ring loads.
FILE_FLAGS = {
    "attn_fwd.hip": ["-fno-slp-vectorize"],
    "moe_w4a16.hip": ["-fno-slp-vectorize"],
    "moe_bf16.hip": ["-fno-slp-vectorize"],
    "moe_persist.hip": ["-fno-slp-vectorize"],
    "mla_decode.hip": ["-fno-slp-vectorize"],
}


def newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def run(cmd):
    t0 = time.time()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), p.stdout))
    return time.time() - t0, p.stdout


# sources whose device assembly is kept next to the object (-save-temps=obj) for check_isa
ISA_CHECKED = ("mla_decode.hip", "gemm_8bit.hip", "attn_fwd.hip", "moe_w4a16.hip", "moe_bf16.hip", "moe_persist.hip")


def _asm_path(src_name, obj_dir=None):
    return os.path.join(obj_dir or OBJ, src_name[:-4] + "-hip-amdgcn-amd-amdhsa-%s.s" % ARCH)


def _functions(asm_text, pattern):
    """yield (mangled name, body lines) of every function whose label matches pattern"""
    import re
    lines = asm_text.splitlines()
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m and re.search(pattern, m.group(1)):
            j = i + 1  # to the function's end label (an early return puts an s_endpgm in the middle of the body)
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            yield m.group(1), lines[i:j + 1]
            i = j
        i += 1


def check_isa(verbose=True):
    """Guards for hand-managed registers, run on the assembly the build just produced.

    * mla_rows128_kernel / mla_rows128x_kernel keep O in the fixed registers a0..a255 named only in inline asm: the compiler must not
      touch the AGPR file itself in that kernel (no AGPR operand outside ;;#ASMSTART..;;#ASMEND), must not spill,
      and the kernel descriptor must allocate 256 AGPRs.
    * gemm_8bit_persist_kernel / gemm_fp8bw_x32_kernel count their LDS waits by hand: a VGPR spill (scratch access = vector-memory
      traffic inside the counted vmcnt window) breaks the counts.
    * attn_prefill_kernel / attn_decode_kernel are sized for two 256-register waves per SIMD: a spill means the tile
      shape no longer fits.
    * the grouped-GEMM kernels (moe_w4a16, moe_bf16) keep several 128-deep blocks of loads in flight in register rings:
      a spill inside the K loop is reloaded through scratch, behind an s_waitcnt vmcnt(0) that empties the rings
      (measured: 2.5 TB/s instead of > 4).
    Returns the list of problems (empty = good)."""
    import re
    problems = []
    agpr = re.compile(r"(?<![\w.])a(\d+|\[\d+:\d+\])(?![\w])|accvgpr")
    path = _asm_path("mla_decode.hip")
    if not os.path.exists(path):
        return ["%s missing (build with this script first)" % path]
    text = open(path).read()
    found = 0
    for name, body in _functions(text, r"mla_rows128[xz]_kernelI"):
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
    path = _asm_path("gemm_8bit.hip")
    if not os.path.exists(path):
        problems.append("%s missing" % path)
    else:
        text = open(path).read()
        n = 0
        for name, body in _functions(text, r"(gemm_8bit_persist2?_kernelI|gemm_fp8bw_x32_kernelI)"):
            n += 1
            # A scratch access is vector-memory traffic: inside a K loop it sits in the hand-counted vmcnt window (the waits
            # stay safe - "all but the N youngest" only ever waits for more - but every block then drains what it should
            # leave in flight). The one-launch kernel's phase loop (depth 1: prologues, tile descriptors) may park values
            # there; a loop inside it (the unit / K-block loops, depth >= 2) may not.
            depth = 0
            for ln in body:
                m = re.search(r"(?:in Loop: Header=\S+|Loop Header:) Depth=(\d+)", ln)
                if m:
                    depth = int(m.group(1))
                elif re.match(r"^\.LBB\w+:", ln):
                    depth = 0  # (a block outside every loop carries no annotation; one inside gets it on the next line)
                code = ln.split(";")[0]
                if "scratch_" in code and (depth >= 2 or "gemm_fp8bw_x32_kernel" not in name):
                    problems.append("%s: spill at loop depth %d: %s" % (name, depth, code.strip()))
                    break
        found += n
        if n == 0:
            problems.append("no gemm_8bit_persist_kernel instantiation found")
    # (moe_w4a16: the group >= 128 instantiations, PB = 1 - AWQ / GPTQ / Mixtral checkpoints - and the mxfp4 ones are
    # build-breaking; the group-32 / 64 int4 tiles, PB = 2 / 4, are only reported - `build.py --check` lists every
    # spilling instantiation as a warning, see check_isa_warnings; none spills with ROCm 7.2.0 at present)
    for src_name, pat in (("attn_fwd.hip", r"attn_prefill_kernelI"), ("attn_fwd.hip", r"attn_decode_kernelI"),
                          ("moe_w4a16.hip", r"moe_w4a16_kernelIDF16.Li\dELi\dELi1ELi\dE"),
                          ("moe_w4a16.hip", r"moe_w4a16_kernelIDF16.Li\dELi\dELi4ELi1E"),
                          ("moe_w4a16.hip", r"moe_w4a16_ksplit_kernelI"),
                          ("moe_bf16.hip", r"moe_bf16_kernelI"), ("moe_persist.hip", r"moe_persist_kernelI")):
        path = _asm_path(src_name)
        if not os.path.exists(path):
            problems.append("%s missing" % path)
            continue
        n = 0
        for name, body in _functions(open(path).read(), pat):
            n += 1
            # attn_prefill_kernel at d = 256 takes the whole register file and may park values the tile loop never touches in
            # scratch - stored in front of the loop, reloaded behind it. That is harmless; a scratch access INSIDE a loop (the
            # assembler's block annotations say which blocks those are) is the failure this check exists for.
            loop_only = src_name == "attn_fwd.hip" and "attn_prefill_kernel" in pat and "Li256E" in name
            in_loop = False
            for ln in body:
                if "; in Loop:" in ln or "Loop Header:" in ln:
                    in_loop = True
                elif re.match(r"^\.LBB\w+:", ln):
                    in_loop = False
                code = ln.split(";")[0]
                if "scratch_" in code and (in_loop or not loop_only):
                    problems.append("%s: spill: %s" % (name, code.strip()))
                    break
        found += n
        if n == 0:
            problems.append("no %s instantiation found" % pat)
    if verbose:
        print("[build] check_isa: %d kernels checked, %d problems" % (found, len(problems)), flush=True)
        for pr in problems[:20]:
            print("  " + pr)
    return problems


def check_isa_warnings(verbose=True):
    """Non-fatal pass (build.py --check): every moe_w4a16_kernel instantiation that spills, whatever its group size."""
    path = _asm_path("moe_w4a16.hip")
    if not os.path.exists(path):
        return []
    warn = []
    for name, body in _functions(open(path).read(), r"moe_w4a16_kernelI"):
        n = sum(1 for ln in body if "scratch_" in ln.split(";")[0])
        if n:
            warn.append("%s: %d scratch accesses (spills)" % (name, n))
    if verbose:
        print("[build] check_isa (warnings only): %d moe_w4a16_kernel instantiations spill" % len(warn), flush=True)
        for w in warn:
            print("  warning: " + w)
    return warn


def _compile_all(obj_dir, extra_flags, jobs, force, verbose, headers):
    os.makedirs(obj_dir, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    todo, objs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s[:-4] + ".o")
        objs.append(obj)
        if force or newer(obj, [src] + headers) or (s in ISA_CHECKED and not os.path.exists(_asm_path(s, obj_dir))):
            todo.append((s, src, obj))
    if todo:
        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            futs = {}
            for s, src, obj in todo:
                flags = HIP_FLAGS + FILE_FLAGS.get(s, []) + extra_flags + (["-save-temps=obj"] if s in ISA_CHECKED else [])
                futs[ex.submit(run, [HIPCC] + flags + ["-c", src, "-o", obj])] = src
            for f in cf.as_completed(futs):
                dt, out = f.result()
                if verbose:
                    print("[build] hipcc %-32s %5.1fs" % (os.path.basename(futs[f]), dt), flush=True)
                    if out.strip():
                        print(out)
        # -save-temps=obj leaves large intermediates next to the objects: keep only the device assembly
        for f in os.listdir(obj_dir):
            if f.endswith((".bc", ".hipi", ".out", ".resolution.txt", ".hipfb")) or (f.endswith(".s") and "-host-" in f) or \
               (f.endswith(".o") and "-hip-amdgcn" in f) or (f.endswith(".o") and "-host-" in f):
                os.remove(os.path.join(obj_dir, f))
    return objs, bool(todo)


def build_probes(jobs=None, force=False, verbose=True):
    """Diagnostic library + kbench (not shipped, not loaded by the Python package)."""
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    headers.append(os.path.abspath(__file__))
    jobs = jobs or min(8, os.cpu_count() or 1)
    bdir = os.path.join(HERE, "build")
    objs, changed = _compile_all(os.path.join(bdir, "obj_probes"), ["-DSGLK_PROBES"], jobs, force, verbose, headers)
    lib = os.path.join(bdir, "libsglk_probes.so")
    if force or changed or newer(lib, objs):
        run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + objs)
    kb_src = os.path.join(ROOT, "tools", "kbench.cpp")
    kb = os.path.join(bdir, "kbench")
    if force or newer(kb, [kb_src, lib]):
        dt, out = run([HIPCC, f"--offload-arch={ARCH}", "-O2", "-std=c++17", kb_src, f"-I{INCLUDE}", f"-L{bdir}",
                       "-lsglk_probes", "-Wl,-rpath,$ORIGIN", "-o", kb])
        if verbose:
            print("[build] kbench %5.1fs" % dt, flush=True)
    return lib


def build_ceiling(force=False, verbose=True):
    """bench.py's MFMA ceiling probe (tools/mfma_ceiling.hip): its own small library, not linked into the product."""
    src = os.path.join(ROOT, "tools", "mfma_ceiling.hip")
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    lib = os.path.join(bdir, "libsglk_ceiling.so")
    if force or newer(lib, [src]):
        dt, out = run([HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", src, "-o", lib])
        if verbose:
            print("[build] hipcc %-32s %5.1fs" % ("mfma_ceiling.hip", dt), flush=True)
    return lib


def build(jobs=None, force=False, with_torch=True, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    headers.append(os.path.abspath(__file__))
    jobs = jobs or min(8, os.cpu_count() or 1)
    # objects of sources that no longer exist must not ship to the GPU box
    live = {f[:-4] + ".o" for f in os.listdir(CSRC) if f.endswith(".hip")}
    for f in os.listdir(OBJ):
        if f.endswith(".o") and f not in live:
            os.remove(os.path.join(OBJ, f))
    objs, todo = _compile_all(OBJ, [], jobs, force, verbose, headers)

    lib = os.path.join(PKG, "libsglk.so")
    if force or todo or newer(lib, objs):
        dt, out = run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + objs)
        if verbose:
            print("[build] link  %-32s %5.1fs" % ("libsglk.so", dt), flush=True)
    build_ceiling(force, verbose)
    problems = check_isa(verbose)
    if problems:
        raise RuntimeError("ISA check failed (hand-managed registers are not safe with this compiler output):\n  "
                           + "\n  ".join(problems))

    if with_torch:
        import torch  # noqa: F401  (paths only; no GPU needed)
        from torch.utils import cpp_extension as ce

        ext_src = os.path.join(CSRC, "torch_extension_hip.cc")
        ext = os.path.join(PKG, "common_ops.abi3.so")
        if force or newer(ext, [ext_src, lib] + headers):
            inc = []
            for p in ce.include_paths():
                inc += ["-isystem", p]
            tlib = ce.library_paths()[0]
            cmd = [
                HIPCC, "-x", "c++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
                "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DPy_LIMITED_API=0x03090000",
                "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
                f"-I{INCLUDE}", "-isystem", sysconfig.get_paths()["include"], "-isystem", "/opt/rocm/include",
            ] + inc + [
                ext_src, "-o", ext, f"-L{tlib}", f"-L{PKG}", "-lsglk", "-ltorch", "-ltorch_cpu", "-lc10",
                "-lc10_hip", "-ltorch_hip", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}", "-Wno-unused-value",
            ]
            dt, out = run(cmd)
            if verbose:
                print("[build] torch %-32s %5.1fs" % ("common_ops.abi3.so", dt), flush=True)
                if out.strip():
                    print(out[-2000:])
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--no-torch", action="store_true")
    ap.add_argument("--probes", action="store_true", help="also build build/libsglk_probes.so and build/kbench")
    ap.add_argument("--check", action="store_true", help="only run the ISA check on the last build's assembly")
    a = ap.parse_args()
    if a.check:
        rc = 1 if check_isa() else 0
        check_isa_warnings()
        sys.exit(rc)
    build(a.jobs, a.force, not a.no_torch)
    if a.probes:
        build_probes(a.jobs, a.force)
    print("[build] ok")
