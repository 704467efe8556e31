#!/usr/bin/env python3
"""hipcc build of the gfx950 kernels and the torch operator registry.

Replaces the reference's cmake/SYCL.cmake + src/BuildOnLinux.cmake (icpx, one
.so per SYCL TU, AOT for `bmg`) with a direct hipcc build for gfx950:

  csrc/*.hip                -> build/obj/*.o  -> python/sgl_kernel/libsglk.so        (C-ABI, torch-free)
  csrc/torch_extension_hip.cc                 -> python/sgl_kernel/common_ops.abi3.so (TORCH_LIBRARY)

Both outputs are in-tree (git-ignored) so that they travel to the GPU box.
Incremental: a source is recompiled when it, a header or this script is newer
than its object. Usage: python build.py [--jobs N] [--force] [--no-torch]
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
    "-Wno-implicit-fallthrough", "-Wno-unused-variable",
]


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


def build(jobs=None, force=False, with_torch=True, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    headers.append(os.path.abspath(__file__))
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    jobs = jobs or min(8, os.cpu_count() or 1)

    todo = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        objs.append(obj)
        if force or newer(obj, [src] + headers):
            todo.append((src, obj))
    if todo:
        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            futs = {ex.submit(run, [HIPCC] + HIP_FLAGS + ["-c", src, "-o", obj]): src for src, obj in todo}
            for f in cf.as_completed(futs):
                dt, out = f.result()
                if verbose:
                    print("[build] hipcc %-32s %5.1fs" % (os.path.basename(futs[f]), dt), flush=True)
                    if out.strip():
                        print(out)

    lib = os.path.join(PKG, "libsglk.so")
    if force or todo or newer(lib, objs):
        dt, out = run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + objs)
        if verbose:
            print("[build] link  %-32s %5.1fs" % ("libsglk.so", dt), flush=True)

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
    a = ap.parse_args()
    build(a.jobs, a.force, not a.no_torch)
    print("[build] ok")
