"""Board power and shader clock while the headline GEMM runs back to back for a few seconds, for three operand
distributions (the clock the part holds under fp8 MFMA load depends on the operand bits: MI355X_MICROARCH.md, DVFS).
Usage (GPU box): python tools/power_probe.py"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sgl-kernel-xpu_amd", "python")]
import torch
import sgl_kernel

M, N, K = 4096, 14336, 4096
FP8 = torch.float8_e4m3fn
dev = torch.device("cuda:0")


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            out.append(r.strip().splitlines()[-1])
        except Exception as e:  # noqa
            out.append("err %s" % e)
        time.sleep(0.3)


def run(name, a, b):
    sa = (torch.rand(K // 128, M, device=dev) * 1e-3 + 1e-4).t()
    sb = (torch.rand(N // 128, K // 128, device=dev) * 1e-3 + 1e-4).t()
    f = lambda: sgl_kernel.fp8_blockwise_scaled_mm(a, b, sa, sb, torch.bfloat16)
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(200):
            f()
        torch.cuda.synchronize()
        n += 200
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    print("%-14s %.1f us per GEMM (%d launches back to back)" % (name, dt / n * 1e6, n))
    for ln in out[1:-1][:6]:
        print("    ", ln[:200])


g = torch.Generator(device="cpu").manual_seed(1)
x = ((torch.rand(M, K, generator=g) - 0.5) * 2 * 448).to(FP8).to(dev)
w = ((torch.rand(N, K, generator=g) - 0.5) * 2 * 448).to(FP8).to(dev).t()
run("uniform e4m3", x, w)
xb = torch.randint(0, 255, (M, K), dtype=torch.uint8, generator=g)
wb = torch.randint(0, 255, (N, K), dtype=torch.uint8, generator=g)
xb[(xb & 0x7F) == 0x7F] ^= 1
wb[(wb & 0x7F) == 0x7F] ^= 1
run("random bytes", xb.view(FP8).to(dev), wb.view(FP8).to(dev).t())
run("zeros", torch.zeros(M, K, device=dev).to(FP8), torch.zeros(N, K, device=dev).to(FP8).t())
