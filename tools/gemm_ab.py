"""fp8_blockwise_scaled_mm at the headline shape (4096, 14336, 4096), 20 calls in one HIP graph, for A / B runs of two builds of the
library in one gpurun call (LD_PRELOAD=<other libsglk.so> selects the build): prints the median of 9 replays."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
import sgl_kernel
dev = "cuda"
M, N, K = 4096, 14336, 4096
FP8 = torch.float8_e4m3fn
g = torch.Generator(device="cpu").manual_seed(0)
a = ((torch.rand(M, K, generator=g) - 0.5) * 2 * 448).to(FP8).to(dev)
b = ((torch.rand(N, K, generator=g) - 0.5) * 2 * 448).to(FP8).to(dev).t()
sa = (torch.rand(M, K // 128, generator=g) + 0.5).to(dev)
sb = (torch.rand(K // 128, N // 128, generator=g) + 0.5).to(dev)
f = lambda: sgl_kernel.fp8_blockwise_scaled_mm(a, b, sa, sb, torch.bfloat16)
import time
t0 = time.time()
while time.time() - t0 < 0.3: f()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(20): f()
ms = []
for _ in range(9):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1) / 20)
m = statistics.median(ms)
print(f"fp8_blockwise (4096, 14336, 4096): {m:.4f} ms  {2.0 * M * N * K / m / 1e9:.0f} TFLOP/s   min {min(ms):.4f}")
