"""fused_experts (int4 W4A16) timing at Mixtral-8x7B shapes (BASELINE configs[4]) with per-kernel breakdown hints."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
import sgl_kernel
dev = "cuda"
E, Hd, I, gs, topk = 8, 4096, 14336, 128, 2
if os.environ.get("MOE_SPLITK"):  # diagnostic build only (LD_PRELOAD=.../build/libsglk_probes.so): 0 = the down projection's K split off
    import ctypes
    ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "build", "libsglk_probes.so")
                ).sglk_debug_set_moe_splitk(int(os.environ["MOE_SPLITK"]))
if os.environ.get("MOE_MIN_ROWS"):  # diagnostic build only: average rows per expert from which the tile pipeline uses 256-row blocks
    import ctypes
    ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "build", "libsglk_probes.so")
                ).sglk_debug_set_moe_persist_min_rows(int(os.environ["MOE_MIN_ROWS"]))
w1 = torch.randint(0, 256, (E, 2 * I, Hd // 2), device=dev, dtype=torch.uint8)
w2 = torch.randint(0, 256, (E, Hd, I // 2), device=dev, dtype=torch.uint8)
s1 = torch.rand(E, 2 * I, Hd // gs, device=dev).to(torch.bfloat16) * 0.01
s2 = torch.rand(E, Hd, I // gs, device=dev).to(torch.bfloat16) * 0.01
for T in (int(x) for x in (sys.argv[1:] or ["1", "16", "64", "256", "2048"])):
    x = torch.randn(T, Hd, device=dev, dtype=torch.bfloat16) * 0.1
    logits = torch.randn(T, E, device=dev, dtype=torch.bfloat16)
    tw = torch.empty(T, topk, device=dev, dtype=torch.float32)
    ti = torch.empty(T, topk, device=dev, dtype=torch.int32)
    sgl_kernel.topk_softmax(tw, ti, logits, True)
    f = lambda: sgl_kernel.fused_experts(x, w1, w2, tw, ti, use_int4_w4a16=True, w1_scale=s1, w2_scale=s2)
    for _ in range(30): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50 if T <= 256 else 10
    for _ in range(n): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    print(f"fused_experts T={T}: {ms:.3f} ms  {2.0*T*topk*3*Hd*I/ms/1e9:.1f} TFLOP/s  weights {(w1.numel()+w2.numel())/ms/1e6:.0f} GB/s")

if os.environ.get("MOE_BENCH_INT4_ONLY"):
    sys.exit(0)

# 16-bit weights (moe_grouped_mm_nt_xe20)
del w1, w2, s1, s2
w1b = (torch.randn(E, 2 * I, Hd, device=dev) * 0.02).to(torch.bfloat16)
w2b = (torch.randn(E, Hd, I, device=dev) * 0.02).to(torch.bfloat16)
for T in (64, 2048):
    x = torch.randn(T, Hd, device=dev, dtype=torch.bfloat16) * 0.1
    logits = torch.randn(T, E, device=dev, dtype=torch.bfloat16)
    tw = torch.empty(T, topk, device=dev, dtype=torch.float32)
    ti = torch.empty(T, topk, device=dev, dtype=torch.int32)
    sgl_kernel.topk_softmax(tw, ti, logits, True)
    f = lambda: sgl_kernel.fused_experts(x, w1b, w2b, tw, ti)
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30 if T <= 256 else 10
    for _ in range(n): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    print(f"fused_experts bf16 T={T}: {ms:.3f} ms  {2.0*T*topk*3*Hd*I/ms/1e9:.1f} TFLOP/s  weights {(w1b.numel()+w2b.numel())*2/ms/1e6:.0f} GB/s")

# mxfp4 weights (e2m1 codes, E8M0 scale bytes per 32)
del w1b, w2b
w1m = torch.randint(0, 256, (E, 2 * I, Hd // 2), device=dev, dtype=torch.uint8)
w2m = torch.randint(0, 256, (E, Hd, I // 2), device=dev, dtype=torch.uint8)
s1m = torch.randint(118, 124, (E, 2 * I, Hd // 32), device=dev, dtype=torch.uint8)
s2m = torch.randint(118, 124, (E, Hd, I // 32), device=dev, dtype=torch.uint8)
for T in (1, 64, 2048):
    x = torch.randn(T, Hd, device=dev, dtype=torch.bfloat16) * 0.1
    logits = torch.randn(T, E, device=dev, dtype=torch.bfloat16)
    tw = torch.empty(T, topk, device=dev, dtype=torch.float32)
    ti = torch.empty(T, topk, device=dev, dtype=torch.int32)
    sgl_kernel.topk_softmax(tw, ti, logits, True)
    f = lambda: sgl_kernel.fused_experts(x, w1m, w2m, tw, ti, use_mxfp4_w4a16=True, w1_scale=s1m, w2_scale=s2m)
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30 if T <= 256 else 10
    for _ in range(n): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    print(f"fused_experts mxfp4 T={T}: {ms:.3f} ms  {2.0*T*topk*3*Hd*I/ms/1e9:.1f} TFLOP/s  weights {(w1m.numel()+w2m.numel())/ms/1e6:.0f} GB/s")
