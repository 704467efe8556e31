"""Diagnostic build only (LD_PRELOAD=.../build/libsglk_probes.so): fp8_blockwise_scaled_mm at the headline shape under a probe variant
against the default variant - outputs must agree to fp32 summation order."""
import ctypes, os, sys
import torch
R = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, os.path.join(R, "sgl-kernel-xpu_amd", "python"))
import sgl_kernel
lib = ctypes.CDLL(os.path.join(R, "sgl-kernel-xpu_amd", "build", "libsglk_probes.so"))
dev = "cuda"
M, N, K = 4096, 14336, 4096
FP8 = torch.float8_e4m3fn
g = torch.Generator(device="cpu").manual_seed(0)
a = ((torch.rand(M, K, generator=g) - 0.5) * 2 * 448).to(FP8).to(dev)
b = ((torch.rand(N, K, generator=g) - 0.5) * 2 * 448).to(FP8).to(dev).t()
sa = (torch.rand(M, K // 128, generator=g) + 0.5).to(dev)
sb = (torch.rand(K // 128, N // 128, generator=g) + 0.5).to(dev)
ref = sgl_kernel.fp8_blockwise_scaled_mm(a, b, sa, sb, torch.bfloat16).float()
for v in [int(x) for x in sys.argv[1:]]:
    lib.sglk_debug_set_gemm_variant(v)
    o = sgl_kernel.fp8_blockwise_scaled_mm(a, b, sa, sb, torch.bfloat16).float()
    lib.sglk_debug_set_gemm_variant(4)
    d = (o - ref).abs()
    print(f"variant {v}: max |diff| {d.max().item():.4g} of max |ref| {ref.abs().max().item():.4g}; elements differing {(d > 0).float().mean().item():.4f}; "
          f"beyond one bf16 ulp {(d > ref.abs() * 2 ** -7).float().mean().item():.6f}")
