"""Compare two main-loop variants of fp8_blockwise_scaled_mm on the GPU and print a per-tile mismatch map."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
import sgl_kernel  # noqa: E402

lib = ctypes.CDLL(os.path.join(os.path.dirname(sgl_kernel.__file__), "libsglk.so"))
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 14336, 4096)
va, vb = (int(x) for x in sys.argv[4:6]) if len(sys.argv) > 5 else (4, 1)
torch.manual_seed(0)
dev = "cuda"
a = (torch.randn(M, K, device=dev) * 2).to(torch.float8_e4m3fn)
b = (torch.randn(N, K, device=dev) * 2).to(torch.float8_e4m3fn).t()
sa = torch.rand(M, K // 128, device=dev) + 0.5
sb = torch.rand(K // 128, (N + 127) // 128, device=dev) + 0.5
outs = []
for v in (va, vb):
    lib.sglk_debug_set_gemm_variant(v)
    outs.append(sgl_kernel.fp8_blockwise_scaled_mm(a, b, sa, sb, torch.bfloat16).float())
torch.cuda.synchronize()
d = (outs[0] - outs[1]).abs() > 1e-2 * outs[1].abs() + 1e-3
print("mismatching elements:", int(d.sum()), "of", d.numel())
tm, tn = (M + 255) // 256, (N + 255) // 256
pad = torch.zeros(tm * 256, tn * 256, dtype=torch.bool, device=dev)
pad[:M, :N] = d
grid = pad.view(tm, 256, tn, 256).any(dim=3).any(dim=1).cpu()
for i in range(tm):
    print("".join("X" if grid[i, jn] else "." for jn in range(tn)))
bad = d.nonzero()
if len(bad):
    r, c = bad[0].tolist()
    t = pad.view(tm, 256, tn, 256)[r // 256, :, c // 256, :]
    print("first bad tile", r // 256, c // 256, "bad rows in tile:", t.any(dim=1).nonzero().flatten().tolist()[:40])
    print("bad cols in tile:", t.any(dim=0).nonzero().flatten().tolist()[:70])
    print("values", outs[0][r, c].item(), outs[1][r, c].item())
