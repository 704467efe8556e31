import os, sys, torch
R = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, os.path.join(R, "sgl-kernel-xpu_amd", "python")); sys.path.insert(0, R)
import sgl_kernel
from oracle import moe as omoe
dev = "cuda"
g = torch.Generator().manual_seed(1)
rows, N, K, gs, dtype = [128] * 8, 512, 1024, 128, torch.bfloat16
E, total = len(rows), sum(rows)
act = (torch.randn(total, K, generator=g) * 0.1).to(dtype)
codes = torch.randint(-8, 8, (E, N, K), generator=g, dtype=torch.int16)
scales = (torch.rand(E, N, K // gs, generator=g) * 0.02 + 0.005).to(dtype)
nib = codes & 0xF
packed = (nib[..., 0::2] | (nib[..., 1::2] << 4)).to(torch.uint8)
rows_t = torch.tensor(rows, dtype=torch.int32, device=dev)
op = torch.ops.sgl_kernel
y = torch.full((total, N), float("nan"), dtype=dtype, device=dev)
ws = torch.full((2, total, N), float("nan"), dtype=torch.float32, device=dev)
used = op.moe_grouped_mm_nt_w4a16_splitk(y, ws, act.to(dev), packed.to(dev), scales.to(dev), None, rows_t, E, True, gs)
print("used", used)
w = omoe.dequant_w4(packed, scales, None, gs).float()  # [E, N, K]
r0 = 0
for h in range(2):
    ref = torch.empty(total, N)
    r0 = 0
    for e, r in enumerate(rows):
        ref[r0:r0 + r] = act[r0:r0 + r, h * K // 2:(h + 1) * K // 2].float() @ w[e][:, h * K // 2:(h + 1) * K // 2].t()
        r0 += r
    d = (ws[h].cpu() - ref).abs()
    print("half", h, "max err", d.max().item(), "mean", d.mean().item(), "ref absmax", ref.abs().max().item())
    bad = d > 0.02
    print("  bad frac", bad.float().mean().item(), "bad rows frac", bad.any(1).float().mean().item(), "bad cols frac", bad.any(0).float().mean().item())
    if bad.any():
        br = bad.any(1).nonzero().flatten(); bc = bad.any(0).nonzero().flatten()
        print("  bad rows", br[:20].tolist(), "...", br[-5:].tolist()); print("  bad cols", bc[:40].tolist(), "...", bc[-5:].tolist())
        # is it the other half / a column shift?
        for h2 in range(2):
            ref2 = torch.empty(total, N); r0 = 0
            for e, r in enumerate(rows):
                ref2[r0:r0 + r] = act[r0:r0 + r, h2 * K // 2:(h2 + 1) * K // 2].float() @ w[e][:, h2 * K // 2:(h2 + 1) * K // 2].t(); r0 += r
            print("  vs half", h2, (ws[h].cpu() - ref2).abs().max().item())
torch.set_printoptions(precision=4, linewidth=200)
ref = torch.empty(total, N); r0 = 0
for e, r in enumerate(rows):
    ref[r0:r0 + r] = act[r0:r0 + r, :K // 2].float() @ w[e][:, :K // 2].t(); r0 += r
print("ws0[0,:36]", ws[0][0, :36].cpu())
print("ref[0,:36]", ref[0, :36])
# find for each of the first 36 columns of row 0 which ref column of row 0 it matches
for c in range(36):
    m = (ref[0] - ws[0][0, c].cpu()).abs()
    print(c, "->", m.argmin().item(), f"{m.min().item():.4f}", end=" | ")
print()
