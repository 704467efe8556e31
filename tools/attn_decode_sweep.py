"""fwd decode at BASELINE configs[2] (bs=16, 32 q heads / 8 kv heads, d=128, seq=4096, paged 64, bf16): time vs num_splits."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
from sgl_kernel.flash_attn import flash_attn_with_kvcache
dev = "cuda"
bs, hq, hk, d, seq, page = 16, 32, 8, 128, 4096, 64
n_pages = bs * seq // page
kc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
vc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
qd = torch.randn(bs, 1, hq, d, device=dev, dtype=torch.bfloat16)
def timeit(f, warm=30, it=100):
    for _ in range(warm): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / it * 1e3
for name, pt in (("random pages", torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, -1)),
                 ("sequential pages", torch.arange(n_pages, device=dev, dtype=torch.int32).view(bs, -1))):
    for splits in (0, 1, 2, 4, 8, 16, 32, 64):
        ms = timeit(lambda: flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens, page_table=pt, num_splits=splits))
        print(f"{name}: num_splits={splits:2d}: {ms:.4f} ms  {(kc.numel()+vc.numel())*2/ms/1e6:.0f} GB/s")
