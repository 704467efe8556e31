"""fwd (flash attention) timing at BASELINE configs[2]: bs=16, 32 q heads / 8 kv heads, d=128, seq=4096, paged 64, bf16."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
from sgl_kernel.flash_attn import flash_attn_with_kvcache
dev = "cuda"
bs, hq, hk, d, seq, page = 16, 32, 8, 128, 4096, 64
n_pages = bs * seq // page
kc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
vc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
pt = torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, seq // page)
lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
def timeit(f, warm, it):
    for _ in range(warm): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / it * 1e3
qd = torch.randn(bs, 1, hq, d, device=dev, dtype=torch.bfloat16)
ms = timeit(lambda: flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens, page_table=pt, causal=True), 30, 100)
print(f"decode: {ms:.4f} ms  {(kc.numel()+vc.numel())*2/ms/1e6:.0f} GB/s")
qp = torch.randn(bs * seq, hq, d, device=dev, dtype=torch.bfloat16)
cu = torch.arange(0, bs + 1, device=dev, dtype=torch.int32) * seq
for causal in (True, False):
    ms = timeit(lambda: flash_attn_with_kvcache(qp, kc, vc, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu,
                                                max_seqlen_q=seq, causal=causal), 3, 5)
    fl = 4.0 * bs * hq * d * seq * seq / (2 if causal else 1)
    print(f"prefill causal={causal}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s")
