"""Is fwd decode at d = 64 (0.46 of HBM) bound by the kernel or by the cache layout of the contract? The same number of
(sequence, kv head) pairs, tiles and bytes twice: (a) bs = 16, 8 kv heads - a head's 128-byte row of a token lies at a 1-KB
stride inside [pages, page, Hk, D]; (b) bs = 128, ONE kv head - the head's rows are contiguous. Graph-timed."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
from sgl_kernel.flash_attn import flash_attn_with_kvcache
dev = "cuda"
def timeit(f, warm=30, it=100):
    for _ in range(warm): f()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(it): f()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); g.replay(); en.record(); torch.cuda.synchronize()
        ts.append(st.elapsed_time(en) / it)
    return sorted(ts)[1]
seq, page = 4096, 64
for d, kvdt in ((64, torch.bfloat16), (128, torch.float8_e4m3fn), (128, torch.bfloat16)):
    for bs, hq, hk in ((16, 32, 8), (128, 4, 1)):
        n_pages = bs * seq // page
        kc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16).to(kvdt)
        vc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16).to(kvdt)
        lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
        qd = torch.randn(bs, 1, hq, d, device=dev, dtype=torch.bfloat16)
        kw = dict(k_descale=torch.ones(1, device=dev), v_descale=torch.ones(1, device=dev)) if kvdt != torch.bfloat16 else {}
        pt = torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, -1)
        ms = timeit(lambda: flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens, page_table=pt, num_splits=2, **kw))
        print(f"d={d} {str(kvdt)[6:]}: bs={bs} kv heads={hk}: {ms*1e3:.1f} us  {(kc.numel()+vc.numel())*kc.element_size()/ms/1e6:.0f} GB/s")
        del kc, vc
