"""Row-count sweeps ACROSS the dispatch boundaries of the kernel families (graph-timed): the 8-bit GEMMs at a Llama-3-8B FFN
projection and its transpose, fwd attention by query tokens per sequence. Looks for cliffs between neighbouring regimes."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
import sgl_kernel
from sgl_kernel.flash_attn import flash_attn_with_kvcache
dev = "cuda"
FP8 = torch.float8_e4m3fn


def timeit(f, it=20):
    for _ in range(5): f()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(it): f()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); g.replay(); en.record(); torch.cuda.synchronize()
        ts.append(st.elapsed_time(en) / it * 1e3)
    return sorted(ts)[1]


what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
if what == "gemm":
    Ms = [1, 16, 64, 65, 96, 128, 129, 192, 256, 257, 384, 512, 513, 768, 1024, 2048]
    for N, K in ((14336, 4096), (4096, 14336)):
        g = torch.Generator().manual_seed(0)
        b = ((torch.rand(N, K, generator=g) - 0.5) * 2 * 448).to(FP8).to(dev).t()
        sb = (torch.rand(K // 128, N // 128, generator=g) + 0.5).to(dev)
        sbc = (torch.rand(N, generator=g) * 0.01).to(dev)
        bi = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(dev).t()
        for M in Ms:
            a = ((torch.rand(M, K, generator=g) - 0.5) * 2 * 448).to(FP8).to(dev)
            sa = (torch.rand(M, K // 128, generator=g) + 0.5).to(dev)
            sar = (torch.rand(M, generator=g) * 0.01).to(dev)
            ai = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).to(dev)
            t1 = timeit(lambda: sgl_kernel.fp8_blockwise_scaled_mm(a, b, sa, sb, torch.bfloat16))
            t2 = timeit(lambda: sgl_kernel.fp8_scaled_mm(a, b, sar, sbc, torch.bfloat16, None))
            t3 = timeit(lambda: sgl_kernel.int8_scaled_mm(ai, bi, sar, sbc, torch.bfloat16, None))
            print(f"N={N} K={K} M={M}: fp8 blockwise {t1:.1f} us | fp8_scaled_mm {t2:.1f} us | int8_scaled_mm {t3:.1f} us")
else:
    bs, hq, hk, seq, page, d = 16, 32, 8, 4096, 64, 128
    n_pages = bs * seq // page
    kc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
    vc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
    pt = torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, seq // page)
    lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
    for q in (1, 2, 4, 5, 8, 16, 31, 32, 33, 64, 128, 256):
        qq = torch.randn(bs * q, hq, d, device=dev, dtype=torch.bfloat16)
        cu = torch.arange(0, bs + 1, device=dev, dtype=torch.int32) * q
        t = timeit(lambda: flash_attn_with_kvcache(qq, kc, vc, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu,
                                                   max_seqlen_q=q, causal=True))
        by = 2.0 * bs * seq * hk * d * 2
        fl = 4.0 * bs * hq * d * q * seq
        print(f"fwd bs16 seq4096 h32/8 d128, {q} query tokens per sequence ({q * hq // hk} packed rows): {t:.1f} us  {by / t / 1e6:.0f} GB/s  {fl / t / 1e6:.0f} TFLOP/s")
