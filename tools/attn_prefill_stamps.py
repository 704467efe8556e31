"""Where attn_prefill_kernel's tile loop spends its cycles (diagnostic build only: LD_PRELOAD=sgl-kernel-xpu_amd/build/libsglk_probes.so).
In-kernel s_memtime stamps around the loop's segments, summed per wave; prints the shares per segment for the causal 16 x 4096
prefill of BASELINE configs[2] at d = 128 and d = 64. Read the SHARES, not the run time: the stamps' fences forbid overlaps."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
from sgl_kernel.flash_attn import flash_attn_with_kvcache
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "build", "libsglk_probes.so"))
lib.sglk_debug_set_attn_prefill_stamps.argtypes = [ctypes.c_void_p]
dev = "cuda"
bs, hq, hk, seq, page = 16, 32, 8, 4096, 64
n_pages = bs * seq // page
pt = torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, seq // page)
lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
names = ["wait + barrier", "QK (16 / 8 MFMAs + K reads)", "DMA issue + page ids", "softmax", "PV (16 / 8 MFMAs + V reads)"]
for d in (128, 64):
    kc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
    vc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
    q = torch.randn(bs * seq, hq, d, device=dev, dtype=torch.bfloat16)
    cu = torch.arange(0, bs + 1, device=dev, dtype=torch.int32) * seq
    run = lambda: flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu, max_seqlen_q=seq, causal=True)
    for _ in range(3): run()
    n_wg = bs * hk * (seq * (hq // hk) // 128)
    buf = torch.zeros(n_wg * 4 * 8, dtype=torch.int64, device=dev)
    lib.sglk_debug_set_attn_prefill_stamps(ctypes.c_void_p(buf.data_ptr()))
    run()
    torch.cuda.synchronize()
    lib.sglk_debug_set_attn_prefill_stamps(ctypes.c_void_p(0))
    s = buf.view(-1, 8).cpu().double()
    s = s[s[:, 5] > 0]
    tiles = s[:, 5].sum()
    tot = s[:, :5].sum()
    print(f"d={d}: {len(s)} waves, {int(tiles)} wave-tiles, {tot / tiles:.0f} stamped cycles per wave-tile (100 MHz ticks x clock ratio if s_memtime is the constant clock)")
    for k, nm in enumerate(names):
        print(f"   {nm:32s} {s[:, k].sum() / tiles:8.1f} per wave-tile  {100 * s[:, k].sum() / tot:5.1f} %")
    long = s[s[:, 5] >= 48]
    if len(long):
        t2 = long[:, 5].sum()
        print("   waves with >= 48 tiles:", " ".join(f"{long[:, k].sum() / t2:.0f}" for k in range(5)))
