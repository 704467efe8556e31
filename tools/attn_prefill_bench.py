"""fwd (flash attention) PREFILL legs only, for profiling attn_prefill_kernel (BASELINE configs[2]: bs=16, 32 q / 8 kv heads,
seq=4096, paged 64, bf16): causal d=128, causal d=64, q=128 chunk prefill over 4096 keys."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
from sgl_kernel.flash_attn import flash_attn_with_kvcache
dev = "cuda"
if os.environ.get("ATTN_PREFILL_PROBE"):  # diagnostic build only: LD_PRELOAD=sgl-kernel-xpu_amd/build/libsglk_probes.so
    import ctypes
    ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "build", "libsglk_probes.so")
                ).sglk_debug_set_attn_prefill_probe(int(os.environ["ATTN_PREFILL_PROBE"]))
if os.environ.get("ATTN_PREFILL_WAVES"):
    import ctypes
    ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "build", "libsglk_probes.so")
                ).sglk_debug_set_attn_prefill_waves(int(os.environ["ATTN_PREFILL_WAVES"]))
ONLY = os.environ.get("ATTN_PREFILL_ONLY")  # "128" / "64" / "256" / "chunk" / "softcap": that leg alone (one rocprofv3 run per shape)
bs, hq, hk, seq, page = 16, 32, 8, 4096, 64
n_pages = bs * seq // page
pt = torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, seq // page)
lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
def timeit(f, warm, it):
    for _ in range(warm): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / it * 1e3
for d in ((128,) if ONLY in ("128", "chunk", "softcap", "fp8") else (64,) if ONLY == "64" else (256,) if ONLY == "256" else
          (96, 192) if ONLY == "96" else (128, 64, 256)):
    hq = 16 if d >= 192 else 32  # (d = 192 / 256: 16 q heads / 8 kv heads)
    kc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
    vc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
    qp = torch.randn(bs * seq, hq, d, device=dev, dtype=torch.bfloat16)
    cu = torch.arange(0, bs + 1, device=dev, dtype=torch.int32) * seq
    if ONLY not in ("chunk", "softcap", "fp8"):
        ms = timeit(lambda: flash_attn_with_kvcache(qp, kc, vc, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu,
                                                    max_seqlen_q=seq, causal=True), 5, 10)
        print(f"prefill causal d={d}: {ms:.3f} ms  {4.0 * bs * hq * d * seq * seq / 2 / ms / 1e9:.1f} TFLOP/s")
    if d == 128 and ONLY in (None, "softcap"):
        ms = timeit(lambda: flash_attn_with_kvcache(qp, kc, vc, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu,
                                                    max_seqlen_q=seq, causal=True, softcap=50.0), 5, 10)
        print(f"prefill causal d={d} softcap 50: {ms:.3f} ms  {4.0 * bs * hq * d * seq * seq / 2 / ms / 1e9:.1f} TFLOP/s")
    if d == 128 and ONLY in (None, "fp8"):
        k8, v8 = kc.to(torch.float8_e4m3fn), vc.to(torch.float8_e4m3fn)
        one = torch.ones(1, device=dev)
        ms = timeit(lambda: flash_attn_with_kvcache(qp, k8, v8, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu,
                                                    max_seqlen_q=seq, causal=True, k_descale=one, v_descale=one), 5, 10)
        print(f"prefill causal d={d} fp8 e4m3 KV: {ms:.3f} ms  {4.0 * bs * hq * d * seq * seq / 2 / ms / 1e9:.1f} TFLOP/s")
        del k8, v8
    if d == 128 and ONLY in (None, "chunk"):
        qc = torch.randn(bs * 128, hq, d, device=dev, dtype=torch.bfloat16)
        cuc = torch.arange(0, bs + 1, device=dev, dtype=torch.int32) * 128
        ms = timeit(lambda: flash_attn_with_kvcache(qc, kc, vc, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cuc,
                                                    max_seqlen_q=128, causal=True), 10, 30)
        print(f"chunk prefill q=128 d=128: {ms:.4f} ms  {4.0 * bs * hq * d * (128 * (seq - 128) + 128 * 129 / 2) / ms / 1e9:.1f} TFLOP/s")
