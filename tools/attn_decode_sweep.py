"""fwd decode at BASELINE configs[2] geometry (bs=16, 32 q heads / 8 kv heads, seq=4096, paged 64): time vs num_splits for
the head dims / cache types of the decode kernel (d = 64 / 128 / 256 bf16, d = 128 fp8 e4m3)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
from sgl_kernel.flash_attn import flash_attn_with_kvcache
dev = "cuda"
bs, hq, hk, seq, page = 16, 32, 8, 4096, 64
def timeit(f, warm=30, it=100):
    """device time per call: `it` calls captured in one HIP graph, the median of three replays (an eager loop of 40-us calls
    times the host)"""
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
if os.environ.get("ATTN_WAVES"):  # diagnostic build only: LD_PRELOAD=sgl-kernel-xpu_amd/build/libsglk_probes.so ATTN_WAVES=4|8
    import ctypes
    ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "build", "libsglk_probes.so")
                ).sglk_debug_set_attn_decode_waves(int(os.environ["ATTN_WAVES"]))
cases = [(int(a.split(":")[0]), a.split(":")[1]) for a in sys.argv[1:]] or [(128, "bf16"), (64, "bf16"), (256, "bf16"), (128, "fp8")]
for d, kv in cases:
    n_pages = bs * seq // page
    kvdt = torch.float8_e4m3fn if kv == "fp8" else torch.bfloat16
    kc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16).to(kvdt)
    vc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16).to(kvdt)
    lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
    qd = torch.randn(bs, 1, hq, d, device=dev, dtype=torch.bfloat16)
    kw = dict(k_descale=torch.ones(1, device=dev), v_descale=torch.ones(1, device=dev)) if kv == "fp8" else {}
    pt = torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, -1)
    for splits in [int(x) for x in os.environ.get("SPLITS", "0,1,2,3,4,6,8,16").split(",")]:
        ms = timeit(lambda: flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens, page_table=pt, num_splits=splits, **kw))
        print(f"d={d} {kv}: num_splits={splits:2d}: {ms:.4f} ms  {(kc.numel()+vc.numel())*kc.element_size()/ms/1e6:.0f} GB/s")
