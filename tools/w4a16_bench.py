"""moe_grouped_mm_nt_xe20_w4a16 alone at the two Mixtral-8x7B GEMM shapes (BASELINE configs[4]: E=8, hidden 4096, inter 14336,
int4 group 128): weight-streaming rate per rows-per-expert pattern. Usage: w4a16_bench.py [rows ...] (default 1 4 16 32 64)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
import sgl_kernel  # noqa: F401  (loads the op library)
dev = "cuda"
E, H, I, gs = 8, 4096, 14336, 128
op = torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20_w4a16


def timeit(f, it=50):
    for _ in range(20): f()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(it): f()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / it


patterns = [int(a) for a in sys.argv[1:]] or [1, 4, 16, 32, 64]
for name, N, K in (("gemm1 N=28672 K=4096", 2 * I, H), ("gemm2 N=4096 K=14336", H, I)):
    w = torch.randint(0, 256, (E, N, K // 2), device=dev, dtype=torch.uint8)
    s = (torch.rand(E, N, K // gs, device=dev) * 0.01).to(torch.bfloat16)
    for r in patterns:
        for skew in (False, True):
            rows = [r] * E
            if skew:  # same total, ragged (binomial-like spread around the mean)
                rows = [max(0, r + d) for d in (-r // 2, r // 2, -r // 4, r // 4, 0, 0, -r // 3, r // 3)]
            total = sum(rows)
            if total == 0: continue
            a = torch.randn(total, K, device=dev, dtype=torch.bfloat16) * 0.1
            out = torch.empty(total, N, device=dev, dtype=torch.bfloat16)
            rt = torch.tensor(rows, device=dev, dtype=torch.int32)
            ms = timeit(lambda: op(out, a, w, s, None, None, rt, E, True, gs))
            print(f"{name} rows={rows}: {ms*1e3:.1f} us  weights {w.numel()/ms/1e6:.0f} GB/s  {2.0*total*N*K/ms/1e9:.1f} TFLOP/s")
    del w, s
