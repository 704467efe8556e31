"""silu_and_mul / silu_and_mul_clamp / swiglu_gpt_oss_sigmoid_alpha timing on [rows, 2 d] bf16 (default 4096 x 8192 and the
[rows * topk, 2 I] intermediate of a Mixtral-8x7B fused_experts at 2048 tokens): bytes = two reads + one write per output."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
import sgl_kernel
dev = "cuda"


def timeit(f, it=50):
    for _ in range(10): f()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(it): f()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / it


for rows, d in ((4096, 4096), (4096, 14336), (64, 14336)):
    for dt in (torch.bfloat16, torch.float16):
        x = torch.randn(rows, 2 * d, device=dev, dtype=dt)
        o = torch.empty(rows, d, device=dev, dtype=dt)
        gb = 3 * o.numel() * 2 / 1e6
        a = timeit(lambda: sgl_kernel.silu_and_mul(x, out=o))
        b = timeit(lambda: sgl_kernel.silu_and_mul_clamp(x, o, 10.0))
        c = timeit(lambda: sgl_kernel.swiglu_gpt_oss_sigmoid_alpha(x, 1.702, 7.0))
        print(f"[{rows}, 2 x {d}] {str(dt)[6:]}: silu_and_mul {a*1e3:.1f} us {gb/a:.0f} GB/s | clamp {b*1e3:.1f} us {gb/b:.0f} GB/s | "
              f"gpt-oss {c*1e3:.1f} us {gb/c:.0f} GB/s")
