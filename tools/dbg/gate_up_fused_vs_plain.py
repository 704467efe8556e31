"""int4 gate / up GEMM of fused_experts at decode sizes: activation in the epilogue (with the gather) against the plain GEMM (with the
scatter copy) + silu_and_mul - graph-timed, Mixtral-sized experts, routed rows."""
import os, sys
import torch
R = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, os.path.join(R, "sgl-kernel-xpu_amd", "python"))
import sgl_kernel  # noqa
dev = "cuda"
E, Hd, I, gs, topk = 8, 4096, 14336, 128, 2
op = torch.ops.sgl_kernel


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


w1 = torch.randint(0, 256, (E, 2 * I, Hd // 2), device=dev, dtype=torch.uint8)
s1 = torch.rand(E, 2 * I, Hd // gs, device=dev).to(torch.bfloat16) * 0.01
for T in (int(a) for a in (sys.argv[1:] or ["1", "8", "16", "32", "64", "128"])):
    total = T * topk
    ti = torch.randn(T, E, device=dev).topk(topk, dim=-1).indices.int()
    rows = torch.bincount(ti.flatten(), minlength=E).to(torch.int32)
    tok = torch.randn(T, Hd, device=dev, dtype=torch.bfloat16) * 0.1
    amap = torch.randint(0, T, (total,), device=dev, dtype=torch.int32)
    cmap = torch.randperm(total, device=dev).to(torch.int32)
    x = torch.empty(total, Hd, device=dev, dtype=torch.bfloat16)
    gu = torch.empty(total, 2 * I, device=dev, dtype=torch.bfloat16)
    h = torch.empty(total, I, device=dev, dtype=torch.bfloat16)
    t_f = timeit(lambda: op.moe_grouped_mm_nt_w4a16_act(h, tok, w1, s1, None, None, rows, E, True, gs, 1, 0.0, amap))

    def plain():
        op.scatter_tokens_to_experts(tok, cmap, x)
        op.moe_grouped_mm_nt_xe20_w4a16(gu, x, w1, s1, None, None, rows, E, True, gs)
        op.silu_and_mul(h, gu)
    t_p = timeit(plain)
    t_g = timeit(lambda: op.moe_grouped_mm_nt_xe20_w4a16(gu, x, w1, s1, None, None, rows, E, True, gs))
    print(f"T={T} rows={rows.tolist()}: fused act + gather {t_f:.1f} us | scatter + plain GEMM + silu_and_mul {t_p:.1f} us (the plain GEMM alone {t_g:.1f})")
