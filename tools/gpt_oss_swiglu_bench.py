"""GEMM 1 of fused_experts with the gpt-oss swiglu (gate / up rows interleaved): the reference's W4A16 route - plain grouped GEMM to
[rows, 2I], then swiglu_gpt_oss_sigmoid_alpha - against the swiglu in the GEMM's epilogue (activation 5 of the authored op), and the
same for 16-bit weights (moe_grouped_mm_nt_xe20 with activation_type 2: fuse_act False + the op, against fuse_act True).
Mixtral-sized experts (E = 8, H = 4096, I = 14336), uniform rows."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "python"))
import sgl_kernel  # noqa
dev = "cuda"
E, Hd, I, gs, topk = 8, 4096, 14336, 128, 2
op = torch.ops.sgl_kernel


def timeit(f, it=20):
    """device time per call: `it` calls in one HIP graph, median of three replays"""
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
for T in (int(a) for a in (sys.argv[1:] or ["64", "512", "2048", "4096"])):
    total = T * topk
    rows = torch.full((E,), total // E, dtype=torch.int32, device=dev)
    x = torch.randn(total, Hd, device=dev, dtype=torch.bfloat16) * 0.1
    gu = torch.empty(total, 2 * I, device=dev, dtype=torch.bfloat16)
    h = torch.empty(total, I, device=dev, dtype=torch.bfloat16)

    def two():
        op.moe_grouped_mm_nt_xe20_w4a16(gu, x, w1, s1, None, None, rows, E, True, gs)
        op.swiglu_gpt_oss_sigmoid_alpha(gu, 1.702, 7.0)
    t2 = timeit(two)
    t1 = timeit(lambda: op.moe_grouped_mm_nt_w4a16_act(h, x, w1, s1, None, None, rows, E, True, gs, 5, 7.0, None, 1.702))
    print(f"int4 T={T}: GEMM 1 + swiglu op {t2:.0f} us, swiglu in the epilogue {t1:.0f} us")
del w1, s1
wb = (torch.randn(E, 2 * I, Hd, device=dev) * 0.02).to(torch.bfloat16)
for T in (int(a) for a in (sys.argv[1:] or ["64", "512", "2048", "4096"])):
    total = T * topk
    rows = torch.full((E,), total // E, dtype=torch.int32, device=dev)
    x = torch.randn(total, Hd, device=dev, dtype=torch.bfloat16) * 0.1
    gu = torch.empty(total, 2 * I, device=dev, dtype=torch.bfloat16)
    h = torch.empty(total, I, device=dev, dtype=torch.bfloat16)

    def two():
        op.moe_grouped_mm_nt_xe20(gu, x, wb, None, rows, E, 2, False, 1.702, 7.0)
        op.swiglu_gpt_oss_sigmoid_alpha(gu, 1.702, 7.0)
    t2 = timeit(two)
    t1 = timeit(lambda: op.moe_grouped_mm_nt_xe20(h, x, wb, None, rows, E, 2, True, 1.702, 7.0))
    print(f"bf16 T={T}: GEMM 1 + swiglu op {t2:.0f} us, swiglu in the epilogue {t1:.0f} us")
