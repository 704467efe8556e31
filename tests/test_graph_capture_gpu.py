"""HIP-graph capture of one decode step through the operator library.

The reference's contract for these ops is "enqueue on the current stream, no host sync, no device -> host copy"
(src/sycl/flash_attention.cpp:426-430, :1426-1429): exactly what stream capture requires. This test
captures  rmsnorm -> per-token-group quant -> fp8 block-scaled GEMM -> fwd decode (split-KV) -> flash_mla_decode (split)
-> topk_softmax -> fused_experts (int4 W4A16)  into one torch.cuda.CUDAGraph, replays it with changed inputs and compares
every replay with the eager result of the same inputs (bit for bit: the kernels are deterministic)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

FP8 = torch.float8_e4m3fn


def _int4_weights(E, N, K, gs, dtype, g):
    codes = torch.randint(0, 256, (E, N, K // 2), generator=g, dtype=torch.uint8)
    scales = (torch.rand(E, N, K // gs, generator=g) * 0.02 + 0.005).to(dtype)
    return codes, scales


def test_decode_step_graph_capture(sglk, dev):
    g = torch.Generator().manual_seed(7)
    dt = torch.bfloat16
    B, hidden, ffn = 8, 1024, 1536
    # attention geometry (fwd decode, paged) and MLA geometry
    Hq, Hk, D, page, S = 16, 4, 128, 64, 1024
    Hm, Sm = 128, 512
    E, topk, inter = 8, 2, 512

    x = torch.randn(B, hidden, generator=g).to(dt).to(dev)
    w_norm = torch.randn(hidden, generator=g).to(dt).to(dev)
    wb = ((torch.rand(ffn, hidden, generator=g) - 0.5) * 2 * 448).to(FP8).to(dev)
    sb = (torch.rand(ffn // 128, hidden // 128, generator=g) * 0.01 + 0.001).to(dev)
    # fwd decode inputs
    q = torch.randn(B, 1, Hq, D, generator=g).to(dt).to(dev)
    n_pages = B * S // page
    kc = torch.randn(n_pages, page, Hk, D, generator=g).to(dt).to(dev)
    vc = torch.randn(n_pages, page, Hk, D, generator=g).to(dt).to(dev)
    table = torch.randperm(n_pages, generator=g).view(B, -1).to(torch.int32).to(dev)
    lens = torch.randint(S // 2, S + 1, (B,), generator=g).to(torch.int32).to(dev)
    # MLA decode inputs
    q_nope = torch.randn(B, Hm, 512, generator=g).to(dt).to(dev)
    q_pe = torch.randn(B, Hm, 64, generator=g).to(dt).to(dev)
    m_pages = B * Sm // page
    mcache = torch.randn(m_pages, page, 576, generator=g).to(dt).to(dev)
    mtable = torch.randperm(m_pages, generator=g).view(B, -1).to(torch.int32).to(dev)
    mlens = torch.randint(Sm // 2, Sm + 1, (B,), generator=g).to(torch.int32).to(dev)
    ws = torch.empty(sglk.flash_mla_get_workspace_size(Sm, B, Hm, page, 4), dtype=torch.uint8, device=dev)
    # MoE inputs
    gate = torch.randn(B, E, generator=g).to(dev)
    w1, s1 = _int4_weights(E, 2 * inter, hidden, 128, dt, g)
    w2, s2 = _int4_weights(E, hidden, inter, 128, dt, g)
    w1, s1, w2, s2 = w1.to(dev), s1.to(dev), w2.to(dev), s2.to(dev)

    def step():
        y = sglk.rmsnorm(x, w_norm, 1e-6)
        qa = torch.empty(B, hidden, dtype=FP8, device=dev)
        sa = torch.empty(hidden // 128, B, dtype=torch.float32, device=dev).t()
        sglk.sgl_per_token_group_quant_8bit(y, qa, sa, 128, 1e-10, -448.0, 448.0, False, enable_v2=False)
        h = sglk.fp8_blockwise_scaled_mm(qa, wb.t(), sa, sb.t(), dt)
        attn = sglk.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, page_table=table, num_splits=4)
        mla = sglk.flash_mla_decode(q_nope, q_pe, mcache, mlens, mtable, ws, 576 ** -0.5, 4)
        tw = torch.empty(B, topk, dtype=torch.float32, device=dev)
        ti = torch.empty(B, topk, dtype=torch.int32, device=dev)
        sglk.topk_softmax(tw, ti, gate, True)
        moe = sglk.fused_experts(y, w1, w2, tw, ti, use_int4_w4a16=True, w1_scale=s1, w2_scale=s2)
        return h, attn, mla, tw, ti, moe

    # warm-up on a side stream (allocations, lazily set kernel attributes), then capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = step()
    statics = (x, q, q_nope, q_pe, gate, lens, mlens)
    for it in range(3):
        gi = torch.Generator().manual_seed(100 + it)
        with torch.no_grad():
            x.copy_(torch.randn(B, hidden, generator=gi).to(dt))
            q.copy_(torch.randn(B, 1, Hq, D, generator=gi).to(dt))
            q_nope.copy_(torch.randn(B, Hm, 512, generator=gi).to(dt))
            q_pe.copy_(torch.randn(B, Hm, 64, generator=gi).to(dt))
            gate.copy_(torch.randn(B, E, generator=gi))
            lens.copy_(torch.randint(S // 2, S + 1, (B,), generator=gi).to(torch.int32))
            mlens.copy_(torch.randint(Sm // 2, Sm + 1, (B,), generator=gi).to(torch.int32))
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in captured]
        want = step()
        torch.cuda.synchronize()
        names = ("fp8 gemm", "fwd decode", "flash_mla_decode", "topk weights", "topk ids", "fused_experts")
        for name, a, b in zip(names, got, want):
            assert torch.isfinite(a.float()).all(), name
            assert torch.equal(a, b), f"replay {it}: {name} differs from eager"
    del statics
