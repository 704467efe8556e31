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
    if os.environ.get("GEMM_SPLITK"):  # diagnostic build only (LD_PRELOAD=.../build/libsglk_probes.so): forced slice count (0: never)
        import ctypes
        ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "build", "libsglk_probes.so")
                    ).sglk_debug_set_gemm_splitk(int(os.environ["GEMM_SPLITK"]))
    Ms = [int(m) for m in os.environ.get("GEMM_MS", "1,16,64,65,96,128,129,192,256,257,384,512,513,768,1024,2048").split(",")]
    for N, K in ((14336, 4096), (4096, 14336), (4096, 4096), (6144, 4096)):
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
if what == "moe":
    E, Hd, I, gs, topk = 8, 4096, 14336, 128, 2
    if os.environ.get("MOE_SHAPE"):  # "experts,hidden,inter,topk" - e.g. 128,2048,768,8 (Qwen3-30B-A3B) or 256,7168,2048,8 (DeepSeek-V3)
        E, Hd, I, topk = (int(v) for v in os.environ["MOE_SHAPE"].split(","))
    w1 = torch.randint(0, 256, (E, 2 * I, Hd // 2), device=dev, dtype=torch.uint8)
    w2 = torch.randint(0, 256, (E, Hd, I // 2), device=dev, dtype=torch.uint8)
    s1 = torch.rand(E, 2 * I, Hd // gs, device=dev).to(torch.bfloat16) * 0.01
    s2 = torch.rand(E, Hd, I // gs, device=dev).to(torch.bfloat16) * 0.01
    for T in [int(t) for t in os.environ.get("MOE_TS", "1,2,4,8,16,24,32,40,48,64,80,96,128,160,192,256,320,383,384,448,512,640,767,768,1024,1280,1536,2048").split(",")]:
        x = torch.randn(T, Hd, device=dev, dtype=torch.bfloat16) * 0.1
        logits = torch.randn(T, E, device=dev, dtype=torch.bfloat16)
        tw = torch.empty(T, topk, device=dev, dtype=torch.float32)
        ti = torch.empty(T, topk, device=dev, dtype=torch.int32)
        sgl_kernel.topk_softmax(tw, ti, logits, True)
        t = timeit(lambda: sgl_kernel.fused_experts(x, w1, w2, tw, ti, use_int4_w4a16=True, w1_scale=s1, w2_scale=s2), it=10)
        wbytes = (w1.numel() + w2.numel()) * min(1.0, T * topk / E)
        print(f"fused_experts int4 E={E} hidden={Hd} inter={I} top{topk} T={T}: {t:.0f} us  ({t / T:.2f} us per token)  {2.0 * T * topk * 3 * Hd * I / t / 1e6:.0f} TFLOP/s  ~{wbytes / t / 1e6:.2f} TB/s of weights")
if what == "mla":
    from sgl_kernel.attention import flash_mla_decode, flash_mla_get_workspace_size
    page = 64
    for H in (8, 16, 32, 64, 65, 96, 128):
        for bs, seq in ((1, 8192), (4, 8192), (16, 8192), (32, 8192), (64, 8192), (128, 8192), (128, 1024), (128, 2048), (256, 2048)):
            n_pages = bs * seq // page
            cache = torch.randn(n_pages, page, 576, device=dev, dtype=torch.bfloat16)
            qn = torch.randn(bs, H, 512, device=dev, dtype=torch.bfloat16)
            qp = torch.randn(bs, H, 64, device=dev, dtype=torch.bfloat16)
            lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
            table = torch.arange(n_pages, device=dev, dtype=torch.int32).view(bs, -1)
            ws = torch.empty(flash_mla_get_workspace_size(seq, bs, H, page, -1), device=dev, dtype=torch.uint8)
            t = timeit(lambda: flash_mla_decode(qn, qp, cache, lens, table, ws, 0.1, -1), it=10)
            by = bs * seq * 576 * 2
            print(f"flash_mla_decode H={H} bs={bs} seq={seq}: {t:.1f} us  {by / t / 1e6:.2f} TB/s")
            del cache
if what == "fwdbs":
    hq, hk, page, d = 32, 8, 64, 128
    for bs, seq in ((1, 4096), (2, 4096), (4, 4096), (8, 4096), (16, 4096), (32, 4096), (64, 4096), (128, 4096), (256, 2048), (16, 512), (16, 1024), (16, 16384), (1, 65536)):
        n_pages = bs * seq // page
        kc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
        vc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
        pt = torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, seq // page)
        lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
        qq = torch.randn(bs, 1, hq, d, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: flash_attn_with_kvcache(qq, kc, vc, cache_seqlens=lens, page_table=pt, causal=True))
        by = 2.0 * bs * seq * hk * d * 2
        print(f"fwd decode bs={bs} seq={seq}: {t:.1f} us  {by / t / 1e6:.2f} TB/s")
        del kc, vc
if what == "elem":
    for rows in (1, 16, 64, 256, 1024, 4096, 16384):
        for hidden in (1024, 2048, 4096, 4104, 5120, 7168, 8192, 16384):
            x = torch.randn(rows, hidden, device=dev, dtype=torch.bfloat16)
            w = torch.ones(hidden, device=dev, dtype=torch.bfloat16)
            r = torch.randn(rows, hidden, device=dev, dtype=torch.bfloat16)
            t1 = timeit(lambda: sgl_kernel.rmsnorm(x, w, 1e-6))
            t2 = timeit(lambda: sgl_kernel.fused_add_rmsnorm(x, r, w, 1e-6))
            print(f"rows={rows} hidden={hidden}: rmsnorm {t1:.1f} us {rows * hidden * 4 / t1 / 1e6:.2f} TB/s | fused_add_rmsnorm {t2:.1f} us {rows * hidden * 8 / t2 / 1e6:.2f} TB/s")
if what == "elem2":
    FP8 = torch.float8_e4m3fn
    for rows in (1, 16, 64, 256, 1024, 4096):
        for hidden in (4096, 14336):
            x2 = torch.randn(rows, 2 * hidden, device=dev, dtype=torch.bfloat16)
            o = torch.empty(rows, hidden, device=dev, dtype=torch.bfloat16)
            t1 = timeit(lambda: torch.ops.sgl_kernel.silu_and_mul(o, x2))
            x = torch.randn(rows, hidden, device=dev, dtype=torch.bfloat16)
            q = torch.empty(rows, hidden, device=dev, dtype=FP8)
            s = torch.empty(hidden // 128, rows, device=dev, dtype=torch.float32).t()
            t2 = timeit(lambda: sgl_kernel.sgl_per_token_group_quant_8bit(x, q, s, 128, 1e-10, -448.0, 448.0, False, enable_v2=False))
            qt = torch.empty(rows, hidden, device=dev, dtype=FP8)
            st = torch.empty(rows, 1, device=dev, dtype=torch.float32)
            t3 = timeit(lambda: sgl_kernel.sgl_per_token_quant_fp8(x, qt, st))
            print(f"rows={rows} hidden={hidden}: silu_and_mul {t1:.1f} us | per_token_group_quant_8bit {t2:.1f} us | per_token_quant_fp8 {t3:.1f} us")
if what == "prefill1":
    hq, hk, page, d = 32, 8, 64, 128
    for bs, q, ctx in ((1, 128, 4096), (1, 128, 32768), (1, 512, 4096), (1, 512, 32768), (1, 2048, 2048), (1, 2048, 32768), (1, 8192, 8192),
                       (2, 128, 32768), (4, 128, 32768), (4, 512, 8192)):
        n_pages = bs * ctx // page
        kc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
        vc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
        pt = torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, ctx // page)
        lens = torch.full((bs,), ctx, device=dev, dtype=torch.int32)
        qq = torch.randn(bs * q, hq, d, device=dev, dtype=torch.bfloat16)
        cu = torch.arange(0, bs + 1, device=dev, dtype=torch.int32) * q
        t = timeit(lambda: flash_attn_with_kvcache(qq, kc, vc, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu, max_seqlen_q=q,
                                                   causal=True), it=10)
        fl = 4.0 * bs * hq * d * q * (ctx - q / 2)
        print(f"fwd chunk prefill bs={bs} q={q} ctx={ctx}: {t:.0f} us  {fl / t / 1e6:.0f} TFLOP/s")
        del kc, vc
if what == "sample":
    V = 128256
    for bs in (1, 8, 32, 64, 65, 256):
        probs = torch.softmax(torch.randn(bs, V, device=dev), dim=-1)
        k = torch.full((bs,), 50, device=dev, dtype=torch.int32)
        pp = torch.full((bs,), 0.9, device=dev, dtype=torch.float32)
        t1 = timeit(lambda: sgl_kernel.top_k_renorm_prob(probs, k), it=10)
        t2 = timeit(lambda: sgl_kernel.top_p_renorm_prob(probs, pp), it=10)
        t3 = timeit(lambda: sgl_kernel.top_k_top_p_sampling_from_probs(probs, k, pp), it=10)
        t4 = timeit(lambda: sgl_kernel.min_p_sampling_from_probs(probs, pp * 0.1), it=10)
        t5 = timeit(lambda: sgl_kernel.top_k_top_p_sampling_from_probs(probs, k, pp, filter_apply_order="joint"), it=10)
        t6 = timeit(lambda: sgl_kernel.top_p_sampling_from_probs(probs, pp), it=10)
        t7 = timeit(lambda: sgl_kernel.top_p_renorm_prob(probs, 1.0), it=10)    # normaliser + write only
        t8 = timeit(lambda: sgl_kernel.top_p_sampling_from_probs(probs, 1.0), it=10)  # normaliser + draw only
        print(f"sampling vocab {V} bs={bs}: top_k_renorm {t1:.0f} us | top_p_renorm {t2:.0f} us | top_k_top_p_sampling {t3:.0f} us | min_p_sampling {t4:.0f} us"
              f" | joint {t5:.0f} | top_p_sampling {t6:.0f} | p = 1: renorm {t7:.0f}, sampling {t8:.0f}")
if what == "elem3":
    def safe(f):
        try:
            return timeit(f)
        except Exception as e:  # a refused call: print why once and go on
            print("    [refused]", str(e).splitlines()[0][:160])
            return float("nan")
    # decode- to prefill-sized calls of the small ops around attention and routing (tokens sweep at Llama / DeepSeek widths)
    from sgl_kernel import elementwise as ew, moe as moe_api, attention as att
    for tokens in (1, 8, 64, 256, 1024, 4096, 16384):
        hq, hk, d = 32, 8, 128
        pos = torch.randint(0, 4096, (tokens,), device=dev)
        q = torch.randn(tokens, hq * d, device=dev, dtype=torch.bfloat16)
        k = torch.randn(tokens, hk * d, device=dev, dtype=torch.bfloat16)
        cache = torch.randn(4096, d, device=dev, dtype=torch.bfloat16)
        t1 = safe(lambda: ew.rotary_embedding(pos, q, k, d, cache, True))
        q3, k3 = q.view(tokens, hq, d), k.view(tokens, hk, d)
        wq = torch.ones(d, device=dev, dtype=torch.bfloat16)
        cache32 = torch.randn(4096, d, device=dev, dtype=torch.float32)
        t2 = safe(lambda: ew.fused_inplace_qknorm_rope(q3, k3, wq, wq, cache32, pos, True))
        qkv = torch.randn(tokens, (hq + 2 * hk) * d, device=dev, dtype=torch.bfloat16)
        t3 = safe(lambda: ew.fused_qk_norm_rope(qkv, hq, hk, hk, d, 1e-6, wq, wq, 10000.0, True, pos.int()))
        kc = torch.empty(65536, hk * d, device=dev, dtype=torch.bfloat16)
        vc = torch.empty(65536, hk * d, device=dev, dtype=torch.bfloat16)
        loc = torch.randperm(65536, device=dev)[:tokens]
        t4 = safe(lambda: ew.store_cache_xpu(k, k, kc, vc, loc))
        va = torch.randn(tokens, hq, d, device=dev, dtype=torch.bfloat16); sa = torch.randn(tokens, hq, device=dev)
        vm = torch.empty_like(va); sm = torch.empty_like(sa)
        t5 = safe(lambda: att.merge_state_v2(va, sa, va, sa, vm, sm))
        print(f"tokens={tokens}: rotary_embedding {t1:.1f} us | fused_inplace_qknorm_rope {t2:.1f} | fused_qk_norm_rope {t3:.1f} | "
              f"store_cache {t4:.1f} | merge_state_v2 {t5:.1f} ({tokens * hq * d * 6 / t5 / 1e6:.2f} TB/s)")
        for E, topk in ((8, 2), (128, 8), (256, 8)):
            g = torch.randn(tokens, E, device=dev, dtype=torch.bfloat16)
            tw = torch.empty(tokens, topk, device=dev); ti = torch.empty(tokens, topk, device=dev, dtype=torch.int32)
            r1 = safe(lambda: moe_api.topk_softmax(tw, ti, g, True))
            r2 = safe(lambda: moe_api.topk_sigmoid(tw, ti, g, True))
            r3 = float("nan")
            if E == 256:
                bias = torch.randn(E, device=dev, dtype=torch.bfloat16)
                r3 = safe(lambda: moe_api.moe_fused_gate(g, bias, 8, 4, topk))
            block = 64
            cap = tokens * topk + E * (block - 1)
            sorted_ids = torch.empty(cap, device=dev, dtype=torch.int32); eids = torch.empty(cap // block + 1, device=dev, dtype=torch.int32)
            npost = torch.empty(1, device=dev, dtype=torch.int32); cum = torch.empty(E + 1, device=dev, dtype=torch.int32)
            r4 = safe(lambda: moe_api.moe_align_block_size(ti, E, block, sorted_ids, eids, npost, cum))
            print(f"    E={E} topk={topk}: topk_softmax {r1:.1f} us | topk_sigmoid {r2:.1f} | moe_fused_gate {r3:.1f} | moe_align_block_size {r4:.1f}")
        x = torch.randn(tokens, 4096, device=dev, dtype=torch.bfloat16)
        qo = torch.empty(tokens, 4096, device=dev, dtype=torch.float8_e4m3fn); so = torch.zeros(1, device=dev)
        p1 = safe(lambda: sgl_kernel.sgl_per_tensor_quant_fp8(x, qo, so, False))
        p2 = safe(lambda: sgl_kernel.sgl_per_tensor_quant_fp8(x, qo, so, True))
        print(f"    per_tensor_quant_fp8 dynamic {p1:.1f} us | static {p2:.1f} us")
    for n in (4096, 14336):
        qw = torch.randint(0, 2 ** 31 - 1, (4096, n // 8), device=dev, dtype=torch.int32)
        sc = torch.rand(4096 // 128, n, device=dev, dtype=torch.float16); zz = torch.randint(0, 2 ** 31 - 1, (4096 // 128, n // 8), device=dev, dtype=torch.int32)
        t = safe(lambda: sgl_kernel.awq_dequantize(qw, sc, zz))
        print(f"awq_dequantize 4096 x {n}: {t:.1f} us  {(4096 * n * 2.5) / t / 1e6:.2f} TB/s")
if what == "gemmbw":
    if os.environ.get("GEMM_SPLITK"):  # diagnostic build only (LD_PRELOAD=.../build/libsglk_probes.so): forced slice count (0: never)
        import ctypes
        ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "build", "libsglk_probes.so")
                    ).sglk_debug_set_gemm_splitk(int(os.environ["GEMM_SPLITK"]))
    # fp8_blockwise_scaled_mm only, rows across the dispatch boundaries at the four Llama-3-8B projections (+ a deep square)
    for N, K in ((4096, 14336), (14336, 4096), (4096, 4096), (6144, 4096), (8192, 8192)):
        g = torch.Generator().manual_seed(0)
        b = ((torch.rand(N, K, generator=g) - 0.5) * 2 * 448).to(FP8).to(dev).t()
        sb = (torch.rand(K // 128, N // 128, generator=g) + 0.5).to(dev)
        for M in [int(m) for m in os.environ.get("GEMM_MS", "32,64,65,72,73,96,128,129,192,256,257,384,512,513,768,1024").split(",")]:
            a = ((torch.rand(M, K, generator=g) - 0.5) * 2 * 448).to(FP8).to(dev)
            sa = (torch.rand(M, K // 128, generator=g) + 0.5).to(dev)
            t1 = timeit(lambda: sgl_kernel.fp8_blockwise_scaled_mm(a, b, sa, sb, torch.bfloat16))
            print(f"N={N} K={K} M={M}: fp8 blockwise {t1:.1f} us  {2.0 * M * N * K / t1 / 1e6:.0f} TFLOP/s  weights {N * K / t1 / 1e6:.2f} TB/s")
if what == "moe2":
    # fused_experts with 16-bit and mxfp4 weights across the token counts where the 4-bit int path had its dispatch boundaries
    E, Hd, I, topk = 8, 4096, 14336, 2
    Ts = [int(t) for t in os.environ.get("MOE_TS", "1,4,16,32,48,64,96,128,192,256,383,384,512,640,768,1024,1536,2048").split(",")]
    if os.environ.get("MOE_BF16_WAVES"):  # diagnostic build only: 4 = the 128-row streaming tile on four waves instead of eight
        import ctypes
        ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "sgl-kernel-xpu_amd", "build", "libsglk_probes.so")
                    ).sglk_debug_set_moe_bf16_waves(int(os.environ["MOE_BF16_WAVES"]))
    for fmt in os.environ.get("MOE_FMTS", "bf16,mxfp4").split(","):
        if fmt == "bf16":
            w1 = (torch.randn(E, 2 * I, Hd, device=dev) * 0.02).to(torch.bfloat16)
            w2 = (torch.randn(E, Hd, I, device=dev) * 0.02).to(torch.bfloat16)
            kw = {}
        else:
            w1 = torch.randint(0, 256, (E, 2 * I, Hd // 2), device=dev, dtype=torch.uint8)
            w2 = torch.randint(0, 256, (E, Hd, I // 2), device=dev, dtype=torch.uint8)
            kw = dict(use_mxfp4_w4a16=True, w1_scale=torch.randint(118, 124, (E, 2 * I, Hd // 32), device=dev, dtype=torch.uint8),
                      w2_scale=torch.randint(118, 124, (E, Hd, I // 32), device=dev, dtype=torch.uint8))
        for T in Ts:
            x = torch.randn(T, Hd, device=dev, dtype=torch.bfloat16) * 0.1
            logits = torch.randn(T, E, device=dev, dtype=torch.bfloat16)
            tw = torch.empty(T, topk, device=dev, dtype=torch.float32)
            ti = torch.empty(T, topk, device=dev, dtype=torch.int32)
            sgl_kernel.topk_softmax(tw, ti, logits, True)
            t = timeit(lambda: sgl_kernel.fused_experts(x, w1, w2, tw, ti, **kw), it=10)
            print(f"fused_experts {fmt} Mixtral T={T}: {t:.0f} us  ({t / T:.2f} us per token)  {2.0 * T * topk * 3 * Hd * I / t / 1e6:.0f} TFLOP/s")
        del w1, w2
if what == "fwdcfg":
    # fwd across head layouts and page sizes (the sweeps above hold 32 / 8 heads of 128, 64-token pages): decode bs 16 x 4096 keys,
    # causal prefill bs 4 x 1024, a 128-query chunk over 4096 keys
    for hq, hk, d, page in ((32, 8, 128, 64), (32, 32, 128, 64), (64, 8, 128, 64), (8, 1, 128, 64), (28, 4, 128, 64), (40, 8, 128, 64),
                            (32, 8, 128, 16), (32, 8, 128, 32), (32, 8, 128, 128), (32, 8, 128, 256), (32, 8, 64, 16), (16, 8, 256, 64),
                            (48, 8, 128, 64), (96, 8, 128, 64), (16, 16, 64, 64), (24, 8, 96, 64), (16, 2, 192, 64)):
        bs, seq = 16, 4096
        n_pages = bs * seq // page
        try:
            kc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
            vc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
            pt = torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, seq // page)
            lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
            qd = torch.randn(bs, 1, hq, d, device=dev, dtype=torch.bfloat16)
            t1 = timeit(lambda: flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens, page_table=pt, causal=True))
            qc = torch.randn(bs * 128, hq, d, device=dev, dtype=torch.bfloat16)
            cu = torch.arange(0, bs + 1, device=dev, dtype=torch.int32) * 128
            t2 = timeit(lambda: flash_attn_with_kvcache(qc, kc, vc, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu, max_seqlen_q=128,
                                                        causal=True), it=5)
            b4, s4 = 4, 1024
            lens4 = torch.full((b4,), s4, device=dev, dtype=torch.int32)
            pt4 = pt[:b4, : s4 // page].contiguous()
            qp = torch.randn(b4 * s4, hq, d, device=dev, dtype=torch.bfloat16)
            cu4 = torch.arange(0, b4 + 1, device=dev, dtype=torch.int32) * s4
            t3 = timeit(lambda: flash_attn_with_kvcache(qp, kc, vc, cache_seqlens=lens4, page_table=pt4, cu_seqlens_q=cu4, max_seqlen_q=s4,
                                                        causal=True), it=5)
            by = 2.0 * bs * seq * hk * d * 2
            print(f"fwdcfg hq={hq} hk={hk} d={d} page={page}: decode {t1:.1f} us {by / t1 / 1e6:.2f} TB/s | chunk q128 {t2:.0f} us "
                  f"{4.0 * bs * hq * d * 128 * (seq - 64) / t2 / 1e6:.0f} TFLOP/s | prefill 4x1024 {t3:.0f} us {4.0 * b4 * hq * d * s4 * s4 / 2 / t3 / 1e6:.0f} TFLOP/s")
            del kc, vc
        except Exception as e:
            print(f"fwdcfg hq={hq} hk={hk} d={d} page={page}: refused: {str(e).splitlines()[0][:140]}")
if what == "mlacfg":
    # flash_mla_decode / prefill across page sizes (the mla mode holds 64-token pages): bs 32 x 4096 keys; prefill 4 x 256 queries
    from sgl_kernel.attention import flash_mla_decode, flash_mla_get_workspace_size, flash_mla_prefill, flash_mla_prefill_get_workspace_size
    for H in (16, 128):
        for page in (1, 16, 32, 64, 128, 256):
            bs, seq = 32, 4096
            n_pages = bs * seq // page
            try:
                cache = torch.randn(n_pages, page, 576, device=dev, dtype=torch.bfloat16)
                qn = torch.randn(bs, H, 512, device=dev, dtype=torch.bfloat16)
                qp = torch.randn(bs, H, 64, device=dev, dtype=torch.bfloat16)
                lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
                table = torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, -1)
                ws = torch.empty(flash_mla_get_workspace_size(seq, bs, H, page, -1), device=dev, dtype=torch.uint8)
                t = timeit(lambda: flash_mla_decode(qn, qp, cache, lens, table, ws, 0.1, -1), it=10)
                print(f"mlacfg decode H={H} page={page}: {t:.1f} us  {bs * seq * 576 * 2 / t / 1e6:.2f} TB/s")
                del cache
            except Exception as e:
                print(f"mlacfg decode H={H} page={page}: refused: {str(e).splitlines()[0][:150]}")
if what == "fwdsplit":
    # fwd decode at short contexts: the auto split count against explicit ones (is the reduce launch worth it?)
    hq, hk, page, d = 32, 8, 64, 128
    for bs, seq in ((16, 256), (16, 512), (16, 1024), (16, 2048), (64, 512), (64, 1024), (4, 1024), (4, 4096), (1, 4096), (1, 1024)):
        n_pages = bs * seq // page
        kc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
        vc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
        pt = torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, seq // page)
        lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
        qd = torch.randn(bs, 1, hq, d, device=dev, dtype=torch.bfloat16)
        row = []
        for ns in (0, 1, 2, 4, 8, 16):
            try:
                t = timeit(lambda: flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens, page_table=pt, causal=True, num_splits=ns))
                row.append(f"{ns}: {t:.1f}")
            except Exception as e:
                row.append(f"{ns}: refused")
        print(f"fwdsplit bs={bs} seq={seq}: us by num_splits (0 = auto)  " + " | ".join(row))
if what == "mlasplit":
    # flash_mla_decode: the auto split count (-1) against explicit ones
    from sgl_kernel.attention import flash_mla_decode, flash_mla_get_workspace_size
    page = 64
    for H in (16, 128):
        for bs, seq in ((1, 2048), (1, 8192), (4, 2048), (4, 8192), (16, 2048), (16, 8192), (32, 4096), (64, 2048)):
            n_pages = bs * seq // page
            cache = torch.randn(n_pages, page, 576, device=dev, dtype=torch.bfloat16)
            qn = torch.randn(bs, H, 512, device=dev, dtype=torch.bfloat16)
            qp = torch.randn(bs, H, 64, device=dev, dtype=torch.bfloat16)
            lens = torch.full((bs,), seq, device=dev, dtype=torch.int32)
            table = torch.arange(n_pages, device=dev, dtype=torch.int32).view(bs, -1)
            row = []
            for ns in (-1, 1, 2, 4, 8, 16, 32, 64):
                try:
                    ws = torch.empty(flash_mla_get_workspace_size(seq, bs, H, page, ns), device=dev, dtype=torch.uint8)
                    t = timeit(lambda: flash_mla_decode(qn, qp, cache, lens, table, ws, 0.1, ns), it=10)
                    row.append(f"{ns}: {t:.1f}")
                except Exception as e:
                    row.append(f"{ns}: refused")
            print(f"mlasplit H={H} bs={bs} seq={seq}: us by num_kv_splits (-1 = auto)  " + " | ".join(row))
            del cache
if what == "prefillsplit":
    # chunks of long sequences on the 128-row-block kernel: the auto split count against explicit ones
    hq, hk, page, d = 32, 8, 64, 128
    for bs, q, ctx in ((1, 128, 4096), (1, 128, 32768), (1, 512, 8192), (1, 512, 32768), (2, 128, 32768), (4, 128, 32768), (4, 128, 4096),
                       (1, 2048, 32768), (8, 64, 8192)):
        n_pages = bs * ctx // page
        kc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
        vc = torch.randn(n_pages, page, hk, d, device=dev, dtype=torch.bfloat16)
        pt = torch.randperm(n_pages, device=dev).to(torch.int32).view(bs, ctx // page)
        lens = torch.full((bs,), ctx, device=dev, dtype=torch.int32)
        qq = torch.randn(bs * q, hq, d, device=dev, dtype=torch.bfloat16)
        cu = torch.arange(0, bs + 1, device=dev, dtype=torch.int32) * q
        row = []
        for ns in (0, 1, 2, 4, 8, 16, 32):
            try:
                t = timeit(lambda: flash_attn_with_kvcache(qq, kc, vc, cache_seqlens=lens, page_table=pt, cu_seqlens_q=cu, max_seqlen_q=q,
                                                           causal=True, num_splits=ns), it=5)
                row.append(f"{ns}: {t:.0f}")
            except Exception as e:
                row.append(f"{ns}: refused")
        print(f"prefillsplit bs={bs} q={q} ctx={ctx}: us by num_splits (0 = auto)  " + " | ".join(row))
        del kc, vc
