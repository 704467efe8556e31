#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own torch-eager test references.

Runs only in the authoring container (needs /root/reference). The reference
package cannot be imported (its `common_ops` SYCL extension cannot be built
here), so a stub module named `sgl_kernel` is seeded into sys.modules and the
pure-torch reference functions are imported from /root/reference/tests/*.py.
They are run on small seeded inputs; inputs and expected outputs are written to
tests/golden/*.pt as plain tensors. Nothing from the reference travels: the
fixtures are data only.

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_TESTS = "/root/reference/tests"


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def _missing(*a, **k):
            raise RuntimeError(f"stub sgl_kernel.{name} called while generating golden vectors")

        return _missing


class _TritonStub(types.ModuleType):
    """`import triton, triton.language as tl` of the reference tests: @triton.jit leaves the function alone (it is
    never called here), tl.constexpr etc. are placeholders."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name == "jit":
            return lambda fn=None, **kw: fn if fn is not None else (lambda f: f)
        return object


def _import_ref(name):
    sys.modules.setdefault("sgl_kernel", _Stub("sgl_kernel"))
    if "triton" not in sys.modules:
        tr = _TritonStub("triton")
        tr.language = _TritonStub("triton.language")
        sys.modules["triton"] = tr
        sys.modules["triton.language"] = tr.language
    if REF_TESTS not in sys.path:
        sys.path.insert(0, REF_TESTS)
    return importlib.import_module(name)


def save(name, obj):
    path = os.path.join(HERE, name + ".pt")
    torch.save(obj, path)
    print("wrote %-28s %7.1f KiB" % (name + ".pt", os.path.getsize(path) / 1024))


def gen_norm():
    t = _import_ref("test_norm")
    cases = []
    g = torch.Generator().manual_seed(0)
    for rows, n, dt in [(19, 1024, torch.float16), (7, 111, torch.float16), (5, 3072, torch.bfloat16),
                        (3, 500, torch.float32), (2, 8192, torch.float16)]:
        x = torch.randn(rows, n, generator=g).to(dt)
        r = torch.randn(rows, n, generator=g).to(dt)
        w = torch.randn(n, generator=g).to(dt)
        fa_y, fa_r = t.fused_add_rms_norm(x.clone(), r.clone(), w, 1e-6)
        ga_y, ga_r = t.gemma_fused_add_rms_norm(x.clone(), r.clone(), w, 1e-6)
        cases.append(dict(x=x, residual=r, w=w, eps=1e-6,
                          rmsnorm=t.llama_rms_norm(x, w, 1e-6),
                          gemma_rmsnorm=t.gemma_rms_norm(x, w, 1e-6),
                          fused_add=(fa_y, fa_r), gemma_fused_add=(ga_y, ga_r)))
    save("norm", cases)


def gen_activation():
    """The reference's expected values are inline expressions inside its test functions (tests/test_activation.py:18,
    28, 38), so the TEST FUNCTIONS THEMSELVES are run here: the stub op returns a marker, and the test's own
    torch.testing.assert_close call is intercepted to capture (input, expected)."""
    t = _import_ref("test_activation")
    stub = sys.modules["sgl_kernel"]
    cases = []
    for dim, batch, seq in [(128, 2, 4), (512, 3, 1), (2048, 1, 8), (11008, 2, 2)]:
        case = {}
        for key, test_fn, op in (("silu", t.test_fused_silu_mul, "silu_and_mul"),
                                 ("gelu_tanh", t.test_fused_gelu_tanh_mul, "gelu_tanh_and_mul"),
                                 ("gelu", t.test_fused_gelu_mul, "gelu_and_mul")):
            seen = {}

            def fake_op(x, _seen=seen):
                _seen["x"] = x.clone()
                return "marker"

            def fake_close(expected, actual, **kw):
                assert actual == "marker"
                seen["expected"], seen["tol"] = expected.clone(), kw

            setattr(stub, op, fake_op)
            real = torch.testing.assert_close
            torch.testing.assert_close = fake_close
            try:
                torch.manual_seed(dim + batch)
                test_fn(dim, batch, seq)
            finally:
                torch.testing.assert_close = real
                delattr(stub, op)
            case[key] = dict(x=seen["x"].cpu(), out=seen["expected"].cpu(), rtol=seen["tol"]["rtol"], atol=seen["tol"]["atol"])
        cases.append(case)
    save("activation", cases)


def gen_quant():
    t = _import_ref("test_per_token_group_quant_8bit")
    cases = []
    g = torch.Generator().manual_seed(42)
    for rows, k, gs, dt, scale in [(5, 512, 128, torch.bfloat16, 1.0), (40, 2048, 128, torch.float16, 1.0),
                                   (7, 256, 64, torch.float32, 1.0), (9, 1024, 128, torch.bfloat16, 1e-3),
                                   (9, 1024, 128, torch.bfloat16, 100.0), (4, 256, 32, torch.bfloat16, 1.0)]:
        x = (torch.randn(rows, k, generator=g) * scale).to(dt)
        q8, s8 = t.per_token_group_quant_fp8_ref(x, gs, 1e-10, False)
        qu, su = t.per_token_group_quant_fp8_ref(x, gs, 1e-10, True)
        qi, si = t.per_token_group_quant_int8_ref(x, gs, 1e-10)
        cases.append(dict(x=x, group_size=gs, fp8_q=q8.view(torch.uint8), fp8_s=s8,
                          fp8_ue8m0_q=qu.view(torch.uint8), fp8_ue8m0_s=su, int8_q=qi, int8_s=si))
    save("quant", cases)


def gen_quant_v2():
    """tests/test_per_token_group_quant_8bit_v2.py:408-480: the v2 op's own torch references (fused silu * mul, UE8M0)."""
    t = _import_ref("test_per_token_group_quant_8bit_v2")
    cases = []
    g = torch.Generator().manual_seed(7)
    for rows, hidden, gs, ue, fuse in [(5, 512, 128, False, False), (9, 1024, 128, True, False), (6, 512, 64, False, True),
                                       (4, 2048, 128, True, True), (3, 256, 32, False, False), (7, 128, 16, False, False)]:
        width = hidden * (2 if fuse else 1)
        x = torch.randn(rows, width, generator=g).to(torch.bfloat16)
        q, s = t.per_token_group_quant_fp8_ref(x, gs, 1e-10, ue, fuse)
        qi, si = t.per_token_group_quant_int8_ref(x[:, :hidden].contiguous(), gs, 1e-10)
        cases.append(dict(x=x, group_size=gs, scale_ue8m0=ue, fuse_silu_and_mul=fuse, fp8_q=q.view(torch.uint8), fp8_s=s,
                          int8_q=qi, int8_s=si))
    save("quant_v2", cases)


def gen_merge_state():
    """tests/test_merge_state_v2.py:101-137 (merge_state_torch, natural log) on the test's own input recipe (:198-224:
    randn lse with ~10 % of either side set to +inf, never both)."""
    t = _import_ref("test_merge_state_v2")
    cases = []
    for tokens, heads, d, dt in [(64, 8, 32, torch.half), (37, 16, 48, torch.bfloat16), (33, 8, 128, torch.bfloat16),
                                 (7, 8, 256, torch.half), (5, 4, 64, torch.float32)]:
        torch.manual_seed(tokens + d)
        s_a = torch.randn(tokens, heads)
        s_b = torch.randn(tokens, heads)
        ma, mb = torch.rand(tokens, heads) < 0.1, torch.rand(tokens, heads) < 0.1
        both = ma & mb
        s_a[ma & ~both] = float("inf")
        s_b[mb & ~both] = float("inf")
        v_a = torch.randn(tokens, heads, d).to(dt)
        v_b = torch.randn(tokens, heads, d).to(dt)
        out, lse = t.merge_state_torch(v_a.float(), s_a.clone(), v_b.float(), s_b.clone(), torch.empty_like(v_a, dtype=torch.float32),
                                       torch.empty_like(s_a))
        cases.append(dict(v_a=v_a, s_a=s_a, v_b=v_b, s_b=s_b, v_merged=out, s_merged=lse))
    save("merge_state", cases)


def gen_qknorm_rope():
    """tests/test_fused_qk_norm_rope.py:104-218: fused_qk_norm_rope_reference (analytic / YaRN angles) and
    fused_qk_norm_rope_with_cache_reference (cos_sin_cache from tests/test_rope_utils.py:create_cos_sin_cache)."""
    t = _import_ref("test_fused_qk_norm_rope")
    cases = {"yarn": [], "cache": []}
    for tokens, hq, hk, hv, d, neox, dt, factor, low, high, af, rot in [
            (7, 8, 8, 8, 64, True, torch.bfloat16, 1.0, 1.0, 1.0, 1.0, 64),
            (9, 6, 2, 2, 128, False, torch.float16, 1.0, 1.0, 1.0, 1.0, 128),
            (12, 4, 2, 2, 128, True, torch.bfloat16, 4.0, 1.0, 32.0, 1.2, 128),      # YaRN (:302-377)
            (6, 4, 2, 2, 256, False, torch.bfloat16, 4.0, 1.0, 32.0, 1.2, 256),
            (10, 4, 4, 2, 128, True, torch.bfloat16, 1.0, 1.0, 1.0, 1.0, 32),        # partial rotary (:380-455)
            (10, 4, 4, 2, 128, False, torch.bfloat16, 1.0, 1.0, 1.0, 1.0, 64)]:
        torch.manual_seed(42)
        qkv = torch.randn(tokens, (hq + hk + hv) * d).to(dt)
        qw, kw = torch.randn(d).to(dt), torch.randn(d).to(dt)
        pos = torch.arange(tokens, dtype=torch.int32) * 3
        out = t.fused_qk_norm_rope_reference(qkv.float(), hq, hk, hv, d, 1e-6, qw.float(), kw.float(), 10000.0, neox, pos,
                                             factor, low, high, af, rot)
        cases["yarn"].append(dict(qkv=qkv, Hq=hq, Hk=hk, Hv=hv, head_dim=d, eps=1e-6, q_weight=qw, k_weight=kw, base=10000.0,
                                  is_neox=neox, position_ids=pos, factor=factor, low=low, high=high, attention_factor=af,
                                  rotary_dim=rot, out=out))
    for tokens, hq, hk, d, rope, neox, dt, pdt in [(3, 4, 2, 64, 32, False, torch.bfloat16, torch.int32),
                                                   (5, 8, 4, 128, 64, True, torch.float16, torch.int64),
                                                   (8, 6, 2, 128, 128, False, torch.bfloat16, torch.int32),
                                                   (4, 8, 2, 256, 128, True, torch.float16, torch.int64),
                                                   (6, 4, 4, 64, 8, True, torch.bfloat16, torch.int32)]:
        torch.manual_seed(42)
        q, k = torch.randn(tokens, hq, d).to(dt), torch.randn(tokens, hk, d).to(dt)
        qw, kw = torch.randn(d).to(dt), torch.randn(d).to(dt)
        pos = torch.randperm(tokens + 1)[:tokens].to(pdt)
        cache = t.create_cos_sin_cache(rope, max_position=tokens + 1)
        q_ref, k_ref = t.fused_qk_norm_rope_with_cache_reference(q.float(), k.float(), qw.float(), kw.float(), cache, pos, neox)
        cases["cache"].append(dict(q=q, k=k, q_weight=qw, k_weight=kw, cos_sin_cache=cache, positions=pos, is_neox=neox,
                                   q_out=q_ref, k_out=k_ref))
    save("qknorm_rope", cases)


def gen_moe_gates():
    """tests/test_topk_sigmoid.py:41-92, tests/test_biased_topk.py:13-66, tests/test_moe_fused_gate.py:68-139: the three
    routers' torch references on small seeded inputs (indices + weights)."""
    cases = {"topk_sigmoid": [], "biased_topk": [], "moe_fused_gate": []}
    t = _import_ref("test_topk_sigmoid")
    for T, E, k, shared, renorm, with_bias, rsf in [(7, 8, 2, 0, False, False, 1.0), (32, 32, 4, 1, True, True, 2.5),
                                                    (19, 256, 4, 0, True, True, 2.5), (5, 256, 2, 1, False, False, 2.5)]:
        torch.manual_seed(1024 + E)
        x = torch.randn(T, E)
        bias = torch.randn(E) if with_bias else None
        w, ids = t.fused_topk_sigmoid_torch_native(torch.randn(T, 100), x, k, renorm, bias, routed_scaling_factor=rsf,
                                                   num_fused_shared_experts=shared)
        cases["topk_sigmoid"].append(dict(x=x, bias=bias, topk=k, shared=shared, renormalize=renorm, rsf=rsf, weights=w, ids=ids))
    t = _import_ref("test_biased_topk")
    for T, E, k, scoring, shared, renorm, apply in [(9, 128, 4, "sigmoid", 0, True, False), (64, 384, 6, "sqrtsoftplus", 1, True, True),
                                                    (5, 512, 8, "sigmoid", 1, False, True), (3, 128, 8, "sqrtsoftplus", 0, False, False)]:
        torch.manual_seed(E * 100 + k)
        x = torch.randn(T, E) * 2.0
        bias = torch.randn(E) * 0.5
        w, ids = t.biased_topk_torch_native(torch.randn(T, 100), x, bias, k, renorm, scoring, shared, 2.5, apply)
        cases["biased_topk"].append(dict(x=x, bias=bias, topk=k, scoring=scoring, shared=shared, renormalize=renorm, rsf=2.5,
                                         apply=apply, weights=w, ids=ids))
    t = _import_ref("test_moe_fused_gate")
    for T, (E, G, tg, k), scoring, renorm, apply in [(9, (128, 4, 2, 4), "sigmoid", True, False), (33, (256, 8, 4, 8), "sigmoid", True, True),
                                                     (7, (512, 16, 8, 16), "softmax", False, False), (16, (256, 8, 4, 8), "softmax", True, True)]:
        torch.manual_seed(T)
        x = torch.rand(T, E)
        bias = None if scoring == "softmax" else torch.rand(E)
        w, ids = t.biased_grouped_topk_native(x, x, bias, topk=k, renormalize=renorm, num_expert_group=G, topk_group=tg,
                                              num_fused_shared_experts=0, routed_scaling_factor=2.5,
                                              apply_routed_scaling_factor_on_output=apply, scoring_func=scoring)
        cases["moe_fused_gate"].append(dict(x=x, bias=bias, G=G, topk_group=tg, topk=k, scoring=scoring, renormalize=renorm, rsf=2.5,
                                            apply=apply, weights=w, ids=ids))
    save("moe_gates", cases)


def gen_sampling():
    """tests/test_sampling.py: torch_top_k_renorm_probs (:176-204), torch_top_p_renorm_probs (:104-126; it regenerates its
    input from seed 42, so the same recipe is used here), torch_top_k_top_p_joint_mask (:13-34), torch_min_p_sampling
    (:262-274)."""
    t = _import_ref("test_sampling")
    t.device = "cpu"
    cases = {"top_k_renorm": [], "top_p_renorm": [], "joint_mask": [], "min_p_mask": []}
    for B, V, k in [(3, 111, 10), (5, 3000, 100), (4, 3000, torch.tensor([10, 33, 49, 17]))]:
        torch.manual_seed(42)
        pre = torch.rand(B, V)
        pr = pre / pre.sum(dim=-1, keepdim=True)
        cases["top_k_renorm"].append(dict(probs=pr, k=k, out=t.torch_top_k_renorm_probs(pr, k)))
    for B, V, pp in [(3, 111, 0.1), (5, 3000, 1.0), (4, 3000, torch.tensor([0.2, 0.5, 0.75, 0.9]))]:
        torch.manual_seed(42)
        pre = torch.rand(B, V)
        pr = pre / pre.sum(dim=-1, keepdim=True)
        cases["top_p_renorm"].append(dict(probs=pr, p=pp, out=t.torch_top_p_renorm_probs(pr, pp)))
    for B, V, k, pp in [(3, 111, 55, 0.1), (5, 3000, 300, 0.5), (4, 3000, torch.tensor([10, 49, 150, 70]), torch.tensor([0.1, 0.3, 0.7, 0.5]))]:
        torch.manual_seed(42)
        pre = torch.rand(B, V)
        pr = pre / pre.sum(dim=-1, keepdim=True)
        cases["joint_mask"].append(dict(probs=pr, k=k, p=pp, mask=t.torch_top_k_top_p_joint_mask(B, V, k, pp, pr)))
    for B, V, pp in [(3, 111, 0.05), (5, 3000, 0.7), (4, 3000, torch.tensor([0.05, 0.1, 0.2, 0.6]))]:
        torch.manual_seed(42)
        pre = torch.rand(B, V)
        pr = pre / pre.sum(dim=-1, keepdim=True)
        real = torch.zeros
        cases["min_p_mask"].append(dict(probs=pr, p=pp, mask=_min_p_mask_ref(t, B, V, pp, pr)))
    save("sampling", cases)


def _min_p_mask_ref(t, B, V, pp, pr):
    # torch_min_p_sampling builds its mask with device=f"{device}:0": run it with torch.zeros redirected to the CPU
    real = torch.zeros
    torch.zeros = lambda *a, **k: real(*a, **{**k, "device": "cpu"})
    try:
        return t.torch_min_p_sampling(B, V, pp, pr)
    finally:
        torch.zeros = real


def gen_fp8_blockwise():
    t = _import_ref("test_fp8_blockwise_gemm")
    cases = []
    torch.manual_seed(0)
    fmax = torch.finfo(torch.float8_e4m3fn).max
    for M, N, K, dt in [(1, 128, 512, torch.bfloat16), (5, 512, 1024, torch.float16), (127, 128, 512, torch.bfloat16),
                        (33, 384, 256, torch.bfloat16)]:
        a = ((torch.rand(M, K) - 0.5) * 2 * fmax).clamp(-fmax, fmax).to(torch.float8_e4m3fn)
        b = ((torch.rand(N, K) - 0.5) * 2 * fmax).clamp(-fmax, fmax).to(torch.float8_e4m3fn).t()
        sa = (torch.randn(M, K // 128) * 0.001).t().contiguous().t()
        sb = (torch.randn(K // 128, (N + 127) // 128) * 0.001).t().contiguous().t()
        out = t.baseline_scaled_mm(a, b, sa, sb, dt)
        cases.append(dict(a=a.view(torch.uint8), b_nk=b.t().contiguous().view(torch.uint8), sa=sa.contiguous(),
                          sb=sb.contiguous(), out_dtype=dt, out=out))
    save("fp8_blockwise_gemm", cases)


def gen_scaled_mm():
    t8 = _import_ref("test_fp8_gemm")
    sys.modules["utils"].is_sm10x = lambda: False  # tests/test_int8_gemm.py:7 imports a helper utils.py lacks
    ti = _import_ref("test_int8_gemm")
    cases = []
    torch.manual_seed(0)
    fmax = torch.finfo(torch.float8_e4m3fn).max
    for M, N, K, dt, with_bias in [(1, 16, 512, torch.bfloat16, True), (17, 128, 1024, torch.float16, False),
                                   (128, 512, 512, torch.bfloat16, True)]:
        a = ((torch.rand(M, K) - 0.5) * 2 * fmax).clamp(-fmax, fmax).to(torch.float8_e4m3fn)
        b = ((torch.rand(N, K) - 0.5) * 2 * fmax).clamp(-fmax, fmax).to(torch.float8_e4m3fn).t()
        sa, sb = torch.randn(M) * 0.001, torch.randn(N) * 0.001
        bias = torch.randn(N).to(dt) if with_bias else None
        out = t8.torch_scaled_mm(a, b, sa, sb, dt, bias)
        cases.append(dict(kind="fp8", a=a.view(torch.uint8), b_nk=b.t().contiguous().view(torch.uint8), sa=sa, sb=sb,
                          bias=bias, out_dtype=dt, out=out))
        ai = ti.to_int8(torch.randn(M, K) * 5)
        bi = ti.to_int8(torch.randn(N, K).t() * 5)
        sa, sb = torch.randn(M), torch.randn(N)
        bias = (torch.randn(N).to(dt) * 10) if with_bias else None
        out = ti.torch_scaled_mm(ai, bi, sa, sb, dt, bias)
        cases.append(dict(kind="int8", a=ai, b_nk=bi.t().contiguous(), sa=sa, sb=sb, bias=bias, out_dtype=dt, out=out))
    save("scaled_mm", cases)


def gen_mla_decode():
    # tests/test_flash_mla_decode.py:14-18 skips the whole module without an XPU; patch the probe for the import
    import torch as _t
    if not hasattr(_t, "xpu"):
        raise RuntimeError("torch.xpu missing")
    orig = _t.xpu.is_available
    _t.xpu.is_available = lambda: True
    try:
        t = _import_ref("test_flash_mla_decode")
    finally:
        _t.xpu.is_available = orig
    cases = []
    torch.manual_seed(42)
    for dt, bs, H, page, seqs in [(torch.bfloat16, 2, 16, 64, [77, 200]), (torch.float16, 3, 32, 16, [5, 130, 64]),
                                  (torch.bfloat16, 1, 128, 128, [300])]:
        seq_lens = torch.tensor(seqs, dtype=torch.int32)
        block_num = (max(seqs) + page - 1) // page
        pack = 128 // page
        block_num = (block_num + pack - 1) // pack * pack
        q = torch.randn(bs, H, 576, dtype=dt) * 100
        table = torch.randint(0, bs * block_num, (bs, block_num), dtype=torch.int32)
        cache = torch.randn(table.numel(), page, 576, dtype=dt)
        out = torch.zeros(bs, H, 512, dtype=dt)
        scale = (128 + 64) ** -0.5
        t.ref_mla(out, q, cache, scale, table, seq_lens)
        cases.append(dict(q=q, cache=cache, table=table, seq_lens=seq_lens, scale=scale, out=out))
    save("mla_decode", cases)


def gen_mla_prefill():
    import torch as _t
    orig = _t.xpu.is_available
    _t.xpu.is_available = lambda: True
    try:
        t = _import_ref("test_flash_mla_prefill")
    finally:
        _t.xpu.is_available = orig
    cases = []
    torch.manual_seed(42)
    # shapes from the parameter list of tests/test_flash_mla_prefill.py:103-131 (small ones)
    # (token counts cut down so that the fixture stays small: q and out are [total_q, H, 512])
    for dt, H, page, sqs, sks in [(torch.bfloat16, 16, 64, [17], [128]), (torch.float16, 16, 16, [9, 1, 12], [64, 32, 100]),
                                  (torch.bfloat16, 128, 32, [2, 1], [73, 40]), (torch.float16, 16, 128, [17, 5], [17, 5])]:
        bs = len(sqs)
        cu = torch.tensor([0] + torch.cumsum(torch.tensor(sqs), 0).tolist(), dtype=torch.int32)
        sk = torch.tensor(sks, dtype=torch.int32)
        block_num = (max(sks) + page - 1) // page
        pack = 128 // page
        block_num = (block_num + pack - 1) // pack * pack
        qn = torch.randn(sum(sqs), H, 512, dtype=dt)
        qp = torch.randn(sum(sqs), H, 64, dtype=dt)
        table = torch.randint(0, bs * block_num, (bs, block_num), dtype=torch.int32)
        cache = torch.randn(int(table.max()) + 1, page, 576, dtype=dt)
        scale = (128 + 64) ** -0.5
        out = t.ref_mla_prefill_varlen(qn, qp, cache, scale, table, cu, sk, causal=True)
        cases.append(dict(q_nope=qn, q_pe=qp, cache=cache, table=table, cu_seqlens_q=cu, seq_lens_k=sk, scale=scale,
                          out=out))
    save("mla_prefill", cases)


def gen_qserve():
    """tests/test_qserve_w4a8_per_chn_gemm.py and ..._per_group_gemm.py: quantisers, repacking and references."""
    import types as _types
    sys.modules.setdefault("utils", _types.SimpleNamespace(get_device=lambda: torch.device("cpu")))
    tc = _import_ref("test_qserve_w4a8_per_chn_gemm")
    tg = _import_ref("test_qserve_w4a8_per_group_gemm")
    cases = {"chn": [], "group": []}
    torch.manual_seed(0)
    for M, N, K in [(5, 64, 128), (16, 128, 512), (33, 96, 256)]:
        a = torch.randn(M, K) * 0.01
        b = torch.randn(N, K) * 0.01
        a_q, a_scale = tc.sym_quantize_tensor(a)
        b_q, b_scale, b_zero = tc.asym_quantize_tensor(b)
        w, ws, wsz = tc.convert_to_qserve_format(b_q, b_scale, b_zero)
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):  # the reference function prints shapes
            ref = tc.torch_w4a8_per_chn_gemm(a_q.float(), b_q.float(), a_scale.float(), b_scale.float(), b_zero.float(), torch.float32)
        cases["chn"].append(dict(a=a, b=b, a_q=a_q, a_scale=a_scale, b_q=b_q, b_scale=b_scale, b_zero=b_zero, packed=w,
                                 wscales=ws, w_szs=wsz, a_ssums=a.sum(dim=-1, keepdim=True).to(torch.float16),
                                 out=ref.to(torch.float16)))
    for M, N, K in [(5, 64, 128), (16, 128, 512), (33, 96, 256)]:
        a = torch.randn(M, K) * 0.01
        b = torch.randn(N, K) * 0.01
        a_q, a_scale = tg.sym_quantize_tensor(a)
        b_q, chn, s8, z8 = tg.progressive_group_quantize_tensor(b, 128)
        w, chn_f, s8_f, z8_f = tg.convert_to_qserve_format(b_q, chn, s8, z8, 128)
        ref = tg.torch_w4a8_per_group_gemm(a_q, b_q, a_scale, chn, s8, z8, 128, torch.float16)
        cases["group"].append(dict(a=a, b=b, a_q=a_q, a_scale=a_scale, b_q=b_q, chn_scale=chn, scale_i8=s8, zero_i8=z8,
                                   packed=w, wscales=chn_f, scales_i8=s8_f, zeros=z8_f, out=ref))
    save("qserve_w4a8", cases)


def gen_quant_extra():
    """tests/test_per_token_quant_fp8.py:18-27, test_per_tensor_quant_fp8.py:30-38, test_awq_dequant.py:13-62."""
    import types as _types
    sys.modules.setdefault("utils", _types.SimpleNamespace(get_device=lambda: torch.device("cpu")))
    tt = _import_ref("test_per_token_quant_fp8")
    tp = _import_ref("test_per_tensor_quant_fp8")
    ta = _import_ref("test_awq_dequant")
    cases = {"token": [], "tensor": [], "awq": []}
    torch.manual_seed(42)
    for dt, rows, cols in [(torch.float16, 7, 512), (torch.bfloat16, 5, 1076), (torch.bfloat16, 3, 1368)]:
        x = torch.rand(rows, cols, dtype=dt)
        scale = x.float().abs().amax(dim=-1) / 448.0  # the kernel's scale (per_token_quant_fp8.cpp:109)
        cases["token"].append(dict(x=x, scale=scale, q=tt.torch_per_token_quant_fp8(x, scale).view(torch.uint8)))
    for dt, rows, cols in [(torch.float16, 8, 512), (torch.bfloat16, 4, 2048)]:
        x = torch.rand(rows, cols, dtype=dt)
        scale = (x.float().abs().max() / 448.0).reshape(1)
        # the kernel multiplies by 1 / (scale + 1e-8) (per_tensor_quant_fp8.cpp:66); the reference test compares it with
        # x * scale.reciprocal() at rtol / atol 1e-3
        cases["tensor"].append(dict(x=x, scale=scale, q=tp.torch_scaled_fp8_quant(x, scale).view(torch.uint8)))
    for dt, k, c in [(torch.float16, 128, 16), (torch.bfloat16, 256, 32)]:
        qw = torch.randint(0, torch.iinfo(torch.int32).max, (k, c), dtype=torch.int32)
        sc = torch.rand(1, c * 8, dtype=dt)
        qz = torch.randint(0, torch.iinfo(torch.int32).max, (1, c), dtype=torch.int32)
        cases["awq"].append(dict(qweight=qw, scales=sc, qzeros=qz, out=ta.awq_dequantize_torch(qw, sc, qz, k)))
    save("quant_extra", cases)


def gen_moe():
    t = _import_ref("test_moe_gemm")
    cases = {"grouped_mm": [], "fused": []}
    # tests/test_moe_gemm.py:347-386 (_check_int4_grouped_mm): E=8, 2 rows/expert, N=128, K=256
    for explicit_zero, dt, gs in [(False, torch.bfloat16, 32), (True, torch.bfloat16, 64), (False, torch.float16, 128),
                                  (True, torch.float16, 256)]:
        torch.manual_seed(0)
        E, rpe, n, k = 8, 2, 128, 256
        act = torch.randn(E * rpe, k, dtype=dt) * 0.1
        packed, scales, zeros, weights = t._make_int4_weight(E, n, k, gs, dt, explicit_zero)
        expected = torch.cat([act[e * rpe:(e + 1) * rpe].float() @ weights[e].float().t() for e in range(E)]).to(dt)
        cases["grouped_mm"].append(dict(act=act, packed=packed.view(torch.uint8), scales=scales, zeros=zeros,
                                        group_size=gs, rows_per_expert=rpe, out=expected))
    # tests/test_moe_gemm.py:408-471 (test_fused_experts_int4_w4a16): seed 1, T=5, k=2, E=8, H=128, I=64, g=32
    for explicit_zero in (False, True):
        for dt in (torch.bfloat16, torch.float16):
            for with_bias in (False, True):
                for activation in ("silu", "relu2"):
                    torch.manual_seed(1)
                    T, topk, E, H, I, gs = 5, 2, 8, 128, 64, 32
                    gate = 1 if activation == "relu2" else 2
                    x = torch.randn(T, H, dtype=dt) * 0.1
                    w1, w1s, w1z, w1ref = t._make_int4_weight(E, gate * I, H, gs, dt, explicit_zero)
                    w2, w2s, w2z, w2ref = t._make_int4_weight(E, H, I, gs, dt, explicit_zero)
                    ids = torch.tensor([[0, 1], [2, 3], [4, 5], [6, 7], [0, 2]], dtype=torch.int64)
                    tw = torch.rand(T, topk, dtype=torch.float32)
                    tw /= tw.sum(dim=-1, keepdim=True)
                    b1 = torch.randn(E, gate * I) * 0.005 if with_bias else None
                    b2 = torch.randn(E, H) * 0.005 if with_bias else None
                    exp = t.torch_naive_moe(x, w1ref, w2ref, ids, tw, topk, b1, b2, activations=activation)
                    cases["fused"].append(dict(x=x, w1=w1.view(torch.uint8), w2=w2.view(torch.uint8), w1_scale=w1s,
                                               w2_scale=w2s, w1_zp=w1z, w2_zp=w2z, topk_ids=ids, topk_weights=tw,
                                               b1=b1, b2=b2, activation=activation, out=exp))
    # tests/test_moe_gemm.py:140-236 (test_moe_gemm): 16-bit weights, torch_naive_moe as the reference
    cases["fused16"] = []
    for T, topk, E, H, I, act, with_bias in [(5, 2, 8, 128, 64, "silu", False), (7, 3, 8, 256, 128, "gelu", True),
                                             (4, 1, 8, 128, 64, "relu2", False)]:
        torch.manual_seed(3)
        gate = 1 if act == "relu2" else 2
        x = torch.randn(T, H, dtype=torch.bfloat16) * 0.1
        w1 = torch.randn(E, gate * I, H, dtype=torch.bfloat16) * 0.1
        w2 = torch.randn(E, H, I, dtype=torch.bfloat16) * 0.1
        b1 = torch.randn(E, gate * I) * 0.005 if with_bias else None
        b2 = torch.randn(E, H) * 0.005 if with_bias else None
        score = torch.softmax(torch.randn(T, E, dtype=torch.bfloat16).float(), dim=-1)
        tw, ids = torch.topk(score, topk)
        exp = t.torch_naive_moe(x, w1, w2, ids, tw, topk, b1, b2, activations=act, routed_scaling_factor=2.5)
        cases["fused16"].append(dict(x=x, w1=w1, w2=w2, topk_ids=ids, topk_weights=tw, b1=b1, b2=b2, activation=act,
                                     routed_scaling_factor=2.5, out=exp))
    # Mixtral-shaped small-T case (E=8, top-2, group 128, routing through a softmax top-k as SGLang does; the hidden and
    # intermediate widths are cut to keep the fixture small): torch_naive_moe on the dequantised weights
    torch.manual_seed(11)
    T, topk, E, H, I, gs = 3, 2, 8, 512, 256, 128
    x = torch.randn(T, H, dtype=torch.bfloat16) * 0.1
    w1, w1s, w1z, w1ref = t._make_int4_weight(E, 2 * I, H, gs, torch.bfloat16, False)
    w2, w2s, w2z, w2ref = t._make_int4_weight(E, H, I, gs, torch.bfloat16, False)
    score = torch.softmax(torch.randn(T, E, dtype=torch.bfloat16).float(), dim=-1)
    tw, ids = torch.topk(score, topk)
    tw = tw / tw.sum(dim=-1, keepdim=True)
    exp = t.torch_naive_moe(x, w1ref, w2ref, ids, tw, topk, None, None, activations="silu")
    cases["fused"].append(dict(x=x, w1=w1.view(torch.uint8), w2=w2.view(torch.uint8), w1_scale=w1s, w2_scale=w2s, w1_zp=None,
                               w2_zp=None, topk_ids=ids, topk_weights=tw, b1=None, b2=None, activation="silu", out=exp))
    # mxfp4 (tests/test_moe_gemm.py:245-290 quantise / dequantise helpers; :767-846 op-level check; :555-640 fused check)
    cases["mxfp4_dequant"], cases["mxfp4_grouped_mm"], cases["mxfp4_fused"] = [], [], []
    for dt in (torch.bfloat16, torch.float16):
        torch.manual_seed(0)
        E, rpe, n, k = 8, 3, 64, 256
        act = t.create_random_cpu_tensor((E * rpe, k), dt)
        w = t.create_random_cpu_tensor((E, n, k), dt)
        w[0, 0, :32] = 0           # an all-zero block (scale byte from the zero guard)
        w[1, 1, 32:64] *= 2.0 ** -20  # small / large block exponents
        w[2, 2, 64:96] *= 2.0 ** 9
        packed, scales = t._quantize_weights_mxfp4(w)
        dq = t._dequantize_weights_mxfp4(packed, scales, dtype=dt)
        cases["mxfp4_dequant"].append(dict(packed=packed, scales=scales, out=dq))
        expected = torch.cat([act[e * rpe:(e + 1) * rpe].float() @ dq[e].float().t() for e in range(E)]).to(dt)
        cases["mxfp4_grouped_mm"].append(dict(act=act, packed=packed, scales=scales, rows_per_expert=rpe, out=expected))
    for T, topk, E, H, I in [(1, 1, 8, 128, 128), (9, 2, 8, 256, 128)]:
        torch.manual_seed(0)
        a = t.create_random_cpu_tensor((T, H), torch.bfloat16)
        w1 = t.create_random_cpu_tensor((E, 2 * I, H), torch.bfloat16)
        w2 = t.create_random_cpu_tensor((E, H, I), torch.bfloat16)
        score = torch.softmax(torch.randn([T, E], dtype=torch.bfloat16), dim=-1, dtype=torch.float32)
        tw, ids = torch.topk(score, topk)
        w1p, w1s = t._quantize_weights_mxfp4(w1)
        w2p, w2s = t._quantize_weights_mxfp4(w2)
        exp = t.torch_naive_moe(a, t._dequantize_weights_mxfp4(w1p, w1s), t._dequantize_weights_mxfp4(w2p, w2s), ids, tw,
                                topk, None, None, activations="silu")
        cases["mxfp4_fused"].append(dict(x=a, w1=w1p, w2=w2p, w1_scale=w1s, w2_scale=w2s, topk_ids=ids, topk_weights=tw,
                                         out=exp))
    save("moe_w4a16", cases)


def gen_topk_softmax():
    t = _import_ref("test_topk_softmax")
    cases = []
    torch.manual_seed(1024)
    for dt, n_token, n_expert, topk, renorm in [(torch.bfloat16, 32, 8, 2, True), (torch.float16, 32, 256, 4, False),
                                                (torch.bfloat16, 7, 32, 1, True), (torch.float16, 64, 60, 3, True)]:
        gating = torch.randn(n_token, n_expert).to(dt)  # unit scale: no exact ties, weights are meaningful
        hidden = torch.randn(n_token, 100)
        w, ids = t.fused_topk_torch_native(hidden, gating.float(), topk, renorm)
        cases.append(dict(gating=gating, topk=topk, renormalize=renorm, weights=w, ids=ids.to(torch.int32)))
    save("topk_softmax", cases)


def gen_attention():
    t = _import_ref("test_flash_attention")
    cases = []
    torch.manual_seed(0)
    for dt, b, sq, sk, Hq, Hk, D, causal, window, softcap, use_sink in [
        (torch.bfloat16, 2, 1, 130, 8, 2, 128, False, (-1, -1), 0.0, False),
        (torch.float16, 2, 33, 97, 4, 4, 64, True, (-1, -1), 0.0, False),
        (torch.bfloat16, 1, 64, 64, 4, 1, 64, False, (17, 5), 0.0, True),
        (torch.float16, 2, 20, 40, 6, 2, 80, True, (-1, -1), 30.0, False),
        (torch.bfloat16, 1, 5, 300, 16, 4, 256, True, (-1, -1), 0.0, False),
    ]:
        q = torch.randn(b, sq, Hq, D).to(dt)
        k = torch.randn(b, sk, Hk, D).to(dt)
        v = torch.randn(b, sk, Hk, D).to(dt)
        sink = torch.randn(Hq) if use_sink else None
        scale = D ** -0.5
        out, _ = t.attention_ref(q, k, v, scale, sink=sink, causal=causal, window_size=window, softcap=softcap)
        out_pt, _ = t.attention_ref(q, k, v, scale, sink=sink, causal=causal, window_size=window, softcap=softcap,
                                    upcast=False, reorder_ops=True)
        cases.append(dict(q=q, k=k, v=v, sink=sink, scale=scale, causal=causal, window=window, softcap=softcap,
                          out=out, out_pt=out_pt))
    save("attention", cases)
    # KV-cache addressing of flash_attn_with_kvcache (reference tests/test_flash_attention.py:855-893, :938-990): a cache
    # with more rows than sequences, cache_batch_idx picks the row, keys are the cache positions [leftpad, cache_seqlens)
    kvc = []
    torch.manual_seed(1)
    for dt, b, rows, sq, cache_len, Hq, Hk, D, causal in [(torch.bfloat16, 3, 5, 1, 200, 8, 2, 128, False),
                                                           (torch.float16, 2, 4, 17, 96, 4, 4, 64, True)]:
        q = torch.randn(b, sq, Hq, D).to(dt)
        k_cache = torch.randn(rows, cache_len, Hk, D).to(dt)
        v_cache = torch.randn(rows, cache_len, Hk, D).to(dt)
        batch_idx = torch.randperm(rows)[:b].to(torch.int32)
        seqlens = torch.randint(cache_len // 2, cache_len + 1, (b,), dtype=torch.int32)
        leftpad = torch.tensor([int(torch.randint(0, max(1, int(s) - sq), (1,))) for s in seqlens], dtype=torch.int32)
        ar = torch.arange(cache_len).unsqueeze(0)
        key_mask = (ar < seqlens.unsqueeze(1)) & (ar >= leftpad.unsqueeze(1))
        out, _ = t.attention_ref(q, k_cache[batch_idx.long()], v_cache[batch_idx.long()], D ** -0.5, key_padding_mask=key_mask,
                                 causal=causal, key_leftpad=leftpad)
        out_pt, _ = t.attention_ref(q, k_cache[batch_idx.long()], v_cache[batch_idx.long()], D ** -0.5, key_padding_mask=key_mask,
                                    causal=causal, key_leftpad=leftpad, upcast=False, reorder_ops=True)
        kvc.append(dict(q=q, k_cache=k_cache, v_cache=v_cache, cache_batch_idx=batch_idx, cache_seqlens=seqlens,
                        cache_leftpad=leftpad, causal=causal, out=out, out_pt=out_pt))
    save("attention_kvcache", kvc)


def gen_rope():
    t = _import_ref("test_rotary_embedding")
    cases = []
    torch.manual_seed(0)
    for dt, tokens, hq, hk, head, rot, neox in [(torch.bfloat16, 7, 4, 2, 64, 64, True), (torch.float16, 5, 8, 8, 128, 64, False),
                                                (torch.bfloat16, 9, 2, 1, 96, 32, True)]:
        rope = t.RotaryEmbedding(head, rot, 256, 10000, neox, dt)
        pos = torch.randint(0, 256, (tokens,))
        q = torch.randn(tokens, hq * head).to(dt)
        k = torch.randn(tokens, hk * head).to(dt)
        rope.cos_sin_cache = rope.cos_sin_cache.to(dt)  # what the kernel is handed (forward_xpu :138-148)
        qo, ko = rope.forward_native(pos, q.clone(), k.clone())
        cases.append(dict(positions=pos, q=q, k=k, head_size=head, cache=rope.cos_sin_cache.clone(), is_neox=neox,
                          q_out=qo, k_out=ko))
    save("rope", cases)


def gen_swiglu():
    """swiglu_gpt_oss_sigmoid_alpha and silu_and_mul_clamp: inputs and outputs of the reference tests' own pure-torch
    functions (tests/test_swiglu_with_alpha_limit.py:9-14, tests/test_silu_and_mul_clamp.py:8-91), on the CPU."""
    ta = _import_ref("test_swiglu_with_alpha_limit")
    tc = _import_ref("test_silu_and_mul_clamp")
    g = torch.Generator().manual_seed(11)
    alpha_cases, clamp_cases = [], []
    for rows, hidden, alpha, limit, dt in [(1, 64, 0.5, 1.0, torch.float32), (16, 128, 1.0, 5.0, torch.bfloat16),
                                           (128, 256, 2.0, 10.0, torch.float16), (5, 66, 1.702, 7.0, torch.bfloat16),
                                           (3, 1024, 1.702, 7.0, torch.float32)]:
        x = (torch.randn(rows, hidden, generator=g) * 4).to(dt)
        alpha_cases.append(dict(x=x, alpha=alpha, limit=limit, out=ta.swiglu_gpt_oss_sigmoid_alpha_ref(x, alpha, limit)))
    for M, H, dt in [(16, 32, torch.bfloat16), (128, 64, torch.float16), (16, 64, torch.bfloat16), (7, 40, torch.float16)]:
        x = (torch.randn(M, 2 * H, generator=g) * 6).to(dt)
        out = torch.zeros(M, H, dtype=dt)
        tc.silu_and_mul_clamp_torch(x, out, 10.0)
        clamp_cases.append(dict(x=x, limit=10.0, out=out))
    save("swiglu", dict(alpha=alpha_cases, clamp=clamp_cases))


GENERATORS = {
    "swiglu": gen_swiglu,
    "rope": gen_rope,
    "attention": gen_attention,
    "moe_w4a16": gen_moe,
    "topk_softmax": gen_topk_softmax,
    "mla_decode": gen_mla_decode,
    "mla_prefill": gen_mla_prefill,
    "qserve_w4a8": gen_qserve,
    "quant_extra": gen_quant_extra,
    "norm": gen_norm,
    "activation": gen_activation,
    "moe_gates": gen_moe_gates,
    "sampling": gen_sampling,
    "quant_v2": gen_quant_v2,
    "merge_state": gen_merge_state,
    "qknorm_rope": gen_qknorm_rope,
    "quant": gen_quant,
    "fp8_blockwise_gemm": gen_fp8_blockwise,
    "scaled_mm": gen_scaled_mm,
}

if __name__ == "__main__":
    if not os.path.isdir(REF_TESTS):
        sys.exit("reference tests not found at %s (this script only runs in the authoring container)" % REF_TESTS)
    names = sys.argv[1:] or list(GENERATORS)
    for n in names:
        GENERATORS[n]()
