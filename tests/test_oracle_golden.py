"""CPU: the oracle restatements against golden vectors produced by the reference's own
torch-eager test references (tests/golden/make_golden.py). Tolerances are the
reference tests' (cited per test)."""
import math

import torch
from conftest import load_golden

from oracle import activation as oact
from oracle import attn_aux as oaux
from oracle import qknorm_rope as oqk
from oracle import gemm as ogemm
from oracle import norm as onorm
from oracle import quant as oquant


def norm_tol(dtype):  # reference tests/test_norm.py:45-50
    if dtype == torch.float32:
        return dict(rtol=1e-4, atol=1e-4)
    if dtype == torch.bfloat16:
        return dict(rtol=1e-2, atol=1e-2)
    return dict(rtol=1e-3, atol=1e-3)


def test_norm_family_matches_reference_vectors():
    for c in load_golden("norm"):
        x, r, w, eps = c["x"], c["residual"], c["w"], c["eps"]
        tol = norm_tol(x.dtype)
        torch.testing.assert_close(onorm.rmsnorm(x, w, eps), c["rmsnorm"], **tol)
        torch.testing.assert_close(onorm.gemma_rmsnorm(x, w, eps), c["gemma_rmsnorm"], **tol)
        y, nr = onorm.fused_add_rmsnorm(x, r, w, eps)
        torch.testing.assert_close(y, c["fused_add"][0], **tol)
        torch.testing.assert_close(nr, c["fused_add"][1], **tol)
        y, nr = onorm.gemma_fused_add_rmsnorm(x, r, w, eps)
        torch.testing.assert_close(y, c["gemma_fused_add"][0], **tol)
        torch.testing.assert_close(nr, c["gemma_fused_add"][1], **tol)


def test_activation_matches_reference_vectors():
    # fixtures captured by RUNNING the reference's test functions (tests/test_activation.py:16-40): each entry holds the
    # input the test built, the expected value it computed and the tolerance it asserted with
    for c in load_golden("activation"):
        for key, fn in (("silu", oact.silu_and_mul), ("gelu_tanh", oact.gelu_tanh_and_mul), ("gelu", oact.gelu_and_mul)):
            e = c[key]
            torch.testing.assert_close(fn(e["x"]), e["out"], rtol=e["rtol"], atol=e["atol"])


def test_swiglu_variants_match_reference_vectors():
    # outputs of the reference tests' own pure-torch functions (tests/test_swiglu_with_alpha_limit.py:9-14, which works in
    # the input dtype - hence its 1e-1 tolerance for 16-bit inputs, :42-43 - and tests/test_silu_and_mul_clamp.py:8-91)
    g = load_golden("swiglu")
    for c in g["alpha"]:
        tol = 1e-4 if c["x"].dtype == torch.float32 else 1e-1
        torch.testing.assert_close(oact.swiglu_gpt_oss_sigmoid_alpha(c["x"], c["alpha"], c["limit"]), c["out"], rtol=tol, atol=tol)
    for c in g["clamp"]:
        torch.testing.assert_close(oact.silu_and_mul_clamp(c["x"], c["limit"]), c["out"], rtol=1e-2, atol=1e-2)


def test_quant_matches_reference_vectors():
    for c in load_golden("quant"):
        x, gs = c["x"], c["group_size"]
        rows, k = x.shape
        q, s, _ = oquant.per_token_group_quant_8bit(x, gs, torch.float8_e4m3fn)
        # reference tests/test_per_token_group_quant_8bit.py:260-276
        torch.testing.assert_close(s, c["fp8_s"], rtol=1e-3, atol=1e-5)
        ref_q = c["fp8_q"].view(torch.float8_e4m3fn)
        deq = q.float().view(rows, -1, gs) * s.unsqueeze(-1)
        ref_deq = ref_q.float().view(rows, -1, gs) * c["fp8_s"].unsqueeze(-1)
        torch.testing.assert_close(deq, ref_deq, rtol=1e-1, atol=1e-1)
        # the two only differ where x*(1/s) and x/s round to different fp8 codes (1 ulp)
        assert (q.view(torch.uint8) != c["fp8_q"]).float().mean() < 0.02

        q, s, ue = oquant.per_token_group_quant_8bit(x, gs, torch.float8_e4m3fn, scale_ue8m0=True)
        torch.testing.assert_close(s, c["fp8_ue8m0_s"], rtol=0, atol=0)  # powers of two: exact (:405)
        assert torch.equal(ue.to(torch.int32) - 127, torch.log2(s).round().to(torch.int32))

        q, s, _ = oquant.per_token_group_quant_8bit(x, gs, torch.int8)
        torch.testing.assert_close(s, c["int8_s"], rtol=1e-3, atol=1e-5)
        assert (q.to(torch.int32) - c["int8_q"].to(torch.int32)).abs().max() <= 1


def test_fp8_blockwise_matches_reference_vectors():
    for c in load_golden("fp8_blockwise_gemm"):
        a = c["a"].view(torch.float8_e4m3fn)
        b = c["b_nk"].view(torch.float8_e4m3fn).t()
        out = ogemm.fp8_blockwise_scaled_mm(a, b, c["sa"], c["sb"], c["out_dtype"])
        torch.testing.assert_close(out, c["out"], rtol=0.02, atol=1)  # tests/test_fp8_blockwise_gemm.py:83-85
        torch.testing.assert_close(out.float(), c["out"].float(), rtol=1e-2, atol=1e-4)


def test_scaled_mm_matches_reference_vectors():
    for c in load_golden("scaled_mm"):
        if c["kind"] == "fp8":
            a = c["a"].view(torch.float8_e4m3fn)
            b = c["b_nk"].view(torch.float8_e4m3fn).t()
            out = ogemm.fp8_scaled_mm(a, b, c["sa"], c["sb"], c["out_dtype"], c["bias"])
            torch.testing.assert_close(out, c["out"], rtol=0.02, atol=1)  # tests/test_fp8_gemm.py:38-40
        else:
            out = ogemm.int8_scaled_mm(c["a"], c["b_nk"].t(), c["sa"], c["sb"], c["out_dtype"], c["bias"])
            torch.testing.assert_close(out, c["out"])  # tests/test_int8_gemm.py:36


def test_mla_decode_matches_reference_vectors():
    from oracle import mla as omla
    for c in load_golden("mla_decode"):
        o = omla.mla_decode(c["q"], c["cache"], c["scale"], c["table"], c["seq_lens"])
        tol = 1e-2 if c["q"].dtype == torch.bfloat16 else 1e-3  # tests/test_flash_mla_decode.py:145-146
        torch.testing.assert_close(o.float(), c["out"].float(), atol=tol, rtol=tol)


def test_moe_w4a16_matches_reference_vectors():
    from oracle import moe as omoe
    g = load_golden("moe_w4a16")
    for c in g["grouped_mm"]:
        E = c["packed"].shape[0]
        rows = torch.full((E,), c["rows_per_expert"], dtype=torch.int32)
        o = omoe.moe_grouped_mm_w4a16(c["act"], c["packed"], c["scales"], c["zeros"], None, rows, c["group_size"])
        torch.testing.assert_close(o, c["out"], rtol=5e-2, atol=2e-2)  # tests/test_moe_gemm.py:386
    for c in g["fused"]:
        o = omoe.fused_experts_int4(c["x"], c["w1"], c["w2"], c["topk_weights"], c["topk_ids"], c["w1_scale"],
                                    c["w2_scale"], c["w1_zp"], c["w2_zp"], c["b1"], c["b2"], c["activation"])
        torch.testing.assert_close(o, c["out"], rtol=1e-1, atol=2e-2)  # tests/test_moe_gemm.py:471
    # mxfp4: the dequantisation is exact arithmetic on both sides, so the weights must agree bit for bit
    for c in g["mxfp4_dequant"]:
        assert torch.equal(omoe.dequant_mxfp4(c["packed"], c["scales"], c["out"].dtype), c["out"])
    for c in g["mxfp4_grouped_mm"]:
        E = c["packed"].shape[0]
        rows = torch.full((E,), c["rows_per_expert"], dtype=torch.int32)
        o = omoe.moe_grouped_mm_w4a16(c["act"], c["packed"], c["scales"], None, None, rows, 32, mxfp4=True)
        assert torch.equal(o, c["out"])
    for c in g["mxfp4_fused"]:
        o = omoe.fused_experts_int4(c["x"], c["w1"], c["w2"], c["topk_weights"], c["topk_ids"], c["w1_scale"],
                                    c["w2_scale"], mxfp4=True)
        torch.testing.assert_close(o, c["out"], rtol=1e-1, atol=1e-2)  # tests/test_moe_gemm.py:580


def test_topk_softmax_matches_reference_vectors():
    from oracle import moe as omoe
    for c in load_golden("topk_softmax"):
        w, idx, p = omoe.topk_softmax(c["gating"], c["topk"], c["renormalize"])
        for r in (idx != c["ids"]).any(dim=1).nonzero().flatten().tolist():
            a, b = set(idx[r].tolist()), set(c["ids"][r].tolist())
            # equal-score ties may be broken differently (tests/test_topk_softmax.py:12-37)
            assert sorted(p[r, list(a - b)].tolist()) == sorted(p[r, list(b - a)].tolist())
        torch.testing.assert_close(w.sort(dim=1).values, c["weights"].sort(dim=1).values, rtol=1e-5, atol=1e-6)


def test_moe_align_and_prepare_known_answers():
    import numpy as np
    from oracle import moe as omoe
    ids = np.array([[0, 2], [2, 1], [0, 0], [-1, 2]])
    s, e, total, prefix = omoe.moe_align_block_size(ids, 4, 4)  # buckets: id+1; 3 real experts + the "-1" bucket
    assert prefix.tolist() == [0, 4, 8, 12, 16] and total == 16
    assert e.tolist() == [-1, 0, 1, 2]
    assert s.tolist() == [6, 8, 8, 8, 0, 4, 5, 8, 3, 8, 8, 8, 1, 2, 7, 8]
    cnt, ps1, ps2, a_map, c_map = omoe.prepare_moe_input(np.array([[0, 2], [2, 1], [0, 0]]), 3, 7, 2)
    assert cnt.tolist() == [3, 1, 2] and ps1[0].tolist() == [3, 14, 2] and ps2[2].tolist() == [2, 2, 7]
    assert a_map.tolist() == [0, 2, 2, 1, 0, 1] and c_map.tolist() == [0, 4, 5, 3, 1, 2]


def test_attention_matches_reference_vectors():
    from oracle import attention as oa
    for c in load_golden("attention"):
        q, k, v = c["q"], c["k"], c["v"]
        for b in range(q.shape[0]):
            o, _ = oa.attention_seq(q[b], k[b], v[b], c["scale"], causal=c["causal"], window=c["window"],
                                    softcap=c["softcap"], sinks=c["sink"])
            ref = c["out"][b].float()
            err = (o.to(q.dtype).float() - ref).abs().max().item()
            err_pt = (c["out_pt"][b].float() - ref).abs().max().item()
            assert err <= 2 * err_pt + 1e-5  # tests/test_flash_attention.py:1112-1121


def test_rope_matches_reference_vectors():
    from oracle import rope as orope
    for c in load_golden("rope"):
        qo, ko = orope.rotary_embedding(c["positions"], c["q"], c["k"], c["head_size"], c["cache"], c["is_neox"])
        assert torch.equal(qo, c["q_out"]) and torch.equal(ko, c["k_out"])


def test_mla_prefill_oracle_matches_reference_vectors():
    """oracle.mla.mla_prefill vs ref_mla_prefill_varlen outputs (reference tests/test_flash_mla_prefill.py:30-90)."""
    from oracle import mla as omla

    for c in load_golden("mla_prefill"):
        out = omla.mla_prefill(c["q_nope"], c["q_pe"], c["cache"], c["scale"], c["table"], c["cu_seqlens_q"],
                               c["seq_lens_k"], causal=True)
        tol = 1e-2 if c["q_nope"].dtype == torch.bfloat16 else 1e-3  # reference tolerance (:235-236)
        torch.testing.assert_close(out.float(), c["out"].float(), atol=tol, rtol=tol)


def test_qserve_oracle_matches_reference_vectors():
    """oracle.qserve (quantisers, 32x32 repacking, both GEMM references) vs the outputs of the reference's own test
    functions (tests/test_qserve_w4a8_per_chn_gemm.py, tests/test_qserve_w4a8_per_group_gemm.py)."""
    from oracle import qserve as oq

    g = load_golden("qserve_w4a8")
    for c in g["chn"]:
        a_q, a_scale = oq.sym_quantize(c["a"])
        b_q, b_scale, b_zero = oq.asym_quantize_u4(c["b"])
        assert torch.equal(a_q, c["a_q"]) and torch.equal(a_scale, c["a_scale"])
        assert torch.equal(b_q, c["b_q"]) and torch.equal(b_scale, c["b_scale"]) and torch.equal(b_zero, c["b_zero"])
        w, ws, wsz = oq.per_chn_inputs(b_q, b_scale, b_zero)
        assert torch.equal(w, c["packed"]) and torch.equal(ws, c["wscales"]) and torch.equal(wsz, c["w_szs"])
        out = oq.w4a8_per_chn_gemm(a_q, b_q, a_scale, b_scale, b_zero)
        torch.testing.assert_close(out, c["out"], rtol=1e-3, atol=1e-2)  # reference tolerance (:111)
    for c in g["group"]:
        b_q, chn, s8, z8 = oq.progressive_group_quantize(c["b"])
        assert torch.equal(b_q, c["b_q"]) and torch.equal(chn, c["chn_scale"])
        assert torch.equal(s8, c["scale_i8"]) and torch.equal(z8, c["zero_i8"])
        w, ws, s8f, z8f = oq.per_group_inputs(b_q, chn, s8, z8)
        assert torch.equal(w, c["packed"]) and torch.equal(ws, c["wscales"])
        assert torch.equal(s8f, c["scales_i8"]) and torch.equal(z8f, c["zeros"])
        out = oq.w4a8_per_group_gemm(c["a_q"], b_q, c["a_scale"], chn, s8, z8)
        torch.testing.assert_close(out, c["out"], rtol=1e-3, atol=1e-5)  # reference tolerance (:175)


def test_quant_extra_oracle_matches_reference_vectors():
    """oracle per-token / per-tensor fp8 quant and AWQ dequant vs the reference test references."""
    g = load_golden("quant_extra")
    for c in g["token"]:
        q, s = oquant.per_token_quant_fp8(c["x"])
        assert torch.equal(s, c["scale"])
        # reference tolerance (test_per_token_quant_fp8.py:60-62): rtol = atol = 1e-3 on the dequantised codes
        torch.testing.assert_close(q.float(), c["q"].view(torch.float8_e4m3fn).float(), rtol=1e-3, atol=1e-3)
    for c in g["tensor"]:
        q, s = oquant.per_tensor_quant_fp8(c["x"])
        assert torch.equal(s, c["scale"])
        # 1 / (scale + 1e-8) vs scale.reciprocal() may move a value across a rounding boundary: a handful of codes
        # differ by one step, which the reference accepts through its tolerance on the few that do (:57-59)
        a, b = q.float(), c["q"].view(torch.float8_e4m3fn).float()
        assert ((a - b).abs() <= 0.125 * b.abs() + 1e-3).all() and (a != b).float().mean() < 0.02
    for c in g["awq"]:
        out = oquant.awq_dequantize(c["qweight"], c["scales"], c["qzeros"])
        torch.testing.assert_close(out.float(), c["out"].float(), rtol=1e-3, atol=1e-5)  # reference tolerance (:119-121)


def test_fused_experts_16bit_oracle_matches_torch_naive_moe():
    """oracle.moe.fused_experts_16bit vs torch_naive_moe outputs (reference tests/test_moe_gemm.py:59-137) on 16-bit
    weights; the oracle rounds the two intermediates to bf16 as the op sequence does, the reference keeps them in its
    working dtype, so the comparison uses the W4A16 tolerance of the same file (:471), not the 1e-4 of :236."""
    from oracle import moe as omoe

    for c in load_golden("moe_w4a16")["fused16"]:
        for fused in (False, True):  # both routes of the op sequence (activation kernel / GEMM epilogue)
            out = omoe.fused_experts_16bit(c["x"], c["w1"], c["w2"], c["topk_weights"], c["topk_ids"], c["b1"], c["b2"],
                                           c["activation"], c["routed_scaling_factor"], fused_epilogue=fused)
            torch.testing.assert_close(out.float(), c["out"].float(), rtol=1e-1, atol=2e-2)
            torch.testing.assert_close(out.float(), c["out"].float(), rtol=3e-2, atol=3e-3)


def test_quant_v2_matches_reference_vectors():
    """oracle.quant.per_token_group_quant_8bit_v2 vs the v2 test's own references
    (reference tests/test_per_token_group_quant_8bit_v2.py:408-480), incl. the fused silu * mul input and UE8M0 scales."""
    for c in load_golden("quant_v2"):
        x, gs = c["x"], c["group_size"]
        q, s, ue, _ = oquant.per_token_group_quant_8bit_v2(x, gs, torch.float8_e4m3fn, scale_ue8m0=c["scale_ue8m0"],
                                                           fuse_silu_and_mul=c["fuse_silu_and_mul"])
        ref_q = c["fp8_q"].view(torch.float8_e4m3fn).float()
        if c["scale_ue8m0"]:
            assert torch.equal(ue.to(torch.int32), c["fp8_s"].to(torch.int32))  # exponent bytes, exact
            scale = torch.pow(2.0, c["fp8_s"].float() - 127.0)
        else:
            torch.testing.assert_close(s, c["fp8_s"], rtol=1e-6, atol=0)
            scale = c["fp8_s"]
        # same acceptance as the reference (dequantised values, :599-629): a code may differ by one fp8 step where the
        # fused silu is rounded differently
        deq = q.float() * s.repeat_interleave(gs, dim=-1)
        ref = ref_q * scale.repeat_interleave(gs, dim=-1)
        torch.testing.assert_close(deq, ref, rtol=0.13, atol=1e-2 if c["fuse_silu_and_mul"] else 1e-6)
        if not c["fuse_silu_and_mul"]:
            # the two only differ where x*(1/s) and x/s round to different fp8 codes (1 ulp), as for v1 above
            assert (q.view(torch.uint8) != c["fp8_q"]).float().mean() < 0.02
            qi, si, _, _ = oquant.per_token_group_quant_8bit_v2(x, gs, torch.int8)
            torch.testing.assert_close(si, c["int8_s"], rtol=1e-6, atol=0)
            assert (qi.int() - c["int8_q"].int()).abs().max() <= 1


def test_merge_state_matches_reference_vectors():
    ln2 = math.log(2.0)
    for c in load_golden("merge_state"):
        v, s = oaux.merge_state(c["v_a"], c["s_a"], c["v_b"], c["s_b"], base2=False)
        tol = dict(rtol=1e-5, atol=1e-5) if c["v_a"].dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
        torch.testing.assert_close(v.float(), c["v_merged"].to(v.dtype).float(), **tol)
        torch.testing.assert_close(s, c["s_merged"], rtol=1e-5, atol=1e-5)
        # the base-2 op: same merge in other units (its only reference in the reference's tests is a Triton kernel)
        v2, s2 = oaux.merge_state(c["v_a"], c["s_a"] / ln2, c["v_b"], c["s_b"] / ln2, base2=True)
        torch.testing.assert_close(v2.float(), v.float(), **tol)
        torch.testing.assert_close(s2 * ln2, s, rtol=1e-5, atol=1e-5)


def test_qknorm_rope_matches_reference_vectors():
    prec = {torch.bfloat16: 1e-2, torch.float16: 1e-3}
    g = load_golden("qknorm_rope")
    for c in g["cache"]:
        q, k = oqk.fused_inplace_qknorm_rope(c["q"], c["k"], c["q_weight"], c["k_weight"], c["cos_sin_cache"], c["positions"],
                                             c["is_neox"])
        p = prec[c["q"].dtype]
        torch.testing.assert_close(q, c["q_out"].to(q.dtype), rtol=p, atol=p)  # tests/test_fused_qk_norm_rope.py:622-627
        torch.testing.assert_close(k, c["k_out"].to(k.dtype), rtol=p, atol=p)
    for c in g["yarn"]:
        out = oqk.fused_qk_norm_rope(c["qkv"], c["Hq"], c["Hk"], c["Hv"], c["head_dim"], c["eps"], c["q_weight"], c["k_weight"],
                                     c["base"], c["is_neox"], c["position_ids"], c["factor"], c["low"], c["high"],
                                     c["attention_factor"], c["rotary_dim"])
        p = prec[c["qkv"].dtype] * (2 if c["factor"] != 1.0 else 1)  # :375-377
        torch.testing.assert_close(out, c["out"].to(out.dtype), rtol=p, atol=p)


def _same_routing(w, ids, ref_w, ref_ids, n_cols, rtol=1e-4, atol=1e-5):
    """Order inside a row is not part of the contract (the reference's torch.topk(sorted=False)): compare the weights
    scattered by expert id, as reference tests/test_biased_topk.py:77-86 does; the id SETS must be equal."""
    dense = torch.zeros(w.shape[0], n_cols).scatter_(1, ids.long(), w.float())
    ref = torch.zeros(w.shape[0], n_cols).scatter_(1, ref_ids.long(), ref_w.float())
    assert torch.equal(ids.long().sort(dim=1).values, ref_ids.long().sort(dim=1).values)
    torch.testing.assert_close(dense, ref, rtol=rtol, atol=atol)


def test_moe_gates_match_reference_vectors():
    from oracle import moe_gates as og

    gold = load_golden("moe_gates")
    for c in gold["topk_sigmoid"]:
        w, ids = og.topk_sigmoid(c["x"], c["topk"], c["renormalize"], c["bias"], c["rsf"], c["shared"])
        _same_routing(w, ids, c["weights"], c["ids"], c["x"].shape[1] + 1)
    for c in gold["biased_topk"]:
        w, ids = og.biased_topk(c["x"], c["bias"], c["topk"], c["scoring"], c["shared"], c["renormalize"], c["rsf"], c["apply"])
        _same_routing(w, ids, c["weights"], c["ids"], c["x"].shape[1] + 1)
    for c in gold["moe_fused_gate"]:
        w, ids = og.moe_fused_gate(c["x"], c["bias"], c["G"], c["topk_group"], c["topk"], 0, c["scoring"], c["renormalize"],
                                   c["rsf"], c["apply"])
        _same_routing(w, ids, c["weights"], c["ids"], c["x"].shape[1], rtol=1e-2, atol=1e-3)  # test_moe_fused_gate.py:22-23


def test_sampling_filters_match_reference_vectors():
    from oracle import sampling as osamp

    gold = load_golden("sampling")
    for c in gold["top_k_renorm"]:
        torch.testing.assert_close(osamp.renorm(c["probs"], osamp.top_k_mask(c["probs"], c["k"])), c["out"], rtol=1e-3, atol=1e-3)
    for c in gold["top_p_renorm"]:
        torch.testing.assert_close(osamp.renorm(c["probs"], osamp.top_p_mask(c["probs"], c["p"])), c["out"], rtol=1e-3, atol=1e-3)
    for c in gold["joint_mask"]:
        mine = osamp.top_k_mask(c["probs"], c["k"]) & osamp.top_p_mask(c["probs"], c["p"])
        assert torch.all(c["mask"][mine] == 1)  # the reference's mask has an eps of slack on the top-p side: a superset
        assert (c["mask"].bool() & ~mine).sum() <= mine.shape[0] * 2
    for c in gold["min_p_mask"]:
        assert torch.equal(osamp.min_p_mask(c["probs"], c["p"]), c["mask"].bool())


def test_attention_kvcache_addressing_matches_reference_vectors():
    """cache_batch_idx / cache_leftpad semantics (keys = cache positions [leftpad, cache_seqlens) of row batch_idx[b]) as
    attention_ref computes them from key_padding_mask + key_leftpad (reference tests/test_flash_attention.py:855-990)."""
    from oracle import attention as oa

    for c in load_golden("attention_kvcache"):
        q = c["q"]
        b, sq, Hq, D = q.shape
        for i in range(b):
            row, lo, hi = int(c["cache_batch_idx"][i]), int(c["cache_leftpad"][i]), int(c["cache_seqlens"][i])
            out, _ = oa.attention_seq(q[i], c["k_cache"][row, lo:hi], c["v_cache"][row, lo:hi], D ** -0.5, causal=c["causal"])
            err = (out - c["out"][i].float()).abs().max().item()
            err_pt = (c["out_pt"][i].float() - c["out"][i].float()).abs().max().item()
            assert err <= 2 * err_pt + 1e-5, (err, err_pt)
