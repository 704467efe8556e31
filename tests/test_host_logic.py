"""CPU: host-side behaviour of the python package and the torch registry (no GPU needed):
import, op registration with the reference's schemas, argument errors raised before launch,
and the absence of any CPU fallback."""
import pytest
import torch


def test_import_registers_ops(sglk):
    for name in ["rmsnorm", "fused_add_rmsnorm", "gemma_rmsnorm", "gemma_fused_add_rmsnorm", "silu_and_mul",
                 "gelu_tanh_and_mul", "gelu_and_mul", "silu_and_mul_clamp", "swiglu_gpt_oss_sigmoid_alpha",
                 "sgl_per_token_group_quant_8bit",
                 "fp8_blockwise_scaled_mm"]:
        assert hasattr(torch.ops.sgl_kernel, name), name


def test_schemas_match_reference_contract(sglk):
    # reference src/torch_extension_sycl.cc:41, :29, :395-398
    s = torch.ops.sgl_kernel.rmsnorm.default._schema
    assert str(s) == "sgl_kernel::rmsnorm(Tensor($0! -> ) output, Tensor input, Tensor weight, float eps) -> ()"
    # reference src/torch_extension_sycl.cc:32, :108
    assert str(torch.ops.sgl_kernel.silu_and_mul_clamp.default._schema) == \
        "sgl_kernel::silu_and_mul_clamp(Tensor($0! -> ) out, Tensor input, float swiglu_limit) -> ()"
    assert str(torch.ops.sgl_kernel.swiglu_gpt_oss_sigmoid_alpha.default._schema) == \
        "sgl_kernel::swiglu_gpt_oss_sigmoid_alpha(Tensor x, float alpha, float limit) -> Tensor"
    s = torch.ops.sgl_kernel.sgl_per_token_group_quant_8bit.default._schema
    assert [a.name for a in s.arguments] == ["input", "output_q", "output_s", "group_size", "eps", "fp8_min",
                                             "fp8_max", "scale_ue8m0"]
    s = torch.ops.sgl_kernel.fp8_blockwise_scaled_mm.default._schema
    assert [a.name for a in s.arguments] == ["mat_a", "mat_b", "scales_a", "scales_b", "out_dtype"]


def test_no_cpu_fallback(sglk):
    x, w = torch.randn(2, 8), torch.randn(8)
    with pytest.raises(NotImplementedError):
        sglk.rmsnorm(x, w)
    with pytest.raises(NotImplementedError):
        sglk.silu_and_mul(torch.randn(2, 16))


def test_activation_row_rule(sglk):
    # reference python/sgl_kernel/elementwise.py:216-217
    with pytest.raises(ValueError, match="multiple of 16 bytes"):
        sglk.silu_and_mul(torch.randn(2, 6, dtype=torch.float16))


def test_out_of_scope_names_raise_on_call_only(sglk):
    f = sglk.lightning_attention_decode
    with pytest.raises(NotImplementedError):
        f()
    with pytest.raises(AttributeError):
        sglk.definitely_not_an_op


# ---- wrapper contract: argument normalisation pinned with the op call intercepted ----------

class _Recorder:
    """Stands in for torch.ops.sgl_kernel: records (op name, args) and returns what a caller needs to go on."""

    def __init__(self, answers=None):
        self.calls = []
        self.answers = answers or {}

    def __getattr__(self, name):
        rec = self

        class _Op:
            def __call__(self, *a, **k):
                rec.calls.append((name, a, k))
                if name == "swiglu_gpt_oss_sigmoid_alpha":  # (the one op of these wrappers whose result is used)
                    return a[0].new_empty(a[0].shape[0], a[0].shape[1] // 2)
                if name in rec.answers:  # (host-side predicates: moe_w4a16_splitk_applies, moe_grouped_mm_nt_w4a16_splitk)
                    return rec.answers[name]
                return 0

            default = property(lambda self_: self_)

        return _Op()

    def names(self):
        return [c[0] for c in self.calls]


def test_fwd_slot_table_matches_schema(sglk):
    from sgl_kernel import flash_attn as fa

    schema = torch.ops.sgl_kernel.fwd.default._schema
    assert [a.name for a in schema.arguments] == [n for n, _ in fa._FWD_SLOTS]
    assert len(fa._FWD_SLOTS) == 29


def test_flash_attn_with_kvcache_normalisation(sglk, monkeypatch):
    from sgl_kernel import flash_attn as fa

    seen = {}

    def fake(*pos):
        seen["args"] = pos
        return (torch.zeros(1), torch.zeros(1), None, None)

    monkeypatch.setattr(fa, "_fwd", fake)
    b, s, h, d = 3, 5, 4, 64
    q = torch.randn(b, s, h, d)
    kc = torch.randn(7, 64, 2, d)
    vc = torch.randn(7, 64, 2, d)
    lens = torch.tensor([5, 9, 64], dtype=torch.int32)
    table = torch.zeros(b, 1, dtype=torch.int32)
    out = fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, page_table=table, causal=True, window_size=(7, 0))
    assert isinstance(out, torch.Tensor)  # no LSE unless asked
    a = dict(zip([n for n, _ in fa._FWD_SLOTS], seen["args"]))
    # padded 4-D q -> ragged 3-D rows with an arange cu_seqlens_q and max_seqlen_q = s (reference flash_attn.py:258-263)
    assert a["q"].shape == (b * s, h, d) and a["q"].is_contiguous()
    assert a["cu_seqlens_q"].dtype == torch.int32 and a["cu_seqlens_q"].tolist() == [0, 5, 10, 15]
    assert a["max_seqlen_q"] == s
    # cache_seqlens (lengths) rides in the cu_seqlens_k slot, same object (reference :264-266)
    assert a["cu_seqlens_k"] is lens
    # default scale = headdim^-0.5 (reference :233-236)
    assert a["softmax_scale"] == d ** -0.5
    assert a["is_causal"] is True and a["window_size_left"] == 7 and a["window_size_right"] == 0
    assert a["page_table"] is table and a["out"] is None and a["q_v"] is None
    # an int cache_seqlens becomes one int32 per cache row (reference :238-242); return_softmax_lse passes all through
    res = fa.flash_attn_with_kvcache(q[:, :1], kc[:b], vc[:b], cache_seqlens=11, return_softmax_lse=True, softmax_scale=0.25)
    assert isinstance(res, tuple) and len(res) == 4
    a = dict(zip([n for n, _ in fa._FWD_SLOTS], seen["args"]))
    assert a["cu_seqlens_k"].dtype == torch.int32 and a["cu_seqlens_k"].tolist() == [11] * b
    assert a["softmax_scale"] == 0.25 and a["max_seqlen_q"] == 1
    # ragged q given: passed through untouched
    cu = torch.tensor([0, 2, 6, 15], dtype=torch.int32)
    qr = torch.randn(15, h, d)
    fa.flash_attn_with_kvcache(qr, kc, vc, cache_seqlens=lens, page_table=table, cu_seqlens_q=cu, max_seqlen_q=9)
    a = dict(zip([n for n, _ in fa._FWD_SLOTS], seen["args"]))
    assert a["q"] is qr and a["cu_seqlens_q"] is cu and a["max_seqlen_q"] == 9
    with pytest.raises(AssertionError, match="cache_seqlens"):
        fa.flash_attn_with_kvcache(q, kc, vc)
    with pytest.raises(AssertionError, match="appending"):
        fa.flash_attn_with_kvcache(q, kc, vc, k=kc, v=vc, cache_seqlens=lens)


def test_flash_mla_wrappers(sglk, monkeypatch):
    from sgl_kernel import attention as at

    rec = _Recorder()
    monkeypatch.setattr(at, "_ops", rec)
    B, H = 2, 16
    qn = torch.randn(B, H, 512, dtype=torch.bfloat16)
    qp = torch.randn(B, H, 64, dtype=torch.bfloat16)
    cache = torch.randn(8, 64, 576, dtype=torch.bfloat16)
    lens = torch.tensor([100, 128], dtype=torch.int32)
    table = torch.zeros(B, 2, dtype=torch.int32)
    ws = torch.empty(16, dtype=torch.uint8)
    out = at.flash_mla_decode(qn, qp, cache, lens, table, ws, 0.1, -1)
    assert out.shape == (B, H, 512) and out.dtype == torch.bfloat16 and out.is_contiguous()
    name, args, _ = rec.calls[-1]
    assert name == "flash_mla_decode" and args[0] is out and args[1] is qn and args[6] is ws and args[7:] == (0.1, -1)
    # a transposed q_nope view keeps its strides (only the innermost stride must be 1)
    qn_t = torch.randn(H, B, 512, dtype=torch.bfloat16).transpose(0, 1)
    at.flash_mla_decode(qn_t, qp, cache, lens, table, ws, 0.1)
    assert rec.calls[-1][1][1] is qn_t and rec.calls[-1][1][8] == 1  # default num_kv_splits = 1
    with pytest.raises(AssertionError, match="128-token"):
        at.flash_mla_decode(qn, qp, cache, lens, torch.zeros(B, 1, dtype=torch.int32), ws, 0.1)
    with pytest.raises(AssertionError, match="int32"):
        at.flash_mla_decode(qn, qp, cache, lens.long(), table, ws, 0.1)
    with pytest.raises(AssertionError, match="<= 128"):
        at.flash_mla_decode(torch.randn(B, 129, 512, dtype=torch.bfloat16), torch.randn(B, 129, 64, dtype=torch.bfloat16),
                            cache, lens, table, ws, 0.1)
    with pytest.raises(AssertionError, match="greater than 0"):
        at.flash_mla_get_workspace_size(0, 4)
    # prefill: exactly total_q rows, nothing launched for an empty batch
    cu = torch.tensor([0, 3, 7], dtype=torch.int32)
    n = len(rec.calls)
    o = at.flash_mla_prefill(torch.randn(7, H, 512, dtype=torch.bfloat16), torch.randn(7, H, 64, dtype=torch.bfloat16),
                             cache, cu, lens, 4, table, ws, 0.1)
    assert o.shape == (7, H, 512) and rec.calls[-1][0] == "flash_mla_prefill" and rec.calls[-1][1][10:] == (True, -1)
    o = at.flash_mla_prefill(torch.randn(0, H, 512, dtype=torch.bfloat16), torch.randn(0, H, 64, dtype=torch.bfloat16),
                             cache, torch.zeros(3, dtype=torch.int32), lens, 0, table, ws, 0.1)
    assert o.shape == (0, H, 512) and len(rec.calls) == n + 1


def _moe_case(dtype=torch.bfloat16, T=5, H=128, I=64, E=4, k=2, four_bit=True, gated=True):
    x = torch.randn(T, H, dtype=dtype)
    rows1 = 2 * I if gated else I
    if four_bit:
        w1 = torch.zeros(E, rows1, H // 2, dtype=torch.uint8)
        w2 = torch.zeros(E, H, I // 2, dtype=torch.uint8)
        s1 = torch.ones(E, rows1, H // 32, dtype=dtype)
        s2 = torch.ones(E, H, I // 32, dtype=dtype)
    else:
        w1, w2, s1, s2 = torch.zeros(E, rows1, H, dtype=dtype), torch.zeros(E, H, I, dtype=dtype), None, None
    tw = torch.rand(T, k)
    ti = torch.randint(0, E, (T, k), dtype=torch.int64)
    return x, w1, w2, tw, ti, s1, s2


@pytest.mark.parametrize("applies,used", [(128, 128), (256, 256), (128, 0), (0, 0)])
def test_fused_experts_down_projection_k_split_sequence(sglk, monkeypatch, applies, used):
    """From 96 rows per expert, 4-bit weights, no b2: the down projection asks for its K split; when the op says it split (by
    returning its row block), the combine is the split form reading the workspace with that block, otherwise the plain sequence -
    and the plain GEMM is not run twice."""
    from sgl_kernel import moe

    rec = _Recorder({"moe_w4a16_splitk_applies": applies, "moe_grouped_mm_nt_w4a16_splitk": used})
    monkeypatch.setattr(moe, "_ops", rec)
    monkeypatch.setattr(moe, "is_xe2_arch", lambda: True)
    x, w1, w2, tw, ti, s1, s2 = _moe_case(T=256, E=4, k=2)  # 512 rows: 128 per expert
    out = moe.fused_experts(x, w1, w2, tw, ti, use_int4_w4a16=True, w1_scale=s1, w2_scale=s2, routed_scaling_factor=2.5)
    head = ["prepare_moe_input", "scatter_tokens_to_experts", "moe_grouped_mm_nt_w4a16_act", "moe_w4a16_splitk_applies"]
    if applies and used:
        assert rec.names() == head + ["moe_grouped_mm_nt_w4a16_splitk", "apply_shuffle_mul_sum_splitk"]
        gemm, comb = rec.calls[4][1], rec.calls[5][1]
        assert gemm[1].shape == (2, 512, 128) and gemm[1].dtype == torch.float32 and gemm[0].shape == (512, 128)
        assert comb[0] is gemm[0] and comb[1] is gemm[1] and comb[2] is out and comb[5:7] == (used, 2.5) and comb[7] is tw
        assert comb[4] is rec.calls[0][1][1]  # rows_per_expert of prepare_moe_input
    elif applies:  # (the op ran the plain GEMM itself and said so)
        assert rec.names() == head + ["moe_grouped_mm_nt_w4a16_splitk", "apply_shuffle_mul_sum"]
    else:
        assert rec.names() == head + ["moe_grouped_mm_nt_xe20_w4a16", "apply_shuffle_mul_sum"]
    assert rec.calls[3][1] == (512, 4, 128, 64, 32, True, True)  # rows, E, N = hidden, K = inter, group, int4, bf16
    # a bias on the down projection keeps the plain sequence (the split form has no bias path)
    rec.calls.clear()
    moe.fused_experts(x, w1, w2, tw, ti, b2=torch.zeros(4, 128), use_int4_w4a16=True, w1_scale=s1, w2_scale=s2)
    assert "moe_w4a16_splitk_applies" not in rec.names() and rec.names()[-2:] == ["moe_grouped_mm_nt_xe20_w4a16", "apply_shuffle_mul_sum"]


def test_fused_experts_plan_and_op_sequence(sglk, monkeypatch):
    from sgl_kernel import moe

    rec = _Recorder()
    monkeypatch.setattr(moe, "_ops", rec)
    monkeypatch.setattr(moe, "is_xe2_arch", lambda: True)
    x, w1, w2, tw, ti, s1, s2 = _moe_case()
    b1 = torch.randn(4, 128, dtype=torch.bfloat16)
    out = moe.fused_experts(x, w1, w2, tw, ti, b1=b1, use_int4_w4a16=True, w1_scale=s1, w2_scale=s2, routed_scaling_factor=2.5)
    assert out.shape == x.shape and out is not x
    # GEMM 1 carries the gate / up activation in its epilogue (authored op): no separate act-and-mul launch; at decode
    # sizes (fewer than 96 rows per expert) it also gathers its rows through a_map: no scatter launch, the tokens go in as they are
    assert rec.names() == ["prepare_moe_input", "moe_grouped_mm_nt_w4a16_act", "moe_grouped_mm_nt_xe20_w4a16",
                           "apply_shuffle_mul_sum"]
    prep = rec.calls[0][1]
    assert prep[0].dtype == torch.int32 and prep[2] is None and prep[7:] == (4, 128, 2)  # ids int32, E, hidden, topk
    g1, g2 = rec.calls[1][1], rec.calls[2][1]
    assert g1[0].shape == (10, 64) and g1[1].shape == (5, 128)  # act(gate) * up: [T*k, I] from the tokens [T, H]
    assert g1[5].dtype == torch.float32 and torch.equal(g1[5], b1.float())  # bf16 bias widened to fp32
    assert g1[7:12] == (4, True, 32, 1, 0.0) and g2[9] == 32  # E, is_int4, group size, silu, no clamp
    assert g1[12] is prep[5] and g1[12].shape == (10,) and g1[12].dtype == torch.int32  # the row map = prepare_moe_input's a_map
    assert g2[0].shape == (10, 128) and g2[1].shape == (10, 64) and g2[5] is None
    comb = rec.calls[3][1]
    assert comb[1] is out and comb[3] == 2.5 and comb[4] is tw and comb[2] is prep[6]
    # DeepSeek-V4 clamp: activation 4 with its limit
    rec.calls.clear()
    moe.fused_experts(x, w1, w2, tw, ti, use_int4_w4a16=True, w1_scale=s1, w2_scale=s2, swiglu_limit=10)
    assert rec.calls[1][1][10:12] == (4, 10.0)
    # from 96 rows per expert on (the tile pipeline's sizes) the tokens are copied expert-contiguous first, no map
    rec.calls.clear()
    xl, w1l, w2l, twl, til, s1l, s2l = _moe_case(T=192)
    moe.fused_experts(xl, w1l, w2l, twl, til, use_int4_w4a16=True, w1_scale=s1l, w2_scale=s2l)
    # (96 .. 191 rows per expert: the down projection also asks whether its K split applies; the recorder says no)
    assert rec.names() == ["prepare_moe_input", "scatter_tokens_to_experts", "moe_grouped_mm_nt_w4a16_act",
                           "moe_w4a16_splitk_applies", "moe_grouped_mm_nt_xe20_w4a16", "apply_shuffle_mul_sum"]
    assert rec.calls[2][1][1].shape == (384, 128) and rec.calls[2][1][12] is None
    # gpt-oss swiglu (reference moe.py:692-697; its fused 16-bit GEMM, moe.py:830-846): the interleaved-pairs swiglu in GEMM 1's
    # epilogue (activation 5 with limit and alpha) - no [rows, 2I] intermediate, no separate launch; at decode sizes with the gather
    rec.calls.clear()
    moe.fused_experts(x, w1, w2, tw, ti, use_int4_w4a16=True, w1_scale=s1, w2_scale=s2, gemm1_alpha=1.702, gemm1_limit=7.0)
    assert rec.names() == ["prepare_moe_input", "moe_grouped_mm_nt_w4a16_act", "moe_grouped_mm_nt_xe20_w4a16", "apply_shuffle_mul_sum"]
    g1 = rec.calls[1][1]
    assert g1[0].shape == (10, 64) and g1[10:12] == (5, 7.0) and g1[12] is rec.calls[0][1][5] and g1[13] == 1.702
    assert rec.calls[2][1][1].shape == (10, 64)
    # ... and with 16-bit weights the reference's own fused call: activation_type 2, fuse_act, alpha, limit
    rec.calls.clear()
    xb, w1b, w2b, twb, tib, _, _ = _moe_case(four_bit=False)
    moe.fused_experts(xb, w1b, w2b, twb, tib, gemm1_alpha=1.702, gemm1_limit=7.0)
    assert rec.names() == ["prepare_moe_input", "scatter_tokens_to_experts", "moe_grouped_mm_nt_xe20", "moe_grouped_mm_nt_xe20",
                           "apply_shuffle_mul_sum"]
    assert rec.calls[2][1][0].shape == (10, 64) and rec.calls[2][1][6:] == (2, True, 1.702, 7.0)
    with pytest.raises(AssertionError, match="gemm1_limit must be provided"):
        moe.fused_experts(x, w1, w2, tw, ti, use_int4_w4a16=True, w1_scale=s1, w2_scale=s2, gemm1_alpha=1.702)
    with pytest.raises(AssertionError, match="activation must be silu"):
        moe.fused_experts(x, w1, w2, tw, ti, activation="gelu", use_int4_w4a16=True, w1_scale=s1, w2_scale=s2,
                          gemm1_alpha=1.702, gemm1_limit=7.0)
    # relu2: no gate (gate_factor 1): relu^2 in the same epilogue; inplace writes over the input
    rec.calls.clear()
    x, w1, w2, tw, ti, s1, s2 = _moe_case(gated=False)
    out = moe.fused_experts(x, w1, w2, tw, ti, activation="relu2", inplace=True, use_int4_w4a16=True, w1_scale=s1, w2_scale=s2)
    assert out is x and rec.calls[1][1][0].shape == (10, 64) and rec.calls[1][1][10] == 3 and "silu_and_mul" not in rec.names()
    assert rec.calls[-1][1][3] == 1.0
    # 16-bit weights: the grouped GEMM's fused gate/up epilogue, GEMM 2 without it
    rec.calls.clear()
    x, w1, w2, tw, ti, _, _ = _moe_case(four_bit=False)
    moe.fused_experts(x, w1, w2, tw, ti, activation="gelu")
    assert rec.names() == ["prepare_moe_input", "scatter_tokens_to_experts", "moe_grouped_mm_nt_xe20", "moe_grouped_mm_nt_xe20",
                           "apply_shuffle_mul_sum"]
    assert rec.calls[2][1][0].shape == (10, 64) and rec.calls[2][1][6:8] == (1, True) and rec.calls[3][1][7] is False
    # argument errors
    with pytest.raises(AssertionError, match="mutually exclusive"):
        moe.fused_experts(x, w1, w2, tw, ti, use_int4_w4a16=True, use_mxfp4_w4a16=True)
    with pytest.raises(AssertionError, match="w1_scale"):
        x4, w14, w24, tw4, ti4, s14, s24 = _moe_case()
        moe.fused_experts(x4, w14, w24, tw4, ti4, use_int4_w4a16=True)
    with pytest.raises(AssertionError, match="Only silu"):
        moe.fused_experts(x, w1, w2, tw, ti, activation="tanh")
    with pytest.raises(AssertionError, match="use_fp8_w8a8"):
        moe.fused_experts(x, w1, w2, tw, ti, use_fp8_w8a8=True)


def test_moe_scratch_is_grow_only_and_stable(sglk):
    from sgl_kernel import moe

    dev = torch.device("cpu")
    a = moe._get_moe_ws("t_probe", (10, 8), torch.float32, dev)
    b = moe._get_moe_ws("t_probe", (5, 8), torch.float32, dev)
    assert b.data_ptr() == a.data_ptr() and b.shape == (5, 8)  # a smaller request re-uses the same storage
    c = moe._get_moe_ws("t_probe", (10, 8), torch.float32, dev)
    assert c.data_ptr() == a.data_ptr()
    big = moe._get_moe_ws("t_probe", (100, 8), torch.float32, dev)
    assert big.untyped_storage().nbytes() >= int(800 * 1.1) * 4  # grown with 10 % headroom
    again = moe._get_moe_ws("t_probe", (100, 8), torch.float32, dev)
    assert again.data_ptr() == big.data_ptr()
    other = moe._get_moe_ws("t_probe", (4,), torch.int32, dev)  # another dtype replaces the buffer
    assert other.dtype == torch.int32
    assert moe._get_moe_ws("t_other", (4,), torch.int32, dev).data_ptr() != other.data_ptr()
