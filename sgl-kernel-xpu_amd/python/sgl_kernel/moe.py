"""Mixture-of-experts host layer: thin op wrappers and the `fused_experts` routed-expert MLP.

Contract (reference python/sgl_kernel/moe.py): public names and parameters of `moe_align_block_size` (:48-67),
`topk_softmax` (:70-78), `topk_sigmoid` (:81-98), `moe_fused_gate` (:159-201), `biased_topk` (:204-233), `prepare_moe_input` (:278-301), `apply_shuffle_mul_sum` (:304-318),
`scatter_tokens_to_experts` (:321-322) and `fused_experts` (:403-432); the op sequence
prepare_moe_input -> scatter_tokens_to_experts -> grouped GEMM 1 -> gate/up activation -> grouped GEMM 2 ->
apply_shuffle_mul_sum (:655-868); the grow-only scratch buffers keyed by (name, device) with 10 % headroom
(:372-400); gate/up detection from the UNPACKED inner size of w2 (:723); bf16 biases up-cast to fp32 (:574-587);
`rows_per_expert` (the tensor the reference calls expert_offsets) holds COUNTS.

This build's own structure: one `_Plan` derives every size and route from the arguments, `_Scratch` owns the
buffers, the activation routes are a table. 4-bit (int4 / mxfp4 W4A16) and 16-bit weights alike take GEMM 1 with the
gate / up product (silu, gelu, the DeepSeek-V4 clamped swiglu) or relu2 in its epilogue, then GEMM 2; only the gpt-oss
swiglu (gate / up interleaved in w1's rows) runs the reference's three steps GEMM -> swiglu_gpt_oss_sigmoid_alpha -> GEMM.
"""
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from .utils import is_xe2_arch

_ops = torch.ops.sgl_kernel  # (patched by the host-logic tests)


# ------------------------------------------------------------------------------------------------ op wrappers

def moe_align_block_size(topk_ids, num_experts, block_size, sorted_token_ids, experts_ids, num_tokens_post_pad,
                         cumsum_buffer, pad_sorted_token_ids=False):
    _ops.moe_align_block_size.default(topk_ids, num_experts, block_size, sorted_token_ids, experts_ids,
                                      num_tokens_post_pad, cumsum_buffer, pad_sorted_token_ids)


def topk_softmax(topk_weights: torch.Tensor, topk_ids: torch.Tensor, gating_output: float, renormalize: bool = False) -> None:
    _ops.topk_softmax.default(topk_weights, topk_ids, gating_output, renormalize)


_GATE_SCORING = {"sigmoid": 0, "softmax": 1}          # moe_fused_gate
_BIASED_TOPK_SCORING = {"sigmoid": 0, "sqrtsoftplus": 1}


def _scoring_code(table, scoring_func):
    code = table.get(scoring_func.lower())
    if code is None:
        raise ValueError(f"Unknown scoring_func '{scoring_func}', must be one of {list(table.keys())}")
    return code


def topk_sigmoid(topk_weights: torch.Tensor, topk_ids: torch.Tensor, gating_output: torch.Tensor, renormalize: bool = False,
                 correction_bias: Optional[torch.Tensor] = None, routed_scaling_factor: float = 1.0,
                 num_fused_shared_experts: int = 0) -> None:
    """Top-k of sigmoid(gating) (+ correction_bias for the choice only); weights / ids are written in place
    ([tokens, topk] fp32 / int32). The last num_fused_shared_experts slots name the fused shared expert."""
    _ops.topk_sigmoid.default(topk_weights, topk_ids, gating_output, renormalize, correction_bias, routed_scaling_factor,
                              num_fused_shared_experts)


def moe_fused_gate(input_tensor, bias: Optional[torch.Tensor], num_expert_group, topk_group, topk, renormalize=True,
                   scoring_func="sigmoid", num_fused_shared_experts=0, routed_scaling_factor=0,
                   apply_routed_scaling_factor_on_output=False):
    """DeepSeek-V3 grouped top-k: expert groups are ranked by the sum of their two best (biased) scores, the topk_group
    best groups stay eligible, then top-k over their experts. Returns [weights fp32 [tokens, topk], ids int32]."""
    return _ops.moe_fused_gate.default(input_tensor, bias, num_expert_group, topk_group, topk, num_fused_shared_experts,
                                       _scoring_code(_GATE_SCORING, scoring_func), renormalize, routed_scaling_factor,
                                       apply_routed_scaling_factor_on_output)


def biased_topk(input_tensor, bias, output, indices, topk, scoring_func, num_fused_shared_experts=0, renormalize=False,
                routed_scaling_factor=1.0, apply_routed_scaling_factor_on_output=False):
    """Top-k of score(input) + bias with score = sigmoid or sqrt(softplus); weights are the unbiased scores."""
    _ops.biased_topk.default(input_tensor, bias, output, indices, topk, _scoring_code(_BIASED_TOPK_SCORING, scoring_func),
                             num_fused_shared_experts, renormalize, routed_scaling_factor, apply_routed_scaling_factor_on_output)


def prepare_moe_input(topk_ids, expert_offsets, problem_sizes1, problem_sizes2, input_permutation, output_permutation,
                      num_experts, n, k, blockscale_offsets: Optional[torch.Tensor] = None):
    _ops.prepare_moe_input.default(topk_ids, expert_offsets, blockscale_offsets, problem_sizes1, problem_sizes2,
                                   input_permutation, output_permutation, num_experts, n, k)


def apply_shuffle_mul_sum(input, output, permutation, factors, routed_scaling_factor: Optional[float] = None):
    _ops.apply_shuffle_mul_sum.default(input, output, permutation,
                                       1.0 if routed_scaling_factor is None else routed_scaling_factor, factors)


def swiglu_gpt_oss_sigmoid_alpha(x, gemm1_alpha, gemm1_limit):
    """x [B, 2H] with gate / up interleaved -> [B, H]: gate = min(gate, limit), up = clamp(up, -limit, limit),
    gate * sigmoid(alpha * gate) * (up + 1) (reference moe.py:136-146)."""
    assert gemm1_limit > 0, f"gemm1_limit must be positive, got {gemm1_limit}"
    assert x.dim() == 2, f"x must be 2D [B, 2H], got {x.dim()}D"
    assert x.size(1) % 2 == 0, f"Last dim must be even for gate/up split, got {x.size(1)}"
    return _ops.swiglu_gpt_oss_sigmoid_alpha.default(x, gemm1_alpha, gemm1_limit)


def scatter_tokens_to_experts(input, src2dst_map, output):
    _ops.scatter_tokens_to_experts.default(input, src2dst_map, output)


# ------------------------------------------------------------------------------------------------ scratch

class _Scratch:
    """Process-wide grow-only flat buffers, one per (name, device). A request is a view of the first `numel`
    elements; a buffer is replaced (10 % larger than asked) only when it is too small or of another dtype, so the
    same storage serves every MoE layer and every call and the caching allocator does not collect differently
    sized blocks. Replacing a buffer is safe without a sync: the allocator orders the free on the stream."""

    HEADROOM = 1.1

    def __init__(self):
        self._buffers: Dict[Tuple[str, torch.device], torch.Tensor] = {}

    def get(self, name: str, shape: tuple, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
        numel = 1
        for extent in shape:
            numel *= int(extent)
        key = (name, device)
        flat = self._buffers.get(key)
        if flat is None or flat.dtype != dtype or flat.numel() < numel:
            flat = torch.empty(max(numel, int(numel * self.HEADROOM)), dtype=dtype, device=device)
            self._buffers[key] = flat
        return flat[:numel].view(shape)


_scratch = _Scratch()


def _get_moe_ws(name: str, shape: tuple, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    """The reference's name for a scratch request (moe.py:376)."""
    return _scratch.get(name, shape, dtype, device)


# ------------------------------------------------------------------------------------------------ plan

# activation name -> (activation_type of the grouped GEMM op, gated?, elementwise op used on the unfused route)
_ACTIVATIONS = {
    "silu": (0, True, "silu_and_mul"),
    "gelu": (1, True, "gelu_tanh_and_mul"),
    "relu2": (3, False, None),
}


# average rows per expert from which the grouped GEMMs run on the tile pipeline (csrc/moe_persist.hip: kMinAvgRows128), which stages
# expert-contiguous rows: below it GEMM 1 gathers its rows itself, from it on the tokens are copied expert-contiguous first
_TILE_PIPELINE_MIN_ROWS = 88


@dataclass(frozen=True)
class _Plan:
    tokens: int
    hidden: int        # H: GEMM 1 contracts over it, GEMM 2 produces it
    inter: int         # I: GEMM 2 contracts over it (unpacked)
    experts: int
    topk: int
    gate_factor: int   # 2: w1 holds gate and up halves; 1: no gate (relu2)
    four_bit: bool
    int4: bool
    group1: int        # quantisation group along H (GEMM 1) / along I (GEMM 2); 0 for 16-bit weights
    group2: int
    act_type: int
    act_op: Optional[str]

    @property
    def rows(self) -> int:
        return self.tokens * self.topk


def _plan(hidden_states, w1, w2, topk_weights, topk_ids, activation, use_mxfp4_w4a16, use_int4_w4a16,
          w1_scale, w2_scale) -> _Plan:
    assert activation in _ACTIVATIONS, f"Only silu, gelu and relu2 are supported but got {activation}"
    act_type, _, act_op = _ACTIVATIONS[activation]
    four_bit = use_mxfp4_w4a16 or use_int4_w4a16
    unpack = 2 if four_bit else 1  # codes per stored byte
    assert hidden_states.ndim == 2, "hidden_states must be 2D"
    tokens, hidden = hidden_states.shape
    experts, w1_rows, w1_inner = w1.shape
    experts2, w2_rows, w2_inner = w2.shape
    w1_inner *= unpack
    w2_inner *= unpack
    assert experts == experts2, f"w1 and w2 disagree on the number of experts ({experts} vs {experts2})"
    assert hidden == w1_inner, (
        f"hidden_states shape[-1] {tuple(hidden_states.shape)} must equal w1 inner dim {w1_inner} (w1.shape={tuple(w1.shape)})")
    gated = w1_rows == 2 * w2_inner
    assert gated or (w1_rows == w2_inner and activation == "relu2"), (
        f"w2 inner dim {w2_inner} must be half of w1 shape[1] {w1_rows} except non-gate")
    assert topk_ids.shape == topk_weights.shape and topk_ids.shape[0] == tokens, (
        f"topk_ids shape {tuple(topk_ids.shape)} and topk_weights shape {tuple(topk_weights.shape)} must be equal "
        f"and match hidden_states shape[0] {tokens}")
    return _Plan(tokens=tokens, hidden=hidden, inter=w2_inner, experts=experts, topk=topk_ids.shape[1],
                 gate_factor=2 if gated else 1, four_bit=four_bit, int4=bool(use_int4_w4a16),
                 group1=(hidden // w1_scale.shape[2]) if four_bit else 0,
                 group2=(w2_inner // w2_scale.shape[2]) if four_bit else 0,
                 act_type=act_type, act_op=act_op)


def _check_weight_format(hidden_states, w1, w2, use_mxfp4_w4a16, use_int4_w4a16, w1_scale, w2_scale, w1_zp, w2_zp):
    assert not (use_mxfp4_w4a16 and use_int4_w4a16), "use_mxfp4_w4a16 and use_int4_w4a16 are mutually exclusive"
    act_dtype = hidden_states.dtype
    if not (use_mxfp4_w4a16 or use_int4_w4a16):
        assert w1.dtype == act_dtype and w2.dtype == act_dtype, "w1 / w2 must have the dtype of hidden_states"
        assert w1_scale is None and w2_scale is None, "w1_scale / w2_scale are only supported for 4-bit W4A16 MoE"
        assert w1_zp is None and w2_zp is None, "w1_zp/w2_zp are only supported for 4-bit W4A16 MoE"
        return
    for name, w, s in (("w1", w1, w1_scale), ("w2", w2, w2_scale)):
        assert w.dtype in (torch.int8, torch.uint8), f"4-bit W4A16 requires {name} to be int8 or uint8 (packed [E, N, K/2])"
        assert s is not None, f"{name}_scale must be provided for 4-bit W4A16"
    if use_int4_w4a16:
        assert w1_scale.dtype == act_dtype and w2_scale.dtype == act_dtype, "int4 scales dtype must match hidden_states dtype"
        for name, zp, s in (("w1_zp", w1_zp, w1_scale), ("w2_zp", w2_zp, w2_scale)):
            if zp is not None:
                assert zp.dtype == s.dtype and zp.shape == s.shape, f"{name} must have the same dtype and shape as its scale"
    else:
        e8m0 = getattr(torch, "float8_e8m0fnu", torch.uint8)
        assert w1_scale.dtype in (torch.uint8, e8m0) and w2_scale.dtype in (torch.uint8, e8m0), (
            "mxfp4 scales must be E8M0 bytes (uint8 or float8_e8m0fnu)")
        assert w1_zp is None and w2_zp is None, "mxfp4 weights have no zero points"


def _bias_fp32(bias, name, expect):
    """The kernels add the bias in fp32 and take it as fp32: bf16 biases are widened once here."""
    if bias is None:
        return None
    assert bias.dtype in (torch.bfloat16, torch.float32), f"{name} must be bfloat16 or float32"
    assert tuple(bias.shape) == tuple(expect), f"{name} shape must match w{name[1]} shape[:2]"
    return bias.float() if bias.dtype != torch.float32 else bias


def _gather_channels_per_expert(x, perm, rows_per_expert, num_experts):
    """GPTQ act-order support: row r of expert e becomes x[r, perm[e, :]] (rows are grouped by expert, counts given)."""
    owner = torch.repeat_interleave(torch.arange(num_experts, device=x.device), rows_per_expert.to(torch.int64),
                                    output_size=x.size(0))
    return torch.gather(x, 1, perm[owner])


# ------------------------------------------------------------------------------------------------ fused_experts

def fused_experts(hidden_states: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, topk_weights: torch.Tensor,
                  topk_ids: torch.Tensor, b1: Optional[torch.Tensor] = None, b2: Optional[torch.Tensor] = None,
                  inplace: bool = False, activation: str = "silu", use_fp8_w8a8: bool = False,
                  use_mxfp4_w4a16: bool = False, use_int4_w4a16: bool = False,
                  w1_scale: Optional[torch.Tensor] = None, w2_scale: Optional[torch.Tensor] = None,
                  w1_zp: Optional[torch.Tensor] = None, w2_zp: Optional[torch.Tensor] = None,
                  w1_g_idx_perm: Optional[torch.Tensor] = None, w2_g_idx_perm: Optional[torch.Tensor] = None,
                  a1_scale: Optional[torch.Tensor] = None, a2_scale: Optional[torch.Tensor] = None,
                  block_shape: Optional[list] = None, no_combine: bool = False,
                  routed_scaling_factor: Optional[float] = None, gemm1_alpha: Optional[float] = None,
                  gemm1_limit: Optional[float] = None, swiglu_limit: Optional[float] = None) -> torch.Tensor:
    """out[t] = sum_j topk_weights[t, j] * MLP_{topk_ids[t, j]}(hidden_states[t]) (times routed_scaling_factor).

    Weights: 16-bit w1 [E, 2I, H] (gate rows first, then up; [E, I, H] for relu2) and w2 [E, H, I] in the
    activation dtype, or 4-bit W4A16: the same matrices packed two codes per byte along the last axis (low nibble
    = even index), int4 with scales [E, rows, cols/group] in the activation dtype (+ optional raw zero points of
    the same shape; without them the codes are two's-complement), mxfp4 (e2m1) with E8M0 scale bytes, group 32.
    Biases b1 [E, 2I] / b2 [E, H] are added in fp32. `inplace` writes the result over hidden_states."""
    assert not use_fp8_w8a8, "current MoE does not support use_fp8_w8a8"
    assert a1_scale is None and a2_scale is None, "current MoE does not support a1_scale / a2_scale"
    assert block_shape is None, "current MoE does not support block_shape"
    if gemm1_alpha is not None:  # gpt-oss swiglu (reference moe.py:692-697): silu with alpha / limit, interleaved gate / up
        assert activation == "silu", "gemm1_alpha selects the gpt-oss swiglu: activation must be silu"
        assert gemm1_limit is not None, "gemm1_limit must be provided when gemm1_alpha is set for swiglu for GPT-OSS"
        assert swiglu_limit is None, "gemm1_alpha and swiglu_limit exclude each other"
    if swiglu_limit is not None:  # DeepSeek-V4 clamp (reference moe.py:699-709): silu only, 4-bit weights only
        assert activation == "silu" and swiglu_limit == 10
        assert use_mxfp4_w4a16 or use_int4_w4a16, "swiglu_limit requires use_mxfp4_w4a16=True or use_int4_w4a16=True"
    assert is_xe2_arch(), "this MoE path is built for gfx950 (MI355X) only"
    if w1_g_idx_perm is not None or w2_g_idx_perm is not None:
        assert use_int4_w4a16, "w1_g_idx_perm/w2_g_idx_perm only apply to use_int4_w4a16"

    _check_weight_format(hidden_states, w1, w2, use_mxfp4_w4a16, use_int4_w4a16, w1_scale, w2_scale, w1_zp, w2_zp)
    p = _plan(hidden_states, w1, w2, topk_weights, topk_ids, activation, use_mxfp4_w4a16, use_int4_w4a16, w1_scale, w2_scale)
    b1 = _bias_fp32(b1, "b1", w1.shape[:2])
    b2 = _bias_fp32(b2, "b2", w2.shape[:2])

    dev, dt = hidden_states.device, hidden_states.dtype
    if no_combine:
        assert not inplace, "no_combine and inplace exclude each other"
        result = torch.empty((p.tokens, p.hidden), device=dev, dtype=dt)
    else:
        result = hidden_states if inplace else torch.empty_like(hidden_states)

    def scratch(name, shape, dtype=dt):
        return _scratch.get(name, shape, dtype, dev)

    # ---- routing: expert row counts and the token -> expert-contiguous row map
    ids = (topk_ids.int() if topk_ids.dtype == torch.long else topk_ids).contiguous()
    rows_per_expert = scratch("expert_offsets", (p.experts,), torch.int32)
    src_rows = scratch("a_map", (p.rows,), torch.int32)
    dst_rows = scratch("c_map", (p.rows,), torch.int32)
    _ops.prepare_moe_input.default(ids, rows_per_expert, None,
                                   scratch("problem_sizes1", (p.experts, 3), torch.int32),
                                   scratch("problem_sizes2", (p.experts, 3), torch.int32),
                                   src_rows, dst_rows, p.experts, p.hidden, p.topk)
    # decode sizes with 4-bit weights: GEMM 1 gathers its rows through a_map itself (the streaming kernels' staging loads),
    # no [rows, hidden] copy of the tokens and one launch less. From 88 rows per expert on the tile pipeline takes the
    # GEMM, which stages expert-contiguous rows by LDS-DMA: the copy (reference shuffle_rows, moe.py:739) stays there.
    gather_in_gemm1 = (p.four_bit and w1_g_idx_perm is None and p.rows < _TILE_PIPELINE_MIN_ROWS * p.experts
                       and p.tokens * p.hidden < 2**32)
    if gather_in_gemm1:
        x = hidden_states.contiguous()
    else:
        x = scratch("input_A_shuffle", (p.rows, p.hidden))
        _ops.scatter_tokens_to_experts.default(hidden_states.contiguous(), dst_rows, x)
    if w1_g_idx_perm is not None:
        x = _gather_channels_per_expert(x, w1_g_idx_perm, rows_per_expert, p.experts)

    def grouped_mm(out, a, w, scale, zp, bias, group, fuse_act=False):
        if p.four_bit:
            _ops.moe_grouped_mm_nt_xe20_w4a16(out, a, w, scale, zp, bias, rows_per_expert, p.experts, p.int4, group)
        else:
            _ops.moe_grouped_mm_nt_xe20(out, a, w, bias, rows_per_expert, p.experts, p.act_type, fuse_act, 1.702, 7.0)

    # ---- GEMM 1 with the gate / up activation (or relu2) on its fp32 accumulators, in the epilogue: no [rows, 2I]
    # intermediate, no separate act-and-mul launch (the reference runs them as two launches, moe.py:751-835)
    h = scratch("intermediate_cache1_fused", (p.rows, p.inter))
    if gemm1_alpha is not None:
        # gpt-oss: the rows of w1 are (gate, up) PAIRS; the swiglu runs in the GEMM's epilogue on the fp32 accumulators, as the
        # reference's fused 16-bit GEMM does it (moe_grouped_mm_nt_xe20 with activation_type 2 and fuse_act, moe.py:830-846;
        # kernels/moe/xe20/bf16/moe_kernel.hpp:109-125) - for 4-bit weights too, where the reference writes [rows, 2I] and calls
        # swiglu_gpt_oss_sigmoid_alpha on it (moe.py:748-789): no [rows, 2I] round trip through HBM, one launch less (graph-timed,
        # Mixtral-sized experts, GEMM 1 alone: 111 against 123 us at 16 tokens, 135 / 137 at 64, 362 / 381 at 512, 1.80 / 1.94 ms at 4096)
        if p.four_bit:
            _ops.moe_grouped_mm_nt_w4a16_act(h, x, w1, w1_scale, w1_zp, b1, rows_per_expert, p.experts, p.int4, p.group1, 5,
                                             float(gemm1_limit), src_rows if gather_in_gemm1 else None, float(gemm1_alpha))
        else:
            _ops.moe_grouped_mm_nt_xe20(h, x, w1, b1, rows_per_expert, p.experts, 2, True, float(gemm1_alpha), float(gemm1_limit))
    elif not p.four_bit:
        grouped_mm(h, x, w1, None, None, b1, 0, fuse_act=True)
    else:
        fused = 4 if swiglu_limit is not None else {0: 1, 1: 2, 3: 3}[p.act_type]
        _ops.moe_grouped_mm_nt_w4a16_act(h, x, w1, w1_scale, w1_zp, b1, rows_per_expert, p.experts, p.int4, p.group1,
                                         fused, float(swiglu_limit or 0.0), src_rows if gather_in_gemm1 else None)
    if w2_g_idx_perm is not None:
        h = _gather_channels_per_expert(h, w2_g_idx_perm, rows_per_expert, p.experts)

    # ---- GEMM 2 and the weighted combine over the top-k slots
    y = scratch("intermediate_cache3", (p.rows, p.hidden))
    rsf = 1.0 if routed_scaling_factor is None else routed_scaling_factor
    # 4-bit weights, from 88 rows per expert: where the down projection has fewer tiles than the GPU has CUs (Mixtral at 512 tokens:
    # 128 tiles of 128 x 256 and 224 K blocks on 256 CUs; at 1024 tokens 128 tiles of 256 x 256), the K range of every tile is
    # split over two workgroups; the two fp32
    # partial sums go to a workspace and the combine below adds them (a two-term sum: the same in either order) and rounds once,
    # as the GEMM's own store would have (the reference picks a tile policy per average row count, GroupGemmW4A16Xe20.cpp:266-277)
    # (the op answers for the tile counts and the group alignment; the row-count regime is checked here so that other sizes make
    # no extra call)
    if (p.four_bit and b2 is None and p.hidden % 8 == 0 and _TILE_PIPELINE_MIN_ROWS * p.experts <= p.rows
            and _ops.moe_w4a16_splitk_applies(p.rows, p.experts, p.hidden, p.inter, p.group2, p.int4, dt == torch.bfloat16)):
        ws = scratch("splitk_partials", (2, p.rows, p.hidden), torch.float32)
        block = _ops.moe_grouped_mm_nt_w4a16_splitk(y, ws, h.contiguous(), w2, w2_scale, w2_zp, rows_per_expert, p.experts,
                                                     p.int4, p.group2)
        if block:  # (the row block of the split: 128 or 256; 0 = the op ran the plain GEMM into y)
            _ops.apply_shuffle_mul_sum_splitk.default(y, ws, result, dst_rows, rows_per_expert, block, rsf, topk_weights)
            return result
    else:
        grouped_mm(y, h.contiguous(), w2, w2_scale, w2_zp, b2, p.group2)
    _ops.apply_shuffle_mul_sum.default(y, result, dst_rows, 1.0 if routed_scaling_factor is None else routed_scaling_factor,
                                       topk_weights)
    return result
