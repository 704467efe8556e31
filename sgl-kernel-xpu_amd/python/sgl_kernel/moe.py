"""MoE wrappers and the fused_experts orchestration for the W4A16 (int4) path.

Mirrors reference python/sgl_kernel/moe.py: moe_align_block_size :48-67, topk_softmax :70-78,
prepare_moe_input :278-301, apply_shuffle_mul_sum :304-318, scatter_tokens_to_experts :321-322,
the grow-only workspace cache :372-400 and fused_experts :403-870 (same arguments, asserts and op
sequence: prepare_moe_input -> scatter_tokens_to_experts -> grouped GEMM1 -> activation ->
grouped GEMM2 -> apply_shuffle_mul_sum). Built for gfx950: the 4-bit int4 path named by the
north star; the mxfp4 format and the bf16-weight grouped GEMM (`moe_grouped_mm_nt_xe20`) are
outside this build and raise NotImplementedError.
"""
from typing import Dict, Optional, Tuple

import torch

from .utils import is_xe2_arch


def _apply_per_expert_channel_gather(
    x: torch.Tensor,
    perm: torch.Tensor,
    rows_per_expert: torch.Tensor,
    num_experts: int,
) -> torch.Tensor:
    """GPTQ desc_act/g_idx support (reference moe.py:18-45): out[:, c'] = x[:, perm[e, c']] for the
    row block of expert e; rows_per_expert holds counts, rows are already grouped by expert."""
    expert_ids = torch.repeat_interleave(
        torch.arange(num_experts, device=x.device),
        rows_per_expert.to(torch.int64),
        output_size=x.size(0),
    )
    return x.gather(1, perm.index_select(0, expert_ids))


def moe_align_block_size(
    topk_ids,
    num_experts,
    block_size,
    sorted_token_ids,
    experts_ids,
    num_tokens_post_pad,
    cumsum_buffer,
    pad_sorted_token_ids=False,
):
    torch.ops.sgl_kernel.moe_align_block_size.default(
        topk_ids,
        num_experts,
        block_size,
        sorted_token_ids,
        experts_ids,
        num_tokens_post_pad,
        cumsum_buffer,
        pad_sorted_token_ids,
    )


def topk_softmax(
    topk_weights: torch.Tensor,
    topk_ids: torch.Tensor,
    gating_output: float,
    renormalize: bool = False,
) -> None:
    torch.ops.sgl_kernel.topk_softmax.default(topk_weights, topk_ids, gating_output, renormalize)


def prepare_moe_input(
    topk_ids,
    expert_offsets,
    problem_sizes1,
    problem_sizes2,
    input_permutation,
    output_permutation,
    num_experts,
    n,
    k,
    blockscale_offsets: Optional[torch.Tensor] = None,
):
    torch.ops.sgl_kernel.prepare_moe_input.default(
        topk_ids,
        expert_offsets,
        blockscale_offsets,
        problem_sizes1,
        problem_sizes2,
        input_permutation,
        output_permutation,
        num_experts,
        n,
        k,
    )


def apply_shuffle_mul_sum(
    input,
    output,
    permutation,
    factors,
    routed_scaling_factor: Optional[float] = None,
):
    rsf = 1.0

    if routed_scaling_factor is not None:
        rsf = routed_scaling_factor

    torch.ops.sgl_kernel.apply_shuffle_mul_sum.default(input, output, permutation, rsf, factors)


def scatter_tokens_to_experts(input, src2dst_map, output):
    torch.ops.sgl_kernel.scatter_tokens_to_experts.default(input, src2dst_map, output)


_MOE_WS_HEADROOM = 1.1
_moe_ws_cache: Dict[Tuple[str, torch.device], torch.Tensor] = {}


def _get_moe_ws(
    name: str,
    shape: tuple,
    dtype: torch.dtype,
    device: torch.device,
) -> torch.Tensor:
    """A tensor of `shape`/`dtype` backed by a process-wide grow-only flat scratch buffer
    (reference moe.py:376-400): stable buffers across calls and MoE layers keep the caching
    allocator from piling up differently-shaped blocks. Dropping the old buffer is stream-ordered."""
    numel = 1
    for d in shape:
        numel *= d
    key = (name, device)
    cur = _moe_ws_cache.get(key)
    if cur is None or cur.numel() < numel or cur.dtype != dtype:
        new_numel = max(numel, int(numel * _MOE_WS_HEADROOM))
        cur = torch.empty(new_numel, dtype=dtype, device=device)
        _moe_ws_cache[key] = cur
    return cur.narrow(0, 0, numel).view(shape)


def fused_experts(
    hidden_states: torch.Tensor,
    w1: torch.Tensor,
    w2: torch.Tensor,
    topk_weights: torch.Tensor,
    topk_ids: torch.Tensor,
    b1: Optional[torch.Tensor] = None,
    b2: Optional[torch.Tensor] = None,
    inplace: bool = False,
    activation: str = "silu",
    use_fp8_w8a8: bool = False,
    use_mxfp4_w4a16: bool = False,
    use_int4_w4a16: bool = False,
    w1_scale: Optional[torch.Tensor] = None,
    w2_scale: Optional[torch.Tensor] = None,
    w1_zp: Optional[torch.Tensor] = None,
    w2_zp: Optional[torch.Tensor] = None,
    w1_g_idx_perm: Optional[torch.Tensor] = None,
    w2_g_idx_perm: Optional[torch.Tensor] = None,
    a1_scale: Optional[torch.Tensor] = None,
    a2_scale: Optional[torch.Tensor] = None,
    block_shape: Optional[list] = None,
    no_combine: bool = False,
    routed_scaling_factor: Optional[float] = None,
    gemm1_alpha: Optional[float] = None,
    gemm1_limit: Optional[float] = None,
    swiglu_limit: Optional[float] = None,
) -> torch.Tensor:
    """Routed-expert MLP. Default: 16-bit weights w1 [E, 2I (or I for relu2), H], w2 [E, H, I] in the activation
    dtype. use_int4_w4a16: w1 [E, 2I (or I for relu2), H/2] and w2 [E, H, I/2] are int4-packed
    (two codes per byte, low nibble = even k) with per-group scales [E, N, K/group] in the
    activation dtype and optional raw zero-points of the same shape; topk_weights fp32 [T, k];
    biases fp32 (bf16 is up-cast, as the reference does: moe.py:574-587). Arguments and their
    meaning are the reference's (moe.py:403-508)."""
    assert use_fp8_w8a8 is False, "current MoE does not support use_fp8_w8a8"
    assert a1_scale is None, "current MoE does not support a1_scale"
    assert a2_scale is None, "current MoE does not support a2_scale"
    assert block_shape is None, "current MoE does not support block_shape"
    assert activation in (
        "silu",
        "gelu",
        "relu2",
    ), f"Only silu, gelu and relu2 are supported but got {activation}"

    use_4bit_w4a16 = use_mxfp4_w4a16 or use_int4_w4a16
    assert not (use_mxfp4_w4a16 and use_int4_w4a16), "use_mxfp4_w4a16 and use_int4_w4a16 are mutually exclusive"
    if use_mxfp4_w4a16:
        raise NotImplementedError("fused_experts: the mxfp4 weight format is outside the MI355X build (int4 W4A16 only)")
    if use_4bit_w4a16:
        assert w1.dtype == torch.int8 or w1.dtype == torch.uint8, "4-bit W4A16 requires w1 to be int8 or uint8 (packed [E, N, K/2])"
        assert w2.dtype == torch.int8 or w2.dtype == torch.uint8, "4-bit W4A16 requires w2 to be int8 or uint8 (packed [E, N, K/2])"
        assert w1_scale is not None, "w1_scale must be provided for 4-bit W4A16"
        assert w2_scale is not None, "w2_scale must be provided for 4-bit W4A16"
        assert (
            w1_scale.dtype == hidden_states.dtype and w2_scale.dtype == hidden_states.dtype
        ), "int4 scales dtype must match hidden_states dtype"
        if w1_zp is not None:
            assert w1_zp.dtype == w1_scale.dtype and w1_zp.shape == w1_scale.shape, "w1_zp must have the same dtype and shape as w1_scale"
        if w2_zp is not None:
            assert w2_zp.dtype == w2_scale.dtype and w2_zp.shape == w2_scale.shape, "w2_zp must have the same dtype and shape as w2_scale"
    else:
        # 16-bit weights: w1 [E, 2I (or I for relu2), H], w2 [E, H, I] in the activation dtype (reference moe.py:763-775)
        assert w1.dtype == hidden_states.dtype and w2.dtype == hidden_states.dtype, "w1 / w2 must have the dtype of hidden_states"
        assert w1_scale is None and w2_scale is None and w1_zp is None and w2_zp is None, "scales / zero points need a 4-bit format"
    if b1 is not None:
        assert b1.dtype == torch.bfloat16 or b1.dtype == torch.float32, "b1 must be bfloat16 or float32"
        if b1.dtype == torch.bfloat16:
            b1 = b1.float()  # bias is accumulated in float32 in the kernel
    if b2 is not None:
        assert b2.dtype == torch.bfloat16 or b2.dtype == torch.float32, "b2 must be bfloat16 or float32"
        if b2.dtype == torch.bfloat16:
            b2 = b2.float()

    _pack = 2 if use_4bit_w4a16 else 1  # codes per stored element
    _w1_inner = w1.shape[-1] * _pack
    _w2_inner = w2.shape[-1] * _pack
    assert hidden_states.ndim == 2, "hidden_states must be 2D"
    assert (
        hidden_states.shape[-1] == _w1_inner
    ), f"hidden_states shape[-1] {hidden_states.shape} must equal w1 inner dim {_w1_inner} (w1.shape={w1.shape})"
    assert (2 * _w2_inner == w1.shape[1]) or (
        (_w2_inner == w1.shape[1]) and (activation == "relu2")
    ), f"w2 inner dim {_w2_inner} must be half of w1 shape[1] {w1.shape[1]} except non-gate"
    assert (topk_ids.shape == topk_weights.shape) and (
        topk_ids.shape[0] == hidden_states.shape[0]
    ), f"topk_ids shape {topk_ids.shape} and topk_weights shape {topk_weights.shape} must be equal and match hidden_states shape[0] {hidden_states.shape[0]}"

    num_tokens, hidden_dims = hidden_states.shape
    E, _, K = w1.shape
    E, OutK, N = w2.shape
    K = K * _pack
    N = N * _pack
    w1_group_size = K // w1_scale.shape[2] if use_4bit_w4a16 else 0
    w2_group_size = N // w2_scale.shape[2] if use_4bit_w4a16 else 0
    if b1 is not None:
        assert b1.shape == w1.shape[:2], "b1 shape must match w1 shape[:2]"
    if b2 is not None:
        assert b2.shape == w2.shape[:2], "b2 shape must match w2 shape[:2]"

    M = num_tokens
    TopK = topk_ids.shape[1]
    dev = hidden_states.device

    if no_combine:
        assert not inplace
        out_hidden_states = torch.empty((num_tokens, OutK), device=dev, dtype=hidden_states.dtype)
    elif inplace:
        out_hidden_states = hidden_states
    else:
        out_hidden_states = torch.empty_like(hidden_states)

    topk_ids = topk_ids.int() if topk_ids.dtype == torch.long else topk_ids
    topk_ids = topk_ids.contiguous()
    expert_offsets = _get_moe_ws("expert_offsets", (E,), torch.int32, dev)
    problem_sizes1 = _get_moe_ws("problem_sizes1", (E, 3), torch.int32, dev)
    problem_sizes2 = _get_moe_ws("problem_sizes2", (E, 3), torch.int32, dev)
    a_map = _get_moe_ws("a_map", (topk_ids.numel(),), torch.int32, dev)
    c_map = _get_moe_ws("c_map", (topk_ids.numel(),), torch.int32, dev)
    torch.ops.sgl_kernel.prepare_moe_input.default(
        topk_ids, expert_offsets, None, problem_sizes1, problem_sizes2, a_map, c_map, E, hidden_dims, TopK
    )
    input_A_shuffle = _get_moe_ws("input_A_shuffle", (num_tokens * TopK, K), hidden_states.dtype, dev)
    torch.ops.sgl_kernel.scatter_tokens_to_experts.default(hidden_states.contiguous(), c_map, input_A_shuffle)
    if w1_g_idx_perm is not None:
        input_A_shuffle = _apply_per_expert_channel_gather(input_A_shuffle, w1_g_idx_perm, expert_offsets, E)

    intermediate_cache3 = _get_moe_ws("intermediate_cache3", (M * TopK, OutK), hidden_states.dtype, dev)

    # 0=silu, 1=gelu, 3=relu2 (the gpt-oss / deepseek-v4 clamped swiglus of the reference, types 2 and 4,
    # need activation kernels that are outside this build)
    if activation == "silu":
        if gemm1_alpha is not None or swiglu_limit is not None:
            raise NotImplementedError("fused_experts: clamped swiglu variants are outside the MI355X build")
        activation_type = 0
    elif activation == "gelu":
        activation_type = 1
    else:
        activation_type = 3

    assert is_xe2_arch(), "this MoE path is built for gfx950 (MI355X) only"

    gate_factor = 2 if (2 * N == w1.shape[1]) else 1

    # GEMM1 -> activation kernel -> GEMM2: the reference's unfused route (moe.py:735-810), which it always takes for
    # 4-bit weights; 16-bit weights take it here as well (its fused-epilogue route computes the same thing without
    # rounding the GEMM1 output to the activation dtype first)
    intermediate_cache1 = _get_moe_ws("intermediate_cache1_unfused", (M * TopK, gate_factor * N), hidden_states.dtype, dev)
    intermediate_cache2 = _get_moe_ws("intermediate_cache2", (M * TopK, N), hidden_states.dtype, dev)

    def grouped_mm(out, a, w, w_scale, w_zp, bias, group_size):
        if use_4bit_w4a16:
            torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20_w4a16(out, a, w, w_scale, w_zp, bias, expert_offsets, E, use_int4_w4a16, group_size)
        else:
            torch.ops.sgl_kernel.moe_grouped_mm_nt_xe20(out, a, w, bias, expert_offsets, E, activation_type, False, 1.702, 7.0)

    grouped_mm(intermediate_cache1, input_A_shuffle, w1, w1_scale, w1_zp, b1, w1_group_size)
    if activation_type == 0:
        torch.ops.sgl_kernel.silu_and_mul(intermediate_cache2, intermediate_cache1)
    elif activation_type == 1:
        torch.ops.sgl_kernel.gelu_tanh_and_mul(intermediate_cache2, intermediate_cache1)
    else:
        intermediate_cache2 = torch.square(torch.relu(intermediate_cache1))
    if w2_g_idx_perm is not None:
        intermediate_cache2 = _apply_per_expert_channel_gather(intermediate_cache2, w2_g_idx_perm, expert_offsets, E)
    grouped_mm(intermediate_cache3, intermediate_cache2.contiguous(), w2, w2_scale, w2_zp, b2, w2_group_size)

    rsf = 1.0
    if routed_scaling_factor is not None:
        rsf = routed_scaling_factor
    torch.ops.sgl_kernel.apply_shuffle_mul_sum.default(intermediate_cache3, out_hidden_states, c_map, rsf, topk_weights)
    return out_hidden_states
