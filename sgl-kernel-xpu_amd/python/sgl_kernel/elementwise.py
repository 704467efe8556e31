"""RMSNorm family and activation-and-mul wrappers.

Mirrors reference python/sgl_kernel/elementwise.py:7-140 (norms) and :205-260
(activations): same names, arguments, output allocation and error behaviour;
the op call lands in csrc/torch_extension_hip.cc -> include/sglk.h.
"""
from typing import Optional

import torch


def rmsnorm(
    input: torch.Tensor,
    weight: torch.Tensor,
    eps: float = 1e-6,
    out: Optional[torch.Tensor] = None,
    enable_pdl: Optional[bool] = None,
) -> torch.Tensor:
    """``out[i] = (input[i] / RMS(input)) * weight[i]``; input is (batch, hidden) or
    (batch, seq, hidden), last dim contiguous. ``enable_pdl`` is accepted and unused,
    as on the reference (elementwise.py:12)."""
    if out is None:
        out = torch.empty_like(input)
    torch.ops.sgl_kernel.rmsnorm(out, input, weight, eps)
    return out


def fused_add_rmsnorm(
    input: torch.Tensor,
    residual: torch.Tensor,
    weight: torch.Tensor,
    eps: float = 1e-6,
    enable_pdl: Optional[bool] = None,
) -> None:
    """In place: ``residual += input``; ``input = residual / RMS(residual) * weight``."""
    torch.ops.sgl_kernel.fused_add_rmsnorm(input, residual, weight, eps)


def gemma_rmsnorm(
    input: torch.Tensor,
    weight: torch.Tensor,
    eps: float = 1e-6,
    out: Optional[torch.Tensor] = None,
    enable_pdl: Optional[bool] = None,
) -> torch.Tensor:
    """``out[i] = (input[i] / RMS(input)) * (weight[i] + 1)``."""
    if out is None:
        out = torch.empty_like(input)
    torch.ops.sgl_kernel.gemma_rmsnorm(out, input, weight, eps)
    return out


def gemma_fused_add_rmsnorm(
    input: torch.Tensor,
    residual: torch.Tensor,
    weight: torch.Tensor,
    eps: float = 1e-6,
    enable_pdl: Optional[bool] = None,
) -> None:
    """In place: ``residual += input``; ``input = residual / RMS(residual) * (weight + 1)``."""
    torch.ops.sgl_kernel.gemma_fused_add_rmsnorm(input, residual, weight, eps)


def _check_shape(input: torch.Tensor, output: torch.Tensor) -> None:
    assert input.ndim == output.ndim, f"{input.ndim} != {output.ndim}"
    assert input.shape[:-1] == output.shape[:-1], f"{input.shape[:-1]} != {output.shape[:-1]}"
    assert input.shape[-1] == 2 * output.shape[-1], f"{input.shape[-1]} != {2 * output.shape[-1]}"


def _act_and_mul(op, input: torch.Tensor, out: Optional[torch.Tensor]) -> torch.Tensor:
    # same 16-byte row rule and message as reference elementwise.py:216-217
    if input.shape[-1] * input.dtype.itemsize % 16 != 0:
        raise ValueError("The pointers must be multiple of 16 bytes.")
    if out is not None:
        _check_shape(input, out)
    else:
        out = torch.empty(
            input.shape[:-1] + (input.shape[-1] // 2,),
            device=input.device,
            dtype=input.dtype,
        )
    op(out, input)
    return out


def silu_and_mul(input: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    return _act_and_mul(torch.ops.sgl_kernel.silu_and_mul, input, out)


def gelu_tanh_and_mul(input: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    return _act_and_mul(torch.ops.sgl_kernel.gelu_tanh_and_mul, input, out)


def gelu_and_mul(input: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    return _act_and_mul(torch.ops.sgl_kernel.gelu_and_mul, input, out)


def silu_and_mul_clamp(input: torch.Tensor, out: torch.Tensor, swiglu_limit: float) -> None:
    """The DeepSeek-V4 swiglu (reference elementwise.py:231-255): ``input`` [M, 2H] bf16 / fp16 with the gate half first,
    ``out`` [M, H] pre-allocated. Both halves are clamped in bf16 - gate = min(gate, limit), up = clamp(up, -limit, limit) -
    then ``out = silu(gate) * up`` in fp32, rounded to the input dtype. Returns nothing."""
    if input.shape[-1] * input.dtype.itemsize % 16 != 0:
        raise ValueError("The pointers must be multiple of 16 bytes.")
    _check_shape(input, out)
    torch.ops.sgl_kernel.silu_and_mul_clamp(out, input, swiglu_limit)


def rotary_embedding(
    positions: torch.Tensor,
    query: torch.Tensor,
    key: torch.Tensor,
    head_size: int,
    cos_sin_cache: torch.Tensor,
    is_neox: bool = True,
):
    """Rotary position embedding (op schema of reference src/torch_extension_sycl.cc:117-120; the reference
    has no python wrapper, its tests call torch.ops.sgl_kernel.rotary_embedding directly). 2-D query/key
    [tokens, heads*head_size] are rotated in place and returned; 3-D [tokens, heads, head_size] inputs give new
    tensors (rot_dim must equal head_size there). cos_sin_cache [max_pos, rot_dim] must have query's dtype."""
    return torch.ops.sgl_kernel.rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox)


# ---- attention prologue ops: reference elementwise.py:143-200, :288-317, :372-462 ----------

def fused_inplace_qknorm_rope(q: torch.Tensor, k: torch.Tensor, q_weight: torch.Tensor, k_weight: torch.Tensor,
                              cos_sin_cache: torch.Tensor, positions: torch.Tensor, is_neox: bool, eps: float = 1e-6,
                              head_dim: int = 0, rope_dim: int = 0) -> None:
    """In place on q [tokens, Hq, D] / k [tokens, Hk, D] (or 4-D [batch, seq, heads, D]; last dim contiguous):
    per-head RMSNorm with q_weight / k_weight [D], then rotation of the first rope_dim elements by the fp32
    cos_sin_cache row [max_pos, rope_dim] of each token's position. head_dim / rope_dim = 0: taken from the tensors."""
    torch.ops.sgl_kernel.fused_inplace_qknorm_rope(q, k, q_weight, k_weight, cos_sin_cache, positions, is_neox, eps,
                                                   head_dim, rope_dim)


def fused_qk_norm_rope(qkv: torch.Tensor, num_heads_q: int, num_heads_k: int, num_heads_v: int, head_dim: int, eps: float,
                       q_weight: torch.Tensor, k_weight: torch.Tensor, base: float, is_neox: bool,
                       position_ids: torch.Tensor, factor: float = 1.0, low: float = 1.0, high: float = 1.0,
                       attention_factor: float = 1.0, rotary_dim: int = None) -> None:
    """In place on the packed qkv [tokens, (Hq + Hk + Hv) * head_dim]: per-head RMSNorm of the q and k heads, then
    rotary embedding with angles position * base^(-2j / rotary_dim) (YaRN-blended when factor != 1, scaled by
    attention_factor). V is untouched. rotary_dim defaults to head_dim; position_ids int32."""
    torch.ops.sgl_kernel.fused_qk_norm_rope(qkv, num_heads_q, num_heads_k, num_heads_v, head_dim, eps, q_weight, k_weight,
                                            base, is_neox, position_ids, factor, low, high, attention_factor,
                                            head_dim if rotary_dim is None else rotary_dim)


def store_cache_xpu(k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                    indices: torch.Tensor) -> None:
    """k_cache[indices[t]] = k[t], v_cache[indices[t]] = v[t] in one launch; k / v [tokens, row_dim] may be row-strided
    views (rows contiguous), the caches are dense [cache_size, row_dim], indices[t] = -1 skips token t. (The name is the
    reference's; `store_cache` is the same function.)"""
    torch.ops.sgl_kernel.store_cache(k, v, k_cache, v_cache, indices.long())


store_cache = store_cache_xpu
