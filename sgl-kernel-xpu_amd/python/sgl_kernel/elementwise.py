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
