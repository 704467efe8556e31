"""Quantisation and scaled-GEMM wrappers.

Mirrors reference python/sgl_kernel/gemm.py:13-42 (scaled mm), :85-126
(per-token-group quant): same names, argument order and defaults.
"""
import os
from typing import Optional

import torch


def int8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias=None):
    """``(mat_a @ mat_b) * scales_a[:, None] * scales_b[None, :] (+ bias)``; int8 in, int32 accumulate."""
    return torch.ops.sgl_kernel.int8_scaled_mm.default(
        mat_a,
        mat_b,
        scales_a,
        scales_b,
        out_dtype,
        bias,
    )


def fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias=None):
    """``(mat_a @ mat_b) * scales_a[:, None] * scales_b[None, :] (+ bias)``; e4m3fn in, fp32 accumulate."""
    return torch.ops.sgl_kernel.fp8_scaled_mm.default(
        mat_a,
        mat_b,
        scales_a,
        scales_b,
        out_dtype,
        bias,
    )


def fp8_blockwise_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype):
    """``(scales_a (x) mat_a) @ (scales_b (x) mat_b)`` with 1x128 / 128x128 fp32 block
    scales; mat_a [M,K] e4m3fn row-major, mat_b [K,N] e4m3fn column-major."""
    return torch.ops.sgl_kernel.fp8_blockwise_scaled_mm.default(
        mat_a,
        mat_b,
        scales_a,
        scales_b,
        out_dtype,
    )


def _env_flag(name: str) -> bool:
    return os.environ.get(name, "").strip().lower() in ("1", "true", "yes", "on")


def sgl_per_token_group_quant_8bit(
    input: torch.Tensor,
    output_q: torch.Tensor,
    output_s: torch.Tensor,
    group_size: int,
    eps: float,
    fp8_min: float,
    fp8_max: float,
    scale_ue8m0: bool = False,
    fuse_silu_and_mul: bool = False,
    masked_m: Optional[torch.Tensor] = None,
    enable_v2: Optional[bool] = None,
) -> None:
    if enable_v2 is None:
        # the reference asks sglang.srt.utils.get_bool_env_var (gemm.py:98-101); same variable, no import
        enable_v2 = _env_flag("SGLANG_PER_TOKEN_GROUP_QUANT_8BIT_V2")

    if enable_v2:
        return torch.ops.sgl_kernel.sgl_per_token_group_quant_8bit_v2.default(
            input,
            output_q,
            output_s,
            group_size,
            eps,
            fp8_min,
            fp8_max,
            scale_ue8m0,
            fuse_silu_and_mul,
            masked_m,
        )

    assert not fuse_silu_and_mul, "only v2 support fuse_silu_and_mul"
    assert masked_m is None, "only v2 support masked_m"
    torch.ops.sgl_kernel.sgl_per_token_group_quant_8bit.default(
        input, output_q, output_s, group_size, eps, fp8_min, fp8_max, scale_ue8m0
    )


# legacy names kept by the reference (gemm.py:124-126)
sgl_per_token_group_quant_fp8 = sgl_per_token_group_quant_8bit
sgl_per_token_group_quant_int8 = sgl_per_token_group_quant_8bit


def qserve_w4a8_per_chn_gemm(
    in_feats: torch.Tensor,
    kernel: torch.Tensor,
    wscales: torch.Tensor,
    ascales: torch.Tensor,
    w_szs: torch.Tensor,
    a_ssums: torch.Tensor,
    out_feats: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """Reference python/sgl_kernel/gemm.py:314-333 (out dtype float16 only)."""
    if out_feats is None:
        out_feats = torch.empty((in_feats.shape[0], kernel.shape[0]), device=in_feats.device, dtype=torch.float16)
    torch.ops.sgl_kernel.qserve_w4a8_per_chn_gemm.default(in_feats, kernel, wscales, ascales, w_szs, a_ssums, out_feats)
    return out_feats


def qserve_w4a8_per_group_gemm(
    in_feats: torch.Tensor,
    kernel: torch.Tensor,
    zeros: torch.Tensor,
    scales_i8: torch.Tensor,
    wscales: torch.Tensor,
    ascales: torch.Tensor,
    out_feats: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """Reference python/sgl_kernel/gemm.py:336-355 (out dtype float16 only)."""
    if out_feats is None:
        out_feats = torch.empty((in_feats.shape[0], kernel.shape[0]), device=in_feats.device, dtype=torch.float16)
    torch.ops.sgl_kernel.qserve_w4a8_per_group_gemm.default(in_feats, kernel, zeros, scales_i8, wscales, ascales, out_feats)
    return out_feats


def awq_dequantize(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor) -> torch.Tensor:
    """Reference python/sgl_kernel/gemm.py:7-10."""
    return torch.ops.sgl_kernel.awq_dequantize.default(qweight, scales, qzeros)


def sgl_per_tensor_quant_fp8(input: torch.Tensor, output_q: torch.Tensor, output_s: torch.Tensor, is_static: bool) -> None:
    """Reference python/sgl_kernel/gemm.py (sgl_per_tensor_quant_fp8): output_s must be zero-initialised when dynamic."""
    torch.ops.sgl_kernel.sgl_per_tensor_quant_fp8.default(input, output_q, output_s, is_static)


def sgl_per_token_quant_fp8(input: torch.Tensor, output_q: torch.Tensor, output_s: torch.Tensor) -> None:
    """Reference python/sgl_kernel/gemm.py:236-241."""
    torch.ops.sgl_kernel.sgl_per_token_quant_fp8.default(input, output_q, output_s)
