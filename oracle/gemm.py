"""Scaled GEMMs — restates the reference's test references (the only definition
of these ops: they are declared in include/sgl_kernel_ops.h:567-586 but have no
kernel in the reference):

  fp8_blockwise_scaled_mm : tests/test_fp8_blockwise_gemm.py:23-63 (baseline_scaled_mm)
  fp8_scaled_mm           : tests/test_fp8_gemm.py:11-19  (bias added after the cast)
  int8_scaled_mm          : tests/test_int8_gemm.py:16-22 (bias added in fp32)

fp32 matmul on dequantised operands; tolerances are the reference tests'.
"""
import torch


def fp8_blockwise_scaled_mm(a, b, scale_a, scale_b, out_dtype, block=128):
    """a [M,K] fp8, b [K,N] fp8, scale_a [M,K/block], scale_b [K/block, N/block] fp32."""
    M, K = a.shape
    N = b.shape[1]
    sa = scale_a.float().repeat_interleave(block, dim=1)[:, :K]
    sb = scale_b.float().repeat_interleave(block, dim=0)[:K].repeat_interleave(block, dim=1)[:, :N]
    return torch.mm(sa * a.float(), sb * b.float()).to(out_dtype)


def fp8_scaled_mm(a, b, scale_a, scale_b, out_dtype, bias=None):
    o = torch.matmul(a.float(), b.float())
    o = o * scale_a.float().view(-1, 1) * scale_b.float().view(1, -1)
    o = o.to(out_dtype)
    if bias is not None:
        o = o + bias
    return o


def int8_scaled_mm(a, b, scale_a, scale_b, out_dtype, bias=None):
    o = torch.matmul(a.float(), b.float())
    o = o * scale_a.float().view(-1, 1) * scale_b.float().view(1, -1)
    if bias is not None:
        o = o + bias.float()
    return o.to(out_dtype)
