"""Per-token-group 8-bit quantisation — restates the reference kernel
src/sycl/per_token_group_quant_8bit.cpp:86-207 step for step in IEEE fp32:

  amax = max(eps, max|x|)              :97, :139-141
  y_s  = amax / qmax                   :157
  ue8m0: e = ceil(log2(max(y_s,1e-10))), y_s = 2^e, byte = e + 127    :161-165
  q    = cast(min(max(x * (1 / y_s), qmin), qmax))                    :171, :185
  fp8 e4m3fn cast rounds to nearest even; the int8 cast truncates      :191-196

Every step is an exactly-rounded fp32 operation, so the HIP kernel is compared
BIT-EXACTLY with this (codes and scales). Pinned against golden vectors from
tests/test_per_token_group_quant_8bit.py:17-54 (tolerances of :260-276; ue8m0
bytes exact as :405). ceil(log2) is evaluated on the float's exponent/mantissa
bits, which is the exact value of the reference's ceil(log2(.)) expression.
"""
import torch

FP8_MAX = 448.0


def _ceil_log2_exact(y: torch.Tensor) -> torch.Tensor:
    bits = y.contiguous().view(torch.int32)
    e = ((bits >> 23) & 0xFF) - 127
    return e + ((bits & 0x7FFFFF) != 0).to(torch.int32)


def per_token_group_quant_8bit(x: torch.Tensor, group_size: int, dst_dtype: torch.dtype, eps: float = 1e-10,
                               qmin: float = None, qmax: float = None, scale_ue8m0: bool = False):
    """x [rows, k] -> (q [rows, k] dst_dtype, scales [rows, k/group] fp32, ue8m0 bytes or None)."""
    assert x.dim() == 2 and x.shape[1] % group_size == 0
    if qmax is None:
        qmax = FP8_MAX if dst_dtype == torch.float8_e4m3fn else 127.0
    if qmin is None:
        qmin = -qmax
    rows, k = x.shape
    xf = x.float().view(rows, k // group_size, group_size)
    amax = xf.abs().amax(dim=-1).clamp_min(torch.tensor(eps, dtype=torch.float32))
    y_s = amax / torch.tensor(qmax, dtype=torch.float32)
    ue = None
    if scale_ue8m0:
        e = _ceil_log2_exact(y_s.clamp_min(torch.tensor(1e-10, dtype=torch.float32)))
        ue = (e + 127).to(torch.uint8)
        y_s = ((e + 127) << 23).to(torch.int32).view(torch.float32)
    inv = 1.0 / y_s
    qv = (xf * inv.unsqueeze(-1)).clamp(min=qmin, max=qmax)
    if dst_dtype == torch.int8:
        q = qv.to(torch.int32).to(torch.int8)  # float -> int conversion truncates toward zero
    else:
        q = qv.to(torch.float8_e4m3fn)
    return q.view(rows, k), y_s, ue


def per_token_group_quant_8bit_v2(x: torch.Tensor, group_size: int, dst_dtype: torch.dtype, scale_ue8m0: bool = False,
                                  fuse_silu_and_mul: bool = False, masked_m: torch.Tensor = None):
    """v2 = v1 arithmetic on T(T(silu(x1)) * x2) when fused, silu(v) = h (1 + tanh h), h = v / 2 (reference
    src/sycl/per_token_group_quant_8bit_v2.cpp:113-117, :257-259); masked layout [E, T, *] processes rows
    < masked_m[e] only (the rest of q / scales is left untouched -> returned as zeros here).
    Returns (q [.., H], scales [.., H/group] fp32, ue8m0 bytes or None, row_valid mask [.., T])."""
    T = x.dtype
    lead = x.shape[:-1]
    if fuse_silu_and_mul:
        h2 = x.shape[-1] // 2
        a, b = x[..., :h2].float(), x[..., h2:].float()
        half = 0.5 * a
        sv = (half * (1.0 + torch.tanh(half))).to(T)
        v = (sv.float() * b).to(T)
    else:
        v = x
    hidden = v.shape[-1]
    q, s, ue = per_token_group_quant_8bit(v.reshape(-1, hidden), group_size, dst_dtype, scale_ue8m0=scale_ue8m0)
    q = q.view(*lead, hidden)
    s = s.view(*lead, hidden // group_size)
    ue = ue.view(*lead, hidden // group_size) if ue is not None else None
    valid = torch.ones(lead, dtype=torch.bool)
    if masked_m is not None:
        valid = torch.arange(lead[1]).view(1, -1) < masked_m.view(-1, 1)
    return q, s, ue, valid


# ---- per-token / per-tensor fp8 quantisation and AWQ dequantisation ---------------------------

def per_token_quant_fp8(x: torch.Tensor):
    """Reference src/sycl/per_token_quant_fp8.cpp:36, :96-125: per row scale = rowmax|x| / 448,
    q = e4m3(clamp(x * (scale == 0 ? 0 : 1 / scale), +-448)). Returns (q e4m3fn, scale fp32 [rows])."""
    xf = x.float().reshape(-1, x.shape[-1])
    scale = xf.abs().amax(dim=-1) / 448.0
    inv = torch.where(scale == 0, torch.zeros_like(scale), 1.0 / scale)
    q = (xf * inv[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.reshape(x.shape), scale


def per_tensor_quant_fp8(x: torch.Tensor, scale: torch.Tensor = None):
    """Reference src/sycl/per_tensor_quant_fp8.cpp:46-47, :58-105, :121-161: dynamic scale = max|x| / 448;
    q = e4m3(clamp(x * (1 / (scale + 1e-8)), +-448)) with fp32 arithmetic. Returns (q, scale fp32 [1])."""
    xf = x.float()
    if scale is None:
        scale = (xf.abs().max() / 448.0).reshape(1)
    inv = torch.tensor(1.0, dtype=torch.float32) / (scale.float().reshape(1) + torch.tensor(1e-8, dtype=torch.float32))
    q = (xf * inv).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q, scale.float().reshape(1)


def awq_dequantize(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor) -> torch.Tensor:
    """Reference src/sycl/awq_dequantize.cpp:15-51 (tests/test_awq_dequant.py:13-62): int32 words hold 8 nibbles in the
    order 0,4,1,5,2,6,3,7; out[k, 8c+i] = (w - z) * scale in the dtype of scales."""
    order = torch.tensor([0, 4, 1, 5, 2, 6, 3, 7])
    shifts = (order * 4).view(1, 1, 8)
    w = (qweight.to(torch.int64)[:, :, None] >> shifts) & 15
    z = (qzeros.to(torch.int64)[:, :, None] >> shifts) & 15
    group = qweight.shape[0] // scales.shape[0]
    z = z.repeat_interleave(group, dim=0)
    s = scales.float().repeat_interleave(group, dim=0)
    out = (w - z).reshape(qweight.shape[0], -1).float() * s
    return out.to(scales.dtype)
