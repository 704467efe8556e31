"""QServe W4A8 GEMMs — CPU restatement (test infrastructure only; never imported by the product path).

The reference declares the ops (include/sgl_kernel_ops.h:1132-1148) and pins their meaning, the quantisers and the
weight / scale repacking in its tests:
  tests/test_qserve_w4a8_per_chn_gemm.py:12-56 (convert_to_qserve_format), :59-78 (quantisers), :80-88 (reference)
  tests/test_qserve_w4a8_per_group_gemm.py:12-92, :95-132, :134-145
Everything below restates those functions (fp32 math instead of the device fp16 matmul of the per-channel test) and
is pinned on golden vectors produced by the reference's own functions (tests/golden/qserve_w4a8.pt)."""
import torch


def pack_qserve_weight(qweight: torch.Tensor) -> torch.Tensor:
    """[N, K] codes 0..15 -> [N, K/2] int8 in the QServe 32x32 interleaved layout (per_chn test :20-46)."""
    n, k = qweight.shape
    assert n % 32 == 0 and k % 32 == 0
    r = qweight.reshape(n // 32, 2, 2, 8, k // 32, 2, 4, 4).permute(0, 4, 3, 6, 1, 5, 2, 7).contiguous()
    r = r.permute(0, 1, 2, 3, 5, 6, 7, 4).contiguous().to(torch.int8)
    packed = (r[..., 1] << 4) + r[..., 0]
    return packed.reshape(n // 32, k // 32, 32, 16).reshape(n, k // 2).contiguous()


def sym_quantize(t: torch.Tensor):
    """int8 symmetric per-row activation quantiser (:73-77)."""
    s = t.abs().max(dim=-1, keepdim=True)[0] / 127
    return torch.clamp(torch.round(t / s), -128, 127).to(torch.int8), s.to(torch.float16)


def asym_quantize_u4(t: torch.Tensor):
    """uint4 asymmetric per-row weight quantiser (per_chn test :59-70)."""
    mn, mx = t.min(dim=-1, keepdim=True)[0], t.max(dim=-1, keepdim=True)[0]
    s = (mx - mn) / 15
    z = -torch.round(mn / s)
    q = torch.clamp(torch.round(t / s) + z, 0, 15).to(torch.int8)
    return q, s.to(torch.float16), z.to(torch.int8)


def per_chn_inputs(b_q, b_scale, b_zero):
    """-> (packed weight, wscales fp16 [N], w_szs fp16 [N]) (:48-56)."""
    n = b_q.shape[0]
    scale = b_scale.reshape(n).to(torch.float16).contiguous()
    return pack_qserve_weight(b_q), scale, b_zero.reshape(n).to(torch.float16).contiguous() * scale


def w4a8_per_chn_gemm(a_q, b_q, a_scale, b_scale, b_zero, out_dtype=torch.float16):
    """(:80-88) out = (a_q @ (b_q - zero)^T) * a_scale * b_scale."""
    o = a_q.float() @ (b_q.float() - b_zero.float()).t()
    return (o * a_scale.float().view(-1, 1) * b_scale.float().view(1, -1)).to(out_dtype)


def progressive_group_quantize(t: torch.Tensor, group: int = 128):
    """two-level weight quantiser of the per-group test (:95-125)."""
    chn = t.abs().max(dim=-1, keepdim=True)[0] / 119
    t8 = torch.clamp(torch.round(t / chn), -119, 119).reshape(-1, group)
    mn, mx = t8.min(dim=-1, keepdim=True)[0], t8.max(dim=-1, keepdim=True)[0]
    s = torch.round((mx - mn) / 15)
    z = -torch.round(mn / s)
    q = torch.clamp(torch.round(t8 / s) + z, 0, 15).reshape(t.shape[0], -1).to(torch.int8)
    return q, chn.to(torch.float16), s.reshape(t.shape[0], -1).to(torch.int8), z.reshape(t.shape[0], -1).to(torch.int8)


def per_group_inputs(b_q, chn_scale, scale_i8, zero_i8, group: int = 128):
    """-> (packed weight, wscales [N], scales_i8 [K/g, N] permuted, zeros = -zero*scale [K/g, N] permuted) (:12-92)."""
    n, k = b_q.shape

    def perm(x):
        x = x.reshape(n, k // group).transpose(0, 1).contiguous().reshape(k // group, n // 32, 4, 8)
        return x.transpose(-2, -1).contiguous().reshape(k // group, n).contiguous()

    s = perm(scale_i8)
    return pack_qserve_weight(b_q), chn_scale.reshape(n), s, perm(-zero_i8) * s


def w4a8_per_group_gemm(a_q, b_q, a_scale, chn_scale, scale_i8, zero_i8, group: int = 128, out_dtype=torch.float16):
    """(:134-145) dequantise to fp32 per group, fp32 matmul, scale, cast."""
    dq = (b_q.reshape(-1, group).float() - zero_i8.reshape(-1, 1).float()) * scale_i8.reshape(-1, 1).float()
    o = a_q.float() @ dq.reshape(b_q.shape).t()
    return (o * a_scale.float().view(-1, 1) * chn_scale.float().view(1, -1)).to(out_dtype)
