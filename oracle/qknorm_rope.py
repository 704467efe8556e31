"""TEST INFRASTRUCTURE (CPU oracle) — fused per-head RMSNorm + rotary embedding of q and k.

Restates the arithmetic of reference src/sycl/FusedQKNormRope.cpp in fp32 torch (one rounding to the storage dtype at
the end, as the kernels do):
  * fused_inplace_qknorm_rope: FusedQKNormRopeCacheKernel :617-737 (angles from an fp32 cos_sin_cache)
  * fused_qk_norm_rope       : FusedQKNormRopeKernel :268-398 + computeFreqYarn :42-67 (angles computed, YaRN blend,
                               attention factor on the rotated part only)
Pinned by tests/test_oracle_golden.py against vectors from the reference tests' own references
`fused_qk_norm_rope_with_cache_reference` / `fused_qk_norm_rope_reference` (tests/test_fused_qk_norm_rope.py:104-218)."""
import torch


def _norm(x, w, eps):
    xf = x.float()
    return xf * (torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps) * w.float())


def _rotate(y, cos, sin, rope_dim, is_neox, factor=1.0):
    """y [tokens, heads, D] fp32; cos / sin [tokens, rope_dim / 2] fp32."""
    half = rope_dim // 2
    c, s = cos.unsqueeze(1), sin.unsqueeze(1)
    rot = y[..., :rope_dim]
    if is_neox:
        a, b = rot[..., :half], rot[..., half:]
        out = torch.cat([a * c - b * s, b * c + a * s], dim=-1)
    else:
        a, b = rot[..., 0::2], rot[..., 1::2]
        out = torch.stack([a * c - b * s, a * s + b * c], dim=-1).flatten(-2)
    return torch.cat([out * factor, y[..., rope_dim:]], dim=-1)


def fused_inplace_qknorm_rope(q, k, q_weight, k_weight, cos_sin_cache, positions, is_neox, eps=1e-6):
    """q [tokens, Hq, D], k [tokens, Hk, D] -> (q_out, k_out) in the input dtype."""
    rope_dim = cos_sin_cache.shape[1]
    cs = cos_sin_cache.float()[positions.long()]
    cos, sin = cs[:, : rope_dim // 2], cs[:, rope_dim // 2:]
    qo = _rotate(_norm(q, q_weight, eps), cos, sin, rope_dim, is_neox).to(q.dtype)
    ko = _rotate(_norm(k, k_weight, eps), cos, sin, rope_dim, is_neox).to(k.dtype)
    return qo, ko


def yarn_inv_freq(rotary_dim, base, factor, low, high):
    j = torch.arange(0, rotary_dim // 2, dtype=torch.float32)
    freq = torch.exp2((-2.0 * j / rotary_dim) * torch.log2(torch.tensor(float(base))))
    if factor != 1.0:
        high_adj = high + 0.001 if abs(low - high) <= 1e-6 else high
        ramp = torch.clamp((2.0 * j - low) / (high_adj - low), 0.0, 1.0)
        freq = (freq / factor) * (1.0 - ramp) + freq * ramp
    return freq


def fused_qk_norm_rope(qkv, Hq, Hk, Hv, head_dim, eps, q_weight, k_weight, base, is_neox, position_ids, factor=1.0,
                       low=1.0, high=1.0, attention_factor=1.0, rotary_dim=None):
    rotary_dim = head_dim if rotary_dim is None else rotary_dim
    T = qkv.shape[0]
    x = qkv.view(T, Hq + Hk + Hv, head_dim)
    theta = position_ids.float().unsqueeze(1) * yarn_inv_freq(rotary_dim, base, factor, low, high).unsqueeze(0)
    cos, sin = torch.cos(theta), torch.sin(theta)
    q = _rotate(_norm(x[:, :Hq], q_weight, eps), cos, sin, rotary_dim, is_neox, attention_factor).to(qkv.dtype)
    k = _rotate(_norm(x[:, Hq:Hq + Hk], k_weight, eps), cos, sin, rotary_dim, is_neox, attention_factor).to(qkv.dtype)
    return torch.cat([q, k, x[:, Hq + Hk:]], dim=1).reshape(T, -1)
