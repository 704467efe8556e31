"""rotary_embedding — restates the reference's test reference RotaryEmbedding.forward_native +
_apply_rotary_emb (tests/test_rotary_embedding.py:32-47, :91-129): fp32 math,
  x' = x cos - y sin ; y' = y cos + x sin, pairs (i, i + rot/2) for neox, (2i, 2i+1) otherwise,
elements past rot_dim pass through, result cast to the input dtype. The kernel reads cos/sin from a cache that
was rounded to the input dtype (the reference's forward_xpu passes cos_sin_cache.to(bf16), :138-148), so the
oracle takes that cache as given."""
import torch


def rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox):
    rot = cos_sin_cache.shape[1]
    cs = cos_sin_cache.float().index_select(0, positions.flatten())
    cos, sin = cs.chunk(2, dim=-1)

    def apply(t):
        shape = t.shape
        v = t.float().reshape(positions.numel(), -1, head_size)
        r, p = v[..., :rot], v[..., rot:]
        c, s = cos.unsqueeze(1), sin.unsqueeze(1)
        if is_neox:
            x1, x2 = r.chunk(2, dim=-1)
            o = torch.cat((x1 * c - x2 * s, x2 * c + x1 * s), dim=-1)
        else:
            x1, x2 = r[..., ::2], r[..., 1::2]
            o = torch.stack((x1 * c - x2 * s, x2 * c + x1 * s), dim=-1).flatten(-2)
        return torch.cat((o, p), dim=-1).reshape(shape).to(t.dtype)

    return apply(query), apply(key)
