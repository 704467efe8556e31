"""Flash-attention forward — restates the reference's test reference attention_ref
(tests/test_flash_attention.py:349-476) and construct_local_mask (:312-346) per sequence in fp32:

  scores = (q * scale) . k ; softcap: tanh(scores / cap) * cap
  masked (-inf) where  k > q + sk - sq + right            (right = 0 when causal)
                or     left >= 0 and k < q + sk - sq - left
  optional per-head sink logit concatenated as an extra score column (no value row)
  rows with every key masked give 0.
Pinned against golden vectors produced by attention_ref (tests/golden/make_golden.py); tolerance follows the
reference's rule (:1112-1121): error vs fp32 <= 2x the error of a plain torch bf16/fp16 implementation (+ 1e-5).
"""
import torch


def attention_seq(q, k, v, scale, causal=False, window=(-1, -1), softcap=0.0, sinks=None):
    """q [sq, Hq, D], k / v [sk, Hk, D] (any float dtype) -> (out [sq, Hq, D] fp32, lse [Hq, sq] fp32)."""
    sq, Hq, D = q.shape
    sk, Hk, _ = k.shape
    g = Hq // Hk
    qf = q.float()
    kf = k.float().repeat_interleave(g, dim=1)
    vf = v.float().repeat_interleave(g, dim=1)
    left, right = window
    if causal:
        right = 0
    scores = torch.einsum("thd,shd->hts", qf * scale, kf)
    if softcap > 0:
        scores = torch.tanh(scores / softcap) * softcap
    row = torch.arange(sq).view(-1, 1)
    col = torch.arange(sk).view(1, -1)
    mask = torch.zeros(sq, sk, dtype=torch.bool)
    if left >= 0 or right >= 0:
        if right >= 0:
            mask |= col > row + sk - sq + right
        if left >= 0:
            mask |= col < row + sk - sq - left
    scores = scores.masked_fill(mask.unsqueeze(0), float("-inf"))
    if sinks is not None:
        scores = torch.cat([scores, sinks.float().view(Hq, 1, 1).expand(Hq, sq, 1)], dim=-1)
    lse = torch.logsumexp(scores, dim=-1) if scores.shape[-1] > 0 else torch.full((Hq, sq), float("-inf"))
    attn = torch.softmax(scores, dim=-1)
    attn = torch.nan_to_num(attn, nan=0.0)  # rows with no visible key
    if sinks is not None:
        attn = attn[..., :-1]
    out = torch.einsum("hts,shd->thd", attn, vf)
    return out, lse


def attention_ragged(q, k_seqs, v_seqs, cu_q, scale, **kw):
    """q [total_q, Hq, D] ragged by cu_q; k_seqs / v_seqs: per-sequence [sk_b, Hk, D]. Returns (out, lse [Hq, total_q])."""
    out = torch.zeros(q.shape, dtype=torch.float32)
    lse = torch.full((q.shape[1], q.shape[0]), float("-inf"))
    for b in range(len(k_seqs)):
        s, e = int(cu_q[b]), int(cu_q[b + 1])
        if e > s:
            o, l = attention_seq(q[s:e], k_seqs[b], v_seqs[b], scale, **kw)
            out[s:e] = o
            lse[:, s:e] = l
    return out, lse


def gather_paged(cache, page_table_row, seqlen):
    """cache [pages, page, Hk, D] -> the first seqlen tokens of the pages listed in page_table_row."""
    pages = cache[page_table_row.long()]
    return pages.reshape(-1, cache.shape[2], cache.shape[3])[:seqlen]


def dequant_fp8_cache(cache, descale):
    """fp8 (e4m3fn / e5m2) KV cache with one per-tensor descale -> fp32, as attention_ref does before its fp32 math
    (reference tests/test_flash_attention.py:404-407, :1738-1745: cache = (ref / descale).to(fp8); the kernel and the
    reference both multiply the stored value by descale)."""
    return cache.float() * float(descale)
