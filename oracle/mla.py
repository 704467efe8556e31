"""MLA decode — restates the reference's test reference ref_mla
(tests/test_flash_mla_decode.py:38-59: per batch element gather the pages, keep the first
seq_len rows, scaled-dot-product attention of every head against the 576-wide rows with the
first 512 columns as values) in fp32 torch. Pinned against golden vectors produced by that
function (tests/golden/make_golden.py); tolerance of the reference test (:145-146):
1e-2 for bf16, 1e-3 for fp16."""
import torch


def mla_decode(q: torch.Tensor, kv_cache: torch.Tensor, scale: float, block_tables: torch.Tensor,
               seq_lens: torch.Tensor, v_head_dim: int = 512) -> torch.Tensor:
    """q [bs, H, 576], kv_cache [pages, page, 576], block_tables [bs, n] int, seq_lens [bs] -> [bs, H, 512]."""
    bs, H, D = q.shape
    out = torch.zeros(bs, H, v_head_dim, dtype=q.dtype)
    for i in range(bs):
        n = int(seq_lens[i])
        if n == 0:
            continue
        kv = kv_cache[block_tables[i].long()].reshape(-1, D)[:n].float()
        s = (q[i].float() @ kv.t()) * scale
        p = torch.softmax(s, dim=-1)
        out[i] = (p @ kv[:, :v_head_dim]).to(q.dtype)
    return out


def mla_prefill(q_nope: torch.Tensor, q_pe: torch.Tensor, kv_cache: torch.Tensor, scale: float,
                block_tables: torch.Tensor, cu_seqlens_q: torch.Tensor, seq_lens_k: torch.Tensor,
                causal: bool = True) -> torch.Tensor:
    """Varlen MLA prefill, restating ref_mla_prefill_varlen (reference tests/test_flash_mla_prefill.py:30-90) in
    fp32: per sequence gather the pages, keep seq_lens_k[b] rows, every head of every new token attends to the rows
    k <= (seqlen_k - seqlen_q) + q_idx (causal, prefix unmasked) with the first 512 columns as values.
    q_nope [total_q, H, 512], q_pe [total_q, H, 64] -> [total_q, H, 512]. Pinned on golden vectors from that
    function (tests/golden/mla_prefill.pt); tolerance of the reference test (:235-236): 1e-2 bf16, 1e-3 fp16."""
    total_q, H, dl = q_nope.shape
    out = torch.zeros(total_q, H, dl, dtype=q_nope.dtype)
    for b in range(seq_lens_k.shape[0]):
        q0, q1 = int(cu_seqlens_q[b]), int(cu_seqlens_q[b + 1])
        sq, sk = q1 - q0, int(seq_lens_k[b])
        if sq == 0:
            continue
        kv = kv_cache[block_tables[b].long()].reshape(-1, kv_cache.shape[-1])[:sk].float()
        q = torch.cat([q_nope[q0:q1], q_pe[q0:q1]], dim=-1).float()  # [sq, H, 576]
        s = torch.einsum("qhd,kd->hqk", q, kv) * scale
        if causal:
            qi = torch.arange(sq).unsqueeze(1)
            ki = torch.arange(sk).unsqueeze(0)
            s = s.masked_fill((ki > (sk - sq) + qi).unsqueeze(0), float("-inf"))
        p = torch.softmax(s, dim=-1)
        out[q0:q1] = torch.einsum("hqk,kd->qhd", p, kv[:, :dl]).to(q_nope.dtype)
    return out
