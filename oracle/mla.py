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
