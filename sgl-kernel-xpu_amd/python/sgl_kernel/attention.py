"""MLA decode wrappers.

Mirrors reference python/sgl_kernel/attention.py:54-146 (flash_mla_decode,
flash_mla_get_workspace_size): same names, argument order, asserts and output shape
([B, H, 512], as the reference's XPU branch :113-119). Differences that are deliberate:
q_nope / q_pe are NOT copied to contiguous memory (the HIP kernel takes their strides),
and the CUDA-only pad-to-128-heads branch of the reference (:86-93) does not exist.
"""
import torch


def flash_mla_decode(
    q_nope: torch.Tensor,
    q_pe: torch.Tensor,
    kv_c_and_k_pe_cache: torch.Tensor,
    seq_lens: torch.Tensor,
    page_table: torch.Tensor,
    workspace: torch.Tensor,
    sm_scale: float,
    num_kv_splits: int = 1,
) -> torch.Tensor:
    assert q_nope.ndim == 3, f"q_nope must be a 3D tensor, but got {q_nope.ndim}"
    assert q_pe.ndim == 3, f"q_pe must be a 3D tensor, but got {q_pe.ndim}"
    assert (
        kv_c_and_k_pe_cache.ndim == 3
    ), f"kv_c_and_k_pe_cache must be a 3D tensor, but got {kv_c_and_k_pe_cache.ndim}"

    B_q, H, D_q_nope = q_nope.shape
    B_q_2, H_2, D_q_pe = q_pe.shape
    assert (B_q == B_q_2) and (H == H_2)

    _, PAGE_SIZE, D_ckv = kv_c_and_k_pe_cache.shape

    D_latent = 512
    D_rope = 64
    assert D_q_nope == D_latent
    assert D_q_pe == D_rope
    assert D_ckv == D_latent + D_rope

    MAX_HEADS = 128
    assert H <= MAX_HEADS, f"H must be <= {MAX_HEADS}, but got {H}"
    if q_nope.stride(-1) != 1:
        q_nope = q_nope.contiguous()
    if q_pe.stride(-1) != 1:
        q_pe = q_pe.contiguous()

    assert len(page_table.shape) == 2
    B_block_table, block_num = page_table.shape
    assert B_block_table == B_q
    assert block_num > 0, f"block num must be greater than 0, got {block_num}"
    assert block_num % (128 / PAGE_SIZE) == 0

    assert q_nope.dtype in (
        torch.float16,
        torch.bfloat16,
    ), f"q_nope.dtype needs to be fp16 or bf16 but got {q_nope.dtype}."
    assert q_nope.dtype == q_pe.dtype == kv_c_and_k_pe_cache.dtype
    assert seq_lens.dtype == torch.int32, f"seq_lens.dtype needs to be int32 but got {seq_lens.dtype}."
    assert page_table.dtype == torch.int32, f"page_table.dtype needs to be int32 but got {page_table.dtype}."

    out = q_nope.new_empty((B_q, H, D_latent))

    torch.ops.sgl_kernel.flash_mla_decode.default(
        out,
        q_nope,
        q_pe,
        kv_c_and_k_pe_cache,
        seq_lens,
        page_table,
        workspace,
        sm_scale,
        num_kv_splits,
    )
    return out


def flash_mla_get_workspace_size(
    max_seq_len: int,
    num_batches: int,
    num_heads: int = 0,
    page_size: int = 0,
    num_kv_splits: int = -1,
) -> int:
    assert max_seq_len > 0, f"max_seq_len must be greater than 0, got {max_seq_len}"
    assert num_batches > 0, f"num_batches must be greater than 0, got {num_batches}"
    return torch.ops.sgl_kernel.flash_mla_get_workspace_size.default(
        max_seq_len, num_batches, num_heads, page_size, num_kv_splits
    )


def flash_mla_prefill(
    q_nope: torch.Tensor,
    q_pe: torch.Tensor,
    kv_c_and_k_pe_cache: torch.Tensor,
    cu_seqlens_q: torch.Tensor,
    seq_lens_k: torch.Tensor,
    max_seqlen_q: int,
    page_table: torch.Tensor,
    workspace: torch.Tensor,
    sm_scale: float,
    causal: bool = True,
    num_kv_splits: int = -1,
) -> torch.Tensor:
    """MLA prefill with ragged Q and (bottom-right aligned) causal masking; mirrors reference
    python/sgl_kernel/attention.py:149-233: same arguments, same checks, returns out [total_q, H, 512].

    The reference pads `out` to a multiple of 256 rows because its epilogue writes whole tiles (:212-217); the
    HIP kernel writes exactly total_q rows, so no padding is allocated here."""
    assert q_nope.ndim == 3, f"q_nope must be 3D (total_q, heads, dim), got {q_nope.ndim}"
    assert q_pe.ndim == 3, f"q_pe must be 3D (total_q, heads, dim), got {q_pe.ndim}"
    assert (
        kv_c_and_k_pe_cache.ndim == 3
    ), f"kv_c_and_k_pe_cache must be 3D (pages, page_size, dim), got {kv_c_and_k_pe_cache.ndim}"
    total_q, H, D_latent = q_nope.shape
    _, _, D_rope = q_pe.shape
    _, PAGE_SIZE, D_ckv = kv_c_and_k_pe_cache.shape
    assert D_ckv == D_latent + D_rope, f"kv dim {D_ckv} must equal D_latent({D_latent}) + D_rope({D_rope})"
    assert q_nope.dtype in (torch.float16, torch.bfloat16), f"q_nope.dtype must be fp16 or bf16, got {q_nope.dtype}"
    assert q_nope.dtype == q_pe.dtype == kv_c_and_k_pe_cache.dtype
    assert cu_seqlens_q.dtype == torch.int32
    assert seq_lens_k.dtype == torch.int32
    assert page_table.dtype == torch.int32
    batch_size = cu_seqlens_q.shape[0] - 1
    assert seq_lens_k.shape[0] == batch_size
    out = q_nope.new_empty((total_q, H, D_latent))
    if total_q == 0:
        return out
    torch.ops.sgl_kernel.flash_mla_prefill.default(
        out,
        q_nope,
        q_pe,
        kv_c_and_k_pe_cache,
        cu_seqlens_q,
        seq_lens_k.contiguous(),
        max_seqlen_q,
        page_table,
        workspace,
        sm_scale,
        causal,
        num_kv_splits,
    )
    return out


def flash_mla_prefill_get_workspace_size(
    max_seq_len: int,
    num_batches: int,
    num_heads: int = 0,
    page_size: int = 0,
    num_kv_splits: int = -1,
) -> int:
    """Reference python/sgl_kernel/attention.py:236-247. This library's prefill needs no workspace (0)."""
    assert max_seq_len > 0, f"max_seq_len must be > 0, got {max_seq_len}"
    assert num_batches > 0, f"num_batches must be > 0, got {num_batches}"
    return torch.ops.sgl_kernel.flash_mla_prefill_get_workspace_size.default(
        max_seq_len, num_batches, num_heads, page_size, num_kv_splits
    )
