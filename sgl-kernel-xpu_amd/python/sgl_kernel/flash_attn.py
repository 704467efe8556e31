"""Flash-attention wrappers over torch.ops.sgl_kernel.fwd.

Mirrors reference python/sgl_kernel/flash_attn.py: is_fa3_supported :7-24,
flash_attn_with_kvcache :103-298 and flash_attn_varlen_func :301-372 — same arguments, the same
default softmax_scale ((headdim + headdim_v_of_qv)^-0.5), the 4-D -> ragged 3-D flattening of q,
and the aliasing of `cache_seqlens` (per-sequence lengths, not cumulative) onto the op's
`cu_seqlens_k` slot. Features the reference parses but rejects (qv, in-kernel rotary, appended
k/v) are rejected here too.
"""
from typing import Optional, Union

import torch


def is_fa3_supported(device=None) -> bool:
    """The reference gates on Xe2 (flash_attn.py:7-24); this build answers for gfx950."""
    from sgl_kernel.utils import is_gfx950_arch

    return is_gfx950_arch()


def maybe_contiguous(x):
    return x.contiguous() if x is not None and x.stride(-1) != 1 else x


def flash_attn_with_kvcache(
    q,
    k_cache,
    v_cache,
    k=None,
    v=None,
    qv=None,
    rotary_cos=None,
    rotary_sin=None,
    cache_seqlens: Optional[Union[(int, torch.Tensor)]] = None,
    cache_batch_idx: Optional[torch.Tensor] = None,
    cache_leftpad: Optional[torch.Tensor] = None,
    page_table: Optional[torch.Tensor] = None,
    cu_seqlens_q: Optional[torch.Tensor] = None,
    cu_seqlens_k_new: Optional[torch.Tensor] = None,
    max_seqlen_q: Optional[int] = 0,
    max_seqlen_k: Optional[int] = 0,
    rotary_seqlens: Optional[torch.Tensor] = None,
    q_descale: Optional[torch.Tensor] = None,
    k_descale: Optional[torch.Tensor] = None,
    v_descale: Optional[torch.Tensor] = None,
    softmax_scale=None,
    sinks=None,
    causal=False,
    window_size=(-1, -1),  # -1 means infinite context window
    softcap=0.0,  # 0.0 means deactivated
    rotary_interleaved=True,
    scheduler_metadata=None,
    num_splits=0,  # Can be tuned for speed
    pack_gqa=None,  # Can be tuned for speed
    sm_margin=0,  # Can be tuned if some SMs are used for communication
    return_softmax_lse=False,
    out=None,
):
    """Attention of q against a (paged) KV cache; see the reference docstring (flash_attn.py:137-231)
    for the argument meanings. q: (batch, seqlen, nheads, headdim) or ragged (total_q, nheads, headdim)
    with cu_seqlens_q; k_cache / v_cache: (num_blocks, page_block_size, nheads_k, headdim) with
    page_table (batch, max_blocks) int32 and cache_seqlens (batch,) int32. Causal masks are aligned
    to the bottom-right corner. Returns out (total_q, nheads, headdim) [, softmax_lse (nheads, total_q)]."""
    assert k_cache.stride(-1) == 1, "k_cache must have contiguous last dimension"
    assert v_cache.stride(-1) == 1, "v_cache must have contiguous last dimension"
    assert k is None and v is None, "appending new k/v to the cache inside the kernel is not supported"
    if softmax_scale is None:
        softmax_scale = (q.shape[-1] + (qv.shape[-1] if qv is not None else 0)) ** (-0.5)
    if cache_seqlens is not None and isinstance(cache_seqlens, int):
        cache_seqlens = torch.full((k_cache.shape[0],), cache_seqlens, dtype=torch.int32, device=k_cache.device)
        cache_seqlens = maybe_contiguous(cache_seqlens)

    q, k_cache = [maybe_contiguous(x) for x in (q, k_cache)]
    v_cache = v_cache.contiguous() if v_cache.stride(-1) != 1 and v_cache.stride(-3) != 1 else v_cache
    cu_seqlens_q, cu_seqlens_k_new = [maybe_contiguous(x) for x in (cu_seqlens_q, cu_seqlens_k_new)]
    page_table, cache_batch_idx, cache_leftpad = [maybe_contiguous(x) for x in (page_table, cache_batch_idx, cache_leftpad)]
    rotary_cos, rotary_sin = [maybe_contiguous(x) for x in (rotary_cos, rotary_sin)]
    rotary_seqlens = maybe_contiguous(rotary_seqlens)

    if cu_seqlens_q is None:  # !is_varlen_q
        cu_seqlens_q = torch.arange(0, q.size(0) + 1, dtype=torch.int, device=q.device) * q.size(1)
        max_seqlen_q = q.size(1)
        q = q.view(-1, q.size(-2), q.size(-1)).contiguous()
    assert cache_seqlens is not None, "cache_seqlens is required"
    assert cache_seqlens.size(0) + 1 == cu_seqlens_q.size(0)
    cu_seqlens_k = cache_seqlens
    out, softmax_lse, *rest = torch.ops.sgl_kernel.fwd.default(
        q,
        k_cache,
        v_cache,
        qv,
        cu_seqlens_q,
        cu_seqlens_k,
        max_seqlen_q,
        max_seqlen_k,
        page_table,
        cache_batch_idx,
        cache_leftpad,
        rotary_cos,
        rotary_sin,
        rotary_seqlens,
        q_descale,
        k_descale,
        v_descale,
        softmax_scale,
        sinks,
        causal,
        window_size[0],
        window_size[1],
        softcap,
        rotary_interleaved,
        scheduler_metadata,
        num_splits,
        pack_gqa,
        sm_margin,
        out,
    )
    return (out, softmax_lse, *rest) if return_softmax_lse else out


def flash_attn_varlen_func(
    q,
    k,
    v,
    cu_seqlens_q,
    cu_seqlens_k,
    max_seqlen_q,
    max_seqlen_k,
    seqused_q=None,
    seqused_k=None,
    softmax_scale=None,
    sinks=None,
    causal=False,
    qv=None,
    q_descale=None,
    k_descale=None,
    v_descale=None,
    window_size=(-1, -1),
    softcap=0.0,
    num_splits=0,
    pack_gqa=None,
    sm_margin=0,
    return_softmax_lse=False,
):
    """Ragged (non-paged) attention: q (total_q, nheads, d), k / v (total_k, nheads_k, d) with
    cumulative cu_seqlens_q / cu_seqlens_k (batch + 1,) int32."""
    if not is_fa3_supported():
        raise NotImplementedError("flash_attn of this build is only supported on gfx950 (MI355X)")

    if softmax_scale is None:
        softmax_scale = (q.shape[-1] + (qv.shape[-1] if qv is not None else 0)) ** (-0.5)
    if cu_seqlens_q is None:  # !is_varlen_q
        cu_seqlens_q = torch.arange(0, q.size(0) + 1, dtype=torch.int, device=q.device) * q.size(1)
        max_seqlen_q = q.size(1)
        q = q.view(-1, q.size(-2), q.size(-1)).contiguous()

    out, softmax_lse, *rest = torch.ops.sgl_kernel.fwd.default(
        q,
        k,
        v,
        qv,  # qv
        cu_seqlens_q,
        cu_seqlens_k,
        max_seqlen_q,
        max_seqlen_k,
        None,  # page_table,
        None,  # kv_batch_idx
        None,  # leftpad_k
        None,  # rotary cos
        None,  # rotary sin
        None,  # rotary_seqlens
        q_descale,
        k_descale,
        v_descale,
        softmax_scale,
        sinks,
        causal,
        window_size[0],
        window_size[1],
        softcap,
        False,  # rotary_interleaved
        None,  # scheduler_metadata
        num_splits,
        pack_gqa,
        sm_margin,
    )

    return (out, softmax_lse, *rest) if return_softmax_lse else out
