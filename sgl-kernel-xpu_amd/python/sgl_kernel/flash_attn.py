"""flash_attn_with_kvcache / flash_attn_varlen_func over torch.ops.sgl_kernel.fwd.

Contract (reference python/sgl_kernel/flash_attn.py): public names, parameter names and defaults of
`flash_attn_with_kvcache` (:103-136) and `flash_attn_varlen_func` (:301-324); the 29-slot positional
order of the `fwd` op (src/torch_extension_sycl.cc:328-357); default softmax_scale
(headdim + headdim of qv)^-0.5 (:233-236); a padded 4-D q is flattened to ragged [b*s, h, d] with
an arange cu_seqlens_q (:258-263); `cache_seqlens` (per-sequence lengths, NOT cumulative) travels in
the op's `cu_seqlens_k` slot (:264-266); an int `cache_seqlens` is broadcast to one int32 per cache
row (:238-242). Features the reference parses but its kernels reject (qv, rotary inside the kernel,
appending k/v) are rejected here as well.

Everything below is this build's own structure: the op call is assembled by slot NAME from a table,
so the two entry points only state what they fill in.
"""
from typing import Optional, Union

import torch

# positional slots of sgl_kernel::fwd, in schema order, with the value an entry point gets when it says nothing
_FWD_SLOTS = (
    ("q", None), ("k", None), ("v", None), ("q_v", None),
    ("cu_seqlens_q", None), ("cu_seqlens_k", None), ("max_seqlen_q", 0), ("max_seqlen_k", 0),
    ("page_table", None), ("kv_batch_idx", None), ("leftpad_k", None),
    ("rotary_cos", None), ("rotary_sin", None), ("seqlens_rotary", None),
    ("q_descale", None), ("k_descale", None), ("v_descale", None),
    ("softmax_scale", None), ("sinks", None), ("is_causal", False),
    ("window_size_left", -1), ("window_size_right", -1), ("softcap", 0.0),
    ("is_rotary_interleaved", False), ("scheduler_metadata", None), ("num_kv_splits", 0),
    ("pack_gqa", None), ("sm_margin", 0), ("out", None),
)
_FWD_NAMES = frozenset(n for n, _ in _FWD_SLOTS)


def _fwd(*positional):
    """The one place the op is called (tests patch this to look at the assembled arguments)."""
    return torch.ops.sgl_kernel.fwd.default(*positional)


def _call_fwd(return_lse, **filled):
    unknown = set(filled) - _FWD_NAMES
    assert not unknown, f"not a slot of sgl_kernel::fwd: {sorted(unknown)}"
    res = _fwd(*[filled.get(name, default) for name, default in _FWD_SLOTS])
    return tuple(res) if return_lse else res[0]


def is_fa3_supported(device=None) -> bool:
    """Reference flash_attn.py:7-24 answers for Xe2; this build answers for gfx950."""
    from sgl_kernel.utils import is_gfx950_arch

    return is_gfx950_arch()


def _unit_stride_last(t):
    """None stays None; a tensor whose innermost stride is not 1 is copied."""
    if t is None or t.stride(-1) == 1:
        return t
    return t.contiguous()


def _scale_or_default(softmax_scale, q, qv):
    if softmax_scale is not None:
        return softmax_scale
    width = q.shape[-1] + (0 if qv is None else qv.shape[-1])
    return width ** (-0.5)


def _as_ragged(q, cu_seqlens_q, max_seqlen_q):
    """(q [total, h, d], cu_seqlens_q, max_seqlen_q). A padded [b, s, h, d] batch becomes ragged rows with equal lengths."""
    if cu_seqlens_q is not None:
        # ragged q without a usable bound: the row count is always one (no device read-back: graph-capture safe);
        # it only costs the tuned decode / prefill kernel choice, never correctness
        if max_seqlen_q is None or max_seqlen_q <= 0 or max_seqlen_q > q.size(0):
            max_seqlen_q = q.size(0)
        return q, cu_seqlens_q, max_seqlen_q
    b, s = q.size(0), q.size(1)
    cu = torch.arange(0, b + 1, dtype=torch.int32, device=q.device) * s
    return q.reshape(b * s, q.size(-2), q.size(-1)).contiguous(), cu, s


def flash_attn_with_kvcache(q, k_cache, v_cache, k=None, v=None, qv=None, rotary_cos=None, rotary_sin=None,
                            cache_seqlens: Optional[Union[(int, torch.Tensor)]] = None,
                            cache_batch_idx: Optional[torch.Tensor] = None,
                            cache_leftpad: Optional[torch.Tensor] = None, page_table: Optional[torch.Tensor] = None,
                            cu_seqlens_q: Optional[torch.Tensor] = None,
                            cu_seqlens_k_new: Optional[torch.Tensor] = None, max_seqlen_q: Optional[int] = 0,
                            max_seqlen_k: Optional[int] = 0, rotary_seqlens: Optional[torch.Tensor] = None,
                            q_descale: Optional[torch.Tensor] = None, k_descale: Optional[torch.Tensor] = None,
                            v_descale: Optional[torch.Tensor] = None, softmax_scale=None, sinks=None, causal=False,
                            window_size=(-1, -1), softcap=0.0, rotary_interleaved=True, scheduler_metadata=None,
                            num_splits=0, pack_gqa=None, sm_margin=0, return_softmax_lse=False, out=None):
    """Attention of q against a KV cache.

    q: [batch, seqlen_q, heads, d], or ragged [total_q, heads, d] together with cu_seqlens_q [batch + 1].
    k_cache / v_cache: paged [pages, page_size, heads_k, d] with page_table [batch, max_pages] int32, or (no
    page_table) one row per cache slot [slots, seqlen_cache, heads_k, d], optionally indexed by cache_batch_idx
    [batch] and left-padded by cache_leftpad [batch]. cache_seqlens: valid keys per sequence (int32 [batch], or one
    int for all). Causal / sliding-window masks are aligned to the bottom-right corner of the score matrix.
    Returns out [total_q, heads, d] (and softmax_lse [heads, total_q] etc. with return_softmax_lse)."""
    for name, t in (("k_cache", k_cache), ("v_cache", v_cache)):
        assert t.stride(-1) == 1, f"{name} must have contiguous last dimension"
    assert k is None and v is None, "appending new k/v to the cache inside the kernel is not supported"
    assert cache_seqlens is not None, "cache_seqlens is required"

    scale = _scale_or_default(softmax_scale, q, qv)
    if isinstance(cache_seqlens, int):
        cache_seqlens = torch.full((k_cache.shape[0],), cache_seqlens, dtype=torch.int32, device=k_cache.device)
    q, cu_q, max_q = _as_ragged(_unit_stride_last(q), _unit_stride_last(cu_seqlens_q), max_seqlen_q)
    assert cache_seqlens.size(0) + 1 == cu_q.size(0), "cache_seqlens must hold one length per sequence of q"

    return _call_fwd(
        return_softmax_lse,
        q=q, k=k_cache, v=v_cache, q_v=qv,
        cu_seqlens_q=cu_q, cu_seqlens_k=cache_seqlens,  # lengths, not offsets: the op knows (see module docstring)
        max_seqlen_q=max_q, max_seqlen_k=max_seqlen_k,
        page_table=_unit_stride_last(page_table),
        kv_batch_idx=_unit_stride_last(cache_batch_idx), leftpad_k=_unit_stride_last(cache_leftpad),
        rotary_cos=_unit_stride_last(rotary_cos), rotary_sin=_unit_stride_last(rotary_sin),
        seqlens_rotary=_unit_stride_last(rotary_seqlens),
        q_descale=q_descale, k_descale=k_descale, v_descale=v_descale,
        softmax_scale=scale, sinks=sinks, is_causal=causal,
        window_size_left=window_size[0], window_size_right=window_size[1], softcap=softcap,
        is_rotary_interleaved=rotary_interleaved, scheduler_metadata=scheduler_metadata,
        num_kv_splits=num_splits, pack_gqa=pack_gqa, sm_margin=sm_margin, out=out,
    )


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, seqused_q=None,
                           seqused_k=None, softmax_scale=None, sinks=None, causal=False, qv=None, q_descale=None,
                           k_descale=None, v_descale=None, window_size=(-1, -1), softcap=0.0, num_splits=0,
                           pack_gqa=None, sm_margin=0, return_softmax_lse=False):
    """Ragged attention without a cache: q [total_q, heads, d], k / v [total_k, heads_k, d], cumulative
    cu_seqlens_q / cu_seqlens_k [batch + 1] int32 (here the k slot carries real offsets)."""
    if not is_fa3_supported():
        raise NotImplementedError("flash_attn of this build is only supported on gfx950 (MI355X)")
    scale = _scale_or_default(softmax_scale, q, qv)
    q, cu_q, max_q = _as_ragged(q, cu_seqlens_q, max_seqlen_q)
    return _call_fwd(
        return_softmax_lse,
        q=q, k=k, v=v, q_v=qv, cu_seqlens_q=cu_q, cu_seqlens_k=cu_seqlens_k,
        max_seqlen_q=max_q, max_seqlen_k=max_seqlen_k,
        q_descale=q_descale, k_descale=k_descale, v_descale=v_descale,
        softmax_scale=scale, sinks=sinks, is_causal=causal,
        window_size_left=window_size[0], window_size_right=window_size[1], softcap=softcap,
        num_kv_splits=num_splits, pack_gqa=pack_gqa, sm_margin=sm_margin,
    )
