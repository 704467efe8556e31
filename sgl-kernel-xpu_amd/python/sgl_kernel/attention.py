"""Multi-head latent attention (DeepSeek MLA) over a paged latent cache: decode and ragged prefill.

Contract (reference python/sgl_kernel/attention.py): the public names and parameters of `flash_mla_decode`
(:54-62), `flash_mla_get_workspace_size` (:128-146), `flash_mla_prefill` (:149-161) and
`flash_mla_prefill_get_workspace_size` (:236-247); latent width 512 + rope width 64, at most 128 heads, 16-bit
dtypes, int32 `seq_lens` / `page_table` (:77-111); a page table whose width covers whole 128-token spans (:101);
output [rows, heads, 512] in the query dtype (:113-119); the op argument order of
src/torch_extension_sycl.cc:364-383.

Deliberate differences: q_nope / q_pe are handed to the kernel with their strides (only a non-unit innermost
stride is copied), the reference's pad-to-128-heads CUDA branch (:86-93) and its 256-row output padding for
prefill (:212-217: its epilogue writes whole tiles, this one writes exactly total_q rows) do not exist here.
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

LATENT_DIM = 512
ROPE_DIM = 64
MAX_HEADS = 128
_HALF_TYPES = (torch.float16, torch.bfloat16)

_ops = torch.ops.sgl_kernel  # (patched by the host-logic tests)


@dataclass(frozen=True)
class _MlaShapes:
    rows: int  # batch (decode) or total query tokens (prefill)
    heads: int
    page_size: int


def _check_operands(who, q_nope, q_pe, cache) -> _MlaShapes:
    """Rank / width / dtype rules shared by decode and prefill."""
    for name, t in (("q_nope", q_nope), ("q_pe", q_pe), ("kv_c_and_k_pe_cache", cache)):
        assert t.ndim == 3, f"{who}: {name} must be a 3D tensor, but got {t.ndim}"
    rows, heads, d_nope = q_nope.shape
    assert q_pe.shape[:2] == (rows, heads), f"{who}: q_nope {tuple(q_nope.shape)} and q_pe {tuple(q_pe.shape)} disagree"
    assert d_nope == LATENT_DIM, f"{who}: q_nope last dim must be {LATENT_DIM}, got {d_nope}"
    assert q_pe.shape[2] == ROPE_DIM, f"{who}: q_pe last dim must be {ROPE_DIM}, got {q_pe.shape[2]}"
    assert cache.shape[2] == LATENT_DIM + ROPE_DIM, (
        f"{who}: cache rows must be {LATENT_DIM} + {ROPE_DIM} wide, got {cache.shape[2]}")
    assert heads <= MAX_HEADS, f"{who}: H must be <= {MAX_HEADS}, but got {heads}"
    assert q_nope.dtype in _HALF_TYPES, f"{who}: q_nope.dtype needs to be fp16 or bf16 but got {q_nope.dtype}."
    assert q_nope.dtype == q_pe.dtype == cache.dtype, f"{who}: q_nope, q_pe and the cache must share one dtype"
    return _MlaShapes(rows, heads, cache.shape[1])


def _check_int32(who, **tensors):
    for name, t in tensors.items():
        assert t.dtype == torch.int32, f"{who}: {name}.dtype needs to be int32 but got {t.dtype}."


def _check_positive(who, **values):
    for name, v in values.items():
        assert v > 0, f"{who}: {name} must be greater than 0, got {v}"


def _merge(op_name, v_a, s_a, v_b, s_b, v_merged, s_merged):
    s_a, s_b = s_a.to(torch.float32), s_b.to(torch.float32)
    v_merged = torch.empty_like(v_a) if v_merged is None else v_merged
    s_merged = torch.empty_like(s_a) if s_merged is None else s_merged
    getattr(_ops, op_name).default(v_a, s_a, v_b, s_b, v_merged, s_merged)
    return v_merged, s_merged


def merge_state(v_a: torch.Tensor, s_a: torch.Tensor, v_b: torch.Tensor, s_b: torch.Tensor,
                v_merged: Optional[torch.Tensor] = None,
                s_merged: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge two partial attention states over disjoint key sets: v [tokens, heads, d] normalised outputs, s
    [tokens, heads] log-sum-exp in BASE 2 (the flashinfer convention; reference attention.py:12-28). Outputs are
    allocated unless given."""
    return _merge("merge_state", v_a, s_a, v_b, s_b, v_merged, s_merged)


def merge_state_v2(v_a: torch.Tensor, s_a: torch.Tensor, v_b: torch.Tensor, s_b: torch.Tensor,
                   v_merged: Optional[torch.Tensor] = None,
                   s_merged: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The same merge with natural-log s (what `fwd` returns as softmax_lse; reference attention.py:31-51)."""
    return _merge("merge_state_v2", v_a, s_a, v_b, s_b, v_merged, s_merged)


def _inner_unit_stride(t):
    return t if t.stride(-1) == 1 else t.contiguous()


def flash_mla_decode(q_nope: torch.Tensor, q_pe: torch.Tensor, kv_c_and_k_pe_cache: torch.Tensor,
                     seq_lens: torch.Tensor, page_table: torch.Tensor, workspace: torch.Tensor, sm_scale: float,
                     num_kv_splits: int = 1) -> torch.Tensor:
    """One query token per sequence: out[b, h, :] = softmax(sm_scale * [q_nope, q_pe][b, h] . cache[b]^T) . cache[b][:, :512]
    over the first seq_lens[b] tokens of the pages page_table[b] names. `workspace`: uint8 scratch of
    flash_mla_get_workspace_size(...) bytes (split-KV partials)."""
    who = "flash_mla_decode"
    sh = _check_operands(who, q_nope, q_pe, kv_c_and_k_pe_cache)
    assert page_table.ndim == 2 and page_table.shape[0] == sh.rows, (
        f"{who}: page_table must be [batch={sh.rows}, pages], got {tuple(page_table.shape)}")
    pages = page_table.shape[1]
    assert pages > 0, f"{who}: block num must be greater than 0, got {pages}"
    assert (pages * sh.page_size) % 128 == 0, (
        f"{who}: the page table must cover whole 128-token spans ({pages} pages of {sh.page_size})")
    _check_int32(who, seq_lens=seq_lens, page_table=page_table)

    out = q_nope.new_empty((sh.rows, sh.heads, LATENT_DIM))
    _ops.flash_mla_decode.default(out, _inner_unit_stride(q_nope), _inner_unit_stride(q_pe), kv_c_and_k_pe_cache,
                                  seq_lens, page_table, workspace, sm_scale, num_kv_splits)
    return out


def flash_mla_get_workspace_size(max_seq_len: int, num_batches: int, num_heads: int = 0, page_size: int = 0,
                                 num_kv_splits: int = -1) -> int:
    _check_positive("flash_mla_get_workspace_size", max_seq_len=max_seq_len, num_batches=num_batches)
    return _ops.flash_mla_get_workspace_size.default(max_seq_len, num_batches, num_heads, page_size, num_kv_splits)


def flash_mla_prefill(q_nope: torch.Tensor, q_pe: torch.Tensor, kv_c_and_k_pe_cache: torch.Tensor,
                      cu_seqlens_q: torch.Tensor, seq_lens_k: torch.Tensor, max_seqlen_q: int,
                      page_table: torch.Tensor, workspace: torch.Tensor, sm_scale: float, causal: bool = True,
                      num_kv_splits: int = -1) -> torch.Tensor:
    """Ragged prefill: sequence i owns query rows cu_seqlens_q[i] : cu_seqlens_q[i+1] and the first seq_lens_k[i]
    cached tokens (its new tokens are the LAST ones: the causal mask is aligned bottom-right).
    Returns out [total_q, heads, 512]."""
    who = "flash_mla_prefill"
    sh = _check_operands(who, q_nope, q_pe, kv_c_and_k_pe_cache)
    _check_int32(who, cu_seqlens_q=cu_seqlens_q, seq_lens_k=seq_lens_k, page_table=page_table)
    batch = cu_seqlens_q.shape[0] - 1
    assert seq_lens_k.shape[0] == batch, f"{who}: seq_lens_k must hold one length per sequence ({batch})"

    out = q_nope.new_empty((sh.rows, sh.heads, LATENT_DIM))
    if sh.rows > 0:
        _ops.flash_mla_prefill.default(out, q_nope, q_pe, kv_c_and_k_pe_cache, cu_seqlens_q, seq_lens_k.contiguous(),
                                       max_seqlen_q, page_table, workspace, sm_scale, causal, num_kv_splits)
    return out


def flash_mla_prefill_get_workspace_size(max_seq_len: int, num_batches: int, num_heads: int = 0, page_size: int = 0,
                                         num_kv_splits: int = -1) -> int:
    """This library's prefill keeps everything on chip: the answer is 0 (the op is still asked, as the reference does)."""
    _check_positive("flash_mla_prefill_get_workspace_size", max_seq_len=max_seq_len, num_batches=num_batches)
    return _ops.flash_mla_prefill_get_workspace_size.default(max_seq_len, num_batches, num_heads, page_size, num_kv_splits)
