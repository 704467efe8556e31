"""Top-k / top-p / min-p filtering and sampling from probabilities.

Contract (reference python/sgl_kernel/sampling.py): `top_k_renorm_probs` (:21-50, alias `top_k_renorm_prob`),
`top_p_renorm_probs` (:69-99, alias `top_p_renorm_prob`), `top_p_sampling_from_probs` (:131-185),
`top_k_top_p_sampling_from_probs` (:219-302, filter_apply_order "top_k_first" | "joint"), `min_p_sampling_from_probs`
(:332-386): a threshold is a python scalar or a per-row tensor; probabilities are computed on in float32, samples are
int32 [batch]; `check_nan` raises ValueError on NaN input; `indices` maps output row b to probs row indices[b].

On this build the filters are exact (radix select of the pivots), and a draw is reproducible from the generator's
(seed, offset): see csrc/sampling.hip.
"""
from typing import Optional, Union

import torch

_ops = torch.ops.sgl_kernel  # (patched by the host-logic tests)


def _split(threshold):
    """per-row tensor or scalar -> (tensor or None, scalar)"""
    return (threshold, 0) if isinstance(threshold, torch.Tensor) else (None, threshold)


def _refuse_nan(probs, check_nan):
    if check_nan and torch.any(torch.isnan(probs)):
        raise ValueError("Input probs contains NaN.")


def top_k_renorm_probs(probs: torch.Tensor, top_k: Union[torch.Tensor, int]) -> torch.Tensor:
    """Keep each row's top-k probabilities (ties with the k-th value included), zero the rest, renormalise."""
    arr, val = _split(top_k)
    probs = probs.float()
    out = torch.empty_like(probs)
    _ops.top_k_renorm_probs.default(probs, out, arr.long() if arr is not None else None, val)
    return out


def top_p_renorm_probs(probs: torch.Tensor, top_p: Union[torch.Tensor, float]) -> torch.Tensor:
    """Keep each row's nucleus (the largest probabilities whose mass reaches top_p), zero the rest, renormalise."""
    arr, val = _split(top_p)
    probs = probs.float()
    out = torch.empty_like(probs)
    _ops.top_p_renorm_probs.default(probs, out, arr.float() if arr is not None else None, val)
    return out


top_k_renorm_prob = top_k_renorm_probs
top_p_renorm_prob = top_p_renorm_probs


def _samples_like(probs, indices):
    rows = indices.size(0) if indices is not None else probs.size(0)
    return torch.empty(rows, dtype=torch.int32, device=probs.device)


def top_p_sampling_from_probs(probs: torch.Tensor, top_p: Union[torch.Tensor, float], indices: Optional[torch.Tensor] = None,
                              deterministic: bool = True, generator: Optional[torch.Generator] = None,
                              check_nan: bool = False) -> torch.Tensor:
    _refuse_nan(probs, check_nan)
    arr, val = _split(top_p)
    probs = probs.float()
    samples = _samples_like(probs, indices)
    _ops.top_p_sampling_from_probs.default(probs, samples, indices, arr.float() if arr is not None else None, val, deterministic,
                                           generator)
    return samples


def top_k_top_p_sampling_from_probs(probs: torch.Tensor, top_k: Union[torch.Tensor, int], top_p: Union[torch.Tensor, float],
                                    indices: Optional[torch.Tensor] = None, filter_apply_order: str = "top_k_first",
                                    deterministic: bool = True, generator: Optional[torch.Generator] = None,
                                    check_nan: bool = False) -> torch.Tensor:
    """"top_k_first": top-k renormalisation, then top-p sampling on the result; "joint": one draw from the rows whose
    probability passes both filters of the ORIGINAL distribution."""
    if filter_apply_order == "top_k_first":
        return top_p_sampling_from_probs(top_k_renorm_probs(probs, top_k), top_p, indices, deterministic, generator=generator,
                                         check_nan=check_nan)
    if filter_apply_order != "joint":
        raise ValueError(f"Invalid filter_apply_order: {filter_apply_order}")
    _refuse_nan(probs, check_nan)
    k_arr, k_val = _split(top_k)
    p_arr, p_val = _split(top_p)
    probs = probs.float()
    samples = _samples_like(probs, indices)
    _ops.top_k_top_p_sampling_from_probs.default(probs, samples, indices, k_arr.int() if k_arr is not None else None, k_val,
                                                 p_arr.float() if p_arr is not None else None, p_val, deterministic, generator)
    return samples


def min_p_sampling_from_probs(probs: torch.Tensor, min_p: Union[torch.Tensor, float], indices: Optional[torch.Tensor] = None,
                              deterministic: bool = True, generator: Optional[torch.Generator] = None,
                              check_nan: bool = False) -> torch.Tensor:
    """One draw per row from the probabilities that are at least min_p times the row's largest one."""
    _refuse_nan(probs, check_nan)
    arr, val = _split(min_p)
    probs = probs.float()
    samples = _samples_like(probs, indices)
    _ops.min_p_sampling_from_probs.default(probs, samples, indices, arr.float() if arr is not None else None, val, deterministic,
                                           generator)
    return samples
