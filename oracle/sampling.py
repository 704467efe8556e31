"""TEST INFRASTRUCTURE (CPU oracle) — top-k / top-p / min-p filters of the sampling ops.

Restates the filter definitions the reference's tests pin (tests/test_sampling.py:13-34 joint mask, :104-126 top-p
renorm, :176-204 top-k renorm, :262-274 min-p mask) with explicit pivots in float64 / exact comparisons:
  top-k : x >= (k-th largest value)
  top-p : x >= t_p, t_p = the smallest value of the shortest descending prefix whose mass reaches p (everything if p >= 1)
  min-p : x >= min_p * max(x)
Sampling itself is random: the GPU tests check membership of every draw in these masks, reproducibility from the
generator state and the empirical distribution. Pinned by tests/test_oracle_golden.py on vectors from the reference
functions (tests/golden/make_golden.py: gen_sampling)."""
import torch


def _rowwise(v, rows, dtype):
    if isinstance(v, torch.Tensor):
        return v.to(dtype).view(rows)
    return torch.full((rows,), v, dtype=dtype)


def top_k_mask(probs, k):
    rows, V = probs.shape
    k = _rowwise(k, rows, torch.int64).clamp(max=V)
    srt = torch.sort(probs, dim=-1, descending=True).values
    pivot = srt.gather(1, (k - 1).clamp(min=0).unsqueeze(1))
    return probs >= pivot


def top_p_mask(probs, p):
    rows, V = probs.shape
    p = _rowwise(p, rows, torch.float64)
    srt = torch.sort(probs.double(), dim=-1, descending=True).values
    cum = torch.cumsum(srt, dim=-1)
    reach = (cum >= p.unsqueeze(1))
    first = torch.where(reach.any(dim=1), reach.to(torch.int8).argmax(dim=1), torch.full((rows,), V - 1))
    pivot = srt.gather(1, first.unsqueeze(1)).float()
    mask = probs >= pivot
    mask[p >= 1.0] = True
    return mask


def min_p_mask(probs, min_p):
    rows, _ = probs.shape
    mp = _rowwise(min_p, rows, torch.float32)
    return probs >= (mp * probs.max(dim=-1).values).unsqueeze(1)


def renorm(probs, mask):
    kept = torch.where(mask, probs, torch.zeros_like(probs))
    return kept / kept.sum(dim=-1, keepdim=True)
