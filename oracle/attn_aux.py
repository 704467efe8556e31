"""TEST INFRASTRUCTURE (CPU oracle) — merge_state / merge_state_v2 / store_cache.

Restates the reference kernels' arithmetic in fp32 torch (never imported by the product path):
  * merge_state_v2: MergePrefixSuffix, reference src/sycl/merge_states.cpp:76-137 (natural log)
  * merge_state   : MergeState, reference src/sycl/merge_states.cpp:208-272 (base 2)
  * store_cache   : StoreCacheKernel, reference src/sycl/KVCache.cpp:11-73 (a copy; index < 0 skips)
Pinned by tests/test_oracle_golden.py against vectors produced by the reference's own test reference
`merge_state_torch` (tests/test_merge_state_v2.py:101-137, imported in tests/golden/make_golden.py); the base-2 op
has only a Triton reference there, so it is pinned through the identity
merge_state(v, s) == merge_state_v2(v, s * ln 2) with s_out / ln 2."""
import math

import torch


def merge_state(v_a, s_a, v_b, s_b, base2):
    """v_* [tokens, heads, d] (any float dtype), s_* [tokens, heads] fp32 -> (v_merged in v's dtype, s_merged fp32)."""
    ninf = torch.tensor(float("-inf"))
    sa = torch.where(torch.isfinite(s_a), s_a.float(), ninf)  # merge_states.cpp:94-95 / :231-232
    sb = torch.where(torch.isfinite(s_b), s_b.float(), ninf)
    m = torch.maximum(sa, sb)
    if base2:
        wa, wb = torch.exp2(sa - m), torch.exp2(sb - m)
    else:
        wa, wb = torch.exp(sa - m), torch.exp(sb - m)
    z = torch.clamp(wa + wb, min=torch.finfo(torch.float32).tiny)  # fmax(.., FLT_MIN)
    ca, cb = (wa / z).unsqueeze(-1), (wb / z).unsqueeze(-1)
    v = (v_a.float() * ca + v_b.float() * cb).to(v_a.dtype)
    s = (torch.log2(z) if base2 else torch.log(z)) + m
    return v, s


def merge_state_base2_via_v2(v_a, s_a, v_b, s_b):
    """The base-2 merge expressed through the natural-log one (used to pin `merge_state` on the imported reference)."""
    ln2 = math.log(2.0)
    v, s = merge_state(v_a, s_a * ln2, v_b, s_b * ln2, base2=False)
    return v, s / ln2


def store_cache(k, v, k_cache, v_cache, indices):
    """In place on clones of the caches; returns (k_cache, v_cache)."""
    kc, vc = k_cache.clone(), v_cache.clone()
    for t, slot in enumerate(indices.tolist()):
        if slot >= 0:
            kc[slot] = k[t]
            vc[slot] = v[t]
    return kc, vc
