"""GPU parity: merge_state, merge_state_v2, store_cache, fused_inplace_qknorm_rope, fused_qk_norm_rope (SURVEY 8f
rank 3) vs the CPU oracle and the fixtures generated from the reference tests' own references.

Grids follow reference tests/test_merge_state_v2.py:139-143 (tokens x heads x head size x dtype), tests/test_merge_state.py,
tests/test_store_cache_xpu.py:21-160 and tests/test_fused_qk_norm_rope.py:222-727; tolerances are the reference's
(`precision`: 1e-2 bf16 / 1e-3 fp16; YaRN cases 2x)."""
import math

import pytest
import torch
from conftest import load_golden

from oracle import attn_aux as oaux
from oracle import qknorm_rope as oqk

pytestmark = pytest.mark.gpu
PREC = {torch.bfloat16: 1e-2, torch.float16: 1e-3, torch.float32: 1e-5}


def _states(tokens, heads, d, dt, seed):
    g = torch.Generator().manual_seed(seed)
    s_a, s_b = torch.randn(tokens, heads, generator=g), torch.randn(tokens, heads, generator=g)
    ma, mb = torch.rand(tokens, heads, generator=g) < 0.1, torch.rand(tokens, heads, generator=g) < 0.1
    both = ma & mb
    s_a[ma & ~both] = float("inf")  # "no keys on this side" as the reference test marks it
    s_b[mb & ~both] = float("inf")
    return torch.randn(tokens, heads, d, generator=g).to(dt), s_a, torch.randn(tokens, heads, d, generator=g).to(dt), s_b


@pytest.mark.parametrize("tokens", [1, 256, 512, 613, 1024, 1536])
@pytest.mark.parametrize("heads", [8, 16, 32])
@pytest.mark.parametrize("d", [32, 48, 64, 128, 256, 512])
@pytest.mark.parametrize("dt", [torch.half, torch.bfloat16, torch.float32])
def test_merge_state_both_bases(sglk, dev, tokens, heads, d, dt):
    v_a, s_a, v_b, s_b = _states(tokens, heads, d, dt, tokens + d)
    for fn, base2 in ((sglk.merge_state_v2, False), (sglk.merge_state, True)):
        v, s = fn(v_a.to(dev), s_a.to(dev), v_b.to(dev), s_b.to(dev))
        rv, rs = oaux.merge_state(v_a, s_a, v_b, s_b, base2)
        assert v.dtype == dt and s.dtype == torch.float32
        # one storage-dtype ulp around the oracle (the weights may differ in the last fp32 bit)
        torch.testing.assert_close(v.cpu().float(), rv.float(), rtol=PREC[dt], atol=PREC[dt])
        torch.testing.assert_close(s.cpu(), rs, rtol=1e-5, atol=1e-5)
        if dt != torch.float32:  # (fp32 outputs differ in the last bit wherever the device expf does)
            assert (v.cpu() != rv).float().mean() < 0.02


def test_merge_state_outputs_given_and_golden(sglk, dev):
    for c in load_golden("merge_state"):
        vm = torch.empty_like(c["v_a"], device=dev)
        sm = torch.empty_like(c["s_a"], device=dev)
        v, s = sglk.merge_state_v2(c["v_a"].to(dev), c["s_a"].to(dev), c["v_b"].to(dev), c["s_b"].to(dev), vm, sm)
        assert v.data_ptr() == vm.data_ptr() and s.data_ptr() == sm.data_ptr()
        p = PREC[c["v_a"].dtype]
        torch.testing.assert_close(v.cpu().float(), c["v_merged"].to(v.dtype).float(), rtol=p, atol=p)
        torch.testing.assert_close(s.cpu(), c["s_merged"], rtol=1e-5, atol=1e-5)
        ln2 = math.log(2.0)
        v2, s2 = sglk.merge_state(c["v_a"].to(dev), (c["s_a"] / ln2).to(dev), c["v_b"].to(dev), (c["s_b"] / ln2).to(dev))
        torch.testing.assert_close(v2.cpu().float(), c["v_merged"].to(v.dtype).float(), rtol=p, atol=p)
        torch.testing.assert_close(s2.cpu() * ln2, c["s_merged"], rtol=1e-5, atol=1e-5)


def test_merge_state_is_the_split_kv_merge(sglk, dev):
    """attention over keys [0, n) == merge of attention over [0, m) and [m, n) with the lse `fwd` returns"""
    g = torch.Generator().manual_seed(3)
    Hq, Hk, D, sk, m = 8, 2, 128, 777, 300
    q = torch.randn(4, Hq, D, generator=g).to(torch.bfloat16).to(dev)
    k = torch.randn(sk, Hk, D, generator=g).to(torch.bfloat16).to(dev)
    v = torch.randn(sk, Hk, D, generator=g).to(torch.bfloat16).to(dev)
    cu_q = torch.tensor([0, 4], dtype=torch.int32, device=dev)

    def attn(lo, hi):
        cu_k = torch.tensor([0, hi - lo], dtype=torch.int32, device=dev)
        o, lse, *_ = sglk.flash_attn_varlen_func(q, k[lo:hi].contiguous(), v[lo:hi].contiguous(), cu_q, cu_k, 4, hi - lo,
                                                 return_softmax_lse=True)
        return o, lse.t().contiguous()  # [Hq, tokens] -> [tokens, Hq]

    (oa, la), (ob, lb), (full, lf) = attn(0, m), attn(m, sk), attn(0, sk)
    merged, lse = sglk.merge_state_v2(oa, la, ob, lb)
    torch.testing.assert_close(merged.float(), full.float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(lse, lf, rtol=1e-4, atol=1e-4)


def test_merge_state_errors(sglk, dev):
    v = torch.zeros(2, 4, 20, dtype=torch.bfloat16, device=dev)
    s = torch.zeros(2, 4, device=dev)
    with pytest.raises(RuntimeError, match="multiple of pack_size"):
        sglk.merge_state_v2(v, s, v, s)
    with pytest.raises(RuntimeError, match="same shape"):
        sglk.merge_state(v, s, v[:1], s[:1])


# ------------------------------------------------------------------------------------------------ store_cache
@pytest.mark.parametrize("tokens", [1, 4, 32, 128, 271])
@pytest.mark.parametrize("row_dim", [128, 256, 512, 1024, 576, 40, 7])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32, torch.uint8])
def test_store_cache(sglk, dev, tokens, row_dim, dt):
    g = torch.Generator().manual_seed(tokens + row_dim)
    cache_size = 2048
    mk = (lambda *sh: torch.randint(0, 255, sh, generator=g, dtype=torch.uint8)) if dt == torch.uint8 else \
         (lambda *sh: torch.randn(*sh, generator=g).to(dt))
    k, v = mk(tokens, row_dim), mk(tokens, row_dim)
    idx = torch.randperm(cache_size, generator=g)[:tokens].to(torch.int64)
    idx[::5] = -1  # skipped tokens
    kc0, vc0 = mk(cache_size, row_dim), mk(cache_size, row_dim)
    kc, vc = kc0.to(dev), vc0.to(dev)
    sglk.store_cache_xpu(k.to(dev), v.to(dev), kc, vc, idx.to(dev))
    rk, rv = oaux.store_cache(k, v, kc0, vc0, idx)
    assert torch.equal(kc.cpu().view(torch.uint8), rk.view(torch.uint8)) and torch.equal(vc.cpu().view(torch.uint8), rv.view(torch.uint8))


@pytest.mark.parametrize("heads,head", [(2, 0), (2, 1), (10, 1)])
def test_store_cache_strided_rows_and_int32_indices(sglk, dev, heads, head):
    g = torch.Generator().manual_seed(123)
    tokens, row_dim, cache_size = 271, 256, 2048
    kw = torch.randn(tokens, heads, row_dim, generator=g).to(torch.bfloat16).to(dev)
    vw = torch.randn(tokens, heads, row_dim, generator=g).to(torch.bfloat16).to(dev)
    k, v = kw[:, head, :], vw[:, head, :]
    assert not k.is_contiguous()
    idx = torch.randperm(cache_size, generator=g)[:tokens].to(torch.int32).to(dev)  # the wrapper widens to int64
    kc = torch.zeros(cache_size, row_dim, dtype=torch.bfloat16, device=dev)
    vc = torch.zeros_like(kc)
    sglk.store_cache(k, v, kc, vc, idx)
    rk, rv = torch.zeros_like(kc), torch.zeros_like(vc)
    rk[idx.long()] = k
    rv[idx.long()] = v
    assert torch.equal(kc, rk) and torch.equal(vc, rv)


def test_store_cache_empty_and_errors(sglk, dev):
    kc = torch.zeros(16, 64, dtype=torch.bfloat16, device=dev)
    vc = torch.zeros_like(kc)
    e = torch.empty(0, 64, dtype=torch.bfloat16, device=dev)
    sglk.store_cache_xpu(e, e, kc, vc, torch.empty(0, dtype=torch.int64, device=dev))
    assert torch.all(kc == 0)
    k = torch.zeros(2, 64, dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError, match="row_dim"):
        sglk.store_cache_xpu(k[:, :32], k[:, :32], kc, vc, torch.zeros(2, dtype=torch.int64, device=dev))
    with pytest.raises(RuntimeError, match="same dtype"):
        sglk.store_cache_xpu(k, k.half(), kc, vc, torch.zeros(2, dtype=torch.int64, device=dev))


# --------------------------------------------------------------------------------------- fused qk-norm + rope
def _cos_sin_cache(rope_dim, max_pos, base=10000.0):
    inv = 1.0 / (base ** (torch.arange(0, rope_dim, 2, dtype=torch.float32) / rope_dim))
    f = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv)
    return torch.cat([f.cos(), f.sin()], dim=-1)


@pytest.mark.parametrize("layout", ["3d", "4d", "padded_rows", "qkv_slices"])
@pytest.mark.parametrize("hq,hk,d,rope", [(4, 2, 64, 32), (8, 4, 128, 64), (16, 4, 128, 128), (32, 8, 256, 128),
                                          (4, 4, 64, 8), (4, 1, 128, 24)])
@pytest.mark.parametrize("neox", [True, False])
@pytest.mark.parametrize("dt,pdt", [(torch.bfloat16, torch.int32), (torch.float16, torch.int64), (torch.float32, torch.int64)])
def test_fused_inplace_qknorm_rope(sglk, dev, layout, hq, hk, d, rope, neox, dt, pdt):
    g = torch.Generator().manual_seed(hq * d + rope)
    b, sq = 3, 5
    tokens = b * sq
    q = torch.randn(tokens, hq, d, generator=g).to(dt)
    k = torch.randn(tokens, hk, d, generator=g).to(dt)
    qw, kw = torch.randn(d, generator=g).to(dt), torch.randn(d, generator=g).to(dt)
    pos = torch.randint(0, 4000, (tokens,), generator=g).to(pdt)
    cache = _cos_sin_cache(rope, 4000)
    rq, rk = oqk.fused_inplace_qknorm_rope(q, k, qw, kw, cache, pos, neox)
    if layout == "3d":
        qd, kd = q.to(dev), k.to(dev)
    elif layout == "4d":
        qd, kd = q.to(dev).view(b, sq, hq, d), k.to(dev).view(b, sq, hk, d)
    elif layout == "padded_rows":  # last dim padded in storage (reference :630-727)
        qs = torch.zeros(tokens, hq, d + 16, dtype=dt, device=dev)
        ks = torch.zeros(tokens, hk, d + 32, dtype=dt, device=dev)
        qd, kd = qs[..., :d], ks[..., :d]
        qd.copy_(q)
        kd.copy_(k)
    else:  # q and k as slices of one packed qkv buffer
        buf = torch.zeros(tokens, (hq + 2 * hk) * d, dtype=dt, device=dev)
        qd = buf[:, :hq * d].view(tokens, hq, d)
        kd = buf[:, hq * d:(hq + hk) * d].view(tokens, hk, d)
        qd.copy_(q)
        kd.copy_(k)
    sglk.fused_inplace_qknorm_rope(qd, kd, qw.to(dev), kw.to(dev), cache.to(dev), pos.to(dev), neox)
    p = PREC[dt]
    torch.testing.assert_close(qd.reshape(tokens, hq, d).cpu(), rq, rtol=p, atol=p)
    torch.testing.assert_close(kd.reshape(tokens, hk, d).cpu(), rk, rtol=p, atol=p)
    if layout == "qkv_slices":
        assert torch.all(buf[:, (hq + hk) * d:] == 0)  # V untouched


@pytest.mark.parametrize("tokens", [1, 7, 128])
@pytest.mark.parametrize("hq,hk,hv,d", [(8, 8, 8, 64), (32, 8, 8, 128), (8, 2, 2, 256)])
@pytest.mark.parametrize("neox", [True, False])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("yarn", [False, True])
def test_fused_qk_norm_rope(sglk, dev, tokens, hq, hk, hv, d, neox, dt, yarn):
    g = torch.Generator().manual_seed(tokens + d)
    qkv = torch.randn(tokens, (hq + hk + hv) * d, generator=g).to(dt)
    qw, kw = torch.randn(d, generator=g).to(dt), torch.randn(d, generator=g).to(dt)
    pos = torch.randint(0, 2000, (tokens,), generator=g).to(torch.int32)
    factor, low, high, af = (4.0, 1.0, 32.0, 1.2) if yarn else (1.0, 1.0, 1.0, 1.0)
    for rot in (d, d // 2):
        ref = oqk.fused_qk_norm_rope(qkv, hq, hk, hv, d, 1e-6, qw, kw, 10000.0, neox, pos, factor, low, high, af, rot)
        x = qkv.clone().to(dev)
        sglk.fused_qk_norm_rope(x, hq, hk, hv, d, 1e-6, qw.to(dev), kw.to(dev), 10000.0, neox, pos.to(dev), factor, low, high, af, rot)
        p = PREC[dt] * (2 if yarn else 1)
        torch.testing.assert_close(x.cpu(), ref, rtol=p, atol=p)
        assert torch.equal(x.cpu()[:, (hq + hk) * d:], qkv[:, (hq + hk) * d:])  # V bit-identical


def test_qknorm_rope_golden(sglk, dev):
    gold = load_golden("qknorm_rope")
    for c in gold["cache"]:
        q, k = c["q"].clone().to(dev), c["k"].clone().to(dev)
        sglk.fused_inplace_qknorm_rope(q, k, c["q_weight"].to(dev), c["k_weight"].to(dev), c["cos_sin_cache"].to(dev),
                                       c["positions"].to(dev), c["is_neox"])
        p = PREC[q.dtype]
        torch.testing.assert_close(q.cpu(), c["q_out"].to(q.dtype), rtol=p, atol=p)
        torch.testing.assert_close(k.cpu(), c["k_out"].to(k.dtype), rtol=p, atol=p)
    for c in gold["yarn"]:
        x = c["qkv"].clone().to(dev)
        sglk.fused_qk_norm_rope(x, c["Hq"], c["Hk"], c["Hv"], c["head_dim"], c["eps"], c["q_weight"].to(dev),
                                c["k_weight"].to(dev), c["base"], c["is_neox"], c["position_ids"].to(dev), c["factor"],
                                c["low"], c["high"], c["attention_factor"], c["rotary_dim"])
        p = PREC[x.dtype] * (2 if c["factor"] != 1.0 else 1)
        torch.testing.assert_close(x.cpu(), c["out"].to(x.dtype), rtol=p, atol=p)


def test_qknorm_rope_errors(sglk, dev):
    q = torch.zeros(2, 4, 96, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(96, dtype=torch.bfloat16, device=dev)
    cache = torch.zeros(8, 32, device=dev)
    pos = torch.zeros(2, dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError, match="Unsupported head dimension"):
        sglk.fused_inplace_qknorm_rope(q, q.clone(), w, w, cache, pos, True)
    q = torch.zeros(2, 4, 64, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(64, dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError, match="float32"):
        sglk.fused_inplace_qknorm_rope(q, q.clone(), w, w, cache.half(), pos, True)
    with pytest.raises(RuntimeError, match="rope_dim must match"):
        sglk.fused_inplace_qknorm_rope(q, q.clone(), w, w, cache, pos, True, 1e-6, 64, 16)
    with pytest.raises(RuntimeError, match="int32"):
        sglk.fused_qk_norm_rope(torch.zeros(2, 6 * 64, dtype=torch.bfloat16, device=dev), 4, 1, 1, 64, 1e-6, w, w, 1e4, True,
                                pos.long())
