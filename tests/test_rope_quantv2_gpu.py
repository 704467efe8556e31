"""GPU parity: rotary_embedding and sgl_per_token_group_quant_8bit_v2 vs the CPU oracle.
Rope cases follow reference tests/test_rotary_embedding.py (2-D in-place neox / interleaved, partial rotary dim,
3-D out-of-place); quant v2 cases follow tests/test_per_token_group_quant_8bit_v2.py:22-70 (fused silu-and-mul,
masked layouts, UE8M0 packed scales)."""
import pytest
import torch
from conftest import load_golden

from oracle import quant as oquant
from oracle import rope as orope

pytestmark = pytest.mark.gpu
FP8 = torch.float8_e4m3fn


def make_cache(rot, max_pos, dtype, g):
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, dtype=torch.float) / rot))
    f = torch.einsum("i,j->ij", torch.arange(max_pos, dtype=torch.float), inv)
    return torch.cat((f.cos(), f.sin()), dim=-1).to(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("neox", [True, False])
@pytest.mark.parametrize("tokens,hq,hk,head,rot", [(1, 1, 1, 64, 64), (33, 32, 8, 128, 128), (17, 8, 2, 128, 64),
                                                   (5, 4, 4, 96, 32), (129, 16, 16, 64, 16), (7, 2, 1, 80, 20)])
def test_rope_2d_inplace(sglk, dev, dtype, neox, tokens, hq, hk, head, rot):
    g = torch.Generator().manual_seed(tokens + head)
    cache = make_cache(rot, 512, dtype, g)
    pos = torch.randint(0, 512, (tokens,), generator=g)
    q = torch.randn(tokens, hq * head, generator=g).to(dtype)
    k = torch.randn(tokens, hk * head, generator=g).to(dtype)
    rq, rk = orope.rotary_embedding(pos, q, k, head, cache, neox)
    qd, kd = q.to(dev), k.to(dev)
    oq, ok = sglk.rotary_embedding(pos.to(dev), qd, kd, head, cache.to(dev), neox)
    assert oq.data_ptr() == qd.data_ptr() and ok.data_ptr() == kd.data_ptr()
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=2e-3, atol=2e-3)
    if dtype == torch.float32:
        tol = dict(rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(qd.cpu(), rq, **tol)
    torch.testing.assert_close(kd.cpu(), rk, **tol)
    # one rounding of an fp32 result: at most 1 ulp of the storage type away from the oracle
    if dtype != torch.float32:
        assert (qd.cpu().view(torch.int16).int() - rq.view(torch.int16).int()).abs().max() <= 1


def test_rope_strided_rows_and_3d(sglk, dev):
    dtype, head, tokens = torch.bfloat16, 64, 19
    g = torch.Generator().manual_seed(4)
    cache = make_cache(head, 128, dtype, g)
    pos = torch.randint(0, 128, (tokens,), generator=g)
    # q/k as column slices of one packed qkv buffer (row stride != width)
    qkv = torch.randn(tokens, (8 + 2 + 2) * head, generator=g).to(dtype)
    qkv_d = qkv.to(dev)
    q_v, k_v = qkv_d[:, : 8 * head], qkv_d[:, 8 * head: 10 * head]
    rq, rk = orope.rotary_embedding(pos, qkv[:, : 8 * head], qkv[:, 8 * head: 10 * head], head, cache, True)
    sglk.rotary_embedding(pos.to(dev), q_v, k_v, head, cache.to(dev), True)
    torch.testing.assert_close(q_v.cpu(), rq, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(k_v.cpu(), rk, rtol=2e-2, atol=2e-2)
    assert torch.equal(qkv_d[:, 10 * head:].cpu(), qkv[:, 10 * head:]), "v must be untouched"
    # 3-D: out of place
    q3 = torch.randn(tokens, 8, head, generator=g).to(dtype)
    k3 = torch.randn(tokens, 2, head, generator=g).to(dtype)
    q3d, k3d = q3.to(dev), k3.to(dev)
    oq, ok = sglk.rotary_embedding(pos.to(dev), q3d, k3d, head, cache.to(dev), False)
    rq, rk = orope.rotary_embedding(pos, q3, k3, head, cache, False)
    assert oq.data_ptr() != q3d.data_ptr() and torch.equal(q3d.cpu(), q3)
    torch.testing.assert_close(oq.cpu(), rq, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(ok.cpu(), rk, rtol=2e-2, atol=2e-2)


def test_rope_golden(sglk, dev):
    for c in load_golden("rope"):
        q, k = c["q"].to(dev), c["k"].to(dev)
        sglk.rotary_embedding(c["positions"].to(dev), q, k, c["head_size"], c["cache"].to(dev), c["is_neox"])
        t = dict(rtol=2e-2, atol=2e-2) if c["q"].dtype == torch.bfloat16 else dict(rtol=2e-3, atol=2e-3)
        torch.testing.assert_close(q.cpu(), c["q_out"], **t)
        torch.testing.assert_close(k.cpu(), c["k_out"], **t)


# ------------------------------------------------------------------------------------------------ quant v2
def run_v2(sglk, dev, x, gs, dst, ue8m0=False, colmajor=False, fuse=False, masked_m=None):
    hidden = x.shape[-1] // (2 if fuse else 1)
    groups = hidden // gs
    lead = x.shape[:-1]
    q = torch.zeros(*lead, hidden, dtype=dst, device=dev)
    if ue8m0 and colmajor:
        s = torch.zeros(*lead[:-1], (groups + 3) // 4, lead[-1], dtype=torch.int32, device=dev).transpose(-1, -2)
    elif ue8m0:
        s = torch.zeros(*lead, groups, dtype=torch.uint8, device=dev)
    elif colmajor:
        s = torch.zeros(*lead[:-1], groups, lead[-1], dtype=torch.float32, device=dev).transpose(-1, -2)
    else:
        s = torch.zeros(*lead, groups, dtype=torch.float32, device=dev)
    lim = (-448.0, 448.0) if dst == FP8 else (-128.0, 127.0)
    sglk.sgl_per_token_group_quant_8bit(x.to(dev), q, s, gs, 1e-10, lim[0], lim[1], ue8m0, fuse,
                                        masked_m.to(dev) if masked_m is not None else None, enable_v2=True)
    return q.cpu(), s.cpu()


@pytest.mark.parametrize("gs", [16, 32, 64, 128])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("colmajor", [False, True])
def test_quant_v2_float_scales(sglk, dev, gs, dtype, fuse, colmajor):
    g = torch.Generator().manual_seed(gs + fuse)
    x = torch.randn(77, 1024 * (2 if fuse else 1), generator=g).to(dtype)
    q, s = run_v2(sglk, dev, x, gs, FP8, colmajor=colmajor, fuse=fuse)
    oq, os_, _, _ = oquant.per_token_group_quant_8bit_v2(x, gs, FP8, fuse_silu_and_mul=fuse)
    if not fuse:
        assert torch.equal(s, os_) and torch.equal(q.view(torch.uint8), oq.view(torch.uint8))
    else:
        # tanh differs in the last ulp between host and device libms: scales agree to rounding, codes within 1 step
        torch.testing.assert_close(s, os_, rtol=1e-2, atol=1e-7)
        d = (q.view(torch.uint8).int() - oq.view(torch.uint8).int()).abs()
        assert d.max() <= 1 and (d != 0).float().mean() < 0.02


def test_quant_v2_int8_and_ue8m0(sglk, dev):
    x = (torch.randn(40, 2048, generator=torch.Generator().manual_seed(9)) * 3).to(torch.bfloat16)
    q, s = run_v2(sglk, dev, x, 128, torch.int8)
    oq, os_, _, _ = oquant.per_token_group_quant_8bit_v2(x, 128, torch.int8)
    # v2 clamps int8 to [-128, 127] (the dtype limits it insists on); values never reach -128 after x / (amax/127)
    assert torch.equal(s, os_) and torch.equal(q, oq)
    for colmajor in (False, True):
        xx = x[:, :352 * 2].contiguous()[:, :352]  # 11 groups of 32: partly filled last pack
        q, s = run_v2(sglk, dev, xx.contiguous(), 32, FP8, ue8m0=True, colmajor=colmajor)
        oq, os_, ue, _ = oquant.per_token_group_quant_8bit_v2(xx.contiguous(), 32, FP8, scale_ue8m0=True)
        if colmajor:
            b = s.contiguous().view(torch.uint8).view(40, -1, 4).reshape(40, -1)
            assert torch.equal(b[:, :11], ue) and (b[:, 11:] == 0).all()
        else:
            assert torch.equal(s, ue)
        assert torch.equal(q.view(torch.uint8), oq.view(torch.uint8))


@pytest.mark.parametrize("mode", ["balanced", "imbalanced", "extreme"])
def test_quant_v2_masked_layout(sglk, dev, mode):
    E, T, H, gs = 6, 64, 512, 128
    g = torch.Generator().manual_seed(5)
    x = torch.randn(E, T, 2 * H, generator=g).to(torch.bfloat16)
    masked = {"balanced": torch.full((E,), 40), "imbalanced": torch.randint(0, T + 1, (E,), generator=g),
              "extreme": torch.tensor([T, 0, 0, 1, 0, T])}[mode].to(torch.int32)
    q, s = run_v2(sglk, dev, x, gs, FP8, fuse=True, masked_m=masked)
    oq, os_, _, valid = oquant.per_token_group_quant_8bit_v2(x, gs, FP8, fuse_silu_and_mul=True, masked_m=masked)
    torch.testing.assert_close(s[valid], os_[valid], rtol=1e-2, atol=1e-7)
    d = (q.view(torch.uint8).int() - oq.view(torch.uint8).int()).abs()[valid]
    assert d.max() <= 1
    assert (q.view(torch.uint8)[~valid] == 0).all() and (s[~valid] == 0).all(), "rows past masked_m must not be written"
