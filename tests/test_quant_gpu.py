"""GPU parity: sgl_per_token_group_quant_8bit vs the CPU oracle, BIT-EXACT on codes, scales and ue8m0
bytes (the oracle restates the kernel's IEEE fp32 steps). Cases follow reference
tests/test_per_token_group_quant_8bit.py:278-417."""
import pytest
import torch
from conftest import load_golden

from oracle import quant as oquant

pytestmark = pytest.mark.gpu

FP8 = torch.float8_e4m3fn
FP8_MAX = 448.0


def run_op(sglk, dev, x, gs, dst, column_major=False, ue8m0=False, eps=1e-10):
    rows, k = x.shape
    groups = k // gs
    xd = x.to(dev)
    q = torch.empty(rows, k, dtype=dst, device=dev)
    qmax = FP8_MAX if dst == FP8 else 127.0
    if ue8m0 and column_major:
        packed = (groups + 3) // 4
        s = torch.zeros(packed, rows, dtype=torch.int32, device=dev).t()  # [rows, packed], stride (1, rows)
    elif ue8m0:
        s = torch.empty(rows, groups, dtype=torch.uint8, device=dev)
    elif column_major:
        s = torch.empty(groups, rows, dtype=torch.float32, device=dev).t()
    else:
        s = torch.empty(rows, groups, dtype=torch.float32, device=dev)
    sglk.sgl_per_token_group_quant_8bit(xd, q, s, gs, eps, -qmax, qmax, ue8m0, enable_v2=False)
    return q.cpu(), s.cpu()


CASES = [  # (tokens, hidden, group, dst, src, column_major)
    (5120, 512, 128, FP8, torch.bfloat16, False),
    (5120, 2048, 128, FP8, torch.bfloat16, True),
    (1024, 18432, 128, FP8, torch.bfloat16, False),
    (5, 7168, 128, FP8, torch.bfloat16, False),
    (40, 2048, 128, FP8, torch.float16, True),
    (333, 1024, 64, FP8, torch.float32, False),
    (129, 1536, 128, torch.int8, torch.bfloat16, False),
    (40, 2048, 64, torch.int8, torch.float16, True),
    (77, 512, 32, FP8, torch.bfloat16, False),
    (31, 1024, 256, FP8, torch.float16, False),
    (13, 2048, 512, torch.int8, torch.float32, True),
    (1, 128, 128, FP8, torch.bfloat16, False),
]


@pytest.mark.parametrize("tokens,hidden,gs,dst,src,colmajor", CASES)
def test_quant_bit_exact(sglk, dev, tokens, hidden, gs, dst, src, colmajor):
    g = torch.Generator().manual_seed(tokens + hidden)
    x = torch.randn(tokens, hidden, generator=g).to(src)
    q, s = run_op(sglk, dev, x, gs, dst, column_major=colmajor)
    oq, os_, _ = oquant.per_token_group_quant_8bit(x, gs, dst)
    assert torch.equal(s, os_), "scales must be bit-exact"
    if dst == FP8:
        assert torch.equal(q.view(torch.uint8), oq.view(torch.uint8)), "fp8 codes must be bit-exact"
    else:
        assert torch.equal(q, oq), "int8 codes must be bit-exact"


@pytest.mark.parametrize("scale", [1e-3, 100.0, 1e-12, 3e4])
def test_scale_edge_cases(sglk, dev, scale):
    x = (torch.randn(64, 1024, generator=torch.Generator().manual_seed(3)) * scale).to(torch.bfloat16)
    x[0] = 0  # an all-zero group exercises the eps floor
    for dst in (FP8, torch.int8):
        q, s = run_op(sglk, dev, x, 128, dst)
        oq, os_, _ = oquant.per_token_group_quant_8bit(x, 128, dst)
        assert torch.equal(s, os_)
        assert torch.equal(q.view(torch.uint8), oq.view(torch.uint8))


@pytest.mark.parametrize("gs", [32, 64, 128])
@pytest.mark.parametrize("colmajor", [False, True])
def test_ue8m0(sglk, dev, gs, colmajor):
    rows, k = 37, 1024 if gs != 32 else 352  # 352/32 = 11 groups: not a multiple of 4 (packed tail)
    x = (torch.randn(rows, k, generator=torch.Generator().manual_seed(gs)) * 7).to(torch.bfloat16)
    q, s = run_op(sglk, dev, x, gs, FP8, column_major=colmajor, ue8m0=True)
    oq, os_, ue = oquant.per_token_group_quant_8bit(x, gs, FP8, scale_ue8m0=True)
    groups = k // gs
    if colmajor:
        # s is [rows, packed] int32 with strides (1, rows): byte g%4 of element (row, g//4)
        b = s.contiguous().view(torch.uint8).view(rows, -1, 4).reshape(rows, -1)[:, :groups]
    else:
        b = s
    assert torch.equal(b, ue), "ue8m0 scale bytes must be exact (reference test :405)"
    assert torch.equal(q.view(torch.uint8), oq.view(torch.uint8))


def test_full_size_properties(sglk, dev):
    """BASELINE configs[1] quant shape (M=4096, K=4096): sampled rows against the oracle + exact properties:
    every |code| <= 448, each group reaches the max code, dequantised error within half an fp8 ulp."""
    x = torch.randn(4096, 4096, generator=torch.Generator().manual_seed(11)).to(torch.bfloat16)
    q, s = run_op(sglk, dev, x, 128, FP8, column_major=True)
    idx = torch.randint(0, 4096, (128,), generator=torch.Generator().manual_seed(12))
    oq, os_, _ = oquant.per_token_group_quant_8bit(x[idx], 128, FP8)
    assert torch.equal(q[idx].view(torch.uint8), oq.view(torch.uint8)) and torch.equal(s[idx], os_)
    qf = q.float().view(4096, 32, 128)
    assert qf.abs().max() <= 448.0
    assert (qf.abs().amax(dim=-1) == 448.0).all(), "the group maximum must map to +-448"
    deq = qf * s.unsqueeze(-1)
    err = (deq - x.float().view(4096, 32, 128)).abs()
    assert (err <= s.unsqueeze(-1) * 16.0 + 1e-6).all()  # e4m3 spacing near 448 is 32


def test_golden_vectors(sglk, dev):
    for c in load_golden("quant"):
        x, gs = c["x"], c["group_size"]
        rows = x.shape[0]
        q, s = run_op(sglk, dev, x, gs, FP8)
        torch.testing.assert_close(s, c["fp8_s"], rtol=1e-3, atol=1e-5)
        deq = q.float().view(rows, -1, gs) * s.unsqueeze(-1)
        ref = c["fp8_q"].view(FP8).float().view(rows, -1, gs) * c["fp8_s"].unsqueeze(-1)
        torch.testing.assert_close(deq, ref, rtol=1e-1, atol=1e-1)
        q, s = run_op(sglk, dev, x, gs, torch.int8)
        torch.testing.assert_close(s, c["int8_s"], rtol=1e-3, atol=1e-5)
        assert (q.int() - c["int8_q"].int()).abs().max() <= 1


def test_errors(sglk, dev):
    x = torch.randn(4, 256, device=dev, dtype=torch.bfloat16)
    q = torch.empty(4, 256, device=dev, dtype=FP8)
    s = torch.empty(4, 2, device=dev, dtype=torch.float32)
    with pytest.raises(RuntimeError, match="group_size"):
        sglk.sgl_per_token_group_quant_8bit(x, q, torch.empty(4, 5, device=dev), 48, 1e-10, -448.0, 448.0,
                                            enable_v2=False)
    with pytest.raises(RuntimeError, match="Int8 or Float8_e4m3fn"):
        sglk.sgl_per_token_group_quant_8bit(x, torch.empty(4, 256, device=dev, dtype=torch.float16), s, 128,
                                            1e-10, -448.0, 448.0, enable_v2=False)
    with pytest.raises(AssertionError):
        sglk.sgl_per_token_group_quant_8bit(x, q, s, 128, 1e-10, -448.0, 448.0, fuse_silu_and_mul=True,
                                            enable_v2=False)
