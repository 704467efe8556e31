"""GPU parity: qserve_w4a8_per_chn_gemm / qserve_w4a8_per_group_gemm vs the CPU oracle. Input construction as
reference tests/test_qserve_w4a8_per_chn_gemm.py:91-111 and tests/test_qserve_w4a8_per_group_gemm.py:148-175
(randn * 0.01, int8 symmetric activations, uint4 weights in the QServe 32x32 layout); sizes from their parameter
lists (:114-117 / :178-182), the largest ones sampled."""
import pytest
import torch
from conftest import load_golden

from oracle import qserve as oq

pytestmark = pytest.mark.gpu

MS = [1, 16, 32, 64, 65, 128, 129, 300, 512, 1024, 2049]  # (> 128 rows: the persistent pipeline of gemm_8bit.hip)
NKS = [(128, 512), (512, 1024), (1024, 4096), (4096, 512), (96, 192), (2080, 384), (7200, 128)]


def make(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(M, K, generator=g) * 0.01, torch.randn(N, K, generator=g) * 0.01


@pytest.mark.parametrize("M", MS)
@pytest.mark.parametrize("N,K", NKS)
def test_per_chn(sglk, dev, M, N, K):
    if K % 64:
        K = 192 + 64
    a, b = make(M, N, K, M + N + K)
    a_q, a_scale = oq.sym_quantize(a)
    b_q, b_scale, b_zero = oq.asym_quantize_u4(b)
    w, ws, wsz = oq.per_chn_inputs(b_q, b_scale, b_zero)
    a_sum = a.sum(dim=-1, keepdim=True).to(torch.float16)
    out = sglk.qserve_w4a8_per_chn_gemm(a_q.to(dev), w.to(dev), ws.to(dev), a_scale.to(dev), wsz.to(dev), a_sum.to(dev))
    assert out.shape == (M, N) and out.dtype == torch.float16
    ref = oq.w4a8_per_chn_gemm(a_q, b_q, a_scale, b_scale, b_zero)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-3, atol=1e-2)  # reference tolerance (:111)
    # with the exact integer row sums in place of the fp16 sums of the unquantised rows the zero-point term is exact
    a_sum_q = (a_q.float().sum(dim=-1, keepdim=True) * a_scale.float()).to(torch.float16)
    out2 = sglk.qserve_w4a8_per_chn_gemm(a_q.to(dev), w.to(dev), ws.to(dev), a_scale.to(dev), wsz.to(dev), a_sum_q.to(dev))
    torch.testing.assert_close(out2.cpu().float(), ref.float(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("M", MS)
@pytest.mark.parametrize("N,K", NKS)
def test_per_group(sglk, dev, M, N, K):
    if K % 128:
        K = 256
    a, b = make(M, N, K, M * 3 + N + K)
    a_q, a_scale = oq.sym_quantize(a)
    b_q, chn, s8, z8 = oq.progressive_group_quantize(b)
    w, ws, s8f, z8f = oq.per_group_inputs(b_q, chn, s8, z8)
    out = sglk.qserve_w4a8_per_group_gemm(a_q.to(dev), w.to(dev), z8f.to(dev), s8f.to(dev), ws.to(dev), a_scale.to(dev))
    ref = oq.w4a8_per_group_gemm(a_q, b_q, a_scale, chn, s8, z8)
    # reference tolerance (:175) is rtol 1e-3, atol 1e-5 on fp16 outputs; the integer accumulation here is exact, the
    # only difference to the oracle is fp32 rounding order before the single cast to fp16 (<= 1 ulp)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("M", [33, 48, 64])
@pytest.mark.parametrize("N,K", [(12320, 512), (14336, 384), (12288, 128)])
def test_wide_n_33_to_64_rows(sglk, dev, M, N, K):
    """33 - 64 rows at N >= 12288: qserve_w4a8_stream32_kernel (v_mfma_i32_32x32x32_i8, four 32-column blocks per wave, K split
    over eight waves - with K = 128 half of them have no step, with K = 384 half have one more than the others; N = 12320 ends in
    a quad with one real block). Integer accumulation exact: per group against the oracle at 1 fp16 ulp, per channel with the
    exact row sums."""
    a, b = make(M, N, K, M + N + K)
    a_q, a_scale = oq.sym_quantize(a)
    b_q, b_scale, b_zero = oq.asym_quantize_u4(b)
    w, ws, wsz = oq.per_chn_inputs(b_q, b_scale, b_zero)
    a_sum_q = (a_q.float().sum(dim=-1, keepdim=True) * a_scale.float()).to(torch.float16)
    out = sglk.qserve_w4a8_per_chn_gemm(a_q.to(dev), w.to(dev), ws.to(dev), a_scale.to(dev), wsz.to(dev), a_sum_q.to(dev))
    ref = oq.w4a8_per_chn_gemm(a_q, b_q, a_scale, b_scale, b_zero)
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=2e-3, atol=2e-3)
    b_q, chn, s8, z8 = oq.progressive_group_quantize(b)
    w, ws, s8f, z8f = oq.per_group_inputs(b_q, chn, s8, z8)
    out = sglk.qserve_w4a8_per_group_gemm(a_q.to(dev), w.to(dev), z8f.to(dev), s8f.to(dev), ws.to(dev), a_scale.to(dev))
    ref = oq.w4a8_per_group_gemm(a_q, b_q, a_scale, chn, s8, z8)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("M", [129, 300, 513, 800, 1024])
@pytest.mark.parametrize("N,K", [(1024, 14336), (4096, 7168), (640, 2048)])
def test_deep_k_129_to_1024_rows(sglk, dev, M, N, K):
    """few hundred rows over a deep K: the 32-row stream (one or two m-tiles per workgroup, the weights streamed per workgroup row) runs
    where its estimate beats the tile pipeline's, which scales with K too (N = 4096, K = 14336: 125 us whatever the rows before)"""
    a, b = make(M, N, K, M + N + K)
    a_q, a_scale = oq.sym_quantize(a)
    b_q, b_scale, b_zero = oq.asym_quantize_u4(b)
    w, ws, wsz = oq.per_chn_inputs(b_q, b_scale, b_zero)
    a_sum_q = (a_q.float().sum(dim=-1, keepdim=True) * a_scale.float()).to(torch.float16)
    out = sglk.qserve_w4a8_per_chn_gemm(a_q.to(dev), w.to(dev), ws.to(dev), a_scale.to(dev), wsz.to(dev), a_sum_q.to(dev))
    ref = oq.w4a8_per_chn_gemm(a_q, b_q, a_scale, b_scale, b_zero)
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=2e-3, atol=2e-3)
    b_q, chn, s8, z8 = oq.progressive_group_quantize(b)
    w, ws, s8f, z8f = oq.per_group_inputs(b_q, chn, s8, z8)
    out = sglk.qserve_w4a8_per_group_gemm(a_q.to(dev), w.to(dev), z8f.to(dev), s8f.to(dev), ws.to(dev), a_scale.to(dev))
    ref = oq.w4a8_per_group_gemm(a_q, b_q, a_scale, chn, s8, z8)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-3, atol=1e-5)


def test_golden_vectors(sglk, dev):
    g = load_golden("qserve_w4a8")
    for c in g["chn"]:
        out = sglk.qserve_w4a8_per_chn_gemm(c["a_q"].to(dev), c["packed"].to(dev), c["wscales"].to(dev),
                                             c["a_scale"].to(dev), c["w_szs"].to(dev), c["a_ssums"].to(dev))
        torch.testing.assert_close(out.cpu(), c["out"], rtol=1e-3, atol=1e-2)
    for c in g["group"]:
        out = sglk.qserve_w4a8_per_group_gemm(c["a_q"].to(dev), c["packed"].to(dev), c["zeros"].to(dev),
                                               c["scales_i8"].to(dev), c["wscales"].to(dev), c["a_scale"].to(dev))
        torch.testing.assert_close(out.cpu(), c["out"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("M,N,K", [(4096, 4096, 4096), (8192, 16384, 512)])
def test_full_size_linearity(sglk, dev, M, N, K):
    """Sizes of the reference matrix the CPU oracle cannot finish quickly: sampled rows against the oracle, and
    exact linearity in the activation scale (a power-of-two factor must scale every output exactly)."""
    a, b = make(M, N, K, 7)
    a_q, a_scale = oq.sym_quantize(a)
    b_q, chn, s8, z8 = oq.progressive_group_quantize(b)
    w, ws, s8f, z8f = oq.per_group_inputs(b_q, chn, s8, z8)
    args = (w.to(dev), z8f.to(dev), s8f.to(dev), ws.to(dev))
    out = sglk.qserve_w4a8_per_group_gemm(a_q.to(dev), *args, a_scale.to(dev))
    rows = torch.randperm(M, generator=torch.Generator().manual_seed(1))[:16]
    ref = oq.w4a8_per_group_gemm(a_q[rows], b_q, a_scale[rows], chn, s8, z8)
    torch.testing.assert_close(out.cpu()[rows], ref, rtol=1e-3, atol=1e-5)
    out2 = sglk.qserve_w4a8_per_group_gemm(a_q.to(dev), *args, (a_scale * 2).to(dev))
    normal = out.float().abs() >= 2.0 ** -13  # fp16 subnormals (and the first normal binade they round into) round differently
    assert torch.equal(out2.float()[normal], out.float()[normal] * 2)
    torch.testing.assert_close(out2.float(), out.float() * 2, rtol=0, atol=2.0 ** -23)


def test_rejects_bad_shapes(sglk, dev):
    a = torch.zeros(4, 96, dtype=torch.int8, device=dev)
    w = torch.zeros(32, 48, dtype=torch.int8, device=dev)
    h = torch.ones(32, dtype=torch.float16, device=dev)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        sglk.qserve_w4a8_per_chn_gemm(a, w, h, h[:4].clone(), h, h[:4].clone())
