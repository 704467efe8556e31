"""GPU parity: silu/gelu_tanh/gelu _and_mul vs the CPU oracle. Grid follows reference
tests/test_activation.py:13-40 (dim x batch x seq, fp16, 3-D input); bf16/fp32 added."""
import pytest
import torch
from conftest import load_golden

from oracle import activation as oact

pytestmark = pytest.mark.gpu

OPS = [("silu_and_mul", oact.silu_and_mul), ("gelu_tanh_and_mul", oact.gelu_tanh_and_mul),
       ("gelu_and_mul", oact.gelu_and_mul)]


@pytest.mark.parametrize("dim", [128, 256, 512, 2048, 4096, 11008, 16384])
@pytest.mark.parametrize("batch,seq", [(1, 1), (2, 4), (4, 32), (8, 2), (16, 64), (1, 512), (3, 128)])
def test_act_and_mul_fp16(sglk, dev, dim, batch, seq):
    x = torch.randn(batch, seq, 2 * dim, generator=torch.Generator().manual_seed(dim + batch)).to(torch.float16)
    xd = x.to(dev)
    for name, ofn in OPS:
        y = getattr(sglk, name)(xd)
        assert y.shape == (batch, seq, dim)
        torch.testing.assert_close(y.cpu(), ofn(x), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_act_and_mul_other_dtypes(sglk, dev, dtype):
    x = (torch.randn(37, 2 * 1000) * 3).to(dtype)
    t = dict(rtol=1e-2, atol=1e-2) if dtype == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
    for name, ofn in OPS:
        out = torch.empty(37, 1000, dtype=dtype, device=dev)
        r = getattr(sglk, name)(x.to(dev), out=out)
        assert r.data_ptr() == out.data_ptr()
        torch.testing.assert_close(out.cpu(), ofn(x), **t)


def test_full_size_config(sglk, dev):
    # BASELINE.json configs[0]: silu_and_mul on x[4096, 8192] bf16 -> out[4096, 4096]
    x = torch.randn(4096, 8192, generator=torch.Generator().manual_seed(7)).to(torch.bfloat16)
    y = sglk.silu_and_mul(x.to(dev)).cpu()
    idx = torch.randint(0, 4096, (64,), generator=torch.Generator().manual_seed(8))
    torch.testing.assert_close(y[idx], oact.silu_and_mul(x[idx]), rtol=1e-2, atol=1e-2)
    # exact symmetry property of the op: act(a)*b is linear in b -> negating b negates the output bit for bit
    xn = x.clone()
    xn[:, 4096:] = -xn[:, 4096:]
    assert torch.equal(sglk.silu_and_mul(xn.to(dev)).cpu(), -y)


def test_golden_vectors(sglk, dev):
    # (input, expected, tolerance) captured from the reference's own test functions: tests/golden/make_golden.py
    for c in load_golden("activation"):
        for key, fn in (("silu", sglk.silu_and_mul), ("gelu_tanh", sglk.gelu_tanh_and_mul), ("gelu", sglk.gelu_and_mul)):
            e = c[key]
            torch.testing.assert_close(fn(e["x"].to(dev)).cpu(), e["out"], rtol=e["rtol"], atol=e["atol"])


def test_errors(sglk, dev):
    with pytest.raises(ValueError):
        sglk.silu_and_mul(torch.randn(2, 6, dtype=torch.float16, device=dev))
    with pytest.raises(AssertionError):
        sglk.silu_and_mul(torch.randn(2, 64, dtype=torch.float16, device=dev),
                          out=torch.empty(2, 16, dtype=torch.float16, device=dev))


# ------------------------------------------------ swiglu_gpt_oss_sigmoid_alpha (reference tests/test_swiglu_with_alpha_limit.py)
@pytest.mark.parametrize("rows", [1, 16, 128, 512, 1024])
@pytest.mark.parametrize("width", [64, 128, 256, 512, 1024, 2048, 4096])  # (the reference's "hidden_size" is the INPUT width)
@pytest.mark.parametrize("alpha,limit", [(0.5, 1.0), (0.5, 10.0), (1.0, 5.0), (2.0, 1.0), (2.0, 10.0), (1.702, 7.0)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_swiglu_gpt_oss_sigmoid_alpha(sglk, dev, rows, width, alpha, limit, dtype):
    x = (torch.randn(rows, width, generator=torch.Generator().manual_seed(rows + width)) * 3).to(dtype)
    y = sglk.swiglu_gpt_oss_sigmoid_alpha(x.to(dev), alpha, limit)
    assert y.shape == (rows, width // 2) and y.dtype == dtype
    ref = oact.swiglu_gpt_oss_sigmoid_alpha(x, alpha, limit)
    # fp32 arithmetic and one rounding on both sides: an output ulp (the reference test allows 1e-1 / 1e-4, :42-43)
    tol = dict(rtol=2e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-3)
    torch.testing.assert_close(y.cpu(), ref, **tol)


def test_swiglu_gpt_oss_odd_counts_and_views(sglk, dev):
    for rows, width in [(3, 2), (5, 6), (7, 66), (1, 10)]:  # pair counts that are not multiples of four: the scalar form
        x = torch.randn(rows, width, generator=torch.Generator().manual_seed(width)).to(torch.bfloat16)
        torch.testing.assert_close(sglk.swiglu_gpt_oss_sigmoid_alpha(x.to(dev), 1.702, 7.0).cpu(),
                                   oact.swiglu_gpt_oss_sigmoid_alpha(x, 1.702, 7.0), rtol=1e-2, atol=1e-3)
    assert sglk.swiglu_gpt_oss_sigmoid_alpha(torch.empty(0, 8, device=dev), 1.0, 1.0).shape == (0, 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        sglk.swiglu_gpt_oss_sigmoid_alpha(torch.zeros(4, 16, device=dev)[:, ::2], 1.0, 1.0)
    with pytest.raises(RuntimeError, match="2D"):
        torch.ops.sgl_kernel.swiglu_gpt_oss_sigmoid_alpha(torch.zeros(2, 4, 8, device=dev), 1.0, 1.0)
    with pytest.raises(AssertionError, match="positive"):
        sglk.swiglu_gpt_oss_sigmoid_alpha(torch.zeros(4, 16, device=dev), 1.0, 0.0)


# ------------------------------------------------------------- silu_and_mul_clamp (reference tests/test_silu_and_mul_clamp.py)
@pytest.mark.parametrize("M", [1, 16, 128, 1000])
@pytest.mark.parametrize("H", [32, 64, 40, 2048, 7168])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("limit", [10.0, 0.7])
def test_silu_and_mul_clamp(sglk, dev, M, H, dtype, limit):
    x = (torch.randn(M, 2 * H, generator=torch.Generator().manual_seed(M + H)) * 6).to(dtype)
    out = torch.full((M, H), float("nan"), dtype=dtype, device=dev)
    assert sglk.silu_and_mul_clamp(x.to(dev), out, limit) is None
    torch.testing.assert_close(out.cpu(), oact.silu_and_mul_clamp(x, limit), rtol=1e-2, atol=1e-2)  # reference :112
    # the clamp is exact in bf16, the rest is fp32 with one rounding: an output ulp
    torch.testing.assert_close(out.cpu(), oact.silu_and_mul_clamp(x, limit), rtol=8e-3 if dtype == torch.bfloat16 else 1e-3,
                               atol=1e-3)


def test_swiglu_variants_golden_and_errors(sglk, dev):
    g = load_golden("swiglu")  # outputs of the reference tests' own torch functions: tests/golden/make_golden.py gen_swiglu
    for c in g["alpha"]:
        tol = 1e-4 if c["x"].dtype == torch.float32 else 1e-1
        torch.testing.assert_close(sglk.swiglu_gpt_oss_sigmoid_alpha(c["x"].to(dev), c["alpha"], c["limit"]).cpu(), c["out"],
                                   rtol=tol, atol=tol)
    for c in g["clamp"]:
        out = torch.empty_like(c["out"], device=dev)
        sglk.silu_and_mul_clamp(c["x"].to(dev), out, c["limit"])
        torch.testing.assert_close(out.cpu(), c["out"], rtol=1e-2, atol=1e-2)
    with pytest.raises(ValueError, match="16 bytes"):
        sglk.silu_and_mul_clamp(torch.zeros(2, 12, dtype=torch.bfloat16, device=dev), torch.zeros(2, 6, dtype=torch.bfloat16, device=dev), 10.0)
    with pytest.raises(RuntimeError, match="swiglu_limit must be > 0"):
        sglk.silu_and_mul_clamp(torch.zeros(2, 16, dtype=torch.bfloat16, device=dev), torch.zeros(2, 8, dtype=torch.bfloat16, device=dev), 0.0)
    with pytest.raises(RuntimeError, match="Half or BFloat16"):
        sglk.silu_and_mul_clamp(torch.zeros(2, 16, device=dev), torch.zeros(2, 8, device=dev), 10.0)
