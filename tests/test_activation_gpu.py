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
