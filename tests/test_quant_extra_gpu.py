"""GPU parity for the SURVEY 8(f) rank-2 ops: sgl_per_token_quant_fp8, sgl_per_tensor_quant_fp8, awq_dequantize.
Parameter matrices of reference tests/test_per_token_quant_fp8.py:44-49, test_per_tensor_quant_fp8.py:41-45,
test_awq_dequant.py:71-84; integer / fp8 codes are compared bit for bit with the oracle."""
import itertools

import pytest
import torch
from conftest import load_golden

from oracle import quant as oq

pytestmark = pytest.mark.gpu
FP8 = torch.float8_e4m3fn


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,cols", list(itertools.product([1, 128, 512, 8192], [512, 1076, 1368, 2048, 4096, 7, 8195])))
def test_per_token_quant_fp8(sglk, dev, dtype, rows, cols):
    if rows * cols > 8192 * 4096:
        pytest.skip("size")
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.rand(rows, cols, generator=g) - (0.5 if cols % 2 else 0.0)).to(dtype)
    if rows > 1:
        x[1] = 0  # an all-zero row: scale 0, codes 0
    q = torch.empty(rows, cols, dtype=FP8, device=dev)
    s = torch.zeros(rows, dtype=torch.float32, device=dev)
    sglk.sgl_per_token_quant_fp8(x.to(dev), q, s)
    q_ref, s_ref = oq.per_token_quant_fp8(x)
    assert torch.equal(s.cpu(), s_ref)
    assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", list(itertools.product([128, 1024], [512, 2048, 8192])) + [(3, 7), (1, 1)])
def test_per_tensor_quant_fp8(sglk, dev, dtype, rows, cols):
    g = torch.Generator().manual_seed(rows * 3 + cols)
    x = (torch.rand(rows, cols, generator=g) - 0.3).to(dtype)
    q = torch.empty(rows, cols, dtype=FP8, device=dev)
    s = torch.zeros(1, dtype=torch.float32, device=dev)
    sglk.sgl_per_tensor_quant_fp8(x.to(dev), q, s, False)
    q_ref, s_ref = oq.per_tensor_quant_fp8(x)
    assert torch.equal(s.cpu(), s_ref)
    assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))
    # static scale (reference :61-67)
    scale = torch.rand(1, generator=g)
    sglk.sgl_per_tensor_quant_fp8(x.to(dev), q, scale.to(dev), True)
    q_ref, _ = oq.per_tensor_quant_fp8(x, scale)
    assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))


@pytest.mark.parametrize("is_bf16", [True, False])
@pytest.mark.parametrize("k,c", list(itertools.product([3584, 128, 512], [448, 576, 16, 128])))
def test_awq_dequantize(sglk, dev, is_bf16, k, c):
    g = torch.Generator().manual_seed(k + c)
    dt = torch.bfloat16 if is_bf16 else torch.float16
    imax = torch.iinfo(torch.int32).max
    for group in (k, 128):
        qw = torch.randint(0, imax, (k, c), generator=g, dtype=torch.int32)
        sc = torch.rand(k // group, c * 8, generator=g).to(dt)
        qz = torch.randint(0, imax, (k // group, c), generator=g, dtype=torch.int32)
        out = sglk.awq_dequantize(qw.to(dev), sc.to(dev), qz.to(dev))
        assert out.shape == (k, c * 8) and out.dtype == dt
        ref = oq.awq_dequantize(qw, sc, qz)
        assert torch.equal(out.cpu(), ref)


def test_golden_vectors(sglk, dev):
    g = load_golden("quant_extra")
    for c in g["token"]:
        q = torch.empty(c["x"].shape, dtype=FP8, device=dev)
        s = torch.zeros(c["x"].shape[0], dtype=torch.float32, device=dev)
        sglk.sgl_per_token_quant_fp8(c["x"].to(dev), q, s)
        assert torch.equal(s.cpu(), c["scale"])
        torch.testing.assert_close(q.cpu().float(), c["q"].view(FP8).float(), rtol=1e-3, atol=1e-3)
    for c in g["awq"]:
        out = sglk.awq_dequantize(c["qweight"].to(dev), c["scales"].to(dev), c["qzeros"].to(dev))
        torch.testing.assert_close(out.cpu().float(), c["out"].float(), rtol=1e-3, atol=1e-5)
