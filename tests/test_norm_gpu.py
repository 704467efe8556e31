"""GPU parity: RMSNorm family (HIP, through torch.ops.sgl_kernel -> C-ABI) vs the CPU oracle and the
reference-generated golden vectors. Grid and layout cases follow reference tests/test_norm.py:65-547."""
import pytest
import torch
from conftest import load_golden

from oracle import norm as onorm

pytestmark = pytest.mark.gpu


def tol(dtype):  # reference tests/test_norm.py:45-50
    if dtype == torch.float32:
        return dict(rtol=1e-4, atol=1e-4)
    if dtype == torch.bfloat16:
        return dict(rtol=1e-2, atol=1e-2)
    return dict(rtol=1e-3, atol=1e-3)


BATCH = [1, 19, 99, 989]
HIDDEN = [111, 500, 1024, 3072, 3584, 4096, 8192, 16384]


@pytest.mark.parametrize("hidden", HIDDEN)
@pytest.mark.parametrize("batch", BATCH)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("specify_out", [True, False])
def test_rmsnorm(sglk, dev, batch, hidden, dtype, specify_out):
    g = torch.Generator().manual_seed(batch * 131 + hidden)
    x = torch.randn(batch, hidden, generator=g).to(dtype)
    w = torch.randn(hidden, generator=g).to(dtype)
    ref = onorm.rmsnorm(x, w, 1e-6)
    xd, wd = x.to(dev), w.to(dev)
    if specify_out:
        y = torch.empty_like(xd)
        sglk.rmsnorm(xd, wd, out=y)
    else:
        y = sglk.rmsnorm(xd, wd)
    torch.testing.assert_close(y.cpu(), ref, **tol(dtype))
    gy = sglk.gemma_rmsnorm(xd, wd)
    torch.testing.assert_close(gy.cpu(), onorm.gemma_rmsnorm(x, w, 1e-6), **tol(dtype))
    assert torch.equal(xd.cpu(), x), "input must not be modified"


@pytest.mark.parametrize("hidden", HIDDEN)
@pytest.mark.parametrize("batch", BATCH)
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.bfloat16])
def test_fused_add_rmsnorm(sglk, dev, batch, hidden, dtype):
    g = torch.Generator().manual_seed(batch * 17 + hidden)
    x = torch.randn(batch, hidden, generator=g).to(dtype)
    r = torch.randn(batch, hidden, generator=g).to(dtype)
    w = torch.randn(hidden, generator=g).to(dtype)
    for fn, ofn in [(sglk.fused_add_rmsnorm, onorm.fused_add_rmsnorm),
                    (sglk.gemma_fused_add_rmsnorm, onorm.gemma_fused_add_rmsnorm)]:
        ref_y, ref_r = ofn(x, r, w, 1e-6)
        xd, rd = x.to(dev), r.to(dev)
        fn(xd, rd, w.to(dev), 1e-6)
        torch.testing.assert_close(xd.cpu(), ref_y, **tol(dtype))
        # the residual update is a single rounded add: bit-exact
        assert torch.equal(rd.cpu(), ref_r)


def test_configs0_full_size(sglk, dev):
    """BASELINE configs[0] at its own size: rmsnorm / fused_add_rmsnorm (and the gemma forms) on (4096, 4096) bf16, every
    row against the oracle (the reference grid above stops at 989 rows; bench.py times exactly this shape)."""
    g = torch.Generator().manual_seed(4096)
    x = torch.randn(4096, 4096, generator=g).to(torch.bfloat16)
    r = torch.randn(4096, 4096, generator=g).to(torch.bfloat16)
    w = torch.randn(4096, generator=g).to(torch.bfloat16)
    xd, wd = x.to(dev), w.to(dev)
    torch.testing.assert_close(sglk.rmsnorm(xd, wd).cpu(), onorm.rmsnorm(x, w, 1e-6), **tol(torch.bfloat16))
    torch.testing.assert_close(sglk.gemma_rmsnorm(xd, wd).cpu(), onorm.gemma_rmsnorm(x, w, 1e-6), **tol(torch.bfloat16))
    assert torch.equal(xd.cpu(), x), "input must not be modified"
    for fn, ofn in [(sglk.fused_add_rmsnorm, onorm.fused_add_rmsnorm),
                    (sglk.gemma_fused_add_rmsnorm, onorm.gemma_fused_add_rmsnorm)]:
        ref_y, ref_r = ofn(x, r, w, 1e-6)
        yd, rd = x.to(dev), r.to(dev)
        fn(yd, rd, wd, 1e-6)
        torch.testing.assert_close(yd.cpu(), ref_y, **tol(torch.bfloat16))
        assert torch.equal(rd.cpu(), ref_r)  # the residual update is a single rounded add: bit-exact


def test_very_wide_rows(sglk, dev):
    # reference tests/test_norm.py:105-117 has (2, 32768); go past the register-cached range too
    for hidden, dtype in [(32768, torch.float16), (65536 + 1024, torch.bfloat16), (40000, torch.float32)]:
        x = torch.randn(2, hidden).to(dtype)
        w = torch.randn(hidden).to(dtype)
        torch.testing.assert_close(sglk.gemma_rmsnorm(x.to(dev), w.to(dev)).cpu(), onorm.gemma_rmsnorm(x, w),
                                   **tol(dtype))
        r = torch.randn(2, hidden).to(dtype)
        xd, rd = x.to(dev), r.to(dev)
        sglk.fused_add_rmsnorm(xd, rd, w.to(dev))
        ry, rr = onorm.fused_add_rmsnorm(x, r, w)
        torch.testing.assert_close(xd.cpu(), ry, **tol(dtype))
        assert torch.equal(rd.cpu(), rr)


def test_layouts(sglk, dev):
    """row-strided 2-D slice, 3-D, 3-D sliced, non-flattenable 3-D (QKV head slice, incl. rows that are
    not 16-byte aligned), fp32 weight with 16-bit input — reference tests/test_norm.py:172-547."""
    dtype = torch.float16
    hidden = 128
    # row-strided 2-D slice (stride(0) = hidden + 64)
    buf = torch.randn(19, hidden + 64).to(dtype)
    w = torch.randn(hidden).to(dtype)
    bd = buf.to(dev)
    y = sglk.rmsnorm(bd[:, :hidden], w.to(dev))
    torch.testing.assert_close(y.cpu(), onorm.rmsnorm(buf[:, :hidden], w), **tol(dtype))
    # 3-D contiguous and 3-D sliced on the last dim
    x3 = torch.randn(4, 7, hidden + 32).to(dtype)
    y = sglk.rmsnorm(x3.to(dev)[..., :hidden], w.to(dev))
    torch.testing.assert_close(y.cpu(), onorm.rmsnorm(x3[..., :hidden], w), **tol(dtype))
    # non-flattenable: q heads out of a packed qkv buffer [tokens, (hq + 2 hk) * d] viewed as [tokens, heads, d]
    tokens, hq, hk, d = 11, 8, 2, 72  # d = 72 halves: rows 144 B apart -> not all 16-byte aligned
    qkv = torch.randn(tokens, (hq + 2 * hk) * d).to(dtype)
    wq = torch.randn(d).to(dtype)
    qd = qkv.to(dev)
    k_view = qd[:, hq * d:(hq + hk) * d].view(tokens, hk, d)
    assert not k_view.is_contiguous()
    out = torch.empty(tokens, hk, d, dtype=dtype, device=dev)
    sglk.rmsnorm(k_view, wq.to(dev), out=out)
    ref = onorm.rmsnorm(qkv[:, hq * d:(hq + hk) * d].reshape(tokens, hk, d), wq)
    torch.testing.assert_close(out.cpu(), ref, **tol(dtype))
    # in-place on the strided view (output strides differ from a fresh tensor's)
    sglk.gemma_rmsnorm(k_view, wq.to(dev), out=k_view)
    torch.testing.assert_close(k_view.cpu(), onorm.gemma_rmsnorm(
        qkv[:, hq * d:(hq + hk) * d].reshape(tokens, hk, d), wq), **tol(dtype))
    # fp32 weight with 16-bit input
    for dt in (torch.float16, torch.bfloat16):
        x = torch.randn(33, 1024).to(dt)
        wf = torch.randn(1024)
        torch.testing.assert_close(sglk.rmsnorm(x.to(dev), wf.to(dev)).cpu(), onorm.rmsnorm(x, wf), **tol(dt))


def test_golden_vectors(sglk, dev):
    for c in load_golden("norm"):
        x, r, w, eps = c["x"], c["residual"], c["w"], c["eps"]
        t = tol(x.dtype)
        torch.testing.assert_close(sglk.rmsnorm(x.to(dev), w.to(dev), eps).cpu(), c["rmsnorm"], **t)
        torch.testing.assert_close(sglk.gemma_rmsnorm(x.to(dev), w.to(dev), eps).cpu(), c["gemma_rmsnorm"], **t)
        xd, rd = x.to(dev), r.to(dev)
        sglk.fused_add_rmsnorm(xd, rd, w.to(dev), eps)
        torch.testing.assert_close(xd.cpu(), c["fused_add"][0], **t)
        torch.testing.assert_close(rd.cpu(), c["fused_add"][1], **t)


def test_errors(sglk, dev):
    x = torch.randn(4, 64, device=dev, dtype=torch.float16)
    with pytest.raises(RuntimeError):
        sglk.rmsnorm(x, torch.randn(32, device=dev, dtype=torch.float16))  # weight size mismatch
    with pytest.raises(RuntimeError):
        sglk.rmsnorm(x.t(), torch.randn(4, device=dev, dtype=torch.float16))  # last dim not contiguous
    with pytest.raises(RuntimeError):
        sglk.fused_add_rmsnorm(x[:, :32], x[:, 32:], torch.randn(32, device=dev, dtype=torch.float16))
