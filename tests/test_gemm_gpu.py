"""GPU parity: fp8_blockwise_scaled_mm, fp8_scaled_mm, int8_scaled_mm vs the CPU oracle.

Input construction and tolerances follow reference tests/test_fp8_blockwise_gemm.py:66-91,
tests/test_fp8_gemm.py:22-48 and tests/test_int8_gemm.py:25-47. Sizes the CPU oracle finishes in
seconds are compared in full; the full BASELINE sizes are compared on a random sample of rows and
columns (the GEMM is separable: out[rows][:, cols] depends only on a[rows] and b[:, cols]) plus an exact
integer known-answer case that pins the MFMA operand/accumulator lane maps."""
import ctypes
import os

import pytest
import torch
from conftest import PKG, load_golden

from oracle import gemm as ogemm

pytestmark = pytest.mark.gpu

FP8 = torch.float8_e4m3fn
FMAX = 448.0


def make_blockwise(M, N, K, seed, scale_mag=1e-3):
    g = torch.Generator().manual_seed(seed)
    a = ((torch.rand(M, K, generator=g) - 0.5) * 2 * FMAX).clamp(-FMAX, FMAX).to(FP8)
    b = ((torch.rand(N, K, generator=g) - 0.5) * 2 * FMAX).clamp(-FMAX, FMAX).to(FP8).t()
    sa = (torch.randn(M, K // 128, generator=g) * scale_mag).t().contiguous().t()
    sb = (torch.randn(K // 128, (N + 127) // 128, generator=g) * scale_mag).t().contiguous().t()
    return a, b, sa, sb


def to_dev_colmajor(t, dev):
    # keep the reference's layouts: b [K,N] with stride (1,K); scales column-major
    return t.t().contiguous().to(dev).t()


def run_blockwise(sglk, dev, a, b, sa, sb, dtype):
    return sglk.fp8_blockwise_scaled_mm(a.to(dev), to_dev_colmajor(b, dev), to_dev_colmajor(sa, dev),
                                        to_dev_colmajor(sb, dev), dtype).cpu()


def test_mfma_lane_maps_known_answer(sglk, dev):
    """Exact small-integer data, asymmetric in m, n and k: any transposed or permuted operand/accumulator
    map, or a K sub-block pairing error, changes the (exactly representable) result."""
    M, N, K = 256, 256, 256
    m = torch.arange(M).view(M, 1)
    n = torch.arange(N).view(N, 1)
    k = torch.arange(K).view(1, K)
    a = (((m * 7 + k * 3) % 5) - 2).float()          # values in {-2..2}
    b = (((n * 11 + k * 5 + (k // 16)) % 7) - 3).float()  # values in {-3..3}
    sa = torch.ones(M, K // 128)
    sb = torch.ones(K // 128, N // 128)
    sa[:, 1] = 2.0  # makes the two K blocks distinguishable
    out = run_blockwise(sglk, dev, a.to(FP8), b.to(FP8).t(), sa, sb, torch.float16).float()
    ref = a[:, :128] @ b[:, :128].t() + 2.0 * (a[:, 128:] @ b[:, 128:].t())
    assert ref.abs().max() < 2048  # exactly representable in fp16
    assert torch.equal(out, ref)


@pytest.mark.parametrize("M", [1, 3, 5, 17, 40, 100, 127, 128, 512])
@pytest.mark.parametrize("N,K", [(128, 512), (512, 1024), (1024, 4096), (4096, 512), (14080, 1024), (8192, 8192), (640, 1152)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fp8_blockwise(sglk, dev, M, N, K, dtype):
    a, b, sa, sb = make_blockwise(M, N, K, seed=M * 7 + N + K)
    out = run_blockwise(sglk, dev, a, b, sa, sb, dtype)
    ref = ogemm.fp8_blockwise_scaled_mm(a, b, sa, sb, dtype)
    torch.testing.assert_close(out, ref, rtol=0.02, atol=1)  # reference tolerance (:83-85)
    # the reference tolerance is loose for |out| ~ 1e-2; also require agreement to output rounding
    torch.testing.assert_close(out.float(), ref.float(), rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize("M", [129, 255, 257, 384, 1000, 2049, 4100])
@pytest.mark.parametrize("N,K", [(512, 1024), (4096, 512), (640, 1152), (14080, 1024), (1000, 768), (136, 256), (2056, 2048)])
def test_fp8_blockwise_tile_pipeline_edges(sglk, dev, M, N, K):
    """the persistent tile loop (M > 128): full and 128-row half tiles, ragged last row / column blocks (N % 8 == 0 only),
    two-block K, rounds that do and do not fill the 256 workgroups - every output element against the oracle"""
    dtype = torch.bfloat16 if (M + N) % 2 else torch.float16
    a, b, sa, sb = make_blockwise(M, N, K, seed=M + N + K)
    out = run_blockwise(sglk, dev, a, b, sa, sb, dtype)
    ref = ogemm.fp8_blockwise_scaled_mm(a, b, sa, sb, dtype)
    torch.testing.assert_close(out.float(), ref.float(), rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize("M", [65, 100, 128, 129, 200, 256, 300, 385, 512])
@pytest.mark.parametrize("N,K", [(4096, 14336), (1024, 7168), (640, 3072), (2056, 12288), (136, 16384), (4096, 3584)])
def test_fp8_blockwise_k_slices(sglk, dev, M, N, K):
    """few rows over a deep K: tile x K-slice units with fp32 partial tiles in a scratch tensor, added in slice order (8, 4 or 2 slices
    by the number of half tiles; ragged rows / columns; the last shape's 28 K blocks are too few to split) - every output element
    against the oracle, and twice the same bits"""
    dtype = torch.bfloat16 if (M + N) % 2 else torch.float16
    a, b, sa, sb = make_blockwise(M, N, K, seed=M + N + K)
    out = run_blockwise(sglk, dev, a, b, sa, sb, dtype)
    ref = ogemm.fp8_blockwise_scaled_mm(a, b, sa, sb, dtype)
    torch.testing.assert_close(out.float(), ref.float(), rtol=2e-2, atol=2e-3)
    assert torch.equal(out, run_blockwise(sglk, dev, a, b, sa, sb, dtype))


def test_fp8_blockwise_workspace_entry(dev):
    """C-ABI: _workspace_size names the scratch bytes of the K-slice path; _ws with no / a short workspace runs unsplit and agrees
    with the split result to output rounding"""
    lib = ctypes.CDLL(os.path.join(PKG, "sgl_kernel", "libsglk.so"))
    lib.sglk_fp8_blockwise_scaled_mm_workspace_size.restype = ctypes.c_int64
    lib.sglk_fp8_blockwise_scaled_mm_workspace_size.argtypes = [ctypes.c_int64] * 3
    M, N, K = 256, 4096, 14336
    size = lib.sglk_fp8_blockwise_scaled_mm_workspace_size(M, N, K)
    assert size == 8 * M * N * 4
    assert lib.sglk_fp8_blockwise_scaled_mm_workspace_size(4096, 14336, 4096) == 0
    assert lib.sglk_fp8_blockwise_scaled_mm_workspace_size(16, N, K) == 0
    a, b, sa, sb = make_blockwise(M, N, K, seed=5)
    ad, bd, sad, sbd = a.to(dev), to_dev_colmajor(b, dev), to_dev_colmajor(sa, dev), to_dev_colmajor(sb, dev)
    outs = []
    for ws_bytes in (size, size - 16, 0):
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        vp = ctypes.c_void_p
        i64 = ctypes.c_int64
        rc = lib.sglk_fp8_blockwise_scaled_mm_ws(vp(torch.cuda.current_stream().cuda_stream), vp(out.data_ptr()), vp(ad.data_ptr()),
                                                 vp(bd.data_ptr()), vp(sad.data_ptr()), vp(sbd.data_ptr()), i64(M), i64(N), i64(K),
                                                 i64(ad.stride(0)), i64(bd.stride(1)), i64(N), i64(sad.stride(0)), i64(sad.stride(1)),
                                                 i64(sbd.stride(0)), i64(sbd.stride(1)), ctypes.c_int(2),  # (SGLK_BF16)
                                                 vp(ws.data_ptr()) if ws_bytes else vp(None), i64(ws_bytes))
        assert rc == 0
        torch.cuda.synchronize()
        outs.append(out.cpu())
    ref = ogemm.fp8_blockwise_scaled_mm(a, b, sa, sb, torch.bfloat16)
    for o in outs:
        torch.testing.assert_close(o.float(), ref.float(), rtol=2e-2, atol=2e-3)
    assert torch.equal(outs[1], outs[2])  # both unsplit


@pytest.mark.parametrize("M", [1, 3, 5, 127, 128, 512, 1024, 4096])
@pytest.mark.parametrize("N", [128, 512, 1024, 4096, 8192, 14080])
@pytest.mark.parametrize("K", [512, 1024, 4096, 8192, 14080, 16384])
def test_fp8_blockwise_reference_grid(sglk, dev, M, N, K):
    """the reference's whole M x N x K cross product (tests/test_fp8_blockwise_gemm.py:88-91; its out_dtype axis alternates
    over the grid). Operands are drawn on the device; the oracle runs on up to 48 sampled rows x 3 sampled 128-column
    blocks (all rows / blocks of the small cases), and every output must be finite."""
    dtype = torch.bfloat16 if (M + N // 128 + K // 512) % 2 else torch.float16
    g = torch.Generator(device=dev).manual_seed(M * 31 + N + K)
    a = ((torch.rand(M, K, device=dev, generator=g) - 0.5) * 2 * FMAX).clamp(-FMAX, FMAX).to(FP8)
    b = ((torch.rand(N, K, device=dev, generator=g) - 0.5) * 2 * FMAX).clamp(-FMAX, FMAX).to(FP8).t()
    sa = (torch.randn(K // 128, M, device=dev, generator=g) * 1e-3).t()   # column-major, like the reference's
    sb = (torch.randn(N // 128, K // 128, device=dev, generator=g) * 1e-3).t()
    out = sglk.fp8_blockwise_scaled_mm(a, b, sa, sb, dtype)
    assert out.shape == (M, N) and out.dtype == dtype and torch.isfinite(out.float()).all()
    gc = torch.Generator().manual_seed(K + M)
    rows = torch.randperm(M, generator=gc)[:48].sort().values
    nblk = torch.randperm(N // 128, generator=gc)[:3].sort().values
    cols = torch.cat([torch.arange(i * 128, (i + 1) * 128) for i in nblk.tolist()])
    rd, cd, nd = rows.to(dev), cols.to(dev), nblk.to(dev)
    ref = ogemm.fp8_blockwise_scaled_mm(a[rd].cpu(), b[:, cd].cpu(), sa[rd].cpu(), sb[:, nd].cpu(), dtype)
    torch.testing.assert_close(out[rd][:, cd].float().cpu(), ref.float(), rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize("M,N,K", [(4096, 14336, 4096), (4096, 4096, 14336), (1024, 14080, 16384), (1, 14336, 4096),
                                   (16, 14336, 4096), (64, 4096, 14336), (300, 768, 384)])
def test_fp8_blockwise_full_size_sampled(sglk, dev, M, N, K):
    a, b, sa, sb = make_blockwise(M, N, K, seed=5)
    out = run_blockwise(sglk, dev, a, b, sa, sb, torch.bfloat16)
    g = torch.Generator().manual_seed(6)
    rows = torch.randperm(M, generator=g)[:48].sort().values
    nblk = torch.randperm((N + 127) // 128, generator=g)[:3].sort().values
    cols = torch.cat([torch.arange(i * 128, min(N, (i + 1) * 128)) for i in nblk.tolist()])
    ref = ogemm.fp8_blockwise_scaled_mm(a[rows], b[:, cols], sa[rows], sb[:, nblk], torch.bfloat16)
    torch.testing.assert_close(out[rows][:, cols].float(), ref.float(), rtol=2e-2, atol=2e-3)
    # tile edges: last rows / last columns are written, nothing is left unwritten
    assert torch.isfinite(out.float()).all()
    # linearity in the scales: doubling sa doubles every partial exactly (power-of-two scaling is exact in
    # fp32 and bf16), so the outputs must be bit-identical up to the factor
    out2 = run_blockwise(sglk, dev, a, b, sa * 2, sb, torch.bfloat16)
    assert torch.equal(out2.float(), out.float() * 2)


def test_fp8_mfma_encodings_agree(dev):
    """The MX encoding with unit E8M0 scales and the plain K=128 encoding must be the same arithmetic. The switch
    exists only in the diagnostic library (build.py --probes); the call goes straight through its C-ABI."""
    path = os.path.join(os.path.dirname(PKG), "build", "libsglk_probes.so")
    if not os.path.exists(path):
        pytest.skip("diagnostic library not built (python sgl-kernel-xpu_amd/build.py --probes)")
    lib = ctypes.CDLL(path)
    a, b, sa, sb = make_blockwise(200, 640, 1024, seed=9)
    M, K = a.shape
    N = b.shape[1]
    ad, bd, sad, sbd = a.to(dev), b.to(dev), sa.to(dev), sb.to(dev)
    I64, P = ctypes.c_int64, ctypes.c_void_p
    lib.sglk_fp8_blockwise_scaled_mm.argtypes = [P, P, P, P, P, P] + [I64] * 10 + [ctypes.c_int]
    lib.sglk_fp8_blockwise_scaled_mm.restype = ctypes.c_int
    outs = []
    try:
        for form in (1, 0):
            lib.sglk_debug_set_fp8_mfma_form(form)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            rc = lib.sglk_fp8_blockwise_scaled_mm(
                None, out.data_ptr(), ad.data_ptr(), bd.data_ptr(), sad.data_ptr(), sbd.data_ptr(), M, N, K,
                ad.stride(0), bd.stride(1), out.stride(0), sad.stride(0), sad.stride(1), sbd.stride(0), sbd.stride(1), 2)
            assert rc == 0
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        lib.sglk_debug_set_fp8_mfma_form(1)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


def test_fp8_blockwise_golden(sglk, dev):
    for c in load_golden("fp8_blockwise_gemm"):
        a = c["a"].view(FP8)
        b = c["b_nk"].view(FP8).t()
        out = run_blockwise(sglk, dev, a, b, c["sa"], c["sb"], c["out_dtype"])
        torch.testing.assert_close(out, c["out"], rtol=0.02, atol=1)
        torch.testing.assert_close(out.float(), c["out"].float(), rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize("M", [1, 24, 100, 128, 512, 777])
@pytest.mark.parametrize("N,K", [(16, 512), (128, 1024), (512, 4096), (4096, 512), (1024, 8192), (320, 1152),
                                 (12288, 512)])  # (the last: 128-row workgroups of the few-row kernel at M = 100, 128)
@pytest.mark.parametrize("with_bias", [True, False])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fp8_scaled_mm(sglk, dev, M, N, K, with_bias, dtype):
    g = torch.Generator().manual_seed(M + N + K)
    a = ((torch.rand(M, K, generator=g) - 0.5) * 2 * FMAX).clamp(-FMAX, FMAX).to(FP8)
    b = ((torch.rand(N, K, generator=g) - 0.5) * 2 * FMAX).clamp(-FMAX, FMAX).to(FP8).t()
    sa = torch.randn(M, generator=g) * 0.001
    sb = torch.randn(N, generator=g) * 0.001
    bias = torch.randn(N, generator=g).to(dtype) if with_bias else None
    out = sglk.fp8_scaled_mm(a.to(dev), to_dev_colmajor(b, dev), sa.to(dev), sb.to(dev), dtype,
                             bias.to(dev) if with_bias else None).cpu()
    ref = ogemm.fp8_scaled_mm(a, b, sa, sb, dtype, bias)
    torch.testing.assert_close(out, ref, rtol=0.02, atol=1)  # tests/test_fp8_gemm.py:38-40
    torch.testing.assert_close(out.float(), ref.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("M", [1, 16, 64, 90, 512, 1000])
@pytest.mark.parametrize("N,K", [(16, 512), (128, 1024), (1024, 4096), (8192, 512), (16384, 1024), (320, 1152)])
@pytest.mark.parametrize("with_bias", [True, False])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_int8_scaled_mm(sglk, dev, M, N, K, with_bias, dtype):
    g = torch.Generator().manual_seed(M * 3 + N + K)
    a = torch.round((torch.randn(M, K, generator=g) * 5).clamp(-128, 127)).to(torch.int8)
    b = torch.round((torch.randn(N, K, generator=g) * 5).clamp(-128, 127)).to(torch.int8).t()
    sa = torch.randn(M, generator=g)
    sb = torch.randn(N, generator=g)
    bias = (torch.randn(N, generator=g).to(dtype) * 10) if with_bias else None
    out = sglk.int8_scaled_mm(a.to(dev), to_dev_colmajor(b, dev), sa.to(dev), sb.to(dev), dtype,
                              bias.to(dev) if with_bias else None).cpu()
    ref = ogemm.int8_scaled_mm(a, b, sa, sb, dtype, bias)
    torch.testing.assert_close(out, ref)  # default tolerances, as tests/test_int8_gemm.py:36
    # int32 accumulation is exact and the epilogue is the oracle's fp32 expression: expect bit-equality
    # on all but the rare element where fp32 a*b*c association differs
    assert (out != ref).float().mean() < 1e-3


@pytest.mark.parametrize("kind", ["fp8", "int8"])
@pytest.mark.parametrize("M", [129, 200, 256, 300, 512, 1000])
@pytest.mark.parametrize("N,K", [(4096, 14336), (1024, 7168), (640, 4096), (2056, 12288)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_scaled_mm_k_slices(sglk, dev, kind, M, N, K, with_bias):
    """129 .. 1024 rows over a deep K: tile x K-slice units with raw accumulators in a scratch tensor, summed and scaled by a second
    kernel in the epilogue's order (8, 4 or 2 slices; ragged rows / columns) - every output element against the oracle; int8 as exact
    as the unsplit path (integer sums), twice the same bits"""
    dtype = torch.bfloat16 if (M + N) % 2 else torch.float16
    g = torch.Generator().manual_seed(M + N + K)
    if kind == "fp8":
        a = ((torch.rand(M, K, generator=g) - 0.5) * 2 * FMAX).clamp(-FMAX, FMAX).to(FP8)
        b = ((torch.rand(N, K, generator=g) - 0.5) * 2 * FMAX).clamp(-FMAX, FMAX).to(FP8).t()
        sa, sb = torch.randn(M, generator=g) * 0.001, torch.randn(N, generator=g) * 0.001
        bias = torch.randn(N, generator=g).to(dtype) if with_bias else None
        run = lambda: sglk.fp8_scaled_mm(a.to(dev), to_dev_colmajor(b, dev), sa.to(dev), sb.to(dev), dtype,
                                         bias.to(dev) if with_bias else None).cpu()
        out = run()
        ref = ogemm.fp8_scaled_mm(a, b, sa, sb, dtype, bias)
        torch.testing.assert_close(out.float(), ref.float(), rtol=2e-2, atol=2e-2)
    else:
        a = torch.round((torch.randn(M, K, generator=g) * 5).clamp(-128, 127)).to(torch.int8)
        b = torch.round((torch.randn(N, K, generator=g) * 5).clamp(-128, 127)).to(torch.int8).t()
        sa, sb = torch.randn(M, generator=g), torch.randn(N, generator=g)
        bias = (torch.randn(N, generator=g).to(dtype) * 10) if with_bias else None
        run = lambda: sglk.int8_scaled_mm(a.to(dev), to_dev_colmajor(b, dev), sa.to(dev), sb.to(dev), dtype,
                                          bias.to(dev) if with_bias else None).cpu()
        out = run()
        ref = ogemm.int8_scaled_mm(a, b, sa, sb, dtype, bias)
        torch.testing.assert_close(out, ref)
        assert (out != ref).float().mean() < 1e-3
    assert torch.equal(out, run())


@pytest.mark.parametrize("kind", ["fp8", "int8"])
@pytest.mark.parametrize("M,N,K", [(4096, 14336, 4096), (2304, 7424, 1024), (1024, 4096, 14336), (640, 4104, 512)])
def test_scaled_mm_full_size_sampled(sglk, dev, kind, M, N, K):
    """The persistent pipeline in the row / column scale modes at sizes the CPU oracle cannot cover whole: whole rounds
    plus a last partial round of half tiles (896 and 261 tiles), every tile as two halves (64 tiles), an N that is not
    a multiple of 256. Oracle on sampled rows (all columns); finiteness and exact row-scale linearity on everything."""
    g = torch.Generator().manual_seed(M + N + K)
    if kind == "fp8":
        a = ((torch.rand(M, K, generator=g) - 0.5) * 2 * FMAX).clamp(-FMAX, FMAX).to(FP8)
        b = ((torch.rand(N, K, generator=g) - 0.5) * 2 * FMAX).clamp(-FMAX, FMAX).to(FP8).t()
        sa, sb = torch.rand(M, generator=g) * 1e-3 + 1e-4, torch.rand(N, generator=g) * 1e-3 + 1e-4
        op, ref_op = sglk.fp8_scaled_mm, ogemm.fp8_scaled_mm
    else:
        a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8)
        b = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).t()
        sa, sb = torch.rand(M, generator=g) * 1e-2 + 1e-3, torch.rand(N, generator=g) * 1e-2 + 1e-3
        op, ref_op = sglk.int8_scaled_mm, ogemm.int8_scaled_mm
    bias = torch.randn(N, generator=g).to(torch.bfloat16)
    ad, bd, sbd, biasd = a.to(dev), to_dev_colmajor(b, dev), sb.to(dev), bias.to(dev)
    out = op(ad, bd, sa.to(dev), sbd, torch.bfloat16, biasd).cpu()
    assert torch.isfinite(out.float()).all()
    rows = torch.randperm(M, generator=g)[:40].sort().values
    rows = torch.cat([rows, torch.tensor([0, M - 1])]).unique()
    ref = ref_op(a[rows], b, sa[rows], sb, torch.bfloat16, bias)
    torch.testing.assert_close(out[rows].float(), ref.float(), rtol=2e-2, atol=2e-2)
    # doubling the row scales doubles every product exactly (power of two), the bias term stays: (out2 - bias) == 2 (out - bias)
    # holds only up to the output rounding, so check it without the bias, where it is exact
    o1 = op(ad, bd, sa.to(dev), sbd, torch.bfloat16, None).cpu()
    o2 = op(ad, bd, (sa * 2).to(dev), sbd, torch.bfloat16, None).cpu()
    assert torch.equal(o2.float(), o1.float() * 2)


def test_scaled_mm_golden(sglk, dev):
    for c in load_golden("scaled_mm"):
        bias = c["bias"].to(dev) if c["bias"] is not None else None
        if c["kind"] == "fp8":
            out = sglk.fp8_scaled_mm(c["a"].view(FP8).to(dev), c["b_nk"].view(FP8).to(dev).t(), c["sa"].to(dev),
                                     c["sb"].to(dev), c["out_dtype"], bias).cpu()
            torch.testing.assert_close(out, c["out"], rtol=0.02, atol=1)
        else:
            out = sglk.int8_scaled_mm(c["a"].to(dev), c["b_nk"].to(dev).t(), c["sa"].to(dev), c["sb"].to(dev),
                                      c["out_dtype"], bias).cpu()
            torch.testing.assert_close(out, c["out"])


def test_errors(sglk, dev):
    a = torch.zeros(4, 256, dtype=FP8, device=dev)
    b = torch.zeros(128, 256, dtype=FP8, device=dev).t()
    sa = torch.ones(4, 2, device=dev)
    sb = torch.ones(2, 1, device=dev)
    with pytest.raises(RuntimeError, match="column major"):
        sglk.fp8_blockwise_scaled_mm(a, b.contiguous(), sa, sb, torch.bfloat16)
    with pytest.raises(RuntimeError, match="scales_a"):
        sglk.fp8_blockwise_scaled_mm(a, b, torch.ones(4, 3, device=dev), sb, torch.bfloat16)
    with pytest.raises(RuntimeError, match="out_dtype"):
        sglk.fp8_blockwise_scaled_mm(a, b, sa, sb, torch.float32)
    with pytest.raises(RuntimeError, match="multiple of 128"):
        sglk.fp8_blockwise_scaled_mm(a[:, :192], b[:192], torch.ones(4, 1, device=dev), sb, torch.bfloat16)
    assert sglk.fp8_blockwise_scaled_mm(a[:0], b, sa[:0], sb, torch.bfloat16).shape == (0, 128)
