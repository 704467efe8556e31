"""CPU: the oracle restatements against golden vectors produced by the reference's own
torch-eager test references (tests/golden/make_golden.py). Tolerances are the
reference tests' (cited per test)."""
import torch
from conftest import load_golden

from oracle import activation as oact
from oracle import gemm as ogemm
from oracle import norm as onorm
from oracle import quant as oquant


def norm_tol(dtype):  # reference tests/test_norm.py:45-50
    if dtype == torch.float32:
        return dict(rtol=1e-4, atol=1e-4)
    if dtype == torch.bfloat16:
        return dict(rtol=1e-2, atol=1e-2)
    return dict(rtol=1e-3, atol=1e-3)


def test_norm_family_matches_reference_vectors():
    for c in load_golden("norm"):
        x, r, w, eps = c["x"], c["residual"], c["w"], c["eps"]
        tol = norm_tol(x.dtype)
        torch.testing.assert_close(onorm.rmsnorm(x, w, eps), c["rmsnorm"], **tol)
        torch.testing.assert_close(onorm.gemma_rmsnorm(x, w, eps), c["gemma_rmsnorm"], **tol)
        y, nr = onorm.fused_add_rmsnorm(x, r, w, eps)
        torch.testing.assert_close(y, c["fused_add"][0], **tol)
        torch.testing.assert_close(nr, c["fused_add"][1], **tol)
        y, nr = onorm.gemma_fused_add_rmsnorm(x, r, w, eps)
        torch.testing.assert_close(y, c["gemma_fused_add"][0], **tol)
        torch.testing.assert_close(nr, c["gemma_fused_add"][1], **tol)


def test_activation_matches_reference_vectors():
    for c in load_golden("activation"):
        x = c["x"]
        tol = dict(rtol=1e-3, atol=1e-3) if x.dtype == torch.float16 else dict(rtol=1e-2, atol=1e-2)
        torch.testing.assert_close(oact.silu_and_mul(x), c["silu"], **tol)  # tests/test_activation.py:20
        torch.testing.assert_close(oact.gelu_tanh_and_mul(x), c["gelu_tanh"], **tol)
        torch.testing.assert_close(oact.gelu_and_mul(x), c["gelu"], **tol)


def test_quant_matches_reference_vectors():
    for c in load_golden("quant"):
        x, gs = c["x"], c["group_size"]
        rows, k = x.shape
        q, s, _ = oquant.per_token_group_quant_8bit(x, gs, torch.float8_e4m3fn)
        # reference tests/test_per_token_group_quant_8bit.py:260-276
        torch.testing.assert_close(s, c["fp8_s"], rtol=1e-3, atol=1e-5)
        ref_q = c["fp8_q"].view(torch.float8_e4m3fn)
        deq = q.float().view(rows, -1, gs) * s.unsqueeze(-1)
        ref_deq = ref_q.float().view(rows, -1, gs) * c["fp8_s"].unsqueeze(-1)
        torch.testing.assert_close(deq, ref_deq, rtol=1e-1, atol=1e-1)
        # the two only differ where x*(1/s) and x/s round to different fp8 codes (1 ulp)
        assert (q.view(torch.uint8) != c["fp8_q"]).float().mean() < 0.02

        q, s, ue = oquant.per_token_group_quant_8bit(x, gs, torch.float8_e4m3fn, scale_ue8m0=True)
        torch.testing.assert_close(s, c["fp8_ue8m0_s"], rtol=0, atol=0)  # powers of two: exact (:405)
        assert torch.equal(ue.to(torch.int32) - 127, torch.log2(s).round().to(torch.int32))

        q, s, _ = oquant.per_token_group_quant_8bit(x, gs, torch.int8)
        torch.testing.assert_close(s, c["int8_s"], rtol=1e-3, atol=1e-5)
        assert (q.to(torch.int32) - c["int8_q"].to(torch.int32)).abs().max() <= 1


def test_fp8_blockwise_matches_reference_vectors():
    for c in load_golden("fp8_blockwise_gemm"):
        a = c["a"].view(torch.float8_e4m3fn)
        b = c["b_nk"].view(torch.float8_e4m3fn).t()
        out = ogemm.fp8_blockwise_scaled_mm(a, b, c["sa"], c["sb"], c["out_dtype"])
        torch.testing.assert_close(out, c["out"], rtol=0.02, atol=1)  # tests/test_fp8_blockwise_gemm.py:83-85
        torch.testing.assert_close(out.float(), c["out"].float(), rtol=1e-2, atol=1e-4)


def test_scaled_mm_matches_reference_vectors():
    for c in load_golden("scaled_mm"):
        if c["kind"] == "fp8":
            a = c["a"].view(torch.float8_e4m3fn)
            b = c["b_nk"].view(torch.float8_e4m3fn).t()
            out = ogemm.fp8_scaled_mm(a, b, c["sa"], c["sb"], c["out_dtype"], c["bias"])
            torch.testing.assert_close(out, c["out"], rtol=0.02, atol=1)  # tests/test_fp8_gemm.py:38-40
        else:
            out = ogemm.int8_scaled_mm(c["a"], c["b_nk"].t(), c["sa"], c["sb"], c["out_dtype"], c["bias"])
            torch.testing.assert_close(out, c["out"])  # tests/test_int8_gemm.py:36
