"""CPU: host-side behaviour of the python package and the torch registry (no GPU needed):
import, op registration with the reference's schemas, argument errors raised before launch,
and the absence of any CPU fallback."""
import pytest
import torch


def test_import_registers_ops(sglk):
    for name in ["rmsnorm", "fused_add_rmsnorm", "gemma_rmsnorm", "gemma_fused_add_rmsnorm", "silu_and_mul",
                 "gelu_tanh_and_mul", "gelu_and_mul", "sgl_per_token_group_quant_8bit",
                 "fp8_blockwise_scaled_mm"]:
        assert hasattr(torch.ops.sgl_kernel, name), name


def test_schemas_match_reference_contract(sglk):
    # reference src/torch_extension_sycl.cc:41, :29, :395-398
    s = torch.ops.sgl_kernel.rmsnorm.default._schema
    assert str(s) == "sgl_kernel::rmsnorm(Tensor($0! -> ) output, Tensor input, Tensor weight, float eps) -> ()"
    s = torch.ops.sgl_kernel.sgl_per_token_group_quant_8bit.default._schema
    assert [a.name for a in s.arguments] == ["input", "output_q", "output_s", "group_size", "eps", "fp8_min",
                                             "fp8_max", "scale_ue8m0"]
    s = torch.ops.sgl_kernel.fp8_blockwise_scaled_mm.default._schema
    assert [a.name for a in s.arguments] == ["mat_a", "mat_b", "scales_a", "scales_b", "out_dtype"]


def test_no_cpu_fallback(sglk):
    x, w = torch.randn(2, 8), torch.randn(8)
    with pytest.raises(NotImplementedError):
        sglk.rmsnorm(x, w)
    with pytest.raises(NotImplementedError):
        sglk.silu_and_mul(torch.randn(2, 16))


def test_activation_row_rule(sglk):
    # reference python/sgl_kernel/elementwise.py:216-217
    with pytest.raises(ValueError, match="multiple of 16 bytes"):
        sglk.silu_and_mul(torch.randn(2, 6, dtype=torch.float16))


def test_out_of_scope_names_raise_on_call_only(sglk):
    f = sglk.merge_state
    with pytest.raises(NotImplementedError):
        f()
    with pytest.raises(AttributeError):
        sglk.definitely_not_an_op
