"""sgl_kernel for AMD Instinct MI355X (gfx950).

Same module and function names as reference python/sgl_kernel/__init__.py:14-184
for the hot path; every op dispatches through torch.ops.sgl_kernel.* (registered
by the `common_ops` extension at import) into hand-written HIP kernels behind
the C-ABI of include/sglk.h. There is no CPU or eager fallback: if the
extension is missing the import fails.
"""
import os as _os

import torch as _torch  # noqa: F401  (loads the HIP runtime the extension binds to)

_here = _os.path.dirname(_os.path.abspath(__file__))
if not any(f.startswith("common_ops") and f.endswith(".so") for f in _os.listdir(_here)):
    raise ImportError(
        "sgl_kernel: the HIP extension common_ops*.so is not built; run "
        "`python sgl-kernel-xpu_amd/build.py` (or __graft_entry__.build())."
    )

from sgl_kernel import common_ops  # noqa: E402,F401  (TORCH_LIBRARY registration happens at dlopen)
from sgl_kernel.attention import (  # noqa: E402
    flash_mla_decode,
    flash_mla_get_workspace_size,
    flash_mla_prefill,
    flash_mla_prefill_get_workspace_size,
    merge_state,
    merge_state_v2,
)
from sgl_kernel.elementwise import (  # noqa: E402
    fused_add_rmsnorm,
    fused_inplace_qknorm_rope,
    fused_qk_norm_rope,
    gelu_and_mul,
    gelu_tanh_and_mul,
    gemma_fused_add_rmsnorm,
    gemma_rmsnorm,
    rmsnorm,
    rotary_embedding,
    silu_and_mul,
    silu_and_mul_clamp,
    store_cache,
    store_cache_xpu,
)
from sgl_kernel.flash_attn import flash_attn_varlen_func, flash_attn_with_kvcache, is_fa3_supported  # noqa: E402
from sgl_kernel.gemm import (  # noqa: E402
    awq_dequantize,
    sgl_per_tensor_quant_fp8,
    sgl_per_token_quant_fp8,
    fp8_blockwise_scaled_mm,
    fp8_scaled_mm,
    int8_scaled_mm,
    qserve_w4a8_per_chn_gemm,
    qserve_w4a8_per_group_gemm,
    sgl_per_token_group_quant_8bit,
    sgl_per_token_group_quant_fp8,
    sgl_per_token_group_quant_int8,
)
from sgl_kernel.moe import (  # noqa: E402
    apply_shuffle_mul_sum,
    biased_topk,
    fused_experts,
    swiglu_gpt_oss_sigmoid_alpha,
    moe_align_block_size,
    moe_fused_gate,
    prepare_moe_input,
    scatter_tokens_to_experts,
    topk_sigmoid,
    topk_softmax,
)
from sgl_kernel.sampling import (  # noqa: E402
    min_p_sampling_from_probs,
    top_k_renorm_prob,
    top_k_renorm_probs,
    top_k_top_p_sampling_from_probs,
    top_p_renorm_prob,
    top_p_renorm_probs,
    top_p_sampling_from_probs,
)
from sgl_kernel.utils import get_device_capability, is_gfx950_arch, is_xe2_arch  # noqa: E402
from sgl_kernel.version import __version__  # noqa: E402

# Public names of the reference package that are outside the MI355X hot path
# (SURVEY.md section 2.3). They resolve lazily to a stub that raises on call, so
# `from sgl_kernel import X` keeps working for callers that never use X.
_OUT_OF_SCOPE = frozenset(
    """
    bmm_fp8 cutlass_scaled_fp4_mm scaled_fp4_experts_quant scaled_fp4_quant
    sgl_per_token_group_quant_fp4
    lightning_attention_decode flash_mla_sparse_fwd flash_mla_with_kvcache
    apply_rope_with_cos_sin_cache_inplace fused_k_norm_rope_flashmla
    fused_q_norm_rope fused_qk_rope fused_qk_rope_with_cos_sin_cache_inplace
    multimodal_rotary_embedding
    cutlass_fp4_group_mm fp8_blockwise_scaled_grouped_mm hash_topk moe_sum
    moe_sum_reduce
    weak_ref_tensor
    """.split()
)


def __getattr__(name):
    if name in _OUT_OF_SCOPE:

        def _stub(*args, **kwargs):
            raise NotImplementedError(
                f"sgl_kernel.{name} is outside the MI355X hot path of this build (see DESIGN.md)"
            )

        _stub.__name__ = name
        return _stub
    raise AttributeError(f"module 'sgl_kernel' has no attribute {name!r}")
