"""activation-and-mul — restates reference src/sycl/TripleOps.cpp:29-53 (functors)
in fp32 opmath with one rounding, pinned against tests/test_activation.py:18,28,38
golden vectors (tolerance 1e-3 as there)."""
import math

import torch


def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    d = x.shape[-1] // 2
    a, b = x[..., :d].float(), x[..., d:].float()
    return ((a / (1.0 + torch.exp(-a))) * b).to(x.dtype)


def gelu_tanh_and_mul(x: torch.Tensor) -> torch.Tensor:
    d = x.shape[-1] // 2
    a, b = x[..., :d].float(), x[..., d:].float()
    k_beta = math.sqrt(2.0) * (2.0 / math.sqrt(math.pi)) * 0.5
    inner = k_beta * (a + 0.044715 * (a * a * a))
    return ((0.5 * a * (1.0 + torch.tanh(inner))) * b).to(x.dtype)


def gelu_and_mul(x: torch.Tensor) -> torch.Tensor:
    d = x.shape[-1] // 2
    a, b = x[..., :d].float(), x[..., d:].float()
    return ((a * 0.5 * (1.0 + torch.erf(a * math.sqrt(0.5)))) * b).to(x.dtype)


def silu_and_mul_clamp(x: torch.Tensor, limit: float) -> torch.Tensor:
    """reference src/sycl/SiluAndMulClamp.cpp:61-74 (= tests/test_silu_and_mul_clamp.py:8-91): both halves clamped in bf16
    whatever the input dtype - gate = min(gate, limit), up = clamp(up, -limit, limit) - then silu(gate) * up in fp32."""
    d = x.shape[-1] // 2
    lim = torch.tensor(limit, dtype=torch.float32).to(torch.bfloat16).float()
    bf = lambda t: t.to(torch.bfloat16).float()
    g = bf(torch.minimum(bf(x[..., :d].float()), lim))
    u = bf(torch.maximum(-lim, torch.minimum(bf(x[..., d:].float()), lim)))
    return (g * (1.0 / (1.0 + torch.exp(-g))) * u).to(x.dtype)


def swiglu_gpt_oss_sigmoid_alpha(x: torch.Tensor, alpha: float, limit: float) -> torch.Tensor:
    """reference src/sycl/SwigluAlphaLimit.cpp:16-49 (= tests/test_swiglu_with_alpha_limit.py:9-14, in fp32): x [rows, 2 hidden]
    with gate / up interleaved; gate = min(gate, limit), up = clamp(up, -limit, limit), out = gate * sigmoid(alpha gate) * (up + 1)."""
    gate = x[..., ::2].float().clamp(max=limit)
    up = x[..., 1::2].float().clamp(min=-limit, max=limit)
    return (gate * (1.0 / (1.0 + torch.exp(-(gate * alpha)))) * (up + 1.0)).to(x.dtype)
