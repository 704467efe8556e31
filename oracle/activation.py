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
