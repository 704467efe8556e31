"""RMSNorm family — restates the reference kernels' arithmetic in torch (CPU).

Follows reference src/sycl/RMSNorm.cpp:
  RMSNormForward.reduce_combine/:78-95   sum of squares in fp32
  reduce_project :97-103                 rstd = rsqrt(max(sum,0)/N + eps)
  update :105-136                        y = T((gamma * rstd) * x)
  AddRMSNormForward :160-181             x = T(x + add) (rounded), stored to both, then as above
  GemmaRMSNormNoRstdForward :436-446     y = T((x * rstd) * (1 + gamma))
and is pinned (tolerance of tests/test_norm.py:45-50) against golden vectors
made from tests/test_norm.py:13-62.
"""
import torch


def _rstd(xf: torch.Tensor, eps: float) -> torch.Tensor:
    ss = (xf * xf).sum(dim=-1, keepdim=True)
    return torch.rsqrt(ss.clamp_min(0.0) / xf.shape[-1] + eps)


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    xf = x.float()
    return ((w.float() * _rstd(xf, eps)) * xf).to(x.dtype)


def gemma_rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    xf = x.float()
    return ((xf * _rstd(xf, eps)) * (1.0 + w.float())).to(x.dtype)


def fused_add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, w: torch.Tensor, eps: float = 1e-6):
    """Returns (normed, new_residual); the op writes them to x and residual in place."""
    r = (x.float() + residual.float()).to(x.dtype)
    return rmsnorm(r, w, eps), r


def gemma_fused_add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, w: torch.Tensor, eps: float = 1e-6):
    r = (x.float() + residual.float()).to(x.dtype)
    return gemma_rmsnorm(r, w, eps), r
