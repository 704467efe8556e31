"""Device helpers. Mirrors reference python/sgl_kernel/utils.py:22-56 with the
XPU queries re-targeted at ROCm (PyTorch-ROCm exposes HIP devices as "cuda")."""
import functools
from typing import Dict, Tuple

import torch


def get_cuda_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# the reference's name for the same thing (utils.py:22-23)
get_xpu_stream = get_cuda_stream

_cache_buf: Dict[Tuple[str, torch.device], torch.Tensor] = {}


def _get_cache_buf(name: str, bytes: int, device: torch.device) -> torch.Tensor:
    key = (name, device)
    buf = _cache_buf.get(key)
    if buf is None:
        buf = torch.empty(bytes, dtype=torch.uint8, device=device)
        _cache_buf[key] = buf
    return buf


def _to_tensor_scalar_tuple(x):
    if isinstance(x, torch.Tensor):
        return (x, 0)
    return (None, x)


@functools.lru_cache(maxsize=1)
def get_device_capability() -> Tuple[int, int]:
    """(major, minor) of the current device. The reference returns (2, 0) for
    Xe2 (src/sycl/Device.cpp:16-26); here gfx950 reports (9, 5)."""
    if not torch.cuda.is_available():
        return (0, 0)
    return torch.cuda.get_device_capability()


@functools.lru_cache(maxsize=1)
def is_gfx950_arch() -> bool:
    if not torch.cuda.is_available():
        return False
    name = getattr(torch.cuda.get_device_properties(0), "gcnArchName", "")
    return name.split(":")[0] == "gfx950"


def is_xe2_arch() -> bool:
    """Reference gate name (utils.py:45-56, used at moe.py:717). The MI355X build
    answers for its own target so that callers keeping the reference's gate run."""
    return is_gfx950_arch()
