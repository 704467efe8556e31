"""Device queries of the package, answered for ROCm (PyTorch-ROCm exposes HIP devices as "cuda").

Names the reference's callers use (python/sgl_kernel/utils.py:22-23, :45-56): `get_xpu_stream`,
`get_device_capability`, `is_xe2_arch` (the gate at moe.py:717) keep working and answer for gfx950.
"""
import functools
from typing import Tuple

import torch


def get_cuda_stream() -> int:
    """Raw handle of the current HIP stream (what the C-ABI of include/sglk.h takes as sglk_stream_t)."""
    return torch.cuda.current_stream().cuda_stream


get_xpu_stream = get_cuda_stream


@functools.lru_cache(maxsize=1)
def get_device_capability() -> Tuple[int, int]:
    """(major, minor) of the current device; gfx950 reports (9, 5). (0, 0) without a GPU."""
    return torch.cuda.get_device_capability() if torch.cuda.is_available() else (0, 0)


@functools.lru_cache(maxsize=1)
def is_gfx950_arch() -> bool:
    if not torch.cuda.is_available():
        return False
    arch = getattr(torch.cuda.get_device_properties(0), "gcnArchName", "")
    return arch.split(":")[0] == "gfx950"


def is_xe2_arch() -> bool:
    """The reference's gate name: callers that keep asking it get this build's own target."""
    return is_gfx950_arch()
