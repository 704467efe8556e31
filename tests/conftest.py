"""pytest configuration.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks (CPU, minutes).
`-m gpu`      : parity of the HIP path (through torch.ops.sgl_kernel -> C-ABI) with the oracle.
Fixture style follows reference tests/conftest.py: short tensor printing, default
dtype/device reset around every test, device sync + cache release after GPU tests.
"""
import os
import sys

# The GPU boxes are 256-core hosts shared with other jobs; the per-test oracle work is small tensors on the CPU, where a
# thread pool as wide as the host costs more in wake-ups (and in waiting for cores other tenants hold: the same suite took
# 565 .. 1026 s box to box) than it gains. Cap the pools before torch / numpy start them.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, str(min(16, os.cpu_count() or 1)))

import pytest
import torch

torch.set_num_threads(min(16, os.cpu_count() or 1))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sgl-kernel-xpu_amd", "python")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

torch.set_printoptions(threshold=0)  # never format a (possibly poisoned) large tensor in a failure message

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=False)


@pytest.fixture(autouse=True)
def _reset_defaults():
    dt = torch.get_default_dtype()
    yield
    torch.set_default_dtype(dt)


@pytest.fixture(autouse=True)
def _gpu_cleanup(request):
    yield
    if request.node.get_closest_marker("gpu") is not None and torch.cuda.is_available():
        torch.cuda.synchronize()


@pytest.fixture(scope="session")
def sglk():
    """The product package; import fails loudly if the HIP extension is missing."""
    import sgl_kernel

    return sgl_kernel


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda:0")
