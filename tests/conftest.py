import os
import sys

import pytest

# before anything can start the HIP runtime (torch.cuda.is_available() does): see aesmc_amd/__init__.py
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def oracle_backend():
    """Runs aesmc_amd's HOST logic on CPU tensors by substituting the kernel provider with the
    NumPy oracle (tests only; the product has no CPU path)."""
    from aesmc_amd import _kernels
    from tests.oracle_provider import OracleKernels
    previous = _kernels._swap_provider_for_tests(OracleKernels())
    try:
        yield
    finally:
        _kernels._swap_provider_for_tests(previous)


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda", 0)
