"""aesmc_amd — the batched SMC / importance-sampling inner loop of aesmc (tuananhle7/aesmc) on
AMD Instinct MI355X: hand-written HIP kernels (gfx950) behind the reference's Python interface.

Public surface mirrors the reference package: `inference`, `losses`, `state`, `math`,
`statistics`, `train`.  Importing the package does not touch the GPU; the first kernel call loads
libaesmc_hip.so and raises if it has not been built (`python -m aesmc_amd.build`).
"""
import os as _os

import torch as _torch

# hipGraph workaround, needed before the HIP runtime starts.  ROCm 7.0's graph fast path ("AQL
# packet capture") was found to run captured MEMSET nodes out of stream order: in a captured
# backward pass PyTorch's multi-block reductions zero their semaphores with cudaMemsetAsync, and
# from the third or fourth replay on the gradients of small broadcast parameters came out wrong
# (losses stayed right).  With the fast path off every replay equals the eager gradients bit for
# bit (tests/test_gpu_graphs.py) at no cost (4.36-4.40 ms per ELBO at configs[1] with the fast
# path off, 4.41-4.42 with it on).
# The variable is read once, at the first HIP call (torch.cuda.is_available() already makes one,
# without torch.cuda.is_initialized() turning true): export it in the shell or launcher, or import
# this package before anything touches the GPU.  A value the user has set is left alone.
HIPGRAPH_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def _hip_runtime_started():
    """True once this process has opened the GPU driver (/dev/kfd): the HIP runtime has then read its
    environment.  torch.cuda.is_initialized() is not the test — is_available() and device_count()
    start the runtime without setting it."""
    if _torch.cuda.is_initialized():
        return True
    try:
        for fd in _os.listdir("/proc/self/fd"):
            try:
                if _os.readlink("/proc/self/fd/" + fd).startswith("/dev/kfd"):
                    return True
            except OSError:
                continue
    except OSError:
        pass
    return False


if _os.environ.get(HIPGRAPH_ENV) is not None:
    HIPGRAPH_MEMSET_WORKAROUND = "preset:" + _os.environ[HIPGRAPH_ENV]
elif _hip_runtime_started():
    HIPGRAPH_MEMSET_WORKAROUND = "too-late"   # graphs.GraphedLoss(backward=True) refuses to capture
else:
    _os.environ[HIPGRAPH_ENV] = "0"
    HIPGRAPH_MEMSET_WORKAROUND = "set"

from . import inference  # noqa: F401
from . import losses  # noqa: F401
from . import math  # noqa: F401
from . import settings  # noqa: F401
from . import state  # noqa: F401
from . import statistics  # noqa: F401
from . import train  # noqa: F401

__version__ = "0.5.1"      # = the C ABI's aesmc_version() / 100 (include/aesmc_hip.h holds the history)
