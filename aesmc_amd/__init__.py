"""aesmc_amd — the batched SMC / importance-sampling inner loop of aesmc (tuananhle7/aesmc) on
AMD Instinct MI355X: hand-written HIP kernels (gfx950) behind the reference's Python interface.

Public surface mirrors the reference package: `inference`, `losses`, `state`, `math`,
`statistics`, `train`.  Importing the package does not touch the GPU; the first kernel call loads
libaesmc_hip.so and raises if it has not been built (`python -m aesmc_amd.build`).
"""
from . import inference  # noqa: F401
from . import losses  # noqa: F401
from . import math  # noqa: F401
from . import state  # noqa: F401
from . import statistics  # noqa: F401
from . import train  # noqa: F401

__version__ = "0.1.0"
