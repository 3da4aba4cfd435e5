"""PyTorch's own noise stream, addressed instead of materialised.

`torch.empty(shape).normal_()` on a HIP device — the `_standard_normal` call behind `Normal.rsample`
(aesmc/state.py:98) — is a pure function of the generator's (seed, offset), of the launch geometry ATen
picks from the device properties, and of the element index (csrc/philox_normal.hpp).  `reserve(numel, device)`
does the bookkeeping half of such a call WITHOUT launching it: it reads (seed, offset), advances the generator
by what `normal_` would have consumed, and returns the stream descriptor a kernel needs to form the same
values where it consumes them.  Everything drawn afterwards — by PyTorch or by this package — is unchanged.
"""
import collections

import torch

NoiseStream = collections.namedtuple("NoiseStream", "seed offset threads numel")

_BLOCK = 256
_GEOMETRY = {}


def launch_threads(numel, device):
    """Threads of ATen's `distribution_elementwise_grid_stride_kernel` launch for `numel` elements
    (ATen/native/cuda/DistributionTemplates.h `calc_execution_policy`)."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    cap = _GEOMETRY.get(index)
    if cap is None:
        props = torch.cuda.get_device_properties(index)
        cap = props.multi_processor_count * (props.max_threads_per_multi_processor // _BLOCK)
        _GEOMETRY[index] = cap
    blocks = min(cap, (numel + _BLOCK - 1) // _BLOCK)
    return _BLOCK * blocks


def consumed(numel, threads):
    """What one `normal_` of `numel` float32 elements adds to the generator's offset."""
    return 4 * ((numel - 1) // (threads * 4) + 1)


def reserve(numel, device):
    """The stream descriptor of the next `torch.empty(numel, device=device).normal_()` — and the generator
    advanced as if that call had been made."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    if not torch.cuda.default_generators:
        torch.cuda.init()
    generator = torch.cuda.default_generators[index]
    threads = launch_threads(numel, device)
    offset = generator.get_offset()
    stream = NoiseStream(generator.initial_seed(), offset, threads, numel)
    generator.set_offset(offset + consumed(numel, threads))
    return stream
