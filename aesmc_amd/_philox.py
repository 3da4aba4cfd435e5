"""PyTorch's own noise stream, addressed instead of materialised.

`torch.empty(shape).normal_()` on a HIP device — the `_standard_normal` call behind `Normal.rsample`
(aesmc/state.py:98) — is a pure function of the generator's (seed, offset), of the launch geometry ATen
picks from the device properties, and of the element index (csrc/philox_normal.hpp).  `reserve(numel, device)`
does the bookkeeping half of such a call WITHOUT launching it: it reads (seed, offset), advances the generator
by what `normal_` would have consumed, and returns the stream descriptor a kernel needs to form the same
values where it consumes them.  Everything drawn afterwards — by PyTorch or by this package — is unchanged.
`verified(device)` checks these assumptions against the installed PyTorch once per device (tests:
tests/test_gpu_noise_and_lazy_latents.py, the Philox tests).
"""
import collections
import contextlib
import contextvars

import torch

# `state`: None, or — inside a hipGraph capture — the device tensor [seed, offset] the captured launches read when
# they run; `offset` is then relative to it
NoiseStream = collections.namedtuple("NoiseStream", "seed offset threads numel state", defaults=(None,))

# how much of the generator was consumed through this module (eager): `reserved` by launches that draw inside
# kernels, `replaceable` by `state.sample`'s own torch draws, which a capture replaces by the fill kernel.  A
# GraphedLoss compares their sum over one warm-up evaluation with what the generator actually advanced by: equal
# means every draw of the captured region is one this module can place itself.
COUNTERS = {"reserved": 0, "replaceable": 0}

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
    """What one `normal_` of `numel` FLOAT32 elements adds to the generator's offset (float64 draws take another
    route in ATen: callers reserve noise for float32 tensors only, `state.sample` / `_kernels` check the dtype)."""
    return 4 * ((numel - 1) // (threads * 4) + 1)


_VERIFIED = {}


def verified(device):
    """One check per device, on first use: does the installed PyTorch / ROCm still draw `normal_` the way this module
    assumes (ATen's launch geometry, its offset accounting, rocRAND's Box-Muller)?  Two sizes (one trip, several
    trips) are drawn by PyTorch under a forked generator and formed again by aesmc_philox_normal_fill from the
    reservation `reserve` would have made; the values must be equal bit for bit and the generator must have advanced
    by `consumed`.  If not, noise is left to PyTorch for the rest of the process (`state.set_kernel_noise(False)`):
    slower, never wrong.  Inside a hipGraph capture nothing can be checked: `graphs.GraphedLoss` runs the check before it
    captures; a capture that reaches this unchecked gets False (PyTorch draws)."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    ok = _VERIFIED.get(index)
    if ok is None:
        if torch.cuda.is_current_stream_capturing():
            # nothing can be checked inside a capture, and an unchecked restatement must not be baked into a graph: the
            # noise is left to PyTorch for this capture (graphs.GraphedLoss runs the check itself before it captures, so
            # this is only reached by a caller's own torch.cuda.graph without a warm-up)
            return False
        _VERIFIED[index] = True          # (the check itself draws through this module's callers)
        ok = _self_check(torch.device("cuda", index))
        _VERIFIED[index] = ok
        if not ok:
            import warnings
            from . import state
            state.set_kernel_noise(False)
            warnings.warn("aesmc_amd: torch.empty(n).normal_() on {} does not match this build's restatement of "
                          "PyTorch's Philox stream (launch geometry, offset accounting or Box-Muller changed); noise "
                          "is drawn by PyTorch itself from here on".format(device), RuntimeWarning)
    return ok


def _self_check(device):
    from . import _kernels
    provider = _kernels.get()
    if provider.name != "hip":
        return True
    generator = _generator(device)
    with torch.random.fork_rng(devices=[device.index]):
        generator.manual_seed(0x5eed)
        for numel in (1000, 4 * launch_threads(1 << 30, device) * 2 + 17):
            before = generator.get_offset()
            want = torch.empty(numel, dtype=torch.float32, device=device).normal_()
            threads = launch_threads(numel, device)
            if generator.get_offset() - before != consumed(numel, threads):
                return False
            got = provider.philox_normal(NoiseStream(generator.initial_seed(), before, threads, numel), (numel,), device)
            if not torch.equal(got, want):
                return False
    return True


def _generator(device):
    index = device.index if device.index is not None else torch.cuda.current_device()
    if not torch.cuda.default_generators:
        torch.cuda.init()
    return torch.cuda.default_generators[index]


def reserve(numel, device):
    """The stream descriptor of the next `torch.empty(numel, device=device).normal_()` — and the generator
    advanced as if that call had been made.  Inside a hipGraph capture under `graph_noise_scope` the descriptor
    points at the scope's device-resident generator state instead (see GraphNoise)."""
    threads = launch_threads(numel, device)
    used = consumed(numel, threads)
    scope = _GRAPH_NOISE.get()
    if scope is not None and torch.cuda.is_current_stream_capturing():
        stream = NoiseStream(0, scope.consumed, threads, numel, scope.state)
        scope.consumed += used
        return stream
    generator = _generator(device)
    offset = generator.get_offset()
    stream = NoiseStream(generator.initial_seed(), offset, threads, numel)
    generator.set_offset(offset + used)
    COUNTERS["reserved"] += used
    return stream


class GraphNoise:
    """The generator state of one captured ELBO (aesmc_amd/graphs.py).  PyTorch hands its own captured kernels
    (seed, offset) through device memory it refreshes before every replay; the launches of this package that draw
    inside kernels do the same with a buffer of their own: `upload()` before a replay writes the generator's
    current (seed, offset), the captured launches add the offsets they were captured with (`consumed` so far in
    the region), and `advance()` afterwards moves the generator by the region's total — exactly what the eager
    evaluation consumes, in the same order."""

    SLOTS = 4

    def __init__(self, device):
        self.device = device
        # The host runs ahead of the device (replays are enqueued without a sync): every upload writes a pinned slot of
        # its own and is followed by an event; a slot is rewritten only after the copy that last read it has completed.
        # (One slot rewritten per replay — round 3 — let replay n read the offset of replay n + 1: two optimiser steps
        # then drew the same noise.)
        self.host = torch.zeros((self.SLOTS, 2), dtype=torch.int64, pin_memory=True)
        self.copied = [None] * self.SLOTS
        self.turn = 0
        self.state = torch.zeros(2, dtype=torch.int64, device=device)
        self.consumed = 0

    def upload(self):
        generator = _generator(self.device)
        slot = self.turn
        self.turn = (self.turn + 1) % self.SLOTS
        if self.copied[slot] is not None:
            self.copied[slot].synchronize()        # SLOTS uploads ago: long done unless the host is that far ahead
        self.host[slot, 0] = _as_int64(generator.initial_seed())
        self.host[slot, 1] = generator.get_offset()
        self.state.copy_(self.host[slot], non_blocking=True)
        event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(self.device))
        self.copied[slot] = event

    def advance(self):
        generator = _generator(self.device)
        generator.set_offset(generator.get_offset() + self.consumed)


def _as_int64(value):
    return value - (1 << 64) if value >= (1 << 63) else value


_GRAPH_NOISE = contextvars.ContextVar("aesmc_amd_graph_noise", default=None)


def graph_noise():
    return _GRAPH_NOISE.get()


@contextlib.contextmanager
def graph_noise_scope(scope):
    token = _GRAPH_NOISE.set(scope)
    try:
        yield scope
    finally:
        _GRAPH_NOISE.reset(token)
