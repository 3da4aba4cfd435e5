"""Record / replay of the two random streams an `infer` call consumes, so that runs on different
devices (or under different implementations) see identical draws:

  * standard-normal noise of `Normal.rsample` (torch.distributions.normal._standard_normal);
  * the per-resample uniforms (np.random.uniform, numpy's global RandomState).

Recording wraps the real generators; replaying substitutes stored arrays (moved to the device the
caller asks for).  Golden fixtures under tests/golden/ were captured with `record()` around the
reference and are consumed with `replay()` around this package.
"""
import contextlib

import numpy as np
import torch
import torch.distributions.normal as _normal_module


class Tape:
    def __init__(self, normals=None, uniforms=None):
        self.normals = [] if normals is None else list(normals)
        self.uniforms = [] if uniforms is None else list(uniforms)


@contextlib.contextmanager
def record():
    """Yields a Tape that fills with every normal / uniform block drawn inside the context."""
    tape = Tape()
    real_normal, real_uniform = _normal_module._standard_normal, np.random.uniform

    def spy_normal(shape, dtype, device):
        draw = real_normal(shape, dtype=dtype, device=device)
        tape.normals.append(draw.detach().cpu().numpy().copy())
        return draw

    def spy_uniform(*args, **kwargs):
        draw = real_uniform(*args, **kwargs)
        tape.uniforms.append(np.array(draw, copy=True))
        return draw

    _normal_module._standard_normal, np.random.uniform = spy_normal, spy_uniform
    try:
        yield tape
    finally:
        _normal_module._standard_normal, np.random.uniform = real_normal, real_uniform


@contextlib.contextmanager
def replay(tape):
    """Feeds the tape's blocks back in order; shapes are checked so a divergence in RNG
    consumption fails loudly instead of silently shifting the stream."""
    normals, uniforms = iter(tape.normals), iter(tape.uniforms)
    real_normal, real_uniform = _normal_module._standard_normal, np.random.uniform

    def fake_normal(shape, dtype, device):
        block = next(normals)
        if tuple(block.shape) != tuple(shape):
            raise AssertionError("normal replay shape mismatch: tape {} vs request {}".format(
                block.shape, tuple(shape)))
        return torch.as_tensor(block).to(device=device, dtype=dtype)

    def fake_uniform(low=0.0, high=1.0, size=None):
        block = next(uniforms)
        want = () if size is None else tuple(np.atleast_1d(size))
        if tuple(block.shape) != want:
            raise AssertionError("uniform replay shape mismatch: tape {} vs request {}".format(
                block.shape, want))
        return np.array(block, copy=True)

    _normal_module._standard_normal, np.random.uniform = fake_normal, fake_uniform
    try:
        yield
    finally:
        _normal_module._standard_normal, np.random.uniform = real_normal, real_uniform


class StaticReplay:
    """A tape fed through a CAPTURED evaluation (graphs.GraphedLoss): a hipGraph re-issues recorded device work, so the
    noise it reads cannot come from a host call per draw.  Inside a capture every `_standard_normal` request is answered
    with a static device buffer (one per request, in request order); `load()` then copies the tape's next blocks into
    those buffers before a replay.  Outside a capture: while `armed`, requests pop the tape as `replay()` does (data
    generation between replays); while not armed (the warm-up evaluations a capture makes) they are answered by the real
    generators and the tape stays where it is.  Uniforms: popped from the tape while armed (the graph's uniform feed
    draws them on the host before each replay), real otherwise."""

    def __init__(self, tape):
        self.normals, self.uniforms = iter(tape.normals), iter(tape.uniforms)
        self.slots = []
        self.armed = False
        self._real = None

    def _normal(self, shape, dtype, device):
        device = torch.device(device)
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            slot = torch.empty(tuple(shape), dtype=dtype, device=device)
            self.slots.append(slot)
            return slot
        if not self.armed:
            return self._real[0](shape, dtype=dtype, device=device)
        block = next(self.normals)
        if tuple(block.shape) != tuple(shape):
            raise AssertionError("normal replay shape mismatch: tape {} vs request {}".format(block.shape, tuple(shape)))
        return torch.as_tensor(block).to(device=device, dtype=dtype)

    def _uniform(self, low=0.0, high=1.0, size=None):
        if not self.armed:
            return self._real[1](low, high, size)
        block = next(self.uniforms)
        want = () if size is None else tuple(np.atleast_1d(size))
        if tuple(block.shape) != want:
            raise AssertionError("uniform replay shape mismatch: tape {} vs request {}".format(block.shape, want))
        return np.array(block, copy=True)

    def load(self):
        """The tape's next blocks into the captured evaluation's noise buffers, in the order it requested them."""
        for slot in self.slots:
            block = next(self.normals)
            if tuple(block.shape) != tuple(slot.shape):
                raise AssertionError("normal replay shape mismatch: tape {} vs captured request {}".format(
                    block.shape, tuple(slot.shape)))
            slot.copy_(torch.as_tensor(block).to(device=slot.device, dtype=slot.dtype))

    def __enter__(self):
        self._real = (_normal_module._standard_normal, np.random.uniform)
        _normal_module._standard_normal, np.random.uniform = self._normal, self._uniform
        return self

    def __exit__(self, *exc):
        _normal_module._standard_normal, np.random.uniform = self._real
        return False
