"""Linear-Gaussian model terms whose location never goes through HBM (kernels K8 / K9 / K10).

A state-space model written against the reference's callable contract (aesmc/inference.py:20-46)
returns, per timestep, `Normal(loc, scale)` objects whose `loc` is a small linear map of the particles —
the reference's own test model does (test/models/lgssm.py:40 `mult * previous_latents[-1]`, :52, :66-77
a Linear layer of [x_{t-1}, y_t]).  Through PyTorch each such location is a `[B*K, d] x [d, d]` matmul
that writes `[B,K,d]` to HBM only for `state.sample` / `state.log_prob` to read it back: at B=1024,
K=4096, d=10 those matmuls and re-reads are more than half of a step's device time.

`AffineNormal(source, weight, scale, offset)` IS a `torch.distributions.Normal` — same `loc`, `scale`,
`rsample`, `log_prob`, `batch_shape` — with

    loc = source @ weight.T + offset            source [B,K,din], weight [dout,din], offset [dout] | [B,dout]

evaluated lazily.  The callables return it in place of `Normal(source @ weight.T + offset, scale)`;
nothing else in the contract changes.  `state.sample` then draws with kernel K9 (location + noise in
one pass) and `infer` weighs a step whose transition, emission and proposal are all AffineNormal with
kernel K10 (three locations and the log-weight from x_{t-1} and x_t alone); any other consumer reads
`.loc`, which kernel K8 materialises once.  Every location element is one chain of fused multiply-adds
in a fixed order — the same chain in K8, K9 and K10, so a draw is the same bit for bit whichever
route produced it; K10's log-density divides once per term where PyTorch (and K5) divide per element,
so its log-weights agree with the materialised route to rounding (float64: ~1e-15 relative).  Maps
wider than 16 are proper GEMMs and go to the library (`torch.matmul`).

`particle_affine(x, weight, offset)` is K8 as a differentiable operator for models that need the
location itself (e.g. `tanh(particle_affine(x, A))` of a nonlinear transition).
"""
import torch
from torch.distributions import constraints

from . import _kernels
from . import _ops


def particle_affine(x, weight, offset=None):
    """x @ weight.T + offset (offset [dout], or [B, dout] broadcast over particles) for particles
    x [B,K,din]; differentiable in all three.  Kernel K8 when the map is at most 16 x 16, the
    library's matmul otherwise."""
    provider = _kernels.get()
    if provider.affine_covers(x, weight, offset):
        return _ops.particle_affine(x, weight, offset)
    if provider.name == "hip" and not (torch.is_tensor(x) and x.is_cuda):
        raise RuntimeError("aesmc_amd: particle_affine operand lives on '{}'; this package computes only on a "
                           "HIP device (MI355X) and has no CPU fallback.".format(getattr(x, "device", None)))
    out = torch.matmul(x, weight.t())      # wider than 16 x 16 (or not [B,K,d]): the library's GEMM, on the device
    if offset is not None:
        out = out + (offset.unsqueeze(1) if offset.dim() == 2 else offset)
    return out


class AffineNormal(torch.distributions.Normal):
    """Normal(loc = source @ weight.T + offset, scale) with the location evaluated on demand.

    source: [..., din] (the fused kernels take [batch_size, num_particles, din]);
    weight: [dout, din]; offset: None, [dout], or [batch_size, dout] (one row per batch element,
    shared by its particles — e.g. the observation's part of a proposal's mean);
    scale: tensor (or Python number) broadcastable to [..., dout]; the fused kernels take one value.
    defer_draw: for a PROPOSAL whose model is linear-Gaussian throughout.  `infer` draws x_t from it
        (aesmc/inference.py:106) and only then asks the transition and emission callables for their
        distributions of x_t; when those are AffineNormals too, nothing needs x_t's values before the step
        is weighed, and with `defer_draw=True` the draw is left to that launch (kernel K15: K9 and K10 in
        one pass over x_{t-1} and the noise — the same bits as the two).  The noise is drawn where
        `rsample` would draw it, so the RNG stream is unchanged; `infer` fills the values with K9 instead
        whenever the step turns out not to be weighed that way.  The promise the model makes by setting
        it: its transition and emission callables do not READ the values of the newest latent they are
        handed (building an AffineNormal on it does not).  With argument validation on (PyTorch's
        default) the latent holds NaN until its values exist, so a callable that reads it anyway fails
        at once (a NaN parameter, or the NaN log-weight check of `infer`); `validate_args=False` skips
        that fill.
    """

    arg_constraints = {"scale": constraints.positive}

    def __init__(self, source, weight, scale, offset=None, validate_args=None, defer_draw=False):
        if not (torch.is_tensor(source) and torch.is_tensor(weight)):
            raise TypeError("AffineNormal: source and weight must be tensors")
        if weight.dim() != 2 or source.dim() < 1 or source.size(-1) != weight.size(1):
            raise ValueError("AffineNormal: weight {} does not map source {}".format(
                tuple(weight.shape), tuple(source.shape)))
        dout = weight.size(0)
        if offset is not None:
            ok = tuple(offset.shape) == (dout,) or (
                source.dim() == 3 and tuple(offset.shape) == (source.size(0), dout))
            if not ok:
                raise ValueError("AffineNormal: offset must be [{0}] or [batch_size, {0}], got {1}".format(
                    dout, tuple(offset.shape)))
        if not torch.is_tensor(scale):
            scale = torch.as_tensor(scale, dtype=source.dtype, device=source.device)
        batch_shape = torch.Size(tuple(source.shape[:-1]) + (dout,))
        # (checked by hand: torch.broadcast_shapes costs ~20 us of host time per distribution)
        fits = scale.dim() <= len(batch_shape) and all(
            have == 1 or have == want for have, want in zip(reversed(scale.shape), reversed(batch_shape)))
        if not fits:
            raise ValueError("AffineNormal: scale {} does not broadcast to {}".format(
                tuple(scale.shape), tuple(batch_shape)))
        self.source, self.weight, self.offset = source, weight, offset
        self.scale_param = scale
        self.defer_draw = bool(defer_draw)
        self._loc = None
        torch.distributions.Distribution.__init__(self, batch_shape, validate_args=validate_args)

    @property
    def loc(self):
        if self._loc is None:
            self._loc = particle_affine(self.source, self.weight, self.offset)
        return self._loc

    @property
    def scale(self):
        scale = self.scale_param
        return scale if scale.shape == self._batch_shape else scale.expand(self._batch_shape)

    def expand(self, batch_shape, _instance=None):
        batch_shape = torch.Size(batch_shape)
        return torch.distributions.Normal(self.loc.expand(batch_shape), self.scale.expand(batch_shape),
                                          validate_args=False)

    def __repr__(self):
        return "AffineNormal(source: {}, weight: {}, offset: {}, scale: {})".format(
            tuple(self.source.shape), tuple(self.weight.shape),
            None if self.offset is None else tuple(self.offset.shape), tuple(self.scale_param.shape))
