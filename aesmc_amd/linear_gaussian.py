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
from . import _lazy
from . import _ops


def particle_affine(x, weight, offset=None, activation=None):
    """x @ weight.T + offset (offset [dout], or [B, dout] broadcast over particles) for particles
    x [B,K,din]; differentiable in all three.  Kernel K8 when the map is at most 16 x 16, the
    library's matmul otherwise.  `activation="tanh"`: tanh of that — from the same launch where K8 applies (the bits
    `torch.tanh(particle_affine(x, weight, offset))` holds, without the element-wise pass behind it): the nonlinear
    transition `tanh(A x_{t-1})` of BASELINE.json configs[3]."""
    if activation not in (None, "tanh"):
        raise ValueError("aesmc_amd: particle_affine activation must be None or 'tanh', got {!r}".format(activation))
    provider = _kernels.get()
    if provider.affine_covers(x, weight, offset):
        return _ops.particle_affine(x, weight, offset, through_tanh=activation == "tanh")
    if provider.name == "hip" and not (torch.is_tensor(x) and x.is_cuda):
        raise RuntimeError("aesmc_amd: particle_affine operand lives on '{}'; this package computes only on a "
                           "HIP device (MI355X) and has no CPU fallback.".format(getattr(x, "device", None)))
    out = torch.matmul(x, weight.t())      # wider than 16 x 16 (or not [B,K,d]): the library's GEMM, on the device
    if offset is not None:
        out = out + (offset.unsqueeze(1) if offset.dim() == 2 else offset)
    return torch.tanh(out) if activation == "tanh" else out


class _Terms:
    """A linear-Gaussian term as the fused kernels take it: loc = offset + source @ weight.T, one scale value."""
    __slots__ = ("source", "weight", "offset", "scale_param", "_validate_args")

    def __init__(self, source, weight, offset, scale_param, validate_args):
        self.source, self.weight, self.offset, self.scale_param = source, weight, offset, scale_param
        self._validate_args = validate_args

    def same_terms(self, other):
        return other is not None and self.source is other.source and self.weight is other.weight and \
            self.offset is other.offset and self.scale_param is other.scale_param


def affine_terms(distribution):
    """The (source, weight, offset, scale_param) of a distribution that IS a linear-Gaussian term of the particles —
    an `AffineNormal`, or a plain `torch.distributions.Normal` whose location is an affine expression recorded on a
    lazy latent (`_lazy.LazyAffine`: the model wrote `Normal(x @ W.t() + c, s)` in the reference's own style,
    test/models/lgssm.py:40) and whose scale is one value — else None."""
    if type(distribution) is AffineNormal:
        return distribution
    if type(distribution) is not torch.distributions.Normal:
        return None
    cached = distribution.__dict__.get("_aesmc_terms")
    if cached is not None:
        return cached if cached.source is not None and distribution.loc.is_pending else None
    loc = distribution.__dict__.get("loc")
    if type(loc) is not _lazy.LazyAffine or not loc.is_pending:
        return None
    scale = distribution.__dict__.get("scale")
    if not torch.is_tensor(scale) or isinstance(scale, _lazy.LazyParticles):
        return None
    if scale.numel() != 1:
        if any(stride != 0 for stride in scale.stride()):      # one value, expanded by Normal's broadcast_all?
            return None
        # the one-value tensor that was expanded, where it is at hand (the model's own buffer or parameter: the same
        # object every timestep, which the per-step caches recognise; expand's backward is the sum the kernels form)
        base = scale._base
        if base is not None and base.numel() == 1 and base.dtype == scale.dtype and \
                type(base) in (torch.Tensor, torch.nn.Parameter) and _lazy.reaches_by_views(scale, base, 1):
            scale = base
        else:
            scale = scale[(0,) * scale.dim()]
    terms = _Terms(loc.source, loc.weight, loc.offset, scale, distribution._validate_args)
    distribution.__dict__["_aesmc_terms"] = terms
    return terms


def particle_mlp(x, weight1, offset1, weight2, bias2=None):
    """A learned proposal net over the particles: the two-layer tanh MLP
    bias2 + tanh(offset1 + x @ weight1.T) @ weight2.T  for x [B,K,din] (din <= 16), weight1 [H,din] (H <= 64),
    offset1 [H] or [B,H] (the per-row part of the first layer: its bias plus the observation's columns of the weight
    applied to y_t), weight2 [dout,H] (dout <= 16).  Kernel K13 forward — the hidden layer never leaves registers — and
    K13b backward (recomputed hidden layer, weight gradients on the matrix cores); the PyTorch expression wherever the
    kernels do not cover the operands (other extents, fewer than ~43 particles per batch row; backward: K not a multiple
    of 256).  What BASELINE.json's configs[3] (nonlinear SSM with a learned proposal net) evaluates per timestep;
    `torch.cat + nn.Linear + nn.Tanh + nn.Linear` of the same weights gives the same numbers to float rounding."""
    return _ops.particle_mlp(x, weight1, offset1, weight2, bias2)


class AffineNormal(torch.distributions.Normal):
    """Normal(loc = source @ weight.T + offset, scale) with the location evaluated on demand.

    source: [..., din] (the fused kernels take [batch_size, num_particles, din]);
    weight: [dout, din]; offset: None, [dout], or [batch_size, dout] (one row per batch element,
    shared by its particles — e.g. the observation's part of a proposal's mean);
    scale: tensor (or Python number) broadcastable to [..., dout]; the fused kernels take one value.
    defer_draw: None (default) / True — inside `infer`, a PROPOSAL's draw is not formed when `state.sample` is called
        (aesmc/inference.py:106) but left to the launch that weighs the step (K16 / K15: the draw, its noise and the
        log-weight in one pass): `state.sample` returns a `_lazy.LazyDraw`, a tensor without values.  The noise is
        drawn (or reserved in PyTorch's generator) where `rsample` would draw it, so the RNG stream is unchanged.
        Anything that READS the newest latent's values — a callable doing arithmetic on `latents[-1]` other than the
        affine maps the fused kernels evaluate themselves, a step that turns out not to be linear-Gaussian — gets
        the draw formed on the spot (K9), differentiably: no promise is asked of the model.  False — draw at once.
    """

    arg_constraints = {"scale": constraints.positive}

    def __init__(self, source, weight, scale, offset=None, validate_args=None, defer_draw=None):
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
        self.defer_draw = defer_draw
        self._loc = None
        torch.distributions.Distribution.__init__(self, batch_shape, validate_args=validate_args)

    def same_terms(self, other):
        return other is self or (other is not None and self.source is other.source and self.weight is other.weight and
                                 self.offset is other.offset and self.scale_param is other.scale_param)

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
