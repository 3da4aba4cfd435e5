"""Particle tensors whose values are formed only if something reads them.

The reference's inner loop (aesmc/inference.py:102-126) materialises, per timestep, the re-indexed history
(`state.resample` = `torch.gather`, state.py:179), the proposal's draw (`state.sample` = `rsample`, state.py:98) and
every location `x @ W.t() + c` its callables hand to `Normal(...)` (the reference's own model:
test/models/lgssm.py:40, :52, :66-77) — `[B,K,d]` tensors written only to be read back by the next operation.  A
Markov model's callables do not look at those VALUES: they describe distributions in terms of them.  The three
classes here are what `infer` hands out instead — `torch.Tensor`s (so `torch.is_tensor`, `isinstance`, shape /
dtype / device behave) that hold no values:

  LazyResampled(source, index)       previous_latents[-1]: x_{t-1} re-indexed by the newest ancestors
  LazyDraw(location terms, noise)    latents[-1]: the proposal's reparameterised draw  loc_q + s_q * eps
  LazyInitialDraw(proposal, noise)   latents[-1] of the FIRST step: a BATCH_EXPANDED Normal's transposed draw
  LazyAffine(source, weight, offset) what `x @ W.t()`, `F.linear(x, W, b)`, `+ offset`, `scalar * x` of one of the
                                     above evaluate to: a location  offset + source @ weight.T

`__torch_function__` recognises exactly those affine expressions (and the bookkeeping `torch.distributions.Normal`
does with its arguments: `broadcast_tensors` to the shape they already have, the `loc == loc` NaN test of argument
validation) and records them; EVERY other use materialises first — the gather (K3), the draw (K9), the location (K8),
each differentiable, each once — and then runs on the real tensor.  So nothing a model can do gives numbers other
than the reference's eager evaluation; a model written in the reference's own style, `Normal(x @ W.t() + c, s)`,
reaches the fused kernels (K16 / K15 / K14) without being edited, and one that needs the values pays for them.
"""
import torch

_MAX_DIM = 16      # the item kernels' extent limit (aesmc_affine_max_dim): `scalar * x` is recorded as a map up to here
_MAX_MAP = 256     # the matrix-core step's (aesmc_affine_wide_max_dim): `x @ W.t()` / `F.linear` are recorded up to here


class LazyParticles(torch.Tensor):
    """Common machinery: a wrapper tensor (no storage) answering shape-like questions from what it stands for."""

    @staticmethod
    def _make(cls, shape, like):
        return torch.Tensor._make_wrapper_subclass(cls, shape, dtype=like.dtype, device=like.device,
                                                   requires_grad=False)

    # ---- answered without values -------------------------------------------------------------------
    shape = property(lambda self: self._lazy_shape)
    dtype = property(lambda self: self._lazy_like.dtype)
    device = property(lambda self: self._lazy_like.device)
    is_cuda = property(lambda self: self._lazy_like.is_cuda)
    ndim = property(lambda self: len(self._lazy_shape))
    grad_fn = property(lambda self: None if self._lazy_real is None else self._lazy_real.grad_fn)

    @property
    def requires_grad(self):
        if self._lazy_real is not None:
            return self._lazy_real.requires_grad
        return self._lazy_grad_mode and self._lazy_requires_grad()      # made under no_grad: no history, as eager

    def size(self, dim=None):
        return self._lazy_shape if dim is None else self._lazy_shape[dim]

    def dim(self):
        return len(self._lazy_shape)

    def numel(self):
        count = 1
        for extent in self._lazy_shape:
            count *= extent
        return count

    def element_size(self):
        return self._lazy_like.element_size()

    def is_floating_point(self):
        return self._lazy_like.is_floating_point()

    def __repr__(self):
        return "{}({}, {}, {})".format(type(self).__name__, tuple(self._lazy_shape), self._lazy_like.dtype,
                                       "pending" if self._lazy_real is None else "materialised")

    # ---- values ------------------------------------------------------------------------------------
    @property
    def is_pending(self):
        return self._lazy_real is None

    def materialise(self):
        if self._lazy_real is None:
            # The values are kept for every later reader.  A first reader under torch.no_grad() (a logging or
            # diagnostic look at latents[-1] inside a callable) must not leave them without their history: the
            # reference's eager tensors carry it whoever looks first.
            # (... only if it was MADE with autograd on: one made under torch.no_grad() — a pure evaluation — stands
            #  for a tensor that never had a history, and must not start retaining one for every later step)
            if not torch.is_grad_enabled() and self._lazy_grad_mode and self._lazy_requires_grad():
                with torch.enable_grad():
                    self._lazy_real = self._lazy_compute()
            else:
                self._lazy_real = self._lazy_compute()
        return self._lazy_real

    def resolve(self, real):
        """A launch has formed the values elsewhere (the fused step): later readers get that tensor."""
        self._lazy_real = real

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        recorded = _record(func, args, kwargs)
        if recorded is not NotImplemented:
            return recorded
        with torch._C.DisableTorchFunctionSubclass():
            return func(*_unwrap(args), **{key: _unwrap(value) for key, value in kwargs.items()})

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # not reached through the Python API (__torch_function__ has unwrapped by then); C++ callers that
        # bypass it get the same treatment: the values, then the operator on the real tensor
        return func(*_unwrap(args), **{key: _unwrap(value) for key, value in (kwargs or {}).items()})


class LazyResampled(LazyParticles):
    """x_{t-1}[b, index[b,k], ...] — not gathered until read (`pending` = (source, index) for the fused launches)."""

    @staticmethod
    def __new__(cls, source, index):
        return LazyParticles._make(cls, source.shape, source)

    def __init__(self, source, index):
        self._lazy_grad_mode = torch.is_grad_enabled()
        self._lazy_like = source
        self._lazy_source = source
        self._lazy_index = index
        self._lazy_real = None
        self._lazy_shape = source.shape

    def _lazy_requires_grad(self):
        return self._lazy_source.requires_grad

    @property
    def pending(self):
        """(source, index) while nothing has read the values; None afterwards."""
        return None if self._lazy_real is not None else (self._lazy_source, self._lazy_index)

    def _lazy_compute(self):
        from . import state
        return state.resample(self._lazy_source, self._lazy_index)


class LazyAffine(LazyParticles):
    """offset + source @ weight.T for a [B,K,din] particle tensor `source` (itself possibly lazy), weight [dout,din]
    (any 2-D view), offset None, [dout] or [B,dout]: a location nobody has evaluated."""

    @staticmethod
    def __new__(cls, source, weight, offset=None):
        return LazyParticles._make(cls, torch.Size(tuple(source.shape[:-1]) + (weight.size(0),)), weight)

    def __init__(self, source, weight, offset=None):
        self._lazy_grad_mode = torch.is_grad_enabled()
        self._lazy_like = weight
        self.source, self.weight, self.offset = source, weight, offset
        self._lazy_real = None
        self._lazy_shape = torch.Size(tuple(source.shape[:-1]) + (weight.size(0),))

    def _lazy_requires_grad(self):
        return any(t is not None and t.requires_grad for t in (self.source, self.weight, self.offset))

    def _lazy_compute(self):
        from .linear_gaussian import particle_affine
        return particle_affine(real(self.source), self.weight, self.offset)


class LazyDraw(LazyParticles):
    """The reparameterised draw  (offset + source @ weight.T) + scale * eps  of a linear-Gaussian proposal, its
    noise `eps` a tensor or only RESERVED in PyTorch's generator (`_philox.NoiseStream`): formed by the launch that
    weighs the step (K16 / K15) or, the moment anything reads it, by K9."""

    @staticmethod
    def __new__(cls, terms, noise):
        source, weight = terms.source, terms.weight
        return LazyParticles._make(cls, torch.Size(tuple(source.shape[:-1]) + (weight.size(0),)), weight)

    def __init__(self, terms, noise):
        self._lazy_grad_mode = torch.is_grad_enabled()
        self._lazy_like = terms.weight
        self.terms = terms              # the proposal's (source, weight, offset, scale_param)
        self.noise = noise
        self._lazy_real = None
        self._lazy_shape = torch.Size(tuple(terms.source.shape[:-1]) + (terms.weight.size(0),))

    def _lazy_requires_grad(self):
        terms = self.terms
        return any(t is not None and t.requires_grad for t in (terms.source, terms.weight, terms.offset,
                                                               terms.scale_param))

    def _lazy_compute(self):
        from . import _ops, state
        terms = self.terms
        eps = state._noise_tensor(self.noise, self)
        if getattr(self, "wide", False):
            # rows too wide for K9 (a draw deferred for K17 that no launch formed after all): the location as
            # `particle_affine` evaluates it, then K6 — what `Normal(loc, scale).rsample` does with the same noise
            from .linear_gaussian import particle_affine
            loc = particle_affine(real(terms.source), terms.weight, terms.offset)
            return _ops.normal_rsample(eps, loc, terms.scale_param.expand_as(loc))
        return _ops.affine_rsample(real(terms.source), terms.weight, terms.offset, terms.scale_param, eps)


class LazyInitialDraw(LazyParticles):
    """The FIRST timestep's draw of a BATCH_EXPANDED Normal proposal, `rsample((K,))` transposed to [B,K,d]
    (aesmc/state.py:98, :102-103):  loc[b] + eps[k,b] * scale[b]  with its noise `eps` [K,B,d] already drawn.  Formed by
    the launch that weighs the step when that step is linear-Gaussian in it (K20, `state._initial_step`) or, the moment
    anything reads it, by K6 — the transposed draw `state.sample` used to make right away."""

    @staticmethod
    def __new__(cls, distribution, loc, scale, noise):
        return LazyParticles._make(cls, torch.Size((noise.size(1), noise.size(0)) + tuple(noise.shape[2:])), noise)

    def __init__(self, distribution, loc, scale, noise):
        self._lazy_grad_mode = torch.is_grad_enabled()
        self._lazy_like = noise
        self.distribution = distribution      # the proposal this is the draw of
        self.loc, self.scale = loc, scale     # expanded to the noise's shape [K,B,d]
        self.noise = noise
        self._lazy_real = None
        self._lazy_shape = torch.Size((noise.size(1), noise.size(0)) + tuple(noise.shape[2:]))

    def _lazy_requires_grad(self):
        return self.loc.requires_grad or self.scale.requires_grad

    def _lazy_compute(self):
        from . import _ops
        return _ops.normal_rsample(self.noise.transpose(0, 1), self.loc.transpose(0, 1), self.scale.transpose(0, 1))


# ---- what is recorded instead of computed ----------------------------------------------------------------------
def _is_particles(value):
    """A pending lazy [B,K,d] particle tensor an affine map may be recorded on (not a location: maps of maps are
    evaluated)."""
    return type(value) in (LazyResampled, LazyDraw, LazyInitialDraw) and value._lazy_real is None and \
        len(value._lazy_shape) == 3


def _plain(value):
    return isinstance(value, torch.Tensor) and not isinstance(value, LazyParticles)


def _small_map(weight, din):
    """A map some fused launch may take: at most 16 x 16 (the item kernels), or rows of 17 .. 256 values (the matrix-core
    step, since round 6) — anything a launch then declines is evaluated by `particle_affine` (K8, or the library's GEMM
    above 16 x 16) the moment the location's values are read: the eager numbers either way."""
    return _plain(weight) and weight.dim() == 2 and weight.size(1) == din and \
        1 <= weight.size(0) <= _MAX_MAP and 1 <= din <= _MAX_MAP


def _as_offset(affine, other):
    """`other` as an offset of the location `affine` ([dout] or [B,dout]) if `affine + other` is one, else None."""
    if not _plain(other) or other.dtype != affine.dtype or other.device != affine.device:
        return None
    batch, _, dout = affine._lazy_shape
    shape = tuple(other.shape)
    if shape in ((dout,), (1, dout), (1, 1, dout)):
        return other.reshape(dout)
    if shape == (batch, 1, dout):
        return other[:, 0]
    return None


def reaches_by_views(view, base, hops):
    """Does autograd take `view`'s gradient to `base` through exactly `hops` single-input view nodes (or is there no
    autograd between them at all)?  A view cut off from its base (made under no_grad, or of a detached alias) fails."""
    if not base.requires_grad and not view.requires_grad:
        return True
    if base.requires_grad != view.requires_grad:
        return False
    fn = view.grad_fn
    for _ in range(hops):
        if fn is None or len(fn.next_functions) != 1:
            return False
        fn = fn.next_functions[0][0]
    if fn is None:
        return False
    return fn is base.grad_fn if base.grad_fn is not None else getattr(fn, "variable", None) is base


def _own_tensor(view, hops=2):
    """The tensor a view is ALL of, when it is: `x @ W.t()` hands this module `W.t()`, a new view object per call, and
    `W.t().t()` is W itself — the same storage, geometry and autograd identity (the two transposes' backward is the
    identity).  The model's own parameter object is what the per-step caches and the chained weight gradients
    (`_ops.StepLink`) recognise from one timestep to the next."""
    base = view._base
    if base is not None and type(base) in (torch.Tensor, torch.nn.Parameter) and base.shape == view.shape and \
            base.stride() == view.stride() and base.storage_offset() == view.storage_offset() and \
            base.dtype == view.dtype and reaches_by_views(view, base, hops):
        return base
    return view


_EYES = {}


def _scaled_identity(scalar, dim, like):
    key = (dim, like.dtype, like.device)
    eye = _EYES.get(key)
    if eye is None:
        eye = _EYES[key] = torch.eye(dim, dtype=like.dtype, device=like.device)
    return scalar * eye


def _record(func, args, kwargs):
    """The lazy result of `func(*args)` when it is one of the affine expressions (or the bookkeeping) this module
    understands; NotImplemented sends the caller to materialise."""
    name = getattr(func, "__name__", "")
    if kwargs and name not in ("linear",):
        return NotImplemented
    if name in ("matmul", "__matmul__") and len(args) == 2:
        x, w = args
        if _is_particles(x) and _plain(w) and w.dim() == 2 and w.dtype == x.dtype and w.device == x.device:
            weight = _own_tensor(w.t())
            if _small_map(weight, x._lazy_shape[2]):
                return LazyAffine(x, weight)
        return NotImplemented
    if name == "linear":
        x = args[0] if args else kwargs.get("input")
        w = args[1] if len(args) > 1 else kwargs.get("weight")
        b = args[2] if len(args) > 2 else kwargs.get("bias")
        if _is_particles(x) and _small_map(w, x._lazy_shape[2]) and w.dtype == x.dtype and w.device == x.device and \
                (b is None or (_plain(b) and tuple(b.shape) == (w.size(0),) and b.dtype == x.dtype)):
            return LazyAffine(x, w, b)
        return NotImplemented
    if name in ("add", "__add__", "__radd__") and len(args) == 2:
        a, b = args if type(args[0]) is LazyAffine else (args[1], args[0])
        if type(a) is LazyAffine and a._lazy_real is None:
            offset = _as_offset(a, b)
            if offset is not None:
                if a.offset is not None:
                    held = a.offset
                    offset = (held.unsqueeze(0) if held.dim() < offset.dim() else held) + \
                        (offset.unsqueeze(0) if offset.dim() < held.dim() else offset)
                return LazyAffine(a.source, a.weight, offset)
        return NotImplemented
    if name in ("mul", "__mul__", "__rmul__") and len(args) == 2:
        x, s = args if isinstance(args[0], LazyParticles) else (args[1], args[0])
        if _is_particles(x) and x._lazy_shape[2] <= _MAX_DIM and \
                (isinstance(s, (int, float)) or (_plain(s) and s.dim() == 0 and s.device == x.device)):
            if isinstance(s, (int, float)):
                s = torch.as_tensor(s, dtype=x.dtype, device=x.device)
            return LazyAffine(x, _scaled_identity(s.to(x.dtype), x._lazy_shape[2], x._lazy_like))
        return NotImplemented
    if name == "broadcast_tensors":
        tensors = args[0] if len(args) == 1 and isinstance(args[0], (list, tuple)) else args
        shape = None
        for t in tensors:
            if isinstance(t, LazyParticles) and t._lazy_real is None:
                if shape is not None and tuple(t._lazy_shape) != shape:
                    return NotImplemented
                shape = tuple(t._lazy_shape)
        if shape is None:
            return NotImplemented
        out = []
        for t in tensors:
            if isinstance(t, LazyParticles):
                out.append(t if t._lazy_real is None else t._lazy_real.expand(shape))
            else:
                try:
                    out.append(t.expand(shape))
                except RuntimeError:
                    return NotImplemented
        return tuple(out)
    if name in ("eq", "__eq__") and len(args) == 2 and args[0] is args[1] and isinstance(args[0], LazyAffine) and \
            args[0]._lazy_real is None:
        # `constraints.real.check(loc)` of Distribution argument validation: a NaN location is caught where its
        # log-weights are (the resampler's NaN flag -> FloatingPointError), without evaluating it here
        return torch.ones((), dtype=torch.bool)
    if name in ("expand", "expand_as") and isinstance(args[0], LazyParticles) and args[0]._lazy_real is None:
        target = args[1:] if name == "expand" else (tuple(args[1].shape),)
        target = tuple(target[0]) if len(target) == 1 and isinstance(target[0], (tuple, list, torch.Size)) else tuple(target)
        if target == tuple(args[0]._lazy_shape):
            return args[0]
    return NotImplemented


def _unwrap(value):
    if isinstance(value, LazyParticles):
        return value.materialise()
    if isinstance(value, (list, tuple)):
        return type(value)(_unwrap(item) for item in value)
    return value


def real(tensor):
    """The tensor itself; a lazy one evaluated."""
    return tensor.materialise() if isinstance(tensor, LazyParticles) else tensor
