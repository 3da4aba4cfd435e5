"""`torch.distributions` without host round trips, for the duration of an `infer`.

A model written exactly as the reference writes its own (test/models/lgssm.py:27-29, :41, :61-72;
test/models/gaussian.py:14-24) builds `Normal(mult * x, 0.5)` per timestep with PyTorch's default
`validate_args`.  On a HIP device that costs, per distribution:

  * `broadcast_all` turns the Python number into `torch.tensor(0.5, device=...)` — a pageable host-to-device copy
    (a stream synchronisation, and something a hipGraph capture refuses);
  * `Distribution.__init__` checks every parameter with `constraint.check(value).all()` read on the HOST, and
    `log_prob` does the same for the value (`_validate_sample`) — one device synchronisation each.

`scope()` — entered by `inference.infer`, hence around every call of the four callables — replaces those three with
forms that never leave the device:

  * numbers become cached 0-dim device constants (`constant`: one fill the first time a (value, dtype, device) is met,
    the same tensor object ever after — which is also what lets the fused linear-Gaussian route recognise "the same
    scale as last timestep");
  * parameter and value checks run where the data are and OR a bit into the device status word
    (`FLAG_INVALID_PARAMETER` / `FLAG_VALUE_OUTSIDE_SUPPORT`), raised as the reference's ValueError at the end of
    `infer` — the library's existing deferred-validation mode (`state.set_validation_mode`); 'eager' restores
    PyTorch's own checks, syncs included.

A fourth wrapper rides along: `torch.nn.functional.linear` (what `nn.Linear.forward` calls).  The reference's own
proposal is `Linear(2, 1)` over all B K particles (test/models/lgssm.py:61-72): a [B K, 2] x [2, 1] product, for which
the GEMM library picks tiles made for square problems (forward 18 us, the weight gradient — a 262 144-long contraction
onto one workgroup — 747 us at configs[1]'s B K).  Inside a scope such a map (at most 16 x 16, at least 2^14 rows, on a HIP
device) is kernel K8 (`_ops.particle_affine`: one fma chain per output, inputs ascending, started from the bias) with K11
as its backward (`settings.particle_linear`; the numbers differ from the library's by float rounding of another summation
order, as any two GEMM kernels' do).

The wrappers are installed once (first scope) and delegate to PyTorch's originals whenever no scope is active in the
calling context, so code outside `infer` sees stock `torch.distributions`.
"""
import contextlib
import contextvars
import importlib
import pkgutil
import threading

import torch
from torch.distributions import constraints
from torch.distributions.distribution import Distribution
from torch.distributions.utils import lazy_property

from . import settings

_ACTIVE = contextvars.ContextVar("aesmc_amd_syncfree", default=False)
_INSTALL_LOCK = threading.Lock()
_INSTALLED = False
_TORCH_BROADCAST_ALL = torch.distributions.utils.broadcast_all
_TORCH_INIT = Distribution.__init__
_TORCH_VALIDATE_SAMPLE = Distribution._validate_sample
_TORCH_LINEAR = torch.nn.functional.linear
_PARTICLE_LINEAR_MIN_ROWS = 1 << 14

_CONSTANTS = {}
_CONSTANTS_LIMIT = 4096
_NUMBER = (int, float, bool)
# what a deferred parameter check was about, for the message of the ValueError raised later (newest last)
_CHECKED = []
_CHECKED_LIMIT = 8


def constant(value, dtype, device):
    """The Python number `value` as a 0-dim tensor of `dtype` on `device`, rounded as `torch.tensor(value, dtype=...)`
    rounds it.  Made once per (value, dtype, device) by a fill on the device — no host-to-device copy — and cached;
    inside a hipGraph capture a value not met before is made for that capture only (its memory is the graph's)."""
    device = torch.device(device)
    key = (type(value).__name__, repr(value), dtype, device)
    held = _CONSTANTS.get(key)
    if held is None:
        held = torch.full((), value, dtype=dtype, device=device)
        held._aesmc_number = value      # (its value without a device read: parameter checks of a number run on the host)
        if not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            if len(_CONSTANTS) >= _CONSTANTS_LIMIT:      # (a model that anneals a number every step: the cache must not grow
                _CONSTANTS.clear()                       #  without bound; constants still in use stay alive with their users)
            _CONSTANTS[key] = held
    return held


def _broadcast_all(*values):
    """`torch.distributions.utils.broadcast_all` whose number -> tensor step reads the constant cache when the
    tensors it sits beside live on a HIP device."""
    if not _ACTIVE.get():
        return _TORCH_BROADCAST_ALL(*values)
    like = None
    numbers = False
    for value in values:
        if isinstance(value, torch.Tensor):
            if like is None:
                like = value
        elif isinstance(value, _NUMBER):
            numbers = True
        else:
            return _TORCH_BROADCAST_ALL(*values)      # (tensor-likes, or an argument PyTorch will refuse)
    if like is None or not numbers or like.device.type != "cuda":
        return _TORCH_BROADCAST_ALL(*values)
    return torch.broadcast_tensors(*[v if isinstance(v, torch.Tensor) else constant(v, like.dtype, like.device)
                                     for v in values])


def _is_real(constraint):
    while isinstance(constraint, constraints.independent):
        constraint = constraint.base_constraint
    return constraint is constraints.real or type(constraint) is type(constraints.real)


def _one_value(value):
    """A tensor that is ONE element expanded (every stride 0 — what `broadcast_all` makes of a number or of a 0-dim
    parameter) checked as that element: an elementwise constraint holds for all of it or for none."""
    if value.dim() > 0 and value.numel() > 1 and all(stride == 0 for stride in value.stride()):
        return value[(0,) * value.dim()]
    return value


def _defer(valid, flag, what=None):
    """`valid.all()` must hold: read on the host for a host tensor (free), OR-ed into the status word on the device."""
    if not valid.is_cuda:
        return bool(torch._is_all_true(valid))
    from . import _kernels, _lib
    provider = _kernels.get()
    if flag == _lib.FLAG_VALUE_OUTSIDE_SUPPORT:
        provider.defer_support_check(valid)
    else:
        provider.defer_parameter_check(valid)
        if what is not None and what not in _CHECKED:
            _CHECKED.append(what)
            del _CHECKED[:-_CHECKED_LIMIT]
    return True


def checked_parameters():
    """Descriptions of the parameter checks deferred to the device lately ("scale of Normal (GreaterThan(0.0))")."""
    return list(_CHECKED)


def _init(self, batch_shape=torch.Size(), event_shape=torch.Size(), validate_args=None):
    """`Distribution.__init__` (torch/distributions/distribution.py) with the parameter checks left on the device."""
    wanted = self._validate_args if validate_args is None else validate_args
    if not wanted or not _ACTIVE.get() or settings.current().validation_mode != "deferred":
        return _TORCH_INIT(self, batch_shape, event_shape, validate_args)
    _TORCH_INIT(self, batch_shape, event_shape, False)
    self._validate_args = True                        # `log_prob` & co. still validate (through `_validate_sample` below)
    try:
        arg_constraints = self.arg_constraints
    except NotImplementedError:
        return _TORCH_INIT(self, batch_shape, event_shape, True)      # (PyTorch's own warning, no checks to run)
    from . import _lib
    for name, constraint in arg_constraints.items():
        if constraints.is_dependent(constraint):
            continue
        if name not in self.__dict__ and isinstance(getattr(type(self), name, None), lazy_property):
            continue
        value = getattr(self, name)
        if torch.is_tensor(value) and value.is_cuda and _is_real(constraint):
            continue      # only NaN violates it; NaN reaches the log-weights and raises there (FloatingPointError)
        if type(value) in (torch.Tensor, torch.nn.Parameter) and getattr(constraint, "event_dim", 0) == 0:
            base = value if value._base is None else value._base
            number = getattr(base, "_aesmc_number", None)
            if number is not None and base.numel() == 1:
                # a Python number `broadcast_all` turned into a cached device constant: checked on the host, as a
                # host tensor of the same dtype (so the verdict is the one the device values would get)
                value = torch.tensor(number, dtype=value.dtype)
            else:
                value = _one_value(value)      # (plain tensors only: a lazy particle tensor answers `check` without values)
        what = "{} of {} ({})".format(name, type(self).__name__, constraint)
        if not _defer(constraint.check(value), _lib.FLAG_INVALID_PARAMETER, what):
            raise ValueError("Expected parameter {} ({} of shape {}) of distribution {} to satisfy the constraint {}, "
                             "but found invalid values:\n{}".format(name, type(value).__name__, tuple(value.shape),
                                                                     repr(self), repr(constraint), value))


def validate_sample(distribution, value):
    """`Distribution._validate_sample` with the host half done at once (shapes) and the support check on the device;
    what `state.log_prob` calls in 'deferred' mode and what a distribution's own `log_prob` reaches inside `scope()`."""
    if not isinstance(value, torch.Tensor):
        raise ValueError("The value argument to log_prob must be a Tensor")
    event_start = value.dim() - len(distribution.event_shape)
    if value.size()[event_start:] != distribution.event_shape:
        raise ValueError("The right-most size of value must match event_shape: {} vs {}.".format(
            value.size(), distribution.event_shape))
    expected = distribution.batch_shape + distribution.event_shape
    for got, want in zip(reversed(value.size()), reversed(expected)):
        if got != 1 and want != 1 and got != want:
            raise ValueError("Value is not broadcastable with batch_shape+event_shape: {} vs {}."
                             .format(value.size(), expected))
    try:
        support = distribution.support
    except NotImplementedError:
        import warnings
        warnings.warn("{} does not define `support` to enable sample validation. Please "
                      "initialize the distribution with `validate_args=False` to turn off "
                      "validation.".format(distribution.__class__))
        return
    if _is_real(support):
        return  # only NaN violates it; NaN reaches the log-weights and raises there
    from . import _lib
    if not _defer(support.check(value), _lib.FLAG_VALUE_OUTSIDE_SUPPORT):
        raise ValueError("Expected value argument ({} of shape {}) to be within the support ({}) of the distribution "
                         "{}, but found invalid values:\n{}".format(type(value).__name__, tuple(value.shape),
                                                                    repr(support), repr(distribution), value))


def _validate_sample(self, value):
    if not _ACTIVE.get() or settings.current().validation_mode != "deferred" or \
            not (isinstance(value, torch.Tensor) and value.is_cuda):
        return _TORCH_VALIDATE_SAMPLE(self, value)
    return validate_sample(self, value)


def _linear(input, weight, bias=None):
    """`torch.nn.functional.linear` whose small maps over many rows run as K8 / K11 inside a scope."""
    if _ACTIVE.get() and type(input) is torch.Tensor and input.is_cuda and weight.dim() == 2 and input.dim() in (2, 3) and \
            weight.size(0) <= 16 and weight.size(1) <= 16 and input.size(-1) == weight.size(1) and \
            input.numel() >= _PARTICLE_LINEAR_MIN_ROWS * weight.size(1) and input.dtype == weight.dtype and \
            type(weight) in (torch.Tensor, torch.nn.Parameter) and settings.current().particle_linear and \
            (bias is None or (type(bias) in (torch.Tensor, torch.nn.Parameter) and tuple(bias.shape) == (weight.size(0),))):
        from . import _kernels, _ops
        rows = input
        if input.dim() == 2:
            # [N, din] as "batch rows" of a few thousand particles each: K11 leaves one record per tile of 256 particles
            # and the bias' gradient is summed per batch row, a lane walking the row's tiles — 76 us for ONE row of 2^18
            count = input.size(0)
            per_row = next((k for k in (4096, 2048, 1024, 512, 256) if count % k == 0), count)
            rows = input.view(count // per_row, per_row, input.size(1)) if input.is_contiguous() else \
                input.reshape(count // per_row, per_row, input.size(1))
        provider = _kernels.get()
        if provider.name == "hip" and provider.affine_covers(rows, weight, bias):
            out = _ops.particle_affine(rows.contiguous(), weight, bias)
            return out if input.dim() == 3 else out.reshape(input.size(0), weight.size(0))
    return _TORCH_LINEAR(input, weight, bias)


def _install():
    global _INSTALLED
    with _INSTALL_LOCK:
        if _INSTALLED:
            return
        import torch.distributions as package
        for info in pkgutil.iter_modules(package.__path__):
            module = importlib.import_module(package.__name__ + "." + info.name)
            if getattr(module, "broadcast_all", None) is _TORCH_BROADCAST_ALL:
                module.broadcast_all = _broadcast_all
        Distribution.__init__ = _init
        Distribution._validate_sample = _validate_sample
        if torch.nn.functional.linear is _TORCH_LINEAR:
            torch.nn.functional.linear = _linear
        _INSTALLED = True


@contextlib.contextmanager
def scope():
    """While active (this context only), distributions built or scored on a HIP device do not talk to the host."""
    _install()
    token = _ACTIVE.set(True)
    try:
        yield
    finally:
        _ACTIVE.reset(token)


def active():
    return _ACTIVE.get()
