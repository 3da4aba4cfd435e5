"""The package's behaviour switches in ONE place, scoped by context.

Rounds 1-4 kept these as module globals of `state` and `inference` (six of them, plus the setters that wrote them): two
models in one process — or two threads — shared every switch, and a `with` block that flipped one flipped it for
everybody.  Here a `Settings` object holds them all; `current()` is what the code reads:

  * the process-wide defaults (`set_default`, and the module-level setters `state.set_fused_normal`,
    `state.set_validation_mode`, `state.set_kernel_noise`, `inference.set_history_mode`, `inference.set_lazy_gather`,
    which keep their names and meaning);
  * `with override(fused_normal=False, ...)`: a copy with those fields changed, active for the duration of the block in
    THIS context only (contextvars: per thread, per asyncio task) — what `inference.lazy_gather(...)`,
    `inference.fold_gather_backward(...)` and the hipGraph capture use.

The reference has no switches (aesmc/ is one code path); every combination here gives the same numbers — they select
which of this package's equivalent routes runs (see each field) — except `validation_mode`, which chooses WHEN a
violation surfaces.  Environment variables give the defaults' initial values (measurement knobs, read once at import).
"""
import contextlib
import contextvars
import dataclasses
import os


def _env_flag(name, default=True):
    value = os.environ.get(name)
    return default if value is None else value != "0"


@dataclasses.dataclass
class Settings:
    # 'lazy': `previous_latents` gathers an entry on first access; 'eager': the full re-indexed list every step, exactly
    # as aesmc/inference.py:102-104 builds it
    history_mode: str = "lazy"
    # the newest latent is handed to the callables un-gathered and a linear-Gaussian step fetches its rows through the
    # ancestor indices inside its own launch (off: the resampling launch always re-indexes it)  [AESMC_LAZY_GATHER]
    lazy_gather: bool = True
    # consecutive linear-Gaussian steps hand torch.gather's backward from autograd node to autograd node (off: a launch of
    # its own per step)  [AESMC_FOLD_GATHER_BACKWARD]
    fold_gather_backward: bool = True
    # how `state.log_prob` performs the reference's `_validate_sample` (aesmc/state.py:142) on the device: 'deferred'
    # (support checks set a device flag raised at the end of `infer`) or 'eager' (one host sync per call, as the reference)
    validation_mode: str = "deferred"
    # plain `Normal`s go through the fused kernels K4 / K5 / K6 (off: `distribution.log_prob` / `rsample` in eager PyTorch)
    fused_normal: bool = True
    # a deferred draw's float32 noise is formed inside the launch that uses it, from PyTorch's own Philox stream (off:
    # `_standard_normal` materialises it first)  [AESMC_KERNEL_NOISE]
    kernel_noise: bool = True

    # inside `infer`, an `nn.Linear` / `F.linear` of at most 16 x 16 applied to a particle tensor (at least 2^14 rows on the
    # HIP device) runs as kernel K8 — its backward as K11, the weight gradient a matrix-core contraction over the
    # particles — instead of a library GEMM whose picks for such skinny shapes are poor (the reference's own model:
    # Linear(2, 1) over B K rows, test/models/lgssm.py:61-72 — hipBLASLt's weight-gradient GEMM ran 747 us at 262 144
    # rows); off: the library GEMM  [AESMC_PARTICLE_LINEAR]
    particle_linear: bool = True

    # inside `infer`, the FIRST timestep of a model whose time-0 proposal is a BATCH_EXPANDED Normal and whose emission is
    # linear-Gaussian in the latent (the reference's own models: test/models/lgssm.py) runs as ONE launch, K20 — the
    # transposed draw, the emission's location and the three log-densities — with the bits of the three launches it
    # stands for (K6, K8, K5); off: those three  [AESMC_INITIAL_STEP]
    initial_step: bool = True

    def validate(self):
        if self.history_mode not in ("lazy", "eager"):
            raise ValueError("history mode must be 'lazy' or 'eager', got {}".format(self.history_mode))
        if self.validation_mode not in ("deferred", "eager"):
            raise ValueError("validation mode must be 'deferred' or 'eager', got {}".format(self.validation_mode))
        return self


def knob(name, default=None):
    """An environment MEASUREMENT knob (a threshold or a kernel form a timing experiment pins): read only when
    AESMC_MEASUREMENT_KNOBS=1 is set beside it — as the library's own `measurement_knob` (csrc/common.hpp) — so that a
    product process never changes behaviour because some AESMC_* variable happens to be exported."""
    if os.environ.get("AESMC_MEASUREMENT_KNOBS", "0")[:1] != "1":
        return default
    return os.environ.get(name, default)


_DEFAULT = Settings(lazy_gather=_env_flag("AESMC_LAZY_GATHER"), fold_gather_backward=_env_flag("AESMC_FOLD_GATHER_BACKWARD"),
                    kernel_noise=_env_flag("AESMC_KERNEL_NOISE"), particle_linear=_env_flag("AESMC_PARTICLE_LINEAR"),
                    initial_step=_env_flag("AESMC_INITIAL_STEP"))
_SCOPED = contextvars.ContextVar("aesmc_amd_settings", default=None)      # the innermost `override`'s CHANGED fields only
_FIELDS = tuple(field.name for field in dataclasses.fields(Settings))


class _Resolved:
    """What `current()` hands out inside an `override`: every field read is the scoped value if the block (or an
    enclosing one) changed that field, else the process-wide default AS IT IS NOW — so `set_default(...)` and the module
    level setters (`state.set_fused_normal`, ...) called inside a `with override(...)` block take effect at once for the
    fields the block did not pin."""
    __slots__ = ("_changes",)

    def __init__(self, changes):
        object.__setattr__(self, "_changes", changes)

    def __getattr__(self, name):
        changes = object.__getattribute__(self, "_changes")
        if name in changes:
            return changes[name]
        return getattr(_DEFAULT, name)

    def __setattr__(self, name, value):
        raise AttributeError("settings.current() is read-only: use settings.set_default or settings.override")


def current():
    """The settings in force here: per field, the innermost `override` of this context that names it, else the
    process-wide default.  (Context variables are per thread and per asyncio task and are NOT inherited by threads
    started elsewhere: autograd's worker threads and user threads read the process defaults — the autograd functions of
    this package capture what they need in forward.)"""
    scoped = _SCOPED.get()
    return _DEFAULT if scoped is None else _Resolved(scoped)


def set_default(**changes):
    """Changes the process-wide defaults (what every context sees outside an `override`, and inside one for the fields it
    does not name)."""
    candidate = dataclasses.replace(_DEFAULT, **changes).validate()
    for name in changes:
        setattr(_DEFAULT, name, getattr(candidate, name))


@contextlib.contextmanager
def override(**changes):
    """The named fields changed for the duration of the block, in this context only; every other field keeps following
    the defaults (and any enclosing `override`)."""
    unknown = [name for name in changes if name not in _FIELDS]
    if unknown:
        raise TypeError("settings.override: unknown field(s) {}".format(", ".join(unknown)))
    dataclasses.replace(_DEFAULT, **changes).validate()
    token = _SCOPED.set(dict(_SCOPED.get() or {}, **changes))
    try:
        yield
    finally:
        _SCOPED.reset(token)
