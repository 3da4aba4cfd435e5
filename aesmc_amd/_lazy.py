"""A resampled latent that is not gathered until something reads its values.

The reference re-indexes the latent history by the newest ancestors before every proposal / transition call
(aesmc/inference.py:102-104, `state.resample` = `torch.gather`, state.py:179).  A Markov model's callables only
*describe* distributions in terms of `previous_latents[-1]`; when those are linear-Gaussian (`AffineNormal`) the
kernel that weighs the step can fetch the rows of x_{t-1} through the ancestor indices itself, and the
resampled tensor — 168 MB per step at B=1024, K=4096, d=10, written only to be read back — need not exist.

`LazyResampled(source, index)` is what `ResampledHistory` hands out for such an entry: a `torch.Tensor` (so
`torch.is_tensor`, `isinstance` and attribute access behave) of the source's shape, dtype and device that
holds NO values.  Shape-like attributes answer from the source; every other use — any torch function or
method that reaches `__torch_function__` — first materialises the gather (`state.resample`, differentiable,
once) and then runs on the real tensor.  Nothing a model can do with it gives different numbers from the
reference's eager gather; a model that only builds `AffineNormal`s on it never pays for the gather at all.
"""
import torch


class LazyResampled(torch.Tensor):
    @staticmethod
    def __new__(cls, source, index):
        return torch.Tensor._make_wrapper_subclass(cls, source.shape, strides=source.stride(), dtype=source.dtype,
                                                   device=source.device, requires_grad=False)

    def __init__(self, source, index):
        self._lazy_source = source
        self._lazy_index = index
        self._lazy_real = None
        self._lazy_shape = source.shape

    # ---- answered without values -------------------------------------------------------------------
    shape = property(lambda self: self._lazy_shape)
    dtype = property(lambda self: self._lazy_source.dtype)
    device = property(lambda self: self._lazy_source.device)
    is_cuda = property(lambda self: self._lazy_source.is_cuda)
    ndim = property(lambda self: len(self._lazy_shape))
    requires_grad = property(lambda self: self._lazy_source.requires_grad)
    grad_fn = property(lambda self: None if self._lazy_real is None else self._lazy_real.grad_fn)

    def size(self, dim=None):
        return self._lazy_shape if dim is None else self._lazy_shape[dim]

    def dim(self):
        return len(self._lazy_shape)

    def numel(self):
        return self._lazy_source.numel()

    def element_size(self):
        return self._lazy_source.element_size()

    def is_floating_point(self):
        return self._lazy_source.is_floating_point()

    def __repr__(self):
        state = "pending" if self._lazy_real is None else "materialised"
        return "LazyResampled({}, {}, {})".format(tuple(self._lazy_shape), self._lazy_source.dtype, state)

    # ---- everything else needs the values ----------------------------------------------------------
    @property
    def pending(self):
        """(source, index) while nothing has read the values; None afterwards."""
        return None if self._lazy_real is not None else (self._lazy_source, self._lazy_index)

    def materialise(self):
        if self._lazy_real is None:
            from . import state
            self._lazy_real = state.resample(self._lazy_source, self._lazy_index)
        return self._lazy_real

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        with torch._C.DisableTorchFunctionSubclass():
            return func(*_unwrap(args), **{key: _unwrap(value) for key, value in (kwargs or {}).items()})

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # not reached through the Python API (__torch_function__ has unwrapped by then); C++ callers that
        # bypass it get the same treatment: the gather, then the operator on the real tensor
        return func(*_unwrap(args), **{key: _unwrap(value) for key, value in (kwargs or {}).items()})


def _unwrap(value):
    if isinstance(value, LazyResampled):
        return value.materialise()
    if isinstance(value, (list, tuple)):
        return type(value)(_unwrap(item) for item in value)
    return value


def real(tensor):
    """The tensor itself, a LazyResampled gathered."""
    return tensor.materialise() if type(tensor) is LazyResampled else tensor
