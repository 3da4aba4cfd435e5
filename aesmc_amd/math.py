"""Log-space normalisation helpers with the semantics of the reference's aesmc/math.py.

Both functions accept a torch.Tensor or a numpy.ndarray and return the same kind.  The reduction
runs in the fused log-sum-exp kernel (K1) on the HIP device for both kinds: a numpy array is
staged to the current device and the result copied back (the reference used scipy on the host).
"""
import numpy as np
import torch

from . import _ops


def _logsumexp_keepdim(values, dim):
    """logsumexp over `dim` of a HIP tensor, keeping that dim, through K1's row reduction."""
    moved = values.movedim(dim, -1)
    rows = moved.reshape(-1, moved.size(-1))
    lse = _ops.row_logsumexp(rows)
    return lse.reshape(moved.shape[:-1]).unsqueeze(-1).movedim(-1, dim)


def lognormexp(values, dim=0):
    """values - logsumexp(values, dim): the log of exp(values) normalised along `dim`
    (aesmc/math.py:6-30).  Note the reference's default dim is 0."""
    if isinstance(values, np.ndarray):
        device = torch.device("cuda", torch.cuda.current_device())
        if not np.issubdtype(values.dtype, np.floating):
            values = values.astype(np.float64)  # scipy promotes integer input the same way
        result = lognormexp(torch.from_numpy(np.ascontiguousarray(values)).to(device), dim=dim)
        return result.cpu().numpy()
    return values - _logsumexp_keepdim(values, dim)


def exponentiate_and_normalize(values, dim=0):
    """softmax of `values` along `dim` (aesmc/math.py:33-51)."""
    if isinstance(values, np.ndarray):
        return np.exp(lognormexp(values, dim=dim))
    return torch.exp(lognormexp(values, dim=dim))
