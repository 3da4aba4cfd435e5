"""Kernel provider backed by oracle/kernel_oracle.py (NumPy) with the interface of
aesmc_amd._kernels.HipKernels, for exercising the package's host logic on CPU tensors in tests."""
import torch

from oracle import kernel_oracle


def _t(array, like):
    return torch.from_numpy(array).to(like.dtype if array.dtype.kind == "f" else None)


class OracleKernels:
    name = "oracle"

    def __init__(self):
        self._flags = 0

    def discard_flags(self):
        self._flags = 0

    def read_flags(self, device):
        flags, self._flags = self._flags, 0
        return flags

    def logweight_lse(self, a, b=None, c=None, want_lw=True, want_lse=True):
        arrays = [None if t is None else t.detach().numpy() for t in (a, b, c)]
        lw, lse = kernel_oracle.logweight_lse(*arrays)
        return (torch.from_numpy(lw) if want_lw else None,
                torch.from_numpy(lse) if want_lse else None)

    def logweight_accumulate(self, a, b, c, acc, want_lw=True, want_lse=False):
        arrays = [None if t is None else t.detach().numpy() for t in (a, b, c)]
        lw, total, lse = kernel_oracle.logweight_accumulate(*arrays, acc.detach().numpy())
        return (torch.from_numpy(lw) if want_lw else None, torch.from_numpy(total),
                torch.from_numpy(lse) if want_lse else None)

    def logweight_lse_backward(self, lw, lse, grad_lw, grad_lse, want_neg=True):
        g, ng = kernel_oracle.logweight_lse_backward(
            lw.detach().numpy(), lse.detach().numpy(),
            None if grad_lw is None else grad_lw.detach().numpy(),
            None if grad_lse is None else grad_lse.detach().numpy())
        return torch.from_numpy(g), (torch.from_numpy(ng) if want_neg else None)

    def ancestor_index(self, log_w, u):
        idx, flags = kernel_oracle.ancestor_index(log_w.detach().numpy(), u.numpy())
        self._flags |= flags
        out = torch.from_numpy(idx)
        out._aesmc_sorted = True
        return out

    def gather(self, src, idx):
        assert idx.size() == src.size()[:2]
        out, flags = kernel_oracle.gather(src.detach().numpy(), idx.numpy())
        self._flags |= flags
        return torch.from_numpy(out)

    def gather_backward(self, grad_out, idx, sorted_index=False):
        out, flags = kernel_oracle.gather_backward(grad_out.detach().contiguous().numpy(), idx.numpy())
        self._flags |= flags
        return torch.from_numpy(out)

    def step_covers(self, log_w, payload=None):
        return log_w.dim() == 2 and log_w.numel() > 0 and (payload is None or torch.is_tensor(payload))

    def resample_step(self, log_w, u, payload=None, want_lse=False):
        """The fused step as the composition it must equal: K2, then K1's row log-sum-exp and K3."""
        idx = self.ancestor_index(log_w, u)
        lse = self.logweight_lse(log_w, None, None, want_lw=False, want_lse=True)[1] if want_lse else None
        moved = self.gather(payload, idx) if payload is not None else None
        return idx, lse, moved
