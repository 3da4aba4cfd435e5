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

    def affine_backward_collect(self, left, dtype, device, dx, dy, need, scales):
        return list(left[0])

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

    def resample_step(self, log_w, u, payload=None, want_lse=False, want_child_end=False):
        """The fused step as the composition it must equal: K2, then K1's row log-sum-exp and K3."""
        idx = self.ancestor_index(log_w, u)
        if want_child_end and payload is None:      # child_end[b,k] = #{k' : idx[b,k'] <= k}
            K = idx.size(1)
            counts = torch.zeros(idx.size(0), K + 1, dtype=torch.int64).scatter_add_(
                1, idx.clamp(max=K), torch.ones_like(idx))
            idx._aesmc_child_end = counts[:, :K].cumsum(dim=1).to(torch.int32)
        lse = self.logweight_lse(log_w, None, None, want_lw=False, want_lse=True)[1] if want_lse else None
        moved = self.gather(payload, idx) if payload is not None else None
        return idx, lse, moved

    # ---- linear-Gaussian particle propagation (K8 / K9 / K10) on the C oracle ----------------------
    affine_max_dim = 16

    def affine_covers(self, source, weight, offset=None):
        if not (torch.is_tensor(source) and torch.is_tensor(weight) and source.dim() == 3 and weight.dim() == 2):
            return False
        if source.dtype not in (torch.float32, torch.float64) or weight.dtype != source.dtype:
            return False
        dout, din = weight.shape
        if din != source.size(2) or max(dout, din) > self.affine_max_dim or source.numel() == 0:
            return False
        return offset is None or tuple(offset.shape) in ((dout,), (source.size(0), dout))

    def affine_logweight_covers(self, x_prev, x, y_rows, transition, emission, proposal, scales):
        if x_prev.shape != x.shape or y_rows.dim() != 2:
            return False
        if not (self.affine_covers(x_prev, *transition) and self.affine_covers(x, *emission) and
                self.affine_covers(x_prev, *proposal)):
            return False
        dx = x.size(2)
        return transition[0].size(0) == dx and proposal[0].size(0) == dx and \
            y_rows.shape == (x.size(0), emission[0].size(0)) and all(s.numel() == 1 for s in scales)

    @staticmethod
    def _n(t):
        return None if t is None else t.detach().contiguous().numpy()

    def particle_affine(self, x1, w1, offset=None, x2=None, w2=None, base=None, through_tanh=False):
        from oracle import c_oracle
        out = torch.from_numpy(c_oracle.particle_affine(self._n(x1), self._n(w1), self._n(x2), self._n(w2),
                                                        self._n(offset), self._n(base)))
        return torch.tanh(out) if through_tanh else out

    def affine_rsample(self, source, weight, offset, eps, scale, out=None):
        from oracle import c_oracle
        draw = torch.from_numpy(c_oracle.affine_rsample(self._n(source), self._n(weight), self._n(offset),
                                                        self._n(eps), float(scale)))
        if out is None:
            return draw
        out.copy_(draw)
        return out

    def affine_propagate(self, x_prev, eps, y_rows, transition, emission, proposal, scales, out_x, checked=False,
                         ancestors=None):
        """K15 on the C oracle: its draw, then its log-weight of that draw (with `ancestors`: of the gathered
        x_prev — the composition the fused launch must equal)."""
        from oracle import c_oracle
        if ancestors is not None:
            x_prev = self.gather(x_prev, ancestors)
        pair = lambda term: (self._n(term[0]), self._n(term[1]))
        out_x.copy_(torch.from_numpy(c_oracle.affine_rsample(self._n(x_prev), self._n(proposal[0]), self._n(proposal[1]),
                                                             self._n(eps), float(scales[2]))))
        return torch.from_numpy(c_oracle.affine_logweight(
            self._n(x_prev), self._n(out_x), self._n(y_rows), pair(transition), pair(emission), pair(proposal),
            float(scales[0]), float(scales[1]), float(scales[2])))

    def affine_logweight(self, x_prev, x, y_rows, transition, emission, proposal, scales):
        from oracle import c_oracle
        pair = lambda term: (self._n(term[0]), self._n(term[1]))
        return torch.from_numpy(c_oracle.affine_logweight(
            self._n(x_prev), self._n(x), self._n(y_rows), pair(transition), pair(emission), pair(proposal),
            float(scales[0]), float(scales[1]), float(scales[2])))

    # ---- K13 / K13b on the C oracle (forward) and float64 autograd (backward) ----------------------------------
    def particle_mlp_covers(self, x, weight1, offset1, weight2, bias2=None):
        return torch.is_tensor(x) and x.dim() == 3 and x.dtype in (torch.float32, torch.float64) and \
            weight1.dim() == 2 and weight1.size(1) == x.size(2) <= 16 and weight1.size(0) <= 64 and \
            weight2.dim() == 2 and weight2.size(1) == weight1.size(0) and weight2.size(0) <= 16

    def particle_mlp(self, x, weight1, offset1, weight2, bias2=None):
        from oracle import c_oracle
        return torch.from_numpy(c_oracle.particle_mlp(self._n(x), self._n(weight1), self._n(offset1), self._n(weight2),
                                                      None if bias2 is None else self._n(bias2)))

    def particle_mlp_backward(self, grad_out, x, weight1, offset1, weight2, need_x=True):
        with torch.enable_grad():
            leaves = [t.detach().double().requires_grad_(True) for t in (x, weight1, offset1, weight2)]
            hidden = torch.tanh(leaves[0] @ leaves[1].t() + (leaves[2].unsqueeze(1) if leaves[2].dim() == 2 else leaves[2]))
            out = hidden @ leaves[3].t()
            gx, gw1, goff, gw2 = torch.autograd.grad(out, leaves, grad_out.double())
        rows = goff if goff.dim() == 2 else goff.unsqueeze(0).expand(x.size(0), -1) / x.size(0)
        return (gx.to(x.dtype) if need_x else None, gw1.to(x.dtype), rows.to(x.dtype), gw2.to(x.dtype))

    def particle_affine_backward(self, grad, x, weight, need_x=True, need_weight=True, need_offset=False):
        g2, x2 = grad.reshape(-1, grad.size(-1)).double(), x.reshape(-1, x.size(-1)).double()
        gx = (grad.double() @ weight.double()).to(grad.dtype) if need_x else None
        gw = (g2.t() @ x2).to(grad.dtype) if need_weight else None
        goff = grad.double().sum(dim=1).to(grad.dtype) if need_offset else None
        return gx, gw, goff

    def affine_logweight_backward(self, x_prev, x, y_rows, transition, emission, proposal, scales, need,
                                  grad_lw=None, lw=None, lse=None, grad_lse=None):
        """The adjoint by PyTorch's own autograd over the float64 expression the kernel evaluates."""
        operands = [x_prev, x, y_rows, transition[0], transition[1], emission[0], emission[1], proposal[0],
                    proposal[1]] + list(scales)
        with torch.enable_grad():
            return self._affine_logweight_backward(operands, need, grad_lw, lw, lse, grad_lse)

    @staticmethod
    def _affine_logweight_backward(operands, need, grad_lw, lw, lse, grad_lse):
        leaves = [None if t is None else t.detach().double().requires_grad_(True) for t in operands]
        xp, xx, yy, A, op, C, og, Q, oq, sp, sg, sq = leaves

        def loc(source, weight, offset):
            out = source @ weight.t()
            if offset is not None:
                out = out + (offset.unsqueeze(1) if offset.dim() == 2 else offset)
            return out

        normal = torch.distributions.Normal
        value = (normal(loc(xp, A, op), sp).log_prob(xx).sum(-1) +
                 normal(loc(xx, C, og), sg).log_prob(yy.unsqueeze(1)).sum(-1) -
                 normal(loc(xp, Q, oq), sq).log_prob(xx).sum(-1))
        g = torch.zeros_like(value)
        if grad_lw is not None:
            g = g + grad_lw.double()
        if grad_lse is not None:
            g = g + grad_lse.double().unsqueeze(1) * torch.exp(lw.double() - lse.double().unsqueeze(1))
        wanted = [i for i, t in enumerate(leaves) if t is not None and need[i]]
        grads = torch.autograd.grad(value, [leaves[i] for i in wanted], grad_outputs=g, allow_unused=True)
        out = [None] * 12
        for i, grad in zip(wanted, grads):
            out[i] = None if grad is None else grad.to(operands[i].dtype)
        return out

    def gather_backward_ranges(self, child_grad, child_end):
        B, K = child_end.shape
        out = torch.zeros_like(child_grad)
        ends = child_end.to(torch.int64)
        for b in range(B):
            start = 0
            for k in range(K):
                end = int(ends[b, k])
                if end > start:
                    out[b, k] = child_grad[b, start:end].sum(dim=0)
                start = max(start, end)
        return out

    def affine_step_backward(self, x_prev, x, y_rows, transition, emission, proposal, scales, need, lw, lse,
                             grad_lse=None, grad_x=None, grad_lw=None, ancestors=None, child_grad=None, child_end=None,
                             chain=None):
        """K14's contract by PyTorch's autograd in float64: x is rebuilt as the proposal's draw
        loc_q(x_prev) + s_q eps with eps = (x - loc_q) / s_q held fixed, so every path through it is
        differentiated; x's own slot stays None.  `chain`: the shared parameters' gradients are handed from call to
        call (here: as the list of them) instead of being returned by every call."""
        if chain is not None:
            own = OracleKernels.affine_step_backward(self, x_prev, x, y_rows, transition, emission, proposal, scales, need,
                                                     lw, lse, grad_lse=grad_lse, grad_x=grad_x, grad_lw=grad_lw,
                                                     ancestors=ancestors, child_grad=child_grad, child_end=child_end)
            shared = (3, 5, 7, 9, 10, 11)
            if chain["carry"] is not None:
                for slot in shared:
                    carried = chain["carry"][0][slot]
                    if carried is not None:
                        own[slot] = carried if own[slot] is None else own[slot] + carried
            chain["left"] = None
            if chain["defer"]:
                chain["left"] = ([own[slot] if slot in shared else None for slot in range(12)], 1)
                for slot in shared:
                    own[slot] = None
            return own
        if child_grad is not None:      # the next step's per-child gradient: summed into its ancestors, then as grad_x
            summed = self.gather_backward_ranges(child_grad, child_end)
            grad_x = summed if grad_x is None else grad_x + summed
        if ancestors is not None:       # slot 0 of the result: the gradient of the RESAMPLED rows
            x_prev = self.gather(x_prev, ancestors)
        operands = [x_prev, x, y_rows, transition[0], transition[1], emission[0], emission[1], proposal[0],
                    proposal[1]] + list(scales)
        with torch.enable_grad():
            leaves = [None if t is None else t.detach().double().requires_grad_(True) for t in operands]
            xp, xx, yy, A, op, C, og, Q, oq, sp, sg, sq = leaves

            def loc(source, weight, offset):
                out = source @ weight.t()
                if offset is not None:
                    out = out + (offset.unsqueeze(1) if offset.dim() == 2 else offset)
                return out

            with torch.no_grad():
                eps = (xx - loc(xp, Q, oq)) / sq
            draw = loc(xp, Q, oq) + sq * eps
            normal = torch.distributions.Normal
            value = (normal(loc(xp, A, op), sp).log_prob(draw).sum(-1) +
                     normal(loc(draw, C, og), sg).log_prob(yy.unsqueeze(1)).sum(-1) -
                     normal(loc(xp, Q, oq), sq).log_prob(draw).sum(-1))
            g = torch.zeros_like(value)
            if grad_lw is not None:
                g = g + grad_lw.double()
            if grad_lse is not None:
                g = g + grad_lse.double().unsqueeze(1) * torch.exp(lw.double() - lse.double().unsqueeze(1))
            total = (value * g).sum()
            if grad_x is not None:
                total = total + (draw * grad_x.double()).sum()
            wanted = [i for i, t in enumerate(leaves) if t is not None and need[i] and i != 1]
            grads = torch.autograd.grad(total, [leaves[i] for i in wanted], allow_unused=True)
        out = [None] * 12
        for i, grad in zip(wanted, grads):
            out[i] = None if grad is None else grad.to(operands[i].dtype)
        return out
