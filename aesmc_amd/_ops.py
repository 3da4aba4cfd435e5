"""Differentiable operators of the SMC hot path, each a thin autograd wrapper over one kernel.

Gradient contract of the reference that these preserve:
  * no gradient flows through ancestor indices (`.detach()` at aesmc/inference.py:254);
  * gradients flow through the resample gather into earlier latents (torch.gather, state.py:179);
  * gradients flow through the per-step log-sum-exp into the log-weights (inference.py:130).
"""
import torch

from . import _kernels


class _LogWeightLSE(torch.autograd.Function):
    """(lw, lse) = (a + b - c, logsumexp_k(a + b - c)); b and c optional."""

    @staticmethod
    def forward(ctx, a, b, c):
        k = _kernels.get()
        lw, lse = k.logweight_lse(a, b, c, want_lw=True, want_lse=True)
        ctx.save_for_backward(lw, lse)
        ctx.has = (b is not None, c is not None)
        if lw is a:  # pure row-LSE: autograd outputs must not alias inputs
            lw = a.view_as(a)
        return lw, lse

    @staticmethod
    def backward(ctx, grad_lw, grad_lse):
        lw, lse = ctx.saved_tensors
        has_b, has_c = ctx.has
        need_c = has_c and ctx.needs_input_grad[2]
        g, ng = _kernels.get().logweight_lse_backward(lw, lse, grad_lw, grad_lse, want_neg=need_c)
        return (g if ctx.needs_input_grad[0] else None,
                g if (has_b and ctx.needs_input_grad[1]) else None,
                ng if need_c else None)


class _ResampleGather(torch.autograd.Function):
    """value[b, idx[b,k], ...] with a segmented-sum backward; idx carries no gradient."""

    @staticmethod
    def forward(ctx, value, idx):
        out = _kernels.get().gather(value, idx)
        ctx.save_for_backward(idx)
        # set only by kernel K2 on its own outputs; any other index tensor takes the general path
        ctx.sorted_index = bool(getattr(idx, "_aesmc_sorted", False))
        ctx.mark_non_differentiable(idx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _kernels.get().gather_backward(grad_out, idx, sorted_index=ctx.sorted_index), None


class _NormalLogProbSum(torch.autograd.Function):
    """sum over trailing dims of Normal(loc, scale).log_prob(value); all three are views of
    value's shape, so autograd's own expand-backward reduces gradients of broadcast operands."""

    @staticmethod
    def forward(ctx, value, loc, scale):
        ctx.save_for_backward(value, loc, scale)
        return _kernels.get().normal_logprob_sum(value, loc, scale)

    @staticmethod
    def backward(ctx, grad_out):
        value, loc, scale = ctx.saved_tensors
        return _kernels.get().normal_logprob_sum_backward(value, loc, scale, grad_out,
                                                          *ctx.needs_input_grad)


def normal_log_prob_sum(value, loc, scale):
    """[B,K] summed Normal log-density (kernel K4); differentiable in value, loc and scale."""
    if torch.is_grad_enabled() and (value.requires_grad or loc.requires_grad or scale.requires_grad):
        return _NormalLogProbSum.apply(value, loc, scale)
    return _kernels.get().normal_logprob_sum(value.detach(), loc.detach(), scale.detach())


def logweight_lse(a, b=None, c=None):
    """Returns (log_weight [B,K], logsumexp over particles [B]) for log_weight = a + b - c."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (a, b, c)):
        return _LogWeightLSE.apply(a, b, c)
    return _kernels.get().logweight_lse(a, b, c, want_lw=True, want_lse=True)


def row_logsumexp(x):
    """logsumexp over dim 1 of a [B,K] tensor through K1 (differentiable)."""
    return logweight_lse(x)[1]


def resample_gather(value, idx):
    if torch.is_grad_enabled() and value.requires_grad and value.is_floating_point():
        return _ResampleGather.apply(value, idx)
    return _kernels.get().gather(value, idx)


def ancestor_index(log_weight, uniforms):
    """Systematic-resampling ancestor indices; never differentiable."""
    return _kernels.get().ancestor_index(log_weight.detach(), uniforms)
