"""Differentiable operators of the SMC hot path, each a thin autograd wrapper over one kernel.

Gradient contract of the reference that these preserve:
  * no gradient flows through ancestor indices (`.detach()` at aesmc/inference.py:254);
  * gradients flow through the resample gather into earlier latents (torch.gather, state.py:179);
  * gradients flow through the per-step log-sum-exp into the log-weights (inference.py:130).
"""
import torch

from . import _kernels
from . import _philox
from ._lazy import LazyParticles, LazyResampled


class _LogWeightLSE(torch.autograd.Function):
    """(lw, lse) = (a + b - c, logsumexp_k(a + b - c)); b and c optional."""

    @staticmethod
    def forward(ctx, a, b, c):
        ctx.set_materialize_grads(False)   # an unused output costs nothing (no zero tensors made)
        k = _kernels.get()
        lw, lse = k.logweight_lse(a, b, c, want_lw=True, want_lse=True)
        ctx.save_for_backward(lw, lse)
        ctx.has = (b is not None, c is not None)
        if lw is a:  # pure row-LSE: autograd outputs must not alias inputs
            lw = a.view_as(a)
        return lw, lse

    @staticmethod
    def backward(ctx, grad_lw, grad_lse):
        if grad_lw is None and grad_lse is None:
            return None, None, None
        lw, lse = ctx.saved_tensors
        has_b, has_c = ctx.has
        need_c = has_c and ctx.needs_input_grad[2]
        g, ng = _kernels.get().logweight_lse_backward(lw, lse, grad_lw, grad_lse, want_neg=need_c)
        return (g if ctx.needs_input_grad[0] else None,
                g if (has_b and ctx.needs_input_grad[1]) else None,
                ng if need_c else None)


class _LogWeightAccumulate(torch.autograd.Function):
    """(lw, total, lse) = (a + b - c, acc + lw, logsumexp_k total): K1 with the running sum over time
    that importance sampling normalises (aesmc/inference.py:156-159)."""

    @staticmethod
    def forward(ctx, a, b, c, acc, want_lse):
        ctx.set_materialize_grads(False)
        lw, total, lse = _kernels.get().logweight_accumulate(a, b, c, acc, want_lw=True, want_lse=want_lse)
        ctx.save_for_backward(total, lse)
        ctx.has = (b is not None, c is not None)
        if lw is a:
            lw = a.view_as(a)
        return lw, total, lse

    @staticmethod
    def backward(ctx, grad_lw, grad_total, grad_lse):
        if grad_lw is None and grad_total is None and grad_lse is None:
            return None, None, None, None, None
        total, lse = ctx.saved_tensors
        has_b, has_c = ctx.has
        # d total / d lw = d total / d acc = 1;  d lse / d total = softmax(total)
        if grad_lse is not None and lse is not None:
            through, _ = _kernels.get().logweight_lse_backward(total, lse, grad_total, grad_lse, want_neg=False)
        else:
            through = grad_total
        g = through if grad_lw is None else (grad_lw if through is None else through + grad_lw)
        need_c = has_c and ctx.needs_input_grad[2]
        return (g if ctx.needs_input_grad[0] else None,
                g if (has_b and ctx.needs_input_grad[1]) else None,
                (-g if g is not None else None) if need_c else None,
                through if ctx.needs_input_grad[3] else None, None)


def logweight_accumulate(a, b, c, acc, want_lse=False):
    """One importance-sampling step's bookkeeping in one launch: returns (log_weight = a + b - c,
    running sum acc + log_weight, its logsumexp over particles or None)."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (a, b, c, acc)):
        return _LogWeightAccumulate.apply(a, b, c, acc, want_lse)
    return _kernels.get().logweight_accumulate(a, b, c, acc, want_lw=True, want_lse=want_lse)


class _ResampleGather(torch.autograd.Function):
    """value[b, idx[b,k], ...] with a segmented-sum backward; idx carries no gradient."""

    @staticmethod
    def forward(ctx, value, idx):
        out = _kernels.get().gather(value, idx)
        ctx.save_for_backward(idx)
        # set only by kernel K2 on its own outputs; any other index tensor takes the general path
        ctx.sorted_index = getattr(idx, "_aesmc_sorted", False)      # False, True (K2 wrote it) or "inherited"
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


class _Declined(Exception):
    """A fused kernel does not cover the operands it was offered (raised inside an autograd
    Function's forward, caught by the operator that offered them)."""


class _NormalLogWeight(torch.autograd.Function):
    """lw = logN(x; p) + logN(y; g) - logN(x; q) through kernel K5; backward through K4's."""

    @staticmethod
    def forward(ctx, x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q):
        out = _kernels.get().normal_logweight(x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q)
        if out is None:
            raise _Declined()   # caught by normal_log_weight: the caller takes the K4 + K1 route
        ctx.save_for_backward(x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q)
        return out

    @staticmethod
    def backward(ctx, grad):
        x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q = ctx.saved_tensors
        need = ctx.needs_input_grad
        k = _kernels.get()
        grad = grad.contiguous()
        fused = k.normal_logweight_backward(x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q, grad,
                                            list(need))
        if fused is not None:
            return tuple(fused)
        gx_p, g_loc_p, g_scale_p = k.normal_logprob_sum_backward(x, loc_p, scale_p, grad, need[0], need[1], need[2])
        g_y, g_loc_g, g_scale_g = k.normal_logprob_sum_backward(y, loc_g, scale_g, grad, need[3], need[4], need[5])
        gx_q, g_loc_q, g_scale_q = k.normal_logprob_sum_backward(x, loc_q, scale_q, -grad, need[0], need[6], need[7])
        gx = gx_p + gx_q if need[0] else None
        return gx, g_loc_p, g_scale_p, g_y, g_loc_g, g_scale_g, g_loc_q, g_scale_q


class _NormalLogWeightLSE(torch.autograd.Function):
    """Ties a row log-sum-exp that some later launch produced (the next step's resampling kernel, or
    K1 after the last step) to the operands of the K5 launch whose log-weights it reduces.  Forward
    returns the value as it is; backward runs K5's backward with K1's softmax term formed in place
    (`aesmc_normal_logweight_lse_backward`): one launch and one [B,K] round trip less per timestep
    than K1's backward followed by K5's, same numbers bit for bit."""

    @staticmethod
    def forward(ctx, lse, lw, x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q):
        ctx.save_for_backward(lse, lw, x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q)
        return lse.view_as(lse)

    @staticmethod
    def backward(ctx, grad_lse):
        lse, lw, x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q = ctx.saved_tensors
        need = list(ctx.needs_input_grad[2:])
        k = _kernels.get()
        grad_lse = grad_lse.contiguous()
        fused = k.normal_logweight_backward(x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q, None, need,
                                            lw=lw, lse=lse, grad_lse=grad_lse)
        if fused is None:   # not reached with today's kernels (the backward is elementwise); kept for safety
            grad, _ = k.logweight_lse_backward(lw, lse, None, grad_lse, want_neg=False)
            fused = k.normal_logweight_backward(x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q, grad, need)
        return (None, None) + tuple(fused)


# ---- K20: the first timestep in one launch (state._initial_step) -------------------------------------------
class _NormalRsampleGiven(torch.autograd.Function):
    """draw = loc + eps * scale whose VALUES another launch has already formed (K20 wrote `value`): `_NormalRsample`'s
    node without its launch — the same backward."""

    @staticmethod
    def forward(ctx, eps, loc, scale, value):
        ctx.save_for_backward(eps if scale.requires_grad else None)
        return value.view_as(value)

    @staticmethod
    def backward(ctx, grad):
        (eps,) = ctx.saved_tensors
        grad_loc = grad if ctx.needs_input_grad[1] else None
        grad_scale = grad * eps if ctx.needs_input_grad[2] else None
        return None, grad_loc, grad_scale, None


def normal_rsample_given(eps, loc, scale, value):
    """`value` = loc + eps * scale (formed elsewhere) as a differentiable function of loc and scale."""
    if torch.is_grad_enabled() and (loc.requires_grad or scale.requires_grad):
        return _NormalRsampleGiven.apply(eps, loc, scale, value)
    return value


class InitialOperands(tuple):
    """The operands of a K20 launch as `attach_lse` wants them: (x_0, loc_p, scale_p, y, location port, weight, offset,
    scale_g, loc_q, scale_q) — K5's eight with the emission's location, which was never written, replaced by its
    autograd port (`particle_affine_port`) and the map that gives its values."""


class _InitialLogWeightLSE(torch.autograd.Function):
    """`_NormalLogWeightLSE` for a step K20 weighed: the emission's location C x_0 + g exists only inside that launch, so
    the backward evaluates it first (K8: the values K20 used) and then runs K5's backward with K1's softmax term formed
    in place; the location's gradient leaves through the port — `_ParticleAffine`'s node, created where the three-launch
    route creates it — so every parameter's gradient is accumulated from the same contributions in the same order: the
    same bits as that route."""

    @staticmethod
    def forward(ctx, lse, lw, x, loc_p, scale_p, y, port, weight, offset, scale_g, loc_q, scale_q):
        ctx.save_for_backward(lse, lw, x, loc_p, scale_p, y, weight, offset, scale_g, loc_q, scale_q)
        return lse.view_as(lse)

    @staticmethod
    def backward(ctx, grad_lse):
        lse, lw, x, loc_p, scale_p, y, weight, offset, scale_g, loc_q, scale_q = ctx.saved_tensors
        wanted = ctx.needs_input_grad
        need8 = [wanted[2], wanted[3], wanted[4], wanted[5], wanted[6], wanted[9], wanted[10], wanted[11]]
        k = _kernels.get()
        loc_g = k.particle_affine(x, weight, offset)
        grad_lse = grad_lse.contiguous()
        fused = k.normal_logweight_backward(x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q, None, need8,
                                            lw=lw, lse=lse, grad_lse=grad_lse)
        if fused is None:   # not reached with today's kernels (the backward is elementwise); kept for safety
            grad, _ = k.logweight_lse_backward(lw, lse, None, grad_lse, want_neg=False)
            fused = k.normal_logweight_backward(x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q, grad, need8)
        g_x, g_loc_p, g_scale_p, g_y, g_loc_g, g_scale_g, g_loc_q, g_scale_q = fused
        return None, None, g_x, g_loc_p, g_scale_p, g_y, g_loc_g, None, None, g_scale_g, g_loc_q, g_scale_q


def normal_log_weight_deferred(x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q):
    """K5 forward WITHOUT an autograd node: (log-weights [B,K] carrying no gradient, the operands
    as given) — or None when K5 does not cover the operands.  For callers that differentiate the
    log-weights only through their row log-sum-exp, which they attach with `attach_lse`."""
    tensors = (x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q)
    if not _kernels.get().normal_logweight_covers(x, scale_p, y, scale_g, scale_q):
        return None
    lw = _kernels.get().normal_logweight(*[t.detach() for t in tensors])
    if lw is None:
        return None
    return lw, tensors


def attach_lse(lse, lw, operands):
    """The row log-sum-exp `lse` [B] of `lw` as a differentiable function of the `operands` of the
    launch that produced `lw` (K5's eight views, or K10's AffineOperands)."""
    if isinstance(operands, PendingStep):
        return operands.bind(lse.detach())
    if isinstance(operands, AffineOperands):
        return _AffineLogWeightLSE.apply(lse.detach(), lw, *operands)
    if isinstance(operands, InitialOperands):
        return _InitialLogWeightLSE.apply(lse.detach(), lw, *operands)
    return _NormalLogWeightLSE.apply(lse.detach(), lw, *operands)


def operands_require_grad(operands):
    return any(t is not None and t.requires_grad for t in operands)


def normal_log_weight(x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q):
    """[B,K] log-weight of one step for three Normal terms (kernel K5), or None if K5 does not
    cover the operands (the caller then sums three K4 terms through K1: same numbers)."""
    tensors = (x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q)
    if not _kernels.get().normal_logweight_covers(x, scale_p, y, scale_g, scale_q):
        return None
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        try:
            return _NormalLogWeight.apply(*tensors)
        except _Declined:
            return None
    return _kernels.get().normal_logweight(*[t.detach() for t in tensors])


class _NormalRsample(torch.autograd.Function):
    """draw = loc + eps * scale (kernel K6); loc and scale arrive expanded to eps's shape, so the
    gradients returned here are dense and autograd's expand-backward folds the broadcast dims."""

    @staticmethod
    def forward(ctx, eps, loc, scale):
        ctx.save_for_backward(eps if scale.requires_grad else None)
        return _kernels.get().normal_rsample(eps, loc, scale)

    @staticmethod
    def backward(ctx, grad):
        (eps,) = ctx.saved_tensors
        grad_loc = grad if ctx.needs_input_grad[1] else None
        grad_scale = grad * eps if ctx.needs_input_grad[2] else None
        return None, grad_loc, grad_scale


class _NormalRsampleDrawn(torch.autograd.Function):
    """draw = loc + n * scale with the noise n formed in the launch (K6 drawn); only for a scale that needs no
    gradient (its gradient is grad * n, and n is not kept)."""

    @staticmethod
    def forward(ctx, noise, loc, scale, shape):
        out = _kernels.get().normal_rsample_drawn(noise, loc, scale, shape)
        if out is None:
            raise _Declined()
        return out

    @staticmethod
    def backward(ctx, grad):
        return None, grad, None, None


def normal_rsample_drawn(noise, loc, scale, shape):
    """K6 with the noise of the reservation `noise` formed in the launch, or None where that launch does not apply
    (nothing has been launched then: the caller materialises the noise)."""
    if torch.is_grad_enabled() and scale.requires_grad:
        return None
    if torch.is_grad_enabled() and loc.requires_grad:
        try:
            return _NormalRsampleDrawn.apply(noise, loc, scale.detach(), shape)
        except _Declined:
            return None
    return _kernels.get().normal_rsample_drawn(noise, loc.detach(), scale.detach(), shape)


def normal_rsample(eps, loc, scale):
    """Reparameterised Normal draw from standard-normal noise `eps` [B,K,*] (kernel K6)."""
    if torch.is_grad_enabled() and (loc.requires_grad or scale.requires_grad):
        return _NormalRsample.apply(eps, loc, scale)
    return _kernels.get().normal_rsample(eps, loc.detach(), scale.detach())


def logweight_lse(a, b=None, c=None):
    """Returns (log_weight [B,K], logsumexp over particles [B]) for log_weight = a + b - c."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (a, b, c)):
        return _LogWeightLSE.apply(a, b, c)
    return _kernels.get().logweight_lse(a, b, c, want_lw=True, want_lse=True)


def row_logsumexp(x):
    """logsumexp over dim 1 of a [B,K] tensor through K1 (differentiable)."""
    return logweight_lse(x)[1]


class _ResampleStep(torch.autograd.Function):
    """(idx, lse, moved) of the fused resampling step.  idx carries no gradient
    (aesmc/inference.py:254); lse differentiates into the log-weights as K1's does; moved =
    payload[b, idx[b,k]] differentiates into the payload by the sorted segmented sum."""

    @staticmethod
    def forward(ctx, log_w, uniforms, payload, want_lse, carrier=None):
        # `carrier`: a step node's [B] stand-in for this very log-sum-exp (PendingStep): the gradient of `lse`
        # goes to it instead of into the log-weights
        ctx.set_materialize_grads(False)
        out = _kernels.get().resample_step(log_w, uniforms, payload, want_lse)
        if out is None:
            raise RuntimeError("aesmc_amd: fused resampling step rejected operands it was offered")
        idx, lse, moved = out
        ctx.mark_non_differentiable(idx)
        ctx.save_for_backward(log_w, lse, idx)
        return idx, lse, moved

    @staticmethod
    def backward(ctx, _grad_idx, grad_lse, grad_moved):
        log_w, lse, idx = ctx.saved_tensors
        k = _kernels.get()
        grad_w = grad_payload = grad_carrier = None
        if len(ctx.needs_input_grad) > 4 and ctx.needs_input_grad[4]:
            grad_carrier = grad_lse
        elif ctx.needs_input_grad[0] and lse is not None and grad_lse is not None:
            grad_w, _ = k.logweight_lse_backward(log_w, lse, None, grad_lse, want_neg=False)
        if ctx.needs_input_grad[2] and grad_moved is not None:
            grad_payload = k.gather_backward(grad_moved, idx, sorted_index=True)
        return grad_w, None, grad_payload, None, grad_carrier


def resample_step(log_weight, uniforms, payload=None, want_lse=False, pending=None, want_child_end=False):
    """One resampling step: (ancestor indices [B,K], logsumexp over particles [B] or None,
    payload[b, idx[b,k], ...] or None).  One launch when the fused kernel covers the operands;
    `moved` is None when it does not cover the payload (the caller gathers with the indices).
    `pending` (with want_lse): the PendingStep of the step that produced `log_weight` — the log-sum-exp
    comes back bound to it (its gradient reaches that step's node), with no autograd node of its own."""
    k = _kernels.get()
    if not k.step_covers(log_weight):
        idx = ancestor_index(log_weight, uniforms)
        lse = row_logsumexp(log_weight) if want_lse else None
        return idx, (pending.bind(lse.detach()) if (pending is not None and lse is not None) else lse), None
    if payload is not None and not k.step_covers(log_weight, payload):
        payload = None
    bind = pending is not None and want_lse and torch.is_grad_enabled()
    wants_grad = torch.is_grad_enabled() and (
        (want_lse and log_weight.requires_grad and not bind) or
        (payload is not None and payload.requires_grad and payload.is_floating_point()))
    if bind and not wants_grad:
        # the log-sum-exp belongs to a step node: no autograd node here at all — `infer` ties the VALUE to the
        # node's carrier once, for all timesteps together (bind_rows)
        idx, lse, moved = k.resample_step(log_weight.detach(), uniforms, None if payload is None else payload.detach(),
                                          want_lse, want_child_end=want_child_end)
        pending.box[0] = lse
        lse = LseOf(lse, pending)
    elif wants_grad and bind:
        idx, lse, moved = _ResampleStep.apply(log_weight, uniforms, payload, want_lse, pending.carrier)
        pending.box[0] = lse.detach()
    elif wants_grad:
        idx, lse, moved = _ResampleStep.apply(log_weight, uniforms, payload, want_lse)
    else:
        idx, lse, moved = k.resample_step(log_weight.detach(), uniforms,
                                          None if payload is None else payload.detach(), want_lse)
        if pending is not None and lse is not None:
            lse = pending.bind(lse)
    idx._aesmc_sorted = True
    return idx, lse, moved


def resample_gather(value, idx):
    if torch.is_grad_enabled() and value.requires_grad and value.is_floating_point():
        return _ResampleGather.apply(value, idx)
    return _kernels.get().gather(value, idx)


def ancestor_index(log_weight, uniforms):
    """Systematic-resampling ancestor indices; never differentiable."""
    return _kernels.get().ancestor_index(log_weight.detach(), uniforms)


# ---- linear-Gaussian particle propagation (K8 / K9 / K10) --------------------------------------------

class _ParticleAffine(torch.autograd.Function):
    """x @ weight.T + offset through kernel K8 (or tanh of it: the same launch); backward: the adjoint map through K8 on
    the transposed weight view, the weight gradient through the outer-sum kernel — behind tanh's own factor 1 - out^2
    where the location left through it."""

    @staticmethod
    def forward(ctx, x, weight, offset, through_tanh=False):
        out = _kernels.get().particle_affine(x, weight, offset, through_tanh=through_tanh)
        ctx.save_for_backward(x, weight, out if through_tanh else None)
        ctx.offset_shape = None if offset is None else tuple(offset.shape)
        return out

    @staticmethod
    def backward(ctx, grad):
        x, weight, out = ctx.saved_tensors
        k = _kernels.get()
        need_x, need_w, need_off = ctx.needs_input_grad[:3]
        need_off = need_off and ctx.offset_shape is not None
        if out is not None:
            grad = grad * (1 - out * out)      # d tanh(v) / d v, from the stored output as torch.tanh's backward does
        gx, gw, rows = k.particle_affine_backward(grad.contiguous(), x, weight, need_x, need_w, need_off)
        goff = None if not need_off else (rows if len(ctx.offset_shape) == 2 else rows.sum(dim=0))
        return gx, gw, goff, None


class _ParticleAffinePort(_ParticleAffine):
    """`_ParticleAffine`'s node for a location nobody writes (K20 forms it inside its launch): the output holds no values
    — one element expanded to the location's shape — and exists to receive the location's gradient where the written
    location would have; the backward is `_ParticleAffine`'s."""

    @staticmethod
    def forward(ctx, x, weight, offset, through_tanh=False):
        ctx.save_for_backward(x, weight, None)
        ctx.offset_shape = None if offset is None else tuple(offset.shape)
        return x.new_zeros(()).expand(x.size(0), x.size(1), weight.size(0))


def particle_affine_port(x, weight, offset=None):
    """The autograd identity of `particle_affine(x, weight, offset)` without its values (see `_ParticleAffinePort`)."""
    return _ParticleAffinePort.apply(x, weight, offset, False)


def particle_affine(x, weight, offset=None, through_tanh=False):
    """[B,K,dout] location  offset + x @ weight.T  (kernel K8) — or tanh of it from the same launch —, differentiable in
    x, weight and offset."""
    if isinstance(x, LazyParticles):
        x = x.materialise()
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, weight, offset)):
        return _ParticleAffine.apply(x, weight, offset, through_tanh)
    return _kernels.get().particle_affine(x.detach(), weight.detach(), None if offset is None else offset.detach(),
                                          through_tanh=through_tanh)


# ---- K13: a two-layer tanh net over the particles --------------------------------------------------------
def _mlp_reference(x, weight1, offset1, weight2, bias2):
    hidden = torch.matmul(x, weight1.t()) + (offset1.unsqueeze(1) if offset1.dim() == 2 else offset1)
    out = torch.matmul(torch.tanh(hidden), weight2.t())
    return out if bias2 is None else out + bias2


class _MlpDeclined(Exception):
    pass


class _ParticleMlp(torch.autograd.Function):
    """Forward: kernel K13 (the hidden layer never leaves registers).  Backward: kernel K13b — the hidden layer recomputed
    from the saved inputs, the two weight gradients contracted over the particles on the matrix cores; where it declines
    the shape (K not a multiple of 256), PyTorch's autograd over the same expression."""

    @staticmethod
    def forward(ctx, x, weight1, offset1, weight2, bias2):
        out = _kernels.get().particle_mlp(x, weight1, offset1, weight2, bias2)
        if out is None:
            raise _MlpDeclined()
        ctx.save_for_backward(x, weight1, offset1, weight2, bias2)
        return out

    @staticmethod
    def backward(ctx, grad):
        x, weight1, offset1, weight2, bias2 = ctx.saved_tensors
        need = ctx.needs_input_grad
        fused = _kernels.get().particle_mlp_backward(grad, x, weight1, offset1, weight2, need_x=need[0])
        if fused is not None:
            grad_x, grad_w1, grad_rows, grad_w2 = fused
            grad_offset = None
            if need[2]:
                grad_offset = grad_rows if offset1.dim() == 2 else grad_rows.sum(0)
            grad_bias = grad.sum((0, 1)) if (bias2 is not None and need[4]) else None
            return (grad_x if need[0] else None, grad_w1 if need[1] else None, grad_offset,
                    grad_w2 if need[3] else None, grad_bias)
        saved = (x, weight1, offset1, weight2, bias2)
        with torch.enable_grad():
            leaves = [None if t is None else t.detach().requires_grad_(flag) for t, flag in zip(saved, need)]
            out = _mlp_reference(*leaves)
            wanted = [t for t, flag in zip(leaves, need) if t is not None and flag]
            grads = iter(torch.autograd.grad(out, wanted, grad))
        return tuple(next(grads) if (t is not None and flag) else None for t, flag in zip(leaves, need))


def particle_mlp(x, weight1, offset1, weight2, bias2=None):
    """bias2 + tanh(offset1 + x @ weight1.T) @ weight2.T over particles x [B,K,din]; kernels K13 / K13b when they
    cover the shape, the PyTorch expression otherwise (same numbers to rounding)."""
    if isinstance(x, LazyParticles):
        x = x.materialise()
    k = _kernels.get()
    if k.name == "hip" and not (torch.is_tensor(x) and x.is_cuda):
        raise RuntimeError("aesmc_amd: particle_mlp operand lives on '{}'; this package computes only on a "
                           "HIP device (MI355X) and has no CPU fallback.".format(getattr(x, "device", None)))
    if not k.particle_mlp_covers(x, weight1, offset1, weight2, bias2):
        return _mlp_reference(x, weight1, offset1, weight2, bias2)      # (other extents: PyTorch's operators, on the device)
    tensors = (x, weight1, offset1, weight2, bias2)
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        try:
            return _ParticleMlp.apply(*tensors)
        except _MlpDeclined:
            return _mlp_reference(*tensors)
    out = k.particle_mlp(*[None if t is None else t.detach() for t in tensors])
    return out if out is not None else _mlp_reference(*tensors)


class _AffineRsample(torch.autograd.Function):
    """draw = (offset + source @ weight.T) + eps * scale in one launch (kernel K9)."""

    @staticmethod
    def forward(ctx, source, weight, offset, scale, eps):
        ctx.save_for_backward(source, weight, eps if scale.requires_grad else None)
        ctx.offset_shape = None if offset is None else tuple(offset.shape)
        ctx.scale_shape = tuple(scale.shape)
        return _kernels.get().affine_rsample(source, weight, offset, eps, scale)

    @staticmethod
    def backward(ctx, grad):
        source, weight, eps = ctx.saved_tensors
        return _affine_rsample_backward(ctx, grad, source, weight, eps)


def _affine_rsample_backward(ctx, grad, source, weight, eps):
    need_src, need_w, need_off, need_scale = ctx.needs_input_grad[:4]
    need_off = need_off and ctx.offset_shape is not None
    gsrc, gw, rows = _kernels.get().particle_affine_backward(grad.contiguous(), source, weight, need_src, need_w,
                                                            need_off)
    goff = None if not need_off else (rows if len(ctx.offset_shape) == 2 else rows.sum(dim=0))
    gscale = (grad * eps).sum().reshape(ctx.scale_shape) if need_scale else None
    return gsrc, gw, goff, gscale, None


def affine_rsample(source, weight, offset, scale, eps):
    """Reparameterised draw from Normal(offset + source @ weight.T, scale) given the noise (kernel K9)."""
    if isinstance(source, LazyParticles):
        source = source.materialise()
    tensors = (source, weight, offset, scale)
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        return _AffineRsample.apply(source, weight, offset, scale, eps)
    return _kernels.get().affine_rsample(source.detach(), weight.detach(),
                                         None if offset is None else offset.detach(), eps, scale.detach())


class AffineOperands(tuple):
    """The twelve operands of one K10 launch, in the order
    (x_prev, x, y_rows, A, off_p, C, off_g, Q, off_q, s_p, s_g, s_q); offsets may be None.
    `pending_gather` = (x_{t-1}, ancestors) when x_prev is a LazyResampled nothing has gathered yet: the
    launch fetches its rows through the indices (aesmc_affine_normal_propagate_resampled)."""
    pending_gather = None
    is_draw = False
    wide = False      # rows wider than the fused kernels take (K17 / K18's extent): the backward recomputes, nothing is folded

    def requires_grad(self):
        return any(t is not None and t.requires_grad for t in self)

    def gathered(self):
        """The same operands with x_prev as a real tensor (gathered now if it had not been)."""
        if self.pending_gather is None and not isinstance(self[0], LazyParticles):
            return self
        out = AffineOperands((self[0].materialise(),) + tuple(self[1:]))
        out.is_draw = self.is_draw
        return out

    def with_latent(self, x_t):
        """The same operands with the latent being weighed (slot 1) replaced — a lazy draw by the tensor that
        receives / holds its values."""
        out = AffineOperands((self[0], x_t) + tuple(self[2:]))
        out.is_draw, out.pending_gather, out.wide = self.is_draw, self.pending_gather, self.wide
        return out


def _affine_logweight_grads(operands, need, grad_lw=None, lw=None, lse=None, grad_lse=None):
    x_prev, x, y_rows, A, off_p, C, off_g, Q, off_q, s_p, s_g, s_q = operands
    return _kernels.get().affine_logweight_backward(
        x_prev, x, y_rows, (A, off_p), (C, off_g), (Q, off_q), (s_p, s_g, s_q), list(need),
        grad_lw=grad_lw, lw=lw, lse=lse, grad_lse=grad_lse)


class _AffineLogWeight(torch.autograd.Function):
    """K10 with the log-weights themselves differentiable."""

    @staticmethod
    def forward(ctx, *operands):
        x_prev, x, y_rows, A, off_p, C, off_g, Q, off_q, s_p, s_g, s_q = operands
        ctx.save_for_backward(*[t for t in operands if t is not None])
        ctx.present = [t is not None for t in operands]
        return _kernels.get().affine_logweight(x_prev, x, y_rows, (A, off_p), (C, off_g), (Q, off_q),
                                               (s_p, s_g, s_q))

    @staticmethod
    def backward(ctx, grad):
        saved = iter(ctx.saved_tensors)
        operands = [next(saved) if present else None for present in ctx.present]
        return tuple(_affine_logweight_grads(operands, ctx.needs_input_grad, grad_lw=grad.contiguous()))


class _AffineLogWeightLSE(torch.autograd.Function):
    """The counterpart of _NormalLogWeightLSE for a K10 launch: ties a row log-sum-exp produced by a
    later launch to K10's operands; backward forms K1's softmax term where the step's gradient is
    consumed."""

    @staticmethod
    def forward(ctx, lse, lw, *operands):
        ctx.save_for_backward(lse, lw, *[t for t in operands if t is not None])
        ctx.present = [t is not None for t in operands]
        return lse.view_as(lse)

    @staticmethod
    def backward(ctx, grad_lse):
        lse, lw = ctx.saved_tensors[:2]
        saved = iter(ctx.saved_tensors[2:])
        operands = [next(saved) if present else None for present in ctx.present]
        grads = _affine_logweight_grads(operands, ctx.needs_input_grad[2:], lw=lw, lse=lse,
                                        grad_lse=grad_lse.contiguous())
        return (None, None) + tuple(grads)


class LseOf:
    """A step's row log-sum-exp as a plain value plus the step node it belongs to (see bind_rows)."""
    __slots__ = ("value", "pending")

    def __init__(self, value, pending):
        self.value, self.pending = value, pending


class StepLink:
    """What lets two consecutive steps' autograd nodes skip torch.gather's backward between them.  Step t+1 read the
    rows of x_t through ancestors; its backward (K14) has the gradient of those rows, one per CHILD.  Instead of
    summing children into ancestors in a launch of its own and handing the sum to autograd, it leaves the per-child
    gradient HERE and returns no gradient for x_t; step t's backward — which autograd runs later, x_t being its
    output — picks it up and sums each particle's children while it consumes them.  Sound as long as nothing else can
    ask autograd for the gradient of x_t itself: `infer` arranges this only when it hands the latents to nobody
    (return_latents = return_original_latents = False, what `losses.get_loss` asks for).

    The same hand-over carries the gradients of what the two steps SHARE: a time-homogeneous model passes the same A, C,
    Q and scales to every step and autograd only wants the sum over the steps, so step t+1's backward leaves its sums
    for them unfinished (`carry`: K14's per-workgroup records) and step t's launch starts from them — no finishing
    launch and no accumulation per step, one of each per run of steps (`shared`: the parameters this link's step
    received, what the next step compares its own with)."""
    __slots__ = ("deposit", "carry", "shared")

    def __init__(self, shared=None):
        self.deposit = None
        self.carry = None
        self.shared = shared

    def take(self):
        deposit, self.deposit = self.deposit, None
        return deposit

    def take_carry(self):
        carry, self.carry = self.carry, None
        return carry


def _same_parameter(a, b):
    """Do two steps' operands stand for ONE parameter in autograd's eyes — the same tensor, or the same view of the
    same base (x @ W.t() makes a new view object per call) — so that the sum of their gradients may reach either?"""
    if a is b:
        return True
    if a is None or b is None or a._base is None or a._base is not b._base:
        return False
    return (a.shape == b.shape and a.stride() == b.stride() and a.storage_offset() == b.storage_offset() and
            a.requires_grad == b.requires_grad and type(a.grad_fn) is type(b.grad_fn))


class PendingStep:
    """The link between a step's autograd node (_AffineStep, made when the step is weighed) and its row
    log-sum-exp, which the NEXT resampling launch produces: `carrier` is the node's [B] output that takes
    the log-sum-exp's gradient back to the node, `box[0]` the values once `bind` has seen them."""
    __slots__ = ("carrier", "box")

    def __init__(self):
        self.carrier = None
        self.box = [None]       # what the node keeps: the values only (the carrier would close a reference cycle)

    def bind(self, lse):
        self.box[0] = lse
        return _BindLse.apply(self.carrier, lse)


class _BindRows(torch.autograd.Function):
    """stack [T,B] of per-step row log-sum-exps whose rows `positions` are VALUES produced by resampling launches
    for steps that own a `_AffineStep` node: the gradient of row positions[i] goes to carriers[i] (one node per
    ELBO instead of one per timestep); the other rows differentiate through `stack` as they are."""

    @staticmethod
    def forward(ctx, stack, positions, *carriers):
        ctx.positions = positions
        return stack.view_as(stack)

    @staticmethod
    def backward(ctx, grad):
        # (a reduction like `-lml.sum() / n` sends back an EXPANDED gradient — every stride zero —, and each step's node
        #  would make its own dense copy of its row for K14 to read: one launch per timestep; made dense once here)
        if ctx.positions and not grad.is_contiguous():
            grad = grad.contiguous()
        return (grad, None) + tuple(grad[position] for position in ctx.positions)


def bind_rows(stack, bound):
    """`bound`: [(row, PendingStep)] — ties those rows of the [T,B] stack to their steps' nodes."""
    if not bound or not torch.is_grad_enabled():
        return stack
    return _BindRows.apply(stack, [row for row, _ in bound], *[pending.carrier for _, pending in bound])


class _BindLse(torch.autograd.Function):
    """lse [B] (values from the launch that reduced the rows) as a function of a step node's carrier."""

    @staticmethod
    def forward(ctx, carrier, lse):
        return lse.view_as(lse)

    @staticmethod
    def backward(ctx, grad):
        return grad, None


class _AffineStep(torch.autograd.Function):
    """One SMC step of a linear-Gaussian model as ONE autograd node: inputs the step's operands (x_t only
    as a value: it is the proposal's reparameterised draw of the same x_{t-1}), outputs x_t itself — the
    tensor every later consumer (the next resampling gather, the callables, the returned latents) reads —
    and a [B] carrier that stands for the row log-sum-exp of the step's log-weights until the launch
    that reduces the rows (the next resampling step) has produced it (PendingStep.bind).  Backward
    (kernel K14) therefore receives both the ELBO's gradient at the log-sum-exp and whatever arrives at
    x_t from later steps, and carries them through the densities AND the draw to x_{t-1} and the
    parameters in one pass: the draw's own node (K11), K12's x_t gradient and the two [B,K,d]
    accumulations autograd would put between them never run."""

    @staticmethod
    def forward(ctx, lw, x_value, pending, ancestors, links, *operands):
        # `ancestors` (or None): operands[0] is the UN-resampled x_{t-1}; the step read its rows through them
        # `links` = (this step's StepLink or None, the previous step's StepLink + children ranges or None)
        ctx.set_materialize_grads(False)      # an output nobody differentiated arrives as None, not as zeros
        ctx.lse_box = pending.box
        ctx.own_link, ctx.parent_link, ctx.defer_shared = links
        child_end = None if ctx.parent_link is None else ctx.parent_link[1]
        ctx.save_for_backward(lw, x_value, ancestors, child_end, *[t for t in operands if t is not None])
        ctx.present = [t is not None for t in operands]
        return lw.new_empty((lw.size(0),)), x_value.view_as(x_value)     # the carrier's values are never read

    @staticmethod
    def backward(ctx, grad_lse, grad_x):
        lw, x_value, ancestors, parent_child_end = ctx.saved_tensors[:4]
        saved = iter(ctx.saved_tensors[4:])
        operands = [next(saved) if present else None for present in ctx.present]
        x_prev, _, y_rows, A, off_p, C, off_g, Q, off_q, s_p, s_g, s_q = operands
        need = list(ctx.needs_input_grad[5:])
        need[1] = False
        lse = ctx.lse_box[0]
        if grad_lse is not None and lse is None:
            raise RuntimeError("aesmc_amd internal error: a step's log-sum-exp received a gradient but was never bound")
        k = _kernels.get()
        # what the NEXT step left for this one: the gradient of the rows it resampled from x_t, per child
        child = ctx.own_link.take() if ctx.own_link is not None else None
        carry = ctx.own_link.take_carry() if ctx.own_link is not None else None
        if child is not None and ancestors is None:     # (this step did not go through ancestors itself: sum here)
            summed = k.gather_backward_ranges(child[0], child[1])
            grad_x, child = (summed if grad_x is None else grad_x + summed), None
        # the parameters shared with the steps around: start from what the next step left, leave the sums to the step
        # before (StepLink) — only this run's first step finishes them
        chain = None
        if ancestors is not None and (carry is not None or ctx.defer_shared):
            chain = {"carry": carry, "defer": ctx.defer_shared}
        elif carry is not None:      # (no launch of this step's can take them: finished on their own)
            dx, dy = x_value.size(2), y_rows.size(1)
            carried = k.affine_backward_collect(carry, x_value.dtype, x_value.device, dx, dy, need, (s_p, s_g, s_q))
        grads = k.affine_step_backward(
            x_prev, x_value, y_rows, (A, off_p), (C, off_g), (Q, off_q), (s_p, s_g, s_q), need, lw, lse,
            grad_lse=None if grad_lse is None else grad_lse.contiguous(), grad_x=grad_x, ancestors=ancestors,
            child_grad=None if child is None else child[0], child_end=None if child is None else child[1], chain=chain)
        if chain is None and carry is not None:
            for slot, value in enumerate(carried):
                if value is not None:
                    grads[slot] = value if grads[slot] is None else grads[slot] + value
        if chain is not None and chain["left"] is not None:
            ctx.parent_link[0].carry = chain["left"]
        if ancestors is not None and grads[0] is not None:
            if ctx.parent_link is not None:
                # the previous step's node sums these children into their ancestors itself (StepLink): no launch here,
                # and no gradient for x_{t-1} through autograd
                ctx.parent_link[0].deposit = (grads[0], parent_child_end)
                grads[0] = None
            else:
                # torch.gather's backward (state.py:179): children's gradients summed into their ancestors; K2's
                # indices are non-decreasing along k, so this is the atomic-free segmented sum
                grads[0] = k.gather_backward(grads[0], ancestors, sorted_index=True)
        return (None, None, None, None, None) + tuple(grads)


_SHARED_SLOTS = (3, 5, 7, 9, 10, 11)      # A, C, Q, s_p, s_g, s_q among a step's operands
_CHAIN_SHARED = True                      # (test hook: False = every step finishes and returns its own sums)


def affine_step(lw, operands, fold_gather_backward=False):
    """(PendingStep, x_t) of a K10 step whose x_t is the proposal's draw: one autograd node ties x_t and —
    once `attach_lse` binds it — the row log-sum-exp of `lw` to the step's operands (see _AffineStep).
    `operands[1]` (x_t) enters as a value.  `fold_gather_backward`: consecutive such steps hand the gather's
    backward from node to node (StepLink) — only when nobody outside `infer` holds the latents."""
    pending = PendingStep()
    inputs = list(operands)
    inputs[1] = None
    ancestors = None
    if operands.pending_gather is not None:       # the launch fetched x_{t-1}'s rows through the ancestors
        inputs[0], ancestors = operands.pending_gather
    elif isinstance(inputs[0], LazyParticles):
        inputs[0] = inputs[0].materialise()
    own_link = parent_link = None
    defer_shared = False
    if fold_gather_backward and not operands.wide:
        shared = tuple(inputs[slot] for slot in _SHARED_SLOTS)
        own_link = StepLink(shared)
        if ancestors is not None:
            previous = getattr(inputs[0], "_aesmc_step_link", None)      # x_{t-1} is the previous step's output
            child_end = getattr(ancestors, "_aesmc_child_end", None)
            if previous is not None and child_end is not None:
                parent_link = (previous, child_end)
                defer_shared = _CHAIN_SHARED and previous.shared is not None and torch.is_grad_enabled() and \
                    any(t is not None and t.requires_grad for t in shared) and \
                    all(_same_parameter(a, b) for a, b in zip(shared, previous.shared))
    pending.carrier, x_t = _AffineStep.apply(lw, operands[1].detach(), pending, ancestors,
                                             (own_link, parent_link, defer_shared), *inputs)
    if own_link is not None:
        x_t._aesmc_step_link = own_link
    return pending, x_t


def affine_log_weight(operands):
    """[B,K] log-weight of one step whose three terms are affine Normals (kernel K10)."""
    if torch.is_grad_enabled() and operands.requires_grad():
        return _AffineLogWeight.apply(*operands)
    x_prev, x, y_rows, A, off_p, C, off_g, Q, off_q, s_p, s_g, s_q = [
        None if t is None else t.detach() for t in operands]
    return _kernels.get().affine_logweight(x_prev, x, y_rows, (A, off_p), (C, off_g), (Q, off_q), (s_p, s_g, s_q))


def affine_log_weight_deferred(operands):
    """K10 forward WITHOUT an autograd node: (log-weights, operands) for `attach_lse`."""
    x_prev, x, y_rows, A, off_p, C, off_g, Q, off_q, s_p, s_g, s_q = [
        None if t is None else t.detach() for t in operands]
    lw = _kernels.get().affine_logweight(x_prev, x, y_rows, (A, off_p), (C, off_g), (Q, off_q), (s_p, s_g, s_q))
    return lw, operands


def affine_propagate(operands, eps):
    """K15, no autograd node: fills operands[1] (the deferred draw x_t) from the noise `eps` and returns the
    step's log-weights [B,K] — K9 and K10 in one launch, the same bits as the two.  With
    `operands.pending_gather` the launch also performs the resampling gather of x_{t-1} on the way in; if it
    declines the shape, the gather happens first (and `operands.pending_gather` is cleared)."""
    # (no autograd node is made here and the launcher only takes addresses: the operands go in as they are)
    x, y_rows, A, off_p, C, off_g, Q, off_q, s_p, s_g, s_q = operands[1:]
    x = x.detach()
    k = _kernels.get()
    if isinstance(eps, _philox.NoiseStream):
        # the noise was only RESERVED in PyTorch's generator: K16 forms it (and, if pending, the gather) in the launch
        if operands.pending_gather is not None:
            source, ancestors = operands.pending_gather
        else:
            x_prev = operands[0]
            source, ancestors = (x_prev.materialise() if isinstance(x_prev, LazyParticles) else x_prev), None
        lw = k.affine_propagate_drawn(source, eps, y_rows, (A, off_p), (C, off_g), (Q, off_q), (s_p, s_g, s_q),
                                      out_x=x, ancestors=ancestors)
        if lw is not None:
            return lw
        eps = k.philox_normal(eps, tuple(x.shape), x.device)      # not a shape K16 covers: the values as a tensor
    if operands.pending_gather is not None:
        source, ancestors = operands.pending_gather
        lw = k.affine_propagate(source.detach(), eps, y_rows, (A, off_p), (C, off_g), (Q, off_q), (s_p, s_g, s_q),
                                out_x=x, checked=True, ancestors=ancestors)
        if lw is not None:
            return lw
        operands.pending_gather = None
    x_prev = operands[0]
    x_prev = (x_prev.materialise() if isinstance(x_prev, LazyParticles) else x_prev).detach()
    return k.affine_propagate(x_prev, eps, y_rows, (A, off_p), (C, off_g), (Q, off_q),
                              (s_p, s_g, s_q), out_x=x, checked=True)    # state._affine_step_operands did
