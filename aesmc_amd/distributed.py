"""Batch sharding of the SMC ELBO across the GPUs of one node.

Every operation on the hot path is independent per batch row (log-sum-exp, CDF scan and search,
gather all run along the particle dim), so rank r simply owns rows [lo, hi) of the batch with
model parameters replicated: no data-path collective.  The one exchange step is the reduction of
sum_b log Z_b (one scalar per ELBO evaluation) — RCCL all-reduce over xGMI with backend "nccl";
"gloo" on CPU in tests.  For training the parameter gradients are all-reduced once per step as one
flat bucket.

For index parity with an unsharded run every rank draws the FULL [global_batch, 1] uniform block
from numpy's global RandomState (same seed on every rank) and keeps its own rows, see
`shard_scope`.
"""
import contextlib
import contextvars

import torch
import torch.distributed as dist

from . import inference

# (global_batch_size, lo, hi) while inside shard_scope; per thread / per context, so a sharded
# evaluation in one thread does not re-slice the uniforms of another
_ACTIVE_SHARD = contextvars.ContextVar("aesmc_amd_active_shard", default=None)


def shard_bounds(global_batch_size, rank, world_size):
    """Rows [lo, hi) of the batch owned by `rank` (contiguous, sizes differ by at most one)."""
    base, extra = divmod(global_batch_size, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_observations(observations, rank, world_size):
    """Slices every observation [global_batch, ...] (or dict of them) down to this rank's rows."""
    first = observations[0]
    first = next(iter(first.values())) if isinstance(first, dict) else first
    lo, hi = shard_bounds(first.size(0), rank, world_size)

    def cut(obs):
        if isinstance(obs, dict):
            return {key: cut(item) for key, item in obs.items()}
        return obs[lo:hi]

    return [cut(obs) for obs in observations]


@contextlib.contextmanager
def shard_scope(global_batch_size, rank, world_size):
    """While active, the resampler's per-step uniforms are drawn for the whole global batch and
    sliced to this rank's rows, so a sharded run consumes numpy's RNG exactly like an unsharded
    one and produces the same ancestor indices row for row."""
    lo, hi = shard_bounds(global_batch_size, rank, world_size)
    token = _ACTIVE_SHARD.set((global_batch_size, lo, hi))
    try:
        yield
    finally:
        _ACTIVE_SHARD.reset(token)


def active_shard():
    return _ACTIVE_SHARD.get()


def _group_is_live():
    """True inside an initialised process group (also a one-rank group: same code path)."""
    return dist.is_available() and dist.is_initialized()


class _AllReduceSum(torch.autograd.Function):
    """y = sum over ranks of x; dy/dx = 1 on every rank (each rank differentiates the global
    objective with respect to its local contribution)."""

    @staticmethod
    def forward(ctx, x, group):
        y = x.detach().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, grad):
        return grad, None


class _AllGatherRows(torch.autograd.Function):
    """The [global_batch] vector of per-row values from every rank's rows (ranks in order); the
    gradient of a rank's rows is its slice of the incoming gradient."""

    @staticmethod
    def forward(ctx, local, global_batch_size, rank, world_size, group):
        spans = [shard_bounds(global_batch_size, r, world_size) for r in range(world_size)]
        longest = max(hi - lo for lo, hi in spans)
        padded = local.detach().new_zeros(longest)
        padded[:local.numel()] = local.detach()
        pieces = [torch.empty_like(padded) for _ in range(world_size)]
        dist.all_gather(pieces, padded, group=group)
        ctx.span = spans[rank]
        return torch.cat([piece[:hi - lo] for piece, (lo, hi) in zip(pieces, spans)])

    @staticmethod
    def backward(ctx, grad):
        lo, hi = ctx.span
        return grad[lo:hi], None, None, None, None


def sharded_get_loss(local_observations, num_particles, algorithm, initial, transition, emission,
                     proposal, global_batch_size, rank=None, world_size=None, group=None,
                     exact_mean=False):
    """`losses.get_loss` for a batch sharded over the process group: each rank runs `infer` on its
    rows, then ONE all-reduce of the local sum of log Z_b yields -mean over the global batch.
    `exact_mean=True` all-gathers the [global_batch] vector of log Z_b instead and takes
    `-torch.mean` of it, which reproduces the unsharded loss to the last bit (the all-reduced sum
    can differ from it in the final place)."""
    rank = dist.get_rank(group) if rank is None else rank
    world_size = dist.get_world_size(group) if world_size is None else world_size
    with shard_scope(global_batch_size, rank, world_size):
        result = inference.infer(
            {"iwae": "is", "aesmc": "smc"}[algorithm], local_observations, initial, transition,
            emission, proposal, num_particles, return_log_marginal_likelihood=True,
            return_latents=False, return_log_weight=False)
    if exact_mean and _group_is_live():
        everyone = _AllGatherRows.apply(result["log_marginal_likelihood"], global_batch_size, rank,
                                        world_size, group)
        return -torch.mean(everyone)
    local_sum = result["log_marginal_likelihood"].sum()
    total = _AllReduceSum.apply(local_sum, group) if _group_is_live() else local_sum
    return -total / global_batch_size


def all_reduce_gradients(parameters, group=None):
    """Sums parameter gradients over ranks with one flat-bucket all-reduce (the model is tiny
    next to the particle state: one latency-bound collective per optimiser step)."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads or not _group_is_live():
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    offset = 0
    for g in grads:
        g.copy_(flat[offset:offset + g.numel()].view_as(g))
        offset += g.numel()


def train(dataloader, num_particles, algorithm, initial, transition, emission, proposal, num_epochs,
          num_iterations_per_epoch=None, optimizer_algorithm=torch.optim.Adam, optimizer_kwargs={},
          callback=None, group=None, hip_graph=False, verify_replays=4, reverify_every=256):
    """`train.train` (aesmc/train.py:22-41) with one process per GPU: every rank's `dataloader`
    yields ITS OWN rows of each minibatch (equal counts on all ranks), the loss is the mean over
    the global batch (one all-reduce of sum log Z, `sharded_get_loss`) and the parameter gradients
    are summed over ranks by one flat-bucket all-reduce before each optimiser step, so replicas
    stay identical.  Seed numpy identically on all ranks (the resampler draws the global uniform
    block and keeps its rows) and torch differently per rank (independent proposal noise).
    `callback` sees the global loss.  `hip_graph=True` replays each rank's share of loss + backward
    as one captured hipGraph (`graphs.GraphedLoss(shard=...)`, its capture consuming no random numbers: the eager loop's
    trajectory); the two collectives stay outside it.  A replay freezes whatever the callables computed on the HOST at its
    value during the capture (`train.train`'s docstring: the frozen-callables contract): every `reverify_every`-th replay
    is also evaluated eagerly on the same rows with the same draws and compared — the ranks agree on the verdict (one
    all-reduce) — and on a mismatch every rank drops its graph with a RuntimeWarning and the loop is the eager sharded
    loop from that step on."""
    from . import train as _train
    rank = dist.get_rank(group) if _group_is_live() else 0
    world_size = dist.get_world_size(group) if _group_is_live() else 1
    model_parts = (initial, transition, emission, proposal)
    parameters = list(_train.get_chained_params(*model_parts))
    optimizer = optimizer_algorithm(parameters, **optimizer_kwargs)
    graphed = None
    for epoch, iteration, observations in _train._minibatches(dataloader, num_epochs,
                                                              num_iterations_per_epoch):
        if hip_graph:
            from . import graphs
            if graphed is None:
                optimizer.zero_grad(set_to_none=True)
                shard = (observations[0].size(0) * world_size, rank, world_size)
                graphed = graphs.GraphedLoss(observations, num_particles, algorithm, *model_parts,
                                             backward=True, shard=shard, group=group, check_flags=False,
                                             guard_gradients=True, verify_replays=verify_replays,
                                             preserve_random_state=True)
            problems = []
            if reverify_every and (graphed.replays + 1) % reverify_every == 0:
                problems, loss = _train._reverified_step(graphed, observations)      # (the same verdict on every rank)
            else:
                loss = graphed(observations)        # local replay + the all-reduce of the loss
            all_reduce_gradients(parameters, group=group)
            optimizer.step()
            if problems:
                graphed.check()
                graphed, hip_graph = None, False    # (.grad held the eager evaluation's gradients for this step)
                optimizer.zero_grad(set_to_none=True)
            elif graphed.replays % _train._FLAG_CHECK_INTERVAL == 0:
                graphed.check()
            if callback is not None:
                callback(epoch, iteration, loss.clone(), *model_parts)
            continue
        first = observations[0]
        first = next(iter(first.values())) if isinstance(first, dict) else first
        global_batch_size = first.size(0) * world_size
        optimizer.zero_grad()
        loss = sharded_get_loss(observations, num_particles, algorithm, *model_parts,
                                global_batch_size=global_batch_size, rank=rank, world_size=world_size,
                                group=group)
        loss.backward()
        all_reduce_gradients(parameters, group=group)
        optimizer.step()
        if callback is not None:
            callback(epoch, iteration, loss, *model_parts)
    if graphed is not None:
        graphed.check()
