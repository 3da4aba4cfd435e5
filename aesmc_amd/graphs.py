"""hipGraph capture of a whole ELBO evaluation.

At small per-step sizes (BASELINE.json configs[1]: 23 MB per gather) every kernel of the timestep
loop runs for 5-10 us while the Python host needs ~10 us to issue each one, so the eager loop is
host-bound.  `GraphedLoss` records one complete `losses.get_loss` — all T timesteps: the user's
callables, the K1-K4 kernels, the uniform upload, optionally `loss.backward()` — into a single
hipGraph (torch.cuda.CUDAGraph on ROCm) and replays it with one launch.

What stays faithful to the eager path
  * numpy's global RandomState is consumed exactly as before: T-1 blocks of
    `np.random.uniform(size=[batch_size, 1])` per evaluation, drawn on the host just before the
    replay and uploaded by one async copy ahead of it;
  * torch's generator advances per replay (PyTorch registers the philox offset with the graph);
  * the device status word (NaN log-weights, degenerate rows, ...) is read after the replay and
    raises the same exceptions as `inference.infer`.

Requirements on the model (they hold for any capture): callables must not synchronise with the
host (no .item() / .cpu() / float() of device tensors inside, no tensors made from host data per
call), shapes are fixed, parameters and observations stay at the same addresses (update them in
place).  Distributions in the reference's own style — `Normal(mult * x, 0.5)` with a Python-number
scale and PyTorch's default validate_args — are fine: inside `infer` the number is a cached device
constant and the checks run on the device (aesmc_amd/_syncfree.py).  Host-side state the callables
read is FROZEN at its value during the capture (see `train.train`'s frozen-callables contract).  With backward=True no autograd graph built EAGERLY from the
same parameters may still be alive when the capture starts (drop earlier `loss` tensors): PyTorch
would reuse that graph's gradient accumulators, which are bound to the eager stream, and the
capture could not be closed.

ROCm 7.0 caveat: captured memset nodes (PyTorch's reductions issue them) run out of stream order
under the runtime's graph fast path; `import aesmc_amd` switches that path off through
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 if the HIP runtime has not started yet (aesmc_amd/__init__.py).
The runtime reads the variable at its FIRST HIP call — torch.cuda.is_available() already is one — so
export it in the shell / launcher, or import aesmc_amd before anything touches the GPU.  A capture
WITH backward refuses to start when the variable is known not to be in effect, and in every case
verifies itself: the first `verify_replays` replays are compared with an eager evaluation on the
same inputs and random draws (the fault shows from the third or fourth replay on, in the gradients
only), and a mismatch raises.
"""
import contextlib
import gc
import warnings

import numpy as np
import torch
import torch.distributed as dist

from . import _kernels
from . import _philox
from . import distributed
from . import inference


class _StaticUniformFeed:
    """Uniform feed whose device block [T-1, B] lives as long as the graph (its resampling kernels
    read rows of it) and is refreshed from one of two pinned host blocks before each replay, so the
    host can draw the next evaluation's numbers while the previous upload is still in flight."""

    def __init__(self, batch_size, num_draws, device):
        self.batch_size = batch_size
        self.num_draws = max(num_draws, 1)
        self.host = [torch.empty((self.num_draws, batch_size), dtype=torch.float64, pin_memory=True)
                     for _ in range(2)]
        # the pinned blocks as the device addresses them: the upload is a copy kernel on the replay's stream (no DMA
        # engine hand-over in front of every replay)
        self.mapped = [inference._mapped_view(block, device) if device.type == "cuda" else None for block in self.host]
        self.host_rows = [block.numpy() for block in self.host]      # the same pinned memory: draws are written into it
        self.uploaded = [None, None]    # event after the last upload out of each pinned block
        self.turn = 0
        self.dev = torch.empty((self.num_draws, batch_size), dtype=torch.float64, device=device)
        self.cursor = 0

    def refill_and_upload(self):
        """Draws this evaluation's T-1 blocks — one `np.random.uniform(size=[B, 1])` per timestep, the reference's own
        consumption of the global RandomState (aesmc/inference.py:250; inside `distributed.shard_scope`: global batch,
        own rows kept), written straight into the pinned block — and enqueues the host -> device copy ahead of the
        replay, outside the graph (a pinned-memory copy node inside a capture trips the host allocator's event
        tracking).  Row by row on purpose: drawn as ONE [T-1, B, 1] array the block is a host allocation of
        hundreds of KB made and freed per replay, and on this stack each such allocation beside a process with >8 GB
        of device memory stalled the device (and writes to pinned memory) for 60-80 ms every few replays
        (`tools/graph_probe.py`) — what made replays above ~1.3M particles look slower than the eager loop."""
        slot = self.turn
        self.turn ^= 1
        if self.uploaded[slot] is not None:
            self.uploaded[slot].synchronize()   # two evaluations ago: long done
        rows = self.host_rows[slot]
        for row in range(self.num_draws):
            rows[row] = inference.draw_uniform_block(self.batch_size)
        if self.mapped[slot] is not None:
            self.dev.copy_(self.mapped[slot])
        else:
            self.dev.copy_(self.host[slot], non_blocking=True)
        event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(self.dev.device))
        self.uploaded[slot] = event

    def begin(self):
        self.cursor = 0

    def next(self):
        row = self.dev[self.cursor]
        self.cursor += 1
        return row


class GraphedLoss:
    """One hipGraph holding `losses.get_loss(observations, num_particles, algorithm, ...)` and,
    with backward=True, `loss.backward()` for the parameters of the four model parts.

        graphed = GraphedLoss(observations, K, 'aesmc', initial, transition, emission, proposal,
                              backward=True)
        for batch in data:
            loss = graphed(batch)        # copies `batch` into the static inputs, replays
            optimizer.step()             # .grad tensors are static and refreshed by every replay
    """

    def __init__(self, observations, num_particles, algorithm, initial, transition, emission,
                 proposal, backward=False, warmup=2, check_flags=True, shard=None, group=None,
                 verify_replays=4, guard_gradients=False, preserve_random_state=False):
        """`shard=(global_batch_size, rank, world_size)`: `observations` are this rank's rows of a
        batch sharded over the process group; the graph then holds the LOCAL share
        -sum_local(log Z_b) / global_batch_size (and its backward) and every call finishes with
        the one all-reduce of the loss (gradients: `distributed.all_reduce_gradients`).
        `preserve_random_state`: numpy's global RandomState and this device's torch generator are put back where they
        were when the constructor was entered, so the warm-up, the capture and the verification consume nothing: the
        first replay draws what the next eager evaluation would have drawn, and — a replay consuming both streams
        exactly as an eager evaluation does — a seeded run that switches to replays follows the eager loop's trajectory.
        `verify_replays` (backward only): replays compared with eager evaluations right after the
        capture (0 switches the check off).  Each costs one replay plus one full eager forward + backward and,
        while it runs, the eager autograd graph's memory beside the capture's private pool (at B=1024, K=4096,
        T=100, d=10: 24 GB and about 60 ms each); sharded graphs agree on the verdict across ranks.  `guard_gradients` (backward only): the captured backward
        ends by zeroing every gradient when the device status word is set (NaN log-weights, a
        degenerate row, ...), so an optimiser step taken before the host has read the word cannot
        poison the parameters."""
        import aesmc_amd
        if backward and aesmc_amd.HIPGRAPH_MEMSET_WORKAROUND in ("too-late", "preset:1"):
            raise RuntimeError(
                "aesmc_amd: refusing to capture a backward pass into a hipGraph: {}=0 is not in effect "
                "(status '{}': the HIP runtime had started before `import aesmc_amd`, or the variable is "
                "set to 1). On ROCm 7.0 captured memset nodes then run out of stream order and replayed "
                "GRADIENTS go wrong from the third or fourth replay on while losses stay right. Export "
                "{}=0 in the shell or launcher before Python starts, or import aesmc_amd before the first "
                "GPU call (torch.cuda.is_available() is one).".format(
                    aesmc_amd.HIPGRAPH_ENV, aesmc_amd.HIPGRAPH_MEMSET_WORKAROUND, aesmc_amd.HIPGRAPH_ENV))
        self.guard_gradients = bool(guard_gradients and backward)
        first = observations[0]
        self.shard = shard
        self.group = group
        self.global_batch = shard[0] if shard else first.size(0)
        if isinstance(first, dict):
            raise TypeError("GraphedLoss needs tensor observations (fixed addresses)")
        self.device = first.device
        if self.device.type != "cuda":
            raise RuntimeError("aesmc_amd: GraphedLoss captures a hipGraph and needs a HIP device, "
                               "got {}".format(self.device))
        entry_state = (torch.cuda.get_rng_state(self.device), np.random.get_state()) if preserve_random_state else None
        self.static_observations = [obs.clone() for obs in observations]
        self.check_flags = check_flags
        self.backward = backward
        self._args = (num_particles, algorithm, initial, transition, emission, proposal)
        self.parameters = []
        if backward:
            for part in (initial, transition, emission, proposal):
                owner = getattr(part, "__self__", part)   # bound method of an nn.Module, or a Module
                if isinstance(owner, torch.nn.Module):
                    for p in owner.parameters():
                        if p.requires_grad and all(p is not q for q in self.parameters):
                            self.parameters.append(p)
        num_timesteps = len(observations)
        self.feed = _StaticUniformFeed(first.size(0), num_timesteps - 1, self.device) \
            if algorithm == "aesmc" and num_timesteps > 1 else None

        kernels = _kernels.get()
        kernels.flags(self.device)                       # allocate the status word before capture
        _philox.verified(self.device)                    # (eagerly, with warmup=0 too: inside the capture nothing can be checked)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        self.noise = None
        with torch.cuda.stream(side):                    # warm-up on a side stream, as PyTorch asks
            for _ in range(warmup):
                generator = _philox._generator(self.device)
                before, counted = generator.get_offset(), sum(_philox.COUNTERS.values())
                self._evaluate(refill=True)
                advanced, counted = generator.get_offset() - before, sum(_philox.COUNTERS.values()) - counted
                if backward:
                    for p in self.parameters:
                        p.grad = None
            # Launches that draw their noise inside kernels need the generator state on the device during replays
            # (`_philox.GraphNoise`).  That is only sound when EVERY draw of the evaluation is one this package
            # places itself — PyTorch's own captured draws would not see what ours consumed: the generator must have
            # advanced by exactly what went through `_philox` in the last warm-up evaluation.  Otherwise the capture
            # runs with PyTorch drawing all the noise, as in rounds 1-2.
            if warmup > 0 and _philox.COUNTERS["reserved"] > 0 and advanced == counted and counted > 0:
                self.noise = _philox.GraphNoise(self.device)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        inference._raise_for_flags(kernels.read_flags(self.device))
        if backward:
            for p in self.parameters:                    # capture allocates fresh static .grad
                p.grad = None
        gc.collect()                                     # drop dead eager autograd graphs first
        self.graph = torch.cuda.CUDAGraph()
        self._refill()
        # With a process group alive, RCCL's watchdog thread may call the HIP runtime at any time;
        # "thread_local" keeps such calls from other threads from invalidating this capture.
        error_mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        try:
            with torch.cuda.graph(self.graph, capture_error_mode=error_mode):
                self.static_loss = self._evaluate(refill=False, capturing=True)
        except RuntimeError as error:
            if "captur" not in str(error).lower():
                raise
            raise RuntimeError(
                "aesmc_amd: the ELBO could not be captured into a hipGraph — something in the four "
                "callables talks to the host while the stream is capturing. Usual causes: "
                ".item() / float() / .cpu() / print of a device tensor, host-side control flow on tensor "
                "values, a tensor made from host data on every call (torch.tensor(..., device=...): keep it "
                "in a buffer on the device), or validation_mode 'eager' (PyTorch's own argument checks read "
                ".all() on the host; in the default 'deferred' mode Python-number parameters and default "
                "validate_args are fine). Original error: {}".format(error)) from error
        self.replays = 0
        if backward and verify_replays > 0:
            self._verify(verify_replays)
        if entry_state is not None:
            torch.cuda.set_rng_state(entry_state[0], self.device)
            np.random.set_state(entry_state[1])

    def accepts(self, observations):
        """Can `observations` be copied into the captured inputs (same count, shapes, dtypes, device)?"""
        if len(observations) != len(self.static_observations):
            return False
        return all(torch.is_tensor(fresh) and fresh.shape == static.shape and fresh.dtype == static.dtype and
                   fresh.device == static.device for static, fresh in zip(self.static_observations, observations))

    def _replay_against_eager(self, observations=None):
        """One replay and the same evaluation made eagerly from the random state the replay started with (same inputs,
        same draws).  Returns (problems, loss): `problems` lists what differs — loss to rtol 1e-5, every gradient to
        1e-4 of its norm (a user model may contain float atomics whose order differs from run to run; the faults this
        guards against are off by orders of magnitude); `loss` is the replay's.  Afterwards both random streams have
        moved by ONE evaluation; the parameters' `.grad` are the static tensors with the replay's values when nothing
        differs, and the EAGER evaluation's fresh tensors when something does (the caller drops the graph)."""
        params = self.parameters
        static = [p.grad for p in params]
        cuda_state, numpy_state = torch.cuda.get_rng_state(self.device), np.random.get_state()
        if observations is not None:
            for held, fresh in zip(self.static_observations, observations):
                held.copy_(fresh, non_blocking=True)
        self._replay()
        graph_loss = self.static_loss.clone()
        graph_grads = [None if g is None else g.clone() for g in static]
        after_replay = (torch.cuda.get_rng_state(self.device), np.random.get_state())
        for p in params:
            p.grad = None
        torch.cuda.set_rng_state(cuda_state, self.device)
        np.random.set_state(numpy_state)
        try:
            eager_loss = self._evaluate(refill=True)
        except torch.cuda.OutOfMemoryError:
            # the eager autograd graph does not fit beside the capture's private pool: the replay's results stand
            for p, grad in zip(params, static):
                p.grad = grad
            torch.cuda.empty_cache()
            torch.cuda.set_rng_state(after_replay[0], self.device)
            np.random.set_state(after_replay[1])
            warnings.warn("aesmc_amd: no memory for the eager evaluation a replay is compared with; the comparison "
                          "is skipped", RuntimeWarning)
            return [], graph_loss
        problems = []
        if not torch.allclose(graph_loss, eager_loss, rtol=1e-5, atol=1e-6, equal_nan=True):
            problems.append("loss {} vs {}".format(float(graph_loss), float(eager_loss)))
        for position, (p, got) in enumerate(zip(params, graph_grads)):
            want = p.grad
            if (got is None) != (want is None):
                problems.append("parameter {}: gradient present in only one of the two".format(position))
            elif got is not None:
                scale = float(torch.linalg.vector_norm(want.double()))
                worst = float(torch.linalg.vector_norm((got - want).double()))
                if not worst <= 1e-4 * scale + 1e-12:
                    problems.append("parameter {} {}: |difference| {:.3g} against a gradient of norm {:.3g}".format(
                        position, tuple(p.shape), worst, scale))
        if self.shard and dist.is_available() and dist.is_initialized():
            # every rank reaches the same verdict: a rank that went on alone would wait in the next all-reduce for good
            failed = torch.tensor([1.0 if problems else 0.0], device=self.device)
            dist.all_reduce(failed, op=dist.ReduceOp.MAX, group=self.group)
            if float(failed) > 0 and not problems:
                problems.append("another rank's replay did not reproduce its eager evaluation")
        if not problems:
            for p, grad in zip(params, static):
                p.grad = grad
        return problems, (graph_loss if not problems else eager_loss)

    def _verify(self, replays):
        """Replays the fresh graph `replays` times, each against an eager evaluation on the same inputs and random
        draws (`_replay_against_eager`); a mismatch raises.  Leaves both random streams where they were."""
        params = self.parameters
        start_cuda, start_numpy = torch.cuda.get_rng_state(self.device), np.random.get_state()
        static = [p.grad for p in params]
        try:
            for replay in range(replays):
                problems, _ = self._replay_against_eager()
                if problems:
                    import aesmc_amd
                    raise RuntimeError(
                        "aesmc_amd: replay {} of a freshly captured loss + backward hipGraph does not reproduce "
                        "the eager evaluation on the same inputs and random draws: {}. On ROCm 7.0 this is the "
                        "signature of captured memset nodes running out of stream order; {} must be 0 when the "
                        "HIP runtime starts (status here: '{}'). The graph must not be used.".format(
                            replay + 1, "; ".join(problems), aesmc_amd.HIPGRAPH_ENV,
                            aesmc_amd.HIPGRAPH_MEMSET_WORKAROUND))
        finally:
            for p, grad in zip(params, static):
                p.grad = grad
            torch.cuda.set_rng_state(start_cuda, self.device)
            np.random.set_state(start_numpy)
        inference._raise_for_flags(_kernels.get().read_flags(self.device))

    def reverify(self, observations):
        """A training step taken as a replay AND checked against the eager evaluation of the same minibatch with the same
        draws (backward graphs only): what detects that the callables' HOST-side state has moved since the capture — a
        Python-number coefficient annealed from a callback, `module.eval()`, a counter — which a replay cannot see.
        Returns (problems, loss); with problems the parameters' `.grad` hold the eager evaluation's gradients and the
        graph must be dropped.  Consumes both random streams exactly as one evaluation does."""
        if not self.backward:
            raise RuntimeError("aesmc_amd: GraphedLoss.reverify needs a graph captured with backward=True")
        problems, loss = self._replay_against_eager(observations)
        self.replays += 1
        if self.shard and dist.is_available() and dist.is_initialized():
            loss = loss.clone()
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
        return problems, loss

    def _shard_scope(self):
        return distributed.shard_scope(*self.shard) if self.shard else contextlib.nullcontext()

    def _refill(self):
        if self.feed is not None:
            with self._shard_scope():
                self.feed.refill_and_upload()

    def _replay(self):
        """One replay with the random inputs it needs in place: the uniforms (drawn and uploaded), the generator
        state the noise-drawing launches read, and the generator moved on by what they consume."""
        self._refill()      # stream order keeps the upload behind the previous replay's reads
        if self.noise is not None:
            self.noise.upload()
        self.graph.replay()
        if self.noise is not None:
            self.noise.advance()

    def _evaluate(self, refill, capturing=False):
        num_particles, algorithm, initial, transition, emission, proposal = self._args
        if refill:
            self._refill()
        if self.feed is not None:
            self.feed.begin()
        from . import settings
        noise_scope = switches = contextlib.nullcontext()
        if capturing:
            if self.noise is not None:
                noise_scope = _philox.graph_noise_scope(self.noise)
            else:
                switches = settings.override(kernel_noise=False)      # every draw through PyTorch's own captured generator state
        with switches:
            return self._evaluate_body(noise_scope, num_particles, algorithm, initial, transition, emission, proposal)

    def _evaluate_body(self, noise_scope, num_particles, algorithm, initial, transition, emission, proposal):
        with noise_scope, inference.uniform_feed(self.feed):
            result = inference.infer(
                {"iwae": "is", "aesmc": "smc"}[algorithm], self.static_observations, initial,
                transition, emission, proposal, num_particles, return_log_marginal_likelihood=True,
                return_latents=False, return_log_weight=False)
            # == losses.get_loss when unsharded: -mean_b(log Z_b)
            loss = -torch.sum(result["log_marginal_likelihood"]) / self.global_batch
            if self.backward:
                loss.backward()
                if self.guard_gradients:
                    # on the device, no host sync: a step flagged by the kernels contributes nothing
                    healthy = (_kernels.get().flags(self.device) == 0).reshape(())      # (0-dim: broadcasts into scalar parameters too)
                    for p in self.parameters:
                        if p.grad is not None:
                            p.grad.copy_(torch.where(healthy, p.grad, torch.zeros_like(p.grad)))
        return loss.detach()

    def poll(self):
        """The device status word WITHOUT a synchronisation: after each replay a 4-byte copy of the word into pinned host
        memory is enqueued behind it (two slots, an event each); this call looks at the copies that have landed and, if one
        of them is non-zero, does the synchronising `check()` — which raises what `inference.infer` would have raised.  A
        flagged replay therefore surfaces one or two replays later instead of at the next `check()` (`train()` checks every
        `_FLAG_CHECK_INTERVAL` replays), at the cost of one tiny copy per replay and no stall."""
        probes = getattr(self, "_flag_probes", None)
        if probes is None:
            host = torch.zeros(2, dtype=torch.int32).pin_memory()
            probes = self._flag_probes = {"host": host, "events": [None, None], "turn": 0}
        word = _kernels.get().flags(self.device)
        for slot in (0, 1):
            event = probes["events"][slot]
            if event is not None and event.query():
                probes["events"][slot] = None
                if int(probes["host"][slot]) != 0:
                    probes["events"] = [None, None]
                    self.check()
        slot = probes["turn"]
        if probes["events"][slot] is None:      # (a copy still in flight keeps its slot: skip this replay's probe)
            probes["host"][slot:slot + 1].copy_(word, non_blocking=True)
            event = torch.cuda.Event()
            event.record(torch.cuda.current_stream(self.device))
            probes["events"][slot] = event
            probes["turn"] = slot ^ 1

    def check(self):
        """Synchronising read of the device status word; raises what `inference.infer` would have
        raised for any replay since the last check (with check_flags=False call it yourself, e.g.
        once per logging interval, and the replays run back to back without a host sync)."""
        inference._raise_for_flags(_kernels.get().read_flags(self.device))

    def __call__(self, observations=None):
        """Replays the graph (after copying `observations`, if given, into the static inputs) and
        returns the static loss tensor; parameter gradients, when captured, are in `.grad`."""
        if observations is not None:
            for static, fresh in zip(self.static_observations, observations):
                static.copy_(fresh, non_blocking=True)
        self._replay()
        self.replays += 1
        if self.check_flags:
            self.check()
        if self.shard and dist.is_available() and dist.is_initialized():
            total = self.static_loss.clone()
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group)
            return total
        return self.static_loss
