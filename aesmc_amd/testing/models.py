"""State-space models written against the initial / transition / emission / proposal contract.

Used by the parity tests, `__graft_entry__.smoke()` and `bench.py`.  Each model takes the `state`
module it should tag distributions with, so the same definition runs under this package and —
inside oracle/capture_golden.py only — under the imported reference when fixtures are generated.

  * Lgssm1d*  : counterpart of the reference's test/models/lgssm.py (the config-1 plumbing model),
                including its quirk that the proposal uses scale_0 at every time step
                (test/models/lgssm.py:71).
  * Gaussian* : counterpart of the reference's test/models/gaussian.py (one-step IWAE model).
                Both families are stated in the reference's own style ON PURPOSE — Python-number scales,
                PyTorch's default `validate_args`, the `cat` / `view` proposal, a host-resident `std` —
                because that is what a user who switches packages brings along: `inference.infer` makes
                such callables sync-free (`_syncfree`), so they run in the eager loop without a copy or a
                host read per distribution and can be captured into a hipGraph (`train(hip_graph=...)`).
  * LgssmNd   : d-dimensional linear-Gaussian SSM of SURVEY.md section 8(d) (bench workloads).
  * NonlinearSsm : tanh transition + 2-layer MLP proposal (config 4 of BASELINE.json).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import state as _default_state


# ------------------------------------------------------------------------------------------------
# 1-D linear-Gaussian SSM (reference test/models/lgssm.py)
# ------------------------------------------------------------------------------------------------
class Lgssm1dInitial:
    def __init__(self, loc, scale):
        self.loc, self.scale = loc, scale

    def __call__(self):
        return torch.distributions.Normal(self.loc, self.scale)


class Lgssm1dTransition(nn.Module):
    def __init__(self, init_mult, scale, state=_default_state):
        super().__init__()
        self.mult = nn.Parameter(torch.tensor(float(init_mult)))
        self.scale = scale
        self._state = state

    def forward(self, previous_latents=None, time=None, previous_observations=None):
        dist = torch.distributions.Normal(self.mult * previous_latents[-1], self.scale)
        return self._state.set_batch_shape_mode(dist, self._state.BatchShapeMode.FULLY_EXPANDED)


class Lgssm1dEmission(nn.Module):
    def __init__(self, init_mult, scale, state=_default_state):
        super().__init__()
        self.mult = nn.Parameter(torch.tensor(float(init_mult)))
        self.scale = scale
        self._state = state

    def forward(self, latents=None, time=None, previous_observations=None):
        dist = torch.distributions.Normal(self.mult * latents[-1], self.scale)
        return self._state.set_batch_shape_mode(dist, self._state.BatchShapeMode.FULLY_EXPANDED)


class Lgssm1dProposal(nn.Module):
    def __init__(self, scale_0, scale_t, state=_default_state):
        super().__init__()
        self.scale_0, self.scale_t = scale_0, scale_t
        self.lin_0 = nn.Linear(1, 1)
        self.lin_t = nn.Linear(2, 1)
        self._state = state

    def forward(self, previous_latents=None, time=None, observations=None):
        modes = self._state.BatchShapeMode
        if time == 0:
            loc = self.lin_0(observations[0].unsqueeze(-1)).squeeze(-1)
            return self._state.set_batch_shape_mode(
                torch.distributions.Normal(loc=loc, scale=self.scale_0), modes.BATCH_EXPANDED)
        x_prev = previous_latents[-1]
        num_particles = x_prev.shape[1]
        y_now = observations[time].view(-1, 1, 1).expand(-1, num_particles, 1)
        features = torch.cat([x_prev.unsqueeze(-1), y_now], dim=2).view(-1, 2)
        loc = self.lin_t(features).squeeze(-1).view(-1, num_particles)
        # scale_0 at t > 0 too: the reference never reads scale_t (test/models/lgssm.py:71)
        return self._state.set_batch_shape_mode(
            torch.distributions.Normal(loc=loc, scale=self.scale_0), modes.FULLY_EXPANDED)


class ReferenceLgssm1d(nn.Module):
    """The four reference-style 1-D classes above as one module with the bench's model interface (`initial`,
    `transition`, `emission`, `proposal`, `simulate`): x_0 ~ N(0, 1), x_t ~ N(a x_{t-1}, sx^2), y_t ~ N(c x_t, sy^2).
    Nothing is adapted for the device: Python-number scales, default `validate_args`, `cat` / `view` proposal.  The
    proposal's two Linear layers are set to the model's locally optimal proposal (what training converges towards), its
    one scale — the reference reads `scale_0` at every step, test/models/lgssm.py:71 — to that proposal's."""

    def __init__(self, transition_mult=0.9, emission_mult=1.0, transition_scale=1.0, emission_scale=0.5,
                 state=_default_state, **_):
        super().__init__()
        a, c, sx2, sy2 = transition_mult, emission_mult, transition_scale ** 2, emission_scale ** 2
        spread = 1.0 / (1.0 / sx2 + c * c / sy2)
        self.initial = Lgssm1dInitial(0.0, 1.0)
        self.transition = Lgssm1dTransition(a, transition_scale, state=state)
        self.emission = Lgssm1dEmission(c, emission_scale, state=state)
        self.proposal = Lgssm1dProposal(float(np.sqrt(spread)), float(np.sqrt(spread)), state=state)
        with torch.no_grad():
            spread0 = 1.0 / (1.0 + c * c / sy2)
            self.proposal.lin_0.weight.fill_(spread0 * c / sy2)
            self.proposal.lin_0.bias.zero_()
            self.proposal.lin_t.weight.copy_(torch.tensor([[spread * a / sx2, spread * c / sy2]]))
            self.proposal.lin_t.bias.zero_()
        self._numbers = (a, c, transition_scale, emission_scale)

    @torch.no_grad()
    def simulate(self, num_timesteps, batch_size, seed=0):
        """Observations [T] x [B] drawn from the model (float64 host noise, cast), on the parameters' device."""
        a, c, sx, sy = self._numbers
        device = self.transition.mult.device
        gen = torch.Generator().manual_seed(seed)

        def noise():
            return torch.randn(batch_size, generator=gen, dtype=torch.float64)

        x = noise()
        observations = []
        for time in range(num_timesteps):
            if time > 0:
                x = a * x + sx * noise()
            observations.append((c * x + sy * noise()).to(device=device, dtype=torch.float32))
        return observations


# ------------------------------------------------------------------------------------------------
# One-step Gaussian model (reference test/models/gaussian.py)
# ------------------------------------------------------------------------------------------------
class GaussianPrior(nn.Module):
    def __init__(self, init_mean, std):
        super().__init__()
        self.mean = nn.Parameter(torch.tensor(init_mean, dtype=torch.float))
        # a plain attribute, as the reference has it (test/models/gaussian.py:10): NOT a buffer, so `.to(device)`
        # leaves it on the host and `Normal(loc=<device>, scale=<host 0-dim>)` is what the library is handed
        self.std = torch.tensor(std, dtype=torch.float)

    def forward(self):
        return torch.distributions.Normal(loc=self.mean, scale=self.std)


class GaussianLikelihood(nn.Module):
    def __init__(self, init_std):
        super().__init__()
        self.log_std = nn.Parameter(torch.log(torch.tensor(init_std, dtype=torch.float)))

    def forward(self, latents=None, time=None, previous_observations=None):
        return torch.distributions.Normal(loc=latents[-1], scale=torch.exp(self.log_std))


class GaussianInferenceNetwork(nn.Module):
    def __init__(self, init_mult, init_bias, init_std):
        super().__init__()
        self.mult = nn.Parameter(torch.tensor(init_mult, dtype=torch.float))
        self.bias = nn.Parameter(torch.tensor(init_bias, dtype=torch.float))
        self.log_std = nn.Parameter(torch.log(torch.tensor(init_std, dtype=torch.float)))

    def forward(self, previous_latents=None, time=None, observations=None):
        return torch.distributions.Normal(loc=self.mult * observations[0] + self.bias,
                                          scale=torch.exp(self.log_std))


# ------------------------------------------------------------------------------------------------
# d-dimensional linear-Gaussian SSM; latents are [B, K, d]
# ------------------------------------------------------------------------------------------------
class LgssmNd(nn.Module):
    """x_0 ~ N(0, I), x_t ~ N(A x_{t-1}, sx^2 I), y_t ~ N(C x_t, sy^2 I); proposal is a linear
    map of [x_{t-1}, y_t] with a fixed scale.  The four contract callables are the bound methods
    `initial`, `transition`, `emission`, `proposal`."""

    def __init__(self, dim, transition_scale=1.0, emission_scale=0.5, proposal_scale=0.7, seed=0,
                 dtype=torch.float32, state=_default_state, validate_args=None, affine=False, defer_draw=None):
        super().__init__()
        self.validate_args = validate_args  # None = PyTorch default; False skips per-call host syncs
        # defer_draw (default: on with affine): the proposal leaves its draw to the launch that weighs the
        # step (K15) — this model's transition and emission never read the newest latent's values
        self.defer_draw = bool(affine) if defer_draw is None else bool(defer_draw)
        # affine=True: the callables return aesmc_amd.linear_gaussian.AffineNormal(source, weight, ...)
        # in place of Normal(source @ weight.T + ..., ...) — the same distributions, their locations
        # evaluated inside the sampling / weighting kernels instead of by matmuls beforehand
        self.affine = bool(affine)
        gen = torch.Generator().manual_seed(seed)
        eye = torch.eye(dim, dtype=torch.float64)
        self.dim = dim
        self._state = state
        # scales are device buffers, not Python numbers: Normal(loc, 0.7) would upload the scalar on
        # every call (a host-to-device copy, which also cannot be captured into a hipGraph)
        self.register_buffer("transition_scale", torch.tensor(transition_scale, dtype=dtype))
        self.register_buffer("emission_scale", torch.tensor(emission_scale, dtype=dtype))
        self.register_buffer("proposal_scale", torch.tensor(proposal_scale, dtype=dtype))
        g1 = torch.randn(dim, dim, generator=gen, dtype=torch.float64)
        g2 = torch.randn(dim, dim, generator=gen, dtype=torch.float64)
        self.A = nn.Parameter((0.9 * eye + 0.01 * g1).to(dtype))
        self.C = nn.Parameter((eye + 0.01 * g2).to(dtype))
        self.register_buffer("loc0", torch.zeros(dim, dtype=dtype))
        self.register_buffer("scale0", torch.ones(dim, dtype=dtype))
        # learned-proposal stand-in: mean = Wx x_{t-1} + Wy y_t + b (t = 0: W0 y_0 + b0)
        self.W0 = nn.Parameter((0.5 * eye + 0.01 * torch.randn(dim, dim, generator=gen,
                                                               dtype=torch.float64)).to(dtype))
        self.b0 = nn.Parameter(torch.zeros(dim, dtype=dtype))
        self.Wx = nn.Parameter((0.45 * eye + 0.01 * torch.randn(dim, dim, generator=gen,
                                                                dtype=torch.float64)).to(dtype))
        self.Wy = nn.Parameter((0.5 * eye + 0.01 * torch.randn(dim, dim, generator=gen,
                                                               dtype=torch.float64)).to(dtype))
        self.b = nn.Parameter(torch.zeros(dim, dtype=dtype))

    def _tag(self, dist, mode_name):
        return self._state.set_batch_shape_mode(dist, getattr(self._state.BatchShapeMode, mode_name))

    def _normal(self, loc, scale):
        return torch.distributions.Normal(loc, scale, validate_args=self.validate_args)

    def initial(self):
        return self._tag(self._normal(self.loc0, self.scale0), "NOT_EXPANDED")

    def _affine_normal(self, source, weight, scale, offset=None, defer_draw=False):
        from ..linear_gaussian import AffineNormal
        return AffineNormal(source, weight, scale, offset=offset, validate_args=self.validate_args,
                            defer_draw=defer_draw)

    def transition(self, previous_latents=None, time=None, previous_observations=None):
        if self.affine:
            return self._tag(self._affine_normal(previous_latents[-1], self.A, self.transition_scale),
                             "FULLY_EXPANDED")
        loc = previous_latents[-1] @ self.A.t()
        return self._tag(self._normal(loc, self.transition_scale), "FULLY_EXPANDED")

    def emission(self, latents=None, time=None, previous_observations=None):
        if self.affine:
            return self._tag(self._affine_normal(latents[-1], self.C, self.emission_scale), "FULLY_EXPANDED")
        loc = latents[-1] @ self.C.t()
        return self._tag(self._normal(loc, self.emission_scale), "FULLY_EXPANDED")

    def proposal(self, previous_latents=None, time=None, observations=None):
        if time == 0:
            loc = observations[0] @ self.W0.t() + self.b0
            return self._tag(self._normal(loc, self.proposal_scale), "BATCH_EXPANDED")
        from_observation = self._observation_terms(observations, time)[time]   # [B, d]: shared by a row's particles
        if self.affine:
            return self._tag(self._affine_normal(previous_latents[-1], self.Wx, self.proposal_scale,
                                                 offset=from_observation, defer_draw=self.defer_draw),
                             "FULLY_EXPANDED")
        loc = previous_latents[-1] @ self.Wx.t() + from_observation.unsqueeze(1)
        return self._tag(self._normal(loc, self.proposal_scale), "FULLY_EXPANDED")

    def _observation_terms(self, observations, time):
        """Wy y_t + b for every timestep in ONE matmul: the proposal is handed the whole observation
        sequence at each step, so the first step that needs the term (time 1 of every `infer`) computes
        it for all of them and the later steps of that run take their row — two small launches per timestep
        less.  Never kept across runs (its autograd graph belongs to the run that made it)."""
        cached = getattr(self, "_obs_terms", None)
        if time == 1 or cached is None or cached[0] is not observations:
            stacked = observations if torch.is_tensor(observations) else torch.stack(list(observations))
            # one row per timestep as views of ONE unbind: its backward is a single stack of the rows' gradients, where
            # indexing the [T, B, d] tensor at every step costs a zero fill, a copy and an accumulation of that tensor per
            # step (18 us of the backward's 360 per timestep at the bench shape)
            cached = (observations, (stacked @ self.Wy.t() + self.b).unbind(0))
        # dropped with the last step: a tensor that outlives the run keeps the run's autograd graph (and
        # the parameters' AccumulateGrad nodes) alive, which a later hipGraph capture of a backward
        # pass cannot tolerate (aesmc_amd/graphs.py)
        self._obs_terms = None if time + 1 >= len(observations) else cached
        return cached[1]

    @torch.no_grad()
    def tune_proposal(self):
        """Sets the proposal's parameters to the model's locally optimal proposal in closed form:
        p(x_t | x_{t-1}, y_t) = N(S (A x_{t-1} / sx^2 + C^T y_t / sy^2), S) with
        S = (I / sx^2 + C^T C / sy^2)^-1 (time 0: the posterior of x_0 given y_0, same form with the
        prior N(loc0, scale0^2 I)).  The proposal family here has one scalar scale, so S is
        represented by sqrt(mean diag S) — C is close to the identity, S close to a multiple of
        it.  What a trained proposal converges towards; gives a healthy particle system (the
        untrained stand-in of SURVEY.md 8(d) collapses to ~14 % surviving ancestors per step)."""
        dtype, device = self.A.dtype, self.A.device
        A, C = self.A.double().cpu(), self.C.double().cpu()
        eye = torch.eye(self.dim, dtype=torch.float64)
        sx2 = float(self.transition_scale) ** 2
        sy2 = float(self.emission_scale) ** 2
        S = torch.linalg.inv(eye / sx2 + C.t() @ C / sy2)
        self.Wx.copy_((S @ A / sx2).to(device, dtype))
        self.Wy.copy_((S @ C.t() / sy2).to(device, dtype))
        self.b.zero_()
        s02 = self.scale0.double().cpu() ** 2
        S0 = torch.linalg.inv(torch.diag(1.0 / s02) + C.t() @ C / sy2)
        self.W0.copy_((S0 @ C.t() / sy2).to(device, dtype))
        self.b0.copy_((S0 @ (self.loc0.double().cpu() / s02)).to(device, dtype))
        self.proposal_scale.fill_(float(torch.sqrt(torch.diagonal(S).mean())))
        return self

    @torch.no_grad()
    def simulate(self, num_timesteps, batch_size, seed=0):
        """Observations [T] x [B, d] drawn from the model itself on its own device."""
        device, dtype = self.A.device, self.A.dtype
        gen = torch.Generator().manual_seed(seed)

        def noise():
            return torch.randn(batch_size, self.dim, generator=gen, dtype=torch.float64).to(device, dtype)

        x = self.loc0 + self.scale0 * noise()
        observations = []
        for time in range(num_timesteps):
            if time > 0:
                x = x @ self.A.t() + self.transition_scale * noise()
            observations.append(x @ self.C.t() + self.emission_scale * noise())
        return observations


class NonlinearSsm(nn.Module):
    """x_t ~ N(tanh(A x_{t-1}), sx^2 I), y_t ~ N(C x_t, sy^2 I), proposal = 2-layer MLP of
    [x_{t-1}, y_t] (config 4 of BASELINE.json: 'nonlinear SSM with learned proposal net')."""

    def __init__(self, dim, hidden=64, transition_scale=1.0, emission_scale=0.5, proposal_scale=0.7,
                 seed=0, dtype=torch.float32, state=_default_state, validate_args=None, fused=False):
        super().__init__()
        self.validate_args = validate_args
        # fused=True: the d x d maps through aesmc_amd.linear_gaussian (particle_affine / AffineNormal: kernel K8) and the
        # proposal net through particle_mlp (kernels K13 / K13b) instead of PyTorch matmuls
        self.fused = bool(fused)
        gen = torch.Generator().manual_seed(seed)
        eye = torch.eye(dim, dtype=torch.float64)
        self.dim = dim
        self._state = state
        # scales are device buffers, not Python numbers: Normal(loc, 0.7) would upload the scalar on
        # every call (a host-to-device copy, which also cannot be captured into a hipGraph)
        self.register_buffer("transition_scale", torch.tensor(transition_scale, dtype=dtype))
        self.register_buffer("emission_scale", torch.tensor(emission_scale, dtype=dtype))
        self.register_buffer("proposal_scale", torch.tensor(proposal_scale, dtype=dtype))
        self.A = nn.Parameter((0.9 * eye + 0.05 * torch.randn(dim, dim, generator=gen,
                                                              dtype=torch.float64)).to(dtype))
        self.C = nn.Parameter((eye + 0.01 * torch.randn(dim, dim, generator=gen,
                                                        dtype=torch.float64)).to(dtype))
        self.register_buffer("loc0", torch.zeros(dim, dtype=dtype))
        self.register_buffer("scale0", torch.ones(dim, dtype=dtype))
        torch.manual_seed(seed)
        self.net0 = nn.Sequential(nn.Linear(dim, hidden), nn.Tanh(), nn.Linear(hidden, dim)).to(dtype)
        self.net = nn.Sequential(nn.Linear(2 * dim, hidden), nn.Tanh(), nn.Linear(hidden, dim)).to(dtype)

    def _tag(self, dist, mode_name):
        return self._state.set_batch_shape_mode(dist, getattr(self._state.BatchShapeMode, mode_name))

    def _normal(self, loc, scale):
        return torch.distributions.Normal(loc, scale, validate_args=self.validate_args)

    def initial(self):
        return self._tag(self._normal(self.loc0, self.scale0), "NOT_EXPANDED")

    def transition(self, previous_latents=None, time=None, previous_observations=None):
        if self.fused:      # tanh(A x) from K8's own launch
            from ..linear_gaussian import particle_affine
            loc = particle_affine(previous_latents[-1], self.A, activation="tanh")
        else:
            loc = torch.tanh(previous_latents[-1] @ self.A.t())
        return self._tag(self._normal(loc, self.transition_scale), "FULLY_EXPANDED")

    def emission(self, latents=None, time=None, previous_observations=None):
        if self.fused:
            from ..linear_gaussian import AffineNormal
            return self._tag(AffineNormal(latents[-1], self.C, self.emission_scale, validate_args=self.validate_args),
                             "FULLY_EXPANDED")
        return self._tag(self._normal(latents[-1] @ self.C.t(), self.emission_scale),
                         "FULLY_EXPANDED")

    def _proposal_loc(self, x_prev, y_now):
        """The net of [x_{t-1}, y_t]: fused, its first layer splits into the particles' columns (inside kernel K13) and
        the observation's columns + bias (one [B,H] row offset, a small product)."""
        first, second = self.net[0], self.net[2]
        if self.fused:
            from ..linear_gaussian import particle_mlp
            from_observation = y_now @ first.weight[:, self.dim:].t() + first.bias
            return particle_mlp(x_prev, first.weight[:, :self.dim], from_observation, second.weight, second.bias)
        expanded = y_now.unsqueeze(1).expand(-1, x_prev.size(1), -1)
        return self.net(torch.cat([x_prev, expanded], dim=2))

    def proposal(self, previous_latents=None, time=None, observations=None):
        if time == 0:
            return self._tag(self._normal(self.net0(observations[0]), self.proposal_scale),
                             "BATCH_EXPANDED")
        loc = self._proposal_loc(previous_latents[-1], observations[time])
        return self._tag(self._normal(loc, self.proposal_scale), "FULLY_EXPANDED")

    @torch.no_grad()
    def simulate(self, num_timesteps, batch_size, seed=0):
        device, dtype = self.A.device, self.A.dtype
        gen = torch.Generator().manual_seed(seed)

        def noise():
            return torch.randn(batch_size, self.dim, generator=gen, dtype=torch.float64).to(device, dtype)

        x = self.loc0 + self.scale0 * noise()
        observations = []
        for time in range(num_timesteps):
            if time > 0:
                x = torch.tanh(x @ self.A.t()) + self.transition_scale * noise()
            observations.append(x @ self.C.t() + self.emission_scale * noise())
        return observations


class LearnedScaleSsm(NonlinearSsm):
    """The nonlinear SSM with the scales a learned-proposal model has: the proposal network outputs
    location AND scale per particle ([B,K,d] tensors), the transition noise is a learned
    per-dimension vector, the emission noise stays a scalar buffer."""

    def __init__(self, dim, hidden=64, seed=0, dtype=torch.float32, state=_default_state, validate_args=None, **kw):
        super().__init__(dim, hidden=hidden, seed=seed, dtype=dtype, state=state, validate_args=validate_args, **kw)
        torch.manual_seed(seed + 1)
        self.net0 = nn.Sequential(nn.Linear(dim, hidden), nn.Tanh(), nn.Linear(hidden, 2 * dim)).to(dtype)
        self.net = nn.Sequential(nn.Linear(2 * dim, hidden), nn.Tanh(), nn.Linear(hidden, 2 * dim)).to(dtype)
        self.transition_log_scale = nn.Parameter(torch.zeros(dim, dtype=dtype))

    def transition(self, previous_latents=None, time=None, previous_observations=None):
        loc = torch.tanh(previous_latents[-1] @ self.A.t())
        return self._tag(self._normal(loc, torch.exp(self.transition_log_scale)), "FULLY_EXPANDED")

    def _split(self, raw):
        loc, raw_scale = raw[..., :self.dim], raw[..., self.dim:]
        return loc.contiguous(), (torch.nn.functional.softplus(raw_scale) + 0.1).contiguous()

    def proposal(self, previous_latents=None, time=None, observations=None):
        if time == 0:
            return self._tag(self._normal(*self._split(self.net0(observations[0]))), "BATCH_EXPANDED")
        x_prev = previous_latents[-1]
        y_now = observations[time].unsqueeze(1).expand(-1, x_prev.size(1), -1)
        return self._tag(self._normal(*self._split(self.net(torch.cat([x_prev, y_now], dim=2)))),
                         "FULLY_EXPANDED")


class GaussianIwae(nn.Module):
    """The one-step Gaussian model above bundled as one module (config 3 of BASELINE.json: IWAE,
    T = 1, no resampling): x ~ N(mean, 1), y ~ N(x, obs_std), q(x | y) = N(mult y + bias, q_std).
    `transition` is None, as in the reference's test/test_losses.py:45."""

    transition = None

    def __init__(self, prior_mean=0.3, obs_std=0.8, q_mult=0.6, q_bias=0.1, q_std=0.9,
                 dtype=torch.float32, state=_default_state, validate_args=None, **_):
        super().__init__()
        self._state = state
        self.validate_args = validate_args
        self.mean = nn.Parameter(torch.tensor(prior_mean, dtype=dtype))
        self.register_buffer("prior_std", torch.tensor(1.0, dtype=dtype))
        self.obs_log_std = nn.Parameter(torch.log(torch.tensor(obs_std, dtype=dtype)))
        self.q_mult = nn.Parameter(torch.tensor(q_mult, dtype=dtype))
        self.q_bias = nn.Parameter(torch.tensor(q_bias, dtype=dtype))
        self.q_log_std = nn.Parameter(torch.log(torch.tensor(q_std, dtype=dtype)))

    def _normal(self, loc, scale, mode_name):
        dist = torch.distributions.Normal(loc, scale, validate_args=self.validate_args)
        return self._state.set_batch_shape_mode(dist, getattr(self._state.BatchShapeMode, mode_name))

    def initial(self):
        return self._normal(self.mean, self.prior_std, "NOT_EXPANDED")

    def emission(self, latents=None, time=None, previous_observations=None):
        return self._normal(latents[-1], torch.exp(self.obs_log_std), "FULLY_EXPANDED")

    def proposal(self, previous_latents=None, time=None, observations=None):
        return self._normal(self.q_mult * observations[0] + self.q_bias, torch.exp(self.q_log_std),
                            "BATCH_EXPANDED")

    @torch.no_grad()
    def simulate(self, num_timesteps, batch_size, seed=0):
        gen = torch.Generator().manual_seed(seed)
        device, dtype = self.mean.device, self.mean.dtype
        x = self.mean + torch.randn(batch_size, generator=gen, dtype=torch.float64).to(device, dtype)
        noise = torch.randn(batch_size, generator=gen, dtype=torch.float64).to(device, dtype)
        return [x + torch.exp(self.obs_log_std) * noise]


def kalman_log_likelihood(model, observations):
    """Exact log p(y_{1:T}) per batch row of an LgssmNd by the Kalman filter (float64, host):
    an independent statistical oracle — E[exp(log Z_hat)] of IS/SMC equals exp of this."""
    A = model.A.detach().double().cpu().numpy()
    C = model.C.detach().double().cpu().numpy()
    d = A.shape[0]
    Q = np.eye(d) * float(model.transition_scale) ** 2
    R = np.eye(d) * float(model.emission_scale) ** 2
    ys = [y.detach().double().cpu().numpy() for y in observations]
    batch = ys[0].shape[0]
    out = np.zeros(batch)
    for b in range(batch):
        mean, cov = np.zeros(d), np.eye(d)
        for t, y in enumerate(ys):
            if t > 0:
                mean, cov = A @ mean, A @ cov @ A.T + Q
            s = C @ cov @ C.T + R
            resid = y[b] - C @ mean
            sol = np.linalg.solve(s, resid)
            out[b] += -0.5 * (resid @ sol + np.linalg.slogdet(s)[1] + d * np.log(2 * np.pi))
            gain = cov @ C.T @ np.linalg.inv(s)
            mean, cov = mean + gain @ resid, cov - gain @ C @ cov
    return out
