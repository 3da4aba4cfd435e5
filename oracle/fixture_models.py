"""ORACLE — TEST INFRASTRUCTURE ONLY.  The model definitions the golden fixtures are GENERATED with.

`oracle/capture_golden.py` runs the imported reference's `infer` / `get_loss` on these callables and
stores inputs, draws and outputs under tests/golden/.  They are FROZEN: plain `x @ W.t()` callables in
the reference's own style (test/models/lgssm.py:40, :52, :66-77), touched by no performance work —
the bench / test model `aesmc_amd.testing.models.LgssmNd` (which states the same distributions for the
fused kernels, caches the observation term over time, ...) is free to change without the fixtures
moving.  `tests/test_oracle.py::test_fixtures_regenerate_byte_for_byte` re-runs the generator and
compares every array with the committed files.

Parameter names and their seeded initial values equal LgssmNd's (A, C, W0, b0, Wx, Wy, b), so a fixture's
`param_model.*` arrays load into either class.
"""
import torch
import torch.nn as nn


class FixtureLgssmNd(nn.Module):
    """x_0 ~ N(0, I), x_t ~ N(A x_{t-1}, sx^2 I), y_t ~ N(C x_t, sy^2 I); proposal: a linear map of
    [x_{t-1}, y_t] with a fixed scale."""

    def __init__(self, dim, state, transition_scale=1.0, emission_scale=0.5, proposal_scale=0.7, seed=0,
                 dtype=torch.float32):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        eye = torch.eye(dim, dtype=torch.float64)
        self.dim = dim
        self._state = state
        self.register_buffer("transition_scale", torch.tensor(transition_scale, dtype=dtype))
        self.register_buffer("emission_scale", torch.tensor(emission_scale, dtype=dtype))
        self.register_buffer("proposal_scale", torch.tensor(proposal_scale, dtype=dtype))
        g1 = torch.randn(dim, dim, generator=gen, dtype=torch.float64)
        g2 = torch.randn(dim, dim, generator=gen, dtype=torch.float64)
        self.A = nn.Parameter((0.9 * eye + 0.01 * g1).to(dtype))
        self.C = nn.Parameter((eye + 0.01 * g2).to(dtype))
        self.register_buffer("loc0", torch.zeros(dim, dtype=dtype))
        self.register_buffer("scale0", torch.ones(dim, dtype=dtype))
        self.W0 = nn.Parameter((0.5 * eye + 0.01 * torch.randn(dim, dim, generator=gen, dtype=torch.float64)).to(dtype))
        self.b0 = nn.Parameter(torch.zeros(dim, dtype=dtype))
        self.Wx = nn.Parameter((0.45 * eye + 0.01 * torch.randn(dim, dim, generator=gen, dtype=torch.float64)).to(dtype))
        self.Wy = nn.Parameter((0.5 * eye + 0.01 * torch.randn(dim, dim, generator=gen, dtype=torch.float64)).to(dtype))
        self.b = nn.Parameter(torch.zeros(dim, dtype=dtype))

    def _tag(self, dist, mode_name):
        return self._state.set_batch_shape_mode(dist, getattr(self._state.BatchShapeMode, mode_name))

    def initial(self):
        return self._tag(torch.distributions.Normal(self.loc0, self.scale0), "NOT_EXPANDED")

    def transition(self, previous_latents=None, time=None, previous_observations=None):
        loc = previous_latents[-1] @ self.A.t()
        return self._tag(torch.distributions.Normal(loc, self.transition_scale), "FULLY_EXPANDED")

    def emission(self, latents=None, time=None, previous_observations=None):
        loc = latents[-1] @ self.C.t()
        return self._tag(torch.distributions.Normal(loc, self.emission_scale), "FULLY_EXPANDED")

    def proposal(self, previous_latents=None, time=None, observations=None):
        if time == 0:
            loc = observations[0] @ self.W0.t() + self.b0
            return self._tag(torch.distributions.Normal(loc, self.proposal_scale), "BATCH_EXPANDED")
        from_observation = observations[time] @ self.Wy.t() + self.b
        loc = previous_latents[-1] @ self.Wx.t() + from_observation.unsqueeze(1)
        return self._tag(torch.distributions.Normal(loc, self.proposal_scale), "FULLY_EXPANDED")

    @torch.no_grad()
    def simulate(self, num_timesteps, batch_size, seed=0):
        """Observations [T] x [B, d] drawn from the model itself (float64 host noise, cast to the model's dtype)."""
        dtype = self.A.dtype
        gen = torch.Generator().manual_seed(seed)

        def noise():
            return torch.randn(batch_size, self.dim, generator=gen, dtype=torch.float64).to(dtype)

        x = self.loc0 + self.scale0 * noise()
        observations = []
        for time in range(num_timesteps):
            if time > 0:
                x = x @ self.A.t() + self.transition_scale * noise()
            observations.append(x @ self.C.t() + self.emission_scale * noise())
        return observations
