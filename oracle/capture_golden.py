"""ORACLE — TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by importing the reference
from /root/reference (read-only, this container only) and recording, for each case:

  inputs   : observations, model parameters, every standard-normal block and every numpy uniform
             block the reference drew (so the run can be replayed bit-for-bit on any device);
  outputs  : per-step log_weights, ancestral_indices, original_latents, genealogy-resampled
             latents, log_weight, last_latent, log_marginal_likelihood, the loss of
             aesmc.losses.get_loss and its parameter gradients.

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/capture_golden.py
Nothing of the reference's source is stored: fixtures hold arrays and a JSON description only.
The only missing dependency of the reference here is `pykalman` (imported by its
test/models/lgssm.py for a Kalman smoother the fixtures do not use); it is stubbed.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, REFERENCE)
sys.dont_write_bytecode = True
sys.modules.setdefault("pykalman", types.ModuleType("pykalman"))

import aesmc as ref  # noqa: E402  (the reference)
import aesmc.state  # noqa: E402,F401
import aesmc.math  # noqa: E402,F401

from aesmc_amd.testing import replay  # noqa: E402
from oracle import fixture_models  # noqa: E402  (frozen: see its docstring)
from oracle import kernel_oracle  # noqa: E402


def _load_reference_module(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE, relpath))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


ref_lgssm = _load_reference_module("ref_test_models_lgssm", "test/models/lgssm.py")
ref_gaussian = _load_reference_module("ref_test_models_gaussian", "test/models/gaussian.py")


def _np(t):
    return t.detach().cpu().numpy().copy()


def _named_params(parts):
    out = {}
    for part_name, part in parts.items():
        if isinstance(part, torch.nn.Module):
            for name, p in part.named_parameters():
                out["{}.{}".format(part_name, name)] = p
    return out


def _margin(log_weights, uniforms):
    """Smallest |pos_k - c_j| over all resampling steps (float64 CDF): how far every comparison
    of the resampler is from flipping.  Stored so tests know which equalities are robust."""
    worst = np.inf
    for lw, u in zip(log_weights[:-1], uniforms):
        lw = lw.astype(np.float64)
        w = np.exp(lw - lw.max(axis=1, keepdims=True))
        c = np.cumsum(w, axis=1)
        c = c / c[:, -1:]
        K = lw.shape[1]
        pos = (u.reshape(-1, 1) + np.arange(K)) / K
        for b in range(lw.shape[0]):
            j = np.searchsorted(c[b], pos[b])
            lo = np.abs(pos[b] - c[b][np.clip(j - 1, 0, K - 1)])
            hi = np.abs(pos[b] - c[b][np.clip(j, 0, K - 1)])
            worst = min(worst, lo.min(), hi.min())
    return float(worst)


def _flips_per_step(log_weights, uniforms, indices):
    """Per resampling step: how many of the reference's indices differ from the float64-CDF contract
    this library implements (oracle/kernel_oracle.py) on the reference's own log-weights."""
    out = []
    for lw, u, want in zip(log_weights, uniforms, indices):
        mine, _ = kernel_oracle.ancestor_index(lw, np.asarray(u, dtype=np.float64).reshape(-1))
        out.append(int((mine != want).sum()))
    return out


def capture_infer(name, meta, parts, observations, num_particles, algorithm, light=False):
    """Runs the reference's infer (everything returned) and get_loss + backward under one tape.
    `light`: large cases keep every input and the per-step log-weights / indices but not the T
    latent tensors (twice [T,B,K,d])."""
    inference_algorithm = {"iwae": "is", "aesmc": "smc"}[algorithm]
    smc = inference_algorithm == "smc"
    with replay.record() as tape:
        result = ref.inference.infer(
            inference_algorithm, observations, parts["initial"], parts["transition"],
            parts["emission"], parts["proposal"], num_particles,
            return_log_marginal_likelihood=True, return_latents=True,
            return_original_latents=smc, return_log_weight=True, return_log_weights=True,
            return_ancestral_indices=smc)
    params = _named_params(parts)
    for p in params.values():
        p.grad = None
    with replay.replay(tape):
        loss = ref.losses.get_loss(observations, num_particles, algorithm, parts["initial"],
                                   parts["transition"], parts["emission"], parts["proposal"])
    loss.backward()

    arrays = {}
    for t, obs in enumerate(observations):
        arrays["obs_{}".format(t)] = _np(obs)
    for i, block in enumerate(tape.normals):
        arrays["normal_{}".format(i)] = block
    for i, block in enumerate(tape.uniforms):
        arrays["uniform_{}".format(i)] = block
    for pname, p in params.items():
        arrays["param_" + pname] = _np(p)
        arrays["grad_" + pname] = _np(p.grad) if p.grad is not None else np.zeros(p.shape)
    for t, lw in enumerate(result["log_weights"]):
        arrays["out_log_weights_{}".format(t)] = _np(lw)
    if not light:
        for t, x in enumerate(result["latents"]):
            arrays["out_latents_{}".format(t)] = _np(x)
    if smc:
        for t, a in enumerate(result["ancestral_indices"]):
            arrays["out_idx_{}".format(t)] = _np(a)
        if not light:
            for t, x in enumerate(result["original_latents"]):
                arrays["out_original_latents_{}".format(t)] = _np(x)
    arrays["out_lml"] = _np(result["log_marginal_likelihood"])
    arrays["out_log_weight"] = _np(result["log_weight"])
    arrays["out_last_latent"] = _np(result["last_latent"])
    arrays["out_loss"] = _np(loss)
    meta = dict(meta, name=name, algorithm=algorithm, num_particles=num_particles,
                num_timesteps=len(observations), num_normals=len(tape.normals),
                num_uniforms=len(tape.uniforms), param_names=sorted(params),
                versions={"torch": torch.__version__, "numpy": np.__version__})
    if smc and len(observations) > 1:
        meta["margin"] = _margin([arrays["out_log_weights_{}".format(t)]
                                  for t in range(len(observations))], tape.uniforms)
        if light:
            steps = len(observations) - 1
            meta["light"] = True
            meta["flips_vs_float64_cdf"] = _flips_per_step(
                [arrays["out_log_weights_{}".format(t)] for t in range(steps)], tape.uniforms,
                [arrays["out_idx_{}".format(t)] for t in range(steps)])
    arrays["meta"] = np.array(json.dumps(meta))
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("{:32s} loss={:+.6f} margin={:.3g} bytes={}".format(
        name, float(loss), meta.get("margin", float("nan")), os.path.getsize(path)))


# ---- cases ---------------------------------------------------------------------------------------
def lgssm1d_parts(emission_scale, dtype, seed):
    """The reference's own test/models/lgssm.py classes (config 1 of BASELINE.json)."""
    torch.manual_seed(seed)
    parts = {
        "initial": ref_lgssm.Initial(0.0, 1.0),
        "transition": ref_lgssm.Transition(0.9, 1.0),
        "emission": ref_lgssm.Emission(1.0, emission_scale),
        "proposal": ref_lgssm.Proposal(1.0, 0.1),
    }
    for part in parts.values():
        if isinstance(part, torch.nn.Module):
            part.to(dtype)
    return parts


def lgssm1d_observations(T, B, dtype, seed):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, generator=gen, dtype=torch.float64)
    out = []
    for t in range(T):
        if t > 0:
            x = 0.9 * x + torch.randn(B, generator=gen, dtype=torch.float64)
        out.append((x + torch.randn(B, generator=gen, dtype=torch.float64)).to(dtype))
    return out


def case_lgssm1d(name, algorithm, emission_scale, dtype, B=2, K=16, T=8, seed=1):
    parts = lgssm1d_parts(emission_scale, dtype, seed)
    observations = lgssm1d_observations(T, B, dtype, seed + 100)
    np.random.seed(seed)
    torch.manual_seed(seed + 1)
    meta = {"model": "lgssm1d", "dtype": str(dtype).replace("torch.", ""), "batch_size": B,
            "emission_scale": emission_scale, "initial": [0.0, 1.0], "transition_scale": 1.0,
            "proposal_scales": [1.0, 0.1]}
    capture_infer(name, meta, parts, observations, K, algorithm)


def case_lgssm_nd(name, algorithm, dtype, dim=3, B=4, K=256, T=6, seed=3, proposal_scale=0.7, light=False):
    model = fixture_models.FixtureLgssmNd(dim, ref.state, proposal_scale=proposal_scale, seed=seed, dtype=dtype)
    observations = model.simulate(T, B, seed=seed + 100)
    parts = {"initial": model.initial, "transition": model.transition, "emission": model.emission,
             "proposal": model.proposal, "model": model}
    np.random.seed(seed)
    torch.manual_seed(seed + 1)
    meta = {"model": "lgssm_nd", "dtype": str(dtype).replace("torch.", ""), "batch_size": B,
            "dim": dim, "seed": seed, "proposal_scale": proposal_scale}
    capture_infer(name, meta, parts, observations, K, algorithm, light=light)


def case_gaussian(name, B=8, K=64, seed=5):
    """The reference's own test/models/gaussian.py (T = 1, transition None, config 3 shape)."""
    torch.manual_seed(seed)
    prior = ref_gaussian.Prior(0.3, 1.0)
    likelihood = ref_gaussian.Likelihood(0.8)
    network = ref_gaussian.InferenceNetwork(0.6, 0.1, 0.9)
    observations = [torch.randn(B)]
    parts = {"initial": prior, "transition": None, "emission": likelihood, "proposal": network}
    np.random.seed(seed)
    torch.manual_seed(seed + 1)
    meta = {"model": "gaussian", "dtype": "float32", "batch_size": B, "prior_std": 1.0}
    capture_infer(name, meta, parts, observations, K, "iwae")


def case_resampler(name, log_weight, seed):
    """Standalone aesmc.inference.sample_ancestral_index on given log-weights."""
    np.random.seed(seed)
    with replay.record() as tape:
        idx = ref.inference.sample_ancestral_index(torch.as_tensor(log_weight))
    mine, _ = kernel_oracle.ancestor_index(log_weight, tape.uniforms[0])
    mismatches = int((mine != _np(idx)).sum())
    meta = {"name": name, "dtype": str(log_weight.dtype), "mismatches_vs_float64_cdf": mismatches,
            "shape": list(log_weight.shape)}
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, log_weight=log_weight, uniform=tape.uniforms[0], out_idx=_np(idx),
                        meta=np.array(json.dumps(meta)))
    print("{:32s} float64-CDF mismatches={} / {} bytes={}".format(
        name, mismatches, idx.numel(), os.path.getsize(path)))


def case_train(name, algorithm, seed=9):
    """aesmc.train.train driven end to end by the reference (its SyntheticDataset, its optimiser
    loop): pins RNG consumption order of data generation + inference across steps and epochs."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    if algorithm == "iwae":
        true_parts = (ref_gaussian.Prior(1.0, 1.0), None, ref_gaussian.Likelihood(0.5))
        parts = {"initial": ref_gaussian.Prior(0.0, 1.0), "transition": None,
                 "emission": ref_gaussian.Likelihood(1.0),
                 "proposal": ref_gaussian.InferenceNetwork(0.1, 0.0, 1.5)}
        num_timesteps, batch_size, num_particles = 1, 6, 8
        meta = {"model": "gaussian", "prior_std": 1.0, "true": [1.0, 1.0, 0.5]}
    else:
        true_parts = (ref_lgssm.Initial(0.0, 1.0), ref_lgssm.Transition(0.9, 1.0), ref_lgssm.Emission(1.0, 1.0))
        parts = {"initial": ref_lgssm.Initial(0.0, 1.0), "transition": ref_lgssm.Transition(0.5, 1.0),
                 "emission": ref_lgssm.Emission(0.7, 1.0), "proposal": ref_lgssm.Proposal(1.0, 0.1)}
        num_timesteps, batch_size, num_particles = 4, 3, 8
        meta = {"model": "lgssm1d", "initial": [0.0, 1.0], "transition_scale": 1.0, "emission_scale": 1.0,
                "proposal_scales": [1.0, 0.1], "true": [0.9, 1.0]}
    params = _named_params(parts)
    arrays = {"init_" + k: _np(v) for k, v in params.items()}
    torch.manual_seed(seed + 1)   # model construction drew from the stream: restart it here
    np.random.seed(seed + 1)
    dataloader = ref.train.get_synthetic_dataloader(*true_parts, num_timesteps, batch_size)
    losses = []
    # the tape holds every standard-normal block (data generation AND proposal sampling, in the order
    # drawn) and every resampling uniform block of the whole run: a device whose generator differs
    # from the CPU's (the MI355X) can replay the run draw for draw
    with replay.record() as tape:
        ref.train.train(dataloader, num_particles, algorithm, parts["initial"], parts["transition"],
                        parts["emission"], parts["proposal"], num_epochs=2, num_iterations_per_epoch=2,
                        optimizer_algorithm=torch.optim.SGD, optimizer_kwargs={"lr": 0.05},
                        callback=lambda e, i, loss, *rest: losses.append(float(loss)))
    for i, block in enumerate(tape.normals):
        arrays["normal_{}".format(i)] = block
    for i, block in enumerate(tape.uniforms):
        arrays["uniform_{}".format(i)] = block
    for k, v in params.items():
        arrays["final_" + k] = _np(v)
    arrays["losses"] = np.array(losses)
    arrays["torch_state_probe"] = _np(torch.rand(3))          # where the torch stream ended up
    arrays["numpy_state_probe"] = np.random.uniform(size=3)   # and numpy's
    meta = dict(meta, name=name, algorithm=algorithm, seed=seed, num_timesteps=num_timesteps,
                batch_size=batch_size, num_particles=num_particles, dtype="float32",
                param_names=sorted(params))
    arrays["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **arrays)
    print("{:32s} losses={}".format(name, np.round(losses, 5)))


def case_api_signatures():
    """The reference's public surface as data: for every public function / class of its six
    modules, the parameter names, which of them have defaults, and the defaults that are plain
    constants.  tests/test_host_logic.py holds this package's modules against it."""
    import inspect
    import aesmc
    table = {}
    for module_name in ("inference", "losses", "math", "state", "statistics", "train"):
        module = getattr(aesmc, module_name)
        for name, member in sorted(vars(module).items()):
            if name.startswith("_") or getattr(member, "__module__", None) != module.__name__:
                continue
            if inspect.isclass(member) and not issubclass(member, __import__("enum").Enum):
                target = member.__init__
            elif inspect.isfunction(member):
                target = member
            elif inspect.isclass(member):
                table["{}.{}".format(module_name, name)] = {"enum": sorted(m.name for m in member)}
                continue
            else:
                continue
            params = []
            for p in inspect.signature(target).parameters.values():
                default = None
                if p.default is not inspect.Parameter.empty:
                    default = repr(p.default) if isinstance(p.default, (bool, int, float, str, type(None), dict)) \
                        else "<object>"
                params.append([p.name, default])
            table["{}.{}".format(module_name, name)] = {"params": params}
    with open(os.path.join(GOLDEN, "api_signatures.json"), "w") as fh:
        json.dump(table, fh, indent=1, sort_keys=True)
    print("api_signatures.json: {} entries".format(len(table)))


def main(out_dir=None):
    """Writes every fixture into `out_dir` (default: tests/golden/)."""
    global GOLDEN
    if out_dir is not None:
        GOLDEN = out_dir
    os.makedirs(GOLDEN, exist_ok=True)
    case_api_signatures()
    f32, f64 = torch.float32, torch.float64
    # config 1 of BASELINE.json: reference's 1-D LGSSM, B=2, K=16, T=8
    case_lgssm1d("c1_lgssm1d_smc_f32", "aesmc", 1.0, f32)
    case_lgssm1d("c1_lgssm1d_smc_f64", "aesmc", 1.0, f64)
    case_lgssm1d("c1_lgssm1d_smc_stock_f32", "aesmc", 0.01, f32)  # test_losses.py:92 scale: collapses
    case_lgssm1d("c1_lgssm1d_is_f32", "iwae", 1.0, f32)
    # d-dimensional LGSSM (the bench model) at fixture size
    case_lgssm_nd("lgssm3d_smc_f32", "aesmc", f32)
    case_lgssm_nd("lgssm3d_smc_f64", "aesmc", f64)
    case_lgssm_nd("lgssm3d_is_f32", "iwae", f32, K=64, T=4)
    case_lgssm_nd("lgssm10d_smc_f64", "aesmc", f64, dim=10, B=2, K=512, T=4, seed=7)
    # one-step IWAE (config 3 shape at fixture size)
    case_gaussian("gaussian_iwae_f32")
    # standalone resampler
    rng = np.random.RandomState(11)
    case_resampler("resampler_k1000_s1_f64", rng.randn(8, 1000), 21)
    case_resampler("resampler_k1000_s5_f64", 5 * rng.randn(8, 1000), 22)
    case_resampler("resampler_k4096_f64", 2 * rng.randn(4, 4096), 23)
    case_resampler("resampler_k1000_s1_f32", rng.randn(8, 1000).astype(np.float32), 24)
    case_resampler("resampler_k4096_f32", (2 * rng.randn(4, 4096)).astype(np.float32), 25)
    lw = rng.randn(6, 33)
    lw[0, 3:9] = -np.inf
    lw[1, :] = -np.inf
    lw[1, 17] = 0.0
    lw[2, :] = 0.0
    lw[3, 5] = 700.0
    case_resampler("resampler_edge_f64", lw, 26)
    case_resampler("resampler_k1_f64", rng.randn(5, 1), 27)
    lw = rng.randn(3, 7)
    lw[1, :] = -np.inf
    lw[2, 2] = np.inf
    case_resampler("resampler_degenerate_f64", lw, 28)
    # the whole training loop
    case_train("train_iwae_gaussian", "iwae")
    case_train("train_aesmc_lgssm1d", "aesmc")
    # round 2: the resampler at configs[4]'s particle count, and a configs[1]-like float32 run
    # (d=10, K=1024, T=20) whose per-step log-weights pin the float32 flip rate on real weights
    rng = np.random.RandomState(12)
    case_resampler("resampler_k16384_f64", 2 * rng.randn(4, 16384), 31)
    case_resampler("resampler_k16384_f32", (2 * rng.randn(4, 16384)).astype(np.float32), 32)
    case_lgssm_nd("lgssm10d_k1024_smc_f32", "aesmc", f32, dim=10, B=2, K=1024, T=20, seed=11, light=True)
    # round 3: a float32 run of the reference at the north-star particle count (d=10, K=4096)
    case_lgssm_nd("lgssm10d_k4096_smc_f32", "aesmc", f32, dim=10, B=2, K=4096, T=10, seed=13, light=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
