"""Drop-in use of the package in place of the reference (`import aesmc_amd as aesmc`): train a
d-dimensional linear-Gaussian state-space model with the AESMC objective on the MI355X, then
summarise the filtering posterior.

    python examples/lgssm_train.py [--dim 10] [--particles 1024] [--batch 256] [--steps 200] [--graph]
                                   [--callables affine|matmul]

`--graph` runs the optimisation loop on one captured hipGraph (aesmc_amd.train(..., hip_graph=True)).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import aesmc_amd as aesmc                       # was: import aesmc
from aesmc_amd.testing import models            # contract models (initial / transition / emission / proposal)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=10)
    ap.add_argument("--particles", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--timesteps", type=int, default=20)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--callables", default="affine", choices=["affine", "matmul"],
                    help="affine: the model's callables return aesmc_amd.linear_gaussian.AffineNormal (locations "
                         "evaluated inside the sampling / weighting kernels); matmul: Normal(x @ W.T + c, s)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-tunableop", action="store_true",
                    help="keep PyTorch's default GEMM picks for the model's matmuls (slower on MI355X: see DESIGN.md 6)")
    args = ap.parse_args()
    if not args.no_tunableop:      # the model's skinny [B*K, d] x [d, d] matmuls: let PyTorch pick tuned kernels
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "lgssm_train_tunableop.csv"))

    device = torch.device("cuda", 0)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    truth = models.LgssmNd(args.dim, seed=1, validate_args=False).to(device)
    model = models.LgssmNd(args.dim, seed=0, validate_args=False, affine=args.callables == "affine").to(device)
    with torch.no_grad():                       # start from a poor proposal
        for p in (model.W0, model.Wx, model.Wy):
            p.mul_(0.3)

    loader = aesmc.train.get_synthetic_dataloader(truth.initial, truth.transition, truth.emission,
                                                  args.timesteps, args.batch)
    history = []
    started = time.perf_counter()
    aesmc.train.train(loader, args.particles, "aesmc", model.initial, model.transition, model.emission,
                      model.proposal, num_epochs=1, num_iterations_per_epoch=args.steps,
                      optimizer_algorithm=torch.optim.Adam, optimizer_kwargs={"lr": 3e-3},
                      callback=lambda epoch, it, loss, *parts: history.append(loss.detach()),
                      hip_graph=args.graph)
    torch.cuda.synchronize()
    seconds = time.perf_counter() - started
    losses = torch.stack(history).cpu().numpy()
    rate = args.batch * args.particles * args.timesteps * args.steps / seconds
    print("loss {:.3f} -> {:.3f} in {} steps, {:.1f} s ({:.2e} particle-steps/s forward+backward{})".format(
        losses[:10].mean(), losses[-10:].mean(), args.steps, seconds, rate, ", hipGraph" if args.graph else ""))

    # posterior summaries, as the reference's test/models/lgssm.py does: filtering mean / variance, ESS
    observations = truth.simulate(args.timesteps, args.batch, seed=7)
    with torch.no_grad():
        out = aesmc.inference.infer("smc", observations, model.initial, model.transition, model.emission,
                                    model.proposal, args.particles, return_log_marginal_likelihood=True)
    last = out["latents"][-1]
    mean = aesmc.statistics.empirical_mean(last, out["log_weight"])
    variance = aesmc.statistics.empirical_variance(last, out["log_weight"])
    ess = aesmc.statistics.ess(out["log_weight"])
    print("log Z per sequence {:.3f}; last-step ESS {:.1f} of {}; |mean| {:.3f}, mean variance {:.3f}".format(
        out["log_marginal_likelihood"].mean().item(), ess.mean().item(), args.particles,
        mean.abs().mean().item(), variance.mean().item()))


if __name__ == "__main__":
    main()
