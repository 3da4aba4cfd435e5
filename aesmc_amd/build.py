"""Builds libaesmc_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

Usage: ``python -m aesmc_amd.build [--force]``.  hipcc cross-compiles without a GPU.  The .so is
git-ignored but travels with the working tree; nothing is installed into site-packages.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libaesmc_hip.so")
SOURCES = ["abi.hip", "logweight_lse.hip", "ancestor_index.hip", "resample_gather.hip",
           "normal_logprob.hip", "normal_rsample.hip", "particle_summary.hip", "linear_gaussian.hip"]
HEADERS = [os.path.join(CSRC, "common.hpp"),
           os.path.join(os.path.dirname(HERE), "include", "aesmc_hip.h")]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > built for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP source into one shared library; returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=" + ARCH,
           "-ffp-contract=off",  # keep float64 CDF arithmetic as written (no fused a*b+c)
           "-munsafe-fp-atomics",  # hardware float atomics for the gather backward
           "-Wall", "-Wno-unused-function",
           "-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[aesmc_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
