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
           "normal_logprob.hip", "normal_rsample.hip", "particle_summary.hip", "linear_gaussian.hip",
           "linear_gaussian_backward.hip", "philox_normal.hip", "linear_gaussian_noise.hip",
           "linear_gaussian_fused.hip", "linear_gaussian_item.hip", "linear_gaussian_step_backward.hip", "linear_gaussian_wide.hip", "linear_gaussian_wide_backward.hip",
           "linear_gaussian_wide_generic_draw.hip", "linear_gaussian_wide_generic_emit.hip", "particle_mlp.hip",
           "linear_gaussian_initial.hip"]
HEADERS = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "linear_gaussian.hpp"),
           os.path.join(CSRC, "philox_normal.hpp"), os.path.join(CSRC, "linear_gaussian_fused.hpp"),
           os.path.join(CSRC, "linear_gaussian_backward.hpp"), os.path.join(CSRC, "linear_gaussian_wide_generic.hpp"),
           os.path.join(os.path.dirname(HERE), "include", "aesmc_hip.h")]
ARCH = "gfx950"
# per translation unit: the generic matrix-core step's product loops are 96 - 128 operand groups of eight v_mfma each, written
# as ONE loop that must be unrolled in full (the accumulator tiles are register arrays indexed by the loop counter); the
# compiler's default budget for `#pragma unroll` is a few instructions short of the largest instantiations — it then keeps
# the loop and puts the tiles in scratch (784 bytes, five times slower).  The other sources keep the default budget.
SOURCE_FLAGS = {"linear_gaussian_wide_generic_draw.hip": ["-mllvm", "-pragma-unroll-threshold=200000"],
                "linear_gaussian_wide_generic_emit.hip": ["-mllvm", "-pragma-unroll-threshold=200000"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _depends_on(path, seen=None):
    """`path` and every file it includes with quotes, transitively (what a translation unit is rebuilt for)."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path) as handle:
        for line in handle:
            line = line.strip()
            if line.startswith("#include \""):
                _depends_on(os.path.join(os.path.dirname(path), line.split('"')[1]), seen)
    return seen


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
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH,
             "-ffp-contract=off",  # keep float64 CDF arithmetic as written (no fused a*b+c)
             "-munsafe-fp-atomics",  # hardware float atomics for the gather backward
             "-Wall", "-Wno-unused-function"]
    extra = os.environ.get("AESMC_HIPCC_FLAGS", "").split()      # experiments (another -O level, -save-temps, ...); empty in a product build
    flags += extra
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for source in SOURCES:      # one hipcc per translation unit, side by side; unchanged ones are kept
        src, obj = os.path.join(CSRC, source), os.path.join(objdir, source[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(d) for d in _depends_on(src)):
            continue
        cmd = [_hipcc()] + flags + SOURCE_FLAGS.get(source, []) + ["-c", src, "-o", obj]
        if verbose:
            print("[aesmc_amd.build]", " ".join(cmd), flush=True)
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, job in jobs:
        if job.wait() != 0:
            raise subprocess.CalledProcessError(job.returncode, cmd)
    link = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB_PATH] + \
        [os.path.join(objdir, s[:-4] + ".o") for s in SOURCES]
    if verbose:
        print("[aesmc_amd.build]", " ".join(link), flush=True)
    subprocess.run(link, check=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
