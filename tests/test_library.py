"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
that include/aesmc_hip.h declares (no compute without a GPU), and the product path refuses to run
anywhere but on a HIP device."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(ROOT, "include", "aesmc_hip.h")) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(aesmc_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from aesmc_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 9
    for name in names:
        assert hasattr(lib, name), "libaesmc_hip.so does not export " + name
    assert sorted(_lib.SIGNATURES) == names       # ctypes binding covers the header one to one
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        getattr(raw, name)
    assert lib.aesmc_version() == 501
    import aesmc_amd
    assert aesmc_amd.__version__ == "0.{}.{}".format(501 // 100, 501 % 100)      # the package says the ABI's version (501 = 0.5.1)
    assert lib.aesmc_target_arch() == b"gfx950"
    assert lib.aesmc_ancestor_index_lds_max_particles() >= 16384
    assert lib.aesmc_workspace_bytes(4, 1024) == 0
    assert lib.aesmc_workspace_bytes(4, 40000) == 4 * (40000 + 40000 // 8 + 1) * 8


def test_library_is_gfx950_code_object():
    from aesmc_amd import _lib
    with open(_lib.LIB_PATH, "rb") as fh:
        blob = fh.read()
    assert b"gfx950" in blob
    for kernel in (b"logweight_lse_kernel", b"ancestor_index_kernel", b"resample_gather_kernel",
                   b"resample_gather_bwd_kernel"):
        assert kernel in blob


def test_argument_validation_happens_before_any_launch():
    """NULL pointers / negative sizes are rejected by the ABI itself (status 1), no GPU needed."""
    from aesmc_amd import _lib
    lib = _lib.load()
    assert lib.aesmc_logweight_lse(0, None, None, None, None, None, 1, 1, None) == 1
    assert lib.aesmc_logweight_lse(7, 8, None, None, 8, None, 0, 4, None) == 0      # B == 0: no-op
    assert lib.aesmc_ancestor_index(0, None, None, None, None, 1, 1, None, 0, None) == 1
    assert lib.aesmc_ancestor_index(0, 8, 8, 8, None, -1, 4, None, 0, None) == 1
    assert lib.aesmc_resample_gather(None, None, None, None, 1, 1, 4, 4, 4, None) == 1
    assert lib.aesmc_resample_gather(8, 8, 8, None, 1, 1 << 31, 4, 4, 4, None) == 2   # unsupported size
    assert lib.aesmc_resample_gather_backward(5, 8, 8, 8, None, 1, 1, 1, 0, None) == 1   # bad dtype tag
    # linear-Gaussian propagation: NULL operands, a misaligned base pointer, a map wider than 16
    assert lib.aesmc_affine_max_dim() == 16
    assert lib.aesmc_particle_affine(0, None, None, None, None, None, None, 1, 1, None) == 1
    amap = _lib.AffineMap(16, 4, 1, 0, 0, 4, 4)
    assert lib.aesmc_particle_affine(0, 8, ctypes.byref(amap), None, None, None, 16, 1, 1, None) == 1   # x1 misaligned
    wide = _lib.AffineMap(16, 17, 1, 0, 0, 17, 17)
    assert lib.aesmc_particle_affine(0, 16, ctypes.byref(wide), None, None, None, 32, 1, 1, None) == 2
    assert lib.aesmc_particle_affine(0, 16, ctypes.byref(amap), None, None, None, 32, 0, 5, None) == 0   # B == 0: no-op
    assert lib.aesmc_affine_normal_rsample(0, 16, ctypes.byref(amap), 32, 48, 32, 1, 1, None) == 1      # out aliases eps
    assert lib.aesmc_affine_normal_logweight(0, None, None, None, 0, None, None, None, None, None, None, None, 1, 1,
                                             None) == 1
    pairs = 16 + 3 * 256 * 4      # (round 5: behind the records and row sums, the three maps' interleaved weight pairs)
    assert lib.aesmc_affine_backward_workspace_bytes(0, 0, 0) == 1024 * 4 * 256 * 4 + pairs
    assert lib.aesmc_affine_backward_workspace_bytes(1, 2, 300) == (1024 * 4 * 256 + 3 * 3 * 8 * 16) * 8 + pairs
    assert lib.aesmc_particle_affine_backward(0, 16, 32, ctypes.byref(amap), None, 48, None, None, 0, 1, 1, None) == 1
    # the matrix-core step's extents (0.5.0) and its workspace by width
    assert (lib.aesmc_affine_wide_dim(), lib.aesmc_affine_wide_min_dim(), lib.aesmc_affine_wide_max_dim()) == (128, 17, 256)
    assert lib.aesmc_affine_wide_workspace_bytes_for(2, 64, 128, 128) == 2 * 64 * 3 * 4        # two draw sums + one emission sum
    assert lib.aesmc_affine_wide_workspace_bytes_for(2, 64, 256, 256) == 2 * 64 * (2 * 4 + 2) * 4      # four + two chunks
    assert lib.aesmc_affine_wide_workspace_bytes_for(2, 64, 300, 16) == 0                       # beyond the widest row
    # the proposal net (K13 / K13b): NULL operands, a hidden layer wider than 64, sixteen inputs in the backward
    assert lib.aesmc_particle_mlp_max_hidden() == 64
    hidden = _lib.AffineMap(16, 4, 1, 0, 0, 32, 4)        # [32, 4]
    output = _lib.AffineMap(16, 32, 1, 0, 0, 4, 32)       # [4, 32]
    assert lib.aesmc_particle_mlp(0, None, ctypes.byref(hidden), ctypes.byref(output), 32, 1, 256, None) == 1
    assert lib.aesmc_particle_mlp(0, 16, ctypes.byref(hidden), ctypes.byref(output), 16, 1, 256, None) == 1      # out aliases x
    too_wide = _lib.AffineMap(16, 4, 1, 0, 0, 65, 4)
    assert lib.aesmc_particle_mlp(0, 16, ctypes.byref(too_wide), ctypes.byref(output), 32, 1, 256, None) == 2
    assert lib.aesmc_particle_mlp(0, 16, ctypes.byref(hidden), ctypes.byref(output), 32, 0, 256, None) == 0      # B == 0: no-op
    assert lib.aesmc_particle_mlp_backward_records(4, 300) == 0 and lib.aesmc_particle_mlp_backward_records(4, 512) > 0
    assert lib.aesmc_particle_mlp_backward(0, 16, None, ctypes.byref(hidden), ctypes.byref(output), None, None, None, None,
                                           1, 256, None) == 1
    assert lib.aesmc_particle_mlp_backward(0, 16, 32, ctypes.byref(hidden), ctypes.byref(output), None, None, None, None,
                                           1, 300, None) == 2                                    # K not a multiple of 256
    # the weight pairs with the densities' constants behind them (0.5.1), and the first timestep's launch (K20)
    assert lib.aesmc_affine_weight_pairs_floats() == 3 * 256 + 8
    assert lib.aesmc_affine_weight_pairs_scaled(ctypes.byref(amap), ctypes.byref(amap), ctypes.byref(amap), 16, 16, 16, None,
                                                None) == 1                                       # no buffer
    assert lib.aesmc_affine_weight_pairs_scaled(ctypes.byref(amap), ctypes.byref(amap), ctypes.byref(amap), None, 16, 16, 64,
                                                None) == 1                                       # a scale missing
    assert lib.aesmc_affine_weight_pairs_scaled(ctypes.byref(amap), ctypes.byref(wide), ctypes.byref(amap), 16, 16, 16, 64,
                                                None) == 2                                       # a map wider than 16
    row = _lib.View3(16, 4, 0, 1)          # one row per batch element, constant along the particles
    along = _lib.View3(16, 64, 4, 1)       # varies along the particles
    views = [ctypes.byref(row)] * 5
    assert lib.aesmc_affine_normal_initial_step(None, *views, ctypes.byref(amap), ctypes.byref(row), 32, 48, 1, 16, None) == 1
    assert lib.aesmc_affine_normal_initial_step(16, *views, ctypes.byref(amap), ctypes.byref(row), 32, 48, -1, 16, None) == 1
    assert lib.aesmc_affine_normal_initial_step(16, ctypes.byref(along), *views[1:], ctypes.byref(amap), ctypes.byref(row), 32, 48,
                                                1, 16, None) == 2                                # not constant along particles
    assert lib.aesmc_affine_normal_initial_step(16, *views, ctypes.byref(wide), ctypes.byref(row), 32, 48, 1, 16, None) == 2
    assert lib.aesmc_affine_normal_initial_step(16, *views, ctypes.byref(amap), ctypes.byref(row), 32, 48, 1 << 20, 1 << 12,
                                                None) == 2                                       # beyond 32-bit element arithmetic
    assert lib.aesmc_affine_normal_initial_step(16, *views, ctypes.byref(amap), ctypes.byref(row), 32, 48, 0, 16, None) == 0   # B == 0: no-op
    sixteen = _lib.AffineMap(16, 16, 1, 0, 0, 32, 16)
    assert lib.aesmc_particle_mlp_backward(0, 16, 32, ctypes.byref(sixteen), ctypes.byref(output), None, None, None, None,
                                           1, 256, None) == 2                                    # no column left for the ones


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_product_refuses_cpu_tensors_loudly():
    """No CPU fallback: every operator of the hot path raises on host tensors."""
    from aesmc_amd import inference, math, state
    from aesmc_amd.testing import models
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        state.resample(torch.zeros(2, 3), torch.zeros(2, 3, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        inference.sample_ancestral_index(torch.zeros(2, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        math.lognormexp(torch.zeros(2, 3), dim=1)
    model = models.LgssmNd(2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        inference.infer("smc", model.simulate(3, 2), model.initial, model.transition, model.emission,
                        model.proposal, 4)
    from aesmc_amd.linear_gaussian import AffineNormal, particle_affine, particle_mlp
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        particle_affine(torch.zeros(2, 3, 4), torch.zeros(4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        particle_mlp(torch.zeros(2, 256, 4), torch.zeros(8, 4), torch.zeros(8), torch.zeros(3, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        AffineNormal(torch.zeros(2, 3, 4), torch.zeros(4, 4), torch.tensor(1.0)).loc


def test_missing_library_is_an_error_not_a_fallback(monkeypatch):
    from aesmc_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "does_not_exist.so"))
    with pytest.raises(_lib.AesmcLibraryError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under aesmc_amd/ may reference it."""
    package = os.path.join(ROOT, "aesmc_amd")
    for folder, _, files in os.walk(package):
        for name in files:
            if name.endswith((".py", ".hip", ".hpp")):
                with open(os.path.join(folder, name)) as fh:
                    text = fh.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), name
                assert "kernel_oracle" not in text and "reference_port" not in text, name
