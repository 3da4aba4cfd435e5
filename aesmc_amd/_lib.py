"""ctypes binding of libaesmc_hip.so (C ABI declared in include/aesmc_hip.h).

The library is the only compute backend of this package: if it is missing or a tensor is not on
a HIP device the callers raise — there is no CPU or eager-PyTorch fallback.
"""
import ctypes
import os

import torch  # noqa: F401  (loads the HIP runtime that the library's DT_NEEDED resolves against)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaesmc_hip.so")

OK = 0
ERR_NAMES = {1: "invalid argument", 2: "unsupported shape", 3: "kernel launch failed",
             4: "workspace missing or too small"}
FLAG_NAN_LOG_WEIGHT = 1
FLAG_DEGENERATE_ROW = 2
FLAG_INDEX_OUT_OF_RANGE = 4
FLAG_VALUE_OUTSIDE_SUPPORT = 8  # host-side deferred validation (state.log_prob), not a kernel
FLAG_UNSORTED_INDEX = 16
FLAG_INVALID_PARAMETER = 32  # host-side deferred argument validation (_syncfree: Distribution.__init__), not a kernel
F32, F64 = 0, 1

_vp, _i64, _i32, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
_u64 = ctypes.c_uint64


class View3(ctypes.Structure):
    """`aesmc_view3` of include/aesmc_hip.h: a [B,K,D] view by element strides (0 = broadcast)."""
    _fields_ = [("ptr", ctypes.c_void_p), ("stride_b", ctypes.c_int64), ("stride_k", ctypes.c_int64),
                ("stride_d", ctypes.c_int64)]


class AffineMap(ctypes.Structure):
    """`aesmc_affine_map` of include/aesmc_hip.h: loc = offset + weight @ x without materialising it."""
    _fields_ = [("weight", ctypes.c_void_p), ("stride_out", ctypes.c_int64), ("stride_in", ctypes.c_int64),
                ("offset", ctypes.c_void_p), ("offset_stride_b", ctypes.c_int64), ("dout", ctypes.c_int64),
                ("din", ctypes.c_int64)]


_map_p = ctypes.POINTER(AffineMap)


class AffineLogweightGrads(ctypes.Structure):
    """`aesmc_affine_logweight_grads` of include/aesmc_hip.h: K12's optional outputs."""
    _fields_ = [(name, ctypes.c_void_p) for name in (
        "grad_x_prev", "grad_x", "grad_loc_p", "grad_loc_g", "grad_loc_q", "grad_weight_p", "grad_weight_g",
        "grad_weight_q", "grad_scales", "grad_offset_p", "grad_offset_g", "grad_offset_q")]

class AffineChain(ctypes.Structure):
    """`aesmc_affine_chain` of include/aesmc_hip.h: how K14's weight gradients join those of the steps around it."""
    _fields_ = [("carry", ctypes.c_void_p), ("carry_records", ctypes.c_int32), ("defer", ctypes.c_int32),
                ("records", ctypes.c_int32), ("pairs_in", ctypes.c_void_p), ("pairs_out", ctypes.c_void_p)]


# name -> (restype, argtypes); mirrors include/aesmc_hip.h one to one.
SIGNATURES = {
    "aesmc_version": (_i32, []),
    "aesmc_target_arch": (ctypes.c_char_p, []),
    "aesmc_host_device_pointer": (_i32, [_vp, ctypes.POINTER(ctypes.c_void_p)]),
    "aesmc_logweight_lse": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "aesmc_logweight_accumulate": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "aesmc_logweight_lse_backward": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "aesmc_ancestor_index": (_i32, [_i32, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _sz, _vp]),
    "aesmc_ancestor_index_lds_max_particles": (_i64, []),
    "aesmc_workspace_bytes": (_sz, [_i64, _i64]),
    "aesmc_resample_gather": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp]),
    "aesmc_resample_gather_backward": (_i32, [_i32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "aesmc_normal_logprob_sum": (_i32, [_i32, _vp, _vp, _vp, _vp] + [_i64] * 12 + [_vp]),
    "aesmc_normal_logprob_sum_backward": (_i32, [_i32] + [_vp] * 7 + [_i64] * 12 + [_vp]),
    "aesmc_normal_logweight": (_i32, [_i32, ctypes.POINTER(View3), _vp, _i64, _i64, _i64, _i64, _vp]),
    "aesmc_normal_logweight_backward": (_i32, [_i32, ctypes.POINTER(View3)] + [_vp] * 9 + [_i64] * 4 + [_vp]),
    "aesmc_normal_logweight_lse_backward": (_i32, [_i32, ctypes.POINTER(View3)] + [_vp] * 12 + [_i64] * 4 + [_vp]),
    "aesmc_resample_step": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp]),
    "aesmc_particle_summary_workspace_bytes": (_sz, [_i32, _i64, _i64, _i64]),
    "aesmc_particle_summary": (_i32, [_i32, _vp, ctypes.POINTER(View3), _vp, _vp, _vp, _i64, _i64, _i64, _vp, _sz,
                                      _vp]),
    "aesmc_normal_rsample": (_i32, [_i32] + [ctypes.POINTER(View3)] * 3 + [_vp, _i64, _i64, _i64, _vp]),
    "aesmc_affine_max_dim": (_i64, []),
    "aesmc_particle_affine": (_i32, [_i32, _vp, _map_p, _vp, _map_p, _vp, _vp, _i64, _i64, _vp]),
    "aesmc_particle_affine_tanh": (_i32, [_i32, _vp, _map_p, _vp, _map_p, _vp, _vp, _i64, _i64, _vp]),
    "aesmc_affine_normal_rsample": (_i32, [_i32, _vp, _map_p, _vp, _vp, _vp, _i64, _i64, _vp]),
    "aesmc_affine_backward_workspace_bytes": (_sz, [_i32, _i64, _i64]),
    "aesmc_particle_affine_backward": (_i32, [_i32, _vp, _vp, _map_p, _vp, _vp, _vp, _vp, _sz, _i64, _i64, _vp]),
    "aesmc_affine_normal_logweight_backward": (_i32, [_i32, _vp, _vp, _vp, _i64, _map_p, _map_p, _map_p] + [_vp] * 7 +
                                                      [ctypes.POINTER(AffineLogweightGrads), _vp, _sz, _i64, _i64, _vp]),
    "aesmc_affine_normal_propagate": (_i32, [_i32, _vp, _vp, _vp, _i64, _map_p, _map_p, _map_p, _vp, _vp, _vp, _vp, _vp,
                                             _i64, _i64, _vp]),
    "aesmc_affine_step_backward": (_i32, [_i32, _vp, _vp, _vp, _i64, _map_p, _map_p, _map_p] + [_vp] * 8 +
                                          [ctypes.POINTER(AffineLogweightGrads), _vp, _sz, _i64, _i64, _vp]),
    "aesmc_affine_step_backward_resampled": (_i32, [_i32, _vp, _vp, _vp, _vp, _i64, _map_p, _map_p, _map_p] + [_vp] * 10 +
                                                    [ctypes.POINTER(AffineLogweightGrads), _vp, _sz, _vp,
                                                     ctypes.POINTER(AffineChain), _i64, _i64, _vp]),
    "aesmc_affine_backward_collect": (_i32, [_i32, _vp, _i32, _i64, _i64, ctypes.POINTER(AffineLogweightGrads), _vp]),
    "aesmc_resample_step_ranges": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "aesmc_affine_normal_logweight": (_i32, [_i32, _vp, _vp, _vp, _i64, _map_p, _map_p, _map_p, _vp, _vp, _vp, _vp,
                                             _i64, _i64, _vp]),
    "aesmc_affine_normal_propagate_resampled": (_i32, [_i32, _vp, _vp, _vp, _vp, _i64, _map_p, _map_p, _map_p, _vp, _vp,
                                                       _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "aesmc_affine_normal_propagate_drawn": (_i32, [_vp, _vp, _vp, _i64, _map_p, _map_p, _map_p, _vp, _vp, _vp, _vp, _vp,
                                                   _vp, _i64, _i64, _u64, _u64, _i64, _vp, _vp]),
    "aesmc_affine_normal_propagate_drawn_paired": (_i32, [_vp, _vp, _vp, _i64, _map_p, _map_p, _map_p, _vp, _vp, _vp, _vp,
                                                          _vp, _vp, _i64, _i64, _u64, _u64, _i64, _vp, _vp, _vp]),
    "aesmc_affine_weight_pairs_floats": (_i64, []),
    "aesmc_affine_weight_pairs": (_i32, [_map_p, _map_p, _map_p, _vp, _vp]),
    "aesmc_affine_weight_pairs_scaled": (_i32, [_map_p, _map_p, _map_p, _vp, _vp, _vp, _vp, _vp]),
    "aesmc_affine_normal_initial_step": (_i32, [_vp] + [ctypes.POINTER(View3)] * 5 + [_map_p, ctypes.POINTER(View3), _vp, _vp,
                                                _i64, _i64, _vp]),
    "aesmc_particle_mlp_max_hidden": (_i64, []),
    "aesmc_particle_mlp": (_i32, [_i32, _vp, _map_p, _map_p, _vp, _i64, _i64, _vp]),
    "aesmc_particle_mlp_backward_records": (_i64, [_i64, _i64]),
    "aesmc_particle_mlp_backward": (_i32, [_i32, _vp, _vp, _map_p, _map_p, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "aesmc_affine_wide_dim": (_i64, []),
    "aesmc_wide_adjoint_tile": (_i64, []),
    "aesmc_wide_adjoint_scale": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _vp]),
    "aesmc_wide_adjoint_merge": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "aesmc_affine_wide_workspace_bytes": (_sz, [_i64, _i64]),
    "aesmc_affine_wide_min_dim": (_i64, []),
    "aesmc_affine_wide_max_dim": (_i64, []),
    "aesmc_affine_wide_workspace_bytes_for": (_sz, [_i64, _i64, _i64, _i64]),
    "aesmc_affine_normal_propagate_wide": (_i32, [_vp, _vp, _vp, _vp, _i64, _map_p, _map_p, _map_p, _vp, _vp, _vp, _vp, _vp,
                                                  _vp, _sz, _vp, _i64, _i64, _u64, _u64, _i64, _vp, _vp]),
    "aesmc_philox_normal_fill": (_i32, [_vp, _i64, _u64, _u64, _i64, _i32, _vp, _vp]),
    "aesmc_normal_rsample_drawn": (_i32, [_i32, ctypes.POINTER(View3), ctypes.POINTER(View3), _vp, _i64, _i64, _i64, _u64, _u64, _i64, _i32, _vp, _vp]),
}

# measurement / test hooks the library also exports; deliberately NOT in include/aesmc_hip.h (process-wide
# switches are not something a binder of the C ABI should see): which variant of two equivalent kernels runs
TEST_HOOKS = {
    "aesmc_test_set_step_parts": (_i32, [_i32]),
    "aesmc_test_set_sorted_backward_kernel": (_i32, [_i32]),
    "aesmc_test_set_step_backward": (_i32, [_i32, _i32]),
    "aesmc_test_set_k16_form": (_i32, [_i32]),
    "aesmc_test_last_k16_form": (_i32, []),
    "aesmc_test_last_step_backward_form": (_i32, []),
    "aesmc_test_set_k2_form": (_i32, [_i32]),
    "aesmc_test_last_k2_form": (_i32, []),
    "aesmc_test_last_logweight_backward_form": (_i32, []),
}

_lib = None


class AesmcLibraryError(RuntimeError):
    pass


def load():
    """Loads the shared library once and declares every entry point's signature."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AesmcLibraryError(
            "aesmc_amd: HIP library not built ({} missing). Run `python -m aesmc_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.".format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in list(SIGNATURES.items()) + list(TEST_HOOKS.items()):
        fn = getattr(lib, name)  # AttributeError here == the library does not export the symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status, what):
    if status != OK:
        raise AesmcLibraryError("aesmc_amd: {} failed: {} (status {})".format(
            what, ERR_NAMES.get(status, "unknown error"), status))
