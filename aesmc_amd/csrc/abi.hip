// Library identification entry points of the C ABI (include/aesmc_hip.h).
#include "common.hpp"

extern "C" int aesmc_version(void) { return 501; /* 0.5.1: see the history beside the declaration */ }
extern "C" const char *aesmc_target_arch(void) { return "gfx950"; }

// The address a kernel can use for `host_ptr`, a pointer into PINNED host memory (hipHostMalloc; PyTorch's
// `pin_memory=True`): the resampling launch then reads its per-row uniforms — 8 bytes per batch row, written by the host
// just before the launch (aesmc/inference.py:250) — where the host wrote them, instead of behind a copy launch per
// timestep.  AESMC_ERR_UNSUPPORTED when the memory is not mapped into the device's address space.
extern "C" int aesmc_host_device_pointer(const void *host_ptr, void **device_ptr) {
  if (host_ptr == nullptr || device_ptr == nullptr) return AESMC_ERR_INVALID_ARGUMENT;
  void *mapped = nullptr;
  if (hipHostGetDevicePointer(&mapped, const_cast<void *>(host_ptr), 0) != hipSuccess || mapped == nullptr) {
    (void)hipGetLastError();
    return AESMC_ERR_UNSUPPORTED;
  }
  *device_ptr = mapped;
  return AESMC_OK;
}
