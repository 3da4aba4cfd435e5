// aesmc_philox_normal_fill: the tensor `torch.empty(numel).normal_()` would hold, written by this library
// from (seed, offset) — ATen's launch geometry and rocRAND's Philox4x32-10 + Box-Muller restated (see
// philox_normal.hpp).  The propagation kernels draw their noise through the same helpers without ever
// writing it; this entry point exists so that the identity with PyTorch's stream can be tested on its own
// and so that a caller can materialise the noise of a step whose draw was left to a kernel.
#include "common.hpp"
#include "philox_normal.hpp"

namespace aesmc {

template <bool FUSED>
__global__ __launch_bounds__(256) void philox_normal_fill_kernel(float *__restrict__ out, int64_t numel,
                                                                 PhiloxStream stream) {
  const PhiloxStream s = philox_resolve(stream);
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint64_t G = s.threads;
  const uint64_t span = 4 * G;
  uint32_t c = 0;
  for (uint64_t e = t; e < (uint64_t)numel; e += span, ++c) {
    const float4 n = philox_normal4<FUSED>(s, t, c);
    out[e] = n.x;
    if (e + G < (uint64_t)numel) out[e + G] = n.y;
    if (e + 2 * G < (uint64_t)numel) out[e + 2 * G] = n.z;
    if (e + 3 * G < (uint64_t)numel) out[e + 3 * G] = n.w;
  }
}

}  // namespace aesmc

extern "C" int aesmc_philox_normal_fill(void *out, int64_t numel, uint64_t seed, uint64_t offset, int64_t threads,
                                        int variant, const uint64_t *rng_state, void *stream) {
  using namespace aesmc;
  if (out == nullptr || numel < 0 || threads <= 0 || (threads % 256) != 0 || threads > 0x7fffffffLL ||
      (offset & 3u) != 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (numel == 0) return AESMC_OK;
  if ((((uintptr_t)rng_state) & 7u) != 0) return AESMC_ERR_INVALID_ARGUMENT;
  const PhiloxStream s = philox_stream(seed, offset, threads, rng_state);
  const dim3 grid((unsigned)(threads / 256));
  hipStream_t hs = static_cast<hipStream_t>(stream);
  if (variant == 0)
    hipLaunchKernelGGL(philox_normal_fill_kernel<true>, grid, dim3(256), 0, hs, static_cast<float *>(out), numel, s);
  else
    hipLaunchKernelGGL(philox_normal_fill_kernel<false>, grid, dim3(256), 0, hs, static_cast<float *>(out), numel, s);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}
