// K18g's instantiations, the combine launch and their launcher (linear_gaussian_wide_generic.hpp): the emission density
// and the log-weight of a linear-Gaussian step with rows of 20 .. 256 values — aesmc/inference.py:112-126.
#include "linear_gaussian_wide_generic.hpp"

namespace aesmc {

// log w out of the particles' records, when the emission's output rows were cut into chunks (dy > 128)
__global__ __launch_bounds__(256) void wideg_combine_kernel(WideGArgs a) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < (int64_t)a.B * a.K) a.out_lw[p] = wideg_log_weight(a, a.sums + p * a.sums_stride, 0.0f, false);
}

template <int DXP, int MC>
static int wideg_emit_one(const WideGArgs &a, hipStream_t s) {
  static bool raised[64] = {};
  if (!lg_raise_lds_limit(reinterpret_cast<const void *>(&affine_wideg_emission_kernel<DXP, MC>), raised))
    return AESMC_ERR_LAUNCH;
  const int64_t tiles = (int64_t)a.B * a.tiles_per_row;
  const int64_t groups = (tiles + kWgThreads / 64 - 1) / (kWgThreads / 64);
  const int64_t per_chunk = std::max<int64_t>(1, lg_cu_count() / (int64_t)a.chunks_emit);
  const dim3 grid((unsigned)std::min<int64_t>(per_chunk, groups), a.chunks_emit);
  const size_t lds = sizeof(float) * (size_t)MC * (DXP + 4);
  hipLaunchKernelGGL((affine_wideg_emission_kernel<DXP, MC>), grid, dim3(kWgThreads), lds, s, a);
  if (hipGetLastError() != hipSuccess) return AESMC_ERR_LAUNCH;
  if (a.chunks_emit > 1u) {
    const int64_t N = (int64_t)a.B * a.K;
    hipLaunchKernelGGL(wideg_combine_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, a);
    if (hipGetLastError() != hipSuccess) return AESMC_ERR_LAUNCH;
  }
  return AESMC_OK;
}

int wideg_launch_emission(const WideGArgs &a, int dxp, hipStream_t s) {
  const bool small = wideg_emit_chunk(a.dout) == 64;
#define WIDEG_EMIT(DXP)                                                                           \
  case DXP:                                                                                       \
    return small ? wideg_emit_one<DXP, 64>(a, s) : wideg_emit_one<DXP, 128>(a, s)
  switch (dxp) {
    WIDEG_EMIT(32);
    WIDEG_EMIT(48);
    WIDEG_EMIT(64);
    WIDEG_EMIT(96);
    WIDEG_EMIT(128);
    WIDEG_EMIT(192);
    WIDEG_EMIT(256);
    default:
      return AESMC_ERR_UNSUPPORTED;
  }
#undef WIDEG_EMIT
}

}  // namespace aesmc
