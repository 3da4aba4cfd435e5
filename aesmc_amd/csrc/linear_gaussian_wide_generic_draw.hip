// K17g's instantiations and launcher (linear_gaussian_wide_generic.hpp): the draw of a linear-Gaussian step with rows of
// 20 .. 256 values on the fp32 matrix cores — aesmc/inference.py:102-111 (resample, propose, sample) for such a model.
#include "linear_gaussian_wide_generic.hpp"

namespace aesmc {

template <int DXP, int MC, bool GATHER>
static int wideg_draw_one(const WideGArgs &a, hipStream_t s) {
  static bool raised[64] = {};
  if (!lg_raise_lds_limit(reinterpret_cast<const void *>(&affine_wideg_draw_kernel<DXP, MC, GATHER>), raised))
    return AESMC_ERR_LAUNCH;
  const int64_t tiles = (int64_t)a.B * a.tiles_per_row;
  const int64_t groups = (tiles + kWgThreads / 64 - 1) / (kWgThreads / 64);
  // every chunk of a stretch of tiles resident at once (one workgroup per CU: the weights fill its LDS), so that the
  // chunks' reads of the same rows of x_{t-1} meet in L2
  const int64_t per_chunk = std::max<int64_t>(1, lg_cu_count() / (int64_t)a.chunks_draw);
  const dim3 grid((unsigned)std::min<int64_t>(per_chunk, groups), a.chunks_draw);
  const size_t lds = sizeof(float) * 2 * (size_t)MC * (DXP + 4);
  hipLaunchKernelGGL((affine_wideg_draw_kernel<DXP, MC, GATHER>), grid, dim3(kWgThreads), lds, s, a);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

int wideg_launch_draw(const WideGArgs &a, int dxp, bool gather, hipStream_t s) {
#define WIDEG_DRAW(DXP, MC)                                                                       \
  case DXP:                                                                                       \
    return gather ? wideg_draw_one<DXP, MC, true>(a, s) : wideg_draw_one<DXP, MC, false>(a, s)
  switch (dxp) {
    WIDEG_DRAW(32, 32);
    WIDEG_DRAW(48, 48);
    WIDEG_DRAW(64, 64);
    WIDEG_DRAW(96, 96);
    WIDEG_DRAW(128, 128);
    WIDEG_DRAW(192, 64);      // (144 unrolled operand groups at MC = 96: the compiler keeps the loop and spills the tiles)
    WIDEG_DRAW(256, 64);
    default:
      return AESMC_ERR_UNSUPPORTED;
  }
#undef WIDEG_DRAW
}

}  // namespace aesmc
