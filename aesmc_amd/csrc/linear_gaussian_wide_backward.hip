// The element-wise parts of the wide step's backward (rows of 128 float32 values: BASELINE.json configs[4]) — autograd of
// aesmc/state.py:114-155 (`log_prob` of the transition and the emission) and :98 (the reparameterised draw) for one
// timestep, between the matrix products that the GEMM library runs (aesmc_amd/_kernels.py: affine_step_backward_wide).
//
// A residual d = value - location comes out of a product's epilogue as a [B,K,128] tensor.  Turning it into the location's
// adjoint u = g d / s^2 (g: the gradient arriving at the particle's log-weight), taking sum_j d_j^2 per particle (the
// scale's gradient), the adjoint's sum over a batch row's particles (the offset's gradient) and, for the transition,
// folding -u into the gradient that arrives at x_t were eight passes over [B,K,128] tensors as PyTorch operations
// (6.4 GB per timestep at B=64, K=16384); here two launches that read and write each tensor once (3.2 GB).
//
// Mapping: a particle's row is 512 bytes = 32 lanes x 16 bytes, a wavefront moves two rows per instruction, a workgroup of
// 256 lanes a tile of 256 particles in 32 trips (K a multiple of 256: a tile lies inside one batch row).  A lane keeps the
// running sum of its four columns over the tile; the eight lanes that share columns meet in LDS and are added in lane
// order; the tile's 128 column sums go to a [B, K / 256, 128] array that the caller adds over its tiles (a fixed order:
// reproducible).  sum_j d_j^2 of a particle: four squares per lane, then the 32 lanes' values by a butterfly.
#include "linear_gaussian.hpp"

namespace aesmc {

typedef float wb_f4 __attribute__((ext_vector_type(4)));
constexpr int kWbDim = 128;
constexpr int kWbTile = 256;      // particles per workgroup
constexpr int kWbThreads = 256;

__device__ __forceinline__ float wb_half_wave_sum(float v) {      // over the 32 lanes that share a particle, to all of them
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// the tile's column sums out of the lanes' partial sums: lanes (p, c) with p = 0 .. 7 (tid >> 5) and c = tid & 31 hold the
// sums of columns 4 c .. 4 c + 3 over their particles; added in the order p = 0, 1, ... and stored by the first 32 lanes
__device__ __forceinline__ void wb_store_column_sums(const wb_f4 &mine, float *lds, float *out128) {
  const uint32_t tid = threadIdx.x;
  reinterpret_cast<wb_f4 *>(lds)[tid] = mine;
  __syncthreads();
  if (tid < 32u) {
    wb_f4 total = reinterpret_cast<const wb_f4 *>(lds)[tid];
#pragma unroll
    for (int p = 1; p < 8; ++p) {
      const wb_f4 part = reinterpret_cast<const wb_f4 *>(lds)[32 * p + tid];
#pragma unroll
      for (int r = 0; r < 4; ++r) total[r] = total[r] + part[r];
    }
    reinterpret_cast<wb_f4 *>(out128)[tid] = total;
  }
  __syncthreads();
}

// u <- weight d / s^2 in place with d = u as it came, or — `base` given: one row of 128 values per batch row — d = base[b] - u
// (u then holds the LOCATION and base the value minus the map's offset: no [B,K,128] copy of a broadcast row for a product's
// epilogue to start from); sq[n] = sum_j d_j^2; rows[b, tile, :] = sum over the tile of the new u
__global__ __launch_bounds__(kWbThreads) void wide_adjoint_scale_kernel(float *__restrict__ u, const float *__restrict__ weight,
                                                                         const float *__restrict__ scale,
                                                                         const float *__restrict__ base, int64_t base_sb,
                                                                         uint32_t tiles_per_row,
                                                                         float *__restrict__ sq, float *__restrict__ rows) {
  __shared__ __attribute__((aligned(16))) float lds[kWbThreads * 4];
  const uint32_t tid = threadIdx.x, c = tid & 31u, p = tid >> 5;
  const int64_t first = (int64_t)blockIdx.x * kWbTile;
  const float s = scale[0];
  const float var = s * s;
  wb_f4 colsum = {0.0f, 0.0f, 0.0f, 0.0f};
  wb_f4 from = {0.0f, 0.0f, 0.0f, 0.0f};
  if (base != nullptr) from = reinterpret_cast<const wb_f4 *>(base + (int64_t)(blockIdx.x / tiles_per_row) * base_sb)[c];
#pragma unroll 4
  for (int trip = 0; trip < kWbTile / 8; ++trip) {
    const int64_t n = first + 8 * trip + p;
    wb_f4 *at = reinterpret_cast<wb_f4 *>(u + n * kWbDim) + c;
    wb_f4 d = *at;
    if (base != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) d[r] = from[r] - d[r];
    }
    const float f = weight[n] / var;
    float part = 0.0f;
    wb_f4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      part = fma_t(d[r], d[r], part);
      v[r] = d[r] * f;
      colsum[r] = colsum[r] + v[r];
    }
    *at = v;
    if (sq != nullptr) {
      part = wb_half_wave_sum(part);
      if (c == 0u) sq[n] = part;
    }
  }
  if (rows != nullptr) wb_store_column_sums(colsum, lds, rows + (int64_t)blockIdx.x * kWbDim);
}

// the transition's residual d (in u_p; or — `value` given — u_p holds the LOCATION and d = (value - base[b]) - u_p, base one
// row per batch row or null) and the gradient arriving at x_t (in at_x; `add` [B,K,128] or null is added to it first):
//   u_p <- weight d / s^2,  at_x <- (add + at_x) - u_p,  sq[n] = sum_j d_j^2,  rows_p / rows_x[b, tile, :] = the tile's sums of both
__global__ __launch_bounds__(kWbThreads) void wide_adjoint_merge_kernel(float *__restrict__ u_p, float *__restrict__ at_x,
                                                                         const float *__restrict__ weight,
                                                                         const float *__restrict__ scale,
                                                                         const float *__restrict__ value,
                                                                         const float *__restrict__ base, int64_t base_sb,
                                                                         uint32_t tiles_per_row, const float *__restrict__ add,
                                                                         float *__restrict__ sq, float *__restrict__ rows_p,
                                                                         float *__restrict__ rows_x) {
  __shared__ __attribute__((aligned(16))) float lds[kWbThreads * 4];
  const uint32_t tid = threadIdx.x, c = tid & 31u, p = tid >> 5;
  const int64_t first = (int64_t)blockIdx.x * kWbTile;
  const float s = scale[0];
  const float var = s * s;
  wb_f4 sum_p = {0.0f, 0.0f, 0.0f, 0.0f}, sum_x = sum_p;
  wb_f4 from = {0.0f, 0.0f, 0.0f, 0.0f};
  if (base != nullptr) from = reinterpret_cast<const wb_f4 *>(base + (int64_t)(blockIdx.x / tiles_per_row) * base_sb)[c];
#pragma unroll 4
  for (int trip = 0; trip < kWbTile / 8; ++trip) {
    const int64_t n = first + 8 * trip + p;
    wb_f4 *at_p = reinterpret_cast<wb_f4 *>(u_p + n * kWbDim) + c;
    wb_f4 *at_g = reinterpret_cast<wb_f4 *>(at_x + n * kWbDim) + c;
    wb_f4 d = *at_p;
    if (value != nullptr) {
      const wb_f4 v0 = reinterpret_cast<const wb_f4 *>(value + n * kWbDim)[c];
#pragma unroll
      for (int r = 0; r < 4; ++r) d[r] = (v0[r] - from[r]) - d[r];
    }
    wb_f4 arriving = *at_g;
    if (add != nullptr) {      // (what later steps sent to x_t, added here instead of copied for a product's epilogue to start from)
      const wb_f4 more = reinterpret_cast<const wb_f4 *>(add + n * kWbDim)[c];
#pragma unroll
      for (int r = 0; r < 4; ++r) arriving[r] = more[r] + arriving[r];
    }
    const float f = weight[n] / var;
    float part = 0.0f;
    wb_f4 v, w;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      part = fma_t(d[r], d[r], part);
      v[r] = d[r] * f;
      w[r] = arriving[r] - v[r];
      sum_p[r] = sum_p[r] + v[r];
      sum_x[r] = sum_x[r] + w[r];
    }
    *at_p = v;
    *at_g = w;
    if (sq != nullptr) {
      part = wb_half_wave_sum(part);
      if (c == 0u) sq[n] = part;
    }
  }
  if (rows_p != nullptr) wb_store_column_sums(sum_p, lds, rows_p + (int64_t)blockIdx.x * kWbDim);
  if (rows_x != nullptr) wb_store_column_sums(sum_x, lds, rows_x + (int64_t)blockIdx.x * kWbDim);
}

static inline bool wb_shape_ok(int64_t B, int64_t K) { return B > 0 && K > 0 && K % kWbTile == 0 && B * K / kWbTile < (1ll << 31); }

}  // namespace aesmc

using namespace aesmc;

extern "C" int64_t aesmc_wide_adjoint_tile(void) { return kWbTile; }

extern "C" int aesmc_wide_adjoint_scale(void *u, const void *weight, const void *scale, const void *base, int64_t base_stride_b,
                                        void *out_sq, void *out_rows, int64_t B, int64_t K, void *stream) {
  if (u == nullptr || weight == nullptr || scale == nullptr || B < 0 || K < 0) return AESMC_ERR_INVALID_ARGUMENT;
  if (!aligned16(u) || (out_rows != nullptr && !aligned16(out_rows)) ||
      (base != nullptr && (!aligned16(base) || (base_stride_b % 4) != 0)))
    return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0) return AESMC_OK;
  if (!wb_shape_ok(B, K)) return AESMC_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(wide_adjoint_scale_kernel, dim3((unsigned)(B * K / kWbTile)), dim3(kWbThreads), 0,
                     static_cast<hipStream_t>(stream), static_cast<float *>(u), static_cast<const float *>(weight),
                     static_cast<const float *>(scale), static_cast<const float *>(base), base_stride_b,
                     (uint32_t)(K / kWbTile), static_cast<float *>(out_sq), static_cast<float *>(out_rows));
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

extern "C" int aesmc_wide_adjoint_merge(void *u_p, void *at_x, const void *weight, const void *scale, const void *value,
                                        const void *base, int64_t base_stride_b, const void *add, void *out_sq,
                                        void *out_rows_p, void *out_rows_x, int64_t B, int64_t K, void *stream) {
  if (u_p == nullptr || at_x == nullptr || u_p == at_x || weight == nullptr || scale == nullptr || B < 0 || K < 0 ||
      (base != nullptr && value == nullptr) || value == u_p || value == at_x || add == u_p || add == at_x)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (!aligned16(u_p) || !aligned16(at_x) || (out_rows_p != nullptr && !aligned16(out_rows_p)) ||
      (out_rows_x != nullptr && !aligned16(out_rows_x)) || (value != nullptr && !aligned16(value)) ||
      (add != nullptr && !aligned16(add)) ||
      (base != nullptr && (!aligned16(base) || (base_stride_b % 4) != 0)))
    return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0) return AESMC_OK;
  if (!wb_shape_ok(B, K)) return AESMC_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(wide_adjoint_merge_kernel, dim3((unsigned)(B * K / kWbTile)), dim3(kWbThreads), 0,
                     static_cast<hipStream_t>(stream), static_cast<float *>(u_p), static_cast<float *>(at_x),
                     static_cast<const float *>(weight), static_cast<const float *>(scale), static_cast<const float *>(value),
                     static_cast<const float *>(base), base_stride_b, (uint32_t)(K / kWbTile), static_cast<const float *>(add),
                     static_cast<float *>(out_sq), static_cast<float *>(out_rows_p), static_cast<float *>(out_rows_x));
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}
