// K6: reparameterised Normal draw  out[b,k,j] = loc[b,k,j] + eps[b,k,j] * scale[b,k,j].
//
// The second half of torch.distributions.Normal.rsample as `state.sample` reaches it
// (aesmc/state.py:61-111 -> torch/distributions/normal.py rsample: `self.loc + eps * self.scale`):
// eager PyTorch runs a broadcast multiply into a temporary and then an add, 5 passes over
// [B,K,D]; this is one pass (eps + loc in, the draw out).  The standard-normal noise itself stays
// PyTorch's (`_standard_normal`, same generator, same stream position), so a seeded run draws the
// same particles as the reference.  The product is rounded before the sum (the build disables
// contraction), so every element equals the eager result bit for bit.
//
// eps and out are dense [B,K,D]; loc and scale are [B,K,D] views by element strides (0 =
// broadcast), as in K4.
#include "common.hpp"

namespace aesmc {

struct RsStrides {
  int64_t b, k, d;
};

constexpr int kRsBlock = 256;

// 16 bytes of T moved as one access.
template <typename T> struct alignas(16) Pack16 {
  T v[Vec16<T>::N];
};

// loc dense like eps; scale varies along j only (stride_d 0 or 1, the other strides 0).
template <typename T>
__global__ __launch_bounds__(kRsBlock) void normal_rsample_dense_kernel(
    const T *__restrict__ eps, const T *__restrict__ loc, const T *__restrict__ scale,
    T *__restrict__ out, uint64_t n, uint32_t D, uint32_t scale_stride_d) {
  constexpr uint32_t V = Vec16<T>::N;
  const uint64_t first = ((uint64_t)blockIdx.x * kRsBlock + threadIdx.x) * V;
  if (first >= n) return;
  if (first + V <= n) {
    const Pack16<T> e = *reinterpret_cast<const Pack16<T> *>(eps + first);
    const Pack16<T> m = *reinterpret_cast<const Pack16<T> *>(loc + first);
    Pack16<T> r;
    uint32_t j = (uint32_t)(first % D);
#pragma unroll
    for (uint32_t i = 0; i < V; ++i) {
      const T sigma = scale[(uint64_t)j * scale_stride_d];
      r.v[i] = m.v[i] + e.v[i] * sigma;
      j = (j + 1 == D) ? 0u : j + 1;
    }
    *reinterpret_cast<Pack16<T> *>(out + first) = r;
  } else {
    for (uint64_t i = first; i < n; ++i)
      out[i] = loc[i] + eps[i] * scale[(uint64_t)(i % D) * scale_stride_d];
  }
}

// Any broadcast pattern: one element per lane.
template <typename T>
__global__ __launch_bounds__(kRsBlock) void normal_rsample_strided_kernel(
    const T *__restrict__ eps, const T *__restrict__ loc, const T *__restrict__ scale,
    T *__restrict__ out, uint64_t n, uint32_t K, uint32_t D, RsStrides sm, RsStrides ss) {
  const uint64_t i = (uint64_t)blockIdx.x * kRsBlock + threadIdx.x;
  if (i >= n) return;
  const uint64_t row = i / D;
  const int64_t j = (int64_t)(i - row * D);
  const int64_t b = (int64_t)(row / K);
  const int64_t k = (int64_t)(row - (uint64_t)b * K);
  const T mu = loc[b * sm.b + k * sm.k + j * sm.d];
  const T sigma = scale[b * ss.b + k * ss.k + j * ss.d];
  out[i] = mu + eps[i] * sigma;
}

template <typename T>
static int launch_rsample(const void *eps, const aesmc_view3 &loc, const aesmc_view3 &scale, void *out,
                          int64_t B, int64_t K, int64_t D, hipStream_t stream) {
  const uint64_t n = (uint64_t)B * K * D;
  const T *e = static_cast<const T *>(eps);
  const T *m = static_cast<const T *>(loc.ptr);
  const T *s = static_cast<const T *>(scale.ptr);
  T *o = static_cast<T *>(out);
  const bool loc_dense = (D == 1 || loc.stride_d == 1) && (K == 1 || loc.stride_k == D) &&
                         (B == 1 || loc.stride_b == K * D);
  const bool scale_by_column = (B == 1 || scale.stride_b == 0) && (K == 1 || scale.stride_k == 0) &&
                               (D == 1 || scale.stride_d == 0 || scale.stride_d == 1);
  const bool aligned = ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(m) |
                         reinterpret_cast<uintptr_t>(o)) & 15u) == 0;
  if (loc_dense && scale_by_column && aligned) {
    constexpr uint64_t V = Vec16<T>::N;
    const uint64_t threads = (n + V - 1) / V;
    const uint64_t blocks = (threads + kRsBlock - 1) / kRsBlock;
    if (blocks >= (1ull << 31)) return AESMC_ERR_UNSUPPORTED;
    const uint32_t sd = (D == 1) ? 0u : (uint32_t)scale.stride_d;
    hipLaunchKernelGGL(normal_rsample_dense_kernel<T>, dim3((uint32_t)blocks), dim3(kRsBlock), 0, stream,
                       e, m, s, o, n, (uint32_t)D, sd);
  } else {
    const uint64_t blocks = (n + kRsBlock - 1) / kRsBlock;
    if (blocks >= (1ull << 31)) return AESMC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(normal_rsample_strided_kernel<T>, dim3((uint32_t)blocks), dim3(kRsBlock), 0, stream,
                       e, m, s, o, n, (uint32_t)K, (uint32_t)D,
                       RsStrides{loc.stride_b, loc.stride_k, loc.stride_d},
                       RsStrides{scale.stride_b, scale.stride_k, scale.stride_d});
  }
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

}  // namespace aesmc

extern "C" int aesmc_normal_rsample(int dtype, const void *eps, const aesmc_view3 *loc,
                                    const aesmc_view3 *scale, void *out, int64_t B, int64_t K, int64_t D,
                                    void *stream) {
  using namespace aesmc;
  if (!eps || !loc || !scale || !out || !loc->ptr || !scale->ptr || B < 0 || K < 0 || D < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0 || D == 0) return AESMC_OK;
  if (K >= (1ll << 31) || D >= (1ll << 31) || B >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32) return launch_rsample<float>(eps, *loc, *scale, out, B, K, D, s);
  if (dtype == AESMC_F64) return launch_rsample<double>(eps, *loc, *scale, out, B, K, D, s);
  return AESMC_ERR_INVALID_ARGUMENT;
}
