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
// out is dense [B,K,D]; eps, loc and scale are [B,K,D] views by element strides (0 = broadcast),
// as in K4.  Besides dense noise the kernel knows the TRANSPOSED layout: a BATCH_EXPANDED
// distribution is sampled as [K,B,...] and handed on as a transposed view (aesmc/state.py:102-103),
// which every later consumer would read with a stride of B rows; the draw is instead written
// straight into [B,K,...] order through an LDS tile, so the time-0 latent is dense from the start.
#include "common.hpp"
#include "philox_normal.hpp"

namespace aesmc {

struct RsStrides {
  int64_t b, k, d;
};

constexpr int kRsBlock = 256;

// 16 bytes of T moved as one access.
template <typename T> struct alignas(16) Pack16 {
  T v[Vec16<T>::N];
};

// loc dense like eps; scale varies along j only (stride_d 0 or 1, the other strides 0), or is
// dense like loc (SCALE_DENSE: the [B,K,D] output of a proposal network, also read again by K5).
template <typename T, bool SCALE_DENSE>
__global__ __launch_bounds__(kRsBlock) void normal_rsample_dense_kernel(
    const T *__restrict__ eps, const T *__restrict__ loc, const T *__restrict__ scale,
    T *__restrict__ out, uint64_t n, uint32_t D, uint32_t scale_stride_d, int stream) {
  constexpr uint32_t V = Vec16<T>::N;
  const uint64_t first = ((uint64_t)blockIdx.x * kRsBlock + threadIdx.x) * V;
  if (first >= n) return;
  if (first + V <= n) {
    const Pack16<T> e = load16(reinterpret_cast<const Pack16<T> *>(eps + first), stream);   // read once
    const Pack16<T> m = *reinterpret_cast<const Pack16<T> *>(loc + first);                // K5 reads it again
    Pack16<T> r;
    if constexpr (SCALE_DENSE) {
      const Pack16<T> sg = *reinterpret_cast<const Pack16<T> *>(scale + first);
#pragma unroll
      for (uint32_t i = 0; i < V; ++i) r.v[i] = m.v[i] + e.v[i] * sg.v[i];
    } else {
      uint32_t j = (uint32_t)(first % D);
#pragma unroll
      for (uint32_t i = 0; i < V; ++i) {
        const T sigma = scale[(uint64_t)j * scale_stride_d];
        r.v[i] = m.v[i] + e.v[i] * sigma;
        j = (j + 1 == D) ? 0u : j + 1;
      }
    }
    *reinterpret_cast<Pack16<T> *>(out + first) = r;
  } else {
    for (uint64_t i = first; i < n; ++i)
      out[i] = loc[i] + eps[i] * (SCALE_DENSE ? scale[i] : scale[(uint64_t)(i % D) * scale_stride_d]);
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

// eps laid out [K,B,D] dense (read as the view eps[b,k,j] = base[(k B + b) D + j]): a workgroup
// owns a TILE x TILE block of (k, b) rows, reads it in runs along (b, j), parks the finished draws
// in LDS and writes them in runs along (k, j).
template <typename T>
__global__ __launch_bounds__(kRsBlock) void normal_rsample_transposed_kernel(
    const T *__restrict__ eps, const T *__restrict__ loc, const T *__restrict__ scale,
    T *__restrict__ out, uint32_t B, uint32_t K, uint32_t D, uint32_t tile, RsStrides sm, RsStrides ss) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
  T *park = reinterpret_cast<T *>(rs_smem);             // [tile][tile * D + 1]
  const uint32_t b0 = blockIdx.x * tile, k0 = blockIdx.y * tile;
  const uint32_t nb = min(tile, B - b0), nk = min(tile, K - k0);
  const uint32_t pitch = tile * D + 1;
  const uint32_t run_in = nb * D;
  for (uint32_t t = threadIdx.x; t < nk * run_in; t += kRsBlock) {
    const uint32_t kk = t / run_in, rest = t - kk * run_in;
    const uint32_t bb = rest / D, j = rest - bb * D;
    const int64_t b = b0 + bb, k = k0 + kk;
    const T noise = eps[((uint64_t)k * B + (uint64_t)b) * D + j];
    const T mu = loc[b * sm.b + k * sm.k + (int64_t)j * sm.d];
    const T sigma = scale[b * ss.b + k * ss.k + (int64_t)j * ss.d];
    park[bb * pitch + kk * D + j] = mu + noise * sigma;
  }
  __syncthreads();
  const uint32_t run_out = nk * D;
  for (uint32_t t = threadIdx.x; t < nb * run_out; t += kRsBlock) {
    const uint32_t bb = t / run_out, rest = t - bb * run_out;          // rest = kk * D + j
    out[((uint64_t)(b0 + bb) * K + k0) * D + rest] = park[bb * pitch + rest];
  }
}

// One value per particle (the scalar IWAE model: 134 MB of noise at B=4096 K=8192): the classic
// 64 x 64 transpose — lane x walks b on the way in and k on the way out, four rows per trip, no
// index divisions.  For B and K multiples of 64; other shapes take the general kernel above.
template <typename T>
__global__ __launch_bounds__(kRsBlock) void normal_rsample_transposed_d1_kernel(
    const T *__restrict__ eps, const T *__restrict__ loc, const T *__restrict__ scale,
    T *__restrict__ out, uint32_t B, uint32_t K, RsStrides sm, RsStrides ss) {
  constexpr uint32_t TILE = 64;
  __shared__ T park[TILE][TILE + 1];
  const uint32_t b0 = blockIdx.x * TILE, k0 = blockIdx.y * TILE;
  const uint32_t tx = threadIdx.x % TILE, ty = threadIdx.x / TILE;     // 256 lanes: 4 rows per trip
#pragma unroll 4
  for (uint32_t r = ty; r < TILE; r += kRsBlock / TILE) {
    const int64_t b = b0 + tx, k = k0 + r;
    const T noise = eps[(uint64_t)k * B + (uint64_t)b];
    park[tx][r] = loc[b * sm.b + k * sm.k] + noise * scale[b * ss.b + k * ss.k];
  }
  __syncthreads();
#pragma unroll 4
  for (uint32_t r = ty; r < TILE; r += kRsBlock / TILE)
    out[(uint64_t)(b0 + r) * K + k0 + tx] = park[r][tx];
}

template <typename T>
static int launch_rsample(const aesmc_view3 &eps_view, const aesmc_view3 &loc, const aesmc_view3 &scale,
                          void *out, int64_t B, int64_t K, int64_t D, hipStream_t stream) {
  const uint64_t n = (uint64_t)B * K * D;
  const T *e = static_cast<const T *>(eps_view.ptr);
  const T *m = static_cast<const T *>(loc.ptr);
  const T *s = static_cast<const T *>(scale.ptr);
  T *o = static_cast<T *>(out);
  const bool eps_dense = (D == 1 || eps_view.stride_d == 1) && (K == 1 || eps_view.stride_k == D) &&
                         (B == 1 || eps_view.stride_b == K * D);
  if (!eps_dense) {
    const bool eps_transposed = (D == 1 || eps_view.stride_d == 1) && eps_view.stride_b == D &&
                                eps_view.stride_k == B * D;
    uint32_t tile = 0;
    if (eps_transposed) {
      const uint32_t budget = 48 * 1024 / sizeof(T);
      for (uint32_t cand : {64u, 32u, 16u, 8u})
        if ((uint64_t)cand * (cand * D + 1) <= budget) {
          tile = cand;
          break;
        }
    }
    if (eps_transposed && D == 1 && B % 64 == 0 && K % 64 == 0 && K / 64 <= 65535) {
      hipLaunchKernelGGL(normal_rsample_transposed_d1_kernel<T>, dim3((unsigned)(B / 64), (unsigned)(K / 64)),
                         dim3(kRsBlock), 0, stream, e, m, s, o, (uint32_t)B, (uint32_t)K,
                         RsStrides{loc.stride_b, loc.stride_k, loc.stride_d},
                         RsStrides{scale.stride_b, scale.stride_k, scale.stride_d});
      return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
    }
    if (tile == 0) return AESMC_ERR_UNSUPPORTED;    // the caller materialises eps and comes back
    const dim3 grid((unsigned)((B + tile - 1) / tile), (unsigned)((K + tile - 1) / tile));
    if (grid.y > 65535u) return AESMC_ERR_UNSUPPORTED;
    const size_t lds = (size_t)tile * (tile * D + 1) * sizeof(T);
    hipLaunchKernelGGL(normal_rsample_transposed_kernel<T>, grid, dim3(kRsBlock), lds, stream, e, m, s, o,
                       (uint32_t)B, (uint32_t)K, (uint32_t)D, tile,
                       RsStrides{loc.stride_b, loc.stride_k, loc.stride_d},
                       RsStrides{scale.stride_b, scale.stride_k, scale.stride_d});
    return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
  }
  const bool loc_dense = (D == 1 || loc.stride_d == 1) && (K == 1 || loc.stride_k == D) &&
                         (B == 1 || loc.stride_b == K * D);
  const bool scale_by_column = (B == 1 || scale.stride_b == 0) && (K == 1 || scale.stride_k == 0) &&
                               (D == 1 || scale.stride_d == 0 || scale.stride_d == 1);
  const bool aligned = ((reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(m) |
                         reinterpret_cast<uintptr_t>(o)) & 15u) == 0;
  const bool scale_dense = !scale_by_column && (D == 1 || scale.stride_d == 1) &&
                           (K == 1 || scale.stride_k == D) && (B == 1 || scale.stride_b == K * D) &&
                           (reinterpret_cast<uintptr_t>(s) & 15u) == 0;
  if (loc_dense && scale_dense && aligned) {
    constexpr uint64_t V = Vec16<T>::N;
    const uint64_t blocks = ((n + V - 1) / V + kRsBlock - 1) / kRsBlock;
    if (blocks >= (1ull << 31)) return AESMC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((normal_rsample_dense_kernel<T, true>), dim3((uint32_t)blocks), dim3(kRsBlock), 0, stream,
                       e, m, s, o, n, (uint32_t)D, 0u, stream_hint(4 * n * sizeof(T)));
    return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
  }
  if (loc_dense && scale_by_column && aligned) {
    constexpr uint64_t V = Vec16<T>::N;
    const uint64_t threads = (n + V - 1) / V;
    const uint64_t blocks = (threads + kRsBlock - 1) / kRsBlock;
    if (blocks >= (1ull << 31)) return AESMC_ERR_UNSUPPORTED;
    const uint32_t sd = (D == 1) ? 0u : (uint32_t)scale.stride_d;
    hipLaunchKernelGGL((normal_rsample_dense_kernel<T, false>), dim3((uint32_t)blocks), dim3(kRsBlock), 0, stream,
                       e, m, s, o, n, (uint32_t)D, sd, stream_hint(3 * n * sizeof(T)));
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

// K6 with the NOISE formed in the launch (philox_normal.hpp): thread t of the G that ATen's `normal_` would have
// launched forms, trip by trip, the four normals that call hands the elements t + G (4 c + i), and leaves
// loc + n * scale there — the noise tensor is neither written nor read.  A wavefront's lanes hold consecutive
// elements (4-byte accesses, contiguous across the wavefront); an element's (b, k, j) advance by constants from one
// normal to the next, so the strided views of loc and scale cost no division per element.
template <bool FUSED>
__global__ __launch_bounds__(kRsBlock) void normal_rsample_drawn_kernel(
    const float *__restrict__ loc, const float *__restrict__ scale, float *__restrict__ out, uint64_t n, uint32_t K,
    uint32_t D, RsStrides sm, RsStrides ss, PhiloxStream stream) {
  const PhiloxStream s = philox_resolve(stream);
  const uint32_t t = blockIdx.x * (uint32_t)kRsBlock + threadIdx.x;
  const uint64_t G = s.threads;
  if (t >= G) return;
  const uint32_t step_row = (uint32_t)(G / D), dj = (uint32_t)(G - (uint64_t)step_row * D);
  const uint32_t db = step_row / K, dk = step_row - db * K;
  const uint32_t row0 = t / D;
  uint32_t j = t - row0 * D, b = row0 / K, k = row0 - b * K;
  uint64_t e = t;
  for (uint32_t c = 0; e < n; ++c) {
    const float4 four = philox_normal4<FUSED>(s, t, c);
    const float normal[4] = {four.x, four.y, four.z, four.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (e < n) {
        const float mu = loc[(int64_t)b * sm.b + (int64_t)k * sm.k + (int64_t)j * sm.d];
        const float sigma = scale[(int64_t)b * ss.b + (int64_t)k * ss.k + (int64_t)j * ss.d];
        out[e] = mu + normal[i] * sigma;
      }
      e += G;
      j += dj;
      k += dk;
      b += db;
      if (j >= D) {
        j -= D;
        ++k;
      }
      if (k >= K) {
        k -= K;
        ++b;
      }
    }
  }
}

}  // namespace aesmc

extern "C" int aesmc_normal_rsample_drawn(int dtype, const aesmc_view3 *loc, const aesmc_view3 *scale, void *out,
                                          int64_t B, int64_t K, int64_t D, uint64_t seed, uint64_t offset,
                                          int64_t threads, int variant, const uint64_t *rng_state, void *stream) {
  using namespace aesmc;
  if (!loc || !scale || !out || !loc->ptr || !scale->ptr || B < 0 || K < 0 || D < 0 || threads <= 0 ||
      (threads % 256) != 0 || threads > 0x7fffffffLL || (offset & 3u) != 0 || (((uintptr_t)rng_state) & 7u) != 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32) return dtype == AESMC_F64 ? AESMC_ERR_UNSUPPORTED : AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0 || D == 0) return AESMC_OK;
  if (K >= (1ll << 31) || D >= (1ll << 31) || B >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  const uint64_t n = (uint64_t)B * (uint64_t)K * (uint64_t)D;
  if (n >= (1ull << 32)) return AESMC_ERR_UNSUPPORTED;
  const PhiloxStream ps = philox_stream(seed, offset, threads, rng_state);
  const RsStrides sm{loc->stride_b, loc->stride_k, loc->stride_d}, ss{scale->stride_b, scale->stride_k, scale->stride_d};
  const dim3 grid((unsigned)(threads / 256));
  hipStream_t hs = static_cast<hipStream_t>(stream);
  if (variant == 0)
    hipLaunchKernelGGL(normal_rsample_drawn_kernel<true>, grid, dim3(kRsBlock), 0, hs, static_cast<const float *>(loc->ptr),
                       static_cast<const float *>(scale->ptr), static_cast<float *>(out), n, (uint32_t)K, (uint32_t)D, sm,
                       ss, ps);
  else
    hipLaunchKernelGGL(normal_rsample_drawn_kernel<false>, grid, dim3(kRsBlock), 0, hs, static_cast<const float *>(loc->ptr),
                       static_cast<const float *>(scale->ptr), static_cast<float *>(out), n, (uint32_t)K, (uint32_t)D, sm,
                       ss, ps);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

extern "C" int aesmc_normal_rsample(int dtype, const aesmc_view3 *eps, const aesmc_view3 *loc,
                                    const aesmc_view3 *scale, void *out, int64_t B, int64_t K, int64_t D,
                                    void *stream) {
  using namespace aesmc;
  if (!eps || !loc || !scale || !out || !eps->ptr || !loc->ptr || !scale->ptr || B < 0 || K < 0 || D < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0 || D == 0) return AESMC_OK;
  if (K >= (1ll << 31) || D >= (1ll << 31) || B >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32) return launch_rsample<float>(*eps, *loc, *scale, out, B, K, D, s);
  if (dtype == AESMC_F64) return launch_rsample<double>(*eps, *loc, *scale, out, B, K, D, s);
  return AESMC_ERR_INVALID_ARGUMENT;
}
