// K4: summed Normal log-density  out[b,k] = sum_j log N(value[b,k,j]; loc[b,k,j], scale[b,k,j])
// and its backward.
//
// Replaces, inside aesmc/state.py:114-155 (`state.log_prob`), the chain of ~11 elementwise /
// reduction launches that torch.distributions.Normal.log_prob + `.view(B,K,-1).sum(2)` expands to
// (sub, pow, neg, mul, div, log, sub, sub, sum ...), each a full pass over [B,K,d].  One pass here:
// value + loc (+ scale when it is a tensor) in, 4 B per particle out.
//
// Per element the arithmetic is PyTorch's, operation for operation (torch/distributions/normal.py
// log_prob):  -((v - mu)^2) / (2 * sigma^2) - log(sigma) - log(sqrt(2 pi)), so element values agree
// with the eager path to the last place; only the order of the d-sum differs.
//
// Operands are [B,K,D] views given by element strides (0 = broadcast), which covers every
// BatchShapeMode of the reference without materialising an expand: loc [D] (NOT_EXPANDED),
// loc [B,D] (BATCH_EXPANDED), loc [B,K,D] (FULLY_EXPANDED), value = observation expanded over K,
// the transposed time-0 latent, scalar scales.
#include "common.hpp"
namespace aesmc {

struct Strides3 {
  int64_t b, k, d;
};

constexpr int kLpBlock = 256;
constexpr int kTileBytes = 32 * 1024;  // LDS tile budget per workgroup

// Particles per tile: as many as the LDS budget holds, as a power of two (a dense operand's tile
// then starts on a 16-byte boundary whenever its batch row does) between 256 — one per lane — and
// 2048: with a handful of values per particle a 256-particle tile is a few hundred bytes of work
// per workgroup and the launch is all scheduling (d=1, 33 M particles: 346 us -> see DESIGN.md).
static inline uint32_t particles_per_tile(int64_t budget_elems, int64_t elems_per_particle, int64_t K) {
  int64_t fit = budget_elems / elems_per_particle;
  uint32_t P = 1;
  while ((int64_t)P * 2 <= fit && P < 2048u) P *= 2;
  if (P > K) P = (uint32_t)K;
  return P;
}

template <typename T> struct NormConst;
template <> struct NormConst<float> {
  // (float)math.log(math.sqrt(2 * math.pi)) as PyTorch's scalar operand is narrowed
  static __device__ __forceinline__ float half_log_2pi() { return 0.9189385332046727f; }
};
template <> struct NormConst<double> {
  static __device__ __forceinline__ double half_log_2pi() { return 0.9189385332046727; }
};

template <typename T> __device__ __forceinline__ T normal_logpdf(T v, T mu, T sigma) {
  const T diff = v - mu;
  const T var = sigma * sigma;
  return (-(diff * diff)) / (T(2) * var) - Num<T>::log(sigma) - NormConst<T>::half_log_2pi();
}

__device__ __forceinline__ uint32_t pad_index(uint32_t e) { return e + (e >> 5); }

// Small D: a workgroup owns P consecutive particles of one batch row.  Lanes walk the tile's P*D
// elements in memory order (coalesced for dense operands), park the per-element log-densities in
// LDS, then one lane per particle adds its D values.
template <typename T>
__global__ __launch_bounds__(kLpBlock) void normal_lps_tile_kernel(
    const T *__restrict__ value, const T *__restrict__ loc, const T *__restrict__ scale,
    T *__restrict__ out, uint32_t K, uint32_t D, uint32_t P, uint32_t tiles_per_row, Strides3 sv,
    Strides3 sm, Strides3 ss) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lps_smem[];
  T *tile = reinterpret_cast<T *>(lps_smem);
  const uint32_t b = blockIdx.x / tiles_per_row;
  const uint32_t k0 = (blockIdx.x - b * tiles_per_row) * P;
  const uint32_t np = min(P, K - k0);
  const uint32_t ne = np * D;
  const T *vb = value + (int64_t)b * sv.b;
  const T *mb = loc + (int64_t)b * sm.b;
  const T *sb = scale + (int64_t)b * ss.b;

  uint32_t e = threadIdx.x;
  uint32_t kk = e / D, j = e - kk * D;
  const uint32_t dk = kLpBlock / D, dj = kLpBlock - dk * D;
  for (; e < ne; e += kLpBlock) {
    const int64_t k = k0 + kk;
    const T f = normal_logpdf<T>(vb[k * sv.k + (int64_t)j * sv.d], mb[k * sm.k + (int64_t)j * sm.d],
                                 sb[k * ss.k + (int64_t)j * ss.d]);
    tile[pad_index(e)] = f;
    kk += dk;
    j += dj;
    if (j >= D) {
      j -= D;
      ++kk;
    }
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < np; p += kLpBlock) {
    T sum = T(0);
    const uint32_t base = p * D;
    for (uint32_t jj = 0; jj < D; ++jj) sum += tile[pad_index(base + jj)];
    out[(int64_t)b * K + k0 + p] = sum;
  }
}

// Large D: one wavefront per particle, lanes stride over D, shuffle reduction.
template <typename T>
__global__ __launch_bounds__(kLpBlock) void normal_lps_wave_kernel(
    const T *__restrict__ value, const T *__restrict__ loc, const T *__restrict__ scale,
    T *__restrict__ out, int64_t particles, uint32_t K, uint32_t D, Strides3 sv, Strides3 sm,
    Strides3 ss) {
  const int64_t p = (int64_t)blockIdx.x * (kLpBlock / kWave) + threadIdx.x / kWave;
  if (p >= particles) return;
  const int lane = threadIdx.x % kWave;
  const int64_t b = p / K, k = p - b * K;
  const T *vp = value + b * sv.b + k * sv.k;
  const T *mp = loc + b * sm.b + k * sm.k;
  const T *sp = scale + b * ss.b + k * ss.k;
  T sum = T(0);
  for (uint32_t j = lane; j < D; j += kWave)
    sum += normal_logpdf<T>(vp[(int64_t)j * sv.d], mp[(int64_t)j * sm.d], sp[(int64_t)j * ss.d]);
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, kWave);
  if (lane == 0) out[p] = sum;
}

// Backward: with g = grad_out[b,k], z = (v - mu) / sigma^2 :
//   d/dv = -g z,   d/dmu = g z,   d/dsigma = g ((v - mu)^2 / sigma^3 - 1 / sigma).
// Each requested gradient is written densely [B,K,D]; autograd's expand-backward reduces the
// broadcast operands.
template <typename T>
__global__ __launch_bounds__(kLpBlock) void normal_lps_bwd_kernel(
    const T *__restrict__ value, const T *__restrict__ loc, const T *__restrict__ scale,
    const T *__restrict__ grad_out, T *__restrict__ grad_value, T *__restrict__ grad_loc,
    T *__restrict__ grad_scale, int64_t total, uint32_t K, uint32_t D, Strides3 sv, Strides3 sm,
    Strides3 ss) {
  const int64_t stride = (int64_t)gridDim.x * kLpBlock;
  for (int64_t e = (int64_t)blockIdx.x * kLpBlock + threadIdx.x; e < total; e += stride) {
    const int64_t p = e / D;
    const uint32_t j = (uint32_t)(e - p * D);
    const int64_t b = p / K, k = p - b * K;
    const T v = value[b * sv.b + k * sv.k + (int64_t)j * sv.d];
    const T mu = loc[b * sm.b + k * sm.k + (int64_t)j * sm.d];
    const T sigma = scale[b * ss.b + k * ss.k + (int64_t)j * ss.d];
    const T g = grad_out[p];
    const T diff = v - mu;
    const T var = sigma * sigma;
    const T gz = g * (diff / var);
    if (grad_value) grad_value[e] = -gz;
    if (grad_loc) grad_loc[e] = gz;
    if (grad_scale) grad_scale[e] = g * ((diff * diff) / (var * sigma) - T(1) / sigma);
  }
}

// Small D, SCALAR scale (one value for the whole tensor — Normal(loc, 0.7)), and `value` and / or
// `loc` dense ([K, D] contiguous inside a batch row): the dense operands are streamed with 16-byte
// loads; log(scale) and 2*scale^2 are formed once per lane.  Same arithmetic, same LDS reduction.
template <typename T, bool VALUE_DENSE, bool LOC_DENSE>
__global__ __launch_bounds__(kLpBlock) void normal_lps_tile_vec_kernel(
    const T *__restrict__ value, const T *__restrict__ loc, const T *__restrict__ scale,
    T *__restrict__ out, uint32_t K, uint32_t D, uint32_t P, uint32_t tiles_per_row, Strides3 sv,
    Strides3 sm) {
  constexpr int N = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char lps_smem[];
  T *tile = reinterpret_cast<T *>(lps_smem);
  const uint32_t b = blockIdx.x / tiles_per_row;
  const uint32_t k0 = (blockIdx.x - b * tiles_per_row) * P;
  const uint32_t np = min(P, K - k0);
  const uint32_t ne = np * D;
  const T *vb = value + (int64_t)b * sv.b;
  const T *mb = loc + (int64_t)b * sm.b;
  const T sigma = scale[0];
  const T two_var = T(2) * (sigma * sigma);
  const T log_sigma = Num<T>::log(sigma);
  const T half_log_2pi = NormConst<T>::half_log_2pi();

  const uint32_t nvec = ne / N;
  uint32_t e = threadIdx.x * N;
  uint32_t kk = e / D, j = e - kk * D;
  const uint32_t step = kLpBlock * N, dk = step / D, dj = step - dk * D;
  for (uint32_t v = threadIdx.x; v < nvec; v += kLpBlock) {
    T x[N], m[N];
    if constexpr (VALUE_DENSE) {
      const V packed = *reinterpret_cast<const V *>(vb + (int64_t)k0 * sv.k + e);
#pragma unroll
      for (int q = 0; q < N; ++q) x[q] = Vec16<T>::get(packed, q);
    }
    if constexpr (LOC_DENSE) {
      const V packed = *reinterpret_cast<const V *>(mb + (int64_t)k0 * sm.k + e);
#pragma unroll
      for (int q = 0; q < N; ++q) m[q] = Vec16<T>::get(packed, q);
    }
    if constexpr (!VALUE_DENSE || !LOC_DENSE) {
      uint32_t k2 = kk, j2 = j;
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const int64_t k = (int64_t)k0 + k2;
        if constexpr (!VALUE_DENSE) x[q] = vb[k * sv.k + (int64_t)j2 * sv.d];
        if constexpr (!LOC_DENSE) m[q] = mb[k * sm.k + (int64_t)j2 * sm.d];
        if (++j2 == D) {
          j2 = 0;
          ++k2;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const T diff = x[q] - m[q];
      tile[pad_index(e + q)] = (-(diff * diff)) / two_var - log_sigma - half_log_2pi;
    }
    e += step;
    kk += dk;
    j += dj;
    if (j >= D) {
      j -= D;
      ++kk;
    }
  }
  for (uint32_t t = nvec * N + threadIdx.x; t < ne; t += kLpBlock) {  // at most N - 1 leftovers
    const uint32_t k2 = t / D, j2 = t - k2 * D;
    const int64_t k = (int64_t)k0 + k2;
    const T diff = vb[k * sv.k + (int64_t)j2 * sv.d] - mb[k * sm.k + (int64_t)j2 * sm.d];
    tile[pad_index(t)] = (-(diff * diff)) / two_var - log_sigma - half_log_2pi;
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < np; p += kLpBlock) {
    T sum = T(0);
    const uint32_t base = p * D;
    for (uint32_t jj = 0; jj < D; ++jj) sum += tile[pad_index(base + jj)];
    out[(int64_t)b * K + k0 + p] = sum;
  }
}

// Large D, dense value and loc rows, scalar scale: TPR lanes per particle, each streaming 16-byte
// vectors of the row; partial sums meet by shuffles inside the TPR-lane team.
template <typename T, int TPR>
__global__ __launch_bounds__(kLpBlock) void normal_lps_row_vec_kernel(
    const T *__restrict__ value, const T *__restrict__ loc, const T *__restrict__ scale,
    T *__restrict__ out, int64_t particles, uint32_t K, uint32_t D, Strides3 sv, Strides3 sm, int stream) {
  constexpr int N = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const int64_t p = (int64_t)blockIdx.x * (kLpBlock / TPR) + threadIdx.x / TPR;
  const int t = threadIdx.x % TPR;
  T sum = T(0);
  if (p < particles) {
    const int64_t b = p / K, k = p - b * K;
    const V *vp = reinterpret_cast<const V *>(value + b * sv.b + k * sv.k);
    const V *mp = reinterpret_cast<const V *>(loc + b * sm.b + k * sm.k);
    const T sigma = scale[0];
    const T two_var = T(2) * (sigma * sigma);
    const T log_sigma = Num<T>::log(sigma);
    for (uint32_t v = t; v < D / N; v += TPR) {
      const V x = vp[v], m = load16(mp + v, stream);
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const T diff = Vec16<T>::get(x, q) - Vec16<T>::get(m, q);
        sum += (-(diff * diff)) / two_var - log_sigma - NormConst<T>::half_log_2pi();
      }
    }
  }
#pragma unroll
  for (int off = TPR / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, kWave);
  if (t == 0 && p < particles) out[p] = sum;
}

static inline bool is_scalar(const Strides3 &s) { return s.b == 0 && s.k == 0 && s.d == 0; }

template <typename T>
static inline bool is_dense(const void *ptr, const Strides3 &s, int64_t D) {
  constexpr int N = Vec16<T>::N;
  return s.d == 1 && s.k == D && (s.b % N) == 0 && (((uintptr_t)ptr) & 15u) == 0;
}

template <typename T>
static int launch_lps(const void *value, const void *loc, const void *scale, void *out, int64_t B,
                      int64_t K, int64_t D, Strides3 sv, Strides3 sm, Strides3 ss, hipStream_t s) {
  constexpr int N = Vec16<T>::N;
  const bool vdense = is_dense<T>(value, sv, D), ldense = is_dense<T>(loc, sm, D);
  // the row kernel only needs each particle's D values contiguous and 16-byte aligned; the row
  // itself may be anywhere (e.g. an observation broadcast over particles: particle stride 0)
  auto rows_ok = [&](const void *ptr, const Strides3 &st) {
    return st.d == 1 && st.k % N == 0 && st.b % N == 0 && (((uintptr_t)ptr) & 15u) == 0;
  };
  if (D > 64 && D % N == 0 && is_scalar(ss) && rows_ok(value, sv) && rows_ok(loc, sm)) {
    const int64_t particles = B * K;
    const int tpr = D / N <= 16 ? 16 : (D / N <= 32 ? 32 : 64);
    const int64_t blocks = (particles + (kLpBlock / tpr) - 1) / (kLpBlock / tpr);
    if (blocks > 0x7fffffffLL) return AESMC_ERR_UNSUPPORTED;
    dim3 grid((unsigned)blocks), block(kLpBlock);
    const int stream = stream_hint((uint64_t)particles * D * sizeof(T) * 2);
    auto X = (const T *)value;
    auto M = (const T *)loc;
    auto S = (const T *)scale;
    if (tpr == 16)
      hipLaunchKernelGGL((normal_lps_row_vec_kernel<T, 16>), grid, block, 0, s, X, M, S, (T *)out, particles, (uint32_t)K, (uint32_t)D, sv, sm, stream);
    else if (tpr == 32)
      hipLaunchKernelGGL((normal_lps_row_vec_kernel<T, 32>), grid, block, 0, s, X, M, S, (T *)out, particles, (uint32_t)K, (uint32_t)D, sv, sm, stream);
    else
      hipLaunchKernelGGL((normal_lps_row_vec_kernel<T, 64>), grid, block, 0, s, X, M, S, (T *)out, particles, (uint32_t)K, (uint32_t)D, sv, sm, stream);
    return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
  }
  if (D <= 64 && is_scalar(ss) && (vdense || ldense)) {
    const uint32_t P = particles_per_tile(kTileBytes / (int)sizeof(T), D, K);
    const uint32_t tiles = (uint32_t)((K + P - 1) / P);
    // a dense operand's tile must start on a 16-byte boundary: P*D (tile pitch) and K*D (row pitch)
    if ((uint64_t)B * tiles <= 0x7fffffffull && (tiles == 1 || ((uint64_t)P * D) % N == 0)) {
      const uint32_t ne = P * (uint32_t)D;
      const size_t lds = (size_t)(ne + (ne >> 5) + 1) * sizeof(T);
      dim3 grid((unsigned)(B * tiles)), block(kLpBlock);
      auto X = (const T *)value;
      auto M = (const T *)loc;
      auto S = (const T *)scale;
      if (vdense && ldense)
        hipLaunchKernelGGL((normal_lps_tile_vec_kernel<T, true, true>), grid, block, lds, s, X, M, S, (T *)out, (uint32_t)K, (uint32_t)D, P, tiles, sv, sm);
      else if (vdense)
        hipLaunchKernelGGL((normal_lps_tile_vec_kernel<T, true, false>), grid, block, lds, s, X, M, S, (T *)out, (uint32_t)K, (uint32_t)D, P, tiles, sv, sm);
      else
        hipLaunchKernelGGL((normal_lps_tile_vec_kernel<T, false, true>), grid, block, lds, s, X, M, S, (T *)out, (uint32_t)K, (uint32_t)D, P, tiles, sv, sm);
      return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
    }
  }
  if (D <= 64) {
    const uint32_t P = particles_per_tile(kTileBytes / (int)sizeof(T), D, K);
    const uint32_t tiles = (uint32_t)((K + P - 1) / P);
    if ((uint64_t)B * tiles > 0x7fffffffull) return AESMC_ERR_UNSUPPORTED;
    const uint32_t ne = P * (uint32_t)D;
    const size_t lds = (size_t)(ne + (ne >> 5) + 1) * sizeof(T);
    hipLaunchKernelGGL((normal_lps_tile_kernel<T>), dim3((unsigned)(B * tiles)), dim3(kLpBlock), lds,
                       s, (const T *)value, (const T *)loc, (const T *)scale, (T *)out, (uint32_t)K,
                       (uint32_t)D, P, tiles, sv, sm, ss);
  } else {
    const int64_t particles = B * K;
    const int64_t blocks = (particles + (kLpBlock / kWave) - 1) / (kLpBlock / kWave);
    if (blocks > 0x7fffffffLL) return AESMC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((normal_lps_wave_kernel<T>), dim3((unsigned)blocks), dim3(kLpBlock), 0, s,
                       (const T *)value, (const T *)loc, (const T *)scale, (T *)out, particles,
                       (uint32_t)K, (uint32_t)D, sv, sm, ss);
  }
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

template <typename T>
static int launch_lps_bwd(const void *value, const void *loc, const void *scale, const void *go,
                          void *gv, void *gm, void *gs, int64_t B, int64_t K, int64_t D, Strides3 sv,
                          Strides3 sm, Strides3 ss, hipStream_t s) {
  const int64_t total = B * K * D;
  int64_t blocks = (total + kLpBlock - 1) / kLpBlock;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 workgroups per CU
  hipLaunchKernelGGL((normal_lps_bwd_kernel<T>), dim3((unsigned)blocks), dim3(kLpBlock), 0, s,
                     (const T *)value, (const T *)loc, (const T *)scale, (const T *)go, (T *)gv,
                     (T *)gm, (T *)gs, total, (uint32_t)K, (uint32_t)D, sv, sm, ss);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

// K5: the whole log-weight of one SMC step when prior / transition, emission and proposal are all
// Normal (scalar scales are constants of the launch; vector / tensor scales — learned proposals —
// are read like the locations, TENSOR_SCALES):
//   lw[b,k] = sum_j log N(x; mu_p, s_p) + sum_j log N(y; mu_g, s_g) - sum_j log N(x; mu_q, s_q)
// One pass instead of three K4 launches and K1's combine: x is read once for both of its terms.
// The three d-sums are formed separately, each in K4's order, and combined as (p + g) - q exactly
// like aesmc/inference.py:125-126, so the result is bit-identical to the unfused route.
struct View3 {
  const void *ptr;
  Strides3 st;
};

template <typename T>
__device__ __forceinline__ T load_view(const View3 &v, int64_t b, int64_t k, uint32_t j) {
  return reinterpret_cast<const T *>(v.ptr)[b * v.st.b + k * v.st.k + (int64_t)j * v.st.d];
}

// `dense` bit i set: operand i ([K,D] contiguous inside a batch row, 16-byte aligned tiles) is read
// with 16-byte loads.  Bits: 0 x, 1 mu_p, 2 mu_q (extent Dx); 3 y, 4 mu_g (extent Dy).
// STATIC_MASK >= 0 fixes the mask at compile time (the t > 0 layout of a Markov model: x, mu_p,
// mu_q, mu_g dense, y broadcast = 0b10111), so the loads are straight-line code the compiler can
// keep in flight together; -1 takes the mask from the argument.
template <typename T, int STATIC_MASK, bool TENSOR_SCALES = false, bool SCALE_TABLE = false>
__global__ __launch_bounds__(kLpBlock) void normal_logweight_kernel(
    View3 x, View3 mu_p, View3 sc_p, View3 y, View3 mu_g, View3 sc_g, View3 mu_q, View3 sc_q,
    T *__restrict__ out, uint32_t K, uint32_t Dx, uint32_t Dy, uint32_t P, uint32_t tiles_per_row,
    uint32_t dense_arg, int stream, uint32_t perd = 0) {
  const uint32_t dense = STATIC_MASK >= 0 ? (uint32_t)STATIC_MASK : dense_arg;
  // TENSOR_SCALES: a scale that varies along j only (`perd` bit 0 s_p, 1 s_q, 2 s_g — a learned
  // per-dimension vector) is tabulated once per workgroup as (2 sigma_j^2, log sigma_j); `covered`
  // marks the scales that need no per-element load (dense: vector loads, per-d: the table)
  const uint32_t covered = dense | ((perd & 7u) << 5);
  constexpr int N = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char lps_smem[];
  T *term_p = reinterpret_cast<T *>(lps_smem);                 // [P * Dx] (padded)
  const uint32_t span_x = P * Dx + ((P * Dx) >> 5) + 1;
  T *term_q = term_p + span_x;                                 // [P * Dx]
  T *term_g = term_q + span_x;                                 // [P * Dy]
  T *table = term_g + (P * Dy + ((P * Dy) >> 5) + 1);           // [6][64], TENSOR_SCALES with per-d scales
  if constexpr (TENSOR_SCALES && SCALE_TABLE) {
    {
      const uint32_t j = threadIdx.x;
      if (j < 64) {
        if ((perd & 1u) && j < Dx) {
          const T sg = reinterpret_cast<const T *>(sc_p.ptr)[(int64_t)j * sc_p.st.d];
          table[j] = T(2) * (sg * sg);
          table[64 + j] = Num<T>::log(sg);
        }
        if ((perd & 2u) && j < Dx) {
          const T sg = reinterpret_cast<const T *>(sc_q.ptr)[(int64_t)j * sc_q.st.d];
          table[128 + j] = T(2) * (sg * sg);
          table[192 + j] = Num<T>::log(sg);
        }
        if ((perd & 4u) && j < Dy) {
          const T sg = reinterpret_cast<const T *>(sc_g.ptr)[(int64_t)j * sc_g.st.d];
          table[256 + j] = T(2) * (sg * sg);
          table[320 + j] = Num<T>::log(sg);
        }
      }
      __syncthreads();
    }
  }
  const uint32_t b = blockIdx.x / tiles_per_row;
  const uint32_t k0 = (blockIdx.x - b * tiles_per_row) * P;
  const uint32_t np = min(P, K - k0);
  const T half_log_2pi = NormConst<T>::half_log_2pi();

  // The t > 0 layout of a Markov model with equal extents (x, mu_p, mu_q, mu_g dense, y broadcast):
  // ONE sweep issues all four 16-byte loads of a vector slot before any arithmetic, instead of the
  // x-terms' sweep followed by the emission term's (whose loads could only start after the first
  // sweep's had landed).  Same element arithmetic, same LDS layout, same sums.
  if (STATIC_MASK == 23 && Dx == Dy && (np * Dx) % N == 0) {
    const T s_p = reinterpret_cast<const T *>(sc_p.ptr)[0], s_q = reinterpret_cast<const T *>(sc_q.ptr)[0],
            s_g = reinterpret_cast<const T *>(sc_g.ptr)[0];
    const T two_var_p = T(2) * (s_p * s_p), log_p = Num<T>::log(s_p);
    const T two_var_q = T(2) * (s_q * s_q), log_q = Num<T>::log(s_q);
    const T two_var_g = T(2) * (s_g * s_g), log_g = Num<T>::log(s_g);
    const uint32_t ne = np * Dx, nvec = ne / N;
    const T *xt = reinterpret_cast<const T *>(x.ptr) + (int64_t)b * x.st.b + (int64_t)k0 * x.st.k;
    const T *pt = reinterpret_cast<const T *>(mu_p.ptr) + (int64_t)b * mu_p.st.b + (int64_t)k0 * mu_p.st.k;
    const T *qt = reinterpret_cast<const T *>(mu_q.ptr) + (int64_t)b * mu_q.st.b + (int64_t)k0 * mu_q.st.k;
    const T *gt = reinterpret_cast<const T *>(mu_g.ptr) + (int64_t)b * mu_g.st.b + (int64_t)k0 * mu_g.st.k;
    uint32_t e = threadIdx.x * N;
    uint32_t kk = e / Dx, j = e - kk * Dx;
    const uint32_t step = kLpBlock * N, dk = step / Dx, dj = step - dk * Dx;
    for (uint32_t v = threadIdx.x; v < nvec; v += kLpBlock) {
      const V tx = *reinterpret_cast<const V *>(xt + e);
      const V tp = load16(reinterpret_cast<const V *>(pt + e), stream);
      const V tq = load16(reinterpret_cast<const V *>(qt + e), stream);
      const V tg = load16(reinterpret_cast<const V *>(gt + e), stream);
      T yv[N];
      uint32_t k2 = kk, j2 = j;
#pragma unroll
      for (int r = 0; r < N; ++r) {
        yv[r] = load_view<T>(y, b, (int64_t)k0 + k2, j2);
        if (++j2 == Dx) { j2 = 0; ++k2; }
      }
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const T xr = Vec16<T>::get(tx, r);
        const T dp = xr - Vec16<T>::get(tp, r), dq = xr - Vec16<T>::get(tq, r);
        const T dg = yv[r] - Vec16<T>::get(tg, r);
        term_p[pad_index(e + r)] = (-(dp * dp)) / two_var_p - log_p - half_log_2pi;
        term_q[pad_index(e + r)] = (-(dq * dq)) / two_var_q - log_q - half_log_2pi;
        term_g[pad_index(e + r)] = (-(dg * dg)) / two_var_g - log_g - half_log_2pi;
      }
      e += step; kk += dk; j += dj;
      if (j >= Dx) { j -= Dx; ++kk; }
    }
  } else {
  {  // ---- the two terms in x ---------------------------------------------------------------------
    // scalar scales: constants of the launch; tensor scales (TENSOR_SCALES: dense bits 32 s_p,
    // 64 s_q, 128 s_g) are read like the locations and enter element by element, as K4 forms them
    const T s_p = reinterpret_cast<const T *>(sc_p.ptr)[0], s_q = reinterpret_cast<const T *>(sc_q.ptr)[0];
    const T two_var_p = T(2) * (s_p * s_p), log_p = Num<T>::log(s_p);
    const T two_var_q = T(2) * (s_q * s_q), log_q = Num<T>::log(s_q);
    const T *spt = reinterpret_cast<const T *>(sc_p.ptr) + (int64_t)b * sc_p.st.b + (int64_t)k0 * sc_p.st.k;
    const T *sqt = reinterpret_cast<const T *>(sc_q.ptr) + (int64_t)b * sc_q.st.b + (int64_t)k0 * sc_q.st.k;
    const uint32_t ne = np * Dx, nvec = ne / N;
    const T *xt = reinterpret_cast<const T *>(x.ptr) + (int64_t)b * x.st.b + (int64_t)k0 * x.st.k;
    const T *pt = reinterpret_cast<const T *>(mu_p.ptr) + (int64_t)b * mu_p.st.b + (int64_t)k0 * mu_p.st.k;
    const T *qt = reinterpret_cast<const T *>(mu_q.ptr) + (int64_t)b * mu_q.st.b + (int64_t)k0 * mu_q.st.k;
    uint32_t e = threadIdx.x * N;
    uint32_t kk = e / Dx, j = e - kk * Dx;
    const uint32_t step = kLpBlock * N, dk = step / Dx, dj = step - dk * Dx;
    for (uint32_t v = threadIdx.x; v < nvec; v += kLpBlock) {
      T xv[N], pv[N], qv[N];
      if (dense & 1u) { const V t = *reinterpret_cast<const V *>(xt + e);
#pragma unroll
        for (int r = 0; r < N; ++r) xv[r] = Vec16<T>::get(t, r); }
      if (dense & 2u) { const V t = load16(reinterpret_cast<const V *>(pt + e), stream);
#pragma unroll
        for (int r = 0; r < N; ++r) pv[r] = Vec16<T>::get(t, r); }
      if (dense & 4u) { const V t = load16(reinterpret_cast<const V *>(qt + e), stream);
#pragma unroll
        for (int r = 0; r < N; ++r) qv[r] = Vec16<T>::get(t, r); }
      T spv[N], sqv[N];
      if constexpr (TENSOR_SCALES) {
        if (dense & 32u) { const V t = load16(reinterpret_cast<const V *>(spt + e), stream);
#pragma unroll
          for (int r = 0; r < N; ++r) spv[r] = Vec16<T>::get(t, r); }
        if (dense & 64u) { const V t = load16(reinterpret_cast<const V *>(sqt + e), stream);
#pragma unroll
          for (int r = 0; r < N; ++r) sqv[r] = Vec16<T>::get(t, r); }
      }
      if ((dense & 7u) != 7u || (TENSOR_SCALES && (covered & 96u) != 96u)) {
        uint32_t k2 = kk, j2 = j;
#pragma unroll
        for (int r = 0; r < N; ++r) {
          const int64_t k = (int64_t)k0 + k2;
          if (!(dense & 1u)) xv[r] = load_view<T>(x, b, k, j2);
          if (!(dense & 2u)) pv[r] = load_view<T>(mu_p, b, k, j2);
          if (!(dense & 4u)) qv[r] = load_view<T>(mu_q, b, k, j2);
          if constexpr (TENSOR_SCALES) {
            if (!(covered & 32u)) spv[r] = load_view<T>(sc_p, b, k, j2);
            if (!(covered & 64u)) sqv[r] = load_view<T>(sc_q, b, k, j2);
          }
          if (++j2 == Dx) { j2 = 0; ++k2; }
        }
      }
#pragma unroll
      for (int r = 0; r < N; ++r) {
        if constexpr (TENSOR_SCALES) {
          if constexpr (!SCALE_TABLE) {             // no per-dimension scale in this launch
            term_p[pad_index(e + r)] = normal_logpdf(xv[r], pv[r], spv[r]);
            term_q[pad_index(e + r)] = normal_logpdf(xv[r], qv[r], sqv[r]);
            continue;
          }
          uint32_t jr = j + r;                      // this element's dimension index (r < N <= Dx wraps once)
          if (jr >= Dx) jr -= Dx;
          if (Dx < (uint32_t)N) jr = (j + r) % Dx;
          const T dp = xv[r] - pv[r], dq = xv[r] - qv[r];
          const T tvp = (perd & 1u) ? table[jr] : T(2) * (spv[r] * spv[r]);
          const T lgp = (perd & 1u) ? table[64 + jr] : Num<T>::log(spv[r]);
          const T tvq = (perd & 2u) ? table[128 + jr] : T(2) * (sqv[r] * sqv[r]);
          const T lgq = (perd & 2u) ? table[192 + jr] : Num<T>::log(sqv[r]);
          term_p[pad_index(e + r)] = (-(dp * dp)) / tvp - lgp - half_log_2pi;
          term_q[pad_index(e + r)] = (-(dq * dq)) / tvq - lgq - half_log_2pi;
        } else {
          const T dp = xv[r] - pv[r], dq = xv[r] - qv[r];
          term_p[pad_index(e + r)] = (-(dp * dp)) / two_var_p - log_p - half_log_2pi;
          term_q[pad_index(e + r)] = (-(dq * dq)) / two_var_q - log_q - half_log_2pi;
        }
      }
      e += step; kk += dk; j += dj;
      if (j >= Dx) { j -= Dx; ++kk; }
    }
    for (uint32_t t = nvec * N + threadIdx.x; t < ne; t += kLpBlock) {
      const uint32_t k2 = t / Dx, j2 = t - k2 * Dx;
      const int64_t k = (int64_t)k0 + k2;
      const T xv = load_view<T>(x, b, k, j2);
      if constexpr (TENSOR_SCALES) {
        term_p[pad_index(t)] = normal_logpdf(xv, load_view<T>(mu_p, b, k, j2), load_view<T>(sc_p, b, k, j2));
        term_q[pad_index(t)] = normal_logpdf(xv, load_view<T>(mu_q, b, k, j2), load_view<T>(sc_q, b, k, j2));
      } else {
        const T dp = xv - load_view<T>(mu_p, b, k, j2), dq = xv - load_view<T>(mu_q, b, k, j2);
        term_p[pad_index(t)] = (-(dp * dp)) / two_var_p - log_p - half_log_2pi;
        term_q[pad_index(t)] = (-(dq * dq)) / two_var_q - log_q - half_log_2pi;
      }
    }
  }
  {  // ---- the emission term in y ----------------------------------------------------------------
    const T s_g = reinterpret_cast<const T *>(sc_g.ptr)[0];
    const T two_var_g = T(2) * (s_g * s_g), log_g = Num<T>::log(s_g);
    const uint32_t ne = np * Dy, nvec = ne / N;
    const T *yt = reinterpret_cast<const T *>(y.ptr) + (int64_t)b * y.st.b + (int64_t)k0 * y.st.k;
    const T *sgt = reinterpret_cast<const T *>(sc_g.ptr) + (int64_t)b * sc_g.st.b + (int64_t)k0 * sc_g.st.k;
    const T *gt = reinterpret_cast<const T *>(mu_g.ptr) + (int64_t)b * mu_g.st.b + (int64_t)k0 * mu_g.st.k;
    uint32_t e = threadIdx.x * N;
    uint32_t kk = e / Dy, j = e - kk * Dy;
    const uint32_t step = kLpBlock * N, dk = step / Dy, dj = step - dk * Dy;
    for (uint32_t v = threadIdx.x; v < nvec; v += kLpBlock) {
      T yv[N], gv[N];
      if (dense & 8u) { const V t = *reinterpret_cast<const V *>(yt + e);
#pragma unroll
        for (int r = 0; r < N; ++r) yv[r] = Vec16<T>::get(t, r); }
      if (dense & 16u) { const V t = load16(reinterpret_cast<const V *>(gt + e), stream);
#pragma unroll
        for (int r = 0; r < N; ++r) gv[r] = Vec16<T>::get(t, r); }
      T sgv[N];
      if constexpr (TENSOR_SCALES) {
        if (dense & 128u) { const V t = load16(reinterpret_cast<const V *>(sgt + e), stream);
#pragma unroll
          for (int r = 0; r < N; ++r) sgv[r] = Vec16<T>::get(t, r); }
      }
      if ((dense & 24u) != 24u || (TENSOR_SCALES && !(covered & 128u))) {
        uint32_t k2 = kk, j2 = j;
#pragma unroll
        for (int r = 0; r < N; ++r) {
          const int64_t k = (int64_t)k0 + k2;
          if (!(dense & 8u)) yv[r] = load_view<T>(y, b, k, j2);
          if (!(dense & 16u)) gv[r] = load_view<T>(mu_g, b, k, j2);
          if constexpr (TENSOR_SCALES) {
            if (!(covered & 128u)) sgv[r] = load_view<T>(sc_g, b, k, j2);
          }
          if (++j2 == Dy) { j2 = 0; ++k2; }
        }
      }
#pragma unroll
      for (int r = 0; r < N; ++r) {
        if constexpr (TENSOR_SCALES) {
          if constexpr (!SCALE_TABLE) {
            term_g[pad_index(e + r)] = normal_logpdf(yv[r], gv[r], sgv[r]);
            continue;
          }
          uint32_t jr = j + r;
          if (jr >= Dy) jr -= Dy;
          if (Dy < (uint32_t)N) jr = (j + r) % Dy;
          const T dg = yv[r] - gv[r];
          const T tvg = (perd & 4u) ? table[256 + jr] : T(2) * (sgv[r] * sgv[r]);
          const T lgg = (perd & 4u) ? table[320 + jr] : Num<T>::log(sgv[r]);
          term_g[pad_index(e + r)] = (-(dg * dg)) / tvg - lgg - half_log_2pi;
        } else {
          const T dg = yv[r] - gv[r];
          term_g[pad_index(e + r)] = (-(dg * dg)) / two_var_g - log_g - half_log_2pi;
        }
      }
      e += step; kk += dk; j += dj;
      if (j >= Dy) { j -= Dy; ++kk; }
    }
    for (uint32_t t = nvec * N + threadIdx.x; t < ne; t += kLpBlock) {
      const uint32_t k2 = t / Dy, j2 = t - k2 * Dy;
      const int64_t k = (int64_t)k0 + k2;
      if constexpr (TENSOR_SCALES) {
        term_g[pad_index(t)] = normal_logpdf(load_view<T>(y, b, k, j2), load_view<T>(mu_g, b, k, j2),
                                             load_view<T>(sc_g, b, k, j2));
      } else {
        const T dg = load_view<T>(y, b, k, j2) - load_view<T>(mu_g, b, k, j2);
        term_g[pad_index(t)] = (-(dg * dg)) / two_var_g - log_g - half_log_2pi;
      }
    }
  }
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < np; p += kLpBlock) {
    T sum_p = T(0), sum_q = T(0), sum_g = T(0);
    for (uint32_t jj = 0; jj < Dx; ++jj) {
      sum_p += term_p[pad_index(p * Dx + jj)];
      sum_q += term_q[pad_index(p * Dx + jj)];
    }
    for (uint32_t jj = 0; jj < Dy; ++jj) sum_g += term_g[pad_index(p * Dy + jj)];
    out[(int64_t)b * K + k0 + p] = (sum_p + sum_g) - sum_q;
  }
}

// K5 with one value per particle in both x and y (Dx == Dy == 1, e.g. the scalar IWAE model of
// BASELINE.json configs[2]): no d-sum, so no LDS tile — a lane takes four consecutive particles of
// a batch row; each operand is read as one 16-byte load (dense along k), one scalar load
// (broadcast along k) or four strided ones.  Same element arithmetic and the same (p + g) - q
// combine as the tiled kernel; the sums of one term are `0 + term`, as there.
template <typename T>
__device__ __forceinline__ void load_four(const View3 &v, int64_t b, uint32_t k0, uint32_t live, T (&dst)[4]) {
  const T *row = reinterpret_cast<const T *>(v.ptr) + b * v.st.b;
  if (v.st.k == 0) {
    const T value = row[0];
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = value;
  } else if (v.st.k == 1 && live == 4 && ((reinterpret_cast<uintptr_t>(row + k0) & (4 * sizeof(T) - 1)) == 0)) {
    struct alignas(4 * sizeof(T)) Four { T v[4]; };
    const Four packed = *reinterpret_cast<const Four *>(row + k0);
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = packed.v[r];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = row[(int64_t)(k0 + (r < (int)live ? r : 0)) * v.st.k];
  }
}

template <typename T>
__global__ __launch_bounds__(kLpBlock) void normal_logweight_d1_kernel(
    View3 x, View3 mu_p, View3 sc_p, View3 y, View3 mu_g, View3 sc_g, View3 mu_q, View3 sc_q,
    T *__restrict__ out, uint32_t K, uint32_t blocks_per_row) {
  const uint32_t b = blockIdx.x / blocks_per_row;
  const uint32_t k0 = ((blockIdx.x - b * blocks_per_row) * kLpBlock + threadIdx.x) * 4;
  if (k0 >= K) return;
  const uint32_t live = min(4u, K - k0);
  const T s_p = reinterpret_cast<const T *>(sc_p.ptr)[0], s_g = reinterpret_cast<const T *>(sc_g.ptr)[0],
          s_q = reinterpret_cast<const T *>(sc_q.ptr)[0];
  const T two_var_p = T(2) * (s_p * s_p), log_p = Num<T>::log(s_p);
  const T two_var_g = T(2) * (s_g * s_g), log_g = Num<T>::log(s_g);
  const T two_var_q = T(2) * (s_q * s_q), log_q = Num<T>::log(s_q);
  const T half_log_2pi = NormConst<T>::half_log_2pi();
  T xv[4], pv[4], qv[4], yv[4], gv[4], lw[4];
  load_four<T>(x, b, k0, live, xv);
  load_four<T>(mu_p, b, k0, live, pv);
  load_four<T>(mu_q, b, k0, live, qv);
  load_four<T>(y, b, k0, live, yv);
  load_four<T>(mu_g, b, k0, live, gv);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const T dp = xv[r] - pv[r], dq = xv[r] - qv[r], dg = yv[r] - gv[r];
    const T sum_p = T(0) + ((-(dp * dp)) / two_var_p - log_p - half_log_2pi);
    const T sum_q = T(0) + ((-(dq * dq)) / two_var_q - log_q - half_log_2pi);
    const T sum_g = T(0) + ((-(dg * dg)) / two_var_g - log_g - half_log_2pi);
    lw[r] = (sum_p + sum_g) - sum_q;
  }
  T *orow = out + (int64_t)b * K + k0;
  if (live == 4 && ((reinterpret_cast<uintptr_t>(orow) & (4 * sizeof(T) - 1)) == 0)) {
    struct alignas(4 * sizeof(T)) Four { T v[4]; };
    Four packed;
#pragma unroll
    for (int r = 0; r < 4; ++r) packed.v[r] = lw[r];
    *reinterpret_cast<Four *>(orow) = packed;
  } else {
    for (uint32_t r = 0; r < live; ++r) orow[r] = lw[r];
  }
}

// K5 for wide rows (more than 64 values per particle, e.g. d = 128): K4's row mapping — TPR lanes
// per particle stream 16-byte vectors of the row, partial sums meet by shuffles — with the three
// terms accumulated side by side, x read once for two of them.  Same per-lane order and the same
// shuffle tree as normal_lps_row_vec_kernel, so each sum is bit-identical to K4's.
template <typename T, int TPR>
__global__ __launch_bounds__(kLpBlock) void normal_logweight_row_kernel(
    View3 x, View3 mu_p, View3 sc_p, View3 y, View3 mu_g, View3 sc_g, View3 mu_q, View3 sc_q,
    T *__restrict__ out, int64_t particles, uint32_t K, uint32_t Dx, uint32_t Dy, int stream) {
  constexpr int N = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const int64_t p = (int64_t)blockIdx.x * (kLpBlock / TPR) + threadIdx.x / TPR;
  const int t = threadIdx.x % TPR;
  T sum_p = T(0), sum_q = T(0), sum_g = T(0);
  if (p < particles) {
    const int64_t b = p / K, k = p - b * K;
    auto row = [&](const View3 &v) {
      return reinterpret_cast<const V *>(reinterpret_cast<const T *>(v.ptr) + b * v.st.b + k * v.st.k);
    };
    const V *xp = row(x), *pp = row(mu_p), *qp = row(mu_q), *yp = row(y), *gp = row(mu_g);
    const T s_p = reinterpret_cast<const T *>(sc_p.ptr)[0], s_g = reinterpret_cast<const T *>(sc_g.ptr)[0],
            s_q = reinterpret_cast<const T *>(sc_q.ptr)[0];
    const T two_var_p = T(2) * (s_p * s_p), log_p = Num<T>::log(s_p);
    const T two_var_g = T(2) * (s_g * s_g), log_g = Num<T>::log(s_g);
    const T two_var_q = T(2) * (s_q * s_q), log_q = Num<T>::log(s_q);
    const T half_log_2pi = NormConst<T>::half_log_2pi();
    for (uint32_t v = t; v < Dx / N; v += TPR) {
      const V xv = xp[v], pv = load16(pp + v, stream), qv = load16(qp + v, stream);
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const T dp = Vec16<T>::get(xv, r) - Vec16<T>::get(pv, r);
        const T dq = Vec16<T>::get(xv, r) - Vec16<T>::get(qv, r);
        sum_p += (-(dp * dp)) / two_var_p - log_p - half_log_2pi;
        sum_q += (-(dq * dq)) / two_var_q - log_q - half_log_2pi;
      }
    }
    for (uint32_t v = t; v < Dy / N; v += TPR) {
      const V yv = yp[v], gv = load16(gp + v, stream);
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const T dg = Vec16<T>::get(yv, r) - Vec16<T>::get(gv, r);
        sum_g += (-(dg * dg)) / two_var_g - log_g - half_log_2pi;
      }
    }
  }
#pragma unroll
  for (int off = TPR / 2; off > 0; off >>= 1) {
    sum_p += __shfl_xor(sum_p, off, kWave);
    sum_q += __shfl_xor(sum_q, off, kWave);
    sum_g += __shfl_xor(sum_g, off, kWave);
  }
  if (t == 0 && p < particles) out[p] = (sum_p + sum_g) - sum_q;
}

static inline int row_team(int64_t D, int N) { return D / N <= 16 ? 16 : (D / N <= 32 ? 32 : 64); }

template <typename T>
static int launch_logweight(const View3 *v, void *out, int64_t B, int64_t K, int64_t Dx, int64_t Dy,
                            hipStream_t s) {
  constexpr int N = Vec16<T>::N;
  if (Dx < 1 || Dy < 1) return AESMC_ERR_UNSUPPORTED;
  const bool tensor_scales = !is_scalar(v[2].st) || !is_scalar(v[5].st) || !is_scalar(v[7].st);
  if (tensor_scales && (Dx > 64 || Dy > 64)) return AESMC_ERR_UNSUPPORTED;
  if (Dx > 64 || Dy > 64) {
    // wide rows: both extents wide, whole 16-byte vectors, the same lane team as K4 picks for
    // each (so that every sum keeps K4's order), rows contiguous and 16-byte aligned
    auto rows_ok = [&](const View3 &view) {
      return view.st.d == 1 && view.st.k % N == 0 && view.st.b % N == 0 && (((uintptr_t)view.ptr) & 15u) == 0;
    };
    if (Dx <= 64 || Dy <= 64 || Dx % N != 0 || Dy % N != 0 || row_team(Dx, N) != row_team(Dy, N))
      return AESMC_ERR_UNSUPPORTED;
    if (!rows_ok(v[0]) || !rows_ok(v[1]) || !rows_ok(v[6]) || !rows_ok(v[3]) || !rows_ok(v[4]))
      return AESMC_ERR_UNSUPPORTED;
    const int64_t particles = B * K;
    const int tpr = row_team(Dx, N);
    const int64_t blocks = (particles + (kLpBlock / tpr) - 1) / (kLpBlock / tpr);
    if (blocks > 0x7fffffffLL) return AESMC_ERR_UNSUPPORTED;
    dim3 grid((unsigned)blocks), block(kLpBlock);
    const int stream = stream_hint((uint64_t)particles * (3 * Dx + Dy) * sizeof(T));
#define AESMC_ROW_CASE(team)                                                                            \
    hipLaunchKernelGGL((normal_logweight_row_kernel<T, team>), grid, block, 0, s, v[0], v[1], v[2], v[3],  \
                       v[4], v[5], v[6], v[7], (T *)out, particles, (uint32_t)K, (uint32_t)Dx, (uint32_t)Dy, stream)
    if (tpr == 16) AESMC_ROW_CASE(16);
    else if (tpr == 32) AESMC_ROW_CASE(32);
    else AESMC_ROW_CASE(64);
#undef AESMC_ROW_CASE
    return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
  }
  if (Dx == 1 && Dy == 1 && !tensor_scales) {
    const uint32_t bpr = (uint32_t)((K + 4 * kLpBlock - 1) / (4 * kLpBlock));
    if ((uint64_t)B * bpr > 0x7fffffffull) return AESMC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((normal_logweight_d1_kernel<T>), dim3((unsigned)(B * bpr)), dim3(kLpBlock), 0, s, v[0],
                       v[1], v[2], v[3], v[4], v[5], v[6], v[7], (T *)out, (uint32_t)K, bpr);
    return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
  }
  const uint32_t P = particles_per_tile(kTileBytes / (int)sizeof(T), 2 * Dx + Dy, K);
  if (P == 0) return AESMC_ERR_UNSUPPORTED;
  const uint32_t tiles = (uint32_t)((K + P - 1) / P);
  if ((uint64_t)B * tiles > 0x7fffffffull) return AESMC_ERR_UNSUPPORTED;
  auto dense = [&](const View3 &view, int64_t D) {
    return is_dense<T>(view.ptr, view.st, D) && (tiles == 1 || ((uint64_t)P * D) % N == 0);
  };
  uint32_t mask = 0;
  if (dense(v[0], Dx)) mask |= 1u;
  if (dense(v[1], Dx)) mask |= 2u;
  if (dense(v[6], Dx)) mask |= 4u;
  if (dense(v[3], Dy)) mask |= 8u;
  if (dense(v[4], Dy)) mask |= 16u;
  const uint32_t ex = P * (uint32_t)Dx, ey = P * (uint32_t)Dy;
  const size_t lds = (size_t)(2 * (ex + (ex >> 5) + 1) + ey + (ey >> 5) + 1) * sizeof(T);
  const int stream = stream_hint((uint64_t)B * K * (3 * Dx + Dy) * sizeof(T));
  if (tensor_scales) {   // scales read like locations: dense ones by 16-byte loads (bits 32 s_p, 64 s_q, 128 s_g)
    if (dense(v[2], Dx)) mask |= 32u;
    if (dense(v[7], Dx)) mask |= 64u;
    if (dense(v[5], Dy)) mask |= 128u;
    auto per_dimension = [](const View3 &view) { return view.st.b == 0 && view.st.k == 0; };  // scalars included
    const uint32_t perd = (per_dimension(v[2]) ? 1u : 0u) | (per_dimension(v[7]) ? 2u : 0u) |
                          (per_dimension(v[5]) ? 4u : 0u);
    if (perd != 0)
      hipLaunchKernelGGL((normal_logweight_kernel<T, -1, true, true>), dim3((unsigned)(B * tiles)), dim3(kLpBlock),
                         lds + 6 * 64 * sizeof(T), s, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], (T *)out,
                         (uint32_t)K, (uint32_t)Dx, (uint32_t)Dy, P, tiles, mask, stream, perd);
    else
      hipLaunchKernelGGL((normal_logweight_kernel<T, -1, true, false>), dim3((unsigned)(B * tiles)), dim3(kLpBlock),
                         lds, s, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], (T *)out, (uint32_t)K,
                         (uint32_t)Dx, (uint32_t)Dy, P, tiles, mask, stream, 0u);
    return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
  }
  if (mask == 23u)
    hipLaunchKernelGGL((normal_logweight_kernel<T, 23>), dim3((unsigned)(B * tiles)), dim3(kLpBlock), lds, s,
                       v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], (T *)out, (uint32_t)K, (uint32_t)Dx,
                       (uint32_t)Dy, P, tiles, mask, stream);
  else
    hipLaunchKernelGGL((normal_logweight_kernel<T, -1>), dim3((unsigned)(B * tiles)), dim3(kLpBlock), lds, s,
                       v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], (T *)out, (uint32_t)K, (uint32_t)Dx,
                       (uint32_t)Dy, P, tiles, mask, stream);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

// K5 backward: one pass writes, densely, the gradient of every operand that wants one — values,
// locations and scales.  Element arithmetic and the final gx = gx_p + gx_q are K4's backward plus
// the eager add, operation for operation, so the numbers equal the three-launch route bit for bit.
template <typename T>
__global__ __launch_bounds__(kLpBlock) void normal_logweight_bwd_kernel(
    View3 x, View3 mu_p, View3 sc_p, View3 y, View3 mu_g, View3 sc_g, View3 mu_q, View3 sc_q,
    const T *__restrict__ grad_lw, T *__restrict__ gx, T *__restrict__ gmu_p, T *__restrict__ gy,
    T *__restrict__ gmu_g, T *__restrict__ gmu_q, T *__restrict__ gs_p, T *__restrict__ gs_g,
    T *__restrict__ gs_q, int64_t total_x, int64_t total_y, uint32_t K, uint32_t Dx, uint32_t Dy,
    const T *__restrict__ lw, const T *__restrict__ lse, const T *__restrict__ grad_lse) {
  const int64_t stride = (int64_t)gridDim.x * kLpBlock;
  const int64_t first = (int64_t)blockIdx.x * kLpBlock + threadIdx.x;
  // The gradient arriving at the log-weights.  With `grad_lse` it is K1's backward evaluated in
  // place — grad_lse[b] * exp(lw - lse[b]) (+ grad_lw when that is given too), the same operations
  // in the same order as logweight_lse_bwd_kernel — so the [B,K] gradient never round-trips HBM.
  auto incoming = [&](int64_t p, int64_t b) -> T {
    T g = grad_lse != nullptr ? grad_lse[b] * Num<T>::exp(lw[p] - lse[b]) : T(0);
    if (grad_lw != nullptr) g = grad_lse != nullptr ? g + grad_lw[p] : grad_lw[p];
    return g;
  };
  if (gx != nullptr || gmu_p != nullptr || gmu_q != nullptr || gs_p != nullptr || gs_q != nullptr) {
    for (int64_t e = first; e < total_x; e += stride) {
      const int64_t p = e / Dx;
      const uint32_t j = (uint32_t)(e - p * Dx);
      const int64_t b = p / K, k = p - b * K;
      const T v = load_view<T>(x, b, k, j);
      const T g = incoming(p, b);
      const T s_p = load_view<T>(sc_p, b, k, j), s_q = load_view<T>(sc_q, b, k, j);
      const T var_p = s_p * s_p, var_q = s_q * s_q;
      const T dp = v - load_view<T>(mu_p, b, k, j), dq = v - load_view<T>(mu_q, b, k, j);
      const T gq = -g;                                  // log q enters the weight with a minus sign
      const T gz_p = g * (dp / var_p);
      const T gz_q = gq * (dq / var_q);
      if (gmu_p) gmu_p[e] = gz_p;
      if (gmu_q) gmu_q[e] = gz_q;
      if (gx) gx[e] = (-gz_p) + (-gz_q);
      if (gs_p) gs_p[e] = g * ((dp * dp) / (var_p * s_p) - T(1) / s_p);
      if (gs_q) gs_q[e] = gq * ((dq * dq) / (var_q * s_q) - T(1) / s_q);
    }
  }
  if (gy != nullptr || gmu_g != nullptr || gs_g != nullptr) {
    for (int64_t e = first; e < total_y; e += stride) {
      const int64_t p = e / Dy;
      const uint32_t j = (uint32_t)(e - p * Dy);
      const int64_t b = p / K, k = p - b * K;
      const T g = incoming(p, b);
      const T s_g = load_view<T>(sc_g, b, k, j);
      const T var_g = s_g * s_g;
      const T dg = load_view<T>(y, b, k, j) - load_view<T>(mu_g, b, k, j);
      const T gz_g = g * (dg / var_g);
      if (gmu_g) gmu_g[e] = gz_g;
      if (gy) gy[e] = -gz_g;
      if (gs_g) gs_g[e] = g * ((dg * dg) / (var_g * s_g) - T(1) / s_g);
    }
  }
}

// K5 backward, the layout of a Markov model (what every timestep of the bench workloads runs): x, mu_p,
// mu_q dense [B,K,Dx], mu_g dense [B,K,Dy], y dense or one row per batch element, every scale one
// value for the whole tensor.  The generic kernel above addresses each element through strided views
// with two 64-bit divisions and moves 4 bytes per access: 620 us for 1.36 GB at B=1024 K=4096 d=10
// (2.2 TB/s).  Here a lane walks 16-byte vectors of the flat arrays — (particle, column) of a vector's
// first element advance by a constant step per trip, no division in the loop — and every gradient is a
// 16-byte store.  Element arithmetic is the generic kernel's, operation for operation: same bits.
template <typename T>
__global__ __launch_bounds__(kLpBlock) void normal_logweight_bwd_dense_kernel(
    const T *__restrict__ x, const T *__restrict__ mu_p, const T *__restrict__ mu_q, const T *__restrict__ y,
    const T *__restrict__ mu_g, const T *__restrict__ sc_p, const T *__restrict__ sc_g, const T *__restrict__ sc_q,
    int64_t y_stride_b /* elements; y_dense: K * Dy */, int y_dense, const T *__restrict__ grad_lw,
    const T *__restrict__ lw, const T *__restrict__ lse, const T *__restrict__ grad_lse, T *__restrict__ gx,
    T *__restrict__ gmu_p, T *__restrict__ gy, T *__restrict__ gmu_g, T *__restrict__ gmu_q, uint32_t vec_x,
    uint32_t vec_y, uint32_t K, uint32_t Dx, uint32_t Dy) {
  constexpr int N = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const uint32_t stride = gridDim.x * kLpBlock;
  const uint32_t first = blockIdx.x * kLpBlock + threadIdx.x;
  auto incoming = [&](uint32_t p, uint32_t b) -> T {   // as in normal_logweight_bwd_kernel
    T g = grad_lse != nullptr ? grad_lse[b] * Num<T>::exp(lw[p] - lse[b]) : T(0);
    if (grad_lw != nullptr) g = grad_lse != nullptr ? g + grad_lw[p] : grad_lw[p];
    return g;
  };
  auto put = [](V &v, int r, T value) {
    if (r == 0) v.x = value;
    else if (r == 1) v.y = value;
    if constexpr (N == 4) {
      if (r == 2) v.z = value;
      else if (r == 3) v.w = value;
    }
  };
  if (gx != nullptr || gmu_p != nullptr || gmu_q != nullptr) {
    const T s_p = sc_p[0], s_q = sc_q[0];
    const T var_p = s_p * s_p, var_q = s_q * s_q;
    // element e = vector * N: particle p = e / Dx, column j; both advance by a constant per trip
    const uint32_t step = stride * N;
    const uint32_t dp_step = step / Dx, dj_step = step - dp_step * Dx;
    uint32_t p = (first * N) / Dx, j = first * N - p * Dx;
    for (uint32_t v = first; v < vec_x; v += stride) {
      const V xv = reinterpret_cast<const V *>(x)[v];
      const V pv = reinterpret_cast<const V *>(mu_p)[v];
      const V qv = reinterpret_cast<const V *>(mu_q)[v];
      V out_p, out_q, out_x;
      uint32_t pp = p, jj = j;
      T g = incoming(pp, pp / K);
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const T value = Vec16<T>::get(xv, r);
        const T dp = value - Vec16<T>::get(pv, r), dq = value - Vec16<T>::get(qv, r);
        const T gq = -g;                                  // log q enters the weight with a minus sign
        const T gz_p = g * (dp / var_p);
        const T gz_q = gq * (dq / var_q);
        put(out_p, r, gz_p);
        put(out_q, r, gz_q);
        put(out_x, r, (-gz_p) + (-gz_q));
        if (++jj == Dx && r + 1 < N) {
          jj = 0;
          ++pp;
          g = incoming(pp, pp / K);
        }
      }
      if (gmu_p) reinterpret_cast<V *>(gmu_p)[v] = out_p;
      if (gmu_q) reinterpret_cast<V *>(gmu_q)[v] = out_q;
      if (gx) reinterpret_cast<V *>(gx)[v] = out_x;
      p += dp_step;
      j += dj_step;
      if (j >= Dx) {
        j -= Dx;
        ++p;
      }
    }
  }
  if (gy != nullptr || gmu_g != nullptr) {
    const T s_g = sc_g[0];
    const T var_g = s_g * s_g;
    const uint32_t step = stride * N;
    const uint32_t dp_step = step / Dy, dj_step = step - dp_step * Dy;
    uint32_t p = (first * N) / Dy, j = first * N - p * Dy;
    for (uint32_t v = first; v < vec_y; v += stride) {
      const V gv = reinterpret_cast<const V *>(mu_g)[v];
      V out_g, out_y;
      uint32_t pp = p, jj = j;
      uint32_t b = pp / K;
      T g = incoming(pp, b);
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const T yv = y_dense ? y[(uint64_t)pp * Dy + jj] : y[(int64_t)b * y_stride_b + jj];
        const T dg = yv - Vec16<T>::get(gv, r);
        const T gz_g = g * (dg / var_g);
        put(out_g, r, gz_g);
        put(out_y, r, -gz_g);
        if (++jj == Dy && r + 1 < N) {
          jj = 0;
          ++pp;
          b = pp / K;
          g = incoming(pp, b);
        }
      }
      if (gmu_g) reinterpret_cast<V *>(gmu_g)[v] = out_g;
      if (gy) reinterpret_cast<V *>(gy)[v] = out_y;
      p += dp_step;
      j += dj_step;
      if (j >= Dy) {
        j -= Dy;
        ++p;
      }
    }
  }
}

template <typename T>
static bool launch_logweight_bwd_dense(const View3 *v, const void *grad_lw, void *gx, void *gmu_p, void *gy,
                                       void *gmu_g, void *gmu_q, void *gs_p, void *gs_g, void *gs_q, int64_t B,
                                       int64_t K, int64_t Dx, int64_t Dy, hipStream_t s, const void *lw,
                                       const void *lse, const void *grad_lse) {
  constexpr int N = Vec16<T>::N;
  if (gs_p != nullptr || gs_g != nullptr || gs_q != nullptr) return false;       // scale gradients: generic kernel
  if (!is_scalar(v[2].st) || !is_scalar(v[5].st) || !is_scalar(v[7].st)) return false;
  const int64_t total_x = B * K * Dx, total_y = B * K * Dy;
  if (total_x >= (1ll << 32) / 2 || total_y >= (1ll << 32) / 2 || total_x % N != 0 || total_y % N != 0) return false;
  auto flat = [&](const View3 &view, int64_t D) {
    return view.st.d == 1 && view.st.k == D && view.st.b == K * D && (reinterpret_cast<uintptr_t>(view.ptr) & 15u) == 0;
  };
  if (!flat(v[0], Dx) || !flat(v[1], Dx) || !flat(v[6], Dx) || !flat(v[4], Dy)) return false;
  const bool y_dense = flat(v[3], Dy);
  const bool y_rows = v[3].st.k == 0 && (v[3].st.d == 1 || Dy == 1);             // one observation row per batch element
  if (!y_dense && !y_rows) return false;
  if (gy != nullptr && !y_dense) return false;                                    // dense gy only for a dense y
  for (void *out : {gx, gmu_p, gy, gmu_g, gmu_q})
    if (out != nullptr && (reinterpret_cast<uintptr_t>(out) & 15u) != 0) return false;
  const uint32_t vec_x = (uint32_t)(total_x / N), vec_y = (uint32_t)(total_y / N);
  const uint32_t most = vec_x > vec_y ? vec_x : vec_y;
  uint32_t blocks = (most + kLpBlock - 1) / kLpBlock;
  if (blocks > 256u * 16u) blocks = 256u * 16u;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((normal_logweight_bwd_dense_kernel<T>), dim3(blocks), dim3(kLpBlock), 0, s,
                     (const T *)v[0].ptr, (const T *)v[1].ptr, (const T *)v[6].ptr, (const T *)v[3].ptr,
                     (const T *)v[4].ptr, (const T *)v[2].ptr, (const T *)v[5].ptr, (const T *)v[7].ptr,
                     y_dense ? K * Dy : v[3].st.b, y_dense ? 1 : 0, (const T *)grad_lw, (const T *)lw,
                     (const T *)lse, (const T *)grad_lse, (T *)gx, (T *)gmu_p, (T *)gy, (T *)gmu_g, (T *)gmu_q, vec_x,
                     vec_y, (uint32_t)K, (uint32_t)Dx, (uint32_t)Dy);
  return true;
}

// K5 backward, the layout of a FIRST timestep (aesmc/inference.py:79-98: `initial()` NOT_EXPANDED, the time-0 proposal
// BATCH_EXPANDED): x dense [B,K,Dx], mu_g dense [B,K,Dy], and every other operand — the prior's and the proposal's
// location and scale, the observation, the emission's scale — constant along the particles (particle stride 0: a scalar,
// a per-column vector, one row per batch element).  The generic kernel addresses each element through strided views
// with two 64-bit divisions: 523 us at B=1024 K=4096 d=10.  Here a workgroup lies inside ONE batch row and a lane walks
// 16-byte vectors of that row's flat elements (coalesced loads of x / mu_g, 16-byte stores of every gradient) in 32-bit
// arithmetic; the row-constant operands are 4-byte reads of a few cache lines.  Element arithmetic is the generic
// kernel's, operation for operation: same bits.
template <typename T>
__global__ __launch_bounds__(kLpBlock) void normal_logweight_bwd_rows_kernel(
    const T *__restrict__ x, const T *__restrict__ mu_g, View3 mu_p, View3 sc_p, View3 y, View3 sc_g, View3 mu_q, View3 sc_q,
    const T *__restrict__ grad_lw, const T *__restrict__ lw, const T *__restrict__ lse, const T *__restrict__ grad_lse,
    T *__restrict__ gx, T *__restrict__ gmu_p, T *__restrict__ gy, T *__restrict__ gmu_g, T *__restrict__ gmu_q,
    T *__restrict__ gs_p, T *__restrict__ gs_g, T *__restrict__ gs_q, uint32_t K, uint32_t Dx, uint32_t Dy) {
  constexpr int N = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const uint32_t b = blockIdx.y;
  const uint32_t v = blockIdx.x * kLpBlock + threadIdx.x;      // this lane's vector of the row
  auto incoming = [&](uint32_t p) -> T {   // as in normal_logweight_bwd_kernel
    T g = grad_lse != nullptr ? grad_lse[b] * Num<T>::exp(lw[p] - lse[b]) : T(0);
    if (grad_lw != nullptr) g = grad_lse != nullptr ? g + grad_lw[p] : grad_lw[p];
    return g;
  };
  auto row = [&](const View3 &view, uint32_t j) {
    return reinterpret_cast<const T *>(view.ptr)[(int64_t)b * view.st.b + (int64_t)j * view.st.d];
  };
  auto put = [](V &out, int r, T value) {
    if (r == 0) out.x = value;
    else if (r == 1) out.y = value;
    if constexpr (N == 4) {
      if (r == 2) out.z = value;
      else if (r == 3) out.w = value;
    }
  };
  if ((gx != nullptr || gmu_p != nullptr || gmu_q != nullptr || gs_p != nullptr || gs_q != nullptr) && v * N < K * Dx) {
    const uint32_t at = (b * K * Dx) / N + v;      // (a row is whole vectors: the host admits K Dx % N == 0)
    uint32_t k = (v * N) / Dx, j = v * N - k * Dx;
    const V xv = reinterpret_cast<const V *>(x)[at];
    V out_p, out_q, out_x, out_sp, out_sq;
    T g = incoming(b * K + k);
#pragma unroll
    for (int r = 0; r < N; ++r) {
      const T value = Vec16<T>::get(xv, r);
      const T s_p = row(sc_p, j), s_q = row(sc_q, j);
      const T var_p = s_p * s_p, var_q = s_q * s_q;
      const T dp = value - row(mu_p, j), dq = value - row(mu_q, j);
      const T gq = -g;                                  // log q enters the weight with a minus sign
      const T gz_p = g * (dp / var_p);
      const T gz_q = gq * (dq / var_q);
      put(out_p, r, gz_p);
      put(out_q, r, gz_q);
      put(out_x, r, (-gz_p) + (-gz_q));
      if (gs_p) put(out_sp, r, g * ((dp * dp) / (var_p * s_p) - T(1) / s_p));
      if (gs_q) put(out_sq, r, gq * ((dq * dq) / (var_q * s_q) - T(1) / s_q));
      if (++j == Dx && r + 1 < N) {
        j = 0;
        ++k;
        g = incoming(b * K + k);
      }
    }
    if (gmu_p) reinterpret_cast<V *>(gmu_p)[at] = out_p;
    if (gmu_q) reinterpret_cast<V *>(gmu_q)[at] = out_q;
    if (gx) reinterpret_cast<V *>(gx)[at] = out_x;
    if (gs_p) reinterpret_cast<V *>(gs_p)[at] = out_sp;
    if (gs_q) reinterpret_cast<V *>(gs_q)[at] = out_sq;
  }
  if ((gy != nullptr || gmu_g != nullptr || gs_g != nullptr) && v * N < K * Dy) {
    const uint32_t at = (b * K * Dy) / N + v;
    uint32_t k = (v * N) / Dy, j = v * N - k * Dy;
    const V gv = reinterpret_cast<const V *>(mu_g)[at];
    V out_g, out_y, out_sg;
    T g = incoming(b * K + k);
#pragma unroll
    for (int r = 0; r < N; ++r) {
      const T s_g = row(sc_g, j);
      const T var_g = s_g * s_g;
      const T dg = row(y, j) - Vec16<T>::get(gv, r);
      const T gz_g = g * (dg / var_g);
      put(out_g, r, gz_g);
      put(out_y, r, -gz_g);
      if (gs_g) put(out_sg, r, g * ((dg * dg) / (var_g * s_g) - T(1) / s_g));
      if (++j == Dy && r + 1 < N) {
        j = 0;
        ++k;
        g = incoming(b * K + k);
      }
    }
    if (gmu_g) reinterpret_cast<V *>(gmu_g)[at] = out_g;
    if (gy) reinterpret_cast<V *>(gy)[at] = out_y;
    if (gs_g) reinterpret_cast<V *>(gs_g)[at] = out_sg;
  }
}

template <typename T>
static bool launch_logweight_bwd_rows(const View3 *v, const void *grad_lw, void *gx, void *gmu_p, void *gy,
                                      void *gmu_g, void *gmu_q, void *gs_p, void *gs_g, void *gs_q, int64_t B,
                                      int64_t K, int64_t Dx, int64_t Dy, hipStream_t s, const void *lw,
                                      const void *lse, const void *grad_lse) {
  constexpr int N = Vec16<T>::N;
  if (B * K * std::max(Dx, Dy) >= (1ll << 31) || B > 65535 || K < 2) return false;
  if ((K * Dx) % N != 0 || (K * Dy) % N != 0) return false;      // a batch row: whole 16-byte vectors
  auto flat = [&](const View3 &view, int64_t D) {
    return (D == 1 || view.st.d == 1) && view.st.k == D && view.st.b == K * D &&
           (reinterpret_cast<uintptr_t>(view.ptr) & 15u) == 0;
  };
  if (!flat(v[0], Dx) || !flat(v[4], Dy)) return false;
  for (int i : {1, 2, 3, 5, 6, 7})
    if (v[i].st.k != 0) return false;
  for (void *out : {gx, gmu_p, gy, gmu_g, gmu_q, gs_p, gs_g, gs_q})
    if (out != nullptr && (reinterpret_cast<uintptr_t>(out) & 15u) != 0) return false;
  const int64_t vectors = K * std::max(Dx, Dy) / N;
  const dim3 grid((unsigned)((vectors + kLpBlock - 1) / kLpBlock), (unsigned)B);
  hipLaunchKernelGGL((normal_logweight_bwd_rows_kernel<T>), grid, dim3(kLpBlock), 0, s, (const T *)v[0].ptr,
                     (const T *)v[4].ptr, v[1], v[2], v[3], v[5], v[6], v[7], (const T *)grad_lw, (const T *)lw,
                     (const T *)lse, (const T *)grad_lse, (T *)gx, (T *)gmu_p, (T *)gy, (T *)gmu_g, (T *)gmu_q, (T *)gs_p,
                     (T *)gs_g, (T *)gs_q, (uint32_t)K, (uint32_t)Dx, (uint32_t)Dy);
  return true;
}

static int g_lw_bwd_last = 0;      // which form K5's backward took last: 1 dense, 2 rows, 3 generic (test hook below)

template <typename T>
static int launch_logweight_bwd(const View3 *v, const void *grad_lw, void *gx, void *gmu_p, void *gy,
                                void *gmu_g, void *gmu_q, void *gs_p, void *gs_g, void *gs_q, int64_t B,
                                int64_t K, int64_t Dx, int64_t Dy, hipStream_t s, const void *lw = nullptr,
                                const void *lse = nullptr, const void *grad_lse = nullptr) {
  g_lw_bwd_last = 1;
  if (launch_logweight_bwd_dense<T>(v, grad_lw, gx, gmu_p, gy, gmu_g, gmu_q, gs_p, gs_g, gs_q, B, K, Dx, Dy, s, lw,
                                    lse, grad_lse))
    return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
  g_lw_bwd_last = 2;
  if (launch_logweight_bwd_rows<T>(v, grad_lw, gx, gmu_p, gy, gmu_g, gmu_q, gs_p, gs_g, gs_q, B, K, Dx, Dy, s, lw, lse,
                                   grad_lse))
    return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
  g_lw_bwd_last = 3;
  const int64_t total_x = B * K * Dx, total_y = B * K * Dy;
  const int64_t most = total_x > total_y ? total_x : total_y;
  int64_t blocks = (most + kLpBlock - 1) / kLpBlock;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((normal_logweight_bwd_kernel<T>), dim3((unsigned)blocks), dim3(kLpBlock), 0, s, v[0], v[1],
                     v[2], v[3], v[4], v[5], v[6], v[7], (const T *)grad_lw, (T *)gx, (T *)gmu_p, (T *)gy,
                     (T *)gmu_g, (T *)gmu_q, (T *)gs_p, (T *)gs_g, (T *)gs_q, total_x, total_y, (uint32_t)K,
                     (uint32_t)Dx, (uint32_t)Dy, (const T *)lw, (const T *)lse, (const T *)grad_lse);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

}  // namespace aesmc

extern "C" int aesmc_test_last_logweight_backward_form(void) { return aesmc::g_lw_bwd_last; }

extern "C" int aesmc_normal_logprob_sum(int dtype, const void *value, const void *loc,
                                        const void *scale, void *out, int64_t B, int64_t K,
                                        int64_t D, int64_t v_sb, int64_t v_sk, int64_t v_sd,
                                        int64_t m_sb, int64_t m_sk, int64_t m_sd, int64_t s_sb,
                                        int64_t s_sk, int64_t s_sd, void *stream) {
  using namespace aesmc;
  if (!value || !loc || !scale || !out || B < 0 || K < 0 || D < 0) return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0) return AESMC_OK;
  if (K >= (1ll << 31) || D >= (1ll << 31) || B >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  if (D == 0) {  // empty event: the sum is 0
    return zero_fill_async(out, (size_t)B * K * (dtype == AESMC_F32 ? 4 : 8), (hipStream_t)stream)
               ? AESMC_OK : AESMC_ERR_LAUNCH;
  }
  Strides3 sv{v_sb, v_sk, v_sd}, sm{m_sb, m_sk, m_sd}, ss{s_sb, s_sk, s_sd};
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32) return launch_lps<float>(value, loc, scale, out, B, K, D, sv, sm, ss, s);
  if (dtype == AESMC_F64) return launch_lps<double>(value, loc, scale, out, B, K, D, sv, sm, ss, s);
  return AESMC_ERR_INVALID_ARGUMENT;
}

extern "C" int aesmc_normal_logprob_sum_backward(
    int dtype, const void *value, const void *loc, const void *scale, const void *grad_out,
    void *grad_value, void *grad_loc, void *grad_scale, int64_t B, int64_t K, int64_t D, int64_t v_sb,
    int64_t v_sk, int64_t v_sd, int64_t m_sb, int64_t m_sk, int64_t m_sd, int64_t s_sb, int64_t s_sk,
    int64_t s_sd, void *stream) {
  using namespace aesmc;
  if (!value || !loc || !scale || !grad_out || B < 0 || K < 0 || D < 0) return AESMC_ERR_INVALID_ARGUMENT;
  if (!grad_value && !grad_loc && !grad_scale) return AESMC_OK;
  if (B == 0 || K == 0 || D == 0) return AESMC_OK;
  if (K >= (1ll << 31) || D >= (1ll << 31) || B >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  Strides3 sv{v_sb, v_sk, v_sd}, sm{m_sb, m_sk, m_sd}, ss{s_sb, s_sk, s_sd};
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32)
    return launch_lps_bwd<float>(value, loc, scale, grad_out, grad_value, grad_loc, grad_scale, B, K, D,
                                 sv, sm, ss, s);
  if (dtype == AESMC_F64)
    return launch_lps_bwd<double>(value, loc, scale, grad_out, grad_value, grad_loc, grad_scale, B, K, D,
                                  sv, sm, ss, s);
  return AESMC_ERR_INVALID_ARGUMENT;
}

extern "C" int aesmc_normal_logweight(int dtype, const aesmc_view3 *views, void *out_lw, int64_t B,
                                      int64_t K, int64_t Dx, int64_t Dy, void *stream) {
  using namespace aesmc;
  if (views == nullptr || out_lw == nullptr || B < 0 || K < 0) return AESMC_ERR_INVALID_ARGUMENT;
  View3 v[8];
  for (int i = 0; i < 8; ++i) {
    if (views[i].ptr == nullptr) return AESMC_ERR_INVALID_ARGUMENT;
    v[i].ptr = views[i].ptr;
    v[i].st = Strides3{views[i].stride_b, views[i].stride_k, views[i].stride_d};
  }
  if (B == 0 || K == 0) return AESMC_OK;
  if (K >= (1ll << 31) || B >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32) return launch_logweight<float>(v, out_lw, B, K, Dx, Dy, s);
  if (dtype == AESMC_F64) return launch_logweight<double>(v, out_lw, B, K, Dx, Dy, s);
  return AESMC_ERR_INVALID_ARGUMENT;
}

extern "C" int aesmc_normal_logweight_backward(int dtype, const aesmc_view3 *views, const void *grad_lw,
                                               void *grad_x, void *grad_mu_p, void *grad_y, void *grad_mu_g,
                                               void *grad_mu_q, void *grad_s_p, void *grad_s_g, void *grad_s_q,
                                               int64_t B, int64_t K, int64_t Dx, int64_t Dy, void *stream) {
  using namespace aesmc;
  if (views == nullptr || grad_lw == nullptr || B < 0 || K < 0 || Dx < 1 || Dy < 1) return AESMC_ERR_INVALID_ARGUMENT;
  View3 v[8];
  for (int i = 0; i < 8; ++i) {
    if (views[i].ptr == nullptr) return AESMC_ERR_INVALID_ARGUMENT;
    v[i].ptr = views[i].ptr;
    v[i].st = Strides3{views[i].stride_b, views[i].stride_k, views[i].stride_d};
  }
  if (!grad_x && !grad_mu_p && !grad_y && !grad_mu_g && !grad_mu_q && !grad_s_p && !grad_s_g && !grad_s_q)
    return AESMC_OK;
  if (B == 0 || K == 0) return AESMC_OK;
  if (K >= (1ll << 31) || B >= (1ll << 31) || Dx >= (1ll << 31) || Dy >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32)
    return launch_logweight_bwd<float>(v, grad_lw, grad_x, grad_mu_p, grad_y, grad_mu_g, grad_mu_q, grad_s_p,
                                       grad_s_g, grad_s_q, B, K, Dx, Dy, s);
  if (dtype == AESMC_F64)
    return launch_logweight_bwd<double>(v, grad_lw, grad_x, grad_mu_p, grad_y, grad_mu_g, grad_mu_q, grad_s_p,
                                        grad_s_g, grad_s_q, B, K, Dx, Dy, s);
  return AESMC_ERR_INVALID_ARGUMENT;
}

extern "C" int aesmc_normal_logweight_lse_backward(int dtype, const aesmc_view3 *views, const void *lw,
                                                   const void *lse, const void *grad_lse, const void *grad_lw,
                                                   void *grad_x, void *grad_mu_p, void *grad_y, void *grad_mu_g,
                                                   void *grad_mu_q, void *grad_s_p, void *grad_s_g,
                                                   void *grad_s_q, int64_t B, int64_t K, int64_t Dx,
                                                   int64_t Dy, void *stream) {
  using namespace aesmc;
  if (views == nullptr || lw == nullptr || lse == nullptr || grad_lse == nullptr || B < 0 || K < 0 || Dx < 1 ||
      Dy < 1)
    return AESMC_ERR_INVALID_ARGUMENT;
  View3 v[8];
  for (int i = 0; i < 8; ++i) {
    if (views[i].ptr == nullptr) return AESMC_ERR_INVALID_ARGUMENT;
    v[i].ptr = views[i].ptr;
    v[i].st = Strides3{views[i].stride_b, views[i].stride_k, views[i].stride_d};
  }
  if (!grad_x && !grad_mu_p && !grad_y && !grad_mu_g && !grad_mu_q && !grad_s_p && !grad_s_g && !grad_s_q)
    return AESMC_OK;
  if (B == 0 || K == 0) return AESMC_OK;
  if (K >= (1ll << 31) || B >= (1ll << 31) || Dx >= (1ll << 31) || Dy >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32)
    return launch_logweight_bwd<float>(v, grad_lw, grad_x, grad_mu_p, grad_y, grad_mu_g, grad_mu_q, grad_s_p,
                                       grad_s_g, grad_s_q, B, K, Dx, Dy, s, lw, lse, grad_lse);
  if (dtype == AESMC_F64)
    return launch_logweight_bwd<double>(v, grad_lw, grad_x, grad_mu_p, grad_y, grad_mu_g, grad_mu_q, grad_s_p,
                                        grad_s_g, grad_s_q, B, K, Dx, Dy, s, lw, lse, grad_lse);
  return AESMC_ERR_INVALID_ARGUMENT;
}
