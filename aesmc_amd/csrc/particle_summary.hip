// K7: weighted particle summaries in one pass over the particles.
//
//   w[b,k]        = softmax_k log_w[b,k]
//   mean[b,j]     = sum_k w[b,k] value[b,k,j]                  aesmc/statistics.py:47-60
//   second[b,j]   = sum_k w[b,k] value[b,k,j]^2                aesmc/statistics.py:63-76 (variance =
//                                                              second - mean^2, formed by the caller)
//   log_ess[b]    = 2 lse_k(log_w) - lse_k(2 log_w)            aesmc/statistics.py:79-91
//
// The reference loops over the K particles in Python (statistics.py:32-42: two elementwise
// launches per particle) after a softmax, and takes two log-sum-exps for the ESS.  Here a batch
// row is cut into S slices of particles (S = 1 when there are rows enough to fill the chip); one
// workgroup owns a slice: slice max, then ONE sweep in which TX = min(D, 256) lanes walk a
// particle's values (consecutive lanes read consecutive elements, also across particle
// boundaries) while TY = 256 / TX lane groups stride over k, four particles in flight per lane.
// Every lane keeps its columns' sums of e = exp(log_w - max) times v and v^2 in registers; lane 0
// of a group also sum e and sum e^2.  Partials meet in LDS and are added in a fixed order.  With
// S > 1 the slice records (max, sum e, sum e^2, column sums) go to a workspace and a second small
// kernel merges them by rescaling to the row max — again in a fixed order, so results are
// reproducible.  Sums are divided by sum e at the end (the reference multiplies by the normalised
// weight first: same value up to rounding; tolerance in tests/test_gpu_kernels.py).
// HBM-bound: value is read once (4 D B per particle), log_w twice (L2 hit).
#include "common.hpp"

namespace aesmc {

constexpr int kSumBlock = 256;
constexpr int kSumRegs = 4;    // columns per lane per sweep: one sweep covers TX * kSumRegs columns
constexpr int kSumUnroll = 4;  // particles in flight per lane

struct SumStrides {
  int64_t b, k, d;
};

// Record of one slice in the workspace: [max, sum e, sum e^2, A_0..A_{D-1}, Q_0..Q_{D-1}]
__host__ __device__ __forceinline__ int64_t record_elems(int64_t D) { return 3 + 2 * D; }

static inline uint32_t pick_slices(int64_t B, int64_t K) {
  // at least ~4 workgroups per CU in total, at least 512 particles per slice
  int64_t want = (1024 + B - 1) / B;
  int64_t most = K / 512;
  if (most < 1) most = 1;
  if (want > most) want = most;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  return (uint32_t)want;
}

// VEC: the values of a particle are dense and 16-byte aligned — a lane then owns kSumRegs CONSECUTIVE
// columns and fetches them with 16-byte loads (TX = D / kSumRegs lanes per particle: 32 at D = 128, so
// eight particles per pass and four passes in flight), instead of kSumRegs columns TX apart by 4-byte
// loads.  Same sums in the same order per column.
template <typename T, bool VEC>
__global__ __launch_bounds__(kSumBlock) void particle_summary_kernel(
    const T *__restrict__ log_w, const T *__restrict__ value, SumStrides sv, T *__restrict__ out_log_ess,
    T *__restrict__ out_mean, T *__restrict__ out_second, T *__restrict__ records, uint32_t K, uint32_t D,
    uint32_t TX, uint32_t S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sum_smem[];
  T *partial = reinterpret_cast<T *>(sum_smem);  // [TY][TX * kSumRegs][2] column sums / [kSumBlock][2]
  __shared__ T red_m[kSumBlock / kWave];
  __shared__ int red_nan[kSumBlock / kWave];
  __shared__ T slice_stats[2];                   // sum e, sum e^2

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid % kWave, wave = tid / kWave;
  const int64_t b = blockIdx.x / S;
  const uint32_t slice = blockIdx.x - (uint32_t)b * S;
  const uint32_t per_slice = (K + S - 1) / S;
  const uint32_t k_lo = min(K, slice * per_slice), k_hi = min(K, k_lo + per_slice);
  const T *lw = log_w + b * (int64_t)K;

  // ---- slice max and NaN scan --------------------------------------------------------------------
  T m = Num<T>::neg_inf();
  int has_nan = 0;
  for (uint32_t k = k_lo + tid; k < k_hi; k += kSumBlock) {
    const T v = lw[k];
    has_nan |= (v != v);
    m = Num<T>::max(m, v);
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    m = Num<T>::max(m, __shfl_xor(m, off, kWave));
    has_nan |= __shfl_xor(has_nan, off, kWave);
  }
  if (lane == 0) {
    red_m[wave] = m;
    red_nan[wave] = has_nan;
  }
  __syncthreads();
  m = red_m[0];
  has_nan = red_nan[0];
  for (int w = 1; w < kSumBlock / kWave; ++w) {
    m = Num<T>::max(m, red_m[w]);
    has_nan |= red_nan[w];
  }
  if (has_nan) m = Num<T>::nan();
  // exp(x - max) needs a finite max; an empty or all -inf slice contributes nothing, +inf / NaN
  // poison the row (handled where the row is finalised)
  const bool usable = m > Num<T>::neg_inf() && m < Num<T>::pos_inf();

  const uint32_t TY = kSumBlock / TX;
  const uint32_t tx = tid % TX, ty = tid / TX;   // lanes with ty >= TY (TX not a power of two) idle
  const uint32_t cols_per_sweep = TX * kSumRegs;
  const uint32_t num_cols = value != nullptr ? D : 0;
  const T *vrow = value != nullptr ? value + b * sv.b : nullptr;
  T *record = records != nullptr ? records + ((int64_t)b * S + slice) * record_elems(D) : nullptr;

  for (uint32_t c0 = 0; c0 == 0 || c0 < num_cols; c0 += cols_per_sweep) {
    T acc1[kSumRegs], acc2[kSumRegs];
#pragma unroll
    for (int r = 0; r < kSumRegs; ++r) acc1[r] = acc2[r] = T(0);
    T s1 = T(0), s2 = T(0);
    if (usable && ty < TY) {
      for (uint32_t k0 = k_lo + ty; k0 < k_hi; k0 += TY * kSumUnroll) {
        T e[kSumUnroll];
        T v[kSumUnroll][kSumRegs];
#pragma unroll
        for (int q = 0; q < kSumUnroll; ++q) {
          const uint32_t k = k0 + q * TY;
          const bool live = k < k_hi;
          e[q] = live ? Num<T>::exp(lw[k] - m) : T(0);
          if constexpr (VEC) {
            constexpr int N = Vec16<T>::N;
            using V = typename Vec16<T>::type;
            const uint32_t j = c0 + tx * kSumRegs;            // num_cols is a multiple of kSumRegs
            if (live && j < num_cols) {
              const V *src = reinterpret_cast<const V *>(vrow + (int64_t)k * sv.k + j);
#pragma unroll
              for (int h = 0; h < kSumRegs / N; ++h) {
                const V packed = src[h];
#pragma unroll
                for (int r = 0; r < N; ++r) v[q][h * N + r] = Vec16<T>::get(packed, r);
              }
            } else {
#pragma unroll
              for (int r = 0; r < kSumRegs; ++r) v[q][r] = T(0);
            }
          } else {
#pragma unroll
            for (int r = 0; r < kSumRegs; ++r) {
              const uint32_t j = c0 + r * TX + tx;
              v[q][r] = (live && j < num_cols) ? vrow[(int64_t)k * sv.k + (int64_t)j * sv.d] : T(0);
            }
          }
        }
#pragma unroll
        for (int q = 0; q < kSumUnroll; ++q) {
          if (tx == 0 && c0 == 0) {
            s1 += e[q];
            s2 += e[q] * e[q];
          }
#pragma unroll
          for (int r = 0; r < kSumRegs; ++r) {
            const T ev = e[q] * v[q][r];
            acc1[r] += ev;
            acc2[r] += ev * v[q][r];
          }
        }
      }
    }
    // ---- weight sums (first sweep only): one pair per lane group, added in group order ------------
    if (c0 == 0) {
      __syncthreads();
      partial[2 * tid] = s1;
      partial[2 * tid + 1] = s2;
      __syncthreads();
      if (tid == 0) {
        T t1 = T(0), t2 = T(0);
        for (uint32_t i = 0; i < TY; ++i) {   // only tx == 0 lanes hold non-zero sums
          t1 += partial[2 * (i * TX)];
          t2 += partial[2 * (i * TX) + 1];
        }
        slice_stats[0] = t1;
        slice_stats[1] = t2;
        if (record != nullptr) {
          record[0] = m;
          record[1] = t1;
          record[2] = t2;
        }
      }
      __syncthreads();
    }
    if (num_cols == 0) break;
    // ---- column sums of this sweep ----------------------------------------------------------------
    __syncthreads();
    if (ty < TY) {
#pragma unroll
      for (int r = 0; r < kSumRegs; ++r) {
        const uint32_t col = VEC ? tx * kSumRegs + r : r * TX + tx;
        partial[(ty * cols_per_sweep + col) * 2] = acc1[r];
        partial[(ty * cols_per_sweep + col) * 2 + 1] = acc2[r];
      }
    }
    __syncthreads();
    for (uint32_t c = tid; c < cols_per_sweep && c0 + c < num_cols; c += kSumBlock) {
      T t1 = T(0), t2 = T(0);
      for (uint32_t i = 0; i < TY; ++i) {
        t1 += partial[(i * cols_per_sweep + c) * 2];
        t2 += partial[(i * cols_per_sweep + c) * 2 + 1];
      }
      if (record != nullptr) {
        record[3 + c0 + c] = t1;
        record[3 + D + c0 + c] = t2;
      } else {
        const T total = slice_stats[0];
        const int64_t o = b * (int64_t)D + c0 + c;
        if (out_mean != nullptr) out_mean[o] = usable ? t1 / total : Num<T>::nan();
        if (out_second != nullptr) out_second[o] = usable ? t2 / total : Num<T>::nan();
      }
    }
  }
  if (record == nullptr && out_log_ess != nullptr && tid == 0) {
    // 2 (m + log S1) - (2 m + log S2): the max cancels, so huge offsets do not lose digits
    out_log_ess[b] = usable ? T(2) * Num<T>::log(slice_stats[0]) - Num<T>::log(slice_stats[1]) : Num<T>::nan();
  }
}

// Few values per particle (D <= kSmallD): the sweep above would leave most of a wavefront's lanes
// computing the same exp and issue 4-byte loads.  Instead a tile of 256 particles is copied to LDS
// in memory order (16-byte loads when the slice is dense), then lane p owns particle p: one exp,
// D products, all sums in registers; the 2 D + 2 lane sums meet by wavefront shuffles and four
// LDS slots.  Same slicing, records and merge as the general kernel.
constexpr int kSmallD = 16;

template <typename T>
__global__ __launch_bounds__(kSumBlock) void particle_summary_small_kernel(
    const T *__restrict__ log_w, const T *__restrict__ value, SumStrides sv, T *__restrict__ out_log_ess,
    T *__restrict__ out_mean, T *__restrict__ out_second, T *__restrict__ records, uint32_t K, uint32_t D,
    uint32_t S, int dense, uint32_t R /* particles per lane per tile: R * D <= kSmallD */) {
  constexpr int N = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  __shared__ T tile[kSumBlock * kSmallD + (kSumBlock * kSmallD) / 32 + 1];
  __shared__ T red_m[kSumBlock / kWave];
  __shared__ int red_nan[kSumBlock / kWave];
  __shared__ T red_sum[kSumBlock / kWave][2 * kSmallD + 2];

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid % kWave, wave = tid / kWave;
  const int64_t b = blockIdx.x / S;
  const uint32_t slice = blockIdx.x - (uint32_t)b * S;
  const uint32_t per_slice = (K + S - 1) / S;
  const uint32_t k_lo = min(K, slice * per_slice), k_hi = min(K, k_lo + per_slice);
  const T *lw = log_w + b * (int64_t)K;

  T m = Num<T>::neg_inf();
  int has_nan = 0;
  for (uint32_t k = k_lo + tid; k < k_hi; k += kSumBlock) {
    const T v = lw[k];
    has_nan |= (v != v);
    m = Num<T>::max(m, v);
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    m = Num<T>::max(m, __shfl_xor(m, off, kWave));
    has_nan |= __shfl_xor(has_nan, off, kWave);
  }
  if (lane == 0) {
    red_m[wave] = m;
    red_nan[wave] = has_nan;
  }
  __syncthreads();
  m = red_m[0];
  has_nan = red_nan[0];
  for (int w = 1; w < kSumBlock / kWave; ++w) {
    m = Num<T>::max(m, red_m[w]);
    has_nan |= red_nan[w];
  }
  if (has_nan) m = Num<T>::nan();
  const bool usable = m > Num<T>::neg_inf() && m < Num<T>::pos_inf();

  T acc[2 * kSmallD + 2];   // [0..D) sum e v, [kSmallD..kSmallD+D) sum e v^2, then sum e, sum e^2
#pragma unroll
  for (int i = 0; i < 2 * kSmallD + 2; ++i) acc[i] = T(0);
  const T *vrow = value + b * sv.b;
  if (usable) {
    const uint32_t per_tile = kSumBlock * R;
    for (uint32_t t0 = k_lo; t0 < k_hi; t0 += per_tile) {
      const uint32_t np = min(per_tile, k_hi - t0);
      const uint32_t ne = np * D;
      __syncthreads();   // the previous tile has been consumed
      if (dense) {       // the slice's rows are one contiguous, 16-byte aligned run
        const T *src = vrow + (int64_t)t0 * D;
        const uint32_t nvec = ne / N;
        for (uint32_t i = tid; i < nvec; i += kSumBlock) {
          const V packed = reinterpret_cast<const V *>(src)[i];
#pragma unroll
          for (int r = 0; r < N; ++r) tile[i * N + r + ((i * N + r) >> 5)] = Vec16<T>::get(packed, r);
        }
        for (uint32_t e = nvec * N + tid; e < ne; e += kSumBlock) tile[e + (e >> 5)] = src[e];
      } else {
        for (uint32_t e = tid; e < ne; e += kSumBlock) {
          const uint32_t p = e / D, j = e - p * D;
          tile[e + (e >> 5)] = vrow[(int64_t)(t0 + p) * sv.k + (int64_t)j * sv.d];
        }
      }
      __syncthreads();
      for (uint32_t p = tid; p < np; p += kSumBlock) {
        const T e = Num<T>::exp(lw[t0 + p] - m);
        acc[2 * kSmallD] += e;
        acc[2 * kSmallD + 1] += e * e;
        const uint32_t base = p * D;
#pragma unroll
        for (int j = 0; j < kSmallD; ++j) {
          if ((uint32_t)j < D) {
            const T v = tile[base + j + ((base + j) >> 5)];
            const T ev = e * v;
            acc[j] += ev;
            acc[kSmallD + j] += ev * v;
          }
        }
      }
    }
  }
  // ---- 256 lane sums -> one, per quantity: shuffles inside a wavefront, then the 4 wavefronts ----
#pragma unroll
  for (int i = 0; i < 2 * kSmallD + 2; ++i) {
    T x = acc[i];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
    if (lane == 0) red_sum[wave][i] = x;
  }
  __syncthreads();
  if (tid < 2 * kSmallD + 2) {
    T total = red_sum[0][tid];
    for (int w = 1; w < kSumBlock / kWave; ++w) total += red_sum[w][tid];
    red_sum[0][tid] = total;
  }
  __syncthreads();
  const T s1 = red_sum[0][2 * kSmallD], s2 = red_sum[0][2 * kSmallD + 1];
  T *record = records != nullptr ? records + ((int64_t)b * S + slice) * record_elems(D) : nullptr;
  if (record != nullptr) {
    if (tid == 0) {
      record[0] = m;
      record[1] = s1;
      record[2] = s2;
    }
    if (tid < D) {
      record[3 + tid] = red_sum[0][tid];
      record[3 + D + tid] = red_sum[0][kSmallD + tid];
    }
  } else {
    if (tid < D) {
      const int64_t o = b * (int64_t)D + tid;
      if (out_mean != nullptr) out_mean[o] = usable ? red_sum[0][tid] / s1 : Num<T>::nan();
      if (out_second != nullptr) out_second[o] = usable ? red_sum[0][kSmallD + tid] / s1 : Num<T>::nan();
    }
    if (out_log_ess != nullptr && tid == 0)
      out_log_ess[b] = usable ? T(2) * Num<T>::log(s1) - Num<T>::log(s2) : Num<T>::nan();
  }
}

// Merges the S slice records of a batch row: everything rescaled to the row max, slices in order.
template <typename T>
__global__ __launch_bounds__(kSumBlock) void particle_summary_merge_kernel(
    const T *__restrict__ records, T *__restrict__ out_log_ess, T *__restrict__ out_mean,
    T *__restrict__ out_second, uint32_t D, uint32_t S) {
  const int64_t b = blockIdx.x;
  const int64_t stride = record_elems(D);
  const T *row = records + b * (int64_t)S * stride;
  T m = Num<T>::neg_inf();
  bool poisoned = false;
  for (uint32_t s = 0; s < S; ++s) {
    const T ms = row[s * stride];
    poisoned |= (ms != ms) || ms == Num<T>::pos_inf();
    m = Num<T>::max(m, ms);
  }
  const bool usable = !poisoned && m > Num<T>::neg_inf();
  T s1 = T(0), s2 = T(0);
  if (usable) {
    for (uint32_t s = 0; s < S; ++s) {
      const T ms = row[s * stride];
      if (ms == Num<T>::neg_inf()) continue;   // slice of zero-weight particles
      const T f = Num<T>::exp(ms - m);
      s1 += row[s * stride + 1] * f;
      s2 += row[s * stride + 2] * (f * f);
    }
  }
  for (uint32_t c = threadIdx.x; c < D; c += kSumBlock) {
    T t1 = T(0), t2 = T(0);
    if (usable) {
      for (uint32_t s = 0; s < S; ++s) {
        const T ms = row[s * stride];
        if (ms == Num<T>::neg_inf()) continue;
        const T f = Num<T>::exp(ms - m);
        t1 += row[s * stride + 3 + c] * f;
        t2 += row[s * stride + 3 + D + c] * f;
      }
    }
    if (out_mean != nullptr) out_mean[b * (int64_t)D + c] = usable ? t1 / s1 : Num<T>::nan();
    if (out_second != nullptr) out_second[b * (int64_t)D + c] = usable ? t2 / s1 : Num<T>::nan();
  }
  if (out_log_ess != nullptr && threadIdx.x == 0)
    out_log_ess[b] = usable ? T(2) * Num<T>::log(s1) - Num<T>::log(s2) : Num<T>::nan();
}

template <typename T>
static int launch_summary(const void *log_w, const aesmc_view3 *value, void *out_log_ess, void *out_mean,
                          void *out_second, int64_t B, int64_t K, int64_t D, void *ws, size_t ws_bytes,
                          hipStream_t s) {
  uint32_t TX = 1;
  if (value != nullptr) TX = (uint32_t)(D < kSumBlock ? D : kSumBlock);
  // 16-byte loads: dense, aligned rows whose length is a whole number of lane shares
  const bool vec = value != nullptr && D > kSmallD && D % kSumRegs == 0 && value->stride_d == 1 &&
                   (reinterpret_cast<uintptr_t>(value->ptr) & 15u) == 0 &&
                   (value->stride_b * (int64_t)sizeof(T)) % 16 == 0 && (value->stride_k * (int64_t)sizeof(T)) % 16 == 0;
  if (vec) TX = (uint32_t)(D / kSumRegs < kSumBlock ? D / kSumRegs : kSumBlock);
  const uint32_t TY = kSumBlock / TX;
  size_t lds = (size_t)TY * TX * kSumRegs * 2 * sizeof(T);
  const size_t lds_weights = (size_t)kSumBlock * 2 * sizeof(T);
  if (lds < lds_weights) lds = lds_weights;
  SumStrides sv{0, 0, 0};
  const T *v = nullptr;
  if (value != nullptr) {
    v = static_cast<const T *>(value->ptr);
    sv = SumStrides{value->stride_b, value->stride_k, value->stride_d};
  }
  const uint32_t S = pick_slices(B, K);
  T *records = nullptr;
  if (S > 1) {
    if (ws == nullptr || ws_bytes < (size_t)B * S * record_elems(value != nullptr ? D : 0) * sizeof(T))
      return AESMC_ERR_WORKSPACE;
    records = static_cast<T *>(ws);
  }
  if ((uint64_t)B * S > 0x7fffffffull) return AESMC_ERR_UNSUPPORTED;
  const uint32_t cols = value != nullptr ? (uint32_t)D : 0u;
  if (value != nullptr && D >= 3 && D <= kSmallD) {   // D <= 2: the lane-per-element sweep is already dense
    constexpr int N = Vec16<T>::N;
    // dense: particle rows back to back inside a batch row, and every slice / tile start 16-byte aligned
    const uint32_t per_slice = (uint32_t)((K + S - 1) / S);
    const bool dense = (D == 1 || sv.d == 1) && (K == 1 || sv.k == D) &&
                       ((reinterpret_cast<uintptr_t>(v) & 15u) == 0) && ((sv.b * (int64_t)sizeof(T)) % 16 == 0) &&
                       (((uint64_t)per_slice * D) % N == 0) && (((uint64_t)kSumBlock * D) % N == 0);
    const uint32_t R = (uint32_t)(kSmallD / D);
    hipLaunchKernelGGL(particle_summary_small_kernel<T>, dim3((unsigned)(B * S)), dim3(kSumBlock), 0, s,
                       static_cast<const T *>(log_w), v, sv, static_cast<T *>(out_log_ess),
                       static_cast<T *>(out_mean), static_cast<T *>(out_second), records, (uint32_t)K, cols, S,
                       dense ? 1 : 0, R);
  } else if (vec)
    hipLaunchKernelGGL((particle_summary_kernel<T, true>), dim3((unsigned)(B * S)), dim3(kSumBlock), lds, s,
                       static_cast<const T *>(log_w), v, sv, static_cast<T *>(out_log_ess),
                       static_cast<T *>(out_mean), static_cast<T *>(out_second), records, (uint32_t)K, cols, TX,
                       S);
  else
    hipLaunchKernelGGL((particle_summary_kernel<T, false>), dim3((unsigned)(B * S)), dim3(kSumBlock), lds, s,
                       static_cast<const T *>(log_w), v, sv, static_cast<T *>(out_log_ess),
                       static_cast<T *>(out_mean), static_cast<T *>(out_second), records, (uint32_t)K, cols, TX,
                       S);
  if (S > 1)
    hipLaunchKernelGGL(particle_summary_merge_kernel<T>, dim3((unsigned)B), dim3(kSumBlock), 0, s, records,
                       static_cast<T *>(out_log_ess), static_cast<T *>(out_mean),
                       static_cast<T *>(out_second), cols, S);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

}  // namespace aesmc

extern "C" size_t aesmc_particle_summary_workspace_bytes(int dtype, int64_t B, int64_t K, int64_t D) {
  if (B <= 0 || K <= 0 || D < 0) return 0;
  const uint32_t S = aesmc::pick_slices(B, K);
  if (S <= 1) return 0;
  return (size_t)B * S * (size_t)aesmc::record_elems(D) * (dtype == AESMC_F64 ? 8 : 4);
}

extern "C" int aesmc_particle_summary(int dtype, const void *log_w, const aesmc_view3 *value,
                                      void *out_log_ess, void *out_mean, void *out_second, int64_t B,
                                      int64_t K, int64_t D, void *ws, size_t ws_bytes, void *stream) {
  using namespace aesmc;
  if (log_w == nullptr || B < 0 || K < 0 || D < 0) return AESMC_ERR_INVALID_ARGUMENT;
  if (value != nullptr && value->ptr == nullptr) return AESMC_ERR_INVALID_ARGUMENT;
  if (value == nullptr && (out_mean != nullptr || out_second != nullptr)) return AESMC_ERR_INVALID_ARGUMENT;
  if (out_log_ess == nullptr && out_mean == nullptr && out_second == nullptr) return AESMC_OK;
  if (B == 0) return AESMC_OK;
  if (K == 0) return AESMC_ERR_INVALID_ARGUMENT;  // no particles: the weights are undefined
  if (K >= (1ll << 31) || D >= (1ll << 31) || B >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  if (value != nullptr && D == 0) value = nullptr;  // empty rows: nothing to average
  if (value == nullptr) {
    out_mean = out_second = nullptr;
    D = 0;
  }
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32)
    return launch_summary<float>(log_w, value, out_log_ess, out_mean, out_second, B, K, D, ws, ws_bytes, s);
  if (dtype == AESMC_F64)
    return launch_summary<double>(log_w, value, out_log_ess, out_mean, out_second, B, K, D, ws, ws_bytes, s);
  return AESMC_ERR_INVALID_ARGUMENT;
}
