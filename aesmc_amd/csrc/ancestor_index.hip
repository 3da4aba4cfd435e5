// K2: systematic ancestral resampling as a per-sequence segmented prefix scan over the weight CDF.
//
// Replaces aesmc/inference.py:234-269 (+ aesmc/math.py:33-51, numpy branch): the reference copies
// the log-weights to the host, runs scipy logsumexp / np.exp / np.cumsum and a Python loop of
// np.digitize per batch row, then copies int64 indices back.  Here one workgroup owns one batch row:
//
//   pass 1  row maximum + NaN scan                       (coalesced global reads, shuffle + LDS reduce)
//   pass 2  w = exp(lw - max) in float64, inclusive scan  (each wavefront scans a contiguous span in
//           64-element steps with lane shuffles and a running carry; span totals go through LDS)
//   pass 3  c = (local + span offset) / total             (float64 true division; c[K-1] == 1.0)
//   pass 4  idx[k] = upper_bound(c, (u + k) / K)          (binary search in the LDS-resident CDF)
//
// The CDF lives in LDS (8 B x K, up to 160 KiB -> K <= kLdsMaxParticles); larger K uses a
// caller-supplied global workspace with the same code path.  Float64 inside regardless of the I/O
// dtype: the result then does not depend on the scan's association order (SURVEY.md section 7,
// hard part 1) and matches the reference bit-for-bit on float64 inputs.
#include "common.hpp"

namespace aesmc {

constexpr int kMaxThreads = 1024;
constexpr int kScratchDoubles = 64;  // per-workgroup LDS scratch (wave totals, reduce slots)
// 160 KiB LDS per workgroup on gfx950; keep 1 KiB of headroom beyond CDF + scratch.
constexpr int64_t kLdsMaxParticles = (160 * 1024 - 1024) / 8 - kScratchDoubles;

template <typename T, bool CDF_IN_LDS>
__global__ __launch_bounds__(kMaxThreads) void ancestor_index_kernel(
    const T *__restrict__ log_w, const double *__restrict__ u, int64_t *__restrict__ out_idx,
    int32_t *flags, int K, double *__restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *scratch = smem;
  double *cdf = CDF_IN_LDS ? (smem + kScratchDoubles) : (ws + (size_t)blockIdx.x * (size_t)K);

  const int tid = threadIdx.x;
  const int nt = blockDim.x;
  const int lane = tid % kWave;
  const int wave = tid / kWave;
  const int nwaves = nt / kWave;
  const int64_t row = blockIdx.x;
  const T *lw = log_w + row * (int64_t)K;
  int64_t *idx = out_idx + row * (int64_t)K;

  // ---- pass 1: row max, NaN detection ---------------------------------------------------------
  T m = Num<T>::neg_inf();
  int has_nan = 0;
  for (int k = tid; k < K; k += nt) {
    T v = lw[k];
    has_nan |= (v != v);
    m = Num<T>::max(m, v);
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    m = Num<T>::max(m, __shfl_xor(m, off, kWave));
    has_nan |= __shfl_xor(has_nan, off, kWave);
  }
  int *scratch_i = reinterpret_cast<int *>(scratch + 32);
  if (lane == 0) {
    scratch[wave] = (double)m;
    scratch_i[wave] = has_nan;
  }
  __syncthreads();
  double dm = scratch[0];
  has_nan = scratch_i[0];
  for (int w = 1; w < nwaves; ++w) {
    dm = fmax(dm, scratch[w]);
    has_nan |= scratch_i[w];
  }
  __syncthreads();  // scratch is reused below

  const bool degenerate = has_nan || !(dm > -__builtin_huge_val() && dm < __builtin_huge_val());
  if (degenerate) {
    // Reference: NaN -> FloatingPointError (inference.py:244-245); max = +-inf -> NaN CDF ->
    // np.digitize returns K for every particle.  Both are reported through `flags`.
    if (tid == 0) raise_flag(flags, has_nan ? AESMC_FLAG_NAN_LOG_WEIGHT : AESMC_FLAG_DEGENERATE_ROW);
    for (int k = tid; k < K; k += nt) idx[k] = (int64_t)K;
    return;
  }

  // ---- pass 2: float64 weights + wavefront-blocked inclusive scan -----------------------------
  // Wavefront w owns the contiguous span [w*span, min(K, (w+1)*span)), span a multiple of 64.
  const int span = ((K + nwaves - 1) / nwaves + kWave - 1) / kWave * kWave;
  const int begin = wave * span;
  const int end = min(K, begin + span);
  double carry = 0.0;
  for (int base = begin; base < end; base += kWave) {
    const int k = base + lane;
    double x = (k < end) ? ::exp((double)lw[k] - dm) : 0.0;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      double y = __shfl_up(x, off, kWave);
      if (lane >= off) x += y;
    }
    x += carry;
    if (k < end) cdf[k] = x;
    carry = __shfl(x, kWave - 1, kWave);
  }
  if (lane == 0) scratch[wave] = carry;
  __syncthreads();

  // ---- pass 3: span offsets, normalise by the row total ---------------------------------------
  double offset = 0.0, total = 0.0;
  for (int w = 0; w < nwaves; ++w) {
    if (w == wave) offset = total;
    total += scratch[w];
  }
  for (int k = begin + lane; k < end; k += kWave) cdf[k] = (cdf[k] + offset) / total;
  __syncthreads();

  // ---- pass 4: idx[k] = #{ j : c[j] <= (u + k) / K } -------------------------------------------
  const double ub = u[row];
  const double dK = (double)K;
  for (int k = tid; k < K; k += nt) {
    const double pos = (ub + (double)k) / dK;
    int lo = 0, hi = K;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= pos)
        lo = mid + 1;
      else
        hi = mid;
    }
    idx[k] = (int64_t)lo;
  }
}

static int pick_threads(int64_t K) {
  if (K <= 512) return 128;
  if (K <= 2048) return 256;
  if (K <= 8192) return 512;
  return 1024;
}

template <typename T>
static int launch(const void *log_w, const double *u, int64_t *idx, int32_t *flags, int64_t B,
                  int64_t K, void *ws, size_t ws_bytes, hipStream_t s) {
  const int nt = pick_threads(K);
  if (K <= kLdsMaxParticles) {
    const size_t lds = (size_t)(K + kScratchDoubles) * sizeof(double);
    static bool attr_set = false;  // raise the dynamic-LDS cap once per process and instantiation
    if (!attr_set) {
      if (hipFuncSetAttribute((const void *)ancestor_index_kernel<T, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return AESMC_ERR_LAUNCH;
      attr_set = true;
    }
    hipLaunchKernelGGL((ancestor_index_kernel<T, true>), dim3((unsigned)B), dim3(nt), lds, s,
                       (const T *)log_w, u, idx, flags, (int)K, (double *)nullptr);
  } else {
    if (ws == nullptr || ws_bytes < (size_t)B * (size_t)K * sizeof(double)) return AESMC_ERR_WORKSPACE;
    const size_t lds = (size_t)kScratchDoubles * sizeof(double);
    hipLaunchKernelGGL((ancestor_index_kernel<T, false>), dim3((unsigned)B), dim3(nt), lds, s,
                       (const T *)log_w, u, idx, flags, (int)K, (double *)ws);
  }
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

}  // namespace aesmc

extern "C" int64_t aesmc_ancestor_index_lds_max_particles(void) { return aesmc::kLdsMaxParticles; }

extern "C" size_t aesmc_workspace_bytes(int64_t B, int64_t K) {
  if (B <= 0 || K <= aesmc::kLdsMaxParticles) return 0;
  return (size_t)B * (size_t)K * sizeof(double);
}

extern "C" int aesmc_ancestor_index(int dtype, const void *log_w, const double *u, int64_t *out_idx,
                                    int32_t *flags, int64_t B, int64_t K, void *ws, size_t ws_bytes,
                                    void *stream) {
  if (log_w == nullptr || u == nullptr || out_idx == nullptr || B < 0 || K < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0) return AESMC_OK;
  if (K > 0x3fffffffLL || B > 0x7fffffffLL) return AESMC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32) return aesmc::launch<float>(log_w, u, out_idx, flags, B, K, ws, ws_bytes, s);
  if (dtype == AESMC_F64) return aesmc::launch<double>(log_w, u, out_idx, flags, B, K, ws, ws_bytes, s);
  return AESMC_ERR_INVALID_ARGUMENT;
}
