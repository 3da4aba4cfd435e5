// K2: systematic ancestral resampling as a per-sequence segmented prefix scan over the weight CDF.
//
// Replaces aesmc/inference.py:234-269 (+ aesmc/math.py:33-51, numpy branch): the reference copies
// the log-weights to the host, runs scipy logsumexp / np.exp / np.cumsum and a Python loop of
// np.digitize per batch row, then copies int64 indices back.  Here one workgroup owns one batch row
// and every lane owns kChunk (8, 4 or 2) CONSECUTIVE particles (blocked layout: a wavefront reads /
// writes one contiguous span, and the scan needs one cross-lane step per kChunk particles):
//
//   pass 1  row maximum + NaN scan                        (coalesced reads, shuffle + LDS reduce)
//   pass 2  w = exp(lw - max) in float64; per-lane running sums in registers; one wavefront scan
//           of the lane totals (shuffles); wavefront totals through LDS; unnormalised CDF -> LDS
//   pass 3  c = CDF / total, correctly rounded (c[K-1] == 1.0 as in the reference's c / max(c))
//   pass 4  idx[k] = #{ j : c[j] <= (u + k) / K }: binary search for a lane's first particle,
//           exponential (galloping) search from the previous answer for the next ones — ancestors
//           are monotone in k, so consecutive particles land a few slots apart
//
// The CDF lives in LDS (8 B x K, padded; up to 160 KiB -> K <= kLdsMaxParticles); larger K uses a
// caller-supplied global workspace with the same code path.  Float64 inside regardless of the I/O
// dtype: the result then does not depend on the scan's association order (SURVEY.md section 7,
// hard part 1) and matches the reference bit-for-bit on float64 inputs.
//
// The kernel is VALU-bound on float64 arithmetic (rocprofv3: ~230 VALU instructions per particle
// in the first, strided version), hence: a short exp() specialised to arguments <= 0, and both
// per-particle divisions done as reciprocal + two FMAs, which still yields the correctly rounded
// quotient (Markstein's theorem; verified against true division in tests/test_oracle.py).
#include "../../aesmc_amd/csrc/common.hpp"

namespace aesmc {

constexpr int kMaxThreads = 1024;
constexpr int kScratchDoubles = 64;  // per-workgroup LDS scratch (wavefront totals, reduce slots)
// The CDF is stored with one padding slot per 8 entries: lane t writes entries 8t..8t+7, and the
// pad turns the 64-byte lane stride into 72 bytes, which spreads the lanes over the LDS banks.
__host__ __device__ __forceinline__ int64_t cdf_slot(int64_t e) { return e + (e >> 3); }
__host__ __device__ __forceinline__ int64_t cdf_row_slots(int64_t K) { return cdf_slot(K) + 1; }
// 160 KiB LDS per workgroup on gfx950, 1 KiB kept free: K + K/8 + 1 + scratch doubles must fit.
constexpr int64_t kLdsMaxParticles = (((160 * 1024 - 1024) / 8 - kScratchDoubles - 2) * 8) / 9;

// exp(x) for x <= 0 in float64: Cody-Waite reduction x = n ln2 + r, |r| <= ln2 / 2, degree-13
// Taylor polynomial (truncation < 5e-18 relative), scaling by v_ldexp_f64.  exp(0) == 1 exactly.
__device__ __forceinline__ double exp_nonpositive(double x) {
  if (!(x > -745.2)) return 0.0;  // exp underflows to zero below ~ -745.13 (also catches -inf)
  const double n = __builtin_rint(x * 1.4426950408889634074);
  double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);
  r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;               // 1/13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);   // 1/12!
  p = __builtin_fma(p, r, 2.505210838544172e-08);  // 1/11!
  p = __builtin_fma(p, r, 2.755731922398589e-07);  // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06); // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);   // 1/8!
  p = __builtin_fma(p, r, 1.984126984126984e-04);  // 1/7!
  p = __builtin_fma(p, r, 1.3888888888888889e-03); // 1/6!
  p = __builtin_fma(p, r, 8.333333333333333e-03);  // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666664e-02); // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666666e-01); // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)n);
}

// a / b given y = 1 / b (a true, correctly rounded division done once per row): q0 = a y,
// r = a - b q0 (exact in an FMA), q = q0 + r y is the correctly rounded quotient.
__device__ __forceinline__ double divide_with_reciprocal(double a, double b, double y) {
  const double q0 = a * y;
  const double r = __builtin_fma(-b, q0, a);
  return __builtin_fma(r, y, q0);
}

template <typename T, int kChunk, bool CDF_IN_LDS, int STOP>
__global__ __launch_bounds__(kMaxThreads) void exp_kernel(
    const T *__restrict__ log_w, const double *__restrict__ u, int64_t *__restrict__ out_idx,
    int32_t *flags, int K, double *__restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *scratch = smem;
  double *cdf = CDF_IN_LDS ? (smem + kScratchDoubles) : (ws + (size_t)blockIdx.x * (size_t)cdf_row_slots(K));

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

  if (STOP == 1) { if (tid == 0) idx[0] = (int64_t)dm; return; }
  // ---- pass 2: float64 weights, blocked scan ---------------------------------------------------
  // Round r covers particles [r * nt * kChunk, (r + 1) * nt * kChunk); lane `tid` owns kChunk
  // consecutive ones.  `carry` is the sum of all earlier rounds.
  const int per_round = nt * kChunk;
  double carry = 0.0;
  for (int round_base = 0; round_base < K; round_base += per_round) {
    const int first = round_base + tid * kChunk;
    double s[kChunk];
    double run = 0.0;
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
      const int k = first + i;
      run += (k < K) ? exp_nonpositive((double)lw[k < K ? k : 0] - dm) : 0.0;
      s[i] = run;
    }
    // inclusive scan of the lane totals across the wavefront
    double incl = run;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const double y = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += y;
    }
    double excl = __shfl_up(incl, 1, kWave);  // exclusive prefix of this lane inside its wavefront
    if (lane == 0) excl = 0.0;
    if (lane == kWave - 1) scratch[wave] = incl;
    __syncthreads();
    double base = carry, round_total = 0.0;
    for (int w = 0; w < nwaves; ++w) {
      if (w == wave) base = carry + round_total;
      round_total += scratch[w];
    }
    base += excl;
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
      const int k = first + i;
      if (k < K) cdf[cdf_slot(k)] = base + s[i];
    }
    carry += round_total;
    __syncthreads();  // scratch is rewritten by the next round
  }

  if (STOP == 2) { __syncthreads(); if (tid == 0) idx[0] = (int64_t)cdf[cdf_slot(K - 1)]; return; }
  // ---- pass 3: normalise by the row total (every lane rereads only what it wrote) -------------
  // The last particle's CDF entry was formed by exactly the additions that formed `carry`'s
  // summands in a different association; dividing by that entry itself keeps c[K-1] == 1.0.
  __syncthreads();
  const double total = cdf[cdf_slot(K - 1)];
  __syncthreads();
  const double inv_total = 1.0 / total;
  for (int round_base = 0; round_base < K; round_base += per_round) {
    const int first = round_base + tid * kChunk;
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
      const int k = first + i;
      if (k < K) cdf[cdf_slot(k)] = divide_with_reciprocal(cdf[cdf_slot(k)], total, inv_total);
    }
  }
  __syncthreads();

  if (STOP == 3) { if (tid == 0) idx[0] = (int64_t)(cdf[cdf_slot(K / 2)] * 1000.0); return; }
  // ---- pass 4: idx[k] = #{ j : c[j] <= (u + k) / K } -------------------------------------------
  const double ub = u[row];
  const double dK = (double)K;
  const double inv_K = 1.0 / dK;
  for (int round_base = 0; round_base < K; round_base += per_round) {
    const int first = round_base + tid * kChunk;
    if (first >= K) break;
    int64_t found[kChunk];
    int answer = 0;
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
      const int k = first + i;
      const double pos = divide_with_reciprocal(ub + (double)k, dK, inv_K);
      int left, right;
      if (i == 0) {
        left = 0;
        right = K;
      } else {  // gallop from the previous particle's answer
        left = answer;
        int probe = answer, step = 1;
        while (probe < K && cdf[cdf_slot(probe)] <= pos) {
          left = probe + 1;
          probe += step;
          step <<= 1;
        }
        right = probe < K ? probe : K;
      }
      while (left < right) {
        const int mid = (left + right) >> 1;
        if (cdf[cdf_slot(mid)] <= pos)
          left = mid + 1;
        else
          right = mid;
      }
      answer = left;
      found[i] = (int64_t)left;
    }
    if (first + kChunk <= K && (((uintptr_t)(idx + first)) & 15u) == 0) {
#pragma unroll
      for (int i = 0; i < kChunk; i += 2) {
        longlong2 pair;
        pair.x = found[i];
        pair.y = found[i + 1];
        *reinterpret_cast<longlong2 *>(idx + first + i) = pair;
      }
    } else {
#pragma unroll
      for (int i = 0; i < kChunk; ++i)
        if (first + i < K) idx[first + i] = found[i];
    }
  }
}


}  // namespace aesmc
#include <cstdio>
#include <vector>
#include <cmath>
using namespace aesmc;
template <int CH, int STOP> float run(const float* lw, const double* u, int64_t* idx, int B, int K, int nt) {
  size_t lds = (size_t)(cdf_row_slots(K) + kScratchDoubles) * sizeof(double);
  hipFuncSetAttribute((const void*)exp_kernel<float, CH, true, STOP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((exp_kernel<float, CH, true, STOP>), dim3(B), dim3(nt), lds, 0, lw, u, idx, (int32_t*)nullptr, K, (double*)nullptr);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((exp_kernel<float, CH, true, STOP>), dim3(B), dim3(nt), lds, 0, lw, u, idx, (int32_t*)nullptr, K, (double*)nullptr);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / 20;
}
int main() {
  for (auto shape : std::vector<std::pair<int,int>>{{1024, 4096}, {256, 1024}, {4096, 8192}}) {
    int B = shape.first, K = shape.second;
    std::vector<float> h((size_t)B * K); std::vector<double> hu(B);
    unsigned s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; float a = (s >> 8) / 16777216.f; s = s * 1664525u + 1013904223u; float b = (s >> 8) / 16777216.f; v = sqrtf(-2.f * logf(a + 1e-7f)) * cosf(6.2831853f * b); }
    for (auto& v : hu) { s = s * 1664525u + 1013904223u; v = (s >> 8) / 16777216.0; }
    float* lw; double* u; int64_t* idx;
    hipMalloc(&lw, h.size() * 4); hipMalloc(&u, B * 8); hipMalloc(&idx, (size_t)B * K * 8);
    hipMemcpy(lw, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(u, hu.data(), B * 8, hipMemcpyHostToDevice);
    int nt8 = std::min(1024, ((K + 7) / 8 + 63) / 64 * 64), nt4 = std::min(1024, ((K + 3) / 4 + 63) / 64 * 64);
    printf("B=%d K=%d chunk8 nt=%d: pass1 %.1f  +pass2 %.1f  +pass3 %.1f  full %.1f us\n", B, K, nt8,
           run<8, 1>(lw, u, idx, B, K, nt8), run<8, 2>(lw, u, idx, B, K, nt8), run<8, 3>(lw, u, idx, B, K, nt8), run<8, 0>(lw, u, idx, B, K, nt8));
    printf("B=%d K=%d chunk4 nt=%d: pass1 %.1f  +pass2 %.1f  +pass3 %.1f  full %.1f us\n", B, K, nt4,
           run<4, 1>(lw, u, idx, B, K, nt4), run<4, 2>(lw, u, idx, B, K, nt4), run<4, 3>(lw, u, idx, B, K, nt4), run<4, 0>(lw, u, idx, B, K, nt4));
    hipFree(lw); hipFree(u); hipFree(idx);
  }
  return 0;
}
