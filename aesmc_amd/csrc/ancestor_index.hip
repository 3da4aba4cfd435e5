// K2: systematic ancestral resampling as a per-sequence segmented prefix scan over the weight CDF.
//
// Replaces aesmc/inference.py:234-269 (+ aesmc/math.py:33-51, numpy branch): the reference copies
// the log-weights to the host, runs scipy logsumexp / np.exp / np.cumsum and a Python loop of
// np.digitize per batch row, then copies int64 indices back.  Here one workgroup owns one batch row
// and every lane owns C CONSECUTIVE particles (blocked layout: a wavefront reads / writes one
// contiguous span, and a scan needs one cross-lane step per C particles, not per particle).
//
// Two kernels share the arithmetic helpers below:
//   * ancestor_index_inv_kernel (K <= 32768, the one that runs in practice): CDF entries stay in
//     registers; ancestor indices come from inverting the CDF per SOURCE particle and an int32
//     max-scan over the positions — no search (see the comment above the kernel);
//   * ancestor_index_kernel (larger K): the float64 CDF is stored (caller workspace) and every
//     position searches it, galloping from the previous particle's answer.
//
// Float64 inside regardless of the I/O dtype: the result then does not depend on the scan's
// association order (hard part 1) and matches the reference bit-for-bit on
// float64 inputs.  The kernels are VALU-bound on float64 arithmetic (rocprofv3: ~230 VALU
// instructions per particle in the first, strided + binary-search version), hence: a short exp()
// specialised to arguments <= 0, and the per-particle divisions done as reciprocal + two FMAs,
// which still yields the correctly rounded quotient (Markstein's theorem; verified against true
// division in exact rational arithmetic in tests/test_oracle.py).
#include "common.hpp"

namespace aesmc {

constexpr int kMaxThreads = 1024;
constexpr int kScratchDoubles = 64;  // per-workgroup LDS scratch (wavefront totals, reduce slots)
constexpr int kScanSlot = 40;        // inv kernel: [0,16) maxima, [32,40) int flags / max-scan, [40,58) the scan's
// Stored-CDF kernel (K > 32768): the row's float64 CDF lives in the caller's workspace with one
// padding slot per 8 entries (lane t writes entries 8t..8t+7: a 72-byte lane stride keeps the
// lanes of a wavefront in different memory channels); aesmc_workspace_bytes accounts for the pad.
__host__ __device__ __forceinline__ int64_t cdf_slot(int64_t e) { return e + (e >> 3); }
__host__ __device__ __forceinline__ int64_t cdf_row_slots(int64_t K) { return cdf_slot(K) + 1; }

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

template <typename T, int kChunk>
__global__ __launch_bounds__(kMaxThreads) void ancestor_index_kernel(
    const T *__restrict__ log_w, const double *__restrict__ u, int64_t *__restrict__ out_idx,
    int32_t *flags, int K, double *__restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *scratch = smem;
  double *cdf = ws + (size_t)blockIdx.x * (size_t)cdf_row_slots(K);   // this row's slice of the workspace

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

// ---- primary kernel: everything in registers, no per-particle search ------------------------------
//
// Each lane owns C consecutive particles j and keeps their CDF entries in registers.  Instead of
// searching the CDF for every position k, it inverts the question: first[j] = the first k whose
// position (u + k) / K reaches c[j], i.e. ceil(c[j] K - u) corrected against the exactly rounded
// positions.  Since idx[k] = #{j : first[j] <= k}, lane j drops the marker j + 1 at k = first[j]
// (when any particle starts there) into an LDS array, and an inclusive max-scan over k — the same
// blocked scan as for the CDF, on int32 — turns the markers into the ancestor indices.  O(1) work
// per particle, no data-dependent loops, 4 B of LDS per particle (K <= 32768 in one workgroup).
constexpr int kInvMaxChunk = 32;
constexpr int64_t kInvMaxParticles = (int64_t)kMaxThreads * kInvMaxChunk;

// Optional tail of the kernel (the fused resampling step): with the row's ancestor indices still
// in LDS, copy the payload rows  dst[b,k,:] = src[b, idx[b,k], :]  — K3's chunking (16-byte
// stores, G-byte source pieces), minus the round trip of the int64 indices through HBM and one
// launch.  `src == nullptr` switches it off.
struct StepPayload {
  const char *src;
  char *dst;
  int64_t stride_b, stride_k;  // bytes
  uint32_t ppp;                // G-byte pieces per particle row
  int G;                       // 4, 8 or 16
};

template <int G, int U>
__device__ __forceinline__ void gather_row_from_lds(const int *anc, const char *srow, char *drow,
                                                    uint32_t K, uint32_t ppp, int64_t stride_k,
                                                    uint32_t tid, uint32_t nt, uint32_t part,
                                                    uint32_t parts) {
  constexpr int V = 16 / G;
  // U chunks in flight per lane: 5 for the small kernels (K=1024 d=10 has 10 per lane: two trips), 4
  // where a fifth would cost the 8-particles-per-lane kernel its third workgroup per CU (85 registers)
  using P = typename Piece<G>::type;
  const uint64_t row_pieces = (uint64_t)K * ppp;             // a multiple of V (checked on the host)
  const uint32_t row_chunks = (uint32_t)(row_pieces / V);
  // a batch row shared by `parts` workgroups: this one copies chunks [begin, chunks)
  const uint32_t per_part = (row_chunks + parts - 1) / parts;
  const uint32_t begin = min(row_chunks, part * per_part);
  const uint32_t chunks = min(row_chunks, begin + per_part);
  // A lane's chunks lie nt apart: (particle, piece) of a chunk's first piece advance by a constant
  // (dk, dr) from one to the next — one integer division per lane instead of one per chunk.
  const uint32_t step = nt * V;
  const uint32_t dk = step / ppp, dr = step - dk * ppp;
  uint32_t k = ((begin + tid) * V) / ppp;
  uint32_t r = (begin + tid) * V - k * ppp;
  for (uint32_t c0 = begin; c0 < chunks; c0 += nt * U) {
    P piece[U][V];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t chunk = c0 + j * nt + tid;
      const bool live = chunk < chunks;
      uint32_t kk = live ? k : 0u, rr = live ? r : 0u;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        piece[j][i] = *reinterpret_cast<const P *>(srow + (int64_t)anc[kk] * stride_k + (uint64_t)rr * G);
        if (++rr == ppp) {
          rr = 0;
          ++kk;
        }
      }
      k += dk;
      r += dr;
      if (r >= ppp) {
        r -= ppp;
        ++k;
      }
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t chunk = c0 + j * nt + tid;
      if (chunk < chunks) {
        uint4 packed;
        __builtin_memcpy(&packed, piece[j], 16);
        *reinterpret_cast<uint4 *>(drow + (uint64_t)chunk * 16) = packed;
      }
    }
  }
}

// min over lanes >= this one of x (inclusive), by six bpermute steps: used only where the children ranges are written
// (training), to make them monotone on knife-edge rows — see the clamp in ancestor_index_inv_kernel.
__device__ __forceinline__ int wave_suffix_min(int x, int lane) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int other = __shfl_down(x, d, kWave);
    if (lane + d < kWave) x = min(x, other);
  }
  return x;
}

// (Holding the kernel to 64 registers — four 512-lane workgroups per CU instead of three, so that
// 1024 batch rows are resident at once — was measured and bought nothing: 78.4 vs 77.7 us at B=1024
// K=4096, with a 20-byte spill.)
// PAYLOAD = false: the instantiation without the tail (K2 alone, what a step whose propagation kernel fetches the rows
// itself launches).  Without the copy's chunks in flight it is held to 64 registers: eight wavefronts per SIMD, so four
// 512-lane workgroups per CU and 1024 batch rows resident at once instead of 768 and a second round.
template <typename T, int C, bool PAYLOAD>
__global__ __launch_bounds__(kMaxThreads, (PAYLOAD || C > 8) ? 1 : 8) void ancestor_index_inv_kernel(
    const T *__restrict__ log_w, const double *__restrict__ u, int64_t *__restrict__ out_idx,
    int32_t *flags, int K, T *__restrict__ out_lse, StepPayload payload, int B, int parts,
    int32_t *__restrict__ out_child_end) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *scratch = smem;                                        // [64]
  int *scratch_i = reinterpret_cast<int *>(scratch + 32);
  int *marker = reinterpret_cast<int *>(smem + kScratchDoubles);  // [K rounded up to C]
  const int tid = threadIdx.x;
  const int nt = blockDim.x;
  const int lane = tid % kWave;
  const int wave = tid / kWave;
  const int nwaves = nt / kWave;
  int *first_of_lane = marker + nt * C;                           // [nwaves]: first[] of each wavefront's lane 0
  // `parts` workgroups share a batch row (grid = parts * B, see launch_inv): each repeats the
  // row's scan — 4 B per particle, L2-resident after the first reader — and owns 1/parts of the
  // OUTPUT: the lanes [part, part + 1) * nt / parts store their indices, and the payload copy is
  // cut by 16-byte chunks.  No workgroup waits for another.
  const int64_t row = blockIdx.x % (unsigned)B;
  const uint32_t part = blockIdx.x / (unsigned)B;
  const bool owns_idx = (uint32_t)tid * parts / nt == part;       // nt is a multiple of parts
  const T *lw = log_w + row * (int64_t)K;
  int64_t *idx = out_idx + row * (int64_t)K;
  const int j0 = tid * C;                                         // nt * C >= K: one round
  // the row's uniform is needed only after the scan, but a load issued there (behind three barriers,
  // which the compiler may not move it across) would stall every lane for a full memory latency
  const double ub = u[row];

  // ---- load once, row max + NaN scan ---------------------------------------------------------
  T v[C];
  constexpr int NV = Vec16<T>::N;
  if (C % NV == 0 && j0 + C <= K && (((uintptr_t)(lw + j0)) & 15u) == 0) {
    using V = typename Vec16<T>::type;                       // the lane's C values as 16-byte loads
#pragma unroll
    for (int q = 0; q < C / NV; ++q) {
      const V packed = reinterpret_cast<const V *>(lw + j0)[q];
#pragma unroll
      for (int r = 0; r < NV; ++r) v[q * NV + r] = Vec16<T>::get(packed, r);
    }
  } else {
#pragma unroll
    for (int i = 0; i < C; ++i) v[i] = (j0 + i < K) ? lw[j0 + i] : Num<T>::neg_inf();
  }
  T m = Num<T>::neg_inf();
  int has_nan = 0;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    has_nan |= (v[i] != v[i]);
    m = Num<T>::max(m, v[i]);
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    m = Num<T>::max(m, __shfl_xor(m, off, kWave));
    has_nan |= __shfl_xor(has_nan, off, kWave);
  }
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
  // (no barrier here: the scan below publishes into scratch slots of its own, kScanSlot onwards)
  const bool degenerate = has_nan || !(dm > -__builtin_huge_val() && dm < __builtin_huge_val());
  if (degenerate) {  // same conventions as the reference: see include/aesmc_hip.h, K2
    if (tid == 0 && part == 0) {
      raise_flag(flags, has_nan ? AESMC_FLAG_NAN_LOG_WEIGHT : AESMC_FLAG_DEGENERATE_ROW);
      // torch.logsumexp's values for such rows (K1 returns the same)
      if (out_lse != nullptr) out_lse[row] = has_nan ? Num<T>::nan() : (T)dm;
    }
    if (owns_idx)
      for (int i = 0; i < C; ++i)
        if (j0 + i < K) {
          idx[j0 + i] = (int64_t)K;
          if (out_child_end != nullptr) out_child_end[row * (int64_t)K + j0 + i] = 0;     // nobody has children
        }
    if (!PAYLOAD || payload.src == nullptr) return;
    // the unfused route would clamp the out-of-range index K to K - 1 in K3: same bytes here
    for (int k = tid; k < nt * C; k += nt) marker[k] = K - 1;
    __syncthreads();
    const char *srow = payload.src + row * payload.stride_b;
    char *drow = payload.dst + (uint64_t)row * K * payload.ppp * payload.G;
    if (payload.G == 16)
      gather_row_from_lds<16, (C <= 4 ? 5 : 4)>(marker, srow, drow, K, payload.ppp, payload.stride_k, tid, nt, part, parts);
    else if (payload.G == 8)
      gather_row_from_lds<8, (C <= 4 ? 5 : 4)>(marker, srow, drow, K, payload.ppp, payload.stride_k, tid, nt, part, parts);
    else
      gather_row_from_lds<4, (C <= 4 ? 5 : 4)>(marker, srow, drow, K, payload.ppp, payload.stride_k, tid, nt, part, parts);
    return;
  }

  // ---- float64 weights, blocked inclusive scan ---------------------------------------------------
  double s[C];
  double run = 0.0;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    run += (j0 + i < K) ? exp_nonpositive((double)v[i] - dm) : 0.0;
    s[i] = run;
  }
  double incl = run;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const double y = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += y;
  }
  double base = __shfl_up(incl, 1, kWave);
  if (lane == 0) base = 0.0;
  double *scan = scratch + kScanSlot;                          // [16] wavefront totals, [16] / [17]: see below
  if (lane == kWave - 1) scan[wave] = incl;
  // The CDF's last entry is the normaliser, so that c[K-1] == 1.0 exactly (reference: c / max(c)).
  // Its owner publishes the two terms only it has — its exclusive prefix inside the wavefront and
  // its running sum up to particle K - 1 — BEFORE the barrier; every lane then adds the earlier
  // wavefronts' totals in the owner's own order: the same value bit for bit, one barrier fewer.
  const int last_wave = ((K - 1) / C) / kWave;
  if (j0 <= K - 1 && K - 1 < j0 + C) {
    scan[16] = base;
    scan[17] = s[K - 1 - j0];
  }
  __syncthreads();
  for (int w = 0; w < wave; ++w) base += scan[w];
  double total = scan[16];
  for (int w = 0; w < last_wave; ++w) total += scan[w];
  total += scan[17];
  const double inv_total = 1.0 / total;
  // by-product: logsumexp of the row (the step's contribution to log Z), float64 inside
  if (out_lse != nullptr && tid == 0 && part == 0) out_lse[row] = (T)(dm + ::log(total));

  // ---- first[j] = min{ k : (u + k) / K >= c[j] } ---------------------------------------------------
  const double dK = (double)K;
  const double inv_K = 1.0 / dK;
  int first[C];
#pragma unroll
  for (int i = 0; i < C; ++i) {
    if (j0 + i < K) {
      // (an entry behind which only negligible weights lie may come out one ulp ABOVE the last one, 1.0, under a
      //  tree-ordered scan; the reference's sequential cumsum is monotone and its c / max(c) never exceeds 1: clamped, as in
      //  the lean form, so that first[] — the children ranges — stays non-decreasing on knife-edge rows too)
      const double c = fmin(divide_with_reciprocal(base + s[i], total, inv_total), 1.0);
      // c <= (u + k) / K  <=>  k >= c K - u, up to the rounding of the position and of this
      // product: both are below K * 2^-52, so unless c K - u sits within K * 1e-15 of an integer
      // its ceiling IS the answer; only that rare case is settled against the exact positions.
      const double x = __builtin_fma(c, dK, -ub);
      const double t = __builtin_ceil(x);
      int k0 = t < 0.0 ? 0 : (t > dK ? K : (int)t);
      if (__builtin_fabs(x - __builtin_rint(x)) <= dK * 1e-15) {
        while (k0 > 0 && divide_with_reciprocal(ub + (double)(k0 - 1), dK, inv_K) >= c) --k0;
        while (k0 < K && divide_with_reciprocal(ub + (double)k0, dK, inv_K) < c) ++k0;
      }
      first[i] = k0;
    } else {
      first[i] = K;
    }
  }
  if (lane == 0) first_of_lane[wave] = first[0];            // the next wavefront's first entry, via LDS
  int wave_min = K;
  if (out_child_end != nullptr) {                           // (each wavefront's smallest first entry: see the clamp below)
    wave_min = wave_suffix_min(first[0], lane);             // min over lanes >= this one, this wavefront
    if (lane == 0) first_of_lane[16 + wave] = wave_min;
  }
  if constexpr (C % 4 == 0) {
#pragma unroll
    for (int q = 0; q < C / 4; ++q) reinterpret_cast<int4 *>(marker + j0)[q] = make_int4(0, 0, 0, 0);
  } else {
#pragma unroll
    for (int i = 0; i < C; ++i) marker[j0 + i] = 0;
  }
  __syncthreads();
  int next_lane_first = __shfl_down(first[0], 1, kWave);     // the next lane's first entry, in-register
  if (lane == kWave - 1) next_lane_first = (wave + 1 < nwaves) ? first_of_lane[wave + 1] : K;
  if (out_child_end != nullptr) {
    // The children ranges must be monotone.  Within a lane first[] is (the lane's running sum is sequential); ACROSS lanes
    // the CDF is assembled from tree-ordered partial sums, and where the weights in between underflow to exact zeros two
    // lanes hold the same sum associated differently — one ulp apart in either order — so on a knife-edge position a
    // later lane's first[] can come out one BELOW an earlier lane's.  The indices the markers produce are then the
    // running maximum's, i.e. those of the SUFFIX MINIMUM of first[]; the ranges are made to say the same: every entry
    // is clamped to the smallest first[] of all later lanes (a reverse scan over each lane's first entry: six bpermutes
    // inside the wavefront, the later wavefronts' minima through LDS).
    int bound = __shfl_down(wave_min, 1, kWave);                       // min over the lanes BEHIND this one
    if (lane == kWave - 1) bound = K;
    for (int w = wave + 1; w < nwaves; ++w) bound = min(bound, first_of_lane[16 + w]);
#pragma unroll
    for (int i = 0; i < C; ++i) first[i] = min(first[i], bound);
    next_lane_first = min(next_lane_first, bound);
  }
  // By-product for the gather's backward (after the clamp above): first[j] = how many positions precede the CDF at j = where the children
  // of particles 0..j end, so the children of particle j are the positions [first[j-1], first[j]) — one run, because
  // the indices are non-decreasing.  (aesmc_affine_step_backward_resampled sums a particle's children with it.)
  if (out_child_end != nullptr && owns_idx) {
    int32_t *ends = out_child_end + row * (int64_t)K + j0;
    // a lane's C entries are consecutive: whole 16-byte stores where the row allows (K a multiple of 4 keeps every
    // lane's first entry on a 16-byte boundary), else entry by entry
    if (C % 4 == 0 && (K & 3) == 0 && j0 + C <= K && (reinterpret_cast<uintptr_t>(out_child_end) & 15u) == 0) {
#pragma unroll
      for (int q = 0; q < C / 4; ++q)
        reinterpret_cast<int4 *>(ends)[q] = make_int4(first[4 * q], first[4 * q + 1], first[4 * q + 2], first[4 * q + 3]);
    } else {
#pragma unroll
      for (int i = 0; i < C; ++i)
        if (j0 + i < K) ends[i] = first[i];
    }
  }
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const int j = j0 + i;
    if (j < K) {
      int next = (i + 1 < C) ? first[i + 1 < C ? i + 1 : i] : next_lane_first;
      if (j == K - 1) next = K;
      if (first[i] < next) marker[first[i]] = j + 1;       // distinct j write distinct slots
    }
  }
  __syncthreads();

  // ---- idx[k] = running maximum of the markers ------------------------------------------------
  int best[C];
  if constexpr (C % 4 == 0) {
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
      const int4 packed = reinterpret_cast<const int4 *>(marker + j0)[q];
      best[4 * q] = packed.x;
      best[4 * q + 1] = packed.y;
      best[4 * q + 2] = packed.z;
      best[4 * q + 3] = packed.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < C; ++i) best[i] = marker[j0 + i];
  }
  int acc = 0;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    acc = max(acc, best[i]);
    best[i] = acc;
  }
  int incl_max = acc;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int y = __shfl_up(incl_max, off, kWave);
    if (lane >= off) incl_max = max(incl_max, y);
  }
  int before = __shfl_up(incl_max, 1, kWave);
  if (lane == 0) before = 0;
  if (lane == kWave - 1) scratch_i[wave] = incl_max;
  __syncthreads();
  for (int w = 0; w < wave; ++w) before = max(before, scratch_i[w]);
#pragma unroll
  for (int i = 0; i < C; ++i) best[i] = max(before, best[i]);
  if (!owns_idx) {
    // another workgroup of this row stores these indices
  } else if (j0 + C <= K && (((uintptr_t)(idx + j0)) & 15u) == 0) {
#pragma unroll
    for (int i = 0; i < C; i += 2) {
      longlong2 pair;
      pair.x = (int64_t)best[i];
      pair.y = (int64_t)best[i + 1];
      *reinterpret_cast<longlong2 *>(idx + j0 + i) = pair;
    }
  } else {
#pragma unroll
    for (int i = 0; i < C; ++i)
      if (j0 + i < K) idx[j0 + i] = (int64_t)best[i];
  }
  if (!PAYLOAD || payload.src == nullptr) return;

  // ---- fused step: the payload rows follow their ancestors ------------------------------------
  // Every lane read its marker slots before the last barrier, so they can now hold the indices.
  if constexpr (C % 4 == 0) {
#pragma unroll
    for (int q = 0; q < C / 4; ++q)
      reinterpret_cast<int4 *>(marker + j0)[q] =
          make_int4(best[4 * q], best[4 * q + 1], best[4 * q + 2], best[4 * q + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < C; ++i) marker[j0 + i] = best[i];
  }
  char *drow = payload.dst + (uint64_t)row * K * payload.ppp * payload.G;
  __syncthreads();
  const char *srow = payload.src + row * payload.stride_b;
  if (payload.G == 16)
    gather_row_from_lds<16, (C <= 4 ? 5 : 4)>(marker, srow, drow, K, payload.ppp, payload.stride_k, tid, nt, part, parts);
  else if (payload.G == 8)
    gather_row_from_lds<8, (C <= 4 ? 5 : 4)>(marker, srow, drow, K, payload.ppp, payload.stride_k, tid, nt, part, parts);
  else
    gather_row_from_lds<4, (C <= 4 ? 5 : 4)>(marker, srow, drow, K, payload.ppp, payload.stride_k, tid, nt, part, parts);
}

// ---- the lean form of the primary kernel: whole rows, no payload ------------------------------------------------------
//
// What a step whose propagation launch fetches the rows itself runs (aesmc_resample_step[_ranges] without a payload) when
// a row is exactly blockDim.x * C particles (K a multiple of 64 C: every BASELINE shape).  The same algorithm as
// ancestor_index_inv_kernel — float64 weights, blocked scan, CDF inversion, int32 max-scan — written for the instruction
// count: that kernel issues ~290 vector instructions per particle (2 336 per lane at C = 8; 490 of them float64
// arithmetic), which at four workgroups per CU makes the launch issue-bound (20 us of its 25 at B = 1024, K = 4096).  Here:
//   * wavefront scans and reductions run on the DPP crossbar (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31: two moves and an
//     add per step for a float64) instead of ds_bpermute + compare + select per step;
//   * every "j < K" predicate is gone (whole rows), the row maximum and the prefixes of the earlier wavefronts are
//     wavefront-uniform scalars (v_readlane of a 16-lane DPP scan over the wavefronts' totals: no loops over LDS);
//   * the rare comparison that sits within rounding of flipping is settled out of line (one uniform branch per wavefront
//     instead of two data-dependent loops per particle);
//   * markers are stored without exchanging neighbours: a lane knows where its own next particle starts; the one marker
//     per wavefront that depends on the NEXT wavefront's first particle is an LDS atomic max (a slot's final value is the
//     largest particle that starts there whichever order the stores land in) — one barrier fewer, four in all.
// The scan associates differently from ancestor_index_inv_kernel's (a tree over rows of 16 lanes instead of
// Hillis-Steele over 64), so CDF entries may differ in the last place of a float64: indices agree except where a
// comparison sits within ~1e-16 of flipping (tests/test_gpu_kernels.py holds both to the same oracle and fixtures).
template <int CTRL, int ROW = 0xf> __device__ __forceinline__ int dpp_i32(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW, 0xf, false);
}
template <int CTRL, int ROW = 0xf> __device__ __forceinline__ float dpp_f32(float old, float src) {
  return __builtin_bit_cast(float, dpp_i32<CTRL, ROW>(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src)));
}
template <int CTRL, int ROW = 0xf> __device__ __forceinline__ double dpp_f64(double old, double src) {
  const long long o = __builtin_bit_cast(long long, old), v = __builtin_bit_cast(long long, src);
  const int lo = dpp_i32<CTRL, ROW>((int)o, (int)v), hi = dpp_i32<CTRL, ROW>((int)(o >> 32), (int)(v >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}
constexpr int kDppShr1 = 0x111, kDppShr2 = 0x112, kDppShr4 = 0x114, kDppShr8 = 0x118, kDppBcast15 = 0x142,
              kDppBcast31 = 0x143, kDppWaveShr1 = 0x138, kDppWaveShl1 = 0x130;

// inclusive sum over the lanes of a row of 16 (what the first four steps of a wavefront scan leave)
__device__ __forceinline__ double row16_scan_add(double x) {
  x += dpp_f64<kDppShr1>(0.0, x);
  x += dpp_f64<kDppShr2>(0.0, x);
  x += dpp_f64<kDppShr4>(0.0, x);
  x += dpp_f64<kDppShr8>(0.0, x);
  return x;
}
__device__ __forceinline__ double wave_scan_add(double x) {
  x = row16_scan_add(x);
  x += dpp_f64<kDppBcast15, 0xa>(0.0, x);
  x += dpp_f64<kDppBcast31, 0xc>(0.0, x);
  return x;
}
__device__ __forceinline__ int row16_scan_max(int x) {      // values >= 0
  x = max(x, dpp_i32<kDppShr1>(0, x));
  x = max(x, dpp_i32<kDppShr2>(0, x));
  x = max(x, dpp_i32<kDppShr4>(0, x));
  x = max(x, dpp_i32<kDppShr8>(0, x));
  return x;
}
__device__ __forceinline__ int wave_scan_max(int x) {
  x = row16_scan_max(x);
  x = max(x, dpp_i32<kDppBcast15, 0xa>(0, x));
  x = max(x, dpp_i32<kDppBcast31, 0xc>(0, x));
  return x;
}
__device__ __forceinline__ float lanes_max(float x, bool whole_wave) {      // lane 15 (row of 16) / lane 63 holds the maximum
  const float ninf = -__builtin_huge_valf();
  x = fmaxf(x, dpp_f32<kDppShr1>(ninf, x));
  x = fmaxf(x, dpp_f32<kDppShr2>(ninf, x));
  x = fmaxf(x, dpp_f32<kDppShr4>(ninf, x));
  x = fmaxf(x, dpp_f32<kDppShr8>(ninf, x));
  if (whole_wave) {
    x = fmaxf(x, dpp_f32<kDppBcast15, 0xa>(ninf, x));
    x = fmaxf(x, dpp_f32<kDppBcast31, 0xc>(ninf, x));
  }
  return x;
}
__device__ __forceinline__ double lanes_max(double x, bool whole_wave) {
  const double ninf = -__builtin_huge_val();
  x = fmax(x, dpp_f64<kDppShr1>(ninf, x));
  x = fmax(x, dpp_f64<kDppShr2>(ninf, x));
  x = fmax(x, dpp_f64<kDppShr4>(ninf, x));
  x = fmax(x, dpp_f64<kDppShr8>(ninf, x));
  if (whole_wave) {
    x = fmax(x, dpp_f64<kDppBcast15, 0xa>(ninf, x));
    x = fmax(x, dpp_f64<kDppBcast31, 0xc>(ninf, x));
  }
  return x;
}
__device__ __forceinline__ float read_lane(float x, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}
__device__ __forceinline__ double read_lane(double x, int lane) {
  const long long v = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_readlane((int)v, lane), hi = __builtin_amdgcn_readlane((int)(v >> 32), lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}

// exp_nonpositive's constants held in SCALAR registers for the whole kernel: a float64 literal cannot be an operand, and
// left to itself the compiler rebuilds each one in a vector register pair in front of the multiply-add that uses it (two
// v_mov per Horner step: 26 per particle).  v_fma_f64 takes one scalar pair as an operand.
struct ExpScalars {
  double log2e, ln2_hi, ln2_lo, c[13];
  __device__ __forceinline__ void init() {
    const double values[16] = {1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
                               1.6059043836821613e-10, 2.08767569878681e-09, 2.505210838544172e-08,
                               2.755731922398589e-07, 2.7557319223985893e-06, 2.48015873015873e-05,
                               1.984126984126984e-04, 1.3888888888888889e-03, 8.333333333333333e-03,
                               4.1666666666666664e-02, 1.6666666666666666e-01, 0.5, 1.0};
    log2e = values[0]; ln2_hi = values[1]; ln2_lo = values[2];
#pragma unroll
    for (int i = 0; i < 13; ++i) c[i] = values[3 + i];
    asm volatile("" : "+s"(log2e), "+s"(ln2_hi), "+s"(ln2_lo));
#pragma unroll
    for (int i = 0; i < 13; ++i) asm volatile("" : "+s"(c[i]));
  }
};
// exp(x) for x <= 0, not NaN: exp_nonpositive's arithmetic, operation for operation, on x clamped at -746 instead of a
// test for x <= -745.2 — the same values, bit for bit: above -745.2 nothing changes, and from there down to the clamp the
// polynomial's p <= exp(-0.0668) = 0.9354 is scaled by 2^-1075 or 2^-1076, below half the smallest denormal, which
// v_ldexp_f64 rounds to the 0.0 that function answers (a zero weight must stay an exact zero: with u = 0 a CDF entry
// of 0 is reached by the first position, a denormal one is not) — without the compare, the two selects and the branch
// around the polynomial.
__device__ __forceinline__ double exp_nonpositive_s(double x, const ExpScalars &k) {
  x = fmax(x, -746.0);
  const double n = __builtin_rint(x * k.log2e);
  double r = __builtin_fma(-n, k.ln2_hi, x);
  r = __builtin_fma(-n, k.ln2_lo, r);
  double p = k.c[0];
#pragma unroll
  for (int i = 1; i < 13; ++i) p = __builtin_fma(p, r, k.c[i]);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)n);
}

// (Held to 64 registers like the kernel above: every row of the north-star batch resident at once.  Measured against a
//  96-register build without the slow path's spills, which runs the batch in two rounds of workgroups: 20.2 against 19.9 us
//  at B = 1024, K = 4096 and 4.0 against 5.7 us at B = 256, K = 1024 — profiles/r05_k2_forms.txt.  Also measured and not
//  kept: a workgroup walking two to four rows so that one row's index stores drain under the next row's scan — at 64
//  registers the loop's live state spilt 72 of them (31 - 33 us); as a kernel of its own at 107 registers, two workgroups
//  per CU: 21.9 / 26.4 / 25.5 us for 2 / 3 / 4 rows per workgroup against 20.2 at B = 1024, 39.8 against 36.7 at B = 2048 —
//  where two rounds of the plain kernel already overlap by themselves (0.46 of HBM): profiles/r05_k2_row_walk.txt.)
template <typename T, int C>
__global__ __launch_bounds__(kMaxThreads, C > 8 ? 1 : 8) void ancestor_index_rows_kernel(
    const T *__restrict__ log_w, const double *__restrict__ u, int64_t *__restrict__ out_idx, int32_t *flags, int K,
    T *__restrict__ out_lse, int32_t *__restrict__ out_child_end) {
  static_assert(C % 4 == 0, "a lane's particles are moved in 16-byte pieces");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *scratch = smem;                                          // [0,16) wavefront maxima, [16,32) wavefront totals, [32], [33]
  int *scratch_i = reinterpret_cast<int *>(smem + 40);             // [0,16) NaN seen, [16,32) wavefront marker maxima
  int *marker = reinterpret_cast<int *>(smem + kScratchDoubles);   // [K] + a spare slot for markers nobody needs
  const int tid = threadIdx.x;
  const int nt = blockDim.x;                                       // nt * C == K
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = nt >> 6;
  const int64_t row = blockIdx.x;
  const T *lw = log_w + row * (int64_t)K;
  int64_t *idx = out_idx + row * (int64_t)K;
  const int j0 = tid * C;
  const double ub = u[row];      // (needed after the scan; sent for now)

  // ---- load once; the markers' zero fill rides in front of the first barrier ---------------------------------------
  T v[C];
  constexpr int NV = Vec16<T>::N;
  {
    using V = typename Vec16<T>::type;
#pragma unroll
    for (int q = 0; q < C / NV; ++q) {
      const V packed = reinterpret_cast<const V *>(lw + j0)[q];
#pragma unroll
      for (int r = 0; r < NV; ++r) v[q * NV + r] = Vec16<T>::get(packed, r);
    }
  }
#pragma unroll
  for (int q = 0; q < C / 4; ++q) reinterpret_cast<int4 *>(marker + j0)[q] = make_int4(0, 0, 0, 0);
  T m = v[0];
  bool nan_here = v[0] != v[0];
#pragma unroll
  for (int i = 1; i < C; ++i) {
    nan_here |= v[i] != v[i];
    m = Num<T>::max(m, v[i]);
  }
  {
    const T wave_m = read_lane(lanes_max(m, true), kWave - 1);
    const int wave_nan = __any(nan_here) ? 1 : 0;
    if (lane == 0) {
      scratch[wave] = (double)wave_m;
      scratch_i[wave] = wave_nan;
    }
    if (tid == 0) *reinterpret_cast<int *>(scratch + 34) = 0;      // (the row's "ranges descend somewhere" flag, see below)
  }
  __syncthreads();
  const int slot = lane < nwaves ? lane : 0;
  const double dm = read_lane(lanes_max(scratch[slot], false), 15);      // (nwaves <= 16: one row of lanes)
  const bool has_nan = __any(scratch_i[slot] != 0);
  const bool degenerate = has_nan || !(dm > -__builtin_huge_val() && dm < __builtin_huge_val());
  if (degenerate) {  // same conventions as the reference: see include/aesmc_hip.h, K2
    if (tid == 0) {
      raise_flag(flags, has_nan ? AESMC_FLAG_NAN_LOG_WEIGHT : AESMC_FLAG_DEGENERATE_ROW);
      if (out_lse != nullptr) out_lse[row] = has_nan ? Num<T>::nan() : (T)dm;      // torch.logsumexp's values for such rows
    }
    for (int i = 0; i < C; ++i) {
      idx[j0 + i] = (int64_t)K;
      if (out_child_end != nullptr) out_child_end[row * (int64_t)K + j0 + i] = 0;     // nobody has children
    }
    return;
  }

  // ---- float64 weights, blocked inclusive scan -----------------------------------------------------------------------
  double s[C];
  double run = 0.0;
  ExpScalars ek;
  ek.init();
#pragma unroll
  for (int i = 0; i < C; ++i) {
    run += exp_nonpositive_s((double)v[i] - dm, ek);
    s[i] = run;
  }
  const double incl = wave_scan_add(run);
  const double excl = dpp_f64<kDppWaveShr1>(0.0, incl);           // the lanes before this one, inside the wavefront
  if (lane == kWave - 1) scratch[16 + wave] = incl;
  // The CDF's last entry is the normaliser, so that c[K-1] == 1.0 exactly (reference: c / max(c)): its owner — the last
  // lane — publishes the two terms only it has, and every lane forms  (excl + earlier wavefronts) + s  the way it does.
  if (tid == nt - 1) {
    scratch[32] = excl;
    scratch[33] = run;
  }
  __syncthreads();
  double before_waves, before_last;
  {
    const double totals = row16_scan_add(lane < nwaves ? scratch[16 + lane] : 0.0);      // inclusive, wavefronts 0 .. lane
    before_waves = read_lane(totals, wave > 0 ? wave - 1 : 0);
    before_last = read_lane(totals, nwaves > 1 ? nwaves - 2 : 0);
    if (wave == 0) before_waves = 0.0;
    if (nwaves == 1) before_last = 0.0;
  }
  const double base = excl + before_waves;
  const double total = (scratch[32] + before_last) + scratch[33];
  const double inv_total = 1.0 / total;
  // by-product: logsumexp of the row (the step's contribution to log Z), float64 inside
  if (out_lse != nullptr && tid == 0) out_lse[row] = (T)(dm + ::log(total));

  // ---- first[j] = min{ k : (u + k) / K >= c[j] } ---------------------------------------------------------------------
  const double dK = (double)K;
  int first[C];
  bool edge = false;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    // (a tree-ordered scan may leave an entry one ulp ABOVE the last one where the weights behind it are negligible; the
    //  reference's sequential cumsum is monotone, its c / max(c) never exceeds 1: clamped, so first[] never exceeds
    //  first[K-1] and stays non-decreasing where it matters)
    const double c = fmin(divide_with_reciprocal(base + s[i], total, inv_total), 1.0);
    // c <= (u + k) / K  <=>  k >= c K - u, up to the rounding of the position and of this product: both are below
    // K * 2^-52, so unless c K - u sits within K * 1e-15 of an integer its ceiling IS the answer
    const double x = __builtin_fma(c, dK, -ub);
    const double t = __builtin_ceil(x);
    first[i] = t < 0.0 ? 0 : (t > dK ? K : (int)t);
    edge |= __builtin_fabs(x - __builtin_rint(x)) <= dK * 1e-15;
  }
  if (__any(edge)) {      // rare: settled against the exactly rounded positions, out of line
    const double inv_K = 1.0 / dK;
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const double c = fmin(divide_with_reciprocal(base + s[i], total, inv_total), 1.0);
      const double x = __builtin_fma(c, dK, -ub);
      if (__builtin_fabs(x - __builtin_rint(x)) <= dK * 1e-15) {
        int k0 = first[i];
        while (k0 > 0 && divide_with_reciprocal(ub + (double)(k0 - 1), dK, inv_K) >= c) --k0;
        while (k0 < K && divide_with_reciprocal(ub + (double)k0, dK, inv_K) < c) ++k0;
        first[i] = k0;
      }
    }
  }
  // By-product for the gather's backward: first[j] = where the children of particles 0..j end (see the kernel above).
  // The ranges must be monotone; first[] is, except where two lanes hold the same partial sum associated differently (the
  // general kernel says how: exact-zero weights in between, a knife-edge position) — then the consistent ranges are those
  // of first[]'s SUFFIX MINIMUM.  Here: written as they are; whether any NEIGHBOURING pair of lanes (or of wavefronts)
  // descends is one comparison and a vote — a row-wide flag in LDS — and only a flagged row (about one in 1e9 with random
  // uniforms) repairs its ranges, out of line, behind the last barrier.
  int *inverted = reinterpret_cast<int *>(scratch + 34);      // (a free slot of the scratch area; cleared in front of the first barrier)
  int range_last = 0;
  const int next_lane_first = dpp_i32<kDppWaveShl1>(K, first[0]);      // the next lane's first particle (lane 63: unknown)
  if (out_child_end != nullptr) {
    if (lane == 0) scratch_i[32 + wave] = first[0];
    range_last = first[C - 1];
    if (__any(lane != kWave - 1 && range_last > next_lane_first) && lane == 0) *inverted = 1;
    int32_t *ends = out_child_end + row * (int64_t)K + j0;
#pragma unroll
    for (int q = 0; q < C / 4; ++q)
      reinterpret_cast<int4 *>(ends)[q] = make_int4(first[4 * q], first[4 * q + 1], first[4 * q + 2], first[4 * q + 3]);
  }
  // ---- markers: slot first[j] holds j + 1 for the LAST particle that starts there ---------------------------------------
  {
#pragma unroll
    for (int i = 0; i < C - 1; ++i) marker[first[i] < first[i + 1] ? first[i] : K] = j0 + i + 1;      // (slot K: the spare)
    const int last = first[C - 1];
    if (lane != kWave - 1) {
      marker[last < next_lane_first ? last : K] = j0 + C;
    } else {
      // the next wavefront's first particle is not known here: the largest particle that starts at a slot wins
      // whichever order the stores land in (every other store to it is a plain store of a larger value, or another max)
      atomicMax(marker + (last < K ? last : K), j0 + C);
    }
  }
  __syncthreads();
  if (out_child_end != nullptr && lane == kWave - 1 && wave + 1 < nwaves && range_last > scratch_i[32 + wave + 1])
    *inverted = 1;      // (a wavefront's last particle against the next wavefront's first; seen by all behind the next barrier)

  // ---- idx[k] = running maximum of the markers ----------------------------------------------------------------------
  int best[C];
#pragma unroll
  for (int q = 0; q < C / 4; ++q) {
    const int4 packed = reinterpret_cast<const int4 *>(marker + j0)[q];
    best[4 * q] = packed.x;
    best[4 * q + 1] = packed.y;
    best[4 * q + 2] = packed.z;
    best[4 * q + 3] = packed.w;
  }
#pragma unroll
  for (int i = 1; i < C; ++i) best[i] = max(best[i], best[i - 1]);
  const int incl_max = wave_scan_max(best[C - 1]);
  const int before_lanes = dpp_i32<kDppWaveShr1>(0, incl_max);
  if (lane == kWave - 1) scratch_i[16 + wave] = incl_max;
  __syncthreads();
  int before;
  {
    const int maxima = row16_scan_max(lane < nwaves ? scratch_i[16 + lane] : 0);
    before = __builtin_amdgcn_readlane(maxima, wave > 0 ? wave - 1 : 0);
    if (wave == 0) before = 0;
  }
  if (out_child_end != nullptr && *inverted != 0) {      // (uniform: read behind the barrier above)
    // the rare row: every entry clamped to the smallest first[] behind it — the later lanes of the wavefront by a reverse
    // scan over each lane's first entry, the later wavefronts' minima through LDS; the entries are read back from where
    // this lane stored them
    int32_t *ends = out_child_end + row * (int64_t)K + j0;
    const int wave_min = wave_suffix_min(ends[0], lane);
    int bound = __shfl_down(wave_min, 1, kWave);
    if (lane == kWave - 1) bound = K;
    __syncthreads();      // (scratch_i[32 ..) was last read in front of the barrier above)
    if (lane == 0) scratch_i[32 + wave] = wave_min;
    __syncthreads();
    for (int w = wave + 1; w < nwaves; ++w) bound = min(bound, scratch_i[32 + w]);
    for (int i = 0; i < C; ++i)
      if (ends[i] > bound) ends[i] = bound;
  }
  before = max(before, before_lanes);
#pragma unroll
  for (int i = 0; i < C; i += 2) {
    longlong2 pair;
    pair.x = (int64_t)max(before, best[i]);
    pair.y = (int64_t)max(before, best[i + 1]);
    *reinterpret_cast<longlong2 *>(idx + j0 + i) = pair;
  }
}

// which kernel a payload-free step launches: 0 by shape, 1 ancestor_index_inv_kernel always, 2 the lean form wherever it
// applies (AESMC_K2_FORM=general / rows in the environment, or the test hook aesmc_test_set_k2_form)
static int g_k2_form = [] {
  const char *v = measurement_knob("AESMC_K2_FORM");
  return v == nullptr ? 0 : (v[0] == 'g' ? 1 : (v[0] == 'r' ? 2 : 0));
}();
static int g_k2_last_form = 0;

static int pick_threads(int64_t K, int chunk) {
  int64_t nt = (K + chunk - 1) / chunk;  // one round when it fits
  nt = (nt + kWave - 1) / kWave * kWave;
  if (nt < kWave) nt = kWave;
  if (nt > kMaxThreads) nt = kMaxThreads;
  return (int)nt;
}

// Workgroups per batch row of the fused step.  One workgroup per row leaves half the CUs idle below
// 256 rows; sharing a row's OUTPUT between two workgroups (each repeating the scan) fills them.
// Measured (tools/stepbench.py, N(0,1) weights): B=128 K=4096 d=10: 16.6 us with 1, 15.2 with 2, 16.6
// with 4, 22.8 with 8 workgroups per row; B=256 K=1024: 9.2 / 8.95 / 10.0 / 10.1; B=1024 K=4096:
// 81.6 / 84.9 / 98.4 / 141.8 — the repeated scan (float64 exp + two scans per particle) costs more
// than the shorter copy returns beyond two.  0 = automatic; AESMC_STEP_PARTS in the environment (or the test
// hook aesmc_test_set_step_parts, which is not part of the C ABI of include/aesmc_hip.h) pins a value.
static int g_step_parts = [] {
  const char *v = measurement_knob("AESMC_STEP_PARTS");
  const int parts = v != nullptr ? atoi(v) : 0;
  return (parts > 0 && (parts & (parts - 1)) == 0) ? parts : 0;
}();

static int pick_parts(int64_t B, int nt, bool has_payload) {
  int limit = nt / kWave;                        // at least one wavefront of index stores per part
  if (limit > 8) limit = 8;
  if (g_step_parts > 0) return g_step_parts < limit ? g_step_parts : limit;
  if (!has_payload) return 1;
  return (B <= 256 && limit >= 2) ? 2 : 1;
}

template <typename T, int C>
static int launch_inv(const void *log_w, const double *u, int64_t *idx, int32_t *flags, int64_t B,
                      int64_t K, hipStream_t s, void *out_lse = nullptr,
                      const StepPayload &payload = StepPayload{nullptr, nullptr, 0, 0, 0, 0},
                      int32_t *child_end = nullptr) {
  if constexpr (C % 4 == 0) {
    // whole rows and no payload: the lean form (ancestor_index_rows_kernel)
    const int64_t lanes = K / C;
    const auto aligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (g_k2_form != 1 && payload.src == nullptr && lanes * C == K && lanes % kWave == 0 && lanes <= kMaxThreads &&
        aligned(log_w) && aligned(idx) && aligned(child_end)) {
      const size_t lds_rows = (size_t)kScratchDoubles * sizeof(double) + (size_t)(K + 4) * sizeof(int);
      static bool rows_attr_set[64] = {};
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return AESMC_ERR_LAUNCH;
      if (!rows_attr_set[dev]) {
        if (hipFuncSetAttribute((const void *)ancestor_index_rows_kernel<T, C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
          return AESMC_ERR_LAUNCH;
        rows_attr_set[dev] = true;
      }
      g_k2_last_form = 2;
      hipLaunchKernelGGL((ancestor_index_rows_kernel<T, C>), dim3((unsigned)B), dim3((unsigned)lanes), lds_rows, s,
                         (const T *)log_w, u, idx, flags, (int)K, (T *)out_lse, child_end);
      return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
    }
  }
  g_k2_last_form = 1;
  const int nt = pick_threads(K, C);
  const size_t lds = (size_t)kScratchDoubles * sizeof(double) + (size_t)(nt * C + nt + 8) * sizeof(int);
  // raise the dynamic-LDS cap once per device and instantiation (a process may drive several GPUs)
  static bool attr_set[64] = {};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) return AESMC_ERR_LAUNCH;
  if (!attr_set[device]) {
    if (hipFuncSetAttribute((const void *)ancestor_index_inv_kernel<T, C, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void *)ancestor_index_inv_kernel<T, C, false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return AESMC_ERR_LAUNCH;
    attr_set[device] = true;
  }
  int parts = pick_parts(B, nt, payload.src != nullptr);
  while (parts > 1 && (nt % parts != 0 || B * parts > 0x7fffffffLL)) parts /= 2;
  if (payload.src != nullptr)
    hipLaunchKernelGGL((ancestor_index_inv_kernel<T, C, true>), dim3((unsigned)(B * parts)), dim3(nt), lds, s,
                       (const T *)log_w, u, idx, flags, (int)K, (T *)out_lse, payload, (int)B, parts, child_end);
  else
    hipLaunchKernelGGL((ancestor_index_inv_kernel<T, C, false>), dim3((unsigned)(B * parts)), dim3(nt), lds, s,
                       (const T *)log_w, u, idx, flags, (int)K, (T *)out_lse, payload, (int)B, parts, child_end);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

template <typename T>
static int launch_step(const void *log_w, const double *u, int64_t *idx, void *out_lse, int32_t *flags,
                       int64_t B, int64_t K, const StepPayload &payload, hipStream_t s, int32_t *child_end = nullptr) {
  if (K <= 512) return launch_inv<T, 2>(log_w, u, idx, flags, B, K, s, out_lse, payload, child_end);
  if (K <= 2048) return launch_inv<T, 4>(log_w, u, idx, flags, B, K, s, out_lse, payload, child_end);
  // (rows of up to 4096 particles without a payload: four per lane on up to 1024 lanes rather than eight on 512 — half the
  //  serial work per lane in a kernel bound by its phases' latency: 18.5 against 20.1 us at B = 1024, 6.2 / 8.1 / 11.7 against
  //  6.6 / 8.8 / 13.0 at B = 128 / 256 / 512, profiles/r05_k2_forms_final.txt)
  if (K <= 4096 && K % 256 == 0 && payload.src == nullptr)      // (whole wavefronts of four: the lean form's shapes)
    return launch_inv<T, 4>(log_w, u, idx, flags, B, K, s, out_lse, payload, child_end);
  if (K <= 8192) return launch_inv<T, 8>(log_w, u, idx, flags, B, K, s, out_lse, payload, child_end);
  if (K <= 16384) return launch_inv<T, 16>(log_w, u, idx, flags, B, K, s, out_lse, payload, child_end);
  if (K <= kInvMaxParticles) return launch_inv<T, 32>(log_w, u, idx, flags, B, K, s, out_lse, payload, child_end);
  return AESMC_ERR_UNSUPPORTED;
}

// Particles per lane grow with the row so that one workgroup (<= 1024 lanes) covers it; beyond
// 32768 particles the CDF no longer fits registers + LDS and goes through the caller's workspace.
template <typename T>
static int launch(const void *log_w, const double *u, int64_t *idx, int32_t *flags, int64_t B,
                  int64_t K, void *ws, size_t ws_bytes, hipStream_t s) {
  if (K <= 512) return launch_inv<T, 2>(log_w, u, idx, flags, B, K, s);
  if (K <= 2048) return launch_inv<T, 4>(log_w, u, idx, flags, B, K, s);
  if (K <= 8192) return launch_inv<T, 8>(log_w, u, idx, flags, B, K, s);
  if (K <= 16384) return launch_inv<T, 16>(log_w, u, idx, flags, B, K, s);
  if (K <= kInvMaxParticles) return launch_inv<T, 32>(log_w, u, idx, flags, B, K, s);
  if (ws == nullptr || ws_bytes < aesmc_workspace_bytes(B, K)) return AESMC_ERR_WORKSPACE;
  const size_t lds = (size_t)kScratchDoubles * sizeof(double);
  hipLaunchKernelGGL((ancestor_index_kernel<T, 8>), dim3((unsigned)B), dim3(kMaxThreads), lds, s,
                     (const T *)log_w, u, idx, flags, (int)K, (double *)ws);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

}  // namespace aesmc

extern "C" int64_t aesmc_ancestor_index_lds_max_particles(void) { return aesmc::kInvMaxParticles; }

// Test hooks (not part of the C ABI of include/aesmc_hip.h): which kernel a payload-free resampling step launches
// (0 by shape, 1 ancestor_index_inv_kernel, 2 ancestor_index_rows_kernel wherever it applies), and which one the last did.
extern "C" int aesmc_test_set_k2_form(int form) {
  if (form < 0 || form > 2) return AESMC_ERR_INVALID_ARGUMENT;
  aesmc::g_k2_form = form;
  return AESMC_OK;
}
extern "C" int aesmc_test_last_k2_form(void) { return aesmc::g_k2_last_form; }

extern "C" int aesmc_test_set_step_parts(int parts) {
  if (parts < 0 || (parts & (parts - 1)) != 0) return AESMC_ERR_INVALID_ARGUMENT;
  aesmc::g_step_parts = parts;
  return AESMC_OK;
}

extern "C" size_t aesmc_workspace_bytes(int64_t B, int64_t K) {
  if (B <= 0 || K <= aesmc::kInvMaxParticles) return 0;
  return (size_t)B * (size_t)aesmc::cdf_row_slots(K) * sizeof(double);
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

static inline int64_t step_pow2_divisor(uint64_t x, int64_t cap) {
  while (cap > 1 && (x % (uint64_t)cap) != 0) cap >>= 1;
  return cap;
}

extern "C" int aesmc_resample_step(int dtype, const void *log_w, const double *u, int64_t *out_idx,
                                   void *out_lse, const void *src, void *dst, int32_t *flags, int64_t B,
                                   int64_t K, int64_t row_bytes, int64_t src_stride_b,
                                   int64_t src_stride_k, void *stream) {
  using namespace aesmc;
  if (log_w == nullptr || u == nullptr || out_idx == nullptr || B < 0 || K < 0 || row_bytes < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if ((src == nullptr) != (dst == nullptr)) return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0) return AESMC_OK;
  if (K > kInvMaxParticles || B > 0x7fffffffLL) return AESMC_ERR_UNSUPPORTED;
  StepPayload payload{nullptr, nullptr, 0, 0, 0, 0};
  if (src != nullptr && row_bytes > 0) {
    if ((uint64_t)K * (uint64_t)row_bytes >= (1ull << 32)) return AESMC_ERR_UNSUPPORTED;
    int64_t G = step_pow2_divisor((uint64_t)row_bytes, 16);
    G = step_pow2_divisor((uint64_t)(uintptr_t)src, G);
    G = step_pow2_divisor((uint64_t)(src_stride_b < 0 ? -src_stride_b : src_stride_b), G);
    G = step_pow2_divisor((uint64_t)(src_stride_k < 0 ? -src_stride_k : src_stride_k), G);
    // 16-byte stores: aligned destination and batch-row pitch; pieces of at least 4 bytes
    if (G < 4 || ((uintptr_t)dst & 15u) != 0 || ((uint64_t)K * (uint64_t)row_bytes) % 16 != 0)
      return AESMC_ERR_UNSUPPORTED;
    payload = StepPayload{(const char *)src, (char *)dst, src_stride_b, src_stride_k,
                          (uint32_t)(row_bytes / G), (int)G};
  }
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32) return launch_step<float>(log_w, u, out_idx, out_lse, flags, B, K, payload, s);
  if (dtype == AESMC_F64) return launch_step<double>(log_w, u, out_idx, out_lse, flags, B, K, payload, s);
  return AESMC_ERR_INVALID_ARGUMENT;
}

extern "C" int aesmc_resample_step_ranges(int dtype, const void *log_w, const double *u, int64_t *out_idx,
                                          void *out_lse, int32_t *out_child_end, int32_t *flags, int64_t B,
                                          int64_t K, void *stream) {
  using namespace aesmc;
  if (log_w == nullptr || u == nullptr || out_idx == nullptr || out_child_end == nullptr || B < 0 || K < 0 ||
      (((uintptr_t)out_child_end) & 3u) != 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0) return AESMC_OK;
  if (K > kInvMaxParticles || B > 0x7fffffffLL) return AESMC_ERR_UNSUPPORTED;
  const StepPayload none{nullptr, nullptr, 0, 0, 0, 0};
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32) return launch_step<float>(log_w, u, out_idx, out_lse, flags, B, K, none, s, out_child_end);
  if (dtype == AESMC_F64) return launch_step<double>(log_w, u, out_idx, out_lse, flags, B, K, none, s, out_child_end);
  return AESMC_ERR_INVALID_ARGUMENT;
}
