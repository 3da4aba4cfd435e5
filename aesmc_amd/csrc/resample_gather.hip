// K3: resample gather  dst[b,k,:] = src[b, idx[b,k], :]  and its backward (segmented sum).
//
// Replaces torch.gather at aesmc/state.py:179 (element-granular gather with an int64 index
// expanded to the value's full shape) and its scatter_add autograd.  HBM-bound:
// 8 B index + row_bytes read + row_bytes write per particle.
//
// Forward mapping: a batch row's output is one dense run of K*row_bytes bytes.  It is cut into
// 16-byte chunks, one per lane, so every wavefront store instruction writes 1 KiB contiguously.
// A chunk is assembled from 16/G source pieces of G bytes (G = largest power of two <= 16 dividing
// row_bytes and the source strides), each piece lying inside one particle's row.  Systematic
// resampling returns non-decreasing indices, so neighbouring lanes read neighbouring (often the
// same) source rows: surviving rows come from HBM once, repeats are L1/L2 hits.
#include "common.hpp"

namespace aesmc {

constexpr int kGatherBlock = 256;

// V pieces of G bytes per 16-byte chunk (V == 1 on the unaligned fallback) and U chunks per lane,
// 256 lanes apart so each store instruction of a wavefront still covers 1 KiB contiguously.  The
// U*V index loads are issued together, then the U*V row loads: the kernel is a chain of two
// dependent global loads per piece, and at the small per-step sizes of BASELINE.json configs[1]
// (23 MB per launch) it is that latency, not bandwidth, that bounds it — more loads in flight per
// lane shorten the chain count per CU.
template <int G, int V, int U>
__global__ __launch_bounds__(kGatherBlock) void resample_gather_kernel(
    const char *__restrict__ src, const int64_t *__restrict__ idx, char *__restrict__ dst,
    int32_t *flags, uint32_t K, uint32_t ppp /* pieces per particle */,
    uint64_t row_pieces /* K * ppp */, uint32_t chunks_per_row, uint32_t blocks_per_row,
    int64_t stride_b, int64_t stride_k) {
  using P = typename Piece<G>::type;
  const uint32_t b = blockIdx.x / blocks_per_row;
  const uint32_t cb = blockIdx.x - b * blocks_per_row;
  const int64_t *irow = idx + (uint64_t)b * K;
  const char *srow = src + (int64_t)b * stride_b;
  char *drow = dst + (uint64_t)b * row_pieces * G;

  uint32_t kk[U][V], rr[U][V];
  bool live[U][V];
  // (particle, piece) of a lane's first chunk by ONE 32-bit division (K * row_bytes < 2^32 is
  // checked on the host); its other chunks lie 256 apart: a constant step (dk, dr) each
  const uint32_t step = kGatherBlock * V;
  const uint32_t dk = step / ppp, dr = step - dk * ppp;
  const uint32_t first = ((cb * U) * kGatherBlock + threadIdx.x) * (uint32_t)V;
  uint32_t k = first / ppp;
  uint32_t r = first - k * ppp;
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const uint32_t chunk = (cb * U + j) * kGatherBlock + threadIdx.x;
    const uint64_t p0 = (uint64_t)chunk * V;
    uint32_t k_i = k, r_i = r;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      live[j][i] = chunk < chunks_per_row && p0 + i < row_pieces;
      kk[j][i] = live[j][i] ? k_i : 0;
      rr[j][i] = live[j][i] ? r_i : 0;
      if (++r_i == ppp) {
        r_i = 0;
        ++k_i;
      }
    }
    k += dk;
    r += dr;
    if (r >= ppp) {
      r -= ppp;
      ++k;
    }
  }
  int64_t anc[U][V];
#pragma unroll
  for (int j = 0; j < U; ++j)
#pragma unroll
    for (int i = 0; i < V; ++i) anc[j][i] = irow[kk[j][i]];
  int bad = 0;
  P piece[U][V];
#pragma unroll
  for (int j = 0; j < U; ++j)
#pragma unroll
    for (int i = 0; i < V; ++i) {
      int64_t a = anc[j][i];
      if ((uint64_t)a >= (uint64_t)K) {  // torch.gather would raise; never fault, report instead
        bad |= live[j][i] ? 1 : 0;
        a = a < 0 ? 0 : (int64_t)K - 1;
      }
      piece[j][i] = *reinterpret_cast<const P *>(srow + a * stride_k + (uint64_t)rr[j][i] * G);
    }
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const uint32_t chunk = (cb * U + j) * kGatherBlock + threadIdx.x;
    char *out = drow + (uint64_t)chunk * V * G;
    if (live[j][V - 1]) {
      if constexpr (V * G == 16 && V > 1) {
        uint4 packed;
        __builtin_memcpy(&packed, piece[j], 16);
        *reinterpret_cast<uint4 *>(out) = packed;
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) reinterpret_cast<P *>(out)[i] = piece[j][i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i)
        if (live[j][i]) reinterpret_cast<P *>(out)[i] = piece[j][i];
    }
  }
  if (bad) raise_flag(flags, AESMC_FLAG_INDEX_OUT_OF_RANGE);
}

// Backward, general form (any index order): one lane per gradient element; a lane that starts a
// run of equal indices (or a kRunCap-aligned piece of a long run) sums the run and issues ONE
// hardware float atomic for it.  With sorted indices and runs shorter than kRunCap every
// destination receives exactly one add onto zero, so the result is then bitwise reproducible.
constexpr uint32_t kRunCap = 32;

template <typename T>
__global__ __launch_bounds__(kGatherBlock) void resample_gather_bwd_kernel(
    const T *__restrict__ grad_out, const int64_t *__restrict__ idx, T *grad_src, int32_t *flags,
    uint32_t K, uint32_t D, uint64_t row_elems /* K * D */, uint32_t blocks_per_row) {
  const uint32_t b = blockIdx.x / blocks_per_row;
  const uint32_t cb = blockIdx.x - b * blocks_per_row;
  const uint64_t e = (uint64_t)cb * kGatherBlock + threadIdx.x;
  if (e >= row_elems) return;
  const uint32_t k = (uint32_t)(e / D);
  const uint32_t c = (uint32_t)(e - (uint64_t)k * D);
  const int64_t *irow = idx + (uint64_t)b * K;
  const int64_t a = irow[k];
  const bool head = (k % kRunCap == 0) || (irow[k - 1] != a);
  if (!head) return;
  if ((uint64_t)a >= (uint64_t)K) {
    raise_flag(flags, AESMC_FLAG_INDEX_OUT_OF_RANGE);
    return;
  }
  const T *grow = grad_out + (uint64_t)b * row_elems;
  T sum = grow[e];
  const uint32_t stop = min(K, (k / kRunCap + 1) * kRunCap);
  for (uint32_t kk = k + 1; kk < stop && irow[kk] == a; ++kk) sum += grow[(uint64_t)kk * D + c];
  unsafeAtomicAdd(grad_src + ((uint64_t)b * K + (uint64_t)a) * D + c, sum);
}

// Backward for SORTED indices (what systematic resampling produces): no atomics, and results that
// are bitwise reproducible from launch to launch.
//
// With idx non-decreasing along k the offspring of source row j form one contiguous run of k, so
// grad_src[b,j,:] is a segmented sum over consecutive rows.  grad_src is zero-filled first (rows
// without offspring stay zero); then one workgroup takes ONE tile of TK consecutive particles of
// one batch row and writes the sums of exactly those runs that END inside its tile:
//   1. stage the tile's indices (+ the particle before and after) in LDS, find run heads and
//      tails with wavefront ballots, compact the tails; no tail -> the tile lies inside one run
//      that flows on, and the workgroup exits before touching any gradient row;
//   2. stage the tile's gradient rows in LDS (coalesced, 16-byte loads);
//   3. if the first run started in an earlier tile, add its earlier rows straight from global
//      memory (`lead`; its start is found by one coalesced look-back, else a binary search);
//   4. every (tail, column) pair sums its run's rows from LDS and stores one element.
// No workgroup waits for another; every gradient row is staged once.
constexpr int kSortedBlock = 256;

template <typename T, bool VEC_LOAD>
__global__ __launch_bounds__(kSortedBlock) void resample_gather_bwd_sorted_kernel(
    const T *__restrict__ grad_out, const int64_t *__restrict__ idx, T *__restrict__ grad_src,
    int32_t *flags, uint32_t K, uint32_t D, uint32_t TK, uint32_t tiles_per_row) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bwd_smem[];
  T *G = reinterpret_cast<T *>(bwd_smem);                      // [TK * D] staged gradient rows
  T *lead = G + (size_t)TK * D;                                // [D] rows of the first run before the tile
  T *partials = lead + D;                                      // [256] per-lane partial sums of `lead`
  int *ids = reinterpret_cast<int *>(partials + kSortedBlock); // [TK + 2]: before, tile, after
  int *tails = ids + TK + 2;                                   // [TK] positions of run tails, compacted
  int *head_of = tails + TK;                                   // [TK] position of the run head at or before i (-1: earlier tile)
  int *wave_tails = head_of + TK;                              // [4] tails per wavefront
  int *wave_head = wave_tails + 4;                             // [4] last head position per wavefront
  int *shared_lo = wave_head + 4;                              // [1]
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid % kWave, wave = tid / kWave;
  const uint32_t b = blockIdx.x / tiles_per_row;
  const uint32_t k0 = (blockIdx.x - b * tiles_per_row) * TK;
  const uint32_t n = min(TK, K - k0);
  const uint64_t row_base = (uint64_t)b * K;
  const int64_t *irow = idx + row_base;
  const T *grow = grad_out + row_base * D;
  T *drow = grad_src + row_base * D;

  // ---- 1. indices: the tile, the particle before it and the one after ---------------------------
  int bad = 0;
  for (uint32_t i = tid; i < n + 2; i += kSortedBlock) {
    const int64_t k = (int64_t)k0 + i - 1;                     // ids[0] = particle k0 - 1, ids[n + 1] = k0 + n
    int64_t a = (k < 0) ? -1 : (k >= (int64_t)K ? (int64_t)K : irow[k]);
    if (k >= 0 && k < (int64_t)K && (uint64_t)a >= (uint64_t)K) {
      bad |= AESMC_FLAG_INDEX_OUT_OF_RANGE;
      a = a < 0 ? -1 : (int64_t)K;                             // contributes nothing, stays memory-safe
    }
    ids[i] = (int)a;
  }
  __syncthreads();
  const bool live = tid < n;                                   // TK <= 256: one particle per lane
  const int mine = live ? ids[1 + tid] : -2;
  const bool is_head = live && ids[tid] != mine;
  const bool is_tail = live && ids[2 + tid] != mine;
  if (live && mine < ids[tid]) bad |= AESMC_FLAG_UNSORTED_INDEX;
  if (live && tid == n - 1 && k0 + n < K && ids[n + 1] < mine) bad |= AESMC_FLAG_UNSORTED_INDEX;
  if (bad) raise_flag(flags, bad);
  const unsigned long long tail_mask = __ballot(is_tail);
  const unsigned long long head_mask = __ballot(is_head);
  const unsigned long long below = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);  // lanes <= mine
  if (lane == 0) {
    wave_tails[wave] = __popcll(tail_mask);
    wave_head[wave] = head_mask ? (int)(wave * kWave + 63 - __clzll(head_mask)) : -1;
  }
  __syncthreads();
  int tail_base = 0, head_before = -1;
  for (uint32_t w = 0; w < wave; ++w) {
    tail_base += wave_tails[w];
    if (wave_head[w] >= 0) head_before = wave_head[w];
  }
  const int num_tails = wave_tails[0] + wave_tails[1] + wave_tails[2] + wave_tails[3];
  if (num_tails == 0) return;                                  // wholly inside one run that flows on
  if (live) {
    const unsigned long long heads_here = head_mask & below;
    head_of[tid] = heads_here ? (int)(wave * kWave + 63 - __clzll(heads_here)) : head_before;
    if (is_tail) tails[tail_base + __popcll(tail_mask & below) - 1] = (int)tid;
  }

  // ---- 2. stage the gradient rows ------------------------------------------------------------------
  const uint32_t ne = n * D;
  if constexpr (VEC_LOAD) {
    using V = typename Vec16<T>::type;
    const V *src = reinterpret_cast<const V *>(grow + (uint64_t)k0 * D);
    V *dst = reinterpret_cast<V *>(G);
    for (uint32_t v = tid; v < ne / Vec16<T>::N; v += kSortedBlock) dst[v] = src[v];
  } else {
    for (uint32_t e = tid; e < ne; e += kSortedBlock) G[e] = grow[(uint64_t)k0 * D + e];
  }

  // ---- 3. rows of the first run that lie before the tile -----------------------------------------
  const int before = ids[0], first_id = ids[1];
  const bool has_lead = (k0 > 0) && (first_id == before) && first_id >= 0 && first_id < (int)K;
  if (has_lead) {
    if (tid == 0) *shared_lo = -1;
    __syncthreads();
    const int64_t back = (int64_t)k0 - 1 - tid;                // lane's candidate for "last particle before the run"
    if (tid > 0 && back >= -1) {
      const bool differs = back < 0 || irow[back] != (int64_t)first_id;
      if (differs && irow[back + 1] == (int64_t)first_id)
        *shared_lo = (int)(back + 1);                          // unique lane (ids are sorted): the run's first particle
    }
    __syncthreads();
    uint32_t lo;
    if (*shared_lo >= 0) {
      lo = (uint32_t)*shared_lo;
    } else {                                                   // longer than the look-back: binary search the rest
      uint32_t l = 0, h = k0 - kSortedBlock;
      while (l < h) {
        const uint32_t mid = (l + h) >> 1;
        if (irow[mid] < (int64_t)first_id) l = mid + 1; else h = mid;
      }
      lo = l;
    }
    // `slots` lanes share a column and take every slots-th row; their partial sums are combined
    // through LDS in slot order (fixed order: reproducible).  Control flow is workgroup-uniform.
    const uint32_t cols = min(D, (uint32_t)kSortedBlock), slots = kSortedBlock / cols;
    const uint32_t slot = tid / cols, col = tid - slot * cols;
    for (uint32_t cbase = 0; cbase < D; cbase += cols) {
      const uint32_t c = cbase + col;
      T acc = T(0);
      if (slot < slots && c < D) {
        // eight independent accumulators keep eight global loads in flight on long runs
        T a[8] = {T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)};
        uint32_t i = lo + slot;
        for (; (uint64_t)i + 7ull * slots < k0; i += 8 * slots) {
#pragma unroll
          for (int q = 0; q < 8; ++q) a[q] += grow[(uint64_t)(i + q * slots) * D + c];
        }
        for (; i < k0; i += slots) a[0] += grow[(uint64_t)i * D + c];
        acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
      }
      if (slot < slots) partials[slot * cols + col] = acc;
      __syncthreads();
      if (tid < cols && cbase + tid < D) {
        T sum = T(0);
        for (uint32_t s2 = 0; s2 < slots; ++s2) sum += partials[s2 * cols + tid];
        lead[cbase + tid] = sum;
      }
      __syncthreads();
    }
  }
  __syncthreads();

  // ---- 4. one element per (tail, column) ---------------------------------------------------------
  const uint32_t total = (uint32_t)num_tails * D;
  for (uint32_t e = tid; e < total; e += kSortedBlock) {
    const uint32_t t = e / D, c = e - t * D;
    const int i = tails[t];
    const int id = ids[1 + i];
    if ((uint32_t)id >= K) continue;                           // out-of-range index: reported above
    const int h = head_of[i];
    T sum = (h < 0 && has_lead) ? lead[c] : T(0);
    for (int r = h < 0 ? 0 : h; r <= i; ++r) sum += G[(uint32_t)r * D + c];
    drow[(uint64_t)id * D + c] = sum;
  }
}

// Output stage of the range kernel: `rows` destination rows of D elements, row j taken from staged
// row slot_of[j] (0xffff: zeros), stored W elements at a time; a lane's pieces lie 256 apart, so
// (row, piece) advance by a constant step instead of a division per piece.
template <typename T, int W>
__device__ __forceinline__ void store_range(T *__restrict__ out, const T *G, const unsigned short *slot_of,
                                            uint32_t rows, uint32_t D, uint32_t tid) {
  struct alignas(sizeof(T) * W) Pack {
    T v[W];
  };
  const uint32_t per_row = D / W;
  const uint32_t qstep = kSortedBlock / per_row, rstep = kSortedBlock - qstep * per_row;
  uint32_t j = tid / per_row, r = tid - j * per_row;
  const uint32_t total = rows * per_row;
  for (uint32_t e = tid; e < total; e += kSortedBlock) {
    const uint32_t slot = slot_of[j];
    Pack p;
    if (slot != 0xffffu) {
      p = *reinterpret_cast<const Pack *>(G + slot * D + r * W);
    } else {
#pragma unroll
      for (int i = 0; i < W; ++i) p.v[i] = T(0);
    }
    reinterpret_cast<Pack *>(out)[e] = p;
    j += qstep;
    r += rstep;
    if (r >= per_row) {
      r -= per_row;
      ++j;
    }
  }
}

// Backward for SORTED indices without a zero-fill launch (the one that runs).  Same source tiles,
// run detection and look-back as the kernel above, but every destination row is written exactly
// once: idx being non-decreasing, the rows a tile is responsible for form ONE contiguous range
//     ( idx[k0 - 1], idx[k0 + n - 1] ]        (from the row of the particle before the tile,
//                                              exclusive, to the row of its last particle)
// trimmed at either end: the first row belongs to it only when that run, begun in an earlier tile,
// ends here; the last row only when its run ends here; the tile holding particle K - 1 also owns
// everything up to row K - 1.  Rows of the range with a run ending here receive their sum, the
// others — particles without offspring — zero.  The sums are left in the staging buffer at their
// run's last row; a small map (destination row -> staged row, `cap` entries of 16 bits in LDS, the
// range is walked in pieces of `cap` rows) then lets the workgroup store the whole range as one
// dense, coalesced stream.
template <typename T, bool VEC_LOAD>
__global__ __launch_bounds__(kSortedBlock) void resample_gather_bwd_range_kernel(
    const T *__restrict__ grad_out, const int64_t *__restrict__ idx, T *__restrict__ grad_src,
    int32_t *flags, uint32_t K, uint32_t D, uint32_t TK, uint32_t tiles_per_row, uint32_t cap, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char range_smem[];
  T *G = reinterpret_cast<T *>(range_smem);                    // [TK * D] staged gradient rows
  T *lead = G + (size_t)TK * D;                                // [D] rows of the first run before the tile
  T *partials = lead + D;                                      // [256] per-lane partial sums of `lead`
  int *ids = reinterpret_cast<int *>(partials + kSortedBlock); // [TK + 2]: before, tile, after
  int *tails = ids + TK + 2;                                   // [TK] positions of run tails, compacted
  int *head_of = tails + TK;                                   // [TK] position of the run head at or before i (-1: earlier tile)
  int *wave_tails = head_of + TK;                              // [4] tails per wavefront
  int *wave_head = wave_tails + 4;                             // [4] last head position per wavefront
  int *shared_lo = wave_head + 4;                              // [1]
  unsigned short *slot_of = reinterpret_cast<unsigned short *>(shared_lo + 1);  // [cap] staged row holding a
                                                               // destination row's sum (0xffff: none)
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid % kWave, wave = tid / kWave;
  const uint32_t b = blockIdx.x / tiles_per_row;
  const uint32_t k0 = (blockIdx.x - b * tiles_per_row) * TK;
  const uint32_t n = min(TK, K - k0);
  const uint64_t row_base = (uint64_t)b * K;
  const int64_t *irow = idx + row_base;
  const T *grow = grad_out + row_base * D;
  T *drow = grad_src + row_base * D;

  // ---- 1. indices: the tile, the particle before it and the one after ---------------------------
  int bad = 0;
  for (uint32_t i = tid; i < n + 2; i += kSortedBlock) {
    const int64_t k = (int64_t)k0 + i - 1;                     // ids[0] = particle k0 - 1, ids[n + 1] = k0 + n
    int64_t a = (k < 0) ? -1 : (k >= (int64_t)K ? (int64_t)K : irow[k]);
    if (k >= 0 && k < (int64_t)K && (uint64_t)a >= (uint64_t)K) {
      bad |= AESMC_FLAG_INDEX_OUT_OF_RANGE;
      a = a < 0 ? -1 : (int64_t)K;                             // contributes nothing, stays memory-safe
    }
    ids[i] = (int)a;
  }
  __syncthreads();
  const bool live = tid < n;                                   // TK <= 256: one particle per lane
  const int mine = live ? ids[1 + tid] : -2;
  const bool is_head = live && ids[tid] != mine;
  const bool is_tail = live && ids[2 + tid] != mine;
  if (live && mine < ids[tid]) bad |= AESMC_FLAG_UNSORTED_INDEX;
  if (live && tid == n - 1 && k0 + n < K && ids[n + 1] < mine) bad |= AESMC_FLAG_UNSORTED_INDEX;
  if (bad) raise_flag(flags, bad);
  const unsigned long long tail_mask = __ballot(is_tail);
  const unsigned long long head_mask = __ballot(is_head);
  const unsigned long long below = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);  // lanes <= mine
  if (lane == 0) {
    wave_tails[wave] = __popcll(tail_mask);
    wave_head[wave] = head_mask ? (int)(wave * kWave + 63 - __clzll(head_mask)) : -1;
  }
  __syncthreads();
  int tail_base = 0, head_before = -1;
  for (uint32_t w = 0; w < wave; ++w) {
    tail_base += wave_tails[w];
    if (wave_head[w] >= 0) head_before = wave_head[w];
  }
  const int num_tails = wave_tails[0] + wave_tails[1] + wave_tails[2] + wave_tails[3];
  if (live) {
    const unsigned long long heads_here = head_mask & below;
    head_of[tid] = heads_here ? (int)(wave * kWave + 63 - __clzll(heads_here)) : head_before;
    if (is_tail) tails[tail_base + __popcll(tail_mask & below) - 1] = (int)tid;
  }

  // ---- the destination rows this tile writes: [jlo, jhi] -------------------------------------------
  const int before = ids[0], first_id = ids[1], last_id = ids[n], after = ids[n + 1];
  const bool has_lead = (k0 > 0) && (first_id == before) && first_id >= 0 && first_id < (int)K;
  // the run entering from the left ends here unless the whole tile belongs to it and it flows on
  const bool lead_ends_here = has_lead && !(last_id == first_id && after == last_id);
  int jlo = (has_lead && lead_ends_here) ? before : before + 1;
  int jhi = (k0 + n == K) ? (int)K - 1 : ((after == last_id) ? last_id - 1 : last_id);
  if (jlo < 0) jlo = 0;
  if (jhi > (int)K - 1) jhi = (int)K - 1;
  const uint32_t L = jhi >= jlo ? (uint32_t)(jhi - jlo + 1) : 0u;
  if (L == 0) return;                                          // wholly inside one run that flows on

  // ---- 2. stage the gradient rows ------------------------------------------------------------------
  const uint32_t ne = n * D;
  if (num_tails > 0) {
    if constexpr (VEC_LOAD) {
      using V = typename Vec16<T>::type;
      const V *src = reinterpret_cast<const V *>(grow + (uint64_t)k0 * D);
      V *dst = reinterpret_cast<V *>(G);
      for (uint32_t v = tid; v < ne / Vec16<T>::N; v += kSortedBlock) dst[v] = src[v];
    } else {
      for (uint32_t e = tid; e < ne; e += kSortedBlock) G[e] = grow[(uint64_t)k0 * D + e];
    }
  }

  // ---- 3. rows of the first run that lie before the tile -----------------------------------------
  if (has_lead && lead_ends_here) {
    if (tid == 0) *shared_lo = -1;
    __syncthreads();
    const int64_t back = (int64_t)k0 - 1 - tid;                // lane's candidate for "last particle before the run"
    if (tid > 0 && back >= -1) {
      const bool differs = back < 0 || irow[back] != (int64_t)first_id;
      if (differs && irow[back + 1] == (int64_t)first_id)
        *shared_lo = (int)(back + 1);                          // unique lane (ids are sorted): the run's first particle
    }
    __syncthreads();
    uint32_t lo;
    if (*shared_lo >= 0) {
      lo = (uint32_t)*shared_lo;
    } else {                                                   // longer than the look-back: binary search the rest
      uint32_t l = 0, h = k0 - kSortedBlock;
      while (l < h) {
        const uint32_t mid = (l + h) >> 1;
        if (irow[mid] < (int64_t)first_id) l = mid + 1; else h = mid;
      }
      lo = l;
    }
    // `slots` lanes share a column and take every slots-th row; their partial sums are combined
    // through LDS in slot order (fixed order: reproducible).  Control flow is workgroup-uniform.
    const uint32_t cols = min(D, (uint32_t)kSortedBlock), slots = kSortedBlock / cols;
    const uint32_t slot = tid / cols, col = tid - slot * cols;
    for (uint32_t cbase = 0; cbase < D; cbase += cols) {
      const uint32_t c = cbase + col;
      T sum = T(0);
      if (slot < slots && c < D) {
        // eight independent accumulators keep eight global loads in flight on long runs
        T a[8] = {T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)};
        uint32_t i = lo + slot;
        for (; (uint64_t)i + 7ull * slots < k0; i += 8 * slots) {
#pragma unroll
          for (int q = 0; q < 8; ++q) a[q] += grow[(uint64_t)(i + q * slots) * D + c];
        }
        for (; i < k0; i += slots) a[0] += grow[(uint64_t)i * D + c];
        sum = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
      }
      if (slot < slots) partials[slot * cols + col] = sum;
      __syncthreads();
      if (tid < cols && cbase + tid < D) {
        T total = T(0);
        for (uint32_t s2 = 0; s2 < slots; ++s2) total += partials[s2 * cols + tid];
        lead[cbase + tid] = total;
      }
      __syncthreads();
    }
  }
  __syncthreads();

  // ---- 4. one element per (tail, column): the run's sum replaces its last staged row ---------------
  const uint32_t total = (uint32_t)num_tails * D;
  for (uint32_t e = tid; e < total; e += kSortedBlock) {
    const uint32_t t = e / D, c = e - t * D;
    const int i = tails[t];
    const int h = head_of[i];
    T sum = (h < 0 && has_lead) ? lead[c] : T(0);
    for (int r = h < 0 ? 0 : h; r <= i; ++r) sum += G[(uint32_t)r * D + c];
    G[(uint32_t)i * D + c] = sum;                              // only this lane touches column c of rows h..i
  }

  // ---- 5. the range goes out: sums where a run ended, zeros where a particle left no offspring --------
  for (uint32_t piece = 0; piece < L; piece += cap) {
    const uint32_t np = min(cap, L - piece);
    const int first_row = jlo + (int)piece;
    __syncthreads();
    for (uint32_t j = tid; j < np; j += kSortedBlock) slot_of[j] = 0xffffu;
    __syncthreads();
    for (uint32_t t = tid; t < (uint32_t)num_tails; t += kSortedBlock) {
      const int i = tails[t];
      const int id = ids[1 + i];                               // ids outside [jlo, jhi]: out of range, reported above
      if (id >= first_row && id < first_row + (int)np && id <= jhi) slot_of[id - first_row] = (unsigned short)i;
    }
    __syncthreads();
    T *out = drow + (uint64_t)first_row * D;
    // W elements per store: the widest of 16 / 8 / 4 bytes that divides a row (and the base address)
    if (W == 4)
      store_range<T, 4>(out, G, slot_of, np, D, tid);
    else if (W == 2)
      store_range<T, 2>(out, G, slot_of, np, D, tid);
    else
      store_range<T, 1>(out, G, slot_of, np, D, tid);
  }
}

static inline int low_pow2(uint64_t x, int cap) {  // largest power of two <= cap dividing x
  int g = cap;
  while (g > 1 && (x % (uint64_t)g) != 0) g >>= 1;
  return g;
}

template <int G, int V>
static void launch_gather(const void *src, const int64_t *idx, void *dst, int32_t *flags, int64_t B,
                          int64_t K, int64_t row_bytes, int64_t sb, int64_t sk, hipStream_t s) {
  const uint32_t ppp = (uint32_t)(row_bytes / G);
  const uint64_t row_pieces = (uint64_t)K * ppp;
  const uint32_t chunks = (uint32_t)((row_pieces + V - 1) / V);
  // four chunks per lane once a row holds at least that much work for a full workgroup
  if (chunks >= 4u * kGatherBlock) {
    const uint32_t bpr = (chunks + 4 * kGatherBlock - 1) / (4 * kGatherBlock);
    hipLaunchKernelGGL((resample_gather_kernel<G, V, 4>), dim3((unsigned)(B * bpr)), dim3(kGatherBlock),
                       0, s, (const char *)src, idx, (char *)dst, flags, (uint32_t)K, ppp, row_pieces,
                       chunks, bpr, sb, sk);
  } else {
    const uint32_t bpr = (chunks + kGatherBlock - 1) / kGatherBlock;
    hipLaunchKernelGGL((resample_gather_kernel<G, V, 1>), dim3((unsigned)(B * bpr)), dim3(kGatherBlock),
                       0, s, (const char *)src, idx, (char *)dst, flags, (uint32_t)K, ppp, row_pieces,
                       chunks, bpr, sb, sk);
  }
}

}  // namespace aesmc

using namespace aesmc;

// 0: the range kernel for sorted indices (default: every row written once, no zero fill); 1: the
// source-tile kernel behind a zero fill (kept for rows the first declines, and selectable for A/B
// timing: tools/kbench.py)
// (a measurement / test hook, not part of the C ABI of include/aesmc_hip.h: AESMC_SORTED_BACKWARD_KERNEL in the
// environment sets the default, aesmc_test_set_sorted_backward_kernel switches inside one process)
static int g_sorted_backward_kernel = [] {
  const char *v = measurement_knob("AESMC_SORTED_BACKWARD_KERNEL");
  return (v != nullptr && v[0] == '1') ? 1 : 0;
}();
extern "C" int aesmc_test_set_sorted_backward_kernel(int which) {
  if (which != 0 && which != 1) return AESMC_ERR_INVALID_ARGUMENT;
  g_sorted_backward_kernel = which;
  return AESMC_OK;
}

extern "C" int aesmc_resample_gather(const void *src, const int64_t *idx, void *dst, int32_t *flags,
                                     int64_t B, int64_t K, int64_t row_bytes, int64_t src_stride_b,
                                     int64_t src_stride_k, void *stream) {
  using namespace aesmc;
  if (src == nullptr || idx == nullptr || dst == nullptr || B < 0 || K < 0 || row_bytes < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0 || row_bytes == 0) return AESMC_OK;
  // 32-bit piece arithmetic inside a batch row; grid is B * blocks_per_row workgroups.
  if ((uint64_t)K * (uint64_t)row_bytes >= (1ull << 32) || K >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  // Piece size: every source piece address must be G-aligned.
  int G = low_pow2((uint64_t)row_bytes, 16);
  G = low_pow2((uint64_t)(uintptr_t)src, G);
  G = low_pow2((uint64_t)(src_stride_b < 0 ? -src_stride_b : src_stride_b), G);
  G = low_pow2((uint64_t)(src_stride_k < 0 ? -src_stride_k : src_stride_k), G);
  G = low_pow2((uint64_t)(uintptr_t)dst, G);
  // 16-byte stores need a 16-byte aligned dst base and batch-row pitch.
  const bool vec = (((uintptr_t)dst & 15u) == 0) && (((uint64_t)K * (uint64_t)row_bytes) % 16 == 0);
  {
    const uint64_t pieces = (uint64_t)K * (uint64_t)(row_bytes / G);
    const int v = vec ? 16 / G : 1;
    const uint64_t bpr = ((pieces + v - 1) / v + kGatherBlock - 1) / kGatherBlock;
    if ((uint64_t)B * bpr > 0x7fffffffull) return AESMC_ERR_UNSUPPORTED;
  }
#define AESMC_GATHER_CASE(g)                                                                        \
  case g:                                                                                           \
    if (vec)                                                                                        \
      launch_gather<g, 16 / g>(src, idx, dst, flags, B, K, row_bytes, src_stride_b, src_stride_k, s); \
    else                                                                                            \
      launch_gather<g, 1>(src, idx, dst, flags, B, K, row_bytes, src_stride_b, src_stride_k, s);    \
    break;
  switch (G) {
    AESMC_GATHER_CASE(16)
    AESMC_GATHER_CASE(8)
    AESMC_GATHER_CASE(4)
    AESMC_GATHER_CASE(2)
    AESMC_GATHER_CASE(1)
    default:
      return AESMC_ERR_INVALID_ARGUMENT;
  }
#undef AESMC_GATHER_CASE
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

template <typename T>
static bool launch_sorted_backward(const void *grad_out, const int64_t *idx, void *grad_src,
                                   int32_t *flags, int64_t B, int64_t K, int64_t D, hipStream_t s) {
  // tile: as many particles as fit 32 KiB of staged gradient rows, at most 256
  const int64_t row_bytes = D * (int64_t)sizeof(T);
  int64_t TK = (32 * 1024) / row_bytes;
  if (TK > 256) TK = 256;
  if (TK < 8) return false;                                    // very wide rows: use the general kernel
  if (TK > K) TK = K;
  const int64_t tiles = (K + TK - 1) / TK;
  if (B * tiles > 0x7fffffffLL) return false;
  const size_t lds = (size_t)(TK * D + D + kSortedBlock) * sizeof(T) + (size_t)(3 * TK + 2 + 12) * sizeof(int);
  constexpr int N = Vec16<T>::N;
  const bool vec = (((uintptr_t)grad_out & 15u) == 0) && ((K * D) % N == 0) && ((TK * D) % N == 0);
  dim3 grid((unsigned)(B * tiles)), block(kSortedBlock);
  if (vec)
    hipLaunchKernelGGL((resample_gather_bwd_sorted_kernel<T, true>), grid, block, lds, s,
                       (const T *)grad_out, idx, (T *)grad_src, flags, (uint32_t)K, (uint32_t)D,
                       (uint32_t)TK, (uint32_t)tiles);
  else
    hipLaunchKernelGGL((resample_gather_bwd_sorted_kernel<T, false>), grid, block, lds, s,
                       (const T *)grad_out, idx, (T *)grad_src, flags, (uint32_t)K, (uint32_t)D,
                       (uint32_t)TK, (uint32_t)tiles);
  return true;
}

// The range kernel: needs no zero fill.  Declines rows too wide to stage (the caller then zero-fills
// and runs one of the kernels above).
template <typename T>
static bool launch_range_backward(const void *grad_out, const int64_t *idx, void *grad_src,
                                  int32_t *flags, int64_t B, int64_t K, int64_t D, hipStream_t s) {
  constexpr int N = Vec16<T>::N;
  const int64_t row_bytes = D * (int64_t)sizeof(T);
  int64_t TK = (28 * 1024) / row_bytes;                        // staged rows: at most 28 KiB
  if (TK > 256) TK = 256;
  if (TK < 8) return false;
  if (TK > K) TK = K;
  int64_t cap = K < 2048 ? K : 2048;                           // rows of the range mapped at a time (2 B each)
  const int64_t tiles = (K + TK - 1) / TK;
  if (B * tiles > 0x7fffffffLL) return false;
  const size_t floats = (size_t)(TK * D + D + kSortedBlock);
  const size_t lds = floats * sizeof(T) + (size_t)(3 * TK + 2 + 9) * sizeof(int) + (size_t)((cap + 1) / 2 * 2) * 2;
  const bool vec = (((uintptr_t)grad_out & 15u) == 0) && ((K * D) % N == 0) && ((TK * D) % N == 0);
  int W = (int)(16 / sizeof(T));                                // elements per store of the output stage
  while (W > 1 && (D % W != 0 || ((uintptr_t)grad_src % (W * sizeof(T))) != 0)) W /= 2;
  dim3 grid((unsigned)(B * tiles)), block(kSortedBlock);
  if (vec)
    hipLaunchKernelGGL((resample_gather_bwd_range_kernel<T, true>), grid, block, lds, s, (const T *)grad_out,
                       idx, (T *)grad_src, flags, (uint32_t)K, (uint32_t)D, (uint32_t)TK, (uint32_t)tiles,
                       (uint32_t)cap, W);
  else
    hipLaunchKernelGGL((resample_gather_bwd_range_kernel<T, false>), grid, block, lds, s, (const T *)grad_out,
                       idx, (T *)grad_src, flags, (uint32_t)K, (uint32_t)D, (uint32_t)TK, (uint32_t)tiles,
                       (uint32_t)cap, W);
  return true;
}

extern "C" int aesmc_resample_gather_backward(int dtype, const void *grad_out, const int64_t *idx,
                                              void *grad_src, int32_t *flags, int64_t B, int64_t K,
                                              int64_t row_elems, int index_is_sorted, void *stream) {
  using namespace aesmc;
  if (grad_out == nullptr || idx == nullptr || grad_src == nullptr || B < 0 || K < 0 || row_elems < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0 || row_elems == 0) return AESMC_OK;
  if ((uint64_t)K * (uint64_t)row_elems >= (1ull << 32) || K >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype == AESMC_F32 ? 4 : 8;
  const uint64_t re = (uint64_t)K * (uint64_t)row_elems;
  if (index_is_sorted && g_sorted_backward_kernel == 0) {
    const bool launched = dtype == AESMC_F32
        ? launch_range_backward<float>(grad_out, idx, grad_src, flags, B, K, row_elems, s)
        : launch_range_backward<double>(grad_out, idx, grad_src, flags, B, K, row_elems, s);
    if (launched) return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
  }
  // the kernels below write only rows that have offspring: everything else must read as zero
  if (!zero_fill_async(grad_src, (size_t)B * re * esz, s)) return AESMC_ERR_LAUNCH;
  if (index_is_sorted) {
    const bool launched = dtype == AESMC_F32
        ? launch_sorted_backward<float>(grad_out, idx, grad_src, flags, B, K, row_elems, s)
        : launch_sorted_backward<double>(grad_out, idx, grad_src, flags, B, K, row_elems, s);
    if (launched) return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
  }
  const uint64_t bpr = (re + kGatherBlock - 1) / kGatherBlock;
  if ((uint64_t)B * bpr > 0x7fffffffull) return AESMC_ERR_UNSUPPORTED;
  dim3 grid((unsigned)((uint64_t)B * bpr)), block(kGatherBlock);
  if (dtype == AESMC_F32)
    hipLaunchKernelGGL((resample_gather_bwd_kernel<float>), grid, block, 0, s, (const float *)grad_out,
                       idx, (float *)grad_src, flags, (uint32_t)K, (uint32_t)row_elems, re,
                       (uint32_t)bpr);
  else
    hipLaunchKernelGGL((resample_gather_bwd_kernel<double>), grid, block, 0, s,
                       (const double *)grad_out, idx, (double *)grad_src, flags, (uint32_t)K,
                       (uint32_t)row_elems, re, (uint32_t)bpr);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}
