// Shared pieces of the linear-Gaussian propagation kernels (linear_gaussian.hip: K8, K9, K10 / K15;
// linear_gaussian_backward.hip: K11, K12, K14): tile layout and staging, the fma-chain maps, the per-row
// vectors of a tile, register prefetch, and the host-side launch helpers.  See linear_gaussian.hip.
#pragma once
#include <algorithm>

#include "common.hpp"
namespace aesmc {

constexpr int kLgBlock = 256;
constexpr int kLgMaxDim = 16;

// A staged tile keeps its rows apart by `rs` elements: rs = d, a flat copy of the [np, d] block in HBM,
// unless the rows are whole 16-byte vectors of floats (d % 4 == 0), whose strides would put a
// wavefront's row reads on very few LDS banks (d = 12 padded to 16: two banks) — those get one or two
// 16-byte pads per row so that rs / 4 is odd (rs = 12, 12, 20, 20 for d = 4, 8, 12, 16: the best a
// 16-byte-aligned row can do, 8 banks).  Either way a 16-byte vector of the block is ONE 16-byte LDS access
// and a lane's element (p, i) sits at p * rs + i: a row base per particle, immediate offsets per
// element, no per-element index arithmetic.
struct LgLayout {
  uint32_t rs;    // row stride in elements
  uint32_t mul;   // padded rows: ceil(2^17 / (d / 4)), so (v * mul) >> 17 == v / (d / 4) for v < 2^15; else 0
  uint32_t padv;  // padded rows: 16-byte pads per row (1 or 2)
};
template <typename T> __host__ __device__ __forceinline__ LgLayout lg_layout(uint32_t d) {
  LgLayout l;
  const bool padded = sizeof(T) == 4 && d != 0 && (d & 3u) == 0;
  l.padv = padded ? ((((d >> 2) + 1) & 1u) ? 1u : 2u) : 0u;      // d / 4 + pads odd
  l.rs = d + 4 * l.padv;
  l.mul = padded ? (131072u + d / 4 - 1) / (d / 4) : 0u;
  return l;
}
// vector v of the flat block -> vector slot in the tile
__device__ __forceinline__ uint32_t lg_slot(uint32_t v, const LgLayout &l) {
  return l.mul != 0 ? v + ((v * l.mul) >> 17) * l.padv : v;
}
// elements a tile of `particles` rows occupies (+16: the matrix-core operand reads run past a row's end)
template <typename T> static inline size_t lg_tile_elems(size_t particles, size_t d) {
  return particles * lg_layout<T>((uint32_t)d).rs + 16;
}

// threadIdx.x through an opaque move: inside the persistent tile loops everything derived from the lane's
// index is loop-invariant, and the compiler would hoist a hundred LDS addresses out of the loop and hold
// them in registers for the whole kernel (measured: 223 VGPRs for 4-value rows); recomputing them per
// tile costs a few adds.
// The forward kernels keep the hoisting (they have registers to spare and run faster with it: K10
// 99 against 130 us); the backward kernels set LG_OPAQUE_TID.
template <bool OPAQUE> __device__ __forceinline__ uint32_t lg_tid_impl() {
  uint32_t t = threadIdx.x;
  if constexpr (OPAQUE) asm volatile("" : "+v"(t));
  return t;
}
#define lg_tid() lg_tid_impl<LG_OPAQUE>()

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---- two particles per lane on the packed fp32 pipe ------------------------------------------------------------
// A lane that owns two particles keeps element j of both in ONE 64-bit register pair and advances both chains with one
// v_pk_fma_f32.  The weight that multiplies them is one float of a pair read from LDS (weights j, j+1 of input i lie
// side by side), selected by the instruction's op_sel bits — written as inline assembly because the compiler, left to
// vectorise `acc[j][r] = fma(w, x[r], acc[j][r])` itself, pairs DIFFERENT weights instead (A's and Q's, from two LDS
// arrays) and spends a v_mov per multiply-add putting them side by side (measured: 299 v_mov beside 330 v_pk_fma in
// K16's loop).  Same operation, same operands, same rounding as fmaf: the same bits.
typedef float lg_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lg_f2 lg_pk_fma_lo(lg_f2 w, lg_f2 x, lg_f2 acc) {      // acc + w.x * x
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(x));
  return acc;
}
__device__ __forceinline__ lg_f2 lg_pk_fma_hi(lg_f2 w, lg_f2 x, lg_f2 acc) {      // acc + w.y * x
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(x));
  return acc;
}
template <typename T, int PPL> struct LgPacked { static constexpr bool value = false; };
template <> struct LgPacked<float, 2> { static constexpr bool value = true; };

// acc[j] (both particles) += W[j][i] * x_i (both particles) for the DP outputs of input element i; `wrow` = wt + i * DP
template <int DP>
__device__ __forceinline__ void lg_pk_column(const float *__restrict__ wrow, lg_f2 xv, lg_f2 (&acc)[DP]) {
  static_assert(DP % 2 == 0, "weights are read in pairs");
  const lg_f2 *w2 = reinterpret_cast<const lg_f2 *>(wrow);
#pragma unroll
  for (int jj = 0; jj < DP / 2; ++jj) {
    const lg_f2 w = w2[jj];
    acc[2 * jj] = lg_pk_fma_lo(w, xv, acc[2 * jj]);
    acc[2 * jj + 1] = lg_pk_fma_hi(w, xv, acc[2 * jj + 1]);
  }
}

template <typename T> struct LgConst;
template <> struct LgConst<float> {
  static __device__ __forceinline__ float half_log_2pi() { return 0.9189385332046727f; }
};
template <> struct LgConst<double> {
  static __device__ __forceinline__ double half_log_2pi() { return 0.9189385332046727; }
};

// Device-side copy of aesmc_affine_map.
struct LgMap {
  const void *w;
  int64_t sj, si;      // element strides of the weight [dout, din]
  const void *off;     // nullptr, or off[b * off_sb + j]
  int64_t off_sb;
  int32_t dout, din;
};

// [np, d] rows, contiguous in HBM from `src` (16-byte aligned: tiles start at multiples of 256
// particles), into a tile: 16-byte loads, 16-byte LDS stores.
template <typename T, bool LG_OPAQUE = false>
__device__ __forceinline__ void lg_stage_rows(const T *__restrict__ src, uint32_t ne, T *__restrict__ tile,
                                              const LgLayout &l, int stream) {
  constexpr int N = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const uint32_t nvec = ne / N;
#pragma unroll 4
  for (uint32_t v = lg_tid(); v < nvec; v += kLgBlock)
    reinterpret_cast<V *>(tile)[lg_slot(v, l)] = load16(reinterpret_cast<const V *>(src) + v, stream);
#pragma unroll 1
  for (uint32_t e = nvec * N + threadIdx.x; e < ne; e += kLgBlock) tile[e] = src[e];   // unpadded layouts only
}

template <typename T, bool LG_OPAQUE = false>
__device__ __forceinline__ void lg_store_rows(T *__restrict__ dst, uint32_t ne, const T *__restrict__ tile,
                                              const LgLayout &l) {
  constexpr int N = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const uint32_t nvec = ne / N;
#pragma unroll 4
  for (uint32_t v = lg_tid(); v < nvec; v += kLgBlock)
    reinterpret_cast<V *>(dst)[v] = reinterpret_cast<const V *>(tile)[lg_slot(v, l)];
#pragma unroll 1
  for (uint32_t e = nvec * N + threadIdx.x; e < ne; e += kLgBlock) dst[e] = tile[e];
}

// Weight [dout, din] (any strides) zero-padded and TRANSPOSED in LDS: wt[i * DP + j] = W[j][i], so the
// DP weights that multiply input element i are one contiguous (broadcast) read.
template <typename T, int DP>
__device__ __forceinline__ void lg_stage_weight(const LgMap &m, T *__restrict__ wt) {
  const T *w = reinterpret_cast<const T *>(m.w);
#pragma unroll 1
  for (uint32_t e = threadIdx.x; e < DP * DP; e += kLgBlock) {
    const int i = e / DP, j = e - i * DP;
    wt[e] = (j < m.dout && i < m.din) ? w[(int64_t)j * m.sj + (int64_t)i * m.si] : T(0);
  }
}

// Two weights INTERLEAVED: wt[(i * DP + j) * 2 + which] = W_which[j][i].  A kernel that applies both maps to the same
// input (the transition's and the proposal's locations of x_{t-1}) reads the two weights of (i, j) as one 8-byte pair —
// what the compiler's own pairing of the two multiply-adds onto v_pk_fma_f32 wants side by side (from two separate
// arrays it spent a v_mov per multiply-add on putting them there).
template <typename T, int DP>
__device__ __forceinline__ void lg_stage_weight_pair(const LgMap &m0, const LgMap &m1, T *__restrict__ wt) {
#pragma unroll 1
  for (uint32_t e = threadIdx.x; e < 2 * DP * DP; e += kLgBlock) {
    const uint32_t which = e & 1u, ij = e >> 1;
    const int i = ij / DP, j = ij - i * DP;
    const LgMap &m = which ? m1 : m0;
    const T *w = reinterpret_cast<const T *>(m.w);
    wt[e] = (j < m.dout && i < m.din) ? w[(int64_t)j * m.sj + (int64_t)i * m.si] : T(0);
  }
}

// Which batch row each of a lane's PPL particles lies in (flat particle index n = b K + k).  Lanes past
// the tile's end take particle 0 of the tile: they compute on valid addresses and store nothing.
template <int PPL, bool LG_OPAQUE = false>
__device__ __forceinline__ void lg_rows(int64_t n0, uint32_t np, uint32_t K, uint32_t (&p)[PPL], bool (&live)[PPL],
                                        uint32_t (&brow)[PPL]) {
  const uint32_t b0 = (uint32_t)(n0 / K);
  const uint32_t k0 = (uint32_t)(n0 - (int64_t)b0 * K);
#pragma unroll
  for (int r = 0; r < PPL; ++r) {
    const uint32_t q = lg_tid() + r * kLgBlock;
    live[r] = q < np;
    p[r] = live[r] ? q : 0u;
    brow[r] = b0 + (k0 + p[r]) / K;
  }
}

// acc[j][r] = off[b(r)][j] for j < dout (the chain's starting value); zero without an offset.  Elements
// j >= dout repeat the last one: their weights are zero and nothing reads them.
template <typename T, int DP, int PPL>
__device__ __forceinline__ void lg_offsets(const LgMap &m, const uint32_t (&brow)[PPL], T (&acc)[DP][PPL]) {
  const T *off = reinterpret_cast<const T *>(m.off);
  if (off != nullptr) {
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const T *row = off + (int64_t)brow[r] * m.off_sb;
#pragma unroll
      for (int j = 0; j < DP; ++j) acc[j][r] = row[min(j, m.dout - 1)];
    }
  } else {
#pragma unroll
    for (int j = 0; j < DP; ++j)
#pragma unroll
      for (int r = 0; r < PPL; ++r) acc[j][r] = T(0);
  }
}

// acc[j][r] = fma(W[j][i], x[r][i], acc[j][r]) for i = 0 .. din-1 in turn, x read from a staged tile
// (`base[r]` = the particle's first element in it).
template <typename T, int DP, int PPL>
__device__ __forceinline__ void lg_apply_tile(const T *__restrict__ wt, const T *__restrict__ tile,
                                              const uint32_t (&base)[PPL], int din, T (&acc)[DP][PPL]) {
  if constexpr (LgPacked<T, PPL>::value) {
    lg_f2 pk[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) pk[j] = lg_f2{acc[j][0], acc[j][1]};
#pragma unroll
    for (int i = 0; i < DP; ++i)
      if (i < din) lg_pk_column<DP>(wt + i * DP, lg_f2{tile[base[0] + i], tile[base[1] + i]}, pk);
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      acc[j][0] = pk[j].x;
      acc[j][1] = pk[j].y;
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < DP; ++i) {
    if (i < din) {
      T xv[PPL];
  #pragma unroll
      for (int r = 0; r < PPL; ++r) xv[r] = tile[base[r] + i];
  #pragma unroll
      for (int j = 0; j < DP; ++j) {
        const T w = wt[i * DP + j];
  #pragma unroll
        for (int r = 0; r < PPL; ++r) acc[j][r] = fma_t(w, xv[r], acc[j][r]);
      }
    }
  }
}

// The same chains as a LOOP over the input elements (two per trip): one column of weights live at a time
// instead of the whole matrix, a few hundred bytes of code instead of DP^2 unrolled multiply-adds —
// what the register-heavy backward kernel needs.  Same order of operations, same bits.
template <typename T, int DP, int PPL, int UNROLL = 2>
__device__ __forceinline__ void lg_apply_loop(const T *__restrict__ wt, const T *__restrict__ tile,
                                              const uint32_t (&base)[PPL], uint32_t din, T (&acc)[DP][PPL]) {
#pragma unroll UNROLL
  for (uint32_t i = 0; i < din; ++i) {
    T xv[PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r) xv[r] = tile[base[r] + i];
    const T *w = wt + i * DP;
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      const T wj = w[j];
#pragma unroll
      for (int r = 0; r < PPL; ++r) acc[j][r] = fma_t(wj, xv[r], acc[j][r]);
    }
  }
}

// ---- per-batch-row vectors (offsets, the observation) of a tile ------------------------------------------
// A tile of TP consecutive particles spans the batch rows b0 .. b0 + nrows - 1.  When they are few
// (K >= TP / 6, every BASELINE shape) their vectors are staged once per tile in an LDS table
// tab[(slot * NA + a) * DP + j] and lanes pick theirs with broadcast reads; otherwise (tiny K) every
// lane loads its own row's vectors from global memory.  Elements past a vector's length are zero in
// the table, repeats of the last one from global memory: nothing reads them.
constexpr int kLgRowsMax = 8;

template <typename T> struct LgRowVec {
  const T *ptr;      // nullptr: absent (zeros)
  int64_t sb;        // element stride between batch rows (0: one vector shared by all)
  int len;
};

template <typename T, int DP, int NA, bool LG_OPAQUE = false>
__device__ __forceinline__ void lg_stage_table(const LgRowVec<T> (&vec)[NA], uint32_t b0, uint32_t nrows,
                                               T *__restrict__ tab) {
#pragma unroll 1
  for (uint32_t idx = lg_tid(); idx < nrows * NA * DP; idx += kLgBlock) {
    const uint32_t j = idx % DP, a = (idx / DP) % NA, row = idx / (DP * NA);
    T value = T(0);
#pragma unroll
    for (int c = 0; c < NA; ++c)
      if (a == (uint32_t)c && vec[c].ptr != nullptr && (int)j < vec[c].len)
        value = vec[c].ptr[(int64_t)(b0 + row) * vec[c].sb + j];
    tab[idx] = value;
  }
}

// out[j][r] = vector A of the batch row of the lane's particle r.
template <typename T, int DP, int PPL, int NA, int A>
__device__ __forceinline__ void lg_row_values(const LgRowVec<T> (&vec)[NA], bool use_tab, const T *__restrict__ tab,
                                              uint32_t b0, const uint32_t (&brow)[PPL], T (&out)[DP][PPL]) {
  if (use_tab) {
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const T *row = tab + ((brow[r] - b0) * NA + A) * DP;
#pragma unroll
      for (int j = 0; j < DP; ++j) out[j][r] = row[j];
    }
  } else if (vec[A].ptr != nullptr) {
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const T *row = vec[A].ptr + (int64_t)brow[r] * vec[A].sb;
#pragma unroll
      for (int j = 0; j < DP; ++j) out[j][r] = row[min(j, vec[A].len - 1)];
    }
  } else {
#pragma unroll
    for (int j = 0; j < DP; ++j)
#pragma unroll
      for (int r = 0; r < PPL; ++r) out[j][r] = T(0);
  }
}

template <typename T> __device__ __forceinline__ LgRowVec<T> lg_offset_vec(const LgMap &m) {
  LgRowVec<T> v;
  v.ptr = reinterpret_cast<const T *>(m.off);
  v.sb = m.off_sb;
  v.len = m.dout;
  return v;
}

// ---- persistent tiles with register prefetch ----------------------------------------------------------
// A workgroup that loads a tile, waits, computes and stores keeps its share of HBM idle while it
// computes: with two or three workgroups per CU the loaded latency (~6 us at these rates) is not
// covered.  K9 / K10 therefore run a fixed grid of workgroups over the tiles; each holds the NEXT tile's
// 16-byte vectors in registers, issued right after the current tile went into LDS, so the loads fly
// during the arithmetic.  Barriers inside the loop wait for LDS traffic only (a full __syncthreads()
// would also drain the prefetch).
__device__ __forceinline__ void lg_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <typename T, int NV, bool LG_OPAQUE = false>
__device__ __forceinline__ void lg_prefetch(const T *__restrict__ src, uint32_t ne, int stream,
                                            typename Vec16<T>::type (&regs)[NV]) {
  using V = typename Vec16<T>::type;
  const uint32_t nvec = ne / Vec16<T>::N;
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    const uint32_t v = lg_tid() + s * kLgBlock;
    if (v < nvec) regs[s] = load16(reinterpret_cast<const V *>(src) + v, stream);
  }
}

template <typename T, int NV, bool LG_OPAQUE = false>
__device__ __forceinline__ void lg_commit(const T *__restrict__ src, uint32_t ne,
                                          const typename Vec16<T>::type (&regs)[NV], T *__restrict__ tile,
                                          const LgLayout &l) {
  constexpr int N = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const uint32_t nvec = ne / N;
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    const uint32_t v = lg_tid() + s * kLgBlock;
    if (v < nvec) reinterpret_cast<V *>(tile)[lg_slot(v, l)] = regs[s];
  }
#pragma unroll 1
  for (uint32_t e = nvec * N + threadIdx.x; e < ne; e += kLgBlock) tile[e] = src[e];   // last tile only
}

// [ne] elements, contiguous from a 16-byte aligned `src`, into a tile whose layout is FLAT (rs == d), by loads that
// write LDS directly (gfx950 global_load_lds_dwordx4: wave-uniform LDS base + lane x 16 bytes, per-lane source
// address) — no registers in between, so a block can be sent for long before it is needed; the LDS writes are
// complete once the workgroup has passed a barrier that waits for its vector-memory counter.
template <typename T>
__device__ __forceinline__ void lg_stage_rows_async(const T *__restrict__ src, uint32_t ne, T *__restrict__ tile) {
  constexpr uint32_t N = Vec16<T>::N;
  const uint32_t nvec = ne / N;
  const uint32_t wave_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~(uint32_t)(kWave - 1)));
  const uint32_t lane = threadIdx.x & (uint32_t)(kWave - 1);
#pragma unroll 1
  for (uint32_t v0 = wave_base; v0 < nvec; v0 += kLgBlock) {      // wave-uniform trip count
    const uint32_t v = v0 + lane;
#if defined(__HIP_DEVICE_COMPILE__)      /* (the builtin exists in the device pass only; the host pass never runs this body) */
    if (v < nvec)
      __builtin_amdgcn_global_load_lds(src + (size_t)v * N, (__attribute__((address_space(3))) void *)(tile + (size_t)v0 * N),
                                       16, 0, 0);
#else
    (void)v;
#endif
  }
#pragma unroll 1
  for (uint32_t e = nvec * N + threadIdx.x; e < ne; e += kLgBlock) tile[e] = src[e];      // the array's last rows only
}

// ---- rows fetched through ancestor indices -------------------------------------------------------------
// The resampled latent  x_{t-1}[b, idx[b,k], :]  (aesmc/inference.py:102-111, state.py:179) need not exist in
// HBM for a kernel that stages x_{t-1} through LDS anyway: every lane fetches the rows of ITS OWN particles
// (tid + 256 r) from where their ancestors lie — the lane loads its particles' indices itself, one tile
// earlier still, so no index ever goes through LDS — in pieces of PB bytes (16 when rows are whole 16-byte
// vectors, else 8, else 4: the widest unit that never straddles a row; compile time), at immediate offsets
// from one 64-bit row address per particle, and parks them in the tile at its own slot.
struct LgGather {
  const int64_t *idx;   // [B, K] ancestor indices; nullptr: no gather
  int32_t *flags;       // status word (out-of-range indices)
  uint32_t ppr;         // pieces per row
  uint32_t pb;          // bytes per piece: 4, 8 or 16
  uint32_t row_bytes;
};

template <int PB> struct LgPiece;
template <> struct LgPiece<4> { using type = uint32_t; };
template <> struct LgPiece<8> { using type = uint2; };
template <> struct LgPiece<16> { using type = uint4; };

// the raw indices of the lane's particles of the tile at n0 -> registers
template <int PPL, bool LG_OPAQUE = false>
__device__ __forceinline__ void lg_anc_prefetch(const LgGather &g, int64_t n0, uint32_t np, int64_t (&raw)[PPL]) {
#pragma unroll
  for (int r = 0; r < PPL; ++r) {
    const uint32_t q = lg_tid() + r * kLgBlock;
    raw[r] = q < np ? g.idx[n0 + q] : 0;
  }
}

// MAXQ pieces of PB bytes hold a row of the kernel's extent class; regs: PPL * MAXQ * PB / 4 words
template <int PPL, int MAXQ, int PB, bool LG_OPAQUE = false>
__device__ __forceinline__ void lg_gather_prefetch(const char *__restrict__ src, const LgGather &g, int64_t n0,
                                                   uint32_t np, uint32_t K, const int64_t (&raw)[PPL],
                                                   uint32_t (&regs)[PPL * MAXQ * (PB / 4)]) {
  using P = typename LgPiece<PB>::type;
  constexpr int W = PB / 4;
  const uint32_t b0 = (uint32_t)(n0 / K);
  const uint32_t k0 = (uint32_t)(n0 - (int64_t)b0 * K);
#pragma unroll
  for (int r = 0; r < PPL; ++r) {
    const uint32_t q = lg_tid() + r * kLgBlock;
    if (q < np) {
      int64_t a = raw[r];
      if (a < 0 || a >= (int64_t)K) {     // K2 writes K for a degenerate row (flagged there); never fault on it
        raise_flag(g.flags, AESMC_FLAG_INDEX_OUT_OF_RANGE);
        a = a < 0 ? 0 : (int64_t)K - 1;
      }
      const uint64_t row = (uint64_t)(b0 + (k0 + q) / K) * K + (uint64_t)a;
      const char *at = src + row * g.row_bytes;
#pragma unroll
      for (int c = 0; c < MAXQ; ++c) {
        if ((uint32_t)c < g.ppr) {
          const P x = *reinterpret_cast<const P *>(at + c * PB);
          __builtin_memcpy(&regs[(r * MAXQ + c) * W], &x, PB);
        }
      }
    }
  }
}

// the lane's rows into the tile: row p starts at element p * rs (16-byte aligned whenever PB == 16)
template <typename T, int PPL, int MAXQ, int PB, bool LG_OPAQUE = false>
__device__ __forceinline__ void lg_gather_commit(const LgGather &g, uint32_t np,
                                                 const uint32_t (&regs)[PPL * MAXQ * (PB / 4)], T *__restrict__ tile,
                                                 const LgLayout &l) {
  using P = typename LgPiece<PB>::type;
  constexpr int W = PB / 4;
  char *base = reinterpret_cast<char *>(tile);
  const uint32_t row_pitch = l.rs * (uint32_t)sizeof(T);
#pragma unroll
  for (int r = 0; r < PPL; ++r) {
    const uint32_t q = lg_tid() + r * kLgBlock;
    if (q < np) {
#pragma unroll
      for (int c = 0; c < MAXQ; ++c) {
        if ((uint32_t)c < g.ppr) {
          P x;
          __builtin_memcpy(&x, &regs[(r * MAXQ + c) * W], PB);
          *reinterpret_cast<P *>(base + q * row_pitch + c * PB) = x;
        }
      }
    }
  }
}

// A tile's rows through the ancestors straight into the tile (no register double buffer): for kernels that do
// not prefetch a tile ahead.  The piece width is a launch-uniform branch here (one kernel for all three).
template <typename T, int PPL, int DP, bool LG_OPAQUE = false>
__device__ __forceinline__ void lg_gather_stage(const char *__restrict__ src, const LgGather &g, int64_t n0,
                                                uint32_t np, uint32_t K, const int64_t (&raw)[PPL],
                                                T *__restrict__ tile, const LgLayout &l) {
  char *base = reinterpret_cast<char *>(tile);
  const uint32_t row_pitch = l.rs * (uint32_t)sizeof(T);
  const uint32_t b0 = (uint32_t)(n0 / K);
  const uint32_t k0 = (uint32_t)(n0 - (int64_t)b0 * K);
  constexpr int BYTES = DP * (int)sizeof(T);
#pragma unroll
  for (int r = 0; r < PPL; ++r) {
    const uint32_t q = lg_tid() + r * kLgBlock;
    if (q < np) {
      int64_t a = raw[r];
      if (a < 0 || a >= (int64_t)K) {
        raise_flag(g.flags, AESMC_FLAG_INDEX_OUT_OF_RANGE);
        a = a < 0 ? 0 : (int64_t)K - 1;
      }
      const uint64_t row = (uint64_t)(b0 + (k0 + q) / K) * K + (uint64_t)a;
      const char *at = src + row * g.row_bytes;
      char *to = base + q * row_pitch;
      if (g.pb == 16) {
        uint4 x[(BYTES + 15) / 16];
#pragma unroll
        for (int c = 0; c < (BYTES + 15) / 16; ++c)
          if ((uint32_t)c < g.ppr) x[c] = *reinterpret_cast<const uint4 *>(at + c * 16);
#pragma unroll
        for (int c = 0; c < (BYTES + 15) / 16; ++c)
          if ((uint32_t)c < g.ppr) *reinterpret_cast<uint4 *>(to + c * 16) = x[c];
      } else if (g.pb == 8) {
        uint2 x[(BYTES + 7) / 8];
#pragma unroll
        for (int c = 0; c < (BYTES + 7) / 8; ++c)
          if ((uint32_t)c < g.ppr) x[c] = *reinterpret_cast<const uint2 *>(at + c * 8);
#pragma unroll
        for (int c = 0; c < (BYTES + 7) / 8; ++c)
          if ((uint32_t)c < g.ppr) *reinterpret_cast<uint2 *>(to + c * 8) = x[c];
      } else {
        uint32_t x[BYTES / 4];
#pragma unroll
        for (int c = 0; c < BYTES / 4; ++c)
          if ((uint32_t)c < g.ppr) x[c] = *reinterpret_cast<const uint32_t *>(at + c * 4);
#pragma unroll
        for (int c = 0; c < BYTES / 4; ++c)
          if ((uint32_t)c < g.ppr) *reinterpret_cast<uint32_t *>(to + c * 4) = x[c];
      }
    }
  }
}

static inline LgGather lg_gather(const int64_t *idx, int32_t *flags, size_t row_bytes) {
  LgGather g;
  g.idx = idx;
  g.flags = flags;
  g.pb = row_bytes % 16 == 0 ? 16u : row_bytes % 8 == 0 ? 8u : 4u;
  g.ppr = (uint32_t)(row_bytes / g.pb);
  g.row_bytes = (uint32_t)row_bytes;
  return g;
}

// ---- host side ---------------------------------------------------------------------------------
static inline bool lg_map_ok(const aesmc_affine_map *m) {
  return m != nullptr && m->weight != nullptr && m->dout >= 1 && m->din >= 1 && m->dout <= kLgMaxDim &&
         m->din <= kLgMaxDim;
}
static inline LgMap lg_map(const aesmc_affine_map *m) {
  LgMap out;
  out.w = m->weight; out.sj = m->stride_out; out.si = m->stride_in;
  out.off = m->offset; out.off_sb = m->offset_stride_b;
  out.dout = (int32_t)m->dout; out.din = (int32_t)m->din;
  return out;
}
// compile-time extents the kernels are built for: the smallest one that holds d (10 is there for the
// BASELINE shapes: padding 10 to 12 costs 44 % more multiply-adds)
static inline int lg_pad_dim(int64_t d) { return d <= 4 ? 4 : d <= 8 ? 8 : d <= 10 ? 10 : d <= 12 ? 12 : 16; }
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// PPL = 2 only while the tiles fit 64 KiB (two workgroups per CU); one particle per lane may take up to 144 KiB
constexpr size_t kLgLdsBudget = 64 * 1024;

// Workgroups of a persistent launch: as many as are resident at once (by LDS; at most 8 per CU), so
// each walks tiles blockIdx.x, blockIdx.x + grid, ... with the next one prefetched.
static inline int lg_cu_count() {
  // per device, asked once (a launch per timestep must not pay two runtime calls for a constant)
  static int cached[64] = {};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) return 256;
  if (cached[device] == 0) {
    int value = 0;
    cached[device] = (hipDeviceGetAttribute(&value, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
                      value > 0) ? value : 256;
  }
  return cached[device];
}

static inline unsigned lg_persistent_grid(int64_t tiles, size_t lds_bytes, int max_per_cu = 8) {
  const int cus = lg_cu_count();
  int per_cu = (int)((size_t)160 * 1024 / (lds_bytes > 0 ? lds_bytes : 1));
  per_cu = per_cu < 1 ? 1 : (per_cu > max_per_cu ? max_per_cu : per_cu);
  const int64_t resident = (int64_t)cus * per_cu;
  return (unsigned)(tiles < resident ? tiles : resident);
}
constexpr size_t kLgLdsLimit = 144 * 1024;

// Launches `KERNEL<T, DP, PPL>` with DP from `dp` (4, 8, 12, 16) and PPL from `ppl` (1, 2).  Tiles beyond
// 64 KiB of LDS (float64 rows of 10 and more values) need the opt-in; it is per kernel and per device,
// cheap, and only taken for those shapes.
// (the opt-in is raised to the limit once per kernel and device — `lg_raise_lds_limit` remembers — and a failure
// surfaces as a launch error of that launch)
static inline bool lg_raise_lds_limit(const void *kernel, bool (&done)[64]) {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) return false;
  if (!done[device]) {
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    done[device] = true;
  }
  return true;
}
#define LG_LAUNCH(KERNEL, T, DP_, PPL_, grid, lds, stream, ...)                                              \
  do {                                                                                                       \
    bool lds_ok = true;                                                                                      \
    if ((lds) > 64 * 1024) {                                                                                 \
      static bool raised[64] = {};                                                                           \
      lds_ok = lg_raise_lds_limit(reinterpret_cast<const void *>(KERNEL<T, DP_, PPL_>), raised);             \
    }                                                                                                        \
    /* (an empty grid is an invalid launch: the caller's hipGetLastError() reports AESMC_ERR_LAUNCH) */      \
    hipLaunchKernelGGL((KERNEL<T, DP_, PPL_>), lds_ok ? dim3(grid) : dim3(0), dim3(kLgBlock), lds, stream,   \
                       __VA_ARGS__);                                                                         \
  } while (0)
#ifdef AESMC_LG_FAST_BUILD   /* experiments only: one extent, so a translation unit compiles in seconds */
#define LG_DISPATCH(KERNEL, T, dp, ppl, grid, lds, stream, ...)                                              \
  do {                                                                                                       \
    if constexpr (sizeof(T) == 4) {                                                                          \
      if (ppl == 2) LG_LAUNCH(KERNEL, T, 10, 2, grid, lds, stream, __VA_ARGS__);                             \
      else LG_LAUNCH(KERNEL, T, 10, 1, grid, lds, stream, __VA_ARGS__);                                      \
    }                                                                                                        \
  } while (0)
#else
#define LG_DISPATCH(KERNEL, T, dp, ppl, grid, lds, stream, ...)                                              \
  do {                                                                                                       \
    if (ppl == 2) {                                                                                          \
      switch (dp) {                                                                                          \
        case 4: LG_LAUNCH(KERNEL, T, 4, 2, grid, lds, stream, __VA_ARGS__); break;                           \
        case 8: LG_LAUNCH(KERNEL, T, 8, 2, grid, lds, stream, __VA_ARGS__); break;                           \
        case 10: LG_LAUNCH(KERNEL, T, 10, 2, grid, lds, stream, __VA_ARGS__); break;                         \
        case 12: LG_LAUNCH(KERNEL, T, 12, 2, grid, lds, stream, __VA_ARGS__); break;                         \
        default: LG_LAUNCH(KERNEL, T, 16, 2, grid, lds, stream, __VA_ARGS__); break;                         \
      }                                                                                                      \
    } else {                                                                                                 \
      switch (dp) {                                                                                          \
        case 4: LG_LAUNCH(KERNEL, T, 4, 1, grid, lds, stream, __VA_ARGS__); break;                           \
        case 8: LG_LAUNCH(KERNEL, T, 8, 1, grid, lds, stream, __VA_ARGS__); break;                           \
        case 10: LG_LAUNCH(KERNEL, T, 10, 1, grid, lds, stream, __VA_ARGS__); break;                         \
        case 12: LG_LAUNCH(KERNEL, T, 12, 1, grid, lds, stream, __VA_ARGS__); break;                         \
        default: LG_LAUNCH(KERNEL, T, 16, 1, grid, lds, stream, __VA_ARGS__); break;                         \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)
#endif

// With 512-particle tiles a launch of fewer than ~1M particles leaves each CU with at most two or three
// workgroups of one tile each: all latency.  256-particle tiles double the workgroups.
static inline bool lg_few_tiles(int64_t N) { return N < ((int64_t)1 << 20); }

// AESMC_LG_FWD_PPL=1: one particle per lane in K9 / K10 whatever the size (a measurement knob).
static inline int lg_forward_ppl() {
  static const int v = [] { const char *e = measurement_knob("AESMC_LG_FWD_PPL"); return e != nullptr ? atoi(e) : 0; }();
  return v;
}

// Batch rows a tile of `tp` consecutive particles can span.
static inline int64_t lg_rows_spanned(int64_t tp, int64_t K) { return (tp - 1) / K + 2; }

}  // namespace aesmc
