// K8 - K13: particle propagation for linear-Gaussian model terms (and a learned proposal net).
//
// The reference's own state-space model (test/models/lgssm.py:40, :52, :74) and every LGSSM written
// against its callable contract (aesmc/inference.py:20-46) build each step's distributions as
//     Normal(loc = W x + c, scale)            x = previous_latents[-1] or latents[-1]  [B,K,d]
// with one small matrix per term.  Through PyTorch that is, per timestep, three skinny matmuls
// ([B*K, d] x [d, d]: 2 x 4 d bytes per particle each for ~2 d^2 flops — pure HBM traffic), a
// broadcast add, and then `state.sample` / `state.log_prob` (aesmc/state.py:61-155) reading the
// materialised locations back: at B=1024 K=4096 d=10 more than half of a step's device time.
//
// Here the location is never written to HBM.  One lane owns one particle; the tile's rows are staged
// through LDS with 16-byte loads, the (at most 16 x 16) matrices sit zero-padded in LDS and are read
// as broadcasts, and every location element is ONE chain of fused multiply-adds in a fixed order
//     loc[j] = fma(W[j][d-1], x[d-1], ... fma(W[j][1], x[1], fma(W[j][0], x[0], c[j])) ...)
// — the same chain in every kernel of this file, so a location is the same bit for bit whether K9 / K10 /
// K12 evaluate it in passing or K8 materialises it, and oracle/smc_core.c restates it with fma().
//
//   K8  aesmc_particle_affine            out = base + (c + W1 x1 + W2 x2)        (materialise / adjoint)
//   K9  aesmc_affine_normal_rsample      x' = (c + W x) + eps * scale           (state.sample)
//   K10 aesmc_affine_normal_logweight    log N(x'; A x + a, s_p) + log N(y; C x' + g, s_g)
//                                        - log N(x'; Q x + q, s_q)              (inference.py:112-126)
//   K11 aesmc_particle_affine_backward   grad W, grad x of a location           (weight gradients: MFMA)
//   K12 aesmc_affine_normal_logweight_backward   K10's backward in one pass
//   K13 aesmc_particle_mlp               b2 + W2 tanh(c1 + W1 x)                (a learned proposal net)
//
// After the location, K9 follows K6 operation for operation (product rounded before the sum); K10 takes
// PyTorch's per-element log-density with the common factors out of the d-sum (one division per term:
// see the kernel) and combines the terms as (p + g) - q, as K5 does.
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

// ---- K8 ----------------------------------------------------------------------------------------
template <typename T, int DP, int PPL>
__global__ __launch_bounds__(kLgBlock) void particle_affine_kernel(const T *__restrict__ x1, LgMap m1,
                                                                    const T *__restrict__ x2, LgMap m2,
                                                                    const T *__restrict__ base, T *__restrict__ out,
                                                                    int64_t N, uint32_t K, int stream) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lg_smem[];
  constexpr uint32_t TP = kLgBlock * PPL;
  const uint32_t d1 = m1.din, d2 = x2 != nullptr ? m2.din : 0, dout = m1.dout;
  T *w1 = reinterpret_cast<T *>(lg_smem);
  T *w2 = w1 + DP * DP;
  T *t1 = w2 + DP * DP;
  const LgLayout l1 = lg_layout<T>(d1), l2 = lg_layout<T>(d2), lo = lg_layout<T>(dout);
  T *t2 = t1 + (TP * l1.rs + 16);
  T *to = t2 + (TP * l2.rs + 16);
  const int64_t n0 = (int64_t)blockIdx.x * TP;
  const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
  lg_stage_weight<T, DP>(m1, w1);
  lg_stage_rows(x1 + n0 * d1, np * d1, t1, l1, stream);
  if (x2 != nullptr) {
    lg_stage_weight<T, DP>(m2, w2);
    lg_stage_rows(x2 + n0 * d2, np * d2, t2, l2, stream);
  }
  if (base != nullptr) lg_stage_rows(base + n0 * dout, np * dout, to, lo, stream);
  uint32_t p[PPL], brow[PPL], at[PPL];
  bool live[PPL];
  lg_rows<PPL>(n0, np, K, p, live, brow);
  T acc[DP][PPL];
  lg_offsets<T, DP, PPL>(m1, brow, acc);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < PPL; ++r) at[r] = p[r] * l1.rs;
  lg_apply_tile<T, DP, PPL>(w1, t1, at, (int)d1, acc);
  if (x2 != nullptr) {
#pragma unroll
    for (int r = 0; r < PPL; ++r) at[r] = p[r] * l2.rs;
    lg_apply_tile<T, DP, PPL>(w2, t2, at, (int)d2, acc);
  }
#pragma unroll
  for (int j = 0; j < DP; ++j) {
    if ((uint32_t)j < dout) {
  #pragma unroll
      for (int r = 0; r < PPL; ++r) {
        if (live[r]) {
          const uint32_t slot = p[r] * lo.rs + j;
          to[slot] = base != nullptr ? to[slot] + acc[j][r] : acc[j][r];
        }
      }
    }
  }
  __syncthreads();
  lg_store_rows(out + n0 * dout, np * dout, to, lo);
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

// ---- K9 ----------------------------------------------------------------------------------------
template <typename T, int DP, int PPL, bool TAB>
__global__ __launch_bounds__(kLgBlock, 3) void affine_rsample_kernel(const T *__restrict__ src, LgMap m,
                                                                   const T *__restrict__ eps,
                                                                   const T *__restrict__ scale_ptr,
                                                                   T *__restrict__ out, int64_t N, uint32_t K,
                                                                   int stream) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lg_smem[];
  constexpr uint32_t TP = kLgBlock * PPL;
  constexpr int NV = (PPL * DP + Vec16<T>::N - 1) / Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const uint32_t din = m.din, dout = m.dout;
  T *wl = reinterpret_cast<T *>(lg_smem);
  T *tab = wl + DP * DP;                             // [kLgRowsMax][1][DP]
  T *ts = tab + kLgRowsMax * DP;
  const LgLayout ls = lg_layout<T>(din), le = lg_layout<T>(dout);
  T *te = ts + (TP * ls.rs + 16);   // the noise, then the draw in its place
  const T scale = scale_ptr[0];
  lg_stage_weight<T, DP>(m, wl);
  const LgRowVec<T> vec[1] = {lg_offset_vec<T>(m)};
  const int64_t tiles = (N + TP - 1) / TP;
  V rs[NV], re[NV];
  {
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
    lg_prefetch<T, NV>(src + n0 * din, np * din, 0, rs);
    lg_prefetch<T, NV>(eps + n0 * dout, np * dout, stream, re);
  }
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TP;
    const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
    lg_commit<T, NV>(src + n0 * din, np * din, rs, ts, ls);
    lg_commit<T, NV>(eps + n0 * dout, np * dout, re, te, le);
    uint32_t p[PPL], brow[PPL], at[PPL];
    bool live[PPL];
    lg_rows<PPL>(n0, np, K, p, live, brow);
    const uint32_t b0 = (uint32_t)(n0 / K), nrows = (uint32_t)((n0 + np - 1) / K) - b0 + 1;
    constexpr bool use_tab = TAB;      // the host guarantees nrows <= kLgRowsMax when it picks TAB
    T acc[DP][PPL];
    if (use_tab) lg_stage_table<T, DP, 1>(vec, b0, nrows, tab);
    else lg_row_values<T, DP, PPL, 1, 0>(vec, false, tab, b0, brow, acc);
    lg_lds_barrier();
    const int64_t next = tile + gridDim.x;
    if (next < tiles) {
      const int64_t m0 = next * TP;
      const uint32_t mp = (uint32_t)min((int64_t)TP, N - m0);
      lg_prefetch<T, NV>(src + m0 * din, mp * din, 0, rs);
      lg_prefetch<T, NV>(eps + m0 * dout, mp * dout, stream, re);
    }
    if (use_tab) lg_row_values<T, DP, PPL, 1, 0>(vec, true, tab, b0, brow, acc);
#pragma unroll
    for (int r = 0; r < PPL; ++r) at[r] = p[r] * ls.rs;
    lg_apply_tile<T, DP, PPL>(wl, ts, at, (int)din, acc);
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      if ((uint32_t)j < dout) {
#pragma unroll
        for (int r = 0; r < PPL; ++r) {
          if (live[r]) {
            const uint32_t slot = p[r] * le.rs + j;
            te[slot] = acc[j][r] + te[slot] * scale;   // the product rounded before the sum, as K6
          }
        }
      }
    }
    lg_lds_barrier();
    lg_store_rows(out + n0 * dout, np * dout, te, le);
    lg_lds_barrier();
  }
}

// ---- K10 ---------------------------------------------------------------------------------------
// Per term: q = sum_j (v_j - loc_j)^2 as one fma chain from 0 (j ascending), then
//   log N = (-q) / (2 sigma^2) - d (log sigma + log sqrt(2 pi))
// — ONE division per term and particle instead of PyTorch's one per element (K5 keeps those: it is
// HBM-bound either way; this kernel would be VALU-bound on 2 d divisions per particle).
// DRAW (K15): `x` holds the proposal's NOISE instead of x_t; the kernel forms the draw
//   x_t = loc_q + eps * scale_q        (K9's arithmetic on K9's chain: the same bits)
// from the proposal location it computes anyway, writes it to `out_x` through the tile and weighs it —
// one pass over x_{t-1} and the noise instead of K9's and K10's two passes over three arrays.
template <typename T, int DP, int PPL, bool TAB, bool PREFETCH, bool DRAW = false>
__global__ __launch_bounds__(kLgBlock, 3) void affine_logweight_kernel(
    const T *__restrict__ xprev, const T *__restrict__ x, const T *__restrict__ y, int64_t y_sb, LgMap mp, LgMap mg,
    LgMap mq, const T *__restrict__ sp_ptr, const T *__restrict__ sg_ptr, const T *__restrict__ sq_ptr,
    T *__restrict__ out_lw, int64_t N, uint32_t K, T *__restrict__ out_x) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lg_smem[];
  constexpr uint32_t TP = kLgBlock * PPL;
  constexpr int NV = (PPL * DP + Vec16<T>::N - 1) / Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const uint32_t dx = mp.dout, dy = mg.dout;
  T *wp = reinterpret_cast<T *>(lg_smem);
  T *wg = wp + DP * DP;
  T *wq = wg + DP * DP;
  T *tab = wq + DP * DP;                             // [kLgRowsMax][4][DP]: offsets p, q, g and the observation
  T *tprev = tab + kLgRowsMax * 4 * DP;
  const LgLayout lx = lg_layout<T>(dx);
  T *tx = tprev + (TP * lx.rs + 16);
  const T s_p = sp_ptr[0], s_g = sg_ptr[0], s_q = sq_ptr[0];
  lg_stage_weight<T, DP>(mp, wp);
  lg_stage_weight<T, DP>(mg, wg);
  lg_stage_weight<T, DP>(mq, wq);
  LgRowVec<T> vec[4] = {lg_offset_vec<T>(mp), lg_offset_vec<T>(mq), lg_offset_vec<T>(mg), {y, y_sb, (int)dy}};
  const T half_log_2pi = LgConst<T>::half_log_2pi();
  const T two_var_p = T(2) * (s_p * s_p), const_p = T(dx) * (Num<T>::log(s_p) + half_log_2pi);
  const T two_var_g = T(2) * (s_g * s_g), const_g = T(dy) * (Num<T>::log(s_g) + half_log_2pi);
  const T two_var_q = T(2) * (s_q * s_q), const_q = T(dx) * (Num<T>::log(s_q) + half_log_2pi);
  const int64_t tiles = (N + TP - 1) / TP;
  V rp[PREFETCH ? NV : 1], rx[PREFETCH ? NV : 1];
  if constexpr (PREFETCH) {
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const uint32_t ne = (uint32_t)min((int64_t)TP, N - n0) * dx;
    lg_prefetch<T, NV>(xprev + n0 * dx, ne, 0, rp);
    lg_prefetch<T, NV>(x + n0 * dx, ne, 0, rx);
  }
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TP;
    const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
    if constexpr (PREFETCH) {
      lg_commit<T, NV>(xprev + n0 * dx, np * dx, rp, tprev, lx);
      lg_commit<T, NV>(x + n0 * dx, np * dx, rx, tx, lx);
    } else {
      lg_stage_rows(xprev + n0 * dx, np * dx, tprev, lx, 0);
      lg_stage_rows(x + n0 * dx, np * dx, tx, lx, 0);
    }
    uint32_t p[PPL], brow[PPL], at[PPL];
    bool live[PPL];
    lg_rows<PPL>(n0, np, K, p, live, brow);
    const uint32_t b0 = (uint32_t)(n0 / K), nrows = (uint32_t)((n0 + np - 1) / K) - b0 + 1;
    constexpr bool use_tab = TAB;      // the host guarantees nrows <= kLgRowsMax when it picks TAB
    T locp[DP][PPL], locq[DP][PPL], locg[DP][PPL], yv[DP][PPL];
    if (use_tab) {
      lg_stage_table<T, DP, 4>(vec, b0, nrows, tab);
    } else {    // tiny K: the global loads go out before the prefetch, so waiting for them does not drain it
      lg_row_values<T, DP, PPL, 4, 0>(vec, false, tab, b0, brow, locp);
      lg_row_values<T, DP, PPL, 4, 1>(vec, false, tab, b0, brow, locq);
      lg_row_values<T, DP, PPL, 4, 2>(vec, false, tab, b0, brow, locg);
      lg_row_values<T, DP, PPL, 4, 3>(vec, false, tab, b0, brow, yv);
    }
    lg_lds_barrier();
    if constexpr (PREFETCH) {
      const int64_t next = tile + gridDim.x;
      if (next < tiles) {
        const int64_t m0 = next * TP;
        const uint32_t ne = (uint32_t)min((int64_t)TP, N - m0) * dx;
        lg_prefetch<T, NV>(xprev + m0 * dx, ne, 0, rp);
        lg_prefetch<T, NV>(x + m0 * dx, ne, 0, rx);
      }
    }
    if (use_tab) {
      lg_row_values<T, DP, PPL, 4, 0>(vec, true, tab, b0, brow, locp);
      lg_row_values<T, DP, PPL, 4, 1>(vec, true, tab, b0, brow, locq);
    }
#pragma unroll
    for (int r = 0; r < PPL; ++r) at[r] = p[r] * lx.rs;
    // transition and proposal locations from x_prev, one pass over its elements
#pragma unroll
    for (int i = 0; i < DP; ++i) {
      if ((uint32_t)i < dx) {
        T xv[PPL];
#pragma unroll
        for (int r = 0; r < PPL; ++r) xv[r] = tprev[at[r] + i];
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          const T a = wp[i * DP + j], q = wq[i * DP + j];
#pragma unroll
          for (int r = 0; r < PPL; ++r) {
            locp[j][r] = fma_t(a, xv[r], locp[j][r]);
            locq[j][r] = fma_t(q, xv[r], locq[j][r]);
          }
        }
      }
    }
    // squared distances of x to both; x kept for the emission map
    T xx[DP][PPL], qp[PPL], qq[PPL], qg[PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r) qp[r] = qq[r] = qg[r] = T(0);
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      if ((uint32_t)j < dx) {
#pragma unroll
        for (int r = 0; r < PPL; ++r) {
          if constexpr (DRAW) {
            xx[j][r] = locq[j][r] + tx[at[r] + j] * s_q;      // the product rounded before the sum, as K9 / K6
            if (live[r]) tx[at[r] + j] = xx[j][r];             // the lane's own row: the draw leaves through the tile
          } else {
            xx[j][r] = tx[at[r] + j];
          }
          const T dp = xx[j][r] - locp[j][r], dq = xx[j][r] - locq[j][r];
          qp[r] = fma_t(dp, dp, qp[r]);
          qq[r] = fma_t(dq, dq, qq[r]);
        }
      }
    }
    // emission location from x
    if (use_tab) lg_row_values<T, DP, PPL, 4, 2>(vec, true, tab, b0, brow, locg);
#pragma unroll
    for (int i = 0; i < DP; ++i) {
      if ((uint32_t)i < dx) {
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          const T c = wg[i * DP + j];
#pragma unroll
          for (int r = 0; r < PPL; ++r) locg[j][r] = fma_t(c, xx[i][r], locg[j][r]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      if ((uint32_t)j < dy) {
#pragma unroll
        for (int r = 0; r < PPL; ++r) {
          const T yj = use_tab ? tab[((brow[r] - b0) * 4 + 3) * DP + j] : yv[j][r];
          const T dg = yj - locg[j][r];
          qg[r] = fma_t(dg, dg, qg[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      if (live[r]) {
        const T lp = (-qp[r]) / two_var_p - const_p;
        const T lg = (-qg[r]) / two_var_g - const_g;
        const T lq = (-qq[r]) / two_var_q - const_q;
        out_lw[n0 + p[r]] = (lp + lg) - lq;
      }
    }
    lg_lds_barrier();     // every lane is done with the tiles before the next commit
    if constexpr (DRAW) {
      lg_store_rows(out_x + n0 * dx, np * dx, tx, lx);
      lg_lds_barrier();
    }
  }
}

// ---- K13: a learned proposal net over the particles ----------------------------------------------------
//   out[b,k,:] = b2 + W2 tanh( c1[b,:] + W1 x[b,k,:] )          W1 [H, din], W2 [dout, H], H <= 64
// — the two-layer tanh MLP of [x_{t-1}, y_t] that BASELINE.json's nonlinear state-space model uses as its
// proposal (the y_t part of the first layer and its bias arrive as the per-row offset c1).  Through
// PyTorch this is a concatenation, two GEMMs and a tanh with [B,K,H] round trips through HBM between
// them; here the hidden layer lives in registers (one particle per lane, H accumulators), both weight
// matrices in LDS.  Chains as everywhere in this file: fused multiply-adds, inputs ascending, started
// from the offset / bias; tanh is the device library's (the one torch.tanh calls).
template <typename T> __device__ __forceinline__ T lg_tanh(T x);
template <> __device__ __forceinline__ float lg_tanh<float>(float x) { return ::tanhf(x); }
template <> __device__ __forceinline__ double lg_tanh<double>(double x) { return ::tanh(x); }

template <typename T, int DP>
__global__ __launch_bounds__(kLgBlock) void particle_mlp_kernel(const T *__restrict__ x, LgMap m1, LgMap m2,
                                                                 T *__restrict__ out, int64_t N, uint32_t K,
                                                                 uint32_t HP) {
  // HP: the hidden width rounded up to a multiple of 16; the hidden layer is taken 16 units at a time
  // (first-layer chains, tanh, their share of the second-layer chains), so a lane holds 16 hidden values
  // and the output accumulators, whatever H is
  extern __shared__ __attribute__((aligned(16))) unsigned char lg_smem[];
  constexpr uint32_t TP = kLgBlock;
  const uint32_t din = m1.din, hid = m1.dout, dout = m2.dout;
  T *w1 = reinterpret_cast<T *>(lg_smem);         // [DP][HP]: w1[i * HP + h] = W1[h][i]
  T *w2 = w1 + DP * HP;                           // [HP][DP]: w2[h * DP + o] = W2[o][h]
  T *tab = w2 + HP * DP;                          // [kLgRowsMax][HP]: the rows' first-layer offsets
  T *tx = tab + kLgRowsMax * HP;
  const LgLayout lx = lg_layout<T>(din), lo = lg_layout<T>(dout);
  T *to = tx + (TP * lx.rs + 16);
  {
    const T *a = reinterpret_cast<const T *>(m1.w), *b = reinterpret_cast<const T *>(m2.w);
#pragma unroll 1
    for (uint32_t e = threadIdx.x; e < DP * HP; e += kLgBlock) {
      const uint32_t i = e / HP, h = e - i * HP;
      w1[e] = (h < hid && i < din) ? a[(int64_t)h * m1.sj + (int64_t)i * m1.si] : T(0);
    }
#pragma unroll 1
    for (uint32_t e = threadIdx.x; e < HP * DP; e += kLgBlock) {
      const uint32_t h = e / DP, o = e - h * DP;
      w2[e] = (h < hid && o < dout) ? b[(int64_t)o * m2.sj + (int64_t)h * m2.si] : T(0);
    }
  }
  const T *off1 = reinterpret_cast<const T *>(m1.off), *off2 = reinterpret_cast<const T *>(m2.off);
  const int64_t tiles = (N + TP - 1) / TP;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TP;
    const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
    lg_stage_rows(x + n0 * din, np * din, tx, lx, 0);
    const uint32_t b0 = (uint32_t)(n0 / K), nrows = (uint32_t)((n0 + np - 1) / K) - b0 + 1;
#pragma unroll 1
    for (uint32_t idx = threadIdx.x; idx < nrows * HP; idx += kLgBlock) {     // the host guarantees nrows <= kLgRowsMax
      const uint32_t row = idx / HP, h = idx - row * HP;
      tab[idx] = (off1 != nullptr && h < hid) ? off1[(int64_t)(b0 + row) * m1.off_sb + h] : T(0);
    }
    __syncthreads();
    const uint32_t q = threadIdx.x;
    const bool live = q < np;
    const uint32_t p = live ? q : 0u;
    const uint32_t k0 = (uint32_t)(n0 - (int64_t)b0 * K);
    const T *orow = tab + ((k0 + p) / K) * HP;
    const uint32_t base = p * lx.rs;
    T acc[DP];
#pragma unroll
    for (int o = 0; o < DP; ++o) acc[o] = (off2 != nullptr && (uint32_t)o < dout) ? off2[o] : T(0);
#pragma unroll 1
    for (uint32_t c = 0; c < HP; c += 16) {
      T hidden[16];
#pragma unroll
      for (int h = 0; h < 16; ++h) hidden[h] = orow[c + h];
#pragma unroll 2
      for (uint32_t i = 0; i < din; ++i) {
        const T xv = tx[base + i];
        const T *w = w1 + i * HP + c;
#pragma unroll
        for (int h = 0; h < 16; ++h) hidden[h] = fma_t(w[h], xv, hidden[h]);
      }
#pragma unroll
      for (int h = 0; h < 16; ++h) hidden[h] = lg_tanh<T>(hidden[h]);
#pragma unroll
      for (int h = 0; h < 16; ++h) {
        const T *w = w2 + (c + h) * DP;
#pragma unroll
        for (int o = 0; o < DP; ++o) acc[o] = fma_t(w[o], hidden[h], acc[o]);
      }
    }
    if (live) {
#pragma unroll
      for (int o = 0; o < DP; ++o)
        if ((uint32_t)o < dout) to[p * lo.rs + o] = acc[o];
    }
    __syncthreads();
    lg_store_rows(out + n0 * dout, np * dout, to, lo);
    __syncthreads();
  }
}

// ---- per-batch-row sums of a tile (offset gradients) -------------------------------------------------------
// The gradient of an offset c[b] is the sum over the row's particles of the gradient of the location.
// A tile spans the batch rows b0 .. b0 + nrows - 1 (nrows <= kLgRowsMax: the callers' table condition);
// 16 lane groups sum 1/16 of the tile's particles each, column by column, flushing at row boundaries,
// then one lane per (row slot, column) adds the 16 partials in order and leaves the tile's record
//   out[slot * 16 + j]                                   (fixed order: reproducible)
// for a second launch to add up the few tiles that cover each batch row.  `part` holds 16 x 8 x 16 values.
constexpr int kLgRowPart = 16 * kLgRowsMax * 16;
template <typename T>
__device__ __forceinline__ void lg_row_sums(const T *__restrict__ tile, uint32_t rs, uint32_t d, uint32_t np,
                                            uint32_t k0, uint32_t K, T *__restrict__ part, T *__restrict__ out) {
  const uint32_t t = lg_tid_impl<true>(), j = t & 15u, c = t >> 4;
#pragma unroll 1
  for (uint32_t i = t; i < (uint32_t)kLgRowPart; i += kLgBlock) part[i] = T(0);
  lg_lds_barrier();
  const uint32_t chunk = (np + 15) / 16, p0 = c * chunk, p1 = min(np, p0 + chunk);
  if (j < d && p0 < p1) {
    uint32_t slot = (k0 + p0) / K;
    uint32_t next = (slot + 1) * K - k0;        // first particle of the next batch row, tile-relative
    T acc = T(0);
#pragma unroll 1
    for (uint32_t p = p0; p < p1; ++p) {
      if (p >= next) {
        part[(c * kLgRowsMax + slot) * 16 + j] = acc;
        acc = T(0);
        ++slot;
        next += K;
      }
      acc += tile[p * rs + j];
    }
    part[(c * kLgRowsMax + slot) * 16 + j] = acc;
  }
  lg_lds_barrier();
  if (t < (uint32_t)kLgRowsMax * 16) {
    const uint32_t slot = t >> 4;
    T sum = T(0);
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) sum += part[(cc * kLgRowsMax + slot) * 16 + j];
    out[t] = sum;
  }
}

// ---- K11: the adjoint of an affine location ----------------------------------------------------------
// The weight gradient  dW[j][i] = sum over particles of g[p][j] x[p][i]  is a contraction over the
// particle index: it runs on the matrix cores (v_mfma_*_16x16x4: A = 4 particles x 16 values of g,
// B = 4 particles x 16 values of x, f32 / f64 inputs and accumulation — exact IEEE fma chains), which
// keeps the 16 x 16 accumulator in four registers per lane instead of d^2 per particle-owning lane.
template <typename T> struct Mfma;
template <> struct Mfma<float> {
  typedef float Acc __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ Acc fma(float a, float b, Acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};
template <> struct Mfma<double> {
  typedef double Acc __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ Acc fma(double a, double b, Acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};

constexpr int kLgRecord = 256;        // one 16 x 16 partial per matrix and workgroup
constexpr int kLgMaxGrid = 1024;      // most persistent workgroups of the reducing kernels (one record each)

// acc[j][i] += sum_{p < np} tg[p][j] * tx[p][i] over a staged tile; the four wavefronts take particles
// 4 w .. 4 w + 3 of every group of 16 (fixed assignment: the sums are reproducible).
template <typename T>
__device__ __forceinline__ void lg_outer_accumulate(const T *__restrict__ tg, uint32_t dg, const T *__restrict__ tx,
                                                    uint32_t dxx, uint32_t np, typename Mfma<T>::Acc &acc) {
  // dg, dxx: the ROW STRIDES of the two tiles (LgLayout::rs)
  constexpr bool LG_OPAQUE = true;
  // Lane (quad, col) feeds value `col` of particle 4 w + quad (+ 16 per trip).  Columns at or past a row's
  // extent read the neighbouring row: that only reaches accumulator rows / columns >= the extents,
  // which nobody reads, so there is no per-column mask; particles past the tile's end are masked.
  const uint32_t tid = lg_tid(), lane = tid & 63u, wave = tid >> 6;
  const uint32_t col = lane & 15u;
  uint32_t p = wave * 4 + (lane >> 4);
  uint32_t eg = p * dg + col, ex = p * dxx + col;
  const uint32_t step_g = 16 * dg, step_x = 16 * dxx;
  // four trips' operands are fetched before their four multiply-accumulates: the LDS latency is paid
  // once per group, not once per MFMA (the accumulator chain is sequential either way)
  for (uint32_t p0 = wave * 4; p0 < np; p0 += 64) {
    T a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool live = p + 16 * t < np;
      a[t] = live ? tg[eg + t * step_g] : T(0);
      b[t] = live ? tx[ex + t * step_x] : T(0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = Mfma<T>::fma(a[t], b[t], acc);
    p += 64;
    eg += 4 * step_g;
    ex += 4 * step_x;
  }
}

// The same sum over a wavefront's OWN 64 particles (lane = particle mapping of the one-particle-per-lane
// kernels: rows 64 w .. 64 w + 63 of the tiles were written and are read by this wavefront alone), so the
// caller needs no workgroup barrier around it — a wavefront-level fence orders its LDS writes and reads.
// ONES: the lanes of column 15 feed 1 instead of tx's (unused, extent < 16) column, so acc[j][15] gathers
// sum_p tg[p][j] — the tile's column sums at no extra pass (lg_flush_column_sums).
template <typename T, int PPL, bool ONES = false>
__device__ __forceinline__ void lg_outer_accumulate_own(const T *__restrict__ tg, uint32_t dg,
                                                        const T *__restrict__ tx, uint32_t dxx, uint32_t np,
                                                        typename Mfma<T>::Acc &acc) {
  // dg, dxx: the ROW STRIDES of the two tiles (LgLayout::rs); a wavefront's lanes own particles
  // 256 r + 64 w .. + 63 for r < PPL
  constexpr bool LG_OPAQUE = true;
  const uint32_t tid = lg_tid(), lane = tid & 63u, wave = tid >> 6;
  const uint32_t col = lane & 15u;
  const uint32_t step_g = 4 * dg, step_x = 4 * dxx;
#pragma unroll
  for (int r = 0; r < PPL; ++r) {
    uint32_t p = r * kLgBlock + wave * 64 + (lane >> 4);
    uint32_t eg = p * dg + col, ex = p * dxx + col;
    if (np == (uint32_t)(PPL * kLgBlock)) {      // a whole tile (all but the last): no particle masks
#pragma unroll
      for (int group = 0; group < 4; ++group) {
        T a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          a[t] = tg[eg + (4 * group + t) * step_g];
          b[t] = tx[ex + (4 * group + t) * step_x];
          if (ONES) b[t] = col == 15u ? T(1) : b[t];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = Mfma<T>::fma(a[t], b[t], acc);
      }
      continue;
    }
#pragma unroll
    for (int group = 0; group < 4; ++group) {
      T a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool live = p + 4 * t < np;
        a[t] = live ? tg[eg + t * step_g] : T(0);
        b[t] = live ? tx[ex + t * step_x] : T(0);
        if (ONES) b[t] = col == 15u ? T(1) : b[t];      // a particle past the end contributes a = 0
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = Mfma<T>::fma(a[t], b[t], acc);
      p += 16;
      eg += 4 * step_g;
      ex += 4 * step_x;
    }
  }
}

// A tile that lies inside one batch row needs no row bookkeeping: its column sums are what the ONES
// accumulate gathered in column 15 since the last flush.  Each wavefront writes its 16 sums to slot
// `wave` of the tile's row-sum record and clears them; the finishing launch adds the four slots (same
// test there: lg_single_row).  No barrier, no pass over the tile.
__host__ __device__ __forceinline__ bool lg_single_row(int64_t n0, uint32_t np, uint32_t K) {
  return (uint32_t)(n0 % K) + np <= K;
}
template <typename T>
__device__ __forceinline__ void lg_flush_column_sums(typename Mfma<T>::Acc &acc, T *__restrict__ record, bool keep) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((lane & 15) == 15) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (keep) record[wave * 16 + Mfma<T>::row(lane, r)] = acc[r];
      acc[r] = T(0);
    }
  }
}

__device__ __forceinline__ void lg_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// The workgroup's four partial accumulators summed (wavefront 0 .. 3 in turn) into record[0 .. 255],
// element j * 16 + i.  `scratch` holds 4 x 256 values.
template <typename T>
__device__ __forceinline__ void lg_outer_publish(const typename Mfma<T>::Acc &acc, T *__restrict__ scratch,
                                                 T *__restrict__ record) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 4; ++r) scratch[wave * 256 + Mfma<T>::row(lane, r) * 16 + (lane & 15)] = acc[r];
  __syncthreads();
  const int e = threadIdx.x;
  record[e] = ((scratch[e] + scratch[256 + e]) + scratch[512 + e]) + scratch[768 + e];
  __syncthreads();
}

template <typename T, int DP, int PPL>
__global__ __launch_bounds__(kLgBlock) void particle_affine_backward_kernel(const T *__restrict__ g,
                                                                             const T *__restrict__ x, LgMap adjoint,
                                                                             T *__restrict__ gx, T *__restrict__ ws,
                                                                             T *__restrict__ row_ws, int64_t N,
                                                                             uint32_t K, int want_w) {
  constexpr bool LG_OPAQUE = true;   // see lg_tid_impl
  extern __shared__ __attribute__((aligned(16))) unsigned char lg_smem[];
  constexpr uint32_t TP = kLgBlock * PPL;
  const uint32_t dg = adjoint.din, dxx = adjoint.dout;     // g has the location's extent, x (and gx) the input's
  T *wt = reinterpret_cast<T *>(lg_smem);
  T *scratch = wt + DP * DP;                               // 4 x 256 (records) / 16 x 8 x 16 (row sums)
  T *tg = scratch + kLgRowPart;
  const LgLayout lg = lg_layout<T>(dg), lxx = lg_layout<T>(dxx);
  T *tx = tg + (TP * lg.rs + 16);
  typename Mfma<T>::Acc acc = {T(0), T(0), T(0), T(0)};
  if (gx != nullptr) lg_stage_weight<T, DP>(adjoint, wt);
  const int64_t tiles = (N + TP - 1) / TP;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TP;
    const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
    lg_stage_rows<T, true>(g + n0 * dg, np * dg, tg, lg, 1);
    if (want_w) lg_stage_rows<T, true>(x + n0 * dxx, np * dxx, tx, lxx, 0);
    __syncthreads();
    if (row_ws != nullptr) {      // the offset's gradient: per-row sums of the incoming gradient
      const uint32_t b0 = (uint32_t)(n0 / K);
      lg_row_sums<T>(tg, lg.rs, dg, np, (uint32_t)(n0 - (int64_t)b0 * K), K, scratch,
                     row_ws + tile * (kLgRowsMax * 16));
      lg_lds_barrier();
    }
    T out[DP][PPL];
    uint32_t p[PPL], at[PPL];
    bool live[PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const uint32_t q = lg_tid() + r * kLgBlock;
      live[r] = q < np;
      p[r] = live[r] ? q : 0u;
      at[r] = p[r] * lg.rs;
    }
    if (gx != nullptr) {
#pragma unroll
      for (int j = 0; j < DP; ++j)
#pragma unroll
        for (int r = 0; r < PPL; ++r) out[j][r] = T(0);
      lg_apply_tile<T, DP, PPL>(wt, tg, at, (int)dg, out);
    }
    if (want_w) lg_outer_accumulate<T>(tg, lg.rs, tx, lxx.rs, np, acc);
    if (gx != nullptr) {
      __syncthreads();                                     // every wavefront is done reading tx
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        if ((uint32_t)j < dxx) {
#pragma unroll
          for (int r = 0; r < PPL; ++r)
            if (live[r]) tx[p[r] * lxx.rs + j] = out[j][r];
        }
      }
      __syncthreads();
      lg_store_rows<T, true>(gx + n0 * dxx, np * dxx, tx, lxx);
    }
    __syncthreads();
  }
  if (want_w) lg_outer_publish<T>(acc, scratch, ws + (int64_t)blockIdx.x * kLgRecord);
}

// ---- K12: backward of K10 in one pass ----------------------------------------------------------------
// With g = the incoming gradient of lw[b,k] (grad_lw, and / or K1's softmax term grad_lse[b] exp(lw - lse[b])
// formed here), diff_* = value - location and u_p = g diff_p / s_p^2, u_q = -g diff_q / s_q^2,
// u_g = g diff_g / s_g^2 (the gradients with respect to the three locations):
//   grad_x      = -u_p - u_q + C^T u_g          grad_x_prev = A^T u_p + Q^T u_q
//   dA = sum u_p (x) x_prev    dQ = sum u_q (x) x_prev    dC = sum u_g (x) x        (matrix cores, as K11)
//   ds_p = sum g (|diff_p|^2 / s_p^3 - dx / s_p),  ds_g likewise,  ds_q with the opposite sign
// x_prev and x are read once, the two latent gradients written once.  The three terms are taken in turn —
// location (a loop over the input elements: one column of weights live at a time), u, its adjoint, u
// through one spare LDS tile for the outer products (each wavefront over its own particles' rows:
// wavefront fences, no workgroup barriers) and, where an offset's gradient is wanted, for the per-row sums
// (lg_row_sums) — so a lane holds one u, the two latent gradients and little else.
template <typename T, int DP, int PPL>
__device__ __forceinline__ void lg_rows_to_tile(const T (&v)[DP][PPL], uint32_t d, const uint32_t (&p)[PPL],
                                                const bool (&live)[PPL], T *__restrict__ tile, const LgLayout &l) {
#pragma unroll
  for (int j = 0; j < DP; ++j) {
    if ((uint32_t)j < d) {
#pragma unroll
      for (int r = 0; r < PPL; ++r)
        if (live[r]) tile[p[r] * l.rs + j] = v[j][r];
    }
  }
}

// A wavefront's rows of the u tile are its own (lane = particle), so only a launch that also STORES the
// tile (cooperatively, all lanes) needs workgroup barriers around it.
__device__ __forceinline__ void lg_u_ready(bool stored) {
  if (stored) lg_lds_barrier();
  else lg_wave_fence();
}

struct LgBackwardOut {
  void *gxprev, *gx, *up, *ug, *uq, *ws;
  void *rows;        // nullptr, or the tiles' row-sum records [tile][3 terms p, g, q][kLgRowsMax][16]
  int row_terms;     // bit 0 / 1 / 2: term p / g / q wants its row sums
  const void *gx_in; // step kernel only: the gradient that arrives at x_t from later steps, or nullptr
  int want_scale_q;  // step kernel only: the proposal scale's gradient is wanted (costs the proposal's location)
};

template <typename T, int DP, int PPL>
__global__ __launch_bounds__(kLgBlock, (PPL == 2 || sizeof(T) == 8) ? 2 : 3) void affine_logweight_backward_kernel(
    const T *__restrict__ xprev, const T *__restrict__ x, const T *__restrict__ y, int64_t y_sb, LgMap mp, LgMap mg,
    LgMap mq, const T *__restrict__ sp_ptr, const T *__restrict__ sg_ptr, const T *__restrict__ sq_ptr,
    const T *__restrict__ lw, const T *__restrict__ lse, const T *__restrict__ grad_lse,
    const T *__restrict__ grad_lw, LgBackwardOut out, int64_t N, uint32_t K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lg_smem[];
  constexpr uint32_t TP = kLgBlock * PPL;
  const uint32_t dx = mp.dout, dy = mg.dout;
  T *wf = reinterpret_cast<T *>(lg_smem);        // [3][DP*DP] input-major: locations (p, g, q)
  T *wn = wf + 3 * DP * DP;                      // [3][DP*DP] output-major: adjoints
  T *scratch = wn + 3 * DP * DP;                 // 4 x 256 (records) / 16 x 8 x 16 (row sums)
  T *tab = scratch + kLgRowPart;                 // [kLgRowsMax][4][DP]: offsets p, q, g and the observation
  T *tprev = tab + kLgRowsMax * 4 * DP;
  const LgLayout lx = lg_layout<T>(dx), ly = lg_layout<T>(dy);
  T *tx = tprev + (TP * lx.rs + 16);
  T *tu = tx + (TP * lx.rs + 16);  // [TP * max(dx, dy)]
  {
    lg_stage_weight<T, DP>(mp, wf);
    lg_stage_weight<T, DP>(mg, wf + DP * DP);
    lg_stage_weight<T, DP>(mq, wf + 2 * DP * DP);
    LgMap t = mp;
    t.sj = mp.si; t.si = mp.sj; t.dout = mp.din; t.din = mp.dout;
    lg_stage_weight<T, DP>(t, wn);
    t = mg; t.sj = mg.si; t.si = mg.sj; t.dout = mg.din; t.din = mg.dout;
    lg_stage_weight<T, DP>(t, wn + DP * DP);
    t = mq; t.sj = mq.si; t.si = mq.sj; t.dout = mq.din; t.din = mq.dout;
    lg_stage_weight<T, DP>(t, wn + 2 * DP * DP);
  }
  const LgRowVec<T> vec[4] = {lg_offset_vec<T>(mp), lg_offset_vec<T>(mq), lg_offset_vec<T>(mg), {y, y_sb, (int)dy}};
  const T s_p = sp_ptr[0], s_g = sg_ptr[0], s_q = sq_ptr[0];
  const T inv_var_p = T(1) / (s_p * s_p), inv_var_g = T(1) / (s_g * s_g), inv_var_q = T(1) / (s_q * s_q);
  const T inv_s_p = T(1) / s_p, inv_s_g = T(1) / s_g, inv_s_q = T(1) / s_q;
  typename Mfma<T>::Acc acc_a = {T(0), T(0), T(0), T(0)}, acc_c = acc_a, acc_q = acc_a;
  T scale_acc[3] = {T(0), T(0), T(0)};
  T *gxprev = reinterpret_cast<T *>(out.gxprev), *gx = reinterpret_cast<T *>(out.gx);
  T *up_out = reinterpret_cast<T *>(out.up), *ug_out = reinterpret_cast<T *>(out.ug),
    *uq_out = reinterpret_cast<T *>(out.uq);
  T *rows = reinterpret_cast<T *>(out.rows);
  const int row_terms = rows != nullptr ? out.row_terms : 0;
  const int64_t tiles = (N + TP - 1) / TP;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TP;
    const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
    lg_stage_rows<T, true>(xprev + n0 * dx, np * dx, tprev, lx, 0);
    lg_stage_rows<T, true>(x + n0 * dx, np * dx, tx, lx, 0);
    uint32_t p[PPL], brow[PPL], at[PPL];
    bool live[PPL];
    lg_rows<PPL, true>(n0, np, K, p, live, brow);
    const uint32_t b0 = (uint32_t)(n0 / K), nrows = (uint32_t)((n0 + np - 1) / K) - b0 + 1;
    lg_stage_table<T, DP, 4, true>(vec, b0, nrows, tab);      // the host guarantees nrows <= kLgRowsMax
    const uint32_t k0_tile = (uint32_t)(n0 - (int64_t)b0 * K);
    T g[PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const int64_t n = n0 + p[r];
      T value = grad_lw != nullptr ? grad_lw[n] : T(0);
      if (grad_lse != nullptr) value = value + grad_lse[brow[r]] * Num<T>::exp(lw[n] - lse[brow[r]]);
      g[r] = live[r] ? value : T(0);
      at[r] = p[r] * lx.rs;
    }
    lg_lds_barrier();
    T u[DP][PPL], gprev[DP][PPL], gcur[DP][PPL];
    uint32_t au[PPL], ay[PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      au[r] = p[r] * lx.rs;
      ay[r] = p[r] * ly.rs;
    }
#pragma unroll
    for (int j = 0; j < DP; ++j)
#pragma unroll
      for (int r = 0; r < PPL; ++r) gprev[j][r] = T(0);
    // ---- transition term: u = g (x - loc_p) / s_p^2
    lg_row_values<T, DP, PPL, 4, 0>(vec, true, tab, b0, brow, u);
    lg_apply_loop<T, DP, PPL>(wf, tprev, at, dx, u);
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      T q = T(0);
      const T scaled = g[r] * inv_var_p;
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const T diff = (uint32_t)j < dx ? tx[at[r] + min(j, (int)dx - 1)] - u[j][r] : T(0);
        q = fma_t(diff, diff, q);
        u[j][r] = scaled * diff;
        gcur[j][r] = -u[j][r];
      }
      scale_acc[0] += g[r] * (q * inv_var_p * inv_s_p - T(dx) * inv_s_p);
    }
    lg_rows_to_tile<T, DP, PPL>(u, dx, p, live, tu, lx);
    lg_u_ready(up_out != nullptr || (row_terms & 1));
    if (up_out != nullptr) lg_store_rows<T, true>(up_out + n0 * dx, np * dx, tu, lx);
    if (row_terms & 1) lg_row_sums<T>(tu, lx.rs, dx, np, k0_tile, K, scratch, rows + (tile * 3 + 0) * (kLgRowsMax * 16));
    if (gxprev != nullptr) lg_apply_loop<T, DP, PPL>(wn, tu, au, dx, gprev);
    lg_outer_accumulate_own<T, PPL>(tu, lx.rs, tprev, lx.rs, np, acc_a);
    lg_u_ready(up_out != nullptr || (row_terms & 1));
    // ---- proposal term (enters the log-weight with a minus sign): u = -g (x - loc_q) / s_q^2
    lg_row_values<T, DP, PPL, 4, 1>(vec, true, tab, b0, brow, u);
    lg_apply_loop<T, DP, PPL>(wf + 2 * DP * DP, tprev, at, dx, u);
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      T q = T(0);
      const T scaled = g[r] * inv_var_q;
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const T diff = (uint32_t)j < dx ? tx[at[r] + min(j, (int)dx - 1)] - u[j][r] : T(0);
        q = fma_t(diff, diff, q);
        u[j][r] = -(scaled * diff);
        gcur[j][r] = gcur[j][r] - u[j][r];
      }
      scale_acc[2] -= g[r] * (q * inv_var_q * inv_s_q - T(dx) * inv_s_q);
    }
    lg_rows_to_tile<T, DP, PPL>(u, dx, p, live, tu, lx);
    lg_u_ready(uq_out != nullptr || (row_terms & 4));
    if (uq_out != nullptr) lg_store_rows<T, true>(uq_out + n0 * dx, np * dx, tu, lx);
    if (row_terms & 4) lg_row_sums<T>(tu, lx.rs, dx, np, k0_tile, K, scratch, rows + (tile * 3 + 2) * (kLgRowsMax * 16));
    if (gxprev != nullptr) lg_apply_loop<T, DP, PPL>(wn + 2 * DP * DP, tu, au, dx, gprev);
    lg_outer_accumulate_own<T, PPL>(tu, lx.rs, tprev, lx.rs, np, acc_q);
    lg_u_ready(uq_out != nullptr || (row_terms & 4));
    // ---- emission term: u = g (y - loc_g) / s_g^2
    lg_row_values<T, DP, PPL, 4, 2>(vec, true, tab, b0, brow, u);
    lg_apply_loop<T, DP, PPL>(wf + DP * DP, tx, at, dx, u);
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      T q = T(0);
      const T scaled = g[r] * inv_var_g;
      const T *yrow = tab + ((brow[r] - b0) * 4 + 3) * DP;
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const T diff = (uint32_t)j < dy ? yrow[j] - u[j][r] : T(0);
        q = fma_t(diff, diff, q);
        u[j][r] = scaled * diff;
      }
      scale_acc[1] += g[r] * (q * inv_var_g * inv_s_g - T(dy) * inv_s_g);
    }
    if (lx.rs != ly.rs) lg_lds_barrier();     // the u tile changes layout: rows of other wavefronts move under it
    lg_rows_to_tile<T, DP, PPL>(u, dy, p, live, tu, ly);
    lg_u_ready(ug_out != nullptr || (row_terms & 2));
    if (ug_out != nullptr) lg_store_rows<T, true>(ug_out + n0 * dy, np * dy, tu, ly);
    if (row_terms & 2) lg_row_sums<T>(tu, ly.rs, dy, np, k0_tile, K, scratch, rows + (tile * 3 + 1) * (kLgRowsMax * 16));
    if (gx != nullptr) lg_apply_loop<T, DP, PPL>(wn + DP * DP, tu, ay, dy, gcur);
    lg_outer_accumulate_own<T, PPL>(tu, ly.rs, tx, lx.rs, np, acc_c);
    lg_lds_barrier();
    // ---- the two latent gradients leave through the input tiles
    if (gxprev != nullptr) lg_rows_to_tile<T, DP, PPL>(gprev, dx, p, live, tprev, lx);
    if (gx != nullptr) lg_rows_to_tile<T, DP, PPL>(gcur, dx, p, live, tx, lx);
    lg_lds_barrier();
    if (gxprev != nullptr) lg_store_rows<T, true>(gxprev + n0 * dx, np * dx, tprev, lx);
    if (gx != nullptr) lg_store_rows<T, true>(gx + n0 * dx, np * dx, tx, lx);
    lg_lds_barrier();
  }
  T *record = reinterpret_cast<T *>(out.ws) + (int64_t)blockIdx.x * 4 * kLgRecord;
  lg_outer_publish<T>(acc_a, scratch, record);
  lg_outer_publish<T>(acc_c, scratch, record + kLgRecord);
  lg_outer_publish<T>(acc_q, scratch, record + 2 * kLgRecord);
  // the three scale gradients: lanes -> wavefronts (shuffles) -> workgroup, fixed order
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    T v = scale_acc[m];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    if ((threadIdx.x & 63) == 0) scratch[(threadIdx.x >> 6) * 4 + m] = v;
  }
  __syncthreads();
  if (threadIdx.x < kLgRecord) {
    const int m = threadIdx.x;
    record[3 * kLgRecord + m] = m < 3 ? ((scratch[m] + scratch[4 + m]) + scratch[8 + m]) + scratch[12 + m] : T(0);
  }
}

// K12 for a step whose x_t IS the proposal's reparameterised draw, x_t = loc_q(x_{t-1}) + s_q eps (kernel K9):
// the whole step's backward in one pass.  With w = gx_in + dL/dx_t through the transition and emission
// densities, the draw carries w to the proposal's parameters and to x_{t-1}; the proposal's own density
// depends on them only through eps = (x_t - loc_q) / s_q, which the draw holds fixed — its location terms
// cancel identically and only -d log s_q survives.  So: the emission term first (its adjoint starts w), the
// transition term (w -= u_p), then w itself takes the place K12 gives u_q: grad W_q = sum w (x) x_{t-1},
// grad offset_q = row sums of w, grad x_{t-1} = A^T u_p + Q^T w, grad s_q = sum g d / s_q + w . eps.  Neither
// a gradient for x_t nor K9's own backward launch (K11) nor the two [B,K,d] accumulations between them exist.
#ifndef LG_STEP_UNROLL
#define LG_STEP_UNROLL 2
#endif
// EXACT: both extents equal DP (the host checks) — every extent test, row stride and LDS offset folds.
template <typename T, int DP, int PPL, bool EXACT>
__global__ __launch_bounds__(kLgBlock, (PPL == 2 || sizeof(T) == 8) ? 2 : 3) void affine_step_backward_kernel(
    const T *__restrict__ xprev, const T *__restrict__ x, const T *__restrict__ y, int64_t y_sb, LgMap mp, LgMap mg,
    LgMap mq, const T *__restrict__ sp_ptr, const T *__restrict__ sg_ptr, const T *__restrict__ sq_ptr,
    const T *__restrict__ lw, const T *__restrict__ lse, const T *__restrict__ grad_lse,
    const T *__restrict__ grad_lw, LgBackwardOut out, int64_t N, uint32_t K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lg_smem[];
  constexpr uint32_t TP = kLgBlock * PPL;
  constexpr int kUnroll = EXACT ? LG_STEP_UNROLL : 2;
  constexpr bool ONES = EXACT && DP < 16;      // the latent has no column 15: it carries the column sums
  const uint32_t dx = EXACT ? (uint32_t)DP : (uint32_t)mp.dout, dy = EXACT ? (uint32_t)DP : (uint32_t)mg.dout;
  T *wf = reinterpret_cast<T *>(lg_smem);        // [3][DP*DP] input-major: locations (p, g, q)
  T *wn = wf + 3 * DP * DP;                      // [3][DP*DP] output-major: adjoints
  T *scratch = wn + 3 * DP * DP;
  T *tab = scratch + kLgRowPart;                 // [kLgRowsMax][4][DP]: offsets p, q, g and the observation
  T *tprev = tab + kLgRowsMax * 4 * DP;
  const LgLayout lx = lg_layout<T>(dx), ly = lg_layout<T>(dy);
  T *tx = tprev + (TP * lx.rs + 16);
  T *tu = tx + (TP * lx.rs + 16);
  const bool want_sq = out.want_scale_q != 0;
  {
    lg_stage_weight<T, DP>(mp, wf);
    lg_stage_weight<T, DP>(mg, wf + DP * DP);
    if (want_sq) lg_stage_weight<T, DP>(mq, wf + 2 * DP * DP);
    LgMap t = mp;
    t.sj = mp.si; t.si = mp.sj; t.dout = mp.din; t.din = mp.dout;
    lg_stage_weight<T, DP>(t, wn);
    t = mg; t.sj = mg.si; t.si = mg.sj; t.dout = mg.din; t.din = mg.dout;
    lg_stage_weight<T, DP>(t, wn + DP * DP);
    t = mq; t.sj = mq.si; t.si = mq.sj; t.dout = mq.din; t.din = mq.dout;
    lg_stage_weight<T, DP>(t, wn + 2 * DP * DP);
  }
  const LgRowVec<T> vec[4] = {lg_offset_vec<T>(mp), lg_offset_vec<T>(mq), lg_offset_vec<T>(mg), {y, y_sb, (int)dy}};
  const T s_p = sp_ptr[0], s_g = sg_ptr[0], s_q = sq_ptr[0];
  const T inv_var_p = T(1) / (s_p * s_p), inv_var_g = T(1) / (s_g * s_g);
  const T inv_s_p = T(1) / s_p, inv_s_g = T(1) / s_g, inv_s_q = T(1) / s_q;
  typename Mfma<T>::Acc acc_a = {T(0), T(0), T(0), T(0)}, acc_c = acc_a, acc_q = acc_a;
  T scale_acc[3] = {T(0), T(0), T(0)};
  T *gxprev = reinterpret_cast<T *>(out.gxprev);
  const T *gx_in = reinterpret_cast<const T *>(out.gx_in);
  T *rows = reinterpret_cast<T *>(out.rows);
  const int row_terms = rows != nullptr ? out.row_terms : 0;
  const int64_t tiles = (N + TP - 1) / TP;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TP;
    const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
    lg_stage_rows<T, true>(xprev + n0 * dx, np * dx, tprev, lx, 0);
    lg_stage_rows<T, true>(x + n0 * dx, np * dx, tx, lx, 0);
    if (gx_in != nullptr) lg_stage_rows<T, true>(gx_in + n0 * dx, np * dx, tu, lx, 0);
    uint32_t p[PPL], brow[PPL], at[PPL];
    bool live[PPL];
    lg_rows<PPL, true>(n0, np, K, p, live, brow);
    const uint32_t b0 = (uint32_t)(n0 / K), nrows = (uint32_t)((n0 + np - 1) / K) - b0 + 1;
    lg_stage_table<T, DP, 4, true>(vec, b0, nrows, tab);      // the host guarantees nrows <= kLgRowsMax
    const uint32_t k0_tile = (uint32_t)(n0 - (int64_t)b0 * K);
    // offsets' gradients: a tile inside one batch row takes its sums from the matrix cores' spare column
    const bool column_sums = ONES && lg_single_row(n0, np, K);
    const int row_pass = column_sums ? 0 : row_terms;      // the terms that need the pass over the tile
    T g[PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const int64_t n = n0 + p[r];
      T value = grad_lw != nullptr ? grad_lw[n] : T(0);
      if (grad_lse != nullptr) value = value + grad_lse[brow[r]] * Num<T>::exp(lw[n] - lse[brow[r]]);
      g[r] = live[r] ? value : T(0);
      at[r] = p[r] * lx.rs;
    }
    lg_lds_barrier();
    T u[DP][PPL], gprev[DP][PPL], w[DP][PPL];
    uint32_t au[PPL], ay[PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      au[r] = p[r] * lx.rs;
      ay[r] = p[r] * ly.rs;
    }
    // a lane's row of the spare tile is its own particle's: read before the lane overwrites it below
#pragma unroll
    for (int j = 0; j < DP; ++j)
#pragma unroll
      for (int r = 0; r < PPL; ++r) {
        gprev[j][r] = T(0);
        w[j][r] = (gx_in != nullptr && (uint32_t)j < dx && live[r]) ? tu[au[r] + min(j, (int)dx - 1)] : T(0);
      }
    if (gx_in != nullptr && lx.rs != ly.rs) lg_lds_barrier();   // the tile changes layout under the other wavefronts
    // ---- emission term: u = g (y - loc_g) / s_g^2;  w += C^T u
    lg_row_values<T, DP, PPL, 4, 2>(vec, true, tab, b0, brow, u);
    lg_apply_loop<T, DP, PPL, kUnroll>(wf + DP * DP, tx, at, dx, u);
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      T q = T(0);
      const T scaled = g[r] * inv_var_g;
      const T *yrow = tab + ((brow[r] - b0) * 4 + 3) * DP;
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const T diff = (uint32_t)j < dy ? yrow[j] - u[j][r] : T(0);
        q = fma_t(diff, diff, q);
        u[j][r] = scaled * diff;
      }
      scale_acc[1] += g[r] * (q * inv_var_g * inv_s_g - T(dy) * inv_s_g);
    }
    lg_rows_to_tile<T, DP, PPL>(u, dy, p, live, tu, ly);
    lg_u_ready((row_pass & 2) != 0);
    if (row_pass & 2) lg_row_sums<T>(tu, ly.rs, dy, np, k0_tile, K, scratch, rows + (tile * 3 + 1) * (kLgRowsMax * 16));
    lg_apply_loop<T, DP, PPL, kUnroll>(wn + DP * DP, tu, ay, dy, w);
    lg_outer_accumulate_own<T, PPL, ONES>(tu, ly.rs, tx, lx.rs, np, acc_c);
    if (ONES) lg_flush_column_sums<T>(acc_c, rows + (tile * 3 + 1) * (kLgRowsMax * 16), column_sums && (row_terms & 2));
    if (lx.rs != ly.rs) lg_lds_barrier();     // back to the latent's layout
    else lg_u_ready((row_pass & 2) != 0);
    // ---- transition term: u = g (x - loc_p) / s_p^2;  w -= u
    lg_row_values<T, DP, PPL, 4, 0>(vec, true, tab, b0, brow, u);
    lg_apply_loop<T, DP, PPL, kUnroll>(wf, tprev, at, dx, u);
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      T q = T(0);
      const T scaled = g[r] * inv_var_p;
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const T diff = (uint32_t)j < dx ? tx[at[r] + min(j, (int)dx - 1)] - u[j][r] : T(0);
        q = fma_t(diff, diff, q);
        u[j][r] = scaled * diff;
        w[j][r] = w[j][r] - u[j][r];
      }
      scale_acc[0] += g[r] * (q * inv_var_p * inv_s_p - T(dx) * inv_s_p);
    }
    lg_rows_to_tile<T, DP, PPL>(u, dx, p, live, tu, lx);
    lg_u_ready((row_pass & 1) != 0);
    if (row_pass & 1) lg_row_sums<T>(tu, lx.rs, dx, np, k0_tile, K, scratch, rows + (tile * 3 + 0) * (kLgRowsMax * 16));
    if (gxprev != nullptr) lg_apply_loop<T, DP, PPL, kUnroll>(wn, tu, au, dx, gprev);
    lg_outer_accumulate_own<T, PPL, ONES>(tu, lx.rs, tprev, lx.rs, np, acc_a);
    if (ONES) lg_flush_column_sums<T>(acc_a, rows + (tile * 3 + 0) * (kLgRowsMax * 16), column_sums && (row_terms & 1));
    lg_u_ready((row_pass & 1) != 0);
    // ---- the draw: w reaches the proposal's parameters and x_{t-1};  grad s_q = g d / s_q + w . eps
    if (want_sq) {
      lg_row_values<T, DP, PPL, 4, 1>(vec, true, tab, b0, brow, u);
      lg_apply_loop<T, DP, PPL, kUnroll>(wf + 2 * DP * DP, tprev, at, dx, u);
#pragma unroll
      for (int r = 0; r < PPL; ++r) {
        T dot = T(0);
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          const T diff = (uint32_t)j < dx ? tx[at[r] + min(j, (int)dx - 1)] - u[j][r] : T(0);
          dot = fma_t(w[j][r], diff, dot);
        }
        if (live[r]) scale_acc[2] += g[r] * (T(dx) * inv_s_q) + dot * inv_s_q;   // a spare lane's w is particle 0's
      }
    }
    lg_rows_to_tile<T, DP, PPL>(w, dx, p, live, tu, lx);
    lg_u_ready((row_pass & 4) != 0);
    if (row_pass & 4) lg_row_sums<T>(tu, lx.rs, dx, np, k0_tile, K, scratch, rows + (tile * 3 + 2) * (kLgRowsMax * 16));
    if (gxprev != nullptr) lg_apply_loop<T, DP, PPL, kUnroll>(wn + 2 * DP * DP, tu, au, dx, gprev);
    lg_outer_accumulate_own<T, PPL, ONES>(tu, lx.rs, tprev, lx.rs, np, acc_q);
    if (ONES) lg_flush_column_sums<T>(acc_q, rows + (tile * 3 + 2) * (kLgRowsMax * 16), column_sums && (row_terms & 4));
    if constexpr (EXACT && (DP * sizeof(T)) % 8 == 0) {
      // rows of whole 8-byte pairs: each lane stores its own particles' rows (a wavefront's stores cover
      // one contiguous span) — no trip through the tile, no barriers around it
      if (gxprev != nullptr) {
        typedef T Pair __attribute__((ext_vector_type(8 / sizeof(T))));
        constexpr int PER = 8 / sizeof(T);
#pragma unroll
        for (int r = 0; r < PPL; ++r) {
          if (live[r]) {
            Pair *dst = reinterpret_cast<Pair *>(gxprev + (n0 + p[r]) * DP);
#pragma unroll
            for (int j = 0; j < DP / PER; ++j) {
              Pair v;
#pragma unroll
              for (int e = 0; e < PER; ++e) v[e] = gprev[j * PER + e][r];
              dst[j] = v;
            }
          }
        }
      }
      lg_lds_barrier();
    } else {
      lg_lds_barrier();
      if (gxprev != nullptr) {
        lg_rows_to_tile<T, DP, PPL>(gprev, dx, p, live, tprev, lx);
        lg_lds_barrier();
        lg_store_rows<T, true>(gxprev + n0 * dx, np * dx, tprev, lx);
      }
      lg_lds_barrier();
    }
  }
  T *record = reinterpret_cast<T *>(out.ws) + (int64_t)blockIdx.x * 4 * kLgRecord;
  lg_outer_publish<T>(acc_a, scratch, record);
  lg_outer_publish<T>(acc_c, scratch, record + kLgRecord);
  lg_outer_publish<T>(acc_q, scratch, record + 2 * kLgRecord);
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    T v = scale_acc[m];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    if ((threadIdx.x & 63) == 0) scratch[(threadIdx.x >> 6) * 4 + m] = v;
  }
  __syncthreads();
  if (threadIdx.x < kLgRecord) {
    const int m = threadIdx.x;
    record[3 * kLgRecord + m] = m < 3 ? ((scratch[m] + scratch[4 + m]) + scratch[8 + m]) + scratch[12 + m] : T(0);
  }
}

// Sums the workgroups' records in workgroup order: out[m][j * din + i] = sum_b ws[b][m][j * 16 + i].
struct LgFinish {
  void *out[4];
  int32_t rows[4], cols[4];
  // offset gradients: goff[t][b][j] = sum over the tiles covering batch row b of their row-sum records
  const void *row_ws;     // [tile][row_terms][kLgRowsMax][16], or nullptr
  void *goff[3];
  int32_t goff_d[3];
  int32_t row_terms, matrices, row_blocks;    // records per tile; matrix blocks; row blocks per term (64 rows each)
  int32_t column_sums;                        // single-row tiles hold four wavefront sums (lg_flush_column_sums)
  int64_t B, N;
  uint32_t K, TP;
};
template <typename T>
__global__ __launch_bounds__(1024) void lg_finish_kernel(const T *__restrict__ ws, int nblocks, int record, LgFinish f) {
  if ((int)blockIdx.x >= f.matrices) {
    // ---- one lane per (batch row, column): the few tiles that cover the row, in tile order
    const int r = (int)blockIdx.x - f.matrices, term = r / f.row_blocks;
    T *goff = reinterpret_cast<T *>(f.goff[term]);
    if (goff == nullptr) return;
    const int64_t b = (int64_t)(r - term * f.row_blocks) * 64 + (threadIdx.x >> 4);
    const uint32_t j = threadIdx.x & 15u, d = (uint32_t)f.goff_d[term];
    if (b >= f.B || j >= d) return;
    const T *rows = reinterpret_cast<const T *>(f.row_ws);
    const int64_t first = b * f.K / f.TP, last = ((b + 1) * f.K - 1) / f.TP;
    T sum = T(0);
    for (int64_t tile = first; tile <= last; ++tile) {
      const int64_t n0 = tile * f.TP, b0 = n0 / f.K;
      const T *record = rows + (tile * f.row_terms + term) * (kLgRowsMax * 16);
      if (f.column_sums && lg_single_row(n0, (uint32_t)min((int64_t)f.TP, f.N - n0), f.K))
        sum += ((record[j] + record[16 + j]) + record[32 + j]) + record[48 + j];     // the four wavefronts' sums
      else
        sum += record[(b - b0) * 16 + j];
    }
    goff[b * d + j] = sum;
    return;
  }
  // element e of matrix m: four lanes each sum a quarter of the workgroups' records (in workgroup order,
  // eight loads in flight), then the quarters are added in order — fixed association, reproducible
  __shared__ T part[4 * 256];
  const int m = blockIdx.x, e = threadIdx.x & 255, seg = threadIdx.x >> 8;
  const int per = (nblocks + 3) / 4, b0 = seg * per, b1 = min(nblocks, b0 + per);
  T sum = T(0);
  int b = b0;
  for (; b + 8 <= b1; b += 8) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ws[(int64_t)(b + u) * record + m * 256 + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += v[u];
  }
  for (; b < b1; ++b) sum += ws[(int64_t)b * record + m * 256 + e];
  part[seg * 256 + e] = sum;
  __syncthreads();
  if (seg == 0) {
    const T total = ((part[e] + part[256 + e]) + part[512 + e]) + part[768 + e];
    const int j = e >> 4, i = e & 15;
    T *out = reinterpret_cast<T *>(f.out[m]);
    if (out != nullptr && j < f.rows[m] && i < f.cols[m]) out[j * f.cols[m] + i] = total;
  }
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
static inline unsigned lg_persistent_grid(int64_t tiles, size_t lds_bytes, int max_per_cu = 8) {
  int device = 0, cus = 256;
  if (hipGetDevice(&device) == hipSuccess) {
    int value = 0;
    if (hipDeviceGetAttribute(&value, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && value > 0)
      cus = value;
  }
  int per_cu = (int)((size_t)160 * 1024 / (lds_bytes > 0 ? lds_bytes : 1));
  per_cu = per_cu < 1 ? 1 : (per_cu > max_per_cu ? max_per_cu : per_cu);
  const int64_t resident = (int64_t)cus * per_cu;
  return (unsigned)(tiles < resident ? tiles : resident);
}
constexpr size_t kLgLdsLimit = 144 * 1024;

// Launches `KERNEL<T, DP, PPL>` with DP from `dp` (4, 8, 12, 16) and PPL from `ppl` (1, 2).  Tiles beyond
// 64 KiB of LDS (float64 rows of 10 and more values) need the opt-in; it is per kernel and per device,
// cheap, and only taken for those shapes.
#define LG_LAUNCH(KERNEL, T, DP_, PPL_, grid, lds, stream, ...)                                              \
  do {                                                                                                       \
    if ((lds) > 64 * 1024)                                                                                   \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL<T, DP_, PPL_>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds));                     \
    hipLaunchKernelGGL((KERNEL<T, DP_, PPL_>), grid, dim3(kLgBlock), lds, stream, __VA_ARGS__);              \
  } while (0)
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

template <typename T>
static int launch_particle_affine(const void *x1, const aesmc_affine_map *m1, const void *x2,
                                  const aesmc_affine_map *m2, const void *base, void *out, int64_t B, int64_t K,
                                  hipStream_t stream) {
  const int64_t N = B * K;
  const int64_t d1 = m1->din, d2 = x2 != nullptr ? m2->din : 0, dout = m1->dout;
  const int dp = lg_pad_dim(std::max(std::max(d1, d2), dout));
  int ppl = sizeof(T) == 4 ? 2 : 1;
  size_t lds = 0;
  for (; ppl >= 1; --ppl) {
    const size_t tp = (size_t)kLgBlock * ppl;
    lds = sizeof(T) * (2 * (size_t)dp * dp + lg_tile_elems<T>(tp, d1) + lg_tile_elems<T>(tp, d2) + lg_tile_elems<T>(tp, dout));
    if (lds <= (ppl > 1 ? kLgLdsBudget : kLgLdsLimit)) break;
  }
  if (ppl < 1) return AESMC_ERR_UNSUPPORTED;
  const int64_t tiles = (N + (int64_t)kLgBlock * ppl - 1) / ((int64_t)kLgBlock * ppl);
  if (tiles > 0x7fffffff) return AESMC_ERR_UNSUPPORTED;
  LgMap a = lg_map(m1), b = x2 != nullptr ? lg_map(m2) : a;
  const int hint = stream_hint((uint64_t)N * (d1 + d2 + dout) * sizeof(T));
  LG_DISPATCH(particle_affine_kernel, T, dp, ppl, dim3((unsigned)tiles), lds, stream, static_cast<const T *>(x1), a,
              static_cast<const T *>(x2), b, static_cast<const T *>(base), static_cast<T *>(out), N, (uint32_t)K,
              hint);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

// The persistent kernels with (`_tab`) and without the per-row LDS table, as three-parameter names for
// the dispatch macro.
template <typename T, int DP, int PPL>
static constexpr auto affine_rsample_tab = &affine_rsample_kernel<T, DP, PPL, true>;
template <typename T, int DP, int PPL>
static constexpr auto affine_rsample_rows = &affine_rsample_kernel<T, DP, PPL, false>;
template <typename T, int DP, int PPL>
static constexpr auto affine_logweight_tab = &affine_logweight_kernel<T, DP, PPL, true, false>;
template <typename T, int DP, int PPL>
static constexpr auto affine_logweight_tab_prefetch = &affine_logweight_kernel<T, DP, PPL, true, true>;
template <typename T, int DP, int PPL>
static constexpr auto affine_logweight_rows = &affine_logweight_kernel<T, DP, PPL, false, false>;
template <typename T, int DP, int PPL>
static constexpr auto affine_propagate_tab = &affine_logweight_kernel<T, DP, PPL, true, false, true>;
template <typename T, int DP, int PPL>
static constexpr auto affine_propagate_tab_prefetch = &affine_logweight_kernel<T, DP, PPL, true, true, true>;
template <typename T, int DP, int PPL>
static constexpr auto affine_propagate_rows = &affine_logweight_kernel<T, DP, PPL, false, false, true>;

// With 512-particle tiles a launch of fewer than ~1M particles leaves each CU with at most two or three
// workgroups of one tile each: all latency.  256-particle tiles double the workgroups.
static inline bool lg_few_tiles(int64_t N) { return N < ((int64_t)1 << 20); }

// AESMC_LG_FWD_PPL=1: one particle per lane in K9 / K10 whatever the size (a measurement knob).
static inline int lg_forward_ppl() {
  static const int v = [] { const char *e = getenv("AESMC_LG_FWD_PPL"); return e != nullptr ? atoi(e) : 0; }();
  return v;
}

// Batch rows a tile of `tp` consecutive particles can span.
static inline int64_t lg_rows_spanned(int64_t tp, int64_t K) { return (tp - 1) / K + 2; }

template <typename T>
static int launch_affine_rsample(const void *src, const aesmc_affine_map *m, const void *eps, const void *scale,
                                 void *out, int64_t B, int64_t K, hipStream_t stream) {
  const int64_t N = B * K;
  const int dp = lg_pad_dim(std::max(m->din, m->dout));
  int ppl = sizeof(T) == 4 ? 2 : 1;
  size_t lds = 0;
  for (; ppl >= 1; --ppl) {
    const size_t tp = (size_t)kLgBlock * ppl;
    lds = sizeof(T) * ((size_t)dp * dp + (size_t)kLgRowsMax * dp + lg_tile_elems<T>(tp, m->din) + lg_tile_elems<T>(tp, m->dout));
    if (lds <= (ppl > 1 ? kLgLdsBudget : kLgLdsLimit)) break;
  }
  if (ppl < 1) return AESMC_ERR_UNSUPPORTED;
  if (ppl == 2 && (lg_few_tiles(N) || lg_forward_ppl() == 1)) ppl = 1;      // small launches: twice the workgroups, half the tile each
  bool tab = lg_rows_spanned((int64_t)kLgBlock * ppl, K) <= kLgRowsMax;
  if (!tab && ppl == 2) {      // few particles per batch row: one particle per lane, rows from global memory
    ppl = 1;
    tab = lg_rows_spanned(kLgBlock, K) <= kLgRowsMax;
  }
  {
    const size_t tp = (size_t)kLgBlock * ppl;
    lds = sizeof(T) * ((size_t)dp * dp + (size_t)kLgRowsMax * dp + lg_tile_elems<T>(tp, m->din) + lg_tile_elems<T>(tp, m->dout));
  }
  const int64_t tiles = (N + (int64_t)kLgBlock * ppl - 1) / ((int64_t)kLgBlock * ppl);
  if (tiles > 0x7fffffff) return AESMC_ERR_UNSUPPORTED;
  const int hint = stream_hint((uint64_t)N * (m->din + 2 * m->dout) * sizeof(T));
  const unsigned grid = lg_persistent_grid(tiles, lds);
  if (tab)
    LG_DISPATCH(affine_rsample_tab, T, dp, ppl, dim3(grid), lds, stream, static_cast<const T *>(src), lg_map(m),
                static_cast<const T *>(eps), static_cast<const T *>(scale), static_cast<T *>(out), N, (uint32_t)K, hint);
  else
    LG_DISPATCH(affine_rsample_rows, T, dp, 1, dim3(grid), lds, stream, static_cast<const T *>(src), lg_map(m),
                static_cast<const T *>(eps), static_cast<const T *>(scale), static_cast<T *>(out), N, (uint32_t)K, hint);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

template <typename T>
static int launch_affine_logweight(const void *xprev, const void *x, const void *y, int64_t y_sb,
                                   const aesmc_affine_map *mp, const aesmc_affine_map *mg, const aesmc_affine_map *mq,
                                   const void *sp, const void *sg, const void *sq, void *out, int64_t B, int64_t K,
                                   hipStream_t stream, void *out_x = nullptr) {
  // out_x != nullptr: `x` is the proposal's noise and the draw is formed here (K15)
  const int64_t N = B * K;
  const int64_t dx = mp->dout, dy = mg->dout;
  const int dp = lg_pad_dim(std::max(dx, dy));
  int ppl = sizeof(T) == 4 ? 2 : 1;
  size_t lds = 0;
  for (; ppl >= 1; --ppl) {
    const size_t tp = (size_t)kLgBlock * ppl;
    lds = sizeof(T) * (3 * (size_t)dp * dp + (size_t)kLgRowsMax * 4 * dp + 2 * lg_tile_elems<T>(tp, dx));
    if (lds <= (ppl > 1 ? kLgLdsBudget : kLgLdsLimit)) break;
  }
  if (ppl < 1) return AESMC_ERR_UNSUPPORTED;
  if (ppl == 2 && (lg_few_tiles(N) || lg_forward_ppl() == 1)) ppl = 1;
  bool tab = lg_rows_spanned((int64_t)kLgBlock * ppl, K) <= kLgRowsMax;
  if (!tab && ppl == 2) {
    ppl = 1;
    tab = lg_rows_spanned(kLgBlock, K) <= kLgRowsMax;
  }
  {
    const size_t tp = (size_t)kLgBlock * ppl;
    lds = sizeof(T) * (3 * (size_t)dp * dp + (size_t)kLgRowsMax * 4 * dp + 2 * lg_tile_elems<T>(tp, dx));
  }
  const int64_t tiles = (N + (int64_t)kLgBlock * ppl - 1) / ((int64_t)kLgBlock * ppl);
  if (tiles > 0x7fffffff) return AESMC_ERR_UNSUPPORTED;
  // persistent workgroups with the next tile prefetched (B=1024 K=4096 d=10: 106 us against 145 us with one
  // tile per workgroup); AESMC_LG_PREFETCH=0 selects the latter: a measurement knob
  static const bool prefetch = [] { const char *v = getenv("AESMC_LG_PREFETCH"); return v == nullptr || v[0] != '0'; }();
  const unsigned grid = prefetch ? lg_persistent_grid(tiles, lds) : (unsigned)tiles;
#define LG_LOGWEIGHT_ARGS                                                                                          \
  static_cast<const T *>(xprev), static_cast<const T *>(x), static_cast<const T *>(y), y_sb, lg_map(mp), lg_map(mg), \
      lg_map(mq), static_cast<const T *>(sp), static_cast<const T *>(sg), static_cast<const T *>(sq),              \
      static_cast<T *>(out), N, (uint32_t)K, static_cast<T *>(out_x)
  if (out_x != nullptr) {
    if (tab && prefetch) LG_DISPATCH(affine_propagate_tab_prefetch, T, dp, ppl, dim3(grid), lds, stream, LG_LOGWEIGHT_ARGS);
    else if (tab) LG_DISPATCH(affine_propagate_tab, T, dp, ppl, dim3(grid), lds, stream, LG_LOGWEIGHT_ARGS);
    else LG_DISPATCH(affine_propagate_rows, T, dp, 1, dim3(grid), lds, stream, LG_LOGWEIGHT_ARGS);
  } else {
    if (tab && prefetch) LG_DISPATCH(affine_logweight_tab_prefetch, T, dp, ppl, dim3(grid), lds, stream, LG_LOGWEIGHT_ARGS);
    else if (tab) LG_DISPATCH(affine_logweight_tab, T, dp, ppl, dim3(grid), lds, stream, LG_LOGWEIGHT_ARGS);
    else LG_DISPATCH(affine_logweight_rows, T, dp, 1, dim3(grid), lds, stream, LG_LOGWEIGHT_ARGS);
  }
#undef LG_LOGWEIGHT_ARGS
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

// Workspace of the reducing kernels: kLgMaxGrid records of 4 x 256 values (weight-gradient partials), then
// one 8 x 16 record per 256-particle tile and term (offset-gradient row sums).
static inline size_t lg_record_elems() { return (size_t)kLgMaxGrid * 4 * kLgRecord; }
static inline size_t lg_row_elems(int64_t N, int terms) {
  return (size_t)((N + kLgBlock - 1) / kLgBlock) * terms * kLgRowsMax * 16;
}

template <typename T>
static int launch_particle_affine_backward(const void *g, const void *x, const aesmc_affine_map *m, void *gx, void *gw,
                                           void *goff, void *ws, size_t ws_bytes, int64_t B, int64_t K,
                                           hipStream_t stream) {
  const int64_t N = B * K;
  const int64_t dout = m->dout, din = m->din;
  const int dp = lg_pad_dim(std::max(dout, din));
  int ppl = sizeof(T) == 4 ? 2 : 1;
  size_t lds = 0;
  for (; ppl >= 1; --ppl) {
    const size_t tp = (size_t)kLgBlock * ppl;
    lds = sizeof(T) * (16 + (size_t)dp * dp + kLgRowPart + lg_tile_elems<T>(tp, dout) + lg_tile_elems<T>(tp, din));
    if (lds <= (ppl > 1 ? kLgLdsBudget : kLgLdsLimit) &&
        (goff == nullptr || lg_rows_spanned((int64_t)tp, K) <= kLgRowsMax))
      break;
  }
  if (ppl < 1) return AESMC_ERR_UNSUPPORTED;     // offset gradient with fewer than ~43 particles per row: caller sums
  const int64_t tp = (int64_t)kLgBlock * ppl;
  const int64_t tiles = (N + tp - 1) / tp;
  const int grid = (int)std::min<int64_t>(lg_persistent_grid(tiles, lds), kLgMaxGrid);
  const size_t need = (gw != nullptr || goff != nullptr) ? lg_record_elems() + lg_row_elems(N, 1) : 0;
  if (ws_bytes < need * sizeof(T)) return AESMC_ERR_WORKSPACE;
  T *records = static_cast<T *>(ws);
  T *rows = goff != nullptr ? records + lg_record_elems() : nullptr;
  LgMap adjoint;           // gx = g W: the map from the location's extent back to the input's
  adjoint.w = m->weight; adjoint.sj = m->stride_in; adjoint.si = m->stride_out;
  adjoint.off = nullptr; adjoint.off_sb = 0; adjoint.dout = (int32_t)din; adjoint.din = (int32_t)dout;
  LG_DISPATCH(particle_affine_backward_kernel, T, dp, ppl, dim3((unsigned)grid), lds, stream,
              static_cast<const T *>(g), static_cast<const T *>(x), adjoint, static_cast<T *>(gx), records, rows, N,
              (uint32_t)K, gw != nullptr ? 1 : 0);
  if (hipGetLastError() != hipSuccess) return AESMC_ERR_LAUNCH;
  if (gw != nullptr || goff != nullptr) {      // one finishing launch: the weight gradient's records, the rows' tiles
    LgFinish f = {};
    f.out[0] = gw; f.rows[0] = (int32_t)dout; f.cols[0] = (int32_t)din;
    f.matrices = gw != nullptr ? 1 : 0;
    f.row_ws = rows; f.goff[0] = goff; f.goff_d[0] = (int32_t)dout; f.row_terms = 1;
    f.row_blocks = goff != nullptr ? (int32_t)((B + 63) / 64) : 0;
    f.B = B; f.K = (uint32_t)K; f.TP = (uint32_t)tp;
    hipLaunchKernelGGL(lg_finish_kernel<T>, dim3((unsigned)(f.matrices + f.row_blocks)), dim3(1024), 0, stream,
                       static_cast<const T *>(records), grid, kLgRecord, f);
    if (hipGetLastError() != hipSuccess) return AESMC_ERR_LAUNCH;
  }
  return AESMC_OK;
}

template <typename T, int DP, int PPL>
static constexpr auto affine_step_backward_exact = &affine_step_backward_kernel<T, DP, PPL, true>;
template <typename T, int DP, int PPL>
static constexpr auto affine_step_backward_any = &affine_step_backward_kernel<T, DP, PPL, false>;

template <typename T>
static int launch_affine_logweight_backward(const void *xprev, const void *x, const void *y, int64_t y_sb,
                                            const aesmc_affine_map *mp, const aesmc_affine_map *mg,
                                            const aesmc_affine_map *mq, const void *sp, const void *sg, const void *sq,
                                            const void *lw, const void *lse, const void *grad_lse, const void *grad_lw,
                                            const aesmc_affine_logweight_grads *o, void *ws, size_t ws_bytes, int64_t B,
                                            int64_t K, hipStream_t stream, bool step = false,
                                            const void *gx_in = nullptr) {
  const int64_t N = B * K;
  const int64_t dx = mp->dout, dy = mg->dout;
  const int dp = lg_pad_dim(std::max(dx, dy));
  static const int forced = [] { const char *v = getenv("AESMC_LG_BWD_PPL"); return v != nullptr ? atoi(v) : 0; }();   // measurement knob
  int ppl = (sizeof(T) == 4 && dp <= 12 && !lg_few_tiles(N)) ? 2 : 1;
  if (forced == 1 || forced == 2) ppl = (sizeof(T) == 4 && dp <= 12) ? forced : 1;
  size_t lds = 0;
  for (; ppl >= 1; --ppl) {
    const size_t tp = (size_t)kLgBlock * ppl;
    lds = sizeof(T) * (16 + 6 * (size_t)dp * dp + kLgRowPart + (size_t)kLgRowsMax * 4 * dp + 2 * lg_tile_elems<T>(tp, dx) +
                       std::max(lg_tile_elems<T>(tp, dx), lg_tile_elems<T>(tp, dy)));
    if (lds <= (ppl > 1 ? (size_t)78 * 1024 : kLgLdsLimit) && lg_rows_spanned((int64_t)tp, K) <= kLgRowsMax) break;
  }
  if (ppl < 1) return AESMC_ERR_UNSUPPORTED;   // fewer than ~43 particles per batch row: the caller takes the unfused route
  const int64_t tiles = (N + (int64_t)kLgBlock * ppl - 1) / ((int64_t)kLgBlock * ppl);
  const int grid = (int)std::min<int64_t>(lg_persistent_grid(tiles, lds, (ppl == 2 || sizeof(T) == 8) ? 2 : 3), kLgMaxGrid);   // what the registers allow
  const int row_terms = (o->grad_offset_p != nullptr ? 1 : 0) | (o->grad_offset_g != nullptr ? 2 : 0) |
                        (o->grad_offset_q != nullptr ? 4 : 0);
  const size_t need = lg_record_elems() + (row_terms != 0 ? lg_row_elems(N, 3) : 0);
  if (ws_bytes < need * sizeof(T)) return AESMC_ERR_WORKSPACE;
  T *row_ws = row_terms != 0 ? static_cast<T *>(ws) + lg_record_elems() : nullptr;
  LgBackwardOut out;
  out.gxprev = o->grad_x_prev; out.gx = o->grad_x; out.up = o->grad_loc_p; out.ug = o->grad_loc_g;
  out.uq = o->grad_loc_q; out.ws = ws; out.rows = row_ws; out.row_terms = row_terms;
  out.gx_in = gx_in; out.want_scale_q = o->grad_scales != nullptr ? 1 : 0;
#define LG_BACKWARD_ARGS                                                                                            \
  static_cast<const T *>(xprev), static_cast<const T *>(x), static_cast<const T *>(y), y_sb, lg_map(mp), lg_map(mg), \
      lg_map(mq), static_cast<const T *>(sp), static_cast<const T *>(sg), static_cast<const T *>(sq),               \
      static_cast<const T *>(lw), static_cast<const T *>(lse), static_cast<const T *>(grad_lse),                    \
      static_cast<const T *>(grad_lw), out, N, (uint32_t)K
  if (step && dx == dp && dy == dp) {
    LG_DISPATCH(affine_step_backward_exact, T, dp, ppl, dim3((unsigned)grid), lds, stream, LG_BACKWARD_ARGS);
  } else if (step) {
    LG_DISPATCH(affine_step_backward_any, T, dp, ppl, dim3((unsigned)grid), lds, stream, LG_BACKWARD_ARGS);
  } else {
    LG_DISPATCH(affine_logweight_backward_kernel, T, dp, ppl, dim3((unsigned)grid), lds, stream, LG_BACKWARD_ARGS);
  }
#undef LG_BACKWARD_ARGS
  if (hipGetLastError() != hipSuccess) return AESMC_ERR_LAUNCH;
  LgFinish f = {};
  f.out[0] = o->grad_weight_p; f.rows[0] = (int32_t)dx; f.cols[0] = (int32_t)dx;
  f.out[1] = o->grad_weight_g; f.rows[1] = (int32_t)dy; f.cols[1] = (int32_t)dx;
  f.out[2] = o->grad_weight_q; f.rows[2] = (int32_t)dx; f.cols[2] = (int32_t)dx;
  f.out[3] = o->grad_scales; f.rows[3] = 1; f.cols[3] = 3;
  f.matrices = 4;
  f.row_ws = row_ws; f.row_terms = 3;
  f.goff[0] = o->grad_offset_p; f.goff[1] = o->grad_offset_g; f.goff[2] = o->grad_offset_q;
  f.goff_d[0] = (int32_t)dx; f.goff_d[1] = (int32_t)dy; f.goff_d[2] = (int32_t)dx;
  f.row_blocks = row_terms != 0 ? (int32_t)((B + 63) / 64) : 0;
  f.B = B; f.N = N; f.K = (uint32_t)K; f.TP = (uint32_t)(kLgBlock * ppl);
  f.column_sums = (step && dx == dp && dy == dp && dp < 16) ? 1 : 0;
  hipLaunchKernelGGL(lg_finish_kernel<T>, dim3((unsigned)(4 + 3 * f.row_blocks)), dim3(1024), 0, stream,
                     static_cast<const T *>(ws), grid, 4 * kLgRecord, f);   // one finishing launch for everything
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

template <typename T>
static int launch_particle_mlp(const void *x, const aesmc_affine_map *m1, const aesmc_affine_map *m2, void *out,
                               int64_t B, int64_t K, hipStream_t stream) {
  const int64_t N = B * K;
  const int64_t din = m1->din, hid = m1->dout, dout = m2->dout;
  const int dp = lg_pad_dim(std::max(din, dout));
  const uint32_t hp = (uint32_t)((hid + 15) / 16 * 16);
  if (lg_rows_spanned(kLgBlock, K) > kLgRowsMax) return AESMC_ERR_UNSUPPORTED;   // fewer than ~43 particles per row
  const size_t lds = sizeof(T) * (2 * (size_t)dp * hp + (size_t)kLgRowsMax * hp + lg_tile_elems<T>(kLgBlock, din) +
                                  lg_tile_elems<T>(kLgBlock, dout));
  if (lds > kLgLdsLimit) return AESMC_ERR_UNSUPPORTED;
  const int64_t tiles = (N + kLgBlock - 1) / kLgBlock;
  const unsigned grid = lg_persistent_grid(tiles, lds, 8);
  const T *xp = static_cast<const T *>(x);
  T *op = static_cast<T *>(out);
#define LG_MLP_CASE(DP_)                                                                                            \
  do {                                                                                                              \
    if (lds > 64 * 1024)                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&particle_mlp_kernel<T, DP_>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
    hipLaunchKernelGGL((particle_mlp_kernel<T, DP_>), dim3(grid), dim3(kLgBlock), lds, stream, xp, lg_map(m1),      \
                       lg_map(m2), op, N, (uint32_t)K, hp);                                                         \
  } while (0)
  switch (dp) {
    case 4: LG_MLP_CASE(4); break;
    case 8: LG_MLP_CASE(8); break;
    case 10: LG_MLP_CASE(10); break;
    case 12: LG_MLP_CASE(12); break;
    default: LG_MLP_CASE(16); break;
  }
#undef LG_MLP_CASE
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

}  // namespace aesmc

using namespace aesmc;

extern "C" int64_t aesmc_particle_mlp_max_hidden(void) { return 64; }

extern "C" int aesmc_particle_mlp(int dtype, const void *x, const aesmc_affine_map *layer1,
                                  const aesmc_affine_map *layer2, void *out, int64_t B, int64_t K, void *stream) {
  if (x == nullptr || out == nullptr || layer1 == nullptr || layer2 == nullptr || layer1->weight == nullptr ||
      layer2->weight == nullptr || B < 0 || K < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  if (!aligned16(x) || !aligned16(out) || x == out) return AESMC_ERR_INVALID_ARGUMENT;
  if (layer1->din < 1 || layer1->din > kLgMaxDim || layer2->dout < 1 || layer2->dout > kLgMaxDim ||
      layer1->dout < 1 || layer1->dout > 64 || layer2->din != layer1->dout)
    return AESMC_ERR_UNSUPPORTED;
  if (B == 0 || K == 0) return AESMC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AESMC_F32 ? launch_particle_mlp<float>(x, layer1, layer2, out, B, K, s)
                            : launch_particle_mlp<double>(x, layer1, layer2, out, B, K, s);
}


extern "C" int aesmc_affine_normal_logweight_backward(
    int dtype, const void *x_prev, const void *x, const void *y, int64_t y_stride_b, const aesmc_affine_map *transition,
    const aesmc_affine_map *emission, const aesmc_affine_map *proposal, const void *scale_p, const void *scale_g,
    const void *scale_q, const void *lw, const void *lse, const void *grad_lse, const void *grad_lw,
    const aesmc_affine_logweight_grads *out, void *ws, size_t ws_bytes, int64_t B, int64_t K, void *stream) {
  if (x_prev == nullptr || x == nullptr || y == nullptr || transition == nullptr || emission == nullptr ||
      proposal == nullptr || scale_p == nullptr || scale_g == nullptr || scale_q == nullptr || out == nullptr ||
      ws == nullptr || B < 0 || K < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (grad_lw == nullptr && grad_lse == nullptr) return AESMC_ERR_INVALID_ARGUMENT;
  if (grad_lse != nullptr && (lw == nullptr || lse == nullptr)) return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  const void *aligned[] = {x_prev, x, ws, out->grad_x_prev, out->grad_x, out->grad_loc_p, out->grad_loc_g,
                           out->grad_loc_q};
  for (const void *ptr : aligned)
    if (ptr != nullptr && !aligned16(ptr)) return AESMC_ERR_INVALID_ARGUMENT;
  if (!lg_map_ok(transition) || !lg_map_ok(emission) || !lg_map_ok(proposal)) return AESMC_ERR_UNSUPPORTED;
  const int64_t dx = transition->dout;
  if (transition->din != dx || proposal->dout != dx || proposal->din != dx || emission->din != dx)
    return AESMC_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0 || K == 0) {      // empty sums
    const size_t esz = dtype == AESMC_F64 ? 8 : 4;
    bool ok = true;
    if (out->grad_weight_p != nullptr) ok = ok && zero_fill_async(out->grad_weight_p, (size_t)(dx * dx) * esz, s);
    if (out->grad_weight_g != nullptr) ok = ok && zero_fill_async(out->grad_weight_g, (size_t)(emission->dout * dx) * esz, s);
    if (out->grad_weight_q != nullptr) ok = ok && zero_fill_async(out->grad_weight_q, (size_t)(dx * dx) * esz, s);
    if (out->grad_scales != nullptr) ok = ok && zero_fill_async(out->grad_scales, 3 * esz, s);
    return ok ? AESMC_OK : AESMC_ERR_LAUNCH;
  }
  return dtype == AESMC_F32
             ? launch_affine_logweight_backward<float>(x_prev, x, y, y_stride_b, transition, emission, proposal,
                                                       scale_p, scale_g, scale_q, lw, lse, grad_lse, grad_lw, out, ws,
                                                       ws_bytes, B, K, s)
             : launch_affine_logweight_backward<double>(x_prev, x, y, y_stride_b, transition, emission, proposal,
                                                        scale_p, scale_g, scale_q, lw, lse, grad_lse, grad_lw, out, ws,
                                                        ws_bytes, B, K, s);
}


extern "C" int aesmc_affine_step_backward(
    int dtype, const void *x_prev, const void *x, const void *y, int64_t y_stride_b, const aesmc_affine_map *transition,
    const aesmc_affine_map *emission, const aesmc_affine_map *proposal, const void *scale_p, const void *scale_g,
    const void *scale_q, const void *lw, const void *lse, const void *grad_lse, const void *grad_lw,
    const void *grad_x, const aesmc_affine_logweight_grads *out, void *ws, size_t ws_bytes, int64_t B, int64_t K,
    void *stream) {
  if (out == nullptr) return AESMC_ERR_INVALID_ARGUMENT;
  // the draw leaves no gradient for x_t and the location gradients have no meaning here
  if (out->grad_x != nullptr || out->grad_loc_p != nullptr || out->grad_loc_g != nullptr || out->grad_loc_q != nullptr)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (x_prev == nullptr || x == nullptr || y == nullptr || transition == nullptr || emission == nullptr ||
      proposal == nullptr || scale_p == nullptr || scale_g == nullptr || scale_q == nullptr || ws == nullptr || B < 0 ||
      K < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (grad_lw == nullptr && grad_lse == nullptr && grad_x == nullptr) return AESMC_ERR_INVALID_ARGUMENT;
  if (grad_lse != nullptr && (lw == nullptr || lse == nullptr)) return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  const void *aligned[] = {x_prev, x, ws, grad_x, out->grad_x_prev};
  for (const void *ptr : aligned)
    if (ptr != nullptr && !aligned16(ptr)) return AESMC_ERR_INVALID_ARGUMENT;
  if (grad_x != nullptr && (grad_x == out->grad_x_prev)) return AESMC_ERR_INVALID_ARGUMENT;
  if (!lg_map_ok(transition) || !lg_map_ok(emission) || !lg_map_ok(proposal)) return AESMC_ERR_UNSUPPORTED;
  const int64_t dx = transition->dout;
  if (transition->din != dx || proposal->dout != dx || proposal->din != dx || emission->din != dx)
    return AESMC_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0 || K == 0) {      // empty sums
    const size_t esz = dtype == AESMC_F64 ? 8 : 4;
    bool ok = true;
    if (out->grad_weight_p != nullptr) ok = ok && zero_fill_async(out->grad_weight_p, (size_t)(dx * dx) * esz, s);
    if (out->grad_weight_g != nullptr) ok = ok && zero_fill_async(out->grad_weight_g, (size_t)(emission->dout * dx) * esz, s);
    if (out->grad_weight_q != nullptr) ok = ok && zero_fill_async(out->grad_weight_q, (size_t)(dx * dx) * esz, s);
    if (out->grad_scales != nullptr) ok = ok && zero_fill_async(out->grad_scales, 3 * esz, s);
    return ok ? AESMC_OK : AESMC_ERR_LAUNCH;
  }
  return dtype == AESMC_F32
             ? launch_affine_logweight_backward<float>(x_prev, x, y, y_stride_b, transition, emission, proposal,
                                                       scale_p, scale_g, scale_q, lw, lse, grad_lse, grad_lw, out, ws,
                                                       ws_bytes, B, K, s, true, grad_x)
             : launch_affine_logweight_backward<double>(x_prev, x, y, y_stride_b, transition, emission, proposal,
                                                        scale_p, scale_g, scale_q, lw, lse, grad_lse, grad_lw, out, ws,
                                                        ws_bytes, B, K, s, true, grad_x);
}


extern "C" size_t aesmc_affine_backward_workspace_bytes(int dtype, int64_t B, int64_t K) {
  const int64_t N = (B > 0 && K > 0) ? B * K : 0;
  return (lg_record_elems() + lg_row_elems(N, 3)) * (dtype == AESMC_F64 ? 8 : 4);
}

extern "C" int aesmc_particle_affine_backward(int dtype, const void *grad, const void *x, const aesmc_affine_map *map,
                                              void *out_grad_x, void *out_grad_weight, void *out_grad_offset, void *ws,
                                              size_t ws_bytes, int64_t B, int64_t K, void *stream) {
  const bool reduces = out_grad_weight != nullptr || out_grad_offset != nullptr;
  if (grad == nullptr || map == nullptr || map->weight == nullptr || B < 0 || K < 0 ||
      (out_grad_weight != nullptr && x == nullptr) || (reduces && ws == nullptr))
    return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  if (!aligned16(grad) || (x != nullptr && !aligned16(x)) || (out_grad_x != nullptr && !aligned16(out_grad_x)) ||
      (ws != nullptr && !aligned16(ws)))
    return AESMC_ERR_INVALID_ARGUMENT;
  if (!lg_map_ok(map)) return AESMC_ERR_UNSUPPORTED;
  if (out_grad_x == nullptr && !reduces) return AESMC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0 || K == 0) {   // empty sums: the weight gradient is zero, there are no rows
    if (out_grad_weight != nullptr &&
        !zero_fill_async(out_grad_weight, (size_t)(map->dout * map->din) * (dtype == AESMC_F64 ? 8 : 4), s))
      return AESMC_ERR_LAUNCH;
    if (out_grad_offset != nullptr && B > 0 &&
        !zero_fill_async(out_grad_offset, (size_t)(B * map->dout) * (dtype == AESMC_F64 ? 8 : 4), s))
      return AESMC_ERR_LAUNCH;
    return AESMC_OK;
  }
  return dtype == AESMC_F32
             ? launch_particle_affine_backward<float>(grad, x, map, out_grad_x, out_grad_weight, out_grad_offset, ws,
                                                      ws_bytes, B, K, s)
             : launch_particle_affine_backward<double>(grad, x, map, out_grad_x, out_grad_weight, out_grad_offset, ws,
                                                       ws_bytes, B, K, s);
}


extern "C" int64_t aesmc_affine_max_dim(void) { return kLgMaxDim; }

extern "C" int aesmc_particle_affine(int dtype, const void *x1, const aesmc_affine_map *m1, const void *x2,
                                     const aesmc_affine_map *m2, const void *base, void *out, int64_t B, int64_t K,
                                     void *stream) {
  if (x1 == nullptr || out == nullptr || m1 == nullptr || m1->weight == nullptr || B < 0 || K < 0 ||
      (x2 != nullptr && (m2 == nullptr || m2->weight == nullptr)))
    return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  if (!aligned16(x1) || !aligned16(out) || (x2 != nullptr && !aligned16(x2)) || (base != nullptr && !aligned16(base)))
    return AESMC_ERR_INVALID_ARGUMENT;
  if (!lg_map_ok(m1) || (x2 != nullptr && (!lg_map_ok(m2) || m2->dout != m1->dout))) return AESMC_ERR_UNSUPPORTED;
  if (B == 0 || K == 0) return AESMC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AESMC_F32 ? launch_particle_affine<float>(x1, m1, x2, m2, base, out, B, K, s)
                            : launch_particle_affine<double>(x1, m1, x2, m2, base, out, B, K, s);
}

extern "C" int aesmc_affine_normal_rsample(int dtype, const void *source, const aesmc_affine_map *map,
                                           const void *eps, const void *scale, void *out, int64_t B, int64_t K,
                                           void *stream) {
  if (source == nullptr || map == nullptr || map->weight == nullptr || eps == nullptr || scale == nullptr ||
      out == nullptr || B < 0 || K < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  if (!aligned16(source) || !aligned16(eps) || !aligned16(out) || out == eps || out == source)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (!lg_map_ok(map)) return AESMC_ERR_UNSUPPORTED;
  if (B == 0 || K == 0) return AESMC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AESMC_F32 ? launch_affine_rsample<float>(source, map, eps, scale, out, B, K, s)
                            : launch_affine_rsample<double>(source, map, eps, scale, out, B, K, s);
}

extern "C" int aesmc_affine_normal_propagate(int dtype, const void *x_prev, const void *eps, const void *y,
                                             int64_t y_stride_b, const aesmc_affine_map *transition,
                                             const aesmc_affine_map *emission, const aesmc_affine_map *proposal,
                                             const void *scale_p, const void *scale_g, const void *scale_q,
                                             void *out_x, void *out_lw, int64_t B, int64_t K, void *stream) {
  if (x_prev == nullptr || eps == nullptr || y == nullptr || transition == nullptr || emission == nullptr ||
      proposal == nullptr || scale_p == nullptr || scale_g == nullptr || scale_q == nullptr || out_lw == nullptr ||
      out_x == nullptr || B < 0 || K < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  if (!aligned16(x_prev) || !aligned16(eps) || !aligned16(out_x) || out_x == x_prev) return AESMC_ERR_INVALID_ARGUMENT;
  if (!lg_map_ok(transition) || !lg_map_ok(emission) || !lg_map_ok(proposal)) return AESMC_ERR_UNSUPPORTED;
  const int64_t dx = transition->dout;
  if (transition->din != dx || proposal->dout != dx || proposal->din != dx || emission->din != dx)
    return AESMC_ERR_UNSUPPORTED;
  if (B == 0 || K == 0) return AESMC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AESMC_F32
             ? launch_affine_logweight<float>(x_prev, eps, y, y_stride_b, transition, emission, proposal, scale_p,
                                              scale_g, scale_q, out_lw, B, K, s, out_x)
             : launch_affine_logweight<double>(x_prev, eps, y, y_stride_b, transition, emission, proposal, scale_p,
                                               scale_g, scale_q, out_lw, B, K, s, out_x);
}

extern "C" int aesmc_affine_normal_logweight(int dtype, const void *x_prev, const void *x, const void *y,
                                             int64_t y_stride_b, const aesmc_affine_map *transition,
                                             const aesmc_affine_map *emission, const aesmc_affine_map *proposal,
                                             const void *scale_p, const void *scale_g, const void *scale_q,
                                             void *out_lw, int64_t B, int64_t K, void *stream) {
  if (x_prev == nullptr || x == nullptr || y == nullptr || transition == nullptr || emission == nullptr ||
      proposal == nullptr || scale_p == nullptr || scale_g == nullptr || scale_q == nullptr || out_lw == nullptr ||
      B < 0 || K < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  if (!aligned16(x_prev) || !aligned16(x)) return AESMC_ERR_INVALID_ARGUMENT;
  if (!lg_map_ok(transition) || !lg_map_ok(emission) || !lg_map_ok(proposal)) return AESMC_ERR_UNSUPPORTED;
  // one latent extent: x' and x_prev are [B,K,dx]; the emission maps x' to the observation's dy values
  const int64_t dx = transition->dout;
  if (transition->din != dx || proposal->dout != dx || proposal->din != dx || emission->din != dx)
    return AESMC_ERR_UNSUPPORTED;
  if (B == 0 || K == 0) return AESMC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AESMC_F32
             ? launch_affine_logweight<float>(x_prev, x, y, y_stride_b, transition, emission, proposal, scale_p,
                                              scale_g, scale_q, out_lw, B, K, s)
             : launch_affine_logweight<double>(x_prev, x, y, y_stride_b, transition, emission, proposal, scale_p,
                                               scale_g, scale_q, out_lw, B, K, s);
}
