// Pieces shared by the backward kernels of the linear-Gaussian step (linear_gaussian_backward.hip: K11, K12, K14;
// linear_gaussian_step_backward.hip: K14 for rows of ten values): the matrix-core accumulators of the weight
// gradients and their records, the offsets' column sums, the step kernel's argument block.
#pragma once
#include "linear_gaussian.hpp"
namespace aesmc {

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

// A tile that lies inside one batch row needs no row bookkeeping: its column sums are what the ONES
// accumulate gathered in column 15 since the last flush.  Each wavefront writes its 16 sums to slot
// `wave` of the tile's row-sum record and clears them; the finishing launch adds the four slots (same
// test there: lg_single_row).  No barrier, no pass over the tile.
__host__ __device__ __forceinline__ bool lg_single_row(int64_t n0, uint32_t np, uint32_t K) {
  return (uint32_t)(n0 % K) + np <= K;
}
// (`slot`: 4 x 16 values in LDS, the four wavefronts' sums of one term; lg_store_column_sums adds them at the tile's end)
template <typename T>
__device__ __forceinline__ void lg_flush_column_sums(typename Mfma<T>::Acc &acc, T *__restrict__ slot, bool keep) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((lane & 15) == 15) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (keep) slot[wave * 16 + Mfma<T>::row(lane, r)] = acc[r];
      acc[r] = T(0);
    }
  }
}
// Behind the barrier that ends the tile: the tile's record of each wanted term, record[j] = the four wavefronts' sums
// added in wavefront order (what the finishing launch used to add, from four times the bytes).
template <typename T>
__device__ __forceinline__ void lg_store_column_sums(const T *__restrict__ slots, T *__restrict__ records, int terms) {
  if (threadIdx.x < 48) {
    const int t = threadIdx.x >> 4, j = threadIdx.x & 15;
    if ((terms >> t) & 1) {
      const T *c = slots + t * 64 + j;
      records[t * (kLgRowsMax * 16) + j] = ((c[0] + c[16]) + c[32]) + c[48];
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

struct LgBackwardOut {
  void *gxprev, *gx, *up, *ug, *uq, *ws;
  void *rows;        // nullptr, or the tiles' row-sum records [tile][3 terms p, g, q][kLgRowsMax][16]
  int row_terms;     // bit 0 / 1 / 2: term p / g / q wants its row sums
  const void *gx_in; // step kernel only: the gradient that arrives at x_t from later steps, or nullptr
  int want_scale_q;  // step kernel only: the proposal scale's gradient is wanted (costs the proposal's location)
  LgGather gat;      // step kernel only: `xprev` is the un-resampled latent, its rows fetched through gat.idx
  // step kernel only: the NEXT step's gradient with respect to the rows it resampled from x_t, one row per child
  // [B,K,dx], and where each particle's children end (child_end[b,k] = number of children of particles 0..k of row b):
  // torch.gather's backward (the sum of a particle's children) is formed here, where it is consumed
  const void *child_grad;
  const int32_t *child_end;
  int child_stage;   // a fourth LDS tile exists: the tile's children rows — one contiguous block — are staged through it
  int child_align;   // rows per 16 bytes' worth of alignment: a staged block starts at a multiple of this many rows
  // step kernel only: the records an earlier launch with the same parameters left in ITS workspace (aesmc_affine_chain):
  // workgroup w adds records w, w + grid, ... to its own, so the weights' gradients of a run of steps are finished once
  const void *carry;
  int carry_records;
  // step kernel, rows form: room behind the records and row sums for the three maps' interleaved weight pairs (3 x
  // kLgPairFloats floats; nullptr: a workspace of the older size — the location chains then run one output at a time)
  float *pairs;
  int pairs_ready;      // the pairs at `pairs` are an earlier call's of the same run (aesmc_affine_chain::pairs_in): not rebuilt
};
constexpr int kLgPairFloats = (kLgMaxDim / 2) * kLgMaxDim * 2;      // (== kPairFloats of linear_gaussian_fused.hpp)

constexpr int kLgChildLimit = 32;   // children a lane sums by itself; longer runs (a collapsed system) take the wavefront
                                    // (8: the bench shape's healthy ancestry 338 -> 348 us, a collapsed one 456 -> 437)
constexpr int kLgChildTrip = 1;     // of them per trip out of the staged block (2: the same time, ten more registers; 4: slower)

// linear_gaussian_step_backward.hip: the step's backward for rows of an even number (2 .. 14) of float32 values (the second form of K14)
bool affine_step_backward_rows_covers(const aesmc_affine_map *mp, const aesmc_affine_map *mg, const aesmc_affine_map *mq,
                                      int64_t B, int64_t K);
unsigned affine_step_backward_rows_grid(int64_t B, int64_t K, int64_t d);
bool affine_step_backward_rows_pairs();      // does the rows form take the interleaved weight pairs?
int affine_step_backward_forced_grid();      // test hook: > 0 pins the grid of both forms
int launch_affine_step_backward_rows(const float *xprev, const float *x, const float *y, int64_t y_sb, const LgMap &mp,
                                     const LgMap &mg, const LgMap &mq, const float *sp, const float *sg, const float *sq,
                                     const float *lw, const float *lse, const float *grad_lse, const float *grad_lw,
                                     const LgBackwardOut &out, int64_t N, uint32_t K, unsigned grid, hipStream_t stream);

}  // namespace aesmc
