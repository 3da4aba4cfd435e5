// K14 for rows of D float32 values, D even, 2 .. 14 (ten: the BASELINE shapes): the whole backward of a linear-Gaussian SMC step whose x_t
// is the proposal's reparameterised draw — affine_step_backward_kernel's arithmetic (linear_gaussian_backward.hip: the
// same fma chains, the same matrix-core accumulation order, the same records: every output equal bit for bit when the
// two run on the same grid), rearranged the way the forward step was (linear_gaussian_fused.hip):
//   * a wavefront works on its own 64 particles from the rows' arrival to the gradient's store: the rows of x_{t-1}
//     (through the ancestors) and of x_t come straight into the lane's registers, one tile ahead; what the matrix cores
//     need transposed (u, x) goes through an LDS area only this wavefront touches — no workgroup barrier inside a tile
//     (one at its end, where the four wavefronts' column sums meet, when an offset's gradient is wanted);
//   * the weights are SCALAR operands (s_load + v_fmac_f32 v, s, v): the launch's maps are the same for every particle,
//     so neither LDS reads nor vector registers are spent on them; offsets, the observation, lse and its gradient are
//     per batch row, a tile lies inside one row (K a multiple of 256): scalar loads as well;
//   * the children's rows (the gather's backward, folded in) are one contiguous block per WAVEFRONT, sent for a tile
//     ahead by loads that write LDS directly; a lane sums its run out of it in k order.
// Reference: autograd of aesmc/state.py:114-155,179 and aesmc/inference.py:108-130 for one timestep.
#include "linear_gaussian_backward.hpp"
#include "linear_gaussian_fused.hpp"

#include <atomic>

namespace aesmc {

typedef const float __attribute__((address_space(4))) sb_cfloat;
typedef float sb_f2 __attribute__((ext_vector_type(2)));
typedef float sb_f4 __attribute__((ext_vector_type(4)));
typedef sb_f4 sb_f4_a4 __attribute__((aligned(4)));      // a 16-byte global access at 4-byte alignment (hardware: unaligned mode)
typedef sb_f2 sb_f2_a4 __attribute__((aligned(4)));

#ifdef AESMC_K14_NO_PACKED      /* measurement build: the adjoints as v_fmac_f32 chains (round 4's form) */
constexpr bool kSbPacked = false;
#else
constexpr bool kSbPacked = true;  // the three adjoints advance two outputs per instruction (v_pk_fma_f32: sb_pk_tx2)
#endif
constexpr int kSbChildRows = 96;         // rows of a wavefront's staged block of children (mean 64; what does not fit: from HBM)
constexpr int kSbSlots = 2 * 192;        // the tile's column sums (three terms x four wavefronts x 16), two tiles' worth
template <int D> struct Sb {             // D: the rows' extent (both the latent's and the observation's)
  static_assert(D >= 2 && D <= 14 && D % 2 == 0, "even extents below 16: rows in 8-byte pieces, column 15 of a matrix tile free");
  static constexpr int kTile = 64 * D + 16;  // a wavefront's u / x area (+16: the matrix-core operand reads run past a row's end)
  static constexpr int kWave = 2 * kTile + kSbChildRows * D;      // floats per wavefront
  static constexpr int kOnes = 15 * 4 * D + 8;      // 1.0 wherever a lane of column 15 reads its "x" operand (sb_outer)
  static constexpr size_t kLds = sizeof(float) * (kSbSlots + kOnes + 4 * (size_t)kWave);
  static constexpr int kPerCu = D <= 10 ? 3 : 2;      // workgroups per CU (= wavefronts per SIMD) the registers allow
  static_assert(kLds * kPerCu <= 160 * 1024, "LDS of the resident workgroups");
};

// tile / tpr with mul = floor(2^32 / tpr): the high product is the quotient or one less (tile < 2^23)
__device__ __forceinline__ uint32_t sb_row_of(uint32_t tile, uint32_t tpr, uint32_t mul) {
  uint32_t q = __umulhi(tile, mul);
  q += (tile - q * tpr >= tpr) ? 1u : 0u;
  return q;
}

// the adjoint's step: acc_i += W[j][i] u_j then acc_i += W[j+1][i] u_{j+1} for N consecutive i — rows j, j + 1 of W as they
// lie (the forward chains' counterpart, fused_fmac_sx2, is linear_gaussian_fused.hpp's)
template <int N>
__device__ __forceinline__ void sb_fmac_tx2(float *acc, const float *w0, const float *w1, float u0, float u1) {
  static_assert(N >= 1 && N <= 5, "one to five inputs per statement");
  if constexpr (N == 5) {
    asm("v_fmac_f32 %0, %5, %15\n\tv_fmac_f32 %1, %6, %15\n\tv_fmac_f32 %2, %7, %15\n\tv_fmac_f32 %3, %8, %15\n\t"
        "v_fmac_f32 %4, %9, %15\n\t"
        "v_fmac_f32 %0, %10, %16\n\tv_fmac_f32 %1, %11, %16\n\tv_fmac_f32 %2, %12, %16\n\tv_fmac_f32 %3, %13, %16\n\t"
        "v_fmac_f32 %4, %14, %16"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4])
        : "s"(w0[0]), "s"(w0[1]), "s"(w0[2]), "s"(w0[3]), "s"(w0[4]), "s"(w1[0]), "s"(w1[1]), "s"(w1[2]), "s"(w1[3]),
          "s"(w1[4]), "v"(u0), "v"(u1));
  } else if constexpr (N == 4) {
    asm("v_fmac_f32 %0, %4, %12\n\tv_fmac_f32 %1, %5, %12\n\tv_fmac_f32 %2, %6, %12\n\tv_fmac_f32 %3, %7, %12\n\t"
        "v_fmac_f32 %0, %8, %13\n\tv_fmac_f32 %1, %9, %13\n\tv_fmac_f32 %2, %10, %13\n\tv_fmac_f32 %3, %11, %13"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
        : "s"(w0[0]), "s"(w0[1]), "s"(w0[2]), "s"(w0[3]), "s"(w1[0]), "s"(w1[1]), "s"(w1[2]), "s"(w1[3]), "v"(u0), "v"(u1));
  } else if constexpr (N == 3) {
    asm("v_fmac_f32 %0, %3, %9\n\tv_fmac_f32 %1, %4, %9\n\tv_fmac_f32 %2, %5, %9\n\t"
        "v_fmac_f32 %0, %6, %10\n\tv_fmac_f32 %1, %7, %10\n\tv_fmac_f32 %2, %8, %10"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2])
        : "s"(w0[0]), "s"(w0[1]), "s"(w0[2]), "s"(w1[0]), "s"(w1[1]), "s"(w1[2]), "v"(u0), "v"(u1));
  } else if constexpr (N == 2) {
    asm("v_fmac_f32 %0, %2, %6\n\tv_fmac_f32 %1, %3, %6\n\t"
        "v_fmac_f32 %0, %4, %7\n\tv_fmac_f32 %1, %5, %7"
        : "+v"(acc[0]), "+v"(acc[1])
        : "s"(w0[0]), "s"(w0[1]), "s"(w1[0]), "s"(w1[1]), "v"(u0), "v"(u1));
  } else {
    asm("v_fmac_f32 %0, %1, %3\n\tv_fmac_f32 %0, %2, %4" : "+v"(acc[0]) : "s"(w0[0]), "s"(w1[0]), "v"(u0), "v"(u1));
  }
}
// The same step on the packed pipe: (acc_i, acc_{i+1}) += (W[j][i], W[j][i+1]) u_j, then the same with row j + 1 — the two
// weights of a pair are neighbours in a row of W as it lies, u_j / u_{j+1} the low / high half of one register pair.  Each
// half is v_fmac_f32's fused multiply-add, rows ascending: the same bits, half the instructions.  NP pairs of outputs
// (1 .. 3) per statement.
template <int NP>
__device__ __forceinline__ void sb_pk_tx2(lg_f2 *acc, const float *w0, const float *w1, lg_f2 u) {
  static_assert(NP >= 1 && NP <= 3, "one to three output pairs per statement");
#define SB_W(row, i) "s"(lg_f2{row[2 * (i)], row[2 * (i) + 1]})
  if constexpr (NP == 3) {
    asm("v_pk_fma_f32 %0, %3, %9, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %4, %9, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %2, %5, %9, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %6, %9, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %1, %7, %9, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %2, %8, %9, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2])
        : SB_W(w0, 0), SB_W(w0, 1), SB_W(w0, 2), SB_W(w1, 0), SB_W(w1, 1), SB_W(w1, 2), "v"(u));
  } else if constexpr (NP == 2) {
    asm("v_pk_fma_f32 %0, %2, %6, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %3, %6, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %4, %6, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %1, %5, %6, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "+v"(acc[0]), "+v"(acc[1])
        : SB_W(w0, 0), SB_W(w0, 1), SB_W(w1, 0), SB_W(w1, 1), "v"(u));
  } else {
    asm("v_pk_fma_f32 %0, %1, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "+v"(acc[0]) : SB_W(w0, 0), SB_W(w1, 0), "v"(u));
  }
#undef SB_W
}
template <int D, int P0 = 0>
__device__ __forceinline__ void sb_adjoint_pk(lg_f2 *acc, const float *w0, const float *w1, lg_f2 u) {
  if constexpr (2 * P0 < D) {
    constexpr int REM = D / 2 - P0, NP = REM >= 3 && REM != 4 ? 3 : (REM >= 2 ? 2 : 1);      // (4 = 2 + 2, never 3 + 1)
    sb_pk_tx2<NP>(acc + P0, w0 + 2 * P0, w1 + 2 * P0, u);
    sb_adjoint_pk<D, P0 + NP>(acc, w0, w1, u);
  }
}

template <int D, int I0 = 0>
__device__ __forceinline__ void sb_adjoint_pair(float *acc, const float *w0, const float *w1, float u0, float u1) {
  if constexpr (I0 < D) {
    constexpr int N = fused_group(D - I0);
    sb_fmac_tx2<N>(acc + I0, w0 + I0, w1 + I0, u0, u1);
    sb_adjoint_pair<D, I0 + N>(acc, w0, w1, u0, u1);
  }
}

__device__ __forceinline__ float sb_uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// a row of D values at 8-byte alignment: 16-byte pieces, then an 8-byte one where D is not a multiple of four
template <int D> __device__ __forceinline__ void sb_load_row(const float *__restrict__ src, float (&v)[D]) {
#pragma unroll
  for (int e = 0; e + 4 <= D; e += 4) {
    const sb_f4 a = *reinterpret_cast<const sb_f4_a4 *>(src + e);
    v[e] = a[0]; v[e + 1] = a[1]; v[e + 2] = a[2]; v[e + 3] = a[3];
  }
  if constexpr (D % 4 != 0) {
    const sb_f2 c = *reinterpret_cast<const sb_f2_a4 *>(src + (D & ~3));
    v[D - 2] = c[0]; v[D - 1] = c[1];
  }
}
template <int D> __device__ __forceinline__ void sb_store_row(float *__restrict__ dst, const float (&v)[D]) {
#pragma unroll
  for (int e = 0; e + 4 <= D; e += 4) {
    sb_f4 a;
    a[0] = v[e]; a[1] = v[e + 1]; a[2] = v[e + 2]; a[3] = v[e + 3];
    *reinterpret_cast<sb_f4_a4 *>(dst + e) = a;
  }
  if constexpr (D % 4 != 0) {
    sb_f2 c;
    c[0] = v[D - 2]; c[1] = v[D - 1];
    *reinterpret_cast<sb_f2_a4 *>(dst + (D & ~3)) = c;
  }
}
// the same row in LDS: 8-byte pieces (rows of 4 D bytes)
template <int D> __device__ __forceinline__ void sb_lds_row(const float *row, float (&v)[D]) {
#pragma unroll
  for (int j = 0; j < D / 2; ++j) {
    const sb_f2 q = reinterpret_cast<const sb_f2 *>(row)[j];
    v[2 * j] = q[0];
    v[2 * j + 1] = q[1];
  }
}
template <int D> __device__ __forceinline__ void sb_lds_put(float *row, const float (&v)[D]) {
#pragma unroll
  for (int j = 0; j < D / 2; ++j) {
    sb_f2 q;
    q[0] = v[2 * j];
    q[1] = v[2 * j + 1];
    reinterpret_cast<sb_f2 *>(row)[j] = q;
  }
}
// values a compiler barrier pins where they are (one empty statement per register: any extent)
template <int N> __device__ __forceinline__ void sb_pin(float (&v)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) asm volatile("" : "+v"(v[j]));
}

// acc[j][i] += sum over the wavefront's 64 particles of tg[p][j] tx[p][i]: lg_outer_accumulate_own's whole-tile branch
// with ONES (the lanes of column 15 feed 1: acc[j][15] gathers the column sums of tg), on the wavefront's own area
// (`ones`: an area holding 1.0 at every offset the sixteen reads use — the lanes of column 15 read it instead of tx)
template <int D>
__device__ __forceinline__ void sb_outer(const float *tg, const float *tx, const float *ones, uint32_t lane,
                                         Mfma<float>::Acc &acc) {
  const uint32_t col = lane & 15u, e = (lane >> 4) * D + col;
  const float *ta = tg + e, *tb = col == 15u ? ones : tx + e;
  // the NEXT group's eight operands are sent for before this group's four products: an LDS round trip per group passes
  // under matrix work instead of in front of it (counters: 55 % of this kernel's wave cycles are waits, three wavefronts
  // per SIMD do not cover one another's)
  float a[2][4], b[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    a[0][t] = ta[t * 4 * D];
    b[0][t] = tb[t * 4 * D];
  }
#pragma unroll
  for (int group = 0; group < 4; ++group) {
    if (group + 1 < 4) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[(group + 1) & 1][t] = ta[(4 * (group + 1) + t) * 4 * D];
        b[(group + 1) & 1][t] = tb[(4 * (group + 1) + t) * 4 * D];
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = Mfma<float>::fma(a[group & 1][t], b[group & 1][t], acc);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The launch's operands.  The kernel reads them where it uses them, through the kernel-argument segment (scalar loads
// out of the constant cache), instead of holding a hundred scalar registers of pointers across the tile loop — those
// registers are what the weights' rows travel in.
struct SbArgs {
  const float *xprev, *x, *y;
  int64_t y_sb;
  const float *wp, *wg, *wq, *offp, *offg, *offq;
  int64_t offp_sb, offg_sb, offq_sb;
  const float *sp, *sg, *sq;
  const float *lw_src, *glw_src, *lse, *grad_lse;      // (lw_src / glw_src: a readable stand-in when the term is absent)
  float *gxprev;
  const float *gx_in;
  float *rows, *ws;
  const float *carry;
  const float *child_rows;
  const int32_t *child_end;
  const float *pairs;      // PAIRED: the maps' interleaved weight pairs (transition, emission, proposal), else unused
  const int64_t *anc;
  int32_t *flags;
  int64_t N, tiles;
  uint32_t K, tpr, tpr_mul;      // tiles per batch row (K / 256) and floor(2^32 / tpr)
  int32_t row_terms, want_sq, has_lse, has_glw, carry_records;
};
typedef const SbArgs __attribute__((address_space(4))) sb_cargs;

// PAIRED: the location chains advance two outputs per instruction too (weights from the interleaved pairs the host's
// launch wrote into the workspace just before this one: fused_pk_pair, linear_gaussian_fused.hpp) — the same bits.
template <int D, bool GATHER, bool FOLDS, bool PAIRED>
__global__ __launch_bounds__(kLgBlock, Sb<D>::kPerCu) void affine_step_backward_rows_kernel(SbArgs unused_by_name) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char sb_smem[];
  constexpr int kSbTile = Sb<D>::kTile, kSbWave = Sb<D>::kWave, kSbOnes = Sb<D>::kOnes;
  // the argument block, re-derived (opaquely) wherever it is read: a field is loaded where it is used, never hoisted
  unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
#define SB_A() ([&]() -> sb_cargs * { asm volatile("" : "+s"(ka)); return (sb_cargs *)ka; }())
  float *slots = reinterpret_cast<float *>(sb_smem);
  uint32_t tid = threadIdx.x;
  asm volatile("" : "+v"(tid));      // (what derives from the lane's index is recomputed per tile, not held across the loop)
  const uint32_t lane = tid & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float *ones = slots + kSbSlots;
  float *tu = ones + kSbOnes + wave * kSbWave, *tx = tu + kSbTile, *tc = tx + kSbTile;
  for (uint32_t i = threadIdx.x; i < (uint32_t)kSbOnes; i += kLgBlock) ones[i] = 1.0f;
  __syncthreads();
  float inv_var_p, inv_var_g, inv_s_p, inv_s_g, inv_s_q;
  {
    sb_cargs *A = SB_A();
    const float s_p = ((sb_cfloat *)A->sp)[0], s_g = ((sb_cfloat *)A->sg)[0], s_q = ((sb_cfloat *)A->sq)[0];
    inv_var_p = sb_uniform(1.0f / (s_p * s_p));
    inv_var_g = sb_uniform(1.0f / (s_g * s_g));
    inv_s_p = sb_uniform(1.0f / s_p);
    inv_s_g = sb_uniform(1.0f / s_g);
    inv_s_q = sb_uniform(1.0f / s_q);
  }
  Mfma<float>::Acc acc_a = {0.0f, 0.0f, 0.0f, 0.0f}, acc_c = acc_a, acc_q = acc_a;
  float scale_acc[3] = {0.0f, 0.0f, 0.0f};
  const uint32_t step = gridDim.x;
  int bad = 0;

  // ---- what waits in registers for the NEXT tile ----------------------------------------------------------------
  float xp_n[D], xt_n[D], lw_n = 0.0f, glw_n = 0.0f;
  int64_t anc_n = 0;                                   // the tile after the next one's ancestors (raw)
  int32_t rb_c = 0, re_c = 0, rb_n = 0, re_n = 0;      // child_end entries (before the lane's particle, at it): this tile's, the next one's
  uint32_t blk_lo = 0, blk_hi = 0;                     // the staged block of THIS tile: flat rows [blk_lo, blk_hi)
  auto anc_load = [&](uint32_t tile) -> int64_t { return SB_A()->anc[(uint64_t)tile * kLgBlock + tid]; };
  auto raw_load = [&](uint32_t tile, int32_t &before, int32_t &end) {
    sb_cargs *A = SB_A();
    const uint32_t n = tile * kLgBlock + tid;
    const int32_t *child_end = A->child_end;
    end = child_end[n];
    const uint32_t k0 = (tile - sb_row_of(tile, A->tpr, A->tpr_mul) * A->tpr) * kLgBlock;
    before = child_end[k0 + tid == 0u ? n : n - 1u];      // (no branch around the load: a row's first particle re-reads its own entry)
  };
  auto rows_prefetch = [&](uint32_t tile, int64_t araw) {
    sb_cargs *A = SB_A();
    const uint32_t n = tile * kLgBlock + tid, K = A->K;
    const float *src;
    if constexpr (GATHER) {
      int64_t a = araw;
      if (a < 0 || a >= (int64_t)K) {      // K2 writes K for a degenerate row (flagged there); never fault on it
        bad = 1;
        a = a < 0 ? 0 : (int64_t)K - 1;
      }
      const uint32_t b = sb_row_of(tile, A->tpr, A->tpr_mul);
      src = A->xprev + ((uint64_t)(b * K + (uint32_t)a)) * D;
    } else {
      src = A->xprev + (uint64_t)n * D;
    }
    sb_load_row(src, xp_n);
    sb_load_row(A->x + (uint64_t)n * D, xt_n);
    lw_n = A->lw_src[n];
    glw_n = A->glw_src[n];
  };
  // flat rows [lo, hi) of the children of the lane's particle of `tile`, from its two entries (child_range of the first form)
  auto child_range = [&](uint32_t tile, int32_t before, int32_t end_, uint32_t &lo, uint32_t &hi) {
    sb_cargs *A = SB_A();
    const uint32_t K = A->K;
    const uint32_t n0 = tile * kLgBlock;
    const uint32_t b0 = sb_row_of(tile, A->tpr, A->tpr_mul), k0 = n0 - b0 * K;
    const bool first_of_row = k0 + tid == 0u;
    const uint32_t base = b0 * K;
    const uint32_t end = (uint32_t)min(max(end_, 0), (int32_t)K);
    hi = base + end;
    lo = base + min((uint32_t)max(first_of_row ? 0 : before, 0), end);
  };
  // the wavefront's block: from its first particle's run to its last one's, begun at an even row (16 bytes), cut at
  // kSbChildRows; sent for by loads that write LDS directly
  auto block_send = [&](uint32_t lo, uint32_t hi, uint32_t &blo, uint32_t &bhi) {
    sb_cargs *A = SB_A();
    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)lo, 0), last = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
    blo = first & ~1u;
    bhi = last > blo ? min(min((last + 1u) & ~1u, (uint32_t)A->N), blo + (uint32_t)kSbChildRows) : blo;
    const uint32_t nvec = (bhi - blo) * D / 4;      // an even number of rows: whole 16-byte vectors
    const float *src = A->child_rows + (uint64_t)blo * D;
#pragma unroll 1
    for (uint32_t v0 = 0; v0 < nvec; v0 += 64) {
      const uint32_t v = v0 + lane;
      if (v < nvec)
        __builtin_amdgcn_global_load_lds(src + (size_t)v * 4, (__attribute__((address_space(3))) void *)(tc + (size_t)v0 * 4), 16, 0, 0);
    }
  };

  const uint32_t tiles = (uint32_t)SB_A()->tiles;      // (the host: K, hence N, a multiple of 256; N < 2^31)
  const uint32_t first_tile = blockIdx.x;
  if (first_tile < tiles) {
    if constexpr (FOLDS) {
      raw_load(first_tile, rb_c, re_c);
      if (first_tile + step < tiles) raw_load(first_tile + step, rb_n, re_n);
      uint32_t lo, hi;
      child_range(first_tile, rb_c, re_c, lo, hi);
      block_send(lo, hi, blk_lo, blk_hi);
      asm volatile("" ::: "memory");
    }
    int64_t a0 = 0;
    if constexpr (GATHER) {
      a0 = anc_load(first_tile);
      if (first_tile + step < tiles) anc_n = anc_load(first_tile + step);
    }
    rows_prefetch(first_tile, a0);
  }
  float g_old[D];      // the last tile's gradient rows: stored behind the next tile's first wait, not in front of it
#pragma unroll
  for (int j = 0; j < D; ++j) g_old[j] = 0.0f;
  uint32_t old_tile = 0xffffffffu, trip = 0;

  for (uint32_t tile = first_tile; tile < tiles; tile += step) {
    // everything sent for during the last tile has landed (the block of children writes LDS from the vector-memory side)
    // (the prefetched registers are operands of the wait: the compiler then asks for them in front of it and never
    //  again behind it, where its own wait would also cover the stores that follow)
    if constexpr (D == 10) {
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(xp_n[0]), "+v"(xp_n[1]), "+v"(xp_n[2]), "+v"(xp_n[3]), "+v"(xp_n[4]), "+v"(xp_n[5]), "+v"(xp_n[6]),
                     "+v"(xp_n[7]), "+v"(xp_n[8]), "+v"(xp_n[9]), "+v"(xt_n[0]), "+v"(xt_n[1]), "+v"(xt_n[2]), "+v"(xt_n[3]),
                     "+v"(xt_n[4]), "+v"(xt_n[5]), "+v"(xt_n[6]), "+v"(xt_n[7]), "+v"(xt_n[8]), "+v"(xt_n[9]), "+v"(lw_n),
                     "+v"(glw_n), "+v"(rb_n), "+v"(re_n), "+v"(anc_n)
                   :
                   : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(lw_n), "+v"(glw_n), "+v"(rb_n), "+v"(re_n), "+v"(anc_n) : : "memory");
      sb_pin(xp_n);
      sb_pin(xt_n);
    }
    uint32_t b;
    float g;
    float xp[D], xt[D], w[D];
    {
      sb_cargs *A = SB_A();
      float *gxprev = A->gxprev;
      if (old_tile != 0xffffffffu && gxprev != nullptr) sb_store_row(gxprev + ((uint64_t)old_tile * kLgBlock + tid) * D, g_old);
      b = sb_row_of(tile, A->tpr, A->tpr_mul);
#pragma unroll
      for (int j = 0; j < D; ++j) {
        xp[j] = xp_n[j];
        xt[j] = xt_n[j];
      }
      g = A->has_glw ? glw_n : 0.0f;
      if (A->has_lse) {
        const float lse_b = ((sb_cfloat *)A->lse)[b], glse_b = ((sb_cfloat *)A->grad_lse)[b];
        g = g + glse_b * Num<float>::exp(lw_n - lse_b);
      }
      // ---- w = the gradient arriving at x_t: from later steps' densities (gx_in) and from the particle's children ----
      const float *gx_in = A->gx_in;
      if (gx_in != nullptr) {
        sb_load_row(gx_in + ((uint64_t)tile * kLgBlock + tid) * D, w);
      } else {
#pragma unroll
        for (int j = 0; j < D; ++j) w[j] = 0.0f;
      }
    }
    if (FOLDS) {
      const float *child_rows = SB_A()->child_rows;
      uint32_t lo, hi;
      child_range(tile, rb_c, re_c, lo, hi);
      const uint32_t own_last = min(hi, lo + (uint32_t)kLgChildLimit);
      float acc[D];
#pragma unroll
      for (int j = 0; j < D; ++j) acc[j] = 0.0f;
      uint32_t c = lo;
      const uint32_t in_lds = lo >= blk_lo ? min(own_last, blk_hi) : lo;
      for (; c < in_lds; ++c) {      // in k order: ((0 + c0) + c1) + c2 ...
        float row[D];
        sb_lds_row(tc + (c - blk_lo) * D, row);
#pragma unroll
        for (int j = 0; j < D; ++j) acc[j] = acc[j] + row[j];
      }
      c = min(c, max(in_lds, lo));
      for (; c < own_last; ++c) {      // (rows the block had no room for)
        float row[D];
        sb_load_row(child_rows + (uint64_t)c * D, row);
#pragma unroll
        for (int j = 0; j < D; ++j) acc[j] = acc[j] + row[j];
      }
      uint64_t todo = __ballot(own_last < hi);
      while (todo != 0) {        // wavefront-uniform: every lane helps the lane whose run is long (a collapsed system)
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t from = (uint32_t)__shfl((int)own_last, leader, kWave), to = (uint32_t)__shfl((int)hi, leader, kWave);
        float part[D];
#pragma unroll
        for (int j = 0; j < D; ++j) part[j] = 0.0f;
        for (uint32_t cc = from + lane; cc < to; cc += kWave) {
          float row[D];
          sb_load_row(child_rows + (uint64_t)cc * D, row);
#pragma unroll
          for (int j = 0; j < D; ++j) part[j] += row[j];
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
#pragma unroll
          for (int off = kWave / 2; off > 0; off >>= 1) part[j] += __shfl_xor(part[j], off, kWave);
          if ((int)lane == leader) acc[j] += part[j];
        }
        todo &= todo - 1;
      }
#pragma unroll
      for (int j = 0; j < D; ++j) w[j] = w[j] + acc[j];
    }
    // ---- the next tile's inputs go out now and fly during this tile's arithmetic ------------------------------------
    {
      const uint32_t next = tile + step, after = next + step;
      // (w is complete here: nothing that waits for a load of THIS tile may sink behind the loads sent for the next one,
      //  where its wait would be a wait for all of them)
      if constexpr (D == 10) {
        asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]),
                          "+v"(w[9]), "+v"(g)
                     :
                     : "memory");
      } else {
        asm volatile("" : "+v"(g) : : "memory");
        sb_pin(w);
      }
      if (next < tiles) {
        if constexpr (FOLDS) {
          // (this wavefront has taken its sums out of its block: the area is free; a wavefront's LDS accesses keep their order)
          lg_wave_fence();
          uint32_t lo, hi;
          child_range(next, rb_n, re_n, lo, hi);
          block_send(lo, hi, blk_lo, blk_hi);
          asm volatile("" ::: "memory");
          rb_c = rb_n;
          re_c = re_n;
          if (after < tiles) raw_load(after, rb_n, re_n);
        }
        rows_prefetch(next, anc_n);
        if constexpr (GATHER) {
          if (after < tiles) anc_n = anc_load(after);
        }
      }
      asm volatile("" ::: "memory");
    }
    // acc_j = offset_j + sum_i W[j][i] in_i (ascending i): a location.  The weights are scalar operands, two rows of
    // ten at a time, the next two sent for before this pair's multiply-adds.
    auto chain = [&](const float *wptr, int which, const float *off_ptr, int64_t off_sb, const float (&in)[D], float (&acc)[D]) {
      sb_cfloat *W = PAIRED ? (sb_cfloat *)SB_A()->pairs + which * kPairFloats : (sb_cfloat *)wptr;
      if (off_ptr != nullptr) {
        sb_cfloat *off = (sb_cfloat *)off_ptr + (int64_t)b * off_sb;
#pragma unroll
        for (int j = 0; j < D; ++j) acc[j] = off[j];
      } else {
#pragma unroll
        for (int j = 0; j < D; ++j) acc[j] = 0.0f;
      }
      float wa[2 * D], wb[2 * D];
#pragma unroll
      for (int e = 0; e < 2 * D; ++e) wa[e] = W[e];
      lg_f2 in2[D / 2], acc2[D / 2];
      if constexpr (PAIRED) {
#pragma unroll
        for (int i = 0; i < D / 2; ++i) {
          in2[i] = lg_f2{in[2 * i], in[2 * i + 1]};
          acc2[i] = lg_f2{acc[2 * i], acc[2 * i + 1]};
        }
      }
#pragma unroll
      for (int jb = 0; jb < D; jb += 2) {
        float (&cur)[2 * D] = (jb & 2) ? wb : wa;
        float (&nxt)[2 * D] = (jb & 2) ? wa : wb;
        if (jb + 2 < D) {
#pragma unroll
          for (int e = 0; e < 2 * D; ++e) nxt[e] = W[(jb + 2) * D + e];      // (pairs: the next output pair's 2 D floats)
        }
        if constexpr (PAIRED) fused_pk_pair<D>(acc2[jb / 2], cur, in2);
        else fused_pair<D>(acc[jb], acc[jb + 1], cur, cur + D, in);
      }
      if constexpr (PAIRED) {
#pragma unroll
        for (int i = 0; i < D / 2; ++i) {
          acc[2 * i] = acc2[i][0];
          acc[2 * i + 1] = acc2[i][1];
        }
      }
    };
    // acc_i += sum_j W[j][i] u_j (ascending j): an adjoint
    auto adjoint = [&](const float *wptr, const float (&u)[D], float (&acc)[D]) {
      sb_cfloat *W = (sb_cfloat *)wptr;
      float wa[2 * D], wb[2 * D];
#pragma unroll
      for (int e = 0; e < 2 * D; ++e) wa[e] = W[e];
      lg_f2 acc2[D / 2];
      if constexpr (kSbPacked) {
#pragma unroll
        for (int i = 0; i < D / 2; ++i) acc2[i] = lg_f2{acc[2 * i], acc[2 * i + 1]};
      }
#pragma unroll
      for (int jb = 0; jb < D; jb += 2) {
        float (&cur)[2 * D] = (jb & 2) ? wb : wa;
        float (&nxt)[2 * D] = (jb & 2) ? wa : wb;
        if (jb + 2 < D) {
#pragma unroll
          for (int e = 0; e < 2 * D; ++e) nxt[e] = W[(jb + 2) * D + e];
        }
        if constexpr (kSbPacked) sb_adjoint_pk<D>(acc2, cur, cur + D, lg_f2{u[jb], u[jb + 1]});
        else sb_adjoint_pair<D>(acc, cur, cur + D, u[jb], u[jb + 1]);
      }
      if constexpr (kSbPacked) {
#pragma unroll
        for (int i = 0; i < D / 2; ++i) {
          acc[2 * i] = acc2[i][0];
          acc[2 * i + 1] = acc2[i][1];
        }
      }
    };
    float *slot = slots + (trip & 1u) * 192;
    ++trip;
    float u[D], gprev[D];
#pragma unroll
    for (int j = 0; j < D; ++j) gprev[j] = 0.0f;
    // ---- emission term: u = g (y - loc_g) / s_g^2;  w += C^T u ----------------------------------------------------------
    {
      sb_cargs *A = SB_A();
      chain(A->wg, 1, A->offg, A->offg_sb, xt, u);
      sb_cfloat *yrow = (sb_cfloat *)A->y + (int64_t)b * A->y_sb;
      float q = 0.0f;
      const float scaled = g * inv_var_g;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float diff = yrow[j] - u[j];
        q = fma_t(diff, diff, q);
        u[j] = scaled * diff;
      }
      scale_acc[1] += g * (q * inv_var_g * inv_s_g - (float)D * inv_s_g);
      {
        sb_lds_put(tu + lane * D, u);
        sb_lds_put(tx + lane * D, xt);
      }
      lg_wave_fence();
      adjoint(A->wg, u, w);
      sb_outer<D>(tu, tx, ones, lane, acc_c);
      lg_flush_column_sums<float>(acc_c, slot + 64, (A->row_terms & 2) != 0);
      lg_wave_fence();
    }
    // ---- transition term: u = g (x - loc_p) / s_p^2;  w -= u ------------------------------------------------------------
    {
      sb_cargs *A = SB_A();
      chain(A->wp, 0, A->offp, A->offp_sb, xp, u);
      float q = 0.0f;
      const float scaled = g * inv_var_p;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float diff = xt[j] - u[j];
        q = fma_t(diff, diff, q);
        u[j] = scaled * diff;
        w[j] = w[j] - u[j];
      }
      scale_acc[0] += g * (q * inv_var_p * inv_s_p - (float)D * inv_s_p);
      {
        sb_lds_put(tu + lane * D, u);
        sb_lds_put(tx + lane * D, xp);
      }
      lg_wave_fence();
      if (A->gxprev != nullptr) adjoint(A->wp, u, gprev);
      else for (int j = 0; j < D; ++j) gprev[j] += u[j];
      sb_outer<D>(tu, tx, ones, lane, acc_a);
      lg_flush_column_sums<float>(acc_a, slot, (A->row_terms & 1) != 0);
      lg_wave_fence();
    }
    // ---- the draw: w reaches the proposal's parameters and x_{t-1};  grad s_q = g d / s_q + w . eps ----------------------
    {
      sb_cargs *A = SB_A();
      if (A->want_sq) {
        chain(A->wq, 2, A->offq, A->offq_sb, xp, u);
        float dot = 0.0f;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const float diff = xt[j] - u[j];
          dot = fma_t(w[j], diff, dot);
        }
        scale_acc[2] += g * ((float)D * inv_s_q) + dot * inv_s_q;
      }
      sb_lds_put(tu + lane * D, w);
      lg_wave_fence();
      if (A->gxprev != nullptr) adjoint(A->wq, w, gprev);
      else for (int j = 0; j < D; ++j) gprev[j] += w[j];
      sb_outer<D>(tu, tx, ones, lane, acc_q);
      const int row_terms = A->rows != nullptr ? A->row_terms : 0;
      lg_flush_column_sums<float>(acc_q, slot + 128, (row_terms & 4) != 0);
      lg_wave_fence();
#pragma unroll
      for (int j = 0; j < D; ++j) g_old[j] = gprev[j];
      old_tile = tile;
      if (row_terms != 0) {      // the four wavefronts' column sums meet: the tile's record of each wanted term
        // (LDS stores only: a release fence here would also wait for the next tile's block, which is on its way into LDS)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        lg_store_column_sums<float>(slot, A->rows + (uint64_t)tile * 3 * (kLgRowsMax * 16), row_terms);
      }
    }
  }
  sb_cargs *A = SB_A();
  if (old_tile != 0xffffffffu && A->gxprev != nullptr) sb_store_row(A->gxprev + ((uint64_t)old_tile * kLgBlock + tid) * D, g_old);
  if (bad) raise_flag(A->flags, AESMC_FLAG_INDEX_OUT_OF_RANGE);
  __syncthreads();      // the wavefronts' areas become the records' scratch
  float *scratch = slots + kSbSlots + kSbOnes;
  float *record = A->ws + (int64_t)blockIdx.x * 4 * kLgRecord;
  lg_outer_publish<float>(acc_a, scratch, record);
  lg_outer_publish<float>(acc_c, scratch, record + kLgRecord);
  lg_outer_publish<float>(acc_q, scratch, record + 2 * kLgRecord);
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    float v = scale_acc[m];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    if ((threadIdx.x & 63) == 0) scratch[(threadIdx.x >> 6) * 4 + m] = v;
  }
  __syncthreads();
  if (threadIdx.x < kLgRecord) {
    const int m = threadIdx.x;
    record[3 * kLgRecord + m] = m < 3 ? ((scratch[m] + scratch[4 + m]) + scratch[8 + m]) + scratch[12 + m] : 0.0f;
  }
  const float *carry = A->carry;
  if (carry != nullptr && threadIdx.x < kLgRecord) {
    // (each lane wrote the four elements it now reads: no barrier; the records carried are added in record order)
    float own[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) own[m] = record[m * kLgRecord + threadIdx.x];
    const int carry_records = A->carry_records;
    for (int r = blockIdx.x; r < carry_records; r += gridDim.x) {
      const float *c = carry + (int64_t)r * 4 * kLgRecord + threadIdx.x;
      float v[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) v[m] = c[m * kLgRecord];
#pragma unroll
      for (int m = 0; m < 4; ++m) own[m] += v[m];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) record[m * kLgRecord + threadIdx.x] = own[m];
  }
#undef SB_A
#endif
}

// 0: this form where it covers the call; 1: always the first form (tiles through LDS).  AESMC_K14_FORM=tiles in the
// environment, or the test hook below (not part of the C ABI of include/aesmc_hip.h).
static std::atomic<int> g_sb_form{[] { const char *v = measurement_knob("AESMC_K14_FORM"); return (v != nullptr && v[0] == 't') ? 1 : 0; }()};
static std::atomic<int> g_sb_grid{0};      // > 0: workgroups of either form's launch (the records' association follows the grid)
int affine_step_backward_forced_grid() { return g_sb_grid.load(std::memory_order_relaxed); }

// Does this form cover the call?  Rows of the same even number of float32 values (2 .. 14) on both sides, every tile of
// 256 particles inside one batch row, weights whose rows are contiguous (what an nn.Linear holds), 32-bit row numbers.
static std::atomic<int> g_sb_last{0};      // which form the last step's backward took: 1 tiles, 2 rows (test hook below)
static bool sb_covers(const aesmc_affine_map *mp, const aesmc_affine_map *mg, const aesmc_affine_map *mq, int64_t B, int64_t K);
bool affine_step_backward_rows_covers(const aesmc_affine_map *mp, const aesmc_affine_map *mg, const aesmc_affine_map *mq,
                                      int64_t B, int64_t K) {
  const bool covers = sb_covers(mp, mg, mq, B, K);
  g_sb_last.store(covers ? 2 : 1, std::memory_order_relaxed);
  return covers;
}
static bool sb_covers(const aesmc_affine_map *mp, const aesmc_affine_map *mg, const aesmc_affine_map *mq, int64_t B, int64_t K) {
  if (g_sb_form.load(std::memory_order_relaxed) == 1) return false;
  const auto rows_contiguous = [](const aesmc_affine_map *m) {
    return m->stride_in == 1 && m->stride_out == m->din && (reinterpret_cast<uintptr_t>(m->weight) & 3u) == 0 &&
           (m->offset == nullptr || (reinterpret_cast<uintptr_t>(m->offset) & 3u) == 0);
  };
  const int64_t d = mp->dout;
#ifdef AESMC_LG_FAST_BUILD
  if (d != 10) return false;
#endif
  return d >= 2 && d <= 14 && d % 2 == 0 && mp->din == d && mg->dout == d && mg->din == d && mq->dout == d &&
         mq->din == d && K % kLgBlock == 0 && B * K > 0 && B * K < (1ll << 31) && rows_contiguous(mp) &&
         rows_contiguous(mg) && rows_contiguous(mq);
}

template <int D> static unsigned sb_grid(int64_t tiles) {
  return (unsigned)std::min<int64_t>(lg_persistent_grid(tiles, Sb<D>::kLds, Sb<D>::kPerCu), kLgMaxGrid);
}
unsigned affine_step_backward_rows_grid(int64_t B, int64_t K, int64_t d) {
  const int64_t tiles = B * K / kLgBlock;
  switch (d) {
#ifndef AESMC_LG_FAST_BUILD
    case 2: return sb_grid<2>(tiles);
    case 4: return sb_grid<4>(tiles);
    case 6: return sb_grid<6>(tiles);
    case 8: return sb_grid<8>(tiles);
    case 12: return sb_grid<12>(tiles);
    case 14: return sb_grid<14>(tiles);
#endif
    default: return sb_grid<10>(tiles);
  }
}

template <int D, bool PAIRED>
static void sb_launch(bool gathers, bool folds, unsigned grid, hipStream_t stream, const SbArgs &a, bool &ok) {
  constexpr size_t lds = Sb<D>::kLds;
  static bool raised[4][64] = {};
  ok = true;
  if (gathers && folds) {
    if (lds > 64 * 1024) ok = lg_raise_lds_limit(reinterpret_cast<const void *>(&affine_step_backward_rows_kernel<D, true, true, PAIRED>), raised[0]);
    hipLaunchKernelGGL((affine_step_backward_rows_kernel<D, true, true, PAIRED>), ok ? dim3(grid) : dim3(0), dim3(kLgBlock), lds, stream, a);
  } else if (gathers) {
    if (lds > 64 * 1024) ok = lg_raise_lds_limit(reinterpret_cast<const void *>(&affine_step_backward_rows_kernel<D, true, false, PAIRED>), raised[1]);
    hipLaunchKernelGGL((affine_step_backward_rows_kernel<D, true, false, PAIRED>), ok ? dim3(grid) : dim3(0), dim3(kLgBlock), lds, stream, a);
  } else if (folds) {
    if (lds > 64 * 1024) ok = lg_raise_lds_limit(reinterpret_cast<const void *>(&affine_step_backward_rows_kernel<D, false, true, PAIRED>), raised[2]);
    hipLaunchKernelGGL((affine_step_backward_rows_kernel<D, false, true, PAIRED>), ok ? dim3(grid) : dim3(0), dim3(kLgBlock), lds, stream, a);
  } else {
    if (lds > 64 * 1024) ok = lg_raise_lds_limit(reinterpret_cast<const void *>(&affine_step_backward_rows_kernel<D, false, false, PAIRED>), raised[3]);
    hipLaunchKernelGGL((affine_step_backward_rows_kernel<D, false, false, PAIRED>), ok ? dim3(grid) : dim3(0), dim3(kLgBlock), lds, stream, a);
  }
}
template <int D>
static void sb_launch_either(bool paired, bool gathers, bool folds, unsigned grid, hipStream_t stream, const SbArgs &a, bool &ok) {
  if (paired) sb_launch<D, true>(gathers, folds, grid, stream, a, ok);
  else sb_launch<D, false>(gathers, folds, grid, stream, a, ok);
}

// does the rows form take the maps' interleaved weight pairs (AESMC_K14_PAIRS=0 in the environment: a measurement knob)?
bool affine_step_backward_rows_pairs() {
  static const bool no_pairs = [] { const char *v = measurement_knob("AESMC_K14_PAIRS"); return v != nullptr && v[0] == '0'; }();
  return kSbPacked && !no_pairs;
}

int launch_affine_step_backward_rows(const float *xprev, const float *x, const float *y, int64_t y_sb, const LgMap &mp,
                                     const LgMap &mg, const LgMap &mq, const float *sp, const float *sg, const float *sq,
                                     const float *lw, const float *lse, const float *grad_lse, const float *grad_lw,
                                     const LgBackwardOut &out, int64_t N, uint32_t K, unsigned grid, hipStream_t stream) {
  const bool gathers = out.gat.idx != nullptr, folds = out.child_grad != nullptr;
  SbArgs a = {};
  a.xprev = xprev; a.x = x; a.y = y; a.y_sb = y_sb;
  a.wp = static_cast<const float *>(mp.w); a.wg = static_cast<const float *>(mg.w); a.wq = static_cast<const float *>(mq.w);
  a.offp = static_cast<const float *>(mp.off); a.offg = static_cast<const float *>(mg.off); a.offq = static_cast<const float *>(mq.off);
  a.offp_sb = mp.off_sb; a.offg_sb = mg.off_sb; a.offq_sb = mq.off_sb;
  a.sp = sp; a.sg = sg; a.sq = sq;
  a.lw_src = grad_lse != nullptr ? lw : x; a.glw_src = grad_lw != nullptr ? grad_lw : x;      // (x: N readable values)
  a.lse = lse; a.grad_lse = grad_lse;
  a.gxprev = static_cast<float *>(out.gxprev); a.gx_in = static_cast<const float *>(out.gx_in);
  a.rows = static_cast<float *>(out.rows); a.ws = static_cast<float *>(out.ws);
  a.carry = static_cast<const float *>(out.carry);
  a.child_rows = static_cast<const float *>(out.child_grad); a.child_end = out.child_end;
  a.anc = out.gat.idx; a.flags = out.gat.flags;
  a.N = N; a.tiles = N / kLgBlock; a.K = K;
  a.tpr = K / kLgBlock; a.tpr_mul = (uint32_t)((1ull << 32) / a.tpr);      // (tpr == 1: 2^32 does not fit, the quotient's fixup covers it)
  if (a.tpr == 1) a.tpr_mul = 0xffffffffu;
  a.row_terms = out.rows != nullptr ? out.row_terms : 0; a.want_sq = out.want_scale_q;
  a.has_lse = grad_lse != nullptr ? 1 : 0; a.has_glw = grad_lw != nullptr ? 1 : 0;
  a.carry_records = out.carry != nullptr ? out.carry_records : 0;
  // the chains' weights as interleaved pairs, written into the workspace's tail by a small launch in front of this one (the
  // weights may have been stepped since the last call; inside a hipGraph capture the rebuild is captured with it)
  const bool paired = affine_step_backward_rows_pairs() && out.pairs != nullptr;
  if (paired) {
    if (!out.pairs_ready) {      // (a run of steps shares its weights: its first call builds them, the others are handed them)
      const int status = launch_affine_weight_pairs(mp, mg, mq, out.pairs, stream);
      if (status != AESMC_OK) return status;
    }
    a.pairs = out.pairs;
  }
  bool ok = true;
  switch (mp.dout) {
#ifndef AESMC_LG_FAST_BUILD
    case 2: sb_launch_either<2>(paired, gathers, folds, grid, stream, a, ok); break;
    case 4: sb_launch_either<4>(paired, gathers, folds, grid, stream, a, ok); break;
    case 6: sb_launch_either<6>(paired, gathers, folds, grid, stream, a, ok); break;
    case 8: sb_launch_either<8>(paired, gathers, folds, grid, stream, a, ok); break;
    case 12: sb_launch_either<12>(paired, gathers, folds, grid, stream, a, ok); break;
    case 14: sb_launch_either<14>(paired, gathers, folds, grid, stream, a, ok); break;
#endif
    case 10: sb_launch_either<10>(paired, gathers, folds, grid, stream, a, ok); break;
    default: return AESMC_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

}  // namespace aesmc

extern "C" int aesmc_test_last_step_backward_form(void) { return aesmc::g_sb_last.load(std::memory_order_relaxed); }

extern "C" int aesmc_test_set_step_backward(int form, int grid) {
  if ((form != 0 && form != 1) || grid < 0 || grid > aesmc::kLgMaxGrid) return AESMC_ERR_INVALID_ARGUMENT;
  aesmc::g_sb_form.store(form, std::memory_order_relaxed);
  aesmc::g_sb_grid.store(grid, std::memory_order_relaxed);
  return AESMC_OK;
}
