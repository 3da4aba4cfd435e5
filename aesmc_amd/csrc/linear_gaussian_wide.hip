// K17 / K18: the linear-Gaussian SMC step for WIDE latents (rows of 128 float32 values: BASELINE.json configs[4]) on the
// fp32 matrix cores — what aesmc/inference.py:102-126 costs per timestep when the three maps are 128 x 128: until now
// three library GEMMs, the broadcast adds of their offsets, the draw and a three-Normal log-weight kernel, 17 passes
// over a [B,K,128] tensor (9 GB at B=64, K=16384); here two launches that read x_{t-1} (through the ancestors), the
// noise and x_t once each and write x_t once.
//
//   K17  x_t = (c_q + Q x) + eps s_q,  sum_q, sum_p          (x = x_{t-1}[ancestor]; both maps of x in one pass)
//   K18  log w = (sum_p + sum_g) - sum_q,  sum_g from c_g + C x_t
//
// A location is ONE fma chain per output element, inputs ascending, started from the offset (the arithmetic contract,
// oracle/smc_core.c): v_mfma_f32_16x16x4_f32 accumulates its four k in order into an accumulator initialised with the
// offset, 32 k-steps in a row — bit for bit that chain (tools/probe_mfma.hip) — so x_t equals the C oracle's bits.
// The squared distances are summed per lane over its 32 outputs and then over the four lanes that share a particle:
// another association than the oracle's single chain, equal to it to rounding (tests: 2e-6 relative).
//
// Mapping.  A workgroup of 512 lanes per CU keeps the maps' weights in LDS (two maps: 135 KB) for the whole launch; a
// wavefront walks tiles of 32 consecutive particles (two matrix tiles of 16).  Matrix operands: A[m][k] = W[16 mt + m]
// [4 s + k] out of LDS — a row's inputs are stored permuted, input 4 s + g at g * 32 + s, so that the four k-steps of a
// group come in one 16-byte read, rows 132 floats apart: conflict-free — and B[k][n] = x[particle n][4 s + k] straight
// from HBM into the lane's registers (lane (g, n) holds inputs g, 4 + g, ... of particle n: 32 dword loads per matrix
// tile; the other wavefront on the SIMD multiplies meanwhile).  The accumulators D[row = output][col = particle] leave
// lane (g, n) with outputs 16 mt + 4 g + r of particle n: the element-wise part (draw, residuals) runs in that layout
// on 16-byte pieces, a particle's four lanes add their sums through two cross-lane steps.
#include "linear_gaussian.hpp"
#include "linear_gaussian_wide_generic.hpp"
#include "philox_normal.hpp"

namespace aesmc {

constexpr int kWd = 128;                      // the extent this file is built for
constexpr int kWdRow = kWd + 4;               // LDS row stride in floats
constexpr int kWdTile = 32;                   // particles per wavefront tile
constexpr int kWdThreads = 512;
typedef float wd_f4 __attribute__((ext_vector_type(4)));

struct WideArgs {
  const float *x_in;        // K17: x_{t-1} (un-resampled when `anc` is there); K18: x_t
  const int64_t *anc;       // K17: ancestors [B,K] or nullptr
  const float *eps;         // K17: the draw's noise [B,K,128]
  const float *y;           // K18: observation rows
  int64_t y_sb;
  const float *w[2];        // K17: {Q, A}; K18: {C}: [128,128] row-major
  const float *off[2];      // nullptr, [128] (sb = 0) or [B,128]
  int64_t off_sb[2];
  const float *s_p, *s_g, *s_q;
  float *out_x;             // K17
  float *sums;              // [N,2]: (sum_p, sum_q) — K17 writes, K18 reads
  float *out_lw;            // K18
  int32_t *flags;
  int64_t N;
  uint32_t K;
  // K17 with the noise formed in the launch (eps == nullptr): torch's Philox stream (philox_normal.hpp) and Q = G / 128,
  // the particles one trip's four outputs lie apart
  PhiloxStream ps;
  uint32_t Q;
};

// the maps' weights into LDS, permuted per row (see the header)
template <int NMAPS>
__device__ __forceinline__ void wide_stage_weights(const WideArgs &a, float *wl) {
  for (int m = 0; m < NMAPS; ++m) {
    const float *w = a.w[m];
    for (uint32_t v = threadIdx.x; v < (uint32_t)(kWd * kWd / 4); v += kWdThreads) {
      const uint32_t j = v / (kWd / 4), s = v % (kWd / 4);      // inputs 4 s .. 4 s + 3 of row j
      const wd_f4 q = *reinterpret_cast<const wd_f4 *>(w + (size_t)j * kWd + 4 * s);
      float *row = wl + (size_t)m * kWd * kWdRow + j * kWdRow;
#pragma unroll
      for (int g = 0; g < 4; ++g) row[g * 32 + s] = q[g];
    }
  }
}

// acc[map][mt][nt] += W_map x over all 32 k-steps; bx[nt][s] = x[particle n of tile nt][4 s + g].
// ROLLING PREFETCH: the four inputs a k-group consumed are dead once its last product has issued, so the NEXT tile's
// values for them (next[nt]: that tile's row of particle n, at input g) are sent for right there — eight dword loads per
// k-group — and have the rest of this tile's products and its element-wise part to arrive (rocprofv3 before: the matrix
// pipe busy 63 % of the launch, every wavefront waiting 18 600 cycles per tile for its rows in front of the first product).
template <int NMAPS>
__device__ __forceinline__ void wide_products(const float *wl, uint32_t lane, float (&bx)[2][32], wd_f4 (&acc)[NMAPS][8][2],
                                              const float *next0, const float *next1) {
  const uint32_t m = lane & 15u, g = lane >> 4;
  const float *wa = wl + m * kWdRow + g * 32;
  // One 16-byte read feeds eight multiply-accumulates (four k-steps x two particle tiles); the NEXT read is sent for
  // before them and nothing else may move across the group's end — left to itself the scheduler hoists all 128 reads
  // to the top and spills five hundred registers.
  constexpr int GROUPS = 8 * NMAPS * 8;
  auto operand = [&](int idx) {
    const int sg = idx / (NMAPS * 8), map = (idx / 8) % NMAPS, mt = idx % 8;
    return *reinterpret_cast<const wd_f4 *>(wa + (size_t)map * kWd * kWdRow + mt * 16 * kWdRow + 4 * sg);
  };
  wd_f4 cur = operand(0);
#pragma unroll
  for (int idx = 0; idx < GROUPS; ++idx) {
    const int sg = idx / (NMAPS * 8), map = (idx / 8) % NMAPS, mt = idx % 8;
    wd_f4 nxt = cur;
    if (idx + 1 < GROUPS) nxt = operand(idx + 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[map][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[t], bx[nt][4 * sg + t], acc[map][mt][nt], 0, 0, 0);
    }
    if ((idx + 1) % (NMAPS * 8) == 0) {      // this k-group's inputs are spent: the next tile's take their registers
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bx[0][4 * sg + t] = next0[4 * (4 * sg + t)];
        bx[1][4 * sg + t] = next1[4 * (4 * sg + t)];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
}

template <int NMAPS>
__device__ __forceinline__ void wide_offsets(const WideArgs &a, uint32_t b, uint32_t g, wd_f4 (&acc)[NMAPS][8][2]) {
#pragma unroll
  for (int map = 0; map < NMAPS; ++map) {
    const float *off = a.off[map];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      wd_f4 o = {0.0f, 0.0f, 0.0f, 0.0f};
      if (off != nullptr) o = *reinterpret_cast<const wd_f4 *>(off + (int64_t)b * a.off_sb[map] + 16 * mt + 4 * g);
      acc[map][mt][0] = o;
      acc[map][mt][1] = o;
    }
  }
}

// sum over the four lanes (g = 0 .. 3) that share particle n, in the order ((g0 + g1) + g2) + g3, returned to all
__device__ __forceinline__ float wide_particle_sum(float v, uint32_t lane) {
  const uint32_t n = lane & 15u;
  const float v0 = __shfl(v, (int)n, kWave), v1 = __shfl(v, (int)(n + 16u), kWave), v2 = __shfl(v, (int)(n + 32u), kWave),
              v3 = __shfl(v, (int)(n + 48u), kWave);
  return ((v0 + v1) + v2) + v3;
}

// K17.  DRAWN: the noise is element e of what `torch.empty([B,K,128]).normal_()` would hold, formed here (as K16 forms
// its own).  ATen gives thread t the elements t + G (4 c + i), i = 0 .. 3, of its c-th Philox call; G is a multiple of
// 128, so those are ONE output j = t % 128 of four particles Q = G / 128 apart.  A tile therefore takes its 32
// particles as 8 consecutive ones from each of the four quarters of a trip (matrix tile nt, column n: quarter
// 2 nt + (n >> 3), particle (n & 7) of the eight) — columns are independent, any 16 particles make a matrix tile — and a
// call's four normals all land in this wavefront: the two lanes n and n ^ 8 that hold the same outputs of the same
// eight particles share the calls (two of a group's four outputs each) and swap what the other needs.
template <bool GATHER, bool DRAWN>
__global__ __launch_bounds__(kWdThreads, 2) void affine_wide_draw_kernel(WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wd_smem[];
  float *wl = reinterpret_cast<float *>(wd_smem);
  wide_stage_weights<2>(a, wl);
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t n = lane & 15u, g = lane >> 4;
  const float s_q = a.s_q[0], s_p = a.s_p[0];
  const float two_var_p = 2.0f * (s_p * s_p), two_var_q = 2.0f * (s_q * s_q);
  const float const_p = (float)kWd * (Num<float>::log(s_p) + LgConst<float>::half_log_2pi());
  const float const_q = (float)kWd * (Num<float>::log(s_q) + LgConst<float>::half_log_2pi());
  const int64_t tiles = a.N / kWdTile;
  const uint32_t K = a.K;
  uint32_t bad = 0;
  PhiloxStream ps = a.ps;
  if constexpr (DRAWN) ps = philox_resolve(a.ps);
  const uint32_t h = n >> 3, Q = a.Q, per_trip = Q / 8;
  // A tile's rows of x_{t-1} are sent for a tile AHEAD — its ancestor indices before the tile in hand multiplies, its rows
  // k-group by k-group inside the multiplications as the operand registers fall free (wide_products) — so both memory
  // round trips pass under this wavefront's own work instead of in front of the next tile's first product.
  const int64_t tile_stride = (int64_t)gridDim.x * (kWdThreads / 64);
  auto particles_of = [&](int64_t tile, int64_t (&part)[2], uint32_t &trip, uint32_t &pb) {
    trip = DRAWN ? (uint32_t)(tile / per_trip) : 0u;
    pb = DRAWN ? ((uint32_t)(tile - (int64_t)trip * per_trip)) * 8u + (n & 7u) : 0u;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      part[nt] = DRAWN ? (int64_t)4 * Q * trip + pb + (int64_t)Q * (2 * nt + h) : tile * kWdTile + 16 * nt + n;
  };
  auto batch_row_of = [&](int64_t tile, uint32_t trip) { return (uint32_t)((DRAWN ? (int64_t)4 * Q * trip : tile * kWdTile) / K); };
  auto rows_of = [&](const int64_t (&part)[2], const int64_t (&anc_in)[2], uint32_t b, const float *(&src)[2]) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      int64_t row = part[nt];
      if constexpr (GATHER) {
        int64_t anc = anc_in[nt];
        if (anc < 0 || anc >= (int64_t)K) {      // K2 writes K for a degenerate row (flagged there); never fault on it
          bad = 1;
          anc = anc < 0 ? 0 : (int64_t)K - 1;
        }
        row = (int64_t)b * K + anc;
      }
      src[nt] = a.x_in + row * kWd + g;
    }
  };
  float bx[2][32];
  int64_t tile = (int64_t)blockIdx.x * (kWdThreads / 64) + wave;
  if (tile < tiles) {
    int64_t part0[2], anc0[2] = {0, 0};
    uint32_t trip0, pb0;
    particles_of(tile, part0, trip0, pb0);
    if constexpr (GATHER) {
      anc0[0] = a.anc[part0[0]];
      anc0[1] = a.anc[part0[1]];
    }
    const float *src0[2];
    rows_of(part0, anc0, batch_row_of(tile, trip0), src0);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int s = 0; s < 32; ++s) bx[nt][s] = src0[nt][4 * s];
  }
  for (; tile < tiles; tile += tile_stride) {
    // the particle of column n of matrix tile nt; a tile lies inside one batch row (the host: K a multiple of 32, and
    // of 4 Q when the launch draws)
    uint32_t trip, pb;
    int64_t part[2];
    particles_of(tile, part, trip, pb);
    const uint32_t b = batch_row_of(tile, trip);
    // the next tile's particles and their ancestors (no branch around loads — the compiler would wait for every load in
    // flight where the branch ends —: behind the last tile the last tile's are fetched again and dropped)
    const int64_t tile_next = tile + tile_stride < tiles ? tile + tile_stride : tile;
    uint32_t trip_next, pb_next;
    int64_t part_next[2], anc_next[2] = {0, 0};
    particles_of(tile_next, part_next, trip_next, pb_next);
    if constexpr (GATHER) {
      anc_next[0] = a.anc[part_next[0]];
      anc_next[1] = a.anc[part_next[1]];
    }
    const float *src_next[2];
    rows_of(part_next, anc_next, batch_row_of(tile_next, trip_next), src_next);
    wd_f4 acc[2][8][2];
    wide_offsets<2>(a, b, g, acc);
    wide_products<2>(wl, lane, bx, acc, src_next[0], src_next[1]);
    // ---- the draw and the two squared distances, outputs 16 mt + 4 g + r of the lane's two particles ------------
    float q_sum[2] = {0.0f, 0.0f}, p_sum[2] = {0.0f, 0.0f};
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      wd_f4 e[2];
      if constexpr (DRAWN) {
        // outputs r = 2 h, 2 h + 1 are this lane's calls, the other two its partner's (lane ^ 8); of a call's four
        // normals quarter h goes to this lane's particle of matrix tile 0, quarter 2 + h to tile 1, the rest across
        const uint32_t t0 = pb * kWd + 16u * mt + 4u * g + 2u * h;
        const float4 c0 = philox_normal4(ps, t0, trip), c1 = philox_normal4(ps, t0 + 1u, trip);
        const float keep[2][2] = {{h ? c0.y : c0.x, h ? c0.w : c0.z}, {h ? c1.y : c1.x, h ? c1.w : c1.z}};      // [call][nt]
        const float send[2][2] = {{h ? c0.x : c0.y, h ? c0.z : c0.w}, {h ? c1.x : c1.y, h ? c1.z : c1.w}};
        float got[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) got[u][nt] = __shfl_xor(send[u][nt], 8, kWave);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          e[nt][0] = h ? got[0][nt] : keep[0][nt];
          e[nt][1] = h ? got[1][nt] : keep[1][nt];
          e[nt][2] = h ? keep[0][nt] : got[0][nt];
          e[nt][3] = h ? keep[1][nt] : got[1][nt];
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) e[nt] = *reinterpret_cast<const wd_f4 *>(a.eps + part[nt] * kWd + 16 * mt + 4 * g);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        wd_f4 x;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float noise = e[nt][r] * s_q;
          x[r] = acc[0][mt][nt][r] + noise;
          const float dq = x[r] - acc[0][mt][nt][r], dp = x[r] - acc[1][mt][nt][r];
          q_sum[nt] = fma_t(dq, dq, q_sum[nt]);
          p_sum[nt] = fma_t(dp, dp, p_sum[nt]);
        }
        *reinterpret_cast<wd_f4 *>(a.out_x + part[nt] * kWd + 16 * mt + 4 * g) = x;
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float qs = wide_particle_sum(q_sum[nt], lane), ps_ = wide_particle_sum(p_sum[nt], lane);
      if (g == 0) {
        float2 out;
        out.x = (-ps_) / two_var_p - const_p;
        out.y = (-qs) / two_var_q - const_q;
        *reinterpret_cast<float2 *>(a.sums + 2 * part[nt]) = out;
      }
    }
  }
  if (bad != 0u) raise_flag(a.flags, AESMC_FLAG_INDEX_OUT_OF_RANGE);
}

// K18
__global__ __launch_bounds__(kWdThreads, 2) void affine_wide_emission_kernel(WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wd_smem[];
  float *wl = reinterpret_cast<float *>(wd_smem);
  wide_stage_weights<1>(a, wl);
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t n = lane & 15u, g = lane >> 4;
  const float s_g = a.s_g[0];
  const float two_var_g = 2.0f * (s_g * s_g);
  const float const_g = (float)kWd * (Num<float>::log(s_g) + LgConst<float>::half_log_2pi());
  const int64_t tiles = a.N / kWdTile;
  const uint32_t K = a.K;
  // (the next tile's rows of x_t are sent for k-group by k-group inside this tile's multiplications, as in K17)
  const int64_t tile_stride = (int64_t)gridDim.x * (kWdThreads / 64);
  auto rows_load = [&](int64_t tile, float (&bx)[2][32]) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float *src = a.x_in + (tile * kWdTile + 16 * nt + n) * kWd + g;
#pragma unroll
      for (int s = 0; s < 32; ++s) bx[nt][s] = src[4 * s];
    }
  };
  float bx[2][32];
  int64_t tile = (int64_t)blockIdx.x * (kWdThreads / 64) + wave;
  if (tile < tiles) rows_load(tile, bx);
  for (; tile < tiles; tile += tile_stride) {
    const int64_t n0 = tile * kWdTile;
    const uint32_t b = (uint32_t)(n0 / K);
    const int64_t tile_next = tile + tile_stride < tiles ? tile + tile_stride : tile;      // (behind the last tile: fetched again, dropped)
    const float *next0 = a.x_in + (tile_next * kWdTile + n) * kWd + g, *next1 = next0 + 16 * kWd;
    wd_f4 acc[1][8][2];
    wide_offsets<1>(a, b, g, acc);
    wide_products<1>(wl, lane, bx, acc, next0, next1);
    wd_f4 yv[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) yv[mt] = *reinterpret_cast<const wd_f4 *>(a.y + (int64_t)b * a.y_sb + 16 * mt + 4 * g);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int64_t p = n0 + 16 * nt + n;
      float g_sum = 0.0f;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = yv[mt][r] - acc[0][mt][nt][r];
          g_sum = fma_t(d, d, g_sum);
        }
      }
      g_sum = wide_particle_sum(g_sum, lane);
      if (g == 0) {
        const float2 pq = *reinterpret_cast<const float2 *>(a.sums + 2 * p);
        const float sum_g = (-g_sum) / two_var_g - const_g;
        a.out_lw[p] = (pq.x + sum_g) - pq.y;
      }
    }
  }
}

}  // namespace aesmc

using namespace aesmc;

extern "C" int64_t aesmc_affine_wide_dim(void) { return kWd; }

extern "C" size_t aesmc_affine_wide_workspace_bytes(int64_t B, int64_t K) {
  return (B > 0 && K > 0) ? (size_t)B * (size_t)K * 2 * sizeof(float) : 0;
}

extern "C" int64_t aesmc_affine_wide_min_dim(void) { return kWgMinDim; }
extern "C" int64_t aesmc_affine_wide_max_dim(void) { return kWgMaxDim; }

// floats per particle between the launches: two raw sums per chunk of the draw's output rows, one per chunk of the emission's
static inline uint32_t wideg_record_floats(int64_t dx, int64_t dy) {
  const int dxp = wideg_padded(dx);
  return 2u * wideg_chunks(dx, wideg_draw_chunk(dxp)) + wideg_chunks(dy, wideg_emit_chunk(dy));
}

extern "C" size_t aesmc_affine_wide_workspace_bytes_for(int64_t B, int64_t K, int64_t dx, int64_t dy) {
  if (B <= 0 || K <= 0 || dx <= 0 || dy <= 0 || wideg_padded(dx) == 0 || wideg_padded(dy) == 0) return 0;
  const size_t floats = std::max<size_t>(2, wideg_record_floats(dx, dy));
  return (size_t)B * (size_t)K * floats * sizeof(float);
}

// K17g + K18g (linear_gaussian_wide_generic.hpp): every shape the 128-wide kernels above do not take
static int wideg_propagate(const void *x_src, const int64_t *ancestors, const void *eps, const void *y, int64_t y_stride_b,
                           const aesmc_affine_map *transition, const aesmc_affine_map *emission,
                           const aesmc_affine_map *proposal, const void *scale_p, const void *scale_g, const void *scale_q,
                           void *out_x, void *out_lw, void *ws, size_t ws_bytes, int32_t *flags, int64_t B, int64_t K,
                           hipStream_t s) {
  const int64_t dx = proposal->dout, dy = emission->dout;
  if (proposal->din != dx || transition->dout != dx || transition->din != dx || emission->din != dx) return AESMC_ERR_UNSUPPORTED;
  if (dx < kWgMinDim || dx > kWgMaxDim || dy < 1 || dy > kWgMaxDim) return AESMC_ERR_UNSUPPORTED;
  if (eps == nullptr) return AESMC_ERR_UNSUPPORTED;      // the noise in the launch: the 128-wide kernels only (the caller fills it)
  const aesmc_affine_map *maps[3] = {transition, emission, proposal};
  for (const aesmc_affine_map *m : maps) {      // weights as an nn.Linear holds them (dense rows)
    if (m->stride_in != 1 || m->stride_out != m->din || (((uintptr_t)m->weight) & 3u) != 0 ||
        (((uintptr_t)m->offset) & 3u) != 0)
      return AESMC_ERR_UNSUPPORTED;
  }
  // pieces of four move as 16-byte accesses where rows, strides and bases allow it; element by element elsewhere
  auto offset_vec = [](const aesmc_affine_map *m) {
    return m->offset == nullptr || (aligned16(m->offset) && (m->offset_stride_b % 4) == 0);
  };
  const bool vec_x = (dx % 4) == 0;
  const uint32_t vec_in = (vec_x && aligned16(transition->weight) && aligned16(proposal->weight) &&
                           aligned16(emission->weight)) ? 1u : 0u;
  const uint32_t vec_draw = (vec_x && offset_vec(transition) && offset_vec(proposal)) ? 1u : 0u;
  const uint32_t vec_emit = ((dy % 4) == 0 && offset_vec(emission) && aligned16(y) && (y_stride_b % 4) == 0) ? 1u : 0u;
  const int64_t N = B * K;
  if (B >= (1ll << 31) || K >= (1ll << 31) || N >= (1ll << 31) / 2) return AESMC_ERR_UNSUPPORTED;
  if (ws_bytes < aesmc_affine_wide_workspace_bytes_for(B, K, dx, dy)) return AESMC_ERR_WORKSPACE;
  if (N == 0) return AESMC_OK;
  const int dxp = wideg_padded(dx);
  WideGArgs a = {};
  a.x_in = static_cast<const float *>(x_src); a.anc = ancestors; a.eps = static_cast<const float *>(eps);
  a.y = static_cast<const float *>(y); a.y_sb = y_stride_b;
  a.w[0] = static_cast<const float *>(proposal->weight); a.w[1] = static_cast<const float *>(transition->weight);
  a.off[0] = static_cast<const float *>(proposal->offset); a.off[1] = static_cast<const float *>(transition->offset);
  a.off_sb[0] = proposal->offset_stride_b; a.off_sb[1] = transition->offset_stride_b;
  a.s_p = static_cast<const float *>(scale_p); a.s_g = static_cast<const float *>(scale_g);
  a.s_q = static_cast<const float *>(scale_q);
  a.out_x = static_cast<float *>(out_x); a.sums = static_cast<float *>(ws); a.out_lw = static_cast<float *>(out_lw);
  a.flags = flags; a.B = (uint32_t)B; a.K = (uint32_t)K; a.tiles_per_row = (uint32_t)((K + kWgTile - 1) / kWgTile);
  a.din = (uint32_t)dx; a.dout = (uint32_t)dx; a.dx = (uint32_t)dx;
  a.chunks_draw = wideg_chunks(dx, wideg_draw_chunk(dxp));
  a.chunks_emit = wideg_chunks(dy, wideg_emit_chunk(dy));
  a.sums_stride = std::max<uint32_t>(2u, wideg_record_floats(dx, dy));
  a.vec_in = vec_in; a.vec_out = vec_draw;
  int status = wideg_launch_draw(a, dxp, ancestors != nullptr, s);
  if (status != AESMC_OK) return status;
  WideGArgs e = a;
  e.x_in = static_cast<const float *>(out_x); e.anc = nullptr; e.eps = nullptr;
  e.w[0] = static_cast<const float *>(emission->weight); e.w[1] = nullptr;
  e.off[0] = static_cast<const float *>(emission->offset); e.off[1] = nullptr;
  e.off_sb[0] = emission->offset_stride_b; e.off_sb[1] = 0;
  e.dout = (uint32_t)dy; e.vec_out = vec_emit;
  return wideg_launch_emission(e, dxp, s);
}

// K17 + K18: see include/aesmc_hip.h
extern "C" int aesmc_affine_normal_propagate_wide(
    const void *x_src, const int64_t *ancestors, const void *eps, const void *y, int64_t y_stride_b,
    const aesmc_affine_map *transition, const aesmc_affine_map *emission, const aesmc_affine_map *proposal,
    const void *scale_p, const void *scale_g, const void *scale_q, void *out_x, void *out_lw, void *ws, size_t ws_bytes,
    int32_t *flags, int64_t B, int64_t K, uint64_t seed, uint64_t offset, int64_t threads, const uint64_t *rng_state,
    void *stream) {
  if (x_src == nullptr || y == nullptr || transition == nullptr || emission == nullptr ||
      proposal == nullptr || scale_p == nullptr || scale_g == nullptr || scale_q == nullptr || out_x == nullptr ||
      out_lw == nullptr || ws == nullptr || B < 0 || K < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  const void *aligned[] = {x_src, eps, out_x, ws};
  for (const void *ptr : aligned)
    if (ptr != nullptr && !aligned16(ptr)) return AESMC_ERR_INVALID_ARGUMENT;
  if (eps == nullptr && (threads <= 0 || (threads % 256) != 0 || threads > 0x7fffffffLL || (offset & 3u) != 0 ||
                         (((uintptr_t)rng_state) & 7u) != 0))
    return AESMC_ERR_INVALID_ARGUMENT;
  if (out_x == x_src || (eps != nullptr && out_x == eps) || (((uintptr_t)ancestors) & 7u) != 0 || (((uintptr_t)y) & 3u) != 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  const aesmc_affine_map *maps[3] = {transition, emission, proposal};
  bool exact = true;      // rows of exactly 128 values on both sides and whole tiles: the kernels of this file
  for (const aesmc_affine_map *m : maps) {
    if (m->weight == nullptr) return AESMC_ERR_INVALID_ARGUMENT;
    exact = exact && m->dout == kWd && m->din == kWd;
  }
  exact = exact && (((uintptr_t)y) & 15u) == 0 && (y_stride_b % 4) == 0;
  if (!exact || K % kWdTile != 0)      // any other width (17 .. 256, dx != dy allowed), any K, any row alignment
    return wideg_propagate(x_src, ancestors, eps, y, y_stride_b, transition, emission, proposal, scale_p, scale_g, scale_q,
                           out_x, out_lw, ws, ws_bytes, flags, B, K, static_cast<hipStream_t>(stream));
  for (const aesmc_affine_map *m : maps) {
    // weights as an nn.Linear holds them, 16-byte aligned operands
    if (m->stride_in != 1 || m->stride_out != kWd || !aligned16(m->weight) ||
        (m->offset != nullptr && (!aligned16(m->offset) || (m->offset_stride_b % 4) != 0)))
      return AESMC_ERR_UNSUPPORTED;
  }
  const int64_t N = B * K;
  if (N >= (1ll << 31) / 2) return AESMC_ERR_UNSUPPORTED;
  // the noise formed in the launch: a trip's four quarters (Q = G / 128 particles each) inside one batch row
  const int64_t Q = eps == nullptr ? threads / kWd : 0;
  if (eps == nullptr && (Q % 8 != 0 || K % (4 * Q) != 0)) return AESMC_ERR_UNSUPPORTED;
  if (ws_bytes < aesmc_affine_wide_workspace_bytes(B, K)) return AESMC_ERR_WORKSPACE;
  if (N == 0) return AESMC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  WideArgs a = {};
  a.x_in = static_cast<const float *>(x_src); a.anc = ancestors; a.eps = static_cast<const float *>(eps);
  a.y = static_cast<const float *>(y); a.y_sb = y_stride_b;
  a.w[0] = static_cast<const float *>(proposal->weight); a.w[1] = static_cast<const float *>(transition->weight);
  a.off[0] = static_cast<const float *>(proposal->offset); a.off[1] = static_cast<const float *>(transition->offset);
  a.off_sb[0] = proposal->offset_stride_b; a.off_sb[1] = transition->offset_stride_b;
  a.s_p = static_cast<const float *>(scale_p); a.s_g = static_cast<const float *>(scale_g);
  a.s_q = static_cast<const float *>(scale_q);
  a.out_x = static_cast<float *>(out_x); a.sums = static_cast<float *>(ws); a.out_lw = static_cast<float *>(out_lw);
  a.flags = flags; a.N = N; a.K = (uint32_t)K;
  a.ps = philox_stream(seed, offset, eps == nullptr ? threads : 256, rng_state); a.Q = (uint32_t)Q;
  const int64_t tiles = N / kWdTile;
  const unsigned grid = (unsigned)std::min<int64_t>(lg_cu_count(), (tiles + kWdThreads / 64 - 1) / (kWdThreads / 64));
  const size_t lds2 = sizeof(float) * 2 * (size_t)kWd * kWdRow, lds1 = sizeof(float) * (size_t)kWd * kWdRow;
  static bool raised[5][64] = {};
#define WIDE_DRAW(G, D, slot)                                                                                          \
  do {                                                                                                                 \
    if (!lg_raise_lds_limit(reinterpret_cast<const void *>(&affine_wide_draw_kernel<G, D>), raised[slot]))           \
      return AESMC_ERR_LAUNCH;                                                                                         \
    hipLaunchKernelGGL((affine_wide_draw_kernel<G, D>), dim3(grid), dim3(kWdThreads), lds2, s, a);                     \
  } while (0)
  if (ancestors != nullptr && eps == nullptr) WIDE_DRAW(true, true, 0);
  else if (ancestors != nullptr) WIDE_DRAW(true, false, 1);
  else if (eps == nullptr) WIDE_DRAW(false, true, 2);
  else WIDE_DRAW(false, false, 3);
#undef WIDE_DRAW
  if (hipGetLastError() != hipSuccess) return AESMC_ERR_LAUNCH;
  WideArgs e = a;
  e.x_in = static_cast<const float *>(out_x); e.anc = nullptr;
  e.w[0] = static_cast<const float *>(emission->weight); e.off[0] = static_cast<const float *>(emission->offset);
  e.off_sb[0] = emission->offset_stride_b;
  if (!lg_raise_lds_limit(reinterpret_cast<const void *>(&affine_wide_emission_kernel), raised[4])) return AESMC_ERR_LAUNCH;
  hipLaunchKernelGGL(affine_wide_emission_kernel, dim3(grid), dim3(kWdThreads), lds1, s, e);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}
