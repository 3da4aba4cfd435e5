// K17g / K18g: the matrix-core linear-Gaussian step of linear_gaussian_wide.hip (rows of exactly 128 values) for ANY row
// width from 17 to 256 (observations: 1 to 256), latent and observation widths free of each other (dx != dy), any K.
// aesmc/state.py:61-183 is dimension-agnostic: before this file every width between the item kernels' 16 and the wide
// kernels' 128 — and everything above 128 — took three library GEMMs, their offsets' broadcast adds, the draw and a
// three-Normal log-weight kernel per timestep (17 passes over [B,K,d] tensors).
//
//   K17g  x_t = (c_q + Q x) + eps s_q,  raw sum_j (x_t - loc_q)_j^2 and sum_j (x_t - loc_p)_j^2     (x = x_{t-1}[ancestor])
//   K18g  raw sum_j (y - (c_g + C x_t))_j^2;  log w = (lp + lg) - lq   (at once, or by the combine launch below)
//
// The arithmetic contract is the 128-wide kernels': a location is ONE fma chain per output element, inputs ascending,
// started from the offset (oracle/smc_core.c) — v_mfma_f32_16x16x4_f32 accumulates its four k in order — so x_t equals the
// C oracle's bits; the squared distances are summed per lane, over a particle's four lanes, and (wide rows) over the
// output chunks in ascending order: equal to the oracle's single chain to rounding.
//
// What is generic here:
//   * the padded extent DXP (32 / 48 / 64 / 96 / 128 / 192 / 256: the smallest that holds the row) is a template
//     parameter; the row's real length `din`, the maps' real output count `dout` are launch arguments.  Inputs beyond
//     `din` are staged as zeros — fma(0, 0, acc) leaves acc as it is — outputs beyond `dout` are neither stored nor summed,
//     and matrix tiles that lie wholly in the padding are skipped (a wavefront-uniform test);
//   * rows wider than 128: both maps' weights no longer fit one CU's LDS (2 x 256 x 260 floats = 532 KB).  The OUTPUT
//     rows are cut into chunks of MC = 64 along the grid's y (three at 192, four at 256): a workgroup keeps its chunk's rows of both
//     maps resident, walks the tiles as before and leaves the chunk's partial sums per particle; x_{t-1} is read once per
//     chunk (L2 / MALL absorb most of it: the chunks of a tile run side by side);
//   * K not a multiple of 32: a batch row's last tile is masked — its missing particles load the row's last particle
//     again and store nothing;
//   * widths that are not multiples of 4: a row is then not a whole number of 16-byte pieces, so the pieces of four (the
//     noise in, x_t out, offsets, the observation, the weights' staging) are moved element by element with an element's own
//     bounds test (`vec_in` / `vec_out`: wavefront-uniform switches; the matrix operands were dword loads all along).
#pragma once
#include "linear_gaussian.hpp"

namespace aesmc {

typedef float wg_f4 __attribute__((ext_vector_type(4)));
constexpr int kWgTile = 32;       // particles per wavefront tile (two matrix tiles of 16)
constexpr int kWgThreads = 512;

struct WideGArgs {
  const float *x_in;        // K17g: x_{t-1} (un-resampled when `anc` is there); K18g: x_t
  const int64_t *anc;       // K17g: ancestors [B,K] or nullptr
  const float *eps;         // K17g: the draw's noise [B,K,dx]
  const float *y;           // K18g: observation rows [B,dy]
  int64_t y_sb;
  const float *w[2];        // K17g: {Q, A}; K18g: {C}: [dout,din] row-major
  const float *off[2];      // nullptr, [dout] (sb = 0) or [B,dout]
  int64_t off_sb[2];
  const float *s_p, *s_g, *s_q;
  float *out_x;             // K17g
  float *sums;              // [N, sums_stride] raw squared distances: p and q per K17g chunk, then g per K18g chunk
  float *out_lw;            // K18g (finish) / the combine launch
  int32_t *flags;
  uint32_t B, K, tiles_per_row;
  uint32_t din, dout;       // the maps' real extents (multiples of 4)
  uint32_t sums_stride;     // floats per particle in `sums`
  uint32_t chunks_draw;     // K17g's chunks (their 2 sums each lead a particle's record)
  uint32_t chunks_emit;     // K18g's chunks
  uint32_t dx;              // latent width (the constant of the two latent densities)
  uint32_t vec_in;          // 1: din is a multiple of 4 and the weights are 16-byte aligned (rows staged in 16-byte pieces)
  uint32_t vec_out;         // 1: dout, the offsets' / the observation's row strides are multiples of 4: pieces of four as such
};

// four consecutive values of a row of `limit` values starting at column `col`: one 16-byte piece, or element by element
// (columns beyond the row read as zero)
__device__ __forceinline__ wg_f4 wideg_load4(const float *row, uint32_t col, uint32_t limit, bool vec) {
  wg_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
  if (vec) {
    if (col < limit) v = *reinterpret_cast<const wg_f4 *>(row + col);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (col + r < limit) v[r] = row[col + r];
  }
  return v;
}
__device__ __forceinline__ void wideg_store4(float *row, uint32_t col, uint32_t limit, bool vec, const wg_f4 &v) {
  if (vec) {
    if (col < limit) *reinterpret_cast<wg_f4 *>(row + col) = v;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (col + r < limit) row[col + r] = v[r];
  }
}

// chunk `chunk`'s MC rows of NMAPS maps into LDS, zero-padded to DXP inputs; a row's inputs permuted — input 4 s + g at
// g * (DXP / 4) + s — so that the four k-steps of a group come in one 16-byte read (linear_gaussian_wide.hip)
template <int DXP, int MC, int NMAPS>
__device__ __forceinline__ void wideg_stage_weights(const WideGArgs &a, float *wl, uint32_t chunk) {
  constexpr int Row = DXP + 4, G4 = DXP / 4;
  for (int m = 0; m < NMAPS; ++m) {
    const float *w = a.w[m];
    for (uint32_t v = threadIdx.x; v < (uint32_t)(MC * G4); v += kWgThreads) {
      const uint32_t j = v / G4, s = v % G4, jr = chunk * MC + j;
      wg_f4 q = {0.0f, 0.0f, 0.0f, 0.0f};
      if (jr < a.dout) q = wideg_load4(w + (size_t)jr * a.din, 4u * s, a.din, a.vec_in != 0u);
      float *row = wl + (size_t)m * MC * Row + j * Row;
#pragma unroll
      for (int g = 0; g < 4; ++g) row[g * G4 + s] = q[g];
    }
  }
}

// acc[map][mt][nt] += W_map x over all k-steps; bx[nt][s] = x[particle n of tile nt][4 s + g].  The rolling prefetch of
// the 128-wide kernel: a k-group's four inputs are dead once its last product has issued, and the NEXT tile's values for
// them are sent for right there (when PREFETCH; a chunked launch's passes over the same tile keep bx).
// `mtiles`: matrix tiles of this chunk that hold real outputs; `kgroups`: k-groups that hold real inputs (both uniform).
template <int DXP, int MC, int NMAPS>
__device__ __forceinline__ void wideg_products(const float *wl, uint32_t lane, float (&bx)[2][DXP / 4],
                                               wg_f4 (&acc)[NMAPS][MC / 16][2], const float *next0, const float *next1,
                                               uint32_t mtiles, uint32_t kgroups, uint32_t din) {
  constexpr int Row = DXP + 4, G4 = DXP / 4, MT = MC / 16, KG = DXP / 16;
  const uint32_t m = lane & 15u, g = lane >> 4;
  const float *wa = wl + m * Row + g * G4;
  constexpr int GROUPS = KG * NMAPS * MT;
  auto operand = [&](int idx) {
    const int sg = idx / (NMAPS * MT), map = (idx / MT) % NMAPS, mt = idx % MT;
    return *reinterpret_cast<const wg_f4 *>(wa + (size_t)map * MC * Row + mt * 16 * Row + 4 * sg);
  };
  wg_f4 cur = operand(0);
#pragma unroll
  for (int idx = 0; idx < GROUPS; ++idx) {
    const int sg = idx / (NMAPS * MT), map = (idx / MT) % NMAPS, mt = idx % MT;
    wg_f4 nxt = cur;
    if (idx + 1 < GROUPS) nxt = operand(idx + 1);
    if ((uint32_t)mt < mtiles && (uint32_t)sg < kgroups) {      // (uniform: tiles of padding multiply nothing)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[map][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[t], bx[nt][4 * sg + t], acc[map][mt][nt], 0, 0, 0);
      }
    }
    if ((idx + 1) % (NMAPS * MT) == 0) {      // this k-group's inputs are spent: the next tile's take their registers
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t s = 4 * sg + t;
        const bool real = 4u * s + g < din;      // (input 4 s + g of the row)
        bx[0][s] = real ? next0[4 * s] : 0.0f;
        bx[1][s] = real ? next1[4 * s] : 0.0f;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
}

template <int MC, int NMAPS>
__device__ __forceinline__ void wideg_offsets(const WideGArgs &a, uint32_t b, uint32_t g, uint32_t chunk,
                                              wg_f4 (&acc)[NMAPS][MC / 16][2]) {
#pragma unroll
  for (int map = 0; map < NMAPS; ++map) {
    const float *off = a.off[map];
#pragma unroll
    for (int mt = 0; mt < MC / 16; ++mt) {
      wg_f4 o = {0.0f, 0.0f, 0.0f, 0.0f};
      const uint32_t col = chunk * MC + 16 * mt + 4 * g;
      if (off != nullptr) o = wideg_load4(off + (int64_t)b * a.off_sb[map], col, a.dout, a.vec_out != 0u);
      acc[map][mt][0] = o;
      acc[map][mt][1] = o;
    }
  }
}

// sum over the four lanes (g = 0 .. 3) that share particle n, in the order ((g0 + g1) + g2) + g3, returned to all
__device__ __forceinline__ float wideg_particle_sum(float v, uint32_t lane) {
  const uint32_t n = lane & 15u;
  const float v0 = __shfl(v, (int)n, kWave), v1 = __shfl(v, (int)(n + 16u), kWave), v2 = __shfl(v, (int)(n + 32u), kWave),
              v3 = __shfl(v, (int)(n + 48u), kWave);
  return ((v0 + v1) + v2) + v3;
}

// a particle's log-weight out of its record of raw squared distances: chunks in ascending order, then the three densities
// as the 128-wide kernels form them
__device__ __forceinline__ float wideg_log_weight(const WideGArgs &a, const float *rec, float g_last, bool g_in_register) {
  float ps = 0.0f, qs = 0.0f, gs = 0.0f;
  for (uint32_t c = 0; c < a.chunks_draw; ++c) {
    ps = ps + rec[2 * c];
    qs = qs + rec[2 * c + 1];
  }
  const uint32_t stored = g_in_register ? a.chunks_emit - 1 : a.chunks_emit;
  for (uint32_t c = 0; c < stored; ++c) gs = gs + rec[2 * a.chunks_draw + c];
  if (g_in_register) gs = gs + g_last;
  const float s_p = a.s_p[0], s_g = a.s_g[0], s_q = a.s_q[0];
  const float half_log_2pi = LgConst<float>::half_log_2pi();
  const float lp = (-ps) / (2.0f * (s_p * s_p)) - (float)a.dx * (Num<float>::log(s_p) + half_log_2pi);
  const float lq = (-qs) / (2.0f * (s_q * s_q)) - (float)a.dx * (Num<float>::log(s_q) + half_log_2pi);
  const float lg = (-gs) / (2.0f * (s_g * s_g)) - (float)a.dout * (Num<float>::log(s_g) + half_log_2pi);
  return (lp + lg) - lq;
}

// tile -> (batch row, first particle of the tile inside it)
__device__ __forceinline__ void wideg_tile_place(const WideGArgs &a, int64_t tile, uint32_t &b, uint32_t &k0) {
  b = (uint32_t)(tile / a.tiles_per_row);
  k0 = (uint32_t)(tile - (int64_t)b * a.tiles_per_row) * kWgTile;
}

// K17g
template <int DXP, int MC, bool GATHER>
__global__ __launch_bounds__(kWgThreads) void affine_wideg_draw_kernel(WideGArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
  float *wl = reinterpret_cast<float *>(wg_smem);
  const uint32_t chunk = blockIdx.y;
  wideg_stage_weights<DXP, MC, 2>(a, wl, chunk);
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t n = lane & 15u, g = lane >> 4;
  const float s_q = a.s_q[0];
  const uint32_t K = a.K, din = a.din;
  const bool vec = a.vec_out != 0u;
  const uint32_t kgroups = (din + 15u) / 16u;
  const uint32_t rows_here = a.dout > chunk * MC ? a.dout - chunk * MC : 0u;      // real outputs in this chunk
  const uint32_t mtiles = (rows_here + 15u) / 16u < (uint32_t)(MC / 16) ? (rows_here + 15u) / 16u : (uint32_t)(MC / 16);
  const int64_t tiles = (int64_t)a.B * a.tiles_per_row;
  const int64_t tile_stride = (int64_t)gridDim.x * (kWgThreads / 64);
  uint32_t bad = 0;
  // the rows of x_{t-1} the lane's two columns read (through the ancestors), clamped inside the batch row for a masked tail
  auto rows_of = [&](int64_t tile, const float *(&src)[2]) {
    uint32_t b, k0;
    wideg_tile_place(a, tile, b, k0);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      uint32_t k = k0 + 16u * nt + n;
      if (k >= K) k = K - 1u;
      int64_t row = (int64_t)b * K + k;
      if constexpr (GATHER) {
        int64_t anc = a.anc[row];
        if (anc < 0 || anc >= (int64_t)K) {      // K2 writes K for a degenerate row (flagged there); never fault on it
          bad = 1;
          anc = anc < 0 ? 0 : (int64_t)K - 1;
        }
        row = (int64_t)b * K + anc;
      }
      src[nt] = a.x_in + row * din + g;
    }
  };
  float bx[2][DXP / 4];
  int64_t tile = (int64_t)blockIdx.x * (kWgThreads / 64) + wave;
  if (tile < tiles) {
    const float *src0[2];
    rows_of(tile, src0);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int s = 0; s < DXP / 4; ++s) bx[nt][s] = 4u * s + g < din ? src0[nt][4 * s] : 0.0f;
  }
  for (; tile < tiles; tile += tile_stride) {
    uint32_t b, k0;
    wideg_tile_place(a, tile, b, k0);
    const int64_t tile_next = tile + tile_stride < tiles ? tile + tile_stride : tile;      // (behind the last: fetched again, dropped)
    const float *src_next[2];
    rows_of(tile_next, src_next);
    wg_f4 acc[2][MC / 16][2];
    wideg_offsets<MC, 2>(a, b, g, chunk, acc);
    wideg_products<DXP, MC, 2>(wl, lane, bx, acc, src_next[0], src_next[1], mtiles, kgroups, din);
    // ---- the draw and the two squared distances, outputs chunk * MC + 16 mt + 4 g + r of the lane's two particles ------
    float q_sum[2] = {0.0f, 0.0f}, p_sum[2] = {0.0f, 0.0f};
    int64_t part[2];
    bool live[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const uint32_t k = k0 + 16u * nt + n;
      live[nt] = k < K;
      part[nt] = (int64_t)b * K + (live[nt] ? k : K - 1u);
    }
#pragma unroll
    for (int mt = 0; mt < MC / 16; ++mt) {
      const uint32_t col = chunk * MC + 16 * mt + 4 * g;
      if (col < a.dout) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const wg_f4 e = wideg_load4(a.eps + part[nt] * a.dout, col, a.dout, vec);
          wg_f4 x;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float noise = e[r] * s_q;
            x[r] = acc[0][mt][nt][r] + noise;
            if (vec || col + r < a.dout) {      // (a row that is not whole pieces of four: its last piece holds padding)
              const float dq = x[r] - acc[0][mt][nt][r], dp = x[r] - acc[1][mt][nt][r];
              q_sum[nt] = fma_t(dq, dq, q_sum[nt]);
              p_sum[nt] = fma_t(dp, dp, p_sum[nt]);
            }
          }
          if (live[nt]) wideg_store4(a.out_x + part[nt] * a.dout, col, a.dout, vec, x);
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float qs = wideg_particle_sum(q_sum[nt], lane), ps = wideg_particle_sum(p_sum[nt], lane);
      if (g == 0 && live[nt]) {
        float *rec = a.sums + part[nt] * a.sums_stride + 2 * chunk;
        rec[0] = ps;
        rec[1] = qs;
      }
    }
  }
  if (bad != 0u) raise_flag(a.flags, AESMC_FLAG_INDEX_OUT_OF_RANGE);
}

// K18g.  FINISH (one emission chunk): the log-weight is formed here from the record K17g left; else the chunk's raw sum
// joins the record and the combine launch forms it.
template <int DXP, int MC>
__global__ __launch_bounds__(kWgThreads) void affine_wideg_emission_kernel(WideGArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
  float *wl = reinterpret_cast<float *>(wg_smem);
  const uint32_t chunk = blockIdx.y;
  wideg_stage_weights<DXP, MC, 1>(a, wl, chunk);
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t n = lane & 15u, g = lane >> 4;
  const uint32_t K = a.K, din = a.din;
  const bool vec = a.vec_out != 0u;
  const uint32_t kgroups = (din + 15u) / 16u;
  const uint32_t rows_here = a.dout > chunk * MC ? a.dout - chunk * MC : 0u;
  const uint32_t mtiles = (rows_here + 15u) / 16u < (uint32_t)(MC / 16) ? (rows_here + 15u) / 16u : (uint32_t)(MC / 16);
  const bool finish = a.chunks_emit == 1u;
  const int64_t tiles = (int64_t)a.B * a.tiles_per_row;
  const int64_t tile_stride = (int64_t)gridDim.x * (kWgThreads / 64);
  auto rows_of = [&](int64_t tile, const float *(&src)[2]) {
    uint32_t b, k0;
    wideg_tile_place(a, tile, b, k0);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      uint32_t k = k0 + 16u * nt + n;
      if (k >= K) k = K - 1u;
      src[nt] = a.x_in + ((int64_t)b * K + k) * din + g;
    }
  };
  float bx[2][DXP / 4];
  int64_t tile = (int64_t)blockIdx.x * (kWgThreads / 64) + wave;
  if (tile < tiles) {
    const float *src0[2];
    rows_of(tile, src0);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int s = 0; s < DXP / 4; ++s) bx[nt][s] = 4u * s + g < din ? src0[nt][4 * s] : 0.0f;
  }
  for (; tile < tiles; tile += tile_stride) {
    uint32_t b, k0;
    wideg_tile_place(a, tile, b, k0);
    const int64_t tile_next = tile + tile_stride < tiles ? tile + tile_stride : tile;
    const float *src_next[2];
    rows_of(tile_next, src_next);
    wg_f4 acc[1][MC / 16][2];
    wideg_offsets<MC, 1>(a, b, g, chunk, acc);
    wideg_products<DXP, MC, 1>(wl, lane, bx, acc, src_next[0], src_next[1], mtiles, kgroups, din);
    float g_sum[2] = {0.0f, 0.0f};
#pragma unroll
    for (int mt = 0; mt < MC / 16; ++mt) {
      const uint32_t col = chunk * MC + 16 * mt + 4 * g;
      if (col < a.dout) {
        const wg_f4 yv = wideg_load4(a.y + (int64_t)b * a.y_sb, col, a.dout, vec);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (vec || col + r < a.dout) {
              const float d = yv[r] - acc[0][mt][nt][r];
              g_sum[nt] = fma_t(d, d, g_sum[nt]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float gs = wideg_particle_sum(g_sum[nt], lane);
      const uint32_t k = k0 + 16u * nt + n;
      if (g == 0 && k < K) {
        const int64_t p = (int64_t)b * K + k;
        float *rec = a.sums + p * a.sums_stride;
        if (finish) a.out_lw[p] = wideg_log_weight(a, rec, gs, true);
        else rec[2 * a.chunks_draw + chunk] = gs;
      }
    }
  }
}

// ---- host side: which instantiation a width takes, and the launchers (one translation unit each: the unrolled kernels
// are slow to compile) ----------------------------------------------------------------------------------------------------
constexpr int kWgMinDim = 17, kWgMaxDim = 256;      // (rows of at most 16 values are the item kernels')

// the smallest padded extent that holds a row of `d` values (0: not covered)
static inline int wideg_padded(int64_t d) {
  const int extents[] = {32, 48, 64, 96, 128, 192, 256};
  for (int e : extents)
    if (d <= e) return e;
  return 0;
}
// output rows per chunk of K17g (two maps resident) and K18g (one) at padded input extent `dxp`
static inline int wideg_draw_chunk(int dxp) { return dxp <= 128 ? dxp : 64; }
static inline int wideg_emit_chunk(int64_t dy) { return dy <= 64 ? 64 : 128; }
static inline uint32_t wideg_chunks(int64_t rows, int chunk) { return (uint32_t)((rows + chunk - 1) / chunk); }

int wideg_launch_draw(const WideGArgs &a, int dxp, bool gather, hipStream_t s);
int wideg_launch_emission(const WideGArgs &a, int dxp, hipStream_t s);      // + the combine launch when chunks_emit > 1

}  // namespace aesmc
