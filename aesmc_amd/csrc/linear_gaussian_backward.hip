// K11, K12, K14: the backward kernels of the linear-Gaussian particle propagation (see linear_gaussian.hip
// for the forward kernels, the arithmetic contract and the reference call sites).  The weight gradients are
// contractions over the particle index and run on the matrix cores; everything per particle stays on the
// vector ALUs, one lane = one particle.
#include "linear_gaussian_backward.hpp"
namespace aesmc {

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
// One row of the next step's per-child gradient, for the lane that owns the children's ancestor.
template <typename T, int DP, bool EXACT>
__device__ __forceinline__ void lg_child_row(const T *__restrict__ row, uint32_t dx, T (&v)[DP]) {
  if constexpr (EXACT && (DP * sizeof(T)) % 8 == 0) {      // rows of whole 8-byte pairs, 8-byte aligned
    typedef T Pair __attribute__((ext_vector_type(8 / sizeof(T))));
    constexpr int PER = 8 / sizeof(T);
    const Pair *src = reinterpret_cast<const Pair *>(row);
#pragma unroll
    for (int j = 0; j < DP / PER; ++j) {
      const Pair q = src[j];
#pragma unroll
      for (int e = 0; e < PER; ++e) v[j * PER + e] = q[e];
    }
  } else {
#pragma unroll
    for (int j = 0; j < DP; ++j) v[j] = (uint32_t)j < dx ? row[min(j, (int)dx - 1)] : T(0);
  }
}

#ifndef LG_STEP_WAVES
#define LG_STEP_WAVES 3
#endif
#ifndef LG_STEP_UNROLL
#define LG_STEP_UNROLL 2
#endif
// EXACT: both extents equal DP (the host checks) — every extent test, row stride and LDS offset folds.
template <typename T, int DP, int PPL, bool EXACT>
__global__ __launch_bounds__(kLgBlock, (PPL == 2 || sizeof(T) == 8) ? 2 : LG_STEP_WAVES) void affine_step_backward_kernel(
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
  T *tchild = tu + (TP * max(lx.rs, ly.rs) + 16);      // only with out.child_stage
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
  // through the ancestors (the forward step never wrote x_{t-1}[ancestors]: neither does this): a lane fetches
  // the rows of its own particles; their indices are loaded one tile ahead
  const bool gathers = out.gat.idx != nullptr;
  int64_t ranc[PPL];
  // Exact extents with rows of whole 8-byte pairs: the NEXT tile's rows of x_t (16-byte vectors) and of x_{t-1}
  // (8-byte pieces through the ancestors) wait in registers while this tile is worked on, as in the forward kernels —
  // the tile's first barrier then waits for LDS stores, not for HBM.
  constexpr bool PREF = EXACT && (DP * sizeof(T)) % 8 == 0;
  constexpr int GQ = PREF ? (int)(DP * sizeof(T) / 8) : 1;
  constexpr int NVX = PREF ? (int)((PPL * DP + Vec16<T>::N - 1) / Vec16<T>::N) : 1;
  uint32_t rg[PPL * GQ * 2];
  typename Vec16<T>::type rx[NVX];
  LgGather gat8 = out.gat;
  gat8.pb = 8;
  gat8.ppr = (uint32_t)GQ;
  const char *xprev_bytes = reinterpret_cast<const char *>(xprev);
  // ... and so do the tile's per-row vectors (offsets, observation: a few values per lane) and what the particles'
  // share of the log-sum-exp's gradient is made of
  constexpr int kTabTrips = PREF ? (kLgRowsMax * 4 * DP + kLgBlock - 1) / kLgBlock : 1;
  T held_tab[kTabTrips], held_lw[PPL], held_glw[PPL], held_lse[PPL], held_glse[PPL];
  auto small_prefetch = [&](int64_t n0, uint32_t np) {
    const uint32_t b0 = (uint32_t)(n0 / K), nrows = (uint32_t)((n0 + np - 1) / K) - b0 + 1;
    const uint32_t k0 = (uint32_t)(n0 - (int64_t)b0 * K);
#pragma unroll
    for (int trip = 0; trip < kTabTrips; ++trip) {
      const uint32_t idx = lg_tid_impl<true>() + trip * kLgBlock;
      const uint32_t j = idx % DP, a = (idx / DP) % 4, row = idx / (DP * 4);
      T value = T(0);
      if (idx < nrows * 4 * DP) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (a == (uint32_t)c && vec[c].ptr != nullptr && (int)j < vec[c].len)
            value = vec[c].ptr[(int64_t)(b0 + row) * vec[c].sb + j];
      }
      held_tab[trip] = value;
    }
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const uint32_t q = lg_tid_impl<true>() + r * kLgBlock, qq = q < np ? q : 0u;
      const int64_t n = n0 + qq;
      const uint32_t b = b0 + (k0 + qq) / K;
      held_glw[r] = grad_lw != nullptr ? grad_lw[n] : T(0);
      held_lw[r] = grad_lse != nullptr ? lw[n] : T(0);
      held_lse[r] = grad_lse != nullptr ? lse[b] : T(0);
      held_glse[r] = grad_lse != nullptr ? grad_lse[b] : T(0);
    }
  };
  if ((int64_t)blockIdx.x < tiles) {
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
    if (gathers) lg_anc_prefetch<PPL, true>(out.gat, n0, np, ranc);
    if constexpr (PREF) {
      small_prefetch(n0, np);
      lg_prefetch<T, NVX, true>(x + n0 * dx, np * dx, 0, rx);
      if (gathers) {
        lg_gather_prefetch<PPL, GQ, 8, true>(xprev_bytes, gat8, n0, np, K, ranc, rg);
        const int64_t m0 = n0 + (int64_t)gridDim.x * TP;
        if (m0 < N) lg_anc_prefetch<PPL, true>(out.gat, m0, (uint32_t)min((int64_t)TP, N - m0), ranc);
      }
    }
  }
  // The children of the lane's particles (the gather's backward folded in, below), as FLAT row numbers b K + c of the
  // next step's per-child gradient: where each particle's run begins and ends, and where the whole tile's does — the
  // indices are non-decreasing, so a tile's children are one contiguous block of rows.  Loaded one tile ahead, like
  // the ancestors.
  const bool folds = out.child_grad != nullptr;
  const bool stages = folds && out.child_stage != 0;
  const T *child_rows = reinterpret_cast<const T *>(out.child_grad);
  // (RAW entries are kept across the tile: turning them into row numbers where they are loaded would make the
  //  wavefront wait for the loads on the spot)
  int32_t raw_end[PPL], raw_before[PPL], raw_tile_before = 0, raw_tile_end = 0;
  auto lane_raw_prefetch = [&](int64_t n0, uint32_t np) {
    const uint32_t k0 = (uint32_t)(n0 % K);
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const uint32_t q = lg_tid_impl<true>() + r * kLgBlock, qq = q < np ? q : 0u;
      const int64_t n = n0 + qq;
      // (no lane-dependent branch around the loads: a particle without a predecessor re-reads its own entry)
      raw_end[r] = out.child_end[n];
      raw_before[r] = out.child_end[(k0 + qq) % K == 0 ? n : n - 1];
    }
  };
  auto tile_raw_prefetch = [&](int64_t n0, uint32_t np) {      // uniform addresses: every lane the same two entries
    raw_tile_before = out.child_end[n0 % K == 0 ? n0 : n0 - 1];
    raw_tile_end = out.child_end[n0 + np - 1];
  };
  // flat row numbers [lo, hi) of particle q of the tile that starts at n0, from its two entries
  auto child_range = [&](int64_t n0, uint32_t q, bool lives, int32_t before, int32_t end_, uint32_t &lo, uint32_t &hi) {
    const uint32_t b0 = (uint32_t)(n0 / K), k0 = (uint32_t)(n0 - (int64_t)b0 * K);
    const uint32_t kk = k0 + q, rel = kk / K;
    const bool first_of_row = kk - rel * K == 0;
    const uint32_t base = (b0 + rel) * K;
    const uint32_t end = lives ? (uint32_t)min(max(end_, 0), (int32_t)K) : 0u;
    hi = base + end;
    lo = base + min((uint32_t)max(first_of_row ? 0 : before, 0), end);
  };
  // the block of rows a tile's children occupy, from the tile's two entries: it starts at a row whose address is a
  // multiple of 16 bytes and holds a whole number of 16-byte vectors; what does not fit the LDS tile (a tile whose
  // particles have more than TP children between them) is fetched by the lanes themselves
  auto staged_range = [&](int64_t n0, uint32_t np, uint32_t &lo, uint32_t &hi) {
    uint32_t tile_lo, tile_hi, unused;
    child_range(n0, 0u, true, raw_tile_before, raw_tile_before, tile_lo, unused);
    child_range(n0, np - 1, true, raw_tile_end, raw_tile_end, unused, tile_hi);
    const uint32_t align = (uint32_t)out.child_align;
    lo = tile_lo & ~(align - 1u);
    hi = tile_hi > lo ? min(min((tile_hi + align - 1u) & ~(align - 1u), (uint32_t)N), lo + TP) : lo;
  };
  // Where the tile's layout is flat (rows end to end: every extent that is not a multiple of 4), the NEXT tile's block
  // is sent for as soon as this tile's sums have been taken from the LDS tile — loads that write LDS directly, its two
  // entries having come a tile earlier still — and lands during the rest of this tile's arithmetic.
  const bool ahead = PREF && stages && lx.mul == 0;
  uint32_t cur_lo = 0, cur_hi = 0, nxt_lo = 0, nxt_hi = 0;
  if (folds && (int64_t)blockIdx.x < tiles) {
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
    lane_raw_prefetch(n0, np);
    tile_raw_prefetch(n0, np);
    if (ahead) {
      staged_range(n0, np, cur_lo, cur_hi);
      lg_stage_rows<T, true>(child_rows + (int64_t)cur_lo * dx, (cur_hi - cur_lo) * dx, tchild, lx, 0);
      const int64_t m0 = n0 + (int64_t)gridDim.x * TP;
      if (m0 < N) tile_raw_prefetch(m0, (uint32_t)min((int64_t)TP, N - m0));
    }
  }
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TP;
    const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
    if constexpr (PREF) {
      if (gathers) lg_gather_commit<T, PPL, GQ, 8, true>(gat8, np, rg, tprev, lx);
      else lg_stage_rows<T, true>(xprev + n0 * dx, np * dx, tprev, lx, 0);
      lg_commit<T, NVX, true>(x + n0 * dx, np * dx, rx, tx, lx);
    } else {
      if (gathers) {
        lg_gather_stage<T, PPL, DP, true>(xprev_bytes, out.gat, n0, np, K, ranc, tprev, lx);
        const int64_t m0 = (tile + gridDim.x) * TP;
        if (m0 < N) lg_anc_prefetch<PPL, true>(out.gat, m0, (uint32_t)min((int64_t)TP, N - m0), ranc);
      } else {
        lg_stage_rows<T, true>(xprev + n0 * dx, np * dx, tprev, lx, 0);
      }
      lg_stage_rows<T, true>(x + n0 * dx, np * dx, tx, lx, 0);
    }
    if (gx_in != nullptr) lg_stage_rows<T, true>(gx_in + n0 * dx, np * dx, tu, lx, 0);
    uint32_t p[PPL], brow[PPL], at[PPL];
    bool live[PPL];
    lg_rows<PPL, true>(n0, np, K, p, live, brow);
    const uint32_t b0 = (uint32_t)(n0 / K), nrows = (uint32_t)((n0 + np - 1) / K) - b0 + 1;
    if constexpr (PREF) {
#pragma unroll
      for (int trip = 0; trip < kTabTrips; ++trip) {
        const uint32_t idx = lg_tid_impl<true>() + trip * kLgBlock;
        if (idx < nrows * 4 * DP) tab[idx] = held_tab[trip];
      }
    } else {
      lg_stage_table<T, DP, 4, true>(vec, b0, nrows, tab);      // the host guarantees nrows <= kLgRowsMax
    }
    // (after the commits above: the registers that held this tile's rows are free for the next tile's block of children)
    uint32_t own_lo[PPL], own_hi[PPL], staged_lo = 0, staged_hi = 0;      // staged rows: [staged_lo, staged_hi)
    if (folds) {
#pragma unroll
      for (int r = 0; r < PPL; ++r) {
        const uint32_t q = lg_tid_impl<true>() + r * kLgBlock;
        child_range(n0, q < np ? q : 0u, q < np, raw_before[r], raw_end[r], own_lo[r], own_hi[r]);
      }
      const int64_t m0 = (tile + gridDim.x) * TP;
      const uint32_t mp_ = m0 < N ? (uint32_t)min((int64_t)TP, N - m0) : 0u;
      if (ahead) {
        staged_lo = cur_lo;
        staged_hi = cur_hi;
        if (m0 < N) {
          staged_range(m0, mp_, nxt_lo, nxt_hi);
          const int64_t mm0 = m0 + (int64_t)gridDim.x * TP;
          if (mm0 < N) tile_raw_prefetch(mm0, (uint32_t)min((int64_t)TP, N - mm0));
        }
      } else if (stages) {
        staged_range(n0, np, staged_lo, staged_hi);
        lg_stage_rows<T, true>(child_rows + (int64_t)staged_lo * dx, (staged_hi - staged_lo) * dx, tchild, lx, 0);
        if (m0 < N) tile_raw_prefetch(m0, mp_);
      }
      if (m0 < N) lane_raw_prefetch(m0, mp_);
    }
    const uint32_t k0_tile = (uint32_t)(n0 - (int64_t)b0 * K);
    // offsets' gradients: a tile inside one batch row takes its sums from the matrix cores' spare column
    const bool column_sums = ONES && lg_single_row(n0, np, K);
    const int row_pass = column_sums ? 0 : row_terms;      // the terms that need the pass over the tile
    T g[PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const int64_t n = n0 + p[r];
      T value;
      if constexpr (PREF) {
        value = held_glw[r];
        if (grad_lse != nullptr) value = value + held_glse[r] * Num<T>::exp(held_lw[r] - held_lse[r]);
      } else {
        value = grad_lw != nullptr ? grad_lw[n] : T(0);
        if (grad_lse != nullptr) value = value + grad_lse[brow[r]] * Num<T>::exp(lw[n] - lse[brow[r]]);
      }
      g[r] = live[r] ? value : T(0);
      at[r] = p[r] * lx.rs;
    }
    // (the block of children sent for during the last tile writes LDS from the vector-memory side: it has landed once
    //  that counter is drained — the register prefetches committed above have drained it anyway)
    if (ahead) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    if (folds) {
      // ---- the gather's backward folded in: w += sum of the particle's children's rows ------------------------
      // The lane adds its run in k order (fixed, so reproducible; the stand-alone segmented-sum kernel associates
      // differently: equal to rounding) — out of the staged block where the rows are there, from HBM otherwise; a run
      // longer than kLgChildLimit (a collapsed particle system) is shared out over the wavefront.
      const int lane = (int)(threadIdx.x & (kWave - 1));
#pragma unroll
      for (int r = 0; r < PPL; ++r) {
        const uint32_t lo = own_lo[r], hi = own_hi[r];
        const uint32_t own_last = min(hi, lo + (uint32_t)kLgChildLimit);
        T acc[DP];
#pragma unroll
        for (int j = 0; j < DP; ++j) acc[j] = T(0);
        uint32_t c = lo;
        // (lo >= staged_lo when the ranges are what the resampling launch wrote: the block begins with the tile's first run)
        const uint32_t in_lds = lo >= staged_lo ? min(own_last, staged_hi) : lo;
        // kLgChildTrip rows in flight per trip, added in k order: ((0 + c0) + c1) + c2 ...  (rows past the run's end are
        // re-reads of its last one, added as zero: no branch between the loads and the adds)
        for (; c < in_lds; c += kLgChildTrip) {
          T row[kLgChildTrip][DP];
#pragma unroll
          for (int i = 0; i < kLgChildTrip; ++i)
            lg_child_row<T, DP, EXACT>(tchild + (min(c + i, in_lds - 1) - staged_lo) * lx.rs, dx, row[i]);
#pragma unroll
          for (int i = 0; i < kLgChildTrip; ++i) {
            const bool there = c + i < in_lds;
#pragma unroll
            for (int j = 0; j < DP; ++j) acc[j] = acc[j] + (there ? row[i][j] : T(0));
          }
        }
        c = min(c, max(in_lds, lo));
        for (; c < own_last; ++c) {      // (rows the tile had no room for)
          T a[DP];
          lg_child_row<T, DP, EXACT>(child_rows + (int64_t)c * dx, dx, a);
#pragma unroll
          for (int j = 0; j < DP; ++j) acc[j] = acc[j] + a[j];
        }
        uint64_t todo = __ballot(own_last < hi);
        while (todo != 0) {        // wavefront-uniform: every lane helps the lane whose run is long
          const int leader = __ffsll((long long)todo) - 1;
          const uint32_t from = (uint32_t)__shfl((int)own_last, leader, kWave), to = (uint32_t)__shfl((int)hi, leader, kWave);
          T part[DP];
#pragma unroll
          for (int j = 0; j < DP; ++j) part[j] = T(0);
          for (uint32_t cc = from + lane; cc < to; cc += kWave) {
            T row[DP];
            lg_child_row<T, DP, EXACT>(child_rows + (int64_t)cc * dx, dx, row);
#pragma unroll
            for (int j = 0; j < DP; ++j) part[j] += row[j];
          }
#pragma unroll
          for (int j = 0; j < DP; ++j) {
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) part[j] += __shfl_xor(part[j], off, kWave);
            if (lane == leader) acc[j] += part[j];
          }
          todo &= todo - 1;
        }
#pragma unroll
        for (int j = 0; j < DP; ++j)
          if ((uint32_t)j < dx && live[r]) w[j][r] = w[j][r] + acc[j];
      }
    }
    if constexpr (PREF) {
      const int64_t next = tile + gridDim.x;
      if (ahead) {
        // every wavefront has taken its sums: the LDS tile is free for the next tile's block
        lg_lds_barrier();
        if (next < tiles) lg_stage_rows_async<T>(child_rows + (int64_t)nxt_lo * dx, (nxt_hi - nxt_lo) * dx, tchild);
        cur_lo = nxt_lo;
        cur_hi = nxt_hi;
      }
      // the next tile's rows go out now and fly during the rest of this tile's arithmetic
      if (next < tiles) {
        const int64_t m0 = next * TP;
        const uint32_t mp_ = (uint32_t)min((int64_t)TP, N - m0);
        small_prefetch(m0, mp_);
        lg_prefetch<T, NVX, true>(x + m0 * dx, mp_ * dx, 0, rx);
        if (gathers) {
          lg_gather_prefetch<PPL, GQ, 8, true>(xprev_bytes, gat8, m0, mp_, K, ranc, rg);      // its indices came a tile ago
          const int64_t nn0 = (next + gridDim.x) * TP;
          if (nn0 < N) lg_anc_prefetch<PPL, true>(out.gat, nn0, (uint32_t)min((int64_t)TP, N - nn0), ranc);
        }
      }
    }
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
    if (ONES) lg_flush_column_sums<T>(acc_c, scratch + 64, column_sums && (row_terms & 2));
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
    if (ONES) lg_flush_column_sums<T>(acc_a, scratch, column_sums && (row_terms & 1));
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
    if (ONES) lg_flush_column_sums<T>(acc_q, scratch + 128, column_sums && (row_terms & 4));
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
    // (scratch[0 .. 191] is written next by this tile's successor's flushes, several barriers on, or by wavefront 0's
    //  own part of the closing records)
    if (ONES && column_sums && row_terms != 0)
      lg_store_column_sums<T>(scratch, rows + tile * 3 * (kLgRowsMax * 16), row_terms);
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
  if (out.carry != nullptr && threadIdx.x < kLgRecord) {
    // (each lane wrote the four elements it now reads: no barrier; the records carried are added in record order)
    const T *carry = reinterpret_cast<const T *>(out.carry);
    T own[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) own[m] = record[m * kLgRecord + threadIdx.x];
    for (int r = blockIdx.x; r < out.carry_records; r += gridDim.x) {
      const T *c = carry + (int64_t)r * 4 * kLgRecord + threadIdx.x;
      T v[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) v[m] = c[m * kLgRecord];
#pragma unroll
      for (int m = 0; m < 4; ++m) own[m] += v[m];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) record[m * kLgRecord + threadIdx.x] = own[m];
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
  int32_t row_terms, matrices, row_blocks;    // records per tile; matrix blocks; row blocks per term (kLgFinishRows rows each)
  int32_t column_sums;                        // single-row tiles hold their sums in slot 0 (lg_store_column_sums)
  int64_t B, N;
  uint32_t K, TP;
};
constexpr int kLgFinishSplit = 4;      // workgroups per matrix of the finishing launch
constexpr int kLgFinishRows = 16;      // batch rows per workgroup of its row part (64: 10.7 us at c4; 16: 9.1)
template <typename T>
__global__ __launch_bounds__(1024) void lg_finish_kernel(const T *__restrict__ ws, int nblocks, int record, LgFinish f) {
  if ((int)blockIdx.x >= f.matrices * kLgFinishSplit) {
    // ---- one lane per (batch row, column): the few tiles that cover the row, in tile order
    const int r = (int)blockIdx.x - f.matrices * kLgFinishSplit, term = r / f.row_blocks;
    T *goff = reinterpret_cast<T *>(f.goff[term]);
    if (goff == nullptr) return;
    if (threadIdx.x >= 16 * kLgFinishRows) return;
    const int64_t b = (int64_t)(r - term * f.row_blocks) * kLgFinishRows + (threadIdx.x >> 4);
    const uint32_t j = threadIdx.x & 15u, d = (uint32_t)f.goff_d[term];
    if (b >= f.B || j >= d) return;
    const T *rows = reinterpret_cast<const T *>(f.row_ws);
    const int64_t first = b * f.K / f.TP, last = ((b + 1) * f.K - 1) / f.TP;
    // (sixteen tiles' records in flight, then added in tile order: the same association as one at a time)
    auto tile_value = [&](int64_t tile) {
      const int64_t n0 = tile * f.TP, b0 = n0 / f.K;
      const T *record = rows + (tile * f.row_terms + term) * (kLgRowsMax * 16);
      if (f.column_sums && lg_single_row(n0, (uint32_t)min((int64_t)f.TP, f.N - n0), f.K))
        return record[j];     // (slot 0 whichever row it is: lg_store_column_sums)
      return record[(b - b0) * 16 + j];
    };
    T sum = T(0);
    int64_t tile = first;
    for (; tile + 16 <= last + 1; tile += 16) {
      T v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = tile_value(tile + u);
#pragma unroll
      for (int u = 0; u < 16; ++u) sum += v[u];
    }
    for (; tile + 4 <= last + 1; tile += 4) {
      T v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = tile_value(tile + u);
#pragma unroll
      for (int u = 0; u < 4; ++u) sum += v[u];
    }
    for (; tile <= last; ++tile) sum += tile_value(tile);
    goff[b * d + j] = sum;
    return;
  }
  // element e of matrix m: a matrix is shared out over kLgFinishSplit workgroups (64 elements each); sixteen lanes per
  // element each sum a sixteenth of the workgroups' records (in workgroup order, eight loads in flight), then the
  // sixteenths are added in order — fixed association, reproducible
  __shared__ T part[16 * 64];
  const int m = blockIdx.x / kLgFinishSplit, lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int e = (blockIdx.x - m * kLgFinishSplit) * 64 + lane;
  const int per = (nblocks + 15) / 16, b0 = seg * per, b1 = min(nblocks, b0 + per);
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
  part[seg * 64 + lane] = sum;
  __syncthreads();
  if (seg == 0) {
    T total = part[lane];
#pragma unroll
    for (int q = 1; q < 16; ++q) total += part[q * 64 + lane];
    const int j = e >> 4, i = e & 15;
    T *out = reinterpret_cast<T *>(f.out[m]);
    if (out != nullptr && j < f.rows[m] && i < f.cols[m]) out[j * f.cols[m] + i] = total;
  }
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
    f.row_blocks = goff != nullptr ? (int32_t)((B + kLgFinishRows - 1) / kLgFinishRows) : 0;
    f.B = B; f.K = (uint32_t)K; f.TP = (uint32_t)tp;
    hipLaunchKernelGGL(lg_finish_kernel<T>, dim3((unsigned)(f.matrices * kLgFinishSplit + f.row_blocks)), dim3(1024), 0, stream,
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
                                            const void *gx_in = nullptr, const int64_t *anc_idx = nullptr,
                                            int32_t *flags = nullptr, const void *child_grad = nullptr,
                                            const int32_t *child_end = nullptr, aesmc_affine_chain *chain = nullptr) {
  const int64_t N = B * K;
  const int64_t dx = mp->dout, dy = mg->dout;
  const int dp = lg_pad_dim(std::max(dx, dy));
  static const int forced = [] { const char *v = measurement_knob("AESMC_LG_BWD_PPL"); return v != nullptr ? atoi(v) : 0; }();   // measurement knob
  // rows of ten float32 values, tiles inside one batch row, an nn.Linear's weights: the second form of the step kernel
  bool rows_form = false;
  if constexpr (sizeof(T) == 4) rows_form = step && N % kLgBlock == 0 && affine_step_backward_rows_covers(mp, mg, mq, B, K);
  int ppl = (sizeof(T) == 4 && dp <= 12 && !lg_few_tiles(N)) ? 2 : 1;
  // the step kernel is latency-bound: one particle per lane and three workgroups per CU where the registers
  // allow it without spills (d = 10: 322 -> 300 us; d = 8: no difference; d = 12: 448 -> 477, kept at two)
  if (step && dp <= 10) ppl = 1;
  if (forced == 1 || forced == 2) ppl = (sizeof(T) == 4 && dp <= 12) ? forced : 1;
  if (rows_form) ppl = 1;
  size_t lds = 0;
  for (; ppl >= 1; --ppl) {
    const size_t tp = (size_t)kLgBlock * ppl;
    lds = sizeof(T) * (16 + 6 * (size_t)dp * dp + kLgRowPart + (size_t)kLgRowsMax * 4 * dp + 2 * lg_tile_elems<T>(tp, dx) +
                       std::max(lg_tile_elems<T>(tp, dx), lg_tile_elems<T>(tp, dy)));
    if (lds <= (ppl > 1 ? (size_t)78 * 1024 : kLgLdsLimit) && lg_rows_spanned((int64_t)tp, K) <= kLgRowsMax) break;
  }
  if (ppl < 1) return AESMC_ERR_UNSUPPORTED;   // fewer than ~43 particles per batch row: the caller takes the unfused route
  // the children's rows go through a tile of their own where that costs no resident workgroup (else lanes fetch them)
  bool child_stage = false;
  if (child_grad != nullptr) {
    const int per_cu = (ppl == 2 || sizeof(T) == 8) ? 2 : (step ? LG_STEP_WAVES : 3);
    const size_t with_tile = lds + sizeof(T) * lg_tile_elems<T>((size_t)kLgBlock * ppl, dx);
    static const bool off = [] { const char *v = measurement_knob("AESMC_LG_CHILD_STAGE"); return v != nullptr && atoi(v) == 0; }();
    if (!off && with_tile * per_cu <= (size_t)160 * 1024 && with_tile <= kLgLdsLimit) {
      child_stage = true;
      lds = with_tile;
    }
  }
  const int64_t tiles = (N + (int64_t)kLgBlock * ppl - 1) / ((int64_t)kLgBlock * ppl);
  const int forced_grid = step ? affine_step_backward_forced_grid() : 0;      // test hook (aesmc_test_set_step_backward)
  int grid = (int)std::min<int64_t>(lg_persistent_grid(tiles, lds, (ppl == 2 || sizeof(T) == 8) ? 2 : (step ? LG_STEP_WAVES : 3)), kLgMaxGrid);   // what the registers allow
  if (rows_form) grid = (int)affine_step_backward_rows_grid(B, K, mp->dout);
  if (forced_grid > 0) grid = (int)std::min<int64_t>(std::min<int64_t>(forced_grid, tiles), kLgMaxGrid);
  const int row_terms = (o->grad_offset_p != nullptr ? 1 : 0) | (o->grad_offset_g != nullptr ? 2 : 0) |
                        (o->grad_offset_q != nullptr ? 4 : 0);
  const size_t need = lg_record_elems() + (row_terms != 0 ? lg_row_elems(N, 3) : 0);
  if (ws_bytes < need * sizeof(T)) return AESMC_ERR_WORKSPACE;
  T *row_ws = row_terms != 0 ? static_cast<T *>(ws) + lg_record_elems() : nullptr;
  LgBackwardOut out;
  out.gxprev = o->grad_x_prev; out.gx = o->grad_x; out.up = o->grad_loc_p; out.ug = o->grad_loc_g;
  out.uq = o->grad_loc_q; out.ws = ws; out.rows = row_ws; out.row_terms = row_terms;
  out.gx_in = gx_in; out.want_scale_q = (o->grad_scales != nullptr || (chain != nullptr && chain->defer > 1)) ? 1 : 0;
  out.gat = lg_gather(anc_idx, flags, (size_t)dx * sizeof(T));
  out.child_grad = child_grad;
  out.child_end = child_end;
  out.child_stage = child_stage ? 1 : 0;
  {
    size_t row_bytes = (size_t)dx * sizeof(T), align = 1;
    while ((row_bytes * align) % 16 != 0) align *= 2;
    out.child_align = (int)align;
  }
  out.carry = chain != nullptr ? chain->carry : nullptr;
  out.carry_records = chain != nullptr && chain->carry != nullptr ? chain->carry_records : 0;
  {
    const size_t used = (need * sizeof(T) + 15) & ~(size_t)15;
    out.pairs = ws_bytes >= used + 3 * kLgPairFloats * sizeof(float) ? reinterpret_cast<float *>(static_cast<char *>(ws) + used)
                                                                     : nullptr;
    out.pairs_ready = 0;
    if (chain != nullptr && chain->pairs_in != nullptr) {      // an earlier call of this run of steps built them
      out.pairs = static_cast<float *>(const_cast<void *>(chain->pairs_in));
      out.pairs_ready = 1;
    }
    if (chain != nullptr) chain->pairs_out = nullptr;
  }
  if (chain != nullptr && !step) return AESMC_ERR_UNSUPPORTED;
  if ((child_grad != nullptr) != (child_end != nullptr) || (child_grad != nullptr && !step)) return AESMC_ERR_UNSUPPORTED;
  if (anc_idx != nullptr && (!step || N > 0x7fffffffLL)) return AESMC_ERR_UNSUPPORTED;
#define LG_BACKWARD_ARGS                                                                                            \
  static_cast<const T *>(xprev), static_cast<const T *>(x), static_cast<const T *>(y), y_sb, lg_map(mp), lg_map(mg), \
      lg_map(mq), static_cast<const T *>(sp), static_cast<const T *>(sg), static_cast<const T *>(sq),               \
      static_cast<const T *>(lw), static_cast<const T *>(lse), static_cast<const T *>(grad_lse),                    \
      static_cast<const T *>(grad_lw), out, N, (uint32_t)K
  if (rows_form) {
    if constexpr (sizeof(T) == 4) {
      const int status = launch_affine_step_backward_rows(
          static_cast<const float *>(xprev), static_cast<const float *>(x), static_cast<const float *>(y), y_sb, lg_map(mp),
          lg_map(mg), lg_map(mq), static_cast<const float *>(sp), static_cast<const float *>(sg),
          static_cast<const float *>(sq), static_cast<const float *>(lw), static_cast<const float *>(lse),
          static_cast<const float *>(grad_lse), static_cast<const float *>(grad_lw), out, N, (uint32_t)K, (unsigned)grid,
          stream);
      if (status != AESMC_OK) return status;
      if (chain != nullptr && affine_step_backward_rows_pairs()) chain->pairs_out = out.pairs;
    }
  } else if (step && dx == dp && dy == dp) {
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
  // (a deferring call leaves the weights' and scales' sums as records in `ws` for the call that carries them on)
  const bool defer = chain != nullptr && chain->defer != 0;
  if (chain != nullptr) chain->records = grid;
  f.matrices = defer ? 0 : 4;
  f.row_ws = row_ws; f.row_terms = 3;
  f.goff[0] = o->grad_offset_p; f.goff[1] = o->grad_offset_g; f.goff[2] = o->grad_offset_q;
  f.goff_d[0] = (int32_t)dx; f.goff_d[1] = (int32_t)dy; f.goff_d[2] = (int32_t)dx;
  f.row_blocks = row_terms != 0 ? (int32_t)((B + kLgFinishRows - 1) / kLgFinishRows) : 0;
  f.B = B; f.N = N; f.K = (uint32_t)K; f.TP = (uint32_t)(kLgBlock * ppl);
  f.column_sums = (step && dx == dp && dy == dp && dp < 16) ? 1 : 0;
  const unsigned finishing = (unsigned)(f.matrices * kLgFinishSplit + 3 * f.row_blocks);
  if (finishing == 0) return AESMC_OK;
  // (only row blocks — the weights' records are carried on: their 16 lanes x kLgFinishRows rows are the whole workgroup)
  hipLaunchKernelGGL(lg_finish_kernel<T>, dim3(finishing), dim3(f.matrices == 0 ? 16 * kLgFinishRows : 1024), 0, stream,
                     static_cast<const T *>(ws), grid, 4 * kLgRecord, f);   // one finishing launch for everything
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

// The finishing launch alone, for records a deferring call left behind and nobody carried on.
template <typename T>
static int launch_affine_collect(const void *ws, int records, int64_t dx, int64_t dy,
                                 const aesmc_affine_logweight_grads *o, hipStream_t stream) {
  LgFinish f = {};
  f.out[0] = o->grad_weight_p; f.rows[0] = (int32_t)dx; f.cols[0] = (int32_t)dx;
  f.out[1] = o->grad_weight_g; f.rows[1] = (int32_t)dy; f.cols[1] = (int32_t)dx;
  f.out[2] = o->grad_weight_q; f.rows[2] = (int32_t)dx; f.cols[2] = (int32_t)dx;
  f.out[3] = o->grad_scales; f.rows[3] = 1; f.cols[3] = 3;
  f.matrices = 4;
  hipLaunchKernelGGL(lg_finish_kernel<T>, dim3((unsigned)(4 * kLgFinishSplit)), dim3(1024), 0, stream,
                     static_cast<const T *>(ws), records, 4 * kLgRecord, f);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

}  // namespace aesmc

using namespace aesmc;

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


static int affine_step_backward_entry(
    int dtype, const void *x_prev, const int64_t *ancestors, int32_t *flags, const void *x, const void *y,
    int64_t y_stride_b, const aesmc_affine_map *transition,
    const aesmc_affine_map *emission, const aesmc_affine_map *proposal, const void *scale_p, const void *scale_g,
    const void *scale_q, const void *lw, const void *lse, const void *grad_lse, const void *grad_lw,
    const void *grad_x, const void *child_grad, const int32_t *child_end, const aesmc_affine_logweight_grads *out,
    void *ws, size_t ws_bytes, int64_t B, int64_t K, void *stream, aesmc_affine_chain *chain);

extern "C" int aesmc_affine_step_backward(
    int dtype, const void *x_prev, const void *x, const void *y, int64_t y_stride_b, const aesmc_affine_map *transition,
    const aesmc_affine_map *emission, const aesmc_affine_map *proposal, const void *scale_p, const void *scale_g,
    const void *scale_q, const void *lw, const void *lse, const void *grad_lse, const void *grad_lw,
    const void *grad_x, const aesmc_affine_logweight_grads *out, void *ws, size_t ws_bytes, int64_t B, int64_t K,
    void *stream) {
  return affine_step_backward_entry(dtype, x_prev, nullptr, nullptr, x, y, y_stride_b, transition, emission, proposal,
                                    scale_p, scale_g, scale_q, lw, lse, grad_lse, grad_lw, grad_x, nullptr, nullptr, out,
                                    ws, ws_bytes, B, K, stream, nullptr);
}

extern "C" int aesmc_affine_step_backward_resampled(
    int dtype, const void *x_src, const int64_t *ancestors, const void *x, const void *y, int64_t y_stride_b,
    const aesmc_affine_map *transition, const aesmc_affine_map *emission, const aesmc_affine_map *proposal,
    const void *scale_p, const void *scale_g, const void *scale_q, const void *lw, const void *lse,
    const void *grad_lse, const void *grad_lw, const void *grad_x, const void *child_grad, const int32_t *child_end,
    const aesmc_affine_logweight_grads *out, void *ws, size_t ws_bytes, int32_t *flags, aesmc_affine_chain *chain,
    int64_t B, int64_t K, void *stream) {
  if (ancestors == nullptr || (((uintptr_t)ancestors) & 7u) != 0) return AESMC_ERR_INVALID_ARGUMENT;
  if (chain != nullptr) {
    chain->records = 0;
    if (chain->carry != nullptr && (chain->carry_records <= 0 || chain->carry_records > kLgMaxGrid || chain->carry == ws ||
                                    !aligned16(chain->carry)))
      return AESMC_ERR_INVALID_ARGUMENT;
    if (chain->defer != 0 && out != nullptr &&
        (out->grad_weight_p != nullptr || out->grad_weight_g != nullptr || out->grad_weight_q != nullptr ||
         out->grad_scales != nullptr))
      return AESMC_ERR_INVALID_ARGUMENT;      // a deferring call writes none of them
  }
  if ((child_grad == nullptr) != (child_end == nullptr) || (((uintptr_t)child_end) & 3u) != 0 ||
      (child_grad != nullptr && child_grad == (out != nullptr ? out->grad_x_prev : nullptr)))
    return AESMC_ERR_INVALID_ARGUMENT;
  return affine_step_backward_entry(dtype, x_src, ancestors, flags, x, y, y_stride_b, transition, emission, proposal,
                                    scale_p, scale_g, scale_q, lw, lse, grad_lse, grad_lw, grad_x, child_grad, child_end,
                                    out, ws, ws_bytes, B, K, stream, chain);
}

static int affine_step_backward_entry(
    int dtype, const void *x_prev, const int64_t *ancestors, int32_t *flags, const void *x, const void *y,
    int64_t y_stride_b, const aesmc_affine_map *transition,
    const aesmc_affine_map *emission, const aesmc_affine_map *proposal, const void *scale_p, const void *scale_g,
    const void *scale_q, const void *lw, const void *lse, const void *grad_lse, const void *grad_lw,
    const void *grad_x, const void *child_grad, const int32_t *child_end, const aesmc_affine_logweight_grads *out,
    void *ws, size_t ws_bytes, int64_t B, int64_t K, void *stream, aesmc_affine_chain *chain) {
  if (out == nullptr) return AESMC_ERR_INVALID_ARGUMENT;
  // the draw leaves no gradient for x_t and the location gradients have no meaning here
  if (out->grad_x != nullptr || out->grad_loc_p != nullptr || out->grad_loc_g != nullptr || out->grad_loc_q != nullptr)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (x_prev == nullptr || x == nullptr || y == nullptr || transition == nullptr || emission == nullptr ||
      proposal == nullptr || scale_p == nullptr || scale_g == nullptr || scale_q == nullptr || ws == nullptr || B < 0 ||
      K < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (grad_lw == nullptr && grad_lse == nullptr && grad_x == nullptr && child_grad == nullptr)
    return AESMC_ERR_INVALID_ARGUMENT;
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
  if ((B == 0 || K == 0) && chain != nullptr && (chain->carry != nullptr || chain->defer != 0))
    return AESMC_ERR_UNSUPPORTED;      // nothing to launch: the caller collects what it carries
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
                                                       ws_bytes, B, K, s, true, grad_x, ancestors, flags, child_grad,
                                                       child_end, chain)
             : launch_affine_logweight_backward<double>(x_prev, x, y, y_stride_b, transition, emission, proposal,
                                                        scale_p, scale_g, scale_q, lw, lse, grad_lse, grad_lw, out, ws,
                                                        ws_bytes, B, K, s, true, grad_x, ancestors, flags, child_grad,
                                                        child_end, chain);
}

extern "C" int aesmc_affine_backward_collect(int dtype, const void *ws, int32_t records, int64_t dx, int64_t dy,
                                             const aesmc_affine_logweight_grads *out, void *stream) {
  if (ws == nullptr || out == nullptr || records <= 0 || records > kLgMaxGrid || dx < 1 || dx > kLgMaxDim || dy < 1 ||
      dy > kLgMaxDim || !aligned16(ws))
    return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AESMC_F32 ? launch_affine_collect<float>(ws, records, dx, dy, out, s)
                            : launch_affine_collect<double>(ws, records, dx, dy, out, s);
}


extern "C" size_t aesmc_affine_backward_workspace_bytes(int dtype, int64_t B, int64_t K) {
  const int64_t N = (B > 0 && K > 0) ? B * K : 0;
  // (+ the rows form's interleaved weight pairs behind them: 3 maps x 256 floats, 16-byte aligned)
  return (lg_record_elems() + lg_row_elems(N, 3)) * (dtype == AESMC_F64 ? 8 : 4) + 16 + 3 * kLgPairFloats * sizeof(float);
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


