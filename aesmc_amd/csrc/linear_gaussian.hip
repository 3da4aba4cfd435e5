// K8 - K10, K15: particle propagation for linear-Gaussian model terms;
// their backward kernels K11, K12, K14 live in linear_gaussian_backward.hip, shared pieces in linear_gaussian.hpp.
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
//
// After the location, K9 follows K6 operation for operation (product rounded before the sum); K10 takes
// PyTorch's per-element log-density with the common factors out of the d-sum (one division per term:
// see the kernel) and combines the terms as (p + g) - q, as K5 does.
#include "linear_gaussian.hpp"
namespace aesmc {

// ---- K8 ----------------------------------------------------------------------------------------
// (`stream`: bit 0 the streaming-load hint; bit 1: the location leaves through tanh — `tanh(A x)` of a nonlinear
//  transition, aesmc_particle_affine_tanh — the device library's function, the one torch.tanh calls, applied to the very
//  bits the plain launch would store)
template <typename T> __device__ __forceinline__ T lg_tanh(T x);
template <> __device__ __forceinline__ float lg_tanh<float>(float x) { return ::tanhf(x); }
template <> __device__ __forceinline__ double lg_tanh<double>(double x) { return ::tanh(x); }

template <typename T, int DP, int PPL>
__global__ __launch_bounds__(kLgBlock) void particle_affine_kernel(const T *__restrict__ x1, LgMap m1,
                                                                    const T *__restrict__ x2, LgMap m2,
                                                                    const T *__restrict__ base, T *__restrict__ out,
                                                                    int64_t N, uint32_t K, int stream_and_activation) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lg_smem[];
  constexpr uint32_t TP = kLgBlock * PPL;
  const int stream = stream_and_activation & 1;
  const bool through_tanh = (stream_and_activation & 2) != 0;
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
          const T location = base != nullptr ? to[slot] + acc[j][r] : acc[j][r];
          to[slot] = through_tanh ? lg_tanh<T>(location) : location;
        }
      }
    }
  }
  __syncthreads();
  lg_store_rows(out + n0 * dout, np * dout, to, lo);
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
// GATHER (with DRAW, TAB and PREFETCH): `xprev` is the UN-resampled x_{t-1} and the tile's rows are fetched
// through the ancestor indices `gat.idx` — the resampled latent (aesmc/inference.py:102-111) is never written:
// the indices of tile t+2 are in flight while the rows of tile t+1 are, while tile t computes.
template <typename T, int DP, int PPL, bool TAB, bool PREFETCH, bool DRAW = false, int GATHER = 0>
__global__ __launch_bounds__(kLgBlock, 3) void affine_logweight_kernel(
    const T *__restrict__ xprev, const T *__restrict__ x, const T *__restrict__ y, int64_t y_sb, LgMap mp, LgMap mg,
    LgMap mq, const T *__restrict__ sp_ptr, const T *__restrict__ sg_ptr, const T *__restrict__ sq_ptr,
    T *__restrict__ out_lw, int64_t N, uint32_t K, T *__restrict__ out_x, LgGather gat) {
  // GATHER: 0, or the bytes per piece of a gathered row (4, 8, 16)
  static_assert(!GATHER || (TAB && PREFETCH), "the gathering variant is the persistent one");
  constexpr int PB = GATHER ? GATHER : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lg_smem[];
  constexpr uint32_t TP = kLgBlock * PPL;
  constexpr int NV = (PPL * DP + Vec16<T>::N - 1) / Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const uint32_t dx = mp.dout, dy = mg.dout;
  T *wpq = reinterpret_cast<T *>(lg_smem);            // [DP][DP][2]: the transition's and the proposal's weight, interleaved
  T *wg = wpq + 2 * DP * DP;
  T *tab = wg + DP * DP;                             // [kLgRowsMax][4][DP]: offsets p, q, g and the observation
  T *tprev = tab + kLgRowsMax * 4 * DP;
  const LgLayout lx = lg_layout<T>(dx);
  T *tx = tprev + (TP * lx.rs + 16);
  const T s_p = sp_ptr[0], s_g = sg_ptr[0], s_q = sq_ptr[0];
  lg_stage_weight_pair<T, DP>(mp, mq, wpq);
  lg_stage_weight<T, DP>(mg, wg);
  LgRowVec<T> vec[4] = {lg_offset_vec<T>(mp), lg_offset_vec<T>(mq), lg_offset_vec<T>(mg), {y, y_sb, (int)dy}};
  const T half_log_2pi = LgConst<T>::half_log_2pi();
  const T two_var_p = T(2) * (s_p * s_p), const_p = T(dx) * (Num<T>::log(s_p) + half_log_2pi);
  const T two_var_g = T(2) * (s_g * s_g), const_g = T(dy) * (Num<T>::log(s_g) + half_log_2pi);
  const T two_var_q = T(2) * (s_q * s_q), const_q = T(dx) * (Num<T>::log(s_q) + half_log_2pi);
  const int64_t tiles = (N + TP - 1) / TP;
  V rp[(PREFETCH && !GATHER) ? NV : 1], rx[PREFETCH ? NV : 1];
  constexpr int MAXQ = (DP * (int)sizeof(T) + PB - 1) / PB;      // pieces that hold a row of this extent class
  uint32_t rg[GATHER ? PPL * MAXQ * (PB / 4) : 1];
  int64_t ranc[GATHER ? PPL : 1];
  const char *xprev_bytes = reinterpret_cast<const char *>(xprev);
  if constexpr (GATHER) {
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
    lg_anc_prefetch<PPL, true>(gat, n0, np, ranc);
    lg_gather_prefetch<PPL, MAXQ, PB, true>(xprev_bytes, gat, n0, np, K, ranc, rg);
    lg_prefetch<T, NV>(x + n0 * dx, np * dx, 0, rx);
    const int64_t m0 = n0 + (int64_t)gridDim.x * TP;
    if (m0 < N) lg_anc_prefetch<PPL, true>(gat, m0, (uint32_t)min((int64_t)TP, N - m0), ranc);
  } else if constexpr (PREFETCH) {
    const int64_t n0 = (int64_t)blockIdx.x * TP;
    const uint32_t ne = (uint32_t)min((int64_t)TP, N - n0) * dx;
    lg_prefetch<T, NV>(xprev + n0 * dx, ne, 0, rp);
    lg_prefetch<T, NV>(x + n0 * dx, ne, 0, rx);
  }
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TP;
    const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
    if constexpr (GATHER) {
      lg_gather_commit<T, PPL, MAXQ, PB, true>(gat, np, rg, tprev, lx);
      lg_commit<T, NV>(x + n0 * dx, np * dx, rx, tx, lx);
    } else if constexpr (PREFETCH) {
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
    if constexpr (GATHER) {
      const int64_t next = tile + gridDim.x;
      if (next < tiles) {
        const int64_t m0 = next * TP;
        const uint32_t mp_ = (uint32_t)min((int64_t)TP, N - m0);
        lg_gather_prefetch<PPL, MAXQ, PB, true>(xprev_bytes, gat, m0, mp_, K, ranc, rg);      // its indices came a tile ago
        lg_prefetch<T, NV>(x + m0 * dx, mp_ * dx, 0, rx);
        const int64_t nn0 = (next + gridDim.x) * TP;
        if (nn0 < N) lg_anc_prefetch<PPL, true>(gat, nn0, (uint32_t)min((int64_t)TP, N - nn0), ranc);
      }
    } else if constexpr (PREFETCH) {
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
          const T a = wpq[(i * DP + j) * 2], q = wpq[(i * DP + j) * 2 + 1];
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

template <typename T>
static int launch_particle_affine(const void *x1, const aesmc_affine_map *m1, const void *x2,
                                  const aesmc_affine_map *m2, const void *base, void *out, int64_t B, int64_t K,
                                  hipStream_t stream, bool through_tanh = false) {
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
  const int hint = stream_hint((uint64_t)N * (d1 + d2 + dout) * sizeof(T)) | (through_tanh ? 2 : 0);
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
template <typename T, int DP, int PPL>
static constexpr auto affine_propagate_gather4 = &affine_logweight_kernel<T, DP, PPL, true, true, true, 4>;
template <typename T, int DP, int PPL>
static constexpr auto affine_propagate_gather8 = &affine_logweight_kernel<T, DP, PPL, true, true, true, 8>;
template <typename T, int DP, int PPL>
static constexpr auto affine_propagate_gather16 = &affine_logweight_kernel<T, DP, PPL, true, true, true, 16>;

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
                                   hipStream_t stream, void *out_x = nullptr, const int64_t *anc_idx = nullptr,
                                   int32_t *flags = nullptr) {
  // out_x != nullptr: `x` is the proposal's noise and the draw is formed here (K15)
  // anc_idx != nullptr (with out_x): `xprev` is the un-resampled latent, its rows fetched through the indices
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
  const LgGather gat = lg_gather(anc_idx, flags, (size_t)dx * sizeof(T));
  if (anc_idx != nullptr && (!tab || N > 0x7fffffffLL || out_x == nullptr)) return AESMC_ERR_UNSUPPORTED;
  // persistent workgroups with the next tile prefetched (B=1024 K=4096 d=10: 106 us against 145 us with one
  // tile per workgroup); AESMC_LG_PREFETCH=0 selects the latter: a measurement knob
  static const bool prefetch = [] { const char *v = measurement_knob("AESMC_LG_PREFETCH"); return v == nullptr || v[0] != '0'; }();
  const unsigned grid = prefetch ? lg_persistent_grid(tiles, lds) : (unsigned)tiles;
#define LG_LOGWEIGHT_ARGS                                                                                          \
  static_cast<const T *>(xprev), static_cast<const T *>(x), static_cast<const T *>(y), y_sb, lg_map(mp), lg_map(mg), \
      lg_map(mq), static_cast<const T *>(sp), static_cast<const T *>(sg), static_cast<const T *>(sq),              \
      static_cast<T *>(out), N, (uint32_t)K, static_cast<T *>(out_x), gat
  if (anc_idx != nullptr) {
    const dim3 pgrid(lg_persistent_grid(tiles, lds));
    if (gat.pb == 16) LG_DISPATCH(affine_propagate_gather16, T, dp, ppl, pgrid, lds, stream, LG_LOGWEIGHT_ARGS);
    else if (gat.pb == 8) LG_DISPATCH(affine_propagate_gather8, T, dp, ppl, pgrid, lds, stream, LG_LOGWEIGHT_ARGS);
    else LG_DISPATCH(affine_propagate_gather4, T, dp, ppl, pgrid, lds, stream, LG_LOGWEIGHT_ARGS);
  } else if (out_x != nullptr) {
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

}  // namespace aesmc

using namespace aesmc;

extern "C" int64_t aesmc_affine_max_dim(void) { return kLgMaxDim; }

static int particle_affine_entry(int dtype, const void *x1, const aesmc_affine_map *m1, const void *x2,
                                 const aesmc_affine_map *m2, const void *base, void *out, int64_t B, int64_t K,
                                 void *stream, bool through_tanh) {
  if (x1 == nullptr || out == nullptr || m1 == nullptr || m1->weight == nullptr || B < 0 || K < 0 ||
      (x2 != nullptr && (m2 == nullptr || m2->weight == nullptr)))
    return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  if (!aligned16(x1) || !aligned16(out) || (x2 != nullptr && !aligned16(x2)) || (base != nullptr && !aligned16(base)))
    return AESMC_ERR_INVALID_ARGUMENT;
  if (!lg_map_ok(m1) || (x2 != nullptr && (!lg_map_ok(m2) || m2->dout != m1->dout))) return AESMC_ERR_UNSUPPORTED;
  if (B == 0 || K == 0) return AESMC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AESMC_F32 ? launch_particle_affine<float>(x1, m1, x2, m2, base, out, B, K, s, through_tanh)
                            : launch_particle_affine<double>(x1, m1, x2, m2, base, out, B, K, s, through_tanh);
}

extern "C" int aesmc_particle_affine(int dtype, const void *x1, const aesmc_affine_map *m1, const void *x2,
                                     const aesmc_affine_map *m2, const void *base, void *out, int64_t B, int64_t K,
                                     void *stream) {
  return particle_affine_entry(dtype, x1, m1, x2, m2, base, out, B, K, stream, false);
}

// tanh(location): the nonlinear transition of BASELINE.json configs[3] (`tanh(A x_{t-1})`) without the element-wise
// launch behind K8 (a read and a write of [B,K,d] more)
extern "C" int aesmc_particle_affine_tanh(int dtype, const void *x1, const aesmc_affine_map *m1, const void *x2,
                                          const aesmc_affine_map *m2, const void *base, void *out, int64_t B, int64_t K,
                                          void *stream) {
  return particle_affine_entry(dtype, x1, m1, x2, m2, base, out, B, K, stream, true);
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

extern "C" int aesmc_affine_normal_propagate_resampled(
    int dtype, const void *x_src, const int64_t *ancestors, const void *eps, const void *y, int64_t y_stride_b,
    const aesmc_affine_map *transition, const aesmc_affine_map *emission, const aesmc_affine_map *proposal,
    const void *scale_p, const void *scale_g, const void *scale_q, void *out_x, void *out_lw, int32_t *flags,
    int64_t B, int64_t K, void *stream) {
  if (x_src == nullptr || ancestors == nullptr || eps == nullptr || y == nullptr || transition == nullptr ||
      emission == nullptr || proposal == nullptr || scale_p == nullptr || scale_g == nullptr || scale_q == nullptr ||
      out_lw == nullptr || out_x == nullptr || B < 0 || K < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  if (!aligned16(x_src) || !aligned16(eps) || !aligned16(out_x) || out_x == x_src || (((uintptr_t)ancestors) & 7u) != 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (!lg_map_ok(transition) || !lg_map_ok(emission) || !lg_map_ok(proposal)) return AESMC_ERR_UNSUPPORTED;
  const int64_t dx = transition->dout;
  if (transition->din != dx || proposal->dout != dx || proposal->din != dx || emission->din != dx)
    return AESMC_ERR_UNSUPPORTED;
  if (B == 0 || K == 0) return AESMC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AESMC_F32
             ? launch_affine_logweight<float>(x_src, eps, y, y_stride_b, transition, emission, proposal, scale_p,
                                              scale_g, scale_q, out_lw, B, K, s, out_x, ancestors, flags)
             : launch_affine_logweight<double>(x_src, eps, y, y_stride_b, transition, emission, proposal, scale_p,
                                               scale_g, scale_q, out_lw, B, K, s, out_x, ancestors, flags);
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
