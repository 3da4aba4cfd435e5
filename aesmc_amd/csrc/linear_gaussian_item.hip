// K16, third form ("item form"): one SMC step's propagation — resampling gather, the proposal's draw with its noise
// formed in the launch, the three log-densities (aesmc/inference.py:102-126, state.py:98, :179 for a linear-Gaussian
// model) — with ONE work item per workgroup and no roles.
//
//   x_t[b,k,:] = loc_q(x_{t-1}[b, anc[b,k], :]) + s_q * eps[b,k,:]
//   lw[b,k]    = log N(x_t; A x + a, s_p) + log N(y_b; C x_t + g, s_g) - log N(x_t; loc_q, s_q)
//
// The persistent form (linear_gaussian_fused.hip) keeps 512 workgroups resident, each walking items blockIdx.x,
// blockIdx.x + grid, ... with four wavefronts drawing the NEXT item's noise while four propagate this one.  That
// pipeline needs many items per workgroup to pay: a launch of 1 239 items (one GPU's shard of the north-star batch at 8
// GPUs, B = 128) is a serial first draw plus three lock-step rounds, the last one 42 % full.  Here an item is a
// workgroup: all eight wavefronts draw its normals (a lane two or three Philox calls instead of five), one barrier,
// then wavefront w propagates chunk (w & 1) of window (w >> 1) — 64 particles, lane = particle, the maps as fma chains
// with scalar-register weights exactly as in the persistent form's scalar-weight branch.  The ancestor index is sent
// for first and the row of x_{t-1} behind the first draw, so both memory round trips fly under the draws; the hardware
// puts the next workgroup on a CU the moment one ends, so nothing runs in rounds.  A wavefront's rows of x_t are staged
// for their contiguous store in the part of the noise tile it alone read.
//
// Same item geometry (ATen's Philox launch: linear_gaussian_fused.hpp), same chains, same order: every output bit equals
// the persistent form's (tests/test_gpu_propagation_forms.py) and x_t the C oracle's.
#include "linear_gaussian_fused.hpp"

namespace aesmc {

// DXC: the latent's extent (compile time).  DYC: the observation's, or 0 = run time (<= 16).  PAIRED: the maps' `w` point
// at their interleaved copies (aesmc_affine_weight_pairs) and the chains of two outputs advance together, one
// v_pk_fma_f32 per input (linear_gaussian_fused.hpp: the same bits as the v_fmac_f32 chains).
template <int DXC, int DYC, bool GATHER, bool PAIRED>
__global__ __launch_bounds__(512, 4) void affine_propagate_item_kernel(
    const float *__restrict__ xsrc, const float *__restrict__ y, int64_t y_sb, LgMap mp, LgMap mg, LgMap mq,
    const float *__restrict__ sp_ptr, const float *__restrict__ sg_ptr, const float *__restrict__ sq_ptr,
    float *__restrict__ out_lw, uint32_t K, uint32_t Bn, float *__restrict__ out_x,
    const int64_t *__restrict__ anc_idx, int32_t *flags, PhiloxStream ps_in, FusedPlan plan) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr uint32_t dx = DXC;
  constexpr uint32_t kRunStride = kRunP * (uint32_t)DXC + 4u;      // floats between two windows' areas of the noise tile
  constexpr int DPX = 4 * ((DXC + 3) / 4);
  constexpr int DPY = DYC != 0 ? 4 * ((DYC + 3) / 4) : 16;
  const uint32_t dy = DYC != 0 ? (uint32_t)DYC : (uint32_t)mg.dout;
  extern __shared__ __attribute__((aligned(16))) unsigned char item_smem[];
  float *tab = reinterpret_cast<float *>(item_smem);            // [kTabF]: [window][batch row 0 / 1][p, q, g, y][16]
  float *noise = tab + kTabF;                                   // [4][kRunP * dx + 4]: window i's run at i * kRunStride, rows end to end
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint32_t wi = w >> 1, c = w & 1u;                       // this wavefront's window and chunk
  const uint32_t G = plan.G;
  const uint32_t item = blockIdx.x;

  // ---- the item's four windows (wavefront-uniform) --------------------------------------------------------------
  uint32_t head[4], limit[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const FusedWin v = fused_window(plan, item, (uint32_t)i, dx, K);
    head[i] = v.head;
    limit[i] = v.count * dx;
  }
  const FusedWin cur = fused_window(plan, item, wi, dx, K);
  const uint32_t rl = 64u * c + lane;                                           // the lane's row of the window
  const uint32_t rr = min(rl, cur.count != 0 ? cur.count - 1 : 0u);           // ... clamped: idle lanes repeat the last one
  const bool live = rl < cur.count;

  // ---- the ancestor first: its round trip, and the row's behind it, fly under the draws ---------------------------
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 araw;
  if constexpr (GATHER) {
    araw = __builtin_bit_cast(u32x2, anc_idx[cur.nf + rr]);
  } else {
    const uint32_t k = cur.k0 + rr;
    araw = u32x2{k >= K ? k - K : k, 0u};
  }

  // ---- this wavefront's block of the table: window wi, batch row c; lane = (vector, element) ----------------------
  float held = 0.0f;
  {
    const LgRowVec<float> vec[4] = {lg_offset_vec<float>(mp), lg_offset_vec<float>(mq), lg_offset_vec<float>(mg),
                                    {y, y_sb, (int)dy}};
    const uint32_t tab_a = lane >> 4, tab_j = lane & 15u;
    const float *tab_src = vec[0].ptr;
    int64_t tab_sb = vec[0].sb;
    int tab_len = vec[0].ptr != nullptr ? vec[0].len : 0;
#pragma unroll
    for (int a = 1; a < 4; ++a) {
      const bool mine = tab_a == (uint32_t)a;
      tab_src = mine ? vec[a].ptr : tab_src;
      tab_sb = mine ? vec[a].sb : tab_sb;
      tab_len = mine ? (vec[a].ptr != nullptr ? vec[a].len : 0) : tab_len;
    }
    // (no branch around the load — the compiler would wait for every load in flight where the branch ends —: an entry
    //  the table does not have reads a word that is always there, and is then replaced by zero)
    const uint32_t b = cur.b0 + c;
    const bool present = cur.count != 0 && b < Bn && (int)tab_j < tab_len;
    const float *at = present ? tab_src + ((int64_t)b * tab_sb + tab_j) : y;
    const float value = *at;
    held = present ? value : 0.0f;
  }

  // (PAIRED: the densities' constants behind the pairs — eight scalar registers sent for here, read behind the barrier)
  typedef float item_f8 __attribute__((ext_vector_type(8)));
  item_f8 consts = {};
  if constexpr (PAIRED)
    consts = *reinterpret_cast<const item_f8 __attribute__((address_space(4))) *>((fused_cfloat *)(unsigned long long)mp.w + 3 * kPairFloats);

  // ---- the item's normals: element G (4 trip + i) + t0 + j is output i of thread t0 + j ------------------------------
  const PhiloxStream ps = philox_resolve(ps_in);
  const uint32_t t0 = cur.t0, trip = cur.c, span = cur.tl + dx - 1;
  auto place = [&](uint32_t j, const float4 &first, const float4 &second, bool wraps) {
    float n4[4] = {first.x, first.y, first.z, first.w};
    if (wraps) {
      n4[0] = first.y; n4[1] = first.z; n4[2] = first.w; n4[3] = second.x;
    }
    // rows of a window lie end to end: element v of its run is at v — one unsigned minimum places a normal (j < head wraps
    // around to a huge v; thread ids past the block's span fall behind every limit): a normal that belongs to nobody goes
    // to the word behind the window's run, which nobody reads (a full window's is the four spare words between the
    // windows' areas) — no branch around the store, no comparison and select in front of it
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t v = j - head[i];
      noise[(uint32_t)i * kRunStride + min(v, limit[i])] = n4[i];
    }
  };

  // the row of x_{t-1} (sent for between the first draw and the rest)
  uint32_t bad = 0;
  float xin[DXC];
  auto rows_load = [&]() {
    uint32_t a = araw[0];
    // K2 writes K for a degenerate row (flagged there); never fault on it
    bad |= (a >= K ? 1u : 0u) | araw[1];
    a = a < K ? a : ((int32_t)araw[1] < 0 ? 0u : K - 1);
    const uint32_t Kdx = K * dx;
    const uint32_t base = cur.b0 * Kdx + ((cur.k0 + rr) >= K ? Kdx : 0u);      // the batch row's first element
    const float *at = xsrc + (base + __umul24(a, dx));
    constexpr int PW = DXC % 4 == 0 ? 4 : (DXC % 2 == 0 ? 2 : 1);               // naturally aligned pieces of a row
#pragma unroll
    for (int e0 = 0; e0 < DXC; e0 += PW) {
      if constexpr (PW == 4) {
        const fz4 piece = *reinterpret_cast<const fz4 *>(at + e0);
#pragma unroll
        for (int e = 0; e < 4; ++e) xin[e0 + e] = piece[e];
      } else if constexpr (PW == 2) {
        const float2 piece = *reinterpret_cast<const float2 *>(at + e0);
        xin[e0] = piece.x;
        xin[e0 + 1] = piece.y;
      } else {
        xin[e0] = at[e0];
      }
    }
  };

  constexpr uint32_t SC = ((kRunP + 1u) * (uint32_t)DXC - 1u) / 256u;           // == plan.S: 256 SC thread ids cover a block
  constexpr uint32_t IT = (SC * 256u + 511u) / 512u;                             // Philox calls of a lane
  // The lane's first call, then the row of x_{t-1} (the ancestor has had that call's time to arrive; the row has the
  // other calls').  Past a trip's last thread id (its last block only, at most dx - 1 lanes) element G (4c + i) + t is
  // thread t - G's output i + 1 of this trip, or its output 0 of the next.
  const uint32_t t_first = t0 + tid;
  const bool wraps_first = t_first >= G;
  const uint32_t tt_first = wraps_first ? t_first - G : t_first;
  const float4 first = philox_normal4(ps, tt_first, trip);
  __builtin_amdgcn_sched_barrier(0);
  rows_load();
  __builtin_amdgcn_sched_barrier(0);
  if (t0 + span <= G) {
    // no thread id wraps (all but a trip's last block; then span == 256 SC): the other calls written out as one straight
    // line — independent dependency chains for the vector ALU to interleave
    float4 drawn[IT];
#pragma unroll
    for (uint32_t s = 1; s < IT; ++s)
      if (s * 512u + 64u * w < SC * 256u) drawn[s] = philox_normal4(ps, t_first + s * 512u, trip);      // (uniform)
    place(tid, first, first, false);
#pragma unroll
    for (uint32_t s = 1; s < IT; ++s)
      if (s * 512u + 64u * w < SC * 256u) place(tid + s * 512u, drawn[s], drawn[s], false);
  } else {
    float4 second = first;
    if (__any(wraps_first)) second = philox_normal4(ps, tt_first, trip + 1);      // (uniform branch, rarely taken)
    place(tid, first, second, wraps_first);      // (thread ids past the block's span fall behind every limit)
#pragma unroll 1
    for (uint32_t j = tid + 512u; j < span; j += 512u) {
      const uint32_t t = t0 + j;
      const bool wraps = t >= G;
      const uint32_t tt = wraps ? t - G : t;
      const float4 a = philox_normal4(ps, tt, trip);
      float4 b = a;
      if (__any(wraps)) b = philox_normal4(ps, tt, trip + 1);
      place(j, a, b, wraps);
    }
  }
  tab[64u * w + lane] = held;
  lg_lds_barrier();

  // ---- propagate: lane = particle; the maps' weights in scalar registers --------------------------------------------
  auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
  const float s_q = sq_ptr[0];
  float *mine = noise + wi * kRunStride + 64u * c * dx;                         // this chunk's 64 rows of noise; later of x_t
  const float *trow = tab + wi * (2u * 4u * 16u) + ((cur.k0 + rr) >= K ? 64u : 0u);      // the lane's batch row's vectors
  const unsigned long long wq_a = (unsigned long long)mq.w, wp_a = (unsigned long long)mp.w, wg_a = (unsigned long long)mg.w;
  // the densities' constants: behind the pairs when the launch that interleaved them had the scales (scalar loads; the tag
  // names the extents they were formed for), else from the scales here — the same expressions, the same bits
  float two_var_p, const_p, two_var_g, const_g, two_var_q, const_q;
  bool at_hand = false;
  if constexpr (PAIRED) {
    at_hand = __float_as_uint(consts[6]) == fused_consts_tag(dx, dy);
    if (at_hand) {
      two_var_p = consts[0]; const_p = consts[1];
      two_var_g = consts[2]; const_g = consts[3];
      two_var_q = consts[4]; const_q = consts[5];
    }
  }
  if (!at_hand) {
    const float s_p = sp_ptr[0], s_g = sg_ptr[0];
    const float half_log_2pi = LgConst<float>::half_log_2pi();
    two_var_p = uniform(2.0f * (s_p * s_p)); const_p = uniform((float)dx * (Num<float>::log(s_p) + half_log_2pi));
    two_var_g = uniform(2.0f * (s_g * s_g)); const_g = uniform((float)dy * (Num<float>::log(s_g) + half_log_2pi));
    two_var_q = uniform(2.0f * (s_q * s_q)); const_q = uniform((float)dx * (Num<float>::log(s_q) + half_log_2pi));
  }

  float locq[DPX], locp[DPX];
  if constexpr (PAIRED) {
    lg_f2 xin2[(DXC + 1) / 2], q2[DPX / 2], p2[DPX / 2];
#pragma unroll
    for (int i = 0; i < (DXC + 1) / 2; ++i) xin2[i] = lg_f2{xin[2 * i], 2 * i + 1 < DXC ? xin[(2 * i + 1) % DXC] : 0.0f};
    fused_chain_pk<DXC, DPX>(wq_a, trow + 16, dx, xin2, q2);
    fused_chain_pk<DXC, DPX>(wp_a, trow, dx, xin2, p2);
#pragma unroll
    for (int j = 0; j < DPX; ++j) {
      locq[j] = q2[j / 2][j % 2];
      locp[j] = p2[j / 2][j % 2];
    }
  } else {
    fused_chain<DXC, DPX>(wq_a, trow + 16, dx, xin, locq);
    fused_chain<DXC, DPX>(wp_a, trow, dx, xin, locp);
  }
  // the lane's noise (its row of the chunk: lanes past the window's end read the last row's, or — a chunk past the
  // window's end — the other chunk's region, which its owner may already be overwriting: values nobody keeps)
  float nz[DXC];
  {
    // (whole 16- / 8-byte pieces where rows are: a wavefront's reads then fall on distinct LDS banks for every even
    //  extent but 8 and 16 — two- and four-way there —; odd extents read dwords, conflict-free)
    const float *nrow = noise + wi * kRunStride + rr * dx;
    constexpr int PW = DXC % 4 == 0 ? 4 : (DXC % 2 == 0 ? 2 : 1);
#pragma unroll
    for (int e0 = 0; e0 < DXC; e0 += PW) {
      if constexpr (PW == 4) {
        const fz4 piece = *reinterpret_cast<const fz4 *>(nrow + e0);
#pragma unroll
        for (int e = 0; e < 4; ++e) nz[e0 + e] = piece[e];
      } else if constexpr (PW == 2) {
        const float2 piece = *reinterpret_cast<const float2 *>(nrow + e0);
        nz[e0] = piece.x;
        nz[e0 + 1] = piece.y;
      } else {
        nz[e0] = nrow[e0];
      }
    }
  }
  float xx[DXC], qp = 0.0f, qq = 0.0f, qg = 0.0f;
#pragma unroll
  for (int j = 0; j < DXC; ++j) {
    xx[j] = locq[j] + nz[j] * s_q;      // the product rounded before the sum, as K9 / K6
    const float ep = xx[j] - locp[j], eq = xx[j] - locq[j];
    qp = fma_t(ep, ep, qp);
    qq = fma_t(eq, eq, qq);
  }
  float locg[DPY];
  if constexpr (PAIRED) {
    lg_f2 xx2[(DXC + 1) / 2], g2[DPY / 2];
#pragma unroll
    for (int i = 0; i < (DXC + 1) / 2; ++i) xx2[i] = lg_f2{xx[2 * i], 2 * i + 1 < DXC ? xx[(2 * i + 1) % DXC] : 0.0f};
    fused_chain_pk<DXC, DPY>(wg_a, trow + 32, dy, xx2, g2);
#pragma unroll
    for (int j = 0; j < DPY; ++j) locg[j] = g2[j / 2][j % 2];
  } else {
    fused_chain<DXC, DPY>(wg_a, trow + 32, dy, xx, locg);
  }
#pragma unroll
  for (int v = 0; v < DPY / 4; ++v) {
    const fz4 y4 = *reinterpret_cast<const fz4 *>(trow + 48 + 4 * v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if ((uint32_t)(4 * v + e) < dy) {
        const float eg = y4[e] - locg[4 * v + e];
        qg = fma_t(eg, eg, qg);
      }
    }
  }
  if (live) {
    const float lp = (-qp) / two_var_p - const_p;
    const float lg = (-qg) / two_var_g - const_g;
    const float lq = (-qq) / two_var_q - const_q;
    out_lw[cur.nf + rl] = (lp + lg) - lq;
  }
  // ---- the chunk's rows of x_t leave as one contiguous run, staged where the chunk's noise was (every lane of this
  //      wavefront has read its noise: a wavefront's LDS accesses execute in order) ------------------------------------
  {
    float *srow = mine + lane * dx;
    constexpr int PW = DXC % 4 == 0 ? 4 : (DXC % 2 == 0 ? 2 : 1);
#pragma unroll
    for (int e0 = 0; e0 < DXC; e0 += PW) {
      if constexpr (PW == 4) {
        *reinterpret_cast<fz4 *>(srow + e0) = fz4{xx[e0], xx[e0 + 1], xx[e0 + 2], xx[e0 + 3]};
      } else if constexpr (PW == 2) {
        *reinterpret_cast<float2 *>(srow + e0) = make_float2(xx[e0], xx[e0 + 1]);
      } else {
        srow[e0] = xx[e0];
      }
    }
    const uint32_t rows = cur.count > 64u * c ? min(cur.count - 64u * c, 64u) : 0u;
    const uint32_t words = rows * dx;
    float *run = out_x + (size_t)(cur.nf + 64u * c) * dx;
#pragma unroll
    for (int u = 0; u < (DXC + 3) / 4; ++u) {
      const uint32_t q = lane + 64u * u;
      if (4u * q + 4u <= words) *reinterpret_cast<fz4_a4 *>(run + 4u * q) = *reinterpret_cast<const fz4 *>(mine + 4u * q);
    }
    if ((words & 3u) != 0u && lane < (words & 3u)) {
      const uint32_t e = (words & ~3u) + lane;
      run[e] = mine[e];
    }
  }
  if (bad != 0u) raise_flag(flags, AESMC_FLAG_INDEX_OUT_OF_RANGE);
#endif
}

template <int DXC, int DYC, bool PAIRED>
static int item_launch(dim3 grid, size_t lds, hipStream_t stream, const float *xsrc, const float *y, int64_t y_sb,
                       const LgMap &mp, const LgMap &mg, const LgMap &mq, const float *sp, const float *sg, const float *sq,
                       float *out_lw, uint32_t K, uint32_t Bn, float *out_x, const int64_t *anc, int32_t *flags,
                       const PhiloxStream &ps, const FusedPlan &plan) {
  static bool raised[2][64] = {};
  if (anc != nullptr) {
    if (lds > 64 * 1024 &&
        !lg_raise_lds_limit(reinterpret_cast<const void *>(&affine_propagate_item_kernel<DXC, DYC, true, PAIRED>), raised[0]))
      return AESMC_ERR_LAUNCH;
    hipLaunchKernelGGL((affine_propagate_item_kernel<DXC, DYC, true, PAIRED>), grid, dim3(512), lds, stream, xsrc, y, y_sb, mp,
                       mg, mq, sp, sg, sq, out_lw, K, Bn, out_x, anc, flags, ps, plan);
  } else {
    if (lds > 64 * 1024 &&
        !lg_raise_lds_limit(reinterpret_cast<const void *>(&affine_propagate_item_kernel<DXC, DYC, false, PAIRED>), raised[1]))
      return AESMC_ERR_LAUNCH;
    hipLaunchKernelGGL((affine_propagate_item_kernel<DXC, DYC, false, PAIRED>), grid, dim3(512), lds, stream, xsrc, y, y_sb, mp,
                       mg, mq, sp, sg, sq, out_lw, K, Bn, out_x, anc, flags, ps, plan);
  }
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

// The maps' interleaved copies: pairs[jp][i] = (W[2 jp][i], W[2 jp + 1][i]) (zero where the second row does not exist), one
// region of kPairFloats floats per map in the order transition, emission, proposal.  Any strides: a transposed view too.
// `tail`: 0 nothing behind the regions is written; 1 the constants' tag is cleared; 2 the constants of the three
// densities and their tag (linear_gaussian_fused.hpp: kPairConsts) from the scales `sp`, `sg`, `sq`.
__global__ __launch_bounds__(256) void affine_weight_pairs_kernel(LgMap mp, LgMap mg, LgMap mq, float *__restrict__ out,
                                                                  const float *__restrict__ sp, const float *__restrict__ sg,
                                                                  const float *__restrict__ sq, int tail) {
  const LgMap &m = blockIdx.x == 0 ? mp : (blockIdx.x == 1 ? mg : mq);
  const float *w = reinterpret_cast<const float *>(m.w);
  float *dst = out + blockIdx.x * kPairFloats;
  const int din = m.din, dout = m.dout;
  for (int e = threadIdx.x; e < kPairFloats; e += 256) {
    const int jp = e / (2 * din), rem = e - jp * 2 * din, i = rem >> 1, j = 2 * jp + (rem & 1);
    dst[e] = (j < dout && i < din) ? w[(int64_t)j * m.sj + (int64_t)i * m.si] : 0.0f;
  }
  if (tail != 0 && blockIdx.x == 0 && threadIdx.x < (unsigned)kPairConsts) {
    const int t = threadIdx.x, which = t >> 1;
    const uint32_t dx = (uint32_t)mp.dout, dy = (uint32_t)mg.dout;
    float value = 0.0f;
    if (tail == 2 && t < 6) {
      // (the expressions of affine_propagate_item_kernel, in this translation unit: the same instructions)
      const float s = which == 0 ? sp[0] : (which == 1 ? sg[0] : sq[0]);
      const float extent = (float)(which == 1 ? dy : dx);
      value = (t & 1) == 0 ? 2.0f * (s * s) : extent * (Num<float>::log(s) + LgConst<float>::half_log_2pi());
    } else if (tail == 2 && t == 6) {
      value = __uint_as_float(fused_consts_tag(dx, dy));
    }
    out[3 * kPairFloats + t] = value;
  }
}

int launch_affine_weight_pairs(const LgMap &mp, const LgMap &mg, const LgMap &mq, float *out, hipStream_t stream,
                               const float *sp, const float *sg, const float *sq, bool tail) {
  const bool scaled = sp != nullptr && sg != nullptr && sq != nullptr;
  hipLaunchKernelGGL(affine_weight_pairs_kernel, dim3(3), dim3(256), 0, stream, mp, mg, mq, out, sp, sg, sq,
                     tail ? (scaled ? 2 : 1) : 0);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

// AESMC_ERR_UNSUPPORTED: the caller takes another form.  Covers latent extents 2 .. 16 with observation extents 1 .. 16;
// without `weight_pairs` the weights' rows must be contiguous ([dout, din] row-major, what an nn.Linear holds).
int launch_affine_propagate_item(const void *xsrc, const int64_t *anc_idx, const void *y, int64_t y_sb,
                                 const aesmc_affine_map *mp, const aesmc_affine_map *mg, const aesmc_affine_map *mq,
                                 const void *sp, const void *sg, const void *sq, void *out_x, void *out_lw, int32_t *flags,
                                 int64_t B, int64_t K, uint64_t seed, uint64_t offset, int64_t threads,
                                 const uint64_t *rng_state, const float *weight_pairs, hipStream_t stream) {
  const int64_t dx = mp->dout, dy = mg->dout;
  if (dx < 2 || dx > 16 || dy < 1 || dy > 16) return AESMC_ERR_UNSUPPORTED;
  const auto rows_contiguous = [](const aesmc_affine_map *m) {
    return m->stride_in == 1 && m->stride_out == m->din && (reinterpret_cast<uintptr_t>(m->weight) & 3u) == 0;
  };
  const bool paired = weight_pairs != nullptr;
  if (!paired && !(rows_contiguous(mp) && rows_contiguous(mg) && rows_contiguous(mq))) return AESMC_ERR_UNSUPPORTED;
  FusedPlan plan;
  const int planned = fused_make_plan(plan, B, K, dx, threads);
  if (planned != AESMC_OK) return planned;
  const size_t lds = sizeof(float) * ((size_t)kTabF + 4 * ((size_t)kRunP * (size_t)dx + 4));      // (four windows' areas)
  if (lds > kLgLdsLimit) return AESMC_ERR_UNSUPPORTED;
  const PhiloxStream ps = philox_stream(seed, offset, threads, rng_state);
  const dim3 grid(plan.items);
  LgMap p = lg_map(mp), gm = lg_map(mg), q = lg_map(mq);
  if (paired) {
    p.w = weight_pairs;
    gm.w = weight_pairs + kPairFloats;
    q.w = weight_pairs + 2 * kPairFloats;
  }
#define ITEM_ARGS                                                                                                    \
  grid, lds, stream, static_cast<const float *>(xsrc), static_cast<const float *>(y), y_sb, p, gm, q,                  \
      static_cast<const float *>(sp), static_cast<const float *>(sg), static_cast<const float *>(sq),                  \
      static_cast<float *>(out_lw), (uint32_t)K, (uint32_t)B, static_cast<float *>(out_x), anc_idx, flags, ps, plan
#define ITEM_CASE(D)                                                                                                 \
  case D:                                                                                                            \
    if (paired) return dy == D ? item_launch<D, D, true>(ITEM_ARGS) : item_launch<D, 0, true>(ITEM_ARGS);              \
    return dy == D ? item_launch<D, D, false>(ITEM_ARGS) : item_launch<D, 0, false>(ITEM_ARGS);
  switch (dx) {
#ifdef AESMC_LG_FAST_BUILD
    ITEM_CASE(10)
#else
    ITEM_CASE(2) ITEM_CASE(3) ITEM_CASE(4) ITEM_CASE(5) ITEM_CASE(6) ITEM_CASE(7) ITEM_CASE(8) ITEM_CASE(9) ITEM_CASE(10)
    ITEM_CASE(11) ITEM_CASE(12) ITEM_CASE(13) ITEM_CASE(14) ITEM_CASE(15) ITEM_CASE(16)
#endif
    default: return AESMC_ERR_UNSUPPORTED;
  }
#undef ITEM_CASE
#undef ITEM_ARGS
}

}  // namespace aesmc

using namespace aesmc;

extern "C" int64_t aesmc_affine_weight_pairs_floats(void) { return 3 * (int64_t)kPairFloats + kPairConsts; }

extern "C" int aesmc_affine_weight_pairs(const aesmc_affine_map *transition, const aesmc_affine_map *emission,
                                         const aesmc_affine_map *proposal, void *out_pairs, void *stream) {
  if (out_pairs == nullptr || (reinterpret_cast<uintptr_t>(out_pairs) & 15u) != 0) return AESMC_ERR_INVALID_ARGUMENT;
  if (!lg_map_ok(transition) || !lg_map_ok(emission) || !lg_map_ok(proposal)) return AESMC_ERR_UNSUPPORTED;
  return launch_affine_weight_pairs(lg_map(transition), lg_map(emission), lg_map(proposal), static_cast<float *>(out_pairs),
                                    static_cast<hipStream_t>(stream), nullptr, nullptr, nullptr, true);
}

extern "C" int aesmc_affine_weight_pairs_scaled(const aesmc_affine_map *transition, const aesmc_affine_map *emission,
                                                const aesmc_affine_map *proposal, const void *scale_p, const void *scale_g,
                                                const void *scale_q, void *out_pairs, void *stream) {
  if (out_pairs == nullptr || (reinterpret_cast<uintptr_t>(out_pairs) & 15u) != 0) return AESMC_ERR_INVALID_ARGUMENT;
  if (scale_p == nullptr || scale_g == nullptr || scale_q == nullptr) return AESMC_ERR_INVALID_ARGUMENT;
  if (!lg_map_ok(transition) || !lg_map_ok(emission) || !lg_map_ok(proposal)) return AESMC_ERR_UNSUPPORTED;
  return launch_affine_weight_pairs(lg_map(transition), lg_map(emission), lg_map(proposal), static_cast<float *>(out_pairs),
                                    static_cast<hipStream_t>(stream), static_cast<const float *>(scale_p),
                                    static_cast<const float *>(scale_g), static_cast<const float *>(scale_q), true);
}
