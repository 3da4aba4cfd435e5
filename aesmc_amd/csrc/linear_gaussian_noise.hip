// K16: one SMC step's propagation with NOTHING but the state read and written:
//   x_t[b,k,:] = loc_q(x_{t-1}[b, anc[b,k], :]) + s_q * eps[b,k,:]        eps drawn INSIDE the launch
//   lw[b,k]    = log N(x_t; A x + a, s_p) + log N(y_b; C x_t + g, s_g) - log N(x_t; loc_q, s_q)
// — aesmc/inference.py:102-126 for a linear-Gaussian model: the resampling gather (state.py:179), the
// proposal's `rsample` (state.py:98: `_standard_normal` + loc + eps * scale) and the three log-densities, from one
// read of the surviving rows of x_{t-1} and of the ancestor indices, one write of x_t and of the log-weights.
// Per particle 8 + 4d (in) + 4d + 4 (out) bytes instead of K3's 8d + 8, `normal_`'s 4d and K15's 12d + 4.
//
// The noise is PyTorch's own stream (philox_normal.hpp): element e of the tensor `torch.empty([B,K,d]).normal_()`
// would hold.  ATen hands the four normals of one Philox call to elements G apart (e = t + G (4c + i)), so a
// work item here is a block of `tl` thread ids t in one trip c: its four WINDOWS of `tl` consecutive elements, G
// apart, are four runs of consecutive particles (those whose first element lies in the window; d - 1 more
// thread ids are drawn past the block's end for the last particle's tail), 4 x RUNP tile rows.  A lane draws S
// Philox calls, scatters the 4 S normals into the noise tile by (window, row, column) and, after one barrier,
// reads back the rows of ITS particles.  Nothing else crosses lanes: x_{t-1}'s rows are fetched by the lane
// that owns the particle through the particle's ancestor (prefetched one item ahead, the index two ahead),
// and x_t leaves from registers.
#include <type_traits>

#include "linear_gaussian.hpp"
#include "philox_normal.hpp"

namespace aesmc {

struct LgNoisePlan {
  uint64_t numel;       // B K d
  uint64_t magic;       // ceil(2^40 / d): (v * magic) >> 40 == v / d for v < 2^32
  uint32_t L;           // thread ids per block (elements per window): 256 S - (d - 1)
  uint32_t S;           // Philox calls per lane and item
  uint32_t blocks;      // blocks per trip: ceil(G / L)
  uint32_t trips;       // ceil(numel / (4 G))
  uint32_t small_magic; // ceil(2^20 / d): (v * small_magic) >> 20 == v / d for v < 2^15
};

__device__ __forceinline__ uint64_t lg_div_d(uint64_t v, uint64_t magic) {   // v < 2^32
  return (uint64_t)(((unsigned __int128)v * magic) >> 40);
}

constexpr int kLgSegRows = 2;     // batch rows a window's run of particles can span (K >= RUNP)
constexpr int kNoiseThreads = 2 * kLgBlock;

// Workgroups of 512 lanes in two roles.  Wavefronts 0-3 ("particles": lane = tile slot tid + 256 r) fetch
// x_{t-1}'s rows and do the step's arithmetic; wavefronts 4-7 ("noise": pure vector arithmetic, never waiting for
// memory) draw the NEXT item's normals and stage its per-batch-row vectors, into the other half of a
// double-buffered LDS tile.  One barrier per item hands the buffers over.  Every SIMD then holds, per resident
// workgroup, one wavefront that always has arithmetic to issue beside one that mostly waits for HBM.
template <typename T, int DP, int PPL, int PB>
__global__ __launch_bounds__(kNoiseThreads, 4) void affine_propagate_noise_kernel(
    const T *__restrict__ xsrc, const T *__restrict__ y, int64_t y_sb, LgMap mp, LgMap mg, LgMap mq,
    const T *__restrict__ sp_ptr, const T *__restrict__ sg_ptr, const T *__restrict__ sq_ptr, T *__restrict__ out_lw,
    int64_t N, uint32_t K, uint32_t Bn, T *__restrict__ out_x, LgGather gat, PhiloxStream ps_in, LgNoisePlan plan) {
  static_assert(sizeof(T) == 4, "the in-kernel noise is torch's float32 stream");
  const PhiloxStream ps = philox_resolve(ps_in);
  extern __shared__ __attribute__((aligned(16))) unsigned char lg_smem[];
  constexpr uint32_t TP = kLgBlock * PPL, RUNP = TP / 4;
  constexpr int MAXQ = (DP * (int)sizeof(T) + PB - 1) / PB;
  constexpr int W = PB / 4;
  constexpr uint32_t kTab = 4 * kLgSegRows * 4 * DP;
  using P = typename LgPiece<PB>::type;
  const uint32_t dx = mp.dout, dy = mg.dout;
  const LgLayout lx = lg_layout<T>(dx);
  const uint32_t tile_elems = TP * lx.rs + 16;
  T *wp = reinterpret_cast<T *>(lg_smem);
  T *wg = wp + DP * DP;
  T *wq = wg + DP * DP;
  T *tabs = wq + DP * DP;                            // [2][4][kLgSegRows][4][DP]: offsets p, q, g and the observation
  T *tprev = tabs + 2 * kTab;
  T *noise = tprev + tile_elems;                     // [2] tiles
  const uint32_t tid = threadIdx.x & (kLgBlock - 1);
  const bool draws = threadIdx.x >= kLgBlock;        // wavefront-uniform role
  const uint64_t G = ps.threads;
  const uint32_t items = plan.trips * plan.blocks;

  // where the four windows of an item lie: first particle, particle count, head (elements of the window that
  // belong to a particle owned elsewhere)
  struct Item {
    uint32_t nf[4], count[4], head[4], t0, tl, c;
  };
  auto locate = [&](uint32_t item) {
    Item it;
    it.c = item / plan.blocks;
    const uint32_t tb = item - it.c * plan.blocks;
    it.t0 = tb * plan.L;
    it.tl = (uint32_t)min((uint64_t)plan.L, G - it.t0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint64_t lo = G * (4ull * it.c + i) + it.t0;
      if (lo >= plan.numel) {
        it.nf[i] = 0; it.count[i] = 0; it.head[i] = 0;
      } else {
        const uint64_t hi = min(lo + it.tl, plan.numel);
        const uint64_t nf = lg_div_d(lo + dx - 1, plan.magic), nl = lg_div_d(hi + dx - 1, plan.magic);
        it.count[i] = (uint32_t)(nl - nf);
        it.nf[i] = nl != nf ? (uint32_t)nf : 0u;       // (a window that owns no particle points at a valid one)
        it.head[i] = (uint32_t)(nf * dx - lo);
      }
    }
    return it;
  };
  auto pick = [](const uint32_t (&v)[4], uint32_t s) { return s == 0 ? v[0] : s == 1 ? v[1] : s == 2 ? v[2] : v[3]; };

  if (!draws) {
#pragma unroll 1
    for (uint32_t e = tid; e < 3 * DP * DP; e += kLgBlock) {      // the three maps, zero-padded and transposed
      const uint32_t m = e / (DP * DP), rest = e - m * (DP * DP);
      const int i = rest / DP, j = rest - i * DP;
      const LgMap &map = m == 0 ? mp : m == 1 ? mg : mq;
      const T *w = reinterpret_cast<const T *>(map.w);
      wp[e] = (j < map.dout && i < map.din) ? w[(int64_t)j * map.sj + (int64_t)i * map.si] : T(0);
    }
  }

  // ---- the noise role: item -> noise tile + row table, one item ahead of the particles ------------------------
  auto draw_item = [&](uint32_t item, uint32_t slot) {
    const Item it = locate(item);
    T *tx = noise + slot * tile_elems;
    T *tab = tabs + slot * kTab;
    const LgRowVec<T> vec[4] = {lg_offset_vec<T>(mp), lg_offset_vec<T>(mq), lg_offset_vec<T>(mg), {y, y_sb, (int)dy}};
    uint32_t b0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b0[i] = it.nf[i] / K;
    // (the table's values are loaded first and written after the draws: the loads fly during the arithmetic)
    constexpr int kTabTrips = (kTab + kLgBlock - 1) / kLgBlock;
    T held[kTabTrips];
#pragma unroll
    for (int trip = 0; trip < kTabTrips; ++trip) {
      const uint32_t idx = tid + trip * kLgBlock;
      const uint32_t j = idx % DP, a = (idx / DP) % 4, rel = (idx / (4 * DP)) % kLgSegRows, s = idx / (4 * DP * kLgSegRows);
      const uint32_t b = pick(b0, s) + rel;
      T value = T(0);
      if (idx < kTab && pick(it.count, s) != 0 && b < Bn) {
#pragma unroll
        for (int cidx = 0; cidx < 4; ++cidx)
          if (a == (uint32_t)cidx && vec[cidx].ptr != nullptr && (int)j < vec[cidx].len)
            value = vec[cidx].ptr[(int64_t)b * vec[cidx].sb + j];
      }
      held[trip] = value;
    }
    const bool flat = lx.rs == dx;      // workgroup-uniform
    uint32_t limit[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) limit[i] = it.count[i] * dx;
#pragma unroll 1
    for (uint32_t s = 0; s < plan.S; ++s) {
      const uint32_t j = tid + s * kLgBlock;
      if (j >= it.tl + dx - 1) break;
      const uint64_t t = (uint64_t)it.t0 + j;
      float n4[4];
      if (t < G) {
        const float4 n = philox_normal4(ps, (uint32_t)t, it.c);
        n4[0] = n.x; n4[1] = n.y; n4[2] = n.z; n4[3] = n.w;
      } else {      // past the last thread id: the elements belong to the next window's first thread ids
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint64_t e = G * (4ull * it.c + i) + t;
          n4[i] = e < plan.numel ? philox_normal_element(ps, e) : 0.0f;
        }
      }
      if (flat) {
        // rows of the tile lie end to end (rs == d): element v of a window's run is at v — one unsigned comparison
        // places a normal (j < head wraps around to a huge v), instead of a division into (row, column)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t v = j - it.head[i];
          if (v < limit[i]) tx[i * RUNP * dx + v] = n4[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (j >= it.head[i]) {
            const uint32_t v = j - it.head[i];
            const uint32_t rr = (v * plan.small_magic) >> 20, col = v - rr * dx;
            if (rr < it.count[i]) tx[(i * RUNP + rr) * lx.rs + col] = n4[i];
          }
        }
      }
    }
#pragma unroll
    for (int trip = 0; trip < kTabTrips; ++trip)
      if (tid + trip * kLgBlock < kTab) tab[tid + trip * kLgBlock] = held[trip];
  };

  if (draws) {
    if (blockIdx.x < items) draw_item(blockIdx.x, 0);
    lg_lds_barrier();
    uint32_t slot = 0;
    for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
      const uint32_t next = item + gridDim.x;
      if (next < items) draw_item(next, slot ^ 1u);
      lg_lds_barrier();
      slot ^= 1u;
    }
    return;
  }

  // ---- the particle role ---------------------------------------------------------------------------------------
  const T s_p = sp_ptr[0], s_g = sg_ptr[0], s_q = sq_ptr[0];
  const T half_log_2pi = LgConst<T>::half_log_2pi();
  const T two_var_p = T(2) * (s_p * s_p), const_p = T(dx) * (Num<T>::log(s_p) + half_log_2pi);
  const T two_var_g = T(2) * (s_g * s_g), const_g = T(dy) * (Num<T>::log(s_g) + half_log_2pi);
  const T two_var_q = T(2) * (s_q * s_q), const_q = T(dx) * (Num<T>::log(s_q) + half_log_2pi);
  const char *src_bytes = reinterpret_cast<const char *>(xsrc);
  char *out_bytes = reinterpret_cast<char *>(out_x);
  // a lane's tile slots: slot = tid + 256 r  ->  window slot / RUNP, row slot % RUNP
  uint32_t seg[PPL], row[PPL];
#pragma unroll
  for (int r = 0; r < PPL; ++r) {
    const uint32_t slot = tid + r * kLgBlock;
    // RUNP is a multiple of the wavefront: a wavefront's lanes share their window, so everything picked by it
    // (first particle, count, batch row) is uniform and lives on the scalar unit
    static_assert(RUNP % kWave == 0, "a window's rows are whole wavefronts");
    seg[r] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(slot / RUNP));
    row[r] = slot - seg[r] * RUNP;
  }
  // (generic over "are there ancestor indices": as a runtime branch around the index loads it made the compiler
  // wait for each of them on the spot)
  auto particles = [&](auto has_idx) {
  uint32_t rg[PPL * MAXQ * W];
  int64_t ranc[PPL];
  // Idle lanes (a window holds fewer particles than it has rows) DUPLICATE the window's last particle: same
  // loads, same arithmetic, same stores of the same values to the same addresses.  No lane-dependent branch
  // surrounds a load or a store then, so the loads stay in flight across the arithmetic instead of being waited
  // for at the end of an `if` (measured: four exposed memory round trips per item with the branches in place).
  // (A window without any particle — past the tensor's end — is wavefront-uniform: those wavefronts skip.)
  auto row_of = [&](const Item &it, int r) {
    const uint32_t count = pick(it.count, seg[r]);
    return min(row[r], count != 0 ? count - 1 : 0u);
  };
  // the lane's ancestors of `it`'s particles -> registers
  auto anc_prefetch = [&](const Item &it) {
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const uint32_t first = pick(it.nf, seg[r]), rr = row_of(it, r);
      if constexpr (decltype(has_idx)::value) {
        ranc[r] = gat.idx[(int64_t)first + rr];       // (an empty window reads particle `first` = 0: valid, unused)
      } else {      // no resampling in front of this step: a particle is its own ancestor
        const uint32_t k0 = first % K;
        ranc[r] = (k0 + rr) >= K ? (int64_t)(k0 + rr - K) : (int64_t)(k0 + rr);
      }
    }
  };
  // the rows of x_{t-1} they point at -> registers
  auto rows_prefetch = [&](const Item &it) {
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const uint32_t first = pick(it.nf, seg[r]), rr = row_of(it, r);      // uniform per window: scalar divisions
      const uint32_t b0 = first / K, k0 = first - b0 * K;
      int64_t a = ranc[r];
      if (a < 0 || a >= (int64_t)K) {
        raise_flag(gat.flags, AESMC_FLAG_INDEX_OUT_OF_RANGE);
        a = a < 0 ? 0 : (int64_t)K - 1;
      }
      const uint64_t source = (uint64_t)(b0 + ((k0 + rr) >= K ? 1u : 0u)) * K + (uint64_t)a;
      const char *at = src_bytes + source * gat.row_bytes;
#pragma unroll
      for (int c = 0; c < MAXQ; ++c) {
        // (pieces past the row's end — an extent below the kernel's class — re-read its last piece: no branch)
        const P x = *reinterpret_cast<const P *>(at + min((uint32_t)c, gat.ppr - 1) * PB);
        __builtin_memcpy(&rg[(r * MAXQ + c) * W], &x, PB);
      }
    }
  };

  Item cur = locate(blockIdx.x < items ? blockIdx.x : 0);
  Item ahead = cur;       // the item after `cur` (located once: its ancestors are sent for an item before its rows are)
  if (blockIdx.x < items) {
    anc_prefetch(cur);
    rows_prefetch(cur);
    if (blockIdx.x + gridDim.x < items) {
      ahead = locate(blockIdx.x + gridDim.x);
      anc_prefetch(ahead);
    }
  }
  lg_lds_barrier();       // the first item's noise and table are there; so are the maps
  uint32_t slot = 0;
  for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
    const T *tx = noise + slot * tile_elems;
    const T *tab = tabs + slot * kTab;
    // ---- park the rows fetched for this item (own slots: no other lane reads them); send for the next item's
    {
      char *base = reinterpret_cast<char *>(tprev);
      const uint32_t row_pitch = lx.rs * (uint32_t)sizeof(T);
#pragma unroll
      for (int r = 0; r < PPL; ++r) {
#pragma unroll
        for (int c = 0; c < MAXQ; ++c) {
          if ((uint32_t)c < gat.ppr) {
            P x;
            __builtin_memcpy(&x, &rg[(r * MAXQ + c) * W], PB);
            *reinterpret_cast<P *>(base + (tid + r * kLgBlock) * row_pitch + c * PB) = x;
          }
        }
      }
    }
    const uint32_t next = item + gridDim.x;
    Item nxt = cur;
    if (next < items) {
      nxt = ahead;
      rows_prefetch(nxt);                                        // its ancestors came an item ago
      if (next + gridDim.x < items) {
        ahead = locate(next + gridDim.x);
        anc_prefetch(ahead);
      }
    }
    // ---- the lane's particles ----------------------------------------------------------------------------
    bool live[PPL];
    uint32_t at[PPL], trow[PPL];
    int64_t n_of[PPL];
    uint32_t an[PPL];       // the particle's row in the noise tile (an idle lane: the particle it duplicates)
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      const uint32_t first = pick(cur.nf, seg[r]), rr = row_of(cur, r);
      live[r] = pick(cur.count, seg[r]) != 0;                             // wavefront-uniform
      at[r] = (tid + r * kLgBlock) * lx.rs;                               // x_{t-1}: the lane's own slot
      an[r] = (seg[r] * RUNP + rr) * lx.rs;
      n_of[r] = (int64_t)first + rr;
      const uint32_t k0 = first - (first / K) * K;
      trow[r] = ((seg[r] * kLgSegRows + ((k0 + rr) >= K ? 1u : 0u)) * 4) * DP;
    }
    T locp[DP][PPL], locq[DP][PPL], locg[DP][PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r)
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        locp[j][r] = tab[trow[r] + 0 * DP + j];
        locq[j][r] = tab[trow[r] + 1 * DP + j];
      }
    if constexpr (LgPacked<T, PPL>::value) {      // both particles of the lane per multiply-add (linear_gaussian.hpp)
      lg_f2 ap[DP], aq[DP];
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        ap[j] = lg_f2{locp[j][0], locp[j][1]};
        aq[j] = lg_f2{locq[j][0], locq[j][1]};
      }
#pragma unroll
      for (int i = 0; i < DP; ++i) {
        if ((uint32_t)i < dx) {
          const lg_f2 xv = lg_f2{tprev[at[0] + i], tprev[at[1] + i]};
          lg_pk_column<DP>(wp + i * DP, xv, ap);
          lg_pk_column<DP>(wq + i * DP, xv, aq);
        }
      }
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        locp[j][0] = ap[j].x; locp[j][1] = ap[j].y;
        locq[j][0] = aq[j].x; locq[j][1] = aq[j].y;
      }
    } else {
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
    }
    T xx[DP][PPL], qp[PPL], qq[PPL], qg[PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r) qp[r] = qq[r] = qg[r] = T(0);
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      if ((uint32_t)j < dx) {
#pragma unroll
        for (int r = 0; r < PPL; ++r) {
          xx[j][r] = locq[j][r] + tx[an[r] + j] * s_q;      // the product rounded before the sum, as K9 / K6
          const T dp = xx[j][r] - locp[j][r], dq = xx[j][r] - locq[j][r];
          qp[r] = fma_t(dp, dp, qp[r]);
          qq[r] = fma_t(dq, dq, qq[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < PPL; ++r) xx[j][r] = T(0);
      }
    }
    // x_t leaves from the registers of the lane that owns the particle
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
      if (live[r]) {
        char *to = out_bytes + (uint64_t)n_of[r] * gat.row_bytes;
#pragma unroll
        for (int c = 0; c < MAXQ; ++c) {
          if ((uint32_t)c < gat.ppr) {
            T piece[W];
#pragma unroll
            for (int e = 0; e < W; ++e) piece[e] = xx[(c * W + e) < DP ? (c * W + e) : DP - 1][r];
            P packed;
            __builtin_memcpy(&packed, piece, PB);
            *reinterpret_cast<P *>(to + c * PB) = packed;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < PPL; ++r)
#pragma unroll
      for (int j = 0; j < DP; ++j) locg[j][r] = tab[trow[r] + 2 * DP + j];
    if constexpr (LgPacked<T, PPL>::value) {
      lg_f2 ag[DP];
#pragma unroll
      for (int j = 0; j < DP; ++j) ag[j] = lg_f2{locg[j][0], locg[j][1]};
#pragma unroll
      for (int i = 0; i < DP; ++i)
        if ((uint32_t)i < dx) lg_pk_column<DP>(wg + i * DP, lg_f2{xx[i][0], xx[i][1]}, ag);
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        locg[j][0] = ag[j].x; locg[j][1] = ag[j].y;
      }
    } else {
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
    }
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      if ((uint32_t)j < dy) {
#pragma unroll
        for (int r = 0; r < PPL; ++r) {
          const T dg = tab[trow[r] + 3 * DP + j] - locg[j][r];
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
        out_lw[n_of[r]] = (lp + lg) - lq;
      }
    }
    lg_lds_barrier();     // hand-over: the other half of the buffers now holds the next item's noise and table
    slot ^= 1u;
    cur = nxt;
  }
  };
  if (gat.idx != nullptr) particles(std::true_type{});
  else particles(std::false_type{});
}

static int launch_affine_propagate_noise(const void *xsrc, const int64_t *anc_idx, const void *y, int64_t y_sb,
                                         const aesmc_affine_map *mp, const aesmc_affine_map *mg,
                                         const aesmc_affine_map *mq, const void *sp, const void *sg, const void *sq,
                                         void *out_x, void *out_lw, int32_t *flags, int64_t B, int64_t K,
                                         uint64_t seed, uint64_t offset, int64_t threads, const uint64_t *rng_state,
                                         hipStream_t stream) {
  using T = float;
  const int64_t N = B * K;
  const int64_t dx = mp->dout, dy = mg->dout;
  const int dp = lg_pad_dim(std::max(dx, dy));
  const uint64_t numel = (uint64_t)N * (uint64_t)dx;
  if (numel >= (1ull << 32) || N > 0x7fffffffLL) return AESMC_ERR_UNSUPPORTED;
  int ppl = 2;
  size_t lds = 0;
  for (; ppl >= 1; --ppl) {
    const size_t tp = (size_t)kLgBlock * ppl;
    lds = sizeof(T) * (3 * (size_t)dp * dp + (size_t)2 * 4 * kLgSegRows * 4 * dp + 3 * lg_tile_elems<T>(tp, dx));
    if (lds <= (size_t)80 * 1024) break;      // two workgroups per CU
  }
  if (ppl < 1) return AESMC_ERR_UNSUPPORTED;
  const auto calls = [&](int p) { return ((uint64_t)(kLgBlock * p / 4 + 1) * dx - 1) / kLgBlock; };
  if (ppl == 2 && (lg_few_tiles(N) || lg_forward_ppl() == 1) && calls(1) >= 1) {      // small launches: twice the workgroups
    ppl = 1;
    lds = sizeof(T) * (3 * (size_t)dp * dp + (size_t)2 * 4 * kLgSegRows * 4 * dp + 3 * lg_tile_elems<T>(kLgBlock, dx));
  }
  const uint32_t runp = (uint32_t)(kLgBlock * ppl / 4);
  if ((uint64_t)K < runp) return AESMC_ERR_UNSUPPORTED;        // a run of particles may span two batch rows, not more
  LgNoisePlan plan;
  plan.numel = numel;
  plan.magic = ((1ull << 40) + (uint64_t)dx - 1) / (uint64_t)dx;
  plan.small_magic = (uint32_t)(((1u << 20) + (uint32_t)dx - 1) / (uint32_t)dx);
  plan.S = (uint32_t)(((uint64_t)(runp + 1) * dx - 1) / kLgBlock);      // 256 S <= (RUNP + 1) d - 1: at most RUNP particles per window
  if (plan.S < 1) return AESMC_ERR_UNSUPPORTED;
  plan.L = plan.S * kLgBlock - ((uint32_t)dx - 1);
  plan.blocks = (uint32_t)(((uint64_t)threads + plan.L - 1) / plan.L);
  plan.trips = (uint32_t)((numel + 4ull * (uint64_t)threads - 1) / (4ull * (uint64_t)threads));
  const uint64_t items = (uint64_t)plan.trips * plan.blocks;
  if (items > 0x7fffffffull) return AESMC_ERR_UNSUPPORTED;
  const PhiloxStream ps = philox_stream(seed, offset, threads, rng_state);
  const LgGather gat = lg_gather(anc_idx, flags, (size_t)dx * sizeof(T));
  const dim3 grid(lg_persistent_grid((int64_t)items, lds, 2));
#define LG_NOISE_ARGS                                                                                                \
  static_cast<const T *>(xsrc), static_cast<const T *>(y), y_sb, lg_map(mp), lg_map(mg), lg_map(mq),                   \
      static_cast<const T *>(sp), static_cast<const T *>(sg), static_cast<const T *>(sq), static_cast<T *>(out_lw), N, \
      (uint32_t)K, (uint32_t)B, static_cast<T *>(out_x), gat, ps, plan
#define LG_NOISE_LAUNCH(DP_, PPL_, PB_)                                                                              \
  do {                                                                                                               \
    static bool raised[64] = {};                                                                                     \
    if (lds > 64 * 1024 &&                                                                                           \
        !lg_raise_lds_limit(reinterpret_cast<const void *>(&affine_propagate_noise_kernel<T, DP_, PPL_, PB_>), raised)) \
      return AESMC_ERR_LAUNCH;                                                                                       \
    hipLaunchKernelGGL((affine_propagate_noise_kernel<T, DP_, PPL_, PB_>), grid, dim3(kNoiseThreads), lds, stream,   \
                       LG_NOISE_ARGS);                                                                               \
  } while (0)
#define LG_NOISE_PB(DP_, PPL_)                                                                                       \
  do {                                                                                                               \
    if (gat.pb == 16) LG_NOISE_LAUNCH(DP_, PPL_, 16);                                                                \
    else if (gat.pb == 8) LG_NOISE_LAUNCH(DP_, PPL_, 8);                                                             \
    else LG_NOISE_LAUNCH(DP_, PPL_, 4);                                                                              \
  } while (0)
#define LG_NOISE_DP(PPL_)                                                                                            \
  do {                                                                                                               \
    switch (dp) {                                                                                                    \
      LG_NOISE_CASES(PPL_)                                                                                           \
    }                                                                                                                \
  } while (0)
#ifdef AESMC_LG_FAST_BUILD
#define LG_NOISE_CASES(PPL_) default: LG_NOISE_PB(10, PPL_); break;
#else
#define LG_NOISE_CASES(PPL_)                                                                                         \
  case 4: LG_NOISE_PB(4, PPL_); break;                                                                               \
  case 8: LG_NOISE_PB(8, PPL_); break;                                                                               \
  case 10: LG_NOISE_PB(10, PPL_); break;                                                                             \
  case 12: LG_NOISE_PB(12, PPL_); break;                                                                             \
  default: LG_NOISE_PB(16, PPL_); break;
#endif
  if (ppl == 2) LG_NOISE_DP(2);
  else LG_NOISE_DP(1);
#undef LG_NOISE_ARGS
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

// the second form of this launch (linear_gaussian_fused.hip: the maps on the matrix cores); AESMC_ERR_UNSUPPORTED
// for what it does not cover
int launch_affine_propagate_fused(const void *xsrc, const int64_t *anc_idx, const void *y, int64_t y_sb,
                                  const aesmc_affine_map *mp, const aesmc_affine_map *mg, const aesmc_affine_map *mq,
                                  const void *sp, const void *sg, const void *sq, void *out_x, void *out_lw,
                                  int32_t *flags, int64_t B, int64_t K, uint64_t seed, uint64_t offset,
                                  int64_t threads, const uint64_t *rng_state, const float *weight_pairs, hipStream_t stream);

}  // namespace aesmc

using namespace aesmc;

extern "C" int aesmc_affine_normal_propagate_drawn(
    const void *x_src, const int64_t *ancestors, const void *y, int64_t y_stride_b, const aesmc_affine_map *transition,
    const aesmc_affine_map *emission, const aesmc_affine_map *proposal, const void *scale_p, const void *scale_g,
    const void *scale_q, void *out_x, void *out_lw, int32_t *flags, int64_t B, int64_t K, uint64_t seed,
    uint64_t offset, int64_t threads, const uint64_t *rng_state, void *stream) {
  return aesmc_affine_normal_propagate_drawn_paired(x_src, ancestors, y, y_stride_b, transition, emission, proposal, scale_p,
                                                    scale_g, scale_q, out_x, out_lw, flags, B, K, seed, offset, threads,
                                                    rng_state, nullptr, stream);
}

extern "C" int aesmc_affine_normal_propagate_drawn_paired(
    const void *x_src, const int64_t *ancestors, const void *y, int64_t y_stride_b, const aesmc_affine_map *transition,
    const aesmc_affine_map *emission, const aesmc_affine_map *proposal, const void *scale_p, const void *scale_g,
    const void *scale_q, void *out_x, void *out_lw, int32_t *flags, int64_t B, int64_t K, uint64_t seed,
    uint64_t offset, int64_t threads, const uint64_t *rng_state, const void *weight_pairs, void *stream) {
  if ((reinterpret_cast<uintptr_t>(weight_pairs) & 15u) != 0) return AESMC_ERR_INVALID_ARGUMENT;
  if (x_src == nullptr || y == nullptr || transition == nullptr || emission == nullptr || proposal == nullptr ||
      scale_p == nullptr || scale_g == nullptr || scale_q == nullptr || out_lw == nullptr || out_x == nullptr || B < 0 ||
      K < 0 || threads <= 0 || (threads % 256) != 0 || threads > 0x7fffffffLL || (offset & 3u) != 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (!aligned16(x_src) || !aligned16(out_x) || out_x == x_src || (((uintptr_t)ancestors) & 7u) != 0 ||
      (((uintptr_t)rng_state) & 7u) != 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (!lg_map_ok(transition) || !lg_map_ok(emission) || !lg_map_ok(proposal)) return AESMC_ERR_UNSUPPORTED;
  const int64_t dx = transition->dout;
  if (transition->din != dx || proposal->dout != dx || proposal->din != dx || emission->din != dx)
    return AESMC_ERR_UNSUPPORTED;
  if (B == 0 || K == 0) return AESMC_OK;
  // AESMC_K16_FORM=roles: the first form whatever the shape (a measurement knob; both forms give the same bits)
  static const bool first_form = [] { const char *v = measurement_knob("AESMC_K16_FORM"); return v != nullptr && v[0] == 'r'; }();
  if (!first_form) {
    const int status = launch_affine_propagate_fused(x_src, ancestors, y, y_stride_b, transition, emission, proposal,
                                                     scale_p, scale_g, scale_q, out_x, out_lw, flags, B, K, seed, offset,
                                                     threads, rng_state, static_cast<const float *>(weight_pairs),
                                                     static_cast<hipStream_t>(stream));
    if (status != AESMC_ERR_UNSUPPORTED) return status;
  }
  return launch_affine_propagate_noise(x_src, ancestors, y, y_stride_b, transition, emission, proposal, scale_p,
                                       scale_g, scale_q, out_x, out_lw, flags, B, K, seed, offset, threads, rng_state,
                                       static_cast<hipStream_t>(stream));
}
