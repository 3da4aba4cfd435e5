// K16, second form: one SMC step's propagation — resampling gather, the proposal's draw with its noise formed in
// the launch, the three log-densities (aesmc/inference.py:102-126, state.py:98, :179 for a linear-Gaussian model) —
// with the three location maps on the matrix cores.
//
//   x_t[b,k,:] = loc_q(x_{t-1}[b, anc[b,k], :]) + s_q * eps[b,k,:]
//   lw[b,k]    = log N(x_t; A x + a, s_p) + log N(y_b; C x_t + g, s_g) - log N(x_t; loc_q, s_q)
//
// The arithmetic contract is K8-K15's (oracle/smc_core.c): a location element is ONE fma chain over the inputs in
// ascending order started from its offset.  v_mfma_f32_16x16x4_f32 IS that chain (exact float32 multiply-adds, k
// ascending, accumulator started from C — tools/probe_mfma.hip holds it against fmaf bit for bit on the device), so
// the maps leave the vector ALU — which torch's Philox rounds and Box-Muller keep busy — for a pipe that runs beside
// it, and the weights live in nine registers per lane for the whole launch instead of being re-read from LDS per
// multiply-add.
//
// Work items are the first form's (linear_gaussian_noise.hip): ATen hands the four normals of one Philox call to
// elements G apart, so an item is a block of thread ids of one trip and its four WINDOWS of <= 128 consecutive
// particles.  512 lanes in two roles.  Wavefronts 4-7 ("noise", pure vector arithmetic) draw the NEXT item's normals
// into the other half of a double-buffered LDS tile and stage its per-batch-row vectors; one barrier per item.
// Wavefront w of 0-3 ("particles") owns window w, in two chunks of 64 particles = 4 matrix tiles of 16:
//   * a tile's operand B[k][particle] is loaded straight from x_{t-1} through the ancestor index in the matrix
//     layout (lane = 16 (k mod 4) + particle: one dword per lane and k-step), one item ahead; the indices two ahead;
//   * D[j][particle] = offset_j + sum_k W[j][k] B[k][particle]; lanes hold 4 consecutive j of one particle: one
//     16-byte LDS store per tile into the wavefront's own scratch, read back as rows (lane = particle) for the
//     element-wise part: draw, residuals, the quadratic chains in ascending j (K10's order), the three divisions;
//   * x_t's rows go through the same scratch: they are the emission map's operand (dword reads in matrix layout) and
//     leave for HBM as ONE contiguous run per chunk in 16-byte pieces (1 KiB per store instruction) instead of
//     five 8-byte pieces per lane at a 40-byte stride.
// No lane-dependent branch surrounds a load: idle rows duplicate the window's last particle (same loads, same
// arithmetic, nothing stored), an empty window computes on particle 0 and stores nothing.
#include "linear_gaussian_fused.hpp"

namespace aesmc {


// Workgroup: four particle wavefronts and four noise wavefronts, two workgroups per CU (four wavefronts per SIMD).
// (Measured and not kept: SIX noise wavefronts for the scalar-weight form, which fits 96 registers — ten wavefronts
// per workgroup, five per SIMD, a lane's draws cut from five calls per item to four / three: 167 us against 127 at
// B=1024 K=4096 d=10 — the particle wavefronts are the critical path and lose issue slots to the extra noise ones.)
template <bool SCALARW> constexpr int fused_threads() { return 512; }

template <int KS, int DXC, int DYC, bool GATHER, bool SCALARW>
__global__ __launch_bounds__(fused_threads<SCALARW>(), 4) void affine_propagate_fused_kernel(
    const float *__restrict__ xsrc, const float *__restrict__ y, int64_t y_sb, LgMap mp, LgMap mg, LgMap mq,
    const float *__restrict__ sp_ptr, const float *__restrict__ sg_ptr, const float *__restrict__ sq_ptr,
    float *__restrict__ out_lw, uint32_t K, uint32_t Bn, float *__restrict__ out_x,
    const int64_t *__restrict__ anc_idx, int32_t *flags, PhiloxStream ps_in, FusedPlan plan) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int DP = 4 * KS;
  constexpr uint32_t RS = KS == 1 ? 4 : (KS <= 3 ? 12 : 20);      // scratch row stride: 16-byte rows on distinct banks
  const uint32_t dx = DXC ? (uint32_t)DXC : (uint32_t)mp.dout, dy = DYC ? (uint32_t)DYC : (uint32_t)mg.dout;
  extern __shared__ __attribute__((aligned(16))) unsigned char fused_smem[];
  float *tabs = reinterpret_cast<float *>(fused_smem);          // [2][kTabF]
  float *noise = tabs + 2 * kTabF;                              // [2][plan.tile_f]: window i's run at i * kRunP * dx, rows end to end
  float *scratch = noise + 2 * plan.tile_f;                     // [4 wavefronts][2][64 * RS]
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t G = plan.G;
  const uint32_t last_item = plan.items - 1;

  // window `i` of `item`: first particle, particle count, elements in front of the first particle
  auto window = [&](uint32_t item, uint32_t i) {
    FusedWin v;
    const uint32_t c = fused_div(item, plan.blocks, plan.blocks_mul);
    const uint32_t t0 = (item - c * plan.blocks) * plan.L;
    const uint32_t tl = min(plan.L, G - t0);
    const uint32_t lo = G * (4u * c + i) + t0;
    if (lo >= plan.numel) {
      v.nf = 0; v.count = 0; v.head = 0;
    } else {
      const uint32_t hi = min(lo + tl, plan.numel);
      const uint32_t nf = fused_div(lo + dx - 1, dx, plan.dx_mul), nl = fused_div(hi + dx - 1, dx, plan.dx_mul);
      v.count = nl - nf;
      v.nf = nl != nf ? nf : 0u;
      v.head = nf * dx - lo;
    }
    v.b0 = fused_div(v.nf, K, plan.K_mul);
    v.k0 = v.nf - v.b0 * K;
    v.c = c; v.t0 = t0; v.tl = tl;
    return v;
  };
  // What the noise role needs of an item, 16 words: head[4], limit[4] (= count * dx), trip, t0, span, -, row[4] (the
  // windows' first batch row; ~0 for an empty window).  The particle wavefronts compute their window of every item
  // anyway (three items ahead, for their prefetches) and leave it in a ring of four records in LDS; the noise
  // wavefronts read it one and two items later instead of locating all four windows themselves (scalar divisions: a
  // third of their instruction stream before).
  uint32_t *ring = reinterpret_cast<uint32_t *>(scratch + 4u * 2u * 64u * RS);      // [4][16]
  auto publish = [&](uint32_t it, const FusedWin &v, uint32_t i) {
    uint32_t *rec = ring + (it & 3u) * 16u;
    rec[i] = v.head;
    rec[4 + i] = v.count * dx;
    rec[12 + i] = v.count != 0 ? v.b0 : ~0u;
    if (i == 0) {
      rec[8] = v.c;
      rec[9] = v.t0;
      rec[10] = v.tl + dx - 1;
    }
  };

  if (w >= 4) {
    // ================================ the noise role ======================================================
    const PhiloxStream ps = philox_resolve(ps_in);
    const uint32_t tid = threadIdx.x - 256u;
    const LgRowVec<float> vec[4] = {lg_offset_vec<float>(mp), lg_offset_vec<float>(mq), lg_offset_vec<float>(mg),
                                    {y, y_sb, (int)dy}};
    // the table's values of an item -> two registers per lane (the loads fly while the caller draws: they are sent for one
    // item before they are written).  Wavefront nw covers entries [64 nw, 64 nw + 64) of each 256-entry trip: window
    // nw / 2 + 2 trip, batch row nw & 1 — uniform; the lane picks the vector (lane >> 4) and the element (lane & 15).
    // `row` = the window's first batch row out of the item's record.
    constexpr uint32_t NW = (uint32_t)fused_threads<SCALARW>() / 64u - 4u, NL = 64u * NW;       // noise wavefronts, their lanes
    const uint32_t nw = w - 4u, tab_a = lane >> 4, tab_j = lane & 15u;
    // (which vector a lane reads, where and how long it is: fixed for the launch)
    const float *tab_src = vec[0].ptr;
    int64_t tab_sb = vec[0].sb;
    int tab_len = vec[0].ptr != nullptr ? vec[0].len : 0;
#pragma unroll
    for (int cidx = 1; cidx < 4; ++cidx) {
      const bool mine = tab_a == (uint32_t)cidx;
      tab_src = mine ? vec[cidx].ptr : tab_src;
      tab_sb = mine ? vec[cidx].sb : tab_sb;
      tab_len = mine ? (vec[cidx].ptr != nullptr ? vec[cidx].len : 0) : tab_len;
    }
    const bool tab_live = (int)tab_j < tab_len;
    tab_src += tab_live ? tab_j : 0u;
    // the table's eight 64-entry blocks (window = block / 2, batch row = block & 1) go round the wavefronts
    auto table_load = [&](const uint32_t (&row)[2], float (&held)[2]) {
#pragma unroll
      for (int trip = 0; trip < 2; ++trip) {
        const uint32_t block = nw + NW * trip;
        held[trip] = 0.0f;
        if (block >= 8u) continue;       // uniform
        const uint32_t b = row[trip] + (block & 1u);
        const bool row_ok = row[trip] != ~0u && b < Bn;      // uniform
        float value = 0.0f;
        if (row_ok && tab_live) value = tab_src[(int64_t)b * tab_sb];
        held[trip] = value;
      }
    };
    struct Record {
      uint32_t head[4], limit[4], c, t0, span, row[4];
    };
    auto locate = [&](uint32_t item) {       // before the first barrier only: nobody has published anything yet
      Record r;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const FusedWin v = window(item, (uint32_t)i);
        r.head[i] = v.head; r.limit[i] = v.count * dx; r.row[i] = v.count != 0 ? v.b0 : ~0u;
        r.c = v.c; r.t0 = v.t0; r.span = v.tl + dx - 1;
      }
      return r;
    };
    auto read_record = [&](uint32_t it) {
      const uint32_t *rec = ring + (it & 3u) * 16u;
      Record r;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        r.head[i] = rec[i]; r.limit[i] = rec[4 + i];
        r.row[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[12 + i]);
      }
      r.c = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[8]);
      r.t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[9]);
      r.span = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[10]);
      return r;
    };
    auto draw_item = [&](const Record &r, uint32_t slot, const float (&held)[2]) {
      float *tx = noise + slot * plan.tile_f;
      float *tab = tabs + slot * kTabF;
      auto place = [&](uint32_t j, const float4 &first, const float4 &second, bool wraps) {
        float n4[4] = {first.x, first.y, first.z, first.w};
        if (wraps) {
          n4[0] = first.y; n4[1] = first.z; n4[2] = first.w; n4[3] = second.x;
        }
        // rows of the tile lie end to end: element v of a window's run is at v — one unsigned comparison places a
        // normal (j < head wraps around to a huge v; thread ids past the block's span fall behind every limit), and
        // a normal that belongs to nobody goes to a spare word behind the tile: no branch around the store
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t v = j - r.head[i];
          tx[v < r.limit[i] ? (uint32_t)i * kRunP * dx + v : 4u * kRunP * dx] = n4[i];
        }
      };
      auto draw = [&](uint32_t j, float4 &first, float4 &second, bool &wraps) {
        // past the last thread id (the last block of a trip, at most dx - 1 lanes): element G (4c + i) + t is thread
        // t - G's output i + 1 of this trip, or its output 0 of the next
        const uint32_t t = r.t0 + j;
        wraps = t >= G;
        const uint32_t tt = wraps ? t - G : t;
        first = make_float4(0.5f, 0.25f, 0.125f, 1.0f);
        second = first;
        {
          first = philox_normal4(ps, tt, r.c);
          if (__any(wraps)) second = philox_normal4(ps, tt, r.c + 1);      // (uniform branch, rarely taken)
        }
      };
      // A lane's calls are independent dependency chains (ten Philox rounds, then Box-Muller's logarithm, root, sine and
      // cosine): with the extent known at compile time all of an item's calls are written out as one straight line for the
      // vector ALU to interleave (one call at a time a noise wavefront issued an instruction every ~7 cycles even with
      // the SIMD to itself).  No lane-dependent branch: a lane past the span draws a value that lands on the spare word.
      constexpr uint32_t SC = DXC != 0 ? ((kRunP + 1u) * (uint32_t)DXC - 1u) / 256u : 0u;      // == plan.S
      if (SC != 0 && NL == 256u && r.t0 + r.span <= G) {      // (uniform; no thread id wraps: all but a trip's last block)
        float4 drawn[SC != 0 ? SC : 1];
#pragma unroll
        for (uint32_t s = 0; s < SC; ++s) drawn[s] = philox_normal4(ps, r.t0 + tid + s * NL, r.c);
#pragma unroll
        for (uint32_t s = 0; s < SC; ++s) place(tid + s * NL, drawn[s], drawn[s], false);
      } else {
      // two Philox calls per trip
#pragma unroll 1
      for (uint32_t s = 0; s * NL < r.span; s += 2) {
        const uint32_t j0 = tid + s * NL, j1 = j0 + NL;
        float4 a0, b0, a1, b1;
        bool w0, w1 = false;
        draw(j0, a0, b0, w0);
        const bool two = (s + 1) * NL < r.span;      // uniform
        if (two) draw(j1, a1, b1, w1);
        place(j0, a0, b0, w0);
        if (two) place(j1, a1, b1, w1);
      }
      }
#pragma unroll
      for (int trip = 0; trip < 2; ++trip)
        if (nw + NW * trip < 8u) tab[64u * (nw + NW * trip) + lane] = held[trip];
    };
    float held[2], held_next[2];
    {
      const Record first = locate(min(blockIdx.x, last_item)), second = locate(min(blockIdx.x + gridDim.x, last_item));
      const uint32_t rows0[2] = {first.row[nw >> 1], first.row[((nw + NW) >> 1) & 3u]};
      const uint32_t rows1[2] = {second.row[nw >> 1], second.row[((nw + NW) >> 1) & 3u]};
      table_load(rows0, held);
      table_load(rows1, held_next);
      draw_item(first, 0, held);
    }
    lg_lds_barrier();
    uint32_t slot = 0, it = 0;
    for (uint32_t item = blockIdx.x; item < plan.items; item += gridDim.x, ++it) {
      const uint32_t next = item + gridDim.x;
      held[0] = held_next[0];
      held[1] = held_next[1];
      {
        const uint32_t *rec = ring + ((it + 2u) & 3u) * 16u;      // the record of item + 2 strides
        const uint32_t rows[2] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)rec[12 + (nw >> 1)]),
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[12 + (((nw + NW) >> 1) & 3u)])};
        table_load(rows, held_next);
      }
      if (next < plan.items) draw_item(read_record(it + 1u), slot ^ 1u, held);
      lg_lds_barrier();
      slot ^= 1u;
    }
    return;
  }

  // ==================================== the particle role ==================================================
  // (g, n, ln are refreshed through an opaque move per chunk: everything derived from the lane's index is loop-
  // invariant, and the compiler would otherwise hold dozens of LDS addresses in registers across the whole loop)
  uint32_t ln = lane, g = lane >> 4, n = lane & 15u;
  // (s_setprio for the particle wavefronts — measured with 2 and without: 120.6 against 120.6-121.1 us — is left out.
  //  Also measured and not kept: both chunks of a window through every phase together — four chains per scalar
  //  weight pair instead of two, a weight fetched once per item — needs more registers than 128: 45 spilled.)
  float *scr_q = scratch + w * (2u * 64u * RS);
  float *scr_p = scr_q + 64u * RS;
  // the maps as matrix operands A[m = output j][k = input i]: lane holds W[n][4 s + g] of k-step s
  float aq[KS], ap[KS], ag[KS];
#pragma unroll
  for (int s = 0; s < (SCALARW ? 0 : KS); ++s) {
    const uint32_t i = 4u * s + g;
    const float *wq = reinterpret_cast<const float *>(mq.w), *wp = reinterpret_cast<const float *>(mp.w),
                *wg = reinterpret_cast<const float *>(mg.w);
    aq[s] = (n < dx && i < dx) ? wq[(int64_t)n * mq.sj + (int64_t)i * mq.si] : 0.0f;
    ap[s] = (n < dx && i < dx) ? wp[(int64_t)n * mp.sj + (int64_t)i * mp.si] : 0.0f;
    ag[s] = (n < dy && i < dx) ? wg[(int64_t)n * mg.sj + (int64_t)i * mg.si] : 0.0f;
  }
  // (the step's constants through readfirstlane: wavefront-uniform values the compiler would otherwise keep in nine
  //  vector registers for the whole loop)
  auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
  const float s_p = sp_ptr[0], s_g = sg_ptr[0], s_q = sq_ptr[0];
  const float half_log_2pi = LgConst<float>::half_log_2pi();
  const float two_var_p = uniform(2.0f * (s_p * s_p)), const_p = uniform((float)dx * (Num<float>::log(s_p) + half_log_2pi));
  const float two_var_g = uniform(2.0f * (s_g * s_g)), const_g = uniform((float)dy * (Num<float>::log(s_g) + half_log_2pi));
  const float two_var_q = uniform(2.0f * (s_q * s_q)), const_q = uniform((float)dx * (Num<float>::log(s_q) + half_log_2pi));

  // row of tile T (0..7) this lane feeds as matrix operand, clamped to the window's last particle
  auto tile_row = [&](const FusedWin &win, int T) { return min(16u * (uint32_t)T + n, win.count != 0 ? win.count - 1 : 0u); };
  // row of chunk c this lane owns in the element-wise part (and fetches: below)
  auto lane_row = [&](const FusedWin &win, int c) { return min(64u * (uint32_t)c + ln, win.count != 0 ? win.count - 1 : 0u); };
  // Loads go through buffer descriptors: a 32-bit byte offset per lane plus an immediate per piece instead of a
  // 64-bit address per load, and a read past the array's end returns zero instead of faulting.
  const __amdgpu_buffer_rsrc_t x_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xsrc), 0, (int)(plan.numel * 4u), 0x00020000);
  // Rows of x_{t-1} are fetched by the lane that owns the particle in the element-wise part, in pieces from one row
  // address, and reach the matrix layout through LDS at the top of their chunk.  (The first version of this kernel loaded
  // the matrix layout directly, one dword per lane and k-step: the four lanes that share a row are 16 lanes apart,
  // so nothing coalesced — 64 separate requests per instruction, four times the requests of whole rows, and the
  // vector-memory pipe, not the arithmetic, set the launch's time.)
  // The ancestor (identity without resampling in front of the step): the whole int64, two of them per lane and item.
  // (GATHER is a template parameter: as a runtime branch around the load it made the compiler wait for each index on
  // the spot — exposed memory round trips per item)
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  auto anc_load = [&](const FusedWin &win, int c) -> u32x2 {
    const uint32_t r = lane_row(win, c);
    if constexpr (GATHER) {
      // (a plain 8-byte load: __builtin_amdgcn_raw_buffer_load_b64 compiles to a ONE-dword load with ROCm 7.2's hipcc)
      return __builtin_bit_cast(u32x2, anc_idx[win.nf + r]);
    } else {
      const uint32_t k = win.k0 + r;
      return u32x2{k >= K ? k - K : k, 0u};
    }
  };
  uint32_t bad = 0;       // out-of-range indices seen by this lane: reported once, after the loop (no atomic inside it)
  const uint32_t Kdx = K * dx;
  constexpr int XN = SCALARW ? DXC : DP;      // a row's registers (the matrix form pads to whole k-steps)
  auto rows_load = [&](const FusedWin &win, int c, u32x2 raw, float (&x)[XN]) {
    const uint32_t r = lane_row(win, c);
    uint32_t a = raw[0];
    // K2 writes K for a degenerate row (flagged there); never fault on it
    bad |= (a >= K ? 1u : 0u) | raw[1];
    a = a < K ? a : ((int32_t)raw[1] < 0 ? 0u : K - 1);
    const uint32_t base = win.b0 * Kdx + ((win.k0 + r) >= K ? Kdx : 0u);      // the batch row's first element
    const uint32_t off = (base + __umul24(a, dx)) << 2;
    // naturally aligned pieces that cover the row exactly when the extent is known at compile time and even — 16
    // bytes when rows are whole 16-byte vectors, else 8; otherwise dwords through the buffer descriptor (elements past
    // the row's end come from the next row, or are the zeros behind the array: unused)
    if constexpr (DXC != 0 && DXC % 2 == 0) {
      constexpr int PW = DXC % 4 == 0 ? 4 : 2;
      const float *at = xsrc + (base + __umul24(a, dx));
#pragma unroll
      for (int e0 = 0; e0 < XN; e0 += PW) {
        if (e0 >= DXC) {
#pragma unroll
          for (int e = 0; e < PW; ++e) x[e0 + e] = 0.0f;
        } else if constexpr (PW == 4) {
          const fz4 piece = *reinterpret_cast<const fz4 *>(at + e0);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e0 + e] = piece[e];
        } else {
          const float2 piece = *reinterpret_cast<const float2 *>(at + e0);
          x[e0] = piece.x;
          x[e0 + 1] = piece.y;
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < XN; ++e)
        x[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off + 4u * e, 0, 0));
    }
  };

  const uint32_t stride = gridDim.x;
  FusedWin cur = window(min(blockIdx.x, last_item), w);
  FusedWin nxt = window(min(blockIdx.x + stride, last_item), w);
  FusedWin ahd = window(min(blockIdx.x + 2 * stride, last_item), w);
  u32x2 araw[2];
  float xr[2][XN];
#pragma unroll
  for (int c = 0; c < 2; ++c) araw[c] = anc_load(cur, c);
#pragma unroll
  for (int c = 0; c < 2; ++c) rows_load(cur, c, araw[c], xr[c]);
#pragma unroll
  for (int c = 0; c < 2; ++c) araw[c] = anc_load(nxt, c);
  publish(1, nxt, w);
  publish(2, ahd, w);
  lg_lds_barrier();       // the first item's noise and table are there
  uint32_t slot = 0, it = 0;
  for (uint32_t item = blockIdx.x; item < plan.items; item += stride, ++it) {
    const float *tx = noise + slot * plan.tile_f + w * kRunP * dx;
    const float *tab = tabs + slot * kTabF + w * (2u * 4u * 16u);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      ln = lane;
      asm volatile("" : "+v"(ln));
      g = ln >> 4;
      n = ln & 15u;
      if constexpr (SCALARW) {
        // ---- the maps on the vector ALU with their weights in scalar registers (rows of W contiguous in memory: scalar
        //      loads fetch them, nothing is staged): lane = particle from the first instruction on, no transposes; LDS
        //      holds the chunk's rows of x_t on their way out only.  Same chains, same order (input ascending, started
        //      from the offset), same bits as the matrix-core form below.
        // (the chunk's rows are read where the prefetch left them; the next item's are sent for once the two chains that
        //  read them are done)
        float (&xin)[XN] = xr[c];
        const uint32_t rl = 64u * c + ln;
        const uint32_t rr = lane_row(cur, c);
        const bool live = rl < cur.count;
        const float *trow = tab + ((cur.k0 + rr) >= K ? 64u : 0u);
        // (the pointers through an opaque move per chunk: the 3 d^2 weights are loop-invariant, and hoisted out of the
        //  item loop they would be spilled into vector registers and read back lane by lane)
        unsigned long long wq_a = (unsigned long long)mq.w, wp_a = (unsigned long long)mp.w, wg_a = (unsigned long long)mg.w;
        asm volatile("" : "+s"(wq_a), "+s"(wp_a), "+s"(wg_a));
        // Two outputs at a time (two independent chains back to back), their 2 din weights — two rows of W, contiguous —
        // in scalar registers; the NEXT pair's rows are sent for before this pair's multiply-adds (scalar loads come
        // back out of order, so the only wait there is waits for all of them: it must sit behind a block of work).
        auto chain = [&](unsigned long long base, const float *offsets, uint32_t dout, const float *in, float (&acc)[DP]) {
          fused_cfloat *W = (fused_cfloat *)base;
#pragma unroll
          for (int v = 0; v < KS; ++v) {
            const fz4 o4 = *reinterpret_cast<const fz4 *>(offsets + 4 * v);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * v + e] = o4[e];
          }
          constexpr int DIN = DXC;       // (this path is instantiated for compile-time extents only)
          float wa[2 * DIN], wb[2 * DIN];
#pragma unroll
          for (int e = 0; e < 2 * DIN; ++e) wa[e] = W[e];
#pragma unroll
          for (int jb = 0; jb < DP; jb += 2) {
            if ((uint32_t)jb >= dout) break;
            float (&cur_w)[2 * DIN] = (jb & 2) ? wb : wa;
            float (&next_w)[2 * DIN] = (jb & 2) ? wa : wb;
            if ((uint32_t)(jb + 2) < dout) {
#pragma unroll
              for (int e = 0; e < 2 * DIN; ++e) next_w[e] = W[(jb + 2) * DIN + min(e, (int)(dout - jb - 2) * DIN - 1)];
            }
            if constexpr (DIN % 5 == 0) {
              if ((uint32_t)(jb + 1) < dout) {
#pragma unroll
                for (int i0 = 0; i0 < DIN; i0 += 5) fused_fmac_s5x2(acc[jb], acc[jb + 1], cur_w + i0, cur_w + DIN + i0, in + i0);
                continue;
              }
            }
#pragma unroll
            for (int i = 0; i < DIN; ++i) {
              acc[jb] = fused_fmac_s(acc[jb], cur_w[i], in[i]);
              if ((uint32_t)(jb + 1) < dout) acc[jb + 1] = fused_fmac_s(acc[jb + 1], cur_w[DIN + i], in[i]);
            }
          }
        };
        float locq[DP], locp[DP];
        float xx[DP], qp = 0.0f, qq = 0.0f, qg = 0.0f;
        {
          chain(wq_a, trow + 16, dx, xin, locq);
          chain(wp_a, trow, dx, xin, locp);
          rows_load(nxt, c, araw[c], xr[c]);
          araw[c] = anc_load(ahd, c);
#pragma unroll
          for (int j = 0; j < DP; ++j) {
            if ((uint32_t)j < dx) {
              xx[j] = locq[j] + tx[rr * dx + j] * s_q;      // the product rounded before the sum, as K9 / K6
              const float ep = xx[j] - locp[j], eq = xx[j] - locq[j];
              qp = fma_t(ep, ep, qp);
              qq = fma_t(eq, eq, qq);
            } else {
              xx[j] = 0.0f;
            }
          }
          float locg[DP];
          chain(wg_a, trow + 32, dy, xx, locg);
#pragma unroll
          for (int v = 0; v < KS; ++v) {
            const fz4 y4 = *reinterpret_cast<const fz4 *>(trow + 48 + 4 * v);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if ((uint32_t)(4 * v + e) < dy) {
                const float eg = y4[e] - locg[4 * v + e];
                qg = fma_t(eg, eg, qg);
              }
            }
          }
#pragma unroll
          for (int j = 0; j < DP; ++j)
            if ((uint32_t)j < dx) scr_q[ln * dx + j] = xx[j];
        }
        if (live) {
          const float lp = (-qp) / two_var_p - const_p;
          const float lg = (-qg) / two_var_g - const_g;
          const float lq = (-qq) / two_var_q - const_q;
          out_lw[cur.nf + rl] = (lp + lg) - lq;
        }
        {
          const uint32_t rows = cur.count > 64u * c ? min(cur.count - 64u * c, 64u) : 0u;
          const uint32_t words = rows * dx;
          float *run = out_x + (size_t)(cur.nf + 64u * c) * dx;
#pragma unroll
          for (int u = 0; u < KS; ++u) {
            const uint32_t q = ln + 64u * u;
            if (4u * q + 4u <= words)
              *reinterpret_cast<fz4_a4 *>(run + 4u * q) = *reinterpret_cast<const fz4 *>(scr_q + 4u * q);
          }
          if ((words & 3u) != 0u && ln < (words & 3u)) {
            const uint32_t e = (words & ~3u) + ln;
            run[e] = scr_q[e];
          }
        }
      } else {
      // ---- proposal and transition locations of the chunk's 4 tiles on the matrix cores: D[j][particle] (lanes hold
      //      4 consecutive j of one particle) -> the wavefront's scratch rows [particle][RS].  Two tiles at a time:
      //      four independent accumulation chains keep the matrix pipe issuing back to back (one tile's three
      //      dependent products of each map would wait out the accumulator latency) ---------------------------------
      // the chunk's rows (lane = particle) -> scr_p, end to end; back as the matrix operand B[k = 4 s + g][particle n]
      // of its four tiles (a wavefront's LDS accesses execute in order)
      float bx[4][KS];
      {
#pragma unroll
        for (int j = 0; j < XN; ++j)
          if ((uint32_t)j < dx) scr_p[ln * dx + j] = xr[c][j];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const uint32_t i = 4u * s + g;
            const float b = scr_p[(16u * t + n) * dx + min(i, dx - 1)];
            bx[t][s] = ((DXC != 0 && 4 * s + 3 < DXC) || i < dx) ? b : 0.0f;
          }
      }
      // ---- the next item's rows into the registers just emptied; then the item after's ancestors ----------------
      rows_load(nxt, c, araw[c], xr[c]);
      araw[c] = anc_load(ahd, c);
#pragma unroll
      for (int pair = 0; pair < 2; ++pair) {
        fz4 dq4[2], dp4[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint32_t r = tile_row(cur, 4 * c + 2 * pair + h);
          const float *row = tab + ((cur.k0 + r) >= K ? 64u : 0u) + 4u * g;
          dp4[h] = *reinterpret_cast<const fz4 *>(row);             // offsets: the chains' starting values
          dq4[h] = *reinterpret_cast<const fz4 *>(row + 16);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float b = bx[2 * pair + h][s];
            dq4[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s], b, dq4[h], 0, 0, 0);
            dp4[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[s], b, dp4[h], 0, 0, 0);
          }
        }
        if (g < (uint32_t)KS) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t t = 2 * pair + h;
            *reinterpret_cast<fz4 *>(scr_q + (16u * t + n) * RS + 4u * g) = dq4[h];
            *reinterpret_cast<fz4 *>(scr_p + (16u * t + n) * RS + 4u * g) = dp4[h];
          }
        }
      }
      // ---- lane = particle: draw, residuals, quadratic chains (ascending j); x_t's rows, end to end, go into the
      //      scratch the proposal's locations came from (a wavefront's LDS accesses execute in order: every lane
      //      has read its locations before any row is overwritten) ------------------------------------------------
      const uint32_t rl = 64u * c + ln;
      const uint32_t rr = lane_row(cur, c);
      const bool live = rl < cur.count;
      const uint32_t relL = (cur.k0 + rr) >= K ? 1u : 0u;
      float xx[DP], qp = 0.0f, qq = 0.0f;
      {
        fz4 q4[KS], p4[KS];
#pragma unroll
        for (int v = 0; v < KS; ++v) {
          q4[v] = *reinterpret_cast<const fz4 *>(scr_q + ln * RS + 4 * v);
          p4[v] = *reinterpret_cast<const fz4 *>(scr_p + ln * RS + 4 * v);
        }
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          if ((uint32_t)j < dx) {
            const float lq = q4[j / 4][j % 4], lp = p4[j / 4][j % 4];
            xx[j] = lq + tx[rr * dx + j] * s_q;      // the product rounded before the sum, as K9 / K6
            const float ep = xx[j] - lp, eq = xx[j] - lq;
            qp = fma_t(ep, ep, qp);
            qq = fma_t(eq, eq, qq);
          } else {
            xx[j] = 0.0f;
          }
        }
#pragma unroll
        for (int j = 0; j < DP; ++j)
          if ((uint32_t)j < dx) scr_q[ln * dx + j] = xx[j];
      }
      // ---- emission locations: operand B from those rows; the four tiles' chains interleaved ----------------------
      {
        fz4 dg4[4];
        float bg[4][KS];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t r = tile_row(cur, 4 * c + t);
          dg4[t] = *reinterpret_cast<const fz4 *>(tab + ((cur.k0 + r) >= K ? 64u : 0u) + 32u + 4u * g);
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const uint32_t i = 4u * s + g;
            const float b = scr_q[(16u * t + n) * dx + min(i, dx - 1)];
            bg[t][s] = ((DXC != 0 && 4 * s + 3 < DXC) || i < dx) ? b : 0.0f;
          }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int t = 0; t < 4; ++t) dg4[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag[s], bg[t][s], dg4[t], 0, 0, 0);
        if (g < (uint32_t)KS) {
#pragma unroll
          for (int t = 0; t < 4; ++t) *reinterpret_cast<fz4 *>(scr_p + (16u * t + n) * RS + 4u * g) = dg4[t];
        }
      }
      float qg = 0.0f;
      {
#pragma unroll
        for (int v = 0; v < KS; ++v) {
          const fz4 g4 = *reinterpret_cast<const fz4 *>(scr_p + ln * RS + 4 * v);
          const fz4 y4 = *reinterpret_cast<const fz4 *>(tab + relL * 64u + 48u + 4 * v);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if ((uint32_t)(4 * v + e) < dy) {
              const float eg = y4[e] - g4[e];
              qg = fma_t(eg, eg, qg);
            }
          }
        }
      }
      if (live) {
        const float lp = (-qp) / two_var_p - const_p;
        const float lg = (-qg) / two_var_g - const_g;
        const float lq = (-qq) / two_var_q - const_q;
        out_lw[cur.nf + rl] = (lp + lg) - lq;
      }
      // ---- the chunk's rows of x_t leave as one contiguous run ----------------------------------------------------
      {
        const uint32_t rows = cur.count > 64u * c ? min(cur.count - 64u * c, 64u) : 0u;
        const uint32_t words = rows * dx;
        float *run = out_x + (size_t)(cur.nf + 64u * c) * dx;
#pragma unroll
        for (int u = 0; u < KS; ++u) {
          const uint32_t q = ln + 64u * u;
          if (4u * q + 4u <= words)
            *reinterpret_cast<fz4_a4 *>(run + 4u * q) = *reinterpret_cast<const fz4 *>(scr_q + 4u * q);
        }
        if ((words & 3u) != 0u && ln < (words & 3u)) {      // (uniform test first: a run of whole 16-byte pieces has no tail)
          const uint32_t e = (words & ~3u) + ln;
          run[e] = scr_q[e];
        }
      }
    
      }
    }
    const FusedWin far = window(min(item + 3 * stride, last_item), w);
    publish(it + 3u, far, w);
    lg_lds_barrier();     // hand-over: the other half of the buffers now holds the next item's noise and table
    slot ^= 1u;
    cur = nxt;
    nxt = ahd;
    ahd = far;
  }
  if (bad != 0u) raise_flag(flags, AESMC_FLAG_INDEX_OUT_OF_RANGE);
#endif
}

template <int KS, int DXC, int DYC, bool SCALARW>
static int fused_launch_g(dim3 grid, size_t lds, hipStream_t stream, const float *xsrc, const float *y, int64_t y_sb,
                          const LgMap &mp, const LgMap &mg, const LgMap &mq, const float *sp, const float *sg,
                          const float *sq, float *out_lw, uint32_t K, uint32_t Bn, float *out_x, const int64_t *anc,
                          int32_t *flags, const PhiloxStream &ps, const FusedPlan &plan) {
  static bool raised[2][64] = {};
  if (anc != nullptr) {
    if (lds > 64 * 1024 &&
        !lg_raise_lds_limit(reinterpret_cast<const void *>(&affine_propagate_fused_kernel<KS, DXC, DYC, true, SCALARW>), raised[0]))
      return AESMC_ERR_LAUNCH;
    hipLaunchKernelGGL((affine_propagate_fused_kernel<KS, DXC, DYC, true, SCALARW>), grid, dim3(fused_threads<SCALARW>()), lds, stream, xsrc,
                       y, y_sb, mp, mg, mq, sp, sg, sq, out_lw, K, Bn, out_x, anc, flags, ps, plan);
  } else {
    if (lds > 64 * 1024 &&
        !lg_raise_lds_limit(reinterpret_cast<const void *>(&affine_propagate_fused_kernel<KS, DXC, DYC, false, SCALARW>), raised[1]))
      return AESMC_ERR_LAUNCH;
    hipLaunchKernelGGL((affine_propagate_fused_kernel<KS, DXC, DYC, false, SCALARW>), grid, dim3(fused_threads<SCALARW>()), lds, stream, xsrc,
                       y, y_sb, mp, mg, mq, sp, sg, sq, out_lw, K, Bn, out_x, anc, flags, ps, plan);
  }
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

int launch_affine_propagate_item(const void *xsrc, const int64_t *anc_idx, const void *y, int64_t y_sb,
                                 const aesmc_affine_map *mp, const aesmc_affine_map *mg, const aesmc_affine_map *mq,
                                 const void *sp, const void *sg, const void *sq, void *out_x, void *out_lw, int32_t *flags,
                                 int64_t B, int64_t K, uint64_t seed, uint64_t offset, int64_t threads,
                                 const uint64_t *rng_state, const float *weight_pairs, hipStream_t stream);

int g_fused_form = [] {
  const char *v = measurement_knob("AESMC_K16_FORM");      // ("roles": the first form whatever the shape — linear_gaussian_noise.hip)
  return v == nullptr ? 0 : (v[0] == 'p' ? 1 : (v[0] == 'i' ? 2 : 0));
}();
int g_fused_last_form = 0;
// One item per workgroup wherever that form applies: measured ahead of the persistent form at every size and extent tried
// (profiles/r05_k16_forms.txt: B = 128 / 256 / 512 / 1024 at K = 4096, d = 10: 19.3 / 31.3 / 58.5 / 110.7 against 25.2 / 38.8 /
// 63.7 / 113.8 us; d = 4 / 8 / 12 at B = 1024: 48.8 / 86.9 / 132.5 against 79.6 / 133.2 / 165.8 us).  The persistent form stays
// for weights whose rows are not contiguous (its matrix-core branch) and as the comparison the tests pin the bits to.
constexpr uint32_t kItemFormMaxItems = 0xffffffffu;

// AESMC_ERR_UNSUPPORTED: the caller takes the first form (linear_gaussian_noise.hip)
int launch_affine_propagate_fused(const void *xsrc, const int64_t *anc_idx, const void *y, int64_t y_sb,
                                  const aesmc_affine_map *mp, const aesmc_affine_map *mg, const aesmc_affine_map *mq,
                                  const void *sp, const void *sg, const void *sq, void *out_x, void *out_lw,
                                  int32_t *flags, int64_t B, int64_t K, uint64_t seed, uint64_t offset,
                                  int64_t threads, const uint64_t *rng_state, const float *weight_pairs, hipStream_t stream) {
  const int64_t dx = mp->dout, dy = mg->dout;
  g_fused_last_form = 1;      // (or, where this file declines, the first form: the caller's)
  if (dx < 2 || dx > 16 || dy < 1 || dy > 16) return AESMC_ERR_UNSUPPORTED;
  FusedPlan plan;
  const int planned = fused_make_plan(plan, B, K, dx, threads);
  if (planned != AESMC_OK) return planned;
  // One item per workgroup (linear_gaussian_item.hip); the same bits either way.
  if (g_fused_form == 2 || (g_fused_form == 0 && plan.items < kItemFormMaxItems)) {
    const int status = launch_affine_propagate_item(xsrc, anc_idx, y, y_sb, mp, mg, mq, sp, sg, sq, out_x, out_lw, flags, B,
                                                    K, seed, offset, threads, rng_state, weight_pairs, stream);
    if (status != AESMC_ERR_UNSUPPORTED) {
      g_fused_last_form = 2;
      return status;
    }
  }
  const int ks = (int)((std::max(dx, dy) + 3) / 4);
  // extents above 12: the first form is faster (rows of 16 values put a wavefront's noise reads on two LDS banks
  // here: 371 against 337 us at B=1024 K=4096 d=16, profiles/r04_k16bench_sweep.txt)
  if (ks > 3) return AESMC_ERR_UNSUPPORTED;
  const uint32_t rs = ks == 1 ? 4 : (ks <= 3 ? 12 : 20);
  const size_t lds = sizeof(float) * (2 * (size_t)kTabF + 2 * (size_t)plan.tile_f + 4 * 2 * 64 * (size_t)rs + 4 * 16);
  if (lds > kLgLdsLimit) return AESMC_ERR_UNSUPPORTED;
  const PhiloxStream ps = philox_stream(seed, offset, threads, rng_state);
  const dim3 grid(lg_persistent_grid((int64_t)plan.items, lds, 2));
  const LgMap p = lg_map(mp), gm = lg_map(mg), q = lg_map(mq);
#define FUSED_ARGS                                                                                                   \
  grid, lds, stream, static_cast<const float *>(xsrc), static_cast<const float *>(y), y_sb, p, gm, q,                  \
      static_cast<const float *>(sp), static_cast<const float *>(sg), static_cast<const float *>(sq),                  \
      static_cast<float *>(out_lw), (uint32_t)K, (uint32_t)B, static_cast<float *>(out_x), anc_idx, flags, ps, plan
  // the maps' weights as scalar operands: rows of W contiguous ([dout, din] row-major, what an nn.Linear holds);
  // AESMC_K16_MAPS=matrix keeps the matrix-core form (a measurement knob; both give the same bits)
  static const bool matrix_only = [] { const char *v = measurement_knob("AESMC_K16_MAPS"); return v != nullptr && v[0] == 'm'; }();
  const auto rows_contiguous = [](const aesmc_affine_map *m) {
    return m->stride_in == 1 && m->stride_out == m->din && (reinterpret_cast<uintptr_t>(m->weight) & 3u) == 0;
  };
  const bool scalar_w = !matrix_only && rows_contiguous(mp) && rows_contiguous(mg) && rows_contiguous(mq);
#ifdef AESMC_LG_FAST_BUILD
  if (ks == 3 && dx == 10 && dy == 10)
    return scalar_w ? fused_launch_g<3, 10, 10, true>(FUSED_ARGS) : fused_launch_g<3, 10, 10, false>(FUSED_ARGS);
  return AESMC_ERR_UNSUPPORTED;
#else
  if (ks == 3 && dx == 10 && dy == 10)
    return scalar_w ? fused_launch_g<3, 10, 10, true>(FUSED_ARGS) : fused_launch_g<3, 10, 10, false>(FUSED_ARGS);
  switch (ks) {
    case 1: return fused_launch_g<1, 0, 0, false>(FUSED_ARGS);
    case 2: return fused_launch_g<2, 0, 0, false>(FUSED_ARGS);
    default: return fused_launch_g<3, 0, 0, false>(FUSED_ARGS);
  }
#endif
#undef FUSED_ARGS
}

}  // namespace aesmc

// Test hooks (not part of the C ABI of include/aesmc_hip.h): 0 = the form is chosen by shape, 1 = the persistent form,
// 2 = one item per workgroup wherever that form applies; and which of the two the last launch through
// launch_affine_propagate_fused asked for (1 also stands for "this file declined": the first form ran).
extern "C" int aesmc_test_last_k16_form(void) { return aesmc::g_fused_last_form; }

extern "C" int aesmc_test_set_k16_form(int form) {
  if (form < 0 || form > 2) return AESMC_ERR_INVALID_ARGUMENT;
  aesmc::g_fused_form = form;
  return AESMC_OK;
}
