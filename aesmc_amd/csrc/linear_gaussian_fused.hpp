// What the two forms of the fused propagation launch share (linear_gaussian_fused.hip: persistent workgroups in two roles;
// linear_gaussian_item.hip: one work item per workgroup): the item geometry ATen's Philox launch imposes, the plan the host
// makes of it, and the multiply-adds with scalar-register weights.
#pragma once
#include "linear_gaussian.hpp"
#include "philox_normal.hpp"

namespace aesmc {

typedef float fz4 __attribute__((ext_vector_type(4)));
typedef fz4 fz4_a4 __attribute__((aligned(4)));      // a 16-byte global access at 4-byte alignment (hardware: unaligned mode)

constexpr uint32_t kRunP = 128;                       // rows per window: two chunks of 64
constexpr uint32_t kTabF = 4 * 2 * 4 * 16;            // floats per table slot: [window][row 0/1][p, q, g, y][16]

struct FusedPlan {
  uint32_t numel;       // B K dx < 2^29
  uint32_t G;           // thread ids of ATen's launch
  uint32_t L, S;        // thread ids per block, Philox calls per lane and item
  uint32_t blocks, items;
  uint32_t blocks_mul, dx_mul, K_mul;      // floor(2^32 / divisor)
  uint32_t tile_f;      // floats per noise tile
};

// v / d for d >= 2 with mul = floor(2^32 / d): the estimate is the quotient or one less
__device__ __forceinline__ uint32_t fused_div(uint32_t v, uint32_t d, uint32_t mul) {
  const uint32_t q = __umulhi(v, mul);
  return (v - q * d) >= d ? q + 1 : q;
}


// acc += w * x with the weight in a SCALAR register (one per wavefront: the launch's maps are the same for every
// particle).  Written as an instruction because the compiler, left to itself, pairs two outputs per v_pk_fma_f32 and
// spends two s_mov per multiply-add on putting their weights side by side.
typedef const float __attribute__((address_space(4))) fused_cfloat;
__device__ __forceinline__ float fused_fmac_s(float acc, float w, float x) {
  asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "s"(w), "v"(x));
  return acc;
}
// five inputs of two chains in one statement (the compiler pads every asm statement's end with an s_nop: one per
// ten multiply-adds instead of one each)
__device__ __forceinline__ void fused_fmac_s5x2(float &a0, float &a1, const float *w0, const float *w1, const float *x) {
  asm("v_fmac_f32 %0, %2, %12\n\tv_fmac_f32 %1, %7, %12\n\t"
      "v_fmac_f32 %0, %3, %13\n\tv_fmac_f32 %1, %8, %13\n\t"
      "v_fmac_f32 %0, %4, %14\n\tv_fmac_f32 %1, %9, %14\n\t"
      "v_fmac_f32 %0, %5, %15\n\tv_fmac_f32 %1, %10, %15\n\t"
      "v_fmac_f32 %0, %6, %16\n\tv_fmac_f32 %1, %11, %16"
      : "+v"(a0), "+v"(a1)
      : "s"(w0[0]), "s"(w0[1]), "s"(w0[2]), "s"(w0[3]), "s"(w0[4]), "s"(w1[0]), "s"(w1[1]), "s"(w1[2]), "s"(w1[3]),
        "s"(w1[4]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]));
}

struct FusedWin {
  uint32_t nf, count, head, b0, k0;      // first particle, particles, elements in front of the first, its batch row, its k
  uint32_t c, t0, tl;                    // the item's trip, first thread id and thread-id count
};


// N inputs of two chains in one statement, N = 1 .. 5 (see fused_fmac_s5x2)
template <int N>
__device__ __forceinline__ void fused_fmac_sx2(float &a0, float &a1, const float *w0, const float *w1, const float *x) {
  static_assert(N >= 1 && N <= 5, "one to five inputs per statement");
  if constexpr (N == 5) {
    fused_fmac_s5x2(a0, a1, w0, w1, x);
  } else if constexpr (N == 4) {
    asm("v_fmac_f32 %0, %2, %10\n\tv_fmac_f32 %1, %6, %10\n\t"
        "v_fmac_f32 %0, %3, %11\n\tv_fmac_f32 %1, %7, %11\n\t"
        "v_fmac_f32 %0, %4, %12\n\tv_fmac_f32 %1, %8, %12\n\t"
        "v_fmac_f32 %0, %5, %13\n\tv_fmac_f32 %1, %9, %13"
        : "+v"(a0), "+v"(a1)
        : "s"(w0[0]), "s"(w0[1]), "s"(w0[2]), "s"(w0[3]), "s"(w1[0]), "s"(w1[1]), "s"(w1[2]), "s"(w1[3]), "v"(x[0]),
          "v"(x[1]), "v"(x[2]), "v"(x[3]));
  } else if constexpr (N == 3) {
    asm("v_fmac_f32 %0, %2, %8\n\tv_fmac_f32 %1, %5, %8\n\t"
        "v_fmac_f32 %0, %3, %9\n\tv_fmac_f32 %1, %6, %9\n\t"
        "v_fmac_f32 %0, %4, %10\n\tv_fmac_f32 %1, %7, %10"
        : "+v"(a0), "+v"(a1)
        : "s"(w0[0]), "s"(w0[1]), "s"(w0[2]), "s"(w1[0]), "s"(w1[1]), "s"(w1[2]), "v"(x[0]), "v"(x[1]), "v"(x[2]));
  } else if constexpr (N == 2) {
    asm("v_fmac_f32 %0, %2, %6\n\tv_fmac_f32 %1, %4, %6\n\t"
        "v_fmac_f32 %0, %3, %7\n\tv_fmac_f32 %1, %5, %7"
        : "+v"(a0), "+v"(a1)
        : "s"(w0[0]), "s"(w0[1]), "s"(w1[0]), "s"(w1[1]), "v"(x[0]), "v"(x[1]));
  } else {
    asm("v_fmac_f32 %0, %2, %4\n\tv_fmac_f32 %1, %3, %4" : "+v"(a0), "+v"(a1) : "s"(w0[0]), "s"(w1[0]), "v"(x[0]));
  }
}
// how many of `rem` remaining inputs the next statement takes: never leaves a single input behind
constexpr int fused_group(int rem) { return rem <= 5 ? rem : (rem == 6 ? 3 : (rem == 7 || rem == 8 ? 4 : 5)); }

// acc0 / acc1 += W[j0][:] . in, W[j0 + 1][:] . in  — two independent chains, inputs ascending (the arithmetic contract)
template <int DIN, int I0 = 0>
__device__ __forceinline__ void fused_pair(float &a0, float &a1, const float *w0, const float *w1, const float *in) {
  if constexpr (I0 < DIN) {
    constexpr int N = fused_group(DIN - I0);
    fused_fmac_sx2<N>(a0, a1, w0 + I0, w1 + I0, in + I0);
    fused_pair<DIN, I0 + N>(a0, a1, w0, w1, in);
  }
}

// acc[j] = offsets[j] + sum_i W[j][i] in[i], j < dout <= DP, one fma chain per output, inputs ascending, started from the
// offset; W's rows contiguous ([dout, DIN] row-major) behind the 64-bit address `base`, fetched by scalar loads two rows
// at a time — the NEXT pair's rows are sent for before this pair's multiply-adds (scalar loads come back out of order,
// so the only wait there is waits for all of them: it must sit behind a block of work).  `offsets`: 16-byte aligned LDS.
template <int DIN, int DP>
__device__ __forceinline__ void fused_chain(unsigned long long base, const float *offsets, uint32_t dout, const float *in,
                                            float (&acc)[DP]) {
  fused_cfloat *W = (fused_cfloat *)base;
#pragma unroll
  for (int v = 0; v < DP / 4; ++v) {
    const fz4 o4 = *reinterpret_cast<const fz4 *>(offsets + 4 * v);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[4 * v + e] = o4[e];
  }
  float wa[2 * DIN], wb[2 * DIN];
  // rows jb and jb + 1 (the last row twice where dout is odd: nothing is read past the weights' end)
  auto load_pair = [&](float (&dst)[2 * DIN], int jb) {
    const uint32_t second = min((uint32_t)jb + 1u, dout - 1u);
#pragma unroll
    for (int e = 0; e < DIN; ++e) dst[e] = W[jb * DIN + e];
#pragma unroll
    for (int e = 0; e < DIN; ++e) dst[DIN + e] = W[second * DIN + e];
  };
  load_pair(wa, 0);
#pragma unroll
  for (int jb = 0; jb < DP; jb += 2) {
    if ((uint32_t)jb >= dout) break;
    float (&cur_w)[2 * DIN] = (jb & 2) ? wb : wa;
    float (&next_w)[2 * DIN] = (jb & 2) ? wa : wb;
    if ((uint32_t)(jb + 2) < dout) load_pair(next_w, jb + 2);
    if ((uint32_t)(jb + 1) < dout) {
      fused_pair<DIN>(acc[jb], acc[jb + 1], cur_w, cur_w + DIN, in);
    } else {
#pragma unroll
      for (int i = 0; i < DIN; ++i) acc[jb] = fused_fmac_s(acc[jb], cur_w[i], in[i]);
    }
  }
}

// ---- two chains per instruction: v_pk_fma_f32 with the weights of outputs j, j + 1 side by side in a scalar pair -------------
// acc = (chain of output j, chain of output j + 1); one instruction advances both by input i: (w[j][i], w[j+1][i]) * x_i.
// The weights come from an interleaved copy of the map — pairs[jp][i] = (W[2 jp][i], W[2 jp + 1][i]), zero where the
// second row does not exist (aesmc_affine_weight_pairs) — so a pair is two consecutive scalar registers; x_i is the low or
// the high half of a vector register pair, picked by the instruction's op_sel bits.  Each half is the IEEE fused
// multiply-add v_fmac_f32 computes, in the same order (inputs ascending, started from the offset): the same bits.
// N inputs (1 .. 4) of one output pair per statement; `w` = the pair's scalar registers from input I0 on, `x` = the inputs
// as register pairs (x[i / 2], half i & 1).
template <int N>
__device__ __forceinline__ void fused_pk_group(lg_f2 &acc, const float *w, const lg_f2 *x) {
  static_assert(N >= 1 && N <= 4, "one to four inputs per statement");
#define FUSED_W(i) "s"(lg_f2{w[2 * (i)], w[2 * (i) + 1]})
  if constexpr (N == 4) {
    asm("v_pk_fma_f32 %0, %1, %5, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %5, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %0, %3, %6, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %4, %6, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "+v"(acc) : FUSED_W(0), FUSED_W(1), FUSED_W(2), FUSED_W(3), "v"(x[0]), "v"(x[1]));
  } else if constexpr (N == 3) {
    asm("v_pk_fma_f32 %0, %1, %4, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %0, %3, %5, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]"
        : "+v"(acc) : FUSED_W(0), FUSED_W(1), FUSED_W(2), "v"(x[0]), "v"(x[1]));
  } else if constexpr (N == 2) {
    asm("v_pk_fma_f32 %0, %1, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "+v"(acc) : FUSED_W(0), FUSED_W(1), "v"(x[0]));
  } else {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : FUSED_W(0), "v"(x[0]));
  }
#undef FUSED_W
}
template <int DIN, int I0 = 0>
__device__ __forceinline__ void fused_pk_pair(lg_f2 &acc, const float *w, const lg_f2 *x) {
  if constexpr (I0 < DIN) {
    constexpr int N = DIN - I0 >= 4 ? 4 : DIN - I0;      // (I0 stays even: groups of four start on a register pair)
    fused_pk_group<N>(acc, w + 2 * I0, x + I0 / 2);
    fused_pk_pair<DIN, I0 + N>(acc, w, x);
  }
}
// fused_chain with both chains of an output pair in one instruction stream: acc[jp] = (output 2 jp, output 2 jp + 1).
// `base`: the map's interleaved copy; the NEXT pair's weights are sent for before this pair's multiply-adds.
template <int DIN, int DP>
__device__ __forceinline__ void fused_chain_pk(unsigned long long base, const float *offsets, uint32_t dout, const lg_f2 *in,
                                               lg_f2 (&acc)[DP / 2]) {
  fused_cfloat *W = (fused_cfloat *)base;
#pragma unroll
  for (int v = 0; v < DP / 4; ++v) {
    const fz4 o4 = *reinterpret_cast<const fz4 *>(offsets + 4 * v);
    acc[2 * v] = lg_f2{o4[0], o4[1]};
    acc[2 * v + 1] = lg_f2{o4[2], o4[3]};
  }
  float wa[2 * DIN], wb[2 * DIN];
#pragma unroll
  for (int e = 0; e < 2 * DIN; ++e) wa[e] = W[e];
#pragma unroll
  for (int jp = 0; jp < DP / 2; ++jp) {
    if ((uint32_t)(2 * jp) >= dout) break;
    float (&cur_w)[2 * DIN] = (jp & 1) ? wb : wa;
    float (&next_w)[2 * DIN] = (jp & 1) ? wa : wb;
    if ((uint32_t)(2 * jp + 2) < dout) {
#pragma unroll
      for (int e = 0; e < 2 * DIN; ++e) next_w[e] = W[(jp + 1) * 2 * DIN + e];
    }
    fused_pk_pair<DIN>(acc[jp], cur_w, in);
  }
}

// window `i` of work item `item`: first particle, particle count, elements in front of the first particle
__device__ __forceinline__ FusedWin fused_window(const FusedPlan &plan, uint32_t item, uint32_t i, uint32_t dx, uint32_t K) {
  FusedWin v;
  const uint32_t G = plan.G;
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
}

// The host's plan of a launch (both forms): AESMC_OK, or AESMC_ERR_UNSUPPORTED for what the item geometry does not cover.
static inline int fused_make_plan(FusedPlan &plan, int64_t B, int64_t K, int64_t dx, int64_t threads) {
  const uint64_t numel = (uint64_t)B * (uint64_t)K * (uint64_t)dx;
  if (dx < 2 || dx > 16 || K < (int64_t)kRunP || K >= (1ll << 24) || B >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  // 32-bit element arithmetic throughout (window bounds run up to numel + 4 G + L; byte offsets up to 8 N)
  if (numel >= (1ull << 29) || (uint64_t)threads >= (1ull << 24)) return AESMC_ERR_UNSUPPORTED;
  plan.numel = (uint32_t)numel;
  plan.G = (uint32_t)threads;
  plan.S = (uint32_t)(((uint64_t)(kRunP + 1) * dx - 1) / 256);      // 256 S <= (128 + 1) dx - 1: at most 128 particles per window
  if (plan.S < 1) return AESMC_ERR_UNSUPPORTED;
  plan.L = plan.S * 256u - ((uint32_t)dx - 1);
  plan.blocks = (uint32_t)(((uint64_t)threads + plan.L - 1) / plan.L);
  const uint64_t trips = (numel + 4ull * (uint64_t)threads - 1) / (4ull * (uint64_t)threads);
  const uint64_t items = trips * plan.blocks;
  if (items > 0x3fffffffull || plan.blocks < 2) return AESMC_ERR_UNSUPPORTED;
  plan.items = (uint32_t)items;
  plan.blocks_mul = (uint32_t)((1ull << 32) / plan.blocks);
  plan.dx_mul = (uint32_t)((1ull << 32) / (uint64_t)dx);
  plan.K_mul = (uint32_t)((1ull << 32) / (uint64_t)K);
  plan.tile_f = (uint32_t)((4 * kRunP * dx + 4 + 3) & ~3ull);      // + the spare word unplaced normals go to
  return AESMC_OK;
}

// The maps' interleaved copies (aesmc_affine_weight_pairs; linear_gaussian_item.hip): one region of kPairFloats floats per
// map in the order transition, emission, proposal.
constexpr int kPairFloats = (kLgMaxDim / 2) * kLgMaxDim * 2;
// Behind the three regions of a buffer of aesmc_affine_weight_pairs_floats() values: the launch's constants of the three
// densities — (2 s^2, d (log s + log(2 pi) / 2)) for transition, emission, proposal, the very expressions the propagating
// kernels evaluate — then a tag: fused_consts_tag(dx, dy) when aesmc_affine_weight_pairs_scaled wrote them for these
// extents, 0 when nobody did (aesmc_affine_weight_pairs).  The item form reads them instead of taking three logarithms
// per wavefront (36 of its 660 vector instructions, the pipe it saturates).
constexpr int kPairConsts = 8;
__host__ __device__ constexpr uint32_t fused_consts_tag(uint32_t dx, uint32_t dy) { return 0x5c000000u | (dy << 8) | dx; }
// `sp` .. `sq`: the three scales (device, one float32 each), or all nullptr; `tail`: `out` has the kPairConsts values'
// room behind the regions (they are written — the tag 0 without scales); without it nothing behind the regions is touched
int launch_affine_weight_pairs(const LgMap &mp, const LgMap &mg, const LgMap &mq, float *out, hipStream_t stream,
                               const float *sp = nullptr, const float *sg = nullptr, const float *sq = nullptr,
                               bool tail = false);

// which form `aesmc_affine_normal_propagate_drawn` launches: 0 by shape, 1 the persistent form, 2 one item per workgroup
// (AESMC_K16_FORM=persistent / item in the environment, or the test hook aesmc_test_set_k16_form)
extern int g_fused_form;

}  // namespace aesmc
