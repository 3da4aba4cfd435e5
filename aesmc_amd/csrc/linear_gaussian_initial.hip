// K20: the FIRST step of a run whose proposal and prior do not look at a latent and whose emission is linear-Gaussian in
// the latent being drawn (aesmc/inference.py:79-98 with the reference's own model, test/models/lgssm.py: `initial()` a
// Normal, the time-0 proposal a Normal of the observation, BATCH_EXPANDED, the emission Normal(C x_0 + g, s_g)):
//
//   x_0[b,k,:] = mu_q[b,:] + eps[k,b,:] * s_q[b,:]                      (state.py:98, :102-103: rsample((K,)) transposed)
//   lw[b,k]    = (sum_j log N(x_0j; mu_p, s_p) + sum_j log N(y_bj; (C x_0 + g)_j, s_g)) - sum_j log N(x_0j; mu_q, s_q)
//
// in one launch, where the library used to take three — K6 (the transposed draw), K8 (the emission's location as a
// [B,K,dy] tensor, written to be read once) and K5 (three log-densities with their operands fetched through general
// strides) — 515 us of an 11.4 ms ELBO at the north-star shape for one timestep of a hundred.  Every value has the
// bits those three give: the draw is K6's `mu + eps * s` (the product rounded before the sum), the location K8's fma
// chain (inputs ascending, started from the offset), an element's log-density K5's
// (-(d d)) / (2 s s) - log s - log(2 pi) / 2, summed over j ascending from zero, combined as (p + g) - q
// (tests/test_gpu_initial_step.py holds the launch to the three, bit for bit).
//
// The noise comes in the reference's order — [K, B, d], what `Normal(loc [B,d], s).rsample((K,))` draws — and the draw
// leaves as [B, K, d]: a workgroup owns 16 batch rows x 16 particles (one per lane; 32 particles per row: 190 against 178 us), reads its noise in runs along (b, j) — 16-byte pieces at 4-byte alignment —, parks the
// draws in LDS, weighs one particle per lane out of there, and writes x_0 and the log-weights in runs along (k, j).
// What depends on the batch row only — the proposal's and the prior's location and scale, the observation, the
// emission's offset, and per scale 2 s^2 and log s — is tabulated once per workgroup (16 rows x 16 columns = its 256
// lanes), so scalar, per-column and per-row parameters cost the same and no logarithm is taken per particle.
#include "linear_gaussian.hpp"

namespace aesmc {

constexpr uint32_t kInB = 16, kInK = 16, kInBlock = 256;
static_assert(kInB * 16 == kInBlock, "one lane per (batch row, column) of the tables");

typedef float in_f4 __attribute__((ext_vector_type(4)));
typedef in_f4 in_f4_a4 __attribute__((aligned(4)));      // a 16-byte global access at 4-byte alignment (hardware: unaligned mode)

// an operand that is constant along the particles: value(b, j) = ptr[b * sb + j * sd] (0 strides: broadcast)
struct InView {
  const float *ptr;
  int64_t sb, sd;
};
enum { kInMuQ, kInSQ, kInTvQ, kInLgQ, kInMuP, kInTvP, kInLgP, kInY, kInOffG, kInTvG, kInLgG, kInTables };

template <int DX> struct InLds {
  static constexpr uint32_t pitch = kInK * DX + 1;      // floats per batch row of the parked draws (+1: rows on distinct banks)
  static constexpr uint32_t park = kInB * pitch, tab = kInTables * kInB * 16, wts = 16 * DX, lw = kInB * (kInK + 1);
  static constexpr size_t bytes = sizeof(float) * (park + tab + wts + lw);
};

template <int DX>
__global__ __launch_bounds__(kInBlock) void affine_initial_step_kernel(const float *__restrict__ eps, InView mu_q, InView s_q,
                                                                      InView mu_p, InView s_p, InView obs, InView s_g,
                                                                      LgMap mg, float *__restrict__ out_x,
                                                                      float *__restrict__ out_lw, uint32_t B, uint32_t K) {
  extern __shared__ __attribute__((aligned(16))) float in_smem[];
  constexpr uint32_t pitch = InLds<DX>::pitch;
  float *park = in_smem;                       // [kInB][pitch]: x_0 of the tile, a batch row's particles end to end
  float *tab = park + InLds<DX>::park;         // [kInTables][kInB][16]
  float *wts = tab + InLds<DX>::tab;           // [dy][DX]: the emission's weights, rows end to end
  float *lwt = wts + InLds<DX>::wts;           // [kInB][kInK + 1]
  const uint32_t dy = (uint32_t)mg.dout;
  const uint32_t b0 = blockIdx.x * kInB, k0 = blockIdx.y * kInK;
  const uint32_t nb = min(kInB, B - b0), nk = min(kInK, K - k0);
  const uint32_t tid = threadIdx.x;
  const float half_log_2pi = LgConst<float>::half_log_2pi();

  // ---- the tables: lane = (batch row, column) ----------------------------------------------------------------------------
  {
    const uint32_t bb = tid >> 4, j = tid & 15u;
    float *t = tab + bb * 16 + j;
    if (bb < nb) {
      const int64_t b = (int64_t)b0 + bb;
      if (j < (uint32_t)DX) {
        const float sq = s_q.ptr[b * s_q.sb + (int64_t)j * s_q.sd], sp = s_p.ptr[b * s_p.sb + (int64_t)j * s_p.sd];
        t[kInMuQ * kInBlock] = mu_q.ptr[b * mu_q.sb + (int64_t)j * mu_q.sd];
        t[kInSQ * kInBlock] = sq;
        t[kInTvQ * kInBlock] = 2.0f * (sq * sq);
        t[kInLgQ * kInBlock] = Num<float>::log(sq);
        t[kInMuP * kInBlock] = mu_p.ptr[b * mu_p.sb + (int64_t)j * mu_p.sd];
        t[kInTvP * kInBlock] = 2.0f * (sp * sp);
        t[kInLgP * kInBlock] = Num<float>::log(sp);
      }
      if (j < dy) {
        const float sg = s_g.ptr[b * s_g.sb + (int64_t)j * s_g.sd];
        t[kInY * kInBlock] = obs.ptr[b * obs.sb + (int64_t)j * obs.sd];
        t[kInOffG * kInBlock] = mg.off != nullptr ? static_cast<const float *>(mg.off)[b * mg.off_sb + j] : 0.0f;
        t[kInTvG * kInBlock] = 2.0f * (sg * sg);
        t[kInLgG * kInBlock] = Num<float>::log(sg);
      }
    }
    const float *w = static_cast<const float *>(mg.w);
    for (uint32_t e = tid; e < dy * (uint32_t)DX; e += kInBlock) {
      const uint32_t jo = e / (uint32_t)DX, i = e - jo * (uint32_t)DX;
      wts[e] = w[(int64_t)jo * mg.sj + (int64_t)i * mg.si];
    }
  }
  __syncthreads();

  // ---- the draws: the noise in runs along (b, j), K6's arithmetic ------------------------------------------------------------
  auto draw = [&](uint32_t run) {      // run = nb * DX (a constant for a whole tile: the divisions are multiplications)
    const uint32_t total = nk * run;
    for (uint32_t t = tid; t < total; t += kInBlock) {
      const uint32_t kk = t / run, rest = t - kk * run;
      const uint32_t bb = rest / (uint32_t)DX, j = rest - bb * (uint32_t)DX;
      const float noise = eps[((k0 + kk) * B + b0) * (uint32_t)DX + rest];      // (32-bit: the host admits B K d < 2^31)
      park[bb * pitch + kk * (uint32_t)DX + j] = tab[kInMuQ * kInBlock + bb * 16 + j] + noise * tab[kInSQ * kInBlock + bb * 16 + j];
    }
  };
  if (nb == kInB) {      // a whole tile: 16 DX values per run = 4 DX pieces of 16 bytes (at 4-byte alignment: any extent, any B)
    constexpr uint32_t pieces = kInB * (uint32_t)DX / 4u;
    for (uint32_t t = tid; t < nk * pieces; t += kInBlock) {
      const uint32_t kk = t / pieces, rest = 4u * (t - kk * pieces);
      const in_f4 noise = *reinterpret_cast<const in_f4_a4 *>(eps + (((k0 + kk) * B + b0) * (uint32_t)DX + rest));
      uint32_t bb = rest / (uint32_t)DX, j = rest - bb * (uint32_t)DX;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        park[bb * pitch + kk * (uint32_t)DX + j] = tab[kInMuQ * kInBlock + bb * 16 + j] + noise[r] * tab[kInSQ * kInBlock + bb * 16 + j];
        if (++j == (uint32_t)DX) {
          j = 0;
          ++bb;
        }
      }
    }
  } else {
    draw(nb * (uint32_t)DX);
  }
  __syncthreads();

  // ---- one particle per lane: the emission's location (K8's chain), the three sums (K5's terms, j ascending) ---------------
#pragma unroll 1
  for (uint32_t p = tid; p < kInB * kInK; p += kInBlock) {
    const uint32_t kk = p & (kInK - 1u), bb = p / kInK;
    if (bb >= nb || kk >= nk) continue;
    const float *row = park + bb * pitch + kk * (uint32_t)DX;
    const float *tb = tab + bb * 16;
    float x[DX];
#pragma unroll
    for (int i = 0; i < DX; ++i) x[i] = row[i];
    float sum_p = 0.0f, sum_q = 0.0f, sum_g = 0.0f;
#pragma unroll
    for (int j = 0; j < DX; ++j) {
      const float dp = x[j] - tb[kInMuP * kInBlock + j], dq = x[j] - tb[kInMuQ * kInBlock + j];
      sum_p += (-(dp * dp)) / tb[kInTvP * kInBlock + j] - tb[kInLgP * kInBlock + j] - half_log_2pi;
      sum_q += (-(dq * dq)) / tb[kInTvQ * kInBlock + j] - tb[kInLgQ * kInBlock + j] - half_log_2pi;
    }
#pragma unroll
    for (int jo = 0; jo < 16; ++jo) {
      if ((uint32_t)jo < dy) {      // (uniform)
        float loc = tb[kInOffG * kInBlock + jo];
#pragma unroll
        for (int i = 0; i < DX; ++i) loc = fma_t(wts[jo * DX + i], x[i], loc);
        const float dg = tb[kInY * kInBlock + jo] - loc;
        sum_g += (-(dg * dg)) / tb[kInTvG * kInBlock + jo] - tb[kInLgG * kInBlock + jo] - half_log_2pi;
      }
    }
    lwt[bb * (kInK + 1u) + kk] = (sum_p + sum_g) - sum_q;
  }
  __syncthreads();

  // ---- out: a batch row's particles are one contiguous run of x_0 and one of the log-weights --------------------------------
  {
    const uint32_t run = nk * (uint32_t)DX, total = nb * run;
    if (nk == kInK) {      // whole runs: 16-byte stores (at 4-byte alignment)
      constexpr uint32_t pieces = kInK * (uint32_t)DX / 4u;
      for (uint32_t t = tid; t < nb * pieces; t += kInBlock) {
        const uint32_t bb = t / pieces, rest = 4u * (t - bb * pieces);
        const float *from = park + bb * pitch + rest;
        *reinterpret_cast<in_f4_a4 *>(out_x + (((b0 + bb) * K + k0) * (uint32_t)DX + rest)) = in_f4{from[0], from[1], from[2], from[3]};
      }
    } else {
      for (uint32_t t = tid; t < total; t += kInBlock) {
        const uint32_t bb = t / run, rest = t - bb * run;
        out_x[((b0 + bb) * K + k0) * (uint32_t)DX + rest] = park[bb * pitch + rest];
      }
    }
    for (uint32_t t = tid; t < nb * kInK; t += kInBlock) {
      const uint32_t bb = t / kInK, kk = t & (kInK - 1u);
      if (kk < nk) out_lw[(b0 + bb) * K + k0 + kk] = lwt[bb * (kInK + 1u) + kk];
    }
  }
}

template <int DX>
static int initial_launch(dim3 grid, hipStream_t stream, const float *eps, const InView *v, const LgMap &mg, float *out_x,
                          float *out_lw, uint32_t B, uint32_t K) {
  hipLaunchKernelGGL(affine_initial_step_kernel<DX>, grid, dim3(kInBlock), InLds<DX>::bytes, stream, eps, v[0], v[1], v[2], v[3],
                     v[4], v[5], mg, out_x, out_lw, B, K);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

}  // namespace aesmc

using namespace aesmc;

extern "C" int aesmc_affine_normal_initial_step(const void *eps, const aesmc_view3 *loc_q, const aesmc_view3 *scale_q,
                                                const aesmc_view3 *loc_p, const aesmc_view3 *scale_p, const aesmc_view3 *y,
                                                const aesmc_affine_map *emission, const aesmc_view3 *scale_g, void *out_x,
                                                void *out_lw, int64_t B, int64_t K, void *stream) {
  if (eps == nullptr || out_x == nullptr || out_lw == nullptr || loc_q == nullptr || scale_q == nullptr || loc_p == nullptr ||
      scale_p == nullptr || y == nullptr || scale_g == nullptr || emission == nullptr)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (B < 0 || K < 0) return AESMC_ERR_INVALID_ARGUMENT;
  if (!lg_map_ok(emission)) return AESMC_ERR_UNSUPPORTED;
  const aesmc_view3 *views[6] = {loc_q, scale_q, loc_p, scale_p, y, scale_g};
  InView v[6];
  for (int i = 0; i < 6; ++i) {
    if (views[i]->ptr == nullptr) return AESMC_ERR_INVALID_ARGUMENT;
    if (K > 1 && views[i]->stride_k != 0) return AESMC_ERR_UNSUPPORTED;      // not constant along the particles: K6 + K8 + K5
    v[i].ptr = static_cast<const float *>(views[i]->ptr);
    v[i].sb = views[i]->stride_b;
    v[i].sd = views[i]->stride_d;
  }
  if (B == 0 || K == 0) return AESMC_OK;
  const int64_t tiles_b = (B + kInB - 1) / kInB, tiles_k = (K + kInK - 1) / kInK;
  if (tiles_k > 65535 || tiles_b > 0x7fffffff || B >= (1ll << 31) || K >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  if (B * K * std::max<int64_t>(emission->din, emission->dout) >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;      // 32-bit element arithmetic
  for (int i = 0; i < 6; ++i)      // ... and the per-row operands' rows within it
    if (B * std::max<int64_t>(v[i].sb < 0 ? -v[i].sb : v[i].sb, 1) >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)tiles_b, (unsigned)tiles_k);
  const LgMap mg = lg_map(emission);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float *e = static_cast<const float *>(eps);
  float *ox = static_cast<float *>(out_x), *ol = static_cast<float *>(out_lw);
#define INITIAL_CASE(D) \
  case D: return initial_launch<D>(grid, s, e, v, mg, ox, ol, (uint32_t)B, (uint32_t)K);
  switch (emission->din) {
    INITIAL_CASE(1) INITIAL_CASE(2) INITIAL_CASE(3) INITIAL_CASE(4) INITIAL_CASE(5) INITIAL_CASE(6) INITIAL_CASE(7)
    INITIAL_CASE(8) INITIAL_CASE(9) INITIAL_CASE(10) INITIAL_CASE(11) INITIAL_CASE(12) INITIAL_CASE(13) INITIAL_CASE(14)
    INITIAL_CASE(15) INITIAL_CASE(16)
    default: return AESMC_ERR_UNSUPPORTED;
  }
#undef INITIAL_CASE
}
