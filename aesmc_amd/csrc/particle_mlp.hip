// K13 / K13b: a learned proposal net over the particles and its backward — the two-layer tanh MLP of [x_{t-1}, y_t] that
// BASELINE.json's configs[3] (nonlinear state-space model with a learned proposal net) evaluates once per timestep
// (aesmc/inference.py:102-106: `proposal(previous_latents=..., observations=...)`; the reference's own proposal,
// test/models/lgssm.py:66-77, is its one-layer case):
//
//   out[b,k,:] = b2 + W2 tanh( c1[b,:] + W1 x[b,k,:] )          W1 [H, din], W2 [dout, H], din, dout <= 16, H <= 64
//
// (the y_t columns of the first layer and its bias arrive as the per-row offset c1).  Through PyTorch this is a
// concatenation, two GEMMs and a tanh with [B,K,H] round trips through HBM between them — 173 us of a 260 us timestep at
// B = 128, K = 4096, d = 10, H = 64 (profiles/r06b_rocprof_c4nl_before_k13.csv) — and, backward, two weight-gradient GEMMs that
// contract over B K = 524 288 particles onto a handful of workgroups (hipBLASLt's default picks: 898 + 822 us).
//
//   K13   forward: the hidden layer lives in registers (one particle per lane, 16 hidden units at a time), both weight
//         matrices in LDS.  Chains as everywhere in this library: fused multiply-adds, inputs ascending, started from the
//         offset / bias; tanh is the device library's (the one torch.tanh calls).
//   K13b  backward, RECOMPUTING the hidden layer from x and c1 (nothing of [B,K,H] is ever stored): per particle
//           dh = (W2^T g) (1 - h^2),   grad_x = W1^T dh
//         and, contracted over the particles on the matrix cores (v_mfma 16x16x4, exact fma accumulation),
//           grad_W2 = sum g (x) h,     grad_W1 = sum dh (x) x,     grad_c1[b] = sum_k dh   (a column of ones beside x)
//         A wavefront owns 64 particles from their rows' arrival to the gradient's store and keeps its own accumulators
//         over all its tiles; what the matrix cores need transposed goes through an LDS area only that wavefront touches.
//         Each wavefront leaves its 16 x 16 partials as records that the binder adds (a fixed order: reproducible); the
//         per-row sums are left per tile (K a multiple of 256: a tile lies inside one batch row).
#include "linear_gaussian.hpp"
#include "linear_gaussian_backward.hpp"

namespace aesmc {

template <typename T> __device__ __forceinline__ T mlp_tanh(T x);
template <> __device__ __forceinline__ float mlp_tanh<float>(float x) { return ::tanhf(x); }
template <> __device__ __forceinline__ double mlp_tanh<double>(double x) { return ::tanh(x); }

// The backward RECOMPUTES the hidden layer only to form 1 - h^2 and the outer products: float32 there takes
// tanh(v) = 1 - 2 / (1 + e^{2 v}) on the hardware's exp2 / reciprocal (absolute error ~1e-7, saturating correctly at both
// ends) instead of the device library's 40-instruction tanhf — two thirds of the kernel's vector instructions were that
// call (142 us -> see profiles/README.md).  The FORWARD keeps the library's tanhf: its values are the ones torch.tanh gives.
template <typename T> __device__ __forceinline__ T mlp_tanh_backward(T x) { return mlp_tanh<T>(x); }
template <> __device__ __forceinline__ float mlp_tanh_backward<float>(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);      // e^{2 x}
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}

constexpr int kMlpMaxHidden = 64;

template <typename T, int DP>
__global__ __launch_bounds__(kLgBlock) void particle_mlp_kernel(const T *__restrict__ x, LgMap m1, LgMap m2,
                                                                 T *__restrict__ out, int64_t N, uint32_t K,
                                                                 uint32_t HP) {
  // HP: the hidden width rounded up to a multiple of 16; the hidden layer is taken 16 units at a time
  // (first-layer chains, tanh, their share of the second-layer chains), so a lane holds 16 hidden values
  // and the output accumulators, whatever H is
  extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];
  constexpr uint32_t TP = kLgBlock;
  const uint32_t din = m1.din, hid = m1.dout, dout = m2.dout;
  T *w1 = reinterpret_cast<T *>(mlp_smem);        // [DP][HP]: w1[i * HP + h] = W1[h][i]
  T *w2 = w1 + DP * HP;                           // [HP][DP]: w2[h * DP + o] = W2[o][h]
  T *tab = w2 + HP * DP;                          // [kLgRowsMax][HP]: the rows' first-layer offsets
  T *tx = tab + kLgRowsMax * HP;
  const LgLayout lx = lg_layout<T>(din), lo = lg_layout<T>(dout);
  T *to = tx + (TP * lx.rs + 16);
  {
    const T *a = reinterpret_cast<const T *>(m1.w), *b = reinterpret_cast<const T *>(m2.w);
#pragma unroll 1
    for (uint32_t e = threadIdx.x; e < DP * HP; e += kLgBlock) {
      const uint32_t i = e / HP, h = e - i * HP;
      w1[e] = (h < hid && i < din) ? a[(int64_t)h * m1.sj + (int64_t)i * m1.si] : T(0);
    }
#pragma unroll 1
    for (uint32_t e = threadIdx.x; e < HP * DP; e += kLgBlock) {
      const uint32_t h = e / DP, o = e - h * DP;
      w2[e] = (h < hid && o < dout) ? b[(int64_t)o * m2.sj + (int64_t)h * m2.si] : T(0);
    }
  }
  const T *off1 = reinterpret_cast<const T *>(m1.off), *off2 = reinterpret_cast<const T *>(m2.off);
  const int64_t tiles = (N + TP - 1) / TP;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t n0 = tile * TP;
    const uint32_t np = (uint32_t)min((int64_t)TP, N - n0);
    lg_stage_rows(x + n0 * din, np * din, tx, lx, 0);
    const uint32_t b0 = (uint32_t)(n0 / K), nrows = (uint32_t)((n0 + np - 1) / K) - b0 + 1;
#pragma unroll 1
    for (uint32_t idx = threadIdx.x; idx < nrows * HP; idx += kLgBlock) {     // the host guarantees nrows <= kLgRowsMax
      const uint32_t row = idx / HP, h = idx - row * HP;
      tab[idx] = (off1 != nullptr && h < hid) ? off1[(int64_t)(b0 + row) * m1.off_sb + h] : T(0);
    }
    __syncthreads();
    const uint32_t q = threadIdx.x;
    const bool live = q < np;
    const uint32_t p = live ? q : 0u;
    const uint32_t k0 = (uint32_t)(n0 - (int64_t)b0 * K);
    const T *orow = tab + ((k0 + p) / K) * HP;
    const uint32_t base = p * lx.rs;
    T acc[DP];
#pragma unroll
    for (int o = 0; o < DP; ++o) acc[o] = (off2 != nullptr && (uint32_t)o < dout) ? off2[o] : T(0);
#pragma unroll 1
    for (uint32_t c = 0; c < HP; c += 16) {
      T hidden[16];
#pragma unroll
      for (int h = 0; h < 16; ++h) hidden[h] = orow[c + h];
#pragma unroll 2
      for (uint32_t i = 0; i < din; ++i) {
        const T xv = tx[base + i];
        const T *w = w1 + i * HP + c;
#pragma unroll
        for (int h = 0; h < 16; ++h) hidden[h] = fma_t(w[h], xv, hidden[h]);
      }
#pragma unroll
      for (int h = 0; h < 16; ++h) hidden[h] = mlp_tanh<T>(hidden[h]);
#pragma unroll
      for (int h = 0; h < 16; ++h) {
        const T *w = w2 + (c + h) * DP;
#pragma unroll
        for (int o = 0; o < DP; ++o) acc[o] = fma_t(w[o], hidden[h], acc[o]);
      }
    }
    if (live) {
#pragma unroll
      for (int o = 0; o < DP; ++o)
        if ((uint32_t)o < dout) to[p * lo.rs + o] = acc[o];
    }
    __syncthreads();
    lg_store_rows(out + n0 * dout, np * dout, to, lo);
    __syncthreads();
  }
}

template <typename T>
static int launch_particle_mlp(const void *x, const aesmc_affine_map *m1, const aesmc_affine_map *m2, void *out,
                               int64_t B, int64_t K, hipStream_t stream) {
  const int64_t N = B * K;
  const int64_t din = m1->din, hid = m1->dout, dout = m2->dout;
  const int dp = lg_pad_dim(std::max(din, dout));
  const uint32_t hp = (uint32_t)((hid + 15) / 16 * 16);
  if (lg_rows_spanned(kLgBlock, K) > kLgRowsMax) return AESMC_ERR_UNSUPPORTED;   // fewer than ~43 particles per row
  const size_t lds = sizeof(T) * (2 * (size_t)dp * hp + (size_t)kLgRowsMax * hp + lg_tile_elems<T>(kLgBlock, din) +
                                  lg_tile_elems<T>(kLgBlock, dout));
  if (lds > kLgLdsLimit) return AESMC_ERR_UNSUPPORTED;
  const int64_t tiles = (N + kLgBlock - 1) / kLgBlock;
  const unsigned grid = lg_persistent_grid(tiles, lds, 8);
  const T *xp = static_cast<const T *>(x);
  T *op = static_cast<T *>(out);
#define MLP_CASE(DP_)                                                                                               \
  do {                                                                                                              \
    if (lds > 64 * 1024)                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&particle_mlp_kernel<T, DP_>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
    hipLaunchKernelGGL((particle_mlp_kernel<T, DP_>), dim3(grid), dim3(kLgBlock), lds, stream, xp, lg_map(m1),      \
                       lg_map(m2), op, N, (uint32_t)K, hp);                                                         \
  } while (0)
  switch (dp) {
    case 4: MLP_CASE(4); break;
    case 8: MLP_CASE(8); break;
    case 10: MLP_CASE(10); break;
    case 12: MLP_CASE(12); break;
    default: MLP_CASE(16); break;
  }
#undef MLP_CASE
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

// ---- K13b ----------------------------------------------------------------------------------------------------------------
constexpr int kMlpStride = 20;      // elements between two particles' rows in a wavefront's transposing areas (16 + a pad)
constexpr int kMlpWaves = kLgBlock / 64;

struct MlpBackwardArgs {
  const void *x, *g;       // [N, din], [N, dout]
  LgMap m1, m2;
  void *gx;                // [N, din] or nullptr
  void *rec_w1;            // [records][HP / 16][256]: grad_W1's 16 (hidden) x 16 (input; column 15: the rows' sums) partials
  void *rec_w2;            // [records][HP / 16][256]: grad_W2's 16 (output) x 16 (hidden) partials
  void *rows;              // [tiles][kMlpWaves][HP]: sum over a wavefront's 64 particles of dh (or nullptr)
  int64_t N;
  uint32_t K, HP;
};

// The transposing areas belong to ONE wavefront: what it wrote must have landed before its lanes read each other's rows
// (LDS serves a wavefront's accesses in order; the wait makes that explicit and keeps the compiler from moving accesses
// across it) — no workgroup barrier, the four wavefronts of a workgroup never wait for each other inside a tile.
__device__ __forceinline__ void mlp_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T, int DP>
__global__ __launch_bounds__(kLgBlock, sizeof(T) == 4 ? 2 : 1) void particle_mlp_backward_kernel(MlpBackwardArgs a) {      // (float32: two workgroups per CU — 256 registers a wavefront)
  extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];
  using Acc = typename Mfma<T>::Acc;
  const uint32_t HP = a.HP, din = a.m1.din, hid = a.m1.dout, dout = a.m2.dout;
  const uint32_t chunks = HP / 16;
  T *w1 = reinterpret_cast<T *>(mlp_smem);        // [DP][HP]: w1[i * HP + h] = W1[h][i]
  T *w2 = w1 + DP * HP;                           // [HP][DP]: w2[h * DP + o] = W2[o][h]
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  T *area = w2 + HP * DP + wave * (3 * 64 * kMlpStride + kMlpMaxHidden);      // this wavefront's own
  T *buf_g = area, *buf_x = buf_g + 64 * kMlpStride, *buf_t = buf_x + 64 * kMlpStride, *tab = buf_t + 64 * kMlpStride;
  {
    const T *wa = reinterpret_cast<const T *>(a.m1.w), *wb = reinterpret_cast<const T *>(a.m2.w);
#pragma unroll 1
    for (uint32_t e = threadIdx.x; e < DP * HP; e += kLgBlock) {
      const uint32_t i = e / HP, h = e - i * HP;
      w1[e] = (h < hid && i < din) ? wa[(int64_t)h * a.m1.sj + (int64_t)i * a.m1.si] : T(0);
    }
#pragma unroll 1
    for (uint32_t e = threadIdx.x; e < HP * DP; e += kLgBlock) {
      const uint32_t h = e / DP, o = e - h * DP;
      w2[e] = (h < hid && o < dout) ? wb[(int64_t)o * a.m2.sj + (int64_t)h * a.m2.si] : T(0);
    }
  }
  __syncthreads();
  const T *x = reinterpret_cast<const T *>(a.x), *g = reinterpret_cast<const T *>(a.g);
  const T *off1 = reinterpret_cast<const T *>(a.m1.off);
  T *gx_out = reinterpret_cast<T *>(a.gx);
  T *rows_out = reinterpret_cast<T *>(a.rows);
  Acc acc1[kMlpMaxHidden / 16], acc2[kMlpMaxHidden / 16];
#pragma unroll
  for (int c = 0; c < kMlpMaxHidden / 16; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc1[c][r] = acc2[c][r] = T(0);
  const uint32_t m = lane & 15u, kq = lane >> 4;      // the lane's row / column of a matrix operand, its k among four
  const int64_t tiles = a.N / kLgBlock;                // (the host: K, hence N, a multiple of 256)
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t n = tile * kLgBlock + threadIdx.x;
    const uint32_t b = (uint32_t)((tile * kLgBlock) / a.K);
    T xv[DP], gv[DP], gxv[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) {
      xv[i] = (uint32_t)i < din ? x[n * din + i] : T(0);
      gv[i] = (uint32_t)i < dout ? g[n * dout + i] : T(0);
      gxv[i] = T(0);
    }
    // the wavefront's operands as the matrix cores read them: row = particle, 16 columns (g: outputs, zero beyond dout;
    // x: inputs, zero beyond din, ONE in column 15 — the rows' sums ride in the last column of grad_W1's partials)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      buf_g[lane * kMlpStride + i] = i < DP ? gv[i < DP ? i : 0] : T(0);
      buf_x[lane * kMlpStride + i] = i == 15 ? T(1) : (i < DP ? xv[i < DP ? i : 0] : T(0));
    }
    if (lane < HP) tab[lane] = (off1 != nullptr && lane < hid) ? off1[(int64_t)b * a.m1.off_sb + lane] : T(0);
    mlp_wave_sync();
#pragma unroll
    for (int c = 0; c < kMlpMaxHidden / 16; ++c) {
      if ((uint32_t)c < chunks) {      // (uniform)
        T hv[16], dh[16];
#pragma unroll
        for (int h = 0; h < 16; ++h) hv[h] = tab[16 * c + h];
#pragma unroll
        for (int i = 0; i < DP; ++i) {
          const T *w = w1 + i * HP + 16 * c;
#pragma unroll
          for (int h = 0; h < 16; ++h) hv[h] = fma_t(w[h], xv[i], hv[h]);
        }
#pragma unroll
        for (int h = 0; h < 16; ++h) {
          hv[h] = mlp_tanh_backward<T>(hv[h]);
          const T *w = w2 + (16 * c + h) * DP;
          T t = T(0);
#pragma unroll
          for (int o = 0; o < DP; ++o) t = fma_t(w[o], gv[o], t);
          dh[h] = t * (T(1) - hv[h] * hv[h]);
        }
#pragma unroll
        for (int i = 0; i < DP; ++i) {
          const T *w = w1 + i * HP + 16 * c;
#pragma unroll
          for (int h = 0; h < 16; ++h) gxv[i] = fma_t(w[h], dh[h], gxv[i]);
        }
        // grad_W2's chunk: D[o][h] += sum over the 64 particles of g[p][o] h[p][h]
#pragma unroll
        for (int h = 0; h < 16; ++h) buf_t[lane * kMlpStride + h] = hv[h];
        mlp_wave_sync();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const uint32_t p = 4 * s + kq;
          acc2[c] = Mfma<T>::fma(buf_g[p * kMlpStride + m], buf_t[p * kMlpStride + m], acc2[c]);
        }
        mlp_wave_sync();
        // grad_W1's chunk: D[h][i] += sum over the particles of dh[p][h] x[p][i]   (i = 15: the ones)
#pragma unroll
        for (int h = 0; h < 16; ++h) buf_t[lane * kMlpStride + h] = dh[h];
        mlp_wave_sync();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const uint32_t p = 4 * s + kq;
          acc1[c] = Mfma<T>::fma(buf_t[p * kMlpStride + m], buf_x[p * kMlpStride + m], acc1[c]);
        }
        mlp_wave_sync();
        __builtin_amdgcn_sched_barrier(0);      // (a chunk's weights and hidden values are not sent for during the chunk before)
      }
    }
    if (gx_out != nullptr) {
#pragma unroll
      for (int i = 0; i < DP; ++i)
        if ((uint32_t)i < din) gx_out[n * din + i] = gxv[i];
    }
    // the rows' sums so far are column 15 of grad_W1's partials: handed out per tile, cleared (each wavefront its own)
    if (m == 15u) {
#pragma unroll
      for (int c = 0; c < kMlpMaxHidden / 16; ++c) {
        if ((uint32_t)c < chunks) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (rows_out != nullptr)
              rows_out[((int64_t)tile * kMlpWaves + wave) * HP + 16 * c + Mfma<T>::row((int)lane, r)] = acc1[c][r];
            acc1[c][r] = T(0);
          }
        }
      }
    }
  }
  // this wavefront's partials: record (workgroup, wavefront), chunk c, element [row][column]
  T *rec1 = reinterpret_cast<T *>(a.rec_w1), *rec2 = reinterpret_cast<T *>(a.rec_w2);
  const int64_t record = ((int64_t)blockIdx.x * kMlpWaves + wave) * chunks;
#pragma unroll
  for (int c = 0; c < kMlpMaxHidden / 16; ++c) {
    if ((uint32_t)c < chunks) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int at = Mfma<T>::row((int)lane, r) * 16 + (int)m;
        if (rec1 != nullptr) rec1[(record + c) * 256 + at] = acc1[c][r];
        if (rec2 != nullptr) rec2[(record + c) * 256 + at] = acc2[c][r];
      }
    }
  }
}

template <typename T> static size_t mlp_backward_lds(int dp, uint32_t hp) {
  return sizeof(T) * (2 * (size_t)dp * hp + (size_t)kMlpWaves * (3 * 64 * kMlpStride + kMlpMaxHidden));
}

static inline int64_t mlp_backward_grid(int64_t B, int64_t K) {
  const int64_t tiles = B * K / kLgBlock;
  return std::min<int64_t>(tiles, (int64_t)lg_cu_count() * 2);
}

template <typename T>
static int launch_particle_mlp_backward(const void *x, const void *g, const aesmc_affine_map *m1, const aesmc_affine_map *m2,
                                        void *gx, void *rec_w1, void *rec_w2, void *rows, int64_t B, int64_t K,
                                        hipStream_t stream) {
  const int64_t din = m1->din, hid = m1->dout, dout = m2->dout;
  const int dp = lg_pad_dim(std::max(din, dout));
  const uint32_t hp = (uint32_t)((hid + 15) / 16 * 16);
  const size_t lds = mlp_backward_lds<T>(dp, hp);
  if (lds > kLgLdsLimit) return AESMC_ERR_UNSUPPORTED;
  MlpBackwardArgs a;
  a.x = x; a.g = g; a.m1 = lg_map(m1); a.m2 = lg_map(m2); a.gx = gx; a.rec_w1 = rec_w1; a.rec_w2 = rec_w2; a.rows = rows;
  a.N = B * K; a.K = (uint32_t)K; a.HP = hp;
  const unsigned grid = (unsigned)mlp_backward_grid(B, K);
#define MLPB_CASE(DP_)                                                                                              \
  do {                                                                                                              \
    if (lds > 64 * 1024)                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&particle_mlp_backward_kernel<T, DP_>),              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
    hipLaunchKernelGGL((particle_mlp_backward_kernel<T, DP_>), dim3(grid), dim3(kLgBlock), lds, stream, a);         \
  } while (0)
  switch (dp) {
    case 4: MLPB_CASE(4); break;
    case 8: MLPB_CASE(8); break;
    case 10: MLPB_CASE(10); break;
    case 12: MLPB_CASE(12); break;
    default: MLPB_CASE(16); break;
  }
#undef MLPB_CASE
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

}  // namespace aesmc

using namespace aesmc;

extern "C" int64_t aesmc_particle_mlp_max_hidden(void) { return kMlpMaxHidden; }

static int mlp_shape_status(const aesmc_affine_map *layer1, const aesmc_affine_map *layer2) {
  if (layer1 == nullptr || layer2 == nullptr || layer1->weight == nullptr || layer2->weight == nullptr)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (layer1->din < 1 || layer1->din > kLgMaxDim || layer2->dout < 1 || layer2->dout > kLgMaxDim ||
      layer1->dout < 1 || layer1->dout > kMlpMaxHidden || layer2->din != layer1->dout)
    return AESMC_ERR_UNSUPPORTED;
  return AESMC_OK;
}

extern "C" int aesmc_particle_mlp(int dtype, const void *x, const aesmc_affine_map *layer1,
                                  const aesmc_affine_map *layer2, void *out, int64_t B, int64_t K, void *stream) {
  if (x == nullptr || out == nullptr || B < 0 || K < 0) return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  if (!aligned16(x) || !aligned16(out) || x == out) return AESMC_ERR_INVALID_ARGUMENT;
  const int status = mlp_shape_status(layer1, layer2);
  if (status != AESMC_OK) return status;
  if (B == 0 || K == 0) return AESMC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AESMC_F32 ? launch_particle_mlp<float>(x, layer1, layer2, out, B, K, s)
                            : launch_particle_mlp<double>(x, layer1, layer2, out, B, K, s);
}

// records the backward leaves for a batch of B x K particles: one per wavefront of its launch (the binder allocates
// rec_w1 / rec_w2 as [records][ceil(H / 16)][256] and adds them over the first dimension)
extern "C" int64_t aesmc_particle_mlp_backward_records(int64_t B, int64_t K) {
  if (B <= 0 || K <= 0 || K % kLgBlock != 0) return 0;
  return mlp_backward_grid(B, K) * kMlpWaves;
}

extern "C" int aesmc_particle_mlp_backward(int dtype, const void *x, const void *grad_out, const aesmc_affine_map *layer1,
                                           const aesmc_affine_map *layer2, void *grad_x, void *rec_w1, void *rec_w2,
                                           void *rows, int64_t B, int64_t K, void *stream) {
  if (x == nullptr || grad_out == nullptr || B < 0 || K < 0) return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  const int status = mlp_shape_status(layer1, layer2);
  if (status != AESMC_OK) return status;
  if (B == 0 || K == 0) return AESMC_OK;
  // a tile inside one batch row; a free sixteenth input column for the ones that gather the rows' sums
  if (K % kLgBlock != 0 || B * K >= (1ll << 31) || layer1->din > 15) return AESMC_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AESMC_F32
             ? launch_particle_mlp_backward<float>(x, grad_out, layer1, layer2, grad_x, rec_w1, rec_w2, rows, B, K, s)
             : launch_particle_mlp_backward<double>(x, grad_out, layer1, layer2, grad_x, rec_w1, rec_w2, rows, B, K, s);
}
