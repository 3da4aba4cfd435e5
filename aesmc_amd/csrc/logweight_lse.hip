// K1: fused log-weight combine + per-row log-sum-exp, and its backward.
//
// Replaces aesmc/inference.py:97-98, :125-126 (lw = log f + log g - log q), :130 / :158
// (torch.logsumexp over particles) and aesmc/math.py:27-30.  HBM-bound: 3 reads + 1 write of
// [B,K] (16 B per particle-step in fp32), one streaming pass, no stack of the T weight tensors.
//
// Mapping: TPR threads cooperate on one row (TPR = 64: one wavefront per row, 4 rows per 256-thread
// workgroup; TPR = 256: the whole workgroup).  Each lane streams 16-byte vectors, keeps a running
// (max, sum exp) pair, and the pairs are merged by wavefront shuffles and, for TPR = 256, through
// 4 LDS slots.
#include "common.hpp"

namespace aesmc {

constexpr int kBlock = 256;

template <typename T, int TPR, bool VEC>
__global__ __launch_bounds__(kBlock) void logweight_lse_kernel(
    const T *__restrict__ a, const T *__restrict__ b, const T *__restrict__ c, T *out_lw,
    T *__restrict__ out_lse, int64_t B, int64_t K, int stream, const T *__restrict__ acc_in,
    T *__restrict__ out_acc) {
  constexpr int ROWS = kBlock / TPR;
  constexpr int N = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const int tid = threadIdx.x;
  const int sub = tid / TPR;       // row within the workgroup
  const int t = tid % TPR;         // lane within the row team
  const int64_t row = (int64_t)blockIdx.x * ROWS + sub;
  const bool live = row < B;

  LseState<T> st;
  st.init();
  if (live) {
    const int64_t base = row * K;
    if constexpr (VEC) {
      const int64_t nvec = K / N;
      const V *av = reinterpret_cast<const V *>(a + base);
      const V *bv = b ? reinterpret_cast<const V *>(b + base) : nullptr;
      const V *cv = c ? reinterpret_cast<const V *>(c + base) : nullptr;
      V *ov = out_lw ? reinterpret_cast<V *>(out_lw + base) : nullptr;
      // running sum over time (importance sampling, aesmc/inference.py:157): total = acc_in + lw is
      // written to out_acc and is what the row log-sum-exp is taken of
      const V *accv = acc_in ? reinterpret_cast<const V *>(acc_in + base) : nullptr;
      V *oaccv = acc_in ? reinterpret_cast<V *>(out_acc + base) : nullptr;
      // U vectors per input in flight per lane before any arithmetic: the running (max, sum) pair
      // is a serial chain, so without this the loads of the next trip wait behind it.
      constexpr int U = 4;
      for (int64_t i0 = t; i0 < nvec; i0 += (int64_t)TPR * U) {
        V x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t i = i0 + (int64_t)u * TPR;
          if (i < nvec) x[u] = av[i];
        }
        if (bv) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + (int64_t)u * TPR;
            if (i < nvec) {
              const V y = load16(bv + i, stream);
              x[u].x += y.x;
              x[u].y += y.y;
              if constexpr (N == 4) {
                x[u].z += y.z;
                x[u].w += y.w;
              }
            }
          }
        }
        if (cv) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + (int64_t)u * TPR;
            if (i < nvec) {
              const V y = load16(cv + i, stream);
              x[u].x -= y.x;
              x[u].y -= y.y;
              if constexpr (N == 4) {
                x[u].z -= y.z;
                x[u].w -= y.w;
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t i = i0 + (int64_t)u * TPR;
          if (i < nvec) {
            if (ov) ov[i] = x[u];
            if (accv) {
              const V y = accv[i];
              x[u].x = y.x + x[u].x;
              x[u].y = y.y + x[u].y;
              if constexpr (N == 4) {
                x[u].z = y.z + x[u].z;
                x[u].w = y.w + x[u].w;
              }
              oaccv[i] = x[u];
            }
            T vals[N];
#pragma unroll
            for (int j = 0; j < N; ++j) vals[j] = Vec16<T>::get(x[u], j);
            st.template push_many<N>(vals);
          }
        }
      }
    } else {
      for (int64_t k = t; k < K; k += TPR) {
        T x = a[base + k];
        if (b) x += b[base + k];
        if (c) x -= c[base + k];
        if (out_lw) out_lw[base + k] = x;
        if (acc_in) {
          x = acc_in[base + k] + x;
          out_acc[base + k] = x;
        }
        st.push(x);
      }
    }
  }
  if (out_lse == nullptr) return;  // uniform across the grid

  wave_merge(st);
  if constexpr (TPR > kWave) {
    __shared__ T sm[kBlock / kWave];
    __shared__ T ss[kBlock / kWave];
    __shared__ int sn[kBlock / kWave];
    const int wave = tid / kWave, lane = tid % kWave;
    if (lane == 0) {
      sm[wave] = st.m;
      ss[wave] = st.s;
      sn[wave] = st.nan;
    }
    __syncthreads();
    if (tid == 0) {
      LseState<T> acc;
      acc.m = sm[0];
      acc.s = ss[0];
      acc.nan = sn[0];
      for (int w = 1; w < kBlock / kWave; ++w) acc.merge(sm[w], ss[w], sn[w]);
      if (live) out_lse[row] = acc.value();
    }
  } else {
    if (t == 0 && live) out_lse[row] = st.value();
  }
}

template <typename T, int TPR, bool VEC>
__global__ __launch_bounds__(kBlock) void logweight_lse_bwd_kernel(
    const T *__restrict__ lw, const T *__restrict__ lse, const T *__restrict__ grad_lw,
    const T *__restrict__ grad_lse, T *__restrict__ out_g, T *__restrict__ out_neg_g, int64_t B,
    int64_t K) {
  constexpr int ROWS = kBlock / TPR;
  constexpr int N = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  const int tid = threadIdx.x;
  const int sub = tid / TPR;
  const int t = tid % TPR;
  const int64_t row = (int64_t)blockIdx.x * ROWS + sub;
  if (row >= B) return;
  const int64_t base = row * K;
  const T l = lse[row];
  const T gl = grad_lse ? grad_lse[row] : T(0);
  if constexpr (VEC) {
    const int64_t nvec = K / N;
    const V *xv = reinterpret_cast<const V *>(lw + base);
    const V *gv = grad_lw ? reinterpret_cast<const V *>(grad_lw + base) : nullptr;
    V *og = reinterpret_cast<V *>(out_g + base);
    V *on = out_neg_g ? reinterpret_cast<V *>(out_neg_g + base) : nullptr;
    for (int64_t i = t; i < nvec; i += TPR) {
      V x = xv[i];
      V g;
      g.x = grad_lse ? gl * Num<T>::exp(x.x - l) : T(0);
      g.y = grad_lse ? gl * Num<T>::exp(x.y - l) : T(0);
      if constexpr (N == 4) {
        g.z = grad_lse ? gl * Num<T>::exp(x.z - l) : T(0);
        g.w = grad_lse ? gl * Num<T>::exp(x.w - l) : T(0);
      }
      if (gv) {
        V u = gv[i];
        g.x += u.x;
        g.y += u.y;
        if constexpr (N == 4) {
          g.z += u.z;
          g.w += u.w;
        }
      }
      og[i] = g;
      if (on) {
        V n;
        n.x = -g.x;
        n.y = -g.y;
        if constexpr (N == 4) {
          n.z = -g.z;
          n.w = -g.w;
        }
        on[i] = n;
      }
    }
  } else {
    for (int64_t k = t; k < K; k += TPR) {
      T g = grad_lse ? gl * Num<T>::exp(lw[base + k] - l) : T(0);
      if (grad_lw) g += grad_lw[base + k];
      out_g[base + k] = g;
      if (out_neg_g) out_neg_g[base + k] = -g;
    }
  }
}

// One 256-thread workgroup per row for long rows, or whenever one wavefront per row would leave
// most of the 256 CUs without a workgroup; otherwise one wavefront per row (4 rows per workgroup).
static inline bool use_whole_workgroup_per_row(int64_t B, int64_t K) {
  return K > 1024 || (K >= 256 && B < 2048);
}

static inline bool aligned16(const void *p) { return p == nullptr || ((uintptr_t)p & 15u) == 0; }

template <typename T>
static int launch_fwd(const void *a, const void *b, const void *c, void *lw, void *lse, int64_t B,
                      int64_t K, hipStream_t s, const void *acc_in = nullptr, void *out_acc = nullptr) {
  constexpr int N = Vec16<T>::N;
  const bool vec = (K % N == 0) && aligned16(a) && aligned16(b) && aligned16(c) && aligned16(lw) &&
                   aligned16(acc_in) && aligned16(out_acc);
  const bool wide = use_whole_workgroup_per_row(B, K);
  const int rows = wide ? 1 : kBlock / kWave;
  const int64_t grid64 = (B + rows - 1) / rows;
  if (grid64 > 0x7fffffffLL) return AESMC_ERR_UNSUPPORTED;
  dim3 grid((unsigned)grid64), block(kBlock);
  auto A = (const T *)a;
  auto Bp = (const T *)b;
  auto C = (const T *)c;
  auto L = (T *)lw;
  auto S = (T *)lse;
  auto AI = (const T *)acc_in;
  auto AO = (T *)out_acc;
  const int stream = stream_hint((uint64_t)B * (uint64_t)K * sizeof(T) *
                                 (1 + (b != nullptr) + (c != nullptr) + (lw != nullptr) + 2 * (acc_in != nullptr)));
  if (wide) {
    if (vec)
      hipLaunchKernelGGL((logweight_lse_kernel<T, 256, true>), grid, block, 0, s, A, Bp, C, L, S, B, K, stream, AI, AO);
    else
      hipLaunchKernelGGL((logweight_lse_kernel<T, 256, false>), grid, block, 0, s, A, Bp, C, L, S, B, K, stream, AI, AO);
  } else {
    if (vec)
      hipLaunchKernelGGL((logweight_lse_kernel<T, 64, true>), grid, block, 0, s, A, Bp, C, L, S, B, K, stream, AI, AO);
    else
      hipLaunchKernelGGL((logweight_lse_kernel<T, 64, false>), grid, block, 0, s, A, Bp, C, L, S, B, K, stream, AI, AO);
  }
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

template <typename T>
static int launch_bwd(const void *lw, const void *lse, const void *glw, const void *glse, void *g,
                      void *ng, int64_t B, int64_t K, hipStream_t s) {
  constexpr int N = Vec16<T>::N;
  const bool vec = (K % N == 0) && aligned16(lw) && aligned16(glw) && aligned16(g) && aligned16(ng);
  const bool wide = use_whole_workgroup_per_row(B, K);
  const int rows = wide ? 1 : kBlock / kWave;
  const int64_t grid64 = (B + rows - 1) / rows;
  if (grid64 > 0x7fffffffLL) return AESMC_ERR_UNSUPPORTED;
  dim3 grid((unsigned)grid64), block(kBlock);
  auto X = (const T *)lw;
  auto S = (const T *)lse;
  auto GW = (const T *)glw;
  auto GS = (const T *)glse;
  auto G = (T *)g;
  auto NG = (T *)ng;
  if (wide) {
    if (vec)
      hipLaunchKernelGGL((logweight_lse_bwd_kernel<T, 256, true>), grid, block, 0, s, X, S, GW, GS, G, NG, B, K);
    else
      hipLaunchKernelGGL((logweight_lse_bwd_kernel<T, 256, false>), grid, block, 0, s, X, S, GW, GS, G, NG, B, K);
  } else {
    if (vec)
      hipLaunchKernelGGL((logweight_lse_bwd_kernel<T, 64, true>), grid, block, 0, s, X, S, GW, GS, G, NG, B, K);
    else
      hipLaunchKernelGGL((logweight_lse_bwd_kernel<T, 64, false>), grid, block, 0, s, X, S, GW, GS, G, NG, B, K);
  }
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

}  // namespace aesmc

extern "C" int aesmc_logweight_lse(int dtype, const void *lp_a, const void *lp_b, const void *lp_c,
                                   void *out_lw, void *out_lse, int64_t B, int64_t K,
                                   void *stream) {
  if (lp_a == nullptr || B < 0 || K < 0) return AESMC_ERR_INVALID_ARGUMENT;
  if (out_lw == nullptr && out_lse == nullptr) return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0) return AESMC_OK;
  if (K == 0 && out_lse == nullptr) return AESMC_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32) return aesmc::launch_fwd<float>(lp_a, lp_b, lp_c, out_lw, out_lse, B, K, s);
  if (dtype == AESMC_F64) return aesmc::launch_fwd<double>(lp_a, lp_b, lp_c, out_lw, out_lse, B, K, s);
  return AESMC_ERR_INVALID_ARGUMENT;
}

extern "C" int aesmc_logweight_accumulate(int dtype, const void *lp_a, const void *lp_b, const void *lp_c,
                                          const void *acc_in, void *out_lw, void *out_acc, void *out_lse,
                                          int64_t B, int64_t K, void *stream) {
  if (lp_a == nullptr || acc_in == nullptr || out_acc == nullptr || B < 0 || K < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0) return AESMC_OK;
  if (K == 0 && out_lse == nullptr) return AESMC_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32)
    return aesmc::launch_fwd<float>(lp_a, lp_b, lp_c, out_lw, out_lse, B, K, s, acc_in, out_acc);
  if (dtype == AESMC_F64)
    return aesmc::launch_fwd<double>(lp_a, lp_b, lp_c, out_lw, out_lse, B, K, s, acc_in, out_acc);
  return AESMC_ERR_INVALID_ARGUMENT;
}

extern "C" int aesmc_logweight_lse_backward(int dtype, const void *lw, const void *lse,
                                            const void *grad_lw, const void *grad_lse, void *out_g,
                                            void *out_neg_g, int64_t B, int64_t K, void *stream) {
  if (lw == nullptr || lse == nullptr || out_g == nullptr || B < 0 || K < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0) return AESMC_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AESMC_F32)
    return aesmc::launch_bwd<float>(lw, lse, grad_lw, grad_lse, out_g, out_neg_g, B, K, s);
  if (dtype == AESMC_F64)
    return aesmc::launch_bwd<double>(lw, lse, grad_lw, grad_lse, out_g, out_neg_g, B, K, s);
  return AESMC_ERR_INVALID_ARGUMENT;
}
