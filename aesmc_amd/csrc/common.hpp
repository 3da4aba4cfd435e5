// Shared device helpers for the aesmc gfx950 kernels.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/aesmc_hip.h"

namespace aesmc {

constexpr int kWave = 64;

template <typename T> struct Num;
template <> struct Num<float> {
  static __device__ __forceinline__ float neg_inf() { return -__builtin_huge_valf(); }
  static __device__ __forceinline__ float pos_inf() { return __builtin_huge_valf(); }
  static __device__ __forceinline__ float nan() { return __builtin_nanf(""); }
  static __device__ __forceinline__ float exp(float x) { return ::expf(x); }
  static __device__ __forceinline__ float log(float x) { return ::logf(x); }
  static __device__ __forceinline__ float max(float a, float b) { return ::fmaxf(a, b); }
};
template <> struct Num<double> {
  static __device__ __forceinline__ double neg_inf() { return -__builtin_huge_val(); }
  static __device__ __forceinline__ double pos_inf() { return __builtin_huge_val(); }
  static __device__ __forceinline__ double nan() { return __builtin_nan(""); }
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double log(double x) { return ::log(x); }
  static __device__ __forceinline__ double max(double a, double b) { return ::fmax(a, b); }
};

// 16-byte vector of T (float4 / double2) for coalesced 1 KiB-per-wave accesses.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  using type = float4;
  static __device__ __forceinline__ float get(const float4 &v, int i) {
    return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
  }
  static __device__ __forceinline__ float4 make(float a, float b, float c, float d) {
    return make_float4(a, b, c, d);
  }
};
template <> struct Vec16<double> {
  static constexpr int N = 2;
  using type = double2;
  static __device__ __forceinline__ double get(const double2 &v, int i) { return i == 0 ? v.x : v.y; }
};

// Running (max, sum-of-exp) pair of a streaming log-sum-exp, plus a sticky NaN marker.
template <typename T> struct LseState {
  T m;
  T s;
  int nan;
  __device__ __forceinline__ void init() {
    m = Num<T>::neg_inf();
    s = T(0);
    nan = 0;
  }
  // Fold one value in.
  __device__ __forceinline__ void push(T v) {
    nan |= (v != v);
    if (v > m) {
      s = (m == Num<T>::neg_inf()) ? T(0) : s * Num<T>::exp(m - v);
      m = v;
    }
    if (m != Num<T>::neg_inf()) s += Num<T>::exp(v - m);
  }
  // Fold N values in with at most one rescale: the same (max, sum) pair as N single pushes up to
  // rounding, without a data-dependent branch per value.
  template <int N> __device__ __forceinline__ void push_many(const T (&v)[N]) {
    T top = v[0];
    nan |= (v[0] != v[0]);
#pragma unroll
    for (int i = 1; i < N; ++i) {
      nan |= (v[i] != v[i]);
      top = Num<T>::max(top, v[i]);
    }
    if (top > m) {
      s = (m == Num<T>::neg_inf()) ? T(0) : s * Num<T>::exp(m - top);
      m = top;
    }
    if (m != Num<T>::neg_inf()) {
#pragma unroll
      for (int i = 0; i < N; ++i) s += Num<T>::exp(v[i] - m);
    }
  }
  // Fold another partial state in.
  __device__ __forceinline__ void merge(T m2, T s2, int nan2) {
    nan |= nan2;
    T mm = Num<T>::max(m, m2);
    if (mm == Num<T>::neg_inf()) {
      s = T(0);
    } else {
      T a = (m == Num<T>::neg_inf()) ? T(0) : s * Num<T>::exp(m - mm);
      T b = (m2 == Num<T>::neg_inf()) ? T(0) : s2 * Num<T>::exp(m2 - mm);
      s = a + b;
    }
    m = mm;
  }
  // torch.logsumexp conventions for the special rows (see include/aesmc_hip.h, K1).
  __device__ __forceinline__ T value() const {
    if (nan) return Num<T>::nan();
    if (m == Num<T>::pos_inf() || m == Num<T>::neg_inf()) return m;
    return m + Num<T>::log(s);
  }
};

template <typename T> __device__ __forceinline__ void wave_merge(LseState<T> &st) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    T m2 = __shfl_xor(st.m, off, kWave);
    T s2 = __shfl_xor(st.s, off, kWave);
    int n2 = __shfl_xor(st.nan, off, kWave);
    st.merge(m2, s2, n2);
  }
}

// 16-byte load of data that is read exactly once (locations produced by the user's callables,
// noise): the non-temporal hint keeps it from displacing lines that WILL be read again (the
// latent, the log-weights) from L2 / MALL.  K5 at B=1024 K=4096 d=10: 152 -> 140 us.
typedef unsigned int u32x4_native __attribute__((ext_vector_type(4)));
template <typename V> __device__ __forceinline__ V stream_load16(const V *p) {
  static_assert(sizeof(V) == 16, "16-byte vectors only");
  const u32x4_native raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4_native *>(p));
  V out;
  __builtin_memcpy(&out, &raw, 16);
  return out;
}

// ... but only when the launch's operands are too large to be cache-resident anyway: at
// configs[1] sizes (10 MB per tensor, produced by the kernel before and still in L2) the hint made
// the whole ELBO 3 % SLOWER, at B=1024 K=4096 (168 MB per tensor) 3 % faster.  `load16` takes the
// decision as a launch-uniform flag; `stream_hint` makes it from the bytes a launch touches.
template <typename V> __device__ __forceinline__ V load16(const V *p, int stream) {
  return stream ? stream_load16(p) : *p;
}

// Measurement knobs (kernel forms, tile shapes, thresholds a timing experiment wants to pin) are read from the
// environment ONLY when AESMC_MEASUREMENT_KNOBS=1 is set beside them: a product process never changes behaviour because
// some AESMC_* variable happens to be exported.  (tools/*.py set it; tests select forms through the aesmc_test_* hooks.)
static inline const char *measurement_knob(const char *name) {
  static const bool enabled = [] { const char *v = getenv("AESMC_MEASUREMENT_KNOBS"); return v != nullptr && v[0] == '1'; }();
  return enabled ? getenv(name) : nullptr;
}

static inline int stream_hint(uint64_t launch_bytes) {
  static const uint64_t threshold = [] {
    const char *mb = measurement_knob("AESMC_STREAM_MB");      // tuning knob; default: beyond L2 + half the MALL
    return (uint64_t)(mb != nullptr ? atoll(mb) : 160) << 20;
  }();
  return launch_bytes >= threshold ? 1 : 0;
}

// G-byte unit of a row copy (K3 and the fused step): the widest power of two <= 16 that divides
// the row length and every address involved.
template <int G> struct Piece;
template <> struct Piece<1> { using type = uint8_t; };
template <> struct Piece<2> { using type = uint16_t; };
template <> struct Piece<4> { using type = uint32_t; };
template <> struct Piece<8> { using type = uint2; };
template <> struct Piece<16> { using type = uint4; };

// Zero fill as an ordinary kernel.  hipMemsetAsync is avoided on purpose: captured into a large
// hipGraph (a whole backward pass) its memset node was observed NOT to be ordered after the
// nodes captured before it on the same stream, so a block recycled inside the graph's memory pool
// was zeroed too early and read back stale (rows without offspring in K3's backward came out as
// garbage from the second graph of a process on).  A kernel node keeps the stream order.
__global__ __launch_bounds__(256) static void zero_fill_kernel(uint4 *__restrict__ dst16, uint64_t n16,
                                                               unsigned char *__restrict__ tail,
                                                               uint32_t ntail) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
    dst16[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0 && threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

static inline bool zero_fill_async(void *dst, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return true;
  // bytes up to the first 16-byte boundary go with the tail bytes' mechanism (at most 15 lanes)
  const uint32_t head = (uint32_t)((16 - (reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u);
  if (head != 0) {
    const uint32_t nhead = head < bytes ? head : (uint32_t)bytes;
    hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(256), 0, stream, static_cast<uint4 *>(nullptr),
                       (uint64_t)0, static_cast<unsigned char *>(dst), nhead);
    dst = static_cast<unsigned char *>(dst) + nhead;
    bytes -= nhead;
    if (bytes == 0) return hipGetLastError() == hipSuccess;
  }
  const uint64_t n16 = bytes / 16;
  const uint32_t ntail = (uint32_t)(bytes - n16 * 16);
  uint64_t blocks = (n16 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<uint4 *>(dst),
                     n16, static_cast<unsigned char *>(dst) + n16 * 16, ntail);
  return hipGetLastError() == hipSuccess;
}

__device__ __forceinline__ void raise_flag(int32_t *flags, int32_t bit) {
  if (flags != nullptr) atomicOr(flags, bit);
}

}  // namespace aesmc
