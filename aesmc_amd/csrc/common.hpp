// Shared device helpers for the aesmc gfx950 kernels.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// G-byte unit of a row copy (K3 and the fused step): the widest power of two <= 16 that divides
// the row length and every address involved.
template <int G> struct Piece;
template <> struct Piece<1> { using type = uint8_t; };
template <> struct Piece<2> { using type = uint16_t; };
template <> struct Piece<4> { using type = uint32_t; };
template <> struct Piece<8> { using type = uint2; };
template <> struct Piece<16> { using type = uint4; };

__device__ __forceinline__ void raise_flag(int32_t *flags, int32_t bit) {
  if (flags != nullptr) atomicOr(flags, bit);
}

}  // namespace aesmc
