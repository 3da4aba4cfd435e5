// Standard-normal noise drawn inside a kernel from the stream PyTorch's own generator would have produced.
//
// The reference draws a step's particles with `Normal.rsample` (aesmc/state.py:98), i.e. one
// `torch.empty(shape).normal_()` per timestep.  On a HIP device that call is ATen's
// `distribution_elementwise_grid_stride_kernel` over Philox4x32-10 (rocRAND's `hiprand_normal4`):
//
//   G      = 256 * min(#CU * (maxThreadsPerCU / 256), ceil(numel / 256))      threads of the launch
//   thread t, trip c:  (r0, r1, r2, r3) = Philox4x32-10(counter = (offset / 4 + c, 0, t, 0), key = seed)
//                      (n0, n1) = box_muller(r0, r1),  (n2, n3) = box_muller(r2, r3)
//   element e = t + G * (4 c + i)  receives  n_i                                i = 0 .. 3
//   the generator's offset then advances by 4 * ceil(numel / (4 G))
//
// so a value is a pure function of (seed, offset, G, e): any kernel that knows those four can form the
// noise where it is consumed instead of reading it back from HBM — the same bits, and the generator is
// advanced by what `normal_` would have consumed (aesmc_amd/_philox.py), so everything drawn afterwards
// is unchanged too.  `tests/test_gpu_round3.py` (the Philox tests) holds this against `torch.empty(n).normal_()` bit for bit.
//
// Box-Muller as rocRAND writes it (rocrand_normal.h `box_muller`):
//   u = 2^-32 + r0 * 2^-32,  v = 2^-32 * 2pi + r1 * (2^-32 * 2pi),  s = sqrtf(-2 logf(u)),
//   (sin v * s, cos v * s) with the hardware sine / cosine (`__sincosf`).
// The library this file is part of is compiled with -ffp-contract=off; rocRAND inside PyTorch is compiled
// with hipcc's default (contraction on), which fuses the two affine maps into one fma each: written out here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aesmc {

struct PhiloxStream {
  uint32_t key0, key1;      // seed
  uint32_t base_lo, base_hi;  // offset / 4: the counter of a thread's first trip
  uint32_t threads;         // G
  // hipGraph replays: (seed, offset) live in device memory, refreshed by the host before each replay; the
  // launch's own offset above is then relative to it (what the captured region consumed before this launch)
  const uint64_t *state;
};

static inline PhiloxStream philox_stream(uint64_t seed, uint64_t offset, int64_t threads, const uint64_t *state) {
  PhiloxStream s;
  s.key0 = (uint32_t)seed;
  s.key1 = (uint32_t)(seed >> 32);
  s.base_lo = (uint32_t)(offset >> 2);
  s.base_hi = (uint32_t)(offset >> 34);
  s.threads = (uint32_t)threads;
  s.state = state;
  return s;
}

// first thing in a kernel: the stream with a device-resident generator state folded in
__device__ __forceinline__ PhiloxStream philox_resolve(PhiloxStream s) {
  if (s.state != nullptr) {
    const uint64_t seed = s.state[0];
    const uint64_t base = (s.state[1] >> 2) + (((uint64_t)s.base_hi << 32) | s.base_lo);
    s.key0 = (uint32_t)seed;
    s.key1 = (uint32_t)(seed >> 32);
    s.base_lo = (uint32_t)base;
    s.base_hi = (uint32_t)(base >> 32);
  }
  return s;
}

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    // (as one 64-bit product each: v_mad_u64_u32 gives both halves in one instruction)
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    // (a ^ b ^ c in one instruction: v_bitop3_b32 with the truth table of a three-way exclusive or — gfx950; the compiler
    //  writes two v_xor_b32)
#if defined(__HIP_DEVICE_COMPILE__)
    c0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
    c2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
#else
    c0 = hi1 ^ c1 ^ k0;
    c2 = hi0 ^ c3 ^ k1;
#endif
    c1 = lo1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}

// the four raw words of (thread t, trip c)
__device__ __forceinline__ uint4 philox_words(const PhiloxStream &s, uint32_t t, uint32_t c) {
  const uint32_t lo = s.base_lo + c;
  const uint32_t hi = s.base_hi + (lo < s.base_lo ? 1u : 0u);
  return philox4x32_10(lo, hi, t, 0u, s.key0, s.key1);
}

// logf and sqrtf as the device library evaluates them, WITHOUT the steps that exist for arguments Box-Muller never
// has: u lies in [2^-32, 1], so logf's rescaling of subnormal inputs (compare, two selects, ldexp) and its infinity
// check never act, and -2 log u is -0 or in [1.1e-7, 44.4], so sqrtf's rescaling of inputs below 2^-96 never acts.
// What remains is, instruction for instruction, the sequence `__builtin_logf` / `__builtin_sqrtf` compile to with
// -ffp-contract=off on gfx950 (v_log_f32 times ln 2 in two pieces; v_sqrt_f32 and its one-ulp correction): the same
// bits (the tests that hold the stream against torch.empty(n).normal_() bit for bit cover it), 22 fewer vector
// instructions per Philox call.
__device__ __forceinline__ float box_muller_log(float u) {
  const float y = __builtin_amdgcn_logf(u);                             // log2
  const float hi = __builtin_bit_cast(float, 0x3f317217u), lo = __builtin_bit_cast(float, 0x3377d1cfu);      // ln 2
  const float p = hi * y;
  float e = __builtin_fmaf(y, hi, -p);
  e = __builtin_fmaf(y, lo, e);
  return p + e;
}
__device__ __forceinline__ float box_muller_sqrt(float a) {
  const float r = __builtin_amdgcn_sqrtf(a);
  const float below = __builtin_bit_cast(float, __builtin_bit_cast(int, r) - 1);
  const float above = __builtin_bit_cast(float, __builtin_bit_cast(int, r) + 1);
  const float err_below = __builtin_fmaf(-below, r, a), err_above = __builtin_fmaf(-above, r, a);
  float s = (0.0f >= err_below) ? below : r;
  s = (0.0f < err_above) ? above : s;
  // (a = -0.0 — u = 1 — needs no case of its own: v_sqrt_f32 gives -0.0, `below` is then a NaN and `above` a denormal
  //  whose error term is -0.0: neither comparison holds and s = r = a, the value the device library's own `a == 0 ? a : s`
  //  returns; +0.0 does not occur, and every other a is at least 1.1e-7)
  return s;
}

// FUSED = false (separate multiply and add) exists for the probe that established which one PyTorch's build
// of rocRAND uses (aesmc_philox_normal_fill's `variant`).
template <bool FUSED = true> __device__ __forceinline__ float2 box_muller_f32(uint32_t x, uint32_t y) {
  float u, v;
  if constexpr (FUSED) {
    u = __builtin_fmaf((float)x, 2.3283064e-10f, 2.3283064e-10f);
    v = __builtin_fmaf((float)y, 1.46291807e-09f, 1.46291807e-09f);
  } else {
    u = 2.3283064e-10f + ((float)x * 2.3283064e-10f);
    v = 1.46291807e-09f + ((float)y * 1.46291807e-09f);
  }
  const float s = box_muller_sqrt(-2.0f * box_muller_log(u));
  float2 out;
  // `normal_`'s own transform, rand * std + mean with std = 1 and mean = 0, follows: the sum turns -0.0 into +0.0
  // (the product with 1.0f changes no value and is left out)
  out.x = (__ocml_native_sin_f32(v) * s) + 0.0f;
  out.y = (__ocml_native_cos_f32(v) * s) + 0.0f;
  return out;
}

// (n0, n1, n2, n3) of (thread t, trip c): what ATen's kernel hands its elements t + G (4c + i)
template <bool FUSED = true>
__device__ __forceinline__ float4 philox_normal4(const PhiloxStream &s, uint32_t t, uint32_t c) {
  const uint4 w = philox_words(s, t, c);
  const float2 a = box_muller_f32<FUSED>(w.x, w.y), b = box_muller_f32<FUSED>(w.z, w.w);
  return make_float4(a.x, a.y, b.x, b.y);
}

// One element on its own (edges of a tile): element e of the tensor.
__device__ __forceinline__ float philox_normal_element(const PhiloxStream &s, uint64_t e) {
  const uint64_t m = e / s.threads;
  const uint32_t t = (uint32_t)(e - m * s.threads);
  const float4 n = philox_normal4(s, t, (uint32_t)(m >> 2));
  const uint32_t i = (uint32_t)m & 3u;
  return i == 0 ? n.x : i == 1 ? n.y : i == 2 ? n.z : n.w;
}

}  // namespace aesmc
