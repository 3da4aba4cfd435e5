// Hardware probe (not product code): two facts the fused propagation kernel relies on.
//  1. v_mfma_f32_16x16x4_f32 chained over k-steps, accumulator started from an offset, equals the ascending
//     fmaf chain  acc = fmaf(W[j][i], x[i], acc)  bit for bit (the arithmetic contract of K8-K16, oracle/smc_core.c).
//  2. 16-byte global loads / stores at 8-byte aligned addresses (rows of 40 bytes) return / write the right bytes,
//     and how fast a row gather runs with (16, 16, 8)-byte pieces against five 8-byte pieces.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/probe_mfma.hip -o tools/bin/probe_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef u4 u4a8 __attribute__((aligned(8)));

constexpr int D = 10, KS = 3;

__global__ void chain_valu(const float* __restrict__ W /*[16][12]*/, const float* __restrict__ off, const float* __restrict__ X /*[N][12]*/,
                           float* __restrict__ out /*[N][16]*/, int N) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  for (int j = 0; j < 16; ++j) {
    float acc = off[j];
    for (int i = 0; i < D; ++i) acc = __builtin_fmaf(W[j * 12 + i], X[(size_t)p * 12 + i], acc);
    out[(size_t)p * 16 + j] = acc;
  }
}

// one wavefront per 64 particles: 4 tiles of 16
__global__ void chain_mfma(const float* __restrict__ W, const float* __restrict__ off, const float* __restrict__ X,
                           float* __restrict__ out, int N) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int g = lane >> 4, n = lane & 15;
  float a[KS];
  for (int s = 0; s < KS; ++s) a[s] = W[n * 12 + 4 * s + g];      // A[m = lane & 15][k = lane >> 4]
  for (int tile = 0; tile < 4; ++tile) {
    const int p = wave * 64 + tile * 16 + n;
    if (wave * 64 + tile * 16 >= N) break;
    f4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = off[4 * g + r];
    for (int s = 0; s < KS; ++s) {
      const float b = X[(size_t)p * 12 + 4 * s + g];                 // B[k = lane >> 4][n = lane & 15]
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) out[(size_t)p * 16 + 4 * g + r] = acc[r];
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void gather_rows(const char* __restrict__ src, char* __restrict__ dst, const int* __restrict__ idx, int N) {
  for (int t = blockIdx.x * 256 + threadIdx.x; t < N; t += gridDim.x * 256) {
    const char* p = src + (size_t)idx[t] * 40;
    char* q = dst + (size_t)t * 40;
    if (MODE == 0) {
      u4 a = *reinterpret_cast<const u4a8*>(p);
      u4 b = *reinterpret_cast<const u4a8*>(p + 16);
      uint2 c = *reinterpret_cast<const uint2*>(p + 32);
      *reinterpret_cast<u4a8*>(q) = a;
      *reinterpret_cast<u4a8*>(q + 16) = b;
      *reinterpret_cast<uint2*>(q + 32) = c;
    } else {
      uint2 v[5];
      for (int c = 0; c < 5; ++c) v[c] = *reinterpret_cast<const uint2*>(p + 8 * c);
      for (int c = 0; c < 5; ++c) *reinterpret_cast<uint2*>(q + 8 * c) = v[c];
    }
  }
}

int main() {
  const int N = 1 << 16;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> W(16 * 12, 0.f), off(16, 0.f), X((size_t)N * 12, 0.f);
  for (int j = 0; j < D; ++j) for (int i = 0; i < D; ++i) W[j * 12 + i] = nd(rng) * (j == 3 ? 1e4f : 1.f);
  for (int j = 0; j < D; ++j) off[j] = nd(rng) * (j == 5 ? 1e-30f : 1.f);
  for (int p = 0; p < N; ++p) for (int i = 0; i < D; ++i) X[(size_t)p * 12 + i] = nd(rng) * ((p & 7) == 0 ? 1e-20f : (p & 7) == 1 ? 1e20f : 1.f);
  float *dW, *doff, *dX, *o1, *o2;
  CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&doff, 64)); CK(hipMalloc(&dX, X.size() * 4));
  CK(hipMalloc(&o1, (size_t)N * 64)); CK(hipMalloc(&o2, (size_t)N * 64));
  CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(doff, off.data(), 64, hipMemcpyHostToDevice));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(chain_valu, dim3(N / 256), dim3(256), 0, 0, dW, doff, dX, o1, N);
  hipLaunchKernelGGL(chain_mfma, dim3(N / 256), dim3(256), 0, 0, dW, doff, dX, o2, N);
  CK(hipDeviceSynchronize());
  std::vector<float> h1((size_t)N * 16), h2((size_t)N * 16);
  CK(hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost));
  size_t diff = 0, nonfinite = 0;
  for (size_t e = 0; e < h1.size(); ++e) {
    if ((e & 15) >= (size_t)D) continue;
    uint32_t a, b; memcpy(&a, &h1[e], 4); memcpy(&b, &h2[e], 4);
    if (a != b) { if (diff < 5) printf("  differ at p=%zu j=%zu: valu %.9g (%08x) mfma %.9g (%08x)\n", e / 16, e & 15, h1[e], a, h2[e], b); ++diff; }
    if (!std::isfinite(h1[e])) ++nonfinite;
  }
  printf("mfma_chain_vs_fmaf_chain: %zu of %zu elements differ (%zu non-finite)\n", diff, (size_t)N * D, nonfinite);

  // ---- unaligned 16-byte accesses on 40-byte rows + gather timing
  const int M = 1 << 22;      // 4M rows of 40 B = 168 MB
  char *src, *dst; int* idx;
  CK(hipMalloc(&src, (size_t)M * 40)); CK(hipMalloc(&dst, (size_t)M * 40)); CK(hipMalloc(&idx, (size_t)M * 4));
  std::vector<uint32_t> hs((size_t)M * 10);
  for (size_t e = 0; e < hs.size(); ++e) hs[e] = (uint32_t)(e * 2654435761u);
  std::vector<int> hi(M);
  // sorted ancestors per block of 4096 with repeats (systematic-resampling-like: ~57 % survive)
  { std::uniform_real_distribution<float> ud(0.f, 1.f);
    for (int b = 0; b < M / 4096; ++b) { std::vector<float> w(4096); double tot = 0; for (auto& x : w) { x = std::exp(nd(rng)); tot += x; }
      double c = 0, u = ud(rng); int a = 0; c = w[0]; for (int k = 0; k < 4096; ++k) { double pos = (u + k) / 4096.0 * tot; while (c < pos && a < 4095) c += w[++a]; hi[(size_t)b * 4096 + k] = b * 4096 + a; } } }
  CK(hipMemcpy(src, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(idx, hi.data(), (size_t)M * 4, hipMemcpyHostToDevice));
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(dst, 0, (size_t)M * 40));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(gather_rows<0>, dim3(2048), dim3(256), 0, 0, src, dst, idx, M);
      else hipLaunchKernelGGL(gather_rows<1>, dim3(2048), dim3(256), 0, 0, src, dst, idx, M);
    }
    CK(hipEventRecord(e0, 0));
    for (int rep = 0; rep < 20; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(gather_rows<0>, dim3(2048), dim3(256), 0, 0, src, dst, idx, M);
      else hipLaunchKernelGGL(gather_rows<1>, dim3(2048), dim3(256), 0, 0, src, dst, idx, M);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint32_t> hd((size_t)M * 10);
    CK(hipMemcpy(hd.data(), dst, hd.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int t = 0; t < M; ++t) for (int c = 0; c < 10; ++c) if (hd[(size_t)t * 10 + c] != hs[(size_t)hi[t] * 10 + c]) ++bad;
    printf("gather mode %d (%s): %.1f us per launch, %.2f TB/s on %d MB (idx 4B + 40 in + 40 out), %zu wrong words\n", mode,
           mode == 0 ? "16+16+8 B pieces at 8-byte alignment" : "5 x 8 B pieces", ms * 1000 / 20,
           (double)M * 84 / (ms / 20 * 1e-3) / 1e12, (int)((size_t)M * 84 >> 20), bad);
  }
  return diff == 0 ? 0 : 1;
}
