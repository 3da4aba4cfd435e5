// K3: resample gather  dst[b,k,:] = src[b, idx[b,k], :]  and its backward (segmented sum).
//
// Replaces torch.gather at aesmc/state.py:179 (element-granular gather with an int64 index
// expanded to the value's full shape) and its scatter_add autograd.  HBM-bound:
// 8 B index + row_bytes read + row_bytes write per particle.
//
// Forward mapping: a batch row's output is one dense run of K*row_bytes bytes.  It is cut into
// 16-byte chunks, one per lane, so every wavefront store instruction writes 1 KiB contiguously.
// A chunk is assembled from 16/G source pieces of G bytes (G = largest power of two <= 16 dividing
// row_bytes and the source strides), each piece lying inside one particle's row.  Systematic
// resampling returns non-decreasing indices, so neighbouring lanes read neighbouring (often the
// same) source rows: surviving rows come from HBM once, repeats are L1/L2 hits.
#include "common.hpp"

namespace aesmc {

constexpr int kGatherBlock = 256;

template <int G> struct Piece;
template <> struct Piece<1> { using type = uint8_t; };
template <> struct Piece<2> { using type = uint16_t; };
template <> struct Piece<4> { using type = uint32_t; };
template <> struct Piece<8> { using type = uint2; };
template <> struct Piece<16> { using type = uint4; };

// V pieces of G bytes per lane; V*G == 16 on the vector path, V == 1 on the unaligned fallback.
template <int G, int V>
__global__ __launch_bounds__(kGatherBlock) void resample_gather_kernel(
    const char *__restrict__ src, const int64_t *__restrict__ idx, char *__restrict__ dst,
    int32_t *flags, uint32_t K, uint32_t ppp /* pieces per particle */,
    uint64_t row_pieces /* K * ppp */, uint32_t chunks_per_row, uint32_t blocks_per_row,
    int64_t stride_b, int64_t stride_k) {
  using P = typename Piece<G>::type;
  const uint32_t b = blockIdx.x / blocks_per_row;
  const uint32_t cb = blockIdx.x - b * blocks_per_row;
  const uint32_t chunk = cb * kGatherBlock + threadIdx.x;
  if (chunk >= chunks_per_row) return;

  const uint64_t p0 = (uint64_t)chunk * V;
  uint32_t k = (uint32_t)(p0 / ppp);
  uint32_t r = (uint32_t)(p0 - (uint64_t)k * ppp);
  const int64_t *irow = idx + (uint64_t)b * K;
  const char *srow = src + (int64_t)b * stride_b;
  char *drow = dst + ((uint64_t)b * row_pieces + p0) * G;

  P piece[V];
  int bad = 0;
  uint32_t cur_k = 0xffffffffu;
  const char *prow = nullptr;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    if (p0 + i < row_pieces) {
      if (k != cur_k) {
        int64_t a = irow[k];
        if ((uint64_t)a >= (uint64_t)K) {  // torch.gather would raise; never fault, report instead
          bad = 1;
          a = a < 0 ? 0 : (int64_t)K - 1;
        }
        prow = srow + a * stride_k;
        cur_k = k;
      }
      piece[i] = *reinterpret_cast<const P *>(prow + (uint64_t)r * G);
      if (++r == ppp) {
        r = 0;
        ++k;
      }
    }
  }
  if (p0 + V <= row_pieces) {
    if constexpr (V * G == 16 && V > 1) {
      uint4 out;
      __builtin_memcpy(&out, piece, 16);
      *reinterpret_cast<uint4 *>(drow) = out;
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) reinterpret_cast<P *>(drow)[i] = piece[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i)
      if (p0 + i < row_pieces) reinterpret_cast<P *>(drow)[i] = piece[i];
  }
  if (bad) raise_flag(flags, AESMC_FLAG_INDEX_OUT_OF_RANGE);
}

// Backward, general form (any index order): one lane per gradient element; a lane that starts a
// run of equal indices (or a kRunCap-aligned piece of a long run) sums the run and issues ONE
// hardware float atomic for it.  With sorted indices and runs shorter than kRunCap every
// destination receives exactly one add onto zero, so the result is then bitwise reproducible.
constexpr uint32_t kRunCap = 32;

template <typename T>
__global__ __launch_bounds__(kGatherBlock) void resample_gather_bwd_kernel(
    const T *__restrict__ grad_out, const int64_t *__restrict__ idx, T *grad_src, int32_t *flags,
    uint32_t K, uint32_t D, uint64_t row_elems /* K * D */, uint32_t blocks_per_row) {
  const uint32_t b = blockIdx.x / blocks_per_row;
  const uint32_t cb = blockIdx.x - b * blocks_per_row;
  const uint64_t e = (uint64_t)cb * kGatherBlock + threadIdx.x;
  if (e >= row_elems) return;
  const uint32_t k = (uint32_t)(e / D);
  const uint32_t c = (uint32_t)(e - (uint64_t)k * D);
  const int64_t *irow = idx + (uint64_t)b * K;
  const int64_t a = irow[k];
  const bool head = (k % kRunCap == 0) || (irow[k - 1] != a);
  if (!head) return;
  if ((uint64_t)a >= (uint64_t)K) {
    raise_flag(flags, AESMC_FLAG_INDEX_OUT_OF_RANGE);
    return;
  }
  const T *grow = grad_out + (uint64_t)b * row_elems;
  T sum = grow[e];
  const uint32_t stop = min(K, (k / kRunCap + 1) * kRunCap);
  for (uint32_t kk = k + 1; kk < stop && irow[kk] == a; ++kk) sum += grow[(uint64_t)kk * D + c];
  unsafeAtomicAdd(grad_src + ((uint64_t)b * K + (uint64_t)a) * D + c, sum);
}

static inline int low_pow2(uint64_t x, int cap) {  // largest power of two <= cap dividing x
  int g = cap;
  while (g > 1 && (x % (uint64_t)g) != 0) g >>= 1;
  return g;
}

template <int G, int V>
static void launch_gather(const void *src, const int64_t *idx, void *dst, int32_t *flags, int64_t B,
                          int64_t K, int64_t row_bytes, int64_t sb, int64_t sk, hipStream_t s) {
  const uint32_t ppp = (uint32_t)(row_bytes / G);
  const uint64_t row_pieces = (uint64_t)K * ppp;
  const uint32_t chunks = (uint32_t)((row_pieces + V - 1) / V);
  const uint32_t bpr = (chunks + kGatherBlock - 1) / kGatherBlock;
  hipLaunchKernelGGL((resample_gather_kernel<G, V>), dim3((unsigned)(B * bpr)), dim3(kGatherBlock),
                     0, s, (const char *)src, idx, (char *)dst, flags, (uint32_t)K, ppp, row_pieces,
                     chunks, bpr, sb, sk);
}

}  // namespace aesmc

extern "C" int aesmc_resample_gather(const void *src, const int64_t *idx, void *dst, int32_t *flags,
                                     int64_t B, int64_t K, int64_t row_bytes, int64_t src_stride_b,
                                     int64_t src_stride_k, void *stream) {
  using namespace aesmc;
  if (src == nullptr || idx == nullptr || dst == nullptr || B < 0 || K < 0 || row_bytes < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0 || row_bytes == 0) return AESMC_OK;
  // 32-bit piece arithmetic inside a batch row; grid is B * blocks_per_row workgroups.
  if ((uint64_t)K * (uint64_t)row_bytes >= (1ull << 32) || K >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  // Piece size: every source piece address must be G-aligned.
  int G = low_pow2((uint64_t)row_bytes, 16);
  G = low_pow2((uint64_t)(uintptr_t)src, G);
  G = low_pow2((uint64_t)(src_stride_b < 0 ? -src_stride_b : src_stride_b), G);
  G = low_pow2((uint64_t)(src_stride_k < 0 ? -src_stride_k : src_stride_k), G);
  G = low_pow2((uint64_t)(uintptr_t)dst, G);
  // 16-byte stores need a 16-byte aligned dst base and batch-row pitch.
  const bool vec = (((uintptr_t)dst & 15u) == 0) && (((uint64_t)K * (uint64_t)row_bytes) % 16 == 0);
  {
    const uint64_t pieces = (uint64_t)K * (uint64_t)(row_bytes / G);
    const int v = vec ? 16 / G : 1;
    const uint64_t bpr = ((pieces + v - 1) / v + kGatherBlock - 1) / kGatherBlock;
    if ((uint64_t)B * bpr > 0x7fffffffull) return AESMC_ERR_UNSUPPORTED;
  }
#define AESMC_GATHER_CASE(g)                                                                        \
  case g:                                                                                           \
    if (vec)                                                                                        \
      launch_gather<g, 16 / g>(src, idx, dst, flags, B, K, row_bytes, src_stride_b, src_stride_k, s); \
    else                                                                                            \
      launch_gather<g, 1>(src, idx, dst, flags, B, K, row_bytes, src_stride_b, src_stride_k, s);    \
    break;
  switch (G) {
    AESMC_GATHER_CASE(16)
    AESMC_GATHER_CASE(8)
    AESMC_GATHER_CASE(4)
    AESMC_GATHER_CASE(2)
    AESMC_GATHER_CASE(1)
    default:
      return AESMC_ERR_INVALID_ARGUMENT;
  }
#undef AESMC_GATHER_CASE
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}

extern "C" int aesmc_resample_gather_backward(int dtype, const void *grad_out, const int64_t *idx,
                                              void *grad_src, int32_t *flags, int64_t B, int64_t K,
                                              int64_t row_elems, void *stream) {
  using namespace aesmc;
  if (grad_out == nullptr || idx == nullptr || grad_src == nullptr || B < 0 || K < 0 || row_elems < 0)
    return AESMC_ERR_INVALID_ARGUMENT;
  if (dtype != AESMC_F32 && dtype != AESMC_F64) return AESMC_ERR_INVALID_ARGUMENT;
  if (B == 0 || K == 0 || row_elems == 0) return AESMC_OK;
  if ((uint64_t)K * (uint64_t)row_elems >= (1ull << 32) || K >= (1ll << 31)) return AESMC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype == AESMC_F32 ? 4 : 8;
  const uint64_t re = (uint64_t)K * (uint64_t)row_elems;
  const uint64_t bpr = (re + kGatherBlock - 1) / kGatherBlock;
  if ((uint64_t)B * bpr > 0x7fffffffull) return AESMC_ERR_UNSUPPORTED;
  if (hipMemsetAsync(grad_src, 0, (size_t)B * re * esz, s) != hipSuccess) return AESMC_ERR_LAUNCH;
  dim3 grid((unsigned)((uint64_t)B * bpr)), block(kGatherBlock);
  if (dtype == AESMC_F32)
    hipLaunchKernelGGL((resample_gather_bwd_kernel<float>), grid, block, 0, s, (const float *)grad_out,
                       idx, (float *)grad_src, flags, (uint32_t)K, (uint32_t)row_elems, re,
                       (uint32_t)bpr);
  else
    hipLaunchKernelGGL((resample_gather_bwd_kernel<double>), grid, block, 0, s,
                       (const double *)grad_out, idx, (double *)grad_src, flags, (uint32_t)K,
                       (uint32_t)row_elems, re, (uint32_t)bpr);
  return hipGetLastError() == hipSuccess ? AESMC_OK : AESMC_ERR_LAUNCH;
}
