/* ORACLE — TEST INFRASTRUCTURE ONLY.  Plain scalar C restatement of the index / byte work of the
 * hot path, written independently of both the HIP kernels and the NumPy oracle
 * (oracle/kernel_oracle.py); tests/test_oracle.py pins it to the fixtures captured from the
 * reference (tests/golden/resampler_*.npz) and to the NumPy oracle, GPU tests compare the HIP
 * library against it.  Nothing under aesmc_amd/ may link or load this file.
 *
 * Each function cites the reference lines it restates (paths relative to /root/reference).
 *
 * Build: make -C oracle   ->  oracle/_build/libsmc_oracle.so   (gcc, no fast-math: IEEE semantics)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { FLAG_NAN_LOG_WEIGHT = 1, FLAG_DEGENERATE_ROW = 2, FLAG_INDEX_OUT_OF_RANGE = 4 };

/* Systematic resampling with the uniforms passed in.
 * aesmc/inference.py:250-264:  positions (u_b + k) / K, normalised weights w = exp(lw - lse),
 * c = cumsum(w) / max(cumsum(w)), index = np.digitize(position, c) = #{ j : c[j] <= position }.
 * The K2 contract evaluates the weights and the CDF in float64 whatever the input dtype and
 * normalises by the CDF's own last entry (c[K-1] == 1 exactly, like c / max(c)); a row with a NaN,
 * or without a finite maximum, yields index K everywhere and a flag (inference.py:244-245 raises
 * FloatingPointError for NaN; an all -inf row makes np.digitize return K).
 * `lw` holds float64 values (float32 inputs are widened by the caller: exact). */
int smc_oracle_ancestor_index(const double *lw, const double *u, int64_t *idx, int64_t B, int64_t K) {
  int flags = 0;
  double *cdf = (double *)malloc((size_t)(K > 0 ? K : 1) * sizeof(double));
  for (int64_t b = 0; b < B; ++b) {
    const double *row = lw + b * K;
    int64_t *out = idx + b * K;
    int has_nan = 0;
    double m = -INFINITY;
    for (int64_t k = 0; k < K; ++k) {
      if (row[k] != row[k]) has_nan = 1;
      if (row[k] > m) m = row[k];
    }
    if (has_nan || !(m > -INFINITY && m < INFINITY)) {
      flags |= has_nan ? FLAG_NAN_LOG_WEIGHT : FLAG_DEGENERATE_ROW;
      for (int64_t k = 0; k < K; ++k) out[k] = K;
      continue;
    }
    double run = 0.0;
    for (int64_t k = 0; k < K; ++k) { /* sequential sum, as np.cumsum */
      run += exp(row[k] - m);
      cdf[k] = run;
    }
    const double total = cdf[K - 1];
    for (int64_t k = 0; k < K; ++k) cdf[k] = cdf[k] / total;
    /* both sequences are non-decreasing: one merge pass gives every count */
    int64_t j = 0;
    for (int64_t k = 0; k < K; ++k) {
      const double position = (u[b] + (double)k) / (double)K;
      while (j < K && cdf[j] <= position) ++j;
      out[k] = j;
    }
  }
  free(cdf);
  return flags;
}

/* aesmc/state.py:158-183 (torch.gather along dim 1 with the index expanded over the row):
 * dst[b,k,:] = src[b, idx[b,k], :], rows of row_bytes bytes.  An index outside [0, K) — which
 * torch.gather rejects — is clamped and flagged, as the K3 contract says. */
int smc_oracle_gather(const unsigned char *src, const int64_t *idx, unsigned char *dst, int64_t B,
                      int64_t K, int64_t row_bytes) {
  int flags = 0;
  for (int64_t b = 0; b < B; ++b)
    for (int64_t k = 0; k < K; ++k) {
      int64_t a = idx[b * K + k];
      if (a < 0 || a >= K) {
        flags |= FLAG_INDEX_OUT_OF_RANGE;
        a = a < 0 ? 0 : K - 1;
      }
      memcpy(dst + (b * K + k) * row_bytes, src + (b * K + a) * row_bytes, (size_t)row_bytes);
    }
  return flags;
}

/* Adjoint of the gather (autograd of torch.gather, state.py:179): grad_src[b,j,:] = sum over
 * { k : idx[b,k] == j } of grad_out[b,k,:], summed in increasing k in float64. */
int smc_oracle_gather_backward(const double *grad_out, const int64_t *idx, double *grad_src, int64_t B,
                               int64_t K, int64_t D) {
  int flags = 0;
  memset(grad_src, 0, (size_t)(B * K * D) * sizeof(double));
  for (int64_t b = 0; b < B; ++b)
    for (int64_t k = 0; k < K; ++k) {
      const int64_t a = idx[b * K + k];
      if (a < 0 || a >= K) {
        flags |= FLAG_INDEX_OUT_OF_RANGE;
        continue;
      }
      for (int64_t d = 0; d < D; ++d) grad_src[(b * K + a) * D + d] += grad_out[(b * K + k) * D + d];
    }
  return flags;
}

/* aesmc/inference.py:196-231 (get_resampled_latents), index part: the lineage of every final
 * particle.  indices: T-1 arrays [B,K] back to back; lineage: T arrays [B,K] back to back, with
 * lineage[T-1] = identity and lineage[t-1][b,k] = indices[t-1][b, lineage[t][b,k]]. */
void smc_oracle_lineage(const int64_t *indices, int64_t *lineage, int64_t T, int64_t B, int64_t K) {
  const int64_t n = B * K;
  for (int64_t b = 0; b < B; ++b)
    for (int64_t k = 0; k < K; ++k) lineage[(T - 1) * n + b * K + k] = k;
  for (int64_t t = T - 1; t > 0; --t)
    for (int64_t b = 0; b < B; ++b)
      for (int64_t k = 0; k < K; ++k)
        lineage[(t - 1) * n + b * K + k] = indices[(t - 1) * n + b * K + lineage[t * n + b * K + k]];
}

/* aesmc/inference.py:125-126, :130: lw = a + b - c, lse = logsumexp over particles, in float64
 * (max-shifted; rows of -inf give -inf, a +inf gives +inf, NaN propagates — torch.logsumexp's
 * conventions).  b and c may be NULL. */
void smc_oracle_logweight_lse(const double *a, const double *b, const double *c, double *lw, double *lse,
                              int64_t B, int64_t K) {
  for (int64_t r = 0; r < B; ++r) {
    double m = -INFINITY;
    int has_nan = 0;
    for (int64_t k = 0; k < K; ++k) {
      double x = a[r * K + k];
      if (b) x = x + b[r * K + k];
      if (c) x = x - c[r * K + k];
      lw[r * K + k] = x;
      if (x != x) has_nan = 1;
      if (x > m) m = x;
    }
    if (has_nan) {
      lse[r] = NAN;
    } else if (!(m > -INFINITY && m < INFINITY)) {
      lse[r] = m;
    } else {
      double s = 0.0;
      for (int64_t k = 0; k < K; ++k) s += exp(lw[r * K + k] - m);
      lse[r] = m + log(s);
    }
  }
}
