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

/* ---- linear-Gaussian particle propagation (kernels K8 / K9 / K10) ----------------------------------
 *
 * What the reference's LGSSM callables compute with PyTorch — test/models/lgssm.py:40 (transition:
 * Normal(mult * previous_latents[-1], scale)), :52 (emission), :66-77 (proposal: a Linear layer of
 * [x_{t-1}, y_t]) — followed by state.sample (aesmc/state.py:61-111), state.log_prob
 * (aesmc/state.py:114-155) and the log-weight combine (aesmc/inference.py:112-126), for a model whose
 * locations are affine in the particles:  loc[b,k,j] = off[b,j] + sum_i w[j,i] x[b,k,i].
 * The kernels' contract fixes the arithmetic of the location to one chain of fused multiply-adds,
 * i ascending, starting from the offset (0 without one); fmaf()/fma() of libm are exactly that.
 * TYPE-generic through the macro below: _f32 works in float, _f64 in double.
 */
#define DEFINE_AFFINE(SUFFIX, T, FMA, LOG)                                                              \
  static void chain_##SUFFIX(const T *w, int64_t sj, int64_t si, const T *x, int64_t dout, int64_t din, \
                             T *acc) {                                                                  \
    for (int64_t j = 0; j < dout; ++j)                                                                  \
      for (int64_t i = 0; i < din; ++i) acc[j] = FMA(w[j * sj + i * si], x[i], acc[j]);                 \
  }                                                                                                     \
  /* K8: out = base + (off + W1 x1 + W2 x2); x2 / w2, off and base optional */                          \
  void smc_oracle_particle_affine_##SUFFIX(const T *x1, const T *w1, int64_t w1_sj, int64_t w1_si,      \
                                           int64_t d1, const T *x2, const T *w2, int64_t w2_sj,         \
                                           int64_t w2_si, int64_t d2, const T *off, int64_t off_sb,     \
                                           const T *base, T *out, int64_t B, int64_t K, int64_t dout) { \
    T acc[256];                                                                                          \
    for (int64_t b = 0; b < B; ++b)                                                                     \
      for (int64_t k = 0; k < K; ++k) {                                                                 \
        const int64_t n = b * K + k;                                                                    \
        for (int64_t j = 0; j < dout; ++j) acc[j] = off ? off[b * off_sb + j] : (T)0;                   \
        chain_##SUFFIX(w1, w1_sj, w1_si, x1 + n * d1, dout, d1, acc);                                   \
        if (x2) chain_##SUFFIX(w2, w2_sj, w2_si, x2 + n * d2, dout, d2, acc);                           \
        for (int64_t j = 0; j < dout; ++j) out[n * dout + j] = base ? base[n * dout + j] + acc[j] : acc[j]; \
      }                                                                                                 \
  }                                                                                                     \
  /* K9: x' = loc + eps * scale, product rounded before the sum (torch's rsample: loc + eps * scale) */ \
  void smc_oracle_affine_rsample_##SUFFIX(const T *src, const T *w, int64_t sj, int64_t si,             \
                                          const T *off, int64_t off_sb, const T *eps, T scale, T *out,  \
                                          int64_t B, int64_t K, int64_t dout, int64_t din) {            \
    T acc[256];                                                                                          \
    for (int64_t b = 0; b < B; ++b)                                                                     \
      for (int64_t k = 0; k < K; ++k) {                                                                 \
        const int64_t n = b * K + k;                                                                    \
        for (int64_t j = 0; j < dout; ++j) acc[j] = off ? off[b * off_sb + j] : (T)0;                   \
        chain_##SUFFIX(w, sj, si, src + n * din, dout, din, acc);                                       \
        for (int64_t j = 0; j < dout; ++j) {                                                            \
          const T noise = eps[n * dout + j] * scale;                                                    \
          out[n * dout + j] = acc[j] + noise;                                                           \
        }                                                                                               \
      }                                                                                                 \
  }                                                                                                     \
  /* K10: per term the squared distance as one fma chain from 0 (j ascending), then                       \
   * log N = (-q) / (2 sigma^2) - d (log sigma + log sqrt(2 pi)) — torch.distributions.Normal.log_prob   \
   * (-((v - mu)^2) / (2 sigma^2) - log sigma - log sqrt(2 pi)) summed over j with the common factor     \
   * taken out; the terms combine as (p + g) - q: aesmc/inference.py:125-126 */                          \
  static T term_sum_##SUFFIX(const T *v, const T *loc, T sigma, int64_t d) {                            \
    const T two_var = (T)2 * (sigma * sigma);                                                           \
    const T constant = (T)d * (LOG(sigma) + (T)0.9189385332046727);                                     \
    T q = (T)0;                                                                                         \
    for (int64_t j = 0; j < d; ++j) {                                                                   \
      const T diff = v[j] - loc[j];                                                                     \
      q = FMA(diff, diff, q);                                                                           \
    }                                                                                                   \
    return (-q) / two_var - constant;                                                                   \
  }                                                                                                     \
  void smc_oracle_affine_logweight_##SUFFIX(                                                            \
      const T *xprev, const T *x, const T *y, int64_t y_sb, const T *wp, const T *offp, int64_t offp_sb, \
      const T *wg, const T *offg, int64_t offg_sb, const T *wq, const T *offq, int64_t offq_sb, T sp,   \
      T sg, T sq, T *lw, int64_t B, int64_t K, int64_t dx, int64_t dy) {                                \
    T loc[256];                                                                                          \
    for (int64_t b = 0; b < B; ++b)                                                                     \
      for (int64_t k = 0; k < K; ++k) {                                                                 \
        const int64_t n = b * K + k;                                                                    \
        for (int64_t j = 0; j < dx; ++j) loc[j] = offp ? offp[b * offp_sb + j] : (T)0;                  \
        chain_##SUFFIX(wp, dx, 1, xprev + n * dx, dx, dx, loc);                                         \
        const T sum_p = term_sum_##SUFFIX(x + n * dx, loc, sp, dx);                                     \
        for (int64_t j = 0; j < dx; ++j) loc[j] = offq ? offq[b * offq_sb + j] : (T)0;                  \
        chain_##SUFFIX(wq, dx, 1, xprev + n * dx, dx, dx, loc);                                         \
        const T sum_q = term_sum_##SUFFIX(x + n * dx, loc, sq, dx);                                     \
        for (int64_t j = 0; j < dy; ++j) loc[j] = offg ? offg[b * offg_sb + j] : (T)0;                  \
        chain_##SUFFIX(wg, dx, 1, x + n * dx, dy, dx, loc);                                             \
        const T sum_g = term_sum_##SUFFIX(y + b * y_sb, loc, sg, dy);                                   \
        lw[n] = (sum_p + sum_g) - sum_q;                                                                \
      }                                                                                                 \
  }

DEFINE_AFFINE(f32, float, fmaf, logf)
DEFINE_AFFINE(f64, double, fma, log)


/* K13: the two-layer tanh net over particles  out = b2 + W2 tanh(c1[b] + W1 x[b,k])  that BASELINE.json's
 * nonlinear state-space model uses as its proposal (torch.cat + Linear + Tanh + Linear through the
 * reference's callable contract; its one-layer case is the reference's own test/models/lgssm.py:66-77).
 * Chains as above (fma, inputs ascending, started from the offset / bias); tanh is libm's. */
#define DEFINE_MLP(SUFFIX, T, FMA, TANH)                                                                \
  void smc_oracle_particle_mlp_##SUFFIX(const T *x, const T *w1, const T *off1, int64_t off1_sb,        \
                                        const T *w2, const T *b2, T *out, int64_t B, int64_t K,         \
                                        int64_t din, int64_t hid, int64_t dout) {                       \
    T hidden[256];                                                                                      \
    for (int64_t b = 0; b < B; ++b)                                                                     \
      for (int64_t k = 0; k < K; ++k) {                                                                 \
        const int64_t n = b * K + k;                                                                    \
        for (int64_t h = 0; h < hid; ++h) {                                                             \
          T acc = off1 ? off1[b * off1_sb + h] : (T)0;                                                  \
          for (int64_t i = 0; i < din; ++i) acc = FMA(w1[h * din + i], x[n * din + i], acc);            \
          hidden[h] = TANH(acc);                                                                        \
        }                                                                                               \
        for (int64_t o = 0; o < dout; ++o) {                                                            \
          T acc = b2 ? b2[o] : (T)0;                                                                    \
          for (int64_t h = 0; h < hid; ++h) acc = FMA(w2[o * hid + h], hidden[h], acc);                 \
          out[n * dout + o] = acc;                                                                      \
        }                                                                                               \
      }                                                                                                 \
  }
DEFINE_MLP(f32, float, fmaf, tanhf)
DEFINE_MLP(f64, double, fma, tanh)
