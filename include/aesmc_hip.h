/*
 * aesmc_hip.h — C ABI of libaesmc_hip.so, the MI355X (gfx950) kernels behind the batched SMC
 * inner loop of aesmc (reference: tuananhle7/aesmc, paths are into /root/reference).
 *
 * The reference has no FFI: its "operators" for this path are third-party CPU library calls made
 * from Python.  Each entry point below replaces one such call site; the Python host binds them
 * with ctypes (aesmc_amd/_lib.py) and INTEGRATION.md shows the stub a reference maintainer adds.
 *
 * Conventions (all entry points):
 *   - extern "C", return int status (AESMC_OK == 0), never throw, never allocate, never sync.
 *   - every pointer is a DEVICE pointer borrowed for the duration of the enqueued work;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *   - tensors are dense row-major [B, K, ...] unless a stride argument says otherwise;
 *   - `flags` is an optional device int32 word that kernels OR status bits into (AESMC_FLAG_*);
 *     the host reads it once per ELBO evaluation instead of synchronising per timestep.
 */
#ifndef AESMC_HIP_H_
#define AESMC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes --------------------------------------------------------------------------- */
#define AESMC_OK 0
#define AESMC_ERR_INVALID_ARGUMENT 1 /* null pointer, negative size, misaligned base pointer    */
#define AESMC_ERR_UNSUPPORTED 2      /* shape outside what the kernels handle (see each entry)  */
#define AESMC_ERR_LAUNCH 3           /* hipGetLastError() != hipSuccess after the launch        */
#define AESMC_ERR_WORKSPACE 4        /* workspace too small / missing                           */

/* ---- bits OR-ed into the device `flags` word ------------------------------------------------ */
#define AESMC_FLAG_NAN_LOG_WEIGHT 1  /* a log-weight was NaN  -> FloatingPointError, inference.py:244-245 */
#define AESMC_FLAG_DEGENERATE_ROW 2  /* a row had max = +-inf -> every index == K (reference: NaN CDF)     */
#define AESMC_FLAG_INDEX_OUT_OF_RANGE 4 /* gather saw idx < 0 or idx >= K (torch.gather would raise)     */
#define AESMC_FLAG_VALUE_OUTSIDE_SUPPORT 8 /* reserved for the host's deferred sample validation         */
#define AESMC_FLAG_UNSORTED_INDEX 16 /* gather backward was promised sorted indices and met a descent   */
#define AESMC_FLAG_INVALID_PARAMETER 32 /* reserved for the host's deferred distribution-argument validation */

/* ---- dtype tags ----------------------------------------------------------------------------- */
#define AESMC_F32 0
#define AESMC_F64 1

/* Library / build identification. */
/* 10000*major + 100*minor + patch.  History of the entry points (a binder checks this before it looks symbols up):
 *   100 (0.1.0)  rounds 1-2.
 *   200 (0.2.0)  removed aesmc_particle_mlp / aesmc_particle_mlp_max_hidden (the user's MLP stays in PyTorch);
 *                aesmc_set_step_parts / aesmc_set_sorted_backward_kernel became test hooks outside this header
 *                (aesmc_test_*); aesmc_affine_normal_propagate_drawn keeps its signature and now runs the fused launch
 *                (gather + noise + draw + log-weight terms in one kernel); added aesmc_affine_normal_propagate_wide
 *                (+ aesmc_affine_wide_dim, aesmc_affine_wide_workspace_bytes).
 *   500 (0.5.0)  aesmc_affine_normal_propagate_wide takes every width 17 .. 256 (observations 1 .. 256), dx != dy, any K
 *                (added aesmc_affine_wide_min_dim, aesmc_affine_wide_max_dim, aesmc_affine_wide_workspace_bytes_for);
 *                AESMC_FLAG_INVALID_PARAMETER reserved for the host's deferred distribution-argument validation;
 *                aesmc_particle_mlp is back (0.1.0 had it, 0.2.0 dropped it) WITH its backward: aesmc_particle_mlp_backward,
 *                aesmc_particle_mlp_backward_records (+ aesmc_particle_mlp_max_hidden); aesmc_particle_affine_tanh.
 *   501 (0.5.1)  added aesmc_affine_weight_pairs_scaled; aesmc_affine_weight_pairs_floats() grew by eight values (the
 *                three densities' constants and their tag behind the pairs: aesmc_affine_weight_pairs clears the tag);
 *                added aesmc_affine_normal_initial_step (K20: the first timestep's draw, emission location and
 *                log-weight in one launch).
 *   400 (0.4.0)  aesmc_affine_chain grew `pairs_in` / `pairs_out` (a run of backward steps builds the weight pairs once);
 *                added aesmc_wide_adjoint_tile, aesmc_wide_adjoint_scale, aesmc_wide_adjoint_merge
 *   300 (0.3.0)  added aesmc_affine_normal_propagate_drawn_paired, aesmc_affine_weight_pairs,
 *                aesmc_affine_weight_pairs_floats (round 5: packed multiply-adds in the fused propagation launch). */
int aesmc_version(void);
const char *aesmc_target_arch(void); /* "gfx950" */

/* The address kernels can use for a pointer into PINNED host memory (hipHostMalloc / PyTorch pin_memory=True), or
 * AESMC_ERR_UNSUPPORTED when it is not mapped into the device's address space.  aesmc/inference.py:250 draws one
 * uniform per batch row on the host before every resampling step: with this, `u` of aesmc_ancestor_index /
 * aesmc_resample_step may point at the pinned block the host wrote them into — 8 bytes per row read over the host
 * link — instead of at a device copy made by a launch of its own per timestep. */
int aesmc_host_device_pointer(const void *host_ptr, void **device_ptr);

/*
 * K1 — fused log-weight combine + per-row log-sum-exp.
 *   lw[b,k]  = lp_a[b,k] + lp_b[b,k] - lp_c[b,k]           (lp_b / lp_c may be NULL: term dropped)
 *   lse[b]   = log sum_k exp(lw[b,k])                      (out_lse may be NULL)
 * Replaces: the two elementwise torch ops at aesmc/inference.py:97-98 and :125-126, and the
 * torch.logsumexp of aesmc/inference.py:130 / :158 and aesmc/math.py:28 (per timestep, no stack).
 * out_lw may be NULL (pure row-LSE of lp_a).  out_lw may alias lp_a.  Inputs dense [B,K].
 * Special values follow torch.logsumexp: row of -inf -> -inf; any +inf -> +inf; any NaN -> NaN.
 */
int aesmc_logweight_lse(int dtype, const void *lp_a, const void *lp_b, const void *lp_c,
                        void *out_lw, void *out_lse, int64_t B, int64_t K, void *stream);

/*
 * K1 with a running sum over time — importance sampling with more than one timestep normalises the
 * SUM of the per-step log-weights (aesmc/inference.py:156-159: torch.sum of a stack of all T weight
 * tensors, then torch.logsumexp).  One launch per step instead:
 *   lw[b,k]      = lp_a + lp_b - lp_c          (as K1; out_lw may be NULL when the step's own
 *                                               weights are not wanted)
 *   out_acc[b,k] = acc_in[b,k] + lw[b,k]       (left to right over time: the order torch.sum takes
 *                                               over the leading dim of the stack)
 *   out_lse[b]   = log sum_k exp(out_acc[b,k]) (may be NULL; asked for at the last step only)
 * out_acc may alias acc_in.  Same special values as K1.
 */
int aesmc_logweight_accumulate(int dtype, const void *lp_a, const void *lp_b, const void *lp_c,
                               const void *acc_in, void *out_lw, void *out_acc, void *out_lse,
                               int64_t B, int64_t K, void *stream);

/*
 * K1 backward.  g[b,k] = grad_lw[b,k] + grad_lse[b] * exp(lw[b,k] - lse[b])   (either grad may be
 * NULL = zero).  Writes out_g (gradient w.r.t. lp_a and lp_b) and, if non-NULL, out_neg_g = -g
 * (gradient w.r.t. lp_c).  Replaces autograd of the ops listed under K1.
 */
int aesmc_logweight_lse_backward(int dtype, const void *lw, const void *lse, const void *grad_lw,
                                 const void *grad_lse, void *out_g, void *out_neg_g, int64_t B,
                                 int64_t K, void *stream);

/*
 * K2 — systematic ancestral resampling, one independent problem per batch row.
 *   w        = exp(log_w[b,:] - max_k log_w[b,:])                 (evaluated in float64)
 *   c[j]     = (w[0] + ... + w[j]) / (w[0] + ... + w[K-1])        (float64 segmented prefix scan)
 *   pos[k]   = (u[b] + k) / K                                     (float64, true division)
 *   idx[b,k] = #{ j : c[j] <= pos[k] }                            (== np.digitize(pos, c))
 * Replaces aesmc/inference.py:234-269 (numpy uniform draw excluded: `u` [B] float64 comes from the
 * host's numpy global RandomState so RNG consumption is unchanged) and aesmc/math.py:33-51 on the
 * numpy branch.  NaN anywhere -> AESMC_FLAG_NAN_LOG_WEIGHT (reference raises FloatingPointError);
 * rows whose max is +-inf -> every idx == K and AESMC_FLAG_DEGENERATE_ROW (reference: NaN CDF makes
 * np.digitize return K).  `ws` is needed only when K > aesmc_ancestor_index_lds_max_particles():
 * then it must hold aesmc_workspace_bytes(B, K) bytes.
 */
int aesmc_ancestor_index(int dtype, const void *log_w, const double *u, int64_t *out_idx,
                         int32_t *flags, int64_t B, int64_t K, void *ws, size_t ws_bytes,
                         void *stream);
int64_t aesmc_ancestor_index_lds_max_particles(void);
size_t aesmc_workspace_bytes(int64_t B, int64_t K);

/*
 * K3 — resample gather:  dst[b,k,:] = src[b, idx[b,k], :]  with `row_bytes` contiguous bytes per
 * particle.  src may be strided in its first two dims (byte strides src_stride_b / src_stride_k;
 * this covers the transposed time-0 latent of aesmc/state.py:102-103); dst is dense.
 * Replaces torch.gather at aesmc/state.py:179 (and aesmc/inference.py:226 with row_bytes = 8).
 * idx outside [0,K) never faults: it is clamped and AESMC_FLAG_INDEX_OUT_OF_RANGE is raised.
 */
int aesmc_resample_gather(const void *src, const int64_t *idx, void *dst, int32_t *flags,
                          int64_t B, int64_t K, int64_t row_bytes, int64_t src_stride_b,
                          int64_t src_stride_k, void *stream);

/*
 * K3 backward:  grad_src[b,j,:] = sum_{k : idx[b,k]==j} grad_out[b,k,:]   (`row_elems` elements of
 * `dtype` per particle, both tensors dense).  grad_src is fully overwritten (zero where a particle
 * has no offspring).  Replaces the scatter_add autograd of torch.gather (aesmc/state.py:179).
 * index_is_sorted != 0 promises idx non-decreasing along k (true for every output of
 * aesmc_ancestor_index): the segmented-sum kernel without atomics is used and a violation raises
 * AESMC_FLAG_UNSORTED_INDEX; 0 selects the order-agnostic kernel (one float atomic per run).
 */
int aesmc_resample_gather_backward(int dtype, const void *grad_out, const int64_t *idx,
                                   void *grad_src, int32_t *flags, int64_t B, int64_t K,
                                   int64_t row_elems, int index_is_sorted, void *stream);

/*
 * K4 — summed Normal log-density:
 *   out[b,k] = sum_{j<D} ( -((v-mu)^2) / (2 sigma^2) - log(sigma) - log(sqrt(2 pi)) )
 * with v = value[b,k,j], mu = loc[b,k,j], sigma = scale[b,k,j]; each operand is a [B,K,D] VIEW given
 * by its element strides (sb, sk, sd), 0 meaning broadcast along that dim.  out is dense [B,K].
 * Replaces, inside aesmc/state.py:114-155 (`state.log_prob`), torch.distributions.Normal.log_prob
 * followed by `.view(B, K, -1).sum(2)` (state.py:143-151) — about eleven launches per call —
 * element arithmetic in PyTorch's own order.
 */
int aesmc_normal_logprob_sum(int dtype, const void *value, const void *loc, const void *scale,
                             void *out, int64_t B, int64_t K, int64_t D, int64_t value_sb,
                             int64_t value_sk, int64_t value_sd, int64_t loc_sb, int64_t loc_sk,
                             int64_t loc_sd, int64_t scale_sb, int64_t scale_sk, int64_t scale_sd,
                             void *stream);

/*
 * K4 backward.  grad_out dense [B,K]; each non-NULL grad_* is written densely [B,K,D]:
 *   grad_value = -g z,  grad_loc = g z,  grad_scale = g ((v-mu)^2 / sigma^3 - 1/sigma),
 *   z = (v-mu)/sigma^2.  Reduction over broadcast dims is left to the caller (autograd's expand).
 */
int aesmc_normal_logprob_sum_backward(int dtype, const void *value, const void *loc,
                                      const void *scale, const void *grad_out, void *grad_value,
                                      void *grad_loc, void *grad_scale, int64_t B, int64_t K,
                                      int64_t D, int64_t value_sb, int64_t value_sk, int64_t value_sd,
                                      int64_t loc_sb, int64_t loc_sk, int64_t loc_sd,
                                      int64_t scale_sb, int64_t scale_sk, int64_t scale_sd,
                                      void *stream);

/*
 * K5 — the whole log-weight of one SMC step when the prior / transition, the emission and the
 * proposal are all Normal:
 *   lw[b,k] = sum_j log N(x; mu_p, s_p) + sum_j log N(y; mu_g, s_g) - sum_j log N(x; mu_q, s_q)
 * `views` points to eight [B,K,D] views (element strides, 0 = broadcast) in this order:
 *   0 x (latent, extent Dx)   1 mu_p   2 s_p      (prior or transition)
 *   3 y (observation, Dy)     4 mu_g   5 s_g      (emission)
 *   6 mu_q (extent Dx)        7 s_q               (proposal, evaluated at x)
 * Replaces three calls of K4 and the combine of K1 (aesmc/inference.py:112-126 via
 * aesmc/state.py:114-155); bit-identical to that route.
 * Covered operands:
 *   - extents Dx, Dy <= 64: any scales — one value for the whole tensor (all strides 0, e.g.
 *     Normal(loc, 0.7): the fastest kernels), a per-dimension vector, or a full [B,K,D] view;
 *   - extents above 64: scalar scales only, and only when Dx and Dy take the same lane team in K4,
 *     each row is a multiple of 16 bytes, contiguous and 16-byte aligned.
 * Anything else returns AESMC_ERR_UNSUPPORTED and the caller takes the K4 + K1 route (same numbers).
 */
typedef struct aesmc_view3 {
  const void *ptr;
  int64_t stride_b, stride_k, stride_d; /* element strides, 0 = broadcast */
} aesmc_view3;

int aesmc_normal_logweight(int dtype, const aesmc_view3 *views, void *out_lw, int64_t B, int64_t K,
                           int64_t Dx, int64_t Dy, void *stream);

/* K5 backward: gradients of aesmc_normal_logweight with respect to the values, the locations and
 * the scales, each written densely over [B,K,Dx] (grad_x, grad_mu_p, grad_mu_q, grad_s_p, grad_s_q)
 * or [B,K,Dy] (grad_y, grad_mu_g, grad_s_g); NULL outputs are skipped; a broadcast operand's
 * gradient is reduced by the caller (autograd's expand-backward).  `views` as for the forward (any
 * scales), grad_lw [B,K].  Bit-identical to three aesmc_normal_logprob_sum_backward launches (with
 * -grad_lw for the proposal term) and the add grad_x = grad_x_p + grad_x_q.
 */
int aesmc_normal_logweight_backward(int dtype, const aesmc_view3 *views, const void *grad_lw, void *grad_x,
                                    void *grad_mu_p, void *grad_y, void *grad_mu_g, void *grad_mu_q,
                                    void *grad_s_p, void *grad_s_g, void *grad_s_q, int64_t B, int64_t K,
                                    int64_t Dx, int64_t Dy, void *stream);

/* K5 backward fused with K1's: the gradient of  lse[b] = logsumexp_k lw[b,k]  (and, optionally, of lw
 * itself) with respect to K5's operands in one launch —
 *   g[b,k] = grad_lse[b] * exp(lw[b,k] - lse[b])  (+ grad_lw[b,k] when grad_lw is not NULL)
 * is formed per element where aesmc_normal_logweight_backward would read grad_lw, with K1's own
 * operations, so the results equal aesmc_logweight_lse_backward followed by
 * aesmc_normal_logweight_backward bit for bit, without the [B,K] round trip and the launch.
 * `lw` is what aesmc_normal_logweight produced for `views`; lse, grad_lse are [B].  Replaces the
 * autograd chain of aesmc/inference.py:112-132 for one timestep. */
int aesmc_normal_logweight_lse_backward(int dtype, const aesmc_view3 *views, const void *lw, const void *lse,
                                        const void *grad_lse, const void *grad_lw, void *grad_x,
                                        void *grad_mu_p, void *grad_y, void *grad_mu_g, void *grad_mu_q,
                                        void *grad_s_p, void *grad_s_g, void *grad_s_q, int64_t B, int64_t K,
                                        int64_t Dx, int64_t Dy, void *stream);

/* Fused resampling step — K2, plus two optional by-products of having the whole batch row in one
 * workgroup:
 *   out_lse[b] = logsumexp_k log_w[b,k]  (dtype of log_w; the row's term of log Z,
 *                aesmc/inference.py:130-132), computed in float64 from K2's own max and sum — the
 *                special rows give what K1 gives (NaN, +inf, -inf);
 *   dst[b,k,:] = src[b, out_idx[b,k], :]  (K3's contract for ONE payload tensor,
 *                aesmc/state.py:158-183), with the indices taken from LDS instead of HBM.
 * `out_lse` may be NULL; `src` and `dst` are NULL together.  out_idx is always written and equals
 * aesmc_ancestor_index's bit for bit; dst equals aesmc_resample_gather's (rows of a degenerate
 * batch row: source row K-1, as K3's clamp gives).  Returns AESMC_ERR_UNSUPPORTED — caller takes
 * K2 then K3 — when K exceeds aesmc_ancestor_index_lds_max_particles(), or the payload rows are
 * not multiples of 4 bytes / dst is not 16-byte aligned with a 16-byte batch-row pitch.
 */
int aesmc_resample_step(int dtype, const void *log_w, const double *u, int64_t *out_idx, void *out_lse,
                        const void *src, void *dst, int32_t *flags, int64_t B, int64_t K,
                        int64_t row_bytes, int64_t src_stride_b, int64_t src_stride_k, void *stream);

/* K2 with the children ranges as a by-product: aesmc_resample_step without a payload, plus
 *   out_child_end[b,k] = #{k' : out_idx[b,k'] <= k}      (int32 [B,K])
 * — the ancestor indices are non-decreasing along k, so the children of particle k are the positions
 * [out_child_end[b,k-1], out_child_end[b,k]) (0 for k = 0): what torch.gather's backward (aesmc/state.py:179) sums
 * per particle, handed to aesmc_affine_step_backward_resampled so that it forms those sums where it consumes them.
 * The scan has the value in a register anyway: 4 more bytes per particle written. */
int aesmc_resample_step_ranges(int dtype, const void *log_w, const double *u, int64_t *out_idx, void *out_lse,
                               int32_t *out_child_end, int32_t *flags, int64_t B, int64_t K, void *stream);

/* K6 — reparameterised Normal draw  out[b,k,j] = loc[b,k,j] + eps[b,k,j] * scale[b,k,j].
 *
 * Replaces the broadcast multiply and the add that follow the noise draw in
 * torch.distributions.Normal.rsample as aesmc/state.py:61-111 (`state.sample`) calls it; the
 * product is rounded before the sum, so the result equals eager PyTorch bit for bit.  `out` is
 * dense [B,K,D]; `eps` (the caller's standard-normal noise), `loc` and `scale` are [B,K,D] views
 * by element strides (0 = broadcast).  eps must be dense in [B,K,D] order or in [K,B,D] order
 * (stride_b = D, stride_k = B*D: the BATCH_EXPANDED draw of aesmc/state.py:102-103, written here
 * directly into [B,K,D] order); other eps layouts return AESMC_ERR_UNSUPPORTED.  `out` must not
 * alias the inputs.
 */
int aesmc_normal_rsample(int dtype, const aesmc_view3 *eps, const aesmc_view3 *loc,
                         const aesmc_view3 *scale, void *out, int64_t B, int64_t K, int64_t D,
                         void *stream);

/* K6 with the noise formed in the launch: out[b,k,j] = loc[b,k,j] + n[b,k,j] * scale[b,k,j], where n is element
 * (b K + k) D + j of the float32 tensor `torch.empty([B,K,D]).normal_()` would hold on this device for a generator at
 * (seed, offset) with ATen's launch geometry `threads` (see aesmc_philox_normal_fill: the same stream, the same
 * Box-Muller, `variant` and `rng_state` as there) — `Normal.rsample` of aesmc/state.py:98 without the noise tensor's
 * round trip through HBM; the caller advances the generator by what `normal_` would have consumed.  float32 only
 * (AESMC_ERR_UNSUPPORTED for float64 and for B K D >= 2^32); `out` is dense [B,K,D], loc and scale [B,K,D] views by
 * element strides (0 = broadcast).  The [K,B,D] noise order of a BATCH_EXPANDED draw is not offered here. */
int aesmc_normal_rsample_drawn(int dtype, const aesmc_view3 *loc, const aesmc_view3 *scale, void *out, int64_t B,
                               int64_t K, int64_t D, uint64_t seed, uint64_t offset, int64_t threads, int variant,
                               const uint64_t *rng_state, void *stream);

/* K7 — weighted particle summaries of a batch row in one pass:
 *   w = softmax_k(log_w[b,:]);  out_mean[b,j] = sum_k w[k] value[b,k,j];
 *   out_second[b,j] = sum_k w[k] value[b,k,j]^2;  out_log_ess[b] = 2 lse(log_w) - lse(2 log_w).
 * Replaces the per-particle Python loop of aesmc/statistics.py:7-76 (empirical_mean; the variance
 * is out_second - out_mean^2 as at statistics.py:75-76) and the two log-sum-exps of
 * aesmc/statistics.py:79-91.  `value` is a [B,K,D] view by element strides (NULL: only the ESS);
 * any of the three outputs may be NULL.  Rows whose weights cannot be normalised (NaN, no finite
 * maximum) give NaN, as the reference's softmax does.  With few batch rows a row is cut into
 * slices of particles, one workgroup each, whose records are merged by a second launch: `ws` must
 * then hold aesmc_particle_summary_workspace_bytes(dtype, B, K, D) bytes (0 = not needed), else
 * AESMC_ERR_WORKSPACE.  Not differentiable: callers that need gradients keep the PyTorch
 * expression.
 */
size_t aesmc_particle_summary_workspace_bytes(int dtype, int64_t B, int64_t K, int64_t D);
int aesmc_particle_summary(int dtype, const void *log_w, const aesmc_view3 *value, void *out_log_ess,
                           void *out_mean, void *out_second, int64_t B, int64_t K, int64_t D, void *ws,
                           size_t ws_bytes, void *stream);

/* ---- linear-Gaussian particle propagation (K8 / K9 / K10) ---------------------------------------
 *
 * An LGSSM written against the reference's callable contract (aesmc/inference.py:20-46) — its own
 * test model included: test/models/lgssm.py:40 (transition, `mult * previous_latents[-1]`), :52
 * (emission) and :66-77 (proposal, a Linear layer of [x_{t-1}, y_t]) — builds every step's
 * distributions as Normal(loc = W x + c, scale) with x the [B,K,d] particles.  `aesmc_affine_map`
 * describes one such location without materialising it:
 *     loc[b,k,j] = offset[b, j] + sum_i weight[j, i] * x[b,k,i]         (dout, din <= aesmc_affine_max_dim())
 * evaluated as ONE chain of fused multiply-adds per element, i ascending, starting from the offset
 * (zero when offset == NULL) — the same chain in all three entry points below, so they agree bit for
 * bit with each other (oracle/smc_core.c restates it with fma()).
 */
typedef struct aesmc_affine_map {
  const void *weight;       /* [dout, din] by element strides; a transposed view is swapped strides */
  int64_t stride_out, stride_in;
  const void *offset;       /* NULL, or offset[b * offset_stride_b + j]: [dout] (stride 0) or [B, dout] */
  int64_t offset_stride_b;
  int64_t dout, din;
} aesmc_affine_map;

int64_t aesmc_affine_max_dim(void); /* 16: larger maps are proper GEMMs and stay library calls */

/* K8 — materialise an affine location (or apply its adjoint):
 *   out[b,k,:] = base[b,k,:] + (offset + W1 x1[b,k,:] + W2 x2[b,k,:])      x2 / m2 and base optional
 * x1 [B,K,m1->din], x2 [B,K,m2->din], base and out [B,K,dout] dense and 16-byte aligned; m2 carries a
 * weight only (its offset is ignored) and m2->dout == m1->dout.  The chain runs through W1's terms,
 * then W2's; base is added last.  Replaces the `[B*K, d] x [d, d]` matmul (+ broadcast add) of the
 * model callables named above, and — with transposed weight views — the input-gradient matmuls of
 * their autograd.  out may alias base. */
int aesmc_particle_affine(int dtype, const void *x1, const aesmc_affine_map *m1, const void *x2,
                          const aesmc_affine_map *m2, const void *base, void *out, int64_t B, int64_t K,
                          void *stream);
/* ... and tanh(location) from the same launch (0.5.0): the device library's tanh — the one torch.tanh calls — applied to the
 * bits the plain launch would store; `tanh(A x_{t-1})` of a nonlinear transition without an element-wise launch behind K8. */
int aesmc_particle_affine_tanh(int dtype, const void *x1, const aesmc_affine_map *m1, const void *x2,
                          const aesmc_affine_map *m2, const void *base, void *out, int64_t B, int64_t K,
                          void *stream);

/* K9 — reparameterised draw from Normal(offset + W source, scale) without the location in HBM:
 *   out[b,k,j] = loc[b,k,j] + eps[b,k,j] * scale          (product rounded before the sum, as K6)
 * `scale` points to ONE value on the device; source [B,K,din], eps and out [B,K,dout] dense, 16-byte
 * aligned, out distinct from both inputs.  Replaces the proposal's matmul + `state.sample`
 * (aesmc/state.py:61-111) for t > 0; equals K8 followed by K6 bit for bit. */
int aesmc_affine_normal_rsample(int dtype, const void *source, const aesmc_affine_map *map, const void *eps,
                                const void *scale, void *out, int64_t B, int64_t K, void *stream);

/* K10 — one SMC step's log-weight (aesmc/inference.py:112-126) for a linear-Gaussian model:
 *   lw[b,k] = sum_j log N(x[b,k,j];  transition(x_prev[b,k,:])_j, scale_p)
 *           + sum_j log N(y[b,j];    emission(x[b,k,:])_j,        scale_g)
 *           - sum_j log N(x[b,k,j];  proposal(x_prev[b,k,:])_j,   scale_q)
 * x_prev, x [B,K,dx] dense and 16-byte aligned; y[b * y_stride_b + j], j < dy = emission->dout (the
 * observation, one row per batch element, not expanded over particles); the three scales point to ONE
 * device value each.  transition and proposal map dx -> dx, emission dx -> dy.  Replaces the three
 * matmuls of the callables and the three `state.log_prob` calls (aesmc/state.py:114-155) + combine.
 * Arithmetic: the three locations by the chain above; per term  q = sum_j (v_j - loc_j)^2  as one fma
 * chain from 0 (j ascending) and  log N = (-q) / (2 scale^2) - d (log scale + log sqrt(2 pi))  —
 * torch.distributions.Normal.log_prob summed over j with the common factors taken out (one division per
 * term instead of one per element); the terms combine as (p + g) - q.  Agrees with K8 x 3 followed by K5
 * to rounding (float64: ~1e-15 relative). */
int aesmc_affine_normal_logweight(int dtype, const void *x_prev, const void *x, const void *y,
                                  int64_t y_stride_b, const aesmc_affine_map *transition,
                                  const aesmc_affine_map *emission, const aesmc_affine_map *proposal,
                                  const void *scale_p, const void *scale_g, const void *scale_q, void *out_lw,
                                  int64_t B, int64_t K, void *stream);

/* K15 — K9 and K10 in one pass: the proposal's reparameterised draw and the step's log-weight together,
 *   out_x[b,k,:] = loc_q(x_prev[b,k,:]) + eps[b,k,:] * scale_q         (K9's arithmetic, the same bits)
 *   out_lw[b,k]  = K10's log-weight of (x_prev, out_x, y)              (the same bits as K10 on out_x)
 * from one read of x_prev and of the noise `eps` [B,K,dx] (dense, 16-byte aligned; out_x likewise and
 * distinct from x_prev; it MAY be the noise's own buffer).  For a model whose transition, emission and
 * proposal are all affine Normals nothing between aesmc/inference.py:106 (`state.sample(proposal)`) and
 * :125-126 (the log-weight) needs x_t's VALUES — the callables only describe distributions in terms of
 * it — so the draw can wait for the launch that weighs it: 520 MB per step at B=1024 K=4096 d=10
 * instead of K9's 503 + K10's 352. */
int aesmc_affine_normal_propagate(int dtype, const void *x_prev, const void *eps, const void *y, int64_t y_stride_b,
                                  const aesmc_affine_map *transition, const aesmc_affine_map *emission,
                                  const aesmc_affine_map *proposal, const void *scale_p, const void *scale_g,
                                  const void *scale_q, void *out_x, void *out_lw, int64_t B, int64_t K, void *stream);

/* K15 through the ancestor indices — the resampling gather folded into the propagation:
 *   x_prev[b,k,:] = x_src[b, ancestors[b,k], :]      (aesmc/inference.py:102-111, state.py:179 `torch.gather`)
 * is formed while the tile is staged and never written: K15 of that x_prev, the same bits as
 * aesmc_resample_gather followed by aesmc_affine_normal_propagate.  `x_src` [B,K,dx] dense and 16-byte
 * aligned, `ancestors` int64 [B,K] (what aesmc_ancestor_index / aesmc_resample_step wrote); an index outside
 * [0, K) is clamped and raises AESMC_FLAG_INDEX_OUT_OF_RANGE in `flags` (may be NULL).  One read of the
 * surviving rows of x_src and of the noise, one write of x_t: 8 dx + 4 dx + 12 B per particle instead of the
 * gather's 8 dx + 8 plus K15's 12 dx + 4.  AESMC_ERR_UNSUPPORTED (caller: gather, then K15) with fewer than
 * ~43 particles per batch row or B K >= 2^31. */
int aesmc_affine_normal_propagate_resampled(
    int dtype, const void *x_src, const int64_t *ancestors, const void *eps, const void *y, int64_t y_stride_b,
    const aesmc_affine_map *transition, const aesmc_affine_map *emission, const aesmc_affine_map *proposal,
    const void *scale_p, const void *scale_g, const void *scale_q, void *out_x, void *out_lw, int32_t *flags,
    int64_t B, int64_t K, void *stream);

/* K16 — K15 through the ancestor indices with the noise drawn INSIDE the launch:
 *   eps = the float32 tensor `torch.empty([B,K,dx]).normal_()` holds for a generator at (seed, offset) — see
 *         aesmc_philox_normal_fill below: ATen's launch geometry `threads`, rocRAND's Philox4x32-10 + Box-Muller
 *   x_prev[b,k,:] = x_src[b, ancestors[b,k], :]        (ancestors NULL: x_prev = x_src)
 *   out_x, out_lw = aesmc_affine_normal_propagate(x_prev, eps, ...)        — the same bits
 * Replaces, for one SMC step of a linear-Gaussian model, aesmc/inference.py:102-126: `state.resample` of the
 * newest latent (state.py:179), `state.sample(proposal)` (state.py:98: `_standard_normal`, loc + eps * scale)
 * and the three `state.log_prob` calls.  Neither the noise nor the resampled latent nor any location touches
 * HBM: 8 + 4 dx bytes in, 4 dx + 4 out per particle.  The caller advances the generator by
 * 4 * ceil(B K dx / (4 threads)), as `normal_` would.  float32 only (PyTorch draws float64 noise by another
 * route).  `rng_state` (NULL outside a hipGraph): two uint64 in DEVICE memory, (seed, offset), read by the kernel
 * when it runs — a captured launch replays with whatever generator state the host uploaded before the replay;
 * `seed` is then ignored and `offset` is relative to it (what the captured region had consumed before this
 * launch).  AESMC_ERR_UNSUPPORTED (caller: aesmc_philox_normal_fill, then the launches above) with fewer than
 * 128 particles per batch row (64 below 2^20 particles), dx = 1, or B K dx >= 2^32. */
int aesmc_affine_normal_propagate_drawn(
    const void *x_src, const int64_t *ancestors, const void *y, int64_t y_stride_b,
    const aesmc_affine_map *transition, const aesmc_affine_map *emission, const aesmc_affine_map *proposal,
    const void *scale_p, const void *scale_g, const void *scale_q, void *out_x, void *out_lw, int32_t *flags,
    int64_t B, int64_t K, uint64_t seed, uint64_t offset, int64_t threads, const uint64_t *rng_state, void *stream);

/* The same launch with the three maps' weights also at hand as interleaved pairs — `weight_pairs`:
 * aesmc_affine_weight_pairs_floats() float32 values written by aesmc_affine_weight_pairs for the SAME three maps
 * (their values at the time this launch runs: rebuild after the weights change — once per ELBO evaluation, and inside
 * every hipGraph capture) — so that the location chains of two outputs advance together, one v_pk_fma_f32 per input
 * (each half the fused multiply-add of the scalar chain, same order: the same bits, half the multiply-add
 * instructions), and so that weights of any strides (a transposed view) take this form.  NULL: as
 * aesmc_affine_normal_propagate_drawn.  Reference: aesmc/inference.py:102-126 as above. */
int aesmc_affine_normal_propagate_drawn_paired(
    const void *x_src, const int64_t *ancestors, const void *y, int64_t y_stride_b,
    const aesmc_affine_map *transition, const aesmc_affine_map *emission, const aesmc_affine_map *proposal,
    const void *scale_p, const void *scale_g, const void *scale_q, void *out_x, void *out_lw, int32_t *flags,
    int64_t B, int64_t K, uint64_t seed, uint64_t offset, int64_t threads, const uint64_t *rng_state,
    const void *weight_pairs, void *stream);
/* pairs[map][jp][i] = (W[2 jp][i], W[2 jp + 1][i]), zero where a row or column does not exist, for the transition,
 * emission and proposal maps (float32, extents <= 16, any strides), into `out_pairs` (16-byte aligned device memory of
 * aesmc_affine_weight_pairs_floats() floats).  One small launch. */
int64_t aesmc_affine_weight_pairs_floats(void);
int aesmc_affine_weight_pairs(const aesmc_affine_map *transition, const aesmc_affine_map *emission,
                              const aesmc_affine_map *proposal, void *out_pairs, void *stream);
/* (0.5.1) The same launch with the three scales at hand (device pointers, one float32 each, the values
 * aesmc_affine_normal_propagate_drawn_paired will be handed): behind the pairs it also leaves the launch-wide constants
 * of the three Normal densities of aesmc/state.py:98 — 2 s^2 and d (log s + log(2 pi) / 2) for transition, emission,
 * proposal — and a tag naming the extents they were formed for.  The fused launch then reads them as scalars instead of
 * taking three logarithms in every wavefront (5 % of the vector instructions of a kernel that saturates the vector
 * pipe); the log-weights' bits do not change (the same expressions, evaluated once).  Rebuild whenever a weight OR a
 * scale changes; a buffer written by aesmc_affine_weight_pairs carries a cleared tag and the launch forms the constants
 * itself, as before. */
int aesmc_affine_weight_pairs_scaled(const aesmc_affine_map *transition, const aesmc_affine_map *emission,
                                     const aesmc_affine_map *proposal, const void *scale_p, const void *scale_g,
                                     const void *scale_q, void *out_pairs, void *stream);

/* K20 (0.5.1) — the FIRST step of a run whose proposal and prior do not depend on a latent and whose emission is
 * linear-Gaussian in the latent being drawn (aesmc/inference.py:79-98 with the reference's model style,
 * test/models/lgssm.py: `initial()` a Normal, the time-0 proposal Normal(f(y_0), s) BATCH_EXPANDED, the emission
 * Normal(C x_0 + g, s_g)), float32, latent and observation rows of 1 .. 16 values:
 *   out_x[b,k,:] = loc_q[b,:] + eps[k,b,:] * scale_q[b,:]        (state.py:98, :102-103: `rsample((K,))`, transposed)
 *   out_lw[b,k]  = (sum_j log N(out_x; loc_p, scale_p) + sum_j log N(y_b; C out_x + g, scale_g)) - sum_j log N(out_x; loc_q, scale_q)
 * `eps`: the noise in the reference's order, [K, B, dx] contiguous.  The six views are [B, K, d] views whose stride_k is 0
 * (anything constant along the particles: a scalar, a per-column vector, one row per batch element — any mix); the
 * emission map is C [dy, dx] (any strides) with offset g NULL, [dy] or [B, dy].  One launch in place of
 * aesmc_normal_rsample + aesmc_particle_affine + aesmc_normal_logweight, and the bits of those three: the draw's
 * product is rounded before its sum, the location is one fma chain per output (inputs ascending, started from the
 * offset), an element's log-density (-(d d)) / (2 s s) - log s - log(2 pi) / 2, summed over j ascending from zero.
 * AESMC_ERR_UNSUPPORTED (a view that varies along the particles, rows wider than 16): the caller takes the three. */
int aesmc_affine_normal_initial_step(const void *eps, const aesmc_view3 *loc_q, const aesmc_view3 *scale_q,
                                     const aesmc_view3 *loc_p, const aesmc_view3 *scale_p, const aesmc_view3 *y,
                                     const aesmc_affine_map *emission, const aesmc_view3 *scale_g, void *out_x,
                                     void *out_lw, int64_t B, int64_t K, void *stream);

/* K17 + K18 — one SMC step of a linear-Gaussian model whose latent and observation rows hold 128 float32 values
 * (BASELINE.json configs[4]), the three 128 x 128 maps on the fp32 matrix cores:
 *   x_prev[b,k,:] = x_src[b, ancestors[b,k], :]                              (ancestors NULL: x_prev = x_src)
 *   out_x  = (offset_q + Q x_prev) + eps * s_q                               (each location ONE fma chain per element,
 *   out_lw = (log N(out_x; offset_p + A x_prev, s_p) + log N(y; offset_g + C out_x, s_g)) - log N(out_x; loc_q, s_q)
 *                                                                             inputs ascending: out_x has the C oracle's bits)
 * Replaces aesmc/inference.py:102-126 for one timestep: `state.resample` of the newest latent (state.py:179),
 * `state.sample(proposal)` given its noise `eps` [B,K,128] (state.py:98) and the three `state.log_prob` calls
 * (state.py:114-155) — three library GEMMs, their offsets' broadcast adds, the draw and the log-weight kernel before.
 * Weights [128,128] row-major contiguous (what an nn.Linear holds), offsets NULL / [128] / [B,128], scales one value
 * each, K a multiple of 32; `ws`: aesmc_affine_wide_workspace_bytes(B, K) bytes (two sums per particle between the two
 * launches).  `eps` NULL: the noise is formed in the launch — element e of `torch.empty([B,K,128]).normal_()` for the
 * generator at (seed, offset) with ATen's `threads` (rng_state: as for aesmc_affine_normal_propagate_drawn); covered
 * when K is a multiple of 4 * threads / 128 (configs[4]: threads = 524288, K = 16384), else AESMC_ERR_UNSUPPORTED: the
 * caller fills the noise (aesmc_philox_normal_fill) and passes it.  The squared distances are summed per lane and then over a particle's four lanes: equal to the C oracle's
 * single chain to rounding (2e-6 relative in the tests), not bit for bit.  AESMC_ERR_UNSUPPORTED for every other shape
 * (the caller keeps the route through aesmc_normal_rsample / aesmc_normal_logweight). */
int64_t aesmc_affine_wide_dim(void);      /* 128: the extent with the noise in the launch and the backward pieces below */
size_t aesmc_affine_wide_workspace_bytes(int64_t B, int64_t K);      /* ... at that extent */
/* 0.5.0: the same entry point takes ANY latent width dx with aesmc_affine_wide_min_dim() = 17 <= dx <=
 * aesmc_affine_wide_max_dim() = 256 and any observation width 1 <= dy <= 256 (dx != dy allowed: transition and proposal
 * [dx,dx], emission [dy,dx], y [B,dy], x / eps / out_x [B,K,dx]) and any K — aesmc/state.py:61-183 is dimension-agnostic
 * (rows of at most 16 values are the item kernels': aesmc_affine_normal_propagate_drawn).  Rows are padded to the next of
 * 32 / 48 / 64 / 96 / 128 / 192 / 256 inside the launch (zero inputs leave an fma chain as it is: out_x keeps the C oracle's
 * bits), rows wider than 128 are cut into chunks of output rows (the maps' weights stay LDS-resident per chunk; a
 * particle's partial squared distances meet in `ws`, added in ascending chunk order, for dy > 128 by a third small
 * launch), a batch row's last tile is masked when K is not a multiple of 32, and widths that are not multiples of 4 (rows
 * that are not whole 16-byte pieces) move their pieces of four element by element.  `eps` must be given at these shapes
 * (NULL: AESMC_ERR_UNSUPPORTED — the caller fills the noise with aesmc_philox_normal_fill); `ws`:
 * aesmc_affine_wide_workspace_bytes_for(B, K, dx, dy) bytes. */
int64_t aesmc_affine_wide_min_dim(void);  /* 17 */
int64_t aesmc_affine_wide_max_dim(void);  /* 256 */
size_t aesmc_affine_wide_workspace_bytes_for(int64_t B, int64_t K, int64_t dx, int64_t dy);
int aesmc_affine_normal_propagate_wide(
    const void *x_src, const int64_t *ancestors, const void *eps, const void *y, int64_t y_stride_b,
    const aesmc_affine_map *transition, const aesmc_affine_map *emission, const aesmc_affine_map *proposal,
    const void *scale_p, const void *scale_g, const void *scale_q, void *out_x, void *out_lw, void *ws, size_t ws_bytes,
    int32_t *flags, int64_t B, int64_t K, uint64_t seed, uint64_t offset, int64_t threads, const uint64_t *rng_state,
    void *stream);

/* The element-wise parts of the WIDE step's backward (rows of aesmc_affine_wide_dim() = 128 float32 values), between the
 * [B K,128] x [128,128] products a binder runs through its GEMM library — autograd of aesmc/state.py:114-155 (`log_prob` of
 * the emission and the transition) and :98 (the reparameterised draw) for one timestep, recomputed from x_{t-1}, the
 * ancestors, x_t and the log-weights:
 *   aesmc_wide_adjoint_scale: `u` [B,K,128] holds a residual d = value - location (a product's epilogue) — or, with
 *     `base` [B,128] (row stride base_stride_b, a multiple of 4) given, the LOCATION itself, and d = base[b] - u (base = the
 *     value's row minus the map's offset: no [B,K,128] copy of a broadcast row for a product's epilogue to start from); in place
 *     u <- weight[b,k] d / scale^2  (the location's adjoint; weight = the gradient arriving at the particle's log-weight),
 *     out_sq[b,k] = sum_j d_j^2  (the scale's gradient is weight (out_sq / scale^3 - 128 / scale) summed),
 *     out_rows[b, tile, :] = the new u summed over a tile of aesmc_wide_adjoint_tile() = 256 particles (the offset's
 *     gradient is their sum over a row's K / 256 tiles).  out_sq / out_rows may be NULL.
 *   aesmc_wide_adjoint_merge: the same for the transition's residual in `u_p` (or, `value` [B,K,128] given, its LOCATION:
 *     d = (value - base[b]) - u_p, base NULL or as above), and the gradient arriving at x_t in `at_x` takes the term the
 *     transition's density contributes: at_x <- (add + at_x) - u_p (`add` [B,K,128]: what later steps sent to x_t, or NULL);
 *     out_rows_x holds the tiles' sums of the new at_x (the
 *     proposal's offset: the draw carries what arrives at x_t to it).
 * One read and one write per tensor where the PyTorch operations they replace made eight passes and three copies.  float32,
 * dense, 16-byte aligned, K a multiple of 256 (else AESMC_ERR_UNSUPPORTED: the caller keeps its own operations). */
int64_t aesmc_wide_adjoint_tile(void);      /* 256 */
int aesmc_wide_adjoint_scale(void *u, const void *weight, const void *scale, const void *base, int64_t base_stride_b,
                             void *out_sq, void *out_rows, int64_t B, int64_t K, void *stream);
int aesmc_wide_adjoint_merge(void *u_p, void *at_x, const void *weight, const void *scale, const void *value,
                             const void *base, int64_t base_stride_b, const void *add, void *out_sq, void *out_rows_p,
                             void *out_rows_x, int64_t B, int64_t K, void *stream);

/* K13 — a learned proposal net over the particles: the two-layer tanh MLP
 *   out[b,k,:] = layer2->offset + W2 tanh( layer1->offset[b,:] + W1 x[b,k,:] )
 * with W1 [H, din] (din <= 16, H <= aesmc_particle_mlp_max_hidden() = 64), W2 [dout, H] (dout <= 16);
 * layer1->offset is [H] or [B, H] (the per-row part of the first layer: its bias and the observation's
 * columns of the weight applied to y_t), layer2->offset [dout] or NULL.  x, out dense and 16-byte
 * aligned.  Replaces, in a model whose proposal is such a net of [x_{t-1}, y_t] (BASELINE.json configs[3]'s
 * nonlinear state-space model; the reference's own proposal at test/models/lgssm.py:66-77 is its
 * one-layer case), torch.cat + Linear + tanh + Linear: two GEMMs with [B,K,H] round trips through HBM.
 * Returns AESMC_ERR_UNSUPPORTED (caller keeps the PyTorch expression) beyond those extents or with
 * fewer than ~43 particles per batch row.
 * K13b — its backward, recomputing the hidden layer (nothing of [B,K,H] is stored): for grad_out [B,K,dout]
 *   grad_x[b,k,:] = W1^T dh,  dh = (W2^T grad_out) (1 - h^2)                     (dense [B,K,din]; may be NULL)
 *   rec_w1 / rec_w2: [aesmc_particle_mlp_backward_records(B, K)][ceil(H / 16)][256] — per wavefront of the launch the
 *     16 x 16 partials of grad_W1 (rows: 16 hidden units of the chunk, columns: inputs) and grad_W2 (rows: outputs,
 *     columns: the chunk's hidden units), contracted over the particles on the matrix cores; the binder adds the records
 *     (a fixed order: reproducible) and cuts them to [H, din] / [dout, H];
 *   rows: [B K / 256][4][16 ceil(H / 16)] — sum of dh over each wavefront's 64 particles (the gradient of a per-row
 *     layer1->offset is their sum over a row's K / 64 groups); may be NULL.
 *   The output bias' gradient is grad_out summed over the particles (the binder's reduction).
 * K a multiple of 256 (a tile of 256 particles inside one batch row) and din <= 15, else AESMC_ERR_UNSUPPORTED. */
int64_t aesmc_particle_mlp_max_hidden(void);
int aesmc_particle_mlp(int dtype, const void *x, const aesmc_affine_map *layer1, const aesmc_affine_map *layer2,
                       void *out, int64_t B, int64_t K, void *stream);
int64_t aesmc_particle_mlp_backward_records(int64_t B, int64_t K);
int aesmc_particle_mlp_backward(int dtype, const void *x, const void *grad_out, const aesmc_affine_map *layer1,
                                const aesmc_affine_map *layer2, void *grad_x, void *rec_w1, void *rec_w2, void *rows,
                                int64_t B, int64_t K, void *stream);

/* K11 — the adjoint of an affine location  loc = offset + W x  for an incoming gradient grad [B,K,dout]:
 *   out_grad_x[b,k,i]      = sum_j grad[b,k,j] W[j,i]                       (dense [B,K,din])
 *   out_grad_weight[j,i]   = sum_{b,k} grad[b,k,j] x[b,k,i]                 (dense [dout,din])
 * either output may be NULL.  The weight gradient is a contraction over the particle index and runs
 * on the matrix cores (f32 / f64 MFMA: exact fused multiply-adds), each workgroup leaving one 16 x 16
 * partial in `ws` (aesmc_affine_backward_workspace_bytes(dtype, B, K) bytes, 16-byte aligned), summed in
 * workgroup order by a second launch: reproducible run to run.
 *   out_grad_offset[b,j]   = sum_k grad[b,k,j]                              (dense [B,dout]; NULL: not wanted)
 * Replaces the input- and weight-gradient matmuls ([B*K,dout] x [dout,din] and [dout,B*K] x [B*K,din]) and
 * the broadcast-add's reduction of the callables' autograd.  AESMC_ERR_UNSUPPORTED when the offset
 * gradient is asked for with fewer than ~43 particles per batch row (the caller sums grad itself). */
size_t aesmc_affine_backward_workspace_bytes(int dtype, int64_t B, int64_t K);
int aesmc_particle_affine_backward(int dtype, const void *grad, const void *x, const aesmc_affine_map *map,
                                   void *out_grad_x, void *out_grad_weight, void *out_grad_offset, void *ws,
                                   size_t ws_bytes, int64_t B, int64_t K, void *stream);

/* K12 — backward of K10 in one pass over x_prev and x.  The incoming gradient of lw[b,k] is
 *   g = grad_lw[b,k]  (NULL = 0)  +  grad_lse[b] * exp(lw[b,k] - lse[b])  (NULL = 0; K1's backward formed
 *   in place: `lw` is what K10 produced, `lse` its row log-sum-exp — aesmc/inference.py:130-132)
 * and every output pointer of `out` may be NULL (not wanted):
 *   grad_x_prev, grad_x [B,K,dx]      gradients of the two latents (dense, 16-byte aligned);
 *   grad_loc_p, grad_loc_q [B,K,dx], grad_loc_g [B,K,dy]
 *                                     the gradients with respect to the three locations, written only
 *                                     when an OFFSET's gradient is wanted (the caller sums over
 *                                     particles; the observation's gradient is -sum_k grad_loc_g);
 *   grad_weight_p, grad_weight_q [dx,dx], grad_weight_g [dy,dx]
 *                                     weight gradients (matrix cores, partials in `ws` summed in
 *                                     workgroup order by a second launch: reproducible);
 *   grad_scales [3]                   d / d(scale_p, scale_g, scale_q);
 *   grad_offset_p, grad_offset_q [B,dx], grad_offset_g [B,dy]
 *                                     the location gradients summed over each batch row's particles —
 *                                     the gradient of a [B, d] offset (a shared [d] offset: sum the
 *                                     rows; the observation: minus grad_offset_g) — per tile in a fixed
 *                                     order, the tiles of a row added up by one more small launch.
 * `ws`: aesmc_affine_backward_workspace_bytes(dtype, B, K) bytes.  Replaces, for one timestep, the autograd
 * chain of aesmc/inference.py:112-132 through the callables' matmuls: K5's backward, three
 * weight-gradient and three input-gradient matmuls and the adds between them. */
typedef struct aesmc_affine_logweight_grads {
  void *grad_x_prev, *grad_x;
  void *grad_loc_p, *grad_loc_g, *grad_loc_q;
  void *grad_weight_p, *grad_weight_g, *grad_weight_q;
  void *grad_scales;
  void *grad_offset_p, *grad_offset_g, *grad_offset_q;
} aesmc_affine_logweight_grads;

int aesmc_affine_normal_logweight_backward(
    int dtype, const void *x_prev, const void *x, const void *y, int64_t y_stride_b,
    const aesmc_affine_map *transition, const aesmc_affine_map *emission, const aesmc_affine_map *proposal,
    const void *scale_p, const void *scale_g, const void *scale_q, const void *lw, const void *lse,
    const void *grad_lse, const void *grad_lw, const aesmc_affine_logweight_grads *out, void *ws, size_t ws_bytes,
    int64_t B, int64_t K, void *stream);

/* K14 — the whole backward of one SMC step whose x_t IS the proposal's reparameterised draw
 *   x_t = loc_q(x_{t-1}) + scale_q * eps        (what K9 produced from the same `proposal` map and `x_prev`),
 * i.e. of aesmc/inference.py:106-132 for a linear-Gaussian model: K12 plus the backward of the draw (K11)
 * plus the accumulations autograd puts between them, in one pass.  `grad_x` [B,K,dx] (dense, 16-byte
 * aligned, or NULL) is the gradient arriving at x_t from later timesteps (the next step's resampling
 * gather); g is formed as in K12.  With
 *   w = grad_x + d/dx_t of g * (log p(x_t | x_{t-1}) + log g(y_t | x_t))
 * the draw carries w to the proposal's parameters and to x_{t-1}; the proposal's density itself depends on
 * them only through eps, which the draw holds fixed, so of its terms only -dx log scale_q survives:
 *   grad_x_prev   = A^T u_p + Q^T w                 (u_p: K12's transition location-gradient)
 *   grad_weight_q = sum_{b,k} w (x) x_{t-1},  grad_offset_q[b] = sum_k w,
 *   grad_scales[2] = sum_{b,k} ( g dx / scale_q + w . eps ),   eps = (x_t - loc_q) / scale_q
 * and the transition's and emission's gradients are K12's.  `out->grad_x` and the three `grad_loc_*`
 * must be NULL (x_t gets no gradient of its own: it is not an independent variable here).  Same
 * workspace, same reproducible finishing launch and same AESMC_ERR_UNSUPPORTED shapes as K12. */
int aesmc_affine_step_backward(
    int dtype, const void *x_prev, const void *x, const void *y, int64_t y_stride_b,
    const aesmc_affine_map *transition, const aesmc_affine_map *emission, const aesmc_affine_map *proposal,
    const void *scale_p, const void *scale_g, const void *scale_q, const void *lw, const void *lse,
    const void *grad_lse, const void *grad_lw, const void *grad_x, const aesmc_affine_logweight_grads *out,
    void *ws, size_t ws_bytes, int64_t B, int64_t K, void *stream);

/* K14 through the ancestor indices: `x_src` is the UN-resampled x_{t-1} and the step's x_prev its rows
 * x_src[b, ancestors[b,k], :] fetched while the tile is staged (the forward step — aesmc_affine_normal_propagate_
 * resampled / _drawn — never wrote them, neither does the backward).  Everything else as
 * aesmc_affine_step_backward; `out->grad_x_prev` is the gradient with respect to the RESAMPLED rows
 * ([B,K,dx], one per child): the caller sums children into their ancestors (aesmc_resample_gather_backward),
 * which is torch.gather's backward (aesmc/state.py:179).  An ancestor outside [0, K) is clamped and raises
 * AESMC_FLAG_INDEX_OUT_OF_RANGE in `flags` (may be NULL).
 * `child_grad` / `child_end` (both or neither; NULL = none): the gradient that reaches x_t from the NEXT step when
 * that step, too, resampled through ancestors — its `out->grad_x_prev`, one row per CHILD [B,K,dx] — and the next
 * resampling step's children ranges (aesmc_resample_step_ranges).  The kernel then adds, to whatever `grad_x`
 * brings, the sum of each particle's children's rows (a lane adds its own particle's run in k order; what lies beyond
 * the first 32 children of a run — a collapsed particle system — is shared out over the wavefront, in a fixed order):
 * torch.gather's backward without its launch and without the [B,K,dx] round trip of the summed gradient.
 * `chain` (may be NULL): consecutive steps of a time-homogeneous model share A, C, Q and the scales, and autograd only
 * ever wants the SUM of their gradients over the steps (aesmc/inference.py:60-135 calls the same callables every
 * step).  A call with `chain->defer` non-zero leaves its sums as per-workgroup records in `ws` (grad_weight_* and
 * grad_scales of `out` must be NULL; `chain->records` receives their count; defer = 2 where the scales' gradients are
 * wanted in the end, which costs the proposal's location as a non-NULL grad_scales does) instead of finishing them; the next call
 * names that workspace in `chain->carry` / `carry_records` and each of its workgroups starts from the records it is
 * handed (record w, w + grid, ... in that order), so the run of steps is finished once, by its last call — or by
 * aesmc_affine_backward_collect.  The row sums (offset gradients: per step) are finished by every call. */
typedef struct {
  const void *carry;      /* in: workspace of the call whose records this one continues, or NULL */
  int32_t carry_records;  /* in: how many records it holds (what that call left in `records`) */
  int32_t defer;          /* in: non-zero = leave this call's sums as records in `ws` (2: the scales' among them) */
  int32_t records;        /* out: records this call left in `ws` */
  const void *pairs_in;   /* in: the three maps' interleaved weight pairs as an EARLIER call of this run left them in
                           *     `pairs_out` (the same weights: a run shares them), or NULL: this call writes its own into
                           *     its workspace's tail (one small launch in front) */
  void *pairs_out;        /* out: where the pairs this call used lie (its own, or `pairs_in` handed on), NULL if it used
                           *      none; valid while the workspace that holds them is */
} aesmc_affine_chain;
int aesmc_affine_step_backward_resampled(
    int dtype, const void *x_src, const int64_t *ancestors, const void *x, const void *y, int64_t y_stride_b,
    const aesmc_affine_map *transition, const aesmc_affine_map *emission, const aesmc_affine_map *proposal,
    const void *scale_p, const void *scale_g, const void *scale_q, const void *lw, const void *lse,
    const void *grad_lse, const void *grad_lw, const void *grad_x, const void *child_grad, const int32_t *child_end,
    const aesmc_affine_logweight_grads *out, void *ws, size_t ws_bytes, int32_t *flags, aesmc_affine_chain *chain,
    int64_t B, int64_t K, void *stream);

/* The finishing launch of K14 alone: grad_weight_p / _g / _q and grad_scales of `out` (each may be NULL) from the
 * `records` records a deferring aesmc_affine_step_backward_resampled call left in `ws` — for a run of steps whose last
 * call could not carry them on. */
int aesmc_affine_backward_collect(int dtype, const void *ws, int32_t records, int64_t dx, int64_t dy,
                                  const aesmc_affine_logweight_grads *out, void *stream);

/* Noise — the float32 tensor `torch.empty(numel).normal_()` holds on this device for a generator at
 * (seed, offset): out[e], e < numel.  Replaces, inside a kernel or on its own, the `_standard_normal` draw of
 * `Normal.rsample` (aesmc/state.py:98; torch/distributions/normal.py rsample) WITHOUT leaving PyTorch's
 * stream: ATen's launch geometry (`threads` = 256 * min(#CU * maxThreadsPerCU / 256, ceil(numel / 256)), the
 * caller computes it from the device properties), rocRAND's Philox4x32-10 and Box-Muller restated
 * (csrc/philox_normal.hpp).  The caller advances the generator by 4 * ceil(numel / (4 * threads)), what
 * `normal_` would have consumed.  offset must be a multiple of 4 (PyTorch's always is).  `variant` 0 is
 * the product (Box-Muller's affine maps as fused multiply-adds, as PyTorch's build of rocRAND has them);
 * 1 = separate multiply and add, kept for the test that shows which of the two PyTorch's build uses.
 * `rng_state`: as for aesmc_affine_normal_propagate_drawn (NULL outside a hipGraph). */
int aesmc_philox_normal_fill(void *out, int64_t numel, uint64_t seed, uint64_t offset, int64_t threads, int variant,
                             const uint64_t *rng_state, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AESMC_HIP_H_ */
