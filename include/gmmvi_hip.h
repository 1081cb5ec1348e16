/*
 * gmmvi_hip.h -- C ABI of libgmmvi_hip.so: the MI355X (gfx950) implementation of the per-iteration hot path
 * of OlegArenz/gmmvi (GMMVI.train_iter(): sample selection + background density -> model log-density/gradient
 * -> Stein/MORE natural-gradient estimate -> per-component KL-constrained update -> weight update).
 *
 * The reference has no FFI: its boundary for this path is the Python plug-in surface of
 * src/gmmvi/optimization/gmmvi.py:163-174 and the modules it calls.  Each entry point below names the
 * reference call site (path:line relative to /root/reference/src/gmmvi) whose arithmetic it replaces; the
 * Python classes in gmmvi_amd/ keep the reference's names and argument order and bind these symbols through
 * ctypes (INTEGRATION.md shows the binding a maintainer of the reference would add).
 *
 * Conventions
 *   - plain C: pointers and sizes only, no C++/torch types, no exceptions across the boundary;
 *   - every function returns 0 on success, a negative gmmvi_status otherwise; gmmvi_last_error() gives text;
 *   - numerical rejection (non-positive Cholesky pivot, NaN) is reported through success[] outputs, never as
 *     an error code (reference: NaN-Cholesky -> rejected update, ng_based_component_updater.py:320-324,:493);
 *   - all array arguments are DEVICE pointers obtained from gmmvi_malloc (names end in _dev) unless the
 *     parameter is documented as host; arrays are dense row-major fp32 (int32 for indices/flags);
 *   - calls are asynchronous on the context's HIP stream; gmmvi_download / gmmvi_sync synchronise;
 *   - one context per device per process; a context is not thread-safe (the reference caller is single
 *     threaded).
 */
#ifndef GMMVI_HIP_H
#define GMMVI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gmmvi_ctx gmmvi_ctx;

enum gmmvi_status {
    GMMVI_OK = 0,
    GMMVI_ERR_HIP = -1,        /* a HIP runtime call failed */
    GMMVI_ERR_ARG = -2,        /* invalid argument / unsupported shape */
    GMMVI_ERR_RCCL = -3,       /* an RCCL call failed */
    GMMVI_ERR_STATE = -4       /* call not valid in the context's current state */
};

enum gmmvi_family {
    GMMVI_GAUSS = 0,           /* log N(x; m, L L^T)                           models/full_cov_gmm.py:56-62 */
    GMMVI_STUDENT_T = 1        /* multivariate Student-t, scale operator L     target_distributions/student_t_mixture.py:40-44 */
};

enum gmmvi_stein_flags {
    GMMVI_SELF_NORMALIZED = 1, /* ng_estimator.py:171-188 (else :154-169, not symmetrised) */
    GMMVI_OWN_SAMPLES_ONLY = 2, /* ng_estimator.py:110-118 */
    /* gmmvi_train_iter_samtron only: finish the estimate as gmmvi_stein does (H, g materialised) instead of forming the
     * whitened matrix of the component update directly from the moment sums (same mathematics, different rounding) */
    GMMVI_EXPLICIT_ESTIMATE = 4
};

#define GMMVI_MORE_REGISTER_MAX_DIM 21  /* gmmvi_more: up to here the F x F ridge system (F = D(D+1)/2 + D + 1) is factorised in
                                         * one workgroup's registers; above (D <= 63; components of a blocked-path dimension are re-packed for the call) a tiled Gram launch and a blocked fp64
                                         * Cholesky in global memory take over */
#define GMMVI_MAX_DIM 64       /* register-resident kernels exist for D <= 64; they are used for D <= 50 (environment
                                * GMMVI_BLOCKED_ABOVE, 16..64, moves that threshold) */
#define GMMVI_MAX_DIM_BLOCKED 512   /* above the threshold, D <= 512: blocked kernels (dense L^-1 blocks, fp32 MFMA contractions)
                                     * behind gmmvi_packed_stride / pack_components / cholesky / mixture_eval(_dual) /
                                     * sample_components / stein / update_components_kl / _direct / _iblr */

/* ---- context, errors, memory ------------------------------------------------------------------------ */
int gmmvi_device_count(void);
int gmmvi_ctx_create(gmmvi_ctx** out, int device);
void gmmvi_ctx_destroy(gmmvi_ctx* ctx);
const char* gmmvi_last_error(gmmvi_ctx* ctx);          /* ctx may be NULL: last error of a failed create */
int gmmvi_sync(gmmvi_ctx* ctx);
int gmmvi_malloc(gmmvi_ctx* ctx, size_t nbytes, void** out_dev);
int gmmvi_free(gmmvi_ctx* ctx, void* dev);
int gmmvi_upload(gmmvi_ctx* ctx, void* dst_dev, const void* src_host, size_t nbytes);
int gmmvi_download(gmmvi_ctx* ctx, void* dst_host, const void* src_dev, size_t nbytes);   /* synchronises */
int gmmvi_copy(gmmvi_ctx* ctx, void* dst_dev, const void* src_dev, size_t nbytes);
int gmmvi_fill_f32(gmmvi_ctx* ctx, float* dst_dev, float value, size_t count);
/* dst[i, :] = src[idx[i], :] for rows of row_words 4-byte words (fp32 or int32): SampleDB.remove_every_nth_sample /
 * get_random_sample / background-component gather (optimization/sample_db.py:63-79,137-152,222-224 tf.gather). */
int gmmvi_gather_rows(gmmvi_ctx* ctx, const void* src_dev, const int32_t* idx_dev, int n_rows, int row_words,
                      void* dst_dev);
/* Up to 8 device-to-device copies in ONE launch (the per-iteration SampleDB append of samples, target values,
 * gradients and component snapshots, optimization/sample_db.py:115-124; sizes in bytes, multiples of 4). */
int gmmvi_copy_batch(gmmvi_ctx* ctx, int n, void* const* dst_dev, const void* const* src_dev, const size_t* nbytes);
/* De-interleaves an all-gathered buffer: every rank contributed one chunk of chunk_words 4-byte words made of n_seg <= 8
 * consecutive segments (seg_words[j] words each); dst[j] receives segment j of all ranks back to back,
 * dst[j][r * seg_words[j] + i] = src[r * chunk_words + seg_offset_j + i].  One launch; lets the sharded iteration send
 * several arrays in ONE collective (sharded.py E1-E3). */
/* GMM.replace_weights (models/gmm.py:173-181) on the device: out = lw - logsumexp(lw), the sum in fp64. */
int gmmvi_normalize_logw(gmmvi_ctx* ctx, const float* logw_in_dev, int n, float* logw_out_dev);
/* VipsComponentAdaptation.add_at_best_location (component_adaptation.py:192-226): index_out[0] = argmax_n of
 * log p~(x_n) - max(max_m log q(x_m) - threshold, log q(x_n)) over n candidates (first index on ties, fp64). */
int gmmvi_add_heuristic_argmax(gmmvi_ctx* ctx, const float* model_ld_dev, const float* target_lnpdfs_dev, int n, double threshold,
                               int32_t* index_out_dev);
int gmmvi_unpack_gathered(gmmvi_ctx* ctx, const void* src_dev, int n_ranks, size_t chunk_words, int n_seg,
                          const size_t* seg_words, void* const* dst_dev);
/* dst[i * stride] = value, i < count: a new component's column of the [H, Kcap] reward / weight history rings
 * (models/gmm_wrapper.py:121-124). */
int gmmvi_fill_strided_f32(gmmvi_ctx* ctx, float* dst_dev, size_t stride, size_t count, float value);
/* a[r, idx:K-1] = a[r, idx+1:K] for every row r of a [rows, stride] array: GmmWrapper.remove_component's history
 * update (models/gmm_wrapper.py:145-146). */
int gmmvi_remove_column_f32(gmmvi_ctx* ctx, float* a_dev, int rows, size_t stride, int K, int idx);
/* dst[i] = src[i] + value: the mapping offset of SampleDB.add_samples (optimization/sample_db.py:115). */
int gmmvi_add_scalar_i32(gmmvi_ctx* ctx, int32_t* dst_dev, const int32_t* src_dev, int32_t value, size_t count);
/* dst[i] = exp(src[i])  (GMM.weights, models/gmm.py:171; weight history, models/gmm_wrapper.py:182). */
int gmmvi_exp_f32(gmmvi_ctx* ctx, float* dst_dev, const float* src_dev, size_t count);
/* dst[i] = log(exp(a[i] + ca) + exp(b[i] + cb)): joins the two halves of SampleDB's background density when the window grows by
 * one append -- the mixture of the components that were already in the window (known for the old samples from the previous
 * get_newest_samples call) and the mixture of the new components (optimization/sample_db.py:216-227 evaluates the whole
 * window's mixture on all of its samples every time). */
int gmmvi_logaddexp_f32(gmmvi_ctx* ctx, float* dst_dev, const float* a_dev, float ca, const float* b_dev, float cb,
                        size_t count);
/* Per-append partial densities of SampleDB's background mixture: the components of a window are grouped by the append that
 * contributed them (G groups, offsets [G+1] into the component axis); out[g * out_stride + col0 + n] = log sum_{j in group g}
 * exp(logw[j] + ld[j, n]) from the component log densities ld [Kw, N].  The window's density (optimization/sample_db.py:
 * 216-227) is the log-sum-exp of those rows (gmmvi_combine_partials); rows of appends that stay wholly inside the sliding
 * window are kept from one iteration to the next (gmmvi_copy_2d_f32 moves the surviving block). */
int gmmvi_segment_lse_f32(gmmvi_ctx* ctx, int G, const int32_t* offsets_dev, const float* logw_dev, const float* ld_dev, int N,
                          float* out_dev, size_t out_stride, int col0);
/* dst[r * dst_stride + c] = src[r * src_stride + c], r < rows, c < cols. */
int gmmvi_copy_2d_f32(gmmvi_ctx* ctx, float* dst_dev, size_t dst_stride, const float* src_dev, size_t src_stride, int rows,
                      int cols);
/* timing helpers for bench.py: HIP events on the context's stream */
int gmmvi_event_create(gmmvi_ctx* ctx, void** out_event);
int gmmvi_event_destroy(gmmvi_ctx* ctx, void* event);
int gmmvi_event_record(gmmvi_ctx* ctx, void* event);
int gmmvi_event_elapsed_ms(gmmvi_ctx* ctx, void* start, void* stop, float* out_ms);       /* synchronises on stop */
int gmmvi_event_synchronize(gmmvi_ctx* ctx, void* event);                                   /* waits until the stream has reached it */
/* A read-back that does not wait: dst is pinned host memory from gmmvi_host_alloc; valid once an event recorded behind the copy
 * has been reached (the effective sample sizes of the NEXT iteration's reuse window are fetched this way while the current
 * iteration's last launches still run: optimization/fused.py). */
int gmmvi_host_alloc(gmmvi_ctx* ctx, size_t nbytes, void** out_pinned_host);
int gmmvi_host_free(gmmvi_ctx* ctx, void* pinned_host);
int gmmvi_download_async(gmmvi_ctx* ctx, void* dst_pinned_host, const void* src_dev, size_t nbytes);

/* Grow-in-place device buffers (what optimization/sample_db.py keeps its arrays in; the reference's SampleDB concatenates
   tensors, sample_db.py:97-135): reserve an address range once, map physical memory behind its used part in chunks of
   *chunk_bytes_out; growing copies nothing.  gmmvi_vmm_grow: mapped_bytes / new_mapped_bytes are multiples of the chunk size.
   gmmvi_vmm_release waits for the context's stream, unmaps and frees the range. */
int gmmvi_vmm_reserve(gmmvi_ctx* ctx, size_t max_bytes, void** base_out, size_t* chunk_bytes_out);
int gmmvi_vmm_grow(gmmvi_ctx* ctx, void* base, size_t chunk_bytes, size_t mapped_bytes, size_t new_mapped_bytes);
int gmmvi_vmm_release(gmmvi_ctx* ctx, void* base, size_t chunk_bytes, size_t mapped_bytes, size_t reserved_bytes);

/* Per-kernel timing for bench.py's roofline leg: while enabled, every kernel-launching entry point brackets its
 * launches with HIP events on the context's stream.  gmmvi_profile_report synchronises, writes one line per kernel
 * name ("name count total_ms total_pairs\n"; total_pairs = (sample, component) pairs processed, 0 where not counted) into
 * buf and clears the records. */
int gmmvi_profile_enable(gmmvi_ctx* ctx, int on);
int gmmvi_profile_report(gmmvi_ctx* ctx, char* buf, size_t buf_size);

/* ---- component parameter blocks ---------------------------------------------------------------------- */
/* Number of floats of one packed component block for dimension D (D <= 64: padded dimension, reciprocal diagonal,
 * row- and column-packed strict lower triangle of L, log-normaliser, then for padded D <= 24 the "sweep stream" the packed
 * two-samples-per-lane density kernel reads in order -- mean, log-normaliser, the triangle by columns and by rows with the
 * reciprocal diagonal entries in place -- and for padded D >= 32 L^-1 as matrix-core operand fragments;
 * 64 < D <= 512: mu, log-normaliser, dense L^-1).  0 for an unsupported dimension. */
size_t gmmvi_packed_stride(int D);
/* Pack K components (means[K,D], chols[K,D,D] lower-triangular dense) for the density kernels.
 * family/nu select the log-normaliser (Gaussian: full_cov_gmm.py:60-61; Student-t: student_t_mixture.py:40-44).
 * Also writes inv_chols[K,D,D] = L^-1 when inv_chols_dev != NULL (sample_db.py:121 tf.linalg.inv(chols)). */
int gmmvi_pack_components(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* means_dev,
                          const float* chols_dev, float* packed_dev, float* inv_chols_dev);
/* Batched Cholesky covs[K,D,D] -> chols[K,D,D] (full_cov_gmm.py:23, :64-68); ok[K] = 0 on a non-positive pivot. */
int gmmvi_cholesky(gmmvi_ctx* ctx, int K, int D, const float* covs_dev, float* chols_dev, int32_t* ok_dev);

/* ---- densities ----------------------------------------------------------------------------------------- */
/* For every sample n and component k:  ld[k,n] = log f_k(x_n)  and, over components,
 *   lp[n]   = logsumexp_k(logw[k] + ld[k,n])
 *   grad[n] = sum_k softmax_k(logw[k] + ld[k,n]) * d/dx log f_k(x_n)
 * Replaces FullCovGMM.component_log_densities (models/full_cov_gmm.py:56-62), GMM.log_density /
 * log_densities_also_individual / log_density_and_grad (models/gmm.py:183-216,274-300), SampleDB.evaluate_background
 * (optimization/sample_db.py:154-192, with logw = log(count/N)) and the GMM / Student-t targets with their
 * autodiff gradient (target_distributions/gmm.py:38-40, student_t_mixture.py:66-68, sample_selector.py:74-77).
 * Any of ld_out_dev[K,N], lp_out_dev[N], grad_out_dev[N,D] may be NULL. */
int gmmvi_mixture_eval(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* packed_dev,
                       const float* logw_dev, const float* X_dev, int N, float* ld_out_dev, float* lp_out_dev,
                       float* grad_out_dev);

/* Same pass with a SECOND set of mixture weights over the same components: lp2[n] = logsumexp_k(logw2[k] + ld[k,n]).
 * Fuses SampleDB.get_newest_samples' background density (weights = sample counts, sample_db.py:221-227) with
 * GMM.log_density_and_grad (models/gmm.py:274-300) when the active samples were drawn from the current components
 * (reuse ratio 0): one sweep over the (sample, component) pairs instead of two. */
int gmmvi_mixture_eval_dual(gmmvi_ctx* ctx, int family, float nu, int K, int D, const float* packed_dev,
                            const float* logw_dev, const float* logw2_dev, const float* X_dev, int N, float* ld_out_dev,
                            float* lp_out_dev, float* grad_out_dev, float* lp2_out_dev);

/* Planar n-link robot target (target_distributions/planar_robot.py:29-66) and its gradient. goals_dev[G,2]. */
int gmmvi_target_planar(gmmvi_ctx* ctx, int D, const float* prior_std_dev, int G, const float* goals_dev,
                        float likelihood_std, const float* X_dev, int N, float* lp_out_dev, float* grad_out_dev);

/* ---- sampling -------------------------------------------------------------------------------------------- */
/* x = mu_k + L_k eps for offsets[k] <= n < offsets[k+1] (component order), mapping[n] = k.
 * Replaces GMM.sample_from_components_no_shuffle + FullCovGMM.sample_from_component
 * (models/gmm.py:361-386, models/full_cov_gmm.py:36-39).  eps_dev[N,D] != NULL: host-provided normals
 * (bit-reproducible parity runs); else Philox4x32-10 keyed by seed, counter = first_index + n, given stream. */
int gmmvi_sample_components(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* chols_dev,
                            const int32_t* offsets_dev, int N, uint64_t seed, uint64_t first_index, int stream_id,
                            const float* eps_dev, float* X_out_dev, int32_t* mapping_out_dev);
/* Standard normals eps[N,D] of the same Philox stream (tests, GMM.sample). */
int gmmvi_philox_normals(gmmvi_ctx* ctx, uint64_t seed, uint64_t first_index, int stream_id, int N, int D,
                         float* eps_out_dev);
/* Uniforms u[N] in (0,1): word 0 of block 0 (GMM.sample_categorical, models/gmm.py:134-137). */
int gmmvi_philox_uniforms(gmmvi_ctx* ctx, uint64_t seed, uint64_t first_index, int stream_id, int N,
                          float* u_out_dev);

/* ---- natural-gradient estimate (Stein) --------------------------------------------------------------------- */
/* SteinNgEstimator.get_expected_hessian_and_grad (gmmvi_modules/ng_estimator.py:204-263,171-188,154-169).
 * Inputs: packed model components, samples X[N,D], component log-densities ld[K,N] and model gradient
 * qgrad[N,D] (from gmmvi_mixture_eval on the same X), background densities bg[N], target gradients tgrad[N,D],
 * mapping[N] (only read with GMMVI_OWN_SAMPLES_ONLY; map_offset = K-1-max(mapping), ng_estimator.py:244).
 * Outputs: H_neg[K,D,D] = -E_k[...], g_neg[K,D]. */
int gmmvi_stein(gmmvi_ctx* ctx, int K, int D, const float* packed_dev, const float* X_dev, int N,
                const float* ld_dev, const float* qgrad_dev, const float* bg_dev, const float* tgrad_dev,
                const int32_t* mapping_dev, int map_offset, int flags, float* H_neg_out_dev, float* g_neg_out_dev);

/* MoreNgEstimator.get_expected_hessian_and_grad (gmmvi_modules/ng_estimator.py:296-376) with QuadFunc.fit_quadratic
 * (optimization/least_squares.py:126-191, RegressionFunc.fit :34-76): importance-weighted quadratic ridge regression of
 * the rewards tlp[n] - logq[n] on the samples whitened by each component; flags = enum gmmvi_stein_flags.
 * Inputs: packed blocks + chols[K,D,D] of the model, X[N,D], ld[K,N] and logq[N] (gmmvi_mixture_eval on the same X),
 * bg[N], tlp[N], mapping[N] (GMMVI_OWN_SAMPLES_ONLY only), l2[K] ridge coefficients (GmmWrapper.l2_regularizers).
 * Outputs: H_neg[K,D,D], g_neg[K,D]; NaN for a component whose ridge system is not positive definite.
 * D <= 63.  For a dimension on the blocked path (50 < D by default) the blocks handed over have another layout: the
 * components are re-packed in the register-path layout for the call (means from the blocks, factors from chols_dev);
 * D >= 64: GMMVI_ERR_ARG. */
int gmmvi_more(gmmvi_ctx* ctx, int K, int D, const float* packed_dev, const float* chols_dev, const float* X_dev, int N,
               const float* ld_dev, const float* logq_dev, const float* bg_dev, const float* tlp_dev,
               const int32_t* mapping_dev, int map_offset, int flags, const float* l2_dev, float* H_neg_out_dev,
               float* g_neg_out_dev);

/* ---- component updates --------------------------------------------------------------------------------------- */
/* KLConstrainedNgBasedComponentUpdater.apply_NG_update (gmmvi_modules/ng_based_component_updater.py:431-524,
 * bracketing_search :335-429, kl :244-333) with the exact stop rules of SURVEY.md Appendix A.1.
 * In/out: means[K,D], chols[K,D,D], last_eta[K] (stores eta, not log eta; -1 = none), l2[K].
 * Out (may be NULL): success[K] (int32 0/1), kl[K], n_probes[K], packed[K, gmmvi_packed_stride(D)] = the Gaussian
 * parameter blocks of the updated model (saves the separate gmmvi_pack_components launch). */
int gmmvi_update_components_kl(gmmvi_ctx* ctx, int K, int D, float* means_dev, float* chols_dev,
                               const float* H_neg_dev, const float* g_neg_dev, const float* stepsizes_dev,
                               float temperature, float l2_init, float* last_eta_dev, float* l2_dev,
                               float* num_received_updates_dev, int32_t* success_out_dev, float* kl_out_dev,
                               int32_t* n_probes_out_dev, float* packed_out_dev);
/* Same contract, computed with the reference's own formulation (Q' = (eta Q + R)/eta, Cholesky + triangular inverse
 * per probe, sequential bisection).  Slow; kept as an on-device cross-check of the production kernel. */
int gmmvi_update_components_kl_reference(gmmvi_ctx* ctx, int K, int D, float* means_dev, float* chols_dev,
                                         const float* H_neg_dev, const float* g_neg_dev, const float* stepsizes_dev,
                                         float temperature, float l2_init, float* last_eta_dev, float* l2_dev,
                                         float* num_received_updates_dev, int32_t* success_out_dev, float* kl_out_dev,
                                         int32_t* n_probes_out_dev);
/* DirectNgBasedComponentUpdater (:97-141) and NgBasedComponentUpdaterIblr (:160-223). */
int gmmvi_update_components_direct(gmmvi_ctx* ctx, int K, int D, float* means_dev, float* chols_dev,
                                   const float* H_neg_dev, const float* g_neg_dev, const float* stepsizes_dev,
                                   float l2_init, float* l2_dev, float* num_received_updates_dev,
                                   int32_t* success_out_dev);
int gmmvi_update_components_iblr(gmmvi_ctx* ctx, int K, int D, float* means_dev, float* chols_dev,
                                 const float* H_neg_dev, const float* g_neg_dev, const float* stepsizes_dev,
                                 float l2_init, float* l2_dev, float* num_received_updates_dev,
                                 int32_t* success_out_dev);

/* ---- weights, rewards, stepsizes ---------------------------------------------------------------------------- */
/* WeightUpdater._get_expected_log_ratios (gmmvi_modules/weight_updater.py:56-75), self-normalised branch when
 * self_normalized != 0: E[k] = sum_n softmax_n(ld[k,n] - bg[n]) (tlp[n] - beta*logq[n]);
 * reward[k] = beta*logw[k] + E[k] (:73).  ess_out_dev (may be NULL) receives 1/sum_n w^2 per component
 * (sample_selector.py:154-158 uses the same weights). */
int gmmvi_expected_log_ratios(gmmvi_ctx* ctx, int K, int N, const float* ld_dev, const float* bg_dev,
                              const float* tlp_dev, const float* logq_dev, float beta, const float* logw_dev,
                              int self_normalized, float* E_out_dev, float* reward_out_dev, float* ess_out_dev);
/* TrustRegionBasedWeightUpdater (weight_updater.py:164-279) followed by GMM.replace_weights (models/gmm.py:173-181).
 * stepsize_dev[1] is the KL bound.  kl_eta_out_dev[2] (may be NULL) = (kl, eta). No-op when K == 1 (:275). */
int gmmvi_update_weights_kl(gmmvi_ctx* ctx, int K, float* logw_dev, const float* E_dev, const float* stepsize_dev,
                            float beta, float* kl_eta_out_dev);
/* DirectWeightUpdater (weight_updater.py:123-141). */
int gmmvi_update_weights_direct(gmmvi_ctx* ctx, int K, float* logw_dev, const float* E_dev,
                                const float* stepsize_dev, float beta);
/* ImprovementBasedComponentStepsizeAdaptation.update_stepsize (component_stepsize_adaptation.py:165-188):
 * rewards_prev/rewards_last are reward_history[:, -2] / [:, -1]. */
int gmmvi_component_stepsize_improvement(gmmvi_ctx* ctx, int K, float* stepsizes_dev, const float* rewards_prev_dev,
                                         const float* rewards_last_dev, float min_stepsize, float max_stepsize,
                                         float inc_factor, float dec_factor);
/* ImprovementBasedWeightStepsizeAdaptation._update_stepsize (weight_stepsize_adaptation.py:141-156).
 * state_dev[2] = (stepsize, previous elbo proxy); updated in place. */
int gmmvi_weight_stepsize_improvement(gmmvi_ctx* ctx, int K, const float* logw_dev, const float* rewards_last_dev,
                                      float* state_dev, float min_stepsize, float max_stepsize, float inc_factor,
                                      float dec_factor);

/* ---- single-call iteration ------------------------------------------------------------------------------------- */
/* GMMVI.train_iter() (optimization/gmmvi.py:146-174) for the SAMTRON design choices with a component-based sample
 * selector: Stein estimator, KL-constrained component update, trust-region or direct weight update, improvement-based or
 * fixed stepsizes, built-in target.  The call is exactly the composition of the entry points above in the order the plug-in
 * modules invoke them (one host call instead of ~20), operating on the caller's state arrays.
 * Sample reuse (sample_selector.py:204-219, ratio_reused_samples_to_desired > 0): the caller has chosen the per-component
 * counts of the NEW samples (`offsets`; they follow from the effective sample sizes of the reused ones, sample_selector.py:
 * 160-202) and passes the number of reused samples `n_old`: the active samples are the n_old database rows in front of the
 * append position followed by the N new ones (contiguous in the database), the background mixture is given by
 * (bg_K, bg_packed, bg_logw) over the database's component snapshots of that window (sample_db.py:216-227).  With
 * n_old == 0 and bg_packed == NULL the background components are the model's own (one sweep for both). */
typedef struct gmmvi_samtron_plan {
    int32_t K, D, N;                      /* components, dimension, samples of this iteration (sum of the counts) */
    int32_t target_kind;                  /* 0: mixture family (gmmvi_mixture_eval), 1: planar robot */
    int32_t target_family, target_K;      /* enum gmmvi_family, number of target components */
    float target_nu;
    const float* target_packed;           /* [target_K, stride] */
    const float* target_logw;             /* [target_K] */
    const float* planar_prior_std;        /* [D] */
    const float* planar_goals;            /* [G, 2] */
    int32_t planar_goals_count;
    float planar_likelihood_std;
    /* model state (in/out) */
    float* means; float* chols; float* logw;          /* [K,D], [K,D,D], [K] */
    const float* packed;                  /* parameter blocks of the CURRENT components [K, stride] */
    float* packed_new;                    /* out: blocks of the updated components */
    float* stepsizes; float* last_eta; float* l2; float* num_updates;      /* [K] each */
    int32_t* success_out;                 /* [K] or NULL */
    /* sampling */
    const int32_t* offsets;               /* [K+1] prefix sums of the per-component sample counts */
    int32_t max_per_component;            /* largest per-component count (0: N / K rounded up) */
    int32_t n_old;                        /* reused samples in front of the append position (0: none) */
    int32_t bg_K;                         /* n_old > 0: components the reused samples came from (snapshots; the new ones excluded) */
    const float* bg_packed;               /* [bg_K, stride] their snapshot blocks, or NULL (n_old == 0) */
    const float* bg_logw;                 /* n_old == 0: [K] log(count / N), the background weights (sample_db.py:225-226);
                                           * n_old > 0: [bg_K] log(count / n_old) of the components the reused samples came from */
    const float* bg_old;                  /* n_old > 0: [n_old] background density of the reused samples under those components with
                                           * those weights (what get_newest_samples(n_old) returned for the effective sample sizes) */
    const float* bg_logw_new;             /* n_old > 0: [K] log(count / N) of the new samples */
    float bg_log_share_old, bg_log_share_new;   /* n_old > 0: log(n_old / (n_old + N)), log(N / (n_old + N)) */
    uint64_t seed, first_index;           /* Philox key / global index of the first new sample */
    /* SampleDB append targets, already offset to the first free row (sample_db.py:115-124); snapshots may be NULL */
    float* db_samples; float* db_tlp; float* db_tgrad; int32_t* db_mapping; int32_t mapping_base;
    float* db_means; float* db_chols; float* db_packed;
    /* reward / weight history slots (models/gmm_wrapper.py:150-158,182) */
    const float* reward_prev; const float* reward_last; float* reward_next; float* weight_slot;
    float* wstate;                        /* [2]: weight stepsize, previous ELBO proxy */
    /* hyper-parameters */
    float temperature, l2_init;
    int32_t component_stepsize_mode;      /* 0 fixed, 1 improvement-based */
    float cs_min, cs_max, cs_inc, cs_dec;
    int32_t weight_stepsize_mode;         /* 0 fixed, 1 improvement-based */
    float ws_min, ws_max, ws_inc, ws_dec;
    int32_t weight_update_mode;           /* 0 trust-region, 1 direct */
    int32_t stein_flags;                  /* enum gmmvi_stein_flags (own-samples-only is not supported here;
                                           * GMMVI_EXPLICIT_ESTIMATE: results bit-equal to the module-by-module calls) */
    /* Drawing the NEXT iteration's samples early (n_old == 0 only).  x = mu + L eps needs the updated components, not the
     * weights (sample_selector.py:204-219), so the draw of iteration i + 1 can ride as extra workgroups in the post-update
     * sweep of iteration i instead of opening iteration i + 1 with a launch of its own.
     * presample_next != 0: after the component update, draw the N samples of the next iteration (same offsets, Philox
     *   indices first_index + N ..) into db_samples + N * D and their mapping (base mapping_base + K) into db_mapping + N;
     *   the caller has reserved those rows.
     * presampled != 0: db_samples / db_mapping already hold this iteration's draw (made by the previous call with
     *   presample_next, nothing touched the components, the offsets or the database since): no sampling launch. */
    int32_t presample_next, presampled;
    /* 0: the whole iteration.  1 / 2: the iteration in two calls -- 1 ends behind the component update, 2 is the weight
     * update (post-update sweep, expected log-ratios, weight step).  Between the two the caller may queue work that needs the
     * updated components but not the weights: with sample reuse the effective sample sizes of the NEXT iteration's window
     * (sample_selector.py:140-202), read back asynchronously, so that the next iteration starts without waiting for them. */
    int32_t phase;
} gmmvi_samtron_plan;
int gmmvi_train_iter_samtron(gmmvi_ctx* ctx, const gmmvi_samtron_plan* plan);

/* The same iteration for COMPONENT SHARDS (the reference has no multi-GPU code): rank r owns K components, their
 * samples and their updates; every component needs all N samples of the iteration, so the iteration has three exchange points.
 * It is issued as FOUR calls with one all-gather between consecutive calls (gmmvi_allgather_f32 in place on the exchange
 * buffers; the caller may substitute any exchange, e.g. through the host in tests) -- the launches inside a phase are those of
 * gmmvi_train_iter_samtron (packed density sweeps with carried merges, Stein slab whitened inside the update kernel):
 *   phase 1  draw the local samples and evaluate the target on them, straight into this rank's part of e1
 *   -- all-gather e1 --  [x | log p~ | grad log p~ | E | reward] of every rank
 *   phase 2  de-interleave e1; apply the PREVIOUS iteration's weight step from the gathered (E, reward) (replicated on every
 *            rank: nothing reads the new weights or the new reward column earlier); stepsize rules; dual density sweep over
 *            the local components on all N samples into this rank's part of e2
 *   -- all-gather e2 --  [background partial | log q partial | gradient partial] of every rank
 *   phase 3  combine the ranks' partials; Stein estimate and KL-constrained update of the local components; post-update
 *            sweep, its log q partial into this rank's part of e3
 *   -- all-gather e3 --
 *   phase 4  expected log-ratios and rewards of the local components into this rank's part of e1 (they travel with the next
 *            iteration's first exchange)
 * With n_ranks == 1 the gathers are no-ops and the gathered views may alias the parts. */
typedef struct gmmvi_sharded_plan {
    int32_t n_ranks, rank;
    int32_t K, D, N;                      /* LOCAL components, dimension, samples of ALL ranks (N % n_ranks == 0) */
    int32_t target_kind, target_family, target_K;
    float target_nu;
    const float* target_packed; const float* target_logw;
    const float* planar_prior_std; const float* planar_goals;
    int32_t planar_goals_count; float planar_likelihood_std;
    /* local model state (in/out) */
    float* means; float* chols; const float* packed; float* packed_new;
    float* stepsizes; float* last_eta; float* l2; float* num_updates; int32_t* success_out;
    float* logw_all;                      /* [K * n_ranks] replicated log weights */
    const float* bg_logw;                 /* [K] log(count_k / N) of the local components */
    const int32_t* offsets;               /* [K+1] prefix sums of the local per-component sample counts */
    int32_t max_per_component;
    uint64_t seed, first_index;           /* Philox key / global index of this rank's first new sample */
    /* exchange buffers [n_ranks * stride] floats; this rank's part at rank * stride */
    float* e1;                            /* stride (N / n_ranks) * (2 D + 1) + 2 K */
    float* e2;                            /* stride N * (D + 2) */
    float* e3;                            /* stride N */
    /* gathered arrays written by phase 2 */
    float* x_all; float* tlp_all; float* tgrad_all;      /* [N,D], [N], [N,D] */
    float* E_all; float* reward_all;      /* [K * n_ranks] */
    int32_t has_pending;                  /* e1 carries the previous iteration's (E, reward): phase 2 applies that weight step */
    float* reward_col_pending;            /* [K * n_ranks] reward-history column that receives the gathered rewards */
    const float* reward_prev; const float* reward_last;  /* [K] local slices of the two newest columns AFTER the pending step */
    const float* reward_last_all;         /* [K * n_ranks] the newest column (weight stepsize rule) */
    float* wstate;                        /* [2] weight stepsize, previous ELBO proxy */
    float temperature, l2_init;
    int32_t component_stepsize_mode; float cs_min, cs_max, cs_inc, cs_dec;
    int32_t weight_stepsize_mode; float ws_min, ws_max, ws_inc, ws_dec;
    int32_t stein_flags;
    /* as in gmmvi_samtron_plan: phase 4 draws the NEXT iteration's local samples (Philox indices first_index + N ..) into this
     * rank's part of e1 as riders of the expected-log-ratio launch; phase 1 of a call with presampled != 0 skips its draw */
    int32_t presample_next, presampled;
    /* scratch that lives across the four phases (component log densities, merged mixture arrays): caller-owned so that
     * several plans can take turns on one context; at least gmmvi_sharded_scratch_floats(K, D, N) floats */
    float* scratch;
} gmmvi_sharded_plan;
size_t gmmvi_sharded_scratch_floats(int K, int D, int N);
int gmmvi_train_iter_sharded_phase(gmmvi_ctx* ctx, const gmmvi_sharded_plan* plan, int phase /* 1..4 */);

/* ---- multi-GPU exchange (component shards) ----------------------------------------------------- */
/* RCCL communicator over the ranks of one node; unique_id is the 128-byte ncclUniqueId produced by rank 0. */
int gmmvi_comm_unique_id(char* out_id_128);
int gmmvi_comm_init(gmmvi_ctx* ctx, const char* unique_id_128, int n_ranks, int rank);
int gmmvi_comm_destroy(gmmvi_ctx* ctx);
int gmmvi_allgather_f32(gmmvi_ctx* ctx, const float* send_dev, float* recv_dev, size_t count_per_rank);
int gmmvi_allreduce_f32(gmmvi_ctx* ctx, float* buf_dev, size_t count, int op /* 0 sum, 1 max */);
/* Combine per-rank partial mixtures: given gathered (lp_r[n], grad_r[n,D]) of R ranks, writes
 * lp[n] = LSE_r lp_r[n], grad[n] = sum_r exp(lp_r[n]-lp[n]) grad_r[n]  (the E2/E3 exchange of SURVEY.md 8e). */
int gmmvi_combine_partials(gmmvi_ctx* ctx, int R, int N, int D, const float* lp_parts_dev,
                           const float* grad_parts_dev, float* lp_out_dev, float* grad_out_dev);

/* ---- diagonal-covariance GMMs (models/diagonal_gmm.py:6-59) ---------------------------------------------------- */
/* chol[K,D] = sigma (square roots of the covariance diagonal).  Densities / gradients / sampling / background /
 * Stein CAN use the dense entry points above on L = diag(sigma) (kept for callers that hold dense factors; the Python mirror
 * uses the dedicated kernels below since round 3):
 * gmmvi_diag_embed writes dense[K,D,D] from diag[K,D]; gmmvi_diag_extract reads diag[K,D] = diagonal of dense[K,D,D]
 * (the diagonal Stein estimate of gmmvi_modules/ng_estimator.py:159-162,:178-181 is the diagonal of gmmvi_stein's
 * H_neg); gmmvi_reciprocal_f32 is SampleDB's inv_chols = 1 / chols (optimization/sample_db.py:119,:130). */
int gmmvi_diag_embed(gmmvi_ctx* ctx, int K, int D, const float* diag_dev, float* dense_out_dev);
int gmmvi_diag_extract(gmmvi_ctx* ctx, int K, int D, const float* dense_dev, float* diag_out_dev);
int gmmvi_reciprocal_f32(gmmvi_ctx* ctx, const float* src_dev, size_t n, float* dst_dev);
/* Dedicated O(D)-per-pair kernels for diagonal mixtures (csrc/diag_sweep.hip).  Component block of this path
 * (gmmvi_diag_pack, gmmvi_diag_packed_stride(D) floats): [mu | 1/sigma | 1/sigma^2 | log-normaliser | padding].
 *   gmmvi_diag_mixture_eval  DiagonalGMM.component_log_densities / log_density / log_density_and_grad (models/diagonal_gmm.py:
 *                            40-53, models/gmm.py:183-216,274-300) and SampleDB's background density over diagonal snapshots
 *                            (sample_db.py:164-228); logw2 / lp2_out: a second mixture over the same components (may be NULL)
 *   gmmvi_diag_sample        GMM.sample_from_components_no_shuffle for a DiagonalGMM (x = mu + sigma * eps,
 *                            models/diagonal_gmm.py:43-45): same offsets / Philox arguments as gmmvi_sample_components
 *   gmmvi_diag_stein         SteinNgEstimator, diagonal branches (ng_estimator.py:159-162,:178-181): h_neg_diag[K,D], g_neg[K,D];
 *                            arguments as gmmvi_stein */
size_t gmmvi_diag_packed_stride(int D);
int gmmvi_diag_pack(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* sigma_dev, float* packed_out_dev);
int gmmvi_diag_mixture_eval(gmmvi_ctx* ctx, int K, int D, const float* packed_dev, const float* logw_dev, const float* logw2_dev,
                            const float* X_dev, int N, float* ld_out_dev, float* lp_out_dev, float* grad_out_dev,
                            float* lp2_out_dev);
int gmmvi_diag_sample(gmmvi_ctx* ctx, int K, int D, const float* means_dev, const float* sigma_dev, const int32_t* offsets_dev,
                      int N, uint64_t seed, uint64_t first_index, int stream_id, const float* eps_dev, float* X_out_dev,
                      int32_t* mapping_out_dev);
int gmmvi_diag_stein(gmmvi_ctx* ctx, int K, int D, const float* packed_dev, const float* X_dev, int N, const float* ld_dev,
                     const float* qgrad_dev, const float* bg_dev, const float* tgrad_dev, const int32_t* mapping_dev,
                     int map_offset, int flags, float* h_neg_diag_out_dev, float* g_neg_out_dev);

/* KLConstrainedNgBasedComponentUpdater.apply_NG_update, diagonal branches (gmmvi_modules/
 * ng_based_component_updater.py:447-453, kl() :304-318, :483-490); same contract as gmmvi_update_components_kl with
 * chols[K,D] and H_neg[K,D].  D <= GMMVI_MAX_DIM_BLOCKED. */
int gmmvi_update_components_diag_kl(gmmvi_ctx* ctx, int K, int D, float* means_dev, float* chols_diag_dev,
                                    const float* H_neg_diag_dev, const float* g_neg_dev, const float* stepsizes_dev,
                                    float temperature, float l2_init, float* last_eta_dev, float* l2_dev,
                                    float* num_received_updates_dev, int32_t* success_out_dev, float* kl_out_dev,
                                    int32_t* n_probes_out_dev);
/* NgBasedComponentUpdaterIblr, diagonal branches (:170-174, :188-189, :195-197).  (The reference's direct updater
 * has no diagonal branch.) */
int gmmvi_update_components_diag_iblr(gmmvi_ctx* ctx, int K, int D, float* means_dev, float* chols_diag_dev,
                                      const float* H_neg_diag_dev, const float* g_neg_dev, const float* stepsizes_dev,
                                      float l2_init, float* l2_dev, float* num_received_updates_dev,
                                      int32_t* success_out_dev);

/* ---- MMD (experiments/evaluation/mmd.py:41-58) ------------------------------------------------------------------ */
/* sum_out = sum_{i<Na} sum_{j<Nb} exp(-sum_d inv_bandwidth[d] (A[i,d] - B[j,d])^2)  (fp32 pairs, fp64 sums, fixed
 * summation order).  compute_ustat(sample) is (A, B) = (sample, sample); kernel_mix(sample) is (groundtruth, sample);
 * inv_bandwidth[d] = 1 / (alpha * sigma[d,d]).  scratch_dev holds gmmvi_mmd_scratch_doubles(Na, Nb) doubles. */
size_t gmmvi_mmd_scratch_doubles(int Na, int Nb);
int gmmvi_mmd_pair_sum(gmmvi_ctx* ctx, const float* A_dev, int Na, const float* B_dev, int Nb, int D,
                       const float* inv_bandwidth_dev, double* scratch_dev, double* sum_out_dev);

#ifdef __cplusplus
}
#endif
#endif /* GMMVI_HIP_H */
