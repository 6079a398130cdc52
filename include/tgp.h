/*
 * tgp.h -- C-ABI of the MI355X-native GP-posterior + batch-acquisition engine (libtgp.so).
 *
 * The reference (secondmind-labs/trieste @ 2024-10-16) has NO C/FFI boundary: its plug-in API is a
 * set of duck-typed Python protocols (trieste/models/interfaces.py:38-327,
 * trieste/acquisition/interface.py:27-157, trieste/acquisition/optimizer.py:73-87).  This header
 * is the boundary one level below those protocols: each entry point states the reference
 * interface whose arithmetic it replaces.  The Python host layer (trieste_amd/) binds it with
 * ctypes and re-exposes the reference's protocol names.  See INTEGRATION.md for the stub a
 * trieste maintainer would add.
 *
 * Conventions
 *   - every function returns an int status (TGP_OK == 0); tgp_last_error() gives the message;
 *     nothing aborts: a non-positive-definite K + s2*I returns TGP_ERR_NOT_PD so the host can
 *     raise and the BO loop can record Err (reference bayesian_optimizer.py:855-875,
 *     models/gpflow/models.py:312-315).
 *   - all arrays are float64, row-major, densely packed.
 *   - `where` says where the CALLER's buffers live: TGP_HOST (pageable/pinned host memory; the
 *     library stages them through its own device scratch) or TGP_DEVICE (device pointers on the
 *     handle's GPU, e.g. torch tensors' data_ptr()).  Scalars/small outputs documented as "host"
 *     are always host pointers.
 *   - the handle owns all internal device memory; the caller owns every buffer it passes.
 *   - one host thread per handle at a time; different handles may be driven from different threads
 *     concurrently (tgp_use_private_stream).  Calls are synchronous on return (work is queued on the
 *     handle's HIP stream -- tgp_set_stream / tgp_use_private_stream -- and the stream is
 *     synchronised before returning).  Every call clears the calling thread's sticky HIP error first.
 *   - a trajectory (tgp_traj) must not be used after its model handle is destroyed; destroying it
 *     afterwards is allowed (garbage collectors finalise in arbitrary order).
 *   - acquisition modifiers are handle state: tgp_set_penalization / tgp_set_min_value_samples /
 *     tgp_set_repulsion apply to every later tgp_acq_* call on that handle until cleared; none of them touches
 *     the model (data, hyper-parameters, factorisation).
 *   - no torch types, no C++ types: plain pointers and sizes only.
 */
#ifndef TGP_H
#define TGP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tgp_handle_s* tgp_handle;
typedef struct tgp_traj_s* tgp_traj;

enum tgp_status {
  TGP_OK = 0,
  TGP_ERR_SHAPE = 1,   /* bad sizes / dimension mismatch  (reference: ValueError)             */
  TGP_ERR_NOT_PD = 2,  /* Cholesky pivot <= 0 or NaN      (reference: InvalidArgumentError)   */
  TGP_ERR_ALLOC = 3,   /* hipMalloc failed                (reference: MemoryError hint :864)  */
  TGP_ERR_HIP = 4,     /* any other HIP runtime error                                         */
  TGP_ERR_STATE = 5,   /* call order: e.g. predict before set_data                            */
  TGP_ERR_ARG = 6      /* invalid argument value (negative jitter, unknown kernel, ...)       */
};

/* gpflow.kernels.{SquaredExponential, Matern12, Matern32, Matern52} (chosen by
 * trieste/models/gpflow/builders.py:383-408). */
enum tgp_kernel { TGP_RBF = 0, TGP_MATERN12 = 1, TGP_MATERN32 = 2, TGP_MATERN52 = 3 };

enum tgp_where { TGP_HOST = 0, TGP_DEVICE = 1 };

/* acquisition tails on (mean, var): trieste/acquisition/function/function.py */
enum tgp_acq {
  TGP_ACQ_EI = 0,   /* expected_improvement.__call__            function.py:215-223; param = eta   */
  TGP_ACQ_PI = 1,   /* probability_below_threshold / PoF         function.py:509-510; param = thr  */
  TGP_ACQ_NLCB = 2, /* negative_lower_confidence_bound           function.py:415-416; param = beta */
  TGP_ACQ_AEI = 3,  /* augmented_expected_improvement.__call__  function.py:312-325; param = eta,
                       noise variance = the model's likelihood variance (tgp_set_hyper)        */
  TGP_ACQ_MES = 4,  /* min_value_entropy_search.__call__        entropy.py:195-214; param unused,
                       the min-value samples come from tgp_set_min_value_samples               */
  TGP_ACQ_GIBBON = 5 /* gibbon_quality_term.__call__            entropy.py:479-500 (+ the repulsion
                       term :580-619 while tgp_set_repulsion is in force); param unused         */
};

/* ---- lifetime ------------------------------------------------------------------------- */
/* One handle == one GPR model resident on one GPU (reference: GaussianProcessRegression.__init__,
 * models/gpflow/models.py:88-134).  d = input dimension (1..32). */
int tgp_create(int device_id, int d, int kernel_kind, tgp_handle* out);
int tgp_destroy(tgp_handle h);
const char* tgp_last_error(tgp_handle h); /* h may be NULL: message of the last failed create */
const char* tgp_version(void);
/* Use the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream); NULL = default. */
int tgp_set_stream(tgp_handle h, void* hip_stream);
/* Give the handle its own non-blocking HIP stream (owned and destroyed by the handle).  Several handles
 * driven from several host threads then overlap on the GPU -- used for the concurrent loss evaluations
 * of find_best_model_initialization (models/gpflow/models.py:294-321). */
int tgp_use_private_stream(tgp_handle h);

/* ---- model state ------------------------------------------------------------------------ */
/* Hyper-parameters of kernel + Gaussian likelihood + Constant mean (builders.py:399-443).
 * lengthscales: host [d] (ARD).  Invalidates the factorisation until the next tgp_set_data. */
int tgp_set_hyper(tgp_handle h, double variance, const double* lengthscales, double noise_variance,
                  double mean_const);

/* == GaussianProcessRegression.update_encoded (models.py:171-186) + update_posterior_cache
 * (interface.py:108-112): assign X [N,d], Y [N]; K = k(X,X) + noise*I; L = chol(K) (no jitter);
 * err = Y - c.  Additionally caches W = L^-1 and alpha = K^-1 err.  TGP_ERR_NOT_PD on failure. */
int tgp_set_data(tgp_handle h, const double* X, const double* Y, int64_t N, int where);

/* Rank-k fast path of `update` for the BO loop's usual case -- the new data set is the old one plus k
 * rows, hyper-parameters unchanged (models/gpflow/models.py:171-186 re-assigns the whole data set and
 * gpflow refactorises it; the result is the same posterior): Xnew [k,d], Ynew [k] are appended to the
 * handle's copy of the data and only the trailing block of the factor is recomputed
 * (L21 = A21 W11^T, the Schur complement, its factor, W21): O(k N^2) instead of O(N^3).  When the padded
 * size (N rounded up to 256) changes, the call refactorises everything. */
int tgp_append_data(tgp_handle h, const double* Xnew, const double* Ynew, int64_t k, int where);

/* Make `dst` (same device, input dimension and kernel family) a copy of `src`: hyper-parameters, data and
 * the cached factorisation (device-to-device copies: 3 Npad^2 doubles, ~0.2 ms at N = 4096).  Together with
 * tgp_append_data this is the engine's form of the reference's fantasised model (_fantasized_model,
 * acquisition/function/greedy_batch.py:630-773, built on FastUpdateModel.conditional_predict_f/_joint/_y,
 * models/gpflow/models.py:355-526): instead of re-deriving the conditional posterior of every query batch from
 * the base model, a clone conditioned on the pending points IS an exact GPR on (data + fantasised data) and
 * every sweep on it runs at full speed.  `src` is not modified. */
int tgp_clone_from(tgp_handle dst, tgp_handle src);

/* Local penalization for greedy batches (LocalPenalization / PenalizedAcquisition, greedy_batch.py:54-269):
 * while set, every tgp_acq_values / tgp_acq_argmax / tgp_acq_topk / tgp_acq_value_grad result is
 * acquisition(x) * prod_p phi_p(x) over the P pending points, with dist_p = |x - pending_p|_2:
 *   kind 1 (soft_local_penalizer.__call__, greedy_batch.py:341-354): phi_p = Phi((dist_p - radius_p) / scale_p)
 *   kind 2 (hard_local_penalizer.__call__, greedy_batch.py:376-389): phi_p = ((dist_p / (radius_p + scale_p))^-5 + 1)^(-1/5)
 *   kind 0 or P == 0: clears it.
 * pending host [P,d], radius / scale host [P] (local_penalizer.__init__, greedy_batch.py:287-300:
 * radius = (mean(pending) - eta) / L, scale = sqrt(var(pending)) / L), P <= 1024.  The gradient is the
 * analytic one of the product (the reference differentiates exp(log a + log phi) by autodiff); at dist_p = 0
 * that pending point contributes no gradient. */
int tgp_set_penalization(tgp_handle h, int kind, const double* pending, const double* radius, const double* scale,
                         int64_t P);

/* Entropy-search tails (acquisition/function/entropy.py).  The S samples of the objective's minimum value
 * (MinValueEntropySearch.prepare_acquisition_function :118-139 / GIBBON._update_quality_term :405-419 draw them
 * with a Thompson or Gumbel sampler) are handle state: samples host [S], S <= 4096; S == 0 clears.  With
 * gamma_s = (s - mean) / max(sqrt(var), 1e-8), ratio = pdf(gamma) / Phi(-gamma):
 *   TGP_ACQ_MES:    mean_s [ -gamma ratio / 2 - log Phi(-gamma) ]
 *   TGP_ACQ_GIBBON: -1/2 mean_s log(1 + rho^2 ratio (gamma - ratio)),  rho^2 = var / (var + noise)
 * log Phi as tfp's log_ndtr (asymptotic series below -20).  All tgp_acq_* entry points accept the two kinds;
 * the gradient is analytic. */
int tgp_set_min_value_samples(tgp_handle h, const double* samples, int S);

/* GIBBON's repulsion term (gibbon_repulsion_term.__call__, entropy.py:580-619): while set, TGP_ACQ_GIBBON
 * results are quality(x) + weight / 2 * (log(var_twin(x) + noise) - log(var(x) + noise)), where `twin` is this
 * model conditioned additionally on the m pending points (tgp_clone_from + tgp_append_data with any
 * observations: variances do not depend on them).  The reference forms V_det = yvar - A^T (B + noise I)^-1 A
 * from covariance_between_points(x, pending) and predict_joint(pending); that Schur complement IS the
 * twin's predictive variance + noise.  weight = 1 / m^2 for rescaled_repulsion, else 1.  `twin` is not owned
 * and must outlive the setting; twin == NULL clears.  When the twin is literally this model's data plus m <= 16
 * appended rows with equal hyper-parameters (verified on the device once per data version) its variance is
 * evaluated as the rank-m correction var(x) - sum_r (W'_r . k'(x))^2 over the twin's last m rows of W' = L'^-1
 * -- m kernel sums per candidate -- instead of a second N^2 sweep; any other twin is swept itself. */
int tgp_set_repulsion(tgp_handle h, tgp_handle twin, double weight);

/* The penalization alone, prod_p phi_p(x) at Xq [M,d] -> out [M] (the penalizer objects are callables in the
 * reference: local_penalizer.__call__, greedy_batch.py:341-354, 376-389).  TGP_ERR_STATE if none is set. */
int tgp_penalization_values(tgp_handle h, const double* Xq, int64_t M, double* out, int where);

int tgp_get_sizes(tgp_handle h, int64_t* N, int* d);
/* Negative log marginal likelihood of the current (hyper-parameters, data) and its gradient:
 * value (host scalar); grad (host [d + 3]; NULL = value only, which skips the K^-1 product and the
 * pair reduction) = d/d lengthscales[d], d/d variance,
 * d/d noise_variance, d/d mean_const, all in the natural (constrained) parameters.  This is the
 * likelihood part of the loss gpflow's Scipy optimizer minimises in
 * GaussianProcessRegression.optimize_encoded (models/gpflow/models.py:256-292); priors and
 * parameter transforms are host-side scalars. */
int tgp_nlml(tgp_handle h, double* value, double* grad);
/* TRIAL evaluation: the value tgp_nlml(h, value, NULL) would return after tgp_set_hyper + tgp_set_data with the data
 * already on the device (tgp_set_data has run once on this handle) at the CURRENT hyper-parameters -- what
 * find_best_model_initialization does for every prior draw (reference models.py:294-321: assign the draw, evaluate
 * training_loss).  From N = 257 on (Npad >= 512: where `update` is the persistent kernel; rounds 3 - 5: from 3841 on) only the
 * Cholesky factor is built (half the tile products of an update, so that more
 * trial handles run side by side: tgp_set_update_concurrency), err^T K^-1 err comes from a block forward substitution.
 * The handle is left WITHOUT a posterior: queries fail with TGP_ERR_STATE until the next tgp_set_data.  NOT_PD as for
 * tgp_set_data. */
int tgp_nlml_trial(tgp_handle h, double* value);
/* B trial evaluations at once: hypers [B][d + 3] (variance, lengthscales [d], noise_variance, mean_const per member),
 * values [B], status [B] (TGP_OK or TGP_ERR_NOT_PD per member; a member that breaks down gets a NaN value and does not
 * disturb the others).  From N = 257 on (rounds 4 - 5: 3841) up to 15 members -- 45 up to N = 1024 -- share ONE persistent launch: B chain workgroups, one list of
 * tile tasks over all members' factor-only plans (the start order of a list-scheduling simulation of the B graphs on the
 * workers they share) -- a single factorisation leaves half of the compute units idle behind its chain, and HIP runs at
 * most three such launches side by side.  Each value equals tgp_nlml_trial's at
 * the same hyper-parameters bit for bit.  The handle's own hyper-parameters and posterior are untouched (the members
 * live in scratch matrices, 3 N^2 doubles each: a process-wide scratch per device, allocated on first use and kept --
 * tgp_release_scratch frees it; concurrent calls on one device are serialised).  A launch takes at most 16 (N <= 1024: 45) members, at
 * most a quarter of the compute units as chains, at most 12 GiB and at most four fifths of the device's free memory; a
 * failed allocation halves the group.  Below that size -- and when not even one member's matrices fit, or the device has
 * too few compute units for chains and workers -- the members are evaluated one after the other on the handle
 * (tgp_nlml_trial's arithmetic, the same values).
 * Replaces the loop of find_best_model_initialization (reference models/gpflow/models.py:294-321). */
int tgp_nlml_trial_batch(tgp_handle h, const double* hypers, int B, double* values, int* status);
/* Free the process-wide scratch tgp_nlml_trial_batch keeps on `device_id` (waits for a call in progress; the next batched
 * fit allocates it again).  It is kept between fits on purpose -- a BO loop fits a fresh model every step, and allocating /
 * releasing gigabytes per model cost 10 - 40 ms each way -- so nothing frees it implicitly.  The reference has no
 * counterpart (TensorFlow's allocator owns its arena: models/gpflow/models.py:294-321 never sees memory). */
int tgp_release_scratch(int device_id);
/* Read back the cache (tests / checkpoint-free restore checks): any pointer may be NULL.
 * L [N,N] lower (upper = 0), Winv [N,N] = L^-1, alpha [N].  `where` applies to all three. */
int tgp_get_factor(tgp_handle h, double* L, double* Winv, double* alpha, int where);

/* ---- posterior -------------------------------------------------------------------------- */
/* == GPflowPredictor.predict_encoded (interface.py:119-124): mean [M], var [M] = clip(k** -
 * |L^-1 k*|^2, 1e-12, max).  Either output may be NULL. */
int tgp_predict(tgp_handle h, const double* Xq, int64_t M, double* mean, double* var, int where);

/* Posterior mean only (no N^2 term), used for eta.  mean [M]. */
int tgp_predict_mean(tgp_handle h, const double* Xq, int64_t M, double* mean, int where);

/* == GPflowPredictor.predict_joint_encoded (interface.py:126-133): Xq [G,q,d] ->
 * mean [G,q], cov [G,q,q] = K** - A^T A with the DIAGONAL clipped to >= 1e-12.  q <= 64.  (Round 6: a call of G * q <= 2048
 * points -- and tgp_qei / tgp_reparam_samples of that size -- forms A as one skinny triangular product and A^T A as one Gram product,
 * the arithmetic of tgp_joint_forward, instead of launching the joint kernel built for 10^5 groups: 0.1 - 1 ms instead of 4.9 ms at
 * N = 2048 / 18 ms at N = 4096; same values to the rounding of another summation order; tgp_set_variant bit 10 keeps the kernel.) */
int tgp_predict_joint(tgp_handle h, const double* Xq, int64_t G, int q, double* mean, double* cov,
                      int where);

/* == ExpectedImprovement.prepare/update (function.py:145-149): eta = min_i mean(X_i) over the
 * model's own training inputs.  eta: host scalar. */
int tgp_eta(tgp_handle h, double* eta);

/* ---- acquisition sweep ------------------------------------------------------------------- */
/* acquisition values out [M] (== expected_improvement.__call__ etc. on x [M,1,d]). */
int tgp_acq_values(tgp_handle h, int acq_kind, double param, const double* Xq, int64_t M,
                   double* out, int where);

/* Acquisition value AND its gradient w.r.t. the query point, val [P], grad [P,d] -- what
 * tfp.math.value_and_gradient(_objective_value, x) yields inside the L-BFGS-B refinement of
 * generate_continuous_optimizer (optimizer.py:344-745, call at :628-629).  Analytic: d mean/dx =
 * sum_k alpha_k dk_k/dx, d var/dx = -2 (K^-1 kstar)^T dkstar/dx; zero variance-gradient where the 1e-12 clip is
 * active (tf.clip_by_value). */
int tgp_acq_value_grad(tgp_handle h, int acq_kind, double param, const double* Xq, int64_t P, double* val,
                       double* grad, int where);

/* Fused predict + acquisition + arg-max (== _get_max_discrete_points, optimizer.py:124-150, over
 * `optimize_discrete` / `generate_random_search_optimizer` candidates).  First index wins ties
 * (tf.math.argmax).  Outputs are HOST: best_val, best_idx (= index_base + local index), best_x [d]
 * (may be NULL).  Values are never written to HBM. */
int tgp_acq_argmax(tgp_handle h, int acq_kind, double param, const double* Xq, int64_t M,
                   int64_t index_base, double* best_val, int64_t* best_idx, double* best_x,
                   int where);

/* Streaming top-k (== generate_initial_points, optimizer.py:247-341, one batch): k <= 1024.
 * Descending by value, ties by lower index.  Host outputs vals [k], idx [k]. */
int tgp_acq_topk(tgp_handle h, int acq_kind, double param, const double* Xq, int64_t M,
                 int64_t index_base, int k, double* vals, int64_t* idx, int where);

/* == Box.sample (space.py:843-867) on the device: out [M,d] DEVICE pointer, uniform in
 * [lower, upper) per dimension; element (row first+i, col c) depends only on (seed, first+i, c)
 * (Philox4x32-10), so shards of one logical sample are consistent across GPUs.
 * TF's own Philox stream is not reproducible outside TF: parity tests pass candidates in. */
int tgp_sample_box(tgp_handle h, uint64_t seed, int64_t first, int64_t M, const double* lower,
                   const double* upper, double* out_device);

/* ---- batch Monte-Carlo EI ---------------------------------------------------------------- */
/* == batch_monte_carlo_expected_improvement.__call__ (function.py:1181-1186) with
 * BatchReparametrizationSampler.sample (sampler.py:208-287): Xq [G,q,d], eps [q,S] (the
 * sampler's fixed draws, host or device like Xq), out [G] = mean_S max(eta - min_q(mean +
 * chol(cov + jitter*I) eps), 0).  q <= 64. */
int tgp_qei(tgp_handle h, const double* Xq, int64_t G, int q, const double* eps, int S, double eta,
            double jitter, double* out, int where);

/* == BatchReparametrizationSampler.sample (sampler.py:208-287) itself: out [G,S,q] = mean +
 * (chol(cov + jitter*I) eps)^T for Xq [G,q,d], eps [q,S].  q <= 64. */
int tgp_reparam_samples(tgp_handle h, const double* Xq, int64_t G, int q, const double* eps, int S,
                        double jitter, double* out, int where);

/* == value and gradient of a BATCH acquisition function at the handful of q-batches an L-BFGS-B iteration holds: the reference
 * optimises BatchMonteCarloExpectedImprovement through batchify_joint(generate_continuous_optimizer) (acquisition/optimizer.py:
 * 897-934, 344-560) with tfp.math.value_and_gradient (optimizer.py:628-629), i.e. autodiff THROUGH predict_joint
 * (models/gpflow/interface.py:126-133), the Cholesky factor of the q x q covariance and the reparametrised samples
 * (models/gpflow/sampler.py:276-287).  The engine's share of that derivative, round 6:
 *   tgp_joint_forward: mean [G,q], cov [G,q,q] of Xq [G,q,d] -- tgp_predict_joint's values to the rounding of another summation
 *     order (diagonal clipped at 1e-12), as K*^T, one skinny triangular product and a Gram product instead of the joint kernel
 *     built for 10^5 groups; any q, G * q <= 2048 per call;
 *   tgp_joint_vjp: grad [G,q,d] = d/dXq of  sum_gi gmean[g,i] mean[g,i] + sum_gij gcov[g,i,j] cov[g,i,j]  (gcov need not be
 *     symmetric: both triangles of cov count as written; a clipped diagonal entry has zero gradient at the caller: pass 0).
 *     With these two the q x q factorisation and its adjoint in between are the caller's (host) arithmetic.  TGP_ERR_SHAPE
 *     beyond 2048 points.
 *   tgp_qei_value_grad: both of them with the q x q arithmetic in between ON THE DEVICE -- val [G] = qEI (tgp_qei's value), grad
 *     [G,q,d] = its gradient w.r.t. Xq for eps [q,S], eta, jitter: one wave per group factorises cov + jitter I, draws the samples,
 *     accumulates the adjoints of mean and factor and applies the Cholesky adjoint; one call, one synchronisation.  q <= 64,
 *     G * q <= 2048, 8 (2 q (q|1) + 128) + 4 S bytes of LDS <= 160 KiB (S <= ~23 000 at q = 64), else TGP_ERR_SHAPE: the pair
 *     above takes any q and S.  TGP_ERR_NOT_PD as tgp_qei. */
int tgp_joint_forward(tgp_handle h, const double* Xq, int64_t G, int q, double* mean, double* cov, int where);
int tgp_joint_vjp(tgp_handle h, const double* Xq, int64_t G, int q, const double* gmean, const double* gcov, double* grad,
                  int where);
int tgp_qei_value_grad(tgp_handle h, const double* Xq, int64_t G, int q, const double* eps, int S, double eta, double jitter,
                       double* val, double* grad, int where);

/* == GaussianProcessRegression.covariance_between_points_encoded (models/gpflow/models.py:188-254):
 * out [P1,P2] = k(X1, X2) - (L^-1 k(X, X1))^T (L^-1 k(X, X2)); no clipping (the reference applies
 * none here).  X1 [P1,d], X2 [P2,d]. */
int tgp_cov_between(tgp_handle h, const double* X1, int64_t P1, const double* X2, int64_t P2, double* out,
                    int where);

/* == GPflowPredictor.sample_encoded (models/gpflow/interface.py:135-137 -> gpflow
 * predict_f_samples: predict_f(full_cov=True) then sample_mvn): exact joint posterior samples at
 * Xq [n,d] given the standard-normal draws eps [n,S]:  out [S,n] = mean + chol(cov + jitter I) eps,
 * cov unclipped (gpflow applies no clip here), jitter = gpflow default_jitter = 1e-6.  Any n (the
 * n x n factorisation runs on the device); what ExactThompsonSampler.sample
 * (acquisition/sampler.py:88-123) -- the DEFAULT sampler of DiscreteThompsonSampling
 * (rule.py:935-938) -- draws its minimisers from. */
int tgp_sample_joint(tgp_handle h, const double* Xq, int64_t n, const double* eps, int S, double jitter,
                     double* out, int where);

/* ---- decoupled Thompson trajectories ------------------------------------------------------ */
/* == DecoupledTrajectorySampler._prepare_weight_sampler + weight_sampler(B)
 * (sampler.py:661-738) given the draws: rff_W [F,d], rff_b [F] (the RFF basis, gpflux
 * RandomFourierFeaturesCosine), w [F,B] (prior weights), xi [N,B] (noise draws).
 * v = (K + noise I)^-1 ((Y - c) + sqrt(noise) xi - Phi_Z w) is computed from the CACHED factor
 * (the reference re-factorises per trajectory, sampler.py:730).  All inputs HOST. */
int tgp_traj_create(tgp_handle h, const double* rff_W, const double* rff_b, int F, const double* w,
                    const double* xi, int B, tgp_traj* out);
/* == RandomFourierFeatureTrajectorySampler._prepare_weight_sampler + theta_posterior.sample(B)
 * (sampler.py:518-591) given the standard-normal draws eps [F,B]: the weights theta of the F scaled
 * Fourier features are Gaussian given the data -- "design space" (F < N: an F x F factorisation,
 * sampler.py:529-556) or "gram space" (N <= F: N x N, sampler.py:558-591) -- and
 * theta = mean + chol(cov) eps.  The resulting trajectory is f_b(x) = phi(x) . theta_b + c (no
 * canonical part); it is evaluated / minimised / differentiated by the same tgp_traj_* calls. */
int tgp_traj_create_rff(tgp_handle h, const double* rff_W, const double* rff_b, int F, const double* eps,
                        int B, tgp_traj* out);
/* theta [F,B] (host), for tests. */
int tgp_traj_get_theta(tgp_traj t, double* theta);
int tgp_traj_destroy(tgp_traj t);
/* canonical weights v [N,B] (host), for tests. */
int tgp_traj_get_v(tgp_traj t, double* v);
/* == feature_decomposition_trajectory.__call__ (sampler.py:901-936): Xq [M,d] shared by all B
 * trajectories (per_traj_inputs = 0) or [M,B,d] (= 1); out [M,B]. */
int tgp_traj_eval(tgp_traj t, const double* Xq, int64_t M, int per_traj_inputs, double* out,
                  int where);
/* Value and gradient of each trajectory at ITS OWN point: Xq [P,B,d] -> val [P,B], grad [P,B,d].
 * == what tfp.math.value_and_gradient (acquisition/optimizer.py:628-629) yields for the
 * (negated) trajectories built by Greedy/ParallelContinuousThompsonSampling
 * (acquisition/function/continuous_thompson_sampling.py:30-245); the sign flip is the host's. */
int tgp_traj_value_grad(tgp_traj t, const double* Xq, int64_t P, double* val, double* grad, int where);
/* == ThompsonSamplerFromTrajectory.sample (acquisition/sampler.py:262-271) for B trajectories:
 * arg-min over Xq [M,d]; host outputs best_val [B], best_idx [B] (first index wins ties). */
int tgp_traj_argmin(tgp_traj t, const double* Xq, int64_t M, int64_t index_base, double* best_val,
                    int64_t* best_idx, int where);

/* ---- device-resident winners (one synchronisation per step) ----------------------------------
 * The synchronous arg-max / arg-min above hand their winners back as host scalars.  These forms leave
 * them on the device and only ENQUEUE work on the handle's stream (tgp_set_stream /
 * tgp_use_private_stream): no host synchronisation, so a sharded step is sweep -> all-gather of the
 * pairs (RCCL, on the same stream) -> tgp_merge_winners_async -> ONE device-to-host copy.  Candidates
 * must be device-resident.  Same arithmetic, same first-index tie-break as tgp_acq_argmax /
 * tgp_traj_argmin (== tf.math.argmax in _get_max_discrete_points, acquisition/optimizer.py:149-150;
 * tf.math.argmin in ThompsonSamplerFromTrajectory.sample, acquisition/sampler.py:262-271).
 *   pair_device   DEVICE [2]:    value, global index (int64 stored in the second 8-byte slot)
 *   pairs_device  DEVICE [2][B]: B values, then B global indices (int64 bit patterns)
 * tgp_merge_winners_async: gathered_device [P][2][V] (what an all-gather of P ranks' [2][V] pairs
 * yields) -> out_device [2][V]; larger value wins (smaller if minimize != 0), ties go to the smaller
 * global index, NaN values and negative / "nothing found" indices never win; no valid entry: (NaN, -1).
 * tgp_stream_synchronize waits for everything enqueued on the handle's stream. */
int tgp_acq_argmax_async(tgp_handle h, int acq_kind, double param, const double* Xq_device, int64_t M,
                         int64_t index_base, double* pair_device);
int tgp_traj_argmin_async(tgp_traj t, const double* Xq_device, int64_t M, int64_t index_base,
                          double* pairs_device);
int tgp_merge_winners_async(tgp_handle h, const double* gathered_device, int P, int V, int minimize,
                            double* out_device);
int tgp_stream_synchronize(tgp_handle h);

/* ---- single-controller multi-GPU group -------------------------------
 * One host process, one model replica per device: the BayesianOptimizer / Ask-Tell loop -- and the
 * user's observer, which the reference calls exactly once per step (bayesian_optimizer.py:793-806) --
 * runs ONCE, while the candidate sweep shards over the devices.  The reference has no multi-device
 * path; this is the form its single-process loop can call.
 *   - every member is an ordinary tgp_handle on its own device with a private stream
 *     (tgp_group_member borrows it: posterior queries, gradients, fits go to member 0 or any replica);
 *   - model state is REPLICATED: set_hyper / set_data / append_data run the same deterministic update on
 *     every device concurrently (one host worker thread per member), bit-identical on identical GPUs;
 *   - candidates are SHARDED: member r owns the contiguous rows shard(M, r) = [r*ceil(M/n), ...) of the
 *     logical candidate table, with index_base = its first row, so global indices and the first-index
 *     tie-break survive;
 *   - the only exchange is the winners: every member's sweep leaves its (value, index) pairs on its device,
 *     one all-gather over an in-process RCCL communicator (ncclCommInitAll over xGMI; 16 B per member and
 *     vectorised function) brings them together, a merge kernel on member 0 picks (max value, min index)
 *     and ONE 16-byte copy reaches the host.  merge = TGP_MERGE_PEER replaces the all-gather by 16-byte
 *     peer copies into member 0 (no RCCL needed).  RCCL is resolved at run time (dlopen librccl.so.1):
 *     TGP_MERGE_RCCL fails with TGP_ERR_STATE if it is not loadable.
 * One host thread drives a group at a time.  Group calls are synchronous on return.
 * Test aid: with TGP_GROUP_ALLOW_DUPLICATES set in the environment and merge = TGP_MERGE_PEER a device may be listed
 * several times (several members share one GPU), so the multi-member path can be exercised on a single-GPU box. */
typedef struct tgp_group_s* tgp_group;
typedef struct tgp_group_traj_s* tgp_group_traj;
enum tgp_merge { TGP_MERGE_RCCL = 0, TGP_MERGE_PEER = 1 };

int tgp_group_create(const int* device_ids, int n_dev, int d, int kernel_kind, int merge, tgp_group* out);
int tgp_group_destroy(tgp_group g);
const char* tgp_group_last_error(tgp_group g); /* g may be NULL: message of the last failed create */
/* n_dev, the merge actually in use, and the number of RCCL ranks the communicator reports (0 for PEER) */
int tgp_group_info(tgp_group g, int* n_dev, int* merge, int* rccl_ranks);
int tgp_group_member(tgp_group g, int i, tgp_handle* out); /* borrowed: do not destroy */
/* replicated model state: same arguments as tgp_set_hyper / tgp_set_data / tgp_append_data, HOST buffers */
int tgp_group_set_hyper(tgp_group g, double variance, const double* lengthscales, double noise_variance,
                        double mean_const);
int tgp_group_set_data(tgp_group g, const double* X, const double* Y, int64_t N);
int tgp_group_append_data(tgp_group g, const double* Xnew, const double* Ynew, int64_t k);
/* The group's resident candidate table [M,d], sharded: either scattered from a HOST array
 * (DiscreteSearchSpace.points, space.py:394-407) or generated on the devices as ONE logical Philox sample --
 * element (row, col) depends on (seed, row, col) only, so the table does not depend on n_dev
 * (Box.sample, space.py:843-867; see tgp_sample_box). */
int tgp_group_set_candidates(tgp_group g, const double* Xq, int64_t M);
int tgp_group_sample_candidates(tgp_group g, uint64_t seed, int64_t M, const double* lower, const double* upper);
/* == tgp_acq_argmax over the resident table; best_x [d] (host, may be NULL) is read from the owning member */
int tgp_group_acq_argmax(tgp_group g, int acq_kind, double param, double* best_val, int64_t* best_idx,
                         double* best_x);
/* == tgp_acq_topk over the resident table: every member's top-k (k <= 1024, shard size >= k or the shard's
 * size), merged on the host (value desc, index asc). */
int tgp_group_acq_topk(tgp_group g, int acq_kind, double param, int k, double* vals, int64_t* idx);
/* == tgp_qei with the G q-batches sharded over the members: Xq host [G,q,d], eps host [q,S], out host [G] */
int tgp_group_qei(tgp_group g, const double* Xq, int64_t G, int q, const double* eps, int S, double eta,
                  double jitter, double* out);
/* decoupled Thompson trajectories replicated on every member (same draws): tgp_traj_create per member;
 * tgp_group_traj_argmin = tgp_traj_argmin over the resident table (host outputs [B]). */
int tgp_group_traj_create(tgp_group g, const double* rff_W, const double* rff_b, int F, const double* w,
                          const double* xi, int B, tgp_group_traj* out);
int tgp_group_traj_destroy(tgp_group_traj t);
int tgp_group_traj_argmin(tgp_group_traj t, double* best_val, int64_t* best_idx);
/* duration (ms) of the slowest member's dominant kernel in the most recent sharded sweep */
int tgp_group_last_kernel_ms(tgp_group g, double* ms);

/* ---- measurement ------------------------------------------------------------------------- */
/* Duration (ms, HIP events on the handle's stream) and launch count of the dominant kernel of
 * the most recent sweep-type call (predict / acq_values / acq_argmax / qei / traj_*). */
int tgp_last_kernel_ms(tgp_handle h, double* ms, int* launches);
/* Arithmetic of the plain posterior sweeps (tgp_predict / tgp_acq_values / tgp_acq_argmax(_async) / tgp_acq_topk;
 * joint-mode, gradients, trajectories and `update` always run in float64):
 *   TGP_PREC_F64  (default) W K* on the float64 matrix cores -- the parity path;
 *   TGP_PREC_I8X4 W K* on the int8 matrix cores with both operands split error-free into four signed 8-bit digit
 *                 planes (Ozaki scheme: 10 int8 products with exact int32 accumulation stand in for one float64
 *                 product; row / column scales S_i = (1 + 2^-7) max_k |W_ik|, S' = (1 + 2^-7) s_f^2; what is lost
 *                 are the digit pairs below 2^-32 of S_i S').  K* generation, mean, column norms, acquisition tail
 *                 and arg-max stay float64.  An EMULATED-precision option for throughput: no per-result guarantee,
 *                 never the default and never what the float64 parity claims are made on.
 *   TGP_PREC_I8X5 the same with five digit planes (15 int8 products, digit pairs below 2^-40 dropped); d <= 16.
 *   TGP_PREC_AUTO the split-precision sweep WITH an a-posteriori repair: the int8 kernel prices every candidate's own
 *                 truncation error on the variance (K_SIGMA = 8 standard deviations of  2 2^-32.8 S' sqrt(sum_i c_i^2
 *                 S_i^2 (i+1)), c = W k*), every candidate whose bound exceeds the parity tolerance 1e-5 var + min(64
 *                 eps s_f^2 (1 + N s_f^2 / s^2), 1e-6 s_f^2) -- and, in a fused arg-max, every candidate whose value
 *                 interval reaches the best lower bound -- is recomputed by the float64 kernel in the same call.  The
 *                 bound is a STATISTICAL model of the dropped digit pairs (independent errors; measured against 80-bit
 *                 sums), not a worst case; what it promises -- results inside the parity tolerance candidate by
 *                 candidate, the float64 arg-max -- holds as far as that model does, and float64 stays the only
 *                 arithmetic the parity claims are made on.  Every sweep therefore carries a CANARY in two strata: a
 *                 UNIFORM sample -- one candidate in 4096, at an offset that is a function of (N, hyper-parameters, M, rung)
 *                 -- and an ADVERSARIAL one -- of every 1 / 64 of the sweep the unflagged candidate whose bound sits
 *                 closest to its tolerance, where a failure of the error model would show first.  Both are recomputed
 *                 in float64 whatever their intervals say and |var_f64 - var_int8| is compared with the candidate's
 *                 own bound on the device (+ the float64 kernels' rounding floor); a violation in either moves the
 *                 engine one rung down -- before the next sweep, and for the SYNCHRONISING entry points (tgp_predict,
 *                 tgp_acq_values, tgp_acq_argmax, tgp_acq_topk) before they return: they repeat their sweeps on the
 *                 next rung.  The enqueue-only entry points (tgp_acq_argmax_async, the group / multi-device sweeps,
 *                 tgp_merge_winners_async) return what the rung in effect computed; a violation they meet takes effect
 *                 at the next call (tgp_get_auto_report after the stream is synchronised tells).
 *                 REPRODUCIBILITY: the samples are compared, never written over the int8 results; only what a sweep
 *                 FLAGGED (bound above the tolerance; in a fused arg-max: interval reaching the best lower end) is
 *                 replaced by its float64 value.  tgp_predict / tgp_acq_values under AUTO are therefore a pure function
 *                 of (model state, rung, candidate) -- two identical calls return identical bits, and a candidate's
 *                 value does not depend on which other candidates share its call; the fused arg-max returns the
 *                 float64 winner.  The ladder: four planes, five (d <= 16), float64; a rung is also left when a sweep
 *                 had to recompute more than 10 % of its candidates.  tgp_set_precision restarts the ladder; after
 *                 tgp_set_hyper / tgp_clone_from the next SWEEP does unless a rung was left under hyper-parameters
 *                 within a factor two of the ones then in effect (a fit's trial evaluations do not count).
 *                 N <= 16384 (int32 accumulators), float64 above. */
enum tgp_precision { TGP_PREC_F64 = 0, TGP_PREC_I8X4 = 1, TGP_PREC_I8X5 = 2, TGP_PREC_AUTO = 3 };
int tgp_set_precision(tgp_handle h, int precision);
/* What was asked for, what the next plain sweep will run (never TGP_PREC_AUTO) and, under AUTO, the fraction of its
 * candidates the last completed sweep recomputed in float64 (-1: none yet / not AUTO).  Under AUTO it needs data
 * (TGP_ERR_STATE before tgp_set_data) and synchronises the handle's stream. */
int tgp_get_precision(tgp_handle h, int* requested, int* effective, double* repaired_fraction);
/* K_SIGMA of TGP_PREC_AUTO's per-candidate bound (default 8; restarts the ladder).  Smaller values recompute fewer
 * candidates and make the canary stricter: with a bound tighter than the arithmetic's real error the canary fires and
 * the ladder ends on float64 (tests/test_gpu_i8.py drives it that way). */
int tgp_set_auto_sigma(tgp_handle h, double k_sigma);
/* The canary of TGP_PREC_AUTO since tgp_set_precision / tgp_set_auto_sigma: sampled candidates compared, samples outside their
 * bound, the worst |var_f64 - var_int8| / bound seen, rungs left BECAUSE of a violation, the current rung (0 four
 * planes, 1 five, 2 float64; -1 when the precision is not AUTO).  Synchronises the handle's stream.  The reference has
 * no counterpart: its arithmetic is float64 throughout (trieste/models/gpflow/interface.py:119-124); this is the
 * run-time check that keeps the emulated arithmetic inside the tolerance the reference's own tests state
 * (tests/unit/models/gpflow/test_models.py:363-365). */
int tgp_get_auto_report(tgp_handle h, int64_t* checked, int64_t* violations, double* worst_ratio, int* demotions,
                        int* level);
/* The same report per stratum of the canary (round 6): checked2 / violations2 / worst_ratio2 are arrays of TWO -- [0] the
 * uniform sample, [1] the adversarial one (tgp_get_auto_report returns their sums / maximum); slack_saved: samples that
 * were inside bound + rounding floor but outside the bound alone (float64 rounding of the reference values, not the int8
 * arithmetic).  Accumulated since tgp_set_precision / tgp_set_auto_sigma.  Synchronises the handle's stream.  No reference
 * counterpart (see tgp_get_auto_report). */
int tgp_get_auto_strata(tgp_handle h, int64_t* checked2, int64_t* violations2, double* worst_ratio2, int64_t* slack_saved);
/* Sweep launch policy knob for experiments and tests (0 = default): bit 0 = never use the row-group split of
 * small launches, bit 1 = always use it, bit 2 = joint mode on the first-generation kernel (64-column group slots;
 * dp = 32 always uses it), bit 3 = fused plain launches on the register-staged kernel instead of the LDS-DMA one, bit 4 =
 * `update` through the recursion of dependent launches instead of the persistent task-DAG kernel, bit 5 = the persistent
 * kernel from Npad = 256 on (default: from 512 on since round 6, from 4096 on before), bit 6 = TGP_PREC_AUTO recomputes every
 * flagged candidate through the row-group-split sweep instead of the product path it takes for lists of up to 512, bit 7 =
 * the int8 sweep's workgroups take candidate blocks i, i + #WG, ... instead of drawing them from a counter, bit 8 = the
 * persistent `update` kernel's plan with whole tiles only (default since round 6 at 3 <= Npad / 128 < 48: its two critical
 * single products as half-tile tasks, see tgp_dag_plan), bit 9 = the persistent `update` kernel's chain as ONE workgroup
 * (rounds 3 - 5; default since round 6 at 3 <= Npad / 128 < 48: TWO workgroups swapping the roles of leaf and helper, which
 * forms L(j+1,j) as a blocked triangular solve instead of a product with the inverted diagonal block), bit 10 = tgp_predict at
 * <= 2048 points through a sweep launch and tgp_predict_joint / tgp_qei / tgp_reparam_samples of <= 2048 points (groups x q) through
 * the joint kernel (rounds 1 - 5) instead of skinny triangular products (round 6: the default when no sweep / joint policy bit is
 * set -- and, for tgp_predict, no arithmetic other than float64; joint mode is float64 on every path).  Bits 0-3, 7, 8: every
 * setting computes the same arithmetic on every candidate / matrix entry, bit for bit; bits 4 - 6, 9: the same values up to the
 * rounding of another summation order (bit 10 as well). */
int tgp_set_variant(tgp_handle h, int variant);
/* `update` on SEVERAL handles at once (the prior draws of a hyper-parameter fit: reference models.py:294-321 evaluates
 * them one after the other): the persistent update kernel of a handle takes 1 / n of the compute units (n = 1 ... 16, default
 * 1), so that n handles factorising concurrently on private streams run side by side instead of queueing for the whole
 * device -- one factorisation at N = 4096 keeps only half of the GPU's workgroups busy.  The factor does not depend on n,
 * bit for bit. */
int tgp_set_update_concurrency(tgp_handle h, int n);


/* ---- development / test aid (host only: no device is touched) -------------------------------------------------------
 * The static task list of the persistent `update` kernel (csrc/tgp_kernels_dag.hip) for nb = Npad / 128 block rows of
 * matrices with leading dimension ld: what tgp_set_data uploads for 4096 <= Npad <= 16128.  tests/test_dag_plan.py
 * executes it on numpy blocks (any valid interleaving gives L and W) and checks that every pair of tasks touching the
 * same tile with a write among them is ordered by the flags.  Matrices: 0 = K + s I (tiles carry the partial sums),
 * 1 = L, 2 = W.  flags: bit 0 = B operand natural (else transposed), bit 1 = add the tile already at c_off, bit 2 =
 * negate the product, bit 3 = the task computes HALF of the tile's rows (bit 4: rows 64 .. 127, else 0 .. 63; the split plan
 * of `flags` bit 1 below), bit 5 (upper halves of the plan of `flags` bit 2) = dep3 is NOT waited for before the task starts:
 * it is the flag of the task's lower-half sibling -- an earlier entry of `order` -- which the task waits for at its END,
 * before its own flag goes up; that flag then stands for both halves and a consumer of the whole tile waits for it alone.
 * dep[], dep3: flag ids to wait for (0xffffffff = none); set: the task's own flag (= its position).
 * order[ntasks] (may be NULL): the DISPATCH order -- workers draw positions of it with one atomic and wait for the flags
 * of what they drew; it is a topological order (every flag a task waits for belongs to a chain step or to a task
 * earlier in it), which is what makes that dispatch deadlock-free whatever the residency.  tasks[0 .. *n_urgent) are the
 * tasks on the per-row critical paths (the last burst of a tile and the single-tile products).
 * Flag ids >= ntasks belong to the chain workgroup: ntasks + j = diagonal block j factored and inverted,
 * ntasks + nb + j = L(j+1, j) stored.  chain_dep[3 nb]: chain_dep[2 j], chain_dep[2 j + 1]: what the chain waits for before
 * the leaf of step j and before its L(j+1, j); chain_dep[2 nb + j]: a second flag before L(j+1, j) (the split plan finishes
 * P(j+1, j) in two halves) or none.  Returns TGP_OK, TGP_ERR_ARG, or TGP_ERR_SHAPE when cap < *ntasks (which is always
 * set). */
typedef struct {
  uint32_t a_off, b_off, c_off, o_off, nk, flags;
  uint8_t a_mat, b_mat, c_mat, o_mat;
  uint32_t dep[3], set, dep3;
} tgp_dag_task;
int tgp_dag_plan(int nb, int64_t ld, tgp_dag_task* tasks, int64_t cap, int64_t* ntasks, int64_t* n_urgent,
                 uint32_t* chain_dep, uint32_t* order,
                 int flags /* bit 0: the factor only (tgp_nlml_trial's plan); bit 1: the SPLIT plan of the single update at the
                              chain-bound sizes (round 6): T(i, i-2) and the last burst of tile (i, i-1) -- the two dependent
                              single products between a leaf and the chain's next sub-diagonal product -- as two half-tile
                              tasks each; bit 2 (with bit 1): the plan of the launch whose chain is TWO workgroups
                              (default at those sizes; tgp_set_variant bit 9 = one): EVERY T(i,j) and the last burst of EVERY
                              tile below the diagonal as two halves (task flag bit 5), the order simulated for one worker
                              less and for that chain's timing; bits 8-15: B > 0 -> `order` is the dispatch
                              list of a batched launch of B members (tgp_nlml_trial_batch): B * ntasks entries
                              (member << 24 | task), the members' orders interleaved */);
/* Is `update` at N training points one persistent launch on this handle (size, variant bits)?  *yes = 0 / 1.  The host
 * layer sizes its concurrent evaluations with it (a persistent launch owns its compute units) instead of restating
 * the rule (reference: the fit loop of models/gpflow/models.py:256-321 is what calls `update` that often). */
int tgp_update_is_persistent(tgp_handle h, int64_t N, int* yes);

#ifdef __cplusplus
}
#endif
#endif /* TGP_H */
