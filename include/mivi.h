/*
 * mivi.h -- C ABI of libmivi: the MI355X-native (gfx950) RepGradELBO / ADVI hot path of
 * AdvancedVI.jl v0.7.0.  extern "C", plain pointers and sizes, no exceptions across the ABI.
 *
 * Every entry point cites the reference interface it replaces (paths relative to the
 * AdvancedVI.jl repository).  The Julia-side `ccall` binding is shown in INTEGRATION.md.
 *
 * Conventions
 *   - All functions return mivi_status_t (0 = ok).  mivi_last_error(ctx) gives text.
 *   - T is float (MIVI_F32) or double (MIVI_F64), fixed per context.
 *   - `*_dev` pointers are device (HBM) pointers valid on the context's device; calls taking
 *     only device pointers are ASYNCHRONOUS on the context's stream (no host sync).
 *     `*_host` convenience variants copy in/out and synchronise.
 *   - Sample layout: d x M column-major, one sample per column (src/utils.jl:6 `eachsample=eachcol`).
 *   - Parameter layout (what `Optimisers.destructure(q)` yields):
 *       mean-field: [location (d); diag(scale) (d)]                 src/families/location_scale.jl:39-43
 *       full-rank : [location (d); vec(scale) column-major (d*d)]   src/families/location_scale.jl:21
 *                   entries above the diagonal are ignored on input (LowerTriangular) and the
 *                   gradient written there is exactly 0.
 *   - The objective value is the NEGATIVE ELBO (src/algorithms/repgradelbo.jl:117,148).
 *   - Randomness is explicit: eps is a pure function of (seed, estimate_idx, global sample m, i)
 *     through Philox4x32-10 + Box-Muller (csrc/philox.h), replacing the mutable `rng` threaded
 *     through src/algorithms/repgradelbo.jl:104-110.  One estimate consumes one estimate_idx.
 *   - A context is single-owner (like the reference's rng); distinct contexts are independent.
 */
#ifndef MIVI_H
#define MIVI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIVI_VERSION_MAJOR 0
#define MIVI_VERSION_MINOR 1

typedef int32_t mivi_status_t;
enum {
  MIVI_OK = 0,
  MIVI_ERR_BAD_ARG = 1,
  MIVI_ERR_NONFINITE = 2,      /* objective not finite: host maps to the ErrorException of src/algorithms/common.jl:83-89 */
  MIVI_ERR_NONPOSITIVE_SCALE = 3, /* log of a non-positive scale diagonal: the DomainError ClipScale exists to prevent */
  MIVI_ERR_HIP = 4,
  MIVI_ERR_NO_TARGET = 5,
  MIVI_ERR_UNSUPPORTED = 6
};

typedef enum { MIVI_F32 = 0, MIVI_F64 = 1 } mivi_dtype_t;

/* MeanFieldGaussian / FullRankGaussian: src/families/location_scale.jl:139-141 / :124-128 */
typedef enum { MIVI_MEANFIELD = 0, MIVI_FULLRANK = 1 } mivi_family_t;

/* src/algorithms/entropy.jl: ClosedFormEntropy :25-29, ClosedFormEntropyZeroGradient :11-15,
 * MonteCarloEntropy :40-46, StickingTheLandingEntropy :57-65, StickingTheLandingEntropyZeroGradient :78-90 */
typedef enum {
  MIVI_ENT_CLOSED_FORM = 0,
  MIVI_ENT_CLOSED_FORM_ZERO_GRAD = 1,
  MIVI_ENT_MONTE_CARLO = 2,
  MIVI_ENT_STL = 3,
  MIVI_ENT_STL_ZERO_GRAD = 4
} mivi_entropy_t;

/* RepGradELBO(n_samples; entropy) + family + RNG key: src/algorithms/repgradelbo.jl:21-24,72-74 */
typedef struct {
  int32_t dtype;      /* mivi_dtype_t */
  int32_t family;     /* mivi_family_t */
  int32_t d;          /* LogDensityProblems.dimension(prob) */
  int32_t n_mc;       /* samples THIS context draws per estimate (RepGradELBO.n_samples / world size) */
  int32_t entropy;    /* mivi_entropy_t */
  int32_t device;     /* HIP device ordinal */
  uint64_t seed;      /* Philox key */
  int32_t m_offset;   /* first GLOBAL sample index owned by this context (multi-GPU shard); 0 on one GPU */
  int32_t m_total;    /* GLOBAL n_samples the estimate is normalised by; 0 => n_mc */
  void *stream;       /* hipStream_t to launch on; NULL = the HIP null (legacy default) stream */
  int32_t own_stream; /* != 0: ignore `stream` and create a non-blocking stream owned by the context */
  int32_t reserved;
} mivi_config_t;

typedef struct mivi_ctx mivi_ctx_t;

/* ---- lifetime ------------------------------------------------------------------------------- */
/* replaces AdvancedVI.init(rng, obj::RepGradELBO, adtype, q, prob, params, restructure)
 * (src/algorithms/repgradelbo.jl:41-70): one-time preparation, no AD to prepare. */
mivi_status_t mivi_create(const mivi_config_t *cfg, mivi_ctx_t **out);
mivi_status_t mivi_destroy(mivi_ctx_t *ctx);
const char *mivi_last_error(const mivi_ctx_t *ctx);
int32_t mivi_version(void);
mivi_status_t mivi_set_stream(mivi_ctx_t *ctx, void *hip_stream);
/* Wait for the context's stream.  Device entries never synchronise; a non-finite objective or a non-positive scale
 * diagonal is recorded in a sticky device flag which this call (and the _host entries / mivi_optimize_steps) reads and
 * clears, returning MIVI_ERR_NONFINITE / MIVI_ERR_NONPOSITIVE_SCALE -- the once-per-step isfinite check of
 * src/algorithms/common.jl:83-89. */
mivi_status_t mivi_synchronize(mivi_ctx_t *ctx);
/* length of `params` / gradient: 2d or d + d*d  (test/families/location_scale.jl:146-155) */
int64_t mivi_params_len(const mivi_ctx_t *ctx);
/* length of the shard-additive partials buffer ([sum_m W (d); sum_m W (x) eps; sum ell; sum 0.5|eps|^2]):
 * mean-field 2d + 2; full-rank d + d(d+1)/2 + 2 -- only the lower triangle travels, packed column by column
 * (entry (i, j), j <= i, at d + j*d - j(j-1)/2 + (i - j)) so the all-reduce moves half the bytes of the gradient. */
int64_t mivi_partials_len(const mivi_ctx_t *ctx);

/* ---- targets: the LogDensityProblems plugin seam --------------------------------------------- *
 * replaces LogDensityProblems.logdensity / logdensity_and_gradient / dimension as called from
 * src/algorithms/repgradelbo.jl:84-86 and src/mixedad_logdensity.jl:23-34.  Built-in targets run
 * fused on the device; the callback target is the generic plugin route (batched over columns). */

/* MvNormal(mean, Diagonal(std.^2)): test/models/normal.jl:56-75, bench/benchmarks.jl:43-47. host ptrs, T[d]. */
mivi_status_t mivi_set_target_diag_gauss(mivi_ctx_t *ctx, const void *mean_host, const void *std_host);
/* MvNormal(mean, L*L'): test/models/normal.jl:36-54.  host ptrs: mean T[d], L T[d*d] column-major lower. */
mivi_status_t mivi_set_target_dense_gauss(mivi_ctx_t *ctx, const void *mean_host, const void *chol_L_host);
/* Hierarchical logistic regression over theta = [beta (d-1); s]:
 *   variant 0: docs/src/tutorials/subsampling.md:26-38 (s = log sigma, Normal(0,3) prior on sigma, likeadj = n_data/n)
 *   variant 1: README.md:42-66 under the exp-bijector wrapper README.md:91-106 (LogNormal(0,3) prior, + log|det J| = s)
 * X: n x (d-1) COLUMN-major T (X[r + k*n], Julia's native Matrix layout), y: n bytes {0,1}.  x_on_device != 0 => X,y are device ptrs
 * (borrowed, must outlive the ctx); otherwise host ptrs, copied. */
mivi_status_t mivi_set_target_logreg(mivi_ctx_t *ctx, const void *X, const uint8_t *y, int64_t n,
                                     int32_t variant, double likeadj, int32_t x_on_device);
/* AdvancedVI.subsample(prob, batch) for the built-in logistic regression (docs/src/tutorials/subsampling.md:99-102,
 * src/algorithms/subsampledobjective.jl:85-87): the data set given to mivi_set_target_logreg stays resident, the target
 * becomes its rows idx_host[0..b) (0-based) with the likelihood scaled by `likeadj` (= n_data / b, subsampling.md:37).
 * b = 0 restores the full data set.  Synchronises the stream (the batch buffers are reused from step to step). */
mivi_status_t mivi_logreg_select_rows(mivi_ctx_t *ctx, const int64_t *idx_host, int64_t b, double likeadj);

/* Neal's funnel on the constrained scale + Stacked([log, identity]) bijector (README.md:76-82,102-106):
 * theta_1 = exp(eta_1) ~ LogNormal(0, sigma_v), theta_i ~ Normal(0, theta_1), log|det J| = eta_1. */
mivi_status_t mivi_set_target_funnel(mivi_ctx_t *ctx, double sigma_v);

/* The same funnel WITHOUT the built-in bijector: theta = [s; x] on the constrained scale (s > 0), log p = log LogNormal(s; 0,
 * sigma_v) + sum_i log Normal(x_i; 0, s).  Compose it with mivi_set_bijector_stacked({exp on [0,1), identity on [1,d)}) to obtain
 * the target of mivi_set_target_funnel (tests/test_gpu_bijector.py pins the two against each other). */
mivi_status_t mivi_set_target_funnel_constrained(mivi_ctx_t *ctx, double sigma_v);

/* Bijectors.Stacked over index blocks, wrapping WHATEVER target is set (README.md:76-82,91-119,
 * docs/src/tutorials/constrained.md:154-196: `TransformedLogDensityProblem(prob, binv)`):
 *   logdensity(eta) = log pi(binv(eta)) + logabsdetjac(binv, eta),  binv = identity or exp per block.
 * ranges_host[2 b], ranges_host[2 b + 1] = [begin, end) of block b (0-based, disjoint, inside [0, d)); kinds_host[b]: 0 identity,
 * 1 exp.  Coordinates in no block are identity.  n_blocks = 0 removes the bijector.  The variational family stays on the
 * unconstrained scale; the target (built-in or plugin callback) sees constrained samples binv(z), its gradient g comes back as
 * J' g + d logabsdetjac / d eta  (exp block: x_i g_i + 1), its value gains sum over exp coordinates of eta_i.
 * With a bijector the estimate runs on the explicit-sample route (Z materialised, transformed in place). */
mivi_status_t mivi_set_bijector_stacked(mivi_ctx_t *ctx, int32_t n_blocks, const int32_t *ranges_host, const int32_t *kinds_host);

/* Generic plugin: batched `logdensity_and_gradient` (src/mixedad_logdensity.jl:28) over the columns of Z.
 * Called on the host thread inside mivi_estimate_* with HOST buffers: Z (d x M col-major) in,
 * ell (M) and G (d x M) out.  Return non-zero to abort (-> MIVI_ERR_BAD_ARG). */
typedef int32_t (*mivi_logdensity_and_gradient_fn)(void *user, const void *Z_host, int32_t d, int32_t M,
                                                   void *ell_host, void *G_host);
/* value-only plugin (LogDensityProblems.logdensity), used by mivi_estimate_objective when set; may be NULL */
typedef int32_t (*mivi_logdensity_fn)(void *user, const void *Z_host, int32_t d, int32_t M, void *ell_host);
mivi_status_t mivi_set_target_callback(mivi_ctx_t *ctx, mivi_logdensity_and_gradient_fn fn_grad,
                                       mivi_logdensity_fn fn_value, void *user);

/* ---- the hot path ----------------------------------------------------------------------------- */
/* rand(rng, q, n_mc): src/families/location_scale.jl:71-87 (exposes the sample kernel for parity).
 * Z_dev: T[d*n_mc]; eps_dev: T[d*n_mc] or NULL. */
mivi_status_t mivi_sample(mivi_ctx_t *ctx, const void *params_dev, uint64_t estimate_idx,
                          void *Z_dev, void *eps_dev);

/* estimate_gradient!(rng, obj::RepGradELBO, adtype, out, state, params, restructure):
 * src/algorithms/repgradelbo.jl:151-177 (+ the AD shim src/AdvancedVI.jl:57-67 it replaces).
 * value_dev: T[1] <- -elbo; grad_dev: T[params_len] fully overwritten. Asynchronous for built-in
 * targets; synchronous (host round trip) for the callback target. */
mivi_status_t mivi_estimate_gradient(mivi_ctx_t *ctx, const void *params_dev, uint64_t estimate_idx,
                                     void *value_dev, void *grad_dev);
/* Same, host buffers, synchronous; additionally returns MIVI_ERR_NONFINITE when !isfinite(value)
 * (the check `step` performs at src/algorithms/common.jl:83-89). */
mivi_status_t mivi_estimate_gradient_host(mivi_ctx_t *ctx, const void *params_host, uint64_t estimate_idx,
                                          void *value_host, void *grad_host);
/* `count` consecutive estimates estimate_idx0 .. estimate_idx0+count-1 of the same params replayed as ONE
 * hipGraph launch; value/grad hold the LAST estimate on return.  Built-in targets only.  Mean-field family with the
 * diagonal-Gaussian target (rows independent): all `count` estimates run inside one launch-free kernel instead.  Full-rank f32 family,
 * d and n_mc multiples of 32 in [128, 2048] (multiples of 128 with the dense target or a sticking-the-landing estimator), Gaussian target:
 * the batch engine (kernels_fullrank_batch.hip) -- steps of up to 80 estimates as three launches (draws, one product, one VJP over all of
 * them), no graph. */
mivi_status_t mivi_estimate_gradient_n(mivi_ctx_t *ctx, const void *params_dev, uint64_t estimate_idx0,
                                       int32_t count, void *value_dev, void *grad_dev);
/* The same batch with EVERY estimate's result kept: values_dev T[count] <- -elbo of estimate estimate_idx0 + i;
 * grads_dev T[count * params_len] <- its gradient (row i), or NULL when only the values are wanted.  What a caller of
 * estimate_objective / estimate_gradient! at fixed parameters wants from many estimates -- monitoring with many samples
 * (test/algorithms/klminrepgraddescent.jl:36), a gradient averaged over several estimates
 * (src/algorithms/repgradelbo.jl:151-177 called `count` times on one q).  Estimate i equals mivi_estimate_gradient(estimate_idx0 + i): bitwise on the generic route, to
 * rounding (value 1e-6, gradient relative l2 2e-6) on the batch engine (full-rank f32, d and n_mc multiples of 32, Gaussian targets). */
mivi_status_t mivi_estimate_gradient_each(mivi_ctx_t *ctx, const void *params_dev, uint64_t estimate_idx0,
                                          int32_t count, void *values_dev, void *grads_dev);

/* estimate_objective(rng, obj::RepGradELBO, q, prob; n_samples): src/algorithms/repgradelbo.jl:112-118;
 * `entropy` override mirrors the algorithm-level wrapper src/algorithms/common.jl:29-38 (default there:
 * MonteCarloEntropy).  n_samples may differ from cfg.n_mc: sample m of the call is column m of estimate_idx's eps stream whatever route
 * runs it (chunks of 16384 samples; on batch-engine configurations whole blocks of n_mc samples as engine lanes).  value_dev: T[1]. */
mivi_status_t mivi_estimate_objective(mivi_ctx_t *ctx, const void *params_dev, uint64_t estimate_idx,
                                      int32_t n_samples, int32_t entropy, void *value_dev);
mivi_status_t mivi_estimate_objective_host(mivi_ctx_t *ctx, const void *params_host, uint64_t estimate_idx,
                                           int32_t n_samples, int32_t entropy, void *value_host);

/* gaussian_expectation_gradient_and_hessian!(rng, q, n_samples, grad_buf, hess_buf, prob), the first-order
 * (Stein / Price identity) branch: src/algorithms/gauss_expected_grad_hess.jl:20-60 -- the inner estimator of
 * KLMinWassFwdBwd / KLMinNaturalGradDescent / KLMinSqrtNaturalGradDescent (klminwassfwdbwd.jl:101,
 * klminnaturalgraddescent.jl:120, klminsqrtnaturalgraddescent.jl:104).  Full-rank family only (the reference method
 * takes a triangular scale).  With u = the eps stream of `estimate_idx` (d x n_samples) and z = C u + m:
 *   logpi_avg_dev T[1]    <- mean_b logpi(z_b)
 *   grad_dev      T[d]    <- mean_b grad logpi(z_b)
 *   hess_dev      T[d*d]  <- C' \ mean_b(u_b grad logpi(z_b)')      column-major, NOT symmetrised (as the reference)
 * n_samples <= 0 means cfg.n_mc; any n_samples is processed in chunks of 16384 columns.  Asynchronous for built-in
 * targets.  The second-order branch is mivi_gauss_expected_grad_hess2 below. */
mivi_status_t mivi_gauss_expected_grad_hess(mivi_ctx_t *ctx, const void *params_dev, uint64_t estimate_idx,
                                            int32_t n_samples, void *logpi_avg_dev, void *grad_dev, void *hess_dev);
mivi_status_t mivi_gauss_expected_grad_hess_host(mivi_ctx_t *ctx, const void *params_host, uint64_t estimate_idx,
                                                 int32_t n_samples, void *logpi_avg_host, void *grad_host, void *hess_host);

/* The second-order branch of the same function (src/algorithms/gauss_expected_grad_hess.jl:61-83; the reference's test runs both
 * capabilities, test/general/gauss_expected_grad_hess.jl:45-56): for a target with second-order capability
 * (LogDensityProblems.logdensity_gradient_and_hessian) the Hessian estimate is the SAMPLE AVERAGE of the Hessians at z_b = C u_b + m
 * (the same eps stream as the first-order branch), no Stein identity and no solve:
 *   logpi_avg <- mean_b logpi(z_b)     grad <- mean_b grad logpi(z_b)     hess <- mean_b hess logpi(z_b)   (d x d column-major)
 * Targets with a Hessian: the built-in diagonal / dense Gaussians (constant Hessians -1/sigma^2 / -P: written exactly); the built-in
 * logistic regression (both variants, also on a row selection) and the funnel (unconstrained and constrained) -- their Hessians are linear
 * in per-sample statistics, so the average is one weighted Gram matrix X' diag(mean_b pi (1 - pi)) X resp. an arrow matrix, f64 sums
 * (csrc/kernels_hess2.hip); or a plugin that registered a batched Hessian callback next to its order-1 callback -- Z (d x M) in; ell (M),
 * G (d x M) and Hsum (d x d, column-major) = the SUM over the M columns of hess logpi(z_m) out, host buffers, non-zero return aborts.
 * A Stacked bijector around the target: MIVI_ERR_UNSUPPORTED (use the first-order entry, as the reference does for order-1 problems). */
typedef int32_t (*mivi_logdensity_gradient_and_hessian_fn)(void *user, const void *Z_host, int32_t d, int32_t M,
                                                           void *ell_host, void *G_host, void *Hsum_host);
mivi_status_t mivi_set_target_hess_callback(mivi_ctx_t *ctx, mivi_logdensity_gradient_and_hessian_fn fn, void *user);
mivi_status_t mivi_gauss_expected_grad_hess2(mivi_ctx_t *ctx, const void *params_dev, uint64_t estimate_idx,
                                             int32_t n_samples, void *logpi_avg_dev, void *grad_dev, void *hess_dev);
mivi_status_t mivi_gauss_expected_grad_hess2_host(mivi_ctx_t *ctx, const void *params_host, uint64_t estimate_idx,
                                                  int32_t n_samples, void *logpi_avg_host, void *grad_host, void *hess_host);

/* ---- multi-GPU: shard the MC batch, all-reduce the partials, finalize -------------------------- *
 * No counterpart in the reference (single task).  partials_dev: T[partials_len] un-normalised sums over
 * this context's samples; the caller all-reduces (RCCL sum) and calls mivi_finalize on every rank. */
mivi_status_t mivi_estimate_partials(mivi_ctx_t *ctx, const void *params_dev, uint64_t estimate_idx,
                                     void *partials_dev);
mivi_status_t mivi_finalize(mivi_ctx_t *ctx, const void *params_dev, const void *partials_dev,
                            void *value_dev, void *grad_dev);

/* ---- next to the hot path: projection + optimiser step, device-resident ---------- */
/* ClipScale: scale[diagind] = max(scale[diagind], eps)   src/optimization/clip_scale.jl:18-29 */
mivi_status_t mivi_clip_scale(mivi_ctx_t *ctx, void *params_dev, double epsilon);
/* Optimisers.Descent(eta): params .-= eta .* grad  (the rule of test/algorithms/klminrepgraddescent.jl:110) */
mivi_status_t mivi_descent_update(mivi_ctx_t *ctx, void *params_dev, const void *grad_dev, double eta);
/* Optimisers.Adam(eta, (b1,b2), eps) (bench/benchmarks.jl:64): state_dev T[2*params_len] (m; v), t = step number >= 1 */
mivi_status_t mivi_adam_update(mivi_ctx_t *ctx, void *params_dev, const void *grad_dev, void *state_dev,
                               int64_t t, double eta, double beta1, double beta2, double eps);
/* COCOB(alpha), the "COCOB-Backprop" coin-betting rule (src/optimization/rules.jl:78-96): state_dev T[5*params_len] =
 * (L; G; R; theta; x1), initialised by the caller to (0; 0; 0; 0; params) as Optimisers.init does (:84-86).  Coordinates whose
 * gradient has been exactly zero so far (L = 0; the reference's expression is 0/0 there) are left unchanged. */
mivi_status_t mivi_cocob_update(mivi_ctx_t *ctx, void *params_dev, const void *grad_dev, void *state_dev, double alpha);
/* y <- a*x + b*y over n elements of T.  PolynomialAveraging: x_bar = (1-w) x_bar + w x, src/optimization/averaging.jl:40-47 */
mivi_status_t mivi_axpby(mivi_ctx_t *ctx, void *y_dev, double a, const void *x_dev, double b, int64_t n);
/* DoG (kind 0) / DoWG (kind 1): src/optimization/rules.jl:48-64 / :17-34.  state_dev holds x0 (T[params_len]) followed
 * by two doubles (v, r) at byte offset mivi_dog_state_bytes(ctx)-16.  init: x0 = params, v = 0, r = alpha*(1+norm(params)). */
int64_t mivi_dog_state_bytes(const mivi_ctx_t *ctx);
mivi_status_t mivi_dog_init(mivi_ctx_t *ctx, const void *params_dev, void *state_dev, double alpha);
mivi_status_t mivi_dog_update(mivi_ctx_t *ctx, void *params_dev, const void *grad_dev, void *state_dev, int32_t kind);
/* ProximalLocationScaleEntropy (src/optimization/proximal_location_scale_entropy.jl:44-61), the operator of
 * KLMinRepGradProxDescent (src/algorithms/constructors.jl:122-157): every scale-diagonal entry
 * c <- c + (sqrt(c^2 + 4 gamma) - c) / 2.  gamma = `stepsize` (Descent: eta) when dog_state_dev is NULL, otherwise it is read
 * on the device from the DoG / DoWG state (r / sqrt(v) resp. r^2 / sqrt(v), :26-42) -- no host round trip. */
mivi_status_t mivi_prox_scale_entropy(mivi_ctx_t *ctx, void *params_dev, double stepsize, const void *dog_state_dev,
                                      int32_t dog_kind);
/* `n_steps` iterations of src/algorithms/common.jl:69-104 {estimate_gradient!, update!, ClipScale} with params
 * resident in HBM; rule: 0 = Descent(eta), 1 = Adam(eta).  elbo_dev: T[n_steps] or NULL receives info.elbo (= -value) per
 * iteration.  Returns MIVI_ERR_NONFINITE if any objective was not finite.
 * What runs (first match; every launch-free form keeps parameters and optimiser state in registers for all n_steps):
 *   mean-field, diagonal-Gaussian target                      one launch-free kernel (rows are independent), bitwise the single calls
 *                                                             (sticking-the-landing estimators: a few entries one ulp apart -- the two
 *                                                             kernels' residual terms are contracted differently by the compiler)
 *   full-rank, d <= 32, d n_mc <= 768 (STL 512)              one workgroup for the whole loop (the reference's own benchmark grid), to rounding
 *   full-rank f32, n_mc <= 32, diagonal-Gaussian target,      row-separable launch-free kernel (workgroups own row pairs), to rounding
 *     closed-form / Monte-Carlo entropy, d <= 1126
 *   mean-field, fused funnel target                           one kernel with a per-step grid-wide exchange, bitwise the single calls
 *   logistic-regression target, (d - 1) n_mc <= 256, d <= 64,  one kernel for the whole loop (the reference README's own example: one workgroup;
 *     n (d - 1) n_mc <= 2^20                                  larger data sets: up to 64 that split the rows and exchange partial sums), to rounding
 *   otherwise                                                 one hipGraph of chained estimates (full-rank f32: update + ClipScale in the VJP epilogue)
 * MIVI_NO_FUSED_LOOP=1 forces the hipGraph everywhere (A/B). */
mivi_status_t mivi_optimize_steps(mivi_ctx_t *ctx, void *params_dev, void *opt_state_dev, uint64_t estimate_idx0,
                                  int64_t t0, int32_t n_steps, int32_t rule, double eta, double clip_epsilon,
                                  void *elbo_dev);

/* The general device-resident loop: `n_steps` iterations of `step` (src/algorithms/common.jl:69-104)
 *     estimate_gradient!  ->  Optimisers.update!  ->  operator  ->  averager
 * for every rule / operator / averager the reference's ParamSpaceSGD algorithms combine (constructors.jl:44-157):
 *   rule      0 Descent(eta) | 1 Adam(eta, beta1, beta2, adam_eps) | 2 DoG | 3 DoWG      (src/optimization/rules.jl:17-64)
 *             4 COCOB(alpha = eta)                                                         (src/optimization/rules.jl:66-96)
 *   op        0 IdentityOperator | 1 ClipScale(clip_epsilon) | 2 ProximalLocationScaleEntropy (step size from the rule)
 *   averager  0 NoAveraging | 1 PolynomialAveraging(avg_eta): x_bar <- (1-w_t) x_bar + w_t x, w_t = (eta+1)/(t+eta)
 * opt_state: Adam T[2 params_len] (zeros before the first step) | DoG/DoWG mivi_dog_state_bytes (after mivi_dog_init) |
 * COCOB T[5 params_len] = [L; G; R; theta; x1] (zeros, x1 = the initial parameters: rules.jl:84-86).  COCOB runs as the hipGraph of chained
 * estimates with its update kernel (ClipScale fused), not in the launch-free loops; ProximalLocationScaleEntropy has no step size for it.
 * avg_params: T[params_len] running average, in/out (any content when t0 = 0: w_1 = 1).  t0 = iterations already done
 * (warm start, src/optimize.jl:58-62).  Descent/Adam with Identity/ClipScale and no averaging take the fused paths of
 * mivi_optimize_steps; the other combinations are launch-free as well where mivi_optimize_steps is (mean-field + diagonal-Gaussian
 * target, d <= 4096 (f64: 2048) for DoG / DoWG; full-rank with n_mc <= 32 or d <= 32 and that target), with DoG / DoWG exchanging two norm
 * partials per workgroup and step.  Results are bitwise those of the step-by-step entries on the hipGraph route and for
 * Descent / Adam on the mean-field loop; DoG / DoWG in the launch-free loops to the rounding of the two f64 norm sums, the
 * full-rank launch-free loops to f32 rounding. */
typedef struct mivi_loop {
  int32_t rule, op, averager, n_steps;
  double eta, beta1, beta2, adam_eps;
  double clip_epsilon, avg_eta;
  void *opt_state_dev;
  void *avg_params_dev;
  uint64_t estimate_idx0;
  int64_t t0;
  void *elbo_dev;     /* T[n_steps] or NULL: info.elbo of every iteration */
} mivi_loop_t;
mivi_status_t mivi_optimize_loop(mivi_ctx_t *ctx, void *params_dev, const mivi_loop_t *loop);

/* Estimate-index source: when idx_dev != NULL every estimate uses estimate_idx + *idx_dev (read on the device at
 * kernel time), so a captured graph (e.g. a torch CUDAGraph holding kernels + the RCCL all-reduce) advances the
 * eps stream on replay by bumping one device word.  NULL restores by-value indices. */
mivi_status_t mivi_set_index_source(mivi_ctx_t *ctx, const uint64_t *idx_dev);

/* Tools / tests: which kernels evaluate the built-in logistic regression (f32): 0 = by problem size (default: the VALU
 * kernels below n*p*n_mc = 1.6e7, the matrix-core kernels above), 1 = matrix-core kernels, 2 = VALU kernels.  Results
 * agree to rounding; f64 always takes the VALU kernels. */
mivi_status_t mivi_set_logreg_route(mivi_ctx_t *ctx, int32_t route);

/* ---- sharded finalisation + the collective behind the ABI --------------------------------------------------
 * The Monte-Carlo mean of src/algorithms/repgradelbo.jl:84-86 shards over the sample axis; what crosses GPUs is the partial
 * vector of mivi_estimate_partials.  Instead of all-reduce + a finalisation replicated on every rank:
 *     reduce-scatter  ->  every rank finalises ITS slice (mivi_finalize_slice)  ->  all-gather  ->  unpack (mivi_unpack_final)
 * The slice of rank r is elements [r * n, (r + 1) * n) of the partial vector padded to n * world, n = mivi_slice_len(ctx, world).
 * The packed "final" vector has the layout of the partial vector: [d/dmu; d/dsigma or the column-packed lower triangle of d/dC;
 * value; status bits].  Both functions are plain kernels on the context's stream: a host that owns its collectives
 * (torch.distributed, MPI.jl, RCCL from Julia) calls them around its own reduce-scatter / all-gather. */
int64_t mivi_slice_len(const mivi_ctx_t *ctx, int32_t world);
mivi_status_t mivi_finalize_slice(mivi_ctx_t *ctx, const void *params_dev, const void *slice_sum_dev, int32_t rank, int32_t world,
                                  void *final_slice_dev);
mivi_status_t mivi_unpack_final(mivi_ctx_t *ctx, const void *packed_final_dev, void *value_dev, void *grad_dev);

/* The collective itself, for hosts without one (julia/MIVI.jl): RCCL opened at run time (dlopen of librccl.so, override with
 * MIVI_RCCL_LIB; a copy the process already loaded is reused).  mivi_comm_unique_id fills 128 bytes on rank 0 (ship them to the
 * other ranks any way you like), mivi_comm_init joins `world` ranks -- one process per GPU, the context created with n_mc = the
 * local share, m_offset = its first global column, m_total = n_mc * world.  mivi_estimate_gradient_dist is then estimate_gradient!
 * of the m_total-sample estimate on every rank: {partials kernels, ncclReduceScatter, slice finalise, ncclAllGather, unpack}, all
 * on the context's stream (graph-capturable).  world = 1 without a communicator runs the same kernels without the collectives;
 * world = 1 WITH an id runs them through RCCL (single-GPU test of the whole path).  Route: see mivi_comm_set_route. */
#define MIVI_COMM_ID_BYTES 128
mivi_status_t mivi_comm_unique_id(void *id_host);
mivi_status_t mivi_comm_init(mivi_ctx_t *ctx, const void *id_host, int32_t rank, int32_t world);
mivi_status_t mivi_comm_destroy(mivi_ctx_t *ctx);
mivi_status_t mivi_estimate_gradient_dist(mivi_ctx_t *ctx, const void *params_dev, uint64_t estimate_idx, void *value_dev, void *grad_dev);

/* The exchange written for xGMI (csrc/kernels_p2p.hip), no RCCL involved: every GPU of a node has a direct link to every other one, so
 * the sum of the partial vectors is ONE kernel per rank with two one-hop phases -- every rank stores slice s of its partial vector
 * straight into rank s's staging area; rank s sums the `world` contributions of its slice in rank order, finalises it (the owner of
 * the two scalars also assembles the objective value) and stores the packed final slice into every rank's final buffer; every rank
 * unpacks.  All ranks hold bit-identical results.  One fine-grained allocation per rank, mapped into its peers through HIP IPC:
 *   mivi_p2p_export   allocates this rank's exchange area and fills MIVI_P2P_HANDLE_BYTES describing it (world <= 8: one node);
 *   (the host gathers the `world` blobs in rank order by whatever channel it has: MPI, a file, torch.distributed, ...)
 *   mivi_p2p_attach   maps the peers' areas (ranks inside one process -- tests -- are mapped by pointer); the context then behaves as
 *                     rank `rank` of `world` for mivi_estimate_gradient_dist[_n] even without an RCCL communicator;
 *   mivi_comm_enable_p2p = export + gather through the RCCL communicator of mivi_comm_init + attach, for hosts with no other channel.
 * Every wait inside the kernel is bounded: a peer that never arrives makes the next mivi_synchronize return MIVI_ERR_HIP. */
#define MIVI_P2P_HANDLE_BYTES 256
mivi_status_t mivi_p2p_export(mivi_ctx_t *ctx, int32_t rank, int32_t world, void *handle_host);
mivi_status_t mivi_p2p_attach(mivi_ctx_t *ctx, const void *handles_host /* world x MIVI_P2P_HANDLE_BYTES, rank order */);
mivi_status_t mivi_p2p_detach(mivi_ctx_t *ctx);
/* host-only: out4 = {slice length n, chunk length cn, chunk workgroups G, rank that owns the two scalars} for a partial vector of
 * length L over `world` ranks (no GPU needed) */
void mivi_p2p_geometry(int64_t L, int32_t world, int64_t *out4);
mivi_status_t mivi_comm_enable_p2p(mivi_ctx_t *ctx);
/* One sharded estimate through the RCCL all-reduce route and through the peer-to-peer kernel, every rank collectively; both must agree to 1e-5
 * (value, gradient l2) on EVERY rank (ncclAllReduce(min) of the verdicts).  Only then -- and only with exchange areas of at least one other
 * device attached -- does the automatic route (mivi_comm_set_route 0) take the peer-to-peer kernel at world > 1; until then it is RCCL's
 * reduce-scatter -> mivi_finalize_slice -> all-gather.  rel_out3 (nullable) <- {value rel. difference, gradient rel. l2 difference, 1 if verified}.
 * The reference has no counterpart (single process: src/algorithms/repgradelbo.jl:84-86 is the mean this exchange sums).  A peer-to-peer
 * exchange that does not complete (bounded waits) detaches the areas and returns MIVI_ERR_HIP: the context stays on the RCCL routes. */
mivi_status_t mivi_p2p_selfcheck(mivi_ctx_t *ctx, const void *params_dev, uint64_t estimate_idx, double *rel_out3);
/* Batched sharded estimates on the peer-to-peer route run the exchange as ONE persistent kernel on its own stream BESIDE the compute chain
 * (csrc/kernels_p2p.hip), serving groups of four estimates per epoch.  on = 1 (default); 0 = no pipeline: batched calls run serial steps.
 * The pipeline needs the device to schedule the exchange kernel and the compute chain concurrently; where it does not (streams sharing one
 * hardware queue), the bounded waits end in MIVI_ERR_HIP at the next mivi_synchronize -- the host then switches the pipeline off ON EVERY
 * RANK (all ranks must use the same setting). */
mivi_status_t mivi_p2p_set_pipeline(mivi_ctx_t *ctx, int32_t on);
/* developer: the exchange's device words (per lane {epoch, ticket} at 16-word spacing, ready at word 64, freed[ring] at word 80): out128 = uint32[128] */
mivi_status_t mivi_p2p_debug_words(mivi_ctx_t *ctx, uint32_t *out128_host);
/* how many polls (about 1 us each) a wait inside the exchange may take before the peer counts as lost (default 2^21, about 2 s) */
mivi_status_t mivi_p2p_set_spin_budget(mivi_ctx_t *ctx, int32_t polls);
/* Which exchange mivi_estimate_gradient_dist[_n] uses: 0 = automatic (peer-to-peer when attached, otherwise ONE ncclAllReduce + the whole
 * finalisation on every rank below 16 MB of partials and ncclReduceScatter -> slice finalisation -> ncclAllGather -> unpack above),
 * 1 = ncclAllReduce, 2 = ncclReduceScatter / ncclAllGather, 3 = peer-to-peer.  mivi_comm_route reports the route in force. */
mivi_status_t mivi_comm_set_route(mivi_ctx_t *ctx, int32_t route);
int32_t mivi_comm_route(const mivi_ctx_t *ctx);
/* `count` consecutive sharded estimates of the same params (estimates at fixed parameters are independent: the multi-GPU form of
 * mivi_estimate_gradient_n): the exchange + finalisation of estimate t runs on a second stream UNDER the partial kernels of estimate
 * t + 1 (partial vectors double-buffered), one hipGraph per batch; value / grad hold the last estimate on return.  Every rank calls it
 * with the same arguments.  On a batch-engine shape (full-rank f32, d and n_mc multiples of 128 up to 2048, Gaussian target) and routes 1 / 2
 * -- or one rank without peer-to-peer areas -- the batch runs on the batch engine: up to 80 estimates (24 across ranks) per step as one
 * draw / product / VJP launch each, ONE ncclAllReduce per step over all of the step's partial vectors, one finalisation launch; the last
 * estimate then equals `count` single sharded estimates' to rounding (1e-6 value, 2e-6 gradient l2), not bit for bit. */
mivi_status_t mivi_estimate_gradient_dist_n(mivi_ctx_t *ctx, const void *params_dev, uint64_t estimate_idx0, int32_t count,
                                            void *value_dev, void *grad_dev);
/* Tests: the phases of the peer-to-peer exchange as separate launches (bit 0 push, bit 1 reduce + finalise, bit 2 unpack), so that
 * several ranks living in ONE process can be sequenced from one host thread.  partials_dev: the rank's partial vector, zero padded. */
mivi_status_t mivi_p2p_exchange(mivi_ctx_t *ctx, const void *params_dev, const void *partials_dev, void *value_dev, void *grad_dev,
                                int32_t phases);
/* Diagnostics of the peer-to-peer exchange since the last reset, one line per rank makes a first multi-GPU run readable: out6[0..2] <- microseconds
 * the exchange kernel (its workgroup 0) waited for {the compute chain's hand-over, the peers' pushes, the owners' reduced chunks}, out6[3] <- groups
 * of estimates served, out6[4] <- payload bytes stored into EACH peer per estimate, out6[5] <- slice length (elements).  reset != 0 zeroes them. */
mivi_status_t mivi_p2p_stats(mivi_ctx_t *ctx, double *out6, int32_t reset);
/* Tests: the partial kernels of one estimate with DIRECT staging -- every entry of the rank's partial vector is stored straight into its
 * owner's staging area (no ring slot, no push pass; what mivi_estimate_gradient_dist[_n] do on the peer-to-peer route for the full-rank f32
 * family); follow with mivi_p2p_exchange(ctx, params, NULL, value, grad, phases).  MIVI_ERR_UNSUPPORTED for other configurations. */
mivi_status_t mivi_p2p_partials_direct(mivi_ctx_t *ctx, const void *params_dev, uint64_t estimate_idx);
/* Measurement (bench.py --gpus N): us per estimate of {partial kernels, exchange + finalisation, the serial step, the pipelined step},
 * hipEvents around one hipGraph replay of `reps` estimates each; every rank calls it collectively.  us_host: double[4]. */
mivi_status_t mivi_profile_dist(mivi_ctx_t *ctx, const void *params_dev, int32_t reps, double *us_host);

/* Measurement hook of the batch engine (kernels_fullrank_batch.hip: what mivi_estimate_gradient_n / _each run for the full-rank family with the
 * diagonal- or dense-Gaussian target): `reps` launches of each of a step's kernels for `lanes` estimates, hipEvents on the context's stream.
 * us_out[0..4] (double[5]) <- average launch duration in microseconds of {draws, product (+ the fused diagonal target), VJP + values, the dense
 * target's product (0 with the diagonal target), the sticking-the-landing product (0 with the other estimators)}.  bench.py's roofline leg. */
mivi_status_t mivi_profile_batch(mivi_ctx_t *ctx, const void *params_dev, int32_t lanes, int32_t reps, double *us_out);
/* Estimates per launch ("lanes" of a step) the batch engine uses for a `count`-estimate call: equal steps of at most 80 lanes. */
int32_t mivi_batch_lanes(const mivi_ctx_t *ctx, int32_t count);
/* What a batch of estimates at this context's configuration runs on (measurement / test hook; the reference has no counterpart: every
 * `estimate_gradient!` there is one AD call, src/algorithms/repgradelbo.jl:151-177).  what = 0: 1 when mivi_estimate_gradient_n / _each with
 * these (16-byte aligned) device parameters take the batch engine, 0 otherwise; 1: matrix-pipe products per 32 x 32 x 16 block of the engine's
 * split-operand contractions (3: f16 hi / lo planes); 2: bytes per operand-plane element (4); 3: 1 once a launch-free optimisation loop's grid-wide
 * exchange was lost on this context (mivi_optimize_loop then restored the caller's state, re-ran the steps on the graph of launches and stays there). */
int32_t mivi_batch_info(const mivi_ctx_t *ctx, const void *params_dev, int32_t what);

/* ---- measurement hook (bench.py roofline leg) --------------------------------------------------------- *
 * Times `reps` back-to-back launches of ONE stage of the estimate with hipEvents recorded on the context's
 * stream (after one full warm estimate so every input buffer is populated).
 *   which: 0 = whole estimate, 1 = eps generation, 2 = sample(+fused target) kernel (mean-field: the fused main kernel),
 *          3 = VJP kernel, 4 = dense-target kernel, 5 = the launch-free loop of 100 estimates (mean-field + diagonal
 *          target; what mivi_estimate_gradient_n runs there), 6 / 7 = the split-K product / its reduce kernel alone
 *          (removed), 8 = the sticking-the-landing term W += C^-T eps alone (full-rank, STL estimators),
 *          9 = the LATENCY FLOOR of the full-rank estimate: two empty dependent launches with the grid / block / LDS footprint of the product and VJP kernels,
 *          10 / 11 = the product / VJP launch of FOUR lane-batched estimates (what mivi_estimate_gradient_n issues at the
 *          BASELINE sizes: ms_per_launch is then the time of four estimates' stage).  Stages 1-4, 6, 7, 9-11 are captured `reps` times into one hipGraph and the
 *          replay is timed (eager launches of 2-5 us kernels are host-bound); 0 and 5 are eager.  ms_per_launch_host: double[1]. */
mivi_status_t mivi_profile_kernel(mivi_ctx_t *ctx, int32_t which, const void *params_dev, int32_t reps,
                                  double *ms_per_launch_host);

/* Which kernels the full-rank f32 path runs for `n_samples` per launch on this context (measurement / documentation hook):
 *   bits 0-1: 0 = first generation (32x32 tiles fed from L2, any shape), 1 = unsplit 32x32 product with fused target
 *             (k_fr_prod32) + private-wave VJP (k_fr_vjp32), 2 = split-K LDS-staged product + reduce (k_fr_gemm, k_fr_reduce)
 *             + k_fr_vjp32, 3 = unsplit 64x64 product with fused target (k_fr_prod64, shapes with at least as many tiles as CUs)
 *             + k_fr_vjp32 / k_fr_vjp64;   bit 4: products on the bf16 matrix cores with the exact three-way operand split. */
int32_t mivi_fullrank_route(const mivi_ctx_t *ctx, int32_t n_samples);
/* Which kernels the native logistic-regression target's two data contractions run for `n_samples` per launch (same kind of hook):
 *   0 = vector-ALU kernels (small problems, f64); bit 0: matrix cores (two-way f16 splits); bit 1: the logits on prebuilt operand planes of X
 *   (k_lr_logits_planes); bit 2: X^T R on planes too (k_lr_xtr_planes, residual planes straight from the logits kernel). */
int32_t mivi_logreg_kernels(const mivi_ctx_t *ctx, int32_t n_samples);

/* Developer tool: when buf_dev != NULL every workgroup of the main kernels records wall_clock64() stamps
 * (100 MHz) at buf_dev[block*8 + phase].  NULL switches it off.  Not part of the drop-in surface. */
mivi_status_t mivi_debug_timeline(mivi_ctx_t *ctx, void *buf_dev);

/* ---- host-side RNG restatement (no GPU needed; used by the parity tests) ------------------------- */
void mivi_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* raw words of the eps stream for element (i, global m): out[count] for i = i0 .. i0+count-1 */
void mivi_eps_bits_host(uint64_t seed, uint64_t estimate_idx, int32_t d, int64_t m, int32_t i0, int32_t count,
                        uint32_t *out);
/* eps values themselves evaluated on the host with the same formulas: out_f64[count] */
void mivi_eps_host(uint64_t seed, uint64_t estimate_idx, int32_t d, int64_t m, int32_t i0, int32_t count,
                   int32_t dtype, double *out_f64);

#ifdef __cplusplus
}
#endif
#endif /* MIVI_H */
