/*
 * gpbayes.h — C ABI of the MI355X-native GP-emulator + log-posterior engine.
 *
 * The reference (Hendrik1704/GPBayesTools-HIC) is pure Python and has no FFI of its
 * own; the hot path sits behind three duck-typed Python seams.  This
 * header is the C boundary a maintainer binds with ctypes; each
 * entry point names the reference interface it replaces.
 *
 * Conventions
 *   - plain C, no C++/torch types; all matrices row-major float64; sizes int64_t.
 *   - pointers flagged "host" are caller-owned host memory; pointers flagged "dev"
 *     are caller-owned device (HBM) memory on the context's device.  Functions with
 *     an `on_device` argument accept either.
 *   - every function returns int: 0 = ok, >0 = LAPACK-style info (1-based index of the
 *     first non-positive pivot), <0 = GPB_E_* error; gpb_last_error() gives the text.
 *   - work is enqueued on the context's HIP stream; functions that return host data
 *     synchronise that stream themselves, device-output functions do not
 *     (call gpb_sync or use stream order).
 *   - one context per (process, device, emulator); not thread-safe per context.
 *   - there is NO CPU fallback: if no gfx950 device is present gpb_ctx_create fails.
 *   - device buffers a context releases are kept for the next context of about that size (up to GPB_POOL_MB megabytes per
 *     process, default 8192, 0 = off): on this runtime hipFree + hipMalloc of a large buffer is a driver round trip of 60-100 ms.
 */
#ifndef GPBAYES_H
#define GPBAYES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPB_VERSION 110

/* The library is built with -fvisibility=hidden: these entry points are all it exports. */
#define GPB_API __attribute__((visibility("default")))

/* kernel_id  — sklearn kernels the reference instantiates (src/emulator.py:286-306) */
#define GPB_KERNEL_RBF      0   /* 1.*RBF(l) + White          sk:kernels.py:1525-1575 */
#define GPB_KERNEL_MATERN15 1   /* 1.*Matern(l, nu=1.5)+White sk:kernels.py:1721-1723 */
#define GPB_KERNEL_MATERN25 2   /* nu=2.5 (BASELINE cfg 5)    sk:kernels.py:1724-1726 */

/* observable-transform modes of Emulator.predict (src/emulator.py:558-601) */
#define GPB_MODE_PCA            0
#define GPB_MODE_NO_PCA         1  /* perform_no_PCA=True            :562-565,589-592 */
#define GPB_MODE_EXPDIAG        2  /* exp_and_cov_diagonal=True      :567-568,594-601 */
#define GPB_MODE_NO_PCA_EXPDIAG 3

/* errors */
#define GPB_E_ARG     (-1)
#define GPB_E_STATE   (-2)
#define GPB_E_HIP     (-3)
#define GPB_E_NODEV   (-4)
#define GPB_E_ALLOC   (-5)
#define GPB_E_RCCL    (-6)

/* gpb_gp_get selectors */
#define GPB_GET_K      0  /* [P,N,N] K(X,X)+(noise+alpha)I: blocks of the LOWER 64-block triangle as built (call before gpb_gp_factor, which overwrites them) */
#define GPB_GET_L      1  /* [P,N,N] lower Cholesky factor, upper zeroed  == GPR.L_     */
#define GPB_GET_LINV   2  /* [P,N,N] L^-1 (lower)                                       */
#define GPB_GET_ALPHA  3  /* [P,N]   K^-1 z                               == GPR.alpha_ */
#define GPB_GET_KSTAR  4  /* [P,W,N] K(X*,X) of the most recent predict / likelihood batch of W rows == kernel_(X*, X_train_)  sk:_gpr.py:443 */
#define GPB_GET_FORM   5  /* [P]     distance form each GP's kernel matrices are built in, chosen from theta alone: 0 = Gram form
                           *         |a|^2+|b|^2-2a.b on the centred design, 1 = sklearn's difference form (length scales far below the
                           *         design's extent: sum_k (extent_k/l_k)^2 > 1024, e.g. the Matern lower search bound, src/emulator.py:292-297) */

typedef struct gpb_ctx gpb_ctx;

/* ---- lifetime -------------------------------------------------------------------- */
GPB_API int  gpb_version(void);
GPB_API int  gpb_device_count(void);
/* stream: a hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream), or
 * NULL to create a private one. */
GPB_API int  gpb_ctx_create(int device, void* stream, gpb_ctx** out);
GPB_API int  gpb_ctx_destroy(gpb_ctx* ctx);
/* Re-target the context onto a caller stream; NULL selects the legacy default stream (what
 * torch.cuda.current_stream() is unless the caller changed it). */
GPB_API int  gpb_ctx_set_stream(gpb_ctx* ctx, void* stream);
GPB_API int  gpb_sync(gpb_ctx* ctx);
/* Hand every cached device buffer back to the driver (the cache of the header comment: GPB_POOL_MB megabytes per process, default
 * 8192, 0 = off).  The cache is invisible to other allocators of the process (torch's caching allocator, the caller's own
 * hipMalloc): call this before a large allocation elsewhere.  The library does it itself when one of its own allocations fails.
 * No reference counterpart (memory management of the native side). */
GPB_API int  gpb_pool_trim(void);
GPB_API const char* gpb_last_error(gpb_ctx* ctx);
GPB_API void* gpb_stream(gpb_ctx* ctx);

/* ---- GP state: replaces sklearn GaussianProcessRegressor state ------------------- *
 * gpb_gp_set        <- GPR(kernel, alpha).fit(X, z) inputs          src/emulator.py:309-315
 * gpb_gp_set_theta  <- kernel_.theta                                sk:_gpr.py:332
 * gpb_gp_factor     <- K=kernel_(X); K_ii+=alpha; L_=cholesky(K); alpha_=cho_solve   sk:_gpr.py:346-364
 * gpb_gp_lml        <- GPR.log_marginal_likelihood(theta, eval_gradient)             sk:_gpr.py:537-652
 * gpb_gp_predict    <- GPR.predict(X, return_cov=True) + .diagonal()   sk:_gpr.py:441-469, src/emulator.py:553,573-575
 */
GPB_API int gpb_gp_set(gpb_ctx* ctx, int64_t N, int64_t d, int64_t P,
               const double* X_host /*[N,d]*/, const double* Z_host /*[P,N]*/,
               int kernel_id, double alpha);
/* gpb_gp_set_multi: P GPs, each over ITS OWN design — the GPs of several emulators of a chain (the reference fits dataset
 * after dataset and GP after GP: examples/EmulatorTraining.ipynb:124-138, src/emulator.py:309-315) or the 1 + n_restarts
 * starts of every GP's hyper-parameter search (sk:_gpr.py:318-337) side by side in one batch.  All designs must pad to the
 * same multiple of 64 points.  Such a context is fit-only: gpb_gp_set_theta, gpb_gp_factor, gpb_gp_get, gpb_gp_lml and
 * gpb_gp_lml_subset work on it, the predict / likelihood entry points return GPB_E_STATE.
 * gpb_gp_lml_subset: log-marginal likelihood (+ gradient) of n of the stored GPs — the searches still running — at
 * theta[n, d+2]; a GP's values do not depend on which other GPs share the call (bit for bit).  Works on any context; leaves
 * it without a factorisation (gpb_gp_set_theta + gpb_gp_factor afterwards). */
GPB_API int gpb_gp_set_multi(gpb_ctx* ctx, int64_t P, int64_t d, const int64_t* N_host /*[P]*/,
                     const double* const* X_host /*[P] pointers to [N_p,d]*/,
                     const double* const* Z_host /*[P] pointers to [N_p]*/, int kernel_id, double alpha);
GPB_API int gpb_gp_lml_subset(gpb_ctx* ctx, int64_t n, const int32_t* gp_index /*[n]*/, const double* theta_host /*[n,d+2]*/,
                      double* lml_host /*[n]*/, double* grad_host /*[n,d+2] or NULL*/, int* info_host /*[n] or NULL*/);
GPB_API int gpb_gp_set_theta(gpb_ctx* ctx, const double* theta_host /*[P,d+2]*/);
GPB_API int gpb_gp_factor(gpb_ctx* ctx, int* info_host /*[P], may be NULL*/);
GPB_API int gpb_gp_get(gpb_ctx* ctx, int what, double* out_host);
/* Evaluates at theta_host (state of the context's factorisation is overwritten; call
 * gpb_gp_set_theta+gpb_gp_factor afterwards to restore).  grad_host may be NULL.
 * Non-PD K for GP p: lml[p] = -inf, grad[p,:] = 0, info[p] > 0 (sk:_gpr.py:588-589). */
GPB_API int gpb_gp_lml(gpb_ctx* ctx, const double* theta_host /*[P,d+2]*/,
               double* lml_host /*[P]*/, double* grad_host /*[P,d+2] or NULL*/,
               int* info_host /*[P] or NULL*/);
/* mean/var are [W,P] (reference layout of the concatenated per-GP outputs). */
GPB_API int gpb_gp_predict(gpb_ctx* ctx, const double* Xs, int64_t W, int on_device,
                   double* mean /*[W,P]*/, double* var /*[W,P] or NULL*/);

/* Full predictive covariance between the W query points, per GP (what GPR.predict(return_cov=True) returns
 * and GPR.sample_y draws from: sk:_gpr.py:441-469, 498-540; src/emulator.py:608-633).  cov is [P,W,W].
 * Small batches only (W <= 8192): the MCMC path never forms it. */
GPB_API int gpb_gp_predict_cov(gpb_ctx* ctx, const double* Xs, int64_t W, int on_device,
                       double* mean /*[W,P]*/, double* cov /*[P,W,W]*/);

/* ---- emulator transform: replaces Emulator.predict after the per-GP calls -------- *
 * gpb_emu_set_transform <- _trans_matrix[:npc], scaler.mean_, _cov_trunc, scaler.scale_  src/emulator.py:335-363
 * gpb_emu_predict       <- Emulator.predict(X, return_cov, extra_std)                   src/emulator.py:465-605
 */
GPB_API int gpb_emu_set_transform(gpb_ctx* ctx, int mode, int64_t M,
                          const double* A_host /*[P,M] or NULL (no-PCA)*/,
                          const double* mu_host /*[M]*/,
                          const double* cov_trunc_host /*[M,M] or NULL*/,
                          const double* scale_host /*[M] or NULL (PCA)*/);
GPB_API int gpb_emu_predict(gpb_ctx* ctx, const double* Xs, int64_t W, int on_device,
                    const double* extra_std /*[W] or NULL (=0), same memory space as Xs*/,
                    double* mean /*[W,M]*/, double* cov /*[W,M,M] or NULL*/);

/* ---- likelihood block: replaces Chain._predict + mvn_loglike for ONE emulator ---- *
 * gpb_like_set   <- expdata[i0:i0+M], expdata_cov[i0:i0+M, i0:i0+M]    src/mcmc.py:139,302-324
 * gpb_loglike    <- -1/2 dY^T C^-1 dY - sum log diag chol(C), C = cov_model + cov_exp
 *                   for this emulator's diagonal block                 src/mcmc.py:23-65,153-166,288-293
 * The reference's covariance is block-diagonal over emulators (src/mcmc.py:163-164) and
 * the experimental covariance is diagonal (src/mcmc.py:320-322), so the multivariate
 * normal factorises: log-likelihood = sum over emulators of gpb_loglike blocks.
 * Rows whose block is not positive definite get NaN (the reference yields garbage there,
 * src/mcmc.py:44-54); *n_notpd_host counts them.
 */
GPB_API int gpb_like_set(gpb_ctx* ctx, const double* yexp_host /*[M]*/, const double* cov_exp_host /*[M,M]*/);
GPB_API int gpb_loglike(gpb_ctx* ctx, const double* Xs, int64_t W, int on_device,
                double* ll /*[W], same memory space as Xs*/, int accumulate,
                int* n_notpd_host /*may be NULL; forces a sync when non-NULL*/);

/* gpb_logpost <- Chain.log_posterior / log_likelihood for the LAST (or only) emulator of a chain, all on the
 *                device and asynchronous: block log-likelihood (added onto ll when accumulate != 0), then
 *                inside = all(lo < x < hi) strictly; ll = inside ? ll + inside_const : outside_value
 *                                                                       src/mcmc.py:188-222, 261-299 */
GPB_API int gpb_logpost(gpb_ctx* ctx, const double* Xs_dev /*[W,d]*/, int64_t W, double* ll_dev /*[W]*/, int accumulate,
                const double* lo_dev /*[d]*/, const double* hi_dev /*[d]*/, double outside_value,
                double inside_const);

/* gpb_mvn_loglike <- map(mvn_loglike, dY, cov): generic batched form on caller-provided
 *                    dY[W,M], cov[W,M,M] (any covariance, e.g. from foreign emulators)   src/mcmc.py:23-65,293 */
GPB_API int gpb_mvn_loglike(gpb_ctx* ctx, const double* dY, const double* cov, int64_t W, int64_t M, int on_device,
                    double* ll /*[W]*/, int* n_notpd_host /*may be NULL*/);

/* ---- chain-level helpers (device, for resident sampling loops) ------------------- *
 * gpb_box_finish <- inside=all(min<X<max) (strict); lp[~inside]=-inf|-1e300;
 *                   lp[inside] = ll + const                             src/mcmc.py:194-198,220-221,275-276,296-297
 */
GPB_API int gpb_box_finish(gpb_ctx* ctx, const double* X_dev /*[W,d]*/, int64_t W, int64_t d,
                   const double* lo_dev, const double* hi_dev, double outside_value,
                   double inside_const, double* ll_inout_dev /*[W]*/);

/* ---- parameterTrafoPCA input map (device pre-pass) -------------------------------- *
 * gpb_param_map_set <- the fitted scalers / PCAs of the three parameter groups          src/emulator.py:79-241
 * gpb_param_map     <- the per-row mapping X[W,d_in] -> GP input [W,d_out] that
 *                      Emulator.predict performs with Python loops before the GP calls  src/emulator.py:492-551
 * col_src[j] >= 0: output column j is original column col_src[j]; col_src[j] = -1 - (g*maxpc + c): principal
 * component c of group g.  group_desc[g] = {fn, col0, col1, col2, col3 (-1 = unused), npc}, fn 0 = zeta/s(T)
 * (:102-108), 1 = eta/s(mu_B) (:111-117), 2 = y_loss(y_init) (:120-126).  tables[g] = grid[100] | scaler mean[100]
 * | scaler scale[100] | PCA mean[100] | components[maxpc][100]. */
GPB_API int gpb_param_map_set(gpb_ctx* ctx, int64_t d_in, int64_t d_out, const int32_t* col_src /*[d_out]*/,
                      int32_t n_groups, const int32_t* group_desc /*[G][6]*/,
                      const double* tables /*[G][4+maxpc][100]*/, int32_t maxpc);
GPB_API int gpb_param_map(gpb_ctx* ctx, const double* X_dev /*[W,d_in]*/, int64_t W, double* out_dev /*[W,d_out]*/);

/* ---- emcee-equivalent stretch move (device resident) ------------------------------ *
 * Replaces emcee.EnsembleSampler.sample as driven by LoggingEnsembleSampler.run_mcmc
 * (src/mcmc.py:68-92,372-412): red/blue stretch move, a=2, counter-based Philox RNG
 * replicated on every rank.
 * gpb_stretch_propose: for the walkers of half `half` draw the complementary walker and z, write
 *   proposals q[nhalf,d] and the (d-1) ln z factor[nhalf].  Walker k of the half is index pi(2k+half):
 *   pi = identity when randomize_split = 0 (emcee inds = arange(n) % 2), else a per-step keyed
 *   pseudo-random permutation (emcee's default randomize_split=True), evaluated inline on every rank.
 * gpb_stretch_accept: accept where (d-1)*ln z + lp' - lp > ln u; updates pos, lp, naccept.
 */
GPB_API int gpb_stretch_propose(gpb_ctx* ctx, const double* pos_dev /*[nw,d]*/, int64_t nwalkers, int64_t d,
                        int half, uint64_t seed, uint64_t step, double a,
                        double* q_dev /*[nw/2,d]*/, double* factor_dev /*[nw/2]*/, int randomize_split);
GPB_API int gpb_stretch_accept(gpb_ctx* ctx, double* pos_dev, double* lp_dev /*[nw]*/, int64_t nwalkers, int64_t d,
                       int half, uint64_t seed, uint64_t step,
                       const double* q_dev, const double* factor_dev, const double* lpq_dev /*[nw/2]*/,
                       int64_t* naccept_dev /*[nw]*/, int randomize_split);
/* NaN log-probabilities the accept step has seen since the last reset (emcee raises "Probability function returned
 * NaN" when one occurs, emcee/ensemble.py; on the device such a proposal is rejected and counted).  Synchronises. */
GPB_API int gpb_stretch_nan_count(gpb_ctx* ctx, int64_t* count_host, int reset);

/* gpb_emcee_run <- the loop emcee.EnsembleSampler.sample runs under LoggingEnsembleSampler.run_mcmc
 *                  (src/mcmc.py:68-92, 372-412) with Chain.log_posterior (src/mcmc.py:261-299) as the log-probability,
 * for a chain whose observables come from THIS context's emulator alone: nsteps stretch-move steps (two half-ensemble
 * updates each: propose -> GP predict -> block log-likelihood + prior box -> accept), enqueued back to back on the
 * context's stream with no host involvement per step.  pos/lp hold the ensemble and its log-probabilities on entry
 * and on exit; chain_dev [nsteps, nw, d] / lpchain_dev [nsteps, nw] (either may be NULL) receive the state after every
 * step; steps are numbered step0, step0 + 1, ... in the counter-based generator.  With a communicator installed
 * (gpb_dist_init) every rank evaluates its rows of each half-ensemble batch and one in-stream all-gather per batch
 * completes the vector; nwalkers / 2 must then divide evenly over the ranks.  Asynchronous. */
GPB_API int gpb_emcee_run(gpb_ctx* ctx, double* pos_dev /*[nw,d]*/, double* lp_dev /*[nw]*/, int64_t nwalkers, int64_t nsteps,
                  uint64_t seed, uint64_t step0, double a, int randomize_split,
                  const double* lo_dev /*[d]*/, const double* hi_dev /*[d]*/, double outside_value, double inside_const,
                  double* chain_dev, double* lpchain_dev, int64_t* naccept_dev /*[nw]*/);

/* Chains of several emulators (Chain.emuList, src/mcmc.py:139-166): the covariance is block-diagonal over the emulators,
 * the log-likelihood the sum of their blocks.  ctxs[0..E) are the emulators' contexts in emuList order, all on one device
 * and stream, each with its likelihood block installed (gpb_like_set) and, for parameterTrafoPCA emulators, its
 * parameter map (gpb_param_map_set); rows are in the chain's ORIGINAL parameters [W, ndim].
 * gpb_chain_logpost   <- Chain.log_posterior / log_likelihood for the whole chain, rows inside the box only.
 * gpb_chain_emcee_run <- gpb_emcee_run for such a chain (a communicator, if any, is taken from ctxs[0]).
 * Both need the block likelihood kernels for every emulator (PCA modes with M <= 64 or npc <= 16); GPB_E_STATE
 * otherwise (the caller then sequences gpb_loglike / gpb_box_finish itself).
 * gpb_chain_supported: 1 when the two calls would accept these contexts as they stand, 0 when not, < 0 on bad arguments. */
GPB_API int gpb_chain_supported(gpb_ctx* const* ctxs, int E);
GPB_API int gpb_chain_logpost(gpb_ctx* const* ctxs, int E, const double* Xs_dev /*[W,ndim]*/, int64_t W, double* ll_dev /*[W]*/,
                      const double* lo_dev, const double* hi_dev, double outside_value, double inside_const);
GPB_API int gpb_chain_emcee_run(gpb_ctx* const* ctxs, int E, double* pos_dev, double* lp_dev, int64_t nwalkers, int64_t nsteps,
                        uint64_t seed, uint64_t step0, double a, int randomize_split,
                        const double* lo_dev, const double* hi_dev, double outside_value, double inside_const,
                        double* chain_dev, double* lpchain_dev, int64_t* naccept_dev);
/* gpb_chain_emcee_prepare: everything of gpb_chain_emcee_run that can fail on one rank alone — argument and state checks,
 * workspace allocation — and nothing that is enqueued.  A sharded caller runs it on every rank and lets the ranks agree on
 * the outcome (an all-reduce of the return codes) BEFORE any rank calls gpb_chain_emcee_run: a rank that failed there
 * would leave the others waiting inside the in-stream all-gather. */
GPB_API int gpb_chain_emcee_prepare(gpb_ctx* const* ctxs, int E, int64_t nwalkers);

/* ---- walker sharding over RCCL (one process per GPU) ------------------------------ *
 * gpb_dist_uid: rank 0 obtains a 128-byte ncclUniqueId to broadcast out of band.
 * gpb_dist_init / gpb_dist_allgather: in-stream ncclAllGather of per-walker
 * log-posteriors (count doubles per rank) — the one exchange per log-prob batch.
 */
/* gpb_dist_available: 1 when librccl loads with the four entry points used here (no communicator is created): ranks
 * vote on it before gpb_dist_init, whose ncclCommInitRank is itself collective. */
GPB_API int gpb_dist_available(void);
GPB_API int gpb_dist_uid(void* uid128_host);
GPB_API int gpb_dist_init(gpb_ctx* ctx, int rank, int nranks, const void* uid128_host);
GPB_API int gpb_dist_allgather(gpb_ctx* ctx, const double* send_dev, double* recv_dev, int64_t count);
GPB_API int gpb_dist_finalize(gpb_ctx* ctx);

/* ---- options and measurement ------------------------------------------------------- *
 * gpb_ctx_option: launch-geometry and behaviour options of a context (no reference counterpart; the defaults are what the
 * numbers in DESIGN.md were measured with).  None changes a result except key 18, which moves a GP between two distance forms
 * that agree to ~1e-13, and key 51, which evaluates V = L^-1 K*^T in another arithmetic.  Keys (value ranges are checked;
 * GPB_E_ARG otherwise):
 *   0 XCD affinity of the predict kernel (-1 auto, 0 by walker tile, 1 by row block); 4 outer panel width of the blocked
 *   Cholesky (0: by size); 5 tile order of the static 64-row predict launches (1 sorted, 2 snake, 3 snake of pairs);
 *   7 / 22 switch points of the tile-shape rule (64x64 / 64x128 tiles per 256 CUs), 33 / 34 / 35 the same for compacted
 *   batches; 8 largest batch whose dense block log-likelihood runs one workgroup per walker; 9 / 12 / 14 / 50 tile (64, 128; 0 = by
 *   fill) of the in-panel Cholesky updates / the triangular-inverse levels / the end-of-panel updates / K^-1 of the LML gradient; 10 wave priority of
 *   predict tiles by K-loop length; 11 the block log-likelihood kernels sum the predict partials themselves; 17 skip the all-zero
 *   m-tiles of the predict kernel's diagonal blocks; 18 distance form of the kernel matrices (1: per GP from theta, see
 *   GPB_GET_FORM; 0: difference form for every GP; 2: Gram form for every GP); 19 / 20 design chunks per cross-kernel workgroup /
 *   walkers per lane there; 23 low-rank form of the block log-likelihood when it applies; 25 Cholesky lookahead on a side stream;
 *   27 evaluate only the rows inside the prior box; 28 size the tile rule of a compacted batch by its live rows; 29 / 30 fusions
 *   of the resident step loop (box test and gather in the proposal kernel; accept + next proposal in one launch); 36 balanced
 *   row shares of a sharded step loop (0 off: default, 1 from 8 ranks on, 2 always); 40 the emulators of a chain share one launch
 *   per kernel kind; 42 force the predict tile (0: by rule; 128, 64, 32 = 64 rows x 32 walkers, 65 = 64 x 128: every shape gives
 *   the same bits); 43 route the block log-likelihood through the generic LDS / HBM Cholesky kernel (what M > 64 takes) whatever M;
 *   44 the number of 128x128 predict tiles per 256 CUs from which the rule takes them (0: default 960);
 *   47 Cholesky by column pairs (every second trailing update takes two block columns at once, K = 128): 1 where it is the faster
 *   schedule (default: 1024 <= N <= 3072), 2 always, 0 never; results agree to rounding (another order of the same sums);
 *   49 the block log-likelihoods of a chain of emulators as one workgroup per (walker tile, emulator) and an ordered sum (1,
 *   default) or as one workgroup per walker tile that walks the emulators (0); same bits;
 *   51 V = L^-1 K*^T (sk:_gpr.py:454-460, src/emulator.py:573-575) on the INT8 matrix pipe (csrc/gpb_sliced.hip): 0 never
 *   (default), 1 for every batch of a context whose GPs all have 1 + c / sigma_n^2 <= 128 (the rule reads theta alone; other
 *   contexts keep the fp64 kernel), 2 the rule off (accuracy probes).  Operands as six signed 8-bit digit planes, the 21 digit
 *   products of the upper levels summed exactly in int32, combined in fp64: the variance within ~2e-11 relative of the fp64
 *   kernel's (1e-10 bar kept), a log-posterior within ~1e-11 .. 2e-9 depending on how far its two terms cancel; a walker's bits
 *   still do not depend on batch, tile, compaction or rank count.  1.8-2.1x the fp64 kernel at cfg 4.
 *   Keys 26 / 32 (one rank's share of a sharded step on a single GPU: a measurement hook) take non-zero values in the debug build
 *   only (libgpbayes_debug.so: include/gpbayes_debug.h) and return GPB_E_ARG here.
 * gpb_debug_has_variants: 1 when the loaded library is that debug build (-DGPB_DEBUG_VARIANTS), 0 for the product library.
 * gpb_profile_enable / _read: HIP-event timing of the dominant kernel (k_predict: V = L^-1 K*^T + sum of squares) on the
 *   context's stream — number of timed launches, their summed duration, the (GP, walker) pairs they processed; read resets.
 *   bench.py's roofline block comes from these.
 * gpb_profile_fit_piece: enqueue ONE piece of gpb_gp_factor alone (0 = K(X,X) assembly, 1 = Cholesky, 2 = triangular inverse,
 *   3 = alpha) so that the pieces can be timed apart; leaves the context without a factorisation (gpb_gp_factor afterwards). */
GPB_API int gpb_ctx_option(gpb_ctx* ctx, int key, int value);
GPB_API int gpb_debug_has_variants(void);
GPB_API int gpb_profile_enable(gpb_ctx* ctx, int on);
GPB_API int gpb_profile_read(gpb_ctx* ctx, int64_t* launches, double* total_ms, double* units);
GPB_API int gpb_profile_fit_piece(gpb_ctx* ctx, int piece);

#ifdef __cplusplus
}
#endif
#endif /* GPBAYES_H */
