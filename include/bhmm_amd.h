/*
 * bhmm_amd.h -- C ABI of the MI355X (gfx950) forward-backward / Baum-Welch engine.
 *
 * This is the drop-in boundary for the hot path of bhmm: plain
 * pointers and sizes, no torch / numpy types.  Two families of entry points:
 *
 *  (1) Single-trajectory, host-pointer kernels with the argument meaning of the
 *      reference's native layer (bhmm/hidden/impl_c/_hidden.h:10-63,
 *      bhmm/output_models/impl_c/_gaussian.h:5, discrete.pyx:25).  They are what
 *      bhmm/hidden/impl_c/hidden.pyx binds today; a maintainer can bind these instead
 *      (INTEGRATION.md shows the ctypes / Cython stubs).  Lengths are int64 here; the
 *      reference uses 32-bit int T (T*N < 2^31).
 *
 *  (2) A batched, device-resident context: observations are uploaded once (they are
 *      constant across EM iterations, maximum_likelihood.py:101) and every EM iteration /
 *      Gibbs sweep is ONE call that processes all trajectories on the GPU and returns
 *      only the reduced sufficient statistics.  It replaces the Python loops at
 *      bhmm/estimators/maximum_likelihood.py:221-282,383-385 (E-step),
 *      :332-352 (Viterbi) and bhmm/estimators/bayesian_sampling.py:283-331 (Gibbs
 *      hidden-path step).
 *
 * All functions return an int status (never exit(), unlike _hidden.c:299-304):
 */
#ifndef BHMM_AMD_H_
#define BHMM_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BHMM_OK 0
#define BHMM_ERR_NO_MEM 2      /* same value as _BHMM_ERR_NO_MEM, _hidden.h:5 */
#define BHMM_ERR_INVALID 3     /* bad argument (shape, NULL, unsupported size) */
#define BHMM_ERR_HIP 4         /* HIP runtime error; see bhmm_last_error() */
#define BHMM_ERR_NONFINITE 5   /* log-likelihood not finite (maximum_likelihood.py:385) */
#define BHMM_ERR_CHOICE 6      /* inverse-CDF draw found no state (_hidden.c:299-304) */
#define BHMM_ERR_NO_DEVICE 7   /* no HIP device / library built without one visible */
#define BHMM_ERR_SIGMA 8       /* M-step: a sigma fell below eps (gaussian.py:271-272, RuntimeError) */
#define BHMM_ERR_DISCONNECTED 9 /* reversible sampling of a disconnected count matrix
                                  (bayesian_sampling.py:347-350, NotImplementedError) */

/* emission model kinds */
#define BHMM_EMIT_GAUSSIAN 0 /* obs: double[T];  par0 = means[N], par1 = sigmas[N]            */
#define BHMM_EMIT_DISCRETE 1 /* obs: int32[T];   par0 = B[N*M] row-major, par1 unused         */
#define BHMM_EMIT_EXPLICIT 2 /* obs: double[T*N] = pobs rows (hidden/api.py signatures)       */

/* flags for bhmm_estep */
#define BHMM_FLAG_STORE_GAMMA 1 /* keep gamma (T,N) per trajectory on the device            */

const char *bhmm_last_error(void);
int bhmm_device_count(void);
/* library version / build info: "bhmm_amd <ver> gfx950" */
const char *bhmm_version(void);

/* ------------------------------------------------------------------------------------
 * (1) reference-shaped single-trajectory kernels, host pointers, row-major (T,N) arrays.
 *     Each runs on the current HIP device, synchronously.
 * ---------------------------------------------------------------------------------- */

/* replaces _forward (_hidden.h:10-15, _hidden.c:16-66): fills alpha[T*N], *logprob */
int bhmm_forward(double *alpha, double *logprob, const double *A, const double *pobs,
                 const double *pi, int N, int64_t T);
/* replaces _backward (_hidden.h:17-21, _hidden.c:69-110): fills beta[T*N] */
int bhmm_backward(double *beta, const double *A, const double *pobs, int N, int64_t T);
/* replaces hidden/api.py:133-188 (numpy) / _computeGamma (_hidden.c:113-131) */
int bhmm_state_probabilities(double *gamma, const double *alpha, const double *beta, int N,
                             int64_t T);
/* replaces _compute_transition_counts (_hidden.h:34-40, _hidden.c:148-183); C overwritten */
int bhmm_transition_counts(double *C, const double *A, const double *pobs, const double *alpha,
                           const double *beta, int N, int64_t T);
/* replaces _compute_viterbi (_hidden.h:42-47, _hidden.c:203-281); path is int32[T] */
int bhmm_viterbi(int32_t *path, const double *A, const double *pobs, const double *pi, int N,
                 int64_t T);
/* replaces _sample_path (_hidden.h:49-54, _hidden.c:330-378).  u[T] are the uniforms in
 * [0,1), u[t] used for step t (the reference draws them from libc rand(), T-1 first). */
int bhmm_sample_path(int32_t *path, const double *alpha, const double *A, const double *u,
                     int N, int64_t T);
/* The uniforms the reference's _sample_path would consume after set_seed(seed)
 * (_hidden.c:283-327: r = rand()/(RAND_MAX+1.0), first draw used for t = T-1).  Host-only
 * helper on the C library generator; seed < 0 leaves the generator state untouched
 * (hidden.pyx:181-182 seeds only when a seed is given). */
int bhmm_libc_uniforms(double *u, int64_t T, int seed);
/* replaces _p_obs (_gaussian.h:5, _gaussian.c:45-70) + outputmodel.py:119-131 */
int bhmm_pobs_gaussian(double *pobs, const double *obs, const double *mu, const double *sigma,
                       int N, int64_t T, int ignore_outliers);
/* replaces _update_pout (discrete.pyx:25, _discrete.c:1-32): pout[N*M] += scatter */
int bhmm_update_pout(double *pout, const int32_t *obs, const double *weights, int64_t T, int N,
                     int M);

/* ------------------------------------------------------------------------------------
 * (2) batched device-resident context
 * ---------------------------------------------------------------------------------- */
typedef struct bhmm_ctx bhmm_ctx;

/* device: HIP ordinal.  stream: a hipStream_t to launch on (NULL = the context creates
 * its own).  The context owns every device buffer it allocates. */
int bhmm_ctx_create(bhmm_ctx **out, int device, void *stream);
int bhmm_ctx_destroy(bhmm_ctx *ctx);

/* Upload K trajectories.  obs is the concatenation of all trajectories (element type per
 * `kind`), offsets[K+1] the element offsets of each trajectory in time steps.
 * nstates = N (1 .. 4096: N <= 8 chunk-parallel kernels, 9 .. 64 the lane-per-state family, above
 * that the any-N family -- E-step on the matrix cores up to 512 states, order-faithful kernels of
 * gen_kernels.hpp beyond and for the bit-exact passes; the reference's _hidden.c has no limit either),
 * nsymbols = M (discrete only).  chunk = time-chunk length used for the
 * parallel-in-time decomposition (0 = choose automatically).  obs_on_device != 0 means
 * `obs` is already a device pointer on the context's device (offsets stay on the host). */
int bhmm_ctx_set_observations(bhmm_ctx *ctx, int kind, const void *obs, const int64_t *offsets,
                              int K, int nstates, int nsymbols, int chunk, int obs_on_device);

/* Lagged views of the observations (bhmm/api.py:70-94, lag_observations): view v is the
 * sub-sampled trajectory obs_k[shift::lag] with k = view_traj[v], shift = view_shift[v], and
 * becomes trajectory v of the context.  The ORIGINAL observations (obs, offsets[K+1], as for
 * bhmm_ctx_set_observations) are uploaded once and the views are cut on the device by strided
 * reads, instead of lag host-side copies uploaded one by one. */
int bhmm_ctx_set_observations_lagged(bhmm_ctx *ctx, int kind, const void *obs,
                                     const int64_t *offsets, int K, int lag,
                                     const int32_t *view_traj, const int32_t *view_shift, int V,
                                     int nstates, int nsymbols, int chunk, int obs_on_device);

/* Number of doubles in the packed statistics vector produced by bhmm_estep for the loaded
 * observations:  [0] sum_k logL_k | [1..N] sum_k gamma_k[0] | N*N transition counts C |
 * N state counts sum_t gamma | emission block:
 *   gaussian: N sum gamma*(o-mu_old), N sum gamma*(o-mu_old)^2     (2N)
 *   discrete: N*M weighted symbol counts (row-major, unnormalised) (N*M)
 *   explicit: none. */
int bhmm_ctx_stats_size(const bhmm_ctx *ctx);

/* One E-step over all loaded trajectories with the given model (host pointers):
 * fused emission probabilities + scaled forward + backward + gamma/xi/emission statistics
 * (maximum_likelihood.py:221-282).  Asynchronous on the context's stream.
 *   stats_dev : device buffer of bhmm_ctx_stats_size() doubles, or NULL to use the
 *               context's internal buffer (read back with bhmm_estep_fetch).  A caller
 *               running several ranks all-reduces this buffer (RCCL) before fetching.
 *   flags     : BHMM_FLAG_* */
int bhmm_estep(bhmm_ctx *ctx, const double *A, const double *pi, const double *par0,
               const double *par1, double *stats_dev, int flags);
/* Wait for the last E-step and copy results to the host.  stats (packed vector) and/or
 * logL_k[K] may be NULL.  Returns BHMM_ERR_NONFINITE if any logL_k is not finite.  With more than
 * 4096 trajectories the K log-likelihoods cross the host link only when logL_k is asked for (or the
 * total is not finite and the offending trajectory has to be named): an EM loop needs stats[0].
 * BHMM_ERR_NONFINITE also if the log-likelihoods are finite but a count (sum gamma_0, C, sum gamma) is
 * not even after bhmm_estep's own retry on one chunk per trajectory (reducible
 * transition matrices with emission probabilities hundreds of decades apart) -- refused loudly instead
 * of handing NaN counts to an M-step.
 * ORDER when bhmm_estep wrote into the caller's stats_dev: the FIRST fetch after the launch (with `stats`, or
 * with logL_k only as a sharded caller does) is where this rank's result is inspected, once per launch --
 * the retry above re-runs the local E-step into the same buffer -- so call it BEFORE reducing the buffer in
 * place; later fetches of the same launch do not look at the buffer's counts again. */
int bhmm_estep_fetch(bhmm_ctx *ctx, double *stats, double *logL_k);
/* After an E-step run with BHMM_FLAG_STORE_GAMMA: copy gamma of trajectory k, (T_k,N)
 * row-major, to the host. */
int bhmm_get_gamma(bhmm_ctx *ctx, int k, double *gamma);

/* Viterbi paths of all trajectories (maximum_likelihood.py:332-352).  paths is a host
 * buffer of sum_k T_k int32, trajectory-concatenated like obs. */
int bhmm_viterbi_batch(bhmm_ctx *ctx, const double *A, const double *pi, const double *par0,
                       const double *par1, int32_t *paths);

/* Same paths as bhmm_viterbi_batch, one byte per step (for N <= 256 states -- more return
 * BHMM_ERR_INVALID here and use the int32 form; the int32 form
 * above is the reference's return type, hidden.pyx:161-162, and costs four times the copy).
 * paths_on_device == 0: paths is a host buffer of sum_k T_k bytes -- pass pinned memory
 * (hipHostMalloc / torch pin_memory) for the full link rate, a pageable buffer is pinned for the
 * duration of the copy.  paths_on_device != 0: paths is a device buffer on the context's device
 * and the back-trace kernels write it directly (no copy).  Replaces the Python loop at
 * maximum_likelihood.py:332-352 for callers that keep or post-process paths on the GPU. */
int bhmm_viterbi_batch_u8(bhmm_ctx *ctx, const double *A, const double *pi, const double *par0,
                          const double *par1, uint8_t *paths, int paths_on_device);

/* Gibbs hidden-path step (bayesian_sampling.py:283-331): forward pass + backward sampling
 * of every trajectory.  Uniforms come either from u (host, concatenated like obs; u[t]
 * used at step t) or, when u == NULL, from a counter-based generator seeded with `seed`.
 * Outputs (any may be NULL): paths (host, int32, concatenated), and the hidden-path
 * statistics the sweep needs (generic_hmm.py:297-334,398-431):
 *   counts[N*N] int64 transition counts, n0[N] int64 first-state counts,
 *   emis: gaussian -> 3N doubles: n_i, sum (o - mu_i), sum (o - mu_i)^2 over the steps
 *         assigned to state i (mu = par0, the current means); discrete -> N*M counts. */
int bhmm_sample_paths(bhmm_ctx *ctx, const double *A, const double *pi, const double *par0,
                      const double *par1, const double *u, uint64_t seed, int32_t *paths,
                      int64_t *counts, int64_t *n0, double *emis);

/* The same Gibbs step with the hidden-path statistics left ON THE DEVICE as one packed fp64
 * vector of bhmm_ctx_path_stats_size() doubles,
 *   [ counts N*N | n0 N | emission block (gaussian 3N: n_i, sum d, sum d^2; discrete N*M) ],
 * so that several ranks reduce them with ONE all-reduce (RCCL) and ONE device-to-host copy -- the
 * distributed form of generic_hmm.py:297-334,398-431.  The integer counts are < 2^53, so their
 * fp64 sums are exact whatever the order.  paths (host int32, concatenated) may be NULL. */
int bhmm_ctx_path_stats_size(const bhmm_ctx *ctx);
/* Position of each loaded trajectory in the device random stream used when u == NULL: step t of
 * trajectory k draws uniform(seed, soff[k] + t).  Default (soff == NULL, and after every
 * bhmm_ctx_set_observations): the trajectory's offset in this context.  A caller that shards
 * trajectories over several contexts / GPUs passes their offsets in the UNSHARDED concatenation,
 * so that the sampled paths do not depend on the partition.  soff: host, K entries. */
int bhmm_ctx_set_stream_offsets(bhmm_ctx *ctx, const int64_t *soff);
int bhmm_sample_paths_dev(bhmm_ctx *ctx, const double *A, const double *pi, const double *par0,
                          const double *par1, const double *u, uint64_t seed, int32_t *paths,
                          double *stats_dev);

/* Tuning / introspection knobs by name (returns BHMM_ERR_INVALID for an unknown name):
 *   "spec_enabled"  1/0  use speculative, verified chunk boundaries in bhmm_estep (default 1;
 *                        switched off automatically when verification keeps failing)
 *   "spec_W"        warm-up length in time steps.  By default it is read off the forgetting
 *                   curve that the first E-step on new observations measures for the model at
 *                   hand (two differently started chains on 256 sampled stretches, both
 *                   directions; +15 %, 9..64 states: +50 %); a failed check re-measures and
 *                   lengthens it.  Setting it (or BHMM_AMD_SPEC_W) fixes it
 *   "wide_segments" 1/0  (9..64 states) cut trajectories into time segments with the same
 *                        verified warm-up boundaries; reading it returns the segment count in use
 *   "wide_segment_len"   segment length for the next bhmm_ctx_set_observations (0 = automatic)
 *   "carry"         1/0  (N <= 8) in a sequence of E-steps on slowly changing models (EM) start the
 *                        warm-ups from the PREVIOUS E-step's boundary vectors, a shorter distance
 *                        out, sized from the model change and the measured sensitivity; verified
 *                        like every warm-up, repeated with full warm-ups if the check fails.  Not
 *                        used when the model is identical to the previous call's (default 1;
 *                        BHMM_AMD_CARRY=0)
 *   "carry_W", "carry_ok", "carry_fail"  (read-only) warm-up steps of the last E-step's carried
 *                        starts (0: full warm-ups), E-steps that verified on carried starts /
 *                        had to be repeated
 *   "spec_ok", "spec_fail"  (read-only) E-steps whose boundaries verified / fell back
 *   "spec_last_dev" (read-only) largest relative boundary deviation of the last check
 *   "careful"       (read-only) 1 after an E-step met an all-zero emission row (gaussian
 *                   outlier rule, outputmodel.py:126-130) and switched to the kernel that
 *                   applies the rule per step; 9..64 states: 1 after a lazily scaled vector
 *                   left its range and the E-step was repeated with per-step normalisation
 *   "viterbi_chunked" (read-only) 1 if the last bhmm_viterbi_batch ran parallel over time
 *                   chunks (boundaries verified, no close decision), 0 if it ran serially
 *   "viterbi_margin" 1/0/2  (9..256 states) accept a segment-parallel first pass whose boundaries equal their
 *                   predecessors' vectors to 1e-12 by the margins of the decisions on its path (2: up to 64 states
 *                   from the next call on, not only after a call that needed rounds); "viterbi_mend" 1/0  (9..64
 *                   states) segments further than 1e-12 from their predecessor run again alone up to a kept
 *                   vector of the first pass;  read-only: "viterbi_segments", "viterbi_W", "viterbi_mismatch",
 *                   "viterbi_far", "viterbi_mended", "viterbi_rounds", "viterbi_margin_used",
 *                   "viterbi_margin_close"
 *   "draw_watch"    1/0  (bhmm_sample_paths*) draws whose cumulative sums lie within 64 x the deviation the
 *                   forward pass's boundary check measured are decided again on the serial recursion over a
 *                   long window (csrc/draw_verify.hpp); if one does not stand the call is repeated on exact
 *                   alpha rows (default 1).  read-only: "draw_events", "draw_checked", "draw_redone",
 *                   "draw_alpha_dev";  tests: "draw_watch_tol" (watch tolerance), "draw_test_redo" */
int bhmm_ctx_set_option(bhmm_ctx *ctx, const char *name, double value);
int bhmm_ctx_get_option(bhmm_ctx *ctx, const char *name, double *value);

/* introspection (used by bench.py / tests) */
int64_t bhmm_ctx_total_steps(const bhmm_ctx *ctx);
int bhmm_ctx_num_chunks(const bhmm_ctx *ctx);
int bhmm_ctx_chunk_len(const bhmm_ctx *ctx);
/* time in milliseconds spent in the named kernel during the last bhmm_estep, measured with
 * HIP events on the context's stream.  which: 0 prescan, 1 stitch, 2 forward-backward,
 * 3 finalize, 4 whole E-step. Valid after bhmm_estep_fetch (or a stream sync). */
double bhmm_ctx_last_kernel_ms(bhmm_ctx *ctx, int which);
/* all five at once: out[5] (one call inside a timed loop instead of five) */
int bhmm_ctx_last_kernel_ms_all(bhmm_ctx *ctx, double *out);
void *bhmm_ctx_stream(bhmm_ctx *ctx);
int bhmm_ctx_sync(bhmm_ctx *ctx);
/* ------------------------------------------------------------------------------------
 * (2b) more than one GPU: one process (or thread) per device, trajectories sharded over the
 *      ranks (they are independent given the model: maximum_likelihood.py:383-385,
 *      bayesian_sampling.py:288-290), ONE all-reduce of the packed sufficient statistics per EM
 *      iteration / Gibbs sweep -- the distributed form of the host sums at
 *      bhmm/estimators/maximum_likelihood.py:271-282.  RCCL over xGMI, loaded on first use
 *      (librccl.so; BHMM_ERR_INVALID with a message if it is not there -- the rest of the
 *      library does not need it).  The Python estimators use torch.distributed for the same sum
 *      (bhmm_amd/sharding.py); these entry points give a non-Python binder the same thing.
 *
 *   bhmm_comm_unique_id : rank 0 fills 128 bytes and hands them to the other ranks out of band
 *   bhmm_comm_init_rank : collective over the nranks callers; `device` is this rank's GPU
 *   bhmm_ctx_allreduce_stats : in-place sum over the ranks of count doubles in a DEVICE buffer --
 *                          what bhmm_estep(..., stats_dev) / bhmm_sample_paths_dev left there --
 *                          enqueued on the context's stream (no host synchronisation: follow it
 *                          with the copy to the host on the same stream, or bhmm_ctx_sync)
 * ---------------------------------------------------------------------------------- */
typedef struct bhmm_comm bhmm_comm;
#define BHMM_COMM_ID_BYTES 128
int bhmm_comm_unique_id(void *id);
int bhmm_comm_init_rank(bhmm_comm **out, int device, int nranks, int rank, const void *id);
int bhmm_comm_destroy(bhmm_comm *comm);
int bhmm_comm_size(const bhmm_comm *comm, int *nranks, int *rank);
int bhmm_ctx_allreduce_stats(bhmm_ctx *ctx, bhmm_comm *comm, double *stats_dev, int64_t count);

/* Measurement / test support: K synthetic trajectories of T steps each, drawn ON
 * THE DEVICE from the HMM (A, pi, emission) -- hidden path by inverse CDF of pi / the rows of A,
 * one emission per step (recipe of bhmm/hmm/generic_hmm.py:435-507) -- with the counter-based
 * stream of bhmm_sample_paths: step t of trajectory k uses uniform(seed, 2*(k*T+t)) for the state
 * and uniform(seed, 2*(k*T+t)+1) for the emission, so the result does not depend on launch
 * geometry and a host restatement reproduces discrete trajectories bit for bit.
 *   obs_dev    : device buffer, K*T elements, trajectory-concatenated (double | int32 per kind)
 *   states_dev : optional device buffer of K*T bytes receiving the hidden path (may be NULL)
 *   stream     : hipStream_t or NULL;  par0/par1 as for the emission kinds above. */
int bhmm_synth_observations(void *obs_dev, uint8_t *states_dev, int device, void *stream, int kind,
                            const double *A, const double *pi, const double *par0,
                            const double *par1, int N, int M, int K, int64_t T, uint64_t seed);

/* The same for a SLICE of a larger set: this call's trajectory k is trajectory first_traj + k of
 * the set (stream positions 2 * ((first_traj + k) * T + t)), so ranks that each draw their own shard
 * hold exactly the trajectories one process would draw for the whole set. */
int bhmm_synth_observations_at(void *obs_dev, uint8_t *states_dev, int device, void *stream, int kind,
                               const double *A, const double *pi, const double *par0,
                               const double *par1, int N, int M, int K, int64_t T, uint64_t seed,
                               int64_t first_traj);

/* Host-side M-step helper (no device work): the reversible maximum-likelihood transition matrix
 * of a strongly connected count matrix C[n*n] -- the estimator bhmm takes from msmtools
 * (bhmm/estimators/_tmatrix_disconnected.py:94-105, maximum_likelihood.py:306-320) -- by the
 * fixed point x_ij <- (c_ij + c_ji) / (c_i / x_i + c_j / x_j), P_ij = x_ij / x_i, iterated until the
 * row sums of x move by less than maxerr (or maxiter).  P[n*n] row-major; *iterations (optional)
 * receives the number of iterations.  In numpy this loop costs tens of milliseconds per EM
 * iteration next to a 1 ms E-step. */
int bhmm_mle_reversible(double *P, int64_t *iterations, const double *C, int n, int64_t maxiter,
                        double maxerr);

/* ------------------------------------------------------------------------------------
 * (3) host-side model updates between two passes over the trajectories (no device work).
 *     The reference does these in Python / numpy / msmtools; next to a 0.9 ms E-step or Gibbs path
 *     step they must not cost milliseconds, so each is ONE call.
 * ---------------------------------------------------------------------------------- */

/* The whole M-step of one EM iteration (maximum_likelihood.py:284-330) from the packed statistics
 * vector of bhmm_estep (layout: bhmm_ctx_stats_size):
 *   transition matrix  estimate_P (_tmatrix_disconnected.py:68-123: strongly / weakly connected
 *                      sets, reversible fixed point, partially reversible iteration :126-190,
 *                      row normalisation, estimator with a fixed stationary vector),
 *   initial / stationary distribution (maximum_likelihood.py:310-320, _tmatrix_disconnected.py:229-251),
 *   emission parameters (gaussian.py:214-272 from moments about the old means; discrete.py:202-215).
 *   reversible : 1 / 0, or -1 = "reversible iff T_old is" (the reference passes
 *                self._hmm.is_reversible, _tmatrix_disconnected.py:213-226)
 *   stationary : pi_new = stationary distribution of T_new (else normalised sum_k gamma_k[0])
 *   fixed_pi   : NULL, or the fixed stationary (stationary != 0) / initial (stationary == 0) vector
 *   par0_old / par1_old : current emission parameters (gaussian: means, sigmas; else may be NULL)
 *   outputs    : T_new[n*n], pi_new[n], par0_new (means[n] | B[n*M]), par1_new (sigmas[n] | unused)
 *   info       : optional int32[2]: reversible branch taken, fixed-point iterations
 *   warm_state : optional, 1 + n doubles owned by the caller, zero-initialised and handed to every
 *                call of one EM run: the reversible fixed point then starts from the previous
 *                iteration's solution (same fixed point, same stopping rule, far fewer iterations;
 *                only while all states form one closed connected set).  NULL: cold start.
 * Returns BHMM_ERR_SIGMA if a sigma falls below machine epsilon. */
int bhmm_mstep(int kind, int n, int M, const double *stats, const double *T_old,
               const double *par0_old, const double *par1_old, int reversible, int stationary,
               const double *fixed_pi, int64_t maxiter, double maxerr, double mincount,
               double *T_new, double *pi_new, double *par0_new, double *par1_new, int32_t *info,
               double *warm_state);

/* The parameter draws of one Gibbs sweep (bayesian_sampling.py:333-373) from the packed path
 * statistics of bhmm_sample_paths_dev (layout: bhmm_ctx_path_stats_size), in the reference's order:
 *   emission parameters (gaussian.py:303-318 | discrete.py:243-251; par0 / par1 in and out),
 *   transition matrix   reversible: start at the reversible MLE of C = counts + prior_C, zero
 *                       pattern made consistent (:352-357), then `nsteps` FULL sweeps of the
 *                       element-wise Gibbs sampler of Trendelkamp-Schroer et al. 2015 (what
 *                       msmtools.estimation.sample_tmatrix(nsteps=...) counts); else independent
 *                       Dirichlet rows,
 *   initial distribution Dirichlet(n0 + prior_n0) over the positive entries, or (stationary != 0)
 *                       the stationary distribution of the new matrix.
 * Random numbers come from a counter-based generator: the result is a function of (seed, sweep)
 * and the inputs alone, so every rank of a sharded run draws the SAME parameters from the
 * all-reduced statistics without a broadcast.  prior_* may be NULL (zero).  info: optional
 * int32[1], number of 64-bit draws consumed.  Returns BHMM_ERR_DISCONNECTED for reversible
 * sampling of a count matrix that is not strongly connected. */
int bhmm_gibbs_parameters(int kind, int n, int M, const double *path_stats, const double *prior_C,
                          const double *prior_n0, const double *prior_B, int reversible,
                          int stationary, int64_t nsteps, uint64_t seed, uint64_t sweep, double *T,
                          double *p0, double *par0, double *par1, int32_t *info);

/* building blocks of the two calls above, exported for the parity / property tests:
 * component label per state (sets numbered by decreasing size, _tmatrix_disconnected.py:28-43) */
int bhmm_host_connected_sets(int32_t *label, const double *C, int n, double mincount, int strong);
int bhmm_host_stationary_vector(double *pi, const double *P, int n);
int bhmm_host_estimate_tmatrix(double *P, const double *C, int n, int reversible, const double *fixed_pi,
                         int64_t maxiter, double maxerr, double mincount, int64_t *iterations);
int bhmm_host_is_reversible(const double *P, int n); /* 1 / 0 (-1: bad argument) */
/* `nsweeps` full sweeps of the reversible transition-matrix posterior sampler (the draw of
 * bayesian_sampling.py:341-360 inside bhmm_gibbs_parameters) on the symmetric flux matrix X (n x n, in/out)
 * for the count matrix C.  base keys the per-update random streams; lanes: 0 = the widest instantiation the
 * CPU runs, 1 = one lane, 4 = AVX2 -- the result does not depend on it (that is what the tests check).
 * Returns the number of lanes used, negative on a bad argument. */
int bhmm_host_sample_reversible(double *X, const double *C, int n, int64_t nsweeps, uint64_t base, int lanes);
/* _tmatrix_disconnected.py:126-190: rows in_set != 0 of P (n x n, in/out) */
int bhmm_host_partial_rev(double *P, const double *C, int n, const int32_t *in_set, int64_t maxiter,
                          double maxerr);
/* `count` draws of the counter-based generator: what = 0 uniform [0,1), 1 standard normal,
 * 2 gamma(param), 3 uniform (0,1) */
int bhmm_host_rng_draws(double *out, int64_t count, int what, double param, uint64_t seed,
                        uint64_t stream);

/* diagnostics: y[i] = the E-step kernels' exp() for non-positive arguments (the exponential
 * of the gaussian density, _gaussian.c:18), so that tests can bound its error in ulps */
int bhmm_diag_exp_nonpos(double *y, const double *x, int64_t n);
/* y[i] = the E-step kernels' gaussian density of observation o[i] for one state (mu, sigma)
 * (_gaussian.c:18-20); nansafe = the variant of the per-step-checked kernels */
int bhmm_diag_gauss_pdf(double *y, const double *o, int64_t n, double mu, double sigma,
                        int nansafe);

#ifdef __cplusplus
}
#endif
#endif /* BHMM_AMD_H_ */
