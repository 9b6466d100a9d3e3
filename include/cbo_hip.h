/* cbo_hip.h -- C-ABI of libcbo_hip.so: MI355X (gfx950) GP posterior update + causal acquisition sweep.
 *
 * The reference (ChampiB/CBO_with_OOP, pure Python) has no FFI layer; its operator API for this path
 * is a Python duck type.  Each entry point below names the reference interface it
 * replaces (paths relative to /root/reference/).  The Python host side in cbo_with_oop_amd/ binds
 * exactly these symbols with ctypes (cbo_with_oop_amd/_lib.py); INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions: plain pointers and sizes only; all host arrays are C-contiguous float64 (row-major
 * (n,d) for points); the caller owns every host buffer and it is only touched during the call; every
 * function returns 0 (CBO_OK) or a negative cbo_status and records a message for cbo_last_error();
 * a handle is bound to one HIP device + one stream and is not thread-safe (distinct handles are).
 * There is no CPU fallback anywhere behind this interface.
 */
#ifndef CBO_HIP_H
#define CBO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: cbo_timers gained ms_f32_convert (the library writes sizeof(cbo_timers) bytes into the caller's struct),
 *    cbo_gp_create accepts CBO_DTYPE_F32, cbo_synchronize is device-wide.
 * cbo_abi_version() returns this value from a product build.  Timing-only builds (CBO_DIAG_KNOBS, or a non-zero
 * F32_DBG mask, whose results may be wrong by construction) return CBO_HIP_ABI_DIAG_BASE + this value, so that a
 * consumer checking the version refuses them as the product. */
#define CBO_HIP_ABI_VERSION 5
#define CBO_HIP_ABI_DIAG_BASE 1000
#define CBO_MAX_DIM 8

typedef enum cbo_status {
    CBO_OK = 0,
    CBO_ERR_INVALID = -1,     /* bad argument (shape, NULL, dtype) */
    CBO_ERR_HIP = -2,         /* a HIP runtime call failed; see cbo_last_error() */
    CBO_ERR_NOT_PD = -3,      /* jitchol exhausted its 5 retries (numpy.linalg.LinAlgError in GPy) */
    CBO_ERR_NONPOS_DIAG = -4, /* jitchol: "not pd: non-positive diagonal elements" */
    CBO_ERR_NOT_FITTED = -5,
    CBO_ERR_UNSUPPORTED = -6, /* a request this build cannot serve (e.g. a workspace cap too small for the call) */
    CBO_ERR_NO_DEVICE = -7,
    CBO_ERR_COMM = -8
} cbo_status;

enum { CBO_DTYPE_F64 = 0, CBO_DTYPE_F32 = 1 };
enum { CBO_TASK_MIN = 0, CBO_TASK_MAX = 1 };

typedef struct cbo_ctx cbo_ctx;     /* one HIP device + stream + workspaces */
typedef struct cbo_gp cbo_gp;       /* one GP posterior resident on a ctx */
typedef struct cbo_cands cbo_cands; /* one candidate-intervention set resident on a ctx */

/* Per-phase device time of the calls made since cbo_reset_timers(), from hipEvents recorded on the
 * ctx stream (only while profiling is enabled with cbo_set_profiling).  ms_* are sums, n_* counts. */
typedef struct cbo_timers {
    double ms_kxx;     /* K(X,X) assembly (+ diagonal)                       */
    double ms_chol;    /* jittered Cholesky incl. forward solve z = L^-1 r   */
    double ms_alpha;   /* backward solve alpha = L^-T z                      */
    double ms_kstar;   /* K(X,X*) assembly                                   */
    double ms_trsm;    /* V = L^-1 K*, fused sum(V^2) and V^T z  (dominant)  */
    double ms_acq;     /* variance/EI/cost/argmax epilogue                   */
    int64_t n_fit;     /* number of fits timed                               */
    int64_t n_sweep;   /* number of sweeps / predicts timed                  */
    int64_t n_trsm_launches;
    double trsm_flops; /* algorithmic flops of the timed trsm launches: sum n_pad^2 * m_pad */
    double ms_f32_convert; /* CBO_DTYPE_F32 models: factor / diagonal inverses / z down-converted after a fit */
} cbo_timers;

/* ---- context ---------------------------------------------------------------------------------- */
int cbo_abi_version(void);
const char *cbo_last_error(void);
int cbo_device_count(int *count_out);
/* Bind device `device_id`; fails with CBO_ERR_NO_DEVICE when no gfx950-class GPU is visible. */
int cbo_init(int device_id, cbo_ctx **out);
/* Releases everything the context owns.  Contexts still open when the process exits are shut down by an atexit
 * handler the first cbo_init registers (it runs before the HIP runtime's own exit handlers: a context's CU-masked
 * streams must not outlive the runtime -- tools that hook finalisation crash otherwise); a handle that was already
 * shut down is ignored.  Models, candidate sets and communicators of a context must be destroyed before it. */
void cbo_shutdown(cbo_ctx *ctx);
int cbo_synchronize(cbo_ctx *ctx);
int cbo_set_profiling(cbo_ctx *ctx, int enabled);
int cbo_reset_timers(cbo_ctx *ctx);
int cbo_get_timers(cbo_ctx *ctx, cbo_timers *out);
int cbo_device_name(cbo_ctx *ctx, char *buf, int buflen);
/* Device time (hipEvents on the ctx stream) of everything enqueued between the two calls; the work of the
 * other streams a call uses joins the ctx stream before the call returns, so it is covered. */
int cbo_region_begin(cbo_ctx *ctx);
int cbo_region_end(cbo_ctx *ctx, double *ms_out);

/* ---- GP model ----------------------------------------------------------------------------------
 * Replaces GaussianProcessFactory.create / create_non_causal_gp / create_causal_gp / create_graph_gp
 * (src/GaussianProcessFactory.py:24-73) + GPy GPRegression construction.  prior_mean_X / prior_var_X
 * are the reference's mean_function(X) / variance_adjustment(X) closures (DoCalculus.py:34-66)
 * already evaluated on X (NULL, NULL = non-causal kernel, zero mean).  zero_diag selects GPy's
 * X2=None distance shortcut (plain RBF) vs CausalRBF's explicit-X2 path (causal_kernels.py:53-55).
 * Uploads X, y, priors; does not fit. */
/* dtype (BASELINE.json configs[4], "fp32 path with MFMA"): CBO_DTYPE_F64 -- everything fp64; CBO_DTYPE_F32 -- the
 * FIT stays fp64 (Ky carries a 1e-8 jitter fp32 cannot represent) and every sweep / predict of the model runs its
 * substitution V = L^-1 K* on the f32 MFMA (157 TFLOP/s dense against 78.6 for fp64) from a once-per-fit fp32 copy of
 * the factor; K* is evaluated in fp64 and rounded to fp32, q = sum V^2 and mu = V^T z accumulate in fp64.  Accuracy is
 * that of an fp32 triangular solve: about 6e-8 * sqrt(cond(Ky)) relative in V (tests/test_f32_gpu.py states what it
 * meets on the coral ranges); cbo_gp_append and cbo_cands_keep_solution do not apply (the caller refits). */
int cbo_gp_create(cbo_ctx *ctx, int dtype, int64_t n, int d, const double *X, const double *y,
                  const double *prior_mean_X, const double *prior_var_X, double variance,
                  const double *lengthscale /* 1 value, or d values if ard */, int ard,
                  double noise_var, int zero_diag, cbo_gp **out);
void cbo_gp_destroy(cbo_gp *gp);

/* GPy ExactGaussianInference.inference + util.linalg.jitchol: K(X,X) assembly, Ky = K+(noise+1e-8)I,
 * jittered Cholesky (retry ladder mean(diag)*1e-6 x10, <= 5 retries) with the forward solve
 * z = L^-1 (y - m) carried through the factorisation.  GPy's alpha = L^-T z is materialised on first use
 * (cbo_gp_get_posterior): predict and the sweep form the mean as (L^-1 k*)^T z.  Everything runs on the
 * device from the resident X, y.  jitter_tries_out / jitter_out may be NULL. */
int cbo_gp_fit(cbo_gp *gp, int *jitter_tries_out, double *jitter_out);

/* emukit GPyModelWrapper.set_data -> GPy set_XY (called from src/Monitor.py:160): replace the data
 * and refit. */
int cbo_gp_set_data(cbo_gp *gp, int64_t n, const double *X, const double *y,
                    const double *prior_mean_X, const double *prior_var_X);

/* The upload half of cbo_gp_set_data: replace the data and leave the model unfitted, for a caller that refits
 * together with the next sweep (cbo_gp_fit_sweep).  Any call that needs the posterior returns
 * CBO_ERR_NOT_FITTED until then. */
int cbo_gp_upload_data(cbo_gp *gp, int64_t n, const double *X, const double *y,
                       const double *prior_mean_X, const double *prior_var_X);

/* One more observation for a fitted model -- what every CBO trial does to the set it intervened on
 * (src/Monitor.py:148-160 appends to data_x/data_y, src/CBO.py:224-235 rebuilds the model from them).  Appending
 * row/column n to Ky leaves the first n rows of its factor unchanged: the new column is one forward solve, done as
 * an ordinary one-candidate sweep.  *appended_out = 1: the model is fitted on n+1 points; 0: nothing changed, the
 * shortcut does not apply (jitter in the current factor, padded size exhausted, non-positive pivot) and the caller
 * refits with cbo_gp_set_data.  Same results as a full refit up to rounding. */
int cbo_gp_append(cbo_gp *gp, const double *x_new, double y_new, double prior_mean_new, double prior_var_new,
                  int *appended_out);

/* GPyModelWrapper.predict -> GP.predict -> Posterior._raw_predict (called from
 * src/utils_functions/causal_acquisition_functions.py:33 and src/DoCalculus.py:77):
 * mean = K*^T Ky^-1 (y-m) + m(X*), var = clip(Kdiag - |L^-1 K*|^2, 1e-15) (+ noise). */
int cbo_gp_predict(cbo_gp *gp, int64_t m, const double *Xs, const double *prior_mean_s,
                   const double *prior_var_s, int include_noise, double *mean_out, double *var_out);

/* Hyper-parameter MLE support (GPy model.optimize() reached from src/CBO.py:173 and
 * src/utils_functions/utils.py:44).  cbo_gp_set_hyper replaces kernel variance, lengthscale(s) and noise
 * variance (the model must be refitted with cbo_gp_fit); cbo_gp_log_marginal returns GPy's
 * log_marginal_likelihood of the fitted model, 0.5*(-n log 2pi - logdet Ky - r^T Ky^-1 r), computed on the
 * device from the factor's diagonal and z = L^-1 r.  The optimiser loop itself is host logic. */
int cbo_gp_set_hyper(cbo_gp *gp, double variance, const double *lengthscale, double noise_var);
int cbo_gp_log_marginal(cbo_gp *gp, double *lml_out);
/* The gradients GPy hands its optimiser (ExactGaussianInference dL_dK -> kern.update_gradients_full, dL_dthetaL):
 * d log p(y) / d variance, / d lengthscale (1 value, or d values if ard), / d noise_var, of the fitted model, all on
 * the device (Ky^-1 = L^-T L^-1 through the sweep and GEMM kernels, then one contraction pass).  lml_out may be NULL.
 * An fp64 model of at most 128 observations -- every model the reference builds, and what its per-trial optimize()
 * iterates on -- need not be fitted: one launch goes from the data and the current hyper-parameters to all outputs and
 * leaves the model as it was (if Ky is not positive definite as assembled, the model is fitted with the jitchol
 * ladder and the general path answers).  Larger models: CBO_ERR_NOT_FITTED until fitted. */
int cbo_gp_lml_gradients(cbo_gp *gp, double *lml_out, double *dvariance_out, double *dlengthscale_out,
                         double *dnoise_out);

/* Prediction gradients: emukit GPyModelWrapper.get_prediction_gradients -> GPy
 * predictive_gradients, called from CausalExpectedImprovement.evaluate_with_gradients
 * (src/utils_functions/causal_acquisition_functions.py:54) inside the L-BFGS refinement of
 * src/utils_functions/causal_optimizer.py:59-65.  dmean_out / dvar_out: m*d row-major,
 * d mean / d x and d var / d x.  For the causal kernel GPy differentiates the stationary part only, but the
 * solve Ky^-1 k*(x) behind the variance gradient uses the full kernel, hence prior_var_s = variance_adjustment(Xs)
 * (NULL for the non-causal kernel).  Any number of points: the variance gradient's Ky^-1 k*(x) is a forward and a
 * backward substitution of the whole batch, both by the sweep's strip kernel (the backward one on the factor read
 * in reversed index order, which is lower triangular again), chunked like a sweep. */
int cbo_gp_predict_gradients(cbo_gp *gp, int64_t m, const double *Xs, const double *prior_var_s,
                             double *dmean_out, double *dvar_out);

/* Do-calculus prior (src/DoCalculus.py:34-89): predict at m_groups * group points and
 * average the predictive mean and variance over each consecutive run of `group` rows -- one run per
 * candidate intervention, its rows being the observed inputs with the intervened columns overwritten
 * (DoCalculus.compute_do / get_intervened_inputs, then np.mean over the rows, :59-60).  The reduction runs
 * on the device; only m_groups values per output come back. */
int cbo_gp_predict_grouped(cbo_gp *gp, int64_t m_groups, int64_t group, const double *Xs,
                           const double *prior_mean_s, const double *prior_var_s, int include_noise,
                           double *mean_out /* m_groups */, double *var_out /* m_groups */);

/* The same reduction with the prediction points built on the device: candidate c's rows are the n_obs rows of
 * `observed` (n_obs x d, the graph-level GP's inputs) with column j replaced by values[c * n_iv + iv_index[j]] wherever
 * iv_index[j] >= 0 (DoCalculus.get_intervened_inputs, src/DoCalculus.py:80-89).  Only observed and values cross the
 * host link -- the m * n_obs points (16384 candidates x 1000 rows = 393 MB for d = 3) never exist on the host.
 * Non-causal (graph-level) models only. */
int cbo_gp_predict_do(cbo_gp *gp, int64_t m, int64_t n_obs, const double *observed, int n_iv, const double *values,
                      const int *iv_index /* d */, int include_noise, double *mean_out /* m */, double *var_out /* m */);

/* Posterior state for inspection / tests (GPy posterior.woodbury_chol, .woodbury_vector).
 * L_out: n*n row-major lower triangle (upper part zero); alpha_out: n. Either may be NULL. */
int cbo_gp_get_posterior(cbo_gp *gp, double *L_out, double *alpha_out);
/* Assembled Ky (before factorisation) of the last fit attempt is not kept; this re-assembles
 * K(X,X) + diag into K_out (n*n row-major, symmetric) for tests of the assembly kernel. */
int cbo_gp_assemble_kxx(cbo_gp *gp, double *K_out);
int64_t cbo_gp_n(const cbo_gp *gp);
int cbo_gp_dtype(const cbo_gp *gp);
/* Outcome of the jitchol ladder of the last fit: retries used (0 = none) and jitter added. */
int cbo_gp_jitter(const cbo_gp *gp, int *jitter_tries_out, double *jitter_out);

/* ---- candidate sets ---------------------------------------------------------------------------
 * A candidate-intervention grid resident in HBM (the generalisation of the 100 random anchors of
 * src/utils_functions/causal_optimizer.py:52-55).  index_offset is added to local
 * row numbers when reporting the arg-max (candidate shards of a global grid). */
int cbo_cands_create(cbo_ctx *ctx, int64_t m, int d, const double *Xs, const double *prior_mean_s,
                     const double *prior_var_s, int64_t index_offset, cbo_cands **out);
void cbo_cands_destroy(cbo_cands *c);
/* Keep V = L^-1 K* of this candidate set resident after a sweep (n_pad * m_pad doubles): when the model is then
 * extended with cbo_gp_append, the next sweep adds ONE row to V (O(n m)) instead of redoing the substitution
 * (O(n^2 m)).  Off by default. */
int cbo_cands_keep_solution(cbo_cands *c, int on);

/* ---- acquisition sweep -------------------------------------------------------------------------
 * Replaces the batched `acquisition.evaluate(X)` of the anchor scoring step
 * (causal_optimizer.py:52-55) = CausalExpectedImprovement.evaluate
 * (causal_acquisition_functions.py:27-43) / Cost.evaluate (cost_functions.py:11-17), followed by the
 * top-1 selection.  acq = sign * s (u Phi(u) + phi(u)) / cost, u = (y_best - (mean + ei_jitter))/s;
 * task max returns -EI with the same u (reference quirk).  best_idx is the lowest index attaining
 * the maximum (NaN counts as maximal, like numpy.argmax), offset by the set's index_offset.
 * Outputs stay on the device unless asked for: acq_out / mean_out / var_out (m doubles each) may be
 * NULL. */
int cbo_acq_sweep(cbo_gp *gp, cbo_cands *cands, double y_best, int task, double ei_jitter,
                  double cost, double *acq_out, double *mean_out, double *var_out, double *best_val,
                  int64_t *best_idx);

/* Refit (as cbo_gp_fit, jitchol ladder included) and sweep (as cbo_acq_sweep) in one call, overlapped: the
 * sweep's substitution advances panel by panel on a second stream while the factorisation's chain of short
 * kernels runs.  This is the pair of calls CBO.intervene() makes for the set it has just intervened on
 * (src/Monitor.py:160 set_data -> refit; src/CBO.py:250-257 find_next_y_point).  Same outputs as the two calls
 * in sequence, also on failure: on a non-OK return (CBO_ERR_NOT_PD once jitchol's ladder is exhausted) acq_out /
 * mean_out / var_out / best_val / best_idx are left untouched.  tries_out / jitter_out as in cbo_gp_fit (may be NULL). */
int cbo_gp_fit_sweep(cbo_gp *gp, cbo_cands *cands, double y_best, int task, double ei_jitter, double cost,
                     double *acq_out, double *mean_out, double *var_out, double *best_val,
                     int64_t *best_idx, int *tries_out, double *jitter_out);

/* Every exploration set of a trial in one call: CBO.compute_best_acquisition_values (src/CBO.py:237-260) loops
 * find_next_y_point over the S sets (S = 2 toy, 6 complete, 25 coral).  Pair i is (gps[i], cands[i]) with its own
 * incumbent y_best[i] and batch cost costs[i]; best_vals / best_idxs receive S winners.  Sets whose model has at most
 * 128 observations (every model the reference builds: 10 + <= 40 points) are factored AND swept by one launch inside
 * LDS -- no per-set launch chain, no per-set synchronisation, one copy back; such a model need not be fitted
 * (cbo_gp_upload_data suffices) and its fitted state is left alone.  Larger models, fp32 models and sets whose
 * factorisation needs jitchol's jitter take the general path (cbo_gp_fit_sweep if unfitted, else cbo_acq_sweep).
 * Same numbers as the per-set calls (same device functions, same summation orders).  All pairs on one context. */
int cbo_acq_sweep_sets(int n_sets, cbo_gp *const *gps, cbo_cands *const *cands, const double *y_best, int task,
                       double ei_jitter, const double *costs, double *best_vals, int64_t *best_idxs);

/* The schedule of cbo_gp_fit_sweep is measured, not tabulated: per shape (padded rows, padded candidates) the context
 * times the calls it is given anyway -- the plain sequence first (factorisation alone, sweep alone), then neighbouring
 * splits of the sweep between the pipeline under the factorisation and the closing launch -- and keeps the fastest
 * (cbo_api.hip, schedule_choose / schedule_report; results are the same bits whatever the schedule).  This call writes
 * what was measured and chosen, one text line per shape, into buf (NUL-terminated, truncated to cap; buf may be NULL)
 * and returns the number of shapes still exploring (0 = all settled), or a CBO_ERR_* code (negative as they are).  Every
 * line also says what the shape's last call ran ("last call ran pairs P group G", P = -1: the plain sequence).
 * No reference counterpart: the reference's loop (src/CBO.py:143-173) has no device schedule to choose. */
int cbo_schedule_report(cbo_ctx *ctx, char *buf, int64_t cap);

/* jitchol's ladder walked by the ranks of a communicator side by side (cbo_with_oop_amd/sharding.py, fit_over_ranks).
 * GPy's util.linalg.jitchol (reached from src/GaussianProcessFactory.py:57-73 through GPRegression) tries the plain
 * factorisation, then mean(diag)*1e-6 of jitter, x10 per retry, five retries at most, and keeps the FIRST level that
 * goes through.  With the posterior replicated on G ranks every rank repeats that walk (at config 4: a failed 26 ms
 * attempt, then the 31 ms one, on all eight GPUs).  Instead rank r tries ONE level: the levels below the one expected
 * to succeed get one rank each, all other ranks try the expected level; one small all-gather later every rank knows
 * the lowest level that went through -- the same answer as the sequential walk -- and the ranks that tried a failing
 * level receive the factor from the ones that hold it.
 *   cbo_gp_fit_level: one level (0 = plain, k = the k-th retry's jitter).  *status = 1: factored, the model is fitted
 *     with tries = level; 0: not positive definite at this level; -1: non-positive diagonal entries (jitchol's
 *     "not pd: non-positive diagonal elements", raised before its first retry).  level > 5: CBO_ERR_NOT_PD.
 *   cbo_comm_gather_i64: one integer from every rank, in rank order (out[world]).
 *   cbo_comm_share_factor: called by every rank with the same lists; the needers receive the factor at `level` in
 *     |owners| row slices, one from each owner (ncclSend / ncclRecv in one group), and adopt it. */
int cbo_gp_fit_level(cbo_gp *gp, int level, int *status, double *jitter_out);
/* (cbo_comm_gather_i64 and cbo_comm_share_factor are declared with the communicator, below) */

/* One whole trial of the reference's loop in one call -- what CBO.intervene() (src/CBO.py:143-173) does between two
 * observations, for callers whose models are small enough that three calls' worth of host glue would cost as much as
 * the device work: (1) the model of the set intervened on last takes its new data, as src/CBO.py:224-235
 * (update_gaussian_process_of_last_intervention) / src/Monitor.py:160 (set_data) give it: gps[refit_set] <- (n, X, y,
 * prior mean / variance at X or NULL), left unfitted; refit_set < 0 skips this; (2) cbo_acq_sweep_sets over all pairs
 * (src/CBO.py:237-260); (3) cbo_argmax_sets over the S winners (src/CBO.py:269-277) into *chosen_out.  Errors as those
 * calls; on an error the outputs are unspecified and the model may already hold the new data. */
int cbo_trial_step(int n_sets, cbo_gp *const *gps, cbo_cands *const *cands, int refit_set, int64_t n,
                   const double *X, const double *y, const double *prior_mean_X, const double *prior_var_X,
                   const double *y_best, int task, double ei_jitter, const double *costs, double *best_vals,
                   int64_t *best_idxs, int *chosen_out);

/* Host-buffer convenience form of the same call (uploads Xs first). */
int cbo_acq_sweep_host(cbo_gp *gp, int64_t m, const double *Xs, const double *prior_mean_s,
                       const double *prior_var_s, double y_best, int task, double ei_jitter,
                       double cost, double *acq_out, double *best_val, int64_t *best_idx);

/* src/CBO.py:269-277 select_next_intervention: first index of the maximum over exploration sets.
 * Host-side (S <= 25). */
int cbo_argmax_sets(const double *ys, int s, int *idx_out);

/* Reduce (best_val, best_idx) pairs gathered from all candidate shards (one per GPU) to the global
 * winner with the same tie rule; pure host arithmetic on 16 B per rank (cbo_comm_argmax gathers and calls it). */
int cbo_argmax_pairs(const double *vals, const int64_t *idxs, int n, double *best_val,
                     int64_t *best_idx);

/* ---- arg-max exchange across GPUs ------------------------------------------------------
 * The candidate grid shards over the GPUs of a node (contiguous blocks, index_offset of cbo_cands_create), the
 * posterior is replicated, and the one exchange step is 16 bytes per rank: (best acquisition value, best GLOBAL
 * candidate index), all-gathered over RCCL (xGMI) and reduced identically on every rank with the tie rule of
 * cbo_argmax_pairs -- RCCL has no MAXLOC.  The reference is a single process (src/CBO.py:269-277 picks over sets);
 * this is the cross-GPU counterpart of that pick.  librccl.so.1 is dlopen'ed on first use (CBO_HIP_RCCL_LIB
 * overrides the name); no PyTorch involved.  Every RCCL failure returns CBO_ERR_COMM with RCCL's message.
 *
 * One process per GPU: rank 0 calls cbo_comm_unique_id (128 bytes), the launcher's side channel (a file, MPI_Bcast,
 * a TCP store) hands the bytes to the other ranks, every rank calls cbo_comm_init_rank on its own context.
 * One process driving G devices: cbo_comm_init_all fills out[0..n) (rank i on ctxs[i]); use cbo_comm_argmax_all,
 * which issues the G collectives as one group. */
typedef struct cbo_comm cbo_comm;
#define CBO_COMM_ID_BYTES 128
int cbo_comm_unique_id(void *id_out /* CBO_COMM_ID_BYTES */);
int cbo_comm_init_rank(cbo_ctx *ctx, int world, int rank, const void *id /* CBO_COMM_ID_BYTES */, cbo_comm **out);
int cbo_comm_init_all(int n, cbo_ctx *const *ctxs, cbo_comm **out /* n handles */);
void cbo_comm_destroy(cbo_comm *comm);
int cbo_comm_size(const cbo_comm *comm, int *world_out, int *rank_out);
/* This rank's (val, global idx) in, the global winner out (identical on every rank).  A rank whose shard is empty
 * passes idx = INT64_MAX.  Blocking (the exchange runs on the communicator's own stream). */
int cbo_comm_argmax(cbo_comm *comm, double val, int64_t idx, double *best_val, int64_t *best_idx);
int cbo_comm_argmax_all(int n, cbo_comm *const *comms, const double *vals, const int64_t *idxs, double *best_val,
                        int64_t *best_idx);
/* max over the ranks of one double (the slowest rank's time of a benchmark); doubles as a barrier */
int cbo_comm_max_f64(cbo_comm *comm, double value, double *max_out);
int cbo_comm_barrier(cbo_comm *comm);
/* the ladder walked side by side: see cbo_gp_fit_level above */
int cbo_comm_gather_i64(cbo_comm *comm, int64_t value, int64_t *out);
int cbo_comm_share_factor(cbo_comm *comm, cbo_gp *gp, int level, const int *owners, int n_owners,
                          const int *needers, int n_needers);
/* The needer's side of cbo_comm_share_factor with device copies in the place of ncclRecv: dst (same data and
 * hyper-parameters as src, same context) takes src's factor at `level` in the n_owners row slices the owners would send
 * and adopts it (fitted, tries = level).  For tests on a one-GPU box, where the transfer between ranks cannot run; no
 * reference counterpart (the reference's jitchol, reached from src/GaussianProcessFactory.py:57-73, is one process). */
int cbo_gp_take_factor_slices(cbo_gp *dst, cbo_gp *src, int level, int n_owners);

/* ---- Monte-Carlo interventional target -----------------------------------------
 * Replaces compute_interventions (src/utils_functions/graph_functions.py:48-77): the mean of the target node
 * over num_samples draws of sample_from_model (:8-27) on the mutilated model of intervene_dict (:30-45).
 *
 * The model is an additive structural equation model listed in evaluation order (the reference's
 * OrderedDict order): node k takes
 *     value_k = sum_t  c_t * g_t(a_t * value[parent_t])   [ + eps[eps_index_k] ]
 * terms added left to right and the noise last, g in {x, x^2, exp, cos, sin}.  The closed-form SEM the
 * reference ships has this shape (src/graphs/impl/CompleteGraph.py:57-97).  An intervened node takes its
 * intervention value instead.  The noise matrix eps (n_samples x n_eps, row-major: one row per draw, exactly
 * the `randn(len(model))` vectors the reference draws after np.random.seed(seed)) is generated by the caller
 * with numpy's legacy stream and stays resident on the device, so every call sees the reference's draws. */
#define CBO_SEM_MAX_NODES 16
#define CBO_SEM_MAX_TERMS 64
enum cbo_sem_fn { CBO_FN_ID = 0, CBO_FN_SQUARE = 1, CBO_FN_EXP = 2, CBO_FN_COS = 3, CBO_FN_SIN = 4 };
typedef struct cbo_sem_spec {
    int n_nodes;
    int eps_index[CBO_SEM_MAX_NODES];       /* column of eps added to node k, or -1 */
    int term_begin[CBO_SEM_MAX_NODES + 1];  /* terms of node k are [term_begin[k], term_begin[k+1]) */
    int term_parent[CBO_SEM_MAX_TERMS];     /* index of an EARLIER node */
    int term_fn[CBO_SEM_MAX_TERMS];         /* enum cbo_sem_fn */
    double term_a[CBO_SEM_MAX_TERMS];
    double term_c[CBO_SEM_MAX_TERMS];
} cbo_sem_spec;
typedef struct cbo_sem cbo_sem;

int cbo_sem_create(cbo_ctx *ctx, const cbo_sem_spec *spec, int64_t n_samples, int n_eps,
                   const double *eps /* n_samples * n_eps, row-major */, cbo_sem **out);
void cbo_sem_destroy(cbo_sem *sem);
/* mean_out[i] = mean over the draws of node `target` under do(iv_nodes[j] = values[i*n_iv + j], j < n_iv).
 * n_iv may be 0 (observational mean); m interventions are evaluated in one launch. */
int cbo_sem_target(cbo_sem *sem, int target, int64_t m, int n_iv, const int *iv_nodes,
                   const double *values /* m * n_iv */, double *mean_out /* m */);

/* ---- hardware self-test ------------------------------------------------------------------------
 * Runs the fp64 MFMA lane-layout check (asymmetric operands) used by tests; returns CBO_OK when the
 * v_mfma_f64_16x16x4_f64 A/B/C maps this library assumes hold on the device. */
int cbo_selftest_mfma(cbo_ctx *ctx, double *max_abs_err_out);

#ifdef __cplusplus
}
#endif
#endif /* CBO_HIP_H */
