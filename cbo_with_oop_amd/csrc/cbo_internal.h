// Internal declarations shared by the kernel translation units and the C-ABI host code.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts, v_mfma_f64_16x16x4_f64, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "cbo_hip.h"

namespace cbo {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
typedef __attribute__((address_space(3))) void *lds_ptr_t;
// LDS-DMA (global_load_lds_dwordx4): 64 lanes x 16 B land at (wave-uniform LDS byte address) + lane * 16;
// the global address is per lane.  Issued through inline asm with M0 written in the same statement
// (cdna_hip_programming.md 5.7) so that hipcc does not track it: with the builtin next to ds_reads every
// LDS-read wait degrades to lgkmcnt(0).  The caller counts completion by hand (s_waitcnt vmcnt + barrier).
__device__ __forceinline__ void glds16(const double *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_byte_address(const void *p)
{
    return (unsigned)(unsigned long)(lds_ptr_t)p;
}
#endif

// ---- data layout in HBM ------------------------------------------------------
// * Points: SoA, coordinate k of point i at xs[k * ld + i]; squared norms sq[i]; sqrt(v(x_i)) sv[i].
// * Ky and its Cholesky factor share one row-major buffer A[n_pad][lda].  Only the UPPER triangle is
//   meaningful: Ky = U^T U, U[k][i] = L[i][k].  A right-hand-side strip of 64 columns sits at column
//   n_pad; its first column carries r = y - m(X) and is overwritten by z = L^-1 r during the
//   factorisation.  n_pad = round_up(n, 128); padded rows/cols form an identity block.
// * invDt[b] (b = 16-row block index) holds inv(U_bb) row-major, i.e. invDt[b][k][i] = inv(L_bb)[i][k].
// * V workspace [n_pad][ldv]: K(X, X*) for a chunk of candidates, overwritten by L^-1 K*.
constexpr int kPadN = 128;       // n_pad granularity
constexpr int kStrip = 64;       // candidate columns per workgroup strip
constexpr int kRhsCols = 64;     // width of the right-hand-side strip appended to A
constexpr int kLdExtra = 16;     // extra doubles per row so consecutive rows fall in different channels

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// shared with cbo_comm.hip (the context is defined in cbo_api.hip)
int set_error(int code, const std::string &msg);          // records the message for cbo_last_error(), returns code
hipStream_t ctx_stream(cbo_ctx *c);
int ctx_device(cbo_ctx *c);
// the factor of a model for cbo_comm_share_factor: where it lives; whether this rank holds it at `level` of the ladder;
// its adoption by a rank that has received it
int gp_factor_view(cbo_gp *g, double **A, int64_t *lda, int64_t *n_pad, double **invDt, cbo_ctx **ctx);
bool gp_is_fitted_at(const cbo_gp *g, int level);
int gp_adopt_received_factor(cbo_gp *g, int level);

// GPy constants (GPy 1.10.0 exact_gaussian_inference.py / posterior.py); see oracle/gp_oracle.py.
constexpr double kGpyDiagJitter = 1e-8;
constexpr double kGpyVarClip = 1e-15;

struct PointSet {            // device-resident SoA point set
    double *xs = nullptr;    // [d][ld]
    double *sq = nullptr;    // [ld]
    double *sv = nullptr;    // [ld] sqrt(prior variance) or nullptr
    double *pm = nullptr;    // [ld] prior mean or nullptr
    double *pv = nullptr;    // [ld] prior variance (raw) or nullptr
    int64_t n = 0, ld = 0;
    int d = 0;
};

struct KernelHyper {
    double variance;
    double lengthscale;      // isotropic lengthscale (distance divided after sqrt); 1.0 when ard
    int ard;                 // inputs were pre-scaled per dimension
    int zero_diag;           // GPy X2=None shortcut: r2[i][i] = 0
};

// ---- kernel launchers (kernels_*.hip) -----------------------------------------------------------
// AoS (n,d) raw -> SoA (optionally divided by per-dim lengthscale), squared norms, sqrt(v).
void launch_prep_points(hipStream_t s, const double *raw_aos, int64_t n, int d, const double *ls_dev /*d or null*/,
                        const double *pv_raw /*n or null*/, double *xs, int64_t ld, double *sq, double *sv);

void launch_prep_points_staged(hipStream_t s, const double *stage, int64_t n, int d, const double *ls_dev, bool has_prior,
                               double *raw, double *y, double *pm, double *pv, double *xs, int64_t ld, double *sq,
                               double *sv);

// K(X,X) + diag into the upper 64x64 tiles of A (identity on the padding), and the rhs strip.
void launch_kxx(hipStream_t s, const PointSet &X, const KernelHyper &h, double diag_add, double jitter,
                double *A, int64_t lda, int64_t n_pad);
void launch_zero_pair(hipStream_t s, double *a, double *b, int64_t n);       // a[0:n] = b[0:n] = 0
// (zero, zero_count: ints the launch clears on the way -- the factorisation's status word and counters, cholesky_info_ints())
void launch_rhs(hipStream_t s, const double *y, const double *pm, int64_t n, double *A, int64_t lda, int64_t n_pad,
                int *zero = nullptr, int zero_count = 0);
// K(X, X*) for candidate columns [c_begin, c_begin + m_pad) into V (rows >= n are zero).
void launch_kstar(hipStream_t s, const PointSet &X, const PointSet &C, int64_t c_begin, int64_t m_pad,
                  const KernelHyper &h, double *V, int64_t ldv, int64_t n_pad);

// ---- fp32 sweep (kernels_f32.hip; BASELINE.json configs[4]) -----------------------------------------------
// The fit stays fp64; factor, diagonal inverses and z are down-converted once per fit into a layout whose 16-row
// groups are row-permuted (physical row 4 (k & 3) + (k >> 2) = logical row k) and padded to n32 = round_up(n_pad, 256).
constexpr int kPadN32 = 256;
void launch_factor_to_f32(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt, float *Uf,
                          int64_t ldu, float *invF, int64_t n32);
// K(X, X*) in fp64 arithmetic, stored as fp32; the posterior mean's kernel part mu = K*^T alpha (GPy's own formula)
// is formed on the way from the unrounded values: mu_part is (n32 / 64) * m_pad doubles of workspace
void launch_kstar_f32(hipStream_t s, const PointSet &X, const PointSet &C, int64_t c_begin, int64_t m_pad,
                      const KernelHyper &h, float *V, int64_t ldv, int64_t n32, const double *alpha, double *mu_part,
                      double *mu);
// V <- L^-1 V in fp32 (64-column strips), q[c] = sum V^2 accumulated in fp64
void launch_trsm_strips_f32(hipStream_t s, const float *U, int64_t ldu, const float *invDt, float *V, int64_t ldv,
                            int64_t n32, int64_t m_pad, double *q);
int run_mfma_f32_selftest(hipStream_t s, double *max_err);
int f32_debug_mask();      // the F32_DBG mask this library was built with (0 in a product build)

// Recursive blocked Cholesky (upper) of A[0:n_pad, 0:n_pad] incl. forward solve of the rhs strip.
// Sweep pipelined with the factorisation: as soon as a pair of 128-row panels of U is final, the strip kernel
// solves those rows of V on `stream` (q, mu accumulate) and trsm_update_kernel folds them into the rows below,
// while the factorisation continues on its own streams.  V holds K(X, X*) on entry (assembled on `stream`).
struct SweepPipe {
    hipStream_t stream;                  // in-panel solves + the update of the next panel pair's rows
    hipStream_t bulk;                    // the updates of everything below that
    double *V;
    int64_t ldv, m_pad;
    double *zvec;                        // contiguous copy of z, written panel by panel by the diagonal kernel
    double *q, *mu;                      // zeroed on `stream` by the caller
    int chunk_blocks;                    // row blocks per workgroup of the update kernel
    bool half_lds;                       // 16-row stages (two workgroups per CU) for every kernel of the pipeline
    bool lower_tri;                      // the right-hand sides are lower triangular (identity: V = L^-1), so rows
                                         // [r0, r0+klen) only reach columns < r0+klen: launch just those strips
    int group;                           // G >= 2: updates in groups of G pairs (K = 256 G on `bulk`), see sweep_pipe_pair
    int lead = 0;                        // pairs that go alone AHEAD of the first group (their bulk update, K = 256, can start
                                         // as soon as they are solved: the bulk stream does not idle until a whole group is)
    int tail_begin;                      // rows from here on (a multiple of 256; n_pad = none) are left to ONE
                                         // left-looking strip launch once the factorisation is complete
    std::vector<hipEvent_t> *events;     // factorisation -> sweep dependencies, grown on demand
    void (*mark)(void *user, hipStream_t st, int begin, double flops);   // optional: around every sweep launch (timers)
    void *user;
};
// info_dev: 1 + kCholFlagSlots ints (status word, then one publication counter per 128-row panel)
constexpr int kCholFlagSlots = 1024;
constexpr int kCholFusedTimeout = -2147483647 - 1;     // status word when a strip of a fused launch gave up waiting
// info_zeroed: the caller's launch_rhs has cleared the first cholesky_info_ints(n_pad) ints of info_dev on the same stream
void launch_cholesky(hipStream_t s, hipStream_t side, std::vector<hipEvent_t> &events, double *A, int64_t lda,
                     int64_t n_pad, double *invDt, int *info_dev, const SweepPipe *pipe = nullptr, bool info_zeroed = false);
inline int cholesky_info_ints(int64_t n_pad)
{
    const int np = (int)(n_pad / 128);
    return 2 * np <= kCholFlagSlots ? 1 + 2 * np : 1;
}
// > 0: launch_cholesky of this thread uses that panel form (CBO_HIP_PANEL_FORM's values) whatever the environment says
void set_panel_form_override(int form);
// pair p of the pipelined sweep: rows [r0, r0 + klen) of the factor are final on stream `chain`
void sweep_pipe_pair(const SweepPipe &pipe, hipStream_t chain, const double *A, int64_t lda, const double *invDt,
                     int64_t n_pad, int p, int r0, int klen);
void sweep_pipe_tail(const SweepPipe &pipe, hipStream_t chain, const double *A, int64_t lda, const double *invDt,
                     int64_t n_pad, int pairs_done);
// alpha = U^-1 z  (z = first rhs column of A).
void launch_backsolve(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt, double *alpha);
// out = L^-1 w for one contiguous n_pad vector (w is destroyed): the forward counterpart, one launch per block
void launch_forward_vec(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt, double *w,
                        double *out);
void launch_gather_column(hipStream_t s, const double *V, int64_t ldv, int64_t n_pad, double *dst);
// one-launch forms (a chain of workgroups, one per 128-row block); false = not applicable, nothing was launched.  `info`
// is the model's status word: a give-up leaves kCholFusedTimeout in it and the caller
// repeats the solve with the per-block launches (launch_backsolve_vec / launch_forward_vec) after resetting it.
bool launch_backsolve_chain(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt,
                            const double *src, int64_t src_stride, double *work, double *out, int *info);
bool launch_forward_chain(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt, const double *w,
                          double *out, int *info);
void launch_backsolve_vec(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt,
                          const double *src, int64_t src_stride, double *work, double *out);
// Gradients of the posterior mean and variance w.r.t. the prediction inputs (GPy predictive_gradients), batched:
// dmean[c][k] = sum_i alpha_i dk(x_i, x*_c)/dx*_k, dvar[c][k] = -2 sum_i w_ic dk(x_i, x*_c)/dx*_k (RBF part only) for the
// `cols` workspace columns that hold candidates [c_begin, c_begin + cols); W = Ky^-1 K* in reversed row order.
void launch_pred_gradients(hipStream_t s, const PointSet &X, int64_t n_pad, const PointSet &C, int64_t c_begin,
                           int64_t cols, int64_t m, const KernelHyper &h, const double *inv_ls_dev, const double *alpha,
                           const double *W, int64_t ldw, double *dmean, double *dvar);
// the factor for the backward substitution through the forward strip kernel (kernels_kmat.hip), row reversal, dot pair
void launch_reversed_factor(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt, double *T,
                            int64_t ldt, double *invT);
void launch_reverse_rows(hipStream_t s, const double *V, int64_t ldv, int64_t n_pad, int64_t cols, double *W, int64_t ldw);
void launch_dot2(hipStream_t s, const double *a, const double *b, int64_t n, double *out2);
void launch_sum(hipStream_t s, const double *a, int64_t n, double *out);
void launch_expand_interventions(hipStream_t s, const double *observed, int64_t n_obs, int d, const double *values, int n_iv,
                                 const int *iv_index, int64_t m, double *raw);

// V <- L^-1 V on m_pad columns (64-column strips); optional q[c] = sum_i V[i][c]^2, mu[c] = sum_i V[i][c] z[i].
void launch_trsm_strips(hipStream_t s, const double *U, int64_t ldu, const double *invDt, double *V, int64_t ldv,
                        int64_t n, int64_t m_pad, const double *z, double *q, double *mu, bool accumulate = false,
                        bool half_lds = false);
// C[i0_begin:i0_end, :] -= U[k0:k0+klen, i0_begin:i0_end]^T V[k0:k0+klen, :]  (C and V share the workspace V)
void launch_trsm_update(hipStream_t s, const double *U, int64_t ldu, double *V, int64_t ldv, int k0, int klen,
                        int i0_begin, int i0_end, int64_t m_pad, int chunk_blocks, bool half_lds = true);
// the same kernel as a plain GEMM update with its own output:
// C[i0_begin:i0_end, 0:m_pad] -= U[k0:k0+klen, i0_begin:i0_end]^T V[k0:k0+klen, 0:m_pad]; upper_only skips the
// workgroups that lie entirely below the diagonal
void launch_gemm_update(hipStream_t s, const double *U, int64_t ldu, const double *V, int64_t ldv, double *C,
                        int64_t ldc, int k0, int klen, int i0_begin, int i0_end, int64_t m_pad, int chunk_blocks,
                        bool half_lds, bool upper_only, const int *skip_if = nullptr);

// One (model, candidate set) pair of a multi-set sweep of small models (kernels_chol.hip, small_sets_kernel): every
// pointer is device memory; filled on the host per call and uploaded as an array.
struct cbo_small_set {
    const double *xs, *sq, *sv, *pm, *y;               // model: SoA points (ld), |x|^2, sqrt(v) or null, m(X) or null, targets
    const double *cxs, *csq, *csv, *cpm, *cpv;         // candidates (scaled SoA, ld = cld), prior closures at them or null
    int64_t ld, cld, m, index_offset;
    int n, d, zero_diag, task;
    double variance, lengthscale, noise_var, diag_add, y_best, ei_jitter, cost;
    int ard, pad_;                                     // inputs pre-scaled per dimension (lengthscale gradient per dimension)
    // cbo_trial_step: the model's NEW data have not been uploaded -- they sit in pinned (device-mapped) memory as
    // [X (n,d) | y (n) | prior mean (n) | prior variance (n)] and every workgroup of the set prepares the points from there
    // itself (the arithmetic of prep_points_staged_kernel); the set's first workgroup also fills the resident copies
    // (raw, y, pm, pv and xs, sq, sv above).  nullptr: the resident copies are current.
    const double *stage, *stage_ls;                    // stage_ls: per-dimension lengthscales (ARD) or nullptr
    double *raw, *pv;
};
struct cbo_small_result {
    double best_val;
    int64_t best_idx;
    int info;                                          // first non-positive pivot (1-based) or 0
    int seq;                                           // the call's sequence number, stored last: the record is complete
};
size_t small_sets_scratch_doubles(int n_sets, int blocks_per_set);
// one-launch likelihood + gradients of a small model (kernels_chol.hip): terms[0] = variance sum, terms[1 + k] =
// lengthscale sums per dimension, then z^T z, sum log diag(U), alpha^T alpha, tr(Ky^-1)
constexpr int kSmallLmlTerms = 1 + CBO_MAX_DIM + 4;
struct cbo_small_lml_result {
    double terms[kSmallLmlTerms];
    int info, seq;
};
size_t small_lml_scratch_doubles();
void launch_small_lml(hipStream_t s, const cbo_small_set &st, double *scratch, int *info, cbo_small_lml_result *out, int seq);
// sets / out may be pinned host memory (device-mapped): the kernel then reads the descriptors and writes the results
// across the host link itself and the call needs no copy operation (the host may poll out[].seq instead of
// synchronising the stream); info and ticket (device, n_sets ints each) must be zero on entry and are zero again afterwards
void launch_small_sets(hipStream_t s, const cbo_small_set *sets, int n_sets, int blocks_per_set, double *scratch,
                       double *part_val, int64_t *part_idx, int *info, int *ticket, cbo_small_result *out, int seq);

struct AcqParams {
    double variance, noise_var, y_best, ei_jitter, cost;
    int task, include_noise, want_ei;
};
// var = clip(kss - q) (+ noise), mean = mu + m(X*), acq = +-EI / cost; per-block arg-max partials.
void launch_acq(hipStream_t s, const double *q, const double *mu, const double *pm, const double *pv, int64_t m,
                const AcqParams &p, double *mean_out, double *var_out, double *acq_out, double *part_val,
                int64_t *part_idx, int64_t index_offset, int n_blocks);
void launch_argmax_final(hipStream_t s, const double *part_val, const int64_t *part_idx, int n, double *best_val,
                         int64_t *best_idx, const int *status_src = nullptr, int *status_dst = nullptr);
int acq_blocks_for(int64_t m);
// out[g] = mean of in[g*group .. (g+1)*group)
void launch_group_mean(hipStream_t s, const double *in, int64_t n_groups, int64_t group, double *out);

// out[0] = sum z_i^2, out[1] = sum log U_ii over the n_pad rows (padding contributes 0)
void launch_lml_terms(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *z, double *out2);
// likelihood gradients: out[0] = sum M k, out[1 + k] = sum M k ((x_ik - x_jk)/l_k)^2 over all i, j < n with
// M = alpha alpha^T + negW; partial: lml_grad_tiles(n_pad) * (1 + d) doubles
int lml_grad_tiles(int64_t n_pad);
void launch_lml_grad(hipStream_t s, const PointSet &X, const KernelHyper &h, const double *alpha, const double *negW,
                     int64_t ldw, int64_t n_pad, double *partial, double *out);
void launch_set_identity(hipStream_t s, double *V, int64_t ldv, int64_t n_pad);
void launch_gather_diag(hipStream_t s, const double *A, int64_t lda, int64_t n, double *diag);
void launch_export_lower(hipStream_t s, const double *A, int64_t lda, int64_t n, double *L_rowmajor);
void launch_export_sym(hipStream_t s, const double *A, int64_t lda, int64_t n, double *K_rowmajor);

int run_mfma_selftest(hipStream_t s, double *max_err);

// append-only trial step (kernels_acq.hip): commit column n of the factor (l from column 0 of l_src), the new
// diagonal entry, z_n, the point's coordinates and the diagonal tile's inverse; extend a resident V by row n
void launch_append_commit(hipStream_t s, double *A, int64_t lda, int64_t n, int64_t n_pad, const double *l_src,
                          int64_t ld_src, double d, double zn, double *z, double *lvec, PointSet &X, const PointSet &P,
                          double pm_new, double pv_new, double *y, double y_new, double *invDt);
int append_row_slices(int64_t n);
void launch_append_row(hipStream_t s, double *V, int64_t ldv, int64_t n, const double *lvec, int64_t m_pad,
                       const double *krow, double d, double zn, double *partial, double *q, double *mu);

// Monte-Carlo target of an additive SEM: mean_out[i] = mean over draws of node `target` under intervention i.
// partial: m * sem_partial_blocks(n_draws) doubles of workspace.
int sem_partial_blocks(int64_t n_draws);
void launch_sem_target(hipStream_t s, const cbo_sem_spec &spec, const double *eps_cm, int64_t n_draws, int target,
                       int64_t m, int n_iv, const int *iv_nodes_host, const int *iv_nodes, const double *iv_values,
                       double *partial, double *mean_out);

}  // namespace cbo
