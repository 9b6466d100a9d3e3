// Posterior epilogue + causal Expected Improvement / cost + arg-max for gfx950.
//
// Restates, element by element:
//   GPy Posterior._raw_predict / GP.predict:  var = clip(Kdiag - q, 1e-15, inf) (+ noise), mean = mu + m(X*)
//   CausalRBF.Kdiag                           /root/reference/src/utils_functions/causal_kernels.py:64-79
//   CausalExpectedImprovement.evaluate        /root/reference/src/utils_functions/causal_acquisition_functions.py:27-43
//   get_standard_normal_pdf_cdf               ... :77-88   (scipy.stats.norm.pdf / .cdf = cephes ndtr)
//   emukit Quotient with Cost.evaluate        /root/reference/src/utils_functions/cost_functions.py:11-17
//   top-1 selection (argsort()[::-1][:1] generalised; lowest index wins ties, NaN is maximal as in
//   numpy.argmax)                             /root/reference/src/utils_functions/causal_optimizer.py:52-55
// HBM-bound and tiny next to the TRSM: 2-4 doubles in, up to 3 out per candidate.  The arg-max is a
// wavefront shuffle reduction, one partial per workgroup, then one 256-thread finishing block.
#include "cbo_device.h"

#pragma clang fp contract(off)

namespace cbo {

__device__ __forceinline__ void block_argmax(double v, int64_t i, double *out_v, int64_t *out_i)
{
    __shared__ double sv[4];
    __shared__ int64_t si[4];
    wave_argmax(v, i);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sv[wave] = v; si[wave] = i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bv = sv[0];
        int64_t bi = si[0];
        for (int w = 1; w < 4; ++w)
            if (better(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
        *out_v = bv;
        *out_i = bi;
    }
}

constexpr int64_t kNoIndex = INT64_MAX;

// Two consecutive candidates per lane and iteration (16-byte loads and stores), the next iteration's operands requested
// before this one's arithmetic: with one candidate per lane and the load at the top of the loop body the pass was bound by
// memory LATENCY -- 32 KB in flight per CU -- and ran at 0.29 of the HBM roofline whatever the arithmetic cost (round 5: halving
// the transcendental work changed nothing until the loads were decoupled).
// CAUSAL: the candidates carry a prior mean / variance; MV: mean and / or variance are written out (the plain sweep asks for
// neither: 24 registers less, one more wave per SIMD)
template <bool CAUSAL, bool MV>
__global__ __launch_bounds__(256) void acq_kernel(const double *__restrict__ q, const double *__restrict__ mu,
                                                  const double *__restrict__ pm, const double *__restrict__ pv,
                                                  int64_t m, AcqParams p, double *__restrict__ mean_out,
                                                  double *__restrict__ var_out, double *__restrict__ acq_out,
                                                  double *__restrict__ part_val, int64_t *__restrict__ part_idx,
                                                  int64_t index_offset)
{
    double bv = -INFINITY;
    int64_t bi = kNoIndex;
    constexpr bool causal = CAUSAL;
    if (!MV) { mean_out = nullptr; var_out = nullptr; }
    const int64_t stride = 2 * (int64_t)gridDim.x * blockDim.x;
    // the workgroup's first candidate of an iteration is uniform (scalar registers), the lane's share of the address a constant
    // 16 tid bytes: every load and store is "scalar base + 32-bit lane offset" and the loop advances scalars only
    int64_t cu = 2 * (int64_t)blockIdx.x * blockDim.x;
    const unsigned lane2 = 2 * threadIdx.x;
    const int64_t span = 2 * (int64_t)blockDim.x;
    // operands of the pair at cu + lane2 (the second of an odd tail: a copy of the first, never stored)
    auto fetch = [&](int64_t base, d2 &q2, d2 &mu2, d2 &pm2, d2 &pv2) __attribute__((always_inline)) {
        if (base + span <= m) {                                  // (uniform) every lane has its two candidates
            q2 = *reinterpret_cast<const d2 *>(q + base + lane2);
            mu2 = *reinterpret_cast<const d2 *>(mu + base + lane2);
            if (causal) {
                pm2 = *reinterpret_cast<const d2 *>(pm + base + lane2);
                pv2 = *reinterpret_cast<const d2 *>(pv + base + lane2);
            }
            return;
        }
        const int64_t at = base + lane2;
        if (at + 1 < m) {
            q2 = *reinterpret_cast<const d2 *>(q + at);
            mu2 = *reinterpret_cast<const d2 *>(mu + at);
            if (causal) {
                pm2 = *reinterpret_cast<const d2 *>(pm + at);
                pv2 = *reinterpret_cast<const d2 *>(pv + at);
            }
        } else if (at < m) {
            q2 = d2{q[at], q[at]};
            mu2 = d2{mu[at], mu[at]};
            if (causal) {
                pm2 = d2{pm[at], pm[at]};
                pv2 = d2{pv[at], pv[at]};
            }
        }
    };
    auto store = [&](int64_t base, const d2 &mean2, const d2 &var2, const d2 &acq2) __attribute__((always_inline)) {
        const int64_t c = base + lane2;
        if (base + span <= m) {                                  // (uniform)
            if (mean_out) *reinterpret_cast<d2 *>(mean_out + base + lane2) = mean2;
            if (var_out) *reinterpret_cast<d2 *>(var_out + base + lane2) = var2;
            if (p.want_ei && acq_out) *reinterpret_cast<d2 *>(acq_out + base + lane2) = acq2;
        } else if (c + 1 < m) {
            if (mean_out) *reinterpret_cast<d2 *>(mean_out + c) = mean2;
            if (var_out) *reinterpret_cast<d2 *>(var_out + c) = var2;
            if (p.want_ei && acq_out) *reinterpret_cast<d2 *>(acq_out + c) = acq2;
        } else if (c < m) {
            if (mean_out) mean_out[c] = mean2[0];
            if (var_out) var_out[c] = var2[0];
            if (p.want_ei && acq_out) acq_out[c] = acq2[0];
        }
    };
    // Every memory operation of an iteration is issued in one place, right behind the iteration's only wait: the operands of
    // the NEXT iteration and the results of the PREVIOUS one.  Loads and stores share one counter on gfx9 and the compiler
    // waits for zero whenever both kinds are pending -- with the stores at the end of the body and the loads at its top (the
    // form this loop had until round 5) that wait sat right behind the freshly issued prefetch and took its whole latency,
    // every iteration, plus the stores': the pass ran at "memory floor + arithmetic" (77 + 48 us at 2^24 candidates) instead
    // of the larger of the two.  Now whatever the wait covers was issued a whole iteration of arithmetic earlier.
    d2 qn = {0.0, 0.0}, mun = {0.0, 0.0}, pmn = {0.0, 0.0}, pvn = {0.0, 0.0};
    d2 mean_done = {0.0, 0.0}, var_done = {0.0, 0.0}, acq_done = {0.0, 0.0};
    fetch(cu, qn, mun, pmn, pvn);
    for (int64_t done = -1; cu < m; done = cu, cu += stride) {
        d2 q2 = qn, mu2 = mun, pm2 = pmn, pv2 = pvn;
        // (the operands are in their registers before anything below is issued; nothing memory moves across this line)
        asm volatile("" : "+v"(q2), "+v"(mu2), "+v"(pm2), "+v"(pv2) : : "memory");
        if (done >= 0) store(done, mean_done, var_done, acq_done);
        fetch(cu + stride, qn, mun, pmn, pvn);
        const bool full = cu + span <= m;                        // uniform
        const int64_t c = cu + lane2;
        const bool one = full || c < m, two = full || c + 1 < m;
        d2 mean2, var2, acq2 = {0.0, 0.0};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            double mean, var;
            posterior_of(q2[e], mu2[e], causal ? pm2[e] : 0.0, causal ? pv2[e] : 0.0, causal, p, mean, var);
            mean2[e] = mean;
            var2[e] = var;
        }
        if (p.want_ei) {
            // (one candidate after the other: the two in lockstep -- shared coefficients, independent chains -- needed selects
            // where this takes branches, 190 instead of 160 vector instructions per candidate, and 14 more registers: 113 us
            // against 107 at 2^24 candidates, profiles/r05_ei_pass_ab.txt)
            acq2[0] = acquisition_of(mean2[0], var2[0], p);
            acq2[1] = acquisition_of(mean2[1], var2[1], p);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const double acq = acq2[e];
                const int64_t gi = c + e + index_offset;
                // (a candidate below the lane's best cannot win: better() is only asked when it is not -- one compare on the
                // usual path instead of better()'s NaN classification of both sides)
                if ((e == 0 ? one : two) && !(acq < bv) && better(acq, gi, bv, bi)) { bv = acq; bi = gi; }
            }
        }
        mean_done = mean2;
        var_done = var2;
        acq_done = acq2;
    }
    // (cu has run past m by whole strides: the last iteration's results, if there was one)
    if (cu - stride >= 2 * (int64_t)blockIdx.x * blockDim.x) store(cu - stride, mean_done, var_done, acq_done);
    if (p.want_ei) block_argmax(bv, bi, &part_val[blockIdx.x], &part_idx[blockIdx.x]);
}

// best_val / best_idx may be pinned host memory (the sweep's epilogue writes the winner where the host reads it: no copy
// operation behind the kernel); status_src -> status_dst carries a factorisation's status word the same way.
// (Round 5 folded this reduction into acq_kernel -- the workgroup that draws the last ticket of an agent-scope counter
// reduces the partials -- and took it out again: 2048 tickets on one address serialise, a pass over 2^20 candidates went
// from 21 to 35 us and one over 2^24 gained nothing.)
__global__ __launch_bounds__(256) void argmax_final_kernel(const double *__restrict__ part_val,
                                                           const int64_t *__restrict__ part_idx, int n,
                                                           double *__restrict__ best_val, int64_t *__restrict__ best_idx,
                                                           const int *__restrict__ status_src, int *__restrict__ status_dst)
{
    if (status_src && threadIdx.x == 0) *status_dst = __builtin_nontemporal_load(status_src);
    double bv = -INFINITY;
    int64_t bi = kNoIndex;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        if (better(part_val[i], part_idx[i], bv, bi)) { bv = part_val[i]; bi = part_idx[i]; }
    block_argmax(bv, bi, best_val, best_idx);
}

// Mean of consecutive runs (np.mean over the observed rows of one intervention, DoCalculus.py:59-60): one
// wave per run, 16-byte-free strided reads (runs are short: N_obs ~ 100..1000), shuffle reduction.
__global__ __launch_bounds__(256) void group_mean_kernel(const double *__restrict__ in, int64_t n_groups,
                                                         int64_t group, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const double *p = in + g * group;
    double s = 0.0;
    for (int64_t i = lane; i < group; i += 64) s += p[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[g] = s / (double)group;
}

void launch_group_mean(hipStream_t s, const double *in, int64_t n_groups, int64_t group, double *out)
{
    hipLaunchKernelGGL(group_mean_kernel, dim3((unsigned)((n_groups + 3) / 4)), dim3(256), 0, s, in, n_groups, group, out);
}

// Prediction gradients (GPy GP.predictive_gradients as reached through emukit's
// get_prediction_gradients, /root/reference/src/utils_functions/causal_acquisition_functions.py:54):
//   dk(x_i, x*)/dx*_k = k(x_i, x*) (x_ik - x*_k) / l_k^2      (stationary RBF part; GPy's Stationary.gradients_X,
//   which CausalRBF inherits, ignores the rank-1 causal term -- SURVEY.md §A.2)
//   dmean[c][k] = sum_i alpha_i dk/dx*_k,    dvar[c][k] = -2 sum_i w_ic dk/dx*_k,   w_c = Ky^-1 k*(x_c)
// for a whole batch of points: W holds the w_c as COLUMNS of the workspace the backward substitution wrote, in
// reversed row order (W[n_pad-1-i][c] = w_ic, see reversed_factor_kernel).  One workgroup per 64 points: lane = point
// (coalesced W reads), the four waves split the observations, LDS reduction at the end.
// Coordinates are the (ARD-)scaled SoA copies: u = x / l, so dk/dx*_k = k (u_ik - u*_k) * inv_l_k.
__global__ __launch_bounds__(256) void pred_gradients_kernel(const double *__restrict__ xs, int64_t ldx, int64_t n,
                                                             int64_t n_pad, const double *__restrict__ cs, int64_t ldc,
                                                             int64_t c_begin, int64_t m, int d, double variance,
                                                             double iso_inv_l, const double *__restrict__ inv_ls,
                                                             const double *__restrict__ alpha,
                                                             const double *__restrict__ W, int64_t ldw,
                                                             double *__restrict__ dmean, double *__restrict__ dvar)
{
    __shared__ double red[3][2 * CBO_MAX_DIM][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t cl = (int64_t)blockIdx.x * 64 + lane;             // column of this chunk's workspace
    const int64_t c = c_begin + cl;                                 // candidate
    double xc[CBO_MAX_DIM], gm[CBO_MAX_DIM], gv[CBO_MAX_DIM];
#pragma unroll
    for (int k = 0; k < CBO_MAX_DIM; ++k) {
        xc[k] = (k < d) ? cs[(int64_t)k * ldc + c] : 0.0;
        gm[k] = 0.0;
        gv[k] = 0.0;
    }
    for (int64_t i = g; i < n; i += 4) {
        double diff[CBO_MAX_DIM];
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < CBO_MAX_DIM; ++k) {
            diff[k] = (k < d) ? (xs[(int64_t)k * ldx + i] - xc[k]) * iso_inv_l : 0.0;   // scaled difference
            r2 += diff[k] * diff[k];
        }
        const double kv = variance * exp_nonpositive(-0.5 * r2);
        const double a = alpha[i] * kv, b = W[(n_pad - 1 - i) * ldw + cl] * kv;
#pragma unroll
        for (int k = 0; k < CBO_MAX_DIM; ++k) {
            const double gk = diff[k] * (inv_ls ? inv_ls[k] : iso_inv_l);
            gm[k] += a * gk;
            gv[k] += b * gk;
        }
    }
    if (g > 0) {
#pragma unroll
        for (int k = 0; k < CBO_MAX_DIM; ++k) {
            red[g - 1][k][lane] = gm[k];
            red[g - 1][CBO_MAX_DIM + k][lane] = gv[k];
        }
    }
    __syncthreads();
    if (g == 0 && c < m) {
#pragma unroll
        for (int k = 0; k < CBO_MAX_DIM; ++k) {
            if (k < d) {
                dmean[c * d + k] = ((gm[k] + red[0][k][lane]) + red[1][k][lane]) + red[2][k][lane];
                dvar[c * d + k] = -2.0 * (((gv[k] + red[0][CBO_MAX_DIM + k][lane]) + red[1][CBO_MAX_DIM + k][lane]) +
                                          red[2][CBO_MAX_DIM + k][lane]);
            }
        }
    }
}

void launch_pred_gradients(hipStream_t s, const PointSet &X, int64_t n_pad, const PointSet &C, int64_t c_begin,
                           int64_t cols, int64_t m, const KernelHyper &h, const double *inv_ls_dev, const double *alpha,
                           const double *W, int64_t ldw, double *dmean, double *dvar)
{
    // ARD: coordinates are already divided by l_k (iso factor 1, per-dimension 1/l_k for the chain rule);
    // isotropic: raw coordinates, one 1/l for both
    const double iso = h.ard ? 1.0 : 1.0 / h.lengthscale;
    hipLaunchKernelGGL(pred_gradients_kernel, dim3((unsigned)(cols / 64)), dim3(256), 0, s, X.xs, X.ld, X.n, n_pad, C.xs,
                       C.ld, c_begin, m, X.d, h.variance, iso, h.ard ? inv_ls_dev : nullptr, alpha, W, ldw, dmean, dvar);
}

int acq_blocks_for(int64_t m)
{
    int64_t b = (m + 511) / 512;                      // two candidates per lane
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

void launch_acq(hipStream_t s, const double *q, const double *mu, const double *pm, const double *pv, int64_t m,
                const AcqParams &p, double *mean_out, double *var_out, double *acq_out, double *part_val,
                int64_t *part_idx, int64_t index_offset, int n_blocks)
{
    const bool causal = pv != nullptr, mv = mean_out || var_out;
    auto kernel = causal ? (mv ? acq_kernel<true, true> : acq_kernel<true, false>) : (mv ? acq_kernel<false, true> : acq_kernel<false, false>);
    hipLaunchKernelGGL(kernel, dim3(n_blocks), dim3(256), 0, s, q, mu, pm, pv, m, p, mean_out, var_out, acq_out, part_val,
                       part_idx, index_offset);
}

void launch_argmax_final(hipStream_t s, const double *part_val, const int64_t *part_idx, int n, double *best_val,
                         int64_t *best_idx, const int *status_src, int *status_dst)
{
    hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(256), 0, s, part_val, part_idx, n, best_val, best_idx, status_src,
                       status_dst);
}

// ---- append-only trial step --------------------------------------------------------------------------------
// A CBO trial adds ONE observation to one set (src/Monitor.py:148-160) while hyper-parameters and candidate grid
// stay put.  Appending row/column n to Ky leaves the first n rows of its factor and of V = L^-1 K* unchanged:
//     U[0:n, n] = l = L^-1 k(X, x_new),   U[n, n] = d = sqrt(k(x_new, x_new) + noise + 1e-8 - l^T l),
//     z_n = (y_new - m(x_new) - l^T z) / d,      V[n, :] = (k(x_new, X*) - l^T V[0:n, :]) / d,
//     q += V[n, :]^2,   mu += V[n, :] z_n
// l, l^T l and l^T z come out of an ordinary sweep with x_new as the only candidate; the kernels below commit
// the new column and extend a resident V by one row.

// column n of U, the new diagonal entry, z_n, the new point's coordinates; one thread per row
__global__ void append_commit_kernel(double *__restrict__ A, int64_t lda, int64_t n, int64_t n_pad,
                                     const double *__restrict__ l_src, int64_t ld_src, double d, double zn,
                                     double *__restrict__ z, double *__restrict__ lvec, int dims, double *__restrict__ xs,
                                     int64_t ldx, double *__restrict__ sq, double *__restrict__ sv, double *__restrict__ pm,
                                     double *__restrict__ pv, const double *__restrict__ pxs, int64_t ldp,
                                     const double *__restrict__ psq, const double *__restrict__ psv, double pm_new,
                                     double pv_new, double *__restrict__ y, double y_new)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double li = l_src[i * ld_src];
        A[i * lda + n] = li;
        lvec[i] = li;
    } else if (i == n) {
        A[n * lda + n] = d;
        A[n * lda + n_pad] = zn;             // the rhs column carries z
        z[n] = zn;
        y[n] = y_new;
        for (int k = 0; k < dims; ++k) xs[(int64_t)k * ldx + n] = pxs[(int64_t)k * ldp];
        sq[n] = psq[0];
        if (sv) { sv[n] = psv[0]; pm[n] = pm_new; pv[n] = pv_new; }
    }
}

// Column n%16 of the inverse of the 16x16 upper-triangular diagonal tile that contains row n (row-major, what the
// strip kernel reads).  The other columns do not change: column b of the inverse of an upper-triangular matrix
// depends on its leading (b+1)x(b+1) block only, and the columns right of n are still identity padding.
__global__ void append_tile_inverse_kernel(const double *__restrict__ A, int64_t lda, int64_t tile, int col,
                                           double *__restrict__ invDt)
{
    __shared__ double T[16][17];
    const int t = threadIdx.x;                 // 256 threads: element (a, b)
    const int a = t >> 4, b = t & 15;
    T[a][b] = (b >= a) ? A[(tile * 16 + a) * lda + tile * 16 + b] : 0.0;
    __syncthreads();
    if (t == col) {                            // column t of the inverse by back substitution
        double x[16];
        for (int r = 0; r < 16; ++r) x[r] = 0.0;
        x[t] = 1.0 / T[t][t];
        for (int r = t - 1; r >= 0; --r) {
            double s = 0.0;
            for (int c = r + 1; c <= t; ++c) s = fma(T[r][c], x[c], s);
            x[r] = -s / T[r][r];
        }
        for (int r = 0; r < 16; ++r) invDt[tile * 256 + r * 16 + t] = x[r];
    }
}

// partial[r][j] = sum over the r-th slice of rows i < n of l_i V[i][j]
__global__ __launch_bounds__(256) void append_row_partial_kernel(const double *__restrict__ V, int64_t ldv, int64_t n,
                                                                 int rows_per_slice, const double *__restrict__ lvec,
                                                                 int64_t m_pad, double *__restrict__ partial)
{
    __shared__ double ls[512];
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * rows_per_slice;
    const int64_t i1 = (i0 + rows_per_slice < n) ? i0 + rows_per_slice : n;
    double s = 0.0;
    for (int64_t base = i0; base < i1; base += 512) {
        __syncthreads();
        for (int t = threadIdx.x; t < 512; t += 256) ls[t] = (base + t < i1) ? lvec[base + t] : 0.0;
        __syncthreads();
        const int cnt = (int)((i1 - base < 512) ? i1 - base : 512);
        if (j < m_pad)
            for (int t = 0; t < cnt; ++t) s = fma(ls[t], V[(base + t) * ldv + j], s);
    }
    if (j < m_pad) partial[(int64_t)blockIdx.y * m_pad + j] = s;
}

__global__ void append_row_final_kernel(const double *__restrict__ partial, int slices, int64_t m_pad,
                                        const double *__restrict__ krow, double d, double zn, double *__restrict__ Vrow,
                                        double *__restrict__ q, double *__restrict__ mu)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m_pad) return;
    double s = 0.0;
    for (int r = 0; r < slices; ++r) s += partial[(int64_t)r * m_pad + j];
    const double v = (krow[j] - s) / d;
    Vrow[j] = v;
    q[j] = fma(v, v, q[j]);
    mu[j] = fma(v, zn, mu[j]);
}

void launch_append_commit(hipStream_t s, double *A, int64_t lda, int64_t n, int64_t n_pad, const double *l_src,
                          int64_t ld_src, double d, double zn, double *z, double *lvec, PointSet &X, const PointSet &P,
                          double pm_new, double pv_new, double *y, double y_new, double *invDt)
{
    hipLaunchKernelGGL(append_commit_kernel, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, s, A, lda, n, n_pad,
                       l_src, ld_src, d, zn, z, lvec, X.d, X.xs, X.ld, X.sq, X.sv, X.pm, X.pv, P.xs, P.ld, P.sq, P.sv,
                       pm_new, pv_new, y, y_new);
    hipLaunchKernelGGL(append_tile_inverse_kernel, dim3(1), dim3(256), 0, s, A, lda, n / 16, (int)(n % 16), invDt);
}

int append_row_slices(int64_t n) { const int64_t s = (n + 127) / 128; return (int)(s < 1 ? 1 : (s > 64 ? 64 : s)); }

void launch_append_row(hipStream_t s, double *V, int64_t ldv, int64_t n, const double *lvec, int64_t m_pad,
                       const double *krow, double d, double zn, double *partial, double *q, double *mu)
{
    const int slices = append_row_slices(n);
    const int rows_per_slice = (int)(((n + slices - 1) / slices + 7) / 8 * 8);
    if (n > 0)
        hipLaunchKernelGGL(append_row_partial_kernel, dim3((unsigned)((m_pad + 255) / 256), slices), dim3(256), 0, s, V, ldv,
                           n, rows_per_slice, lvec, m_pad, partial);
    hipLaunchKernelGGL(append_row_final_kernel, dim3((unsigned)((m_pad + 255) / 256)), dim3(256), 0, s, partial,
                       n > 0 ? slices : 0, m_pad, krow, d, zn, V + n * ldv, q, mu);
}

}  // namespace cbo
