// Kernel-matrix assembly for gfx950: K(X,X) (+ diagonal, identity padding) and K(X,X*).
//
// Arithmetic restates GPy's Stationary._unscaled_dist / _scaled_dist and the reference's
// CausalRBF.K (/root/reference/src/utils_functions/causal_kernels.py:45-62) in GPy's operation order
// (GEMM-trick squared distance from the same |x|^2 sums, clip at 0, scale, exp, rank-1 causal term); the
// only deviations from the numpy path are the exp() implementation, the dot-product association inside
// BLAS, and the sqrt -> /l -> square round trip that GPy takes on the way to r^2 (skipped: a couple of
// ulp).  Contraction into FMAs is switched off in this file except where the reference itself goes
// through BLAS (the dot product).
//
// Roofline: HBM-write bound.  One 64x64 output tile per 256-thread workgroup, tile coordinates
// staged in LDS, 16-byte coalesced stores (each wave writes 2 rows x 512 B per instruction).
// Measured (scripts/kxx_roofline.py): 16384 points, d=3 -> 1.08 GB of upper tiles in 0.22 ms = 4.9 TB/s
// (61 % of the 8 TB/s HBM peak); at 4096 points the 34 us launch is ramp/tail dominated (2.0 TB/s).
#include "cbo_device.h"

#pragma clang fp contract(off)

namespace cbo {

// ------------------------------------------------------------------------------------------------
// AoS -> SoA, optional per-dimension scaling (GPy ARD: X / lengthscale), squared norms exactly as
// numpy's np.sum(np.square(X), 1) forms them (sequential for d < 8, the 8-way pairwise tree at d = 8).
__global__ __launch_bounds__(256) void prep_points_kernel(const double *__restrict__ raw, int64_t n, int d,
                                                          const double *__restrict__ ls,
                                                          const double *__restrict__ pv,
                                                          double *__restrict__ xs, int64_t ld,
                                                          double *__restrict__ sq, double *__restrict__ sv)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ld) return;
    double x[CBO_MAX_DIM];
#pragma unroll
    for (int k = 0; k < CBO_MAX_DIM; ++k) x[k] = 0.0;
    if (i < n) {
#pragma unroll
        for (int k = 0; k < CBO_MAX_DIM; ++k)
            if (k < d) {
                double v = raw[i * d + k];
                if (ls) v = v / ls[k];
                x[k] = v;
            }
    }
#pragma unroll
    for (int k = 0; k < CBO_MAX_DIM; ++k)
        if (k < d) xs[(int64_t)k * ld + i] = x[k];
    double s;
    if (d == 8) {
        double r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = __dmul_rn(x[k], x[k]);
        s = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                      __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
    } else {
        s = 0.0;
#pragma unroll
        for (int k = 0; k < CBO_MAX_DIM; ++k)
            if (k < d) s = __dadd_rn(s, __dmul_rn(x[k], x[k]));
    }
    sq[i] = s;
    if (sv) sv[i] = (i < n && pv) ? sqrt(pv[i]) : 0.0;
}

// The same preparation for a small point set whose host arrays sit in ONE pinned (device-mapped) staging buffer
// [X (n,d) | y (n) | prior mean (n) | prior variance (n)]: the kernel reads them across the host link itself and
// also fills the resident device copies (raw AoS, y, prior) -- one launch, no copy operation, nothing to wait for.
__global__ __launch_bounds__(256) void prep_points_staged_kernel(const double *__restrict__ stage, int64_t n, int d,
                                                                 const double *__restrict__ ls, int has_prior,
                                                                 double *__restrict__ raw, double *__restrict__ y,
                                                                 double *__restrict__ pm, double *__restrict__ pv,
                                                                 double *__restrict__ xs, int64_t ld,
                                                                 double *__restrict__ sq, double *__restrict__ sv)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ld) return;
    double x[CBO_MAX_DIM];
#pragma unroll
    for (int k = 0; k < CBO_MAX_DIM; ++k) x[k] = 0.0;
    double pvi = 0.0;
    if (i < n) {
#pragma unroll
        for (int k = 0; k < CBO_MAX_DIM; ++k)
            if (k < d) {
                double v = stage[i * d + k];
                raw[i * d + k] = v;
                if (ls) v = v / ls[k];
                x[k] = v;
            }
        y[i] = stage[n * d + i];
        if (has_prior) {
            pm[i] = stage[n * d + n + i];
            pvi = stage[n * d + 2 * n + i];
            pv[i] = pvi;
        }
    }
#pragma unroll
    for (int k = 0; k < CBO_MAX_DIM; ++k)
        if (k < d) xs[(int64_t)k * ld + i] = x[k];
    double s;
    if (d == 8) {
        double r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = __dmul_rn(x[k], x[k]);
        s = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                      __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
    } else {
        s = 0.0;
#pragma unroll
        for (int k = 0; k < CBO_MAX_DIM; ++k)
            if (k < d) s = __dadd_rn(s, __dmul_rn(x[k], x[k]));
    }
    sq[i] = s;
    if (sv) sv[i] = (i < n && has_prior) ? sqrt(pvi) : 0.0;
}

void launch_prep_points_staged(hipStream_t s, const double *stage, int64_t n, int d, const double *ls_dev, bool has_prior,
                               double *raw, double *y, double *pm, double *pv, double *xs, int64_t ld, double *sq,
                               double *sv)
{
    hipLaunchKernelGGL(prep_points_staged_kernel, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, s, stage, n, d, ls_dev,
                       has_prior ? 1 : 0, raw, y, pm, pv, xs, ld, sq, sv);
}

void launch_prep_points(hipStream_t s, const double *raw_aos, int64_t n, int d, const double *ls_dev,
                        const double *pv_raw, double *xs, int64_t ld, double *sq, double *sv)
{
    const int threads = 256;
    const int blocks = (int)((ld + threads - 1) / threads);
    hipLaunchKernelGGL(prep_points_kernel, dim3(blocks), dim3(threads), 0, s, raw_aos, n, d, ls_dev, pv_raw, xs, ld,
                       sq, sv);
}

// (one kernel-matrix element: kernel_value<D> of cbo_device.h)
struct KmatArgs {
    const double *rx; int64_t ldr; const double *rsq; const double *rsv;   // row points (observations)
    const double *cx; int64_t ldc; const double *csq; const double *csv;   // column points
    int64_t n_rows, n_cols;        // valid rows / cols (beyond: padding)
    int64_t col_begin;             // first column point index of this launch (K* chunks)
    double *out; int64_t ldo;
    double variance, lengthscale, diag_add, jitter;
    int symmetric, zero_diag;
    int sym_tiles;                 // symmetric: tiles per side (grid = upper triangle, linearised)
    float *out32;                  // OUT32: fp32 output (K* of the fp32 sweep), rows of every 16-row group permuted
    const double *alpha;           // OUT32: woodbury vector; mu_part[row tile][column] = sum over the tile's 64 rows of
    double *mu_part;               //        K*[i][c] alpha[i], formed from the fp64 values before they are rounded
    int64_t ld_part;
};

// OUT32: the same fp64 arithmetic, rounded to fp32 on store, into the row-permuted layout of kernels_f32.hip
// (physical row 4 (k & 3) + (k >> 2) of a 16-row group holds logical row k).
template <int D, bool OUT32 = false>
__global__ __launch_bounds__(256) void kmat_tile_kernel(KmatArgs a)
{
    int tj = blockIdx.x, ti = blockIdx.y;
    if (a.symmetric) {
        // only the upper tiles of Ky are ever read: the grid is the nt (nt + 1) / 2 tiles of the upper triangle,
        // row by row (row ti starts at ti nt - ti (ti - 1) / 2)
        const int nt = a.sym_tiles;
        const int t = blockIdx.x;
        ti = (int)((2.0 * nt + 1.0 - sqrt((2.0 * nt + 1.0) * (2.0 * nt + 1.0) - 8.0 * (double)t)) * 0.5);
        while (ti > 0 && ti * nt - ti * (ti - 1) / 2 > t) --ti;               // guard the rounding of the root
        while ((ti + 1) * nt - (ti + 1) * ti / 2 <= t) ++ti;
        tj = ti + (t - (ti * nt - ti * (ti - 1) / 2));
    }
    __shared__ double sx[D][64], sy[D][64];
    __shared__ double sxq[64], syq[64], sxv[64], syv[64];
    __shared__ double sal[OUT32 ? 64 : 1], smu[OUT32 ? 8 * 64 : 1];
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)ti * 64, j0 = (int64_t)tj * 64;
    if (OUT32 && tid >= 128 && tid < 192) {
        const int64_t gi = i0 + (tid - 128);
        sal[tid - 128] = (gi < a.n_rows) ? a.alpha[gi] : 0.0;
    }
    double macc[2] = {0.0, 0.0};
    if (tid < 64) {
        const int64_t gi = i0 + tid;
        const bool in = gi < a.ldr;                    // fp32 sweep: rows are padded to 256, beyond the point set's ld
#pragma unroll
        for (int k = 0; k < D; ++k) sx[k][tid] = in ? a.rx[(int64_t)k * a.ldr + gi] : 0.0;
        sxq[tid] = in ? a.rsq[gi] : 0.0;
        sxv[tid] = (in && a.rsv) ? a.rsv[gi] : 0.0;
    } else if (tid < 128) {
        const int t = tid - 64;
        const int64_t gj = a.col_begin + j0 + t;
#pragma unroll
        for (int k = 0; k < D; ++k) sy[k][t] = a.cx[(int64_t)k * a.ldc + gj];
        syq[t] = a.csq[gj];
        syv[t] = a.csv ? a.csv[gj] : 0.0;
    }
    __syncthreads();
    const int tx = tid & 31, ty = tid >> 5;
    const bool causal = (a.rsv != nullptr) && (a.csv != nullptr);
    const double inv_l2 = 1.0 / (a.lengthscale * a.lengthscale);
    double yj[2][D];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < D; ++k) yj[c][k] = sy[k][2 * tx + c];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int ii = ty + 8 * rr;
        const int64_t gi = i0 + ii;
        double xi[D];
#pragma unroll
        for (int k = 0; k < D; ++k) xi[k] = sx[k][ii];
        d2 o;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int jj = 2 * tx + c;
            const int64_t gj = j0 + jj;                  // column within this launch's output
            const bool row_ok = gi < a.n_rows;
            const bool col_ok = (a.col_begin + gj) < a.n_cols;
            double v;
            if (a.symmetric) {
                if (row_ok && col_ok) {
                    v = kernel_value<D>(xi, yj[c], sxq[ii], syq[jj], a.variance, inv_l2,
                                        a.zero_diag && gi == gj);
                    if (causal) v = __dadd_rn(v, __dmul_rn(sxv[ii], syv[jj]));
                    if (gi == gj) {
                        v = __dadd_rn(v, a.diag_add);                 // Ky = K + (noise + 1e-8) I
                        if (a.jitter != 0.0) v = __dadd_rn(v, a.jitter);   // jitchol: A + jitter I
                    }
                } else {
                    v = (gi == gj) ? 1.0 : 0.0;                       // identity padding
                }
            } else {
                if (row_ok) {
                    v = kernel_value<D>(xi, yj[c], sxq[ii], syq[jj], a.variance, inv_l2, false);
                    if (causal) v = __dadd_rn(v, __dmul_rn(sxv[ii], syv[jj]));
                } else {
                    v = 0.0;                                          // padded observation rows
                }
            }
            o[c] = v;
        }
        if (OUT32) {
            macc[0] = __fma_rn(o[0], sal[ii], macc[0]);
            macc[1] = __fma_rn(o[1], sal[ii], macc[1]);
            const int64_t pr = (gi & ~(int64_t)15) + 4 * (gi & 3) + ((gi >> 2) & 3);
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 of;
            of[0] = (float)o[0];
            of[1] = (float)o[1];
            *reinterpret_cast<f2 *>(&a.out32[pr * a.ldo + j0 + 2 * tx]) = of;
        } else {
            *reinterpret_cast<d2 *>(&a.out[gi * a.ldo + j0 + 2 * tx]) = o;
        }
    }
    if (OUT32) {
        // GPy Posterior._raw_predict: mu = Kx^T woodbury_vector -- the tile's share of it, fixed summation order
        smu[ty * 64 + 2 * tx] = macc[0];
        smu[ty * 64 + 2 * tx + 1] = macc[1];
        __syncthreads();
        if (tid < 64) {
            double t = 0.0;
#pragma unroll
            for (int g = 0; g < 8; ++g) t = __dadd_rn(t, smu[g * 64 + tid]);
            a.mu_part[(int64_t)ti * a.ld_part + j0 + tid] = t;
        }
    }
}

template <int D>
static void launch_kmat_d(hipStream_t s, const KmatArgs &a, dim3 grid)
{
    if (a.out32) hipLaunchKernelGGL((kmat_tile_kernel<D, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((kmat_tile_kernel<D, false>), grid, dim3(256), 0, s, a);
}

static void launch_kmat(hipStream_t s, int d, const KmatArgs &a, dim3 grid)
{
    switch (d) {
        case 1: launch_kmat_d<1>(s, a, grid); break;
        case 2: launch_kmat_d<2>(s, a, grid); break;
        case 3: launch_kmat_d<3>(s, a, grid); break;
        case 4: launch_kmat_d<4>(s, a, grid); break;
        case 5: launch_kmat_d<5>(s, a, grid); break;
        case 6: launch_kmat_d<6>(s, a, grid); break;
        case 7: launch_kmat_d<7>(s, a, grid); break;
        default: launch_kmat_d<8>(s, a, grid); break;
    }
}

void launch_kxx(hipStream_t s, const PointSet &X, const KernelHyper &h, double diag_add, double jitter, double *A,
                int64_t lda, int64_t n_pad)
{
    KmatArgs a;
    a.rx = X.xs; a.ldr = X.ld; a.rsq = X.sq; a.rsv = X.sv;
    a.cx = X.xs; a.ldc = X.ld; a.csq = X.sq; a.csv = X.sv;
    a.n_rows = X.n; a.n_cols = X.n; a.col_begin = 0;
    a.out = A; a.ldo = lda;
    a.variance = h.variance; a.lengthscale = h.lengthscale; a.diag_add = diag_add; a.jitter = jitter;
    a.symmetric = 1; a.zero_diag = h.zero_diag; a.out32 = nullptr; a.alpha = nullptr; a.mu_part = nullptr; a.ld_part = 0;
    const int nt = (int)(n_pad / 64);
    a.sym_tiles = nt;
    launch_kmat(s, X.d, a, dim3((unsigned)(nt * (nt + 1) / 2), 1));
}

void launch_kstar(hipStream_t s, const PointSet &X, const PointSet &C, int64_t c_begin, int64_t m_pad,
                  const KernelHyper &h, double *V, int64_t ldv, int64_t n_pad)
{
    KmatArgs a;
    a.rx = X.xs; a.ldr = X.ld; a.rsq = X.sq; a.rsv = X.sv;
    a.cx = C.xs; a.ldc = C.ld; a.csq = C.sq; a.csv = (X.sv != nullptr) ? C.sv : nullptr;
    a.n_rows = X.n; a.n_cols = C.ld; a.col_begin = c_begin;   // all padded columns are computable
    a.out = V; a.ldo = ldv;
    a.variance = h.variance; a.lengthscale = h.lengthscale; a.diag_add = 0.0; a.jitter = 0.0;
    a.symmetric = 0; a.zero_diag = 0; a.sym_tiles = 0; a.out32 = nullptr; a.alpha = nullptr; a.mu_part = nullptr; a.ld_part = 0;
    launch_kmat(s, X.d, a, dim3((unsigned)(m_pad / 64), (unsigned)(n_pad / 64)));
}

// mu[c] = sum over the row tiles of mu_part[t][c] (fixed order)
__global__ __launch_bounds__(256) void colsum_parts_kernel(const double *__restrict__ part, int64_t ld, int tiles,
                                                           int64_t m, double *__restrict__ mu)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= m) return;
    double t = 0.0;
    for (int i = 0; i < tiles; ++i) t = __dadd_rn(t, part[(int64_t)i * ld + c]);
    mu[c] = t;
}

// K(X, X*) rounded to fp32 for the fp32 sweep: [n32][ldv] floats, rows permuted (kernels_f32.hip), rows >= n zero.
void launch_kstar_f32(hipStream_t s, const PointSet &X, const PointSet &C, int64_t c_begin, int64_t m_pad,
                      const KernelHyper &h, float *V, int64_t ldv, int64_t n32, const double *alpha, double *mu_part,
                      double *mu)
{
    KmatArgs a;
    a.rx = X.xs; a.ldr = X.ld; a.rsq = X.sq; a.rsv = X.sv;
    a.cx = C.xs; a.ldc = C.ld; a.csq = C.sq; a.csv = (X.sv != nullptr) ? C.sv : nullptr;
    a.n_rows = X.n; a.n_cols = C.ld; a.col_begin = c_begin;
    a.out = nullptr; a.ldo = ldv; a.out32 = V;
    a.alpha = alpha; a.mu_part = mu_part; a.ld_part = m_pad;
    a.variance = h.variance; a.lengthscale = h.lengthscale; a.diag_add = 0.0; a.jitter = 0.0;
    a.symmetric = 0; a.zero_diag = 0; a.sym_tiles = 0;
    launch_kmat(s, X.d, a, dim3((unsigned)(m_pad / 64), (unsigned)(n32 / 64)));
    hipLaunchKernelGGL(colsum_parts_kernel, dim3((unsigned)((m_pad + 255) / 256)), dim3(256), 0, s, mu_part, m_pad,
                       (int)(n32 / 64), m_pad, mu);
}

// ------------------------------------------------------------------------------------------------
// Right-hand-side strip: column n_pad of A carries r = y - m(X) (GPy: Y - mean_function.f(X)); the
// other 63 columns of the strip and the padded rows are zero.
// The launch also zeroes the factorisation's status word and publication counters (`zero`, `zero_count` ints): a launch of
// their own cost the chain ~10 us of kernel and boundary in front of its first panel.
__global__ __launch_bounds__(256) void rhs_kernel(const double *__restrict__ y, const double *__restrict__ pm,
                                                  int64_t n, double *__restrict__ A, int64_t lda, int64_t n_pad,
                                                  int *__restrict__ zero, int zero_count)
{
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < zero_count; i += blockDim.x) zero[i] = 0;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pad * kRhsCols) return;
    const int64_t i = idx / kRhsCols;
    const int c = (int)(idx % kRhsCols);
    double v = 0.0;
    if (c == 0 && i < n) v = pm ? __dadd_rn(y[i], -pm[i]) : y[i];
    A[i * lda + n_pad + c] = v;
}

void launch_rhs(hipStream_t s, const double *y, const double *pm, int64_t n, double *A, int64_t lda, int64_t n_pad,
                int *zero, int zero_count)
{
    const int64_t total = n_pad * kRhsCols;
    hipLaunchKernelGGL(rhs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, y, pm, n, A, lda, n_pad, zero,
                       zero ? zero_count : 0);
}

// q = mu = 0 ahead of a sweep that accumulates into them (one launch; the runtime's hipMemsetAsync is a fill kernel plus
// several microseconds of host time per call)
__global__ __launch_bounds__(256) void zero_pair_kernel(double *__restrict__ a, double *__restrict__ b, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = 0.0; b[i] = 0.0; }
}

void launch_zero_pair(hipStream_t s, double *a, double *b, int64_t n)
{
    if (n > 0) hipLaunchKernelGGL(zero_pair_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, n);
}

// ------------------------------------------------------------------------------------------------
__global__ void gather_diag_kernel(const double *__restrict__ A, int64_t lda, int64_t n, double *__restrict__ diag)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) diag[i] = A[i * lda + i];
}

void launch_gather_diag(hipStream_t s, const double *A, int64_t lda, int64_t n, double *diag)
{
    hipLaunchKernelGGL(gather_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, A, lda, n, diag);
}

// Terms of GPy's log marginal likelihood from the factor: sum z^2 (= r^T Ky^-1 r) and sum log U_ii.
__global__ __launch_bounds__(256) void lml_terms_kernel(const double *__restrict__ A, int64_t lda, int64_t n_pad,
                                                        const double *__restrict__ z, double *__restrict__ out2)
{
    __shared__ double s1[256], s2[256];
    double a = 0.0, b = 0.0;
    for (int64_t i = threadIdx.x; i < n_pad; i += 256) {
        const double zi = z[i];
        a = __fma_rn(zi, zi, a);
        b = __dadd_rn(b, log(A[i * lda + i]));
    }
    s1[threadIdx.x] = a;
    s2[threadIdx.x] = b;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            s1[threadIdx.x] = __dadd_rn(s1[threadIdx.x], s1[threadIdx.x + st]);
            s2[threadIdx.x] = __dadd_rn(s2[threadIdx.x], s2[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = s1[0]; out2[1] = s2[0]; }
}

void launch_lml_terms(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *z, double *out2)
{
    hipLaunchKernelGGL(lml_terms_kernel, dim3(1), dim3(256), 0, s, A, lda, n_pad, z, out2);
}

// ---- log marginal likelihood gradients ----------------------------------------------------
// dL/dtheta = 1/2 sum_ij M_ij dK_ij/dtheta with M = alpha alpha^T - Ky^-1 (GPy: dL_dK, passed to
// kern.update_gradients_full).  For the RBF part k = s2 exp(-r2/2), r2 = sum_k (x_ik - x_jk)^2 / l_k^2:
//     dK/ds2 = k / s2,    dK/dl_k = k (x_ik - x_jk)^2 / l_k^3
// so one pass over the upper 64x64 tiles accumulates  S0 = sum M k  and  S_k = sum M k ((x_ik - x_jk)/l_k)^2
// (off-diagonal elements twice), per-tile partials in a fixed order -> reproducible.  negW holds -Ky^-1 (the GEMM
// kernel subtracts).  Coordinates are the resident ones: pre-scaled per dimension when ard, raw otherwise.
template <int D>
__global__ __launch_bounds__(256) void lml_grad_tile_kernel(const double *__restrict__ xs, int64_t ldx, int64_t n,
                                                            const double *__restrict__ alpha,
                                                            const double *__restrict__ negW, int64_t ldw,
                                                            double variance, double inv_l2_iso,
                                                            const double *__restrict__ sv /* sqrt(v(X)) or null */,
                                                            double *__restrict__ partial /* [tiles][1 + D] */)
{
    const int tj = blockIdx.x, ti = blockIdx.y;
    const int tile = ti * gridDim.x + tj;
    __shared__ double sx[D][64], sy[D][64], sa[64], sb[64], svx[64], svy[64];
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double s[1 + D];
#pragma unroll
    for (int k = 0; k <= D; ++k) s[k] = 0.0;
    if (tj >= ti) {
        const int64_t i0 = (int64_t)ti * 64, j0 = (int64_t)tj * 64;
        if (tid < 64) {
#pragma unroll
            for (int k = 0; k < D; ++k) sx[k][tid] = xs[(int64_t)k * ldx + i0 + tid];
            sa[tid] = (i0 + tid < n) ? alpha[i0 + tid] : 0.0;
            svx[tid] = (sv && i0 + tid < n) ? sv[i0 + tid] : 0.0;
        } else if (tid < 128) {
            const int t = tid - 64;
#pragma unroll
            for (int k = 0; k < D; ++k) sy[k][t] = xs[(int64_t)k * ldx + j0 + t];
            sb[t] = (j0 + t < n) ? alpha[j0 + t] : 0.0;
            svy[t] = (sv && j0 + t < n) ? sv[j0 + t] : 0.0;
        }
        __syncthreads();
        const int tx = tid & 63, ty = tid >> 6;                  // column tx, rows ty, ty+4, ...
        const int64_t gj = j0 + tx;
        for (int rr = 0; rr < 16; ++rr) {
            const int ii = ty + 4 * rr;
            const int64_t gi = i0 + ii;
            if (gi >= n || gj >= n || gj < gi) continue;
            double r2 = 0.0, d2k[D];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double df = sx[k][ii] - sy[k][tx];
                d2k[k] = df * df * inv_l2_iso;
                r2 += d2k[k];
            }
            const double kv = variance * exp_nonpositive(-0.5 * r2);
            const double m = (gi == gj ? 1.0 : 2.0) * (sa[ii] * sb[tx] + negW[gi * ldw + gj]);
            const double mk = m * kv;
            // the variance gradient contracts dL_dK with the kernel's own K(X, X) (GPy Stationary.update_gradients_full:
            // sum(self.K(X, X2) * dL_dK) / variance) -- for CausalRBF that K carries the rank-1 causal term too
            // (/root/reference/src/utils_functions/causal_kernels.py:45-62, 153-155); the lengthscale gradient goes
            // through dK_dr, which is the stationary part alone (:84-85)
            s[0] += mk + m * (svx[ii] * svy[tx]);
#pragma unroll
            for (int k = 0; k < D; ++k) s[1 + k] = fma(mk, d2k[k], s[1 + k]);
        }
    }
#pragma unroll
    for (int k = 0; k <= D; ++k) {
        __syncthreads();
        red[tid] = s[k];
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) red[tid] = __dadd_rn(red[tid], red[tid + st]);
            __syncthreads();
        }
        if (tid == 0) partial[(int64_t)tile * (1 + D) + k] = red[0];
    }
}

__global__ void lml_grad_final_kernel(const double *__restrict__ partial, int tiles, int terms, double *__restrict__ out)
{
    const int k = threadIdx.x;
    if (k >= terms) return;
    double t = 0.0;
    for (int i = 0; i < tiles; ++i) t = __dadd_rn(t, partial[(int64_t)i * terms + k]);
    out[k] = t;
}

int lml_grad_tiles(int64_t n_pad) { const int nt = (int)(n_pad / 64); return nt * nt; }

void launch_lml_grad(hipStream_t s, const PointSet &X, const KernelHyper &h, const double *alpha, const double *negW,
                     int64_t ldw, int64_t n_pad, double *partial, double *out)
{
    const int nt = (int)(n_pad / 64);
    const dim3 grid(nt, nt);
    const double inv_l2 = h.ard ? 1.0 : 1.0 / (h.lengthscale * h.lengthscale);
#define CBO_LAUNCH_GRAD(D)                                                                                     \
    hipLaunchKernelGGL(lml_grad_tile_kernel<D>, grid, dim3(256), 0, s, X.xs, X.ld, X.n, alpha, negW, ldw, h.variance, \
                       inv_l2, X.sv, partial)
    switch (X.d) {
        case 1: CBO_LAUNCH_GRAD(1); break;
        case 2: CBO_LAUNCH_GRAD(2); break;
        case 3: CBO_LAUNCH_GRAD(3); break;
        case 4: CBO_LAUNCH_GRAD(4); break;
        case 5: CBO_LAUNCH_GRAD(5); break;
        case 6: CBO_LAUNCH_GRAD(6); break;
        case 7: CBO_LAUNCH_GRAD(7); break;
        default: CBO_LAUNCH_GRAD(8); break;
    }
#undef CBO_LAUNCH_GRAD
    hipLaunchKernelGGL(lml_grad_final_kernel, dim3(1), dim3(16), 0, s, partial, nt * nt, 1 + X.d, out);
}

// V = identity (n_pad x n_pad) in a workspace with leading dimension ldv: the right-hand sides of L^-1.
__global__ void set_identity_kernel(double *__restrict__ V, int64_t ldv, int64_t n_pad)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pad * n_pad) return;
    const int64_t i = idx / n_pad, j = idx % n_pad;
    V[i * ldv + j] = (i == j) ? 1.0 : 0.0;
}

void launch_set_identity(hipStream_t s, double *V, int64_t ldv, int64_t n_pad)
{
    hipLaunchKernelGGL(set_identity_kernel, dim3((unsigned)((n_pad * n_pad + 255) / 256)), dim3(256), 0, s, V, ldv, n_pad);
}

// ---- backward substitution through the forward kernel (prediction gradients for whole grids) -------
// W = U^-1 V (U upper triangular) is a forward substitution in reversed index order: with P the reversal, P U P is
// lower triangular, so the strip kernel (which wants the TRANSPOSED lower factor, row-major, upper part) can be handed
//     T[k][i] = U[n-1-i][n-1-k]   (k <= i)
// and the diagonal-tile inverses of T, which are those of U read backwards in both indices.  Built once per fit,
// on first use.  64x64 tiles through LDS so that reads and writes are both row-contiguous.
__global__ __launch_bounds__(256) void reversed_factor_kernel(const double *__restrict__ A, int64_t lda, int64_t n,
                                                              double *__restrict__ T, int64_t ldt)
{
    __shared__ double tile[64][65];
    const int64_t ti = blockIdx.y, tj = blockIdx.x;            // tile of T: rows 64 ti.., columns 64 tj..
    if (tj < ti) return;                                        // T is upper triangular (by tiles)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    // T[64 ti + a][64 tj + b] = U[n-1-64 tj-b][n-1-64 ti-a]: read U rows (b) contiguous in a
    for (int b = ty; b < 64; b += 4) {
        const int64_t ur = n - 1 - (64 * tj + b), uc = n - 1 - (64 * ti + tx);
        tile[b][tx] = (uc >= ur) ? A[ur * lda + uc] : 0.0;      // tile[b][a] with a = tx
    }
    __syncthreads();
    for (int a = ty; a < 64; a += 4) T[(64 * ti + a) * ldt + 64 * tj + tx] = tile[tx][a];
}

__global__ __launch_bounds__(256) void reversed_inverses_kernel(const double *__restrict__ invDt, int64_t n_tiles,
                                                                double *__restrict__ invT)
{
    const int64_t b = blockIdx.x;
    const int k = threadIdx.x >> 4, i = threadIdx.x & 15;
    invT[b * 256 + k * 16 + i] = invDt[(n_tiles - 1 - b) * 256 + (15 - i) * 16 + (15 - k)];
}

void launch_reversed_factor(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt, double *T,
                            int64_t ldt, double *invT)
{
    const unsigned nt = (unsigned)(n_pad / 64);
    hipLaunchKernelGGL(reversed_factor_kernel, dim3(nt, nt), dim3(256), 0, s, A, lda, n_pad, T, ldt);
    hipLaunchKernelGGL(reversed_inverses_kernel, dim3((unsigned)(n_pad / 16)), dim3(256), 0, s, invDt, n_pad / 16, invT);
}

// W[n-1-r][c] = V[r][c]: the right-hand sides of the reversed system (16-byte accesses, one row pair per block row)
__global__ __launch_bounds__(256) void reverse_rows_kernel(const double *__restrict__ V, int64_t ldv, int64_t n,
                                                           int64_t cols, double *__restrict__ W, int64_t ldw)
{
    const int64_t r = blockIdx.y;
    const int64_t c2 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (c2 >= cols) return;
    *reinterpret_cast<d2 *>(&W[(n - 1 - r) * ldw + c2]) = *reinterpret_cast<const d2 *>(&V[r * ldv + c2]);
}

void launch_reverse_rows(hipStream_t s, const double *V, int64_t ldv, int64_t n_pad, int64_t cols, double *W, int64_t ldw)
{
    hipLaunchKernelGGL(reverse_rows_kernel, dim3((unsigned)((cols / 2 + 255) / 256), (unsigned)n_pad), dim3(256), 0, s, V,
                       ldv, n_pad, cols, W, ldw);
}

// out[0] = sum a_i b_i, out[1] = sum a_i a_i over n entries (one block, fixed order): the two scalars of the
// append step (l^T z, l^T l) and of the likelihood gradients (tr Ky^-1 = sum q, alpha^T alpha)
__global__ __launch_bounds__(256) void dot2_kernel(const double *__restrict__ a, const double *__restrict__ b, int64_t n,
                                                   double *__restrict__ out2)
{
    __shared__ double s1[256], s2[256];
    double x = 0.0, y = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double ai = a[i];
        x = __fma_rn(ai, b[i], x);
        y = __fma_rn(ai, ai, y);
    }
    s1[threadIdx.x] = x;
    s2[threadIdx.x] = y;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            s1[threadIdx.x] = __dadd_rn(s1[threadIdx.x], s1[threadIdx.x + st]);
            s2[threadIdx.x] = __dadd_rn(s2[threadIdx.x], s2[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = s1[0]; out2[1] = s2[0]; }
}

// Do-calculus inputs on the device (src/DoCalculus.py:74-89: the observed inputs of a graph-level GP with the
// intervened columns overwritten, one copy per candidate): raw[(c * n_obs + r) * d + j] =
// iv_index[j] >= 0 ? values[c][iv_index[j]] : observed[r][j].  Only observed (n_obs x d) and values (m x n_iv) ever
// cross the host link.
__global__ __launch_bounds__(256) void expand_interventions_kernel(const double *__restrict__ observed, int64_t n_obs,
                                                                   int d, const double *__restrict__ values, int n_iv,
                                                                   const int *__restrict__ iv_index, int64_t total,
                                                                   double *__restrict__ raw)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const int64_t c = p / n_obs, r = p % n_obs;
    for (int j = 0; j < d; ++j) {
        const int iv = iv_index[j];
        raw[p * d + j] = (iv >= 0) ? values[c * n_iv + iv] : observed[r * d + j];
    }
}

void launch_expand_interventions(hipStream_t s, const double *observed, int64_t n_obs, int d, const double *values, int n_iv,
                                 const int *iv_index, int64_t m, double *raw)
{
    const int64_t total = m * n_obs;
    hipLaunchKernelGGL(expand_interventions_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, observed, n_obs,
                       d, values, n_iv, iv_index, total, raw);
}

// out[0] = sum a_i (one block, fixed order)
__global__ __launch_bounds__(256) void sum_kernel(const double *__restrict__ a, int64_t n, double *__restrict__ out)
{
    __shared__ double s1[256];
    double x = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) x = __dadd_rn(x, a[i]);
    s1[threadIdx.x] = x;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) s1[threadIdx.x] = __dadd_rn(s1[threadIdx.x], s1[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s1[0];
}

void launch_sum(hipStream_t s, const double *a, int64_t n, double *out)
{
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, a, n, out);
}

void launch_dot2(hipStream_t s, const double *a, const double *b, int64_t n, double *out2)
{
    hipLaunchKernelGGL(dot2_kernel, dim3(1), dim3(256), 0, s, a, b, n, out2);
}

// L (row-major lower, upper zero) from the upper factor U: L[i][k] = U[k][i].
__global__ void export_lower_kernel(const double *__restrict__ A, int64_t lda, int64_t n, double *__restrict__ L)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const int64_t i = idx / n, k = idx % n;
    L[idx] = (k <= i) ? A[k * lda + i] : 0.0;
}

void launch_export_lower(hipStream_t s, const double *A, int64_t lda, int64_t n, double *L)
{
    hipLaunchKernelGGL(export_lower_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, s, A, lda, n, L);
}

// Symmetric matrix from its upper triangle.
__global__ void export_sym_kernel(const double *__restrict__ A, int64_t lda, int64_t n, double *__restrict__ K)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const int64_t i = idx / n, j = idx % n;
    K[idx] = (j >= i) ? A[i * lda + j] : A[j * lda + i];
}

void launch_export_sym(hipStream_t s, const double *A, int64_t lda, int64_t n, double *K)
{
    hipLaunchKernelGGL(export_sym_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, s, A, lda, n, K);
}

}  // namespace cbo
