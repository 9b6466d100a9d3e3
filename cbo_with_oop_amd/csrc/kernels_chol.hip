// Recursive blocked Cholesky  Ky = U^T U  (upper factor, row-major) for gfx950, with the forward solve
// z = L^-1 (y - m) carried along as an extra right-hand-side column, and the backward solve for alpha.
//
// Restates LAPACK dpotrf + dpotrs as reached by GPy's jitchol / dpotrs (GPy ExactGaussianInference;
// the reference builds that model at /root/reference/src/GaussianProcessFactory.py:57-73).  A
// non-positive (or NaN) pivot sets *info (first failing 1-based pivot index), which the host-side
// jitter ladder reads after the factorisation.
//
//   potrf(r0, n):  n == 64 -> leaf kernel (one workgroup, LDS-resident 64x64 block + rhs column,
//                             also emits the four 16x16 diagonal inverses the strip TRSM consumes)
//                  else    -> potrf(r0, n1); panel = trsm_strips(U11, A12); syrk(A22 -= A12^T A12,
//                             rhs -= A12^T z1); potrf(r0+n1, n2)
// The SYRK is the fp64-MFMA-bound part (n^3/3 flops overall together with the panel solves).
#include "cbo_internal.h"

namespace cbo {

#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------------
// Leaf: factor the 64x64 diagonal block at (r0, r0) in LDS, right-looking, with the rhs column as a
// 65th column (so z_blk = U_bb^-T r_blk falls out of the same row scalings and rank-1 updates).
constexpr int kLeafLd = 80;   // LDS row stride (doubles): rows r and r+1 are 32 banks apart

__global__ __launch_bounds__(256) void potrf_leaf_kernel(double *A, int64_t lda, int r0, int rcol,
                                                         double *__restrict__ invDt, int *info)
{
    __shared__ double S[64][kLeafLd];
    __shared__ double Y[4][16][17];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int i = idx >> 6, j = idx & 63;
        S[i][j] = A[(int64_t)(r0 + i) * lda + r0 + j];
    }
    if (tid < 64) S[tid][64] = A[(int64_t)(r0 + tid) * lda + rcol];
    __syncthreads();

    const int tr = tid >> 4, tc = tid & 15;
    for (int j = 0; j < 64; ++j) {
        double ajj = S[j][j];
        if (!(ajj > 0.0)) {                       // also catches NaN (LAPACK: ajj <= 0 or isnan)
            if (tid == 0) atomicCAS(info, 0, r0 + j + 1);
            ajj = 1.0;                            // keep going with finite numbers; result is discarded
        }
        const double d = sqrt(ajj);
        if (tid > j && tid <= 64) S[j][tid] = S[j][tid] / d;   // columns j+1..63 and the rhs column 64
        __syncthreads();
        // trailing update of the upper triangle (r > j, c >= r) and of the rhs column (c == 64)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int r = tr + 16 * a;
            if (r > j) {
                const double ujr = S[j][r];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int c = tc + 16 * b;
                    if (c >= r) S[r][c] = fma(-ujr, S[j][c], S[r][c]);
                }
                if (tc == 0) S[r][64] = fma(-ujr, S[j][64], S[r][64]);
            }
        }
        __syncthreads();
    }
    // S[j][j] still holds the (updated) pivot a_jj; the factor's diagonal is its square root.
    if (tid < 64) {
        double ajj = S[tid][tid];
        if (!(ajj > 0.0)) ajj = 1.0;
        S[tid][65] = sqrt(ajj);                   // column 65: diagonal of U
    }
    __syncthreads();

    // Inverses of the four 16x16 diagonal blocks of U (upper triangular), column by column with back
    // substitution, so that U_bb * Y ~= I to working accuracy (the strip TRSM applies Y^T from the left).
    if (tid < 64) {
        const int blk = tid >> 4, j = tid & 15, o = 16 * blk;
        for (int i = 15; i >= 0; --i) {
            double v;
            if (i > j) {
                v = 0.0;
            } else {
                double s = (i == j) ? 1.0 : 0.0;
                for (int k = i + 1; k <= j; ++k) s = fma(-S[o + i][o + k], Y[blk][k][j], s);
                v = s / S[o + i][65];
            }
            Y[blk][i][j] = v;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 4 * 256; idx += 256) {
        const int blk = idx >> 8, i = (idx >> 4) & 15, j = idx & 15;
        invDt[(int64_t)(r0 / 16 + blk) * 256 + i * 16 + j] = Y[blk][i][j];
    }
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int i = idx >> 6, j = idx & 63;
        const double v = (j > i) ? S[i][j] : ((j == i) ? S[i][65] : 0.0);
        A[(int64_t)(r0 + i) * lda + r0 + j] = v;
    }
    if (tid < 64) A[(int64_t)(r0 + tid) * lda + rcol] = S[tid][64];
}

// ------------------------------------------------------------------------------------------------
// SYRK: C[i][j] -= sum_k P[k][i] P[k][j] on the upper tiles of the trailing block, P = the panel rows
// [r0, r0+n1).  TS x TS tile per workgroup, 2x2 waves, each wave (TS/2)^2 via 16x16x4 f64 MFMAs with
// both operand fragments read straight from the panel rows (4 row segments of 128 B per load).
// Extra blocks (blockIdx.x == nt) update the rhs column: r[i] -= sum_k P[k][i] z[k].
template <int TS>
__global__ __launch_bounds__(256) void syrk_kernel(double *A, int64_t lda, int r0, int n1, int c0, int nt, int rcol)
{
    const int tj = blockIdx.x, ti = blockIdx.y;
    const int tid = threadIdx.x;
    const double *P = A + (int64_t)r0 * lda;
    if (tj == nt) {
        // rhs column for the TS rows of tile ti
        constexpr int G = 256 / TS;
        __shared__ double part[256];
        const int i = tid % TS, g = tid / TS;
        const int64_t gi = c0 + (int64_t)ti * TS + i;
        double s = 0.0;
        for (int k = g; k < n1; k += G) s = fma(P[(int64_t)k * lda + gi], P[(int64_t)k * lda + rcol], s);
        part[tid] = s;
        __syncthreads();
        if (g == 0) {
            double tot = 0.0;
#pragma unroll
            for (int gg = 0; gg < G; ++gg) tot += part[gg * TS + i];
            A[gi * lda + rcol] -= tot;
        }
        return;
    }
    if (tj < ti) return;
    constexpr int WT = TS / 2, MT = WT / 16;
    const int lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, kq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    if (ti == tj && wr == 1 && wc == 0) return;      // strictly-lower quadrant of a diagonal tile
    const int64_t ib = c0 + (int64_t)ti * TS + wr * WT;
    const int64_t jb = c0 + (int64_t)tj * TS + wc * WT;
    d4 acc[MT][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nn = 0; nn < MT; ++nn) acc[m][nn] = d4{0.0, 0.0, 0.0, 0.0};
    const double *Pa = P + (int64_t)kq * lda + ib + lc;
    const double *Pb = P + (int64_t)kq * lda + jb + lc;
#pragma unroll 2
    for (int k0 = 0; k0 < n1; k0 += 4) {
        double a[MT], b[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = Pa[(int64_t)k0 * lda + 16 * m];
#pragma unroll
        for (int nn = 0; nn < MT; ++nn) b[nn] = Pb[(int64_t)k0 * lda + 16 * nn];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nn = 0; nn < MT; ++nn) acc[m][nn] = MFMA_F64(a[m], b[nn], acc[m][nn]);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nn = 0; nn < MT; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double *c = &A[(ib + 16 * m + kq + 4 * r) * lda + jb + 16 * nn + lc];
                *c -= acc[m][nn][r];
            }
}

static void launch_syrk(hipStream_t s, double *A, int64_t lda, int r0, int n1, int n2, int rcol)
{
    const int c0 = r0 + n1;
    if (n2 % 128 == 0) {
        const int nt = n2 / 128;
        hipLaunchKernelGGL(syrk_kernel<128>, dim3(nt + 1, nt), dim3(256), 0, s, A, lda, r0, n1, c0, nt, rcol);
    } else {
        const int nt = n2 / 64;
        hipLaunchKernelGGL(syrk_kernel<64>, dim3(nt + 1, nt), dim3(256), 0, s, A, lda, r0, n1, c0, nt, rcol);
    }
}

static void potrf_rec(hipStream_t s, double *A, int64_t lda, int r0, int n, int rcol, double *invDt, int *info)
{
    if (n == 64) {
        hipLaunchKernelGGL(potrf_leaf_kernel, dim3(1), dim3(256), 0, s, A, lda, r0, rcol, invDt, info);
        return;
    }
    // split at a multiple of 128 when possible so the panel solve can use 128-row blocks
    int n1;
    if (n >= 256) n1 = (int)round_up(n / 2, 128);
    else if (n == 192) n1 = 128;
    else n1 = 64;                                  // n == 128
    const int n2 = n - n1;
    potrf_rec(s, A, lda, r0, n1, rcol, invDt, info);
    // panel: A12 <- U11^-T A12   (rows [r0, r0+n1), columns [r0+n1, r0+n))
    launch_trsm_strips(s, A + (int64_t)r0 * lda + r0, lda, invDt + (int64_t)(r0 / 16) * 256,
                       A + (int64_t)r0 * lda + r0 + n1, lda, n1, n2, nullptr, 0, nullptr, nullptr);
    launch_syrk(s, A, lda, r0, n1, n2, rcol);
    potrf_rec(s, A, lda, r0 + n1, n2, rcol, invDt, info);
}

void launch_cholesky(hipStream_t s, double *A, int64_t lda, int64_t n_pad, double *invDt, int *info_dev)
{
    hipMemsetAsync(info_dev, 0, sizeof(int), s);
    potrf_rec(s, A, lda, 0, (int)n_pad, (int)n_pad, invDt, info_dev);
}

// ------------------------------------------------------------------------------------------------
// Backward solve U alpha = z, 64-row blocks from the bottom.  One launch per block: every workgroup
// first solves the 64x64 diagonal system redundantly (one wave, lane = row, no barriers inside), then
// workgroup g folds alpha_blk into the 64 rows of block g above it:  zt[g] -= U[g, blk] alpha_blk.
// zt is a contiguous working copy of z.
__global__ __launch_bounds__(256) void backsolve_step_kernel(const double *__restrict__ A, int64_t lda, int blk,
                                                             double *zt, double *__restrict__ alpha)
{
    __shared__ double D[64][65];
    __shared__ double al[64];
    const int tid = threadIdx.x;
    const int b0 = blk * 64;
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int i = idx >> 6, j = idx & 63;
        D[i][j] = A[(int64_t)(b0 + i) * lda + b0 + j];
    }
    __syncthreads();
    if (tid < 64) {
        double zi = zt[b0 + tid];
        const double dii = D[tid][tid];
        for (int c = 63; c >= 0; --c) {
            const double ac = __shfl(zi, c) / __shfl(dii, c);     // alpha_c (lane c's residual is final)
            if (tid == c) al[c] = ac;
            if (tid < c) zi = fma(-D[tid][c], ac, zi);
        }
    }
    __syncthreads();
    const int g = blockIdx.x;
    if (g == blk) {
        if (tid < 64) alpha[b0 + tid] = al[tid];
        return;
    }
    const int i = tid >> 2, part = tid & 3;
    const double *row = A + (int64_t)(g * 64 + i) * lda + b0 + 16 * part;
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) s = fma(row[c], al[16 * part + c], s);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (part == 0) zt[g * 64 + i] -= s;
}

__global__ void copy_strided_kernel(const double *__restrict__ src, int64_t stride, int64_t n, double *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i * stride];
}

// alpha doubles as the working copy: first alpha <- z, then blocks are finalised bottom-up.
void launch_backsolve(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, double *alpha)
{
    // zt lives in alpha's upper half?  No: keep a separate contiguous copy so alpha is written once.
    // The caller provides alpha with 2*n_pad doubles: [0, n_pad) result, [n_pad, 2 n_pad) working z.
    double *zt = alpha + n_pad;
    hipLaunchKernelGGL(copy_strided_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, s, A + n_pad, lda,
                       n_pad, zt);
    const int nb = (int)(n_pad / 64);
    for (int blk = nb - 1; blk >= 0; --blk)
        hipLaunchKernelGGL(backsolve_step_kernel, dim3(blk + 1), dim3(256), 0, s, A, lda, blk, zt, alpha);
}

}  // namespace cbo
