// Strip-parallel forward substitution  V <- L^-1 V  on fp64 MFMA (v_mfma_f64_16x16x4_f64), gfx950.
//
// This is the dominant kernel of the acquisition sweep (predictive variance = kss - |L^-1 k*|^2,
// GPy Posterior._raw_predict, triangular form) and also the panel solve of the recursive Cholesky.
//
// Decomposition: one 256-thread workgroup per strip of 64 right-hand-side columns; wave w owns the
// 16 columns [16w, 16w+16) of the strip for ALL rows, so the four waves never exchange V data and a
// lane only ever re-reads V elements it stored itself.  Rows are processed in blocks of RB (left-
// looking):   R = V[blk] - L[blk, 0:i0] * V[0:i0]      (MFMA GEMM, K-loop over all previous rows)
//             V[blk] = L[blk,blk]^-1 R                 (16x16 diagonal inverses + MFMA updates)
// The L operand is the transposed factor U (U[k][i] = L[i][k], row-major) so an A fragment
// "A[i = lane&15][k = lane>>4]" is a read of 4 row segments of 128 B; U tiles of 32 x RB are staged
// through LDS (double-buffered, shared by the four waves).  The B fragment "B[k = lane>>4][j = lane&15]"
// comes straight from V in global memory.  The f64 MFMA result map (row = (lane>>4) + 4*reg, col =
// lane&15) is exactly the B-operand map of k-step `reg`, so results feed the next MFMA with no data
// movement.
//
// Roofline: fp64 MFMA bound.  Algorithmic work n^2 flops per column (n^2/2 FMAs); V re-read traffic is
// n^2/(2*RB) * 8 B per column (left-looking), U traffic n^2/2*8 B per strip served from L2/MALL.
#include "cbo_internal.h"

namespace cbo {

#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

constexpr int kKB = 32;   // rows of U staged per LDS stage (8 MFMA k-steps)

template <int RB>
__global__ __launch_bounds__(256) void trsm_strip_kernel(const double *__restrict__ U, int64_t ldu,
                                                         const double *__restrict__ invDt, double *V, int64_t ldv,
                                                         int n, const double *__restrict__ z, int64_t z_stride,
                                                         double *__restrict__ q_out, double *__restrict__ mu_out)
{
    constexpr int T = RB / 16;            // 16-row tiles per row block
    constexpr int LDS_LD = RB + 16;       // row stride: rows kq and kq+1 land 32 banks apart (ds_read_b64)
    constexpr int TPR = RB / 2;           // threads per staged row (16 B each)
    constexpr int RPP = 256 / TPR;        // rows per staging pass
    constexpr int NP = kKB / RPP;         // staging passes per stage
    __shared__ double lds[2][kKB][LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, kq = lane >> 4;
    const int64_t col = (int64_t)blockIdx.x * kStrip + wave * 16 + lc;
    double *Vc = V + col;
    const int s_rr = tid / TPR, s_cc = (tid % TPR) * 2;

    double qacc = 0.0, macc = 0.0;

    for (int i0 = 0; i0 < n; i0 += RB) {
        d4 acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = Vc[(int64_t)(i0 + 16 * t + kq + 4 * r) * ldv];

        const int nst = i0 / kKB;
        if (nst > 0) {
            d2 st[NP];
            double bcur[8], bnext[8];
#pragma unroll
            for (int p = 0; p < NP; ++p)
                st[p] = *reinterpret_cast<const d2 *>(&U[(int64_t)(p * RPP + s_rr) * ldu + i0 + s_cc]);
#pragma unroll
            for (int j = 0; j < 8; ++j) bcur[j] = Vc[(int64_t)(4 * j + kq) * ldv];
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<d2 *>(&lds[0][p * RPP + s_rr][s_cc]) = st[p];
            __syncthreads();
            for (int s = 0; s < nst; ++s) {
                const int cur = s & 1;
                const bool more = (s + 1) < nst;
                if (more) {
                    const int k1 = (s + 1) * kKB;
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        st[p] = *reinterpret_cast<const d2 *>(&U[(int64_t)(k1 + p * RPP + s_rr) * ldu + i0 + s_cc]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) bnext[j] = Vc[(int64_t)(k1 + 4 * j + kq) * ldv];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double nb = -bcur[j];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const double a = lds[cur][4 * j + kq][16 * t + lc];
                        acc[t] = MFMA_F64(a, nb, acc[t]);
                    }
                }
                if (more) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        *reinterpret_cast<d2 *>(&lds[cur ^ 1][p * RPP + s_rr][s_cc]) = st[p];
#pragma unroll
                    for (int j = 0; j < 8; ++j) bcur[j] = bnext[j];
                }
                __syncthreads();
            }
        }

        // Diagonal block: X_s = inv(L_ss) R_s, then R_t -= L_ts X_s for the tiles below.
        const double *Ud = U + (int64_t)i0 * ldu + i0;
        const double *iD = invDt + (int64_t)(i0 / 16) * 256;
#pragma unroll
        for (int s = 0; s < T; ++s) {
            d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double a = iD[s * 256 + (4 * kk + kq) * 16 + lc];
                x = MFMA_F64(a, acc[s][kk], x);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + 16 * s + kq + 4 * r;
                Vc[(int64_t)row * ldv] = x[r];
                qacc = fma(x[r], x[r], qacc);
                if (z) macc = fma(x[r], z[(int64_t)row * z_stride], macc);
            }
#pragma unroll
            for (int t = s + 1; t < T; ++t) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double a = Ud[(int64_t)(16 * s + 4 * kk + kq) * ldu + 16 * t + lc];
                    acc[t] = MFMA_F64(a, -x[kk], acc[t]);
                }
            }
        }
    }

    if (q_out) {
        qacc += __shfl_xor(qacc, 16);
        qacc += __shfl_xor(qacc, 32);
        macc += __shfl_xor(macc, 16);
        macc += __shfl_xor(macc, 32);
        if (kq == 0) {
            q_out[col] = qacc;
            if (mu_out) mu_out[col] = macc;
        }
    }
}

void launch_trsm_strips(hipStream_t s, const double *U, int64_t ldu, const double *invDt, double *V, int64_t ldv,
                        int64_t n, int64_t m_pad, const double *z, int64_t z_stride, double *q, double *mu)
{
    if (n <= 0 || m_pad <= 0) return;
    const dim3 grid((unsigned)(m_pad / kStrip));
    if (n % 128 == 0)
        hipLaunchKernelGGL(trsm_strip_kernel<128>, grid, dim3(256), 0, s, U, ldu, invDt, V, ldv, (int)n, z, z_stride,
                           q, mu);
    else
        hipLaunchKernelGGL(trsm_strip_kernel<64>, grid, dim3(256), 0, s, U, ldu, invDt, V, ldv, (int)n, z, z_stride, q,
                           mu);
}

// ------------------------------------------------------------------------------------------------
// MFMA lane-map self test: C = A * B with asymmetric integer-valued 16x16 operands (exact in fp64).
__global__ void mfma_selftest_kernel(const double *A, const double *B, double *C)
{
    const int l = threadIdx.x;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < 16; k0 += 4)
        acc = MFMA_F64(A[(l & 15) * 16 + k0 + (l >> 4)], B[(k0 + (l >> 4)) * 16 + (l & 15)], acc);
    // Feed the result back as a B operand (k-step r <- register r): D2 = A * C.
    d4 acc2 = {0.0, 0.0, 0.0, 0.0};
    for (int kk = 0; kk < 4; ++kk) acc2 = MFMA_F64(A[(l & 15) * 16 + 4 * kk + (l >> 4)], acc[kk], acc2);
    for (int r = 0; r < 4; ++r) {
        C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
        C[256 + ((l >> 4) + 4 * r) * 16 + (l & 15)] = acc2[r];
    }
}

int run_mfma_selftest(hipStream_t s, double *max_err)
{
    double hA[256], hB[256], hC[512], ref[256], ref2[256];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            hA[i * 16 + j] = (double)((i * 7 + j * 3) % 11 - 5);
            hB[i * 16 + j] = (double)((i * 5 + j * 13 + 1) % 17 - 8);
        }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 16; ++k) acc += hA[i * 16 + k] * hB[k * 16 + j];
            ref[i * 16 + j] = acc;
        }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 16; ++k) acc += hA[i * 16 + k] * ref[k * 16 + j];
            ref2[i * 16 + j] = acc;
        }
    double *dA = nullptr, *dB = nullptr, *dC = nullptr;
    if (hipMalloc(&dA, sizeof(hA)) != hipSuccess || hipMalloc(&dB, sizeof(hB)) != hipSuccess ||
        hipMalloc(&dC, sizeof(hC)) != hipSuccess)
        return -1;
    hipMemcpyAsync(dA, hA, sizeof(hA), hipMemcpyHostToDevice, s);
    hipMemcpyAsync(dB, hB, sizeof(hB), hipMemcpyHostToDevice, s);
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, s, dA, dB, dC);
    hipMemcpyAsync(hC, dC, sizeof(hC), hipMemcpyDeviceToHost, s);
    const hipError_t e = hipStreamSynchronize(s);
    hipFree(dA); hipFree(dB); hipFree(dC);
    if (e != hipSuccess) return -1;
    double m = 0.0;
    for (int i = 0; i < 256; ++i) {
        const double e1 = fabs(hC[i] - ref[i]), e2 = fabs(hC[256 + i] - ref2[i]);
        if (e1 > m) m = e1;
        if (e2 > m) m = e2;
    }
    *max_err = m;
    return 0;
}

}  // namespace cbo
