// Strip-parallel forward substitution  V <- L^-1 V  on fp64 MFMA (v_mfma_f64_16x16x4_f64), gfx950.
//
// This is the dominant kernel of the acquisition sweep (predictive variance = kss - |L^-1 k*|^2,
// GPy Posterior._raw_predict, triangular form) and also the panel solve of the blocked Cholesky.
//
// Decomposition: one 256-thread workgroup per strip of 64 right-hand-side columns; wave w owns the
// 16 columns [16w, 16w+16) of the strip for ALL rows, so the four waves never exchange V data and a
// lane only ever re-reads V elements it stored itself.  Rows are processed in blocks of RB (left-
// looking):   R = V[blk] - L[blk, 0:i0] * V[0:i0]      (MFMA GEMM, K-loop over all previous rows)
//             V[blk] = L[blk,blk]^-1 R                 (16x16 diagonal inverses + MFMA updates)
// The L operand is the transposed factor U (U[k][i] = L[i][k], row-major) so an A fragment
// "A[i = lane&15][k = lane>>4]" is a read of 4 row segments of 128 B; U tiles of 32 x RB are staged
// through LDS (double-buffered, shared by the four waves), and the diagonal block's 16x16 tiles are
// staged through the same LDS for the in-block phase.  The B fragment "B[k = lane>>4][j = lane&15]"
// comes straight from V in global memory.  The f64 MFMA result map (row = (lane>>4) + 4*reg, col =
// lane&15) is exactly the B-operand map of k-step `reg`, so results feed the next MFMA with no data
// movement.  A fragments of k-step j+1 are read from LDS while the MFMAs of k-step j issue.
//
// Roofline: fp64 MFMA bound.  Algorithmic work n^2 flops per column (n^2/2 FMAs); V re-read traffic is
// n^2/(2*RB) * 8 B per column (left-looking), U traffic n^2/2*8 B per strip served from L2/MALL.
#include "cbo_internal.h"

namespace cbo {

#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

#define SCHED_DS(n) __builtin_amdgcn_sched_group_barrier(0x100, (n), 0)
#define SCHED_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, (n), 0)

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;
// LDS-DMA: 64 lanes x 16 B land at (wave-uniform LDS base) + lane * 16; the global address is per lane.
#define GLDS16(gptr, lptr) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gptr), (lds_ptr_t)(lptr), 16, 0, 0)

constexpr int kRB = 128;                  // rows per block
constexpr int kT = kRB / 16;              // 16-row tiles per block
constexpr int kKB = 32;                   // rows of U / V per pipeline stage (8 MFMA k-steps)
constexpr int kLdsLd = kRB + 16;          // U-tile row stride: rows kq and kq+1 land 32 banks apart (ds_read_b64)
constexpr int kNBuf = 3;                  // pipeline depth: DMA of stage s+2 is in flight while stage s computes
constexpr int kABuf = kKB * kLdsLd;       // doubles per U stage buffer
constexpr int kBBuf = 4 * kKB * 16;       // doubles per V stage buffer (4 waves x [32 k][16 cols])
constexpr int kDmaPerStage = 8 + 4;       // LDS-DMA instructions a wave issues per stage (8 U rows + 4 V pieces)
constexpr int kNTile = kT + kT * (kT - 1) / 2;
static_assert(kNTile * 256 <= kNBuf * kABuf, "in-block tile image must fit the U stage buffers");

// position of the 16x16 tile (s, t), s <= t, in the in-block LDS image: the kT diagonal inverses first,
// then the strictly-upper tiles row by row
__host__ __device__ constexpr int tile_slot(int s, int t)
{
    return (s == t) ? s : kT + s * (2 * kT - 1 - s) / 2 + (t - s - 1);
}

template <bool SWEEP>
__global__ __launch_bounds__(256) void trsm_strip_kernel(const double *__restrict__ U, int64_t ldu,
                                                         const double *__restrict__ invDt, double *V, int64_t ldv,
                                                         int n, const double *__restrict__ z,
                                                         double *__restrict__ q_out, double *__restrict__ mu_out)
{
    __shared__ __align__(16) double lds[kNBuf * (kABuf + kBBuf)];      // 159,744 B of the CU's 160 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, kq = lane >> 4;
    const int64_t colw = (int64_t)blockIdx.x * kStrip + wave * 16;     // first column of this wave
    double *Vc = V + colw + lc;
    double *ldsB = lds + kNBuf * kABuf;

    double qacc = 0.0, macc = 0.0;

    for (int i0 = 0; i0 < n; i0 += kRB) {
        d4 acc[kT];
#pragma unroll
        for (int t = 0; t < kT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = Vc[(int64_t)(i0 + 16 * t + kq + 4 * r) * ldv];

        const int nst = i0 / kKB;
        if (nst > 0) {
            // Every operand of the K-loop reaches LDS by LDS-DMA (no register staging), three stages deep.
            const double *ug = U + (int64_t)(wave * 8) * ldu + i0 + lane * 2;
            const double *vg = V + (int64_t)(lane >> 3) * ldv + colw + 2 * (lane & 7);
            auto issue_stage = [&](int k1, int buf) __attribute__((always_inline)) {
                double *la = lds + buf * kABuf + (wave * 8) * kLdsLd;
                const double *g = ug + (int64_t)k1 * ldu;
#pragma unroll
                for (int p = 0; p < 8; ++p) GLDS16(g + (int64_t)p * ldu, la + p * kLdsLd);   // one 1 KiB row each
                double *lb = ldsB + buf * kBBuf + wave * (kKB * 16);
                const double *gv = vg + (int64_t)k1 * ldv;
#pragma unroll
                for (int p = 0; p < 4; ++p) GLDS16(gv + (int64_t)(8 * p) * ldv, lb + p * 128);   // 8 rows x 128 B each
            };
            auto compute_stage = [&](int buf) __attribute__((always_inline)) {
                const double *abase = lds + buf * kABuf + kq * kLdsLd + lc;
                const double *bbase = ldsB + buf * kBBuf + wave * (kKB * 16) + kq * 16 + lc;
                double af[2][kT], bf[2];
#pragma unroll
                for (int t = 0; t < kT; ++t) af[0][t] = abase[16 * t];
                bf[0] = bbase[0];
                SCHED_DS(kT + 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < 7) {
#pragma unroll
                        for (int t = 0; t < kT; ++t) af[(j + 1) & 1][t] = abase[4 * (j + 1) * kLdsLd + 16 * t];
                        bf[(j + 1) & 1] = bbase[4 * (j + 1) * 16];
                    }
                    const double nb = -bf[j & 1];
#pragma unroll
                    for (int t = 0; t < kT; ++t) acc[t] = MFMA_F64(af[j & 1][t], nb, acc[t]);
                    // pin "LDS reads of step j+1, then the MFMAs of step j": the reads complete under the MFMAs
                    if (j < 7) { SCHED_DS(kT + 1); }
                    SCHED_MFMA(kT);
                }
            };
            issue_stage(0, 0);
            if (nst > 1) issue_stage(kKB, 1);
            for (int s = 0; s < nst; ++s) {
                // this wave's DMA of stage s has landed once at most stage s+1's instructions are outstanding
                // (vmcnt retires in order); the barrier then publishes every wave's share of the stage and
                // guarantees that buffer (s+2)%3, read during stage s-1, may be overwritten
                if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaPerStage) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (s + 2 < nst) issue_stage((s + 2) * kKB, (s + 2) % kNBuf);
                compute_stage(s % kNBuf);
            }
            __syncthreads();                                  // all reads of the last stage done before LDS reuse
        }

        // ---- diagonal block: stage its 16x16 tiles (inverses of the diagonal ones) through LDS
        {
            const double *Ud = U + (int64_t)i0 * ldu + i0;
            const double *iD = invDt + (int64_t)(i0 / 16) * 256;
            for (int idx = tid; idx < kNTile * 128; idx += 256) {
                const int p = idx >> 7, e = idx & 127;
                const int k = e >> 3, i2 = (e & 7) * 2;
                const double *src;
                if (p < kT) {
                    src = iD + p * 256 + k * 16 + i2;
                } else {
                    int s = 0, rem = p - kT;                  // invert tile_slot: find (s, t) with slot p
                    while (rem >= kT - 1 - s) { rem -= kT - 1 - s; ++s; }
                    const int t = s + 1 + rem;
                    src = Ud + (int64_t)(16 * s + k) * ldu + 16 * t + i2;
                }
                *reinterpret_cast<d2 *>(&lds[p * 256 + k * 16 + i2]) = *reinterpret_cast<const d2 *>(src);
            }
            __syncthreads();
            const double *tl = &lds[kq * 16 + lc];           // + slot*256 + (4 kk)*16
#pragma unroll
            for (int s = 0; s < kT; ++s) {
                d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) x = MFMA_F64(tl[tile_slot(s, s) * 256 + 64 * kk], acc[s][kk], x);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i0 + 16 * s + kq + 4 * r;
                    Vc[(int64_t)row * ldv] = x[r];
                    if (SWEEP) {
                        qacc = fma(x[r], x[r], qacc);
                        macc = fma(x[r], z[row], macc);
                    }
                }
#pragma unroll
                for (int t = s + 1; t < kT; ++t) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        acc[t] = MFMA_F64(tl[tile_slot(s, t) * 256 + 64 * kk], -x[kk], acc[t]);
                }
            }
            __syncthreads();                                  // the next block's DMA reuses the LDS
        }
    }

    if (SWEEP) {
        qacc += __shfl_xor(qacc, 16);
        qacc += __shfl_xor(qacc, 32);
        macc += __shfl_xor(macc, 16);
        macc += __shfl_xor(macc, 32);
        if (kq == 0) {
            q_out[colw + lc] = qacc;
            mu_out[colw + lc] = macc;
        }
    }
}

void launch_trsm_strips(hipStream_t s, const double *U, int64_t ldu, const double *invDt, double *V, int64_t ldv,
                        int64_t n, int64_t m_pad, const double *z, double *q, double *mu)
{
    if (n <= 0 || m_pad <= 0) return;
    // n is a multiple of 128 at every call site (n_pad of the sweep, the 128-row Cholesky panel)
    const dim3 grid((unsigned)(m_pad / kStrip));
    if (q != nullptr)
        hipLaunchKernelGGL(trsm_strip_kernel<true>, grid, dim3(256), 0, s, U, ldu, invDt, V, ldv, (int)n, z, q, mu);
    else
        hipLaunchKernelGGL(trsm_strip_kernel<false>, grid, dim3(256), 0, s, U, ldu, invDt, V, ldv, (int)n, z, q, mu);
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
