// Forward substitution  V <- L^-1 V  on fp64 MFMA (v_mfma_f64_16x16x4_f64), gfx950: the work of the acquisition
// sweep (predictive variance = kss - |L^-1 k*|^2, GPy Posterior._raw_predict, triangular form) and the panel
// solve of the blocked Cholesky.  Two kernels share one decomposition, LDS staging and MFMA order:
//
//   trsm_strip_kernel<SWEEP, KB>   left-looking: a workgroup owns a strip of 64 right-hand-side columns for ALL
//                                  rows of its (sub-)problem
//   trsm_update_kernel<KB>         the right-looking cut of the same sums, C -= U_panel^T V_panel over strips x
//                                  row chunks, used to run the sweep underneath the factorisation
//
// Decomposition: 256-thread workgroups, 64 columns per strip; wave w owns the 16 columns [16w, 16w+16), so the four
// waves never exchange V data and a lane only ever re-reads V elements it stored itself.  Rows go in blocks of
// 128:    R = V[blk] - L[blk, 0:i0] * V[0:i0]      (MFMA GEMM, K-loop over the previous rows)
//         V[blk] = L[blk,blk]^-1 R                 (16x16 diagonal inverses + MFMA updates)
// The L operand is the transposed factor U (U[k][i] = L[i][k], row-major) so an A fragment
// "A[i = lane&15][k = lane>>4]" is a read of 4 row segments of 128 B; U tiles of KB x 128 are staged through LDS
// by LDS-DMA (3-deep ring shared by the four waves), together with the wave's own V rows (B fragments
// "B[k = lane>>4][j = lane&15]") or, in the diagonal stages, the 16x16 diagonal inverses.  The f64 MFMA result
// map (row = (lane>>4) + 4*reg, col = lane&15) is exactly the B-operand map of k-step `reg`, so results feed
// the next MFMA with no data movement.  A fragments of k-step j+1 are read from LDS while the MFMAs of k-step j
// issue.
//
// Roofline: fp64 MFMA bound.  Algorithmic work n^2 flops per column (n^2/2 FMAs); left-looking V re-read traffic
// n^2/(2*128) * 8 B per column, U traffic n^2/2*8 B per strip served from L2/MALL; right-looking adds one
// read-modify-write of the rows below per panel pair.
#include <cstdlib>
#include <type_traits>

#include "cbo_internal.h"

namespace cbo {

#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

#define SCHED_DS(n) __builtin_amdgcn_sched_group_barrier(0x100, (n), 0)
#define SCHED_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, (n), 0)
#define SCHED_VMEM(n) __builtin_amdgcn_sched_group_barrier(0x010, (n), 0)

constexpr int kRB = 128;                  // rows per block
constexpr int kT = kRB / 16;              // 16-row tiles per block
constexpr int kLdsLd = kRB + 16;          // U-tile row stride: rows kq and kq+1 land 32 banks apart (ds_read_b64)
constexpr int kNBuf = 3;                  // pipeline depth: DMA of stage s+2 is in flight while stage s computes
// Per-stage geometry for KB rows of U / V per pipeline stage (KB/4 MFMA k-steps).  KB = 32: one workgroup fills a
// CU's LDS (159,744 B) -- fewest barriers, the choice when there is one workgroup per CU anyway.  KB = 16: 79,872 B
// and < 256 registers per lane, so two workgroups share a CU (two waves per SIMD) and hide each other's barrier
// and LDS latencies; used when kernels of several streams are in flight at once (the pipelined refit + sweep).
template <int KB>
struct StageGeom {
    static constexpr int kA = KB * kLdsLd;            // doubles per U stage buffer
    static constexpr int kB = 4 * KB * 16;            // doubles per V stage buffer (4 waves x [KB k][16 cols])
    static constexpr int kRA = KB / 4;                // U rows a wave fetches per stage
    static constexpr int kParts = KB / 8;             // DMA groups per stage and wave: 2 U rows + one 1 KiB B piece
    static constexpr int kDma = 3 * kParts;           // LDS-DMA instructions a wave issues per stage
    static constexpr int kKS = KB / 4;                // MFMA k-steps per stage
    static constexpr int kDiagStages = kRB / KB;      // diagonal stages per row block
    static constexpr int kDiagTiles = KB / 16;        // 16x16 diagonal tiles solved per diagonal stage
    static constexpr int kDiagStores = 4 * kDiagTiles;   // V stores a wave issues per diagonal stage
    static_assert(kParts <= kKS - 2, "the DMA groups and the cursor update ride under the first k-steps");
};
#ifdef CBO_DIAG_KNOBS
// Timing-only build (make DIAG=1 -> libcbo_hip_diag.so): waves 0 and 4 of workgroup 0 of trsm_pair_kernel stamp s_memtime at
// the milestones of every stage into a debug buffer read back by cbo_diag_trsm_stamps (scripts/pair_timeline.py).
__device__ unsigned long long g_trsm_stamps[8 * 4096];
extern "C" int cbo_diag_trsm_stamps(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trsm_stamps), sizeof(unsigned long long) * (size_t)n);   // 8 per stage
}
// Timing-only: every workgroup of the last trsm_update_kernel launch leaves [start, end, hw id | xcc id << 32, start, end in
// s_memrealtime's 100 MHz] (s_memtime runs at the shader clock), and
// workgroup (7, gridDim.y / 2) its stage tops: scripts/update_kernel_stamps.py
__device__ unsigned long long g_upd_wg[5 * 65536];
__device__ unsigned long long g_upd_stage[128];
extern "C" int cbo_diag_upd_stamps(unsigned long long *wg, unsigned long long *stage)
{
    int rc = (int)hipMemcpyFromSymbol(wg, HIP_SYMBOL(g_upd_wg), sizeof(unsigned long long) * 5 * 65536);
    if (rc == 0) rc = (int)hipMemcpyFromSymbol(stage, HIP_SYMBOL(g_upd_stage), sizeof(unsigned long long) * 128);
    return rc;
}
#endif

// One continuous software pipeline over "stages" of KB U-rows (described for KB = 32).  Row block b (rows
// i0 = 128 b) consists of nst = i0/32 regular stages (k rows [32 j, 32 j + 32) against the block's 128 columns)
// followed by four diagonal stages (k rows i0 + 32 m: the block's own upper-triangular part).  Every stage's U
// tile [32][128] is fetched by the same LDS-DMA pattern; the per-wave B region receives V rows (regular
// stages) or the two 16x16 diagonal inverses (diagonal stages).  The DMA of stage g+2 is issued while
// stage g computes, across block boundaries, so the pipeline never drains.
struct StageCursor {
    int i0, j, lim;                       // block origin, stage index within the block, stages in the block
    int ai0, aj;                          // the same, clamped to the last stage once the cursor is past the end
    const double *a_src;                  // per-lane source of the U tile's first row handled by this wave
    const double *b_src;                  // per-lane source of the wave's first B piece
    int64_t b_stride;                     // doubles between consecutive B pieces
};

template <bool SWEEP, int KB>
__global__ __launch_bounds__(256) void trsm_strip_kernel(const double *__restrict__ U, int64_t ldu,
                                                         const double *__restrict__ invDt, double *V, int64_t ldv,
                                                         int n, const double *__restrict__ z,
                                                         double *__restrict__ q_out, double *__restrict__ mu_out,
                                                         int accumulate)
{
    using G = StageGeom<KB>;
    constexpr int kKB = KB, kABuf = G::kA, kBBuf = G::kB, kDmaPerStage = G::kDma, kStoresPerDiagStage = G::kDiagStores;
    constexpr int kDS = G::kDiagStages, kDT = G::kDiagTiles, kKS = G::kKS, kParts = G::kParts;
    __shared__ __align__(16) double lds[kNBuf * (kABuf + kBBuf)];      // KB = 32: 159,744 B of the CU's 160 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave-uniform: LDS bases stay scalar
    const int lc = lane & 15, kq = lane >> 4;
    const int64_t colw = (int64_t)blockIdx.x * kStrip + wave * 16;     // first column of this wave
    double *Vc = V + colw + lc;
    double *ldsB = lds + kNBuf * kABuf;
    const unsigned lds_byte0 = lds_byte_address(lds);      // LDS byte address of lds[0]
    const double *ug = U + (int64_t)(wave * G::kRA) * ldu + lane * 2;
    const double *vg = V + (int64_t)(lane >> 3) * ldv + colw + 2 * (lane & 7);

    // Sources of the stage the cursor points at (scalar bookkeeping, done off the MFMA path).  A cursor past
    // the end is clamped to the last stage, whose buffer is free by then, so in-flight counts stay uniform.
    // Branch-free on purpose (selects and masked arithmetic only): the cursor update sits in the middle of a
    // regular stage's MFMA block, where the scheduler can only hide it if it stays in that basic block.
    const double *inv_lane = invDt + lane * 2;
    // (three pieces, so that the regular stage can place each in the shadow of a different MFMA)
    auto locate_a = [&](StageCursor &c) __attribute__((always_inline)) {
        const bool past = c.i0 >= n;
        c.ai0 = past ? n - kRB : c.i0;
        c.aj = past ? (n - kRB) / kKB + kDS - 1 : c.j;
    };
    auto locate_b = [&](StageCursor &c) __attribute__((always_inline)) {
        c.a_src = ug + (int64_t)(kKB * c.aj) * ldu + c.ai0;
    };
    auto locate_c = [&](StageCursor &c) __attribute__((always_inline)) {
        const int nreg = c.ai0 / kKB;
        // diagonal stage: the two 16x16 diagonal inverses go to the B region; regular stage: V rows
        // [32 aj, 32 aj + 32) of this wave's 16 columns
        const int64_t diag = (c.aj >= nreg) ? 1 : 0;
        const int64_t off_diag = ((int64_t)(c.ai0 / 16) + kDT * (c.aj - nreg)) * 256;
        const int64_t off_reg = (int64_t)(kKB * c.aj) * ldv;
        const uintptr_t base = (uintptr_t)vg + ((uintptr_t)inv_lane - (uintptr_t)vg) * (uintptr_t)diag;
        c.b_src = reinterpret_cast<const double *>(base) + (off_reg + (off_diag - off_reg) * diag);
        c.b_stride = 8 * ldv + (128 - 8 * ldv) * diag;
    };
    auto locate = [&](StageCursor &c) __attribute__((always_inline)) {
        locate_a(c);
        locate_b(c);
        locate_c(c);
    };
    auto step = [&](StageCursor &c) __attribute__((always_inline)) {
        const int wrap = (c.j + 1 == c.lim) ? 1 : 0;
        c.i0 += kRB * wrap;
        c.j = (c.j + 1) * (1 - wrap);
        c.lim = c.lim + (c.i0 / kKB + kDS - c.lim) * wrap;
    };
    auto advance = [&](StageCursor &c) __attribute__((always_inline)) {
        step(c);
        locate(c);
    };
    // the LDS-DMA instructions of a stage, split so they can sit between MFMAs: kParts groups of 3
    auto issue_part = [&](const StageCursor &c, int buf, int part) __attribute__((always_inline)) {
        const unsigned la = __builtin_amdgcn_readfirstlane(lds_byte0 + 8u * (unsigned)(buf * kABuf + (wave * G::kRA) * kLdsLd));
        const unsigned lb = __builtin_amdgcn_readfirstlane(
            lds_byte0 + 8u * (unsigned)(kNBuf * kABuf + buf * kBBuf + wave * (kKB * 16)));
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = 2 * part + q;
            glds16(c.a_src + (int64_t)p * ldu, la + 8u * (unsigned)(p * kLdsLd));   // one 1 KiB U row
        }
        glds16(c.b_src + part * c.b_stride, lb + 8u * (unsigned)(part * 128));      // one 1 KiB B piece
    };
    // the same instructions one at a time (q = 0, 1: the group's U rows; q = 2: its B piece)
    auto issue_one = [&](const StageCursor &c, int buf, int part, int q) __attribute__((always_inline)) {
        if (q < 2) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds_byte0 + 8u * (unsigned)(buf * kABuf + (wave * G::kRA) * kLdsLd));
            const int p = 2 * part + q;
            glds16(c.a_src + (int64_t)p * ldu, la + 8u * (unsigned)(p * kLdsLd));
        } else {
            const unsigned lb = __builtin_amdgcn_readfirstlane(
                lds_byte0 + 8u * (unsigned)(kNBuf * kABuf + buf * kBBuf + wave * (kKB * 16)));
            glds16(c.b_src + part * c.b_stride, lb + 8u * (unsigned)(part * 128));
        }
    };
    auto issue_stage = [&](const StageCursor &c, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int part = 0; part < kParts; ++part) issue_part(c, buf, part);
    };

    // acc holds the NEGATED residual  L[blk, 0:k] V[0:k] - V[blk]  so the K-loop needs no operand negation
    d4 acc[kT], accn[kT];
#pragma unroll
    for (int t = 0; t < kT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = -Vc[(int64_t)(16 * t + kq + 4 * r) * ldv];

    StageCursor ahead{0, 0, kDS, 0, 0, nullptr, nullptr, 0};
    locate(ahead);
    issue_stage(ahead, 0);
    advance(ahead);
    issue_stage(ahead, 1);
    advance(ahead);
    int buf = 0;                 // buffer of the current stage; stage g+2 goes to (buf + 2) % 3
    int extra_prev = 0;          // VMEM operations the previous stage issued after its DMA (its V stores)
    // q = sum V^2 and mu = V^T z: lane partials over a PAIR of row blocks (256 rows), reduced over the four
    // lane groups and added to a running total pair by pair -- the grouping the right-looking pipeline produces
    // naturally (one launch of this kernel per pair, accumulate = 1), so both schedules give the same bits
    double qacc = 0.0, macc = 0.0, qtot = 0.0, mtot = 0.0;
    if (SWEEP && accumulate) {            // continue the running totals of the launches before this one
        qtot = q_out[colw + lc];
        mtot = mu_out[colw + lc];
    }

    // top of a stage: this wave's DMA of the stage has landed once only the next stage's 12 DMA instructions
    // (plus whatever the previous stage issued after them) are outstanding -- vmcnt retires in order; the
    // barrier publishes every wave's share and frees buffer (buf+2)%3, read during the previous stage
#define STAGE_TOP()                                                                                       \
    do {                                                                                                  \
        if (extra_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaPerStage + kStoresPerDiagStage) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaPerStage) : "memory");                          \
        __builtin_amdgcn_s_barrier();                                                                     \
    } while (0)

    for (int i0 = 0; i0 < n; i0 += kRB) {
        const int nst = i0 / kKB;
        // Regular stages, software-pipelined ACROSS the stage barrier: the MFMAs of a stage's last k-step are
        // issued after the next stage's barrier and first LDS reads, so barrier skew and the first read latency
        // hide under 8 MFMAs (512 cycles) instead of draining the matrix pipe.
        double af[2][kT], bf[2];
        bool deferred = false;                // af[1]/bf[1] hold the previous stage's k-step 7, MFMAs not yet issued
        for (int j = 0; j < nst; ++j) {
            // every LDS read this wave issued for the previous stage has returned: after the barrier other
            // waves' DMA may overwrite that buffer
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            STAGE_TOP();
            __builtin_amdgcn_sched_barrier(0);
            const int bnext = (buf >= 1) ? buf - 1 : 2;       // (buf + 2) % 3
            extra_prev = 0;
            const double *abase = lds + buf * kABuf + kq * kLdsLd + lc;
            const double *bbase = ldsB + buf * kBBuf + wave * (kKB * 16) + kq * 16 + lc;
            // One wave per SIMD issues in order, and an instruction of any kind takes a few cycles of issue: whatever
            // stands between two MFMAs beyond the 64 cycles the first one executes is a hole in the matrix pipe.  So
            // every MFMA is followed by ONE small piece of the other work -- a ds_read2 of the next k-step's A
            // fragments, its B fragment, one LDS-DMA instruction of stage g+2, a share of the cursor arithmetic --
            // and a scheduling fence pins that order (ISA of round 2: all reads and DMA of a k-step bunched ahead of
            // its eight MFMAs, ~600 cycles per k-step instead of 512).
#pragma unroll
            for (int t = 0; t < kT; ++t) {
                if (deferred) acc[t] = MFMA_F64(af[1][t], bf[1], acc[t]);
                if (t < kT / 2) {
                    af[0][2 * t] = abase[32 * t];
                    af[0][2 * t + 1] = abase[32 * t + 16];
                } else if (t == kT / 2) {
                    bf[0] = bbase[0];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int jj = 0; jj < kKS - 1; ++jj) {
                const double *an = abase + 4 * (jj + 1) * kLdsLd;
#pragma unroll
                for (int t = 0; t < kT; ++t) {
                    acc[t] = MFMA_F64(af[jj & 1][t], bf[jj & 1], acc[t]);
                    if (t < kT / 2) {
                        af[(jj + 1) & 1][2 * t] = an[32 * t];
                        af[(jj + 1) & 1][2 * t + 1] = an[32 * t + 16];
                    } else if (t == kT / 2) {
                        bf[(jj + 1) & 1] = bbase[4 * (jj + 1) * 16];
                    } else if (jj < kParts) {
                        issue_one(ahead, bnext, jj, t - (kT / 2 + 1));    // stage g+2's DMA rides under the MFMAs
                    } else if (jj == kParts) {                            // cursor bookkeeping under the MFMAs too
                        if (t == kT / 2 + 1) step(ahead);
                        if (t == kT / 2 + 2) locate_a(ahead);
                        if (t == kT / 2 + 3) {
                            locate_b(ahead);
                            if (kParts + 1 >= kKS - 1) locate_c(ahead);   // KB = 16: no further k-step in the stage
                        }
                    } else if (jj == kParts + 1) {
                        if (t == kT / 2 + 1) locate_c(ahead);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            deferred = true;                                              // the last k-step sits in af[1], bf[1]
            buf = (buf == 2) ? 0 : buf + 1;
        }
        if (deferred) {
#pragma unroll
            for (int t = 0; t < kT; ++t) acc[t] = MFMA_F64(af[1][t], bf[1], acc[t]);
        }

        // ---- diagonal stages (KB rows each): X_s = inv(L_ss) R_s, then R_t -= L_ts X_s for the tiles below
#pragma unroll
        for (int m = 0; m < kDS; ++m) {
            STAGE_TOP();
            if (m == 0 && i0 + kRB < n) {
                // next block's right-hand sides (K* rows) -- issued before this stage's DMA so that the
                // DMA waits further down never have to cover them early
#pragma unroll
                for (int t = 0; t < kT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) accn[t][r] = -Vc[(int64_t)(i0 + kRB + 16 * t + kq + 4 * r) * ldv];
            }
            // z rows of this stage, fetched ahead of the DMA issue so they are older than it in vmcnt order
            double zr[kDT][4];
            if (SWEEP) {
#pragma unroll
                for (int h = 0; h < kDT; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) zr[h][r] = z[i0 + kKB * m + 16 * h + kq + 4 * r];
                asm volatile("" ::: "memory");
            }
            const int bnext = (buf >= 1) ? buf - 1 : 2;
            issue_stage(ahead, bnext);
            advance(ahead);
            const double *abase = lds + buf * kABuf + kq * kLdsLd + lc;
            const double *ibase = ldsB + buf * kBBuf + wave * (kKB * 16) + kq * 16 + lc;
            // every LDS operand of the stage is fetched up front: the X_s / update chain below is a string of
            // dependent MFMAs and must not wait for an LDS read in between
            double iv[kDT][4], uf[kDT][kT][4];
#pragma unroll
            for (int h = 0; h < kDT; ++h) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) iv[h][kk] = ibase[h * 256 + 64 * kk];
#pragma unroll
                for (int t = kDT * m + h + 1; t < kT; ++t)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) uf[h][t][kk] = abase[(16 * h + 4 * kk) * kLdsLd + 16 * t];
            }
            asm volatile("" ::: "memory");                    // keep the reads ahead of the chain
            // x_s = inv(L_ss) r_s as two independent half-sums (a chain of dependent f64 MFMAs runs at about half the
            // issue rate); every accumulator receives its updates in the same order in every variant of this kernel
            // (tile by tile, k ascending), so the schedules stay bit-identical
            auto emit = [&](int h, int s, const d4 &x) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i0 + 16 * s + kq + 4 * r;
                    Vc[(int64_t)row * ldv] = x[r];
                    if (SWEEP) {
                        qacc = fma(x[r], x[r], qacc);
                        macc = fma(x[r], zr[h][r], macc);
                    }
                }
            };
            if constexpr (kDT == 2) {
                // two tiles per stage: tile s+1 is brought up to date first (with tile s+2 in between, so that no MFMA
                // waits on its predecessor), then ITS solve chain issues with the rest of tile s's updates filling the
                // gaps -- the matrix pipe does not idle through the second chain
                const int s = kDT * m;
                d4 x = {0.0, 0.0, 0.0, 0.0}, x2 = {0.0, 0.0, 0.0, 0.0};
                x = MFMA_F64(iv[0][0], -acc[s][0], x);
                x2 = MFMA_F64(iv[0][1], -acc[s][1], x2);
                x = MFMA_F64(iv[0][2], -acc[s][2], x);
                x2 = MFMA_F64(iv[0][3], -acc[s][3], x2);
                x += x2;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    acc[s + 1] = MFMA_F64(uf[0][s + 1][kk], x[kk], acc[s + 1]);
                    if (s + 2 < kT) acc[s + 2] = MFMA_F64(uf[0][s + 2][kk], x[kk], acc[s + 2]);
                }
                emit(0, s, x);
                const d4 na = -acc[s + 1];
                d4 y1 = {0.0, 0.0, 0.0, 0.0}, y2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (kk & 1) y2 = MFMA_F64(iv[1][kk], na[kk], y2);
                    else y1 = MFMA_F64(iv[1][kk], na[kk], y1);
#pragma unroll
                    for (int t = s + 3; t < kT; ++t) acc[t] = MFMA_F64(uf[0][t][kk], x[kk], acc[t]);
                }
                const d4 y = y1 + y2;
                emit(1, s + 1, y);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int t = s + 2; t < kT; ++t) acc[t] = MFMA_F64(uf[1][t][kk], y[kk], acc[t]);
                }
            } else {
#pragma unroll
                for (int h = 0; h < kDT; ++h) {
                    const int s = kDT * m + h;
                    d4 x = {0.0, 0.0, 0.0, 0.0}, x2 = {0.0, 0.0, 0.0, 0.0};
                    x = MFMA_F64(iv[h][0], -acc[s][0], x);
                    x2 = MFMA_F64(iv[h][1], -acc[s][1], x2);
                    x = MFMA_F64(iv[h][2], -acc[s][2], x);
                    x2 = MFMA_F64(iv[h][3], -acc[s][3], x2);
                    x += x2;
                    emit(h, s, x);
                    // k-major order: consecutive MFMAs go to different accumulators (no dependent back-to-back issue)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                        for (int t = s + 1; t < kT; ++t) acc[t] = MFMA_F64(uf[h][t][kk], x[kk], acc[t]);
                    }
                }
            }
            extra_prev = 1;
            buf = (buf == 2) ? 0 : buf + 1;
        }
#pragma unroll
        for (int t = 0; t < kT; ++t) acc[t] = accn[t];
        if (SWEEP && (((i0 / kRB) & 1) || i0 + kRB >= n)) {
            qacc += __shfl_xor(qacc, 16);
            qacc += __shfl_xor(qacc, 32);
            macc += __shfl_xor(macc, 16);
            macc += __shfl_xor(macc, 32);
            qtot += qacc;
            mtot += macc;
            qacc = 0.0;
            macc = 0.0;
        }
    }
#undef STAGE_TOP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the clamped tail DMA before the LDS goes away
    __builtin_amdgcn_s_barrier();

    if (SWEEP) {
        if (kq == 0) {
            q_out[colw + lc] = qtot;
            mu_out[colw + lc] = mtot;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The strip kernel with TWO waves per SIMD (512 threads): a single wave cannot keep the fp64 matrix pipe busy --
// scripts/probes/stage_probe.hip: back-to-back independent v_mfma_f64_16x16x4_f64 from one wave issue every ~75
// cycles instead of 64, and every barrier, s_waitcnt or dependent chain of that wave is a hole in the pipe; two waves
// per SIMD with the same LDS reads, DMA and stage barrier run at 0.88 of peak where one wave runs at 0.68-0.73.
//
// Same strip (64 columns), same 128-row blocks, same LDS stages (KB = 32), DMA ring and per-element operation order
// as trsm_strip_kernel -- V, q and mu come out bit for bit the same.  The split is by ROWS: waves (cw, 0) and (cw, 1)
// share the 16 columns of column group cw (and therefore every B fragment); wave (cw, h) owns row tiles 4h .. 4h+3 of
// each block, i.e. half of every A tile.  Regular stages need nothing else.  In the diagonal stages the solved tiles
// of the upper half must reach the lower half's wave: the solver wave (h = 0 in diagonal stages 0 and 1) writes x_s,
// x_{s+1} over the two 16x16 inverses it has just consumed in the stage's B region (same size, layout = the B
// operand's), a mid-stage barrier publishes them, and wave (cw, 1) folds them into its four tiles (32 MFMAs) while the
// solver finishes its own updates.  q = sum V^2 and mu = V^T z keep the sequential per-lane order of the one-wave
// kernel: the running lane partials are handed from (cw, 0) to (cw, 1) in the middle of each block and back at its
// end through the padding columns of an LDS stage buffer (the stage barriers order the hand-over).
constexpr int kTH = kT / 2;               // row tiles per wave

template <bool SWEEP>
__global__ __launch_bounds__(512) void trsm_strip8_kernel(const double *__restrict__ U, int64_t ldu,
                                                          const double *__restrict__ invDt, double *V, int64_t ldv,
                                                          int n, const double *__restrict__ z,
                                                          double *__restrict__ q_out, double *__restrict__ mu_out,
                                                          int accumulate)
{
    constexpr int KB = 32;
    using G = StageGeom<KB>;
    constexpr int kABuf = G::kA, kBBuf = G::kB, kKS = G::kKS, kDS = G::kDiagStages, kDT = G::kDiagTiles;
    constexpr int kRows8 = KB / 8;                    // U rows a wave fetches per stage
    constexpr int kDma8 = kRows8 + 2;                 // LDS-DMA instructions a wave issues per stage (+ two B pieces)
    constexpr int kSolverStores = 4 * kDT;            // V stores the solver wave issues in a diagonal stage
    static_assert(kDT == 2 && kDS == 4 && kKS == 8 && kDma8 <= kKS - 2, "geometry the schedule below is written for");
    __shared__ __align__(16) double lds[kNBuf * (kABuf + kBBuf)];      // 159,744 B
    __shared__ __align__(16) double zl[SWEEP ? 2 * kRB : 2];           // z rows of the current and the next block

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave-uniform: LDS bases stay scalar
    const int cw = wave & 3, h = wave >> 2;                            // column group, row half (waves w, w+4 share a SIMD)
    const int lc = lane & 15, kq = lane >> 4;
    const int64_t colw = (int64_t)blockIdx.x * kStrip + cw * 16;       // first column of this wave's column group
    double *Vc = V + colw + lc;
    double *ldsB = lds + kNBuf * kABuf;
    // (q, mu) hand-over slots, two doubles per lane and column group: the 16 padding columns of the first U stage
    // buffer's 32 rows (the DMA writes 128 columns of each 144-double row, the padding is never touched)
    double *hand = lds + (cw * 8 + (lane >> 3)) * kLdsLd + kRB + 2 * (lane & 7);
    const unsigned lds_byte0 = lds_byte_address(lds);
    const double *ug = U + (int64_t)(wave * kRows8) * ldu + lane * 2;
    // B pieces (8 rows x 16 columns, 1 KiB): this wave fetches pieces 2h and 2h+1 of its column group
    const double *vg = V + (int64_t)((lane >> 3) + 16 * h) * ldv + colw + 2 * (lane & 7);
    const double *inv_lane = invDt + lane * 2 + 256 * h;               // diagonal stages: inverse h of the stage's two

    StageCursor ahead{0, 0, kDS, 0, 0, nullptr, nullptr, 0};
    auto locate_a = [&](StageCursor &c) __attribute__((always_inline)) {
        const bool past = c.i0 >= n;
        c.ai0 = past ? n - kRB : c.i0;
        c.aj = past ? (n - kRB) / KB + kDS - 1 : c.j;
    };
    auto locate_b = [&](StageCursor &c) __attribute__((always_inline)) {
        c.a_src = ug + (int64_t)(KB * c.aj) * ldu + c.ai0;
    };
    auto locate_c = [&](StageCursor &c) __attribute__((always_inline)) {
        const int nreg = c.ai0 / KB;
        const int64_t diag = (c.aj >= nreg) ? 1 : 0;
        const int64_t off_diag = ((int64_t)(c.ai0 / 16) + kDT * (c.aj - nreg)) * 256;
        const int64_t off_reg = (int64_t)(KB * c.aj) * ldv;
        const uintptr_t base = (uintptr_t)vg + ((uintptr_t)inv_lane - (uintptr_t)vg) * (uintptr_t)diag;
        c.b_src = reinterpret_cast<const double *>(base) + (off_reg + (off_diag - off_reg) * diag);
        c.b_stride = 8 * ldv + (128 - 8 * ldv) * diag;
    };
    auto step = [&](StageCursor &c) __attribute__((always_inline)) {
        const int wrap = (c.j + 1 == c.lim) ? 1 : 0;
        c.i0 += kRB * wrap;
        c.j = (c.j + 1) * (1 - wrap);
        c.lim = c.lim + (c.i0 / KB + kDS - c.lim) * wrap;
    };
    auto advance = [&](StageCursor &c) __attribute__((always_inline)) {
        step(c);
        locate_a(c);
        locate_b(c);
        locate_c(c);
    };
    // DMA instruction i of the wave's kDma8 for the stage the cursor points at: i < kRows8: U row wave*kRows8 + i;
    // then the two B pieces
    auto issue_one = [&](const StageCursor &c, int buf, int i) __attribute__((always_inline)) {
        if (i < kRows8) {
            const unsigned la = __builtin_amdgcn_readfirstlane(
                lds_byte0 + 8u * (unsigned)(buf * kABuf + (wave * kRows8) * kLdsLd));
            glds16(c.a_src + (int64_t)i * ldu, la + 8u * (unsigned)(i * kLdsLd));
        } else {
            const unsigned lb = __builtin_amdgcn_readfirstlane(
                lds_byte0 + 8u * (unsigned)(kNBuf * kABuf + buf * kBBuf + cw * (KB * 16) + (2 * h) * 128));
            glds16(c.b_src + (i - kRows8) * c.b_stride, lb + 8u * (unsigned)((i - kRows8) * 128));
        }
    };
    auto issue_stage = [&](const StageCursor &c, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < kDma8; ++i) issue_one(c, buf, i);
    };

    // acc holds the NEGATED residual of this wave's four row tiles
    d4 acc[kTH];
#pragma unroll
    for (int t = 0; t < kTH; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = -Vc[(int64_t)(16 * (kTH * h + t) + kq + 4 * r) * ldv];
    // What a block needs from global memory besides its DMA stream: the z rows of the tiles this wave will solve and the
    // block's right-hand sides (K* rows).  vmcnt retires in order and the LDS-DMA instructions are invisible to the
    // compiler's own counting, so a compiler-visible load consumed inside the steady state turns the compiler's wait
    // for it into a wait for the DMA of two stages ahead as well -- a full HBM round trip per diagonal stage (4-5000
    // cycles each in the first timeline of this kernel).  So these loads are issued by hand too, one block ahead, into
    // AGPRs (which the register allocator never moves), right after the DMA issue of diagonal stage 0; the stage-top
    // waits below count them (see wait_top), and they are read back once diagonal stage 3's top has retired them.
    // z travels through LDS: every wave requests the next block's 128 rows (two per lane) with the other ahead loads and
    // writes them to the half of `zl` the current block does not read (all eight waves write the same 1 KiB: no
    // rendezvous beyond the stage barriers); the solver reads its rows from there.
    typedef double d2 __attribute__((ext_vector_type(2)));
    if (SWEEP) {
        const d2 z0 = *reinterpret_cast<const d2 *>(z + 2 * lane);
        *reinterpret_cast<d2 *>(zl + 2 * lane) = z0;           // block 0; published by the first stage barrier
    }
    double accn[kTH][4];                  // next block's right-hand sides, in flight
    d2 zn;                                // next block's z rows 2 lane, 2 lane + 1, in flight
    constexpr int kAhead = kTH * 4 + (SWEEP ? 1 : 0);         // hand-issued loads per block
    // load number sl (0 .. kAhead-1) for the block after the one at i0; the last block asks for its own rows again (no
    // branch around a load: one definition, one use)
    auto request_one = [&](int i0, int sl) __attribute__((always_inline)) {
        const int nb = (i0 + kRB < n) ? i0 + kRB : i0;
        if (sl < kTH * 4) {
            const int t = sl >> 2, r = sl & 3;
            const int row = nb + 16 * (kTH * h + t) + kq + 4 * r;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(accn[t][r]) : "v"(Vc + (int64_t)row * ldv) : "memory");
        } else if (SWEEP) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(zn) : "v"(z + nb + 2 * lane) : "memory");
        }
    };

    // block 0 has no regular stage to spread block 1's ahead loads over: they go first, ahead of the whole DMA stream
    // (the first stage top retires them)
#pragma unroll
    for (int sl = 0; sl < kAhead; ++sl) request_one(0, sl);
    locate_a(ahead);
    locate_b(ahead);
    locate_c(ahead);
    issue_stage(ahead, 0);
    advance(ahead);
    issue_stage(ahead, 1);
    advance(ahead);

    int buf = 0;
    double qacc = 0.0, macc = 0.0, qtot = 0.0, mtot = 0.0;       // totals live in the h = 1 waves
    if (SWEEP && accumulate && h == 1) {
        qtot = q_out[colw + lc];
        mtot = mu_out[colw + lc];
    }

    // Stage-top wait.  This wave's DMA of stage k (issued during stage k-2) has landed once only what is younger than its
    // LAST instruction may still be in flight (vmcnt retires in order): what stage k-2 issued after its DMA (a2), and all
    // of stage k-1 -- what it issued before or among its DMA instructions (b1), the DMA (kDma8), what it issued after
    // (a1).  "After" are the solver's V stores of a diagonal stage; "before" are the
    // hand-issued loads for the next block, spread over the first regular stage of a block.
    int a1 = 0, b1 = 0, a2 = 0;
    auto wait_top = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int extra = a1 + b1 + a2;
#define WAIT_IF(x) if (extra == (x)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma8 + (x)) : "memory")
        if (extra == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma8) : "memory");
        else WAIT_IF(kSolverStores);
        else WAIT_IF(2 * kSolverStores);
        else WAIT_IF(kAhead);
        else WAIT_IF(kAhead + kSolverStores);
        else WAIT_IF(kAhead + 2 * kSolverStores);
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma8) : "memory");       // (any other count: the strict wait)
#undef WAIT_IF
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        a2 = a1;
        a1 = 0;
        b1 = 0;
    };
    static_assert(kDma8 + kAhead + 2 * kSolverStores <= 63, "s_waitcnt vmcnt is a 6-bit count");
#define STAGE8_TOP() wait_top()

    for (int i0 = 0; i0 < n; i0 += kRB) {
        const int nst = i0 / KB;
        double af[2][kTH], bf[2];
        bool deferred = false;                // af[1]/bf[1] hold the previous stage's last k-step, MFMAs not yet issued
        // one regular stage; FIRST: the first one of a block, which also carries the hand-issued loads for the next block,
        // one per MFMA slot ahead of the stage's last DMA instruction
        auto regular_stage = [&](auto first_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            STAGE8_TOP();
            const int bnext = (buf >= 1) ? buf - 1 : 2;       // (buf + 2) % 3
            const double *abase = lds + buf * kABuf + kq * kLdsLd + lc + 64 * h;
            const double *bbase = ldsB + buf * kBBuf + cw * (KB * 16) + kq * 16 + lc;
            // one small piece of the other work after every MFMA, order pinned
#pragma unroll
            for (int t = 0; t < kTH; ++t) {
                if (deferred) acc[t] = MFMA_F64(af[1][t], bf[1], acc[t]);
                if (t < kTH / 2) {
                    af[0][2 * t] = abase[32 * t];
                    af[0][2 * t + 1] = abase[32 * t + 16];
                } else if (t == kTH / 2) {
                    bf[0] = bbase[0];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int jj = 0; jj < kKS - 1; ++jj) {
                const double *an = abase + 4 * (jj + 1) * kLdsLd;
#pragma unroll
                for (int t = 0; t < kTH; ++t) {
                    acc[t] = MFMA_F64(af[jj & 1][t], bf[jj & 1], acc[t]);
                    if (t < kTH / 2) {
                        af[(jj + 1) & 1][2 * t] = an[32 * t];
                        af[(jj + 1) & 1][2 * t + 1] = an[32 * t + 16];
                    } else if (t == kTH / 2) {
                        bf[(jj + 1) & 1] = bbase[4 * (jj + 1) * 16];
                    } else if (jj < kDma8) {
                        issue_one(ahead, bnext, jj);                      // stage g+2's DMA rides under the MFMAs
                    }
                    if (FIRST && t < 3 && 3 * jj + t < kAhead) request_one(i0, 3 * jj + t);
                    if (jj == kDma8) {                                    // cursor bookkeeping under the MFMAs too
                        if (t == 0) step(ahead);
                        if (t == 1) locate_a(ahead);
                        if (t == 2) locate_b(ahead);
                        if (t == 3) locate_c(ahead);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (FIRST) b1 += kAhead;
            deferred = true;                                              // the last k-step sits in af[1], bf[1]
            buf = (buf == 2) ? 0 : buf + 1;
        };
        static_assert(3 * (kDma8 - 1) + 2 >= kAhead - 1, "every ahead load has a slot before the stage's last DMA instruction");
        if (nst > 0) regular_stage(std::true_type{});
        for (int j = 1; j < nst; ++j) regular_stage(std::false_type{});
        if (deferred) {
#pragma unroll
            for (int t = 0; t < kTH; ++t) acc[t] = MFMA_F64(af[1][t], bf[1], acc[t]);
        }

        // ---- diagonal stages (two 16-row tiles each): tiles 2m, 2m+1 belong to the waves of half m >> 1
#pragma unroll
        for (int m = 0; m < kDS; ++m) {
            STAGE8_TOP();
            const int hs = m >> 1;                            // the solving half
            const int ls = kDT * (m & 1);                     // its first tile of the stage, as an index into acc[]
            const bool solver = (h == hs);
            if (SWEEP && solver && (m == 2 || (m == 0 && ((i0 / kRB) & 1)))) {     // take over the running lane partials
                qacc = hand[0];
                macc = hand[1];
            }
            const int bnext = (buf >= 1) ? buf - 1 : 2;
            // the stage's DMA issue and cursor arithmetic: placed where the wave would otherwise wait (behind the first
            // MFMAs of the solve, ahead of the lower half's rendezvous), always ahead of the stage's V stores
            auto stage_dma = [&]() __attribute__((always_inline)) {
                issue_stage(ahead, bnext);
                advance(ahead);
            };
            const double *abase0 = lds + buf * kABuf + kq * kLdsLd + lc;          // U tile of the stage, all 128 columns
            double *xreg = ldsB + buf * kBBuf + cw * (KB * 16);                   // inverses in, solved tiles out
            if (solver) {
                double iv[kDT][4], uf[kDT][kTH][4], zr[kDT][4];
                // the first tile's inverse first: its solve starts as soon as these four values are in; every other LDS
                // operand of the stage is requested behind the solve's first MFMAs
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) iv[0][kk] = xreg[(4 * kk + kq) * 16 + lc];
                auto fetch_rest = [&]() __attribute__((always_inline)) {
#pragma unroll
                    for (int hh = 0; hh < kDT; ++hh) {
                        if (hh > 0) {
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk) iv[hh][kk] = xreg[hh * 256 + (4 * kk + kq) * 16 + lc];
                        }
#pragma unroll
                        for (int t = ls + hh + 1; t < kTH; ++t)
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk)
                                uf[hh][t][kk] = abase0[(16 * hh + 4 * kk) * kLdsLd + 16 * (kTH * hs + t)];
                    }
                    if (SWEEP) {
                        const double *zrow = zl + ((i0 / kRB) & 1) * kRB + KB * m + kq;
#pragma unroll
                        for (int hh = 0; hh < kDT; ++hh)
#pragma unroll
                            for (int r = 0; r < 4; ++r) zr[hh][r] = zrow[16 * hh + 4 * r];
                    }
                };
                auto emit = [&](int hh, int s, const d4 &x) __attribute__((always_inline)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = i0 + 16 * s + kq + 4 * r;
                        Vc[(int64_t)row * ldv] = x[r];
                        if (SWEEP) {
                            qacc = fma(x[r], x[r], qacc);
                            macc = fma(x[r], zr[hh][r], macc);
                        }
                    }
                };
                // the solved tile, in the B-operand layout [row][column], where the inverse just used stood
                auto publish = [&](int hh, const d4 &x) __attribute__((always_inline)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) xreg[hh * 256 + (kq + 4 * r) * 16 + lc] = x[r];
                };
                const int s = kDT * m;
                d4 x = {0.0, 0.0, 0.0, 0.0}, x2 = {0.0, 0.0, 0.0, 0.0};
                x = MFMA_F64(iv[0][0], -acc[ls][0], x);
                x2 = MFMA_F64(iv[0][1], -acc[ls][1], x2);
                x = MFMA_F64(iv[0][2], -acc[ls][2], x);
                x2 = MFMA_F64(iv[0][3], -acc[ls][3], x2);
                __builtin_amdgcn_sched_barrier(0);
                fetch_rest();                                             // under the four MFMAs in flight
                __builtin_amdgcn_sched_barrier(0);
                x += x2;
                asm volatile("" : "+v"(x));
                if (hs == 0) {
                    publish(0, x);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                         // the lower half may read x now
                }
                // the update of the next tile is a chain of four dependent MFMAs: the stage's DMA issue and cursor
                // arithmetic sit in its gaps (and ahead of the stage's V stores, as the stage-top accounting assumes)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    acc[ls + 1] = MFMA_F64(uf[0][ls + 1][kk], x[kk], acc[ls + 1]);
                    if (ls + 2 < kTH) acc[ls + 2] = MFMA_F64(uf[0][ls + 2][kk], x[kk], acc[ls + 2]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kk < 3) {
                        issue_one(ahead, bnext, 2 * kk);
                        issue_one(ahead, bnext, 2 * kk + 1);
                    } else {
                        advance(ahead);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                emit(0, s, x);
                const d4 na = -acc[ls + 1];
                d4 y1 = {0.0, 0.0, 0.0, 0.0}, y2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (kk & 1) y2 = MFMA_F64(iv[1][kk], na[kk], y2);
                    else y1 = MFMA_F64(iv[1][kk], na[kk], y1);
#pragma unroll
                    for (int t = ls + 3; t < kTH; ++t) acc[t] = MFMA_F64(uf[0][t][kk], x[kk], acc[t]);
                }
                d4 y = y1 + y2;
                asm volatile("" : "+v"(y));
                if (hs == 0) {
                    publish(1, y);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                         // ... and y
                }
                emit(1, s + 1, y);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int t = ls + 2; t < kTH; ++t) acc[t] = MFMA_F64(uf[1][t][kk], y[kk], acc[t]);
                }
                a1 += kSolverStores;
                if (SWEEP && (m & 1)) {                                   // done with this half's tiles: hand over
                    if (hs == 0) {
                        hand[0] = qacc;
                        hand[1] = macc;
                        qacc = 0.0;                                       // (an even block starts from zero)
                        macc = 0.0;
                    } else if (((i0 / kRB) & 1) || i0 + kRB >= n) {       // end of a block pair: reduce, add
                        qacc += __shfl_xor(qacc, 16);
                        qacc += __shfl_xor(qacc, 32);
                        macc += __shfl_xor(macc, 16);
                        macc += __shfl_xor(macc, 32);
                        qtot += qacc;
                        mtot += macc;
                        qacc = 0.0;
                        macc = 0.0;
                    } else {
                        hand[0] = qacc;
                        hand[1] = macc;
                    }
                }
            } else if (hs == 0) {
                // lower half while the upper half solves: its U fragments against the two tiles first, then -- as the
                // solver publishes them -- 16 MFMAs per tile
                double uf[kDT][kTH][4], xb[kDT][4];
#pragma unroll
                for (int hh = 0; hh < kDT; ++hh)
#pragma unroll
                    for (int t = 0; t < kTH; ++t)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            uf[hh][t][kk] = abase0[(16 * hh + 4 * kk) * kLdsLd + 16 * (kTH + t)];
                stage_dma();
#pragma unroll
                for (int hh = 0; hh < kDT; ++hh) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                         // tile hh of the stage is published
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) xb[hh][kk] = xreg[hh * 256 + (4 * kk + kq) * 16 + lc];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int t = 0; t < kTH; ++t) acc[t] = MFMA_F64(uf[hh][t][kk], xb[hh][kk], acc[t]);
                }
            } else {
                stage_dma();                                              // upper half, nothing left to solve in this block
            }
            buf = (buf == 2) ? 0 : buf + 1;
        }
        // the hand-issued loads were retired by diagonal stage 3's top (they are older than that stage's DMA)
#pragma unroll
        for (int t = 0; t < kTH; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                asm volatile("" : "+v"(accn[t][r]));
                acc[t][r] = -accn[t][r];
            }
        if (SWEEP) {
            asm volatile("" : "+v"(zn));
            *reinterpret_cast<d2 *>(zl + (((i0 / kRB) + 1) & 1) * kRB + 2 * lane) = zn;
        }
    }
#undef STAGE8_TOP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the clamped tail DMA before the LDS goes away
    __builtin_amdgcn_s_barrier();

    if (SWEEP && h == 1 && kq == 0) {
        q_out[colw + lc] = qtot;
        mu_out[colw + lc] = mtot;
    }
}

// ------------------------------------------------------------------------------------------------
// The strip kernel on 256-row PAIR blocks (round 5).  What a 128-row block of trsm_strip8_kernel costs beyond its regular
// stages is its diagonal phase -- four stages with 60 / 44 / 28 / 12 MFMAs per SIMD between the same barriers and DMA round
// trips as a regular stage's 64, in two of which one wave of every SIMD has nothing to do -- and the refill of the DMA ring
// behind it.  Here two row blocks (b, b+1) go through the ring together and the 128 x 128 square between them is folded
// into the diagonal phase, where it is the work that keeps the matrix pipe busy:
//
//   ownership   wave (cw, h) owns ALL eight row tiles of block b + h for the 16 columns of column group cw (strip8: half
//               of every block).  Both waves of a SIMD still share every B fragment.
//   regular     stages of 16 U rows x 256 columns (the pair's columns): 4 k-steps x 8 tiles = 32 MFMAs per wave, as many
//               as strip8's 8 k-steps x 4 tiles, with ONE B fragment per 8 MFMAs instead of per 4 and half the V re-read
//               traffic (a V row is fetched once per pair of blocks).  Ring of three slots of 43,008 B.
//   D0          eight stages, one 16-row tile of block b each (same U tile shape as a regular stage: the block's rows x
//               the pair's columns).  Wave (cw, 0) solves: x_s = inv(L_ss) r_s, publishes x_s over the inverse it has
//               just consumed (mid-stage barrier), updates its own tiles below.  Wave (cw, 1) folds x_s into the eight
//               tiles of block b+1 -- 32 independent MFMAs per tile, the last k-step deferred past the next stage top so
//               that the solver's serial head (inverse fetch, two dependent MFMA pairs, publication) is covered.
//   D1          four stages, two tiles of block b+1 each (slot row r = U rows r and r + 16 of the stage, the block's own
//               128 columns): wave (cw, 1) solves alone, the arithmetic of trsm_strip_kernel's two-tile diagonal stage;
//               nothing is published.  Wave (cw, 0) has no rows left in the pair: it issues its share of the DMA and
//               requests its next block's right-hand sides.
// Per element the operations and their order are those of trsm_strip_kernel / trsm_strip8_kernel / trsm_update_kernel
// (accumulation over k ascending in k-steps of 4, the solve as two half-sums, updates tile by tile): V, q and mu come out
// bit for bit the same.  The running (q, mu) lane partials pass from (cw, 0) to (cw, 1) once per pair.
// n must be a multiple of 256 (launch_trsm_strips falls back to trsm_strip8_kernel otherwise).
constexpr int kPB = 2 * kRB;                  // rows per pair block
constexpr int kPKB = 16;                      // U rows per stage
constexpr int kPLd = kPB + 16;                // U-tile row stride (doubles): k rows kq, kq+1 land 32 banks apart
constexpr int kPA = kPKB * kPLd;              // doubles per U stage buffer (34,816 B)
constexpr int kPBd = 4 * kPKB * 16;           // doubles per B stage buffer (4 column groups x [16 k][16 cols], 8,192 B)
constexpr int kPDma = 5;                      // LDS-DMA instructions per wave and stage: 4 x 1 KiB of U, one B piece
constexpr int kPD0 = kRB / 16;                // diagonal stages of the pair's first block (one tile each)
constexpr int kPD1 = kRB / 32;                // ... of its second block (two tiles each)

struct PairCursor {
    int i0, j, lim;                           // pair origin, stage index within the pair, stages in the pair
    int ai0, aj;                              // clamped to the last stage once the cursor is past the end
    const double *a_src;                      // per-lane source of this wave's first U piece
    int64_t a_cstride;                        // doubles between the two pieces of a slot row
    const double *b_src;                      // per-lane source of this wave's B piece
    unsigned b_dst;                           // its place in the B stage buffer (doubles)
};

template <bool SWEEP>
__global__ __launch_bounds__(512) void trsm_pair_kernel(const double *__restrict__ U, int64_t ldu,
                                                        const double *__restrict__ invDt, double *V, int64_t ldv,
                                                        int n, const double *__restrict__ z,
                                                        double *__restrict__ q_out, double *__restrict__ mu_out,
                                                        int accumulate)
{
    __shared__ __align__(16) double lds[kNBuf * (kPA + kPBd)];         // 129,024 B
    __shared__ __align__(16) double zl[SWEEP ? 2 * kPB : 2];           // z rows of the current and the next pair
    __shared__ __align__(16) double handbuf[SWEEP ? 4 * 64 * 2 : 2];   // (q, mu) lane partials, (cw, 0) -> (cw, 1)

#ifdef CBO_DIAG_KNOBS
    // timing-only (CBO_HIP_STRIP_MASK=256, diagnostic build): waves 0 and 4 of workgroup 0 stamp every stage -- slot 0 the
    // stage top passed, 1 the solved tile published / final, 2 the stage's last instruction issued (scripts/pair_timeline.py)
    const int dmask = accumulate >> 8;
    accumulate &= 1;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave-uniform: LDS bases stay scalar
    const int cw = wave & 3, h = wave >> 2;                            // column group, block of the pair (waves w, w+4 share a SIMD)
    const int lc = lane & 15, kq = lane >> 4;
    const int64_t colw = (int64_t)blockIdx.x * kStrip + cw * 16;       // first column of this wave's column group
    // V addresses are "per-lane base + wave-uniform row offset": the row offsets stay on the scalar unit and nothing per row
    // is kept in vector registers across the pair loop
    double *Vq = V + colw + lc + (int64_t)kq * ldv;                    // row kq of the lane's column
    double *ldsB = lds + kNBuf * kPA;
    double *hand = handbuf + (cw * 64 + lane) * 2;
    const unsigned lds_byte0 = lds_byte_address(lds);
    const double *ug = U + (int64_t)(2 * wave) * ldu + lane * 2;       // slot rows 2 wave, 2 wave + 1
    const double *vg = V + (int64_t)((lane >> 3) + 8 * h) * ldv + colw + 2 * (lane & 7);   // B piece h: rows 8h .. 8h+7
    const double *inv_d0 = invDt + 128 * h + lane * 2;                 // D0: half h of the stage's inverse
    const double *inv_d1 = invDt + 256 * (cw >> 1) + 128 * (cw & 1) + lane * 2;            // D1: piece cw of its two

    PairCursor ahead{0, 0, kPD0 + kPD1, 0, 0, nullptr, 0, nullptr, 0u};
    auto locate_a = [&](PairCursor &c) __attribute__((always_inline)) {
        const bool past = c.i0 >= n;
        c.ai0 = past ? n - kPB : c.i0;
        c.aj = past ? (n - kPB) / kPKB + kPD0 + kPD1 - 1 : c.j;
    };
    auto locate_b = [&](PairCursor &c) __attribute__((always_inline)) {
        const int m1 = c.aj - c.ai0 / kPKB - kPD0;                     // >= 0: stage m1 of D1
        const bool d1 = m1 >= 0;
        const int64_t row = d1 ? (int64_t)(c.ai0 + kRB + 32 * m1) : (int64_t)kPKB * c.aj;
        const int64_t col = d1 ? (int64_t)(c.ai0 + kRB) : (int64_t)c.ai0;
        c.a_src = ug + row * ldu + col;
        c.a_cstride = d1 ? 16 * ldu : (int64_t)kRB;
    };
    auto locate_c = [&](PairCursor &c) __attribute__((always_inline)) {
        const int nreg = c.ai0 / kPKB;
        const int m1 = c.aj - nreg - kPD0;
        const bool d1 = m1 >= 0, reg = c.aj < nreg;
        const int64_t tile = d1 ? (int64_t)(nreg + kPD0 + 2 * m1) : (int64_t)c.aj;
        const double *inv_p = (d1 ? inv_d1 : inv_d0) + tile * 256;
        const double *v_p = vg + (int64_t)(kPKB * c.aj) * ldv;
        c.b_src = reg ? v_p : inv_p;
        c.b_dst = d1 ? (unsigned)(cw * 128) : (unsigned)(cw * 256 + 128 * h);
    };
    auto step = [&](PairCursor &c) __attribute__((always_inline)) {
        const int wrap = (c.j + 1 == c.lim) ? 1 : 0;
        c.i0 += kPB * wrap;
        c.j = (c.j + 1) * (1 - wrap);
        c.lim = c.lim + (c.i0 / kPKB + kPD0 + kPD1 - c.lim) * wrap;
    };
    auto advance = [&](PairCursor &c) __attribute__((always_inline)) {
        step(c);
        locate_a(c);
        locate_b(c);
        locate_c(c);
    };
    // DMA instruction i of the wave's kPDma for the stage the cursor points at: i < 4: piece i & 1 of slot row
    // 2 wave + (i >> 1); i = 4: the B piece
    auto issue_one = [&](const PairCursor &c, int buf, int i) __attribute__((always_inline)) {
        if (i < 4) {
            const unsigned la = __builtin_amdgcn_readfirstlane(
                lds_byte0 + 8u * (unsigned)(buf * kPA + (2 * wave + (i >> 1)) * kPLd + kRB * (i & 1)));
            glds16(c.a_src + (int64_t)(i >> 1) * ldu + (int64_t)(i & 1) * c.a_cstride, la);
        } else {
            const unsigned lb = __builtin_amdgcn_readfirstlane(lds_byte0 + 8u * ((unsigned)(kNBuf * kPA + buf * kPBd) + c.b_dst));
            glds16(c.b_src, lb);
        }
    };
    auto issue_stage = [&](const PairCursor &c, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < kPDma; ++i) issue_one(c, buf, i);
    };

    // One code path per role (H = 0: the waves of block b, H = 1: of block b + 1), the whole pair loop: the roles keep different
    // things in registers through the diagonal phase, and two straight pipelines are what the register allocator handles
    // without copies or spills.
    auto run = [&](auto role_tag) __attribute__((always_inline)) {
        constexpr int H = decltype(role_tag)::value;
        // acc holds the NEGATED residual of this wave's eight row tiles (block b + h)
        d4 acc[kT];
#pragma unroll
        for (int t = 0; t < kT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = -Vq[(int64_t)(kRB * H + 16 * t + 4 * r) * ldv];
        if (SWEEP) {
            const d2 z0 = *reinterpret_cast<const d2 *>(z + kRB * H + 2 * lane);
            *reinterpret_cast<d2 *>(zl + kRB * H + 2 * lane) = z0;          // pair 0; published by the first stage barrier
        }
        // The next pair's right-hand sides and z rows are requested by hand-issued loads (invisible to the compiler's vmcnt
        // counting like the LDS-DMA instructions, see trsm_strip8_kernel) during D1 -- wave (cw, 0) in its first stage, where it
        // has nothing else to do, wave (cw, 1) in its third, when half of its tiles are solved -- ahead of that stage's DMA, so
        // that the stage top two stages later retires them; they are read back behind the first stage top of the next pair.
        double accn[kT][4];
        d2 zn;
        constexpr int kAhead = kT * 4 + (SWEEP ? 1 : 0);
        // (scalar base + 32-bit lane offset: one vector register of addressing for all of them)
        const unsigned vq_off = (unsigned)(((int64_t)kq * ldv + lc) * 8);
        const unsigned z_off = (unsigned)(lane * 16);
        // in three batches, one per stage of D1's first three (a burst of all of them at once waits for slots in the memory
        // pipeline behind the DMA ring's ten instructions: 2900 cycles of the solver's stage in the first timeline)
        // one at a time, each behind an MFMA of the solver (or a pause of the idle wave): a burst of them at a stage top waits
        // for room in the CU's memory pipeline behind the DMA ring -- 2300-2900 cycles per burst, of 11 or of 33, in the first
        // timelines.  D1's first stage issues kL0 of them, its second the rest, both AFTER the stage's DMA: the top of D1's
        // fourth stage retires the first lot, the first top of the next pair the second.
        constexpr int kL0 = 16, kL1 = kAhead - kL0;
        auto request_one = [&](int i0, int sl) __attribute__((always_inline)) {
            const int nb = (i0 + kPB < n) ? i0 + kPB : i0;               // (the last pair asks for its own rows again: no branch)
            const double *vb = V + colw + (int64_t)(nb + kRB * H) * ldv;   // wave-uniform
            if (sl < kT * 4)
                asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(accn[sl >> 2][sl & 3]) : "v"(vq_off), "s"(vb + (int64_t)(4 * sl) * ldv) : "memory");
            else if (SWEEP)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(zn) : "v"(z_off), "s"(z + nb + kRB * H) : "memory");
        };

        locate_a(ahead);
        locate_b(ahead);
        locate_c(ahead);
        issue_stage(ahead, 0);
        advance(ahead);
        issue_stage(ahead, 1);
        advance(ahead);

        int buf = 0;
        double qacc = 0.0, macc = 0.0, qtot = 0.0, mtot = 0.0;       // totals live in the h = 1 waves
        if (SWEEP && accumulate && H == 1) {
            qtot = q_out[colw + lc];
            mtot = mu_out[colw + lc];
        }

        // Stage-top wait, as in trsm_strip8_kernel: this wave's DMA of stage k (issued during stage k-2) has landed once only
        // what is younger than its last instruction may still be in flight: what stage k-2 issued after its DMA (a2: the
        // solver's V stores), and all of stage k-1 -- what it issued ahead of its DMA (b1: the hand-issued loads), the DMA, what
        // it issued after (a1).
#ifdef CBO_DIAG_KNOBS
        const bool stamp_on = SWEEP && (dmask & 256) && blockIdx.x == 0 && lane == 0 && cw == 0;
        int stamp_i = 0;
#define PSTAMP(slot)                                                                                       \
        do {                                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if (stamp_on && stamp_i < 4096) g_trsm_stamps[8 * stamp_i + 4 * H + (slot)] = __builtin_amdgcn_s_memtime(); \
            __builtin_amdgcn_sched_barrier(0);                                                             \
        } while (0)
#define PSTAMP_NEXT() do { if (stamp_on) ++stamp_i; } while (0)
#else
#define PSTAMP(slot)
#define PSTAMP_NEXT()
#endif
        int a1 = 0, b1 = 0, a2 = 0;
        auto wait_top = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int extra = a1 + b1 + a2;
#define WAIT_IF(x) if (extra == (x)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPDma + (x)) : "memory")
            if (extra == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPDma) : "memory");
            else WAIT_IF(4);
            else WAIT_IF(8);
            else WAIT_IF(16);
            else WAIT_IF(kL1);
            else WAIT_IF(kL0 + 4);
            else WAIT_IF(kL0 + 8);
            else WAIT_IF(kL0 + kL1);
            else WAIT_IF(kL1 + 16);
            else WAIT_IF(kL0 + kL1 + 16);
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPDma) : "memory");       // (any other count: the strict wait)
#undef WAIT_IF
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            a2 = a1;
            a1 = 0;
            b1 = 0;
            PSTAMP(0);
        };
        static_assert(kPDma + kL0 + kL1 + 16 <= 63 && kL1 >= kL0, "s_waitcnt vmcnt is a 6-bit count; the cases of wait_top");

        double af[2][kT], bf[2];
        // ---- regular stages: U rows [16 j, 16 j + 16) against the pair's 256 columns; the last k-step's MFMAs are issued
        // behind the next stage top (barrier skew and the first LDS reads hide under them)
        auto regular_stage = [&](int i0, auto first_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const int zpar = ((i0 / kPB) & 1) * kPB;                 // this pair's half of zl
            wait_top();
            if (FIRST) {
                // the hand-issued loads of the previous pair's D1 were retired by this wait
#pragma unroll
                for (int t = 0; t < kT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        asm volatile("; ahead-pin %0" : "+v"(accn[t][r]));      // (named: scripts/check_hand_issued_loads.py)
                        acc[t][r] = -accn[t][r];
                    }
                if (SWEEP) {
                    asm volatile("; ahead-pin %0" : "+v"(zn));
                    *reinterpret_cast<d2 *>(zl + zpar + kRB * H + 2 * lane) = zn;
                }
            }
            const int bnext = (buf >= 1) ? buf - 1 : 2;       // (buf + 2) % 3
            const double *abase = lds + buf * kPA + kq * kPLd + kRB * H + lc;
            const double *bbase = ldsB + buf * kPBd + cw * 256 + kq * 16 + lc;
#pragma unroll
            for (int t = 0; t < kT; ++t) {
                if (!FIRST) acc[t] = MFMA_F64(af[1][t], bf[1], acc[t]);
                if (t < kT / 2) {
                    af[0][2 * t] = abase[32 * t];
                    af[0][2 * t + 1] = abase[32 * t + 16];
                } else if (t == kT / 2) {
                    bf[0] = bbase[0];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const double *an = abase + 4 * (jj + 1) * kPLd;
#pragma unroll
                for (int t = 0; t < kT; ++t) {
                    acc[t] = MFMA_F64(af[jj & 1][t], bf[jj & 1], acc[t]);
                    if (t < kT / 2) {
                        af[(jj + 1) & 1][2 * t] = an[32 * t];
                        af[(jj + 1) & 1][2 * t + 1] = an[32 * t + 16];
                    } else if (t == kT / 2) {
                        bf[(jj + 1) & 1] = bbase[4 * (jj + 1) * 16];
                    } else {
                        const int slot = 3 * jj + (t - (kT / 2 + 1));       // nine slots: five DMA instructions, the cursor
                        if (slot < kPDma) issue_one(ahead, bnext, slot);
                        else if (slot == kPDma) step(ahead);
                        else if (slot == kPDma + 1) locate_a(ahead);
                        else if (slot == kPDma + 2) locate_b(ahead);
                        else locate_c(ahead);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            PSTAMP(2); PSTAMP_NEXT(); buf = (buf == 2) ? 0 : buf + 1;                   // the last k-step sits in af[1], bf[1]
        };
        // the diagonal phase of the pair at i0; P0: the first pair, which has no regular stage before it (nothing deferred)
        auto diag_phase = [&](int i0, auto first_pair_tag) __attribute__((always_inline)) {
            constexpr bool P0 = decltype(first_pair_tag)::value;
            const int zpar = ((i0 / kPB) & 1) * kPB;                     // this pair's half of zl
            // ---- the diagonal phase, one code path per role (the roles' live registers differ: (cw, 0) carries its next block's
            // right-hand sides through D1, (cw, 1) the fragments of its solve)
            // D0: tile m of block b per stage; (cw, 0) solves, (cw, 1) folds into block b + 1
            // D1: tiles 2m, 2m+1 of block b + 1 per stage; (cw, 1) solves alone
            if constexpr (H == 0) {
#pragma unroll
                for (int m = 0; m < kPD0; ++m) {
                    wait_top();
                    const int bnext = (buf >= 1) ? buf - 1 : 2;
                    const double *abase0 = lds + buf * kPA + kq * kPLd + lc;              // U tile of the stage, all 256 columns
                    double *xreg = ldsB + buf * kPBd + cw * 256;                          // the inverse in, the solved tile out
                    double iv[4], uf[kT][4], zr[4];
                    __builtin_amdgcn_s_setprio(2);          // the solve is the stage's serial head: ahead of (cw, 1)'s deferred MFMAs
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) iv[kk] = xreg[(4 * kk + kq) * 16 + lc];
                    // the regular stages' deferred k-step: the tile about to be solved first
                    if (m == 0 && !P0) acc[0] = MFMA_F64(af[1][0], bf[1], acc[0]);
                    d4 x = {0.0, 0.0, 0.0, 0.0}, x2 = {0.0, 0.0, 0.0, 0.0};
                    x = MFMA_F64(iv[0], -acc[m][0], x);
                    x2 = MFMA_F64(iv[1], -acc[m][1], x2);
                    if (m == 0 && !P0) {
#pragma unroll
                        for (int t = 1; t < kT; ++t) acc[t] = MFMA_F64(af[1][t], bf[1], acc[t]);
                    }
                    x = MFMA_F64(iv[2], -acc[m][2], x);
                    x2 = MFMA_F64(iv[3], -acc[m][3], x2);
                    __builtin_amdgcn_sched_barrier(0);
                    x += x2;
                    asm volatile("" : "+v"(x));
#pragma unroll
                    for (int r = 0; r < 4; ++r) xreg[(kq + 4 * r) * 16 + lc] = x[r];      // B-operand layout [row][column]
                    PSTAMP(3);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                                       // (cw, 1) may read x now
                    __builtin_amdgcn_s_setprio(0);
                    PSTAMP(1);
                    // Its own U fragments only now: the eight waves read the whole 16 x 256 tile once per column group in this
                    // stage -- 128 KB, a thousand cycles of the LDS -- and ahead of the publication those reads were the stage's
                    // serial head (1400-2300 cycles from the stage top to this barrier in the first timeline, (cw, 1) waiting).
                    // The two tiles of the dependent chain first.
#pragma unroll
                    for (int t = m + 1; t < kT; ++t)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) uf[t][kk] = abase0[(4 * kk) * kPLd + 16 * t];
                    if (SWEEP) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) zr[r] = zl[zpar + 16 * m + kq + 4 * r];
                    }
                    // the next tile's update is a chain of four dependent MFMAs: the stage's DMA issue and the cursor
                    // arithmetic sit in its gaps, ahead of the stage's V stores (as the stage-top accounting assumes)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        if (m + 1 < kT) acc[m + 1] = MFMA_F64(uf[m + 1][kk], x[kk], acc[m + 1]);
                        if (m + 2 < kT) acc[m + 2] = MFMA_F64(uf[m + 2][kk], x[kk], acc[m + 2]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (kk == 0) { issue_one(ahead, bnext, 0); issue_one(ahead, bnext, 1); }
                        else if (kk == 1) { issue_one(ahead, bnext, 2); issue_one(ahead, bnext, 3); }
                        else if (kk == 2) issue_one(ahead, bnext, 4);
                        else advance(ahead);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // (scalar row base + the lane's 32-bit offset: no vector address arithmetic next to the MFMAs, which
                        // an fp64 MFMA in flight would hold up)
                        asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(vq_off), "v"(x[r]), "s"(V + colw + (int64_t)(i0 + 16 * m + 4 * r) * ldv) : "memory");
                        if (SWEEP) {
                            qacc = fma(x[r], x[r], qacc);
                            macc = fma(x[r], zr[r], macc);
                        }
                    }
                    // (the partial sums here and now: left alone the compiler sinks both chains to the end of the phase and keeps every
                    // solved tile and z row alive -- in scratch -- until then)
                    if (SWEEP) asm volatile("" : "+v"(qacc), "+v"(macc));
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int t = m + 3; t < kT; ++t) acc[t] = MFMA_F64(uf[t][kk], x[kk], acc[t]);
                    a1 += 4;
                    if (SWEEP && m == kPD0 - 1) {                                       // block b done: hand the lane partials over
                        hand[0] = qacc;
                        hand[1] = macc;
                        qacc = 0.0;
                        macc = 0.0;
                    }
                    PSTAMP(2); PSTAMP_NEXT(); buf = (buf == 2) ? 0 : buf + 1;
                }
#pragma unroll
                for (int m = 0; m < kPD1; ++m) {
                    wait_top();
                    const int bnext = (buf >= 1) ? buf - 1 : 2;
                    issue_stage(ahead, bnext);
                    advance(ahead);
                    if (m < 2) {                    // nothing else to do in D1: its next block's right-hand sides, unhurried
#pragma unroll
                        for (int sl = (m == 0 ? 0 : kL0); sl < (m == 0 ? kL0 : kAhead); ++sl) {
                            request_one(i0, sl);
                            __builtin_amdgcn_s_sleep(2);
                        }
                        a1 += m == 0 ? kL0 : kL1;
                    }
                    PSTAMP(2); PSTAMP_NEXT(); buf = (buf == 2) ? 0 : buf + 1;
                }
            } else {
#pragma unroll
                for (int m = 0; m < kPD0; ++m) {
                    wait_top();
                    const int bnext = (buf >= 1) ? buf - 1 : 2;
                    const double *abase0 = lds + buf * kPA + kq * kPLd + lc;              // U tile of the stage, all 256 columns
                    double *xreg = ldsB + buf * kPBd + cw * 256;                          // the inverse in, the solved tile out
                    double uf[kT][4], xb[4];
                    // the deferred k-step (of the last regular stage, or of the previous tile's fold) under this tile's U reads
#pragma unroll
                    for (int t = 0; t < kT; ++t) {
                        if (m > 0 || !P0) acc[t] = MFMA_F64(af[1][t], bf[1], acc[t]);
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) uf[t][kk] = abase0[(4 * kk) * kPLd + kRB + 16 * t];
                    }
                    PSTAMP(3);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                                       // x is published
                    PSTAMP(1);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) xb[kk] = xreg[(4 * kk + kq) * 16 + lc];
                    // the stage's DMA issue and the cursor arithmetic ride behind the fold's MFMAs, one piece each (ahead of the
                    // mid-stage barrier they made this wave the last to arrive: 1365 cycles after the stage top against the
                    // solver's 530, second timeline)
#pragma unroll
                    for (int kk = 0; kk < (m + 1 < kPD0 ? 3 : 4); ++kk)
#pragma unroll
                        for (int t = 0; t < kT; ++t) {
                            acc[t] = MFMA_F64(uf[t][kk], xb[kk], acc[t]);
                            const int slot = kk * kT + t;
                            if (slot < kPDma) issue_one(ahead, bnext, slot);
                            else if (slot == kPDma) step(ahead);
                            else if (slot == kPDma + 1) locate_a(ahead);
                            else if (slot == kPDma + 2) locate_b(ahead);
                            else if (slot == kPDma + 3) locate_c(ahead);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    if (m + 1 < kPD0) {
#pragma unroll
                        for (int t = 0; t < kT; ++t) af[1][t] = uf[t][3];               // k-step 3: behind the next stage top
                        bf[1] = xb[3];
                    }
                    PSTAMP(2); PSTAMP_NEXT(); buf = (buf == 2) ? 0 : buf + 1;
                }
#pragma unroll
                for (int m = 0; m < kPD1; ++m) {
                    wait_top();
                    const int bnext = (buf >= 1) ? buf - 1 : 2;
                    constexpr int kTl = kT;
                    const int s = 2 * m;
                    const double *ab = lds + buf * kPA + kq * kPLd + lc;          // rows of tile s: columns 0..127, of s+1: 128..255
                    const double *ivb = ldsB + buf * kPBd + kq * 16 + lc;         // the stage's two inverses
                    double iv[2][4], uf[2][kTl][4], zr[2][4];
                    if (SWEEP && m == 0) {                                        // take over the running lane partials
                        qacc = hand[0];
                        macc = hand[1];
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) iv[0][kk] = ivb[(4 * kk) * 16];
                    d4 x = {0.0, 0.0, 0.0, 0.0}, x2 = {0.0, 0.0, 0.0, 0.0};
                    x = MFMA_F64(iv[0][0], -acc[s][0], x);
                    x2 = MFMA_F64(iv[0][1], -acc[s][1], x2);
                    x = MFMA_F64(iv[0][2], -acc[s][2], x);
                    x2 = MFMA_F64(iv[0][3], -acc[s][3], x2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) iv[1][kk] = ivb[256 + (4 * kk) * 16];
#pragma unroll
                    for (int t = s + 1; t < kT; ++t)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) uf[0][t][kk] = ab[(4 * kk) * kPLd + 16 * t];
                    if (SWEEP) {
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                            for (int r = 0; r < 4; ++r) zr[hh][r] = zl[zpar + kRB + 16 * (s + hh) + kq + 4 * r];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    x += x2;
                    asm volatile("" : "+v"(x));
                    PSTAMP(1);
                    auto emit = [&](int hh, const d4 &xx) __attribute__((always_inline)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(vq_off), "v"(xx[r]), "s"(V + colw + (int64_t)(i0 + kRB + 16 * (s + hh) + 4 * r) * ldv) : "memory");
                            if (SWEEP) {
                                qacc = fma(xx[r], xx[r], qacc);
                                macc = fma(xx[r], zr[hh][r], macc);
                            }
                        }
                        if (SWEEP) asm volatile("" : "+v"(qacc), "+v"(macc));
                    };
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        acc[s + 1] = MFMA_F64(uf[0][s + 1][kk], x[kk], acc[s + 1]);
                        if (s + 2 < kT) acc[s + 2] = MFMA_F64(uf[0][s + 2][kk], x[kk], acc[s + 2]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (kk == 0) { issue_one(ahead, bnext, 0); issue_one(ahead, bnext, 1); }
                        else if (kk == 1) { issue_one(ahead, bnext, 2); issue_one(ahead, bnext, 3); }
                        else if (kk == 2) issue_one(ahead, bnext, 4);
                        else advance(ahead);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    emit(0, x);
                    const d4 na = -acc[s + 1];
                    d4 y1 = {0.0, 0.0, 0.0, 0.0}, y2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        if (kk & 1) y2 = MFMA_F64(iv[1][kk], na[kk], y2);
                        else y1 = MFMA_F64(iv[1][kk], na[kk], y1);
#pragma unroll
                        for (int t = s + 3; t < kT; ++t) {
                            acc[t] = MFMA_F64(uf[0][t][kk], x[kk], acc[t]);
                            if (SWEEP && m == 1 && kk == 0 && t == s + 3) request_one(i0, kT * 4);      // (the z rows)
                        }
                    }
                    // (the second tile's U fragments only now: the wave is alone on its SIMD's registers' worth of operands, and
                    // one LDS round trip per stage is what that costs)
#pragma unroll
                    for (int t = s + 2; t < kT; ++t)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) uf[1][t][kk] = ab[(4 * kk) * kPLd + kRB + 16 * t];
                    const d4 y = y1 + y2;
                    emit(1, y);
                    // the next pair's right-hand sides, one load behind each of these MFMAs in the first two stages (24 and 16 of
                    // them): sixteen loads each
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int t = s + 2; t < kT; ++t) {
                            acc[t] = MFMA_F64(uf[1][t][kk], y[kk], acc[t]);
                            const int idx = kk * (kT - s - 2) + (t - s - 2);
                            if (m < 2 && idx < 16) request_one(i0, 16 * m + idx);
                        }
                    a1 += 8 + (m == 0 ? kL0 : m == 1 ? kL1 : 0);
                    if (SWEEP && m == kPD1 - 1) {                                 // end of the pair: reduce over the lane groups, add
                        qacc += __shfl_xor(qacc, 16);
                        qacc += __shfl_xor(qacc, 32);
                        macc += __shfl_xor(macc, 16);
                        macc += __shfl_xor(macc, 32);
                        qtot += qacc;
                        mtot += macc;
                        qacc = 0.0;
                        macc = 0.0;
                    }
                    PSTAMP(2); PSTAMP_NEXT(); buf = (buf == 2) ? 0 : buf + 1;
                }
            }
        };
        // The first pair is peeled: every later pair begins with a regular stage, whose top is where the hand-issued loads of
        // the pair before are read back -- unconditionally, on every path the loop takes (scripts/check_hand_issued_loads.py).
        diag_phase(0, std::true_type{});
        for (int i0 = kPB; i0 < n; i0 += kPB) {
            const int nst = i0 / kPKB;
            regular_stage(i0, std::true_type{});
            for (int j = 1; j < nst; ++j) regular_stage(i0, std::false_type{});
            diag_phase(i0, std::false_type{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the clamped tail DMA before the LDS goes away
        __builtin_amdgcn_s_barrier();

        if (SWEEP && H == 1 && kq == 0) {
            q_out[colw + lc] = qtot;
            mu_out[colw + lc] = mtot;
        }
    };
    if (h == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
#undef PSTAMP
#undef PSTAMP_NEXT
}

// ------------------------------------------------------------------------------------------------
// Right-looking companion of the strip kernel, used when the sweep is pipelined with the factorisation
// (launch_cholesky with a SweepPipe): once rows [k0, k0 + klen) of U and of V are final, every row block below
// them receives its share of the substitution,
//     C[i0 : i0+128, strip] -= U[k0 : k0+klen, i0 : i0+128]^T  V[k0 : k0+klen, strip],
// so that by the time the factorisation reaches a block its right-hand sides only lack the in-block solve.
// Same decomposition, LDS stages, DMA pipeline and MFMA order as the regular stages above (the k-loop of a block
// is simply cut into the pieces [k0, k0+klen) and the partial sums rest in C between launches: fp64 either way,
// the result is bit-identical to the left-looking kernel).  One workgroup = one strip x a chunk of row blocks;
// chunks are short (tens of microseconds) so that the factorisation's own kernels, queued on a higher-priority
// stream, find a free CU quickly.
constexpr int kAccMoves = kT * 4;         // global loads (next block's C) or stores (this block's C) per lane and block

// KB = rows of U / V per pipeline stage.  With KB = 16 a workgroup needs 79,872 B of LDS and 192 registers per
// lane, so two workgroups share a CU (two waves per SIMD): while one waits at its stage barrier or for an LDS
// read the other keeps the matrix pipe busy, and a workgroup's prologue/epilogue hides under its neighbour.
// C may be the V workspace itself (the pipeline: C rows lie below the panel) or a separate array (W = V^T V for
// the likelihood gradients, where U = V = L^-1 and upper_only skips the workgroups entirely below the diagonal).
template <int KB>
__global__ __launch_bounds__(256) void trsm_update_kernel(const double *__restrict__ U, int64_t ldu,
                                                          const double *V, int64_t ldv, double *C, int64_t ldc, int k0,
                                                          int klen, int i0_begin, int i0_end, int chunk_rows,
                                                          int upper_only, const int *__restrict__ skip_if)
{
    // a factorisation that has met a non-positive pivot is abandoned: its remaining launches return at once
    if (skip_if && __builtin_nontemporal_load(skip_if) != 0) return;
    using G = StageGeom<KB>;
    constexpr int kA = G::kA, kB = G::kB, kRA = G::kRA, kParts = G::kParts, kDma = G::kDma, kKS = G::kKS;
    __shared__ __align__(16) double lds[kNBuf * (kA + kB)];

    const int tid = threadIdx.x;
#ifdef CBO_DIAG_KNOBS
    const unsigned long long upd_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long upd_r0 = __builtin_amdgcn_s_memrealtime();
    const int upd_id = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const bool upd_probe = tid == 0 && blockIdx.x == 7 && blockIdx.y == gridDim.y / 2;
    int upd_k = 0;
#define UPD_STAMP() do { if (upd_probe && upd_k < 128) g_upd_stage[upd_k++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define UPD_STAMP() do { } while (0)
#endif
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, kq = lane >> 4;
    const int64_t colw = (int64_t)blockIdx.x * kStrip + wave * 16;
    const int ib = i0_begin + (int)blockIdx.y * chunk_rows;
    const int ie = (ib + chunk_rows < i0_end) ? ib + chunk_rows : i0_end;
    if (ib >= ie) return;                                              // uniform for the workgroup
    if (upper_only && (int)blockIdx.x * kStrip + kStrip <= ib) return;  // every column left of every row: lower part
    const int nst = klen / KB;
    double *Cc = C + colw + lc;
    double *ldsB = lds + kNBuf * kA;
    const unsigned lds_byte0 = lds_byte_address(lds);
    const double *ug = U + (int64_t)(k0 + wave * kRA) * ldu + lane * 2;
    const double *vg = V + (int64_t)(k0 + (lane >> 3)) * ldv + colw + 2 * (lane & 7);
    const int64_t b_stride = 8 * ldv;

    // stage cursor: (row block, KB-row slice of the panel); past the end it stays on the last stage
    int ci0 = ib, cj = 0;
    const double *a_src, *b_src;
    auto locate = [&]() __attribute__((always_inline)) {
        const bool past = ci0 >= ie;
        const int ai0 = past ? ie - kRB : ci0;
        const int aj = past ? nst - 1 : cj;
        a_src = ug + (int64_t)(KB * aj) * ldu + ai0;
        b_src = vg + (int64_t)(KB * aj) * ldv;
    };
    auto advance = [&]() __attribute__((always_inline)) {
        const int wrap = (cj + 1 == nst) ? 1 : 0;
        ci0 += kRB * wrap;
        cj = (cj + 1) * (1 - wrap);
        locate();
    };
    auto issue_one = [&](int buf, int part, int q) __attribute__((always_inline)) {
        if (q < 2) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lds_byte0 + 8u * (unsigned)(buf * kA + (wave * kRA) * kLdsLd));
            const int p = 2 * part + q;
            glds16(a_src + (int64_t)p * ldu, la + 8u * (unsigned)(p * kLdsLd));
        } else {
            const unsigned lb = __builtin_amdgcn_readfirstlane(
                lds_byte0 + 8u * (unsigned)(kNBuf * kA + buf * kB + wave * (KB * 16)));
            glds16(b_src + part * b_stride, lb + 8u * (unsigned)(part * 128));
        }
    };
    auto issue_part = [&](int buf, int part) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 3; ++q) issue_one(buf, part, q);
    };

    // acc = -C (as in the strip kernel: the k-loop then needs no operand negation)
    d4 acc[kT], accn[kT];
#pragma unroll
    for (int t = 0; t < kT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = -Cc[(int64_t)(ib + 16 * t + kq + 4 * r) * ldc];

    UPD_STAMP();                                    // [0] C tile in registers
    locate();
#pragma unroll
    for (int part = 0; part < kParts; ++part) issue_part(0, part);
    advance();
#pragma unroll
    for (int part = 0; part < kParts; ++part) issue_part(1, part);
    advance();
    UPD_STAMP();                                    // [1] two stages of DMA issued

    int buf = 0;
    // vmcnt bookkeeping (in-order retirement): at the top of a stage this wave's DMA of the stage must have
    // landed; younger than it are the next stage's DMA instructions and, around a block boundary, the 32
    // stores of the finished block and the 32 loads of the block after next -- never more than kDma + 32 that
    // may still be in flight (see the order of issue below)
    int boundary = 0;
    for (int i0 = ib; i0 < ie; i0 += kRB) {
        double af[2][kT], bf[2];
        bool deferred = false;
        for (int j = 0; j < nst; ++j) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (boundary) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma + kAccMoves) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma) : "memory");
            __builtin_amdgcn_s_barrier();
            UPD_STAMP();                            // [2 + stage] stage top passed
            boundary = (boundary > 0) ? boundary - 1 : 0;
            if (j == 0 && i0 + kRB < ie) {
                // next block's C, ahead of this stage's DMA in issue order
#pragma unroll
                for (int t = 0; t < kT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) accn[t][r] = -Cc[(int64_t)(i0 + kRB + 16 * t + kq + 4 * r) * ldc];
                asm volatile("" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            const int bnext = (buf >= 1) ? buf - 1 : 2;
            const double *abase = lds + buf * kA + kq * kLdsLd + lc;
            const double *bbase = ldsB + buf * kB + wave * (KB * 16) + kq * 16 + lc;
            // as in the strip kernel: one small piece of the other work after every MFMA, order pinned
#pragma unroll
            for (int t = 0; t < kT; ++t) {
                if (deferred) acc[t] = MFMA_F64(af[1][t], bf[1], acc[t]);
                if (t < kT / 2) {
                    af[0][2 * t] = abase[32 * t];
                    af[0][2 * t + 1] = abase[32 * t + 16];
                } else if (t == kT / 2) {
                    bf[0] = bbase[0];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int jj = 0; jj < kKS - 1; ++jj) {
                const double *an = abase + 4 * (jj + 1) * kLdsLd;
#pragma unroll
                for (int t = 0; t < kT; ++t) {
                    acc[t] = MFMA_F64(af[jj & 1][t], bf[jj & 1], acc[t]);
                    if (t < kT / 2) {
                        af[(jj + 1) & 1][2 * t] = an[32 * t];
                        af[(jj + 1) & 1][2 * t + 1] = an[32 * t + 16];
                    } else if (t == kT / 2) {
                        bf[(jj + 1) & 1] = bbase[4 * (jj + 1) * 16];
                    } else if (jj < kParts) {
                        issue_one(bnext, jj, t - (kT / 2 + 1));
                    } else if (jj == kParts && t == kT / 2 + 1) {
                        advance();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            deferred = true;                       // the last k-step sits in af[1], bf[1] (kKS is even)
            buf = (buf == 2) ? 0 : buf + 1;
        }
#pragma unroll
        for (int t = 0; t < kT; ++t) acc[t] = MFMA_F64(af[1][t], bf[1], acc[t]);
        UPD_STAMP();                                // block's last MFMAs issued
#pragma unroll
        for (int t = 0; t < kT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cc[(int64_t)(i0 + 16 * t + kq + 4 * r) * ldc] = -acc[t][r];
        asm volatile("" ::: "memory");
        UPD_STAMP();                                // stores issued
#pragma unroll
        for (int t = 0; t < kT; ++t) acc[t] = accn[t];
        boundary = 2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef CBO_DIAG_KNOBS
    UPD_STAMP();                                    // drained
    if (tid == 0 && upd_id < 65536) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_upd_wg[5 * upd_id] = upd_t0;
        g_upd_wg[5 * upd_id + 1] = __builtin_amdgcn_s_memtime();
        g_upd_wg[5 * upd_id + 2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        g_upd_wg[5 * upd_id + 3] = upd_r0;
        g_upd_wg[5 * upd_id + 4] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#undef UPD_STAMP
}

void launch_trsm_strips(hipStream_t s, const double *U, int64_t ldu, const double *invDt, double *V, int64_t ldv,
                        int64_t n, int64_t m_pad, const double *z, double *q, double *mu, bool accumulate,
                        bool half_lds)
{
    if (n <= 0 || m_pad <= 0) return;
    // n is a multiple of 128 at every call site (n_pad of the sweep, the 128-row Cholesky panel)
    const dim3 grid((unsigned)(m_pad / kStrip));
    const int acc = accumulate ? 1 : 0;
    if (half_lds) {
        // beside a pipelined sweep: 16-row stages, two half-LDS workgroups per CU, one wave per SIMD each
        if (q != nullptr)
            hipLaunchKernelGGL((trsm_strip_kernel<true, 16>), grid, dim3(256), 0, s, U, ldu, invDt, V, ldv, (int)n, z, q, mu, acc);
        else
            hipLaunchKernelGGL((trsm_strip_kernel<false, 16>), grid, dim3(256), 0, s, U, ldu, invDt, V, ldv, (int)n, z, q, mu, 0);
        return;
    }
    // A workgroup per CU: 256-row pair blocks (trsm_pair_kernel) wherever the row count is a multiple of 256, 128-row blocks
    // (trsm_strip8_kernel) otherwise; CBO_HIP_STRIP_FORM=8 keeps the latter everywhere (A/B: scripts/strip_form_bits.py,
    // scripts/lib_ab.py -- same bits)
    static const bool pairs = [] {
        const char *e = getenv("CBO_HIP_STRIP_FORM");
        return !(e && atoi(e) == 8);
    }();
    if (pairs && n >= kPB && n % kPB == 0) {
#ifdef CBO_DIAG_KNOBS
        static const int pair_mask = [] {
            const char *e = getenv("CBO_HIP_STRIP_MASK");          // 256: stage stamps on (scripts/pair_timeline.py)
            return e ? atoi(e) : 0;
        }();
        const int acc = (accumulate ? 1 : 0) | (pair_mask << 8);
#endif
        if (q != nullptr)
            hipLaunchKernelGGL((trsm_pair_kernel<true>), grid, dim3(512), 0, s, U, ldu, invDt, V, ldv, (int)n, z, q, mu, acc);
        else
            hipLaunchKernelGGL((trsm_pair_kernel<false>), grid, dim3(512), 0, s, U, ldu, invDt, V, ldv, (int)n, z, q, mu, 0);
        return;
    }
    if (q != nullptr)
        hipLaunchKernelGGL((trsm_strip8_kernel<true>), grid, dim3(512), 0, s, U, ldu, invDt, V, ldv, (int)n, z, q, mu, acc);
    else
        hipLaunchKernelGGL((trsm_strip8_kernel<false>), grid, dim3(512), 0, s, U, ldu, invDt, V, ldv, (int)n, z, q, mu, 0);
}

void launch_trsm_update(hipStream_t s, const double *U, int64_t ldu, double *V, int64_t ldv, int k0, int klen,
                        int i0_begin, int i0_end, int64_t m_pad, int chunk_blocks, bool half_lds)
{
    launch_gemm_update(s, U, ldu, V, ldv, V, ldv, k0, klen, i0_begin, i0_end, m_pad, chunk_blocks, half_lds, false, nullptr);
}

void launch_gemm_update(hipStream_t s, const double *U, int64_t ldu, const double *V, int64_t ldv, double *C,
                        int64_t ldc, int k0, int klen, int i0_begin, int i0_end, int64_t m_pad, int chunk_blocks,
                        bool half_lds, bool upper_only, const int *skip_if)
{
    if (i0_begin >= i0_end || m_pad <= 0 || klen <= 0) return;
    // klen: a multiple of the stage height KB (16 or 32) and at least two stages, which the vmcnt bookkeeping assumes --
    // 128 or 256 for a panel (pair), 512 for the bulk groups of launch_cholesky, 256 G for the groups of a pipelined sweep
    // (G = 2..4, cbo_gp_fit_sweep clamps CBO_HIP_PIPE_GROUP to that); the row range a multiple of 128
    const int chunk_rows = chunk_blocks * kRB;
    const unsigned chunks = (unsigned)((i0_end - i0_begin + chunk_rows - 1) / chunk_rows);
    const dim3 grid((unsigned)(m_pad / kStrip), chunks);
    const int up = upper_only ? 1 : 0;
    // (Round 5: the same update on trsm_pair_kernel's regular stages -- one 512-thread workgroup per CU on 256-row pairs,
    // hand-issued loads of the next pair's C tile -- gave the same bits and lost in place: C2 step 5.40 -> 5.48 ms, the
    // 16384-point factorisation 29.5 -> 32.6 ms, worse with more pairs per workgroup (profiles/r05_update_pair_ab.txt):
    // a workgroup that fills a CU leaves the factorisation's chain no slot beside it.  Removed.)
    // (A two-waves-per-SIMD form of this kernel, rows split over a wave pair as in trsm_strip8_kernel, was measured in
    // round 3 and dropped: its K-loops are only 4-8 stages long between C moves, and two independent half-LDS workgroups
    // per CU hide those block boundaries better -- 16384-point factorisation 32.0 ms with KB = 16 x 2 workgroups,
    // 37.6-37.9 ms with either full-LDS form; C2 overlapped step 5.73 vs 6.17-6.27 ms.)
    if (!half_lds)
        hipLaunchKernelGGL(trsm_update_kernel<32>, grid, dim3(256), 0, s, U, ldu, V, ldv, C, ldc, k0, klen, i0_begin,
                           i0_end, chunk_rows, up, skip_if);
    else
        hipLaunchKernelGGL(trsm_update_kernel<16>, grid, dim3(256), 0, s, U, ldu, V, ldv, C, ldc, k0, klen, i0_begin,
                           i0_end, chunk_rows, up, skip_if);
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
