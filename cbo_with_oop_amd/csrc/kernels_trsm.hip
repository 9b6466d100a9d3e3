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
// Timing-only build: workgroup 0 / wave 0 stamps s_memtime around the barrier and at the end of every stage
// (3 stamps per stage) into a debug buffer read back by cbo_diag_trsm_stamps (scripts/trsm_timeline.py).
__device__ unsigned long long g_trsm_stamps[8 * 4096];
#define STAMP(slot)                                                                          \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        if (stamp_on && stamp_i < 4096) g_trsm_stamps[8 * stamp_i + (slot)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#define STAMP_NEXT() do { if (stamp_on) ++stamp_i; } while (0)
extern "C" int cbo_diag_trsm_stamps(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trsm_stamps), sizeof(unsigned long long) * (size_t)n);   // 8 per stage
}
#else
#define STAMP(slot)
#define STAMP_NEXT()
#endif

// One continuous software pipeline over "stages" of KB U-rows (described for KB = 32).  Row block b (rows
// i0 = 128 b) consists of nst = i0/32 regular stages (k rows [32 j, 32 j + 32) against the block's 128 columns)
// followed by four diagonal stages (k rows i0 + 32 m: the block's own upper-triangular part).  Every stage's U
// tile [32][128] is fetched by the same LDS-DMA pattern; the per-wave B region receives V rows (regular
// stages) or the two 16x16 diagonal inverses (diagonal stages).  The DMA of stage g+2 is issued while
// stage g computes, across block boundaries, so the pipeline never drains.
struct StageCursor {
    int i0, j, lim;                       // block origin, stage index within the block, stages in the block
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
    auto locate = [&](StageCursor &c) __attribute__((always_inline)) {
        const bool past = c.i0 >= n;
        const int ai0 = past ? n - kRB : c.i0;
        const int aj = past ? (n - kRB) / kKB + kDS - 1 : c.j;
        const int nreg = ai0 / kKB;
        c.a_src = ug + (int64_t)(kKB * aj) * ldu + ai0;
        // diagonal stage: the two 16x16 diagonal inverses go to the B region; regular stage: V rows
        // [32 aj, 32 aj + 32) of this wave's 16 columns
        const int64_t diag = (aj >= nreg) ? 1 : 0;
        const int64_t off_diag = ((int64_t)(ai0 / 16) + kDT * (aj - nreg)) * 256;
        const int64_t off_reg = (int64_t)(kKB * aj) * ldv;
        const uintptr_t base = (uintptr_t)vg + ((uintptr_t)inv_lane - (uintptr_t)vg) * (uintptr_t)diag;
        c.b_src = reinterpret_cast<const double *>(base) + (off_reg + (off_diag - off_reg) * diag);
        c.b_stride = 8 * ldv + (128 - 8 * ldv) * diag;
    };
    auto advance = [&](StageCursor &c) __attribute__((always_inline)) {
        const int wrap = (c.j + 1 == c.lim) ? 1 : 0;
        c.i0 += kRB * wrap;
        c.j = (c.j + 1) * (1 - wrap);
        c.lim = c.lim + (c.i0 / kKB + kDS - c.lim) * wrap;
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

    StageCursor ahead{0, 0, kDS, nullptr, nullptr, 0};
    locate(ahead);
    issue_stage(ahead, 0);
    advance(ahead);
    issue_stage(ahead, 1);
    advance(ahead);
#ifdef CBO_DIAG_KNOBS
    const bool stamp_on = SWEEP && blockIdx.x == 0 && tid == 0;
    int stamp_i = 0;
#endif
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
            STAMP(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            STAMP(3);
            STAGE_TOP();
            STAMP(1);
            __builtin_amdgcn_sched_barrier(0);
            const int bnext = (buf >= 1) ? buf - 1 : 2;       // (buf + 2) % 3
            extra_prev = 0;
            const double *abase = lds + buf * kABuf + kq * kLdsLd + lc;
            const double *bbase = ldsB + buf * kBBuf + wave * (kKB * 16) + kq * 16 + lc;
#pragma unroll
            for (int t = 0; t < kT; ++t) af[0][t] = abase[16 * t];
            bf[0] = bbase[0];
            if (deferred) {
#pragma unroll
                for (int t = 0; t < kT; ++t) acc[t] = MFMA_F64(af[1][t], bf[1], acc[t]);
            }
            __builtin_amdgcn_sched_barrier(0);
            STAMP(4);
#pragma unroll
            for (int jj = 0; jj < kKS - 1; ++jj) {
#pragma unroll
                for (int t = 0; t < kT; ++t) af[(jj + 1) & 1][t] = abase[4 * (jj + 1) * kLdsLd + 16 * t];
                bf[(jj + 1) & 1] = bbase[4 * (jj + 1) * 16];
                if (jj < kParts) issue_part(ahead, bnext, jj);            // stage g+2's DMA rides under the MFMAs
                if (jj == kParts) advance(ahead);                         // cursor bookkeeping under the MFMAs too
#pragma unroll
                for (int t = 0; t < kT; ++t) acc[t] = MFMA_F64(af[jj & 1][t], bf[jj & 1], acc[t]);
                // pin "LDS reads of step jj+1, DMA, then the MFMAs of step jj": reads and DMA issue complete
                // under the MFMAs
                SCHED_DS(kT + 1);
                if (jj < kParts) { SCHED_VMEM(3); }
                SCHED_MFMA(kT);
            }
            deferred = true;                                              // the last k-step sits in af[1], bf[1]
            STAMP(2);
            STAMP_NEXT();
            buf = (buf == 2) ? 0 : buf + 1;
        }
        if (deferred) {
#pragma unroll
            for (int t = 0; t < kT; ++t) acc[t] = MFMA_F64(af[1][t], bf[1], acc[t]);
        }

        // ---- diagonal stages (KB rows each): X_s = inv(L_ss) R_s, then R_t -= L_ts X_s for the tiles below
#pragma unroll
        for (int m = 0; m < kDS; ++m) {
            STAMP(0);
            STAGE_TOP();
            STAMP(1);
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
            STAMP(2);
            STAMP_NEXT();
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
    auto issue_part = [&](int buf, int part) __attribute__((always_inline)) {
        const unsigned la = __builtin_amdgcn_readfirstlane(lds_byte0 + 8u * (unsigned)(buf * kA + (wave * kRA) * kLdsLd));
        const unsigned lb = __builtin_amdgcn_readfirstlane(
            lds_byte0 + 8u * (unsigned)(kNBuf * kA + buf * kB + wave * (KB * 16)));
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = 2 * part + q;
            glds16(a_src + (int64_t)p * ldu, la + 8u * (unsigned)(p * kLdsLd));
        }
        glds16(b_src + part * b_stride, lb + 8u * (unsigned)(part * 128));
    };

    // acc = -C (as in the strip kernel: the k-loop then needs no operand negation)
    d4 acc[kT], accn[kT];
#pragma unroll
    for (int t = 0; t < kT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = -Cc[(int64_t)(ib + 16 * t + kq + 4 * r) * ldc];

    locate();
#pragma unroll
    for (int part = 0; part < kParts; ++part) issue_part(0, part);
    advance();
#pragma unroll
    for (int part = 0; part < kParts; ++part) issue_part(1, part);
    advance();

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
#pragma unroll
            for (int t = 0; t < kT; ++t) af[0][t] = abase[16 * t];
            bf[0] = bbase[0];
            if (deferred) {
#pragma unroll
                for (int t = 0; t < kT; ++t) acc[t] = MFMA_F64(af[1][t], bf[1], acc[t]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < kKS - 1; ++jj) {
#pragma unroll
                for (int t = 0; t < kT; ++t) af[(jj + 1) & 1][t] = abase[4 * (jj + 1) * kLdsLd + 16 * t];
                bf[(jj + 1) & 1] = bbase[4 * (jj + 1) * 16];
                if (jj < kParts) issue_part(bnext, jj);
                if (jj == kParts) advance();
#pragma unroll
                for (int t = 0; t < kT; ++t) acc[t] = MFMA_F64(af[jj & 1][t], bf[jj & 1], acc[t]);
                SCHED_DS(kT + 1);
                if (jj < kParts) { SCHED_VMEM(3); }
                SCHED_MFMA(kT);
            }
            deferred = true;                       // the last k-step sits in af[1], bf[1] (kKS is even)
            buf = (buf == 2) ? 0 : buf + 1;
        }
#pragma unroll
        for (int t = 0; t < kT; ++t) acc[t] = MFMA_F64(af[1][t], bf[1], acc[t]);
#pragma unroll
        for (int t = 0; t < kT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cc[(int64_t)(i0 + 16 * t + kq + 4 * r) * ldc] = -acc[t][r];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < kT; ++t) acc[t] = accn[t];
        boundary = 2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

void launch_trsm_strips(hipStream_t s, const double *U, int64_t ldu, const double *invDt, double *V, int64_t ldv,
                        int64_t n, int64_t m_pad, const double *z, double *q, double *mu, bool accumulate,
                        bool half_lds)
{
    if (n <= 0 || m_pad <= 0) return;
    // n is a multiple of 128 at every call site (n_pad of the sweep, the 128-row Cholesky panel)
    const dim3 grid((unsigned)(m_pad / kStrip));
    const int acc = accumulate ? 1 : 0;
    if (q != nullptr) {
        if (half_lds)
            hipLaunchKernelGGL((trsm_strip_kernel<true, 16>), grid, dim3(256), 0, s, U, ldu, invDt, V, ldv, (int)n, z, q, mu, acc);
        else
            hipLaunchKernelGGL((trsm_strip_kernel<true, 32>), grid, dim3(256), 0, s, U, ldu, invDt, V, ldv, (int)n, z, q, mu, acc);
    } else {
        if (half_lds)
            hipLaunchKernelGGL((trsm_strip_kernel<false, 16>), grid, dim3(256), 0, s, U, ldu, invDt, V, ldv, (int)n, z, q, mu, 0);
        else
            hipLaunchKernelGGL((trsm_strip_kernel<false, 32>), grid, dim3(256), 0, s, U, ldu, invDt, V, ldv, (int)n, z, q, mu, 0);
    }
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
    // klen is 128 or 256 (a multiple of the 32-row stage and at least two stages, which the vmcnt bookkeeping
    // assumes), the row range a multiple of 128
    const int chunk_rows = chunk_blocks * kRB;
    const unsigned chunks = (unsigned)((i0_end - i0_begin + chunk_rows - 1) / chunk_rows);
    const dim3 grid((unsigned)(m_pad / kStrip), chunks);
    const int up = upper_only ? 1 : 0;
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
