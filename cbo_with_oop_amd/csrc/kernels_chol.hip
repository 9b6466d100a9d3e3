// Blocked right-looking Cholesky  Ky = U^T U  (upper factor, row-major) for gfx950, with the forward
// solve z = L^-1 (y - m) carried along as an extra right-hand-side column, and the backward solve for
// alpha.
//
// Restates LAPACK dpotrf + dpotrs as reached by GPy's jitchol / dpotrs (GPy ExactGaussianInference;
// the reference builds that model at /root/reference/src/GaussianProcessFactory.py:57-73).  A
// non-positive (or NaN) pivot sets *info (first failing 1-based pivot index), which the host-side
// jitter ladder reads after the factorisation.
//
// Per 128-row panel (rows r0 = 128 k; panels go in pairs, launch_cholesky has the schedule):
//   1. potrf_panel_fused_kernel   diagonal block AND row panel in one launch: workgroup 0 factors the 128x128 block in
//                                 LDS (potrf_diag128_v2's device function: one wave runs the chain of 16x16 register
//                                 Choleskys, three waves solve the block's row panel and update its trailing tiles) and
//                                 publishes finished row tiles; the other workgroups are the panel's 64-column strips
//                                 and follow tile by tile.  Separate launches (CBO_HIP_PANEL_FORM=2):
//                                 potrf_diag128_v2_kernel + panel_trsm_kernel
//   2. syrk_rows_kernel           the next panel's (pair's) own rows  -= P^T P   (the piece of the update on the chain)
//   3. syrk_kernel<64> / trsm_update_kernel   the bulk of  A[below, below] -= P^T P  on a second stream (fp64 MFMA),
//                                 rhs -= P^T z  along with it
// The bulk update carries the n^3/3 flops; steps 1-2 are the serial chain.
// Also here: backsolve / forward-solve steps for single vectors, and the one-workgroup kernels for models of at most 128
// observations (small_sets_kernel: factor + sweep of every exploration set of a trial; small_lml_kernel: likelihood and
// gradients of one MLE iterate).
#include <hip/hip_ext.h>
#include <climits>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cbo_device.h"

namespace cbo {

#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------------
constexpr int kDiagLd = 144;   // LDS row stride (doubles): rows kq and kq+1 of a fragment are 32 banks apart

// ------------------------------------------------------------------------------------------------
// Second form of the diagonal-block kernel: the same arithmetic per tile, decoupled waves.
//
// The block's chain is 8 tile factorisations of 16 dependent pivots each (wave 0); everything else -- the row panel
// X = inv(L_d) S[o:o+16, o+16:] and the rank-16 update of the trailing tiles -- is throughput work.  In the first
// form the four waves alternate between the two kinds of work with two barriers per tile, so the chain waits for
// the row panel and the other waves wait for the chain.  Here wave 0 runs one step ahead and touches nothing the
// other waves produce inside an interval:
//   interval jb:  wave 0     X01 = inv(L_jb) S(jb, jb+1);  S(jb+1, jb+1) -= X01^T X01 (registers);  factor tile jb+1
//                            (factor and inverse stay in LDS: the wave never waits on a global store)
//                 waves 1-3  each solves a third of the row panel X(jb, jb+1..8) (tiles ct with ct % 3 == w), leaves
//                            it in LDS and in global memory (finished factor rows), meets the other two at an LDS
//                            counter, reads the whole panel back into registers and updates its share of the trailing
//                            tiles T(ti, tj) -= X_ti^T X_tj from them (the f64 MFMA result map is both operand maps)
//   ONE workgroup barrier per interval; the rendezvous of waves 1-3 in the middle is theirs alone (an LDS counter), so
//   the chain never waits for it.  A solved tile overwrites its own unsolved image in S (only its owner read that),
//   except tile jb+1, which wave 0 reads in the same interval: that one goes to a spare tile below the diagonal
//   (rows 16..31, columns 0..15 -- the lower triangle of S is never loaded or read).  The diagonal-tile inverse is
//   double-buffered (wave 0 writes tile jb+1's while the others read tile jb's).  The right-hand side rides along as column tile 8 (the 16 spare columns of the LDS row
//   stride: column 128 = r, the rest zero), so z = L^-1 r needs no code of its own.
struct Diag2Shared {
    double S[128][kDiagLd];    // the block, upper triangle; columns 128..143: rhs tile (column 128) 
    double Yt[2][16][16];      // inverse of the diagonal factor of tile jb in Yt[jb & 1]: Yt[k][i] = inv(L_d)[i][k]
    int xcount;                // row-panel tiles published by waves 1-3 (their own rendezvous; wave 0 never waits on it)
    int pad_[3];
};

// Register Cholesky of one 16x16 tile given in the MFMA accumulator layout (d[r] = D[kq + 4r][lc], anything below
// the diagonal ignored); returns the factor in the same layout (zeros below the diagonal), writes the inverse to
// LDS (transposed: the A-operand image of the row-panel product) and to global memory (what the strip TRSM reads).
//
// The tile's 16 pivots are one dependent chain, so what counts is the number of dependent instructions per pivot
// (about 13 cycles each for fp64 VALU work in a lone wave; scripts/probes/tile_factor_probe.hip has the forms tried):
//   * pivots go in blocks of four.  One MFMA with a 0/1 selection operand replicates the block's four rows to every
//     16-lane row (t[r] = D[4b + r][lc] on all lane rows), so no cross-lane shuffle sits between two pivots;
//   * the block's 4x4 diagonal sub-block is factored FIRST, on uniform scalars (its ten entries read once with
//     v_readlane, every lane repeating the same arithmetic): per pivot the chain is rsqrt -> multiply -> fma.  The
//     16-wide row scalings and updates follow from those scalars, off the chain;
//   * rsqrt is ocml's instruction sequence (v_rsq_f64, one third-order correction) without its class-check selects,
//     and the positivity test only records the first bad pivot (reported once, after the tile) -- a non-positive pivot
//     lets NaN / Inf through the rest of the tile, which jitchol's retry discards anyway;
//   * a rank-4 MFMA carries the block into the rows below it.
// Every value goes through the same operations in the same order as the plain right-looking form.
// `blocks` (uniform) < 4: only the tile's first `blocks` blocks of four pivots are factored -- the rows of the others are
// identity padding (a model of fewer rows than its tiles hold), whose factor and inverse are the identity they already
// are; nothing a posterior reads depends on them.
__device__ __forceinline__ d4 factor_tile_regs(d4 din, int lane, int pivot_row0, int *info, double (*Yt)[16],
                                               double *__restrict__ invDt_tile, int blocks = 4)
{
    const int lc = lane & 15, kq = lane >> 4;
    d4 d, e;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        d[r] = (kq + 4 * r <= lc) ? din[r] : 0.0;
        e[r] = (kq + 4 * r == lc) ? 1.0 : 0.0;
    }
    const double sel = ((lc >> 2) == kq) ? 1.0 : 0.0;       // A[i = lc][k = kq] = delta(k, i >> 2)
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    int bad = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (b >= blocks) break;
        d4 t = MFMA_F64(sel, d[b], zero);
        d4 s = MFMA_F64(sel, e[b], zero);
        double a[4][4], u[4][4], inv[4], dj[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i; j < 4; ++j) a[i][j] = readlane_f64(t[i], 4 * b + j);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double pj = a[j][j];
            if (!(pj > 0.0) && bad == 0) bad = 4 * b + j + 1;
            const double y0 = __builtin_amdgcn_rsq(pj);
            const double tt = y0 * -pj;
            const double ee = fma(tt, y0, 1.0);
            const double gg = y0 * ee;
            const double hh = fma(ee, 0.375, 0.5);
            inv[j] = fma(gg, hh, y0);
            dj[j] = pj * inv[j];
#pragma unroll
            for (int k = j + 1; k < 4; ++k) u[j][k] = a[j][k] * inv[j];
#pragma unroll
            for (int i = j + 1; i < 4; ++i)
#pragma unroll
                for (int k = i; k < 4; ++k) a[i][k] = fma(-u[j][i], u[j][k], a[i][k]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piv = 4 * b + j;
            t[j] = (lc > piv) ? t[j] * inv[j] : ((lc == piv) ? dj[j] : 0.0);
            s[j] *= inv[j];
#pragma unroll
            for (int i = j + 1; i < 4; ++i) {
                t[i] = fma(-u[j][i], t[j], t[i]);
                s[i] = fma(-u[j][i], s[j], s[i]);
            }
        }
        d[b] = (kq == 0) ? t[0] : (kq == 1) ? t[1] : (kq == 2) ? t[2] : t[3];
        e[b] = (kq == 0) ? s[0] : (kq == 1) ? s[1] : (kq == 2) ? s[2] : s[3];
        if (b < 3) {
            const d4 keep = d, keep_e = e;
            const double na = -d[b];
            d = MFMA_F64(na, d[b], d);
            e = MFMA_F64(na, e[b], e);
#pragma unroll
            for (int r = 0; r <= b; ++r) { d[r] = keep[r]; e[r] = keep_e[r]; }
        }
    }
    if (bad != 0 && lane == 0) atomicCAS(info, 0, pivot_row0 + bad);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Yt[lc][kq + 4 * r] = e[r];
        if (invDt_tile) invDt_tile[lc * 16 + kq + 4 * r] = e[r];
    }
    return d;
}

// The trailing tiles one of waves 1..3 owns: columns {8, 3, 2}, {7, 4, 1}, {6, 5} (12, 12 and 11 tiles, balanced for
// every step since a column loses one tile per step), listed by row so that the tiles still due form a suffix.
template <int W>
struct DiagTiles;
template <>
struct DiagTiles<0> {
    static constexpr int n = 12;
    static constexpr int ti[12] = {1, 1, 1, 2, 2, 2, 3, 3, 4, 5, 6, 7};
    static constexpr int tj[12] = {2, 3, 8, 2, 3, 8, 3, 8, 8, 8, 8, 8};
};
template <>
struct DiagTiles<1> {
    static constexpr int n = 12;
    static constexpr int ti[12] = {1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 6, 7};
    static constexpr int tj[12] = {1, 4, 7, 4, 7, 4, 7, 4, 7, 7, 7, 7};
};
template <>
struct DiagTiles<2> {
    static constexpr int n = 12;      // the last entry repeats a tile and is never written
    static constexpr int ti[12] = {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6};
    static constexpr int tj[12] = {5, 6, 5, 6, 5, 6, 5, 6, 5, 6, 6, 6};
};

template <int W>
__device__ __forceinline__ void diag_trailing(Diag2Shared &sh, const d4 (&x)[9], const d4 (&nx)[9], int jb, int lane,
                                              int tiles)
{
    using L = DiagTiles<W>;
    const int lc = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int g = 0; g < L::n; g += 4) {
        // the group's last tile has the largest row index: nothing due in the group -> skip it (uniform); its first
        // tile the smallest: a group entirely inside the identity padding of a short block has nothing to do either
        if (L::ti[g + 3] <= jb || L::ti[g] >= tiles) continue;
        d4 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[k][r] = sh.S[16 * L::ti[g + k] + kq + 4 * r][16 * L::tj[g + k] + lc];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = MFMA_F64(x[L::ti[g + k]][r], nx[L::tj[g + k]][r], acc[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ti = L::ti[g + k], tj = L::tj[g + k];
            const bool dup = (W == 2 && g + k == 11);
            if (!dup && ti > jb && !(ti == tj && ti == jb + 1)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sh.S[16 * ti + kq + 4 * r][16 * tj + lc] = acc[k][r];
            }
        }
    }
}

#ifdef CBO_DIAG_KNOBS
// Timing-only build: bit 1 = the worker waves of the block factorisation skip their trailing tiles, bit 2 = their
// row-panel thirds as well (what the chain wave's tile factor costs with nothing beside it; results are wrong)
__device__ int g_diag_knob;
extern "C" int cbo_diag_set_knob(int v) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_diag_knob), &v, sizeof(int)); }
#define DIAG_KNOB(bit) (g_diag_knob & (bit))
// Timing-only build: s_memtime stamps of the last diagonal-block launch, [wave][interval][slot] (scripts/diag_stamps.py)
__device__ unsigned long long g_diag_stamps[4 * 9 * 4];
#define DSTAMP(wave_, jb_, slot_) do { if ((threadIdx.x & 63) == 0) g_diag_stamps[((wave_) * 9 + (jb_)) * 4 + (slot_)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int cbo_diag_chol_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_stamps), sizeof(unsigned long long) * 4 * 9 * 4);
}
__device__ unsigned long long g_small_stamps[16];
#define SSTAMP(i_) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_small_stamps[(i_)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int cbo_diag_small_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_small_stamps), sizeof(unsigned long long) * 16);
}
#else
#define DIAG_KNOB(bit) false
#define DSTAMP(wave_, jb_, slot_) do { } while (0)
#define SSTAMP(i_) do { } while (0)
#endif

// A store of the factor that a workgroup of the SAME launch may read (the strips of the fused diagonal + panel launch):
// written through to the coherence point of the device instead of resting in this XCD's L2.
// Fences of the fused diagonal + panel protocol (see potrf_panel_fused_kernel).  The consumer's ACQUIRE is always there
// (1-2 % of the chain).  The producer's RELEASE is a build option: LLVM implements an agent-scope release on gfx950 as
// buffer_wbl2 sc1 -- a write-back of the whole XCD's L2, which at that moment also holds the dirty lines of the bulk
// trailing update running beside the chain -- and it costs 17 % of the factorisation at 4096 points (1.55 -> 1.82 ms;
// 31.9 -> 33.8 ms at 16384; A/B on one box, round 3).  The default producer instead relies on what its stores are on
// this hardware: agent-scope atomic stores (global_store ... sc1, written through to the device's coherence point),
// retired by s_waitcnt vmcnt(0) before the count is incremented.  -DCBO_FORMAL_RELEASE restores the fence.
#ifdef CBO_FORMAL_RELEASE
#define AGENT_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#else
#define AGENT_RELEASE()
#endif
#define AGENT_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
// The chain's kernels share SIMDs with the bulk update's (and a pipelined sweep's) MFMA waves: their instructions go first.
#ifndef CBO_CHAIN_PRIO
#define CBO_CHAIN_PRIO 3
#endif
#define CHAIN_PRIORITY() __builtin_amdgcn_s_setprio(CBO_CHAIN_PRIO)
template <bool PUBLISH>
__device__ __forceinline__ void gstore(double *p, double v)
{
    if (PUBLISH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// One interval of one of waves 1..3 (W = wave - 1): its third of the row panel, the rendezvous, its trailing tiles.
// The wave with W == jb % 3 also carries tile jb's diagonal factor and inverse from LDS (where wave 0 left them) to
// global memory: the chain wave itself never waits on a global store.  PUBLISH: after its last store of the interval
// the wave counts itself in at `flag` (row tile jb of the factor is complete when the count reaches 3 (jb + 1)).
template <int W, bool PUBLISH>
__device__ __forceinline__ void diag_worker(Diag2Shared &sh, double *A, int64_t lda, int r0, int rcol,
                                            double *__restrict__ invDt, double *__restrict__ zvec, int jb,
                                            const double (&af)[4], int lane, int tiles, int *flag)
{
    const int lc = lane & 15, kq = lane >> 4;
    const int o = 16 * jb;
    constexpr int ct0 = (W == 0) ? 3 : W;                 // own column tiles: ct0, ct0 + 3, ct0 + 6 (<= 8; 8 = rhs)
    constexpr int nown = (ct0 + 6 <= 8) ? 3 : 2;
    // tiles that are not due (ct <= jb) are solved along on whatever S holds there (cheaper than branching around a
    // chain of four MFMAs) and stored nowhere
    DSTAMP(W + 1, jb, 0);
    if (W == jb % 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gstore<PUBLISH>(&A[(int64_t)(r0 + o + kq + 4 * r) * lda + r0 + o + lc], sh.S[o + kq + 4 * r][o + lc]);
            gstore<PUBLISH>(&invDt[(int64_t)(r0 / 16 + jb) * 256 + lc * 16 + kq + 4 * r], sh.Yt[jb & 1][lc][kq + 4 * r]);
        }
    }
    d4 xo[nown];
    {
        double bq[4][nown];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int n = 0; n < nown; ++n) bq[kk][n] = sh.S[o + 4 * kk + kq][16 * (ct0 + 3 * n) + lc];
#pragma unroll
        for (int n = 0; n < nown; ++n) xo[n] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int n = 0; n < nown; ++n) xo[n] = MFMA_F64(af[kk], bq[kk][n], xo[n]);
    }
#pragma unroll
    for (int n = 0; n < nown; ++n) {
        const int ct = ct0 + 3 * n;
        if (ct == jb + 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sh.S[16 + kq + 4 * r][lc] = xo[n][r];
        } else if (ct > jb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sh.S[o + kq + 4 * r][16 * ct + lc] = xo[n][r];
        }
    }
    // publish, then wait for the other two (LDS operations of a wave complete in order; the counter only grows)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&sh.xcount, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // rows o .. o+15 of the factor right of the diagonal tile, and z, leave for global memory meanwhile
#pragma unroll
    for (int n = 0; n < nown; ++n) {
        const int ct = ct0 + 3 * n;
        if (ct > jb) {
            if (ct < 8) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    gstore<PUBLISH>(&A[(int64_t)(r0 + o + kq + 4 * r) * lda + r0 + 16 * ct + lc], xo[n][r]);
            } else if (lc == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = r0 + o + kq + 4 * r;
                    A[(int64_t)row * lda + rcol] = xo[n][r];
                    if (zvec) zvec[row] = xo[n][r];
                }
            }
        }
    }
    DSTAMP(W + 1, jb, 1);
    if (PUBLISH && W == jb % 3) {
        // tile jb's inverse (and diagonal factor) are out: flag[1] counts them.  The wait overlaps the rendezvous.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        AGENT_RELEASE();                                       // the count is a release of this wave's stores
        if (lane == 0) __hip_atomic_fetch_add(flag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    while (__hip_atomic_load(&sh.xcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 3 * (jb + 1))
        __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    DSTAMP(W + 1, jb, 2);
    // the whole panel into registers (tiles <= jb are finished rows whose products nobody stores: zeros)
    d4 x[9], nx[9];
#pragma unroll
    for (int ct = 1; ct <= 8; ++ct) {
        if (ct == jb + 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[ct][r] = sh.S[16 + kq + 4 * r][lc];
        } else if (ct > jb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[ct][r] = sh.S[o + kq + 4 * r][16 * ct + lc];
        } else {
            x[ct] = d4{0.0, 0.0, 0.0, 0.0};
        }
        nx[ct] = -x[ct];
    }
    // trailing tiles T(ti, tj) -= X_ti^T X_tj, jb < ti <= 7, ti <= tj <= 8, except the next diagonal tile (wave 0's).
    // Ownership is by column (a compile-time list per wave), tiles go four at a time with their accumulation chains
    // interleaved; a tile that is not due (ti <= jb) is computed on stale operands and simply not written back.
    if (!DIAG_KNOB(1)) diag_trailing<W>(sh, x, nx, jb, lane, tiles);
    DSTAMP(W + 1, jb, 3);
    if (PUBLISH) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have arrived
        AGENT_RELEASE();
        if (lane == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The factorisation of a block that is already in LDS (S: upper triangle + rhs tile; the caller has synchronised).
// `tiles` = 16-row tiles to factor (8 = the whole block; fewer when the rest is identity padding, which the caller
// then writes out itself).  Factor rows, diagonal inverses and z go to global memory (A, invDt, zvec).
template <bool PUBLISH = false>
__device__ __forceinline__ void diag128_factor_in_lds(Diag2Shared &sh, double *A, int64_t lda, int r0, int rcol,
                                                      double *__restrict__ invDt, int *info,
                                                      double *__restrict__ zvec, int tiles, int *flag = nullptr,
                                                      int last_blocks = 4)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, kq = lane >> 4;
    if (tid == 64) sh.xcount = 0;
    if (wave == 0) {
        d4 t0;
#pragma unroll
        for (int r = 0; r < 4; ++r) t0[r] = sh.S[kq + 4 * r][lc];
        // the factor of the tile stays in LDS, in the tile's own place (nobody else touches it): a worker wave takes
        // it and the inverse to global memory in the tile's interval
        const d4 u = factor_tile_regs(t0, lane, r0, info, sh.Yt[0], nullptr, tiles == 1 ? last_blocks : 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) sh.S[kq + 4 * r][lc] = u[r];
    }
    __syncthreads();

    for (int jb = 0; jb < tiles; ++jb) {
        const int o = 16 * jb;
        double af[4];                                       // A operand of the row-panel product: inv(L_jb)[lc][4 kk + kq]
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) af[kk] = sh.Yt[jb & 1][4 * kk + kq][lc];
        if (wave == 0) {
            DSTAMP(0, jb, 0);
            if (jb + 1 < tiles) {
                // two half-sums each: a chain of dependent f64 MFMAs runs at about half the issue rate
                d4 x = {0.0, 0.0, 0.0, 0.0}, xb = {0.0, 0.0, 0.0, 0.0}, acc, accb = {0.0, 0.0, 0.0, 0.0};
                double bq[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) bq[kk] = sh.S[o + 4 * kk + kq][o + 16 + lc];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = sh.S[o + 16 + kq + 4 * r][o + 16 + lc];
                x = MFMA_F64(af[0], bq[0], x);
                xb = MFMA_F64(af[1], bq[1], xb);
                x = MFMA_F64(af[2], bq[2], x);
                xb = MFMA_F64(af[3], bq[3], xb);
                x += xb;
                acc = MFMA_F64(x[0], -x[0], acc);
                accb = MFMA_F64(x[1], -x[1], accb);
                acc = MFMA_F64(x[2], -x[2], acc);
                accb = MFMA_F64(x[3], -x[3], accb);
                acc += accb;
                DSTAMP(0, jb, 1);
                const d4 u = factor_tile_regs(acc, lane, r0 + o + 16, info, sh.Yt[(jb + 1) & 1], nullptr,
                                              jb + 2 == tiles ? last_blocks : 4);
                DSTAMP(0, jb, 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) sh.S[o + 16 + kq + 4 * r][o + 16 + lc] = u[r];
            }
            DSTAMP(0, jb, 3);
        } else {
            const int w = wave - 1;
            if (w == 0) diag_worker<0, PUBLISH>(sh, A, lda, r0, rcol, invDt, zvec, jb, af, lane, tiles, flag);
            else if (w == 1) diag_worker<1, PUBLISH>(sh, A, lda, r0, rcol, invDt, zvec, jb, af, lane, tiles, flag);
            else diag_worker<2, PUBLISH>(sh, A, lda, r0, rcol, invDt, zvec, jb, af, lane, tiles, flag);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void potrf_diag128_v2_kernel(double *A, int64_t lda, int r0, int rcol,
                                                               double *__restrict__ invDt, int *info,
                                                               double *__restrict__ zvec)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Diag2Shared &sh = *reinterpret_cast<Diag2Shared *>(smem_raw);
    if (__builtin_nontemporal_load(info) != 0) return;      // an earlier block met a non-positive pivot: abandoned
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    DSTAMP(wave, 8, 0);
    {
        const unsigned s0 = lds_byte_address(&sh.S[0][0]);
        const double *g = A + (int64_t)(r0 + wave * 32) * lda + r0 + lane * 2;
#pragma unroll 8
        for (int p = 0; p < 32; ++p) {
            // only the upper triangle (by 16-column tiles) is read: lanes left of the row's diagonal tile load nothing
            if (lane >= 8 * ((wave * 32 + p) >> 4))
                glds16(g + (int64_t)p * lda, __builtin_amdgcn_readfirstlane(s0 + 8u * (unsigned)((wave * 32 + p) * kDiagLd)));
        }
    }
    if (tid < 128) {
        sh.S[tid][128] = A[(int64_t)(r0 + tid) * lda + rcol];
#pragma unroll
        for (int c = 129; c < kDiagLd; ++c) sh.S[tid][c] = 0.0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    DSTAMP(wave, 8, 1);
    diag128_factor_in_lds(sh, A, lda, r0, rcol, invDt, info, zvec, 8);
    DSTAMP(wave, 8, 2);
}

// ------------------------------------------------------------------------------------------------
// Row panel of the blocked factorisation, U[r0:r0+128, cols] = U_kk^-T A[r0:r0+128, cols]: the strip kernel's
// arithmetic for ONE 128-row block (same 16x16 diagonal inverses, same tile order), without its three-deep staging
// pipeline -- for 128 rows that pipeline is all prologue.  The whole diagonal block goes to LDS in one burst of
// LDS-DMA, the strip's right-hand sides and the eight inverses go to registers, then the eight tile steps run
// back to back: x_s = inv(L_ss) r_s (two half-sums), tile s+1 brought up to date first, and its own solve chain
// interleaved with the rest of tile s's updates so that the matrix pipe never waits for a dependent result.
// 64 columns per 256-thread workgroup, wave w owns 16 of them (no exchange between waves).
struct PanelShared {
    double U[128][kDiagLd];
};

// The eight (or `tiles`) tile steps of a 128-row block solve for one wave's 16 columns: acc[t] = right-hand sides of
// tile t in the MFMA result layout, iv = the diagonal inverses as A operands, ub = &U[kq][lc] of the block in LDS;
// emit(s, x) receives tile s of the solution.  x_s = inv(L_ss) r_s (two half-sums), tile s+1 brought up to date first,
// its solve chain interleaved with the rest of tile s's updates.
template <typename Emit>
__device__ __forceinline__ void panel_solve_tiles(const double *ub, d4 (&acc)[8], const double (&iv)[8][4], int tiles,
                                                  Emit emit)
{
    d4 x = {0.0, 0.0, 0.0, 0.0}, x2 = {0.0, 0.0, 0.0, 0.0};
    x = MFMA_F64(iv[0][0], acc[0][0], x);
    x2 = MFMA_F64(iv[0][1], acc[0][1], x2);
    x = MFMA_F64(iv[0][2], acc[0][2], x);
    x2 = MFMA_F64(iv[0][3], acc[0][3], x2);
    x += x2;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        emit(s, x);
        if (s == 7 || s + 1 >= tiles) break;
        const d4 nx = -x;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc[s + 1] = MFMA_F64(ub[(16 * s + 4 * kk) * kDiagLd + 16 * (s + 1)], nx[kk], acc[s + 1]);
            if (s + 2 < 8) acc[s + 2] = MFMA_F64(ub[(16 * s + 4 * kk) * kDiagLd + 16 * (s + 2)], nx[kk], acc[s + 2]);
        }
        d4 y1 = {0.0, 0.0, 0.0, 0.0}, y2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk & 1) y2 = MFMA_F64(iv[s + 1][kk], acc[s + 1][kk], y2);
            else y1 = MFMA_F64(iv[s + 1][kk], acc[s + 1][kk], y1);
#pragma unroll
            for (int t = s + 3; t < 8; ++t) acc[t] = MFMA_F64(ub[(16 * s + 4 * kk) * kDiagLd + 16 * t], nx[kk], acc[t]);
        }
        x = y1 + y2;
    }
}


__global__ __launch_bounds__(256) void panel_trsm_kernel(double *A, int64_t lda, int r0, int col0,
                                                         const double *__restrict__ invDt,
                                                         const int *__restrict__ skip_if)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    PanelShared &sh = *reinterpret_cast<PanelShared *>(smem_raw);
    if (__builtin_nontemporal_load(skip_if) != 0) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, kq = lane >> 4;
    {
        const unsigned s0 = lds_byte_address(&sh.U[0][0]);
        const double *g = A + (int64_t)(r0 + wave * 32) * lda + r0 + lane * 2;
#pragma unroll 8
        for (int p = 0; p < 32; ++p) {
            if (lane >= 8 * ((wave * 32 + p) >> 4))          // upper triangle only (by 16-column tiles)
                glds16(g + (int64_t)p * lda, __builtin_amdgcn_readfirstlane(s0 + 8u * (unsigned)((wave * 32 + p) * kDiagLd)));
        }
    }
    double *Ac = A + (int64_t)r0 * lda + col0 + (int64_t)blockIdx.x * kStrip + wave * 16 + lc;
    d4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = Ac[(int64_t)(16 * t + kq + 4 * r) * lda];
    double iv[8][4];
    const double *inv = invDt + (int64_t)(r0 / 16) * 256 + kq * 16 + lc;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) iv[s][kk] = inv[s * 256 + 64 * kk];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    panel_solve_tiles(&sh.U[kq][lc], acc, iv, 8, [&](int s, const d4 &x) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Ac[(int64_t)(16 * s + kq + 4 * r) * lda] = x[r];
    });
}

// ------------------------------------------------------------------------------------------------
// Diagonal block AND its row panel in one launch (a small cooperative group: nothing waits on a grid-wide barrier).
// Workgroup 0 is the diagonal-block kernel above, publishing: its factor stores are written through to the device's
// coherence point and every worker wave counts itself in at `flag` after its last store of an interval, so row tile s
// of the block (the inverse of its diagonal tile and U[s][s+1..7]) is complete when the count reaches 3 (s + 1).
// Workgroups 1.. are the panel's 64-column strips; each wave owns 16 columns, keeps its 128 x 16 right-hand sides in
// registers and follows the block tile by tile: wait for row tile s, read it (coherent loads, straight into MFMA
// operand registers: no LDS, no barrier, the four waves are independent), x_s = inv(L_ss) r_s, fold it into the
// tiles below.  The panel solve thus hides behind the 128 pivots of the block; what is left after the block's last
// interval is one round trip and a sixteenth of the work.  Same chains as panel_solve_tiles: same bits.
// Ordering.  Producer, per interval: write-through stores (agent-scope atomic stores: global_store sc1, performed at the
// device's coherence point), s_waitcnt vmcnt(0) (they have been performed), [a RELEASE fence at agent scope when built
// with -DCBO_FORMAL_RELEASE], the relaxed increment of the interval's count.  Consumer: relaxed polls of the counts;
// once a count covers what it needs, an ACQUIRE fence at agent scope, then its loads of that data (agent-scope atomic
// loads: never served from a stale line of this XCD's L2).  With the release fence, count increment and poll form the
// synchronises-with edge and the two fences extend it to the data: every store a wave made before counting itself in
// happens-before every load a strip makes after seeing that count -- the language-level argument.  Without it (the
// default, see AGENT_RELEASE above for what the fence costs) the producer side is a hardware-level argument: the
// stores were complete at the coherence point before the increment was issued, and the consumer's loads go to that
// same point, after its acquire.  The count invariant: a worker wave counts itself in once per interval after ALL its
// stores of the interval, and row tile s needs the three workers' interval-s stores, so "count >= 3 (s + 1)" means
// row tile s is complete (the inverse of diagonal tile s is one wave's store: its own count, flag[1]).
// Forward progress rests on an assumption about the hardware, not on the language: a dispatch hands out workgroups in
// order of their index and workgroup 0 never waits for a strip, so workgroup 0 is resident and running before any
// strip can spin.  The spin is bounded all the same (`spin_limit` polls, then the status word reports kFusedTimeout,
// the launch and everything after it in the factorisation drains) and the host then repeats the factorisation with the
// separate-launch kernels (CBO_HIP_PANEL_FORM=2's), same bits: cbo_gp_fit / cbo_gp_fit_sweep.
constexpr int kFusedTimeout = kCholFusedTimeout;
constexpr int kFusedSpinLimit = 1 << 22;      // default of CBO_HIP_FUSED_SPIN_LIMIT (negative: give up at the first wait)

__device__ __forceinline__ double coherent_load(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void potrf_panel_fused_kernel(double *A, int64_t lda, int r0, int rcol,
                                                                double *__restrict__ invDt, int *info,
                                                                double *__restrict__ zvec, int col0, int *flag,
                                                                int spin_limit, int strip_base)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    CHAIN_PRIORITY();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (__builtin_nontemporal_load(info) != 0) return;      // an earlier block met a non-positive pivot: abandoned
    // strip_base == 1: workgroup 0 is the diagonal block, workgroups 1.. the strips (one launch); strip_base == 0: a
    // launch of strips only (no LDS at all), behind a one-workgroup launch of the block (the split form, see launch_panel_fused)
    if (strip_base == 1 && blockIdx.x == 0) {
        Diag2Shared &sh = *reinterpret_cast<Diag2Shared *>(smem_raw);
        {
            const unsigned s0 = lds_byte_address(&sh.S[0][0]);
            const double *g = A + (int64_t)(r0 + wave * 32) * lda + r0 + lane * 2;
#pragma unroll 8
            for (int p = 0; p < 32; ++p) {
                if (lane >= 8 * ((wave * 32 + p) >> 4))
                    glds16(g + (int64_t)p * lda, __builtin_amdgcn_readfirstlane(s0 + 8u * (unsigned)((wave * 32 + p) * kDiagLd)));
            }
        }
        if (tid < 128) {
            sh.S[tid][128] = A[(int64_t)(r0 + tid) * lda + rcol];
#pragma unroll
            for (int c = 129; c < kDiagLd; ++c) sh.S[tid][c] = 0.0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        diag128_factor_in_lds<true>(sh, A, lda, r0, rcol, invDt, info, zvec, 8, flag);
        return;
    }
    // ---- a strip: 16 columns per wave
    const int lc = lane & 15, kq = lane >> 4;
    double *Ac = A + (int64_t)r0 * lda + col0 + (int64_t)((int)blockIdx.x - strip_base) * kStrip + wave * 16 + lc;
    d4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = Ac[(int64_t)(16 * t + kq + 4 * r) * lda];
    const double *inv = invDt + (int64_t)(r0 / 16) * 256 + kq * 16 + lc;
    const double *Ub = A + (int64_t)(r0 + kq) * lda + r0 + lc;           // U[kq][lc] of the block
    // flag[0] / 3 = row tiles of the block that are complete, flag[1] = diagonal-tile inverses that are out (inverse s
    // precedes row tile s by most of an interval: x_s does not wait for the row, and the last step needs no row at all)
    int rows_seen = 0, invs_seen = 0;
    // one poll refreshes both counts (the two words share a cache line: one round trip)
    auto wait_for = [&](bool want_row, int s) -> bool {
        int spins = 0;
        if (spin_limit < 0) {                                            // test hook: every strip gives up at once
            if (lane == 0) atomicCAS(info, 0, kFusedTimeout);
            return false;
        }
        const bool polled = (want_row ? rows_seen : invs_seen) <= s;
        while ((want_row ? rows_seen : invs_seen) <= s) {
            const int f0 = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int f1 = __hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            rows_seen = f0 / 3;
            invs_seen = f1;
            if ((want_row ? rows_seen : invs_seen) > s) break;
            if (++spins > spin_limit || __builtin_nontemporal_load(info) != 0) {
                if (spins > spin_limit && lane == 0) atomicCAS(info, 0, kFusedTimeout);
                return false;                                            // uniform: every lane read the same words
            }
            __builtin_amdgcn_s_sleep(2);
        }
        // what the counts cover is visible to the loads that follow (pairs with the producer's release fences)
        if (polled) AGENT_ACQUIRE();
        return true;
    };
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (!wait_for(false, s)) return;
        // a strip that runs behind the block already knows the row is there: its fragments travel with the inverse
        const bool have_row = s < 7 && rows_seen > s;
        double iv[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) iv[kk] = coherent_load(inv + s * 256 + 64 * kk);
        double ub[4][8];
        if (have_row) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int t = s + 1; t < 8; ++t) ub[kk][t] = coherent_load(Ub + (int64_t)(16 * s + 4 * kk) * lda + 16 * t);
        }
        d4 x = {0.0, 0.0, 0.0, 0.0}, x2 = {0.0, 0.0, 0.0, 0.0};
        x = MFMA_F64(iv[0], acc[s][0], x);
        x2 = MFMA_F64(iv[1], acc[s][1], x2);
        x = MFMA_F64(iv[2], acc[s][2], x);
        x2 = MFMA_F64(iv[3], acc[s][3], x2);
        x += x2;
#pragma unroll
        for (int r = 0; r < 4; ++r) Ac[(int64_t)(16 * s + kq + 4 * r) * lda] = x[r];
        if (s == 7) break;
        if (!have_row) {
            if (!wait_for(true, s)) return;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int t = s + 1; t < 8; ++t) ub[kk][t] = coherent_load(Ub + (int64_t)(16 * s + 4 * kk) * lda + 16 * t);
        }
        const d4 nx = -x;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int t = s + 1; t < 8; ++t) acc[t] = MFMA_F64(ub[kk][t], nx[kk], acc[t]);
    }
}

// split: the block as a one-workgroup launch (a whole CU's LDS), then the strips as a launch of their own that needs NO
// LDS: beside a device full of half-LDS update workgroups (the bulk trailing update of a large factorisation, a
// pipelined sweep) a strip workgroup then fits the slot any finishing update workgroup leaves, instead of waiting for
// a CU to drain completely -- 255 strips at 16384 points otherwise wait for the bulk update to end (profiles/r03b_*).
void launch_panel_fused(hipStream_t s, double *A, int64_t lda, int r0, int rcol, double *invDt, int *info, double *zvec,
                        int n_cols, int *flag, int spin_limit, hipEvent_t done = nullptr, bool split = false)
{
    if (!split) {
        hipExtLaunchKernelGGL(potrf_panel_fused_kernel, dim3(1u + (unsigned)(n_cols / kStrip)), dim3(256), sizeof(Diag2Shared),
                              s, nullptr, done, 0, A, lda, r0, rcol, invDt, info, zvec, r0 + 128, flag, spin_limit, 1);
        return;
    }
    hipLaunchKernelGGL(potrf_panel_fused_kernel, dim3(1u), dim3(256), sizeof(Diag2Shared), s, A, lda, r0, rcol, invDt, info,
                       zvec, r0 + 128, flag, spin_limit, 1);
    hipExtLaunchKernelGGL(potrf_panel_fused_kernel, dim3((unsigned)(n_cols / kStrip)), dim3(256), 0, s, nullptr, done, 0, A,
                          lda, r0, rcol, invDt, info, zvec, r0 + 128, flag, spin_limit, 0);
}

// Set by the host around the repeat of a factorisation whose fused launch gave up (kCholFusedTimeout): the repeat uses
// the separate-launch kernels whatever CBO_HIP_PANEL_FORM says.
thread_local int g_panel_form_override = 0;
void set_panel_form_override(int form) { g_panel_form_override = form; }

void launch_panel_trsm(hipStream_t s, double *A, int64_t lda, int r0, int col0, int n_cols, const double *invDt,
                       const int *skip_if, hipEvent_t done = nullptr)
{
    if (n_cols <= 0) { if (done) hipEventRecord(done, s); return; }
    hipExtLaunchKernelGGL(panel_trsm_kernel, dim3((unsigned)(n_cols / kStrip)), dim3(256), sizeof(PanelShared), s, nullptr,
                          done, 0, A, lda, r0, col0, invDt, skip_if);
}

// ------------------------------------------------------------------------------------------------
// Small models, many sets, ONE launch (the reference's own operating point: N = 10..50 observations per exploration
// set, S = 2..25 sets, /root/reference/src/ArgumentParser.py:18,25, src/CBO.py:237-260).  At that size every kernel of
// the general path is launch latency: K(X,X), eight chain launches, K*, the strip solve, the epilogue, a stream
// synchronisation -- per set.  Here one workgroup does all of it for (one set, 64 candidates) inside LDS and
// registers, with the SAME device functions as the general path (kernel_value, the decoupled-wave block
// factorisation, the tile solve, the EI epilogue), so the numbers are the general path's numbers:
//   K(X,X) + diag  ->  LDS block (identity beyond n)       rhs r = y - m(X)  ->  column tile 8
//   factorisation of the ceil(n/16) tiles that are not padding  (factor rows, inverses, z to a per-workgroup scratch)
//   K(X, X*) of the workgroup's 64 candidates straight into the MFMA result registers
//   V = L^-1 K*,  q = sum V^2,  mu = V^T z,  variance, mean, EI / cost, arg-max over the 64 candidates
// A second, tiny launch reduces the per-workgroup winners of every set.  Every workgroup of a set repeats the
// set's factorisation (no inter-workgroup dependency; it is a few microseconds).
struct SmallShared {
    Diag2Shared blk;           // Ky / factor workspace, later the factor itself for the solve
    double xs[CBO_MAX_DIM][128];
    double sq[128], sv[128];
};
static_assert(sizeof(SmallShared) <= 163840, "one workgroup per CU");

constexpr int kSmallLd = kDiagLd;                              // scratch factor rows: [128][144], z in column 128
constexpr int kSmallScratch = 128 * kSmallLd + 8 * 256;        // doubles per workgroup: factor rows + inverses

template <int D>
__device__ __forceinline__ void small_assemble(SmallShared &sh, const cbo_small_set &st, int tiles)
{
    const int tid = threadIdx.x;
    const int i = tid >> 4, j = tid & 15;
    const double inv_l2 = 1.0 / (st.lengthscale * st.lengthscale);
    const bool causal = st.sv != nullptr;
    // Four tile pairs at a time, no branch around a value: one wave per SIMD has nothing to hide an exp's dependent
    // chain behind but the next value's chain (the points beyond n are zeros in LDS: computed, then replaced).
    int ti = 0, tj = 0;                                                // (uniform)
    while (ti < tiles) {
        int gis[4], gjs[4];
        bool due[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            due[u] = ti < tiles;
            gis[u] = 16 * (due[u] ? ti : 0) + i;
            gjs[u] = 16 * (due[u] ? tj : 0) + j;
            if (++tj >= tiles) { ++ti; tj = ti; }
        }
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gi = gis[u], gj = gjs[u];
            double xi[D], xj[D];
#pragma unroll
            for (int k = 0; k < D; ++k) { xi[k] = sh.xs[k][gi]; xj[k] = sh.xs[k][gj]; }
            double w = kernel_value<D>(xi, xj, sh.sq[gi], sh.sq[gj], st.variance, inv_l2, st.zero_diag && gi == gj);
            if (causal) w = __dadd_rn(w, __dmul_rn(sh.sv[gi], sh.sv[gj]));
            const double wd = __dadd_rn(w, st.diag_add);               // Ky = K + (noise + 1e-8) I
            w = (gi == gj) ? wd : w;
            const double pad = (gi == gj) ? 1.0 : 0.0;                 // identity padding
            v[u] = (gi < st.n && gj < st.n) ? w : pad;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (due[u]) sh.blk.S[gis[u]][gjs[u]] = v[u];
    }
}

template <int D>
__device__ __forceinline__ double small_kstar(const SmallShared &sh, const cbo_small_set &st, int row, const double *xc,
                                              double csq, double csv, double inv_l2)
{
    // (no branch around the value -- the four of a tile interleave; rows beyond n are zeros in LDS)
    double xi[D];
#pragma unroll
    for (int k = 0; k < D; ++k) xi[k] = sh.xs[k][row];
    double v = kernel_value<D>(xi, xc, sh.sq[row], csq, st.variance, inv_l2, false);
    const double vc = __dadd_rn(v, __dmul_rn(sh.sv[row], csv));
    v = (st.sv != nullptr) ? vc : v;
    return (row < st.n) ? v : 0.0;
}

template <int D>
__device__ __forceinline__ void small_kstar_tiles(const SmallShared &sh, const cbo_small_set &st, int tiles,
                                                  const double *xc, double csq, double csv, double inv_l2, int kq,
                                                  d4 (&acc)[8])
{
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (t < tiles) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = small_kstar<D>(sh, st, 16 * t + kq + 4 * r, xc, csq, csv, inv_l2);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = 0.0;
        }
    }
}

// The last workgroup of a set to finish (an atomic ticket) reduces the set's per-workgroup winners, hands the result
// record to the host (pinned, device-mapped memory; `seq` is stored last, after a system-scope fence, so that the host
// can poll it) and re-arms the set's status word and ticket for the next call.
__device__ __forceinline__ void small_set_finish(double bv, int64_t bi, int set, int slot, int blocks_per_set,
                                                 double *__restrict__ part_val, int64_t *__restrict__ part_idx,
                                                 int *__restrict__ info, int *__restrict__ ticket,
                                                 cbo_small_result *__restrict__ out, int seq, int *last_flag)
{
    const int tid = threadIdx.x;
    if (tid == 0) {
        part_val[slot] = bv;
        part_idx[slot] = bi;
        __threadfence();
        *last_flag = (atomicAdd(&ticket[set], 1) == blocks_per_set - 1) ? 1 : 0;
    }
    __syncthreads();
    if (*last_flag == 0 || tid >= 64) return;
    __threadfence();
    const int status = (tid == 0) ? atomicAdd(&info[set], 0) : 0;       // (in flight with the loads below)
    bv = -INFINITY;
    bi = INT64_MAX;
    for (int b = tid; b < blocks_per_set; b += 64) {
        const double v = __builtin_nontemporal_load(&part_val[set * blocks_per_set + b]);
        const int64_t i = __builtin_nontemporal_load(&part_idx[set * blocks_per_set + b]);
        if (better(v, i, bv, bi)) { bv = v; bi = i; }
    }
    wave_argmax(bv, bi);
    if (tid == 0) {
        out[set].best_val = bv;
        out[set].best_idx = bi;
        out[set].info = status;
        __threadfence_system();
        *reinterpret_cast<volatile int *>(&out[set].seq) = seq;
        info[set] = 0;
        ticket[set] = 0;
    }
}

// The model side of the one-workgroup kernels: points -> LDS, K(X,X) + diag and the rhs into the block, the
// factorisation of the `tiles` real tiles (factor rows, inverses, z to the workgroup's scratch), the factor back into
// LDS (what a tile solve reads), the inverses and z into registers.  Ends with loads in flight: the caller waits
// (s_waitcnt vmcnt(0) + barrier) before the solve.
// `phases`: 1 = points + assembly + factorisation only (the factor stays in the scratch), 2 = points + the factor from
// the scratch (somebody factored the model before this launch), 3 = both.
__device__ __forceinline__ void small_model_factor(SmallShared &sh, const cbo_small_set &st, int tiles, double *Us,
                                                   double *invs, int *info_word, double (&iv)[8][4], double (&zr)[8][4],
                                                   int phases = 3, bool skip_padding = false)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, kq = lane >> 4;
    // ---- the model's points and K(X,X) + diag, rhs, zero fill of what the factorisation reads beyond the tiles
    // staged: the model's new data are still in the caller's staging buffer (cbo_trial_step): prepared here, from there, with
    // prep_points_staged_kernel's arithmetic; the set's first workgroup also writes the resident copies
    const bool staged = st.stage != nullptr && (phases & 1);
    const bool writer = staged && blockIdx.x == 0;
    const double *ysrc = staged ? st.stage + (int64_t)st.n * st.d : st.y;
    const double *pmsrc = staged ? (st.sv ? st.stage + (int64_t)st.n * st.d + st.n : nullptr) : st.pm;
    double staged_y = 0.0, staged_pm = 0.0;
    if (tid < 128) {
        const bool in = tid < st.n;
        if (staged) {
            double x[CBO_MAX_DIM];
#pragma unroll
            for (int k = 0; k < CBO_MAX_DIM; ++k) x[k] = 0.0;
            double pvi = 0.0;
            if (in) {
                // (y and the prior mean are fetched with the points: one trip across the host link, not two)
                staged_y = ysrc[tid];
                if (pmsrc) staged_pm = pmsrc[tid];
#pragma unroll
                for (int k = 0; k < CBO_MAX_DIM; ++k)
                    if (k < st.d) {
                        double v = st.stage[(int64_t)tid * st.d + k];
                        if (writer) st.raw[(int64_t)tid * st.d + k] = v;
                        if (st.stage_ls) v = v / st.stage_ls[k];
                        x[k] = v;
                    }
                if (st.sv) pvi = st.stage[(int64_t)st.n * st.d + 2 * st.n + tid];
            }
            double sum;
            if (st.d == 8) {
                double r[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = __dmul_rn(x[k], x[k]);
                sum = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                                __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
            } else {
                sum = 0.0;
#pragma unroll
                for (int k = 0; k < CBO_MAX_DIM; ++k)
                    if (k < st.d) sum = __dadd_rn(sum, __dmul_rn(x[k], x[k]));
            }
            const double svi = (in && st.sv) ? sqrt(pvi) : 0.0;
#pragma unroll
            for (int k = 0; k < CBO_MAX_DIM; ++k)
                if (k < st.d) sh.xs[k][tid] = x[k];
            sh.sq[tid] = sum;
            sh.sv[tid] = svi;
            if (writer) {
#pragma unroll
                for (int k = 0; k < CBO_MAX_DIM; ++k)
                    if (k < st.d) const_cast<double *>(st.xs)[(int64_t)k * st.ld + tid] = x[k];
                const_cast<double *>(st.sq)[tid] = sum;
                if (st.sv) const_cast<double *>(st.sv)[tid] = svi;
                if (in) {
                    const_cast<double *>(st.y)[tid] = staged_y;
                    if (st.sv) {
                        const_cast<double *>(st.pm)[tid] = staged_pm;
                        st.pv[tid] = pvi;
                    }
                }
            }
        } else {
            if (in && (phases & 1)) {                     // (with the points: the rhs does not wait for a second trip)
                staged_y = ysrc[tid];
                if (pmsrc) staged_pm = pmsrc[tid];
            }
            for (int k = 0; k < st.d; ++k) sh.xs[k][tid] = in ? st.xs[(int64_t)k * st.ld + tid] : 0.0;
            sh.sq[tid] = in ? st.sq[tid] : 0.0;
            sh.sv[tid] = (in && st.sv) ? st.sv[tid] : 0.0;
        }
    }
    __syncthreads();
    SSTAMP(1);
    if (phases & 1) {
    switch (st.d) {
        case 1: small_assemble<1>(sh, st, tiles); break;
        case 2: small_assemble<2>(sh, st, tiles); break;
        case 3: small_assemble<3>(sh, st, tiles); break;
        case 4: small_assemble<4>(sh, st, tiles); break;
        case 5: small_assemble<5>(sh, st, tiles); break;
        case 6: small_assemble<6>(sh, st, tiles); break;
        case 7: small_assemble<7>(sh, st, tiles); break;
        default: small_assemble<8>(sh, st, tiles); break;
    }
    {
        const int rows = 16 * tiles;
        for (int r = tid >> 4; r < rows; r += 16)
            for (int c = rows + (tid & 15); c < kDiagLd; c += 16) {
                if (c == 128 && r < st.n) continue;                                            // (the rhs: below)
                sh.blk.S[r][c] = 0.0;
            }
        if (tid < st.n) sh.blk.S[tid][128] = pmsrc ? __dadd_rn(staged_y, -staged_pm) : staged_y;   // r = y - m(X)
    }
    __syncthreads();
    SSTAMP(2);
    diag128_factor_in_lds(sh.blk, Us, kSmallLd, 0, 128, invs, info_word, nullptr, tiles, nullptr,
                          skip_padding ? (st.n - 16 * (tiles - 1) + 3) / 4 : 4);
    SSTAMP(3);
    // (ends with a barrier.)  Every wave's stores of factor rows / inverses / z are complete before anyone re-reads them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    }
    if (!(phases & 2)) return;

    // ---- the factor back into LDS (rows of the factored tiles; the solve reads nothing else), inverses and z to registers
    {
        const unsigned s0 = lds_byte_address(&sh.blk.S[0][0]);
        const int rows = 16 * tiles;
        for (int p = wave; p < rows; p += 4)
            glds16(Us + (int64_t)p * kSmallLd + lane * 2, __builtin_amdgcn_readfirstlane(s0 + 8u * (unsigned)(p * kDiagLd)));
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            iv[s][kk] = (s < tiles) ? invs[s * 256 + (4 * kk + kq) * 16 + lc] : 0.0;
            zr[s][kk] = (s < tiles) ? Us[(int64_t)(16 * s + kq + 4 * kk) * kSmallLd + 128] : 0.0;
        }
}

// Up to kSmallByValue descriptors travel as kernel arguments (no read across the host link before the first
// instruction that needs them); longer lists are read from the pinned array.
constexpr int kSmallByValue = 8;
struct SmallSetArgs { cbo_small_set s[kSmallByValue]; };

template <bool BYVAL>
__global__ __launch_bounds__(256) void small_sets_kernel(const SmallSetArgs byval, const cbo_small_set *__restrict__ sets,
                                                         double *scratch, int blocks_per_set,
                                                         double *__restrict__ part_val, int64_t *__restrict__ part_idx,
                                                         int *__restrict__ info, int *__restrict__ ticket,
                                                         cbo_small_result *__restrict__ out, int seq, int phases)
{
    __shared__ int last_flag;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    SmallShared &sh = *reinterpret_cast<SmallShared *>(smem_raw);
    const int set = blockIdx.y, blk = blockIdx.x;
    SSTAMP(0);
    const cbo_small_set st = BYVAL ? byval.s[set] : sets[set];
    const int slot = set * blocks_per_set + blk;
    if (phases == 1) {                                            // one workgroup per set: factor it, nothing else
        double ivx[8][4], zrx[8][4];
        double *fs = scratch + (int64_t)(set * blocks_per_set) * kSmallScratch;
        small_model_factor(sh, st, (st.n + 15) / 16, fs, fs + 128 * kSmallLd, &info[set], ivx, zrx, 1, true);
        return;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, kq = lane >> 4;
    if ((int64_t)blk * 64 >= st.m) {                              // no candidates left for this workgroup
        small_set_finish(-INFINITY, INT64_MAX, set, slot, blocks_per_set, part_val, part_idx, info, ticket, out, seq,
                         &last_flag);
        return;
    }
    const int tiles = (st.n + 15) / 16;
    // phases 3: every workgroup factors the model itself, into its own scratch slot; phases 2: the set's slot 0 holds it
    double *my = scratch + (int64_t)(phases == 2 ? set * blocks_per_set : slot) * kSmallScratch;
    double *Us = my, *invs = my + 128 * kSmallLd;

    // this wave's 16 candidates: fetched now, used after the factorisation (their latency is off the chain)
    const int64_t c = (int64_t)blk * 64 + wave * 16 + lc;
    const int64_t cc = (c < st.m) ? c : st.m - 1;                  // clamped: lanes beyond the set compute, nobody looks
    double xc[CBO_MAX_DIM];
#pragma unroll
    for (int k = 0; k < CBO_MAX_DIM; ++k) xc[k] = (k < st.d) ? st.cxs[(int64_t)k * st.cld + cc] : 0.0;
    const double csq = st.csq[cc], csv = st.csv ? st.csv[cc] : 0.0;
    const double cpm_c = st.cpm ? st.cpm[cc] : 0.0, cpv_c = st.cpv ? st.cpv[cc] : 0.0;

    double iv[8][4], zr[8][4];
    small_model_factor(sh, st, tiles, Us, invs, &info[set], iv, zr, phases, true);
    SSTAMP(4);
    // ---- K(X, X*) of this wave's 16 candidates, straight into the result layout
    const double inv_l2 = 1.0 / (st.lengthscale * st.lengthscale);
    d4 acc[8];
    switch (st.d) {
        case 1: small_kstar_tiles<1>(sh, st, tiles, xc, csq, csv, inv_l2, kq, acc); break;
        case 2: small_kstar_tiles<2>(sh, st, tiles, xc, csq, csv, inv_l2, kq, acc); break;
        case 3: small_kstar_tiles<3>(sh, st, tiles, xc, csq, csv, inv_l2, kq, acc); break;
        case 4: small_kstar_tiles<4>(sh, st, tiles, xc, csq, csv, inv_l2, kq, acc); break;
        case 5: small_kstar_tiles<5>(sh, st, tiles, xc, csq, csv, inv_l2, kq, acc); break;
        case 6: small_kstar_tiles<6>(sh, st, tiles, xc, csq, csv, inv_l2, kq, acc); break;
        case 7: small_kstar_tiles<7>(sh, st, tiles, xc, csq, csv, inv_l2, kq, acc); break;
        default: small_kstar_tiles<8>(sh, st, tiles, xc, csq, csv, inv_l2, kq, acc); break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    SSTAMP(5);
    // ---- V = L^-1 K*, q = sum V^2, mu = V^T z (lane partials, then over the four lane groups: the strip kernel's order)
    double qacc = 0.0, macc = 0.0;
    panel_solve_tiles(&sh.blk.S[kq][lc], acc, iv, tiles, [&](int s, const d4 &x) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            qacc = fma(x[r], x[r], qacc);
            macc = fma(x[r], zr[s][r], macc);
        }
    });
    qacc += __shfl_xor(qacc, 16);
    qacc += __shfl_xor(qacc, 32);
    macc += __shfl_xor(macc, 16);
    macc += __shfl_xor(macc, 32);

    SSTAMP(6);
    // ---- epilogue and the workgroup's arg-max
    AcqParams p;
    p.variance = st.variance; p.noise_var = st.noise_var; p.y_best = st.y_best; p.ei_jitter = st.ei_jitter;
    p.cost = st.cost; p.task = st.task; p.include_noise = 1; p.want_ei = 1;
    double bv = -INFINITY;
    int64_t bi = INT64_MAX;
    if (kq == 0 && c < st.m) {
        double mean, var;
        posterior_of(qacc, macc, cpm_c, cpv_c, st.sv != nullptr, p, mean, var);
        bv = acquisition_of(mean, var, p);
        bi = c + st.index_offset;
    }
    wave_argmax(bv, bi);
    double *red_v = &sh.sq[0];                         // free by now
    int64_t *red_i = reinterpret_cast<int64_t *>(&sh.sv[0]);
    __syncthreads();
    if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (better(red_v[w], red_i[w], bv, bi)) { bv = red_v[w]; bi = red_i[w]; }
    }
    SSTAMP(7);
    small_set_finish(bv, bi, set, slot, blocks_per_set, part_val, part_idx, info, ticket, out, seq, &last_flag);
    SSTAMP(8);
}

// ------------------------------------------------------------------------------------------------
// Log marginal likelihood and its analytic gradients for a model of at most 128 observations -- what every iterate
// of the hyper-parameter MLE asks for (src/CBO.py:173 -> GPy model.optimize()) -- in ONE launch of one workgroup, the
// model not fitted beforehand:
//   K(X,X) + diag -> factorisation (as small_sets_kernel)          sum log diag(U), z^T z
//   V = L^-1 (identity right-hand sides through the same tile solve)  alpha = V^T z, diag(Ky^-1) = column sums of V^2
//   W = V^T V tile by tile on the matrix cores (V from the workgroup's scratch), each tile contracted at once with
//   the kernel and its lengthscale derivatives: the sums of lml_grad_tile_kernel (kernels_kmat.hip), GPy's quirk of
//   the causal term in the variance gradient included.
// The record goes to pinned host memory, closed by the call's sequence number (the host polls it).
template <int D>
__global__ __launch_bounds__(256) void small_lml_kernel(const cbo_small_set st, double *scratch, int *__restrict__ info,
                                                        cbo_small_lml_result *__restrict__ out, int seq)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    SmallShared &sh = *reinterpret_cast<SmallShared *>(smem_raw);
    __shared__ double alpha_s[128];
    __shared__ double red[4][kSmallLmlTerms];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, kq = lane >> 4;
    const int tiles = (st.n + 15) / 16, rows = 16 * tiles;
    double *Us = scratch, *invs = scratch + 128 * kSmallLd, *Vg = scratch + kSmallScratch;
    double iv[8][4], zr[8][4];
    small_model_factor(sh, st, tiles, Us, invs, info, iv, zr);
    if (tid < 128) alpha_s[tid] = 0.0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- V = L^-1: column tile ct of the identity through the tile solve; alpha and diag(Ky^-1) fall out
    double trw = 0.0;
    for (int ct = wave; ct < tiles; ct += 4) {
        d4 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = (t == ct && kq + 4 * r == lc) ? 1.0 : 0.0;
        double qacc = 0.0, macc = 0.0;
        double *Vc = Vg + 16 * ct + lc;
        panel_solve_tiles(&sh.blk.S[kq][lc], acc, iv, tiles, [&](int s2, const d4 &x) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Vc[(int64_t)(16 * s2 + kq + 4 * r) * kSmallLd] = x[r];
                qacc = fma(x[r], x[r], qacc);
                macc = fma(x[r], zr[s2][r], macc);
            }
        });
        qacc += __shfl_xor(qacc, 16);
        qacc += __shfl_xor(qacc, 32);
        macc += __shfl_xor(macc, 16);
        macc += __shfl_xor(macc, 32);
        const int col = 16 * ct + lc;
        if (kq == 0 && col < st.n) {
            alpha_s[col] = macc;
            trw += qacc;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- gradient sums over the upper tile pairs, one pair per wave at a time
    const double inv_l2 = st.ard ? 1.0 : 1.0 / (st.lengthscale * st.lengthscale);
    double sum[1 + D];
#pragma unroll
    for (int k = 0; k <= D; ++k) sum[k] = 0.0;
    int pair = 0;
    for (int ti = 0; ti < tiles; ++ti)
        for (int tj = ti; tj < tiles; ++tj, ++pair) {
            if ((pair & 3) != wave) continue;                       // uniform per wave
            d4 w = {0.0, 0.0, 0.0, 0.0}, w2 = {0.0, 0.0, 0.0, 0.0};
            const double *Va = Vg + 16 * ti + lc, *Vb = Vg + 16 * tj + lc;
            // V is lower triangular: rows above tile tj contribute nothing to column tile tj
            for (int k0 = 16 * tj; k0 < rows; k0 += 8) {
                w = MFMA_F64(Va[(int64_t)(k0 + kq) * kSmallLd], Vb[(int64_t)(k0 + kq) * kSmallLd], w);
                w2 = MFMA_F64(Va[(int64_t)(k0 + 4 + kq) * kSmallLd], Vb[(int64_t)(k0 + 4 + kq) * kSmallLd], w2);
            }
            w += w2;
            const int gj = 16 * tj + lc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = 16 * ti + kq + 4 * r;
                if (gi >= st.n || gj >= st.n || gj < gi) continue;
                double r2 = 0.0, d2k[D];
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double df = sh.xs[k][gi] - sh.xs[k][gj];
                    d2k[k] = df * df * inv_l2;
                    r2 += d2k[k];
                }
                const double kv = st.variance * exp_nonpositive(-0.5 * r2);
                const double m = (gi == gj ? 1.0 : 2.0) * (alpha_s[gi] * alpha_s[gj] - w[r]);
                const double mk = m * kv;
                sum[0] += mk + m * (sh.sv[gi] * sh.sv[gj]);
#pragma unroll
                for (int k = 0; k < D; ++k) sum[1 + k] = fma(mk, d2k[k], sum[1 + k]);
            }
        }
    // ---- z^T z, sum log diag(U), alpha^T alpha, tr(Ky^-1); everything reduced over the workgroup
    double terms[kSmallLmlTerms];
#pragma unroll
    for (int k = 0; k < kSmallLmlTerms; ++k) terms[k] = 0.0;
#pragma unroll
    for (int k = 0; k <= D; ++k) terms[k] = sum[k];
    if (tid < st.n) {
        const double zi = Us[(int64_t)tid * kSmallLd + 128];
        terms[1 + CBO_MAX_DIM + 0] = zi * zi;
        terms[1 + CBO_MAX_DIM + 1] = log(sh.blk.S[tid][tid]);
        terms[1 + CBO_MAX_DIM + 2] = alpha_s[tid] * alpha_s[tid];
    }
    terms[1 + CBO_MAX_DIM + 3] = trw;
#pragma unroll
    for (int k = 0; k < kSmallLmlTerms; ++k) {
        double v = terms[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < kSmallLmlTerms; ++k) out->terms[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        out->info = atomicAdd(info, 0);
        __threadfence_system();
        *reinterpret_cast<volatile int *>(&out->seq) = seq;
        *info = 0;
    }
}

size_t small_lml_scratch_doubles() { return (size_t)kSmallScratch + (size_t)128 * kSmallLd; }

void launch_small_lml(hipStream_t s, const cbo_small_set &st, double *scratch, int *info, cbo_small_lml_result *out, int seq)
{
#define CBO_LAUNCH_LML(D)                                                                                              \
    do {                                                                                                               \
        hipFuncSetAttribute(reinterpret_cast<const void *>(small_lml_kernel<D>),                                       \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmallShared));                     \
        hipLaunchKernelGGL(small_lml_kernel<D>, dim3(1), dim3(256), sizeof(SmallShared), s, st, scratch, info, out,    \
                           seq);                                                                                       \
    } while (0)
    switch (st.d) {
        case 1: CBO_LAUNCH_LML(1); break;
        case 2: CBO_LAUNCH_LML(2); break;
        case 3: CBO_LAUNCH_LML(3); break;
        case 4: CBO_LAUNCH_LML(4); break;
        case 5: CBO_LAUNCH_LML(5); break;
        case 6: CBO_LAUNCH_LML(6); break;
        case 7: CBO_LAUNCH_LML(7); break;
        default: CBO_LAUNCH_LML(8); break;
    }
#undef CBO_LAUNCH_LML
}

size_t small_sets_scratch_doubles(int n_sets, int blocks_per_set) { return (size_t)n_sets * blocks_per_set * kSmallScratch; }

void launch_small_sets(hipStream_t s, const cbo_small_set *sets, int n_sets, int blocks_per_set, double *scratch,
                       double *part_val, int64_t *part_idx, int *info, int *ticket, cbo_small_result *out, int seq)
{
    // once per device and instantiation (several devices in one process each need it; a failed attempt is repeated by
    // the next call; the launch itself reports what is wrong if it never succeeds) -- the call costs a microsecond of
    // the forty a reference-scale trial takes
    {
        static std::atomic<unsigned long long> opted[2];
        int dev = 0;
        const bool byval = n_sets <= kSmallByValue;
        if (hipGetDevice(&dev) != hipSuccess || !((opted[byval].load(std::memory_order_relaxed) >> (dev & 63)) & 1ull)) {
            const void *fn = byval ? reinterpret_cast<const void *>(small_sets_kernel<true>)
                                   : reinterpret_cast<const void *>(small_sets_kernel<false>);
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmallShared)) == hipSuccess)
                opted[byval].fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
        }
    }
    SmallSetArgs args{};
    const dim3 grid((unsigned)blocks_per_set, (unsigned)n_sets);
    // Few candidate blocks per set (the reference's 100-200 candidates): every workgroup factors its set's model itself,
    // ONE launch, no dependency between workgroups.  Many blocks per set (16k-candidate grids on 25 coral sets: 6400
    // workgroups): factoring the model 256 times over costs more than a second launch -- one workgroup per set factors,
    // then the sweep workgroups start from the factor.
    static const int two_phase_from = [] { const char *e = std::getenv("CBO_HIP_SMALL_TWO_PHASE"); return e ? std::atoi(e) : 12; }();
    const bool two_phase = two_phase_from > 0 && blocks_per_set >= two_phase_from;
    auto launch = [&](const dim3 &g, int phases) {
        if (n_sets <= kSmallByValue)
            hipLaunchKernelGGL(small_sets_kernel<true>, g, dim3(256), sizeof(SmallShared), s, args, sets, scratch,
                               blocks_per_set, part_val, part_idx, info, ticket, out, seq, phases);
        else
            hipLaunchKernelGGL(small_sets_kernel<false>, g, dim3(256), sizeof(SmallShared), s, args, sets, scratch,
                               blocks_per_set, part_val, part_idx, info, ticket, out, seq, phases);
    };
    if (n_sets <= kSmallByValue) std::memcpy(args.s, sets, sizeof(cbo_small_set) * (size_t)n_sets);
    if (two_phase) {
        launch(dim3(1u, (unsigned)n_sets), 1);
        launch(grid, 2);
    } else {
        launch(grid, 3);
    }
}

// ------------------------------------------------------------------------------------------------
// SYRK: C[i][j] -= sum_k P[k][i] P[k][j] on the upper tiles of the trailing block, P = the panel rows
// [r0, r0+n1).  TS x TS tile per workgroup, 2x2 waves, each wave (TS/2)^2 via 16x16x4 f64 MFMAs with
// both operand fragments read straight from the panel rows (4 row segments of 128 B per load).
// Extra blocks (blockIdx.x == nt) update the rhs column: r[i] -= sum_k P[k][i] z[k].
template <int TS>
__global__ __launch_bounds__(256) void syrk_kernel(double *A, int64_t lda, int r0, int n1, int c0, int nt, int rcol,
                                                   int ti_begin, const int *__restrict__ skip_if)
{
    if (__builtin_nontemporal_load(skip_if) != 0) return;
    const int tj = blockIdx.x, ti = blockIdx.y + ti_begin;
    const int tid = threadIdx.x;
    const double *P = A + (int64_t)r0 * lda;
    if (tj == nt) {
        // rhs column for the TS rows of tile ti
        constexpr int G = 256 / TS;
        __shared__ double part[256];
        const int i = tid % TS, g = tid / TS;
        const int64_t gi = c0 + (int64_t)ti * TS + i;
        // all loads of a 16-step batch are issued before the first use (L2 latency paid once per batch)
        double s = 0.0;
        for (int k0 = g; k0 < n1; k0 += 16 * G) {
            double pv[16], zv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + u * G;
                const bool ok = k < n1;
                pv[u] = ok ? P[(int64_t)k * lda + gi] : 0.0;
                zv[u] = ok ? P[(int64_t)k * lda + rcol] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) s = fma(pv[u], zv[u], s);
        }
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
    static_assert(MT == 2, "the 16-byte interleaved fragment map below is written for 32x32 wave tiles");
    const int lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, kq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    if (ti == tj && wr == 1 && wc == 0) return;      // strictly-lower quadrant of a diagonal tile
    const int64_t ib = c0 + (int64_t)ti * TS + wr * WT;
    const int64_t jb = c0 + (int64_t)tj * TS + wc * WT;
    // Interleaved tile map: MFMA tile (m, nn) of the wave's 32x32 block covers rows ib + 2i + m and columns
    // jb + 2j + nn, so one 16-byte load per lane yields the fragments of both m (resp. nn) and every access
    // to C is 16 bytes per lane as well.
    d4 acc[MT][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nn = 0; nn < MT; ++nn) acc[m][nn] = d4{0.0, 0.0, 0.0, 0.0};
    const double *Pa = P + (int64_t)kq * lda + ib + 2 * lc;
    const double *Pb = P + (int64_t)kq * lda + jb + 2 * lc;
    // software pipeline over chunks of CH k-steps: the fragments of chunk c+1 are in flight (L2 latency)
    // while the MFMAs of chunk c issue; n1 is a multiple of 4*CH at every call site (128-row panels)
    constexpr int CH = 4;
    d2 a[2][CH], b[2][CH];
    auto load_chunk = [&](int k0, d2 (&aa)[CH], d2 (&bb)[CH]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < CH; ++ks) {
            aa[ks] = *reinterpret_cast<const d2 *>(&Pa[(int64_t)(k0 + 4 * ks) * lda]);
            bb[ks] = *reinterpret_cast<const d2 *>(&Pb[(int64_t)(k0 + 4 * ks) * lda]);
        }
    };
    auto mfma_chunk = [&](const d2 (&aa)[CH], const d2 (&bb)[CH]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < CH; ++ks)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < MT; ++nn) acc[m][nn] = MFMA_F64(aa[ks][m], bb[ks][nn], acc[m][nn]);
    };
    load_chunk(0, a[0], b[0]);
    // the C tile is fetched up front so that its (MALL/HBM) latency hides under the MFMAs
    d2 cv[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            cv[m][r] = *reinterpret_cast<const d2 *>(&A[(ib + 2 * (kq + 4 * r) + m) * lda + jb + 2 * lc]);
    for (int k0 = 0; k0 < n1; k0 += 8 * CH) {
        if (k0 + 4 * CH < n1) load_chunk(k0 + 4 * CH, a[1], b[1]);
        mfma_chunk(a[0], b[0]);
        if (k0 + 8 * CH < n1) load_chunk(k0 + 8 * CH, a[0], b[0]);
        if (k0 + 4 * CH < n1) mfma_chunk(a[1], b[1]);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d2 o = cv[m][r];
            o[0] -= acc[m][0][r];
            o[1] -= acc[m][1][r];
            *reinterpret_cast<d2 *>(&A[(ib + 2 * (kq + 4 * r) + m) * lda + jb + 2 * lc]) = o;
        }
}

// The trailing update of the NEXT panel's 128 rows only -- the one piece of the update that sits on the chain (the
// next diagonal block and row panel wait for it).  It is bound by the matrix pipe of single waves: a 64x64 tile per
// workgroup means 2x2 MFMA tiles x K/4 steps = 256 dependent-free but serial MFMAs per wave at K = 256 (7.8 us).
// Here a workgroup takes 32 rows x 64 columns (wave w: both 16-row tiles x its 16 columns), which halves the MFMAs
// per wave and doubles the workgroups (256 at N = 4096: every CU busy).  blockIdx.x == nt: the rhs column.
__global__ __launch_bounds__(256) void syrk_rows_kernel(double *A, int64_t lda, int r0, int n1, int c0, int nt, int rcol,
                                                        const int *__restrict__ skip_if)
{
    CHAIN_PRIORITY();
    if (__builtin_nontemporal_load(skip_if) != 0) return;
    const int tj = blockIdx.x, ti = blockIdx.y;                  // 64-column tile, 32-row tile (4 of them)
    const int tid = threadIdx.x;
    const double *P = A + (int64_t)r0 * lda;
    if (tj == nt) {
        __shared__ double part[256];
        const int i = tid & 31, g = tid >> 5;
        const int64_t gi = c0 + (int64_t)ti * 32 + i;
        double s = 0.0;
        for (int k0 = g; k0 < n1; k0 += 16 * 8) {
            double pv[16], zv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + u * 8;
                const bool ok = k < n1;
                pv[u] = ok ? P[(int64_t)k * lda + gi] : 0.0;
                zv[u] = ok ? P[(int64_t)k * lda + rcol] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) s = fma(pv[u], zv[u], s);
        }
        part[tid] = s;
        __syncthreads();
        if (g == 0) {
            double tot = 0.0;
#pragma unroll
            for (int gg = 0; gg < 8; ++gg) tot += part[gg * 32 + i];
            A[gi * lda + rcol] -= tot;
        }
        return;
    }
    const int64_t ib = c0 + (int64_t)ti * 32;                    // first row of the tile (= a column of the panel)
    const int64_t jb0 = c0 + (int64_t)tj * 64;
    if (jb0 + 63 < ib) return;                                   // entirely below the diagonal
    const int lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, kq = lane >> 4;
    const int64_t jb = jb0 + wave * 16;
    d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
    const double *Pa = P + (int64_t)kq * lda + ib + lc;          // A[i = lc][k = kq] = P[k][ib + i]  (+16: second row tile)
    const double *Pb = P + (int64_t)kq * lda + jb + lc;          // B[k = kq][j = lc] = P[k][jb + j]
    constexpr int CH = 8;                                        // k-steps per software-pipeline chunk (16: more registers, no faster)
    double a0[2][CH], a1[2][CH], b[2][CH];
    auto load_chunk = [&](int k0, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < CH; ++ks) {
            a0[buf][ks] = Pa[(int64_t)(k0 + 4 * ks) * lda];
            a1[buf][ks] = Pa[(int64_t)(k0 + 4 * ks) * lda + 16];
            b[buf][ks] = Pb[(int64_t)(k0 + 4 * ks) * lda];
        }
    };
    load_chunk(0, 0);
    double cv[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[m][r] = A[(ib + 16 * m + kq + 4 * r) * lda + jb + lc];
    // n1 is 128 or 256: a multiple of 2 chunks of 32 rows
    for (int k0 = 0; k0 < n1; k0 += 8 * CH) {
        load_chunk(k0 + 4 * CH, 1);
#pragma unroll
        for (int ks = 0; ks < CH; ++ks) {
            acc[0] = MFMA_F64(a0[0][ks], b[0][ks], acc[0]);
            acc[1] = MFMA_F64(a1[0][ks], b[0][ks], acc[1]);
        }
        if (k0 + 8 * CH < n1) load_chunk(k0 + 8 * CH, 0);
#pragma unroll
        for (int ks = 0; ks < CH; ++ks) {
            acc[0] = MFMA_F64(a0[1][ks], b[1][ks], acc[0]);
            acc[1] = MFMA_F64(a1[1][ks], b[1][ks], acc[1]);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) A[(ib + 16 * m + kq + 4 * r) * lda + jb + lc] = cv[m][r] - acc[m][r];
}

// the first `rows` rows below the n1 panel rows [r0, r0 + n1) against those panel rows: columns from c0 = r0 + n1 on,
// n2 of them (rows: 128 = the next panel, 256 = the next pair of panels)
// `done`: an event that completes with the launch (the launch carries it: no marker packet on the stream)
static void launch_syrk_rows(hipStream_t s, double *A, int64_t lda, int r0, int n1, int n2, int rcol, const int *skip_if,
                             int rows = 128, hipEvent_t done = nullptr)
{
    const int c0 = r0 + n1;
    const int nt = n2 / 64;
    if (nt <= 0) { if (done) hipEventRecord(done, s); return; }
    if (rows > n2) rows = n2;
    hipExtLaunchKernelGGL(syrk_rows_kernel, dim3(nt + 1, rows / 32), dim3(256), 0, s, nullptr, done, 0, A, lda, r0, n1, c0,
                          nt, rcol, skip_if);
}

// tile rows [ti_begin, ti_end) of the trailing update (64-row tiles counted from the first trailing row)
static void launch_syrk(hipStream_t s, double *A, int64_t lda, int r0, int n1, int n2, int rcol, int ti_begin,
                        int ti_end, const int *skip_if)
{
    const int c0 = r0 + n1;
    const int nt = n2 / 64;
    if (ti_end > nt) ti_end = nt;
    if (ti_end <= ti_begin) return;
    hipLaunchKernelGGL(syrk_kernel<64>, dim3(nt + 1, ti_end - ti_begin), dim3(256), 0, s, A, lda, r0, n1, c0, nt, rcol,
                       ti_begin, skip_if);
}

// Rows [r0, r0 + klen) of U (all columns) and of z are final on stream `chain`: hand them to the sweep.
// Two sweep streams with a look-ahead of one panel pair: `stream` solves the pair's rows of V (sd) and folds
// them into the NEXT pair's rows only (first); `bulk` folds them into everything below that (rest).  The
// bulk launches then follow each other without a gap while sd/first of the following pair run beside them:
//   stream: wait chain[p]; sd(p); record sd[p]; wait rest[p-1]; first(p)      (first(p) rewrites rows that
//   bulk  : wait sd[p]; rest(p); record rest[p]                                rest(p-1) also rewrites)
void sweep_pipe_pair(const SweepPipe &pipe, hipStream_t chain, const double *A, int64_t lda, const double *invDt,
                     int64_t n_pad, int p, int r0, int klen)
{
    auto pipe_event = [&](int kind, int pp) -> hipEvent_t {      // kind 0: chain, 1: sd done, 2: rest done
        std::vector<hipEvent_t> &ev = *pipe.events;
        const size_t slot = (size_t)(3 * pp + kind);
        while (ev.size() <= slot) {
            hipEvent_t e;
            hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
            ev.push_back(e);
        }
        return ev[slot];
    };
    const int64_t cols = (pipe.lower_tri && r0 + klen < pipe.m_pad) ? (int64_t)(r0 + klen) : pipe.m_pad;
    const double m = (double)cols;
    hipEventRecord(pipe_event(0, p), chain);
    hipStreamWaitEvent(pipe.stream, pipe_event(0, p), 0);
    if (pipe.mark) pipe.mark(pipe.user, pipe.stream, 1, (double)klen * (double)klen * m);
    launch_trsm_strips(pipe.stream, A + (int64_t)r0 * lda + r0, lda, invDt + (int64_t)(r0 / 16) * 256,
                       pipe.V + (int64_t)r0 * pipe.ldv, pipe.ldv, klen, cols, pipe.zvec + r0, pipe.q, pipe.mu, true,
                       pipe.half_lds);
    if (pipe.mark) pipe.mark(pipe.user, pipe.stream, 0, 0.0);
    hipEventRecord(pipe_event(1, p), pipe.stream);
    const int below = r0 + klen;
    if (below >= (int)n_pad) return;
    const int first_end = (below + 256 < (int)n_pad) ? below + 256 : (int)n_pad;
    auto update = [&](hipStream_t st, int k0, int kl, int i0, int i1) {
        if (i0 >= i1) return;
        if (pipe.mark) pipe.mark(pipe.user, st, 1, 2.0 * (double)kl * (double)(i1 - i0) * m);
        launch_trsm_update(st, A, lda, pipe.V, pipe.ldv, k0, kl, i0, i1, cols, pipe.chunk_blocks, pipe.half_lds);
        if (pipe.mark) pipe.mark(pipe.user, st, 0, 0.0);
    };
    // Groups of G pairs, as in launch_cholesky's bulk updates (and for the same reason: the update kernel's rate grows
    // with K -- 0.68 of the fp64 MFMA peak at K = 256, 0.76 at 512, 0.83 at 1024 -- and it accumulates sequentially into
    // C, so one K = 256 G pass has the bits of G passes of K = 256).  Group j = pairs [G j, G j + G), G_g = the rows of
    // group g:
    //   stream: after sd(p), p in group j:  first: rows of pair p+1 -= pair p;
    //                                       near:  the rows from pair p+2 to the end of G_{j+1} -= pair p     (K = 256)
    //           (the first near of a group waits for restA of the group before)
    //   bulk  : [wait sd of the group's last pair]  restA(j): G_{j+2} -= group j;  restB(j): everything below -= group j
    // A pair whose group is not entirely in the pipeline (the remainder of the count, a 128-row remainder) goes alone as
    // before.  The caller decides (pipe.group = G): groups pay where the bulk stream is the pipeline's bottleneck (a
    // round of strips or more per CU); with few strips the pairs schedule starts its updates earlier and wins
    // (profiles/r03_schedule_crossover.txt).
    const int G = pipe.group, lead = pipe.lead;
    const int full_pairs = ((pipe.tail_begin < (int)n_pad) ? pipe.tail_begin : (int)n_pad) / 256;
    // pipe.lead pairs go alone ahead of the first group (round 5: with groups from the first pair on, the bulk stream idled
    // until the SECOND pair was solved -- a fifth of the pipelined phase at C2)
    const bool grouped = G >= 2 && !pipe.lower_tri && klen == 256 && p >= lead && ((p - lead) / G + 1) * G + lead <= full_pairs;
    if (grouped) {
        const int j = (p - lead) / G, i = (p - lead) % G;
        const int g0 = 256 * (lead + G * j);                                                       // the group's first row
        const int next_end = (g0 + 512 * G < (int)n_pad) ? g0 + 512 * G : (int)n_pad;              // end of G_{j+1}
        // the first group behind lead pairs: its rows were last written by the bulk update of the pair before it
        if (i == 0 && j == 0 && lead > 0) hipStreamWaitEvent(pipe.stream, pipe_event(2, lead - 1), 0);
        update(pipe.stream, r0, 256, below, first_end);
        if (i == 0 && j >= 1) hipStreamWaitEvent(pipe.stream, pipe_event(2, lead + G * (j - 1)), 0);   // restA of the group before
        update(pipe.stream, r0, 256, r0 + 512, next_end);
        if (i < G - 1) return;
        hipStreamWaitEvent(pipe.bulk, pipe_event(1, p), 0);
        const int a1 = (g0 + 768 * G < (int)n_pad) ? g0 + 768 * G : (int)n_pad;                    // end of G_{j+2}
        update(pipe.bulk, g0, 256 * G, next_end, a1);
        hipEventRecord(pipe_event(2, lead + G * j), pipe.bulk);                      // restA (the first pair's slot)
        update(pipe.bulk, g0, 256 * G, a1, (int)n_pad);
        hipEventRecord(pipe_event(2, p), pipe.bulk);                                 // restB: what a successor waits for
        return;
    }
    if (p > 0) hipStreamWaitEvent(pipe.stream, pipe_event(2, p - 1), 0);
    update(pipe.stream, r0, klen, below, first_end);
    hipStreamWaitEvent(pipe.bulk, pipe_event(1, p), 0);
    update(pipe.bulk, r0, klen, first_end, (int)n_pad);
    hipEventRecord(pipe_event(2, p), pipe.bulk);
}

// The rows the pairs did not take, [tail_begin, n_pad): every earlier pair has been folded into them (first of the
// last pair on `stream`, rest of the last pair on `bulk`) and the factor is complete on `chain`: one left-looking
// launch of the strip kernel on the sub-problem, on the chain stream itself (every CU, full-LDS variant).
void sweep_pipe_tail(const SweepPipe &pipe, hipStream_t chain, const double *A, int64_t lda, const double *invDt,
                     int64_t n_pad, int pairs_done)
{
    std::vector<hipEvent_t> &ev = *pipe.events;
    const size_t slot = (size_t)(3 * pairs_done);
    while (ev.size() <= slot) {
        hipEvent_t e;
        hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
        ev.push_back(e);
    }
    if (pairs_done > 0) {
        hipEventRecord(ev[slot], pipe.stream);                                   // sd and first of the last pair
        hipStreamWaitEvent(chain, ev[slot], 0);
        hipStreamWaitEvent(chain, ev[(size_t)(3 * (pairs_done - 1) + 2)], 0);      // rest of the last pair
    }
    const int t0 = pipe.tail_begin;
    const double rows = (double)((int)n_pad - t0);
    if (pipe.mark) pipe.mark(pipe.user, chain, 1, rows * rows * (double)pipe.m_pad);
    launch_trsm_strips(chain, A + (int64_t)t0 * lda + t0, lda, invDt + (int64_t)(t0 / 16) * 256,
                       pipe.V + (int64_t)t0 * pipe.ldv, pipe.ldv, (int64_t)n_pad - t0, pipe.m_pad, pipe.zvec + t0, pipe.q,
                       pipe.mu, true, false);
    if (pipe.mark) pipe.mark(pipe.user, chain, 0, 0.0);
}

__global__ void zero_ints_kernel(int *p, int n)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0;
}

// Look-ahead of one panel pair: the bulk of pair p-1's trailing update (everything below pair p+1's rows) runs on the
// side stream while the main stream factors and solves pair p; the two meet before pair p's own update of pair
// p+1's rows (schedule inside).
void launch_cholesky(hipStream_t s, hipStream_t side, std::vector<hipEvent_t> &events, double *A, int64_t lda,
                     int64_t n_pad, double *invDt, int *info_dev, const SweepPipe *pipe, bool info_zeroed)
{
    // > 64 KiB of dynamic LDS needs the opt-in on the current device (cheap; done per call so that several
    // devices in one process are all covered)
    hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_diag128_v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)sizeof(Diag2Shared));
    hipFuncSetAttribute(reinterpret_cast<const void *>(panel_trsm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)sizeof(PanelShared));
    hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_panel_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)sizeof(Diag2Shared));
    // 4: diagonal block + row panel in one launch; 2: separate launches (lean panel kernel, beside a pipelined sweep the
    // half-LDS strip kernel); 3: the lean panel kernel also beside a pipelined sweep; 1: the strip kernel as panel solver
    static const int panel_form_env = [] { const char *e = std::getenv("CBO_HIP_PANEL_FORM"); return e ? std::atoi(e) : 4; }();
    const int panel_form = g_panel_form_override > 0 ? g_panel_form_override : panel_form_env;
    // polls a strip of a fused launch makes before it gives up (read per factorisation: a test sets it to -1)
    const int spin_limit = [] { const char *e = std::getenv("CBO_HIP_FUSED_SPIN_LIMIT"); return e ? std::atoi(e) : kFusedSpinLimit; }();
    // the bulk trailing update takes the LDS-staged GEMM form (trsm_update_kernel<16>, one row block per workgroup) while
    // at least this many rows lie below the pair, the 64x64-tile SYRK from L2 fragments below that (round-2 scan)
    constexpr int syrk_gemm_rows = 6144, syrk_gemm_chunk = 1;
    constexpr bool syrk_gemm_half = true;
    auto launch_diag = [&](int rr) {
        hipLaunchKernelGGL(potrf_diag128_v2_kernel, dim3(1), dim3(256), sizeof(Diag2Shared), s, A, lda, rr, (int)n_pad,
                           invDt, info_dev, pipe ? pipe->zvec : nullptr);
    };
    const int np = (int)(n_pad / 128);
    // info_dev[0] is the status word; info_dev[1 + 2p], [2 + 2p] the publication counts of panel p's fused launch
    const bool fused = (panel_form == 4 || panel_form == 5) && 2 * np <= kCholFlagSlots;
    const bool split = panel_form == 5;
    // (a launch, not hipMemsetAsync: the runtime's fill costs two kernels and ~8 us of marker gaps around each)
    if (!info_zeroed) hipLaunchKernelGGL(zero_ints_kernel, dim3(1), dim3(256), 0, s, info_dev, fused ? 1 + 2 * np : 1);
    int *flags = info_dev + 1;
    const int rcol = (int)n_pad;
    while ((int)events.size() < 2 * np + 2) {
        hipEvent_t e;
        hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
        events.push_back(e);
    }
    // the side stream starts after everything queued on the main stream so far (K assembly, rhs)
    hipEventRecord(events[2 * np], s);
    hipStreamWaitEvent(side, events[2 * np], 0);
    // Panels are taken in pairs: the second panel of a pair only needs the first one's update of its own 128
    // rows, so the trailing matrix below the pair is updated once with K = 256 (half the read-modify-write
    // passes); the bulk of that update runs on the side stream under the next pair's diagonal/panel work.
    double *zvec = pipe ? pipe->zvec : nullptr;
    // beside a pipelined sweep the panel solves use the half-LDS kernel, which fits next to a sweep workgroup
    const bool half_lds = pipe && pipe->half_lds;
    const bool lean_panel = ((panel_form == 2 || panel_form == 4 || panel_form == 5) && !pipe) || panel_form == 3;
    int pair = 0;
    auto sweep_rows = [&](int r0, int klen) {
        if (pipe && r0 < pipe->tail_begin) sweep_pipe_pair(*pipe, s, A, lda, invDt, n_pad, pair++, r0, klen);
    };
    // Schedule per pair p (panels A_p, B_p):
    //   chain:  diag A_p, panel A_p, rows B_p -= A_p (K = 128), diag B_p, panel B_p,
    //           [wait bulk(p-1)]  rows A_{p+1}, B_{p+1} -= (A_p, B_p)  (K = 256, 256 rows)
    //   side :  [wait panel B_p]  bulk(p): everything below B_{p+1} -= (A_p, B_p)
    // bulk(p-1) rewrites the rows the K = 256 rows kernel of pair p rewrites, hence the wait; it has the whole chain of
    // pair p to finish, and bulk(p) follows it on the side stream without a gap -- large factorisations are bound by
    // the bulk updates alone, small ones by the chain alone.
    int pending = -1;                      // event index of the bulk update still in flight
    // Large trailing matrices, groups of G pairs (CBO_HIP_BULK_GROUP caps G: 1 = pairs only, 2, 4 = default).  The bulk update is
    // the LDS-staged GEMM kernel, whose rate grows with K (16384 points, all columns: 0.68 / 0.75 / 0.83 of peak at K = 256 /
    // 512 / 1024: scripts/update_kernel_timing.py), and it accumulates into C sequentially from C's own value, so one K = 256 G
    // pass gives the bits of G passes of K = 256.  Group j = pairs [G j, G j + G); S = 256 G; G_g = the S rows of group g:
    //   chain:  pair p of group j: factor; rows of pair p+1 -= pair p (rows kernel, as ever);
    //           [first pair of the group: wait bigA(j-1)]  near(p): the rows from pair p+2 to the end of G_{j+1} -= pair p (K = 256)
    //   side :  [wait the group's last pair]  bigA(j): G_{j+2} -= group j;  bigB(j): everything below G_{j+2} -= group j   (K = S)
    // -- every row still receives the pairs in order, each through the kernel that applied it before (rows kernel for
    // the pair right above, GEMM kernel otherwise): the factor is bit-identical to the pairs-only schedule
    // (scripts/probes/bulk_group_scan.sh: one digest for G = 1, 2, 4).  The chain only ever waits for bigA (S rows, first on
    // the side stream after the previous bigB), so bigB(j) has the whole chain of group j+1 to finish and bigB(j+1) follows it
    // without a gap.  The near updates (K = 256) are the price: they run on the chain stream beside the bulk update.  Groups
    // are used while every pair's bulk update would take the GEMM form anyway; the last ~6000 rows go pair by pair as before.
    // Groups of four while at least CBO_HIP_BULK_GROUP4_ROWS (10240) rows lie below the group, of two below that: round 5,
    // 16384 points 29.61 (G = 2) -> 29.07 ms on one box (thresholds 4096 / 6144 / 8192: 29.63 / 29.82 / 29.28).
    static const int bulk_group = [] { const char *e = std::getenv("CBO_HIP_BULK_GROUP"); return e ? std::atoi(e) : 4; }();
    static const int group4_rows = [] { const char *e = std::getenv("CBO_HIP_BULK_GROUP4_ROWS"); return e ? std::atoi(e) : 10240; }();
    constexpr bool group_split = true;     // inside groups the strips of a fused launch go as an LDS-free launch of their own
    int group_left = 0;                    // pairs of the open group still to come, this one included
    int group_pairs = 0, group_g0 = 0, group_first_k = 0;
    int pending_big_a = -1;                // event index of the last bigA
    for (int k = 0; k < np; k += 2) {
        const int r0 = 128 * k;
        const int n2 = (int)n_pad - r0 - 128;
        const int n3_pair = (int)n_pad - r0 - 256;                   // rows below this pair
        if (group_left == 0 && bulk_group >= 2 && (fused || lean_panel)) {
            // a group opens here if the rows below it still take the GEMM form for every pair of the group
            int G = 0;
            if (bulk_group >= 4 && n3_pair - 1024 >= group4_rows && n3_pair - 1024 >= syrk_gemm_rows && n3_pair - 1024 >= 1024) G = 4;
            else if (n3_pair - 512 >= syrk_gemm_rows && n3_pair - 512 >= 512) G = 2;
            if (G != 0) { group_pairs = G; group_left = G; group_g0 = r0; group_first_k = k; }
        }
        const bool grouped = group_left > 0;
        const bool first_of_group = grouped && group_left == group_pairs;
        const bool last_of_group = grouped && group_left == 1;
        // beside a bulk update that fills the device the strips go as an LDS-free launch of their own (launch_panel_fused)
        const bool split_now = split || (grouped && group_split && panel_form == 4);
        if (fused && n2 > 0) launch_panel_fused(s, A, lda, r0, rcol, invDt, info_dev, zvec, n2, flags + 2 * (r0 / 128), spin_limit, nullptr, split_now);
        else launch_diag(r0);
        if (n2 <= 0) { sweep_rows(r0, 128); break; }
        if (fused) {}
        else if (lean_panel) launch_panel_trsm(s, A, lda, r0, r0 + 128, n2, invDt, info_dev);   // (beside a pipelined sweep: the half-LDS strip kernel)
        else
        launch_trsm_strips(s, A + (int64_t)r0 * lda + r0, lda, invDt + (int64_t)(r0 / 16) * 256,
                           A + (int64_t)r0 * lda + r0 + 128, lda, 128, n2, nullptr, nullptr, nullptr, false, half_lds);
        // rows of the pair's second panel: K = 128 update with the first panel
        launch_syrk_rows(s, A, lda, r0, 128, n2, rcol, info_dev);
        const int r1 = r0 + 128;
        const int n3 = (int)n_pad - r1 - 128;
        const bool bulk = n3 > 256;
        const bool gemm_form = bulk && n3 - 256 >= syrk_gemm_rows;
        // the event the side stream waits for completes WITH the launch it follows (no marker packet on the chain)
        const bool carried = bulk && (fused || lean_panel);
        const hipEvent_t ev_panel = (carried && gemm_form && (!grouped || last_of_group)) ? events[2 * k] : nullptr;
        const hipEvent_t ev_rows = (carried && !gemm_form) ? events[2 * k] : nullptr;
        if (fused && n3 > 0)
            launch_panel_fused(s, A, lda, r1, rcol, invDt, info_dev, zvec, n3, flags + 2 * (r1 / 128), spin_limit, ev_panel, split_now);
        else launch_diag(r1);
        if (n3 <= 0) { sweep_rows(r0, 256); break; }
        if (fused) {}
        else if (lean_panel) launch_panel_trsm(s, A, lda, r1, r1 + 128, n3, invDt, info_dev, ev_panel);
        else
        launch_trsm_strips(s, A + (int64_t)r1 * lda + r1, lda, invDt + (int64_t)(r1 / 16) * 256,
                           A + (int64_t)r1 * lda + r1 + 128, lda, 128, n3, nullptr, nullptr, nullptr, false, half_lds);
        sweep_rows(r0, 256);
        if (grouped) {
            const int S = 256 * group_pairs, g0 = group_g0, n = (int)n_pad;
            const int next_end = (g0 + 2 * S < n) ? g0 + 2 * S : n;                   // end of G_{j+1}
            if (last_of_group) {
                // the group's bulk update (K = S) on the side stream, first the rows the chain needs next
                hipStreamWaitEvent(side, events[2 * k], 0);
                const int a_end = (g0 + 3 * S < n) ? g0 + 3 * S : n;                  // end of G_{j+2}
                launch_gemm_update(side, A, lda, A, lda, A, lda, g0, S, next_end, a_end, n_pad + kRhsCols, syrk_gemm_chunk,
                                   syrk_gemm_half, true, info_dev);
                hipEventRecord(events[2 * group_first_k + 1], side);                  // (the first pair's slot: it has no bulk update of its own)
                pending_big_a = 2 * group_first_k + 1;
                if (a_end < n)
                    launch_gemm_update(side, A, lda, A, lda, A, lda, g0, S, a_end, n, n_pad + kRhsCols, syrk_gemm_chunk,
                                       syrk_gemm_half, true, info_dev);
                hipEventRecord(events[2 * k + 1], side);
                pending = 2 * k + 1;
            }
            // the next pair's rows on the chain as ever, then the rest of the rows up to the end of the next group
            launch_syrk_rows(s, A, lda, r0, 256, n3, rcol, info_dev, 256, nullptr);
            if (first_of_group && pending_big_a >= 0) hipStreamWaitEvent(s, events[pending_big_a], 0);
            launch_gemm_update(s, A, lda, A, lda, A, lda, r0, 256, r0 + 512, next_end, n_pad + kRhsCols, syrk_gemm_chunk,
                               syrk_gemm_half, true, info_dev);
            --group_left;
            continue;
        }
        // both panels against everything below them: the bulk (below the next pair) on the side stream, the next
        // pair's own rows on this stream after the previous pair's bulk update of the same rows.  A large bulk update
        // bounds the factorisation, so it starts as soon as the panels are solved and the rows kernel runs beside its
        // first workgroups; a small one is hidden behind the chain anyway and would only slow the rows kernel (which is
        // ON the chain) down, so it starts after that.
        const int prev = pending;
        auto launch_bulk = [&] {
            if (!carried) hipEventRecord(events[2 * k], s);
            hipStreamWaitEvent(side, events[2 * k], 0);
            // Large trailing blocks go through the LDS-staged GEMM form of the sweep's update kernel (C -= P^T P on
            // 64-column strips x 256-row chunks, upper part only, the rhs strip as one more strip): the 64x64-tile SYRK
            // reads its operands as fragment-shaped loads from L2 and tops out near half the fp64 MFMA rate
            if (gemm_form)
                launch_gemm_update(side, A, lda, A, lda, A, lda, r0, 256, r0 + 256 + 256, (int)n_pad, n_pad + kRhsCols,
                                   syrk_gemm_chunk, syrk_gemm_half, true, info_dev);
            else
                launch_syrk(side, A, lda, r0, 256, n3, rcol, 4, n3 / 64, info_dev);
            hipEventRecord(events[2 * k + 1], side);
            pending = 2 * k + 1;
        };
        if (gemm_form) launch_bulk();
        if (prev >= 0) hipStreamWaitEvent(s, events[prev], 0);
        launch_syrk_rows(s, A, lda, r0, 256, n3, rcol, info_dev, 256, ev_rows);
        if (bulk && !gemm_form) launch_bulk();
    }
    // nothing is left on the side stream that the main stream has not waited for (the last bulk update is
    // awaited before the following panel's syrk); make that explicit for robustness
    hipEventRecord(events[2 * np + 1], side);
    hipStreamWaitEvent(s, events[2 * np + 1], 0);
    if (pipe && pipe->tail_begin < (int)n_pad) sweep_pipe_tail(*pipe, s, A, lda, invDt, n_pad, pair);
}

// ------------------------------------------------------------------------------------------------
// Backward solve U alpha = z, 128-row blocks from the bottom, one launch per block.  Every workgroup
// first solves the 128x128 diagonal system redundantly in LDS (16-row sub-blocks: multiply by the
// stored inv(U_bb), then fold into the rows above), then workgroup g folds alpha_blk into the 128 rows
// of block g above it:  zt[g] -= U[g, blk] alpha_blk.  zt is a contiguous working copy of z.
__global__ __launch_bounds__(256) void backsolve_step_kernel(const double *__restrict__ A, int64_t lda,
                                                             const double *__restrict__ invDt, int blk, double *zt,
                                                             double *__restrict__ alpha)
{
    __shared__ double zs[128];
    __shared__ double al[128];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int b0 = blk * 128;
    if (tid < 128) zs[tid] = zt[b0 + tid];
    __syncthreads();
    for (int s = 7; s >= 0; --s) {
        const int o = 16 * s;
        // alpha_s = inv(U_ss) zs_s : 16 outputs, thread (i, part) sums 4 of the 16 terms
        if (tid < 64) {
            const int i = tid >> 2, part = tid & 3;
            const double *Y = invDt + (int64_t)(b0 / 16 + s) * 256;      // Y[i][k] = inv(U_ss)[i][k]
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = fma(Y[i * 16 + 4 * part + k], zs[o + 4 * part + k], acc);
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            if (part == 0) al[o + i] = acc;
        }
        __syncthreads();
        // rows above inside the block: zs[r] -= sum_c U[b0+r][b0+o+c] al[o+c], r < o
        if (tid < o) {
            const double *row = A + (int64_t)(b0 + tid) * lda + b0 + o;
            double acc = zs[tid];
#pragma unroll
            for (int c = 0; c < 16; ++c) acc = fma(-row[c], al[o + c], acc);
            zs[tid] = acc;
        }
        __syncthreads();
    }
    const int g = blockIdx.x;
    if (g == blk) {
        if (tid < 128) alpha[b0 + tid] = al[tid];
        return;
    }
    // one row per wave iteration: 64 lanes x 2 columns, wave-level reduction
    for (int i = wave; i < 128; i += 4) {
        const d2 u = *reinterpret_cast<const d2 *>(&A[(int64_t)(g * 128 + i) * lda + b0 + 2 * lane]);
        double acc = fma(u[0], al[2 * lane], u[1] * al[2 * lane + 1]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) zt[g * 128 + i] -= acc;
    }
}

__global__ void copy_strided_kernel(const double *__restrict__ src, int64_t stride, int64_t n, double *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i * stride];
}

// The caller provides alpha with 2*n_pad doubles: [0, n_pad) result, [n_pad, 2 n_pad) working copy of z.
void launch_backsolve(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt, double *alpha)
{
    launch_backsolve_vec(s, A, lda, n_pad, invDt, A + n_pad, lda, alpha + n_pad, alpha);
}

// out = U^-1 src for one strided vector (src is left untouched; work is an n_pad scratch vector).
void launch_backsolve_vec(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt,
                          const double *src, int64_t src_stride, double *work, double *out)
{
    hipLaunchKernelGGL(copy_strided_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, s, src, src_stride,
                       n_pad, work);
    const int nb = (int)(n_pad / 128);
    for (int blk = nb - 1; blk >= 0; --blk)
        hipLaunchKernelGGL(backsolve_step_kernel, dim3(blk + 1), dim3(256), 0, s, A, lda, invDt, blk, work, out);
}

// ------------------------------------------------------------------------------------------------
// Forward solve U^T l = k (l = L^-1 k) for ONE right-hand side, 128-row blocks from the top, one launch per block:
// the mirror image of backsolve_step_kernel.  Every workgroup first solves the 128x128 diagonal system redundantly
// in LDS (16-row sub-blocks: multiply by the transposed stored inv(U_ss), then fold into the rows below), then
// workgroup g folds l_blk into the 128 entries of block blk + 1 + g:  w[i] -= sum_k U[blk rows k][i] l_k -- a row of
// U is contiguous in i, so the 128 threads of the update read coalesced.  The strip kernel needs 64 columns and a
// workgroup per strip; for a single column this is what fills the device.
__global__ __launch_bounds__(256) void forward_step_kernel(const double *__restrict__ A, int64_t lda,
                                                           const double *__restrict__ invDt, int blk, int nb, double *w,
                                                           double *__restrict__ out)
{
    __shared__ double rs[128];
    __shared__ double ls[128];
    const int tid = threadIdx.x;
    const int b0 = blk * 128;
    if (tid < 128) rs[tid] = w[b0 + tid];
    __syncthreads();
    for (int s = 0; s < 8; ++s) {
        const int o = 16 * s;
        // l_s = inv(U_ss)^T r_s : thread (i, part) sums 4 of the 16 terms; Y[k][i] = inv(U_ss)[k][i]
        if (tid < 64) {
            const int i = tid >> 2, part = tid & 3;
            const double *Y = invDt + (int64_t)(b0 / 16 + s) * 256;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = fma(Y[(4 * part + k) * 16 + i], rs[o + 4 * part + k], acc);
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            if (part == 0) ls[o + i] = acc;
        }
        __syncthreads();
        // rows below inside the block: rs[c] -= sum_k U[b0+o+k][b0+c] l_{o+k}, c >= o + 16
        if (tid >= o + 16 && tid < 128) {
            double acc = rs[tid];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = fma(-A[(int64_t)(b0 + o + k) * lda + b0 + tid], ls[o + k], acc);
            rs[tid] = acc;
        }
        __syncthreads();
    }
    const int g = blockIdx.x;
    if (g == 0 && tid < 128) out[b0 + tid] = ls[tid];
    // block blk + 1 + g: two threads per entry, each over 64 of the 128 panel rows
    if (blk + 1 + g >= nb) return;                             // the last block has nothing below it (uniform)
    const int c = tid & 127, half = tid >> 7;
    const int64_t col = (int64_t)(blk + 1 + g) * 128 + c;
    double acc = 0.0;
    for (int k = 64 * half; k < 64 * half + 64; ++k) acc = fma(A[(int64_t)(b0 + k) * lda + col], ls[k], acc);
    __syncthreads();
    if (half == 1) rs[c] = acc;
    __syncthreads();
    if (half == 0) w[col] -= acc + rs[c];
}

// out = L^-1 w for one contiguous vector of n_pad entries (w is used as the work vector and destroyed).
void launch_forward_vec(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt, double *w,
                        double *out)
{
    const int nb = (int)(n_pad / 128);
    for (int blk = 0; blk < nb; ++blk) {
        const int below = nb - 1 - blk;                       // blocks that receive this block's contribution
        hipLaunchKernelGGL(forward_step_kernel, dim3(below > 0 ? below : 1), dim3(256), 0, s, A, lda, invDt, blk, nb, w, out);
    }
}

// ------------------------------------------------------------------------------------------------
// The two single-vector solves as ONE launch each: a chain of workgroups, one per 128-row block.
//
// The per-block launches above cost a launch plus a cold diagonal solve per block (42 us: 5.4 ms for alpha at 16384
// points).  Here workgroup w owns block g and, in the backward solve, folds alpha_blk into its 128 entries for every
// blk > g as those become available (the solved block itself is the signal, see chain_wait), then solves its diagonal
// block and publishes alpha_g.  What the chain waits for per block
// is: the arrival of alpha_{g+1}, one 128 x 128 fold, the diagonal solve, the publication.  Everything that does not
// depend on alpha is in place before it arrives: the diagonal block and its 16 x 16 inverses sit in LDS, the block of U
// for the next fold sits in registers (requested while the previous hand-over is awaited).
// Same operations in the same order as the per-block kernels (each fold is one subtraction of the same wave-reduced
// sum, blocks in the same order; the in-block solve is the same code on LDS copies): same bits.
// Forward progress: block g waits only for blocks solved earlier, and the workgroup index grows in solve order, so under
// in-order dispatch every producer is resident before its consumers (no need for all workgroups to be co-resident).
// The polls are bounded; a give-up lands in the status word and the host repeats the solve with the per-block launches.
constexpr int kVecLd = 129;                      // LDS row stride of the diagonal block: column reads by 128 threads

struct VecChainShared {
    double Ud[128][kVecLd];                      // U[g, g] (row-major)
    double inv[8][256];                          // inverses of its 16 x 16 diagonal tiles
    double zs[128], al[128], ab[128];
    int abort_flag;
};

// The solved block IS the signal: the output vector is pre-filled with a sentinel (a NaN payload no computation
// produces; a computed NaN is stored as the canonical quiet NaN), the producer stores its 128 values as agent-scope
// atomics, and each of the consumer's first 128 threads polls ITS value until it is no longer the sentinel -- one
// round trip to the coherence point per hand-over instead of two (count, then data), and no fence: every element is its
// own atomic object and nothing else is communicated through it.
constexpr unsigned long long kVecSentinel = 0x7ff8dead0000beefull;

__device__ __forceinline__ bool chain_wait(const double *src, int *info, int spin_limit, VecChainShared &sh)
{
    int ok = 1;
    if (threadIdx.x < 128) {
        int spins = 0;
        double v;
        for (;;) {
            v = __hip_atomic_load(src + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned long long)__double_as_longlong(v) != kVecSentinel) break;
            if (spin_limit < 0 || ++spins > spin_limit || ((spins & 63) == 0 && __builtin_nontemporal_load(info) != 0)) { ok = 0; break; }
        }
        sh.ab[threadIdx.x] = v;
    }
    if (!__syncthreads_and(ok)) {
        if (threadIdx.x == 0) atomicCAS(info, 0, kFusedTimeout);
        return false;
    }
    return true;
}

__device__ __forceinline__ void chain_publish(double *dst, const double *vals)
{
    if (threadIdx.x < 128) {
        double v = vals[threadIdx.x];
        if (v != v) v = __longlong_as_double(0x7ff8000000000000ll);
        __hip_atomic_store(dst + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void fill_sentinel_kernel(double *p, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = __longlong_as_double((long long)kVecSentinel);
}

__global__ __launch_bounds__(256) void backsolve_chain_kernel(const double *__restrict__ A, int64_t lda,
                                                              const double *__restrict__ invDt, int nb,
                                                              const double *__restrict__ zt, double *alpha, int *info,
                                                              int spin_limit)
{
    extern __shared__ __align__(16) unsigned char vec_smem[];
    VecChainShared &sh = *reinterpret_cast<VecChainShared *>(vec_smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = nb - 1 - (int)blockIdx.x;                       // workgroup 0 owns the last block: solved first
    const int b0 = g * 128;
    if (tid < 128) sh.zs[tid] = zt[b0 + tid];
    for (int i = tid; i < 128 * 128; i += 256) sh.Ud[i >> 7][i & 127] = A[(int64_t)(b0 + (i >> 7)) * lda + b0 + (i & 127)];
    for (int i = tid; i < 8 * 256; i += 256) sh.inv[i >> 8][i & 255] = invDt[(int64_t)(b0 / 16) * 256 + i];
    // The block of the NEXT fold is always in registers before its alpha arrives: U[g, blk] as wave w's rows w, w+4, ...,
    // two columns per lane, requested while the previous hand-over is awaited (32 loads in flight, not 32 round trips).
    d2 un[32];
    auto request_block = [&](int blk) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 32; ++r)
            un[r] = *reinterpret_cast<const d2 *>(&A[(int64_t)(b0 + wave + 4 * r) * lda + blk * 128 + 2 * lane]);
    };
    if (g + 1 < nb) request_block(nb - 1);
    __syncthreads();
    for (int blk = nb - 1; blk > g; --blk) {
        if (!chain_wait(alpha + blk * 128, info, spin_limit, sh)) return;      // alpha_blk is in sh.ab
        const double a0 = sh.ab[2 * lane], a1 = sh.ab[2 * lane + 1];
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            double acc = fma(un[r][0], a0, un[r][1] * a1);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
            if (lane == 0) sh.zs[wave + 4 * r] -= acc;
        }
        if (blk - 1 > g) request_block(blk - 1);
        __syncthreads();
    }
    // the diagonal system, bottom tile first (backsolve_step_kernel's arithmetic on the LDS copies)
    for (int s = 7; s >= 0; --s) {
        const int o = 16 * s;
        if (tid < 64) {
            const int i = tid >> 2, part = tid & 3;
            const double *Y = sh.inv[s];
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = fma(Y[i * 16 + 4 * part + k], sh.zs[o + 4 * part + k], acc);
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            if (part == 0) sh.al[o + i] = acc;
        }
        __syncthreads();
        if (tid < o) {
            double acc = sh.zs[tid];
#pragma unroll
            for (int c = 0; c < 16; ++c) acc = fma(-sh.Ud[tid][o + c], sh.al[o + c], acc);
            sh.zs[tid] = acc;
        }
        __syncthreads();
    }
    chain_publish(alpha + b0, sh.al);
}

__global__ __launch_bounds__(256) void forward_chain_kernel(const double *__restrict__ A, int64_t lda,
                                                            const double *__restrict__ invDt, int nb,
                                                            const double *__restrict__ w, double *out, int *info,
                                                            int spin_limit)
{
    extern __shared__ __align__(16) unsigned char vec_smem[];
    VecChainShared &sh = *reinterpret_cast<VecChainShared *>(vec_smem);
    const int tid = threadIdx.x;
    const int g = (int)blockIdx.x;                                // solved in index order
    const int b0 = g * 128;
    const int c = tid & 127, half = tid >> 7;
    if (tid < 128) sh.zs[tid] = w[b0 + tid];
    for (int i = tid; i < 128 * 128; i += 256) sh.Ud[i >> 7][i & 127] = A[(int64_t)(b0 + (i >> 7)) * lda + b0 + (i & 127)];
    for (int i = tid; i < 8 * 256; i += 256) sh.inv[i >> 8][i & 255] = invDt[(int64_t)(b0 / 16) * 256 + i];
    // the block of the NEXT fold, U[blk rows, g columns], in registers before its l_blk arrives: this thread's column,
    // its half of the 128 rows
    double un[64];
    auto request_block = [&](int blk) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 64; ++k) un[k] = A[(int64_t)(blk * 128 + 64 * half + k) * lda + b0 + c];
    };
    if (g > 0) request_block(0);
    __syncthreads();
    for (int blk = 0; blk < g; ++blk) {
        if (!chain_wait(out + blk * 128, info, spin_limit, sh)) return;        // l_blk is in sh.ab
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 64; ++k) acc = fma(un[k], sh.ab[64 * half + k], acc);
        if (blk + 1 < g) request_block(blk + 1);
        if (half == 1) sh.al[c] = acc;                            // (al is free until the diagonal solve)
        __syncthreads();
        if (half == 0) sh.zs[c] -= acc + sh.al[c];
        __syncthreads();
    }
    // the diagonal system, top tile first (forward_step_kernel's arithmetic on the LDS copies)
    for (int s = 0; s < 8; ++s) {
        const int o = 16 * s;
        if (tid < 64) {
            const int i = tid >> 2, part = tid & 3;
            const double *Y = sh.inv[s];
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = fma(Y[(4 * part + k) * 16 + i], sh.zs[o + 4 * part + k], acc);
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            if (part == 0) sh.al[o + i] = acc;
        }
        __syncthreads();
        if (tid >= o + 16 && tid < 128) {
            double acc = sh.zs[tid];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = fma(-sh.Ud[o + k][tid], sh.al[o + k], acc);
            sh.zs[tid] = acc;
        }
        __syncthreads();
    }
    chain_publish(out + b0, sh.al);
}

// CBO_HIP_VEC_SOLVE_FORM=1: the per-block launches.  `info` is the model's status word (0 after a fit).
static bool vec_chain_enabled(int nb)
{
    static const int form = [] { const char *e = std::getenv("CBO_HIP_VEC_SOLVE_FORM"); return e ? std::atoi(e) : 2; }();
    return form != 1 && nb >= 2;
}
static int vec_spin_limit()
{
    const char *e = std::getenv("CBO_HIP_FUSED_SPIN_LIMIT");
    return e ? std::atoi(e) : kFusedSpinLimit;
}

// ~151 KB of dynamic LDS needs the opt-in ON THE CURRENT DEVICE: done per call (cheap), as launch_cholesky does, so that
// every device of a process that drives several (cbo_comm_init_all) is covered -- a function-local static would cover
// only the device that was current at the first call.  false: the runtime refused, take the per-block launches.
static bool vec_chain_opt_in()
{
    const hipError_t a = hipFuncSetAttribute(reinterpret_cast<const void *>(backsolve_chain_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(VecChainShared));
    const hipError_t b = hipFuncSetAttribute(reinterpret_cast<const void *>(forward_chain_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(VecChainShared));
    if (a == hipSuccess && b == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}

bool launch_backsolve_chain(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt,
                            const double *src, int64_t src_stride, double *work, double *out, int *info)
{
    const int nb = (int)(n_pad / 128);
    if (!vec_chain_enabled(nb)) return false;
    if (!vec_chain_opt_in()) return false;
    hipLaunchKernelGGL(copy_strided_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, s, src, src_stride,
                       n_pad, work);
    hipLaunchKernelGGL(fill_sentinel_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, s, out, n_pad);
    hipLaunchKernelGGL(backsolve_chain_kernel, dim3(nb), dim3(256), sizeof(VecChainShared), s, A, lda, invDt, nb, work, out,
                       info, vec_spin_limit());
    // a launch the runtime refused (nothing ran): the caller takes the per-block launches, which fill `out` themselves
    return hipGetLastError() == hipSuccess;
}

bool launch_forward_chain(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt, const double *w,
                          double *out, int *info)
{
    const int nb = (int)(n_pad / 128);
    if (!vec_chain_enabled(nb)) return false;
    if (!vec_chain_opt_in()) return false;
    hipLaunchKernelGGL(fill_sentinel_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, s, out, n_pad);
    hipLaunchKernelGGL(forward_chain_kernel, dim3(nb), dim3(256), sizeof(VecChainShared), s, A, lda, invDt, nb, w, out,
                       info, vec_spin_limit());
    return hipGetLastError() == hipSuccess;          // (forward_chain_kernel leaves `w` untouched: the fallback can still use it)
}

// dst[i] = V[i * ldv] for i < n_pad: one column of a row-major workspace as a contiguous vector
void launch_gather_column(hipStream_t s, const double *V, int64_t ldv, int64_t n_pad, double *dst)
{
    hipLaunchKernelGGL(copy_strided_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, s, V, ldv, n_pad, dst);
}

}  // namespace cbo
