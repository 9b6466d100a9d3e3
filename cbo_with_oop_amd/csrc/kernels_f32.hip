// fp32 acquisition sweep for gfx950 (BASELINE.json configs[4]: "fp32 path with MFMA"): the substitution
// V <- L^-1 K* on v_mfma_f32_16x16x4_f32 (exact f32 FMA chains, 64 FLOP/clk/SIMD = 157 TFLOP/s dense peak, twice the
// fp64 matrix rate).  The FIT stays fp64 (Ky carries GPy's 1e-8 diagonal jitter next to O(1) entries, which fp32
// cannot represent): after a fit the factor U, the 16x16 diagonal inverses and z are down-converted ONCE into the
// layout below; K(X, X*) is evaluated in fp64 (the coordinates of the coral ranges, T in [2450, 2500], lose the
// pairwise differences in fp32) and rounded to fp32 on store; q = sum V^2 is accumulated in fp64 from the fp32 V.
// The posterior mean does NOT go through the fp32 substitution: mu = V^T z cancels catastrophically when Ky is
// ill-conditioned (both factors carry 1/pivot-sized components), so the fp32 path forms it as GPy itself does,
// mu = K*^T alpha, in fp64 inside the K* assembly kernel (kernels_kmat.hip, OUT32), from alpha = L^-T z.
//
// Same decomposition as trsm_strip_kernel (kernels_trsm.hip): 256-thread workgroups, 64 candidate columns per
// strip, wave w owns 16 columns for all rows, one continuous 3-deep LDS-DMA pipeline of 32-row stages.  What fp32
// changes:
//  * a stage row of U is 1 KiB = 256 floats, so a row block is 256 rows = 16 MFMA tiles (the byte geometry of the
//    fp64 kernel: 64 accumulator registers per lane, 4096 MFMA cycles per stage -- 8 k-steps x 16 tiles x 32 cycles);
//  * the f32 result map is  row = 4 (lane>>4) + reg,  not  (lane>>4) + 4 reg  as for f64, so a result register is
//    NOT the B operand of "k-step reg" in natural row order.  MFMA sums over k in any order as long as both
//    operands agree, so the ROWS of every 16-row group of the fp32 copies (factor, diagonal inverses, K*, V) are
//    stored permuted, physical row 4 (k & 3) + (k >> 2) holding logical row k (an involution): with that storage
//    order every address this kernel computes is the one the fp64 kernel computes, results feed the next MFMA in
//    place, and the LDS reads stay bank-conflict free.  q is a sum over rows: the order does not matter.
//
// Roofline: fp32 MFMA bound.  n32^2 flops per candidate column (n32 = rows padded to 256); V traffic
// n32^2/(2*256)*4 B re-read + 2*n32*4 B per column, U n32^2/2*4 B per strip from L2/MALL.
#include <type_traits>

#include "cbo_internal.h"

namespace cbo {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#ifndef F32_DBG
#define F32_DBG 0          // timing-only builds (scripts/f32_variants.sh): bit mask of pieces to leave out; results are wrong
#endif
int f32_debug_mask() { return F32_DBG; }       // a non-zero mask changes cbo_abi_version(): _lib.load() refuses the build
#if (F32_DBG & 1)
#define SCHED_DS(n)
#define SCHED_MFMA(n)
#define SCHED_VMEM(n)
#else
#define SCHED_DS(n) __builtin_amdgcn_sched_group_barrier(0x100, (n), 0)
#define SCHED_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, (n), 0)
#define SCHED_VMEM(n) __builtin_amdgcn_sched_group_barrier(0x010, (n), 0)
#endif

constexpr int kRB = 256;                  // rows per block
constexpr int kT = kRB / 16;              // 16-row tiles per block
constexpr int kLd = kRB + 16;             // U-tile row stride (floats): rows kq and kq+1 land 16 banks apart (ds_read_b32)
constexpr int kKB = 32;                   // rows of U / V per pipeline stage
constexpr int kNBuf = 3;
constexpr int kABuf = kKB * kLd;          // floats per U stage buffer
constexpr int kBBuf = 4 * kKB * 16;       // floats per V stage buffer (4 waves x [32 k][16 cols])
constexpr int kRA = kKB / 4;              // U rows a wave fetches per stage (one 1 KiB DMA each)
constexpr int kBPieces = kKB * 16 * 4 / 1024;   // 1 KiB B pieces per stage and wave (16 rows x 16 cols each)
constexpr int kParts = 4;                 // DMA groups per stage and wave: 2 U rows (+ one B piece in the first kBPieces)
constexpr int kDma = kRA + kBPieces;      // LDS-DMA instructions a wave issues per stage (U rows, B pieces)
constexpr int kKS = kKB / 4;              // MFMA k-steps per stage
constexpr int kDS = kRB / kKB;            // diagonal stages per row block
constexpr int kDT = kKB / 16;             // 16x16 diagonal tiles solved per diagonal stage
constexpr int kDiagStores = 4 * kDT;
static_assert(kParts * 2 == kRA && kBPieces <= kParts && kParts <= kKS - 2, "DMA grouping");

__device__ __forceinline__ void glds16f(const float *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

__device__ __host__ __forceinline__ int perm16(int k) { return 4 * (k & 3) + (k >> 2); }

struct StageCursor32 {
    int i0, j, lim;
    const float *a_src, *b_src;
    int64_t b_stride;
};

__global__ __launch_bounds__(256) void trsm_strip_f32_kernel(const float *__restrict__ U, int64_t ldu,
                                                             const float *__restrict__ invDt, float *V, int64_t ldv,
                                                             int n, double *__restrict__ q_out)
{
    __shared__ __align__(16) float lds[kNBuf * (kABuf + kBBuf)];       // 129,024 B

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, kq = lane >> 4;
    const int64_t colw = (int64_t)blockIdx.x * kStrip + wave * 16;
    float *Vc = V + colw + lc;
    float *ldsB = lds + kNBuf * kABuf;
    const unsigned lds_byte0 = lds_byte_address(lds);
    const float *ug = U + (int64_t)(wave * kRA) * ldu + lane * 4;                    // one 1 KiB row per instruction
    const float *vg = V + (int64_t)(lane >> 2) * ldv + colw + 4 * (lane & 3);        // 16 rows x 64 B per instruction
    const float *inv_lane = invDt + lane * 4;

    auto locate = [&](StageCursor32 &c) __attribute__((always_inline)) {
        const bool past = c.i0 >= n;
        const int ai0 = past ? n - kRB : c.i0;
        const int aj = past ? (n - kRB) / kKB + kDS - 1 : c.j;
        const int nreg = ai0 / kKB;
        c.a_src = ug + (int64_t)(kKB * aj) * ldu + ai0;
        const int64_t diag = (aj >= nreg) ? 1 : 0;
        const int64_t off_diag = ((int64_t)(ai0 / 16) + kDT * (aj - nreg)) * 256;
        const int64_t off_reg = (int64_t)(kKB * aj) * ldv;
        const uintptr_t base = (uintptr_t)vg + ((uintptr_t)inv_lane - (uintptr_t)vg) * (uintptr_t)diag;
        c.b_src = reinterpret_cast<const float *>(base) + (off_reg + (off_diag - off_reg) * diag);
        c.b_stride = 16 * ldv + (256 - 16 * ldv) * diag;
    };
    auto advance = [&](StageCursor32 &c) __attribute__((always_inline)) {
        const int wrap = (c.j + 1 == c.lim) ? 1 : 0;
        c.i0 += kRB * wrap;
        c.j = (c.j + 1) * (1 - wrap);
        c.lim = c.lim + (c.i0 / kKB + kDS - c.lim) * wrap;
        locate(c);
    };
    auto issue_part = [&](const StageCursor32 &c, int buf, int part) __attribute__((always_inline)) {
        const unsigned la = __builtin_amdgcn_readfirstlane(lds_byte0 + 4u * (unsigned)(buf * kABuf + (wave * kRA) * kLd));
        const unsigned lb = __builtin_amdgcn_readfirstlane(
            lds_byte0 + 4u * (unsigned)(kNBuf * kABuf + buf * kBBuf + wave * (kKB * 16)));
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = 2 * part + q;
            glds16f(c.a_src + (int64_t)p * ldu, la + 4u * (unsigned)(p * kLd));
        }
        if (part < kBPieces) glds16f(c.b_src + part * c.b_stride, lb + 4u * (unsigned)(part * 256));
    };
    auto issue_stage = [&](const StageCursor32 &c, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int part = 0; part < kParts; ++part) issue_part(c, buf, part);
    };

    // acc holds the NEGATED residual (no operand negation in the K-loop); register r of tile t = physical row
    // 16 t + kq + 4 r (= logical row 16 t + 4 kq + r, the f32 result map)
    f4 acc[kT], accn[kT];
#pragma unroll
    for (int t = 0; t < kT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = -Vc[(int64_t)(16 * t + kq + 4 * r) * ldv];

    StageCursor32 ahead{0, 0, kDS, nullptr, nullptr, 0};
    locate(ahead);
    issue_stage(ahead, 0);
    advance(ahead);
    issue_stage(ahead, 1);
    advance(ahead);
    int buf = 0;
    int extra_prev = 0;
    double qacc = 0.0;

#define STAGE_TOP()                                                                                       \
    do {                                                                                                  \
        if (extra_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma + kDiagStores) : "memory");         \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma) : "memory");                                  \
        __builtin_amdgcn_s_barrier();                                                                     \
    } while (0)

    for (int i0 = 0; i0 < n; i0 += kRB) {
        const int nst = i0 / kKB;
        float af[2][kT], bf[2];
        bool deferred = false;
        for (int j = 0; j < nst; ++j) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            STAGE_TOP();
            __builtin_amdgcn_sched_barrier(0);
            const int bnext = (buf >= 1) ? buf - 1 : 2;
            extra_prev = 0;
            const float *abase = lds + buf * kABuf + kq * kLd + lc;
            const float *bbase = ldsB + buf * kBBuf + wave * (kKB * 16) + kq * 16 + lc;
#pragma unroll
            for (int t = 0; t < kT; ++t) af[0][t] = abase[16 * t];
            bf[0] = bbase[0];
            if (deferred) {
#pragma unroll
                for (int t = 0; t < kT; ++t) acc[t] = MFMA_F32(af[1][t], bf[1], acc[t]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < kKS - 1; ++jj) {
#pragma unroll
                for (int t = 0; t < kT; ++t) af[(jj + 1) & 1][t] = abase[4 * (jj + 1) * kLd + 16 * t];
                bf[(jj + 1) & 1] = bbase[4 * (jj + 1) * 16];
                if (!(F32_DBG & 2) && jj < kParts) issue_part(ahead, bnext, jj);
                if (jj == kParts) advance(ahead);
                if (!(F32_DBG & 4)) {
#pragma unroll
                    for (int t = 0; t < kT; ++t) acc[t] = MFMA_F32(af[jj & 1][t], bf[jj & 1], acc[t]);
                } else {
#pragma unroll
                    for (int t = 0; t < kT; ++t) asm volatile("" ::"v"(af[jj & 1][t]), "v"(bf[jj & 1]));
                }
                SCHED_DS(kT + 1);
                if (jj < kBPieces) { SCHED_VMEM(3); } else if (jj < kParts) { SCHED_VMEM(2); }
                SCHED_MFMA(kT);
            }
            deferred = true;
            buf = (buf == 2) ? 0 : buf + 1;
        }
        if (deferred) {
#pragma unroll
            for (int t = 0; t < kT; ++t) acc[t] = MFMA_F32(af[1][t], bf[1], acc[t]);
        }

        // ---- diagonal stages: X_s = inv(L_ss) R_s, then R_t -= L_ts X_s for the tiles below.  The stage index is a
        //      compile-time constant (generic lambda + integral_constant): every register array keeps static indices
        auto diag_stage = [&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value;
            STAGE_TOP();
            if (!(F32_DBG & 16) && m == 0 && i0 + kRB < n) {
                // next block's right-hand sides (K* rows), raw: negated when the block starts, so that no wait for
                // them lands inside the pipeline
#pragma unroll
                for (int t = 0; t < kT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) accn[t][r] = Vc[(int64_t)(i0 + kRB + 16 * t + kq + 4 * r) * ldv];
            }
            asm volatile("" ::: "memory");
            const int bnext = (buf >= 1) ? buf - 1 : 2;
            issue_stage(ahead, bnext);
            advance(ahead);
            const float *abase = lds + buf * kABuf + kq * kLd + lc;
            const float *ibase = ldsB + buf * kBBuf + wave * (kKB * 16) + kq * 16 + lc;
            float iv[kDT][4], uf[kDT][kT][4];
#pragma unroll
            for (int h = 0; h < kDT; ++h) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) iv[h][kk] = ibase[h * 256 + 64 * kk];
#pragma unroll
                for (int t = kDT * m + h + 1; t < kT; ++t)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) uf[h][t][kk] = abase[(16 * h + 4 * kk) * kLd + 16 * t];
            }
            asm volatile("" ::: "memory");
            constexpr int s = kDT * m;
            static_assert(kDT == 2, "the tile-to-tile interleave below is written for two tiles per stage");
            // invF holds the NEGATED inverses: x = (-inv) * (negated residual), no VALU negation inside the chain
            auto solve_tile = [&](int h, const f4 &rneg) __attribute__((always_inline)) -> f4 {
                f4 x = {0.f, 0.f, 0.f, 0.f}, x2 = {0.f, 0.f, 0.f, 0.f};
                x = MFMA_F32(iv[h][0], rneg[0], x);
                x2 = MFMA_F32(iv[h][1], rneg[1], x2);
                x = MFMA_F32(iv[h][2], rneg[2], x);
                x2 = MFMA_F32(iv[h][3], rneg[3], x2);
                return x + x2;
            };
            auto emit_tile = [&](int tile, const f4 &x) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Vc[(int64_t)(i0 + 16 * tile + kq + 4 * r) * ldv] = x[r];
                    const double xd = (double)x[r];
                    qacc = fma(xd, xd, qacc);
                }
            };
            if (!(F32_DBG & 8)) {
                const f4 x = solve_tile(0, acc[s]);
#if (F32_DBG & 32)
                emit_tile(s, x);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int t = s + 1; t < kT; ++t) acc[t] = MFMA_F32(uf[0][t][kk], x[kk], acc[t]);
                const f4 y = solve_tile(1, acc[s + 1]);
                emit_tile(s + 1, y);
#else
                // tile s+1 first (its four updates, interleaved with tile s+2's so that no MFMA waits on its
                // predecessor), then its solve chain with the rest of tile s's updates in between
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    acc[s + 1] = MFMA_F32(uf[0][s + 1][kk], x[kk], acc[s + 1]);
                    if constexpr (s + 2 < kT) acc[s + 2] = MFMA_F32(uf[0][s + 2][kk], x[kk], acc[s + 2]);
                }
                emit_tile(s, x);
                f4 y1 = {0.f, 0.f, 0.f, 0.f}, y2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (kk & 1) y2 = MFMA_F32(iv[1][kk], acc[s + 1][kk], y2);
                    else y1 = MFMA_F32(iv[1][kk], acc[s + 1][kk], y1);
#pragma unroll
                    for (int t = s + 3; t < kT; ++t) acc[t] = MFMA_F32(uf[0][t][kk], x[kk], acc[t]);
                }
                const f4 y = y1 + y2;
                emit_tile(s + 1, y);
#endif
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int t = s + 2; t < kT; ++t) acc[t] = MFMA_F32(uf[1][t][kk], y[kk], acc[t]);
            }
            extra_prev = 1;
            buf = (buf == 2) ? 0 : buf + 1;
        };
        diag_stage(std::integral_constant<int, 0>{});
        diag_stage(std::integral_constant<int, 1>{});
        diag_stage(std::integral_constant<int, 2>{});
        diag_stage(std::integral_constant<int, 3>{});
        diag_stage(std::integral_constant<int, 4>{});
        diag_stage(std::integral_constant<int, 5>{});
        diag_stage(std::integral_constant<int, 6>{});
        diag_stage(std::integral_constant<int, 7>{});
        static_assert(kDS == 8, "eight diagonal stages per 256-row block");
#pragma unroll
        for (int t = 0; t < kT; ++t) acc[t] = -accn[t];
    }
#undef STAGE_TOP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    qacc += __shfl_xor(qacc, 16);
    qacc += __shfl_xor(qacc, 32);
    if (kq == 0) q_out[colw + lc] = qacc;
}

void launch_trsm_strips_f32(hipStream_t s, const float *U, int64_t ldu, const float *invDt, float *V, int64_t ldv,
                            int64_t n32, int64_t m_pad, double *q)
{
    if (n32 <= 0 || m_pad <= 0) return;
    hipLaunchKernelGGL(trsm_strip_f32_kernel, dim3((unsigned)(m_pad / kStrip)), dim3(256), 0, s, U, ldu, invDt, V, ldv,
                       (int)n32, q);
}

// ------------------------------------------------------------------------------------------------
// fp64 factor -> fp32 copies in the row-permuted layout the kernel above reads.  Uf[n32][ldu]: physical row
// 16 g + perm16(k) = logical row 16 g + k of U (upper triangle; zeros below the diagonal; identity beyond n_pad).
// One thread per 4 consecutive columns (16-byte stores).
__global__ __launch_bounds__(256) void factor_to_f32_kernel(const double *__restrict__ A, int64_t lda, int64_t n_pad,
                                                            float *__restrict__ Uf, int64_t ldu, int64_t n32)
{
    const int64_t pr = blockIdx.y;
    const int64_t c4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c4 >= n32) return;
    const int64_t lr = (pr & ~(int64_t)15) + perm16((int)(pr & 15));
    f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t c = c4 + e;
        float v;
        if (lr < n_pad && c < n_pad) v = (c >= lr) ? (float)A[lr * lda + c] : 0.f;
        else v = (lr == c) ? 1.f : 0.f;
        o[e] = v;
    }
    *reinterpret_cast<f4 *>(&Uf[pr * ldu + c4]) = o;
}

// diagonal inverses, negated, rows permuted inside each 16x16 tile (identity beyond n_pad)
__global__ __launch_bounds__(256) void inverses_to_f32_kernel(const double *__restrict__ invDt, int64_t n_pad,
                                                              float *__restrict__ invF, int64_t n32)
{
    const int64_t tile = blockIdx.x;
    const int t = threadIdx.x;                     // physical element (row pk, column i) of the tile
    const int pk = t >> 4, i = t & 15;
    const int k = perm16(pk);
    const int64_t row = tile * 16 + k;
    invF[tile * 256 + t] = (row < n_pad) ? -(float)invDt[tile * 256 + k * 16 + i] : ((k == i) ? -1.f : 0.f);   // NEGATED
}

void launch_factor_to_f32(hipStream_t s, const double *A, int64_t lda, int64_t n_pad, const double *invDt, float *Uf,
                          int64_t ldu, float *invF, int64_t n32)
{
    hipLaunchKernelGGL(factor_to_f32_kernel, dim3((unsigned)((n32 / 4 + 255) / 256), (unsigned)n32), dim3(256), 0, s, A,
                       lda, n_pad, Uf, ldu, n32);
    hipLaunchKernelGGL(inverses_to_f32_kernel, dim3((unsigned)(n32 / 16)), dim3(256), 0, s, invDt, n_pad, invF, n32);
}

// test hook: MFMA f32 lane map (A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15], D row 4 (lane>>4) + reg),
// including the result-as-next-B-operand use with the permuted k order this file relies on
__global__ void mfma_f32_selftest_kernel(const float *A, const float *B, float *C)
{
    const int l = threadIdx.x;
    const int lc = l & 15, kq = l >> 4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < 16; k0 += 4) acc = MFMA_F32(A[lc * 16 + k0 + kq], B[(k0 + kq) * 16 + lc], acc);
    // D2 = A * D with register kk as the B operand of pseudo k-step kk: lane group kq then carries row 4 kq + kk
    f4 acc2 = {0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < 4; ++kk) acc2 = MFMA_F32(A[lc * 16 + 4 * kq + kk], acc[kk], acc2);
    for (int r = 0; r < 4; ++r) {
        C[(4 * kq + r) * 16 + lc] = acc[r];
        C[256 + (4 * kq + r) * 16 + lc] = acc2[r];
    }
}

int run_mfma_f32_selftest(hipStream_t s, double *max_err)
{
    float hA[256], hB[256], hC[512];
    double ref[256], ref2[256];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            hA[i * 16 + j] = (float)((i * 7 + j * 3) % 11 - 5);
            hB[i * 16 + j] = (float)((i * 5 + j * 13 + 1) % 17 - 8);
        }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double a = 0.0;
            for (int k = 0; k < 16; ++k) a += (double)hA[i * 16 + k] * hB[k * 16 + j];
            ref[i * 16 + j] = a;
        }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double a = 0.0;
            for (int k = 0; k < 16; ++k) a += (double)hA[i * 16 + k] * ref[k * 16 + j];
            ref2[i * 16 + j] = a;
        }
    float *dA = nullptr, *dB = nullptr, *dC = nullptr;
    if (hipMalloc(&dA, sizeof(hA)) != hipSuccess || hipMalloc(&dB, sizeof(hB)) != hipSuccess ||
        hipMalloc(&dC, sizeof(hC)) != hipSuccess)
        return -1;
    hipMemcpyAsync(dA, hA, sizeof(hA), hipMemcpyHostToDevice, s);
    hipMemcpyAsync(dB, hB, sizeof(hB), hipMemcpyHostToDevice, s);
    hipLaunchKernelGGL(mfma_f32_selftest_kernel, dim3(1), dim3(64), 0, s, dA, dB, dC);
    hipMemcpyAsync(hC, dC, sizeof(hC), hipMemcpyDeviceToHost, s);
    const hipError_t e = hipStreamSynchronize(s);
    hipFree(dA); hipFree(dB); hipFree(dC);
    if (e != hipSuccess) return -1;
    double m = 0.0;
    for (int i = 0; i < 256; ++i) {                 // integer-valued, |values| < 2^24: exact in fp32
        const double e1 = fabs((double)hC[i] - ref[i]), e2 = fabs((double)hC[256 + i] - ref2[i]);
        if (e1 > m) m = e1;
        if (e2 > m) m = e2;
    }
    *max_err = m;
    return 0;
}

}  // namespace cbo
