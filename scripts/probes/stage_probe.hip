// Hardware probe (gfx950): what one regular stage of the strip kernel costs, piece by piece.  A stage = 64
// v_mfma_f64_16x16x4_f64 (8 k-steps x 8 row tiles) per wave; MODE adds the other work of the real loop:
//   bit 0: the LDS reads (4 ds_read2_b64 + 1 ds_read_b64 per k-step, one after each of the first five MFMAs)
//   bit 1: the LDS-DMA of a later stage (12 global_load_lds_dwordx4 per wave and stage, one after an MFMA)
//   bit 2: the stage barrier (s_waitcnt + s_barrier per stage)
//   bit 3: eight waves per workgroup (two per SIMD), each with four row tiles (the row-halves split), instead of four
// Prints ms, TFLOP/s, and shader cycles per MFMA from s_memtime against the 100 MHz wall clock.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/stage_probe.hip -o scripts/probes/stage_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void glds16(const double *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
constexpr int kLd = 144, kA = 32 * kLd, kB = 4 * 32 * 16;

template <int MODE, int VAR = 0>
__global__ __launch_bounds__((MODE & 8) ? 512 : 256) void probe(const double *src, int64_t ld, double *out, int stages,
                                                               unsigned long long *clk)
{
    constexpr int NT = (MODE & 8) ? 4 : 8;           // row tiles per wave
    constexpr int NW = (MODE & 8) ? 8 : 4;
    __shared__ __align__(16) double lds[3 * (kA + kB)];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, kq = lane >> 4;
    const int half = (MODE & 8) ? (wave >> 2) : 0, cw = wave & 3;
    for (int i = tid; i < 3 * (kA + kB); i += blockDim.x) lds[i] = 1e-3 * (i & 63);
    __syncthreads();
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)lds;
    d4 acc[NT];
    for (int t = 0; t < NT; ++t) acc[t] = d4{0, 0, 0, 0};
    double af[2][NT], bf[2];
    for (int t = 0; t < NT; ++t) af[0][t] = af[1][t] = 1.0 + 1e-9 * lane;
    bf[0] = bf[1] = 0.5;
    double sink = 0.0;
    const double *g = src + (int64_t)(blockIdx.x & 7) * 64 * ld + lane * 2;     // a few MB, L2 resident
    int buf = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long w0 = wall_clock64();
    for (int s = 0; s < stages; ++s) {
        if (MODE & 4) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (MODE & 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(96 / NW) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        FENCE();
        const int bnext = (buf >= 1) ? buf - 1 : 2;
        const double *abase = lds + buf * kA + kq * kLd + lc + 64 * half;
        const double *bbase = lds + 3 * kA + buf * kB + cw * 512 + kq * 16 + lc;
        const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + 8u * (unsigned)(bnext * kA + wave * (32 / NW) * kLd));
        const double *gs = g + (int64_t)((s & 15) * 32 + wave * (32 / NW)) * ld;
        int dma = 0;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const double *an = abase + 4 * ((jj + 1) & 7) * kLd;
            if (VAR == 2 && (MODE & 1)) {                 // all LDS reads of the k-step first
#pragma unroll
                for (int t = 0; t < NT / 2; ++t) {
                    af[(jj + 1) & 1][2 * t] = an[32 * t];
                    af[(jj + 1) & 1][2 * t + 1] = an[32 * t + 16];
                }
                bf[(jj + 1) & 1] = bbase[4 * ((jj + 1) & 7) * 16];
                FENCE();
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = MFMA(af[jj & 1][t], bf[jj & 1], acc[t]);
                if ((MODE & 1) && VAR != 2) {
                    if (VAR == 3) {                       // reads issued, results never used by an MFMA
                        if (t < NT / 2) { sink += an[32 * t] + an[32 * t + 16]; }
                        else if (t == NT / 2) sink += bbase[4 * ((jj + 1) & 7) * 16];
                    } else if (t < NT / 2) {
                        af[(jj + 1) & 1][2 * t] = an[32 * t];
                        af[(jj + 1) & 1][2 * t + 1] = an[32 * t + 16];
                    } else if (t == NT / 2) {
                        bf[(jj + 1) & 1] = bbase[4 * ((jj + 1) & 7) * 16];
                    }
                }
                if ((MODE & 2) && t > NT / 2 && dma < 96 / NW) {
                    glds16(gs + (int64_t)(dma % (32 / NW)) * ld, la + 8u * (unsigned)((dma % (32 / NW)) * kLd));
                    ++dma;
                }
                if (VAR != 1) FENCE();
            }
        }
        buf = (buf == 2) ? 0 : buf + 1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double sum = 0;
    for (int t = 0; t < NT; ++t) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[(int64_t)blockIdx.x * blockDim.x + tid] = sum + lds[tid] + sink;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// Two independent workgroups per CU (own barriers, own LDS rings of 16-row stages): 4 waves x 4 row tiles each, 32-column
// strips.  What the sweep could do at M = 16384 instead of one 8-wave workgroup per CU.
constexpr int kA2 = 16 * kLd, kB2 = 2 * 16 * 16;
__global__ __launch_bounds__(256) void probe2(const double *src, int64_t ld, double *out, int stages, int skew)
{
    __shared__ __align__(16) double lds[3 * (kA2 + kB2)];          // 67,584 B: two workgroups share a CU
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, kq = lane >> 4, half = wave >> 1, cw = wave & 1;
    for (int i = tid; i < 3 * (kA2 + kB2); i += blockDim.x) lds[i] = 1e-3 * (i & 63);
    __syncthreads();
    const unsigned lds0 = (unsigned)(unsigned long)(lds_ptr_t)lds;
    d4 acc[4];
    for (int t = 0; t < 4; ++t) acc[t] = d4{0, 0, 0, 0};
    double af[2][4], bf[2];
    for (int t = 0; t < 4; ++t) af[0][t] = af[1][t] = 1.0 + 1e-9 * lane;
    bf[0] = bf[1] = 0.5;
    const double *g = src + (int64_t)(blockIdx.x & 7) * 64 * ld + lane * 2;
    int buf = 0;
    if (skew && (blockIdx.x & 1)) __builtin_amdgcn_s_sleep(100);    // start the two workgroups of a CU out of phase
    for (int s = 0; s < stages; ++s) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FENCE();
        const int bnext = (buf >= 1) ? buf - 1 : 2;
        const double *abase = lds + buf * kA2 + kq * kLd + lc + 64 * half;
        const double *bbase = lds + 3 * kA2 + buf * kB2 + cw * 256 + kq * 16 + lc;
        const unsigned la = __builtin_amdgcn_readfirstlane(lds0 + 8u * (unsigned)(bnext * kA2 + wave * 4 * kLd));
        const double *gs = g + (int64_t)((s & 31) * 16 + wave * 4) * ld;
        int dma = 0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const double *an = abase + 4 * ((jj + 1) & 3) * kLd;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = MFMA(af[jj & 1][t], bf[jj & 1], acc[t]);
                if (t < 2) {
                    af[(jj + 1) & 1][2 * t] = an[32 * t];
                    af[(jj + 1) & 1][2 * t + 1] = an[32 * t + 16];
                } else if (t == 2) {
                    bf[(jj + 1) & 1] = bbase[4 * ((jj + 1) & 3) * 16];
                }
                if (t >= 2 && dma < 5) {
                    glds16(gs + (int64_t)(dma & 3) * ld, la + 8u * (unsigned)((dma & 3) * kLd));
                    ++dma;
                }
                FENCE();
            }
        }
        buf = (buf == 2) ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double sum = 0;
    for (int t = 0; t < 4; ++t) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[(int64_t)blockIdx.x * blockDim.x + tid] = sum + lds[tid];
}

void run2(const char *name, const double *src, int64_t ld, double *out, int skew)
{
    const int stages = 4000, blocks = 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe2, dim3(blocks), dim3(256), 0, 0, src, ld, out, 20, skew);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe2, dim3(blocks), dim3(256), 0, 0, src, ld, out, stages, skew);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * stages * 16 * 2048;
    printf("%-58s %.3f ms  %.1f TF\n", name, ms, flops / ms / 1e9);
}

template <int MODE, int VAR = 0>
void run(const char *name, const double *src, int64_t ld, double *out, unsigned long long *clk)
{
    const int stages = 2000, threads = (MODE & 8) ? 512 : 256, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, VAR>), dim3(blocks), dim3(threads), 0, 0, src, ld, out, 20, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, VAR>), dim3(blocks), dim3(threads), 0, 0, src, ld, out, stages, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double mfmas_per_wave = (double)stages * 64 * ((MODE & 8) ? 0.5 : 1.0);
    const double flops = (double)blocks * 4 * stages * 64 * 2048;
    printf("%-58s %.3f ms  %.1f TF  memtime ticks/MFMA-slot %.1f  memtime rate %.0f MHz  (wall ticks %llu)\n", name, ms,
           flops / ms / 1e9, (double)h[0] / (stages * 64.0), h[1] ? (double)h[0] / h[1] * 100.0 : 0.0, h[1]);
}

int main()
{
    const int64_t ld = 4096 + 80;
    double *src, *out; unsigned long long *clk;
    hipMalloc(&src, sizeof(double) * ld * 1024);
    hipMemset(src, 0, sizeof(double) * ld * 1024);
    hipMalloc(&out, sizeof(double) * 512 * 256);
    hipMalloc(&clk, 16);
    run<0>("mfma only (8 tiles, 1 wave/SIMD)", src, ld, out, clk);
    run<1>("+ LDS reads", src, ld, out, clk);
    run<2>("+ DMA", src, ld, out, clk);
    run<4>("+ barrier", src, ld, out, clk);
    run<3>("+ LDS reads + DMA", src, ld, out, clk);
    run<5>("+ LDS reads + barrier", src, ld, out, clk);
    run<7>("+ LDS reads + DMA + barrier", src, ld, out, clk);
    run<8>("8 waves: mfma only (4 tiles, 2 waves/SIMD)", src, ld, out, clk);
    run<9>("8 waves + LDS reads", src, ld, out, clk);
    run<11>("8 waves + LDS reads + DMA", src, ld, out, clk);
    run<13>("8 waves + LDS reads + barrier", src, ld, out, clk);
    run<15>("8 waves + LDS reads + DMA + barrier", src, ld, out, clk);
    run<9, 1>("8 waves + LDS reads, compiler's own order (no fences)", src, ld, out, clk);
    run<9, 2>("8 waves + LDS reads, reads of a k-step bunched first", src, ld, out, clk);
    run<9, 3>("8 waves + LDS reads whose results no MFMA uses", src, ld, out, clk);
    run<15, 1>("8 waves everything, compiler's own order", src, ld, out, clk);
    run<15, 2>("8 waves everything, reads bunched first", src, ld, out, clk);
    run2("2 workgroups/CU x 4 waves, 16-row stages, everything", src, ld, out, 0);
    run2("  the same, the two workgroups started out of phase", src, ld, out, 1);
    return 0;
}
