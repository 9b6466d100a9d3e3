// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950, alone and together.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_f64_probe.hip -o scripts/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 = mfma only, 1 = valu only, 2 = even waves mfma / odd waves valu, 3 = both in every wave
__global__ __launch_bounds__(512) void probe(double *out, int iters, double a0, double b0)
{
    const int wave = threadIdx.x >> 6;
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    double v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3 + i;
    const double a = a0 + threadIdx.x * 1e-9, b = b0;
    const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && (wave & 1) == 0);
    const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && (wave & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fma(v[i], a, b);
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int threads, int blocks, int iters, double *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0000001, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64;
    double mf = 0, vf = 0;
    if (MODE == 0 || MODE == 3) mf = waves * iters * 4.0 * 2048;
    if (MODE == 2) mf = waves / 2 * iters * 4.0 * 2048;
    if (MODE == 1 || MODE == 3) vf = waves * iters * 16.0 * 128;
    if (MODE == 2) vf = waves / 2 * iters * 16.0 * 128;
    printf("%-34s threads=%d blocks=%d  %.3f ms  mfma %.1f TF  valu %.1f TF  total %.1f TF\n", name, threads, blocks,
           ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
}

int main()
{
    double *out; hipMalloc(&out, sizeof(double) * 512 * 2048);
    const int it = 20000;
    run<0>("mfma only, 1 wave/SIMD", 256, 256, it, out);
    run<0>("mfma only, 2 waves/SIMD", 512, 256, it, out);
    run<0>("mfma only, 4 waves/SIMD", 512, 512, it, out);
    run<1>("valu fma only, 1 wave/SIMD", 256, 256, it, out);
    run<1>("valu fma only, 2 waves/SIMD", 512, 256, it, out);
    run<1>("valu fma only, 4 waves/SIMD", 512, 512, it, out);
    run<2>("split waves mfma|valu, 2 waves/SIMD", 512, 256, it, out);
    run<2>("split waves mfma|valu, 4 waves/SIMD", 512, 512, it, out);
    run<3>("both in each wave, 1 wave/SIMD", 256, 256, it, out);
    run<3>("both in each wave, 2 waves/SIMD", 512, 256, it, out);
    return 0;
}
