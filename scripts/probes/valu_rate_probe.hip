// Issue rate of fp64 vector instructions on one SIMD (gfx950): cycles per wave-instruction with every SIMD holding 8 waves,
// each wave running 8 independent chains.  build: hipcc -O3 --offload-arch=gfx950 valu_rate_probe.hip -o valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP> __global__ __launch_bounds__(256) void k(double *out, int iters, double seed)
{
    asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" : : : "vcc", "s20", "s21");
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x[i]));
            if (OP == 1) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[i]));
            if (OP == 2) asm volatile("v_rsq_f64 %0, %0" : "+v"(x[i]));
            if (OP == 3) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x[i]));
            if (OP == 4) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(x[i]));
            if (OP == 5) asm volatile("v_add_f64 %0, %0, %0" : "+v"(x[i]));
            if (OP == 6) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x[i]));
            if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(reinterpret_cast<int *>(&x[i])[0]));
            if (OP == 8) asm volatile("v_div_scale_f64 %0, vcc, %0, %0, %0" : "+v"(x[i]) : : "vcc");
            if (OP == 9) asm volatile("v_div_fmas_f64 %0, %0, %0, %0" : "+v"(x[i]));
            if (OP == 10) asm volatile("v_div_fixup_f64 %0, %0, %0, %0" : "+v"(x[i]));
            if (OP == 11) asm volatile("v_rndne_f64 %0, %0" : "+v"(x[i]));
            if (OP == 12) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(reinterpret_cast<int *>(&x[i])[0]) : "v"(x[i]));
            if (OP == 13) asm volatile("v_mov_b64 %0, %0" : "+v"(x[i]));
            if (OP == 14) asm volatile("v_cmp_lt_f64 vcc, %0, %0" : : "v"(x[i]) : "vcc");
            if (OP == 16) asm volatile("v_mov_b32 %0, %0" : "+v"(reinterpret_cast<int *>(&x[i])[0]));
            if (OP == 17) asm volatile("v_add_u32 %0, %0, %0" : "+v"(reinterpret_cast<int *>(&x[i])[0]));
            if (OP == 18) asm volatile("v_cndmask_b32_e64 %0, %0, %0, s[20:21]" : "+v"(reinterpret_cast<int *>(&x[i])[0]));
            if (OP == 19) asm volatile("v_and_b32 %0, %0, %0" : "+v"(reinterpret_cast<int *>(&x[i])[0]));
            if (OP == 20) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(reinterpret_cast<int *>(&x[i])[0]) : "v"(reinterpret_cast<int *>(&x[(i + 1) & 7])[1]), "v"(reinterpret_cast<int *>(&x[(i + 2) & 7])[1]));
            if (OP == 21) asm volatile("v_fma_f64 %0, %0, %0, s[20:21]" : "+v"(x[i]));
            if (OP == 22) asm volatile("v_cndmask_b32_e64 %0, %0, %0, vcc" : "+v"(reinterpret_cast<int *>(&x[i])[0]));
            if (OP == 23) asm volatile("v_cndmask_b32_e64 %0, %0, %0, s[20:21]" : "+v"(reinterpret_cast<int *>(&x[i])[1]));
            if (OP == 24) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(reinterpret_cast<int *>(&x[i])[1]));
            if (OP == 25) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(reinterpret_cast<int *>(&x[i])[0]));
            if (OP == 26) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(reinterpret_cast<int *>(&x[i])[0]) : : "vcc");
            if (OP == 15) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(x[i]));
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, double *out)
{
    const int iters = 4096, blocks = 2048;       // 8 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(out, 16, 1.5);
    hipEventRecord(a); k<OP><<<blocks, 256>>>(out, iters, 1.5); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    int clk; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double per_simd = (double)iters * 8 * 8;            // wave-instructions per SIMD
    printf("%-18s %8.3f ms  %6.2f cycles per wave-instruction at the nominal %d MHz\n", name, ms, ms * 1e-3 * clk * 1e3 / per_simd, clk / 1000);
}
int main()
{
    double *out; hipMalloc(&out, 2048 * 256 * 8);
    run<0>("v_fma_f64", out); run<4>("v_mul_f64", out); run<5>("v_add_f64", out); run<1>("v_rcp_f64", out);
    run<2>("v_rsq_f64", out); run<3>("v_sqrt_f64", out); run<6>("v_ldexp_f64", out); run<7>("v_cndmask_b32", out);
    run<8>("v_div_scale_f64", out); run<9>("v_div_fmas_f64", out); run<10>("v_div_fixup_f64", out); run<11>("v_rndne_f64", out);
    run<12>("v_cvt_i32_f64", out); run<13>("v_mov_b64", out); run<14>("v_cmp_lt_f64", out); run<15>("v_frexp_mant_f64", out);
    run<16>("v_mov_b32", out); run<17>("v_add_u32", out); run<18>("v_cndmask_b32 sgpr", out); run<19>("v_and_b32", out);
    run<20>("v_cndmask 3-reg", out); run<21>("v_fma_f64 sgpr c", out);
    run<22>("cndmask e64 vcc", out); run<23>("cndmask e64 sgpr hi", out); run<24>("cndmask e32 vcc hi", out); run<25>("cndmask e32 0,v", out);
    run<26>("v_addc_co_u32", out);
    return 0;
}
