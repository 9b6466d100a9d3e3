// Does hipExtStreamCreateWithCUMask confine a stream's kernels on this device, and how do mask bits map to
// (XCC, SE, CU)?  Each workgroup spins ~20 us and records where it ran.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#include <cstdint>
__global__ void where(unsigned *out, long long spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("CUs %d\n", prop.multiProcessorCount);
    const int nwg = 2048;
    unsigned *d; hipMalloc(&d, nwg * 8);
    std::vector<unsigned> h(2 * nwg);
    const int words = (prop.multiProcessorCount + 31) / 32;
    for (int trial = 0; trial < 5; ++trial) {
        std::vector<uint32_t> mask(words, 0);
        const char *name = "";
        if (trial == 0) { name = "all"; for (auto &w : mask) w = 0xffffffffu; }
        if (trial == 1) { name = "first 32 bits"; mask[0] = 0xffffffffu; }
        if (trial == 2) { name = "bits 0..7"; mask[0] = 0xffu; }
        if (trial == 3) { name = "all but bits 0..7"; for (auto &w : mask) w = 0xffffffffu; mask[0] = 0xffffff00u; }
        if (trial == 4) { name = "every 32nd bit"; for (auto &w : mask) w = 1u; }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, words, mask.data());
        if (e != hipSuccess) { printf("%s: create failed %s\n", name, hipGetErrorString(e)); continue; }
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(where, dim3(nwg), dim3(64), 0, s, d, 2000LL);
        hipStreamSynchronize(s);
        hipEventRecord(a, s);
        hipLaunchKernelGGL(where, dim3(nwg), dim3(64), 0, s, d, 2000LL);
        hipEventRecord(b, s);
        hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
        std::set<unsigned> cus; std::set<unsigned> xccs;
        for (int i = 0; i < nwg; ++i) {
            const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu); xccs.insert(xcc);
        }
        printf("%-20s: %.3f ms, distinct CUs used %zu, XCCs %zu\n", name, ms, cus.size(), xccs.size());
        if (trial == 2 || trial == 4) { for (auto c : cus) printf(" x%u.se%u.sh%u.cu%u", c >> 12, (c >> 8) & 7, (c >> 4) & 1, c & 15); printf("\n"); }
        hipStreamDestroy(s);
    }
    return 0;
}
