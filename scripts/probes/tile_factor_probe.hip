// Probe: cost of one 16x16 register Cholesky (the chain of the diagonal-block kernel) in several forms.
// hipcc --offload-arch=gfx950 -O3 -o tile_factor_probe tile_factor_probe.hip ; ./tile_factor_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(p) and sqrt(p) for a normal positive p: hardware estimate + two coupled (Goldschmidt) steps
__device__ __forceinline__ void rsqrt_sqrt(double p, double &inv, double &root)
{
    double y = __builtin_amdgcn_rsq(p);
    double g = p * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    root = g;
    inv = h + h;
}

// all four 16-lane rows <- row J of v (two half-row swaps per dword; J is a compile-time constant)
template <int J>
__device__ __forceinline__ double bcast_row(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    int out[2];
    int src[2] = {lo, hi};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        int a = src[w], b = src[w];
        // permlane16_swap(a, b): a.row1 <-> b.row0, a.row3 <-> b.row2
        auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        a = r16[0]; b = r16[1];                       // a = [r0,r0,r2,r2], b = [r1,r1,r3,r3]
        int x = (J & 1) ? b : a;                      // rows [rJ', rJ', rJ'', rJ'']
        int y = x;
        // permlane32_swap(x, y): x.rows(2,3) <-> y.rows(0,1)
        auto r32 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
        out[w] = (J & 2) ? r32[1] : r32[0];
    }
    return __hiloint2double(out[1], out[0]);
}

template <int FORM>
__device__ __forceinline__ d4 factor(d4 din, int lane, d4 &eout)
{
    const int lc = lane & 15, kq = lane >> 4;
    d4 d, e;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        d[r] = (kq + 4 * r <= lc) ? din[r] : 0.0;
        e[r] = (kq + 4 * r == lc) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piv = 4 * b + j;
            double pj = readlane_f64(d[b], 16 * j + piv);
            if (!(pj > 0.0)) pj = 1.0;
            double inv, dj;
            if (FORM == 0) { inv = rsqrt(pj); dj = pj * inv; }
            else rsqrt_sqrt(pj, inv, dj);
            const double scaled = (lc > piv) ? d[b] * inv : ((lc == piv) ? dj : 0.0);
            if (kq == j) {
                d[b] = scaled;
                e[b] *= inv;
            }
            if (j < 3) {
                double ujc, ejc, ujr;
                if (FORM <= 1) {
                    ujc = __shfl(d[b], 16 * j + lc);
                    ejc = __shfl(e[b], 16 * j + lc);
                    ujr = __shfl(d[b], 16 * j + 4 * b + kq);
                } else {
                    ujc = (j == 0) ? bcast_row<0>(d[b]) : (j == 1) ? bcast_row<1>(d[b]) : bcast_row<2>(d[b]);
                    ejc = (j == 0) ? bcast_row<0>(e[b]) : (j == 1) ? bcast_row<1>(e[b]) : bcast_row<2>(e[b]);
                    // U[piv][4b + kq] for the rows below the pivot inside this 4-row sub-block
                    const double s1 = readlane_f64(d[b], 16 * j + 4 * b + 1);
                    const double s2 = readlane_f64(d[b], 16 * j + 4 * b + 2);
                    const double s3 = readlane_f64(d[b], 16 * j + 4 * b + 3);
                    ujr = (kq == 1) ? s1 : ((kq == 2) ? s2 : s3);
                }
                if (kq > j) {
                    d[b] = fma(-ujr, ujc, d[b]);
                    e[b] = fma(-ujr, ejc, e[b]);
                }
            }
        }
        if (b < 3) {
            const d4 keep = d, keep_e = e;
            const double na = -d[b];
            d = MFMA_F64(na, d[b], d);
            e = MFMA_F64(na, e[b], e);
#pragma unroll
            for (int r = 0; r <= b; ++r) { d[r] = keep[r]; e[r] = keep_e[r]; }
        }
    }
    eout = e;
    return d;
}

template <int FORM>
__global__ void probe(const double *T, double *U, double *E, int reps)
{
    const int lane = threadIdx.x, lc = lane & 15, kq = lane >> 4;
    d4 t;
    for (int r = 0; r < 4; ++r) t[r] = T[(kq + 4 * r) * 16 + lc];
    d4 u, e, acc = {0, 0, 0, 0};
    for (int i = 0; i < reps; ++i) {
        d4 in = t;
        in[0] += acc[0] * 1e-300;                 // serialise the repetitions
        u = factor<FORM>(in, lane, e);
        acc += u;
    }
    for (int r = 0; r < 4; ++r) {
        U[(kq + 4 * r) * 16 + lc] = u[r];
        E[(kq + 4 * r) * 16 + lc] = e[r];
    }
}

int main()
{
    std::vector<double> T(256), U(256), E(256), ref(256, 0.0);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) T[i * 16 + j] = exp(-0.05 * (i - j) * (i - j)) + (i == j ? 0.3 : 0.0);
    for (int j = 0; j < 16; ++j) {                    // upper factor: T = R^T R
        for (int i = 0; i <= j; ++i) {
            double s = T[i * 16 + j];
            for (int k = 0; k < i; ++k) s -= ref[k * 16 + i] * ref[k * 16 + j];
            ref[i * 16 + j] = (i == j) ? sqrt(s) : s / ref[i * 16 + i];
        }
    }
    double *dT, *dU, *dE;
    hipMalloc(&dT, 2048); hipMalloc(&dU, 2048); hipMalloc(&dE, 2048);
    hipMemcpy(dT, T.data(), 2048, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 4000;
    for (int form = 0; form < 3; ++form) {
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(a);
            if (form == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, dT, dU, dE, reps);
            if (form == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, dT, dU, dE, reps);
            if (form == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, dT, dU, dE, reps);
            hipEventRecord(b);
            hipEventSynchronize(b);
        }
        float ms;
        hipEventElapsedTime(&ms, a, b);
        hipMemcpy(U.data(), dU, 2048, hipMemcpyDeviceToHost);
        hipMemcpy(E.data(), dE, 2048, hipMemcpyDeviceToHost);
        double err = 0, ierr = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                err = fmax(err, fabs(U[i * 16 + j] - ref[i * 16 + j]));
                double s = 0;                         // E = inv(R)^T  ->  sum_k E[i][k] R[k][j]^T ... check R^T E^T = I
                for (int k = 0; k < 16; ++k) s += E[i * 16 + k] * ref[k * 16 + j] ;
                (void)s;
            }
        // inverse check: E[i][k] = inv(L)[i][k] with L = R^T: sum_k E[i][k] L[k][j] = delta
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = 0;
                for (int k = 0; k < 16; ++k) s += E[i * 16 + k] * ref[j * 16 + k];
                ierr = fmax(ierr, fabs(s - (i == j ? 1.0 : 0.0)));
            }
        printf("form %d: %.3f us per 16x16 factor (%.0f ns per pivot); max |U - ref| %.2e, |E L - I| %.2e\n", form,
               ms * 1e3 / reps, ms * 1e6 / reps / 16, err, ierr);
    }
    return 0;
}
