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

// Form 3: the four rows of a pivot block replicated to every 16-lane row by one MFMA with a 0/1 selection operand
// (t[r] = D[4b + r][lc] on all lane rows); the pivots then need only scalars (v_readlane), no ds_bpermute in the chain.
template <>
__device__ __forceinline__ d4 factor<3>(d4 din, int lane, d4 &eout)
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
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        d4 t = MFMA_F64(sel, d[b], zero);
        d4 s = MFMA_F64(sel, e[b], zero);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piv = 4 * b + j;
            double pj = readlane_f64(t[j], piv);
            if (!(pj > 0.0)) pj = 1.0;
            const double inv = rsqrt(pj);
            const double dj = pj * inv;
            t[j] = (lc > piv) ? t[j] * inv : ((lc == piv) ? dj : 0.0);
            s[j] *= inv;
#pragma unroll
            for (int i = j + 1; i < 4; ++i) {
                const double u = readlane_f64(t[j], 4 * b + i);
                t[i] = fma(-u, t[j], t[i]);
                s[i] = fma(-u, s[j], s[i]);
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
    eout = e;
    return d;
}

// Form 4: as form 3, and the 4x4 diagonal sub-block of the pivot block is factored FIRST on uniform scalars (the ten
// entries read once with v_readlane; every lane repeats the same arithmetic): the dependent chain per pivot is
// rsqrt -> one multiply -> one fma.  The 16-wide row scalings / updates then use those scalars, off the chain.
// Same operations on the same values in the same order as forms 0 and 3 -> the same bits.
template <>
__device__ __forceinline__ d4 factor<4>(d4 din, int lane, d4 &eout)
{
    const int lc = lane & 15, kq = lane >> 4;
    d4 d, e;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        d[r] = (kq + 4 * r <= lc) ? din[r] : 0.0;
        e[r] = (kq + 4 * r == lc) ? 1.0 : 0.0;
    }
    const double sel = ((lc >> 2) == kq) ? 1.0 : 0.0;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        d4 t = MFMA_F64(sel, d[b], zero);
        d4 s = MFMA_F64(sel, e[b], zero);
        double a[4][4], u[4][4], inv[4], dj[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i; j < 4; ++j) a[i][j] = readlane_f64(t[i], 4 * b + j);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double pj = a[j][j];
            if (!(pj > 0.0)) pj = 1.0;
            inv[j] = rsqrt(pj);
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
    eout = e;
    return d;
}

// Form 5: form 4 with ocml's rsqrt sequence spelled out minus its class-check selects (identical bits for normal
// positive pivots), and the positivity test taken off the chain (first bad pivot recorded, reported once).  Form 4: as form 3, and the 4x4 diagonal sub-block of the pivot block is factored FIRST on uniform scalars (the ten
// entries read once with v_readlane; every lane repeats the same arithmetic): the dependent chain per pivot is
// rsqrt -> one multiply -> one fma.  The 16-wide row scalings / updates then use those scalars, off the chain.
// Same operations on the same values in the same order as forms 0 and 3 -> the same bits.
template <>
__device__ __forceinline__ d4 factor<5>(d4 din, int lane, d4 &eout)
{
    const int lc = lane & 15, kq = lane >> 4;
    d4 d, e;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        d[r] = (kq + 4 * r <= lc) ? din[r] : 0.0;
        e[r] = (kq + 4 * r == lc) ? 1.0 : 0.0;
    }
    const double sel = ((lc >> 2) == kq) ? 1.0 : 0.0;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    int bad = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
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
    if (bad) e[0] += 1e300;
    eout = e;
    return d;
}

__device__ unsigned long long g_clk[4];
template <int FORM>
__global__ void probe(const double *T, double *U, double *E, int reps)
{
    const int lane = threadIdx.x, lc = lane & 15, kq = lane >> 4;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    d4 t;
    for (int r = 0; r < 4; ++r) t[r] = T[(kq + 4 * r) * 16 + lc];
    d4 u, e, acc = {0, 0, 0, 0};
    for (int i = 0; i < reps; ++i) {
        d4 in = t;
        in[0] += acc[0] * 1e-300;                 // serialise the repetitions (on the factor AND on its inverse: the
        u = factor<FORM>(in, lane, e);            // kernel needs both before its barrier)
        acc += u + e;
    }
    for (int r = 0; r < 4; ++r) {
        U[(kq + 4 * r) * 16 + lc] = u[r];
        E[(kq + 4 * r) * 16 + lc] = e[r];
    }
    if (lane == 0) {                              // shader cycles and 100 MHz ticks of the whole loop
        g_clk[0] = __builtin_amdgcn_s_memtime() - c0;
        g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
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
    std::vector<double> U0, E0;
    for (int form = 0; form < 6; ++form) {
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(a);
            if (form == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, dT, dU, dE, reps);
            if (form == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, dT, dU, dE, reps);
            if (form == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, dT, dU, dE, reps);
            if (form == 3) hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, dT, dU, dE, reps);
            if (form == 4) hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, dT, dU, dE, reps);
            if (form == 5) hipLaunchKernelGGL(probe<5>, dim3(1), dim3(64), 0, 0, dT, dU, dE, reps);
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
        if (form == 0) { U0 = U; E0 = E; }
        int same = 1;
        for (int i = 0; i < 16; ++i)
            for (int j = i; j < 16; ++j) same &= (U[i * 16 + j] == U0[i * 16 + j]);
        for (int i = 0; i < 256; ++i) same &= (E[i] == E0[i]) ? 1 : 2 * 0;
        unsigned long long clk[4];
        hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
        printf("form %d: %.3f us per 16x16 factor (%.0f ns per pivot; %.0f shader cycles per factor at %.0f MHz); max |U - ref| %.2e, |E L - I| %.2e; bits equal to form 0: %d\n", form,
               ms * 1e3 / reps, ms * 1e6 / reps / 16, (double)clk[0] / reps, (double)clk[0] / (double)clk[1] * 100.0, err, ierr, same);
    }
    return 0;
}
