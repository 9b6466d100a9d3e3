/* Extended-precision (x87 80-bit long double) restatement of the GP posterior maths of
 * oracle/gp_oracle.py.  TEST INFRASTRUCTURE ONLY (see oracle/gp_oracle.py header): it is the
 * arbiter for ill-conditioned cases, where two correct fp64 implementations (LAPACK in the oracle,
 * the HIP kernels in the product) legitimately differ by eps*sqrt(cond(Ky)) in the predictive
 * variance.  Tests compare both against this and require the HIP error to be no worse than a small
 * multiple of the fp64 oracle's own error.
 *
 * Maths restated (citations relative to /root/reference/):
 *   K = s2*exp(-0.5 r^2) + sqrt(v)sqrt(v')^T      src/utils_functions/causal_kernels.py:45-62
 *   Ky = K + diag_add*I ; L = chol(Ky) ; alpha    GPy ExactGaussianInference (src/GaussianProcessFactory.py:57-73 builds it)
 *   mu = Kx^T alpha + m(X*) ; var = clip(Kdiag - |L^-1 Kx|^2, 1e-15) + noise   GPy Posterior._raw_predict / GP.predict
 * Distances use the direct difference form in long double (the exact quantity both fp64 formulas approximate).
 * No jitter ladder here: the caller passes the total diagonal addition the fp64 run ended up with.
 *
 * Build: see oracle/Makefile (gcc -O2 -shared -fPIC).  O(N^3 + N^2 M) scalar loops: keep N <= ~1500.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef long double ld;

static ld kern(const double *a, const double *b, int d, ld s2, const double *ls, int ard)
{
    ld r2 = 0;
    for (int k = 0; k < d; k++) {
        ld l = ard ? (ld)ls[k] : (ld)ls[0];
        ld t = ((ld)a[k] - (ld)b[k]) / l;
        r2 += t * t;
    }
    return s2 * expl(-0.5L * r2);
}

/* returns 0 on success, j+1 if the j-th pivot is not positive */
int gp_truth_predict(long n, int d, const double *X, const double *y, const double *mX,
                     const double *vX, double variance, const double *ls, int ard,
                     double diag_add, double noise_out, long m, const double *Xs,
                     const double *mXs, const double *vXs, int include_noise,
                     double *mean_out, double *var_out, double *alpha_out)
{
    ld *L = (ld *)malloc(sizeof(ld) * n * n);
    ld *z = (ld *)malloc(sizeof(ld) * n);
    ld *al = (ld *)malloc(sizeof(ld) * n);
    ld *kx = (ld *)malloc(sizeof(ld) * n);
    if (!L || !z || !al || !kx) return -1;
    ld s2 = (ld)variance;
    for (long i = 0; i < n; i++)
        for (long j = 0; j <= i; j++) {
            ld k = kern(X + i * d, X + j * d, d, s2, ls, ard);
            if (vX) k += sqrtl((ld)vX[i]) * sqrtl((ld)vX[j]);
            if (i == j) k += (ld)diag_add;
            L[i * n + j] = k;
        }
    /* Cholesky, row-oriented (Cholesky-Banachiewicz) */
    for (long i = 0; i < n; i++) {
        for (long j = 0; j <= i; j++) {
            ld s = L[i * n + j];
            for (long k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
            if (i == j) {
                if (!(s > 0)) { free(L); free(z); free(al); free(kx); return (int)(i + 1); }
                L[i * n + i] = sqrtl(s);
            } else {
                L[i * n + j] = s / L[j * n + j];
            }
        }
    }
    /* alpha = Ky^-1 (y - m) */
    for (long i = 0; i < n; i++) {
        ld s = (ld)y[i] - (mX ? (ld)mX[i] : 0);
        for (long k = 0; k < i; k++) s -= L[i * n + k] * z[k];
        z[i] = s / L[i * n + i];
    }
    for (long i = n - 1; i >= 0; i--) {
        ld s = z[i];
        for (long k = i + 1; k < n; k++) s -= L[k * n + i] * al[k];
        al[i] = s / L[i * n + i];
    }
    if (alpha_out) for (long i = 0; i < n; i++) alpha_out[i] = (double)al[i];
    for (long c = 0; c < m; c++) {
        const double *xs = Xs + c * d;
        ld mu = 0, q = 0;
        ld sv = vXs ? sqrtl((ld)vXs[c]) : 0;
        for (long i = 0; i < n; i++) {
            ld k = kern(X + i * d, xs, d, s2, ls, ard);
            if (vX && vXs) k += sqrtl((ld)vX[i]) * sv;
            mu += k * al[i];
            ld s = k;
            for (long kk = 0; kk < i; kk++) s -= L[i * n + kk] * kx[kk];
            kx[i] = s / L[i * n + i];
            q += kx[i] * kx[i];
        }
        ld var = s2 + (vXs ? (ld)vXs[c] : 0) - q;
        if (var < 1e-15L) var = 1e-15L;
        if (include_noise) var += (ld)noise_out;
        if (mXs) mu += (ld)mXs[c];
        mean_out[c] = (double)mu;
        var_out[c] = (double)var;
    }
    free(L); free(z); free(al); free(kx);
    return 0;
}
