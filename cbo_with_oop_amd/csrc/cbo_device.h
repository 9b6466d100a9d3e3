// Device functions shared by several kernel translation units: one kernel-matrix element in GPy's operation order,
// the causal Expected Improvement of one candidate, the arg-max comparator.  Every function pins its own
// floating-point contraction (no FMA fusion beyond what is written): the callers' results must not depend on
// which translation unit inlined them.
#pragma once

#include "cbo_internal.h"

namespace cbo {

// One kernel-matrix element, GPy operation order (Stationary._unscaled_dist: GEMM-trick squared distance from the
// same |x|^2 sums, clip at 0; RBF.K_of_r), restating /root/reference/src/utils_functions/causal_kernels.py:45-62
// without the rank-1 causal term (added by the caller).
template <int D>
__device__ __forceinline__ double kernel_value(const double *xi, const double *xj, double sqi, double sqj,
                                               double variance, double inv_l2, bool force_zero)
{
#pragma clang fp contract(off)
    // np.dot(X, X2.T): BLAS accumulates a_k*b_k with FMAs from a zero accumulator.
    double dot = __dmul_rn(xi[0], xj[0]);
#pragma unroll
    for (int k = 1; k < D; ++k) dot = __fma_rn(xi[k], xj[k], dot);
    double r2 = __dadd_rn(__dmul_rn(-2.0, dot), __dadd_rn(sqi, sqj));
    if (force_zero) r2 = 0.0;
    r2 = (r2 < 0.0) ? 0.0 : r2;                       // np.clip(r2, 0, inf) (NaN stays NaN)
    // GPy goes r = sqrt(r2) / lengthscale, then r*r.  The round trip through the square root costs ~40 fp64
    // instructions per element and changes r^2 by at most a couple of ulp (far below the 1e-16-level
    // differences between exp() implementations), so the squared scaled distance is formed directly;
    // inv_l2 = 1 / lengthscale^2 is exactly 1 for the reference's lengthscale = 1 (and for ARD, whose inputs
    // are pre-scaled).
    return __dmul_rn(variance, exp(__dmul_rn(-0.5, __dmul_rn(r2, inv_l2))));
}

// scipy.special.ndtr (cephes ndtr.c): 0.5 erfc(-x/sqrt2) split at |x/sqrt2| < sqrt(1/2).
__device__ __forceinline__ double ndtr(double a)
{
#pragma clang fp contract(off)
    const double SQRTH = 7.07106781186547524401E-1;
    if (isnan(a)) return a;
    const double x = a * SQRTH;
    const double zabs = fabs(x);
    double y;
    if (zabs < SQRTH) {
        y = 0.5 + 0.5 * erf(x);
    } else {
        y = 0.5 * erfc(zabs);
        if (x > 0) y = 1.0 - y;
    }
    return y;
}

// (va, ia) beats (vb, ib): larger value, NaN maximal (numpy.argmax), lowest index on ties
__device__ __forceinline__ bool better(double va, int64_t ia, double vb, int64_t ib)
{
    const bool na = isnan(va), nb = isnan(vb);
    if (na != nb) return na;
    if (na || va == vb) return ia < ib;
    return va > vb;
}

__device__ __forceinline__ void wave_argmax(double &v, int64_t &i)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(v, off);
        const int64_t oi = __shfl_down(i, off);
        if (better(ov, oi, v, i)) { v = ov; i = oi; }
    }
}

// Posterior epilogue of one candidate (GPy Posterior._raw_predict / GP.predict + CausalRBF.Kdiag,
// causal_kernels.py:64-79): var = clip(Kdiag - q, 1e-15) (+ noise), mean = mu + m(x*).
__device__ __forceinline__ void posterior_of(double q, double mu, double pm, double pv, bool causal, const AcqParams &p,
                                             double &mean, double &var)
{
#pragma clang fp contract(off)
    const double kss = causal ? (p.variance + pv) : p.variance;
    var = kss - q;
    var = (var < kGpyVarClip) ? kGpyVarClip : var;                  // np.clip(var, 1e-15, inf); NaN stays NaN
    if (p.include_noise) var = var + p.noise_var;                   // Gaussian likelihood predictive_values
    mean = mu;
    if (causal) mean = mean + pm;                                   // GP._raw_predict: mu += mean_function.f(Xnew)
}

// CausalExpectedImprovement.evaluate (causal_acquisition_functions.py:27-43, 77-88) / Cost (emukit Quotient):
// s (u Phi(u) + phi(u)) / cost, task 'max' returns -EI with the same u (reference quirk).
__device__ __forceinline__ double acquisition_of(double mean, double var, const AcqParams &p)
{
#pragma clang fp contract(off)
    const double s = sqrt(var);
    const double mj = mean + p.ei_jitter;
    const double u = (p.y_best - mj) / s;
    const double pdf = exp(-(u * u) / 2.0) / 2.5066282746310002;   // scipy _norm_pdf: exp(-x**2/2)/sqrt(2 pi)
    const double cdf = ndtr(u);
    double imp = s * (u * cdf + pdf);
    if (p.task != CBO_TASK_MIN) imp = -imp;
    return imp / p.cost;
}

}  // namespace cbo
