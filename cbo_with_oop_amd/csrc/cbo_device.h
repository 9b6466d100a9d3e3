// Device functions shared by several kernel translation units: one kernel-matrix element in GPy's operation order,
// the causal Expected Improvement of one candidate, the arg-max comparator.  Every function pins its own
// floating-point contraction (no FMA fusion beyond what is written): the callers' results must not depend on
// which translation unit inlined them.
#pragma once

#include "cbo_internal.h"

namespace cbo {

// c ? a : b through the VOP3 encoding of v_cndmask_b32.  gfx950 issues the VOP2 encoding (mask implicitly in vcc) once per
// ~22 cycles and SIMD, the VOP3 encoding (mask in any scalar pair, vcc included) once per 4.6 like every other full-rate
// instruction (scripts/probes/valu_rate_probe.hip, profiles/r05_valu_rate_probe.txt) -- and the compiler shrinks a select
// to VOP2 wherever its mask lands in vcc.  The EI pass executed ~10 such selects per candidate.
__device__ __forceinline__ unsigned select_u32(unsigned long long mask, unsigned a, unsigned b)
{
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}
__device__ __forceinline__ double select_f64(bool c, double a, double b)
{
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(c);
    return __hiloint2double((int)select_u32(mask, (unsigned)__double2hiint(a), (unsigned)__double2hiint(b)),
                            (int)select_u32(mask, (unsigned)__double2loint(a), (unsigned)__double2loint(b)));
}
// s = sqrt(x) (IEEE, round to nearest) and rs ~ 1 / s from ONE hardware estimate of 1/sqrt(x): the coupled Newton steps of the
// compiler's own sqrt expansion -- without that expansion's scaling of arguments below 2^-767 (three VOP2 selects and two
// v_ldexp per call; a predictive variance is clipped at 1e-15) -- whose second variable h converges to 1 / (2 s).  rs is good
// to an ulp or two: a quotient a / s formed as q = a rs, q += (a - q s) rs is within half an ulp and a bit of the IEEE one
// (4 instructions where the IEEE division is 14).  x = 0 and x = +inf: s = x, rs is not meaningful (`special` says so).
// Negative and NaN arguments give NaN in both.
__device__ __forceinline__ void sqrt_and_reciprocal(double x, double &s, double &rs, bool &special)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    g = fma(fma(-g, g, x), h, g);
    special = __builtin_amdgcn_class(x, 0x260);                      // +-0, +inf
    s = select_f64(special, x, g);
    rs = h + h;
}

// scipy.special.ndtr is cephes ndtr.c: Phi(a) = 0.5 + 0.5 erf(x) for |x| < sqrt(1/2), x = a / sqrt 2, and 0.5 erfc(|x|)
// (reflected for x > 0) beyond, with cephes' own erf / erfc: erf(x) = x T(x^2) / U(x^2) for |x| <= 1; erfc(x) = 1 - erf(x)
// below 1, exp(-x^2) P(x) / Q(x) on [1, 8), exp(-x^2) R(x) / S(x) from 8 on, 0 once x^2 > MAXLOG.  The same rational
// functions here, coefficient for coefficient (checked against scipy.special.erf / erfc on the host: erf bit for bit, erfc to
// 5e-16), Horner steps as FMAs -- and ONE exponential for the candidate: erfc's exp(-x^2) is the density's exp(-a^2 / 2) up
// to the rounding of the argument (|x^2| ulps of the tail: 2e-13 of Phi at a = -35, nothing at the |a| <= 6 an acquisition
// that matters lives at).  The quotients of the rational functions go through recip_plain, the exponential through
// exp_nonpositive: each within an ulp or two of the IEEE division / the library exponential they stand for, as the FMA
// Horner steps are of scipy's -- an Expected Improvement good to a few 1e-16 (times u^2 in the lower tail) either way.
// Round 4 called the device library's erf / erfc / exp per candidate: 485 vector instructions, the exponential
// evaluated twice; the pass ran at 0.29 of the HBM roofline it is priced against.
// One Horner step p z + C with the coefficient in a scalar register pair: left to itself the compiler emits the two-operand
// v_fmac_f64, whose addend is its destination, behind two v_mov_b32 that put the 64-bit constant there -- three vector
// instructions per step (122 of the pass's 431 were such moves); gfx950 has no 64-bit literal operands.
__device__ __forceinline__ double horner(double p, double z, double coefficient)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(coefficient));
    return r;
}
// 1 / q to the last bit or two for a q of moderate size (the denominators of the rational functions below: no scaling,
// no special cases): the hardware estimate and two Newton steps -- 6 instructions where an IEEE division is 14
__device__ __forceinline__ double recip_plain(double q)
{
    double r = __builtin_amdgcn_rcp(q);
    r = fma(fma(-q, r, 1.0), r, r);
    r = fma(fma(-q, r, 1.0), r, r);
    return r;
}
// exp(x) for x <= 0 (the density's -u^2 / 2): 2^k exp(r), k = rint(x log2 e), r = x - k ln 2 in two parts (|r| <= 0.35),
// exp(r) by its Taylor polynomial of degree 12 (truncation 1.7e-16); 4.7e-16 relative against the host's exp over [-745, 0].
// No range checks beyond the underflow: 18 instructions where the device library's exp is ~45.
__device__ __forceinline__ double exp_nonpositive(double x)
{
    const double k = rint(x * 1.4426950408889634);
    const double r = fma(-k, 1.90821492927058770002e-10, fma(-k, 6.93147180369123816490e-01, x));
    double p = 1.0 / 479001600.0;
    p = horner(p, r, 1.0 / 39916800.0);
    p = horner(p, r, 1.0 / 3628800.0);
    p = horner(p, r, 1.0 / 362880.0);
    p = horner(p, r, 1.0 / 40320.0);
    p = horner(p, r, 1.0 / 5040.0);
    p = horner(p, r, 1.0 / 720.0);
    p = horner(p, r, 1.0 / 120.0);
    p = horner(p, r, 1.0 / 24.0);
    p = horner(p, r, 1.0 / 6.0);
    p = horner(p, r, 0.5);
    p = horner(p, r, 1.0);
    p = horner(p, r, 1.0);
    return select_f64(x <= -745.2, 0.0, ldexp(p, (int)k));           // underflow; a NaN argument has made p NaN
}
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
    r2 = select_f64(r2 < 0.0, 0.0, r2);               // np.clip(r2, 0, inf) (NaN stays NaN)
    // GPy goes r = sqrt(r2) / lengthscale, then r*r.  The round trip through the square root costs ~40 fp64
    // instructions per element and changes r^2 by at most a couple of ulp (far below the 1e-16-level
    // differences between exp() implementations), so the squared scaled distance is formed directly;
    // inv_l2 = 1 / lengthscale^2 is exactly 1 for the reference's lengthscale = 1 (and for ARD, whose inputs
    // are pre-scaled).
    // (exp_nonpositive, not the device library's exp: 24 vector instructions where that one is ~60 with 20 32-bit moves and
    // four VOP2 selects -- a kernel-matrix element cost ~300 cycles of a SIMD, two thirds of them the exponential; 4.7e-16
    // against the library's 2e-16, both far inside what separates two hosts' exp())
    return __dmul_rn(variance, exp_nonpositive(__dmul_rn(-0.5, __dmul_rn(r2, inv_l2))));
}

__device__ __forceinline__ double cephes_erf_small(double x)       // |x| <= 1
{
    const double z = x * x;
    double p = 9.60497373987051638749E0;
    p = horner(p, z, 9.00260197203842689217E1);
    p = horner(p, z, 2.23200534594684319226E3);
    p = horner(p, z, 7.00332514112805075473E3);
    p = horner(p, z, 5.55923013010394962768E4);
    double q = z + 3.35617141647503099647E1;
    q = horner(q, z, 5.21357949780152679795E2);
    q = horner(q, z, 4.59432382970980127987E3);
    q = horner(q, z, 2.26290000613890934246E4);
    q = horner(q, z, 4.92673942608635921086E4);
    return x * p * recip_plain(q);
}
// e = exp(-a^2 / 2), computed by the caller (who needs it for the density)
__device__ __forceinline__ double ndtr_with_exp(double a, double e)
{
    const double SQRTH = 7.07106781186547524401E-1, MAXLOG = 7.09782712893383996843E2;
    if (isnan(a)) return a;
    const double x = a * SQRTH;
    const double z = fabs(x);
    double c;
    if (z < 1.0) {
        // one evaluation for both of cephes' branches below 1: erf(x) = x T / U is odd to the bit, so erf(|x|) = |erf(x)|
        const double erf_x = cephes_erf_small(x);
        if (z < SQRTH) return 0.5 + 0.5 * erf_x;
        c = 1.0 - fabs(erf_x);
    } else if (z * z > MAXLOG) {
        c = 0.0;                                       // cephes: underflow
    } else if (z < 8.0) {
        double p = 2.46196981473530512524E-10;
        p = horner(p, z, 5.64189564831068821977E-1);
        p = horner(p, z, 7.46321056442269912687E0);
        p = horner(p, z, 4.86371970985681366614E1);
        p = horner(p, z, 1.96520832956077098242E2);
        p = horner(p, z, 5.26445194995477358631E2);
        p = horner(p, z, 9.34528527171957607540E2);
        p = horner(p, z, 1.02755188689515710272E3);
        p = horner(p, z, 5.57535335369399327526E2);
        double q = z + 1.32281951154744992508E1;
        q = horner(q, z, 8.67072140885989742329E1);
        q = horner(q, z, 3.54937778887819891062E2);
        q = horner(q, z, 9.75708501743205489753E2);
        q = horner(q, z, 1.82390916687909736289E3);
        q = horner(q, z, 2.24633760818710981792E3);
        q = horner(q, z, 1.65666309194161350182E3);
        q = horner(q, z, 5.57535340817727675546E2);
        c = (e * p) * recip_plain(q);
    } else {
        double p = 5.64189583547755073984E-1;
        p = horner(p, z, 1.27536670759978104416E0);
        p = horner(p, z, 5.01905042251180477414E0);
        p = horner(p, z, 6.16021097993053585195E0);
        p = horner(p, z, 7.40974269950448939160E0);
        p = horner(p, z, 2.97886665372100240670E0);
        double q = z + 2.26052863220117276590E0;
        q = horner(q, z, 9.39603524938001434673E0);
        q = horner(q, z, 1.20489539808096656605E1);
        q = horner(q, z, 1.70814450747565897222E1);
        q = horner(q, z, 9.60896809063285878198E0);
        q = horner(q, z, 3.36907645100081516050E0);
        c = (e * p) * recip_plain(q);
    }
    const double y = 0.5 * c;
    return select_f64(x > 0, 1.0 - y, y);
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
    var = select_f64(var < kGpyVarClip, kGpyVarClip, var);          // np.clip(var, 1e-15, inf); NaN stays NaN
    if (p.include_noise) var = var + p.noise_var;                   // Gaussian likelihood predictive_values
    mean = mu;
    if (causal) mean = mean + pm;                                   // GP._raw_predict: mu += mean_function.f(Xnew)
}

// CausalExpectedImprovement.evaluate (causal_acquisition_functions.py:27-43, 77-88) / Cost (emukit Quotient):
// s (u Phi(u) + phi(u)) / cost, task 'max' returns -EI with the same u (reference quirk).
__device__ __forceinline__ double acquisition_of(double mean, double var, const AcqParams &p)
{
#pragma clang fp contract(off)
    double s, rs;
    bool special;
    sqrt_and_reciprocal(var, s, rs, special);
    const double mj = mean + p.ei_jitter;
    const double a = p.y_best - mj;
    double u = a * rs;
    u = fma(fma(-u, s, a), rs, u);
    if (__builtin_expect(special, 0)) u = a / s;                    // (a zero or infinite variance: the IEEE quotient)
    const double e = exp_nonpositive(-(u * u) / 2.0);
    const double pdf = e * 0.3989422804014327;                     // scipy _norm_pdf: exp(-x**2/2)/sqrt(2 pi), to an ulp
    const double cdf = ndtr_with_exp(u, e);
    double imp = s * (u * cdf + pdf);
    if (p.task != CBO_TASK_MIN) imp = -imp;
    // imp / cost, correctly rounded, in three instructions: the cost is the same for every candidate of the launch, so its
    // reciprocal is computed once (hoisted out of the candidate loop) and the quotient corrected by its own remainder
    // (Markstein: RN(q + r rem) with r = RN(1 / c) and q within an ulp is the IEEE quotient for every cost whose significand
    // is not all ones -- the reference's costs are small integers; only the sign of a zero acquisition can differ)
    const double rc = 1.0 / p.cost;
    const double qv = imp * rc;
    return fma(fma(-qv, p.cost, imp), rc, qv);
}

}  // namespace cbo
