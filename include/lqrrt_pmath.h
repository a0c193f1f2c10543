/*
 * lqrrt_pmath.h -- portable, bit-reproducible sin / cos / atan2 / tanh in IEEE-754 double.
 *
 * Why this exists: the reference's problem plugins call np.sin / np.cos / np.arctan2, whose last
 * bit depends on the host (glibc vs SVML/AVX-512 dispatch inside NumPy), and the boat problem's
 * "magic rudder" (demos/demo_boat_advanced.py:101-111) amplifies one-ulp differences by up to
 * ~1e2-1e3 per step once the boat is nearly stopped with saturated thrusters -- see DESIGN.md
 * "Conditioning".  To be able to prove that the wave-parallel engine reproduces the sequential
 * algorithm EXACTLY (bit-for-bit trees at any size) the device code and the C oracle both
 * evaluate elementary functions through this header, which uses only operations that IEEE-754
 * defines exactly -- + - * / fma floor fabs copysign ldexp and comparisons -- in a fixed order.  The
 * same source therefore gives identical bits on gfx950 (hipcc -ffp-contract=off) and on x86-64
 * (gcc -ffp-contract=off; fma() is correctly rounded by the C standard, in hardware or not).
 *
 * Accuracy (measured against 60-digit mpmath, tests/test_pmath.py): sin, cos <= 1 ulp for
 * |x| <= 1e5; atan2 <= 2 ulp; tanh <= 3 ulp.  Domain: finite arguments; |x| < 1.6e6 for sin/cos (beyond that
 * the 33-bit Cody-Waite product n*P1 is no longer exact and libm is used instead).
 *
 * Method: sin/cos -- reduction x = n*(pi/2) + r by a two-level Cody-Waite split of pi/2 with an
 * explicit tail, Taylor kernels in r^2 (reciprocal factorials; truncation < 1e-19 on |r| <= pi/4).
 * atan2 -- t = min/max in [0,1]; t > tan(pi/8) is folded by atan t = pi/4 + atan((t-1)/(t+1));
 * atan z = z + z^3 q(z^2) with q a degree-10 Chebyshev-node interpolant (relative error 5e-18,
 * coefficients generated with mpmath for this file); octant/sign fix-up with hi/lo pi constants.
 */
#ifndef LQRRT_PMATH_H
#define LQRRT_PMATH_H

#include <math.h>

#if defined(__HIPCC__)
#define LQ_HD __host__ __device__ __forceinline__
#else
#define LQ_HD static inline
#endif

/* One Horner step p*w + c with a single rounding.  On the host this is the C standard's fma().  On gfx950 it is the same
 * operation (v_fma_f64 is the IEEE-754 fusedMultiplyAdd), spelled out: left to itself the compiler emits the two-address form,
 * v_mov_b64 tmp, c ; v_fmac_f64 tmp, p, w -- one extra instruction per coefficient, on a wavefront whose speed IS its
 * instruction count (one instruction per ~6 cycles whatever it is). */
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ double lq_fma(double p, double w, double c) {
    double r;
    __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(w), "v"(c));
    return r;
}
#else
#define lq_fma(p, w, c) fma((p), (w), (c))
#endif

#define LQ_PI_HI     0x1.921fb54442d18p+1
#define LQ_PI_LO     0x1.1a62633145c07p-53
#define LQ_PI_2_HI   0x1.921fb54442d18p+0
#define LQ_PI_2_LO   0x1.1a62633145c07p-54
#define LQ_PI_4_HI   0x1.921fb54442d18p-1
#define LQ_PI_4_LO   0x1.1a62633145c07p-55
#define LQ_2_OVER_PI 0x1.45f306dc9c883p-1
#define LQ_TAN_PI_8  0x1.a827999fcef32p-2
/* pi/2 = P1 + P1T (P1: leading 33 bits), P1T = P2 + P2T (P2: next 33 bits) */
#define LQ_P1        0x1.921fb54400000p+0
#define LQ_P1T       0x1.0b4611a626331p-34
#define LQ_P2        0x1.0b4611a600000p-34
#define LQ_P2T       0x1.3198a2e037073p-69

/* sin and cos of x together */
LQ_HD void lq_sincos(double x, double* sn, double* cs) {
    const double ax = fabs(x);
    if (!(ax < 1.6e6)) {                             /* also NaN and infinities */
        *sn = sin(x);
        *cs = cos(x);
        return;
    }
    /* Straight-line on purpose (selects, no branches): on the GPU different lanes of a wavefront evaluate
     * different arguments, and a data-dependent branch costs an exec-mask round trip even when not taken.
     * |x| <= pi/4 is the n = 0 case of the general reduction (r0 = x - 0, w = 0, r = x, rt = +0). */
    const double fn = (ax <= LQ_PI_4_HI) ? 0.0 : floor(x * LQ_2_OVER_PI + 0.5);
    double r0 = x - fn * LQ_P1;                      /* exact: fn*P1 has <= 53 bits, Sterbenz */
    double w = fn * LQ_P1T;
    double r = r0 - w;
    {
        /* x close to a multiple of pi/2: one level deeper (computed always, selected when needed) */
        const int deep = fabs(r) < fabs(r0) * 0x1p-16;
        const double t = r0;
        const double w2 = fn * LQ_P2;
        const double r0d = t - w2;
        const double wd = fn * LQ_P2T - ((t - r0d) - w2);
        const double rd = r0d - wd;
        r0 = deep ? r0d : r0;
        w = deep ? wd : w;
        r = deep ? rd : r;
    }
    const double rt = (r0 - r) - w;
    const int q = (int)(fn - 4.0 * floor(fn * 0.25));   /* fn mod 4 in {0,1,2,3} */
    const double z = r * r;
    /* sin r = r + r z (s3 + z (s5 + ... + z s17)) */
    double ps = 0x1.952c77030ad4ap-49;                          /* 1/17! */
    ps = lq_fma(ps, z, -0x1.ae7f3e733b81fp-41);                    /* -1/15! */
    ps = lq_fma(ps, z, 0x1.6124613a86d09p-33);                     /* 1/13! */
    ps = lq_fma(ps, z, -0x1.ae64567f544e4p-26);                    /* -1/11! */
    ps = lq_fma(ps, z, 0x1.71de3a556c734p-19);                     /* 1/9! */
    ps = lq_fma(ps, z, -0x1.a01a01a01a01ap-13);                    /* -1/7! */
    ps = lq_fma(ps, z, 0x1.1111111111111p-7);                      /* 1/5! */
    ps = lq_fma(ps, z, -0x1.5555555555555p-3);                     /* -1/3! */
    const double hz = 0.5 * z;
    const double ksin = r + (r * z * ps + rt * (1.0 - hz));
    /* cos r = 1 - z/2 + z^2 (c4 + z (c6 + ... + z c16)) */
    double pc = 0x1.ae7f3e733b81fp-45;                          /* 1/16! */
    pc = lq_fma(pc, z, -0x1.93974a8c07c9dp-37);                    /* -1/14! */
    pc = lq_fma(pc, z, 0x1.1eed8eff8d898p-29);                     /* 1/12! */
    pc = lq_fma(pc, z, -0x1.27e4fb7789f5cp-22);                    /* -1/10! */
    pc = lq_fma(pc, z, 0x1.a01a01a01a01ap-16);                     /* 1/8! */
    pc = lq_fma(pc, z, -0x1.6c16c16c16c17p-10);                    /* -1/6! */
    pc = lq_fma(pc, z, 0x1.5555555555555p-5);                      /* 1/4! */
    const double one_m = 1.0 - hz;
    const double kcos = one_m + (((1.0 - one_m) - hz) + (z * z * pc - r * rt));
    double s = ksin, c = kcos;
    if (q == 1) { s = kcos; c = -ksin; }
    else if (q == 2) { s = -ksin; c = -kcos; }
    else if (q == 3) { s = -kcos; c = ksin; }
    *sn = s;
    *cs = c;
}

LQ_HD double lq_sin(double x) { double s, c; lq_sincos(x, &s, &c); return s; }
LQ_HD double lq_cos(double x) { double s, c; lq_sincos(x, &s, &c); return c; }

/* four-quadrant arctangent, C99 sign conventions for zeros.  One body, two spellings of its Horner steps: lq_atan2 uses lq_fma
 * (on gfx950 the explicit three-address v_fma_f64 above, whose coefficients then live in VGPRs -- right for a rollout wavefront
 * that has the register file to itself), lq_atan2_c leaves the step to the compiler's fma (coefficients as literals / SGPRs --
 * right for the NN scan, whose speed is its occupancy: with the VGPR coefficients the scans that carry an atan2 lost a wavefront
 * per SIMD, profiles/r05_nn_regression.txt).  fma is fma: both return the same bits (host: tests/test_pmath.py; device: every scan with an atan2 is compared bit for bit
 * with the C oracle, which only has the first spelling). */
#define LQ_ATAN2_IMPL(NAME, FMA) \
LQ_HD double NAME(double y, double x) { \
    const double ax = fabs(x), ay = fabs(y); \
    const int xneg = copysign(1.0, x) < 0.0; \
    /* the zero cases are patched in at the end by selects (straight-line code, see lq_sincos); the general \
     * path may then see 0/0, whose NaN is discarded */ \
    const int swap = ay > ax; \
    const double mx = swap ? ay : ax, mn = swap ? ax : ay; \
    const double t = mn / mx; \
    const int big = t > LQ_TAN_PI_8; \
    const double z = big ? (t - 1.0) / (t + 1.0) : t; \
    const double w = z * z; \
    double p = -0x1.3a31a1d5ffde0p-6; \
    p = FMA(p, w, 0x1.4162b9ab69c5ap-5); \
    p = FMA(p, w, -0x1.a0999a234950fp-5); \
    p = FMA(p, w, 0x1.dfe6491089bd5p-5); \
    p = FMA(p, w, -0x1.10fa77ab514f0p-4); \
    p = FMA(p, w, 0x1.3b126305dc4dep-4); \
    p = FMA(p, w, -0x1.745d0b28a2eeep-4); \
    p = FMA(p, w, 0x1.c71c71853d607p-4); \
    p = FMA(p, w, -0x1.24924924361fep-3); \
    p = FMA(p, w, 0x1.999999999934cp-3); \
    p = FMA(p, w, -0x1.5555555555555p-2); \
    const double corr = z * w * p;                               /* atan z - z */ \
    double a = big ? LQ_PI_4_HI + (z + (corr + LQ_PI_4_LO)) : z + corr; \
    a = swap ? LQ_PI_2_HI - (a - LQ_PI_2_LO) : a; \
    a = xneg ? LQ_PI_HI - (a - LQ_PI_LO) : a; \
    double res = copysign(a, y); \
    res = (ax == 0.0) ? copysign(LQ_PI_2_HI, y) : res; \
    res = (ay == 0.0) ? (xneg ? copysign(LQ_PI_HI, y) : y) : res; \
    return res; \
}
LQ_ATAN2_IMPL(lq_atan2, lq_fma)
#if defined(__HIP_DEVICE_COMPILE__)
LQ_ATAN2_IMPL(lq_atan2_c, __builtin_fma)
#else
LQ_ATAN2_IMPL(lq_atan2_c, fma)
#endif

/* ln 2 = LN2_HI + LN2_LO, LN2_HI with 32 significant bits so that k*LN2_HI is exact for |k| < 2^20 */
#define LQ_LN2_HI    0x1.62e42fee00000p-1
#define LQ_LN2_LO    0x1.a39ef35793c76p-33
#define LQ_LOG2E     0x1.71547652b82fep+0

/* hyperbolic tangent.  With y = -2|x| = k ln2 + r (|r| <= ln2/2), p = expm1(r) from its Taylor series
 * (truncation < 3e-19 relative) and s = 2^k:  tanh|x| = (1 - s(1+p)) / (1 + s(1+p)), where both
 * (1 -+ s) -+ s p are formed with ONE rounding each (1 -+ s is exact), so there is no cancellation
 * for small |x| (k = 0: -p / (2 + p)).  <= 3 ulp (tests/test_pmath.py). */
LQ_HD double lq_tanh(double x) {
    const double ax = fabs(x);
    if (!(ax < 22.0)) return x != x ? x : copysign(1.0, x);     /* 1 - tanh 22 < 2^-62 */
    const double y = -2.0 * ax;
    const double k = floor(fma(y, LQ_LOG2E, 0.5));
    double r = fma(-k, LQ_LN2_HI, y);
    r = fma(-k, LQ_LN2_LO, r);
    double q = 0x1.ae7f3e733b81fp-41;                          /* 1/15! */
    q = lq_fma(q, r, 0x1.93974a8c07c9dp-37);                      /* 1/14! */
    q = lq_fma(q, r, 0x1.6124613a86d09p-33);                      /* 1/13! */
    q = lq_fma(q, r, 0x1.1eed8eff8d898p-29);                      /* 1/12! */
    q = lq_fma(q, r, 0x1.ae64567f544e4p-26);                      /* 1/11! */
    q = lq_fma(q, r, 0x1.27e4fb7789f5cp-22);                      /* 1/10! */
    q = lq_fma(q, r, 0x1.71de3a556c734p-19);                      /* 1/9! */
    q = lq_fma(q, r, 0x1.a01a01a01a01ap-16);                      /* 1/8! */
    q = lq_fma(q, r, 0x1.a01a01a01a01ap-13);                      /* 1/7! */
    q = lq_fma(q, r, 0x1.6c16c16c16c17p-10);                      /* 1/6! */
    q = lq_fma(q, r, 0x1.1111111111111p-7);                      /* 1/5! */
    q = lq_fma(q, r, 0x1.5555555555555p-5);                      /* 1/4! */
    q = lq_fma(q, r, 0x1.5555555555555p-3);                      /* 1/3! */
    q = lq_fma(q, r, 0x1.0000000000000p-1);                      /* 1/2! */
    const double p = fma(r * r, q, r);                           /* expm1(r) */
    const double s = ldexp(1.0, (int)k);
    const double num = fma(-s, p, 1.0 - s);
    const double den = fma(s, p, 1.0 + s);
    return copysign(num / den, x);
}

#endif /* LQRRT_PMATH_H */
