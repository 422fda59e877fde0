// chub_curves.h -- the EV charge curves (the reference's "FreedomCar" model) for device and host.
//
// What they are: closed-form fits time <-> soc <-> power of a 7 kW-class slow and a 60 kW-class fast
// charger, advanced one 15-minute slot per step (reference: UtilSlow CHS.hpp:467-590, UtilFast
// CHS.hpp:593-726).  The reference evaluates them with float arguments / results and double
// intermediates (pow/exp/log); discrete decisions downstream (ceil of charging time, emergency >= 1.01)
// flip on 1-ulp differences, so the same mixed precision is kept here: f64 arithmetic in the
// reference's association order, one rounding to f32 where the reference has a float.  Integer powers
// are strength-reduced to f64 multiplies (x*x is exact for an f32 x).  Build with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#define CHUB_HD __host__ __device__ __forceinline__

namespace chub {

// x / c for a literal divisor.  Host: the IEEE division itself.  Device: q = x * RN(1/c), one exact residual, one
// correction (Markstein; Brisebarre, Muller, Raina 2004) -- 3 instructions instead of the ~35 of the f64 division
// sequence (~12 for f32).  The sequence returns the correctly rounded quotient except, possibly, for isolated
// divisor-specific dividends; for the divisors used here none turned up in 150 000 (f64) / 40 000 000 (f32) random
// dividends of the shapes the curves produce, and a miss would be one ulp of an intermediate that is rounded to f32 next.
#if defined(__HIP_DEVICE_COMPILE__)
#define CHUB_DIV_K64(x, c) ::chub::div_k64((x), (double) (c), 1.0 / (double) (c))
#define CHUB_DIV_K32(x, c) ::chub::div_k32((x), (float) (c), 1.0f / (float) (c))
__device__ __forceinline__ double div_k64(double x, double c, double rc) {
    const double q = x * rc;
    return __fma_rn(__fma_rn(-q, c, x), rc, q);
}
__device__ __forceinline__ float div_k32(float x, float c, float rc) {
    const float q = x * rc;
    return __fmaf_rn(__fmaf_rn(-q, c, x), rc, q);
}
#else
#define CHUB_DIV_K64(x, c) ((x) / (double) (c))
#define CHUB_DIV_K32(x, c) ((x) / (float) (c))
#endif

// constants the reference recomputes on every call; evaluated once on the host at create
struct CurveConsts {
    float fast_aa1_c;      // (0.7194/0.053)*exp(0)                       CHS.hpp:693
    float fast_aa2_c;      // aa_part1(28.7) - poly(28.7)                 CHS.hpp:698-700
    float fast_soc_scale;  // 100 / aa_part2(51.2)                        CHS.hpp:652,654
    float slow_c2_end;     // c_part2(3.67)                               CHS.hpp:576
};

// ------------------------------------------------------------------------------------ slow (7 kW)
CHUB_HD float slow_b_part1(float xf) {  // CHS.hpp:551-553
    double x = xf, x2 = x * x, x3 = x2 * x, x4 = x2 * x2;
    return (float) (-0.002056 * x4 + 0.00921 * x3 + 0.03562 * x2 + 0.02379 * x + 6.007);
}
CHUB_HD float slow_b_part2(float xf) {  // CHS.hpp:555-557
    double x = xf;
    return (float) ((-4.041 * x + 21.1) / (x - 0.485));
}
CHUB_HD float slow_c_part1(float xf) {  // CHS.hpp:559-562
    double x = xf, x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x;
    return (float) (-0.0004112 * x5 + 0.0023025 * x4 + (0.03562 / 3) * x3 + 0.011895 * x2 + 6.007 * x);
}
CHUB_HD float slow_c_part2(float xf) {  // CHS.hpp:564-566
    double x = xf;
    return (float) (-4.041 * x + 19.140115 * log(fabs(x - 0.485)) + 11.943306312699628);
}
CHUB_HD float slow_c_hole(float x, const CurveConsts &cc) {  // CHS.hpp:568-578
    if (x <= 0) return 0.0f;
    else if ((double) x <= 2.33 * 4) return slow_c_part1(x / 4.0f);
    else if ((double) x <= 3.67 * 4) return slow_c_part2(x / 4.0f);
    else return cc.slow_c2_end;
}
CHUB_HD float slow_time_to_power(float t, bool cp) {  // CHS.hpp:495-511
    if (cp) return (14.68 >= (double) t && t >= 0) ? (float) 5.254973139368931 : 0.0f;
    if ((double) t < 2.33 * 4) return slow_b_part1(t / 4.0f);
    else if ((double) t < 3.67 * 4) return slow_b_part2(t / 4.0f);
    return 0.0f;
}
CHUB_HD float slow_time_to_soc(float t, bool cp, const CurveConsts &cc) {  // CHS.hpp:513-525
    if (cp) {
        if (t <= 0) return 0.0f;
        else if ((double) t >= 14.68) return 100.0f;
        else return (float) CHUB_DIV_K64((double) (100.0f * t), 14.68);
    }
    return (float) CHUB_DIV_K64((double) (100.0f * slow_c_hole(t, cc)), 19.285746346634653);
}
CHUB_HD float slow_soc_to_time(float s, bool cp) {  // CHS.hpp:527-548 (+ s_t_t_1/2 CHS.hpp:580-588)
    if (cp) {
        if (s <= 0) return 0.0f;
        else if (s >= 100) return (float) 14.68;
        else return (float) CHUB_DIV_K64(14.68 * (double) s, 100);
    }
    double x = s;
    if (s < 0) return 0.0f;
    else if (x <= 73.89239629561729) {
        double x2 = x * x;
        return (float) ((-4.276 * 1e-5) * x2 + 0.1295 * x);
    } else if (x <= 100) {
        double x2 = x * x, x3 = x2 * x, x4 = x2 * x2;
        float p = (float) ((4.742 * 1e-6) * x4 - 0.001529 * x3 + 0.1871 * x2 - 10.15 * x + 213.1 + 0.1787983924863248);
        return (float) ((double) p + CHUB_DIV_K64(0.2012016075138625 * (x - 73.89239629561729), 26.10760370438271));
    }
    return (float) 14.68;
}

// ----------------------------------------------------------------------------------- fast (60 kW)
CHUB_HD float fast_a_part1(float xf) {  // CHS.hpp:684-686
    return (float) (0.7194 * exp(0.053 * (double) xf) + 47.78);
}
CHUB_HD float fast_a_part2(float xf) {  // CHS.hpp:688-690
    double x = xf, x2 = x * x, x3 = x2 * x, x4 = x2 * x2;
    return (float) (0.0002253 * x4 - 0.03572 * x3 + 2.016 * x2 - 48.76 * x + 457.7);
}
CHUB_HD float fast_aa_part1(float xf, float aa1_c) {  // CHS.hpp:692-695
    double x = xf;
    return (float) ((0.7194 / 0.053) * exp(0.053 * x) + 50.15 * x - (double) aa1_c);
}
CHUB_HD float fast_aa_part2(float xf, float aa2_c) {  // CHS.hpp:697-704
    double x = xf, x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x;
    return (float) ((0.0002253 / 5) * x5 - (0.03572 / 4) * x4 + (2.016 / 3) * x3 - (48.76 / 2) * x2 + 457.7 * x +
                    (double) aa2_c);
}
CHUB_HD float fast_time_to_power(float t, bool cp) {  // CHS.hpp:621-637
    if (cp) return (3.4133333333333336 >= (double) t && t >= 0) ? (float) 36.44764034125146 : 0.0f;
    if (t >= 0 && (double) t < (28.7 / 15)) return fast_a_part1(t * 15.0f);
    else if (t >= 0 && (double) t < (51.2 / 15)) return fast_a_part2(t * 15.0f);
    return 0.0f;
}
CHUB_HD float fast_time_to_soc(float t, bool cp, const CurveConsts &cc) {  // CHS.hpp:639-659
    if (cp) {
        if (t <= 0) return 0.0f;
        else if ((double) t >= 3.4133333333333336) return 100.0f;
        else return (float) CHUB_DIV_K64((double) (100.0f * t), 3.4133333333333336);
    }
    if (t <= 0) return 0.0f;
    else if ((double) t <= 28.7 / 15) return fast_aa_part1(t * 15.0f, cc.fast_aa1_c) * cc.fast_soc_scale;
    else if ((double) t <= 51.2 / 15) return fast_aa_part2(t * 15.0f, cc.fast_aa2_c) * cc.fast_soc_scale;
    return 100.0f;
}
CHUB_HD float fast_soc_to_time(float s, bool cp) {  // CHS.hpp:661-681 (+ 706-725)
    if (cp) {
        if (s <= 0) return 0.0f;
        else if (s >= 100) return (float) 3.4133333333333336;
        else return (float) CHUB_DIV_K64(3.4133333333333336 * (double) s, 100);
    }
    if (s <= 0) return 0.0f;
    else if (s >= 100) return (float) (51.2 / 15);
    const float mean = 61.43f;  // norm_soc, pure f32: (s - 61.43f) / 31.48f (CHS.hpp:721-725)
    float xn = CHUB_DIV_K32(s - mean, 31.48f);
    const float p1 = -18.18f, p2 = 9.559f, p3 = 48.99f, p4 = -62.97f, p5 = 29.09f;
    const float q1 = -23.9f, q2 = 56.48f, q3 = -50.12f, q4 = 18.96f;
    double x = xn, x2 = x * x, x3 = x2 * x, x4 = x2 * x2;
    double num = (double) p1 * x4 + (double) p2 * x3 + (double) p3 * x2 + (double) (p4 * xn) + (double) p5;
    double den = x4 + (double) q1 * x3 + (double) q2 * x2 + (double) (q3 * xn) + (double) q4;
    return (float) (num / den);
}

// ---------------------------------------------------------------------------- type-generic front
template <int TYPE> CHUB_HD float time_to_power(float t, bool cp) {
    return TYPE == 0 ? fast_time_to_power(t, cp) : slow_time_to_power(t, cp);
}
template <int TYPE> CHUB_HD float time_to_soc(float t, bool cp, const CurveConsts &cc) {
    return TYPE == 0 ? fast_time_to_soc(t, cp, cc) : slow_time_to_soc(t, cp, cc);
}
template <int TYPE> CHUB_HD float soc_to_time(float s, bool cp) {
    return TYPE == 0 ? fast_soc_to_time(s, cp) : slow_soc_to_time(s, cp);
}

// host-only: the per-call constants of the reference, evaluated with the host libm
inline CurveConsts make_curve_consts() {
    CurveConsts c;
    c.fast_aa1_c = (float) ((0.7194 / 0.053) * exp(0.0));
    float aa1_287 = (float) ((0.7194 / 0.053) * exp(0.053 * (double) (float) 28.7) + 50.15 * (double) (float) 28.7 -
                             (double) c.fast_aa1_c);
    c.fast_aa2_c = (float) ((double) aa1_287 -
                            ((0.0002253 / 5) * pow(28.7, 5) - (0.03572 / 4) * pow(28.7, 4) + (2.016 / 3) * pow(28.7, 3) -
                             (48.76 / 2) * pow(28.7, 2) + 457.7 * 28.7));
    float x = (float) 51.2;
    float aa2 = (float) ((0.0002253 / 5) * pow((double) x, 5) - (0.03572 / 4) * pow((double) x, 4) +
                         (2.016 / 3) * pow((double) x, 3) - (48.76 / 2) * pow((double) x, 2) + 457.7 * (double) x +
                         (double) c.fast_aa2_c);
    c.fast_soc_scale = 100.0f / aa2;
    float y = (float) 3.67;
    c.slow_c2_end = (float) (-4.041 * (double) y + 19.140115 * log(fabs((double) y - 0.485)) + 11.943306312699628);
    return c;
}

}  // namespace chub
