// TEST INFRASTRUCTURE (CPU oracle) -- not part of the shipped library.
//
// Deterministic fp32 transcendental functions.  The reference shaders call the
// GLSL built-ins sin, cos, log, pow (PathTracer/assets/shaders/RayTracing/
// Globals.h:68-70, Sampling.glsl:31,41-44,80-81,152-153).  GLSL leaves their
// rounding to the driver, so no two implementations agree bit for bit.  To make
// the CPU oracle and the GPU kernels comparable bit for bit, both sides evaluate
// these functions with the SAME sequence of IEEE-754 binary32 operations
// (+, -, *, /, sqrt, fma: all correctly rounded on x86-64 SSE/FMA and on gfx950).
// This header is the oracle's statement of that sequence ("det_math v1"):
//
//   sincos : q = rne(x * 2/pi) by the 1.5*2^23 trick; 3-term Cody-Waite reduction
//            with fma; degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4];
//            quadrant select.  Defined for |x| < 8192; anything else gives the quiet NaN 0x7fc00000.
//   log    : x = m * 2^k, m in [sqrt(1/2), sqrt(2)); f = m-1; s = f/(2+f);
//            ln m = f - f^2/2 + s*(f^2/2 + R(s^2)); + k*ln2 in two parts.
//   exp    : k = rne(x / ln2); r = x - k*ln2 (two-part, fma); degree-6 polynomial;
//            scaling by 2^k in two exact steps.
//   pow    : exp(y * log(x))   (GLSL: pow(x,y) undefined for x < 0).
//   atan   : |x| > tan(3pi/8): pi/2 - atan(1/|x|); |x| > tan(pi/8): pi/4 + atan((|x|-1)/(|x|+1));
//            odd degree-9 polynomial on the rest (the single-precision Cephes recipe).
//   atan2  : atan(|y/x|), reflected to pi - t for x < 0, negated for y < 0; atan2(0, 0) = 0.
//   acos   : atan2(sqrt((1-x)(1+x)), x) with x clamped to [-1, 1]  (GLSL: undefined outside).
//   tan    : sin / cos.
//   half   : binary32 -> binary16 -> binary32 with round-to-nearest-even, by integer arithmetic.
//
// Accuracy (checked in tests/test_oracle_kat.py against float64 libm): <= 2 ulp for
// sin/cos on [0, 2*pi], log on (0, 1e30), exp on [-87, 88].
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// GLSL min/max/clamp: "max returns y if x < y, otherwise x".
static inline float gmax(float x, float y) { return x < y ? y : x; }
static inline float gmin(float x, float y) { return y < x ? y : x; }
static inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }

static inline void det_sincos(float x, float* s_out, float* c_out)
{
    const float TWO_OVER_PI = 0x1.45f306p-1f;
    const float MAGIC = 12582912.0f;              // 1.5 * 2^23: (v + MAGIC) - MAGIC = rne(v)
    const float P1 = 0x1.921p+0f;                  // pi/2 = P1 + P2 + P3 (P1, P2: 13 bits)
    const float P2 = 0x1.f6ap-13f;
    const float P3 = 0x1.110b46p-26f;
    if (!(u2f(f2u(x) & 0x7fffffffu) < 8192.0f)) {   // outside the reduction's range, inf or NaN: one fixed quiet NaN
        *s_out = *c_out = u2f(0x7fc00000u);
        return;
    }
    float qf = (x * TWO_OVER_PI + MAGIC) - MAGIC;
    float r = fmaf(qf, -P1, x);
    r = fmaf(qf, -P2, r);
    r = fmaf(qf, -P3, r);
    int q = (int)qf;
    float z = r * r;
    // sin(r) = r + r*z*(S1 + z*(S2 + z*S3))
    float sp = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(z, sp, -1.6666654611e-1f);
    float sn = fmaf(r * z, sp, r);
    // cos(r) = 1 - z/2 + z*z*(C1 + z*(C2 + z*C3))
    float cp = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(z, cp, 4.166664568298827e-2f);
    float cs = fmaf(z * z, cp, fmaf(z, -0.5f, 1.0f));
    float s = (q & 1) ? cs : sn;
    float c = (q & 1) ? sn : cs;
    if (q & 2) s = -s;
    if ((q + 1) & 2) c = -c;
    *s_out = s;
    *c_out = c;
}

static inline float det_sin(float x) { float s, c; det_sincos(x, &s, &c); return s; }
static inline float det_cos(float x) { float s, c; det_sincos(x, &s, &c); return c; }

static inline float det_log(float x)
{
    const float LN2_HI = 0x1.62ep-1f;              // 13 bits: k*LN2_HI exact for |k| < 2^11
    const float LN2_LO = 0x1.0bfbe8p-15f;
    const float L1 = 0.66666662693f, L2 = 0.40000972152f, L3 = 0.28498786688f, L4 = 0.24279078841f;
    uint32_t ix = f2u(x);
    if (x != x) return x;                          // NaN
    if ((ix << 1) == 0) return -INFINITY;          // +-0
    if (ix >> 31) return NAN;                      // negative
    if (ix == 0x7f800000u) return x;               // +inf
    int k = 0;
    if (ix < 0x00800000u) {                        // subnormal: scale by 2^25 (exact)
        x = x * 33554432.0f;
        ix = f2u(x);
        k = -25;
    }
    // m in [sqrt(1/2), sqrt(2)): add (1.0 - sqrt(1/2)) in the exponent/mantissa field
    ix += 0x3f800000u - 0x3f3504f3u;
    k += (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    float m = u2f(ix);
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * fmaf(w, L4, L2);
    float t2 = z * fmaf(w, L3, L1);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    return fmaf(dk, LN2_HI, (f - (hfsq - fmaf(s, hfsq + R, dk * LN2_LO))));
}

static inline float det_exp(float x)
{
    const float INV_LN2 = 0x1.715476p+0f;
    const float MAGIC = 12582912.0f;
    const float LN2_HI = 0x1.62ep-1f;
    const float LN2_LO = 0x1.0bfbe8p-15f;
    if (x != x) return x;
    if (x > 88.72283f) return INFINITY;
    if (x < -103.98f) return 0.0f;
    float kf = (x * INV_LN2 + MAGIC) - MAGIC;
    float r = fmaf(kf, -LN2_HI, x);
    r = fmaf(kf, -LN2_LO, r);
    float p = fmaf(r, 1.9875691500e-4f, 1.3981999507e-3f);
    p = fmaf(r, p, 8.3334519073e-3f);
    p = fmaf(r, p, 4.1665795894e-2f);
    p = fmaf(r, p, 1.6666665459e-1f);
    p = fmaf(r, p, 5.0000001201e-1f);
    float e = fmaf(r * r, p, r) + 1.0f;
    int k = (int)kf;                               // in [-150, 128]
    // 2^k in two exact factors so that subnormal results round once, in the last multiply
    int k1 = k / 2, k2 = k - k1;
    float s1 = u2f((uint32_t)(k1 + 127) << 23);
    float s2 = u2f((uint32_t)(k2 + 127) << 23);
    return (e * s1) * s2;
}

static inline float det_pow(float x, float y) { return det_exp(y * det_log(x)); }

static inline float det_atan(float xx)
{
    const float PIO2 = 1.5707963267948966f, PIO4 = 0.7853981633974483f;
    if (xx != xx) return xx;
    float x = u2f(f2u(xx) & 0x7fffffffu);
    float y = 0.0f;
    if (x > 2.414213562373095f) { y = PIO2; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = PIO4; x = (x - 1.0f) / (x + 1.0f); }
    float z = x * x;
    float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
    p = fmaf(z, p, 1.99777106478e-1f);
    p = fmaf(z, p, -3.33329491539e-1f);
    y = y + fmaf(p * z, x, x);
    return (f2u(xx) >> 31) ? -y : y;
}

static inline float det_atan2(float y, float x)
{
    const float PI_F = 3.14159265358979323846f;
    if (y != y || x != x) return u2f(0x7fc00000u);
    float ay = u2f(f2u(y) & 0x7fffffffu), ax = u2f(f2u(x) & 0x7fffffffu);
    if (ay == 0.0f && ax == 0.0f) return 0.0f;
    float t = (ay == __builtin_inff() && ax == __builtin_inff()) ? 0.7853981633974483f : det_atan(ay / ax);   // inf/inf would be a NaN whose sign differs between machines
    if (f2u(x) >> 31) t = PI_F - t;
    return (f2u(y) >> 31) ? -t : t;
}

static inline float det_acos(float x)
{
    if (x != x) return x;
    x = gclamp(x, -1.0f, 1.0f);
    return det_atan2(std::sqrt((1.0f - x) * (1.0f + x)), x);
}

static inline float det_tan(float x) { float s, c; det_sincos(x, &s, &c); return s / c; }

// value of x after a round trip through IEEE binary16 (round to nearest even; overflow to infinity)
static inline float round_to_half(float f)
{
    uint32_t u = f2u(f), sign = u & 0x80000000u, a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return f;                                // inf, NaN
    if (a >= 0x477ff000u) return u2f(sign | 0x7f800000u);          // >= 65520 rounds to infinity
    if (a < 0x33000001u) return u2f(sign);                          // <= 2^-25 rounds to zero
    if (a < 0x38800000u) {                                          // subnormal half: multiples of 2^-24
        float scaled = u2f(a) * 16777216.0f;                        // exact
        float r = (scaled + 12582912.0f) - 12582912.0f;             // rne to integer
        return u2f(f2u(r * 5.9604644775390625e-8f) | sign);
    }
    uint32_t rem = a & 0x1fffu, base = a & ~0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (base & 0x2000u))) base += 0x2000u;
    return u2f(sign | base);
}

} // namespace orc
