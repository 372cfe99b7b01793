// Device-side fp32 helpers for the gfx950 path-tracing kernels.
//
// Every result that reaches a pixel is computed with correctly rounded IEEE-754
// binary32 operations in one fixed order (this translation unit is built with
// -ffp-contract=off; fused multiply-adds appear only where fmaf() is written
// out).  sin, cos, log, exp and pow are evaluated by the fixed operation
// sequences below instead of the device library, so that a CPU restatement using
// the same sequences reproduces the kernels' output bit for bit:
//   sincos : q = rne(x * 2/pi) via the 1.5*2^23 trick, three-term Cody-Waite
//            reduction with fma, degree-7/8 polynomials on [-pi/4, pi/4]; |x| >= 8192, inf
//            and NaN give the quiet NaN 0x7fc00000
//   log    : x = m * 2^k, m in [sqrt(1/2), sqrt(2)), s = f/(2+f) series
//   exp    : k = rne(x / ln2), two-part reduction, degree-6 polynomial, 2^k in two
//            exact factors
//   pow    : exp(y * log(x))
// They stand in for the GLSL built-ins the reference shaders call
// (PathTracer/assets/shaders/RayTracing/Globals.h:68-70, Sampling.glsl:31,41-44,
// 80-81,152-153), whose rounding GLSL leaves to the driver.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__
#define PT_PI 3.14159265358979323f
#define PT_INV_PI 0.31830988618379067f
#define PT_TWO_PI 6.28318530717958648f

namespace ptd {

PT_DEV uint32_t f2u(float f) { return __float_as_uint(f); }
PT_DEV float u2f(uint32_t u) { return __uint_as_float(u); }

// GLSL semantics: max(x,y) = x < y ? y : x ; min(x,y) = y < x ? y : x
PT_DEV float gmax(float x, float y) { return x < y ? y : x; }
PT_DEV float gmin(float x, float y) { return y < x ? y : x; }
PT_DEV float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
PT_DEV float gabs(float x) { return u2f(f2u(x) & 0x7fffffffu); }
PT_DEV float fsqrt(float x) { return __builtin_sqrtf(x); }   // correctly rounded (__fsqrt_rn is the 1-ulp v_sqrt_f32)
PT_DEV float fdiv(float a, float b) { return a / b; }               // correctly rounded IEEE division

PT_DEV void det_sincos(float x, float& s_out, float& c_out)
{
    const float TWO_OVER_PI = 0x1.45f306p-1f;
    const float MAGIC = 12582912.0f;
    const float P1 = 0x1.921p+0f;
    const float P2 = 0x1.f6ap-13f;
    const float P3 = 0x1.110b46p-26f;
    if (!(gabs(x) < 8192.0f)) {   // outside the reduction's range, inf or NaN: one fixed quiet NaN
        s_out = c_out = u2f(0x7fc00000u);
        return;
    }
    float qf = (x * TWO_OVER_PI + MAGIC) - MAGIC;
    float r = fmaf(qf, -P1, x);
    r = fmaf(qf, -P2, r);
    r = fmaf(qf, -P3, r);
    int q = (int)qf;
    float z = r * r;
    float sp = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(z, sp, -1.6666654611e-1f);
    float sn = fmaf(r * z, sp, r);
    float cp = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(z, cp, 4.166664568298827e-2f);
    float cs = fmaf(z * z, cp, fmaf(z, -0.5f, 1.0f));
    float s = (q & 1) ? cs : sn;
    float c = (q & 1) ? sn : cs;
    if (q & 2) s = -s;
    if ((q + 1) & 2) c = -c;
    s_out = s;
    c_out = c;
}

PT_DEV float det_log(float x)
{
    const float LN2_HI = 0x1.62ep-1f;
    const float LN2_LO = 0x1.0bfbe8p-15f;
    const float L1 = 0.66666662693f, L2 = 0.40000972152f, L3 = 0.28498786688f, L4 = 0.24279078841f;
    uint32_t ix = f2u(x);
    if (x != x) return x;
    if ((ix << 1) == 0) return -__builtin_inff();
    if (ix >> 31) return __builtin_nanf("");
    if (ix == 0x7f800000u) return x;
    int k = 0;
    if (ix < 0x00800000u) {
        x = x * 33554432.0f;
        ix = f2u(x);
        k = -25;
    }
    ix += 0x3f800000u - 0x3f3504f3u;
    k += (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    float m = u2f(ix);
    float f = m - 1.0f;
    float s = fdiv(f, 2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * fmaf(w, L4, L2);
    float t2 = z * fmaf(w, L3, L1);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    return fmaf(dk, LN2_HI, (f - (hfsq - fmaf(s, hfsq + R, dk * LN2_LO))));
}

PT_DEV float det_exp(float x)
{
    const float INV_LN2 = 0x1.715476p+0f;
    const float MAGIC = 12582912.0f;
    const float LN2_HI = 0x1.62ep-1f;
    const float LN2_LO = 0x1.0bfbe8p-15f;
    if (x != x) return x;
    if (x > 88.72283f) return __builtin_inff();
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
    int k = (int)kf;
    int k1 = k / 2, k2 = k - k1;
    float s1 = u2f((uint32_t)(k1 + 127) << 23);
    float s2 = u2f((uint32_t)(k2 + 127) << 23);
    return (e * s1) * s2;
}

PT_DEV float det_pow(float x, float y) { return det_exp(y * det_log(x)); }

// atan: |x| > tan(3pi/8): pi/2 - atan(1/|x|); |x| > tan(pi/8): pi/4 + atan((|x|-1)/(|x|+1)); odd polynomial on the rest
PT_DEV float det_atan(float xx)
{
    const float PIO2 = 1.5707963267948966f, PIO4 = 0.7853981633974483f;
    if (xx != xx) return xx;
    float x = u2f(f2u(xx) & 0x7fffffffu);
    float y = 0.0f;
    if (x > 2.414213562373095f) { y = PIO2; x = -fdiv(1.0f, x); }
    else if (x > 0.4142135623730950f) { y = PIO4; x = fdiv(x - 1.0f, x + 1.0f); }
    float z = x * x;
    float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
    p = fmaf(z, p, 1.99777106478e-1f);
    p = fmaf(z, p, -3.33329491539e-1f);
    y = y + fmaf(p * z, x, x);
    return (f2u(xx) >> 31) ? -y : y;
}

PT_DEV float det_atan2(float y, float x)
{
    const float PI_F = 3.14159265358979323846f;
    if (y != y || x != x) return u2f(0x7fc00000u);
    float ay = u2f(f2u(y) & 0x7fffffffu), ax = u2f(f2u(x) & 0x7fffffffu);
    if (ay == 0.0f && ax == 0.0f) return 0.0f;
    float t = (ay == __builtin_inff() && ax == __builtin_inff()) ? 0.7853981633974483f : det_atan(fdiv(ay, ax));   // inf/inf would be a NaN whose sign differs between machines
    if (f2u(x) >> 31) t = PI_F - t;
    return (f2u(y) >> 31) ? -t : t;
}

// GLSL leaves acos outside [-1, 1] undefined: the argument is clamped
PT_DEV float det_acos(float x)
{
    if (x != x) return x;
    x = gclamp(x, -1.0f, 1.0f);
    return det_atan2(fsqrt((1.0f - x) * (1.0f + x)), x);
}

PT_DEV float det_tan(float x) { float s, c; det_sincos(x, s, c); return fdiv(s, c); }

// value of x after a round trip through IEEE binary16 (round to nearest even; overflow to infinity)
PT_DEV float round_to_half(float f)
{
    uint32_t u = f2u(f), sign = u & 0x80000000u, a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return f;
    if (a >= 0x477ff000u) return u2f(sign | 0x7f800000u);
    if (a < 0x33000001u) return u2f(sign);
    if (a < 0x38800000u) {
        float scaled = u2f(a) * 16777216.0f;
        float r = (scaled + 12582912.0f) - 12582912.0f;
        return u2f(f2u(r * 5.9604644775390625e-8f) | sign);
    }
    uint32_t rem = a & 0x1fffu, base = a & ~0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (base & 0x2000u))) base += 0x2000u;
    return u2f(sign | base);
}

// ---- 3-vectors: v3, its operators, dot, cross, normalize, mix, reflect, refract.  The text is included a second time inside
// namespace ptd::fast (pt_fast.h), where fdiv, fsqrt and inversesqrt are the single-instruction approximations
#define PT_INVERSESQRT(x) fdiv(1.0f, fsqrt(x))
#include "pt_vec.inc"
#undef PT_INVERSESQRT

// ---- RNG (PathTracer/assets/shaders/RayTracing/Globals.h:54-64) ----------------
PT_DEV uint32_t pcg_hash(uint32_t& seed)
{
    seed = seed * 747796405u + 2891336453u;
    uint32_t result = ((seed >> ((seed >> 28) + 4u)) ^ seed) * 277803737u;
    return (result >> 22) ^ result;
}
// float(hash) / 4294967295.0: the divisor is a 32-bit float literal = 2^32, an exact scaling
PT_DEV float random_value(uint32_t& seed) { return (float)pcg_hash(seed) * 0x1p-32f; }

} // namespace ptd
