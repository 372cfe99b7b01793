// TEST INFRASTRUCTURE (CPU oracle) -- not part of the shipped library.  PARITY UNPINNED: the reference
// holds no test vectors for these passes; the checks are hand-derived known answers and properties
// (tests/test_oracle_kat.py).
//
// CPU restatement of the reference's three compute passes around the path tracer:
//   PostProcessing.glsl            exposure, ACES fit, gamma 2.2, RGBA8 store
//   EquirectangularToCubeMap.glsl  lat-long radiance image -> cube map (RGBA16F)
//   PreethamSky.glsl               analytic sky -> cube map (RGBA32F)
// Floating point follows the rules of glsl.h / det_math.h: every operation written out once,
// no contraction, transcendental functions from det_math.
//
// Where the shader text leaves something to the implementation, the choice made is:
//   * imageStore to rgba8 / rgba16f: clamp to [0,1] then floor(v*255 + 0.5); binary16 round to nearest even
//   * texture() on the equirectangular image: bilinear, level 0, repeat in u, clamp to edge in v
//     (the sampler belongs to the absent engine)
//   * acos outside [-1,1]: argument clamped
#pragma once
#include "glsl.h"
#include <cstdint>

namespace orc {

// PostProcessing.glsl:15-31
static inline vec3 ACESTonemap(vec3 color)
{
    // mat3 constructors are column-major: m1 * color = col0*r + col1*g + col2*b
    vec3 v = V3(0.59719f * color.x + 0.35458f * color.y + 0.04823f * color.z,
                0.07600f * color.x + 0.90834f * color.y + 0.01566f * color.z,
                0.02840f * color.x + 0.13383f * color.y + 0.83777f * color.z);
    vec3 a = v * (v + 0.0245786f) - V3(0.000090537f);
    vec3 b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    vec3 q = a / b;
    vec3 r = V3(1.60475f * q.x + -0.53108f * q.y + -0.07367f * q.z,
                -0.10208f * q.x + 1.10813f * q.y + -0.00605f * q.z,
                -0.00327f * q.x + -0.07276f * q.y + 1.07602f * q.z);
    return V3(gclamp(r.x, 0.0f, 1.0f), gclamp(r.y, 0.0f, 1.0f), gclamp(r.z, 0.0f, 1.0f));
}

// PostProcessing.glsl:33-36
static inline vec3 GammaCorrect(vec3 color, float gamma)
{
    float e = 1.0f / gamma;
    return V3(det_pow(color.x, e), det_pow(color.y, e), det_pow(color.z, e));
}

static inline uint8_t unorm8(float v)
{
    if (!(v > 0.0f)) return 0;              // also NaN
    if (v >= 1.0f) return 255;
    return (uint8_t)(int)(v * 255.0f + 0.5f);
}

// PostProcessing.glsl:39-55, one pixel
static inline void PostProcessPixel(const float* rgba_in, float exposure, uint8_t* rgba_out)
{
    vec3 p = V3(rgba_in[0], rgba_in[1], rgba_in[2]);
    p = p * exposure;
    p = ACESTonemap(p);
    p = GammaCorrect(p, 2.2f);
    rgba_out[0] = unorm8(p.x); rgba_out[1] = unorm8(p.y); rgba_out[2] = unorm8(p.z); rgba_out[3] = 255;
}

// EquirectangularToCubeMap.glsl:11-25 and PreethamSky.glsl:17-31 (same function)
static inline vec3 GetCubeMapTexCoord(uint32_t gx, uint32_t gy, uint32_t face, float size)
{
    float stx = (float)gx / size, sty = (float)gy / size;
    float ux = 2.0f * stx - 1.0f, uy = 2.0f * (1.0f - sty) - 1.0f;
    vec3 ret;
    if (face == 0) ret = V3(1.0f, uy, -ux);
    else if (face == 1) ret = V3(-1.0f, uy, ux);
    else if (face == 2) ret = V3(ux, 1.0f, -uy);
    else if (face == 3) ret = V3(ux, -1.0f, uy);
    else if (face == 4) ret = V3(ux, uy, 1.0f);
    else ret = V3(-ux, uy, -1.0f);
    return normalize(ret);
}

// texture(sampler2D, uv) on a w x h RGBA32F image: bilinear, repeat in u, clamp in v
static inline vec4 SampleEquirect(const float* img, uint32_t w, uint32_t h, float u, float v)
{
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    if (!(x0f == x0f)) x0f = 0.0f;
    if (!(y0f == y0f)) y0f = 0.0f;
    // wrap x into [0, w): inputs are within a few periods of it
    float wf = (float)w;
    float xw0 = x0f - floorf(x0f / wf) * wf, x1f = x0f + 1.0f, xw1 = x1f - floorf(x1f / wf) * wf;
    int x0 = (int)xw0, x1 = (int)xw1;
    if (x0 < 0) x0 = 0;
    if (x0 >= (int)w) x0 = (int)w - 1;
    if (x1 < 0) x1 = 0;
    if (x1 >= (int)w) x1 = (int)w - 1;
    float top = (float)(h - 1);
    int y0 = y0f < 0.0f ? 0 : (y0f > top ? (int)h - 1 : (int)y0f);
    int y1 = (y0f + 1.0f) < 0.0f ? 0 : ((y0f + 1.0f) > top ? (int)h - 1 : (int)(y0f + 1.0f));
    const float* c00 = img + ((size_t)y0 * w + x0) * 4;
    const float* c10 = img + ((size_t)y0 * w + x1) * 4;
    const float* c01 = img + ((size_t)y1 * w + x0) * 4;
    const float* c11 = img + ((size_t)y1 * w + x1) * 4;
    vec4 r;
    r.x = mix(mix(c00[0], c10[0], fx), mix(c01[0], c11[0], fx), fy);
    r.y = mix(mix(c00[1], c10[1], fx), mix(c01[1], c11[1], fx), fy);
    r.z = mix(mix(c00[2], c10[2], fx), mix(c01[2], c11[2], fx), fy);
    r.w = mix(mix(c00[3], c10[3], fx), mix(c01[3], c11[3], fx), fy);
    return r;
}

// EquirectangularToCubeMap.glsl:28-40, one texel; the image is rgba16f
static inline void EquirectToCubeTexel(const float* img, uint32_t w, uint32_t h, uint32_t gx, uint32_t gy, uint32_t face, uint32_t n, float* out)
{
    const float PI = 3.141592f;
    vec3 tc = GetCubeMapTexCoord(gx, gy, face, (float)n);
    float phi = det_atan2(tc.z, tc.x);
    float theta = det_acos(tc.y);
    float u = phi / (2.0f * PI) + 0.5f, v = theta / PI;
    vec4 c = SampleEquirect(img, w, h, u, v);
    out[0] = round_to_half(c.x); out[1] = round_to_half(c.y); out[2] = round_to_half(c.z); out[3] = round_to_half(c.w);
}

// ---- PreethamSky.glsl -------------------------------------------------------------------
#define ORC_SKY_PI 3.14159265359f     /* the #define at PreethamSky.glsl:33 shadows the const above it */

static inline float saturatedDot(vec3 a, vec3 b) { return gmax(dot(a, b), 0.0f); }

// PreethamSky.glsl:40-50
static inline vec3 YxyToXYZ(vec3 Yxy)
{
    float Y = Yxy.x, x = Yxy.y, y = Yxy.z;
    float X = x * (Y / y);
    float Z = (1.0f - x - y) * (Y / y);
    return V3(X, Y, Z);
}

// PreethamSky.glsl:52-63: XYZ * M = dot(XYZ, column j)
static inline vec3 XYZToRGB(vec3 v)
{
    return V3(v.x * 2.3706743f + v.y * -0.9000405f + v.z * -0.4706338f,
              v.x * -0.5138850f + v.y * 1.4253036f + v.z * 0.0885814f,
              v.x * 0.0052982f + v.y * -0.0146949f + v.z * 1.0093968f);
}

// PreethamSky.glsl:74-81
static inline void calculatePerezDistribution(float t, vec3& A, vec3& B, vec3& C, vec3& D, vec3& E)
{
    A = V3(0.1787f * t - 1.4630f, -0.0193f * t - 0.2592f, -0.0167f * t - 0.2608f);
    B = V3(-0.3554f * t + 0.4275f, -0.0665f * t + 0.0008f, -0.0950f * t + 0.0092f);
    C = V3(-0.0227f * t + 5.3251f, -0.0004f * t + 0.2125f, -0.0079f * t + 0.2102f);
    D = V3(0.1206f * t - 2.5771f, -0.0641f * t - 0.8989f, -0.0441f * t - 1.6537f);
    E = V3(-0.0670f * t + 0.3703f, -0.0033f * t + 0.0452f, -0.0109f * t + 0.0529f);
}

// PreethamSky.glsl:83-105
static inline vec3 calculateZenithLuminanceYxy(float t, float thetaS)
{
    float chi = (4.0f / 9.0f - t / 120.0f) * (ORC_SKY_PI - 2.0f * thetaS);
    float Yz = (4.0453f * t - 4.9710f) * det_tan(chi) - 0.2155f * t + 2.4192f;
    float theta2 = thetaS * thetaS;
    float theta3 = theta2 * thetaS;
    float T = t, T2 = t * t;
    float xz = (0.00165f * theta3 - 0.00375f * theta2 + 0.00209f * thetaS + 0.0f) * T2 +
               (-0.02903f * theta3 + 0.06377f * theta2 - 0.03202f * thetaS + 0.00394f) * T +
               (0.11693f * theta3 - 0.21196f * theta2 + 0.06052f * thetaS + 0.25886f);
    float yz = (0.00275f * theta3 - 0.00610f * theta2 + 0.00317f * thetaS + 0.0f) * T2 +
               (-0.04214f * theta3 + 0.08970f * theta2 - 0.04153f * thetaS + 0.00516f) * T +
               (0.15346f * theta3 - 0.26756f * theta2 + 0.06670f * thetaS + 0.26688f);
    return V3(Yz, xz, yz);
}

// PreethamSky.glsl:107-110
static inline vec3 calculatePerezLuminanceYxy(float theta, float gamma, vec3 A, vec3 B, vec3 C, vec3 D, vec3 E)
{
    float ct = det_cos(theta), cg = det_cos(gamma);
    vec3 e1 = V3(det_exp(B.x / ct), det_exp(B.y / ct), det_exp(B.z / ct));
    vec3 e2 = V3(det_exp(D.x * gamma), det_exp(D.y * gamma), det_exp(D.z * gamma));
    return (A * e1 + 1.0f) * ((C * e2 + 1.0f) + E * cg * cg);
}

// PreethamSky.glsl:112-130
static inline vec3 calculateSkyLuminanceRGB(vec3 s, vec3 e, float t)
{
    vec3 A, B, C, D, E;
    calculatePerezDistribution(t, A, B, C, D, E);
    float thetaS = det_acos(saturatedDot(s, V3(0.0f, 1.0f, 0.0f)));
    float thetaE = det_acos(saturatedDot(e, V3(0.0f, 1.0f, 0.0f)));
    float gammaE = det_acos(saturatedDot(s, e));
    vec3 Yz = calculateZenithLuminanceYxy(t, thetaS);
    vec3 fThetaGamma = calculatePerezLuminanceYxy(thetaE, gammaE, A, B, C, D, E);
    vec3 fZeroThetaS = calculatePerezLuminanceYxy(0.0f, thetaS, A, B, C, D, E);
    vec3 Yp = Yz * (fThetaGamma / fZeroThetaS);
    return XYZToRGB(YxyToXYZ(Yp));
}

// PreethamSky.glsl:133-146, one texel
static inline void PreethamTexel(float turbidity, float azimuth, float inclination, uint32_t gx, uint32_t gy, uint32_t face, uint32_t n, float* out)
{
    vec3 tc = GetCubeMapTexCoord(gx, gy, face, (float)n);
    float si = det_sin(inclination), ci = det_cos(inclination), sa = det_sin(azimuth), ca = det_cos(azimuth);
    vec3 sunDir = normalize(V3(si * ca, ci, si * sa));
    vec3 sky = calculateSkyLuminanceRGB(sunDir, tc, turbidity);
    out[0] = sky.x * 0.05f; out[1] = sky.y * 0.05f; out[2] = sky.z * 0.05f; out[3] = 1.0f;
}

} // namespace orc
