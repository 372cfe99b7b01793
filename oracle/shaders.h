// TEST INFRASTRUCTURE (CPU oracle) -- not part of the shipped library.
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
// this path and cannot be built or run here (its engine submodule is absent and
// its shaders need a Vulkan ray-tracing driver), so this restatement is checked
// only against hand-derived known answers (tests/golden) and not against outputs
// of the reference itself.
//
// Line-by-line fp32 restatement of the reference's ray-tracing shaders
// (all paths relative to PathTracer/assets/shaders/RayTracing/):
//   Globals.h:54-71        PCG_Hash, RandomValue, RandomPointInCircle
//   Sampling.glsl:25-47    GTR1, SampleGTR1
//   Sampling.glsl:70-96    SampleGGXVNDF, GTR2Aniso
//   Sampling.glsl:109-156  SmithG, SmithGAniso, SchlickWeight, DielectricFresnel,
//                          CosineSampleHemisphere
//   Sampling.glsl:179-184  Onb
//   Disney.glsl:42-359     ToWorld, ToLocal, Luminance, TintColors, the four lobe
//                          evaluators, DisneySample, DisneyEval
//   Miss.glsl:8-29         miss sentinel
//   ClosestHit.glsl:50-226 surface fetch + payload
//   RayGen.glsl:106-176    TracePath        RayGen.glsl:312-373  main
// sin/cos/log/pow go through det_math.h; everything else is IEEE +,-,*,/,sqrt.
#pragma once
#include "glsl.h"
#include <vector>

namespace orc {

static const float PI = 3.14159265358979323f;
static const float INV_PI = 0.31830988618379067f;
static const float TWO_PI = 6.28318530717958648f;

// ---- Globals.h:54-71 -------------------------------------------------------
static inline uint32_t PCG_Hash(uint32_t& seed)
{
    seed = seed * 747796405u + 2891336453u;
    uint32_t result = ((seed >> ((seed >> 28) + 4u)) ^ seed) * 277803737u;
    return (result >> 22) ^ result;
}

// The literal 4294967295.0 is a 32-bit float in GLSL and rounds to 2^32, and
// float(0xFFFFFFFF) rounds to 2^32 as well, so the result can be exactly 1.0.
static inline float RandomValue(uint32_t& seed) { return (float)PCG_Hash(seed) / 4294967296.0f; }

static inline vec2 RandomPointInCircle(uint32_t& seed)
{
    float angle = RandomValue(seed) * 2.0f * PI;
    float s, c;
    det_sincos(angle, &s, &c);
    float r = sqrtf(RandomValue(seed));
    return vec2{c * r, s * r};
}

// ---- Globals.h:8-36 (fields that anything downstream reads) -----------------
struct Payload {
    float Distance;
    vec3 Albedo;
    float Metallic, Roughness;
    vec3 Emission;
    vec3 WorldNormal;
    float Anisotropic, Subsurface, SpecularTint, Sheen, SheenTint, Clearcoat, ClearcoatRoughness, SpecTrans, ior;
    float ax, ay, eta;
};

// ---- Sampling.glsl ---------------------------------------------------------
static inline float GTR1(float NDotH, float a)
{
    if (a >= 1.0f) return INV_PI;
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return (a2 - 1.0f) / (PI * det_log(a2) * t);
}

static inline vec3 SampleGTR1(float rgh, float r1, float r2)
{
    float a = gmax(0.001f, rgh);
    float a2 = a * a;
    float phi = r1 * TWO_PI;
    float cosTheta = sqrtf((1.0f - det_pow(a2, 1.0f - r2)) / (1.0f - a2));
    float sinTheta = gclamp(sqrtf(1.0f - (cosTheta * cosTheta)), 0.0f, 1.0f);
    float sinPhi, cosPhi;
    det_sincos(phi, &sinPhi, &cosPhi);
    return V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}

static inline vec3 SampleGGXVNDF(vec3 V, float ax, float ay, float r1, float r2)
{
    vec3 Vh = normalize(V3(ax * V.x, ay * V.y, V.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    vec3 T1 = lensq > 0.0f ? V3(-Vh.y, Vh.x, 0.0f) * inversesqrt(lensq) : V3(1.0f, 0.0f, 0.0f);
    vec3 T2 = cross(Vh, T1);
    float r = sqrtf(r1);
    float phi = 2.0f * PI * r2;
    float sphi, cphi;
    det_sincos(phi, &sphi, &cphi);
    float t1 = r * cphi;
    float t2 = r * sphi;
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    vec3 Nh = t1 * T1 + t2 * T2 + sqrtf(gmax(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalize(V3(ax * Nh.x, ay * Nh.y, gmax(0.0f, Nh.z)));
}

static inline float GTR2Aniso(float NDotH, float HDotX, float HDotY, float ax, float ay)
{
    float a = HDotX / ax;
    float b = HDotY / ay;
    float c = a * a + b * b + NDotH * NDotH;
    return 1.0f / (PI * ax * ay * c * c);
}

static inline float SmithG(float NDotV, float alphaG)
{
    float a = alphaG * alphaG;
    float b = NDotV * NDotV;
    return (2.0f * NDotV) / (NDotV + sqrtf(a + b - a * b));
}

static inline float SmithGAniso(float NDotV, float VDotX, float VDotY, float ax, float ay)
{
    float a = VDotX * ax;
    float b = VDotY * ay;
    float c = NDotV;
    return (2.0f * NDotV) / (NDotV + sqrtf(a * a + b * b + c * c));
}

static inline float SchlickWeight(float u)
{
    float m = gclamp(1.0f - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}

static inline float DielectricFresnel(float cosThetaI, float eta)
{
    float sinThetaTSq = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaTSq > 1.0f) return 1.0f;
    float cosThetaT = sqrtf(gmax(1.0f - sinThetaTSq, 0.0f));
    float rs = (eta * cosThetaT - cosThetaI) / (eta * cosThetaT + cosThetaI);
    float rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    return 0.5f * (rs * rs + rp * rp);
}

static inline vec3 CosineSampleHemisphere(float r1, float r2)
{
    vec3 dir;
    float r = sqrtf(r1);
    float phi = TWO_PI * r2;
    float s, c;
    det_sincos(phi, &s, &c);
    dir.x = r * c;
    dir.y = r * s;
    dir.z = sqrtf(gmax(0.0f, 1.0f - dir.x * dir.x - dir.y * dir.y));
    return dir;
}

static inline void Onb(vec3 N, vec3& T, vec3& B)
{
    vec3 up = gabs(N.z) < 0.9999999f ? V3(0.0f, 0.0f, 1.0f) : V3(1.0f, 0.0f, 0.0f);
    T = normalize(cross(up, N));
    B = cross(N, T);
}

// ---- Disney.glsl -----------------------------------------------------------
static inline vec3 ToWorld(vec3 X, vec3 Y, vec3 Z, vec3 V) { return V.x * X + V.y * Y + V.z * Z; }
static inline vec3 ToLocal(vec3 X, vec3 Y, vec3 Z, vec3 V) { return V3(dot(V, X), dot(V, Y), dot(V, Z)); }
static inline float Luminance(vec3 c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; }

static inline void TintColors(const Payload& p, float eta, float& F0, vec3& Csheen, vec3& Cspec0)
{
    float lum = Luminance(p.Albedo);
    vec3 ctint = lum > 0.0f ? p.Albedo / lum : V3(1.0f);
    F0 = (1.0f - eta) / (1.0f + eta);
    F0 *= F0;
    Cspec0 = F0 * mix(V3(1.0f), ctint, p.SpecularTint);
    Csheen = mix(V3(1.0f), ctint, p.SheenTint);
}

static inline vec3 EvalDisneyDiffuse(const Payload& p, vec3 Csheen, vec3 V, vec3 L, vec3 H, float& pdf)
{
    pdf = 0.0f;
    if (L.z <= 0.0f) return V3(0.0f);
    float LDotH = dot(L, H);
    float Rr = 2.0f * p.Roughness * LDotH * LDotH;
    float FL = SchlickWeight(L.z);
    float FV = SchlickWeight(V.z);
    float Fretro = Rr * (FL + FV + FL * FV * (Rr - 1.0f));
    float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
    float Fss90 = 0.5f * Rr;
    float Fss = mix(1.0f, Fss90, FL) * mix(1.0f, Fss90, FV);
    float ss = 1.25f * (Fss * (1.0f / (L.z + V.z) - 0.5f) + 0.5f);
    float FH = SchlickWeight(LDotH);
    vec3 Fsheen = FH * p.Sheen * Csheen;
    pdf = L.z * INV_PI;
    return INV_PI * p.Albedo * mix(Fd + Fretro, ss, p.Subsurface) + Fsheen;
}

static inline vec3 EvalMicrofacetReflection(const Payload& p, vec3 V, vec3 L, vec3 H, vec3 F, float& pdf)
{
    pdf = 0.0f;
    if (L.z <= 0.0f) return V3(0.0f);
    float D = GTR2Aniso(H.z, H.x, H.y, p.ax, p.ay);
    float G1 = SmithGAniso(gabs(V.z), V.x, V.y, p.ax, p.ay);
    float G2 = G1 * SmithGAniso(gabs(L.z), L.x, L.y, p.ax, p.ay);
    pdf = G1 * D / (4.0f * V.z);
    return F * D * G2 / (4.0f * L.z * V.z);
}

static inline vec3 EvalMicrofacetRefraction(const Payload& p, float eta, vec3 V, vec3 L, vec3 H, vec3 F, float& pdf)
{
    pdf = 0.0f;
    if (L.z >= 0.0f) return V3(0.0f);
    float LDotH = dot(L, H);
    float VDotH = dot(V, H);
    float D = GTR2Aniso(H.z, H.x, H.y, p.ax, p.ay);
    float G1 = SmithGAniso(gabs(V.z), V.x, V.y, p.ax, p.ay);
    float G2 = G1 * SmithGAniso(gabs(L.z), L.x, L.y, p.ax, p.ay);
    float denom = LDotH + VDotH * eta;
    denom *= denom;
    float eta2 = eta * eta;
    float jacobian = gabs(LDotH) / denom;
    pdf = G1 * gmax(0.0f, VDotH) * D * jacobian / V.z;
    vec3 sqrtAlbedo = V3(det_pow(p.Albedo.x, 0.5f), det_pow(p.Albedo.y, 0.5f), det_pow(p.Albedo.z, 0.5f));
    return sqrtAlbedo * (1.0f - F) * D * G2 * gabs(VDotH) * jacobian * eta2 / gabs(L.z * V.z);
}

static inline vec3 EvalClearcoat(const Payload& p, vec3 V, vec3 L, vec3 H, float& pdf)
{
    pdf = 0.0f;
    if (L.z <= 0.0f) return V3(0.0f);
    float VDotH = dot(V, H);
    float F = mix(0.04f, 1.0f, SchlickWeight(VDotH));
    float D = GTR1(H.z, p.ClearcoatRoughness);
    float G = SmithG(L.z, 0.25f) * SmithG(V.z, 0.25f);
    float jacobian = 1.0f / (4.0f * VDotH);
    pdf = D * H.z * jacobian;
    return V3(F) * D * G;
}

static inline vec3 DisneyEval(const Payload& p, vec3 V, vec3 N, vec3 L, float& pdf)
{
    pdf = 0.0f;
    vec3 f = V3(0.0f);
    vec3 T, B;
    Onb(N, T, B);
    V = ToLocal(T, B, N, V);
    L = ToLocal(T, B, N, L);
    vec3 H;
    if (L.z > 0.0f) H = normalize(L + V);
    else H = normalize(L + V * p.eta);
    if (H.z < 0.0f) H = -H;

    vec3 Csheen, Cspec0;
    float F0;
    TintColors(p, p.eta, F0, Csheen, Cspec0);

    float dielectricWt = (1.0f - p.Metallic) * (1.0f - p.SpecTrans);
    float metalWt = p.Metallic;
    float glassWt = (1.0f - p.Metallic) * p.SpecTrans;

    float schlickWt = SchlickWeight(V.z);
    float diffPr = dielectricWt * Luminance(p.Albedo);
    float dielectricPr = dielectricWt * Luminance(mix(Cspec0, V3(1.0f), schlickWt));
    float metalPr = metalWt * Luminance(mix(p.Albedo, V3(1.0f), schlickWt));
    float glassPr = glassWt;
    float clearCtPr = 0.25f * p.Clearcoat;

    float invTotalWt = 1.0f / (diffPr + dielectricPr + metalPr + glassPr + clearCtPr);
    diffPr *= invTotalWt;
    dielectricPr *= invTotalWt;
    metalPr *= invTotalWt;
    glassPr *= invTotalWt;
    clearCtPr *= invTotalWt;

    bool reflect_ = L.z * V.z > 0.0f;
    float tmpPdf = 0.0f;
    float VDotH = gabs(dot(V, H));

    if (diffPr > 0.0f && reflect_) {
        f += EvalDisneyDiffuse(p, Csheen, V, L, H, tmpPdf) * dielectricWt;
        pdf += tmpPdf * diffPr;
    }
    if (dielectricPr > 0.0f && reflect_) {
        float F = (DielectricFresnel(VDotH, 1.0f / p.ior) - F0) / (1.0f - F0);
        f += EvalMicrofacetReflection(p, V, L, H, mix(Cspec0, V3(1.0f), F), tmpPdf) * dielectricWt;
        pdf += tmpPdf * dielectricPr;
    }
    if (metalPr > 0.0f && reflect_) {
        vec3 F = mix(p.Albedo, V3(1.0f), SchlickWeight(VDotH));
        f += EvalMicrofacetReflection(p, V, L, H, F, tmpPdf) * metalWt;
        pdf += tmpPdf * metalPr;
    }
    if (glassPr > 0.0f) {
        float F = DielectricFresnel(VDotH, p.eta);
        if (reflect_) {
            f += EvalMicrofacetReflection(p, V, L, H, V3(F), tmpPdf) * glassWt;
            pdf += tmpPdf * glassPr * F;
        } else {
            f += EvalMicrofacetRefraction(p, p.eta, V, L, H, V3(F), tmpPdf) * glassWt;
            pdf += tmpPdf * glassPr * (1.0f - F);
        }
    }
    if (clearCtPr > 0.0f && reflect_) {
        f += EvalClearcoat(p, V, L, H, tmpPdf) * 0.25f * p.Clearcoat;
        pdf += tmpPdf * clearCtPr;
    }
    return f * gabs(L.z);
}

static inline vec3 DisneySample(const Payload& p, vec3 V, vec3 N, vec3& L, float& pdf, uint32_t& seed)
{
    pdf = 0.0f;
    float r1 = RandomValue(seed);
    float r2 = RandomValue(seed);

    vec3 T, B;
    Onb(N, T, B);
    V = ToLocal(T, B, N, V);

    vec3 Csheen, Cspec0;
    float F0;
    TintColors(p, p.eta, F0, Csheen, Cspec0);

    float dielectricWt = (1.0f - p.Metallic) * (1.0f - p.SpecTrans);
    float metalWt = p.Metallic;
    float glassWt = (1.0f - p.Metallic) * p.SpecTrans;

    float schlickWt = SchlickWeight(V.z);
    float diffPr = dielectricWt * Luminance(p.Albedo);
    float dielectricPr = dielectricWt * Luminance(mix(Cspec0, V3(1.0f), schlickWt));
    float metalPr = metalWt * Luminance(mix(p.Albedo, V3(1.0f), schlickWt));
    float glassPr = glassWt;
    float clearCtPr = 0.25f * p.Clearcoat;

    float invTotalWt = 1.0f / (diffPr + dielectricPr + metalPr + glassPr + clearCtPr);
    diffPr *= invTotalWt;
    dielectricPr *= invTotalWt;
    metalPr *= invTotalWt;
    glassPr *= invTotalWt;
    clearCtPr *= invTotalWt;

    float cdf[5];
    cdf[0] = diffPr;
    cdf[1] = cdf[0] + dielectricPr;
    cdf[2] = cdf[1] + metalPr;
    cdf[3] = cdf[2] + glassPr;
    cdf[4] = cdf[3] + clearCtPr;

    float r3 = RandomValue(seed);

    if (r3 < cdf[0]) {
        L = CosineSampleHemisphere(r1, r2);
    } else if (r3 < cdf[2]) {
        vec3 H = SampleGGXVNDF(V, p.ax, p.ay, r1, r2);
        if (H.z < 0.0f) H = -H;
        L = normalize(reflect(-V, H));
    } else if (r3 < cdf[3]) {
        vec3 H = SampleGGXVNDF(V, p.ax, p.ay, r1, r2);
        float F = DielectricFresnel(gabs(dot(V, H)), p.eta);
        if (H.z < 0.0f) H = -H;
        r3 = (r3 - cdf[2]) / (cdf[3] - cdf[2]);
        if (r3 < F) L = normalize(reflect(-V, H));
        else L = normalize(refract(-V, H, p.eta));
    } else {
        vec3 H = SampleGTR1(p.ClearcoatRoughness, r1, r2);
        if (H.z < 0.0f) H = -H;
        L = normalize(reflect(-V, H));
    }

    L = ToWorld(T, B, N, L);
    V = ToWorld(T, B, N, V);
    return DisneyEval(p, V, N, L, pdf);
}

} // namespace orc
