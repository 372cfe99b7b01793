// Disney BSDF sampling and evaluation for the shade kernel.
// Follows PathTracer/assets/shaders/RayTracing/Disney.glsl:42-359 and the helpers in
// Sampling.glsl:25-47,70-96,109-156,179-184, with the operation order of the GLSL
// source.  The orthonormal basis and the tint colours are computed once per hit
// and shared by sampling and evaluation (the shader recomputes them; they are
// pure functions of the same inputs, so the results are identical).
#pragma once
#include "pt_math.h"

namespace ptd {

#define PT_PI 3.14159265358979323f
#define PT_INV_PI 0.31830988618379067f
#define PT_TWO_PI 6.28318530717958648f

// what ClosestHit.glsl:198-226 leaves in the payload for the BSDF
struct Surf {
    v3 albedo;
    float metallic, roughness, subsurface, specularTint, sheen, sheenTint;
    float clearcoat, clearcoatRoughness, specTrans, ior, ax, ay, eta;
};

PT_DEV float luminance(v3 c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; }

PT_DEV float schlick_weight(float u)
{
    float m = gclamp(1.0f - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}

PT_DEV float dielectric_fresnel(float cosThetaI, float eta)
{
    float sinThetaTSq = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaTSq > 1.0f) return 1.0f;
    float cosThetaT = fsqrt(gmax(1.0f - sinThetaTSq, 0.0f));
    float rs = fdiv(eta * cosThetaT - cosThetaI, eta * cosThetaT + cosThetaI);
    float rp = fdiv(eta * cosThetaI - cosThetaT, eta * cosThetaI + cosThetaT);
    return 0.5f * (rs * rs + rp * rp);
}

PT_DEV float gtr1(float NDotH, float a)
{
    if (a >= 1.0f) return PT_INV_PI;
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return fdiv(a2 - 1.0f, PT_PI * det_log(a2) * t);
}

PT_DEV float gtr2_aniso(float NDotH, float HDotX, float HDotY, float ax, float ay)
{
    float a = fdiv(HDotX, ax);
    float b = fdiv(HDotY, ay);
    float c = a * a + b * b + NDotH * NDotH;
    return fdiv(1.0f, PT_PI * ax * ay * c * c);
}

PT_DEV float smith_g(float NDotV, float alphaG)
{
    float a = alphaG * alphaG;
    float b = NDotV * NDotV;
    return fdiv(2.0f * NDotV, NDotV + fsqrt(a + b - a * b));
}

PT_DEV float smith_g_aniso(float NDotV, float VDotX, float VDotY, float ax, float ay)
{
    float a = VDotX * ax;
    float b = VDotY * ay;
    float c = NDotV;
    return fdiv(2.0f * NDotV, NDotV + fsqrt(a * a + b * b + c * c));
}

PT_DEV v3 sample_gtr1(float rgh, float r1, float r2)
{
    float a = gmax(0.001f, rgh);
    float a2 = a * a;
    float phi = r1 * PT_TWO_PI;
    float cosTheta = fsqrt(fdiv(1.0f - det_pow(a2, 1.0f - r2), 1.0f - a2));
    float sinTheta = gclamp(fsqrt(1.0f - (cosTheta * cosTheta)), 0.0f, 1.0f);
    float sinPhi, cosPhi;
    det_sincos(phi, sinPhi, cosPhi);
    return V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}

PT_DEV v3 sample_ggx_vndf(v3 V, float ax, float ay, float r1, float r2)
{
    v3 Vh = normalize(V3(ax * V.x, ay * V.y, V.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    v3 T1 = lensq > 0.0f ? V3(-Vh.y, Vh.x, 0.0f) * inversesqrt(lensq) : V3(1.0f, 0.0f, 0.0f);
    v3 T2 = cross(Vh, T1);
    float r = fsqrt(r1);
    float phi = 2.0f * PT_PI * r2;
    float sphi, cphi;
    det_sincos(phi, sphi, cphi);
    float t1 = r * cphi;
    float t2 = r * sphi;
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * fsqrt(1.0f - t1 * t1) + s * t2;
    v3 Nh = t1 * T1 + t2 * T2 + fsqrt(gmax(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalize(V3(ax * Nh.x, ay * Nh.y, gmax(0.0f, Nh.z)));
}

PT_DEV v3 cosine_sample_hemisphere(float r1, float r2)
{
    float r = fsqrt(r1);
    float phi = PT_TWO_PI * r2;
    float s, c;
    det_sincos(phi, s, c);
    v3 d;
    d.x = r * c;
    d.y = r * s;
    d.z = fsqrt(gmax(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return d;
}

PT_DEV void onb(v3 N, v3& T, v3& B)
{
    v3 up = gabs(N.z) < 0.9999999f ? V3(0.0f, 0.0f, 1.0f) : V3(1.0f, 0.0f, 0.0f);
    T = normalize(cross(up, N));
    B = cross(N, T);
}

PT_DEV v3 to_world(v3 X, v3 Y, v3 Z, v3 V) { return V.x * X + V.y * Y + V.z * Z; }
PT_DEV v3 to_local(v3 X, v3 Y, v3 Z, v3 V) { return V3(dot(V, X), dot(V, Y), dot(V, Z)); }

struct Tint { float F0; v3 Csheen, Cspec0; };

PT_DEV Tint tint_colors(const Surf& p)
{
    Tint t;
    float lum = luminance(p.albedo);
    v3 ctint = lum > 0.0f ? p.albedo / lum : V3(1.0f);
    float F0 = fdiv(1.0f - p.eta, 1.0f + p.eta);
    F0 *= F0;
    t.F0 = F0;
    t.Cspec0 = F0 * mix(V3(1.0f), ctint, p.specularTint);
    t.Csheen = mix(V3(1.0f), ctint, p.sheenTint);
    return t;
}

struct LobePr { float diff, diel, metal, glass, cc, dielectricWt, metalWt, glassWt; };

PT_DEV LobePr lobe_probabilities(const Surf& p, const Tint& t, float Vz)
{
    LobePr w;
    w.dielectricWt = (1.0f - p.metallic) * (1.0f - p.specTrans);
    w.metalWt = p.metallic;
    w.glassWt = (1.0f - p.metallic) * p.specTrans;
    float schlickWt = schlick_weight(Vz);
    w.diff = w.dielectricWt * luminance(p.albedo);
    w.diel = w.dielectricWt * luminance(mix(t.Cspec0, V3(1.0f), schlickWt));
    w.metal = w.metalWt * luminance(mix(p.albedo, V3(1.0f), schlickWt));
    w.glass = w.glassWt;
    w.cc = 0.25f * p.clearcoat;
    float invTotalWt = fdiv(1.0f, w.diff + w.diel + w.metal + w.glass + w.cc);
    w.diff *= invTotalWt;
    w.diel *= invTotalWt;
    w.metal *= invTotalWt;
    w.glass *= invTotalWt;
    w.cc *= invTotalWt;
    return w;
}

PT_DEV v3 eval_microfacet_reflection(const Surf& p, v3 V, v3 L, v3 H, v3 F, float& pdf)
{
    pdf = 0.0f;
    if (L.z <= 0.0f) return V3(0.0f);
    float D = gtr2_aniso(H.z, H.x, H.y, p.ax, p.ay);
    float G1 = smith_g_aniso(gabs(V.z), V.x, V.y, p.ax, p.ay);
    float G2 = G1 * smith_g_aniso(gabs(L.z), L.x, L.y, p.ax, p.ay);
    pdf = fdiv(G1 * D, 4.0f * V.z);
    return (F * D * G2) / (4.0f * L.z * V.z);
}

// Disney.glsl:252-359 with V and L already in the shading frame (z = normal)
PT_DEV v3 disney_eval_local(const Surf& p, const Tint& tint, v3 V, v3 L, float& pdf)
{
    pdf = 0.0f;
    v3 f = V3(0.0f);
    v3 H;
    if (L.z > 0.0f) H = normalize(L + V);
    else H = normalize(L + V * p.eta);
    if (H.z < 0.0f) H = -H;

    LobePr w = lobe_probabilities(p, tint, V.z);
    bool refl = L.z * V.z > 0.0f;
    float tmpPdf = 0.0f;
    float VDotH = gabs(dot(V, H));

    if (w.diff > 0.0f && refl) {
        v3 fd = V3(0.0f);
        tmpPdf = 0.0f;
        if (!(L.z <= 0.0f)) {
            float LDotH = dot(L, H);
            float Rr = 2.0f * p.roughness * LDotH * LDotH;
            float FL = schlick_weight(L.z);
            float FV = schlick_weight(V.z);
            float Fretro = Rr * (FL + FV + FL * FV * (Rr - 1.0f));
            float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
            float Fss90 = 0.5f * Rr;
            float Fss = mix(1.0f, Fss90, FL) * mix(1.0f, Fss90, FV);
            float ss = 1.25f * (Fss * (fdiv(1.0f, L.z + V.z) - 0.5f) + 0.5f);
            float FH = schlick_weight(LDotH);
            v3 Fsheen = FH * p.sheen * tint.Csheen;
            tmpPdf = L.z * PT_INV_PI;
            fd = PT_INV_PI * p.albedo * mix(Fd + Fretro, ss, p.subsurface) + Fsheen;
        }
        f = f + fd * w.dielectricWt;
        pdf += tmpPdf * w.diff;
    }
    if (w.diel > 0.0f && refl) {
        float F = fdiv(dielectric_fresnel(VDotH, fdiv(1.0f, p.ior)) - tint.F0, 1.0f - tint.F0);
        f = f + eval_microfacet_reflection(p, V, L, H, mix(tint.Cspec0, V3(1.0f), F), tmpPdf) * w.dielectricWt;
        pdf += tmpPdf * w.diel;
    }
    if (w.metal > 0.0f && refl) {
        v3 F = mix(p.albedo, V3(1.0f), schlick_weight(VDotH));
        f = f + eval_microfacet_reflection(p, V, L, H, F, tmpPdf) * w.metalWt;
        pdf += tmpPdf * w.metal;
    }
    if (w.glass > 0.0f) {
        float F = dielectric_fresnel(VDotH, p.eta);
        if (refl) {
            f = f + eval_microfacet_reflection(p, V, L, H, V3(F), tmpPdf) * w.glassWt;
            pdf += tmpPdf * w.glass * F;
        } else {
            v3 fr = V3(0.0f);
            tmpPdf = 0.0f;
            if (!(L.z >= 0.0f)) {
                float LDotH = dot(L, H);
                float VDotH2 = dot(V, H);
                float D = gtr2_aniso(H.z, H.x, H.y, p.ax, p.ay);
                float G1 = smith_g_aniso(gabs(V.z), V.x, V.y, p.ax, p.ay);
                float G2 = G1 * smith_g_aniso(gabs(L.z), L.x, L.y, p.ax, p.ay);
                float denom = LDotH + VDotH2 * p.eta;
                denom *= denom;
                float eta2 = p.eta * p.eta;
                float jacobian = fdiv(gabs(LDotH), denom);
                tmpPdf = fdiv(G1 * gmax(0.0f, VDotH2) * D * jacobian, V.z);
                v3 sqrtAlbedo = V3(det_pow(p.albedo.x, 0.5f), det_pow(p.albedo.y, 0.5f), det_pow(p.albedo.z, 0.5f));
                fr = (sqrtAlbedo * (1.0f - V3(F)) * D * G2 * gabs(VDotH2) * jacobian * eta2) / gabs(L.z * V.z);
            }
            f = f + fr * w.glassWt;
            pdf += tmpPdf * w.glass * (1.0f - F);
        }
    }
    if (w.cc > 0.0f && refl) {
        v3 fc = V3(0.0f);
        tmpPdf = 0.0f;
        if (!(L.z <= 0.0f)) {
            float VDotH2 = dot(V, H);
            float F = mix(0.04f, 1.0f, schlick_weight(VDotH2));
            float D = gtr1(H.z, p.clearcoatRoughness);
            float G = smith_g(L.z, 0.25f) * smith_g(V.z, 0.25f);
            float jacobian = fdiv(1.0f, 4.0f * VDotH2);
            tmpPdf = D * H.z * jacobian;
            fc = V3(F) * D * G;
        }
        f = f + fc * 0.25f * p.clearcoat;
        pdf += tmpPdf * w.cc;
    }
    return f * gabs(L.z);
}

// Disney.glsl:150-250.  V and N are world vectors; returns f, writes L (world) and pdf.
PT_DEV v3 disney_sample(const Surf& p, v3 Vw, v3 N, v3& Lout, float& pdf, uint32_t& seed)
{
    float r1 = random_value(seed);
    float r2 = random_value(seed);
    v3 T, B;
    onb(N, T, B);
    v3 V = to_local(T, B, N, Vw);
    Tint tint = tint_colors(p);
    LobePr w = lobe_probabilities(p, tint, V.z);
    float cdf0 = w.diff;
    float cdf1 = cdf0 + w.diel;
    float cdf2 = cdf1 + w.metal;
    float cdf3 = cdf2 + w.glass;
    float r3 = random_value(seed);

    v3 L;
    if (r3 < cdf0) {
        L = cosine_sample_hemisphere(r1, r2);
    } else if (r3 < cdf2) {
        v3 H = sample_ggx_vndf(V, p.ax, p.ay, r1, r2);
        if (H.z < 0.0f) H = -H;
        L = normalize(reflect(-V, H));
    } else if (r3 < cdf3) {
        v3 H = sample_ggx_vndf(V, p.ax, p.ay, r1, r2);
        float F = dielectric_fresnel(gabs(dot(V, H)), p.eta);
        if (H.z < 0.0f) H = -H;
        r3 = fdiv(r3 - cdf2, cdf3 - cdf2);
        if (r3 < F) L = normalize(reflect(-V, H));
        else L = normalize(refract(-V, H, p.eta));
    } else {
        v3 H = sample_gtr1(p.clearcoatRoughness, r1, r2);
        if (H.z < 0.0f) H = -H;
        L = normalize(reflect(-V, H));
    }
    // the shader goes back to world space and DisneyEval projects again
    v3 Lw = to_world(T, B, N, L);
    v3 Vw2 = to_world(T, B, N, V);
    Lout = Lw;
    return disney_eval_local(p, tint, to_local(T, B, N, Vw2), to_local(T, B, N, Lw), pdf);
}

PT_DEV v3 disney_eval(const Surf& p, v3 Vw, v3 N, v3 Lw, float& pdf)
{
    v3 T, B;
    onb(N, T, B);
    Tint tint = tint_colors(p);
    return disney_eval_local(p, tint, to_local(T, B, N, Vw), to_local(T, B, N, Lw), pdf);
}

} // namespace ptd
