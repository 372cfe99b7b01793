// The reference's three compute passes around the path tracer, as streaming HIP kernels:
//   k_postprocess      PathTracer/assets/shaders/PostProcessing.glsl:14-55  (exposure, ACES fit, gamma 2.2, RGBA8)
//   k_equirect_to_cube PathTracer/assets/shaders/EquirectangularToCubeMap.glsl:11-40 (lat-long image -> RGBA16F cube)
//   k_preetham         PathTracer/assets/shaders/PreethamSky.glsl:17-146     (analytic sky -> RGBA32F cube)
// One thread per pixel / texel; all three are bound by their stores.  Floating point as in pt_math.h: each
// operation written once, no contraction, transcendental functions from the deterministic set, so that results
// can be compared bit for bit with a CPU evaluation of the same expressions.
//
// Left open by the shader text and fixed here: image stores to rgba8 clamp and round half up, stores to rgba16f
// round to nearest even (the cube is kept as RGBA32F holding binary16 values); the equirectangular image is
// sampled bilinearly at level 0, repeating in u and clamping in v; acos clamps its argument.
#include "pt_context.h"
#include "pt_math.h"

namespace ptd {

PT_DEV float aces_in(float r, float g, float b, float m0, float m1, float m2) { return m0 * r + m1 * g + m2 * b; }

// PostProcessing.glsl:15-31
PT_DEV v3 aces_tonemap(v3 c)
{
    v3 v = V3(aces_in(c.x, c.y, c.z, 0.59719f, 0.35458f, 0.04823f),
              aces_in(c.x, c.y, c.z, 0.07600f, 0.90834f, 0.01566f),
              aces_in(c.x, c.y, c.z, 0.02840f, 0.13383f, 0.83777f));
    v3 a = V3(v.x * (v.x + 0.0245786f) - 0.000090537f, v.y * (v.y + 0.0245786f) - 0.000090537f, v.z * (v.z + 0.0245786f) - 0.000090537f);
    v3 b = V3(v.x * (0.983729f * v.x + 0.4329510f) + 0.238081f, v.y * (0.983729f * v.y + 0.4329510f) + 0.238081f,
              v.z * (0.983729f * v.z + 0.4329510f) + 0.238081f);
    v3 q = V3(fdiv(a.x, b.x), fdiv(a.y, b.y), fdiv(a.z, b.z));
    v3 r = V3(1.60475f * q.x + -0.53108f * q.y + -0.07367f * q.z,
              -0.10208f * q.x + 1.10813f * q.y + -0.00605f * q.z,
              -0.00327f * q.x + -0.07276f * q.y + 1.07602f * q.z);
    return V3(gclamp(r.x, 0.0f, 1.0f), gclamp(r.y, 0.0f, 1.0f), gclamp(r.z, 0.0f, 1.0f));
}

PT_DEV uint32_t unorm8(float v)
{
    if (!(v > 0.0f)) return 0u;
    if (v >= 1.0f) return 255u;
    return (uint32_t)(int)(v * 255.0f + 0.5f);
}

__global__ void k_postprocess(uint32_t n, float exposure, const float4* __restrict__ image, uint32_t* __restrict__ out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 px = image[i];
    v3 p = V3(px.x, px.y, px.z) * exposure;
    p = aces_tonemap(p);
    const float e = fdiv(1.0f, 2.2f);
    p = V3(det_pow(p.x, e), det_pow(p.y, e), det_pow(p.z, e));
    out[i] = unorm8(p.x) | (unorm8(p.y) << 8) | (unorm8(p.z) << 16) | 0xFF000000u;
}

// EquirectangularToCubeMap.glsl:11-25, PreethamSky.glsl:17-31
PT_DEV v3 cube_tex_coord(uint32_t gx, uint32_t gy, uint32_t face, float size)
{
    float stx = fdiv((float)gx, size), sty = fdiv((float)gy, size);
    float ux = 2.0f * stx - 1.0f, uy = 2.0f * (1.0f - sty) - 1.0f;
    v3 ret;
    if (face == 0) ret = V3(1.0f, uy, -ux);
    else if (face == 1) ret = V3(-1.0f, uy, ux);
    else if (face == 2) ret = V3(ux, 1.0f, -uy);
    else if (face == 3) ret = V3(ux, -1.0f, uy);
    else if (face == 4) ret = V3(ux, uy, 1.0f);
    else ret = V3(-ux, uy, -1.0f);
    return normalize(ret);
}

PT_DEV float4 sample_equirect(const float4* __restrict__ img, uint32_t w, uint32_t h, float u, float v)
{
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    if (!(x0f == x0f)) x0f = 0.0f;
    if (!(y0f == y0f)) y0f = 0.0f;
    float wf = (float)w;
    float xw0 = x0f - floorf(fdiv(x0f, wf)) * wf, x1f = x0f + 1.0f, xw1 = x1f - floorf(fdiv(x1f, wf)) * wf;
    int x0 = (int)xw0, x1 = (int)xw1;
    if (x0 < 0) x0 = 0;
    if (x0 >= (int)w) x0 = (int)w - 1;
    if (x1 < 0) x1 = 0;
    if (x1 >= (int)w) x1 = (int)w - 1;
    float top = (float)(h - 1);
    int y0 = y0f < 0.0f ? 0 : (y0f > top ? (int)h - 1 : (int)y0f);
    int y1 = (y0f + 1.0f) < 0.0f ? 0 : ((y0f + 1.0f) > top ? (int)h - 1 : (int)(y0f + 1.0f));
    float4 c00 = img[(size_t)y0 * w + x0], c10 = img[(size_t)y0 * w + x1];
    float4 c01 = img[(size_t)y1 * w + x0], c11 = img[(size_t)y1 * w + x1];
    return make_float4(mix(mix(c00.x, c10.x, fx), mix(c01.x, c11.x, fx), fy), mix(mix(c00.y, c10.y, fx), mix(c01.y, c11.y, fx), fy),
                       mix(mix(c00.z, c10.z, fx), mix(c01.z, c11.z, fx), fy), mix(mix(c00.w, c10.w, fx), mix(c01.w, c11.w, fx), fy));
}

// grid: x = texel within the face (row-major), y = face
__global__ void k_equirect_to_cube(const float4* __restrict__ img, uint32_t w, uint32_t h, uint32_t n, float4* __restrict__ cube)
{
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, face = blockIdx.y;
    if (t >= n * n) return;
    uint32_t gx = t % n, gy = t / n;
    const float PI = 3.141592f;                    // EquirectangularToCubeMap.glsl:9
    v3 tc = cube_tex_coord(gx, gy, face, (float)n);
    float phi = det_atan2(tc.z, tc.x);
    float theta = det_acos(tc.y);
    float u = fdiv(phi, 2.0f * PI) + 0.5f, v = fdiv(theta, PI);
    float4 c = sample_equirect(img, w, h, u, v);
    cube[(size_t)face * n * n + t] = make_float4(round_to_half(c.x), round_to_half(c.y), round_to_half(c.z), round_to_half(c.w));
}

// ---- PreethamSky.glsl ----
#define PT_SKY_PI 3.14159265359f      /* the #define at PreethamSky.glsl:33 shadows the const above it */

PT_DEV float saturated_dot(v3 a, v3 b) { return gmax(dot(a, b), 0.0f); }

// PreethamSky.glsl:107-110, one colour channel
PT_DEV float perez(float ct, float cg, float gamma, float A, float B, float C, float D, float E)
{
    return (A * det_exp(fdiv(B, ct)) + 1.0f) * ((C * det_exp(D * gamma) + 1.0f) + E * cg * cg);
}

// PreethamSky.glsl:112-130 with :40-105 inlined
PT_DEV v3 sky_luminance_rgb(v3 s, v3 e, float t)
{
    const float A[3] = {0.1787f * t - 1.4630f, -0.0193f * t - 0.2592f, -0.0167f * t - 0.2608f};
    const float B[3] = {-0.3554f * t + 0.4275f, -0.0665f * t + 0.0008f, -0.0950f * t + 0.0092f};
    const float C[3] = {-0.0227f * t + 5.3251f, -0.0004f * t + 0.2125f, -0.0079f * t + 0.2102f};
    const float D[3] = {0.1206f * t - 2.5771f, -0.0641f * t - 0.8989f, -0.0441f * t - 1.6537f};
    const float E[3] = {-0.0670f * t + 0.3703f, -0.0033f * t + 0.0452f, -0.0109f * t + 0.0529f};
    const v3 up = V3(0.0f, 1.0f, 0.0f);
    float thetaS = det_acos(saturated_dot(s, up));
    float thetaE = det_acos(saturated_dot(e, up));
    float gammaE = det_acos(saturated_dot(s, e));
    // zenith luminance and chromaticity
    float chi = (4.0f / 9.0f - fdiv(t, 120.0f)) * (PT_SKY_PI - 2.0f * thetaS);
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
    const float zen[3] = {Yz, xz, yz};
    float s0, c0, sE, cE, sG, cG, sS, cS;
    det_sincos(thetaE, sE, cE);
    det_sincos(gammaE, sG, cG);
    det_sincos(0.0f, s0, c0);
    det_sincos(thetaS, sS, cS);
    float Yp[3];
    for (int k = 0; k < 3; k++) {
        float fThetaGamma = perez(cE, cG, gammaE, A[k], B[k], C[k], D[k], E[k]);
        float fZeroThetaS = perez(c0, cS, thetaS, A[k], B[k], C[k], D[k], E[k]);
        Yp[k] = zen[k] * fdiv(fThetaGamma, fZeroThetaS);
    }
    // Yxy -> XYZ -> RGB
    float Y = Yp[0], x = Yp[1], y = Yp[2];
    float X = x * fdiv(Y, y);
    float Z = (1.0f - x - y) * fdiv(Y, y);
    return V3(X * 2.3706743f + Y * -0.9000405f + Z * -0.4706338f,
              X * -0.5138850f + Y * 1.4253036f + Z * 0.0885814f,
              X * 0.0052982f + Y * -0.0146949f + Z * 1.0093968f);
}

__global__ void k_preetham(float turbidity, float azimuth, float inclination, uint32_t n, float4* __restrict__ cube)
{
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, face = blockIdx.y;
    if (t >= n * n) return;
    v3 tc = cube_tex_coord(t % n, t / n, face, (float)n);
    float si, ci, sa, ca;
    det_sincos(inclination, si, ci);
    det_sincos(azimuth, sa, ca);
    v3 sun = normalize(V3(si * ca, ci, si * sa));
    v3 sky = sky_luminance_rgb(sun, tc, turbidity);
    cube[(size_t)face * n * n + t] = make_float4(sky.x * 0.05f, sky.y * 0.05f, sky.z * 0.05f, 1.0f);
}

static int replace_env(pt_ctx* ctx, uint32_t n)
{
    PT_SYNC(ctx);
    // the new cube is allocated before the old one goes, so that a failed allocation leaves the scene as it was
    float* fresh = nullptr;
    PT_HIP(ctx, hipMalloc((void**)&fresh, (size_t)6 * n * n * 16));
    if (ctx->d_env) (void)hipFree(ctx->d_env);
    ctx->d_env = fresh;
    ctx->env_n = n;
    return PT_OK;
}

int launch_postprocess(pt_ctx* ctx, float exposure)
{
    int rc = join_chains(ctx);
    if (rc) return rc;
    ctx->main_dirty = true;
    uint32_t n = ctx->width * ctx->height;
    if (!ctx->d_post) PT_HIP(ctx, hipMalloc((void**)&ctx->d_post, (size_t)n * 4));
    k_postprocess<<<(n + 255) / 256, 256, 0, ctx->stream>>>(n, exposure, reinterpret_cast<const float4*>(ctx->d_image), ctx->d_post);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

int env_from_equirect(pt_ctx* ctx, const float* rgba, uint32_t w, uint32_t h, uint32_t n)
{
    int rc = replace_env(ctx, n);
    if (rc) return rc;
    float4* d_img = nullptr;
    PT_HIP(ctx, hipMalloc((void**)&d_img, (size_t)w * h * 16));
    hipError_t e = hipMemcpyAsync(d_img, rgba, (size_t)w * h * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        k_equirect_to_cube<<<dim3((n * n + 255) / 256, 6), 256, 0, ctx->stream>>>(d_img, w, h, n, reinterpret_cast<float4*>(ctx->d_env));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_img);
    PT_HIP(ctx, e);
    return PT_OK;
}

int env_upload(pt_ctx* ctx, const float* rgba, uint32_t n)
{
    int rc = replace_env(ctx, n);
    if (rc) return rc;
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_env, rgba, (size_t)6 * n * n * 16, hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));      // the host array may go away after return
    return PT_OK;
}

int env_preetham(pt_ctx* ctx, float turbidity, float azimuth, float inclination, uint32_t n)
{
    int rc = replace_env(ctx, n);
    if (rc) return rc;
    k_preetham<<<dim3((n * n + 255) / 256, 6), 256, 0, ctx->stream>>>(turbidity, azimuth, inclination, n, reinterpret_cast<float4*>(ctx->d_env));
    PT_HIP(ctx, hipGetLastError());
    ctx->main_dirty = true;
    return PT_OK;
}

} // namespace ptd
