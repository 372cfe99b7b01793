// TEST INFRASTRUCTURE (CPU oracle) -- not part of the shipped library.
//
// The handful of GLSL 4.60 vector types and built-ins the reference shaders use,
// with one fixed evaluation order (plain C left-to-right, no fused contraction:
// build with -ffp-contract=off).  normalize(v) is v * (1 / sqrt(dot(v,v))), so
// normalize(0) yields NaN, which the reference turns into a red pixel
// (PathTracer/assets/shaders/RayTracing/RayGen.glsl:369-370).
#pragma once
#include "det_math.h"

namespace orc {

struct vec2 { float x, y; };
struct vec3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct vec4 { float x, y, z, w; };
struct mat4 { float c[4][4]; };   // c[column][row], glm / GLSL column-major

static inline vec3 V3(float x, float y, float z) { return vec3{x, y, z}; }
static inline vec3 V3(float s) { return vec3{s, s, s}; }

static inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
static inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
static inline vec3 operator+(vec3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
static inline vec3 operator-(float s, vec3 a) { return {s - a.x, s - a.y, s - a.z}; }
static inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
static inline vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
static inline vec3& operator*=(vec3& a, float s) { a = a * s; return a; }
static inline vec3& operator/=(vec3& a, float s) { a = a / s; return a; }

static inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline vec3 cross(vec3 a, vec3 b)
{
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
static inline float inversesqrt(float x) { return 1.0f / sqrtf(x); }
static inline vec3 normalize(vec3 v) { return v * inversesqrt(dot(v, v)); }
static inline float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline vec3 mix(vec3 a, vec3 b, float t) { return a * (1.0f - t) + b * t; }
static inline float gabs(float x) { return u2f(f2u(x) & 0x7fffffffu); }
static inline bool gisnan(float x) { return x != x; }

// reflect(I, N) = I - 2 * dot(N, I) * N
static inline vec3 reflect(vec3 I, vec3 N) { return I - N * (2.0f * dot(N, I)); }

// refract(I, N, eta): zero vector on total internal reflection
static inline vec3 refract(vec3 I, vec3 N, float eta)
{
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return V3(0.0f);
    return I * eta - N * (eta * d + sqrtf(k));
}

static inline vec4 mul(const mat4& m, vec4 v)
{
    vec4 r;
    r.x = m.c[0][0] * v.x + m.c[1][0] * v.y + m.c[2][0] * v.z + m.c[3][0] * v.w;
    r.y = m.c[0][1] * v.x + m.c[1][1] * v.y + m.c[2][1] * v.z + m.c[3][1] * v.w;
    r.z = m.c[0][2] * v.x + m.c[1][2] * v.y + m.c[2][2] * v.z + m.c[3][2] * v.w;
    r.w = m.c[0][3] * v.x + m.c[1][3] * v.y + m.c[2][3] * v.z + m.c[3][3] * v.w;
    return r;
}

} // namespace orc
