/* TEST INFRASTRUCTURE (CPU oracle) -- not part of the shipped library.
 * PARITY UNPINNED: see oracle/shaders.h.
 *
 * C entry points of the CPU oracle, loaded with ctypes by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg only.
 * The scene / parameter structs have the same layout as the product's
 * include/pathtracer_amd.h so a test can hand one description to both.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_texture_desc {
    uint32_t width, height;
    uint32_t flags;            /* sampler state: wrap of u in bits 0-1, of v in bits 2-3 (0 repeat, 1 clamp, 2 mirror), bit 4 nearest */
    uint32_t pad_;
    const uint8_t* rgba8;      /* width*height*4 bytes, row 0 first */
} orc_texture_desc;

typedef struct orc_scene_desc {
    const float* vertices;     /* num_vertices * 12: pos3 uv2 nrm3 tan4 (ClosestHit.glsl:56-81) */
    const uint32_t* indices;
    const uint32_t* submeshes; /* num_submeshes * 4: bufferIndex, vertexOffset, indexOffset, materialIndex (ClosestHit.glsl:147-150) */
    const uint32_t* submesh_index_counts; /* num_submeshes: 3 * triangle count */
    const float* submesh_xforms;          /* num_submeshes * 12: object-to-world, 3 rows of 4 */
    const float* materials;    /* num_materials * 22 words (ClosestHit.glsl:113-142) */
    const orc_texture_desc* textures;
    const float* env_rgba;     /* 6 * face * face * 4 floats (+X,-X,+Y,-Y,+Z,-Z) or NULL = constant (1,1,1) */
    uint32_t num_vertices, num_indices, num_submeshes, num_materials, num_textures, env_face_size;
} orc_scene_desc;

typedef struct orc_params {
    uint32_t spp_per_frame;    /* RayGen.glsl:319  SAMPLE_COUNT = 5 */
    uint32_t max_bounces;      /* RayGen.glsl:111  MAX_BOUNCES = 20 */
    uint32_t rr_depth;         /* RayGen.glsl:163-173, compiled out in the reference: 0 = off */
    float sky_scale;           /* RayGen.glsl:132  10.0 */
    float eps;                 /* RayGen.glsl:159  0.0003 */
    float tmin;                /* RayGen.glsl:341  0.00001 */
    float tmax;                /* RayGen.glsl:342  1e27 */
} orc_params;

typedef struct orc_ctx orc_ctx;

orc_ctx* orc_create(const orc_scene_desc* scene);
void orc_destroy(orc_ctx*);
int orc_update_materials(orc_ctx*, const float* materials, uint32_t num_materials);
int orc_resize(orc_ctx*, uint32_t width, uint32_t height);
int orc_set_camera(orc_ctx*, const float* five_mat4);   /* CameraBuffer, RayTracingLayer.h:17-24 */
int orc_set_params(orc_ctx*, const orc_params*);
int orc_trace(orc_ctx*, uint32_t frame_index, int num_threads, int use_bvh);
int orc_read_image(orc_ctx*, float* rgba);
int orc_read_accum(orc_ctx*, float* rgba);
int orc_write_accum(orc_ctx*, const float* rgba);
int orc_read_primary_ids(orc_ctx*, uint32_t* submesh_prim_pairs);
uint64_t orc_ray_count(orc_ctx*);
void orc_traversal_stats(orc_ctx*, uint64_t* nodes, uint64_t* tris);
uint32_t orc_num_triangles(orc_ctx*);

/* closest hit of arbitrary rays: rays = n * 8 floats (o.xyz, tmin, d.xyz, tmax);
 * hits = n * 5 words: t, u, v (float bits), submesh, prim; a miss is
 * (tmax, 0, 0, 0xFFFFFFFF, 0xFFFFFFFF). */
int orc_intersect(orc_ctx*, const float* rays, uint32_t n, uint32_t* hits, int use_bvh, int num_threads);

/* function-level known-answer hooks */
uint32_t orc_pcg_hash(uint32_t* seed);
float orc_random_value(uint32_t* seed);
/* fn: 0 sin, 1 cos, 2 log, 3 exp, 4 pow(x, y) */
void orc_det_math(int fn, const float* x, const float* y, float* out, uint32_t n);   /* fn: 0 sin 1 cos 2 log 3 exp 4 pow 5 atan2 6 acos 7 tan 8 round-to-half */
/* texture(sampler2D, uv) of one texture at n coordinates (uv = n * 2 floats) -> n * 4 floats */
void orc_sample_texture(const orc_texture_desc* tex, const float* uv, uint32_t n, float* rgba);
/* PostProcessing.glsl, EquirectangularToCubeMap.glsl, PreethamSky.glsl (oracle/passes.h) */
void orc_postprocess(const float* rgba, uint32_t num_pixels, float exposure, uint8_t* rgba8);
void orc_equirect_to_cube(const float* rgba, uint32_t w, uint32_t h, uint32_t face_size, float* cube);
void orc_preetham_sky(float turbidity, float azimuth, float inclination, uint32_t face_size, float* cube);
/* surf = 17 floats: albedo3 metallic roughness subsurface specularTint sheen sheenTint
 * clearcoat clearcoatRoughness specTrans ior ax ay eta anisotropic(unused by BSDF);
 * V, N world vectors; returns f (3), L (3), pdf and the advanced seed. */
void orc_disney_sample(const float* surf, const float* V, const float* N, uint32_t* seed, float* f, float* L, float* pdf);
void orc_disney_eval(const float* surf, const float* V, const float* N, const float* L, float* f, float* pdf);
/* ClosestHit.glsl:144-226 for the hit (submesh, prim, u, v, t) of a ray with direction dir: 26 payload floats
 * (Distance, Albedo3, Roughness, Metallic, Emission3, WorldNormal3, the nine Disney extras, ax, ay, eta) */
int orc_closest_hit(orc_ctx*, uint32_t submesh, uint32_t prim, float u, float v, float t, const float* dir, float* out26);
/* TracePath (RayGen.glsl:106-176) of one ray under the context's parameters; the seed advances as in the shader */
void orc_trace_path(orc_ctx*, const float* origin, const float* dir, uint32_t* seed, int use_bvh, float* radiance);

#ifdef __cplusplus
}
#endif
#endif
