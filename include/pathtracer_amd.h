/* pathtracer_amd -- C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * The reference renderer has no plugin interface; its boundary to the GPU is the
 * ray-tracing descriptor set plus one vkCmdTraceRaysKHR per frame
 * (PathTracer/src/RayTracingLayer.cpp:201-284, bindings 0-11 at :248-263).
 * Every entry point below replaces one piece of that boundary and cites it.
 * Plain pointers and sizes only; the caller owns every host array and the
 * library copies what it needs before returning.  All functions return 0 on
 * success or a PT_ERR_* code; pt_last_error() gives the text.  A context is not
 * thread safe; use one context per GPU.
 */
#ifndef PATHTRACER_AMD_H
#define PATHTRACER_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_ERR_INVALID 1      /* bad argument / call order */
#define PT_ERR_HIP 2          /* a HIP runtime call failed */
#define PT_ERR_NO_DEVICE 3    /* no gfx950 device, or ordinal out of range */
#define PT_ERR_BVH_DEPTH 4    /* built hierarchy deeper than the traversal stack */
#define PT_ERR_TOO_LARGE 5    /* size beyond a documented limit */
#define PT_ERR_TRAVERSAL 6    /* a traversal wave reached its iteration bound: results since the last synchronisation are not valid */

/* binding 9: sampler2D u_Textures[] (ClosestHit.glsl:15).  RGBA8 UNORM texels,
 * row 0 first; sampled at level 0, bilinear with repeat unless flags say otherwise (the
 * reference's samplers are made by its engine from the glTF sampler objects). */
#define PT_TEX_WRAP_REPEAT 0u
#define PT_TEX_WRAP_CLAMP 1u
#define PT_TEX_WRAP_MIRROR 2u
#define PT_TEX_WRAP_S(mode) (mode)          /* wrap mode of u: bits 0-1 */
#define PT_TEX_WRAP_T(mode) ((mode) << 2)   /* wrap mode of v: bits 2-3 */
#define PT_TEX_NEAREST 16u                  /* nearest texel instead of bilinear */
typedef struct pt_texture_desc {
    uint32_t width, height;
    uint32_t flags;            /* PT_TEX_* above; 0 = bilinear, repeat */
    uint32_t pad_;
    const uint8_t* rgba8;
} pt_texture_desc;

/* The scene tables the absent engine's Mesh / AccelerationStructure hand to the
 * shaders (RayTracingLayer.cpp:207,227-260):
 *   binding 4  vertices: 12 floats each = pos3 uv2 nrm3 tan4 (ClosestHit.glsl:56-81)
 *   binding 5  indices:  uint32
 *   binding 6  submeshes: 4 uint32 each = bufferIndex(0), vertexOffset, indexOffset,
 *              materialIndex (ClosestHit.glsl:147-150); one TLAS instance per submesh
 *   binding 8  materials: 22 words each (ClosestHit.glsl:113-142)
 *   binding 10 env cube: 6 faces (+X,-X,+Y,-Y,+Z,-Z) of face*face RGBA32F, or NULL
 *              for a constant white sky
 * plus what the driver keeps inside the TLAS: each submesh's index count and its
 * object-to-world transform (3 rows of 4 floats), i.e.
 * AccelerationStructureSpecification{Mesh, Transform} (RayTracingLayer.cpp:342-348)
 * already multiplied with the submesh's WorldTransform. */
typedef struct pt_scene_desc {
    const float* vertices;
    const uint32_t* indices;
    const uint32_t* submeshes;
    const uint32_t* submesh_index_counts;
    const float* submesh_xforms;
    const float* materials;
    const pt_texture_desc* textures;
    const float* env_rgba;
    uint32_t num_vertices, num_indices, num_submeshes, num_materials, num_textures, env_face_size;
} pt_scene_desc;

/* Constants the reference hard-codes in RayGen.glsl; defaults = reference values. */
typedef struct pt_params {
    uint32_t spp_per_frame;    /* RayGen.glsl:319  SAMPLE_COUNT = 5 */
    uint32_t max_bounces;      /* RayGen.glsl:111  MAX_BOUNCES = 20 */
    uint32_t rr_depth;         /* RayGen.glsl:163-173 Russian roulette from this bounce on; 0 = off (reference) */
    float sky_scale;           /* RayGen.glsl:132  10.0 */
    float eps;                 /* RayGen.glsl:159  0.0003 */
    float tmin;                /* RayGen.glsl:341  0.00001 */
    float tmax;                /* RayGen.glsl:342  1e27 */
} pt_params;

typedef struct pt_accel_stats {
    uint32_t num_triangles;
    uint32_t num_nodes;        /* 8-wide nodes, 128 bytes each */
    uint32_t max_depth;        /* levels of 8-wide nodes */
    uint32_t num_leaves;
    float build_ms;            /* device time of the build kernels */
    float sah_cost;            /* surface-area-heuristic cost of the built hierarchy */
    uint32_t num_references;   /* triangle records in the leaves: num_triangles, or more when long thin triangles were split */
} pt_accel_stats;

/* per-kernel device time of the frames traced since the previous read-out (HIP events on the
 * streams the kernels run on).  A frame runs as several chains side by side (pt_set_chains) and
 * consecutive frames overlap, so launches of one kernel overlap: raygen_ms / extend_ms / shade_ms
 * are sums of the launch durations, extend_busy_ms is the time during which at least one
 * traversal launch was running, total_ms is first start to last end. */
typedef struct pt_trace_timing {
    float total_ms, raygen_ms, extend_ms, shade_ms;
    float extend_busy_ms;
    uint32_t chains;
    uint32_t extend_launches, shade_launches;   /* launches that had rays queued */
    uint64_t rays;             /* ray segments traced by that call */
    uint64_t nodes_visited;    /* hierarchy nodes fetched / triangles tested by the traversal kernel; */
    uint64_t tris_tested;      /* only counted in profiling mode 2, else 0 */
} pt_trace_timing;

typedef struct pt_ctx pt_ctx;

/* Application::GetVulkanDevice() (RayTracingLayer.cpp:18): picks the GPU, creates the stream. */
int pt_create(int device_ordinal, pt_ctx** out);
void pt_destroy(pt_ctx* ctx);
const char* pt_last_error(pt_ctx* ctx);   /* ctx may be NULL: last error of pt_create */
const char* pt_last_warning(pt_ctx* ctx); /* advice left by the last pt_set_chains; empty when there is none */
void pt_default_params(pt_params* out);

/* Mesh vertex/index buffers, submesh table, material buffer, textures, radiance map:
 * descriptor writes for bindings 4,5,6,8,9,10 (RayTracingLayer.cpp:227-263). */
int pt_scene_upload(pt_ctx* ctx, const pt_scene_desc* scene);
/* CreateAccelerationStructure() (RayTracingLayer.cpp:342-348): GPU build of the hierarchy
 * that replaces the driver's BLAS/TLAS (binding 0).  flags: PT_BUILD_* below. */
#define PT_BUILD_DEFAULT 0u
#define PT_BUILD_CLUSTER 8u    /* agglomerative clustering instead of the default tree (Morton radix tree + treelet restructuring) */
#define PT_BUILD_LBVH 1u       /* with PT_BUILD_CLUSTER: build both trees and keep the one with the lower surface-area cost */
#define PT_BUILD_NO_TREELET 16u /* no treelet restructuring passes (eight subtrees at a time, cheapest topology by summed box
                                * area) over the binary tree: PT_BUILD_NO_TREELET alone is the fastest build */
#define PT_BUILD_GREEDY 4u     /* children of a wide node chosen greedily (largest box first) instead of by the surface-area-optimal
                                * dynamic programme: kept for comparison */
#define PT_BUILD_NO_SPLIT 2u   /* every triangle is entered once; by default large triangles whose boxes are mostly empty are
                                * entered once per cell they pass through (same hits, tighter boxes) */
int pt_build_accel(pt_ctx* ctx, uint32_t flags, pt_accel_stats* stats);
/* AccelerationStructure::UpdateMaterialData() (RayTracingLayer.cpp:621). */
int pt_update_materials(pt_ctx* ctx, const float* materials, uint32_t num_materials);
/* Image::Resize of m_Image and m_AccumulationImage (RayTracingLayer.cpp:392-393). */
int pt_resize(pt_ctx* ctx, uint32_t width, uint32_t height);
/* Optional: render into caller-owned device buffers (width*height*4 floats each)
 * instead of the context's own images (bindings 1 and 2), e.g. torch tensors that a
 * collective then reduces.  Pass NULL to go back to the context's own buffers. */
int pt_bind_images(pt_ctx* ctx, void* device_image, void* device_accum);
/* CameraBuffer upload, binding 3 (RayTracingLayer.h:17-24, RayTracingLayer.cpp:403-411):
 * ViewProjection, InverseViewProjection, View, InverseView, InverseProjection,
 * column-major (glm).  Only the last two are read by the shaders. */
int pt_set_camera(pt_ctx* ctx, const float* five_mat4);
int pt_set_params(pt_ctx* ctx, const pt_params* params);
/* Arithmetic of the shading stage (sky look-up, ClosestHit surface fetch, DisneySample / DisneyEval, throughput and the
 * next ray: Miss.glsl, ClosestHit.glsl:144-226, Disney.glsl:150-359, RayGen.glsl:126-173).
 * PT_PRECISION_EXACT (default): correctly rounded IEEE binary32 operations in one fixed order; frames equal the CPU
 *   oracle's bit for bit.
 * PT_PRECISION_FAST: the hardware's reciprocal, square root, reciprocal square root, sin / cos and log2 / exp2 instructions
 *   (1 ulp each, 2^-18 absolute for sin / cos) and fused multiply-adds: what a GLSL compiler is allowed to do with the
 *   reference's shaders.  Camera rays, ray / triangle intersection and accumulation stay bit-exact, so the primary
 *   hit ids of a frame do not change; a BSDF evaluation differs from the bit-exact one by 2e-7 relative at the median and
 *   by less than 5e-4 in 99.9 % of random evaluations (ill-conditioned inputs, e.g. next to total internal reflection,
 *   up to 1e-2), and a converged image (1024 samples per pixel) stays within relRMSE 1e-3 of the CPU oracle's: measured
 *   1.2e-4 (tests/test_gpu_precision.py).  A path whose lobe choice or next hit
 *   flips on a last-bit difference continues elsewhere, as it would between two Vulkan drivers.
 * Frames already queued keep the mode they were traced with. */
#define PT_PRECISION_EXACT 0
#define PT_PRECISION_FAST 1
int pt_set_precision(pt_ctx* ctx, int mode);
/* vkCmdTraceRaysKHR(width, height, 1) with SceneBuffer.FrameIndex = frame_index, binding 7
 * (RayTracingLayer.cpp:274-283, RayGen.glsl:312-373).  Asynchronous: the frame takes the camera
 * and parameters in force at the call and is queued; up to pt_set_batch consecutive frames are
 * launched together, when the batch is full or at the next call that reads, waits for or changes
 * what the queued frames use (every other entry point except pt_set_camera).  Frames are added to
 * the accumulation image in call order whatever the batching, so images do not depend on it.
 * A launch error of a queued frame is returned by the call that launches it.
 * The caller sequences frame_index (1-based; 1 clears the accumulation image). */
int pt_trace(pt_ctx* ctx, uint32_t frame_index);
/* waits for the stream (the reference's Submit + FlushCommandBuffer, RayTracingLayer.cpp:317,424) */
int pt_sync(pt_ctx* ctx);
/* o_Image / o_AccumulationImage read-back and restore (bindings 1 and 2); width*height*4 floats. */
int pt_read_image(pt_ctx* ctx, float* rgba);
int pt_read_accum(pt_ctx* ctx, float* rgba);
int pt_write_accum(pt_ctx* ctx, const float* rgba);
/* o_Image = accum.rgb / accum.a for every pixel (RayGen.glsl:367-372), for an accumulation
 * image that was merged from several GPUs. */
int pt_resolve(pt_ctx* ctx);
/* A submesh transform was edited and the acceleration structure is recreated (RayTracingLayer.cpp:569-604,
 * "Automatically update AS"): new per-submesh object-to-world matrices in the layout of pt_scene_desc.submesh_xforms,
 * then the hierarchy is rebuilt over the moved triangles, as the reference does.  With PT_UPDATE_REFIT in
 * build_flags the hierarchy keeps its topology and only triangles, boxes and node grids are recomputed: several
 * times faster than a rebuild and exact, but the hierarchy's quality decays as objects move far from where they
 * were when it was built (stats->sah_cost tells). */
#define PT_UPDATE_REFIT 0x100u
int pt_update_transforms(pt_ctx* ctx, const float* submesh_xforms, uint32_t num_submeshes, uint32_t build_flags, pt_accel_stats* stats);
/* PostProcessingPass (RayTracingLayer.cpp:286-318, PostProcessing.glsl:14-55): exposure (the reference's
 * default is 0.8, RayTracingLayer.h:92), ACES fit, gamma 2.2, stored as RGBA8 into an image the context owns
 * (m_PostProcessingImage).  Runs on the main stream after the frames issued so far. */
int pt_postprocess(pt_ctx* ctx, float exposure);
int pt_read_postprocessed(pt_ctx* ctx, uint8_t* rgba8);      /* width*height*4 bytes */
void* pt_device_postprocessed(pt_ctx* ctx);
/* Environment sources for binding 10.  All replace the cube map set by pt_scene_upload (call them after it).
 * pt_env_upload: a cube the caller made, 6*face_size*face_size*4 floats, faces +X -X +Y -Y +Z -Z (the descriptor write of
 *   m_RadianceMap, RayTracingLayer.cpp:258, without re-uploading the scene).
 * pt_env_from_equirect: what the engine's TextureCube does with its .hdr radiance image (RayTracingLayer.cpp:50-54,
 *   EquirectangularToCubeMap.glsl:11-40): rgba = width*height*4 floats, row 0 = top; the cube holds binary16 values.
 * pt_env_preetham: the Preetham sky pass (RayTracingLayer.cpp:56-76,366-381, PreethamSky.glsl:17-146; defaults
 *   turbidity 3.14, azimuth 0, inclination 0 at RayTracingLayer.h:88, 2048 texels per face).
 * pt_read_env: the current cube, 6*n*n*4 floats, faces +X -X +Y -Y +Z -Z; rgba may be NULL to query n only. */
int pt_env_upload(pt_ctx* ctx, const float* rgba, uint32_t face_size);
int pt_env_from_equirect(pt_ctx* ctx, const float* rgba, uint32_t width, uint32_t height, uint32_t face_size);
int pt_env_preetham(pt_ctx* ctx, float turbidity, float azimuth, float inclination, uint32_t face_size);
int pt_read_env(pt_ctx* ctx, float* rgba, uint32_t* face_size);
/* (gl_InstanceCustomIndexEXT, gl_PrimitiveID) of the first sample's primary ray of the last
 * frame, per pixel; 0xFFFFFFFF twice for a miss.  width*height*2 words. */
int pt_read_primary_ids(pt_ctx* ctx, uint32_t* submesh_prim_pairs);
/* number of traceRayEXT-equivalents issued since the context was created */
int pt_ray_count(pt_ctx* ctx, uint64_t* rays);
/* A frame is traced as `chains` (1, 2, 4, 8 or 16) independent pipelines, each with a share of the
 * pixel tiles and a HIP stream of its own, so that the tail of one pipeline's traversal launch
 * overlaps the others' work; images do not depend on the setting.  Chains overlap only if HIP has
 * a hardware queue for each.  HIP gives a process 4 unless GPU_MAX_HW_QUEUES says otherwise, and reads the
 * variable when its runtime starts (16 is the measured optimum, more queues than chains cost throughput).
 * A value the caller put in the environment is taken at its word: the default number of chains is then
 * 4, 8 or 16 for < 8, >= 8, >= 16.  With the variable unset the library sets it to 16 when it is loaded,
 * which only helps if no HIP call has been made yet in the process; pt_create therefore measures how many
 * of its streams really run side by side (a millisecond) and chooses the default from that.  After a
 * successful pt_set_chains, pt_last_warning holds a text when chains exceeds those queues, else an empty
 * string; pt_last_error is left alone.  No counterpart in the reference, whose scheduling is the Vulkan
 * driver's. */
int pt_set_chains(pt_ctx* ctx, uint32_t chains);
/* hardware queues the context found for its streams (the caller's GPU_MAX_HW_QUEUES, else measured: 4, 8 or 16) */
uint32_t pt_hardware_queues(pt_ctx* ctx);
/* Up to `frames` (1..8) consecutive pt_trace calls share their kernel launches: the paths of all
 * of them fill one set of ray queues, so a launch has that many times the rays to hide its tail
 * behind.  1 launches every frame at its pt_trace call (lowest latency for an interactive viewer
 * that displays every frame); the default, 8, is the throughput setting for progressive
 * accumulation.  No counterpart in the reference, which records one vkCmdTraceRaysKHR per frame. */
int pt_set_batch(pt_ctx* ctx, uint32_t frames);
/* mode 1: every kernel launch of later pt_trace calls is bracketed by HIP events on the stream
 * it runs on; pt_last_trace_timing waits for the frames issued so far and reports their
 * per-kernel time, summed since its previous call (or since the mode was set).
 * mode 2: as 1, and the traversal kernel also counts the nodes and triangles it touches
 * (slower; for the bytes-per-ray figure, not for timing).  mode 0: off. */
int pt_set_profiling(pt_ctx* ctx, int mode);
int pt_last_trace_timing(pt_ctx* ctx, pt_trace_timing* out);
/* the context's main HIP stream (hipStream_t as void*).  Work the caller enqueues on it after this
 * call runs after every pt_trace issued so far, and later pt_trace calls run after that work. */
void* pt_stream(pt_ctx* ctx);
/* The opposite direction: everything the context does after this call (pt_trace, pt_resolve, pt_postprocess,
 * read-backs) runs after the work enqueued so far on `stream` (a hipStream_t as void*; NULL = the null stream),
 * without blocking the host.  For callers that write the bound images on a stream of their own, e.g. a collective
 * (RCCL reduce of the accumulation image, multi-GPU merge) followed by pt_resolve. */
int pt_wait_stream(pt_ctx* ctx, void* stream);
/* Multi-GPU merge: one ncclReduce (sum, float32, width*height*4 values, in place) of the accumulation image currently
 * rendered into, to rank `root` of `nccl_comm` (an ncclComm_t of RCCL as void*), on the context's main stream after every frame
 * issued so far; asynchronous.  Every rank of the communicator calls it, one context per GPU; afterwards the root calls
 * pt_resolve.  A frame's samples depend only on (pixel, FrameIndex) and frames add (RayGen.glsl:314-315,351-359), so ranks
 * that rendered disjoint sets of frame indices merge to exactly the single-GPU sample set.  The reference is single-GPU
 * and has no counterpart.  RCCL is not linked into this library: ncclReduce is taken from the RCCL already present in the
 * process (linked or loaded), else from librccl.so on the loader's path, or from the file PT_RCCL_LIB names. */
int pt_reduce_accum(pt_ctx* ctx, void* nccl_comm, int root);
/* device pointers of the images currently rendered into */
void* pt_device_image(pt_ctx* ctx);
void* pt_device_accum(pt_ctx* ctx);

/* ---- inspection hooks used by the parity tests ------------------------------------- */
/* closest hit of caller-supplied rays: rays = n*8 floats (o.xyz, tmin, d.xyz, tmax);
 * hits = n*5 words: t,u,v (float bits), submesh, prim; miss = (tmax,0,0,~0,~0).  tmin must be >= 0.
 * counters (may be NULL) = 2 uint64: hierarchy nodes visited, triangles tested. */
int pt_debug_intersect(pt_ctx* ctx, const float* rays, uint32_t n, uint32_t* hits, uint64_t* counters);
/* pt_debug_intersect for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same rays and same hit records
 * (n*8 floats in, n*5 words out; a negative or NaN tmin is taken as 0), but both in DEVICE memory supplied by the caller, and the call allocates
 * nothing: the traversal's fetch cursor and stack spill area live in a fixed 32 MiB array inside the library
 * (one per GPU, held until the process ends).  Limits, refused with PT_ERR_TOO_LARGE: n <= 2^24 rays; a hierarchy
 * of at most 18 levels of 8-wide nodes (pt_accel_stats.max_depth; the 3 M-triangle test scene has 11).
 * Asynchronous on the context's main stream: read device_hits after pt_sync or behind pt_stream.  Frames queued
 * by pt_trace must have been flushed (pt_sync) before the call.  device_hits must not overlap device_rays (PT_ERR_INVALID).
 * Every call on a GPU shares that GPU's array: a call is refused with PT_ERR_INVALID while a query of another context on the
 * same GPU has not finished (queries of one context follow each other on its stream); work the caller orders behind the
 * query on streams of its own is the caller's to keep apart. */
int pt_intersect_noalloc(pt_ctx* ctx, const float* device_rays, uint32_t n, uint32_t* device_hits);
/* the deterministic functions used by the kernels: fn 0 sin, 1 cos, 2 log, 3 exp, 4 pow(x,y), 5 atan2(x,y),
 * 6 acos, 7 tan, 8 binary32 -> binary16 -> binary32 rounding */
int pt_debug_math(pt_ctx* ctx, int fn, const float* x, const float* y, float* out, uint32_t n);
/* DisneySample / DisneyEval (Disney.glsl:150-359) for n independent inputs.
 * surf = n*17 floats: albedo3 metallic roughness subsurface specularTint sheen sheenTint
 * clearcoat clearcoatRoughness specTrans ior ax ay eta anisotropic; V, N, L = n*3;
 * seeds = n (updated in place); f = n*3; pdf = n. */
int pt_debug_disney_sample(pt_ctx* ctx, uint32_t n, const float* surf, const float* V, const float* N,
                           uint32_t* seeds, float* f, float* L, float* pdf);
int pt_debug_disney_eval(pt_ctx* ctx, uint32_t n, const float* surf, const float* V, const float* N,
                         const float* L, float* f, float* pdf);
/* copies the built hierarchy to the host: nodes = num_nodes*32 words, tris = num_references*12 words */
int pt_debug_read_bvh(pt_ctx* ctx, uint32_t* nodes, uint32_t* tris);

#ifdef __cplusplus
}
#endif
#endif
