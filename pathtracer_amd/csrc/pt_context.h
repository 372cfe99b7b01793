// Host-side state of one rendering context (one GPU, one stream).
#pragma once
#include "../../include/pathtracer_amd.h"
#include "pt_types.h"
#include <hip/hip_runtime.h>
#include <string>
#include <vector>

struct pt_ctx {
    int device = 0;
    int num_cus = 0;
    hipStream_t stream = nullptr;
    std::string error;

    // scene tables (device) and their host-side sizes
    float* d_vertices = nullptr;
    uint32_t* d_indices = nullptr;
    uint32_t* d_submeshes = nullptr;
    float* d_xforms = nullptr;
    float* d_materials = nullptr;
    uint8_t* d_tex_pool = nullptr;
    uint32_t* d_tex_table = nullptr;
    float* d_env = nullptr;
    uint32_t* d_sub_tri_offset = nullptr;   // num_submeshes + 1
    uint32_t num_vertices = 0, num_indices = 0, num_submeshes = 0, num_materials = 0, num_textures = 0, env_n = 0;
    uint32_t num_tris = 0;
    bool scene_loaded = false;

    // hierarchy
    ptd::Node8* d_nodes = nullptr;
    ptd::TriRec* d_tris = nullptr;
    uint32_t num_nodes = 0;
    bool accel_built = false;
    pt_accel_stats accel_stats{};

    // images
    uint32_t width = 0, height = 0;
    float* d_image_own = nullptr;
    float* d_accum_own = nullptr;
    float* d_image = nullptr;     // own or bound
    float* d_accum = nullptr;
    uint2* d_primary = nullptr;

    // wavefront state
    float4* d_rayO[2] = {nullptr, nullptr};
    float4* d_rayD[2] = {nullptr, nullptr};
    float4* d_hits = nullptr;
    float4* d_stA = nullptr;
    float4* d_stB = nullptr;
    float4* d_stC = nullptr;
    uint32_t shard_cap = 0;         // ray-queue slots per shard
    uint32_t* d_counts = nullptr;   // PT_SHARDS per round, rounds + 1 rows of queue sizes, then as many rows of fetch cursors
    uint32_t counts_cap = 0;
    uint2* d_spill = nullptr;       // traversal stack spill area, [wave][entry][lane]
    uint32_t spill_waves = 0;
    unsigned long long* d_total_rays = nullptr;
    unsigned long long* d_stats = nullptr;  // [0] nodes visited, [1] triangles tested, [2] fetch cursor of pt_debug_intersect

    pt_params params{};
    float camera[80];
    bool camera_set = false;

    // per-kernel timing of the last frame
    int profiling = 0;           // 0 off, 1 events, 2 events + traversal counters
    std::vector<hipEvent_t> events;
    pt_trace_timing timing{};
    uint32_t timing_rounds = 0;
    bool timing_pending = false;
    unsigned long long rays_before_frame = 0;
};

namespace ptd {
int set_error(pt_ctx* ctx, int code, const std::string& msg);
int build_accel(pt_ctx* ctx, uint32_t flags);
int launch_trace(pt_ctx* ctx, uint32_t frame_index);
int launch_resolve(pt_ctx* ctx);
int debug_intersect(pt_ctx* ctx, const float* rays, uint32_t n, uint32_t* hits, uint64_t* counters);
int debug_math(pt_ctx* ctx, int fn, const float* x, const float* y, float* out, uint32_t n);
int debug_disney(pt_ctx* ctx, bool sample, uint32_t n, const float* surf, const float* V, const float* N,
                 const float* Lin, uint32_t* seeds, float* f, float* Lout, float* pdf);
SceneView scene_view(const pt_ctx* ctx);
} // namespace ptd

#define PT_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return ptd::set_error(ctx, PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
