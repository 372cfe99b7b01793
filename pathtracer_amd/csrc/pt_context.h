// Host-side state of one rendering context (one GPU; one main stream plus the streams of the other chains).
#pragma once
#include "../../include/pathtracer_amd.h"
#include "pt_types.h"
#include <hip/hip_runtime.h>
#include <string>
#include <utility>
#include <vector>

// events and queue sizes of one profiled frame; a few are kept so that profiled frames overlap like unprofiled ones
struct pt_prof_slot {
    std::vector<hipEvent_t> ev[PT_MAX_CHAINS];
    uint32_t* h_counts = nullptr;     // pinned copy of the frame's queue sizes
    size_t h_words = 0;
    uint32_t rounds = 0, chains = 0;
    uint32_t layout_rounds = 0;       // rounds the counter block is laid out for (>= rounds)
    bool pending = false;
};
#define PT_PROF_SLOTS 4

struct pt_ctx {
    int device = 0;
    int num_cus = 0;
    hipStream_t stream = nullptr;            // main stream = stream of chain 0
    std::string error;
    std::string warning;                     // advice that is not an error (pt_last_warning)
    uint32_t hw_queues = 4;                  // hardware queues the process has: from the caller's GPU_MAX_HW_QUEUES, else measured
    std::string hw_queues_source;

    // A frame is traced as `chains` independent pipelines, each owning PT_SHARDS / chains ray-queue shards (and the
    // pixel tiles that map to them) and running raygen -> (extend -> shade)* on its own stream.  The long tail of
    // one chain's traversal launch then overlaps the other chains' work.  Chains are not joined at the end of
    // pt_trace: consecutive frames flow through each chain back to back; anything that reads results joins first.
    uint32_t chains = 1;
    hipStream_t chain_stream[PT_MAX_CHAINS] = {};
    hipEvent_t chain_done[PT_MAX_CHAINS] = {};
    hipEvent_t fork_event = nullptr;
    hipEvent_t wait_event = nullptr;   // pt_wait_stream: marks the caller's stream
    bool main_dirty = true;      // the main stream got work the other chains have not yet been ordered after
    bool join_pending = false;   // the other chains got work the main stream has not yet been ordered after

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
    uint32_t num_refs = 0;                   // triangle records in leaf order: num_tris, or more when the builder split triangles
    bool scene_loaded = false;

    // hierarchy
    ptd::Node8* d_nodes = nullptr;
    ptd::TriRec* d_tris = nullptr;
    ptd::ShadeTri* d_shade_tris = nullptr;   // per triangle in leaf order, see ShadeTri
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
    uint32_t* d_post = nullptr;   // post-processed RGBA8 image, allocated by the first pt_postprocess

    // wavefront state
    float4* d_rayO[2] = {nullptr, nullptr};
    float4* d_rayD[2] = {nullptr, nullptr};
    float4* d_hits = nullptr;
    float* d_hit_det = nullptr;                // determinant of each hit: the barycentric weights are divided out by k_shade
    float4* d_stA[2] = {nullptr, nullptr};     // path state travels with the ray: per queue slot (throughput, seed)
    float4* d_stB[2] = {nullptr, nullptr};     // and (radiance, bounce | sample << 16)
    float4* d_stC = nullptr;                   // per path: colour of the pixel's finished samples
    float4* d_partial = nullptr;    // finished colour per (frame slot, pixel)
    ptd::FrameView* d_views = nullptr;   // per chain PT_MAX_BATCH frame views, written on the chain's stream
    // frames handed to pt_trace and not launched yet: up to `batch` of them share their launches (more rays per launch).
    // Everything that synchronises, reads or changes what the kernels read launches them first.
    std::vector<ptd::FrameView> pending;
    uint32_t batch = PT_MAX_BATCH;
    uint32_t shard_cap = 0;         // ray-queue slots per shard
    uint32_t* d_counts = nullptr;   // per chain: rounds + 1 rows of queue sizes (one per shard of the chain), then as many rows of fetch cursors
    uint32_t counts_cap = 0;
    uint32_t defer_rounds = 2;      // rounds a path may fall behind because its ray was suspended at the tail of a launch (PT_DEFER; 0: off)
    bool defer_forced = false;      // PT_DEFER was set: every batch suspends (or none); otherwise launch_trace decides per batch
    uint32_t* d_suspend = nullptr;  // suspended traversals, [chain][2][shard of the chain][slot][16 + 2 * stack_entries]
    size_t suspend_words = 0;
    uint2* d_spill = nullptr;       // traversal stack spill area, [chain][wave][entry][lane]
    size_t spill_entries = 0;       // allocated entries of d_spill
    std::vector<uint32_t> level_start;  // node numbers at which the levels of the built hierarchy begin, and the node count
    uint32_t stack_entries = 64;    // stack entries a ray can need in the built hierarchy (7 per level + 1), a multiple of 8
    unsigned long long* d_wave_trace = nullptr;   // PT_WAVE_TRACE=<file>: per-wave timeline of the extend launches
    size_t wave_trace_len = 0;
    uint32_t wave_trace_rounds = 0, wave_trace_waves = 0;
    unsigned long long* d_total_rays = nullptr;
    uint32_t* h_fault = nullptr;             // pinned host word the traversal kernels set when a wave reached its iteration bound
    uint32_t* d_fault = nullptr;             // the same word as the device sees it
    uint32_t iter_cap = 1u << 24;            // bound of a wave's traversal loop (PT_ITER_CAP in the environment: tests)
    void* noalloc_scratch = nullptr;         // address, on this context's GPU, of the fixed scratch array of pt_intersect_noalloc
    hipEvent_t noalloc_done = nullptr;       // recorded after each pt_intersect_noalloc launch: other contexts on the GPU check it
    unsigned long long* d_stats = nullptr;  // [0] nodes visited, [1] triangles tested, [2] fetch cursor of pt_debug_intersect

    pt_params params{};
    int precision = PT_PRECISION_EXACT;      // arithmetic of the shading stage (pt_set_precision)
    float camera[80];
    bool camera_set = false;

    // per-kernel timing of the frames traced since the last read-out (the timing window)
    int profiling = 0;           // 0 off, 1 events, 2 events + traversal counters
    pt_prof_slot prof_slot[PT_PROF_SLOTS];
    uint32_t prof_next = 0;
    hipEvent_t prof_ref = nullptr;                    // time zero of the window, on the main stream
    bool window_open = false;
    pt_trace_timing timing{};                         // sums over the window's collected frames
    std::vector<std::pair<float, float>> ext_spans;   // traversal launches of the window, ms since prof_ref
    float win_first = 0.0f, win_last = 0.0f;
    unsigned long long rays_before_frame = 0;
};

namespace ptd {
int set_error(pt_ctx* ctx, int code, const std::string& msg);
int build_accel(pt_ctx* ctx, uint32_t flags);
int refit_accel(pt_ctx* ctx);
int launch_trace(pt_ctx* ctx);     // launches the pending frames
int launch_resolve(pt_ctx* ctx);
int launch_postprocess(pt_ctx* ctx, float exposure);
int env_from_equirect(pt_ctx* ctx, const float* rgba, uint32_t w, uint32_t h, uint32_t face_size);
int env_preetham(pt_ctx* ctx, float turbidity, float azimuth, float inclination, uint32_t face_size);
int env_upload(pt_ctx* ctx, const float* rgba, uint32_t face_size);
int sync_streams(pt_ctx* ctx);     // host waits for every chain
int check_fault(pt_ctx* ctx);      // PT_ERR_TRAVERSAL if a traversal wave reached its iteration bound since the last check
int collect_slot(pt_ctx* ctx, pt_prof_slot& slot);   // waits for that frame, adds it to the timing window
int join_chains(pt_ctx* ctx);      // main stream waits for the other chains (no host wait)
int debug_intersect(pt_ctx* ctx, const float* rays, uint32_t n, uint32_t* hits, uint64_t* counters);
int noalloc_init(pt_ctx* ctx);
void noalloc_release(pt_ctx* ctx);
int intersect_noalloc(pt_ctx* ctx, const float* d_rays, uint32_t n, uint32_t* d_hits);
int debug_math(pt_ctx* ctx, int fn, const float* x, const float* y, float* out, uint32_t n);
int debug_disney(pt_ctx* ctx, bool sample, uint32_t n, const float* surf, const float* V, const float* N,
                 const float* Lin, uint32_t* seeds, float* f, float* Lout, float* pdf);
SceneView scene_view(const pt_ctx* ctx);
} // namespace ptd

// words of one chain's counter block: rounds + 1 rows of queue sizes, as many of fetch cursors, one row whose first
// word counts the deferred rays
inline size_t pt_chain_words(uint32_t rounds, uint32_t shards_per_chain) { return ((size_t)2 * (rounds + 1) + 1) * shards_per_chain; }

#define PT_NOALLOC_MAX_RAYS (1u << 24)

#define PT_SYNC(ctx)                                                                              \
    do {                                                                                          \
        int rc_ = ptd::sync_streams(ctx);                                                         \
        if (rc_) return rc_;                                                                      \
    } while (0)

#define PT_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return ptd::set_error(ctx, PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
