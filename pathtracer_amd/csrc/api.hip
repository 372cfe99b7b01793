// C ABI of the backend (declared in include/pathtracer_amd.h): context life cycle, uploads of the
// reference's descriptor-set contents (PathTracer/src/RayTracingLayer.cpp:248-263), image management
// and read-back.  No exception crosses the boundary; every entry point returns a status code.
#include "pt_context.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <new>

namespace ptd {

static thread_local std::string g_create_error;

// A frame runs as up to 16 chains, each on a HIP stream that needs a hardware queue of its own; HIP gives a process 4
// unless GPU_MAX_HW_QUEUES says otherwise, and reads the variable when its runtime starts (its first API call).  When
// the library is loaded before that (linked into the application, or dlopen'ed before any HIP call) and the variable
// is unset, it asks for 16 here, so that a host application gets the full rate without a launcher script.  Whether the
// request took effect is not known at this point (the runtime may have started earlier, e.g. a host that used torch.cuda
// or another HIP library first): pt_create then measures how many streams really run side by side (probe_hw_queues)
// instead of reading the variable back.  setenv is not thread safe against getenv in other threads: load the library
// before starting threads, or set the variable in the launcher.
static bool g_asked_for_queues = false;
__attribute__((constructor)) static void ask_for_hardware_queues()
{
    if (!getenv("GPU_MAX_HW_QUEUES")) g_asked_for_queues = setenv("GPU_MAX_HW_QUEUES", "16", 0) == 0;
}

// holds a wave for `ticks` of the 100 MHz wall clock; every wave leaves after that time whatever happens
__global__ void k_hold(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

// How many of the context's streams run side by side: one single-wave kernel that holds for 0.4 ms on each of the 16
// streams.  With a hardware queue per stream all of them overlap and the whole takes 0.4 ms; streams that share a queue
// run one after the other (8 queues: 0.8 ms, 4 queues: 1.6 ms).  Costs one to two milliseconds, once per context.
static uint32_t probe_hw_queues(pt_ctx* ctx)
{
    const float hold_ms = 0.4f;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { if (e0) (void)hipEventDestroy(e0); return 4u; }
    for (uint32_t c = 0; c < PT_MAX_CHAINS; c++) k_hold<<<1, 64, 0, ctx->chain_stream[c]>>>(1000ull);     // first launches pay for code upload
    for (uint32_t c = 0; c < PT_MAX_CHAINS; c++) (void)hipStreamSynchronize(ctx->chain_stream[c]);
    (void)hipEventRecord(e0, ctx->stream);
    for (uint32_t c = 1; c < PT_MAX_CHAINS; c++) (void)hipStreamWaitEvent(ctx->chain_stream[c], e0, 0);
    for (uint32_t c = 0; c < PT_MAX_CHAINS; c++) k_hold<<<1, 64, 0, ctx->chain_stream[c]>>>((unsigned long long)(hold_ms * 1e5f));
    for (uint32_t c = 1; c < PT_MAX_CHAINS; c++) {
        (void)hipEventRecord(ctx->chain_done[c], ctx->chain_stream[c]);
        (void)hipStreamWaitEvent(ctx->stream, ctx->chain_done[c], 0);
    }
    (void)hipEventRecord(e1, ctx->stream);
    float ms = 1e9f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) ms = 1e9f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipGetLastError();
    const float rounds = ms / hold_ms;            // streams per hardware queue
    return rounds < 1.5f ? 16u : (rounds < 3.0f ? 8u : 4u);
}

int set_error(pt_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->error = msg;
    else g_create_error = msg;
    return code;
}

template <class T>
static hipError_t dev_free(T*& p)
{
    hipError_t e = hipSuccess;
    if (p) e = hipFree(p);
    p = nullptr;
    return e;
}

static void free_scene(pt_ctx* c)
{
    (void)dev_free(c->d_vertices); (void)dev_free(c->d_indices); (void)dev_free(c->d_submeshes);
    (void)dev_free(c->d_xforms); (void)dev_free(c->d_materials); (void)dev_free(c->d_tex_pool);
    (void)dev_free(c->d_tex_table); (void)dev_free(c->d_env); (void)dev_free(c->d_sub_tri_offset);
    (void)dev_free(c->d_nodes); (void)dev_free(c->d_tris); (void)dev_free(c->d_shade_tris);
    c->scene_loaded = false;
    c->accel_built = false;
}

static void free_images(pt_ctx* c)
{
    (void)dev_free(c->d_image_own); (void)dev_free(c->d_accum_own); (void)dev_free(c->d_primary); (void)dev_free(c->d_post);
    (void)dev_free(c->d_rayO[0]); (void)dev_free(c->d_rayO[1]); (void)dev_free(c->d_rayD[0]); (void)dev_free(c->d_rayD[1]);
    (void)dev_free(c->d_hits); (void)dev_free(c->d_hit_det); (void)dev_free(c->d_stC);
    for (int i = 0; i < 2; i++) { (void)dev_free(c->d_stA[i]); (void)dev_free(c->d_stB[i]); } (void)dev_free(c->d_partial);
    c->d_image = c->d_accum = nullptr;
    c->width = c->height = 0;
}

template <class T>
static int upload(pt_ctx* ctx, T*& dst, const T* src, size_t n)
{
    PT_HIP(ctx, hipMalloc((void**)&dst, (n ? n : 1) * sizeof(T)));
    if (n) PT_HIP(ctx, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return PT_OK;
}

int collect_slot(pt_ctx* ctx, pt_prof_slot& slot)
{
    if (!slot.pending) return PT_OK;
    slot.pending = false;
    pt_trace_timing& t = ctx->timing;
    const uint32_t R = slot.rounds, C = slot.chains, spc = PT_SHARDS / C;
    const size_t chain_words = pt_chain_words(slot.layout_rounds, spc);
    for (uint32_t c = 0; c < C; c++) PT_HIP(ctx, hipEventSynchronize(slot.ev[c][2 + 2 * R]));
    // PT_SPAN_DUMP=<file>: every launch of the profiled frames as a line "chain kind round start_ms end_ms" (kind r = ray
    // generation, e = traversal, s = shading, a = accumulation; times since the window's reference event), appended
    static const char* span_path = getenv("PT_SPAN_DUMP");
    FILE* span_file = span_path ? fopen(span_path, "a") : nullptr;
    auto span = [&](uint32_t c, char kind, uint32_t r, hipEvent_t a, hipEvent_t b) {
        float t0 = 0.0f, dt = 0.0f;
        if (span_file && hipEventElapsedTime(&t0, ctx->prof_ref, a) == hipSuccess && hipEventElapsedTime(&dt, a, b) == hipSuccess)
            fprintf(span_file, "%u %c %u %.4f %.4f\n", c, kind, r, t0, t0 + dt);
    };
    for (uint32_t c = 0; span_file && c < C; c++) {
        std::vector<hipEvent_t>& ev = slot.ev[c];
        span(c, 'r', 0, ev[0], ev[1]);
        for (uint32_t r = 0; r < R; r++) { span(c, 'e', r, ev[1 + 2 * r], ev[2 + 2 * r]); span(c, 's', r, ev[2 + 2 * r], ev[3 + 2 * r]); }
        span(c, 'a', 0, ev[1 + 2 * R], ev[2 + 2 * R]);
    }
    if (span_file) { fputs("#\n", span_file); fclose(span_file); }
    for (uint32_t c = 0; c < C; c++) {
        std::vector<hipEvent_t>& ev = slot.ev[c];
        float ms = 0.0f;
        PT_HIP(ctx, hipEventElapsedTime(&ms, ev[0], ev[1]));
        t.raygen_ms += ms;
        for (uint32_t r = 0; r < R; r++) {
            // the frame's queue sizes tell which rounds had work
            uint64_t queued = 0;
            for (uint32_t k = 0; k < spc; k++) queued += slot.h_counts[c * chain_words + (size_t)r * spc + k];
            float e = 0.0f, sh = 0.0f;
            PT_HIP(ctx, hipEventElapsedTime(&e, ev[1 + 2 * r], ev[2 + 2 * r]));
            PT_HIP(ctx, hipEventElapsedTime(&sh, ev[2 + 2 * r], ev[3 + 2 * r]));
            if (queued > 0) {
                t.extend_ms += e; t.shade_ms += sh;
                t.extend_launches++; t.shade_launches++;
                float b0 = 0.0f;
                PT_HIP(ctx, hipEventElapsedTime(&b0, ctx->prof_ref, ev[1 + 2 * r]));
                ctx->ext_spans.push_back({b0, b0 + e});
            }
            t.rays += queued;
        }
        t.rays -= slot.h_counts[c * chain_words + (size_t)2 * (slot.layout_rounds + 1) * spc];     // deferred rays were queued twice
        float a = 0.0f, b = 0.0f;
        PT_HIP(ctx, hipEventElapsedTime(&a, ctx->prof_ref, ev[0]));
        PT_HIP(ctx, hipEventElapsedTime(&b, ctx->prof_ref, ev[2 + 2 * R]));
        if (t.chains == 0 || a < ctx->win_first) ctx->win_first = a;
        if (t.chains == 0 || b > ctx->win_last) ctx->win_last = b;
        t.chains = C;
    }
    return PT_OK;
}

// collects every frame still in flight and closes the timing window: totals into ctx->timing
static int finish_timing(pt_ctx* ctx)
{
    if (!ctx->pending.empty()) {
        int rc = launch_trace(ctx);
        if (rc) return rc;
    }
    if (!ctx->window_open) return PT_OK;
    for (uint32_t k = 0; k < PT_PROF_SLOTS; k++) {
        int rc = collect_slot(ctx, ctx->prof_slot[(ctx->prof_next + k) % PT_PROF_SLOTS]);
        if (rc) return rc;
    }
    ctx->window_open = false;
    pt_trace_timing& t = ctx->timing;
    t.total_ms = ctx->win_last - ctx->win_first;
    std::sort(ctx->ext_spans.begin(), ctx->ext_spans.end());
    float cur_end = -1e30f;
    t.extend_busy_ms = 0.0f;
    for (auto& sp : ctx->ext_spans) {
        if (sp.first > cur_end) { t.extend_busy_ms += sp.second - sp.first; cur_end = sp.second; }
        else if (sp.second > cur_end) { t.extend_busy_ms += sp.second - cur_end; cur_end = sp.second; }
    }
    ctx->ext_spans.clear();
    if (ctx->profiling == 2) {
        PT_SYNC(ctx);
        unsigned long long hs[16] = {0};
        PT_HIP(ctx, hipMemcpy(hs, ctx->d_stats, 128, hipMemcpyDeviceToHost));
        t.nodes_visited = hs[0];
        t.tris_tested = hs[1];
        if (getenv("PT_WAVE_STATS"))
            fprintf(stderr, "wave stats: node phases %llu (lanes %llu, %.1f/64), lanes without a ray %.1f/64, leaf phases %llu (lanes %llu, %.1f/64), refills %llu, wave ms: draining %.1f of resident %.1f, longest ray %llu steps; node phases by active lanes <=8: %llu, <=16: %llu, <=32: %llu, <=64: %llu\n",
                    hs[3], hs[4], hs[3] ? (double)hs[4] / hs[3] : 0.0, hs[3] ? (double)hs[15] / hs[3] : 0.0, hs[5], hs[6], hs[5] ? (double)hs[6] / hs[5] : 0.0, hs[7], hs[8] * 1e-5, hs[9] * 1e-5, hs[10], hs[11], hs[12], hs[13], hs[14]);
        const char* path = getenv("PT_WAVE_TRACE");
        if (path && ctx->d_wave_trace) {
            size_t len = (size_t)ctx->wave_trace_rounds * ctx->wave_trace_waves * 4;
            std::vector<unsigned long long> buf(len + 2);
            buf[0] = ctx->wave_trace_rounds; buf[1] = ctx->wave_trace_waves;
            PT_HIP(ctx, hipMemcpy(buf.data() + 2, ctx->d_wave_trace, len * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            if (FILE* f = fopen(path, "wb")) { fwrite(buf.data(), sizeof(unsigned long long), buf.size(), f); fclose(f); }
        }
    }
    return PT_OK;
}

int check_fault(pt_ctx* ctx)
{
    if (!ctx->h_fault || *(volatile uint32_t*)ctx->h_fault == 0u) return PT_OK;
    *(volatile uint32_t*)ctx->h_fault = 0u;
    return set_error(ctx, PT_ERR_TRAVERSAL, "a traversal wave reached its iteration bound (" + std::to_string(ctx->iter_cap) +
                     " loop iterations) and left hit records unwritten: "
                     "the frames and ray queries since the last synchronisation are not valid");
}

int sync_streams(pt_ctx* ctx)
{
    if (!ctx->pending.empty()) {
        int rc = launch_trace(ctx);
        if (rc) return rc;
    }
    for (uint32_t c = PT_MAX_CHAINS; c-- > 0;)
        if (ctx->chain_stream[c]) PT_HIP(ctx, hipStreamSynchronize(ctx->chain_stream[c]));
    ctx->join_pending = false;
    return check_fault(ctx);
}

int join_chains(pt_ctx* ctx)
{
    if (!ctx->pending.empty()) {
        int rc = launch_trace(ctx);
        if (rc) return rc;
    }
    if (!ctx->join_pending) return PT_OK;
    for (uint32_t c = 1; c < ctx->chains; c++) {
        PT_HIP(ctx, hipEventRecord(ctx->chain_done[c], ctx->chain_stream[c]));
        PT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->chain_done[c], 0));
    }
    ctx->join_pending = false;
    return PT_OK;
}

} // namespace ptd

using namespace ptd;

extern "C" {

int pt_create(int device_ordinal, pt_ctx** out)
{
    if (!out) return set_error(nullptr, PT_ERR_INVALID, "pt_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) return set_error(nullptr, PT_ERR_NO_DEVICE, "no HIP device is visible");
    if (device_ordinal < 0 || device_ordinal >= count) return set_error(nullptr, PT_ERR_NO_DEVICE, "device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return set_error(nullptr, PT_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return set_error(nullptr, PT_ERR_NO_DEVICE, std::string("kernels are built for gfx950 only; device is ") + prop.gcnArchName);
    if (hipSetDevice(device_ordinal) != hipSuccess) return set_error(nullptr, PT_ERR_HIP, "hipSetDevice failed");
    pt_ctx* ctx = new (std::nothrow) pt_ctx();
    if (!ctx) return set_error(nullptr, PT_ERR_INVALID, "out of host memory");
    ctx->device = device_ordinal;
    ctx->num_cus = prop.multiProcessorCount;
    pt_default_params(&ctx->params);
    hipError_t es = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    ctx->chain_stream[0] = ctx->stream;
    for (uint32_t c = 1; c < PT_MAX_CHAINS && es == hipSuccess; c++) {
        es = hipStreamCreateWithFlags(&ctx->chain_stream[c], hipStreamNonBlocking);
        if (es == hipSuccess) es = hipEventCreateWithFlags(&ctx->chain_done[c], hipEventDisableTiming);
    }
    // Chains only overlap when each has a hardware queue of its own: HIP gives a process 4 unless GPU_MAX_HW_QUEUES
    // (read when the runtime starts) says otherwise.  A value the caller set is taken at its word; when the library set
    // it itself (or nobody did) the overlap is measured
    if (es == hipSuccess) {
        const char* e = getenv("GPU_MAX_HW_QUEUES");
        if (e && !g_asked_for_queues) {
            ctx->hw_queues = (uint32_t)atoi(e);
            ctx->hw_queues_source = "GPU_MAX_HW_QUEUES=" + std::string(e) + " in the environment";
        } else {
            ctx->hw_queues = probe_hw_queues(ctx);
            ctx->hw_queues_source = "measured: " + std::to_string(ctx->hw_queues) + " of 16 streams ran side by side";
        }
    }
    const uint32_t hwq = ctx->hw_queues;
    ctx->chains = hwq >= 16 ? 16u : (hwq >= 8 ? 8u : 4u);
    if (const char* e = getenv("PT_DEFER")) { int v = atoi(e); ctx->defer_rounds = v < 0 ? 0u : (v > 8 ? 8u : (uint32_t)v); ctx->defer_forced = true; }
    if (const char* e = getenv("PT_CHAINS")) {
        uint32_t c = (uint32_t)atoi(e);
        if (c >= 1 && c <= PT_MAX_CHAINS && PT_SHARDS % c == 0) ctx->chains = c;
    }
    if (es == hipSuccess) es = hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming);
    if (const char* e = getenv("PT_BATCH")) { int v = atoi(e); ctx->batch = v < 1 ? 1u : (v > PT_MAX_BATCH ? (uint32_t)PT_MAX_BATCH : (uint32_t)v); }
    if (es == hipSuccess) es = hipMalloc((void**)&ctx->d_views, sizeof(FrameView) * PT_MAX_BATCH * PT_MAX_CHAINS);
    if (es == hipSuccess) es = hipHostMalloc((void**)&ctx->h_fault, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (es == hipSuccess) { *ctx->h_fault = 0u; es = hipHostGetDevicePointer((void**)&ctx->d_fault, ctx->h_fault, 0); }
    if (const char* e = getenv("PT_ITER_CAP")) { long v = atol(e); if (v >= 1 && v <= (1l << 24)) ctx->iter_cap = (uint32_t)v; }
    hipError_t e1 = hipMalloc((void**)&ctx->d_total_rays, 8);
    hipError_t e2 = hipMalloc((void**)&ctx->d_stats, 128);
    if (es != hipSuccess || e1 != hipSuccess || e2 != hipSuccess) { pt_destroy(ctx); return set_error(nullptr, PT_ERR_HIP, "context allocation failed"); }
    (void)hipMemset(ctx->d_total_rays, 0, 8);
    (void)hipMemset(ctx->d_stats, 0, 128);
    if (noalloc_init(ctx) != PT_OK) { std::string msg = ctx->error; pt_destroy(ctx); return set_error(nullptr, PT_ERR_HIP, msg); }
    *out = ctx;
    return PT_OK;
}

void pt_destroy(pt_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)sync_streams(ctx);
    noalloc_release(ctx);
    free_scene(ctx);
    free_images(ctx);
    (void)dev_free(ctx->d_counts);
    (void)dev_free(ctx->d_spill);
    (void)dev_free(ctx->d_views);
    (void)dev_free(ctx->d_suspend);
    (void)dev_free(ctx->d_wave_trace);
    (void)dev_free(ctx->d_total_rays);
    if (ctx->h_fault) (void)hipHostFree(ctx->h_fault);
    (void)dev_free(ctx->d_stats);
    for (pt_prof_slot& slot : ctx->prof_slot) {
        for (auto& v : slot.ev)
            for (hipEvent_t e : v) (void)hipEventDestroy(e);
        if (slot.h_counts) (void)hipHostFree(slot.h_counts);
    }
    if (ctx->prof_ref) (void)hipEventDestroy(ctx->prof_ref);
    for (uint32_t c = 1; c < PT_MAX_CHAINS; c++) {
        if (ctx->chain_done[c]) (void)hipEventDestroy(ctx->chain_done[c]);
        if (ctx->chain_stream[c]) (void)hipStreamDestroy(ctx->chain_stream[c]);
    }
    if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
    if (ctx->wait_event) (void)hipEventDestroy(ctx->wait_event);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* pt_last_error(pt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }
const char* pt_last_warning(pt_ctx* ctx) { return ctx ? ctx->warning.c_str() : ""; }
uint32_t pt_hardware_queues(pt_ctx* ctx) { return ctx ? ctx->hw_queues : 0u; }

void pt_default_params(pt_params* p)
{
    p->spp_per_frame = 5;
    p->max_bounces = 20;
    p->rr_depth = 0;
    p->sky_scale = 10.0f;
    p->eps = 0.0003f;
    p->tmin = 0.00001f;
    p->tmax = 1e27f;
}

int pt_scene_upload(pt_ctx* ctx, const pt_scene_desc* s)
{
    if (!ctx || !s) return PT_ERR_INVALID;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    // every table is checked against its own count
    if ((s->num_vertices && !s->vertices) || (s->num_indices && !s->indices) || (s->num_materials && !s->materials) ||
        (s->num_textures && !s->textures) || (s->num_submeshes && (!s->submeshes || !s->submesh_index_counts || !s->submesh_xforms)))
        return set_error(ctx, PT_ERR_INVALID, "pt_scene_upload: a table with a non-zero count is NULL");
    if (s->env_rgba && s->env_face_size > 8192) return set_error(ctx, PT_ERR_TOO_LARGE, "pt_scene_upload: env_face_size must not exceed 8192");
    // validate every index the kernels will follow, so that no launch can read out of bounds
    std::vector<uint32_t> tri_off(s->num_submeshes + 1, 0);
    for (uint32_t i = 0; i < s->num_submeshes; i++) {
        const uint32_t* sm = s->submeshes + i * 4;
        uint32_t cnt = s->submesh_index_counts[i];
        if (sm[0] != 0) return set_error(ctx, PT_ERR_INVALID, "submesh bufferIndex must be 0 (one vertex and one index buffer)");
        if (cnt % 3) return set_error(ctx, PT_ERR_INVALID, "submesh index count is not a multiple of 3");
        if ((uint64_t)sm[2] + cnt > s->num_indices) return set_error(ctx, PT_ERR_INVALID, "submesh index range exceeds the index buffer");
        if (sm[3] >= s->num_materials) return set_error(ctx, PT_ERR_INVALID, "submesh material index out of range");
        for (uint32_t k = 0; k < cnt; k++)
            if ((uint64_t)s->indices[sm[2] + k] + sm[1] >= s->num_vertices)
                return set_error(ctx, PT_ERR_INVALID, "vertex index out of range in submesh " + std::to_string(i));
        uint64_t total = (uint64_t)tri_off[i] + cnt / 3;
        if (total > 0x0FFFFFFFull) return set_error(ctx, PT_ERR_TOO_LARGE, "more than 2^28-1 triangles");
        tri_off[i + 1] = (uint32_t)total;
    }
    for (uint32_t m = 0; m < s->num_materials; m++) {
        for (int k = 10; k <= 12; k++) {
            int32_t idx;
            memcpy(&idx, &s->materials[(size_t)m * 22 + k], 4);
            if (idx != -1 && (idx < 0 || (uint32_t)idx >= s->num_textures))
                return set_error(ctx, PT_ERR_INVALID, "material " + std::to_string(m) + " names a texture that was not supplied");
        }
    }
    for (uint32_t t = 0; t < s->num_textures; t++)
        if (!s->textures[t].rgba8 || s->textures[t].width == 0 || s->textures[t].height == 0 ||
            (s->textures[t].flags & ~31u) || (s->textures[t].flags & 3u) == 3u || ((s->textures[t].flags >> 2) & 3u) == 3u)
            return set_error(ctx, PT_ERR_INVALID, "texture " + std::to_string(t) + " is empty or has unknown flags");

    PT_SYNC(ctx);
    free_scene(ctx);
    int rc;
    if ((rc = upload(ctx, ctx->d_vertices, s->vertices, (size_t)s->num_vertices * 12))) return rc;
    if ((rc = upload(ctx, ctx->d_indices, s->indices, s->num_indices))) return rc;
    if ((rc = upload(ctx, ctx->d_submeshes, s->submeshes, (size_t)s->num_submeshes * 4))) return rc;
    if ((rc = upload(ctx, ctx->d_xforms, s->submesh_xforms, (size_t)s->num_submeshes * 12))) return rc;
    if ((rc = upload(ctx, ctx->d_materials, s->materials, (size_t)s->num_materials * 22))) return rc;
    if ((rc = upload(ctx, ctx->d_sub_tri_offset, tri_off.data(), tri_off.size()))) return rc;
    // textures: one pool of texels + a table (texel offset, width, height, flags)
    std::vector<uint32_t> table((size_t)s->num_textures * 4 + 4, 0);
    size_t texels = 0;
    for (uint32_t t = 0; t < s->num_textures; t++) {
        table[t * 4 + 0] = (uint32_t)texels;
        table[t * 4 + 1] = s->textures[t].width;
        table[t * 4 + 2] = s->textures[t].height;
        table[t * 4 + 3] = s->textures[t].flags;
        texels += (size_t)s->textures[t].width * s->textures[t].height;
        if (texels > 0xFFFFFFFFull) return set_error(ctx, PT_ERR_TOO_LARGE, "more than 2^32 texels");
    }
    PT_HIP(ctx, hipMalloc((void**)&ctx->d_tex_pool, (texels ? texels : 1) * 4));
    for (uint32_t t = 0; t < s->num_textures; t++)
        PT_HIP(ctx, hipMemcpyAsync(ctx->d_tex_pool + (size_t)table[t * 4] * 4, s->textures[t].rgba8,
                                   (size_t)s->textures[t].width * s->textures[t].height * 4, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = upload(ctx, ctx->d_tex_table, table.data(), table.size()))) return rc;
    if (s->env_rgba && s->env_face_size) {
        ctx->env_n = s->env_face_size;
        if ((rc = upload(ctx, ctx->d_env, s->env_rgba, (size_t)6 * ctx->env_n * ctx->env_n * 4))) return rc;
    } else {
        // the reference's HDR sky is not in its repository: constant white radiance, one texel per face
        float white[24];
        for (float& w : white) w = 1.0f;
        ctx->env_n = 1;
        if ((rc = upload(ctx, ctx->d_env, (const float*)white, 24))) return rc;
    }
    PT_SYNC(ctx);   // the host arrays may go away after return
    ctx->num_vertices = s->num_vertices; ctx->num_indices = s->num_indices; ctx->num_submeshes = s->num_submeshes;
    ctx->num_materials = s->num_materials; ctx->num_textures = s->num_textures;
    ctx->num_tris = tri_off[s->num_submeshes];
    ctx->scene_loaded = true;
    return PT_OK;
}

int pt_build_accel(pt_ctx* ctx, uint32_t flags, pt_accel_stats* stats)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->scene_loaded) return set_error(ctx, PT_ERR_INVALID, "pt_build_accel: no scene uploaded");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = build_accel(ctx, flags);
    if (stats) *stats = ctx->accel_stats;
    return rc;
}

int pt_update_materials(pt_ctx* ctx, const float* materials, uint32_t num_materials)
{
    if (!ctx || !materials) return PT_ERR_INVALID;
    if (!ctx->scene_loaded || num_materials != ctx->num_materials)
        return set_error(ctx, PT_ERR_INVALID, "pt_update_materials: material count differs from the uploaded scene");
    for (uint32_t m = 0; m < num_materials; m++)
        for (int k = 10; k <= 12; k++) {
            int32_t idx;
            memcpy(&idx, &materials[(size_t)m * 22 + k], 4);
            if (idx != -1 && (idx < 0 || (uint32_t)idx >= ctx->num_textures))
                return set_error(ctx, PT_ERR_INVALID, "material names a texture that was not supplied");
        }
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_SYNC(ctx);
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_materials, materials, (size_t)num_materials * 88, hipMemcpyHostToDevice, ctx->stream));
    PT_SYNC(ctx);
    return PT_OK;
}

int pt_update_transforms(pt_ctx* ctx, const float* xforms, uint32_t num_submeshes, uint32_t build_flags, pt_accel_stats* stats)
{
    if (!ctx || !xforms) return PT_ERR_INVALID;
    if (!ctx->scene_loaded || num_submeshes != ctx->num_submeshes)
        return set_error(ctx, PT_ERR_INVALID, "pt_update_transforms: submesh count differs from the uploaded scene");
    for (size_t k = 0; k < (size_t)num_submeshes * 12; k++)
        if (!(xforms[k] == xforms[k]) || xforms[k] > 3e38f || xforms[k] < -3e38f) return set_error(ctx, PT_ERR_INVALID, "pt_update_transforms: matrix entry is not finite");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_SYNC(ctx);
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_xforms, xforms, (size_t)num_submeshes * 48, hipMemcpyHostToDevice, ctx->stream));
    PT_SYNC(ctx);
    int rc = (build_flags & PT_UPDATE_REFIT) ? refit_accel(ctx) : build_accel(ctx, build_flags);
    if (stats) *stats = ctx->accel_stats;
    return rc;
}

int pt_resize(pt_ctx* ctx, uint32_t width, uint32_t height)
{
    if (!ctx) return PT_ERR_INVALID;
    if (width == 0 || height == 0) return set_error(ctx, PT_ERR_INVALID, "pt_resize: empty image");
    if ((uint64_t)width * height > (1ull << 28) / PT_MAX_BATCH) return set_error(ctx, PT_ERR_TOO_LARGE, "pt_resize: more than 2^25 pixels");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_SYNC(ctx);
    free_images(ctx);
    size_t n = (size_t)width * height;
    size_t tiles = (size_t)((width + 7) / 8) * ((height + 7) / 8);
    ctx->shard_cap = (uint32_t)((tiles + PT_SHARDS - 1) / PT_SHARDS) * 64u * PT_MAX_BATCH;
    size_t slots = (size_t)ctx->shard_cap * PT_SHARDS;
    // everything a frame batch needs is reserved here: images (40 B per pixel) and, for PT_MAX_BATCH frames in flight, two ray queues,
    // two path-state queues, hit records and the per-path colours (about 1.4 KB per pixel: 2.9 GB at 1920x1080, 11.7 GB at 3840x2160)
    const size_t total_bytes = n * (16 + 16 + 8) + slots * (4 * 2 * 16 + 16 + 4) + n * PT_MAX_BATCH * 32;
    struct { void** p; size_t bytes; } wanted[] = {
        {(void**)&ctx->d_image_own, n * 16}, {(void**)&ctx->d_accum_own, n * 16}, {(void**)&ctx->d_primary, n * 8},
        {(void**)&ctx->d_rayO[0], slots * 16}, {(void**)&ctx->d_rayD[0], slots * 16}, {(void**)&ctx->d_stA[0], slots * 16}, {(void**)&ctx->d_stB[0], slots * 16},
        {(void**)&ctx->d_rayO[1], slots * 16}, {(void**)&ctx->d_rayD[1], slots * 16}, {(void**)&ctx->d_stA[1], slots * 16}, {(void**)&ctx->d_stB[1], slots * 16},
        {(void**)&ctx->d_hits, slots * 16}, {(void**)&ctx->d_hit_det, slots * 4},
        {(void**)&ctx->d_stC, n * PT_MAX_BATCH * 16}, {(void**)&ctx->d_partial, n * PT_MAX_BATCH * 16}};
    for (auto& w : wanted) {
        hipError_t e = hipMalloc(w.p, w.bytes);
        if (e != hipSuccess) {
            free_images(ctx);
            (void)hipGetLastError();
            return set_error(ctx, PT_ERR_HIP, "pt_resize: " + std::to_string(width) + "x" + std::to_string(height) + " needs " + std::to_string(total_bytes >> 20) +
                             " MiB of device memory for images, ray queues and path state (about 1.4 KB per pixel, reserved up front for batches of " +
                             std::to_string(PT_MAX_BATCH) + " frames); hipMalloc of " + std::to_string(w.bytes >> 20) + " MiB failed: " + hipGetErrorString(e));
        }
    }
    PT_HIP(ctx, hipMemsetAsync(ctx->d_image_own, 0, n * 16, ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->d_accum_own, 0, n * 16, ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->d_primary, 0xFF, n * 8, ctx->stream));
    ctx->main_dirty = true;
    ctx->d_image = ctx->d_image_own;
    ctx->d_accum = ctx->d_accum_own;
    ctx->width = width;
    ctx->height = height;
    return PT_OK;
}

int pt_bind_images(pt_ctx* ctx, void* device_image, void* device_accum)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->width) return set_error(ctx, PT_ERR_INVALID, "pt_bind_images: call pt_resize first");
    PT_SYNC(ctx);
    ctx->d_image = device_image ? (float*)device_image : ctx->d_image_own;
    ctx->d_accum = device_accum ? (float*)device_accum : ctx->d_accum_own;
    return PT_OK;
}

int pt_set_camera(pt_ctx* ctx, const float* m)
{
    if (!ctx || !m) return PT_ERR_INVALID;
    memcpy(ctx->camera, m, sizeof(float) * 80);
    ctx->camera_set = true;
    return PT_OK;
}

int pt_set_params(pt_ctx* ctx, const pt_params* p)
{
    if (!ctx || !p) return PT_ERR_INVALID;
    if (p->spp_per_frame == 0 || p->max_bounces == 0) return set_error(ctx, PT_ERR_INVALID, "spp_per_frame and max_bounces must be >= 1");
    // the traversal orders subtrees by the bit pattern of their entry distance, which is monotonic for non-negative floats only
    if (!(p->tmin >= 0.0f) || !(p->tmax == p->tmax)) return set_error(ctx, PT_ERR_INVALID, "tmin must be >= 0 and tmax must not be NaN");
    if (p->spp_per_frame > 0xFFFFu || p->max_bounces > 0xFFFFu || (uint64_t)p->spp_per_frame * p->max_bounces > 65536ull)
        return set_error(ctx, PT_ERR_TOO_LARGE, "spp_per_frame * max_bounces must not exceed 65536");
    if (!ctx->pending.empty()) {      // frames already handed in keep the parameters they were traced with
        int rc = launch_trace(ctx);
        if (rc) return rc;
    }
    // the chains' shares of the queue counters are laid out by the number of rounds: frames still running keep theirs
    if ((uint64_t)p->spp_per_frame * p->max_bounces != (uint64_t)ctx->params.spp_per_frame * ctx->params.max_bounces) PT_SYNC(ctx);
    ctx->params = *p;
    return PT_OK;
}

int pt_set_precision(pt_ctx* ctx, int mode)
{
    if (!ctx) return PT_ERR_INVALID;
    if (mode != PT_PRECISION_EXACT && mode != PT_PRECISION_FAST) return set_error(ctx, PT_ERR_INVALID, "pt_set_precision: mode must be PT_PRECISION_EXACT or PT_PRECISION_FAST");
    if (!ctx->pending.empty()) {      // frames already handed in keep the arithmetic they were traced with
        int rc = launch_trace(ctx);
        if (rc) return rc;
    }
    ctx->precision = mode;
    return PT_OK;
}

int pt_set_chains(pt_ctx* ctx, uint32_t chains)
{
    if (!ctx) return PT_ERR_INVALID;
    if (chains < 1 || chains > PT_MAX_CHAINS || PT_SHARDS % chains)
        return set_error(ctx, PT_ERR_INVALID, "pt_set_chains: chains must be 1, 2, 4, 8 or 16");
    PT_SYNC(ctx);
    {
        int rc = finish_timing(ctx);
        if (rc) return rc;
    }
    ctx->chains = chains;
    ctx->main_dirty = true;
    // not an error, but worth telling: chains that share a hardware queue run one after the other
    ctx->warning = chains > ctx->hw_queues ? "warning: " + std::to_string(chains) + " chains but the process has " + std::to_string(ctx->hw_queues) +
                                               " hardware queues (" + ctx->hw_queues_source + "); set GPU_MAX_HW_QUEUES >= chains before the HIP runtime starts" : std::string();
    return PT_OK;
}

int pt_set_batch(pt_ctx* ctx, uint32_t frames)
{
    if (!ctx) return PT_ERR_INVALID;
    if (frames < 1 || frames > PT_MAX_BATCH) return set_error(ctx, PT_ERR_INVALID, "pt_set_batch: frames must be 1..8");
    if (!ctx->pending.empty()) {
        int rc = launch_trace(ctx);
        if (rc) return rc;
    }
    ctx->batch = frames;
    return PT_OK;
}

int pt_set_profiling(pt_ctx* ctx, int mode)
{
    if (!ctx) return PT_ERR_INVALID;
    if (mode < 0 || mode > 2) return set_error(ctx, PT_ERR_INVALID, "pt_set_profiling: mode must be 0, 1 or 2");
    int rc = finish_timing(ctx);
    if (rc) return rc;
    ctx->profiling = mode;
    return PT_OK;
}

int pt_trace(pt_ctx* ctx, uint32_t frame_index)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->scene_loaded || !ctx->accel_built) return set_error(ctx, PT_ERR_INVALID, "pt_trace: scene or hierarchy missing");
    if (!ctx->width) return set_error(ctx, PT_ERR_INVALID, "pt_trace: call pt_resize first");
    if (!ctx->camera_set) return set_error(ctx, PT_ERR_INVALID, "pt_trace: call pt_set_camera first");
    if (frame_index == 0) return set_error(ctx, PT_ERR_INVALID, "pt_trace: frame_index is 1-based");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t rounds = ctx->params.spp_per_frame * ctx->params.max_bounces + ctx->defer_rounds;
    if (rounds + 1 > ctx->counts_cap) {
        PT_SYNC(ctx);
        (void)dev_free(ctx->d_counts);
        PT_HIP(ctx, hipMalloc((void**)&ctx->d_counts, pt_chain_words(rounds, PT_SHARDS) * sizeof(uint32_t)));
        ctx->counts_cap = rounds + 1;
    }
    FrameView v;
    v.frame = frame_index;
    memcpy(v.inv_view, ctx->camera + 48, 64);
    memcpy(v.inv_proj, ctx->camera + 64, 64);
    ctx->pending.push_back(v);
    if (ctx->pending.size() >= ctx->batch) return launch_trace(ctx);
    return PT_OK;
}

int pt_sync(pt_ctx* ctx)
{
    if (!ctx) return PT_ERR_INVALID;
    PT_SYNC(ctx);
    return PT_OK;
}

static int read_back(pt_ctx* ctx, void* host, const void* dev, size_t bytes)
{
    if (!host) return PT_ERR_INVALID;
    if (!ctx->width) return set_error(ctx, PT_ERR_INVALID, "no image: call pt_resize first");
    PT_SYNC(ctx);    // every chain, not only the main stream the copy runs on
    PT_HIP(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PT_SYNC(ctx);
    return PT_OK;
}

int pt_read_image(pt_ctx* ctx, float* rgba) { return ctx ? read_back(ctx, rgba, ctx->d_image, (size_t)ctx->width * ctx->height * 16) : PT_ERR_INVALID; }
int pt_read_accum(pt_ctx* ctx, float* rgba) { return ctx ? read_back(ctx, rgba, ctx->d_accum, (size_t)ctx->width * ctx->height * 16) : PT_ERR_INVALID; }
int pt_read_primary_ids(pt_ctx* ctx, uint32_t* ids) { return ctx ? read_back(ctx, ids, ctx->d_primary, (size_t)ctx->width * ctx->height * 8) : PT_ERR_INVALID; }

int pt_write_accum(pt_ctx* ctx, const float* rgba)
{
    if (!ctx || !rgba) return PT_ERR_INVALID;
    if (!ctx->width) return set_error(ctx, PT_ERR_INVALID, "no image: call pt_resize first");
    PT_SYNC(ctx);
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_accum, rgba, (size_t)ctx->width * ctx->height * 16, hipMemcpyHostToDevice, ctx->stream));
    PT_SYNC(ctx);
    return PT_OK;
}

int pt_resolve(pt_ctx* ctx)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->width) return set_error(ctx, PT_ERR_INVALID, "no image: call pt_resize first");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    return launch_resolve(ctx);
}

int pt_postprocess(pt_ctx* ctx, float exposure)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->width) return set_error(ctx, PT_ERR_INVALID, "no image: call pt_resize first");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    return launch_postprocess(ctx, exposure);
}

int pt_read_postprocessed(pt_ctx* ctx, uint8_t* rgba8)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->d_post) return set_error(ctx, PT_ERR_INVALID, "pt_read_postprocessed: call pt_postprocess first");
    return read_back(ctx, rgba8, ctx->d_post, (size_t)ctx->width * ctx->height * 4);
}

void* pt_device_postprocessed(pt_ctx* ctx) { return ctx ? (void*)ctx->d_post : nullptr; }

static int check_env_args(pt_ctx* ctx, uint32_t face_size)
{
    if (!ctx->scene_loaded) return set_error(ctx, PT_ERR_INVALID, "environment: upload the scene first (pt_scene_upload resets the environment)");
    if (face_size == 0 || face_size > 8192) return set_error(ctx, PT_ERR_INVALID, "environment: face size must be 1..8192");
    return PT_OK;
}

int pt_env_from_equirect(pt_ctx* ctx, const float* rgba, uint32_t width, uint32_t height, uint32_t face_size)
{
    if (!ctx || !rgba) return PT_ERR_INVALID;
    if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 28)) return set_error(ctx, PT_ERR_INVALID, "pt_env_from_equirect: bad image size");
    int rc = check_env_args(ctx, face_size);
    if (rc) return rc;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    return env_from_equirect(ctx, rgba, width, height, face_size);
}

int pt_env_preetham(pt_ctx* ctx, float turbidity, float azimuth, float inclination, uint32_t face_size)
{
    if (!ctx) return PT_ERR_INVALID;
    int rc = check_env_args(ctx, face_size);
    if (rc) return rc;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    return env_preetham(ctx, turbidity, azimuth, inclination, face_size);
}

int pt_env_upload(pt_ctx* ctx, const float* rgba, uint32_t face_size)
{
    if (!ctx || !rgba) return PT_ERR_INVALID;
    int rc = check_env_args(ctx, face_size);
    if (rc) return rc;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    return env_upload(ctx, rgba, face_size);
}

int pt_read_env(pt_ctx* ctx, float* rgba, uint32_t* face_size)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->d_env) return set_error(ctx, PT_ERR_INVALID, "pt_read_env: no environment (upload a scene first)");
    if (face_size) *face_size = ctx->env_n;
    if (!rgba) return PT_OK;
    PT_SYNC(ctx);
    PT_HIP(ctx, hipMemcpyAsync(rgba, ctx->d_env, (size_t)6 * ctx->env_n * ctx->env_n * 16, hipMemcpyDeviceToHost, ctx->stream));
    PT_SYNC(ctx);
    return PT_OK;
}

int pt_ray_count(pt_ctx* ctx, uint64_t* rays)
{
    if (!ctx || !rays) return PT_ERR_INVALID;
    unsigned long long v = 0;
    PT_SYNC(ctx);
    PT_HIP(ctx, hipMemcpyAsync(&v, ctx->d_total_rays, 8, hipMemcpyDeviceToHost, ctx->stream));
    PT_SYNC(ctx);
    *rays = v;
    return PT_OK;
}

int pt_last_trace_timing(pt_ctx* ctx, pt_trace_timing* out)
{
    if (!ctx || !out) return PT_ERR_INVALID;
    int rc = finish_timing(ctx);
    if (rc) return rc;
    *out = ctx->timing;
    return PT_OK;
}

void* pt_stream(pt_ctx* ctx)
{
    if (!ctx) return nullptr;
    // the caller is about to order its own work after ours on this stream: bring the other chains in,
    // and order the next frame after whatever the caller enqueues
    if (join_chains(ctx) != PT_OK) return nullptr;
    ctx->main_dirty = true;
    return (void*)ctx->stream;
}
int pt_wait_stream(pt_ctx* ctx, void* stream)
{
    if (!ctx) return PT_ERR_INVALID;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = join_chains(ctx);          // queued frames are launched first: they were recorded before this call
    if (rc) return rc;
    if (!ctx->wait_event) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->wait_event, hipEventDisableTiming));
    PT_HIP(ctx, hipEventRecord(ctx->wait_event, (hipStream_t)stream));
    PT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->wait_event, 0));
    ctx->main_dirty = true;             // the other chains are ordered after the main stream at the next launch
    return PT_OK;
}
// ncclReduce of the RCCL the caller's communicator belongs to.  RCCL is not linked: a single-GPU user needs none.  The
// symbol is looked up in this order: PT_RCCL_LIB (a path), the process's global symbols (an application linked against
// RCCL), an RCCL that is already loaded under its usual names (a Python process whose torch brought its own), and last a
// fresh load of librccl.so from the loader's search path.
typedef int (*nccl_reduce_fn)(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, int root, void* comm, hipStream_t stream);
typedef const char* (*nccl_error_fn)(int);
static nccl_reduce_fn g_nccl_reduce = nullptr;
static nccl_error_fn g_nccl_error = nullptr;
static std::string g_nccl_source;

static bool resolve_rccl(std::string& why)
{
    if (g_nccl_reduce) return true;
    void* h = nullptr;
    if (const char* path = getenv("PT_RCCL_LIB")) {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!h) { why = std::string("PT_RCCL_LIB: ") + dlerror(); return false; }
        g_nccl_source = path;
    }
    if (!h) {
        if (void* f = dlsym(RTLD_DEFAULT, "ncclReduce")) {
            g_nccl_reduce = (nccl_reduce_fn)f;
            g_nccl_error = (nccl_error_fn)dlsym(RTLD_DEFAULT, "ncclGetErrorString");
            g_nccl_source = "global symbols of the process";
            return true;
        }
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)
            if (!h && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) g_nccl_source = std::string(n) + " (already loaded)";
        for (const char* n : names)
            if (!h && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) g_nccl_source = n;
    }
    if (!h) { why = "RCCL not found: link the application against librccl, load it before this call, or set PT_RCCL_LIB"; return false; }
    g_nccl_reduce = (nccl_reduce_fn)dlsym(h, "ncclReduce");
    g_nccl_error = (nccl_error_fn)dlsym(h, "ncclGetErrorString");
    if (!g_nccl_reduce) { why = g_nccl_source + " has no ncclReduce"; return false; }
    return true;
}

int pt_reduce_accum(pt_ctx* ctx, void* nccl_comm, int root)
{
    if (!ctx || !nccl_comm) return PT_ERR_INVALID;
    if (!ctx->width) return set_error(ctx, PT_ERR_INVALID, "pt_reduce_accum: call pt_resize first");
    if (root < 0) return set_error(ctx, PT_ERR_INVALID, "pt_reduce_accum: root must be a rank of the communicator");
    std::string why;
    if (!resolve_rccl(why)) return set_error(ctx, PT_ERR_INVALID, "pt_reduce_accum: " + why);
    PT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = join_chains(ctx);          // queued frames are launched and every chain's work is ordered before the reduce
    if (rc) return rc;
    ctx->main_dirty = true;             // later frames of the other chains run after the reduce
    const size_t count = (size_t)ctx->width * ctx->height * 4;
    const int NCCL_FLOAT32 = 7, NCCL_SUM = 0;
    int nrc = g_nccl_reduce(ctx->d_accum, ctx->d_accum, count, NCCL_FLOAT32, NCCL_SUM, root, nccl_comm, ctx->stream);
    if (nrc != 0)
        return set_error(ctx, PT_ERR_HIP, std::string("pt_reduce_accum: ncclReduce failed: ") + (g_nccl_error ? g_nccl_error(nrc) : std::to_string(nrc).c_str()) +
                         " (RCCL from " + g_nccl_source + ")");
    return PT_OK;
}

void* pt_device_image(pt_ctx* ctx) { return ctx ? (void*)ctx->d_image : nullptr; }
void* pt_device_accum(pt_ctx* ctx) { return ctx ? (void*)ctx->d_accum : nullptr; }

int pt_debug_intersect(pt_ctx* ctx, const float* rays, uint32_t n, uint32_t* hits, uint64_t* counters)
{
    if (!ctx || (n && (!rays || !hits))) return PT_ERR_INVALID;
    if (!ctx->accel_built) return set_error(ctx, PT_ERR_INVALID, "pt_debug_intersect: hierarchy missing");
    for (uint32_t i = 0; i < n; i++)
        if (!(rays[(size_t)i * 8 + 3] >= 0.0f)) return set_error(ctx, PT_ERR_INVALID, "pt_debug_intersect: ray " + std::to_string(i) + " has tmin < 0 or NaN");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    return debug_intersect(ctx, rays, n, hits, counters);
}

int pt_intersect_noalloc(pt_ctx* ctx, const float* device_rays, uint32_t n, uint32_t* device_hits)
{
    if (!ctx || (n && (!device_rays || !device_hits))) return PT_ERR_INVALID;
    if (((uintptr_t)device_rays & 15u) || ((uintptr_t)device_hits & 3u)) return set_error(ctx, PT_ERR_INVALID, "pt_intersect_noalloc: rays must be 16-byte aligned, hits 4-byte aligned");
    if (!ctx->accel_built) return set_error(ctx, PT_ERR_INVALID, "pt_intersect_noalloc: hierarchy missing");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    return intersect_noalloc(ctx, device_rays, n, device_hits);
}

int pt_debug_math(pt_ctx* ctx, int fn, const float* x, const float* y, float* out, uint32_t n)
{
    if (!ctx || (n && (!x || !out))) return PT_ERR_INVALID;
    if (fn < 0 || fn > 8 || ((fn == 4 || fn == 5) && !y)) return set_error(ctx, PT_ERR_INVALID, "pt_debug_math: bad function id");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    return debug_math(ctx, fn, x, y, out, n);
}

int pt_debug_disney_sample(pt_ctx* ctx, uint32_t n, const float* surf, const float* V, const float* N,
                           uint32_t* seeds, float* f, float* L, float* pdf)
{
    if (!ctx || (n && (!surf || !V || !N || !seeds || !f || !L || !pdf))) return PT_ERR_INVALID;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    return debug_disney(ctx, true, n, surf, V, N, nullptr, seeds, f, L, pdf);
}

int pt_debug_disney_eval(pt_ctx* ctx, uint32_t n, const float* surf, const float* V, const float* N,
                         const float* L, float* f, float* pdf)
{
    if (!ctx || (n && (!surf || !V || !N || !L || !f || !pdf))) return PT_ERR_INVALID;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    return debug_disney(ctx, false, n, surf, V, N, L, nullptr, f, nullptr, pdf);
}

int pt_debug_read_bvh(pt_ctx* ctx, uint32_t* nodes, uint32_t* tris)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->accel_built) return set_error(ctx, PT_ERR_INVALID, "pt_debug_read_bvh: hierarchy missing");
    if (nodes && ctx->num_nodes) PT_HIP(ctx, hipMemcpyAsync(nodes, ctx->d_nodes, (size_t)ctx->num_nodes * 128, hipMemcpyDeviceToHost, ctx->stream));
    if (tris && ctx->num_refs) PT_HIP(ctx, hipMemcpyAsync(tris, ctx->d_tris, (size_t)ctx->num_refs * 48, hipMemcpyDeviceToHost, ctx->stream));
    PT_SYNC(ctx);
    return PT_OK;
}

} // extern "C"
