// Wavefront path tracing kernels: the replacement for the reference's ray-tracing pipeline
// (RayGen / ClosestHit / Miss shaders run by vkCmdTraceRaysKHR, PathTracer/src/RayTracingLayer.cpp:274-281).
//
//   k_raygen   RayGen.glsl:312-345   per-pixel seed, first camera ray, path state
//   k_extend   RayGen.glsl:124       closest hit of every queued ray (pt_traverse.h)
//   k_shade    Miss.glsl:8-29, ClosestHit.glsl:144-226, RayGen.glsl:126-173, Disney.glsl:150-359
//              sky on a miss, surface fetch, emission, BSDF sample, throughput, next ray,
//              Russian roulette; finished paths either start the pixel's next sample or
//              store the frame's colour of the pixel; survivors are compacted into the
//              next queue with one ballot + one atomic per wave
//   k_accumulate  RayGen.glsl:347-372  adds the finished frames to the accumulation image in frame
//              order (frame 1 clears it), writes the running mean, NaN -> red
//              and adds the batch's queue sizes to the ray counter
//
// Up to PT_MAX_BATCH consecutive frames are traced by the same launches: a path is a (frame slot,
// pixel) pair, with the slot's own frame index and camera (k_set_views).  A (slot, pixel) owns one
// path at a time (the reference's samples of a pixel are sequential in the RNG state,
// RayGen.glsl:314-345), numbered slot * pixels + pixel.  Rays, hits and the path's running state
// (throughput, seed, radiance, bounce and sample counters) are indexed by queue slot, so that every
// round reads and writes them densely; only the colour of a pixel's finished samples and the
// frame's finished colour are kept per path.  A frame is also cut
// into chains: independent pipelines of these kernels over disjoint pixel tiles and queue shards,
// each on a stream of its own (launch_trace).
//
// The ray queue is cut into PT_SHARDS independent sub-queues.  8x8-pixel tile t starts in shard
// t % PT_SHARDS, a workgroup with blockIdx % PT_SHARDS == s only ever reads and appends to shard s,
// and survivors stay in their shard.  A shard therefore never outgrows its initial capacity, the
// append counters are PT_SHARDS different words instead of one (a single word serialises at about
// 90 atomics per microsecond, which bounded the whole shade stage), and because workgroups b and
// b + 8 share an XCD, a shard's rays, hits and path state are touched by one XCD's L2 only.
#include "pt_math.h"
#include "pt_context.h"
#include "pt_traverse.h"
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace ptd {

static constexpr int EXT_BLOCK = 64;     // one wave per workgroup: its LDS stack is PT_STACK_LDS * 64 * 8 bytes
// Workgroups of one wave: the shading launch shares the chip with the traversal launches of the other chains, whose waves
// hold most wave slots; a single wave is placed as soon as one slot frees up, a workgroup of four waits for room for all
// four on one CU (256 threads: 6 % fewer rays per second on the mid-sized scenes, 3.5 % on the large one)
#ifndef PT_SHADE_BLOCK
#define PT_SHADE_BLOCK 64
#endif
static constexpr int SHADE_BLOCK = PT_SHADE_BLOCK;
static constexpr uint32_t MISS = 0xFFFFFFFFu;

struct Queues {
    float4* rayO_in; float4* rayD_in;
    float4* rayO_out; float4* rayD_out;
    float4* hits;           // per queue slot (t, V, W, triangle); barycentric u = V / hit_det, v = W / hit_det
    const float* hit_det;
    float4* stA_in; float4* stB_in;     // path state of the queued rays, by queue slot like the rays
    float4* stA_out; float4* stB_out;
    float4* stC;                        // per path
    uint32_t* counts;       // of this chain: counts[r * nshards + ls] = rays queued in its shard ls for round r
    uint32_t shard0, nshards;   // the chain owns shards [shard0, shard0 + nshards)
    uint32_t shard_cap;     // slots per shard; shard s owns slots [s * shard_cap, (s + 1) * shard_cap)
    uint2* primary;
    float* image; float* accum;
    float4* partial;            // finished colour of every (frame slot, pixel) of the batch, added up in frame order by k_accumulate
    const FrameView* views;     // the batch's frame indices and cameras
};

PT_DEV v3 mat_mul_dir(const float* m, float x, float y, float z, float w, float& ow)
{
    v3 r;
    r.x = m[0] * x + m[4] * y + m[8] * z + m[12] * w;
    r.y = m[1] * x + m[5] * y + m[9] * z + m[13] * w;
    r.z = m[2] * x + m[6] * y + m[10] * z + m[14] * w;
    ow = m[3] * x + m[7] * y + m[11] * z + m[15] * w;
    return r;
}

// RayGen.glsl:322-342
PT_DEV void camera_ray(const TraceParams& P, const FrameView& V, uint32_t x, uint32_t y, uint32_t sample, uint32_t& seed, v3& o, v3& d)
{
    float pcx = (float)x + 0.5f, pcy = (float)y + 0.5f;
    if (sample > 0) {
        float angle = random_value(seed) * 2.0f * PT_PI;
        float s, c;
        det_sincos(angle, s, c);
        float r = fsqrt(random_value(seed));
        pcx += c * r;
        pcy += s * r;
    }
    float u = fdiv(pcx, (float)P.width), v = fdiv(pcy, (float)P.height);
    float dx = u * 2.0f - 1.0f, dy = v * 2.0f - 1.0f;
    float tw;
    v3 target = mat_mul_dir(V.inv_proj, dx, dy, 1.0f, 1.0f, tw);
    v3 tn = normalize(target / tw);
    float dw;
    v3 dir = mat_mul_dir(V.inv_view, tn.x, tn.y, tn.z, 0.0f, dw);
    o = V3(V.inv_view[12], V.inv_view[13], V.inv_view[14]);
    d = normalize(dir);
}

// Camera rays are always generated with the bit-exact arithmetic, whichever arithmetic shades (primary hits then do not
// depend on the precision mode); plain floats, because each arithmetic namespace has a vector type of its own
PT_DEV void camera_ray_exact(const TraceParams& P, const FrameView& V, uint32_t x, uint32_t y, uint32_t sample, uint32_t& seed, float* od)
{
    v3 o, d;
    camera_ray(P, V, x, y, sample, seed, o, d);
    od[0] = o.x; od[1] = o.y; od[2] = o.z; od[3] = d.x; od[4] = d.y; od[5] = d.z;
}

// slot -> pixel: 8x8 pixel tiles, one tile per wave, so a wave's primary rays are a compact bundle
PT_DEV bool slot_to_pixel(uint32_t slot, uint32_t W, uint32_t H, uint32_t& x, uint32_t& y)
{
    uint32_t tiles_x = (W + 7u) >> 3;
    uint32_t tile = slot >> 6, l = slot & 63u;
    uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    x = tx * 8u + (l & 7u);
    y = ty * 8u + (l >> 3);
    return x < W && y < H;
}

// appends the lanes with `alive` to the queue counted by *counter; returns the slot
PT_DEV uint32_t wave_append(bool alive, uint32_t* counter)
{
    unsigned long long mask = __ballot(alive);
    if (mask == 0) return 0;
    uint32_t lane = threadIdx.x & 63u;
    int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// start of a chain's batch, one wave: the batch's cameras, and the chain's queue counters back to zero (a launch of its own for
// either costs the chain a trip through the command processor)
__global__ void k_set_views(FrameViewBatch b, FrameView* dst, uint32_t* counts, uint32_t words)
{
    if (threadIdx.x < PT_MAX_BATCH) dst[threadIdx.x] = b.v[threadIdx.x];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) counts[i] = 0u;
}

// number of 8x8 tiles of the chain: tile t belongs to shard t % PT_SHARDS
PT_DEV uint32_t chain_tiles(const TraceParams& P, const Queues& q)
{
    uint32_t tiles = ((P.width + 7u) >> 3) * ((P.height + 7u) >> 3);
    uint32_t full = tiles / PT_SHARDS, rest = tiles % PT_SHARDS;
    uint32_t extra = rest > q.shard0 ? (rest - q.shard0 < q.nshards ? rest - q.shard0 : q.nshards) : 0u;
    return full * q.nshards + extra;
}

__global__ void __launch_bounds__(256) k_raygen(TraceParams P, Queues q)
{
    const uint32_t ctiles = chain_tiles(P, q), npix = P.width * P.height;
    const uint32_t nslots = ctiles * P.nframes * 64u;          // one tile of one frame slot per wave
    for (uint32_t base = blockIdx.x * blockDim.x; base < nslots; base += gridDim.x * blockDim.x) {
        uint32_t cs = base + threadIdx.x;
        uint32_t jw = cs >> 6;                         // wave-uniform
        uint32_t k = jw / ctiles, j = jw % ctiles;     // frame slot, the chain's j-th tile
        uint32_t ls = j % q.nshards;
        uint32_t tile = (j / q.nshards) * PT_SHARDS + q.shard0 + ls;
        uint32_t slot = tile * 64u + (cs & 63u);
        uint32_t x = 0, y = 0;
        bool valid = cs < nslots && slot_to_pixel(slot, P.width, P.height, x, y);
        v3 o = V3(0.0f), d = V3(0.0f);
        uint32_t path = 0, seed0 = 0;
        if (valid) {
            uint32_t pixel = x + y * P.width;
            path = k * npix + pixel;
            const FrameView& V = q.views[k];
            uint32_t seed = pixel * V.frame;           // RayGen.glsl:314-315
            camera_ray(P, V, x, y, 0, seed, o, d);
            seed0 = seed;
        }
        uint32_t out = (q.shard0 + ls) * q.shard_cap + wave_append(valid, &q.counts[ls]);
        if (valid) {
            q.stA_out[out] = make_float4(1.0f, 1.0f, 1.0f, u2f(seed0));
            q.stB_out[out] = make_float4(0.0f, 0.0f, 0.0f, u2f(0u));
            q.rayO_out[out] = make_float4(o.x, o.y, o.z, u2f(path));
            q.rayD_out[out] = make_float4(d.x, d.y, d.z, 0.0f);
        }
    }
}

// hit record of a ray that was suspended and handed on to the next round
static constexpr uint32_t DEFERRED = 0xFFFFFFFEu;
// rayD.w (as bits): rounds the path is behind schedule in the low byte, RAY_RESUMES if the ray continues a suspended traversal
static constexpr uint32_t RAY_LAG_MASK = 0xFFu, RAY_RESUMES = 0x100u;

// 4 waves per SIMD: 128 VGPRs, no scratch; the LDS stack would admit 4.5, and 5 (96 VGPRs) measured no faster on a
// mid-sized scene and slower on a large one, whose traversal competes for cache rather than for issue slots
#ifndef PT_EXT_WAVES_PER_SIMD
#define PT_EXT_WAVES_PER_SIMD 4
#endif
#define PT_EXT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(PT_EXT_WAVES_PER_SIMD, PT_EXT_WAVES_PER_SIMD)))

struct ExtendArgs {
    float tmin, tmax;
    const uint32_t* counts;     // queue sizes of this round, one per shard of the chain
    uint32_t* cursors;          // fetch cursors of this round
    uint32_t* next_counts;      // queue sizes of the next round (suspended rays go to its front); nullptr in launches that do not suspend
    uint32_t* deferred;         // running count of suspended rays of the chain's frame
    uint32_t shard0, nshards, shard_cap;
    uint32_t max_lag;           // a path may fall this many rounds behind
    uint32_t susp_cap;          // suspended-traversal records per shard
    uint32_t susp_words;        // words per record: PT_SUSPEND_STATE_WORDS + 2 * stack_entries
    uint32_t stack_entries;     // entries of a lane's column in the spill area = the deepest stack the hierarchy can cause
    const uint32_t* susp_in;    // records of the rays suspended by the previous round, [shard of the chain][slot][susp_words]
    uint32_t* susp_out;         // records for the next round
    const float4* rayO; const float4* rayD;
    float4* next_rayO; float4* next_rayD;
    const float4* stA; const float4* stB;       // path state by queue slot: a suspended ray takes its state along
    float4* next_stA; float4* next_stB;
    float4* hits;
    float* hit_det;
    uint2* spill;
    uint32_t iter_cap;          // bound of a wave's traversal loop
    uint32_t* fault;            // host-visible word, set when a wave reached the bound
};

template <bool STATS, bool CUT>
__global__ void __launch_bounds__(EXT_BLOCK) PT_EXT_WAVES_ATTR k_extend(SceneView sc, ExtendArgs a, unsigned long long* stats, unsigned long long* wave_trace)
{
    __shared__ uint2 stack[PT_STACK_LDS * EXT_BLOCK];
    __shared__ __attribute__((aligned(16))) uint32_t scratch[PT_SCRATCH_WORDS];
    const uint32_t ls = blockIdx.x % a.nshards;
    const uint32_t n = a.counts[ls];
    const uint32_t base = (a.shard0 + ls) * a.shard_cap;
    uint32_t sn = 0, st = 0;
    WaveStats ws;
    auto fetch = [&](uint32_t i, v3& o, v3& d, float& tn, float& tx) {
        float4 ro = a.rayO[base + i], rd = a.rayD[base + i];
        o = V3(ro.x, ro.y, ro.z);
        d = V3(rd.x, rd.y, rd.z);
        tn = a.tmin;
        tx = a.tmax;
        uint32_t w = f2u(rd.w);
        return ((w & RAY_LAG_MASK) < a.max_lag ? PT_RAY_CUTTABLE : 0u) | ((w & RAY_RESUMES) ? PT_RAY_RESUME : 0u);
    };
    // the hit goes out undivided: (t, V, W, triangle) and the determinant; k_shade divides u = V / det, v = W / det
    auto emit = [&](uint32_t i, const Trav& s) {
        a.hits[base + i] = make_float4(s.best.t, s.bV, s.bW, u2f(s.best.tri));
        a.hit_det[base + i] = s.bdet;
    };
    auto suspend = [&](uint32_t i, const Trav& s, const uint2* lds_stack, int stride, const uint2* ovf) {
        // called by all lanes that suspend, inside one branch: one queue slot each at the front of the next round
        unsigned long long m = __ballot(true);
        uint32_t lane = threadIdx.x & 63u;
        int leader = __ffsll((long long)m) - 1;
        uint32_t slot0 = 0;
        if ((int)lane == leader) {
            slot0 = atomicAdd(&a.next_counts[ls], (uint32_t)__popcll(m));
            atomicAdd(a.deferred, (uint32_t)__popcll(m));
        }
        slot0 = __shfl(slot0, leader);
        uint32_t slot = slot0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));     // < susp_cap: a wave suspends at most PT_CUT_LANES rays
        float4 ro = a.rayO[base + i], rd = a.rayD[base + i];
        a.next_rayO[base + slot] = ro;
        a.next_rayD[base + slot] = make_float4(rd.x, rd.y, rd.z, u2f(((f2u(rd.w) & RAY_LAG_MASK) + 1u) | RAY_RESUMES));
        a.next_stA[base + slot] = a.stA[base + i];
        a.next_stB[base + slot] = a.stB[base + i];
        a.hits[base + i] = make_float4(0.0f, 0.0f, 0.0f, u2f(DEFERRED));
        trav_save(s, a.susp_out + ((size_t)ls * a.susp_cap + slot) * a.susp_words, lds_stack, stride, ovf);
    };
    auto resume_record = [&](uint32_t i) { return a.susp_in + ((size_t)ls * a.susp_cap + i) * a.susp_words; };
    trace_wave<STATS, CUT>(sc.nodes, sc.tris, sc.num_tris, n, &a.cursors[ls], stack + threadIdx.x, EXT_BLOCK,
                           a.spill + (size_t)blockIdx.x * (a.stack_entries * 64) + threadIdx.x, scratch, fetch, emit, suspend, resume_record, sn, st, a.iter_cap, a.fault, STATS ? &ws : nullptr);
    if (STATS) {
        atomicAdd(&stats[0], (unsigned long long)sn);
        atomicAdd(&stats[1], (unsigned long long)st);
        if (threadIdx.x == 0) {
            atomicAdd(&stats[3], (unsigned long long)ws.node_phases);
            atomicAdd(&stats[4], (unsigned long long)ws.node_lanes);
            atomicAdd(&stats[5], (unsigned long long)ws.leaf_phases);
            atomicAdd(&stats[6], (unsigned long long)ws.leaf_lanes);
            atomicAdd(&stats[7], (unsigned long long)ws.refills);
            atomicAdd(&stats[8], ws.t_end - ws.t_exhausted);
            atomicAdd(&stats[9], ws.t_end - ws.t_start);
            atomicMax(&stats[10], (unsigned long long)ws.longest_ray);
            for (int k = 0; k < 4; k++) atomicAdd(&stats[11 + k], (unsigned long long)ws.node_hist[k]);
            atomicAdd(&stats[15], (unsigned long long)ws.idle_lanes);
            if (wave_trace) {
                unsigned long long* w = wave_trace + (size_t)blockIdx.x * 4;
                w[0] = ws.t_start; w[1] = ws.t_exhausted; w[2] = ws.t_end;
                w[3] = (unsigned long long)ws.longest_ray | ((unsigned long long)ws.node_phases << 32);
            }
        }
    }
}

// sampler state of a texture (pt_texture_desc.flags): wrap mode of u in bits 0-1 and of v in bits 2-3
// (0 repeat, 1 clamp to edge, 2 mirrored repeat), bit 4 = nearest instead of bilinear
PT_DEV float wrap_coord(float u, uint32_t mode)
{
    if (mode == 0u) {
        float f = u - floorf(u);
        return f >= 0.0f ? f : 0.0f;                  // NaN -> 0
    }
    if (mode == 2u) {
        float h = u * 0.5f;
        float t = (h - floorf(h)) * 2.0f;             // u modulo 2
        float m = t <= 1.0f ? t : 2.0f - t;
        return m >= 0.0f ? m : 0.0f;
    }
    if (!(u >= -1.0f)) u = -1.0f;                     // clamp: the texel indices are clamped below, this keeps them in int range
    return u > 2.0f ? 2.0f : u;
}

PT_DEV int wrap_index(int i, int n, uint32_t mode)
{
    if (mode == 0u) {
        if (i < 0) i += n;
        if (i >= n) i -= n;
        return i;
    }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// RayGen.glsl:347-372 for every pixel of the chain and every frame of the batch, in frame order: the finished colours
// wait in q.partial, so that a pixel's frames are added in the order of the pt_trace calls whatever order they finished in
__global__ void __launch_bounds__(256) k_accumulate(TraceParams P, Queues q, uint32_t count_entries, const uint32_t* deferred, unsigned long long* total)
{
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        // the chain's rays of this batch = sum of all its queue sizes, less the rays queued twice (suspended ones)
        unsigned long long s = 0;
        for (uint32_t r = threadIdx.x; r < count_entries; r += 64) s += q.counts[r];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (threadIdx.x == 0) atomicAdd(total, s - *deferred);
    }
    const uint32_t ctiles = chain_tiles(P, q), npix = P.width * P.height;
    const uint32_t nslots = ctiles * 64u;
    for (uint32_t cs = blockIdx.x * blockDim.x + threadIdx.x; cs < nslots; cs += gridDim.x * blockDim.x) {
        uint32_t j = cs >> 6;
        uint32_t tile = (j / q.nshards) * PT_SHARDS + q.shard0 + j % q.nshards;
        uint32_t x = 0, y = 0;
        if (!slot_to_pixel(tile * 64u + (cs & 63u), P.width, P.height, x, y)) continue;
        uint32_t pixel = x + y * P.width;
        float4* accp = reinterpret_cast<float4*>(q.accum) + pixel;
        float4 acc = *accp;
        v3 shown = V3(0.0f);
        for (uint32_t k = 0; k < P.nframes; k++) {
            float4 c = q.partial[k * npix + pixel];
            v3 color = V3(c.x, c.y, c.z);
            float numPaths = (float)P.spp;
            if (q.views[k].frame > 1) {
                color = color + V3(acc.x, acc.y, acc.z);
                numPaths = acc.w + (float)P.spp;
                acc = make_float4(color.x, color.y, color.z, numPaths);
            } else {
                acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            shown = color / numPaths;
        }
        *accp = acc;
        float4* img = reinterpret_cast<float4*>(q.image) + pixel;
        if (shown.x != shown.x || shown.y != shown.y || shown.z != shown.z) *img = make_float4(1.0f, 0.0f, 0.0f, 1.0f);
        else *img = make_float4(shown.x, shown.y, shown.z, 1.0f);
    }
}

// texture(sampler2D, uv): level 0, RGBA8 UNORM; bilinear + repeat unless the texture's flags say otherwise
PT_DEV float4 sample_texture(const SceneView& sc, int tex, float u, float v)
{
    const uint32_t* tt = sc.tex_table + tex * 4;
    uint32_t off = tt[0], w = tt[1], h = tt[2], flags = tt[3];
    const uint32_t ws = flags & 3u, wt = (flags >> 2) & 3u;
    const bool nearest = (flags & 16u) != 0;
    const uint32_t* px = reinterpret_cast<const uint32_t*>(sc.tex_pool) + off;
    float uu = wrap_coord(u, ws), vv = wrap_coord(v, wt);
    float x = uu * (float)w, y = vv * (float)h;
    if (!nearest) { x = x - 0.5f; y = y - 0.5f; }
    float x0f = floorf(x), y0f = floorf(y);
    float fx = nearest ? 0.0f : x - x0f, fy = nearest ? 0.0f : y - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;
    int x1 = nearest ? x0 : x0 + 1, y1 = nearest ? y0 : y0 + 1;
    x0 = wrap_index(x0, (int)w, ws); x1 = wrap_index(x1, (int)w, ws);
    y0 = wrap_index(y0, (int)h, wt); y1 = wrap_index(y1, (int)h, wt);
    uint32_t p00 = px[(size_t)y0 * w + x0], p10 = px[(size_t)y0 * w + x1];
    uint32_t p01 = px[(size_t)y1 * w + x0], p11 = px[(size_t)y1 * w + x1];
    float out[4];
    for (int ch = 0; ch < 4; ch++) {
        const float k = 1.0f / 255.0f;
        float c00 = (float)((p00 >> (8 * ch)) & 255u) * k, c10 = (float)((p10 >> (8 * ch)) & 255u) * k;
        float c01 = (float)((p01 >> (8 * ch)) & 255u) * k, c11 = (float)((p11 >> (8 * ch)) & 255u) * k;
        out[ch] = mix(mix(c00, c10, fx), mix(c01, c11, fx), fy);
    }
    return make_float4(out[0], out[1], out[2], out[3]);
}

#include "pt_shade.inc"      // sample_env, fetch_surface, k_shade: bit-exact arithmetic
#include "pt_fast.h"         // the same text again as ptd::fast::k_shade with the hardware's approximations

__global__ void k_resolve(uint32_t n, const float* __restrict__ accum, float* __restrict__ image)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    float4 a = reinterpret_cast<const float4*>(accum)[g];
    v3 c = V3(a.x, a.y, a.z) / a.w;
    float4 out = (c.x != c.x || c.y != c.y || c.z != c.z) ? make_float4(1.0f, 0.0f, 0.0f, 1.0f) : make_float4(c.x, c.y, c.z, 1.0f);
    reinterpret_cast<float4*>(image)[g] = out;
}

// global spill area of the traversal stacks: one [stack_entries][64] block of entries per wave
static int ensure_spill(pt_ctx* ctx, uint32_t waves)
{
    size_t need = (size_t)waves * ctx->stack_entries * 64;
    if (need <= ctx->spill_entries) return PT_OK;
    PT_SYNC(ctx);
    if (ctx->d_spill) { (void)hipFree(ctx->d_spill); ctx->d_spill = nullptr; ctx->spill_entries = 0; }
    PT_HIP(ctx, hipMalloc((void**)&ctx->d_spill, need * sizeof(uint2)));
    ctx->spill_entries = need;
    return PT_OK;
}

SceneView scene_view(const pt_ctx* ctx)
{
    SceneView sv;
    sv.vertices = ctx->d_vertices;
    sv.indices = ctx->d_indices;
    sv.submeshes = ctx->d_submeshes;
    sv.xforms = ctx->d_xforms;
    sv.materials = ctx->d_materials;
    sv.tex_pool = ctx->d_tex_pool;
    sv.tex_table = ctx->d_tex_table;
    sv.env = ctx->d_env;
    sv.env_n = ctx->env_n;
    sv.num_tris = ctx->num_refs;
    sv.nodes = ctx->d_nodes;
    sv.tris = ctx->d_tris;
    sv.shade_tris = ctx->d_shade_tris;
    return sv;
}

// launches the frames queued by pt_trace as one batch
int launch_trace(pt_ctx* ctx)
{
    if (ctx->pending.empty()) return PT_OK;
    FrameViewBatch views{};
    const uint32_t nframes = (uint32_t)ctx->pending.size();
    for (uint32_t k = 0; k < nframes; k++) views.v[k] = ctx->pending[k];
    ctx->pending.clear();          // before anything below synchronises, which would flush again
    const pt_params& pp = ctx->params;
    uint32_t npix = ctx->width * ctx->height;
    // Suspension of the last rays of a traversal launch (pt_traverse.h) pays when a launch has little to overlap with: with one
    // or two frames per batch it is worth +2 ... +8 % on deep hierarchies; with the launches of four or more frames in
    // flight on every chain the tail of a launch hides behind the other chains' work anyway, and the suspension records, the
    // two extra rounds and the larger kernel variant cost 1 ... 7 %.  A shallow hierarchy (a ray takes two or three node
    // steps) loses either way.  PT_DEFER in the environment overrides the choice for every batch.
    static const uint32_t defer_max_batch = getenv("PT_DEFER_MAX_BATCH") ? (uint32_t)atoi(getenv("PT_DEFER_MAX_BATCH")) : 2u;
    const uint32_t defer = ctx->defer_forced ? ctx->defer_rounds
                                             : ((nframes <= defer_max_batch && ctx->accel_stats.sah_cost >= 20.0f) ? ctx->defer_rounds : 0u);
    // more rounds than the longest path needs: a path falls a round behind each time its ray is suspended.  The counters
    // are laid out for the most rounds any batch can take, so that batches with and without suspension that follow each other
    // on a chain use the same block
    const uint32_t rounds = pp.spp_per_frame * pp.max_bounces + defer;
    const uint32_t layout_rounds = pp.spp_per_frame * pp.max_bounces + ctx->defer_rounds;
    TraceParams P;
    P.width = ctx->width; P.height = ctx->height;
    P.spp = pp.spp_per_frame; P.max_bounces = pp.max_bounces; P.rr_depth = pp.rr_depth;
    P.sky_scale = pp.sky_scale; P.eps = pp.eps; P.tmin = pp.tmin; P.tmax = pp.tmax;
    P.nframes = nframes;
    SceneView sv = scene_view(ctx);

    const bool prof = ctx->profiling != 0;
    const bool counting = ctx->profiling == 2;
    const bool wave_trace = counting && getenv("PT_WAVE_TRACE");
    const uint32_t chains = ctx->chains;
    const uint32_t spc = PT_SHARDS / chains;                  // shards per chain
    const size_t need_events = 2 + 2 * (size_t)rounds + 1;
    const size_t chain_words = pt_chain_words(layout_rounds, spc);        // queue sizes, fetch cursors, deferred-ray count
    pt_prof_slot* slot = nullptr;
    bool window_start = false;
    if (prof) {
        if (counting) PT_SYNC(ctx);          // counted frames run one at a time
        slot = &ctx->prof_slot[ctx->prof_next++ % PT_PROF_SLOTS];
        {
            int rc = collect_slot(ctx, *slot);     // its previous frame, PT_PROF_SLOTS frames ago
            if (rc) return rc;
        }
        for (uint32_t c = 0; c < chains; c++)
            while (slot->ev[c].size() < need_events) {
                hipEvent_t e;
                PT_HIP(ctx, hipEventCreate(&e));
                slot->ev[c].push_back(e);
            }
        if (slot->h_words < chain_words * chains) {
            if (slot->h_counts) (void)hipHostFree(slot->h_counts);
            slot->h_counts = nullptr;
            PT_HIP(ctx, hipHostMalloc((void**)&slot->h_counts, chain_words * chains * sizeof(uint32_t), hipHostMallocDefault));
            slot->h_words = chain_words * chains;
        }
        slot->rounds = rounds;
        slot->layout_rounds = layout_rounds;
        slot->chains = chains;
        if (!ctx->window_open) {
            if (!ctx->prof_ref) PT_HIP(ctx, hipEventCreate(&ctx->prof_ref));
            PT_HIP(ctx, hipEventRecord(ctx->prof_ref, ctx->stream));
            ctx->timing = pt_trace_timing{};
            ctx->ext_spans.clear();
            ctx->window_open = true;
            window_start = true;
        }
    }

    // workgroup b of a chain's grid serves the chain's shard b % spc
    auto chain_grid = [&](uint32_t block, uint32_t cap_total, uint32_t frames) {
        uint32_t per_shard = (npix * frames / PT_SHARDS + block) / block;
        uint32_t cap_per_shard = cap_total / PT_SHARDS ? cap_total / PT_SHARDS : 1u;
        if (per_shard > cap_per_shard) per_shard = cap_per_shard;
        return per_shard * spc;
    };
    // traversal waves per CU over all chains: enough for every chain's launch to fill its share of the chip, few enough
    // for a wave to refill its lanes several times (measured: 40 for 8 chains; for 16 chains flat from 64 to 96, 48 is 3-4 %
    // slower on a mid-sized and on a large scene, 128 loses 1-3 %, 192 loses 5 %)
    static const uint32_t waves_override = getenv("PT_EXT_WAVES_PER_CU") ? (uint32_t)atoi(getenv("PT_EXT_WAVES_PER_CU")) : 0u;
    // a shallow scene (surface-area cost below 20: a ray is done after two or three node steps, so a wave mostly refills)
    // prefers fewer: the Cornell box loses 11 % at 96 and gains 8 ... 12 % from 48 to 32 (24 and 32 alike, 16 loses 8 %).
    // Deep scenes without suspension are flat from 24 to 64 in long runs at 1920x1080 (Bistro-class +3 % at 32), but a run of
    // twenty frames and 3840x2160 lose 3 % at 32: 64 stays
    const uint32_t ext_waves_per_cu = waves_override ? waves_override : (chains >= 16 ? (ctx->accel_stats.sah_cost >= 20.0f ? 64u : 32u) : 40u);
    uint32_t max_blocks = (uint32_t)ctx->num_cus * 8u;
    uint32_t ext_blocks = chain_grid(EXT_BLOCK, (uint32_t)ctx->num_cus * ext_waves_per_cu, nframes);
    static const uint32_t shade_override = getenv("PT_SHADE_WAVES_PER_CU") ? (uint32_t)atoi(getenv("PT_SHADE_WAVES_PER_CU")) : 0u;
    // shading: 24 waves per CU over all chains (flat from 12 to 32; with workgroups of four waves twice as many were 1.5-3.5 %
    // slower and half as many 1-5 %): the launch waits on gathers, and waves it holds beyond that only take registers from the
    // traversal.  A shallow scene, whose frame is mostly shading, takes 32 (the Cornell box: +3 % over 16)
    uint32_t sh_blocks = chain_grid(SHADE_BLOCK, (uint32_t)ctx->num_cus * (shade_override ? shade_override : (ctx->accel_stats.sah_cost >= 20.0f ? 24u : 32u)) * 64u / SHADE_BLOCK, nframes);
    // a chain's share of the spill area and of the suspension records is laid out for a full batch, so that a
    // short batch launched while a full one is still running stays inside its own chain's share
    const uint32_t ext_blocks_max = chain_grid(EXT_BLOCK, (uint32_t)ctx->num_cus * ext_waves_per_cu, PT_MAX_BATCH);
    {
        int rc = ensure_spill(ctx, ext_blocks_max * chains);
        if (rc) return rc;
    }
    // records of suspended traversals: per chain two pools (the round's own and the next round's), per shard one
    // record for each ray the shard's waves can suspend in a launch
    const uint32_t susp_cap = (ext_blocks_max / spc) * PT_CUT_LANES;
    const uint32_t susp_words = PT_SUSPEND_STATE_WORDS + 2u * ctx->stack_entries;
    {
        size_t need = (size_t)chains * 2 * spc * susp_cap * susp_words;
        if (defer && need > ctx->suspend_words) {
            PT_SYNC(ctx);
            if (ctx->d_suspend) (void)hipFree(ctx->d_suspend);
            ctx->d_suspend = nullptr;
            PT_HIP(ctx, hipMalloc((void**)&ctx->d_suspend, need * sizeof(uint32_t)));
            ctx->suspend_words = need;
        }
    }
    if (counting && window_start) {
        PT_HIP(ctx, hipMemsetAsync(ctx->d_stats, 0, 128, ctx->stream));
        ctx->main_dirty = true;
    }
    if (wave_trace) {
        // per-wave timeline of every extend launch, dumped by the timing read-out: [chain][round][wave]{start, exhausted, end, info}
        size_t need = (size_t)chains * rounds * ext_blocks * 4;
        if (need > ctx->wave_trace_len) {
            if (ctx->d_wave_trace) (void)hipFree(ctx->d_wave_trace);
            ctx->d_wave_trace = nullptr;
            PT_HIP(ctx, hipMalloc((void**)&ctx->d_wave_trace, need * sizeof(unsigned long long)));
            ctx->wave_trace_len = need;
        }
        ctx->wave_trace_rounds = rounds * chains;
        ctx->wave_trace_waves = ext_blocks;
        PT_HIP(ctx, hipMemsetAsync(ctx->d_wave_trace, 0, need * sizeof(unsigned long long), ctx->stream));
        ctx->main_dirty = true;
    }
    if (ctx->main_dirty && chains > 1) {
        PT_HIP(ctx, hipEventRecord(ctx->fork_event, ctx->stream));
        for (uint32_t c = 1; c < chains; c++) PT_HIP(ctx, hipStreamWaitEvent(ctx->chain_stream[c], ctx->fork_event, 0));
    }
    ctx->main_dirty = false;

    uint32_t tiles = ((P.width + 7u) >> 3) * ((P.height + 7u) >> 3);
    for (uint32_t c = 0; c < chains; c++) {
        hipStream_t st = ctx->chain_stream[c];
        size_t ev = 0;
        auto mark = [&]() -> hipError_t { return prof ? hipEventRecord(slot->ev[c][ev++], st) : hipSuccess; };
        uint32_t* counts = ctx->d_counts + c * chain_words;
        uint32_t* cursors = counts + (size_t)(layout_rounds + 1) * spc;
        uint32_t* deferred = cursors + (size_t)(layout_rounds + 1) * spc;
        Queues q;
        q.hits = ctx->d_hits;
        q.hit_det = ctx->d_hit_det;
        q.stC = ctx->d_stC;
        q.counts = counts;
        q.shard0 = c * spc; q.nshards = spc;
        q.shard_cap = ctx->shard_cap;
        q.primary = ctx->d_primary;
        q.image = ctx->d_image; q.accum = ctx->d_accum;
        q.partial = ctx->d_partial;
        q.views = ctx->d_views + c * PT_MAX_BATCH;
        k_set_views<<<1, 64, 0, st>>>(views, ctx->d_views + c * PT_MAX_BATCH, counts, (uint32_t)chain_words);
        uint2* spill = ctx->d_spill + (size_t)c * ext_blocks_max * (ctx->stack_entries * 64);

        PT_HIP(ctx, mark());
        {
            q.rayO_in = q.rayO_out = ctx->d_rayO[0];
            q.rayD_in = q.rayD_out = ctx->d_rayD[0];
            q.stA_in = q.stA_out = ctx->d_stA[0];
            q.stB_in = q.stB_out = ctx->d_stB[0];
            uint32_t blocks = ((tiles / chains + spc) * nframes * 64u + 255u) / 256u;
            if (blocks > max_blocks) blocks = max_blocks;
            k_raygen<<<blocks, 256, 0, st>>>(P, q);
        }
        PT_HIP(ctx, mark());
        for (uint32_t r = 0; r < rounds; r++) {
            int in = (int)(r & 1u), out = in ^ 1;
            ExtendArgs ea;
            ea.tmin = P.tmin; ea.tmax = P.tmax;
            ea.counts = counts + r * spc; ea.cursors = cursors + r * spc;
            ea.next_counts = (defer && r + 1 < rounds) ? counts + (r + 1) * spc : nullptr;
            ea.max_lag = defer;
            ea.susp_cap = susp_cap;
            ea.susp_words = susp_words;
            ea.stack_entries = ctx->stack_entries;
            // records only exist once a batch has suspended rays; a launch that does not suspend never looks at them
            ea.susp_in = defer ? ctx->d_suspend + ((size_t)(c * 2 + in) * spc) * susp_cap * susp_words : nullptr;
            ea.susp_out = defer ? ctx->d_suspend + ((size_t)(c * 2 + out) * spc) * susp_cap * susp_words : nullptr;
            ea.deferred = deferred;
            ea.shard0 = q.shard0; ea.nshards = spc; ea.shard_cap = ctx->shard_cap;
            ea.rayO = ctx->d_rayO[in]; ea.rayD = ctx->d_rayD[in];
            ea.next_rayO = ctx->d_rayO[out]; ea.next_rayD = ctx->d_rayD[out];
            ea.stA = ctx->d_stA[in]; ea.stB = ctx->d_stB[in];
            ea.next_stA = ctx->d_stA[out]; ea.next_stB = ctx->d_stB[out];
            ea.hits = ctx->d_hits;
            ea.hit_det = ctx->d_hit_det;
            ea.spill = spill;
            ea.iter_cap = ctx->iter_cap;
            ea.fault = ctx->d_fault;
            unsigned long long* wt = wave_trace ? ctx->d_wave_trace + ((size_t)c * rounds + r) * ext_blocks * 4 : nullptr;
            if (counting && ea.next_counts) k_extend<true, true><<<ext_blocks, EXT_BLOCK, 0, st>>>(sv, ea, ctx->d_stats, wt);
            else if (counting) k_extend<true, false><<<ext_blocks, EXT_BLOCK, 0, st>>>(sv, ea, ctx->d_stats, wt);
            else if (ea.next_counts) k_extend<false, true><<<ext_blocks, EXT_BLOCK, 0, st>>>(sv, ea, nullptr, nullptr);
            else k_extend<false, false><<<ext_blocks, EXT_BLOCK, 0, st>>>(sv, ea, nullptr, nullptr);
            PT_HIP(ctx, mark());
            q.rayO_in = ctx->d_rayO[in]; q.rayD_in = ctx->d_rayD[in];
            q.rayO_out = ctx->d_rayO[out]; q.rayD_out = ctx->d_rayD[out];
            q.stA_in = ctx->d_stA[in]; q.stB_in = ctx->d_stB[in];
            q.stA_out = ctx->d_stA[out]; q.stB_out = ctx->d_stB[out];
            if (ctx->precision == PT_PRECISION_FAST) fast::k_shade<<<sh_blocks, SHADE_BLOCK, 0, st>>>(sv, P, q, r);
            else k_shade<<<sh_blocks, SHADE_BLOCK, 0, st>>>(sv, P, q, r);
            PT_HIP(ctx, mark());
        }
        {
            uint32_t blocks = ((tiles / chains + spc) * 64u + 255u) / 256u;
            if (blocks > max_blocks) blocks = max_blocks;
            k_accumulate<<<blocks, 256, 0, st>>>(P, q, rounds * spc, deferred, ctx->d_total_rays);
        }
        if (prof) PT_HIP(ctx, hipMemcpyAsync(slot->h_counts + c * chain_words, counts, chain_words * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, mark());
    }
    PT_HIP(ctx, hipGetLastError());
    ctx->join_pending = chains > 1;
    if (slot) slot->pending = true;
    return PT_OK;
}

int launch_resolve(pt_ctx* ctx)
{
    int rc = join_chains(ctx);
    if (rc) return rc;
    ctx->main_dirty = true;
    uint32_t n = ctx->width * ctx->height;
    k_resolve<<<(n + 255) / 256, 256, 0, ctx->stream>>>(n, ctx->d_accum, ctx->d_image);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

// ---- inspection hooks --------------------------------------------------------------------------
__global__ void __launch_bounds__(EXT_BLOCK) k_debug_intersect(SceneView sc, uint32_t n, uint32_t* cursor, const float4* __restrict__ rays,
                                                                uint32_t* __restrict__ out, uint2* spill, uint32_t stack_entries, unsigned long long* stats,
                                                                uint32_t iter_cap, uint32_t* fault)
{
    __shared__ uint2 stack[PT_STACK_LDS * EXT_BLOCK];
    __shared__ __attribute__((aligned(16))) uint32_t scratch[PT_SCRATCH_WORDS];
    uint32_t sn = 0, st = 0;
    trace_wave<true, false>(
        sc.nodes, sc.tris, sc.num_tris, n, cursor, stack + threadIdx.x, EXT_BLOCK,
        spill + (size_t)blockIdx.x * (stack_entries * 64) + threadIdx.x, scratch,
        [&](uint32_t i, v3& o, v3& d, float& tn, float& tx) {
            float4 ro = rays[2 * i], rd = rays[2 * i + 1];
            o = V3(ro.x, ro.y, ro.z);
            d = V3(rd.x, rd.y, rd.z);
            tn = fmaxf(ro.w, 0.0f);        // tmin >= 0 is part of the contract (the host checks host-side rays); NaN -> 0
            tx = rd.w;
            return 0u;
        },
        [&](uint32_t i, const Trav& s) {
            Trav fin = s;
            finish_hit(fin);
            const HitRec& h = fin.best;
            uint32_t* o = out + (size_t)i * 5;
            if (h.tri == MISS) { o[0] = f2u(h.t); o[1] = 0; o[2] = 0; o[3] = MISS; o[4] = MISS; }   // h.t is still tmax
            else { o[0] = f2u(h.t); o[1] = f2u(h.u); o[2] = f2u(h.v); o[3] = sc.tris[h.tri].submesh; o[4] = sc.tris[h.tri].prim; }
        },
        [&](uint32_t, const Trav&, const uint2*, int, const uint2*) {}, [&](uint32_t) { return (const uint32_t*)nullptr; }, sn, st, iter_cap, fault);
    atomicAdd(&stats[0], (unsigned long long)sn);
    atomicAdd(&stats[1], (unsigned long long)st);
}

__global__ void k_debug_math(int fn, uint32_t n, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    switch (fn) {
    case 0: det_sincos(x[i], s, c); out[i] = s; break;
    case 1: det_sincos(x[i], s, c); out[i] = c; break;
    case 2: out[i] = det_log(x[i]); break;
    case 3: out[i] = det_exp(x[i]); break;
    case 4: out[i] = det_pow(x[i], y[i]); break;
    case 5: out[i] = det_atan2(x[i], y[i]); break;
    case 6: out[i] = det_acos(x[i]); break;
    case 7: out[i] = det_tan(x[i]); break;
    default: out[i] = round_to_half(x[i]); break;
    }
}

// one body for both arithmetic namespaces: A = ptd (bit-exact) or ptd::fast
#define PT_DEBUG_DISNEY_BODY(NS)                                                                                          \
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;                                                                    \
    if (i >= n) return;                                                                                                    \
    const float* s = surf + (size_t)i * 17;                                                                                \
    NS::Surf p;                                                                                                            \
    p.albedo = NS::V3(s[0], s[1], s[2]);                                                                                   \
    p.metallic = s[3]; p.roughness = s[4]; p.subsurface = s[5]; p.specularTint = s[6];                                     \
    p.sheen = s[7]; p.sheenTint = s[8]; p.clearcoat = s[9]; p.clearcoatRoughness = s[10];                                  \
    p.specTrans = s[11]; p.ior = s[12]; p.ax = s[13]; p.ay = s[14]; p.eta = s[15];                                         \
    NS::v3 v = NS::V3(V[3 * i], V[3 * i + 1], V[3 * i + 2]), nn = NS::V3(N[3 * i], N[3 * i + 1], N[3 * i + 2]);            \
    float pd;                                                                                                              \
    NS::v3 r;                                                                                                              \
    if (sample) {                                                                                                          \
        uint32_t seed = seeds[i];                                                                                          \
        NS::v3 L;                                                                                                          \
        r = NS::disney_sample(p, v, nn, L, pd, seed);                                                                      \
        seeds[i] = seed;                                                                                                   \
        Lout[3 * i] = L.x; Lout[3 * i + 1] = L.y; Lout[3 * i + 2] = L.z;                                                   \
    } else {                                                                                                               \
        r = NS::disney_eval(p, v, nn, NS::V3(Lin[3 * i], Lin[3 * i + 1], Lin[3 * i + 2]), pd);                             \
    }                                                                                                                      \
    f[3 * i] = r.x; f[3 * i + 1] = r.y; f[3 * i + 2] = r.z;                                                                \
    pdf[i] = pd;

__global__ void k_debug_disney(bool sample, uint32_t n, const float* __restrict__ surf, const float* __restrict__ V,
                               const float* __restrict__ N, const float* __restrict__ Lin, uint32_t* seeds,
                               float* __restrict__ f, float* __restrict__ Lout, float* __restrict__ pdf)
{
    PT_DEBUG_DISNEY_BODY(ptd)
}

__global__ void k_debug_disney_fast(bool sample, uint32_t n, const float* __restrict__ surf, const float* __restrict__ V,
                                    const float* __restrict__ N, const float* __restrict__ Lin, uint32_t* seeds,
                                    float* __restrict__ f, float* __restrict__ Lout, float* __restrict__ pdf)
{
    PT_DEBUG_DISNEY_BODY(ptd::fast)
}
#undef PT_DEBUG_DISNEY_BODY

template <class T>
struct Tmp {
    T* p = nullptr;
    ~Tmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

int debug_intersect(pt_ctx* ctx, const float* rays, uint32_t n, uint32_t* hits, uint64_t* counters)
{
    PT_SYNC(ctx);
    hipStream_t st = ctx->stream;
    Tmp<float4> d_rays;
    Tmp<uint32_t> d_out;
    PT_HIP(ctx, d_rays.alloc(2 * (size_t)n));
    PT_HIP(ctx, d_out.alloc(5 * (size_t)n));
    PT_HIP(ctx, hipMemcpyAsync(d_rays.p, rays, (size_t)n * 32, hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemsetAsync(ctx->d_stats, 0, 128, st));   // two counters + the fetch cursor + wave statistics
    uint32_t blocks = (n + EXT_BLOCK - 1) / EXT_BLOCK;
    uint32_t cap = (uint32_t)ctx->num_cus * 20u;
    if (blocks > cap) blocks = cap;
    {
        int rc = ensure_spill(ctx, blocks);
        if (rc) return rc;
    }
    if (n) k_debug_intersect<<<blocks, EXT_BLOCK, 0, st>>>(scene_view(ctx), n, reinterpret_cast<uint32_t*>(ctx->d_stats + 2), d_rays.p, d_out.p, ctx->d_spill, ctx->stack_entries, ctx->d_stats, ctx->iter_cap, ctx->d_fault);
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipMemcpyAsync(hits, d_out.p, (size_t)n * 20, hipMemcpyDeviceToHost, st));
    unsigned long long hs[2] = {0, 0};
    PT_HIP(ctx, hipMemcpyAsync(hs, ctx->d_stats, 16, hipMemcpyDeviceToHost, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    if (counters) { counters[0] = hs[0]; counters[1] = hs[1]; }
    return check_fault(ctx);
}

// ---- closest hits without any allocation ------------------------------------------------------------
// Scratch of the allocation-free ray query: a 256-byte header (two statistics counters and the fetch cursor of
// the persistent waves, which must start at zero) followed by the traversal-stack spill area of
// PT_NOALLOC_WAVES waves, PT_NOALLOC_STACK entries of 64 lanes each.  One copy per GPU, held until the process
// ends; every call on a GPU uses it, so calls that overlap on one GPU race.
static constexpr uint32_t PT_NOALLOC_WAVES = 512;
static constexpr uint32_t PT_NOALLOC_STACK = 128;    // stack entries per lane: hierarchies of up to 18 levels (7 per level + 1)
static constexpr size_t PT_NOALLOC_HEADER = 256;
static constexpr size_t PT_NOALLOC_BYTES = PT_NOALLOC_HEADER + (size_t)PT_NOALLOC_WAVES * PT_NOALLOC_STACK * 64 * sizeof(uint2);
static __device__ __align__(256) unsigned char scratch[PT_NOALLOC_BYTES];

// Who used a GPU's array last: a query of another context that has not finished yet would share the fetch cursor and
// the spill area with a new one, and the result would be wrong hits, not an error.  Queries of one context follow each
// other on its stream and need no check.
struct NoallocOwner { pt_ctx* ctx = nullptr; hipEvent_t done = nullptr; };
static std::mutex g_noalloc_mutex;
static NoallocOwner g_noalloc_owner[64];

// the array's address on the device that is current; called once per context, at pt_create
int noalloc_init(pt_ctx* ctx)
{
    void* ptr = nullptr;
    PT_HIP(ctx, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return set_error(ctx, PT_ERR_HIP, "hipGetSymbolAddress returned a null address for the ray-query scratch");
    ctx->noalloc_scratch = ptr;
    PT_HIP(ctx, hipEventCreateWithFlags(&ctx->noalloc_done, hipEventDisableTiming));    // made here, so that the query itself creates nothing
    return PT_OK;
}

// pt_destroy: the context's streams are idle by now
void noalloc_release(pt_ctx* ctx)
{
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (ctx->device >= 0 && ctx->device < 64 && g_noalloc_owner[ctx->device].ctx == ctx) g_noalloc_owner[ctx->device] = NoallocOwner();
    if (ctx->noalloc_done) (void)hipEventDestroy(ctx->noalloc_done);
    ctx->noalloc_done = nullptr;
}

int intersect_noalloc(pt_ctx* ctx, const float* d_rays, uint32_t n, uint32_t* d_hits)
{
    if (ctx->stack_entries > PT_NOALLOC_STACK)
        return set_error(ctx, PT_ERR_TOO_LARGE, "pt_intersect_noalloc: the hierarchy has " + std::to_string(ctx->accel_stats.max_depth) +
                         " levels; the fixed scratch holds traversal stacks for at most 18");
    if (n > PT_NOALLOC_MAX_RAYS) return set_error(ctx, PT_ERR_TOO_LARGE, "pt_intersect_noalloc: more than 2^24 rays in one call");
    if (!ctx->pending.empty()) return set_error(ctx, PT_ERR_INVALID, "pt_intersect_noalloc: frames are queued; call pt_sync first");
    if (!ctx->noalloc_scratch || !ctx->noalloc_done) return set_error(ctx, PT_ERR_INVALID, "pt_intersect_noalloc: scratch address missing");
    {
        // the kernel's arguments are __restrict__: the hit records must not overlap the rays
        const uintptr_t r0 = (uintptr_t)d_rays, r1 = r0 + (size_t)n * 32, h0 = (uintptr_t)d_hits, h1 = h0 + (size_t)n * 20;
        if (n && r0 < h1 && h0 < r1) return set_error(ctx, PT_ERR_INVALID, "pt_intersect_noalloc: device_hits overlaps device_rays");
    }
    if (ctx->device < 0 || ctx->device >= 64) return set_error(ctx, PT_ERR_INVALID, "pt_intersect_noalloc: device ordinal beyond 63");
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    NoallocOwner& owner = g_noalloc_owner[ctx->device];
    if (owner.ctx && owner.ctx != ctx && hipEventQuery(owner.done) == hipErrorNotReady)
        return set_error(ctx, PT_ERR_INVALID, "pt_intersect_noalloc: a query of another context is still running on this GPU and uses the same scratch array; "
                                              "wait for it (pt_sync on that context) first");
    (void)hipGetLastError();            // hipEventQuery's "not ready" is not an error of this call
    int rc = join_chains(ctx);          // event record + stream wait only
    if (rc) return rc;
    if (n == 0) return PT_OK;
    hipStream_t st = ctx->stream;
    unsigned char* base = static_cast<unsigned char*>(ctx->noalloc_scratch);
    unsigned long long* stats = reinterpret_cast<unsigned long long*>(base);
    PT_HIP(ctx, hipMemsetAsync(base, 0, PT_NOALLOC_HEADER, st));     // the cursor and the counters start at zero; the spill area needs no clearing
    uint32_t blocks = (n + EXT_BLOCK - 1) / EXT_BLOCK;
    if (blocks > PT_NOALLOC_WAVES) blocks = PT_NOALLOC_WAVES;
    k_debug_intersect<<<blocks, EXT_BLOCK, 0, st>>>(scene_view(ctx), n, reinterpret_cast<uint32_t*>(stats + 2), reinterpret_cast<const float4*>(d_rays), d_hits,
                                                    reinterpret_cast<uint2*>(base + PT_NOALLOC_HEADER), ctx->stack_entries, stats, ctx->iter_cap, ctx->d_fault);
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipEventRecord(ctx->noalloc_done, st));
    owner.ctx = ctx;
    owner.done = ctx->noalloc_done;
    ctx->main_dirty = true;
    return PT_OK;
}

int debug_math(pt_ctx* ctx, int fn, const float* x, const float* y, float* out, uint32_t n)
{
    hipStream_t st = ctx->stream;
    Tmp<float> dx, dy, dout;
    PT_HIP(ctx, dx.alloc(n));
    PT_HIP(ctx, dy.alloc(n));
    PT_HIP(ctx, dout.alloc(n));
    PT_HIP(ctx, hipMemcpyAsync(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
    if (y) PT_HIP(ctx, hipMemcpyAsync(dy.p, y, (size_t)n * 4, hipMemcpyHostToDevice, st));
    else PT_HIP(ctx, hipMemsetAsync(dy.p, 0, (size_t)n * 4, st));
    if (n) k_debug_math<<<(n + 255) / 256, 256, 0, st>>>(fn, n, dx.p, dy.p, dout.p);
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipMemcpyAsync(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    return PT_OK;
}

int debug_disney(pt_ctx* ctx, bool sample, uint32_t n, const float* surf, const float* V, const float* N,
                 const float* Lin, uint32_t* seeds, float* f, float* Lout, float* pdf)
{
    hipStream_t st = ctx->stream;
    Tmp<float> dsurf, dV, dN, dL, df, dLo, dpdf;
    Tmp<uint32_t> dseed;
    PT_HIP(ctx, dsurf.alloc((size_t)n * 17));
    PT_HIP(ctx, dV.alloc((size_t)n * 3));
    PT_HIP(ctx, dN.alloc((size_t)n * 3));
    PT_HIP(ctx, dL.alloc((size_t)n * 3));
    PT_HIP(ctx, df.alloc((size_t)n * 3));
    PT_HIP(ctx, dLo.alloc((size_t)n * 3));
    PT_HIP(ctx, dpdf.alloc(n));
    PT_HIP(ctx, dseed.alloc(n));
    PT_HIP(ctx, hipMemcpyAsync(dsurf.p, surf, (size_t)n * 68, hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(dV.p, V, (size_t)n * 12, hipMemcpyHostToDevice, st));
    PT_HIP(ctx, hipMemcpyAsync(dN.p, N, (size_t)n * 12, hipMemcpyHostToDevice, st));
    if (Lin) PT_HIP(ctx, hipMemcpyAsync(dL.p, Lin, (size_t)n * 12, hipMemcpyHostToDevice, st));
    if (seeds) PT_HIP(ctx, hipMemcpyAsync(dseed.p, seeds, (size_t)n * 4, hipMemcpyHostToDevice, st));
    if (n && ctx->precision == PT_PRECISION_FAST) k_debug_disney_fast<<<(n + 255) / 256, 256, 0, st>>>(sample, n, dsurf.p, dV.p, dN.p, dL.p, dseed.p, df.p, dLo.p, dpdf.p);
    else if (n) k_debug_disney<<<(n + 255) / 256, 256, 0, st>>>(sample, n, dsurf.p, dV.p, dN.p, dL.p, dseed.p, df.p, dLo.p, dpdf.p);
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipMemcpyAsync(f, df.p, (size_t)n * 12, hipMemcpyDeviceToHost, st));
    PT_HIP(ctx, hipMemcpyAsync(pdf, dpdf.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (sample) {
        PT_HIP(ctx, hipMemcpyAsync(Lout, dLo.p, (size_t)n * 12, hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipMemcpyAsync(seeds, dseed.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    }
    PT_HIP(ctx, hipStreamSynchronize(st));
    return PT_OK;
}

} // namespace ptd
