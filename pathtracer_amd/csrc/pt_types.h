// Device data layouts shared by the build kernels, the trace kernels and the host API.
#pragma once
#include <stdint.h>

namespace ptd {

// 8-wide hierarchy node, 128 bytes = one L2 cache line.  Child boxes are quantised to 8 bits per
// plane on a per-node grid: plane = origin + q * 2^(exp - 127), rounded outwards, so a decoded box
// always contains the exact one.  The whole hierarchy of a Sponza-class scene is then ~2 MB and
// stays in every XCD's 4 MB L2.
//   words 0..2   origin x, y, z (float)
//   word  3      exponent bytes: x | y << 8 | z << 16 (biased like a float exponent)
//   words 4..11  child[8]: 0xFFFFFFFF = empty; bit 31 set = leaf with triangles [first, first+count)
//                where first = bits 0..27 and count-1 = bits 28..30; otherwise a node index
//   words 12..23 quantised planes, 8 bytes (one per child) each: lo.x lo.y lo.z hi.x hi.y hi.z
//                (an empty child has lo = 255 and hi = 0 on every axis and can never be entered)
//   words 24..31 unused
struct alignas(128) Node8 {
    float origin[3];
    uint32_t exps;
    uint32_t child[8];
    uint32_t q[12];
    uint32_t pad[8];
};

#define PT_CHILD_EMPTY 0xFFFFFFFFu
#define PT_CHILD_LEAF 0x80000000u
#ifndef PT_LEAF_MAX
#define PT_LEAF_MAX 4u
#endif
#define PT_SHARDS 32u     // independent sub-queues of the wavefront ray queue (multiple of the 8 XCDs)
#ifndef PT_MAX_CHAINS
#define PT_MAX_CHAINS 16    // independent pipelines a frame can be split into (a divisor of PT_SHARDS)
#endif

// world-space triangle, 48 bytes, in leaf order: (v0, submesh) (v1, prim) (v2, part).  Inside a leaf the records are in
// ascending (submesh, prim) order: the traversal's triangle phase settles hits at equal distance by position in the leaf.
// part = 0 for a triangle entered whole; for a reference to a part of a split triangle, the bounds of the part's
// weights in the frame of the triangle's squarest corner (origin vertex + 1 in bits 28-31, four 7-bit bounds below), from
// which a refit bounds the part again
struct alignas(16) TriRec {
    float v0x, v0y, v0z; uint32_t submesh;
    float v1x, v1y, v1z; uint32_t prim;
    float v2x, v2y, v2z; uint32_t pad;      // the part code
};

#ifndef PT_MAX_BATCH
#define PT_MAX_BATCH 8      // frames traced together: consecutive pt_trace calls share their launches
#endif

// what differs between the frames of a batch
struct FrameView {
    uint32_t frame;       // FrameIndex of that pt_trace call
    float inv_view[16];   // column-major
    float inv_proj[16];
};

struct FrameViewBatch { FrameView v[PT_MAX_BATCH]; };

struct TraceParams {
    uint32_t width, height;
    uint32_t spp, max_bounces, rr_depth;
    float sky_scale, eps, tmin, tmax;
    uint32_t nframes;     // frames in this batch; a path is identified by frame slot * width * height + pixel
};

// scene tables as uploaded (reference layouts) + build products
// What shading needs of a triangle, gathered once at build time in leaf order (the order of TriRec): the three
// vertices' texture coordinates and object-space normals, the material, and the ids.  One 80-byte read in place of
// the chain triangle -> submesh record -> three indices -> three vertices.
struct ShadeTri {
    float uv[3][2];
    float n[3][3];
    uint32_t material, submesh, prim;
    uint32_t pad[2];
};
static_assert(sizeof(ShadeTri) == 80, "ShadeTri is five 16-byte words");

struct SceneView {
    const float* vertices;      // 12 floats per vertex
    const uint32_t* indices;
    const uint32_t* submeshes;  // 4 per submesh
    const float* xforms;        // 12 per submesh
    const float* materials;     // 22 words per material
    const uint8_t* tex_pool;    // all textures' RGBA8 texels
    const uint32_t* tex_table;  // per texture: byte offset into tex_pool (in texels), width, height, flags
    const float* env;           // 6 * n * n * 4
    uint32_t env_n;
    uint32_t num_tris;
    const Node8* nodes;
    const TriRec* tris;
    const ShadeTri* shade_tris;
};

} // namespace ptd
