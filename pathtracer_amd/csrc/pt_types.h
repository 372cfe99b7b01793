// Device data layouts shared by the build kernels, the trace kernels and the host API.
#pragma once
#include <stdint.h>

namespace ptd {

// 4-wide hierarchy node, 128 bytes = one L2 cache line.  Child boxes are stored
// structure-of-arrays so four slab tests read eight 16-byte vectors.
// child[k]: 0xFFFFFFFF = empty; bit 31 set = leaf with triangles
// [first, first+count) where first = bits 0..27 and count-1 = bits 28..30;
// otherwise the index of another node.
struct alignas(128) Node4 {
    float lox[4], loy[4], loz[4];
    float hix[4], hiy[4], hiz[4];
    uint32_t child[4];
    uint32_t pad[4];
};

#define PT_CHILD_EMPTY 0xFFFFFFFFu
#define PT_CHILD_LEAF 0x80000000u
#ifndef PT_LEAF_MAX
#define PT_LEAF_MAX 4u
#endif
#define PT_SHARDS 32u     // independent sub-queues of the wavefront ray queue (multiple of the 8 XCDs)

// world-space triangle, 48 bytes, in leaf order: (v0, submesh) (v1, prim) (v2, 0)
struct alignas(16) TriRec {
    float v0x, v0y, v0z; uint32_t submesh;
    float v1x, v1y, v1z; uint32_t prim;
    float v2x, v2y, v2z; uint32_t pad;
};

struct TraceParams {
    uint32_t width, height;
    uint32_t spp, max_bounces, rr_depth;
    float sky_scale, eps, tmin, tmax;
    uint32_t frame;
    float inv_view[16];   // column-major
    float inv_proj[16];
};

// scene tables as uploaded (reference layouts) + build products
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
    const Node4* nodes;
    const TriRec* tris;
};

} // namespace ptd
