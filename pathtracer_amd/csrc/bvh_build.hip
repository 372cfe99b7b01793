// GPU build of the ray-tracing hierarchy: the replacement for the driver-built BLAS/TLAS the
// reference requests in CreateAccelerationStructure() (PathTracer/src/RayTracingLayer.cpp:342-348;
// one TLAS instance per submesh, instance custom index = submesh index, ClosestHit.glsl:147-150).
//
// Pipeline (all on the context's stream):
//   1. k_world_tris   every submesh triangle -> world space (object-to-world 3x4), its box, scene bounds
//   1b. k_split_*     large triangles whose boxes are mostly empty are entered once per cell they pass through (one wave per such triangle)
//   2. k_morton       63-bit Morton code of the box centre
//   3. radix sort     (rocPRIM via hipcub) of (code, triangle)
//   4. binary hierarchy
//      a. default: binary radix tree over the sorted codes (Karras 2012, k_karras), bottom-up boxes with an arrival
//         counter per node (k_refit), then treelet restructuring passes (Karras and Aila 2013, k_treelet_up): every
//         treelet of eight subtrees is put into the topology with the smallest summed box area.  PT_BUILD_NO_TREELET
//         leaves the passes out (fastest build).
//      b. PT_BUILD_CLUSTER: parallel locally-ordered clustering -- every cluster looks PLOC_RADIUS places left and
//         right in Morton order for the neighbour that gives the smallest merged box, mutual nearest neighbours
//         merge, the cluster list is compacted with a prefix scan, repeat until one cluster is left
//         (k_ploc_nearest, k_ploc_flags, scan, k_ploc_merge); the same treelet passes follow.
//      With PT_BUILD_CLUSTER | PT_BUILD_LBVH both are built and the one the collapse below prices cheaper is kept.
//   5. k_collapse_dp  which subtrees become the children of an 8-wide node: surface-area-optimal, by a dynamic programme
//                     over the binary tree (PT_BUILD_GREEDY: open the largest box first)
//      k_collapse     level by level: binary tree -> 8-wide nodes of 128 bytes with 8-bit quantised child
//                     boxes, leaves of <= 4 triangles, triangles are written out in leaf order
#include "pt_context.h"
#include "pt_traverse.h"
#include "pt_math.h"
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

// The bottom-up passes below hand data from the thread that arrives first at a node to the one that arrives second with
// relaxed agent-scope atomics and an inline "s_waitcnt vmcnt(0)" as the release: that is only a release on targets whose
// stores count in vmcnt (the gfx9 family); on a target with a separate store counter the hand-over would publish node
// boxes and children before the stores land.  This file is gfx950 code.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "bvh_build.hip relies on gfx950 memory-counter semantics (s_waitcnt vmcnt(0) as the release of a store): build with --offload-arch=gfx950"
#endif

namespace ptd {

static constexpr uint32_t NONE = 0xFFFFFFFFu;


__device__ __forceinline__ int f2ord(float f)
{
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void k_init_bounds(int* bounds)
{
    if (threadIdx.x < 3) bounds[threadIdx.x] = 0x7fffffff;
    else if (threadIdx.x < 6) bounds[threadIdx.x] = (int)0x80000000;
}

// object-to-world: ((m0*x + m1*y) + m2*z) + m3, no fused operations
__device__ __forceinline__ v3 xform_point(const float* __restrict__ m, v3 p)
{
    return V3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3],
              m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}

// REFIT: tris already holds the records in leaf order and keeps that order; each one is recomputed from the
// (submesh, prim) it names under the current transforms
template <bool REFIT>
__global__ void k_world_tris(uint32_t n, uint32_t ns, const uint32_t* __restrict__ sub_tri_offset,
                             const uint32_t* __restrict__ submeshes, const float* __restrict__ xforms,
                             const uint32_t* __restrict__ indices, const float* __restrict__ vertices,
                             TriRec* tris, float4* __restrict__ tlo, float4* __restrict__ thi, int* bounds)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    if (g < n) {
        uint32_t s, prim, code = 0;
        if (REFIT) {
            s = tris[g].submesh;
            prim = tris[g].prim;
            code = tris[g].pad;         // a reference to a part of a split triangle: barycentric bounds of the part (part_code)
        } else {
            // submesh s with sub_tri_offset[s] <= g < sub_tri_offset[s+1]
            uint32_t a = 0, b = ns;
            while (b - a > 1) {
                uint32_t m = (a + b) >> 1;
                if (sub_tri_offset[m] <= g) a = m; else b = m;
            }
            s = a;
            prim = g - sub_tri_offset[s];
        }
        uint32_t vo = submeshes[s * 4 + 1], io = submeshes[s * 4 + 2];
        const float* m = xforms + s * 12;
        v3 w[3];
        for (int k = 0; k < 3; k++) {
            uint32_t vi = indices[io + prim * 3 + k] + vo;
            const float* v = vertices + (size_t)vi * 12;
            w[k] = xform_point(m, V3(v[0], v[1], v[2]));
        }
        TriRec t;
        t.v0x = w[0].x; t.v0y = w[0].y; t.v0z = w[0].z; t.submesh = s;
        t.v1x = w[1].x; t.v1y = w[1].y; t.v1z = w[1].z; t.prim = prim;
        t.v2x = w[2].x; t.v2y = w[2].y; t.v2z = w[2].z; t.pad = code;
        tris[g] = t;
        lo[0] = fminf(fminf(w[0].x, w[1].x), w[2].x); hi[0] = fmaxf(fmaxf(w[0].x, w[1].x), w[2].x);
        lo[1] = fminf(fminf(w[0].y, w[1].y), w[2].y); hi[1] = fmaxf(fmaxf(w[0].y, w[1].y), w[2].y);
        lo[2] = fminf(fminf(w[0].z, w[1].z), w[2].z); hi[2] = fmaxf(fmaxf(w[0].z, w[1].z), w[2].z);
        if (REFIT && code != 0u) {
            // the part lies inside the parallelogram of its (u, v) bounds (see part_code), whose image under the new transform is
            // spanned by four corner points; their box, widened by the rounding of the corner arithmetic, cut to the triangle's own box
            const float u0 = (float)(code & 127u) * (1.0f / 127.0f), v0 = (float)((code >> 7) & 127u) * (1.0f / 127.0f);
            const float u1 = (float)((code >> 14) & 127u) * (1.0f / 127.0f), v1 = (float)((code >> 21) & 127u) * (1.0f / 127.0f);
            const int org = (int)(code >> 28) - 1;
            const v3 o = w[org], e1 = w[(org + 1) % 3] - o, e2 = w[(org + 2) % 3] - o;
            const v3 c00 = o + e1 * u0 + e2 * v0, c10 = o + e1 * u1 + e2 * v0, c01 = o + e1 * u0 + e2 * v1, c11 = o + e1 * u1 + e2 * v1;
            for (int k = 0; k < 3; k++) {
                const float a0 = comp(c00, k), a1 = comp(c10, k), a2 = comp(c01, k), a3 = comp(c11, k);
                const float pl = fminf(fminf(a0, a1), fminf(a2, a3)), ph = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
                const float slack = (fabsf(lo[k]) + fabsf(hi[k]) + (hi[k] - lo[k])) * 0x1p-20f;
                lo[k] = fmaxf(lo[k], pl - slack);
                hi[k] = fminf(hi[k], ph + slack);
            }
        }
        // a triangle with a NaN or infinite coordinate can never be hit (every test on it compares false): it gets the
        // empty box, the identity of box union, so that it neither moves the scene bounds nor attracts a ray
        float sum = 0.0f;
        for (int k = 0; k < 3; k++) sum += (w[k].x - w[k].x) + (w[k].y - w[k].y) + (w[k].z - w[k].z);   // 0 iff all nine are finite
        if (!(sum == 0.0f)) {
            for (int k = 0; k < 3; k++) { lo[k] = 3.4028235e38f; hi[k] = -3.4028235e38f; }
        }
        tlo[g] = make_float4(lo[0], lo[1], lo[2], 0.0f);
        thi[g] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
    // scene bounds: wave reduction, one atomic per wave and component
    for (int k = 0; k < 3; k++) {
        float l = lo[k], h = hi[k];
        for (int off = 32; off > 0; off >>= 1) {
            l = fminf(l, __shfl_xor(l, off));
            h = fmaxf(h, __shfl_xor(h, off));
        }
        if ((threadIdx.x & 63) == 0) {
            if (l <= h) {
                atomicMin(&bounds[k], f2ord(l));
                atomicMax(&bounds[3 + k], f2ord(h));
            }
        }
    }
}

// ---- triangle splitting -------------------------------------------------------------------------------------------
// A long thin triangle that runs diagonally through the scene has a box that is almost empty, and every ray that
// crosses the box pays for it.  Such a triangle is entered into the hierarchy several times: its box is cut into
// cubic cells (an eighth of its longest side, a quarter if that gives more than PT_SPLIT_MAX pieces) and every cell the
// triangle actually passes through becomes a reference of its own, bounded by the part of the triangle inside the
// cell.  All references carry the same triangle record, so hits (t, barycentrics, submesh, primitive) do not change.
// A triangle is split when its box is large (more than 1/4096 of the scene's box area) and wasteful (more than 4
// times the two-sided area of the triangle: an axis-aligned right triangle has 2; a limit of 2.5 also splits the
// diagonal halves of large axis-aligned quads, which measured 1 to 4 % slower).
#define PT_SPLIT_MAX 64u
#ifndef PT_SPLIT_AREA_DIV
#define PT_SPLIT_AREA_DIV 4096.0f
#endif
#ifndef PT_SPLIT_WASTE
#define PT_SPLIT_WASTE 4.0f
#endif

// A reference's part of its triangle as bounds on the weights (u, v) of the part's points in P = o + u e1 + v e2, where o is the
// triangle's vertex whose angle is closest to a right one and e1, e2 the edges leaving it (large split triangles are mostly
// halves of rectangles, cut along their legs: with the legs as axes the bounds of a cell's part are the cell itself).
// Seven bits per bound, rounded outwards: (origin vertex + 1) << 28 | u0 | v0 << 7 | u1 << 14 | v1 << 21; never 0, which
// marks a triangle entered whole.  The weights of a point do not change under the affine object-to-world map, so a refit
// can bound the part again after any transform edit: the parts of a triangle cover it in (u, v) wherever its vertices move.
__device__ uint32_t part_code(v3 a, v3 b, v3 c, const float (*poly)[3], int np)
{
    const v3 p[3] = {a, b, c};
    int org = 0;
    float best = __builtin_inff();
    for (int i = 0; i < 3; i++) {
        const v3 ea = p[(i + 1) % 3] - p[i], eb = p[(i + 2) % 3] - p[i];
        const float d = dot(ea, eb), score = d * d / (dot(ea, ea) * dot(eb, eb));     // cos^2 of the angle at vertex i
        if (score < best) { best = score; org = i; }
    }
    const v3 o = p[org], e1 = p[(org + 1) % 3] - o, e2 = p[(org + 2) % 3] - o, n = cross(e1, e2);
    const float nn = dot(n, n);
    float u0 = 1.0f, u1 = 0.0f, v0 = 1.0f, v1 = 0.0f;
    for (int i = 0; i < np; i++) {
        const v3 d = V3(poly[i][0], poly[i][1], poly[i][2]) - o;
        const float u = dot(cross(d, e2), n) / nn, v = dot(cross(e1, d), n) / nn;
        u0 = fminf(u0, u); u1 = fmaxf(u1, u); v0 = fminf(v0, v); v1 = fmaxf(v1, v);
    }
    const uint32_t whole = ((uint32_t)(org + 1) << 28) | (127u << 14) | (127u << 21);
    if (!(u0 <= u1) || !(v0 <= v1)) return whole;                  // a degenerate triangle (NaN weights): the whole range
    const float eps = 1e-5f;                                       // rounding of the weights themselves
    const int q0 = (int)floorf((u0 - eps) * 127.0f), q1 = (int)ceilf((u1 + eps) * 127.0f);
    const int r0 = (int)floorf((v0 - eps) * 127.0f), r1 = (int)ceilf((v1 + eps) * 127.0f);
    return ((uint32_t)(org + 1) << 28) | (uint32_t)min(max(q0, 0), 127) | ((uint32_t)min(max(r0, 0), 127) << 7) |
           ((uint32_t)min(max(q1, 0), 127) << 14) | ((uint32_t)min(max(r1, 0), 127) << 21);
}

// bounds of the part of triangle (a, b, c) inside the box [clo, chi]; false if there is none.  code (may be null): part_code of the part
__device__ bool clip_bounds(v3 a, v3 b, v3 c, const float* clo, const float* chi, float* olo, float* ohi, uint32_t* code = nullptr)
{
    float px[2][10][3];
    int np = 3, cur = 0;
    px[0][0][0] = a.x; px[0][0][1] = a.y; px[0][0][2] = a.z;
    px[0][1][0] = b.x; px[0][1][1] = b.y; px[0][1][2] = b.z;
    px[0][2][0] = c.x; px[0][2][1] = c.y; px[0][2][2] = c.z;
    for (int plane = 0; plane < 6 && np > 0; plane++) {
        const int ax = plane >> 1;
        const bool upper = plane & 1;
        const float lim = upper ? chi[ax] : clo[ax];
        int nq = 0;
        float (*in)[3] = px[cur];
        float (*out)[3] = px[cur ^ 1];
        for (int i = 0; i < np; i++) {
            const float* p = in[i];
            const float* q = in[(i + 1) % np];
            const bool pin = upper ? p[ax] <= lim : p[ax] >= lim;
            const bool qin = upper ? q[ax] <= lim : q[ax] >= lim;
            if (pin) { out[nq][0] = p[0]; out[nq][1] = p[1]; out[nq][2] = p[2]; nq++; }
            if (pin != qin) {
                float t = (lim - p[ax]) / (q[ax] - p[ax]);
                for (int k = 0; k < 3; k++) out[nq][k] = k == ax ? lim : p[k] + t * (q[k] - p[k]);
                nq++;
            }
        }
        np = nq;
        cur ^= 1;
    }
    if (np == 0) return false;
    for (int k = 0; k < 3; k++) { olo[k] = 3.4028235e38f; ohi[k] = -3.4028235e38f; }
    for (int i = 0; i < np; i++)
        for (int k = 0; k < 3; k++) {
            olo[k] = fminf(olo[k], px[cur][i][k]);
            ohi[k] = fmaxf(ohi[k], px[cur][i][k]);
        }
    // the clipped vertices carry rounding; the part can neither leave the cell nor the triangle's own box
    for (int k = 0; k < 3; k++) { olo[k] = fmaxf(olo[k], clo[k]); ohi[k] = fminf(ohi[k], chi[k]); }
    if (code) *code = part_code(a, b, c, px[cur], np);
    return true;
}

// Pass 1, one thread per triangle: mode[g] = 0 for a triangle that stays whole (counts[g] = 1), 0xFF for a candidate, which
// is also appended to the candidate list.
__global__ void k_split_classify(uint32_t n, const TriRec* __restrict__ tris, const float4* __restrict__ tlo, const float4* __restrict__ thi,
                                 const int* __restrict__ bounds, uint32_t* __restrict__ counts, unsigned char* __restrict__ mode,
                                 uint32_t* __restrict__ cand, uint32_t* ncand)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    counts[g] = 1u;
    mode[g] = 0;
    const float4 lo = tlo[g], hi = thi[g];
    if (!(lo.x <= hi.x)) return;                                  // the empty box of a triangle that cannot be hit
    const TriRec t = tris[g];
    const float e[3] = {hi.x - lo.x, hi.y - lo.y, hi.z - lo.z};
    const float sa = 2.0f * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]);
    const float sx = ord2f(bounds[3]) - ord2f(bounds[0]), sy = ord2f(bounds[4]) - ord2f(bounds[1]), sz = ord2f(bounds[5]) - ord2f(bounds[2]);
    const float scene_sa = 2.0f * (sx * sy + sy * sz + sz * sx);
    const v3 a = V3(t.v0x, t.v0y, t.v0z), b = V3(t.v1x, t.v1y, t.v1z), c = V3(t.v2x, t.v2y, t.v2z);
    const v3 cr = cross(b - a, c - a);
    const float two_sided = sqrtf(dot(cr, cr));                    // 2 * (|cross| / 2)
    if (!(sa > scene_sa * (1.0f / PT_SPLIT_AREA_DIV)) || !(sa > PT_SPLIT_WASTE * two_sided)) return;
    mode[g] = 0xFF;
    cand[atomicAdd(ncand, 1u)] = g;
}

// One wave per candidate; the lanes share out the cells of the grid over the triangle's box (up to 8 x 8 x 8).
// WRITE = false: decides the grid (an eighth of the longest side; a quarter if that gives more than PT_SPLIT_MAX pieces;
// else the triangle stays whole) and leaves it in mode[g] (8, 4 or 0) with the number of pieces in counts[g].
// WRITE = true: writes the pieces of the grid in mode[g] at offsets[g], in cell order.
template <bool WRITE>
__global__ void __launch_bounds__(64) k_split_cells(uint32_t ncand, const uint32_t* __restrict__ cand, const TriRec* __restrict__ tris,
                                                    const float4* __restrict__ tlo, const float4* __restrict__ thi, uint32_t* __restrict__ counts,
                                                    unsigned char* __restrict__ mode, const uint32_t* __restrict__ offsets,
                                                    TriRec* __restrict__ out_tris, float4* __restrict__ out_lo, float4* __restrict__ out_hi)
{
    if (blockIdx.x >= ncand) return;
    const uint32_t g = cand[blockIdx.x], lane = threadIdx.x;
    const TriRec t = tris[g];
    const float4 lo4 = tlo[g], hi4 = thi[g];
    const float lo[3] = {lo4.x, lo4.y, lo4.z}, hi[3] = {hi4.x, hi4.y, hi4.z};
    const float e[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    const v3 a = V3(t.v0x, t.v0y, t.v0z), b = V3(t.v1x, t.v1y, t.v1z), c = V3(t.v2x, t.v2y, t.v2z);
    const float emax = fmaxf(fmaxf(e[0], e[1]), e[2]);
    const uint32_t w = WRITE ? offsets[g] : 0u;
    // visits the cells of a (emax / div) grid over the box that the triangle passes through; returns how many (wave-uniform)
    auto cells = [&](int div) {
        const float cell = emax / (float)div;
        int nc[3];
        for (int k = 0; k < 3; k++) {
            int m = (int)ceilf(e[k] / cell);
            nc[k] = m < 1 ? 1 : (m > div ? div : m);
        }
        const int total = nc[0] * nc[1] * nc[2];
        uint32_t found = 0;
        for (int c0 = 0; c0 < total; c0 += 64) {
            const int ci = c0 + (int)lane;
            bool hit = false;
            float olo[3], ohi[3];
            uint32_t code = 0;
            if (ci < total) {
                const int idx[3] = {ci % nc[0], (ci / nc[0]) % nc[1], ci / (nc[0] * nc[1])};
                float clo[3], chi[3];
                for (int k = 0; k < 3; k++) {
                    clo[k] = lo[k] + (float)idx[k] * cell;
                    chi[k] = idx[k] + 1 == nc[k] ? hi[k] : lo[k] + (float)(idx[k] + 1) * cell;
                }
                hit = clip_bounds(a, b, c, clo, chi, olo, ohi, WRITE ? &code : nullptr);
            }
            const unsigned long long m = __ballot(hit);
            if (WRITE && hit) {
                const uint32_t pos = w + found + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                TriRec part = t;
                part.pad = code;
                out_tris[pos] = part;
                out_lo[pos] = make_float4(olo[0], olo[1], olo[2], 0.0f);
                out_hi[pos] = make_float4(ohi[0], ohi[1], ohi[2], 0.0f);
            }
            found += (uint32_t)__popcll(m);
        }
        return found;
    };
    if (WRITE) {
        if (mode[g]) cells((int)mode[g]);        // 0: the candidate stays whole and k_split_copy_whole has written it
        return;
    }
    uint32_t pieces = 1u;
    unsigned char md = 0;
    for (int div = 8; div >= 4; div >>= 1) {
        const uint32_t found = cells(div);
        if (found == 0) break;                                     // rounding left nothing: keep the triangle whole
        if (found <= PT_SPLIT_MAX) { pieces = found; md = (unsigned char)div; break; }
    }
    if (lane == 0) { counts[g] = pieces; mode[g] = md; }
}

// the triangles that stay whole, at their place among the references
__global__ void k_split_copy_whole(uint32_t n, const TriRec* __restrict__ tris, const float4* __restrict__ tlo, const float4* __restrict__ thi,
                                   const unsigned char* __restrict__ mode, const uint32_t* __restrict__ offsets,
                                   TriRec* __restrict__ out_tris, float4* __restrict__ out_lo, float4* __restrict__ out_hi)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n || mode[g] != 0) return;
    const uint32_t w = offsets[g];
    out_tris[w] = tris[g];
    out_lo[w] = tlo[g];
    out_hi[w] = thi[g];
}

__device__ __forceinline__ unsigned long long spread21(uint32_t v)
{
    unsigned long long x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void k_morton(uint32_t n, const float4* __restrict__ tlo, const float4* __restrict__ thi, const int* __restrict__ bounds,
                         unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    float blx = ord2f(bounds[0]), bly = ord2f(bounds[1]), blz = ord2f(bounds[2]);
    float bhx = ord2f(bounds[3]), bhy = ord2f(bounds[4]), bhz = ord2f(bounds[5]);
    float ext = fmaxf(fmaxf(bhx - blx, bhy - bly), bhz - blz);
    float scale = ext > 0.0f ? 2097152.0f / ext : 0.0f;
    float4 lo = tlo[g], hi = thi[g];
    float cx = ((lo.x + hi.x) * 0.5f - blx) * scale;
    float cy = ((lo.y + hi.y) * 0.5f - bly) * scale;
    float cz = ((lo.z + hi.z) * 0.5f - blz) * scale;
    uint32_t ix = (uint32_t)fminf(fmaxf(cx, 0.0f), 2097151.0f);
    uint32_t iy = (uint32_t)fminf(fmaxf(cy, 0.0f), 2097151.0f);
    uint32_t iz = (uint32_t)fminf(fmaxf(cz, 0.0f), 2097151.0f);
    keys[g] = (spread21(ix) << 2) | (spread21(iy) << 1) | spread21(iz);
    vals[g] = g;
}

// common-prefix length of sorted keys i and j (j may be out of range)
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// internal node i of the binary radix tree; node ids: internal 0..n-2, leaf j -> (n-1)+j
__global__ void k_karras(int n, const unsigned long long* __restrict__ keys, uint2* __restrict__ children, uint32_t* __restrict__ parent)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    uint32_t left = (lo == gamma) ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;
    uint32_t right = (hi == gamma + 1) ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    children[i] = make_uint2(left, right);
    parent[left] = (uint32_t)i;
    parent[right] = (uint32_t)i;
    if (i == 0) parent[0] = NONE;
}

// b2lo.w carries the subtree's triangle count (as bits)
// Hand-off between threads of one launch (bottom-up passes over the radix tree: whoever arrives second at a node reads what
// the first one wrote).  The L2 of each XCD is not coherent with the others', and a release fence per node writes a whole L2
// back (two such fences per node made these passes take 19 ms each for 3 M triangles).  Instead the handed-over words are
// stored and loaded at agent scope (write-through / read-through), the writer waits for its stores before it announces its
// arrival with an agent-scope atomic, and the reader's loads follow that atomic in program order.
__device__ __forceinline__ void st_agent(float4* p, float4 v)
{
    union { float4 f; unsigned long long u[2]; } x;
    x.f = v;
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    __hip_atomic_store(q, x.u[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, x.u[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 ld_agent(const float4* p)
{
    union { float4 f; unsigned long long u[2]; } x;
    unsigned long long* q = reinterpret_cast<unsigned long long*>(const_cast<float4*>(p));
    x.u[0] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x.u[1] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return x.f;
}
// true when this thread is the second to arrive at the node (and goes on with it); its own stores have landed before
__device__ __forceinline__ bool arrive_second(uint32_t* flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t old = __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::: "memory");
    return old != 0;
}

__global__ void k_refit(int n, const uint32_t* __restrict__ vals, const float4* __restrict__ tlo, const float4* __restrict__ thi,
                        const uint2* __restrict__ children, const uint32_t* __restrict__ parent, uint32_t* flags,
                        float4* b2lo, float4* b2hi)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t t = vals[j];
    float4 lo = tlo[t], hi = thi[t];
    uint32_t node = (uint32_t)(n - 1 + j);
    lo.w = __uint_as_float(1u);
    st_agent(&b2lo[node], lo);
    st_agent(&b2hi[node], hi);
    if (n == 1) return;
    uint32_t cur = parent[node];
    while (cur != NONE) {
        if (!arrive_second(&flags[cur])) return;     // the sibling subtree is not finished: its thread continues
        uint2 ch = children[cur];
        float4 llo = ld_agent(&b2lo[ch.x]), lhi = ld_agent(&b2hi[ch.x]);
        float4 rlo = ld_agent(&b2lo[ch.y]), rhi = ld_agent(&b2hi[ch.y]);
        float4 nlo = make_float4(fminf(llo.x, rlo.x), fminf(llo.y, rlo.y), fminf(llo.z, rlo.z),
                                 __uint_as_float(__float_as_uint(llo.w) + __float_as_uint(rlo.w)));
        float4 nhi = make_float4(fmaxf(lhi.x, rhi.x), fmaxf(lhi.y, rhi.y), fmaxf(lhi.z, rhi.z), 0.0f);
        st_agent(&b2lo[cur], nlo);
        st_agent(&b2hi[cur], nhi);
        cur = parent[cur];
    }
}


// ---- treelet restructuring (after Karras and Aila, "Fast parallel construction of high-quality bounding volume
// hierarchies", 2013) -----------------------------------------------------------------------------------------------
// Bottom-up over a binary tree: at every node with at least eight triangles below it, the node and the six descendants with
// the largest boxes are taken out (a treelet of seven internal nodes over eight subtrees) and put back in the topology that
// minimises the summed box area of the seven, found by a dynamic programme over the 255 subsets of the eight subtrees.
// Three passes of eight against two of seven: 1 to 2 % more rays per second for a quarter more build time; further passes
// change nothing measurable.
// One lane per leaf climbs the tree; the treelets the lanes of a wave arrive at are restructured one after the other by the
// whole wave (subsets and their partitions dealt out over the lanes).  Data is handed between waves like in the other
// bottom-up passes (st_agent / ld_agent).
#ifndef PT_TREELET_LEAVES
#define PT_TREELET_LEAVES 8      // 7 or 8
#endif
#ifndef PT_TREELET_PASSES
#define PT_TREELET_PASSES 3
#endif
__device__ __forceinline__ float dp_area(float4 lo, float4 hi);

__device__ __forceinline__ uint2 ld_agent2(const uint2* p)
{
    union { uint2 v; unsigned long long u; } x;
    x.u = __hip_atomic_load(reinterpret_cast<unsigned long long*>(const_cast<uint2*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return x.v;
}
__device__ __forceinline__ void st_agent2(uint2* p, uint2 v)
{
    union { uint2 v; unsigned long long u; } x;
    x.v = v;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), x.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the subsets of PT_TREELET_LEAVES elements in order of their size: v[first[k] .. first[k + 1]) are those with k elements
struct SubsetTable { unsigned char v[1 << PT_TREELET_LEAVES]; unsigned short first[PT_TREELET_LEAVES + 2]; };
constexpr SubsetTable make_subset_table()
{
    SubsetTable t{};
    int at = 0;
    for (int k = 0; k <= PT_TREELET_LEAVES; k++) {
        t.first[k] = (unsigned short)at;
        for (int sset = 0; sset < (1 << PT_TREELET_LEAVES); sset++) {
            int pc = 0;
            for (int b = 0; b < PT_TREELET_LEAVES; b++) pc += (sset >> b) & 1;
            if (pc == k) t.v[at++] = (unsigned char)sset;
        }
    }
    t.first[PT_TREELET_LEAVES + 1] = (unsigned short)at;
    return t;
}
__constant__ SubsetTable c_subsets = make_subset_table();
// log2 of the lanes that share the partitions of one subset of k elements: as many as fit the wave and have work
constexpr int treelet_group_log2(int k)
{
    int nsub = 1;
    for (int i = 0; i < k; i++) nsub = nsub * (PT_TREELET_LEAVES - i) / (i + 1);
    const int nparts = (1 << (k - 1)) - 1;
    int g = 0;
    while ((nsub << (g + 1)) <= 64 && (1 << (g + 1)) <= nparts + 1) g++;
    return g;
}

struct TreeletLds {
    float area[1 << PT_TREELET_LEAVES], cost[1 << PT_TREELET_LEAVES];
    unsigned char part[1 << PT_TREELET_LEAVES];
    float lo[PT_TREELET_LEAVES][3], hi[PT_TREELET_LEAVES][3];
    uint32_t count[PT_TREELET_LEAVES], leaf[PT_TREELET_LEAVES], inner[PT_TREELET_LEAVES - 1];
    unsigned char node_set[PT_TREELET_LEAVES - 1], node_kid[PT_TREELET_LEAVES - 1][2];
    float cur_cost;
    int ok;
};

// bits of i spread over the set bits of mask, lowest first
__device__ __forceinline__ int deposit_bits(int i, int mask)
{
    int r = 0;
#pragma unroll
    for (int b = 0; b < PT_TREELET_LEAVES - 1; b++) {
        int low = mask & -mask;
        if (i & (1 << b)) r |= low;
        mask ^= low;
    }
    return r;
}

// all 64 lanes of the wave (one workgroup) restructure the treelet rooted at R
__device__ void treelet_optimize(uint32_t R, int lane, int n, uint2* children, uint32_t* parent, float4* b2lo, float4* b2hi, TreeletLds& S)
{
    constexpr int NL = PT_TREELET_LEAVES;
    constexpr int FULL = (1 << NL) - 1;
    // the treelet: expand the subtree with the largest box until there are eight (one lane; the loads of a step go out together)
    if (lane == 0) {
        uint2 kids[NL];                    // children of subtree k when it is an internal node
        float larea[NL];
        auto load_leaf = [&](int k, uint32_t id) {
            S.leaf[k] = id;
            float4 lo = ld_agent(&b2lo[id]), hi = ld_agent(&b2hi[id]);
            kids[k] = id < (uint32_t)(n - 1) ? ld_agent2(&children[id]) : make_uint2(NONE, NONE);
            S.lo[k][0] = lo.x; S.lo[k][1] = lo.y; S.lo[k][2] = lo.z;
            S.hi[k][0] = hi.x; S.hi[k][1] = hi.y; S.hi[k][2] = hi.z;
            S.count[k] = __float_as_uint(lo.w);
            larea[k] = dp_area(lo, hi);
        };
        S.inner[0] = R;
        int ni = 1, nl = 2;
        float cur_cost = dp_area(ld_agent(&b2lo[R]), ld_agent(&b2hi[R]));
        uint2 ch = ld_agent2(&children[R]);
        load_leaf(0, ch.x);
        load_leaf(1, ch.y);
        while (nl < NL) {
            int best = -1;
            float ba = -1.0f;
#pragma unroll
            for (int k = 0; k < NL; k++)
                if (k < nl && kids[k].x != NONE && larea[k] > ba) { ba = larea[k]; best = k; }
            if (best < 0) break;                    // fewer subtrees than a treelet takes: leave it
            uint2 c2 = make_uint2(NONE, NONE);
#pragma unroll
            for (int k = 0; k < NL; k++)
                if (k == best) { c2 = kids[k]; S.inner[ni] = S.leaf[k]; }
            ni++;
            cur_cost += ba;
#pragma unroll
            for (int k = 0; k < NL; k++)
                if (k == best) load_leaf(k, c2.x);
#pragma unroll
            for (int k = 0; k < NL; k++)
                if (k == nl) load_leaf(k, c2.y);
            nl++;
        }
        S.cur_cost = cur_cost;
        S.ok = nl == NL;
    }
    __syncthreads();
    if (!S.ok) { __syncthreads(); return; }
    // box area of every subset: lanes take them in turn
#pragma unroll
    for (int h = 0; h < (1 << NL) / 64; h++) {
        const int sset = lane + 64 * h;
        float lo[3] = {3.4028235e38f, 3.4028235e38f, 3.4028235e38f}, hi[3] = {-3.4028235e38f, -3.4028235e38f, -3.4028235e38f};
#pragma unroll
        for (int k = 0; k < NL; k++)
            if ((sset & (1 << k)) && S.lo[k][0] <= S.hi[k][0]) {       // not the empty box of triangles that cannot be hit
#pragma unroll
                for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], S.lo[k][a]); hi[a] = fmaxf(hi[a], S.hi[k][a]); }
            }
        S.area[sset] = dp_area(make_float4(lo[0], lo[1], lo[2], 0.0f), make_float4(hi[0], hi[1], hi[2], 0.0f));
        S.cost[sset] = 0.0f;               // single subtrees cost nothing here; the others are overwritten below
    }
    __syncthreads();
    // cheapest topology of every subset, by size: cost = own box + the two parts'.  2^gl lanes share the partitions of a subset
#pragma unroll
    for (int k = 2; k <= NL; k++) {
        const int gl = treelet_group_log2(k);
        const int q = lane & ((1 << gl) - 1);
        const int nsub = (int)c_subsets.first[k + 1] - (int)c_subsets.first[k];
        const int nparts = (1 << (k - 1)) - 1;        // the parts that hold the lowest element, the whole set left out
        for (int base = 0; base < nsub; base += 64 >> gl) {
            const int idx = base + (lane >> gl);
            const bool active = idx < nsub;
            const int sset = active ? (int)c_subsets.v[c_subsets.first[k] + idx] : 3;
            const int low = sset & -sset, rest = sset ^ low;
            float best = __builtin_inff();
            int bp = FULL;
            for (int i = q; i < nparts; i += 1 << gl) {
                const int pp = low | deposit_bits(i, rest);
                const float c = S.cost[pp] + S.cost[sset ^ pp];
                if (c < best || (c == best && pp < bp)) { best = c; bp = pp; }
            }
            for (int off = (1 << gl) >> 1; off > 0; off >>= 1) {
                const float ob = __shfl_xor(best, off);
                const int op = __shfl_xor(bp, off);
                if (ob < best || (ob == best && op < bp)) { best = ob; bp = op; }
            }
            if (active && q == 0) { S.cost[sset] = S.area[sset] + best; S.part[sset] = (unsigned char)bp; }
        }
        __syncthreads();
    }
    if (!(S.cost[FULL] < S.cur_cost * 0.9999f)) { __syncthreads(); return; }
    // the internal nodes in that topology: node t covers node_set[t]; a child is subtree (0x80 | k) or internal node t'
    if (lane == 0) {
        int sp = 1, next = 1;
        S.node_set[0] = (unsigned char)FULL;
        // the stack of (subset, node) pairs to do: a byte per entry in two 64-bit words, so that it stays in registers
        unsigned long long stack_sets = FULL, stack_ts = 0;
        while (sp) {
            sp--;
            const int sset = (int)((stack_sets >> (8 * sp)) & 0xFF), t = (int)((stack_ts >> (8 * sp)) & 0xFF);
            const int pp = S.part[sset];
            const int sub[2] = {pp, sset ^ pp};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if ((sub[h] & (sub[h] - 1)) == 0) {
                    S.node_kid[t][h] = (unsigned char)(0x80 | (__ffs(sub[h]) - 1));
                } else {
                    S.node_kid[t][h] = (unsigned char)next;
                    S.node_set[next] = (unsigned char)sub[h];
                    stack_sets = (stack_sets & ~(0xFFull << (8 * sp))) | ((unsigned long long)sub[h] << (8 * sp));
                    stack_ts = (stack_ts & ~(0xFFull << (8 * sp))) | ((unsigned long long)next << (8 * sp));
                    sp++;
                    next++;
                }
            }
        }
    }
    __syncthreads();
    if (lane < NL - 1) {
        const int sset = S.node_set[lane];
        const uint32_t X = S.inner[lane];
        uint32_t kid[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int ref = S.node_kid[lane][h];
            kid[h] = (ref & 0x80) ? S.leaf[ref & 7] : S.inner[ref];
            __hip_atomic_store(&parent[kid[h]], X, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        st_agent2(&children[X], make_uint2(kid[0], kid[1]));
        float lo[3] = {3.4028235e38f, 3.4028235e38f, 3.4028235e38f}, hi[3] = {-3.4028235e38f, -3.4028235e38f, -3.4028235e38f};
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < NL; k++)
            if (sset & (1 << k)) {
                cnt += S.count[k];
                if (!(S.lo[k][0] <= S.hi[k][0])) continue;
#pragma unroll
                for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], S.lo[k][a]); hi[a] = fmaxf(hi[a], S.hi[k][a]); }
            }
        st_agent(&b2lo[X], make_float4(lo[0], lo[1], lo[2], __uint_as_float(cnt)));
        st_agent(&b2hi[X], make_float4(hi[0], hi[1], hi[2], 0.0f));
    }
    __syncthreads();
}

// one pass (flags zeroed), workgroups of one wave: every lane climbs from its leaf; whoever arrives second at a node owns its
// treelet, and the wave restructures the treelets its lanes own one after the other before those lanes climb on
__global__ __launch_bounds__(64) void k_treelet_up(int n, uint2* children, uint32_t* parent, uint32_t* flags, float4* b2lo, float4* b2hi)
{
    __shared__ TreeletLds S;
    const int lane = threadIdx.x;
    const int j = blockIdx.x * 64 + lane;
    uint32_t cur = (j < n && n > 1) ? parent[(uint32_t)(n - 1 + j)] : NONE;
    for (;;) {
        if (cur != NONE && !arrive_second(&flags[cur])) cur = NONE;
        unsigned long long m = __ballot(cur != NONE);
        if (!m) break;
        while (m) {
            const int src = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t R = (uint32_t)__builtin_amdgcn_readlane((int)cur, src);
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(ld_agent(&b2lo[R]).w)) >= (uint32_t)PT_TREELET_LEAVES)
                treelet_optimize(R, lane, n, children, parent, b2lo, b2hi, S);
        }
        if (cur != NONE) cur = __hip_atomic_load(&parent[cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// parent pointers of a binary tree given by its children (the clustered tree is built without them)
__global__ void k_parents(int n, uint32_t root, const uint2* __restrict__ children, uint32_t* __restrict__ parent)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    uint2 ch = children[i];
    parent[ch.x] = (uint32_t)i;
    parent[ch.y] = (uint32_t)i;
    if ((uint32_t)i == root) parent[i] = NONE;
}

// ---- agglomerative build (parallel locally-ordered clustering) ------------------------------------
#ifndef PT_PLOC_RADIUS
#define PT_PLOC_RADIUS 16
#endif
static constexpr int PLOC_RADIUS = PT_PLOC_RADIUS;
static constexpr int PLOC_BLOCK = 256;

__global__ void k_ploc_init(int n, const uint32_t* __restrict__ vals, const float4* __restrict__ tlo, const float4* __restrict__ thi,
                            uint32_t* __restrict__ cid, float4* __restrict__ b2lo, float4* __restrict__ b2hi)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t t = vals[j];
    float4 lo = tlo[t], hi = thi[t];
    lo.w = __uint_as_float(1u);
    uint32_t node = (uint32_t)(n - 1 + j);
    b2lo[node] = lo;
    b2hi[node] = hi;
    cid[j] = node;
}

// nearest neighbour (smallest surface area of the merged box) within PLOC_RADIUS places
__global__ void __launch_bounds__(PLOC_BLOCK) k_ploc_nearest(int nc, const uint32_t* __restrict__ cid, const float4* __restrict__ b2lo,
                                                             const float4* __restrict__ b2hi, uint32_t* __restrict__ nearest)
{
    __shared__ float slo[PLOC_BLOCK + 2 * PLOC_RADIUS][3];
    __shared__ float shi[PLOC_BLOCK + 2 * PLOC_RADIUS][3];
    int base = blockIdx.x * PLOC_BLOCK - PLOC_RADIUS;
    for (int k = threadIdx.x; k < PLOC_BLOCK + 2 * PLOC_RADIUS; k += PLOC_BLOCK) {
        int g = base + k;
        if (g >= 0 && g < nc) {
            uint32_t id = cid[g];
            float4 lo = b2lo[id], hi = b2hi[id];
            slo[k][0] = lo.x; slo[k][1] = lo.y; slo[k][2] = lo.z;
            shi[k][0] = hi.x; shi[k][1] = hi.y; shi[k][2] = hi.z;
        }
    }
    __syncthreads();
    int i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
    if (i >= nc) return;
    int li = threadIdx.x + PLOC_RADIUS;
    float lx = slo[li][0], ly = slo[li][1], lz = slo[li][2], hx = shi[li][0], hy = shi[li][1], hz = shi[li][2];
    float best = __builtin_inff();
    int bestj = -1;
    for (int dlt = -PLOC_RADIUS; dlt <= PLOC_RADIUS; dlt++) {
        int j = i + dlt;
        if (dlt == 0 || j < 0 || j >= nc) continue;
        int lj = li + dlt;
        float ex = fmaxf(hx, shi[lj][0]) - fminf(lx, slo[lj][0]);
        float ey = fmaxf(hy, shi[lj][1]) - fminf(ly, slo[lj][1]);
        float ez = fmaxf(hz, shi[lj][2]) - fminf(lz, slo[lj][2]);
        float a = ex * ey + ey * ez + ez * ex;
        // equal areas (duplicated geometry): prefer the partner with the smallest index xor.  The pair that is smallest in
        // (area, xor) over the whole array then chooses each other, so every round merges, and a run of identical boxes
        // pairs up (0,1)(2,3)... and halves per round instead of merging one pair at its end
        if (bestj < 0 || a < best || (a == best && (i ^ j) < (i ^ bestj))) { best = a; bestj = j; }
    }
    nearest[i] = (uint32_t)bestj;
}

// flags[i]: low word = 1 when cluster i survives (unmerged, or the lower index of a merging pair),
//           high word = 1 when cluster i is the lower index of a merging pair (creates a node)
__global__ void k_ploc_flags(int nc, const uint32_t* __restrict__ nearest, unsigned long long* __restrict__ flags)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    uint32_t j = nearest[i];
    bool mutual = j != 0xFFFFFFFFu && nearest[j] == (uint32_t)i;
    unsigned long long f = 1ull;
    if (mutual) f = (uint32_t)i < j ? (1ull | (1ull << 32)) : 0ull;
    flags[i] = f;
}

__global__ void k_ploc_merge(int nc, uint32_t next_id, const uint32_t* __restrict__ nearest, const unsigned long long* __restrict__ flags,
                             const unsigned long long* __restrict__ scan, const uint32_t* __restrict__ cid_in, uint32_t* __restrict__ cid_out,
                             uint2* __restrict__ children, float4* b2lo, float4* b2hi, uint32_t* __restrict__ totals)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    unsigned long long f = flags[i], sc = scan[i];
    if (i == nc - 1) {
        unsigned long long tot = sc + f;
        totals[0] = (uint32_t)tot;          // clusters after this round
        totals[1] = (uint32_t)(tot >> 32);  // nodes created
    }
    if (!(f & 1ull)) return;
    uint32_t pos = (uint32_t)sc;
    if (f >> 32) {
        uint32_t a = cid_in[i], b = cid_in[nearest[i]];
        uint32_t id = next_id + (uint32_t)(sc >> 32);
        float4 alo = b2lo[a], ahi = b2hi[a], blo = b2lo[b], bhi = b2hi[b];
        b2lo[id] = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z),
                               __uint_as_float(__float_as_uint(alo.w) + __float_as_uint(blo.w)));
        b2hi[id] = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0.0f);
        children[id] = make_uint2(a, b);
        cid_out[pos] = id;
    } else {
        cid_out[pos] = cid_in[i];
    }
}

__device__ __forceinline__ float box_area(float4 lo, float4 hi)
{
    float ex = hi.x - lo.x, ey = hi.y - lo.y, ez = hi.z - lo.z;
    return ex * ey + ey * ez + ez * ex;
}

struct CollapseItem { uint32_t node2, slot, tstart, pad; };

// smallest biased exponent e with extent <= 255 * 2^(e - 127)
__device__ __forceinline__ uint32_t grid_exponent(float extent)
{
    float f = extent * (1.0f / 255.0f);
    if (!(f > 1e-30f)) return 27u;                 // flat or empty axis: any tiny positive step works
    uint32_t bits = __float_as_uint(f);
    uint32_t e = (bits >> 23) & 255u;
    if (bits & 0x007fffffu) e++;                   // not an exact power of two: round the step up
    return e > 253u ? 253u : e;                    // 255 steps of 2^126 span every finite extent; the inverse step stays normal
}

// the grid of a node: origin = lower corner of the padded union of its child boxes, one power-of-two step per axis
struct NodeGrid { float lo[3]; uint32_t e[3]; float inv[3]; };

__device__ __forceinline__ NodeGrid node_grid(const float4* clo, const float4* chi, int nc, float pad)
{
    NodeGrid g;
    float hi3[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    g.lo[0] = g.lo[1] = g.lo[2] = __builtin_inff();
    for (int k = 0; k < nc; k++) {
        g.lo[0] = fminf(g.lo[0], clo[k].x - pad); g.lo[1] = fminf(g.lo[1], clo[k].y - pad); g.lo[2] = fminf(g.lo[2], clo[k].z - pad);
        hi3[0] = fmaxf(hi3[0], chi[k].x + pad); hi3[1] = fmaxf(hi3[1], chi[k].y + pad); hi3[2] = fmaxf(hi3[2], chi[k].z + pad);
    }
    for (int a = 0; a < 3; a++) {
        g.e[a] = grid_exponent(hi3[a] - g.lo[a]);
        g.inv[a] = __uint_as_float((254u - g.e[a]) << 23);       // 2^-(e - 127), exact
    }
    return g;
}

// a child box, padded, on the node's grid: lower planes rounded down, upper planes up
__device__ __forceinline__ void quantise_child(const NodeGrid& g, float4 lo, float4 hi, float pad, uint32_t* qlo, uint32_t* qhi)
{
    float l3[3] = {lo.x - pad, lo.y - pad, lo.z - pad}, h3[3] = {hi.x + pad, hi.y + pad, hi.z + pad};
    for (int a = 0; a < 3; a++) {
        // in double: the difference of two large floats must not overflow or round the wrong way
        double ql = floor(((double)l3[a] - (double)g.lo[a]) * (double)g.inv[a]);
        double qh = ceil(((double)h3[a] - (double)g.lo[a]) * (double)g.inv[a]);
        qlo[a] = (uint32_t)fmin(fmax(ql, 0.0), 255.0);
        qhi[a] = (uint32_t)fmin(fmax(qh, 0.0), 255.0);
    }
}

// plane p of child k is byte (k & 3) of word p * 2 + (k >> 2)
__device__ __forceinline__ void store_child_planes(Node8& nd, int k, const uint32_t* qlo, const uint32_t* qhi)
{
    uint32_t sh = (uint32_t)(k & 3) * 8u;
    int wo = k >> 2;
    for (int a = 0; a < 3; a++) {
        nd.q[a * 2 + wo] |= qlo[a] << sh;
        nd.q[(3 + a) * 2 + wo] |= qhi[a] << sh;
    }
}

// ---- which subtrees of the binary tree become the children of an 8-wide node ---------------------------------------
// Surface-area-optimal choice by a dynamic programme over the binary tree (after Ylitie, Karras, Laine: "Efficient
// incoherent ray traversal on GPUs through compressed wide BVHs", 2017, section 3.2).  c(n, i) = cheapest way to present
// subtree n to a parent as at most i children:
//   c(n, 1) = min(leaf(n), node(n))         leaf(n) = A(n) (C_leaf + P(n) C_tri) if n holds at most PT_LEAF_MAX triangles
//                                            node(n) = A(n) C_node + split(n, 8)
//   c(n, i) = min(split(n, i), c(n, i - 1))  split(n, j) = min over k of c(left, k) + c(right, j - k)
// with A the box area relative to the root's.  The cost ratios are measured, not derived: by instruction counts a triangle
// test is 0.8 of a node step, but throughput on three scenes is flat between C_tri = 0.2 and 0.3 with C_leaf = 0.3 to 0.5 and
// 1 to 3 % lower at 0.8 (the surface-area model overrates how often a leaf's triangles are reached behind a closer hit).
// The greedy rule it replaces (open the child with the largest box until eight are open, every subtree of at most four
// triangles a leaf) left 39 % of all nodes with two children: a subtree of 5 to 8 triangles beside large siblings.
#ifndef PT_COST_NODE
#define PT_COST_NODE 1.0f
#define PT_COST_TRI 0.25f
#define PT_COST_LEAF 0.4f
#endif
// per internal node of the binary tree: c(n, 1..7) and the choices behind them.  choice bits: 3 * (j - 2) + {0, 1, 2} =
// k of split(n, j), j = 2..8; bit 21 = a leaf is cheaper than a node; bit 22 + (i - 2) = c(n, i) takes c(n, i - 1), i = 2..7
struct alignas(16) CollapseDp { float c[7]; uint32_t choice; };

__device__ __forceinline__ float dp_area(float4 lo, float4 hi)
{
    float ex = hi.x - lo.x, ey = hi.y - lo.y, ez = hi.z - lo.z;
    if (!(ex >= 0.0f && ey >= 0.0f && ez >= 0.0f)) return 0.0f;      // the empty box of triangles that cannot be hit
    return ex * ey + ey * ez + ez * ex;
}

// c(n, .) and the choices of internal node id of the binary tree; its children's are final.  AGENT: the children's records
// were written by other threads of this launch (see st_agent)
template <bool AGENT>
__device__ void collapse_dp_node(uint32_t id, int n, float inv_root, const uint2* __restrict__ children,
                                 const float4* __restrict__ b2lo, const float4* __restrict__ b2hi, CollapseDp* dp)
{
    const uint2 ch = children[id];
    float cl[8], cr[8];
    auto load = [&](uint32_t c, float* out) {
        if (c >= (uint32_t)(n - 1)) {
            float v = dp_area(b2lo[c], b2hi[c]) * inv_root * (PT_COST_LEAF + PT_COST_TRI);
            for (int i = 1; i < 8; i++) out[i] = v;
        } else if (AGENT) {
            const float4* rec = reinterpret_cast<const float4*>(dp + c);
            float4 a = ld_agent(rec), b = ld_agent(rec + 1);
            out[1] = a.x; out[2] = a.y; out[3] = a.z; out[4] = a.w; out[5] = b.x; out[6] = b.y; out[7] = b.z;
        } else {
            for (int i = 1; i < 8; i++) out[i] = dp[c].c[i - 1];
        }
    };
    load(ch.x, cl);
    load(ch.y, cr);
    const float4 lo = b2lo[id], hi = b2hi[id];
    const float A = dp_area(lo, hi) * inv_root;
    const uint32_t P = __float_as_uint(lo.w);
    float split[9];
    uint32_t choice = 0;
    for (int j = 2; j <= 8; j++) {
        float best = __builtin_inff();
        uint32_t bk = 1;
        for (int k = 1; k < j; k++) {
            float v = cl[k] + cr[j - k];
            if (v < best) { best = v; bk = (uint32_t)k; }
        }
        split[j] = best;
        choice |= bk << (3 * (j - 2));
    }
    const float c_node = A * PT_COST_NODE + split[8];
    const float c_leaf = P <= PT_LEAF_MAX ? A * (PT_COST_LEAF + (float)P * PT_COST_TRI) : __builtin_inff();
    CollapseDp r;
    r.c[0] = fminf(c_leaf, c_node);
    if (c_leaf <= c_node) choice |= 1u << 21;
    for (int i = 2; i <= 7; i++) {
        if (r.c[i - 2] <= split[i]) { r.c[i - 1] = r.c[i - 2]; choice |= 1u << (22 + i - 2); }
        else r.c[i - 1] = split[i];
    }
    r.choice = choice;
    if (AGENT) {
        float4* rec = reinterpret_cast<float4*>(dp + id);
        st_agent(rec, make_float4(r.c[0], r.c[1], r.c[2], r.c[3]));
        st_agent(rec + 1, make_float4(r.c[4], r.c[5], r.c[6], __uint_as_float(choice)));
    } else {
        dp[id] = r;
    }
}

__device__ __forceinline__ float dp_inv_root(uint32_t root, const float4* __restrict__ b2lo, const float4* __restrict__ b2hi)
{
    float root_area = dp_area(b2lo[root], b2hi[root]);
    return root_area > 0.0f ? 1.0f / root_area : 1.0f;
}

// clustered trees: nodes [first, first + count), whose children are leaves or have smaller numbers than first (the nodes of
// one clustering round)
__global__ void k_collapse_dp(uint32_t first, uint32_t count, int n, uint32_t root, const uint2* __restrict__ children,
                              const float4* __restrict__ b2lo, const float4* __restrict__ b2hi, CollapseDp* dp)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= count) return;
    collapse_dp_node<false>(first + g, n, dp_inv_root(root, b2lo, b2hi), children, b2lo, b2hi, dp);
}

// Morton radix trees: bottom-up from every leaf, the second thread to arrive at a node computes it (flags zeroed)
__global__ void k_collapse_dp_up(int n, uint32_t root, const uint2* __restrict__ children, const uint32_t* __restrict__ parent, uint32_t* flags,
                                 const float4* __restrict__ b2lo, const float4* __restrict__ b2hi, CollapseDp* dp)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || n == 1) return;
    const float inv_root = dp_inv_root(root, b2lo, b2hi);
    uint32_t cur = parent[(uint32_t)(n - 1 + j)];
    while (cur != NONE) {
        if (!arrive_second(&flags[cur])) return;     // the sibling subtree is not finished: its thread continues
        collapse_dp_node<true>(cur, n, inv_root, children, b2lo, b2hi, dp);
        cur = parent[cur];
    }
}

// one thread per frontier item: builds one Node8 and queues its internal children.  dp == nullptr: greedy choice of the
// children (open the one with the largest box), every subtree of at most PT_LEAF_MAX triangles a leaf
__global__ void k_collapse(uint32_t nitems, const CollapseItem* __restrict__ in, CollapseItem* __restrict__ out, uint32_t* out_count,
                           uint32_t* node_counter, uint32_t* leaf_counter, int n, const uint2* __restrict__ children,
                           const float4* __restrict__ b2lo, const float4* __restrict__ b2hi, const uint32_t* __restrict__ vals,
                           const TriRec* __restrict__ tris_in, TriRec* __restrict__ tris_out, Node8* __restrict__ nodes, float pad,
                           const CollapseDp* __restrict__ dp)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nitems) return;
    CollapseItem it = in[g];
    uint32_t cand[8];
    int nc;
    auto count_of = [&](uint32_t id) { return __float_as_uint(b2lo[id].w); };
    auto is_leaf2 = [&](uint32_t id) { return id >= (uint32_t)(n - 1); };
    // does subtree c, presented as one child, become a leaf?
    auto leafy = [&](uint32_t c) {
        if (is_leaf2(c)) return true;
        if (count_of(c) > PT_LEAF_MAX) return false;
        return dp ? ((dp[c].choice >> 21) & 1u) != 0 : true;
    };
    if (leafy(it.node2)) {
        cand[0] = it.node2;   // only the root can get here: the whole scene fits one leaf
        nc = 1;
    } else if (dp) {
        // follow the programme's choices from split(node, 8) down to subtrees that are presented as one child
        uint32_t st_id[8];
        int st_j[8], sp = 0;
        nc = 0;
        {
            uint2 ch = children[it.node2];
            int k = (int)((dp[it.node2].choice >> 18) & 7u);
            st_id[sp] = ch.y; st_j[sp++] = 8 - k;
            st_id[sp] = ch.x; st_j[sp++] = k;
        }
        while (sp) {
            uint32_t id = st_id[--sp];
            int j = st_j[sp];
            if (is_leaf2(id) || j == 1) { cand[nc++] = id; continue; }
            uint32_t choice = dp[id].choice;
            while (j >= 2 && ((choice >> (22 + j - 2)) & 1u)) j--;
            if (j == 1) { cand[nc++] = id; continue; }
            int k = (int)((choice >> (3 * (j - 2))) & 7u);
            uint2 ch = children[id];
            st_id[sp] = ch.y; st_j[sp++] = j - k;
            st_id[sp] = ch.x; st_j[sp++] = k;
        }
    } else {
        uint2 ch = children[it.node2];
        cand[0] = ch.x; cand[1] = ch.y;
        nc = 2;
        while (nc < 8) {
            // open the candidate with the largest box that is not going to be a leaf
            int bestk = -1;
            float besta = -1.0f;
            for (int k = 0; k < nc; k++) {
                if (is_leaf2(cand[k]) || count_of(cand[k]) <= PT_LEAF_MAX) continue;
                float a = box_area(b2lo[cand[k]], b2hi[cand[k]]);
                if (a > besta) { besta = a; bestk = k; }
            }
            if (bestk < 0) break;
            uint2 c2 = children[cand[bestk]];
            cand[bestk] = c2.x;
            cand[nc++] = c2.y;
        }
    }
    NodeGrid gr;
    {
        float4 clo[8], chi[8];
        for (int k = 0; k < nc; k++) { clo[k] = b2lo[cand[k]]; chi[k] = b2hi[cand[k]]; }
        gr = node_grid(clo, chi, nc, pad);
    }
    Node8 nd;
    nd.origin[0] = gr.lo[0]; nd.origin[1] = gr.lo[1]; nd.origin[2] = gr.lo[2];
    nd.exps = gr.e[0] | (gr.e[1] << 8) | (gr.e[2] << 16);
    for (int w = 0; w < 12; w++) nd.q[w] = 0u;
    for (int w = 0; w < 8; w++) nd.pad[w] = 0u;
    uint32_t tstart = it.tstart;
    for (int k = 0; k < 8; k++) {
        uint32_t qlo[3] = {255u, 255u, 255u}, qhi[3] = {0u, 0u, 0u};
        if (k >= nc) {
            nd.child[k] = PT_CHILD_EMPTY;
        } else {
            uint32_t c = cand[k];
            float4 lo = b2lo[c], hi = b2hi[c];
            uint32_t cnt = __float_as_uint(lo.w);
            quantise_child(gr, lo, hi, pad, qlo, qhi);
            if (leafy(c)) {
                nd.child[k] = PT_CHILD_LEAF | ((cnt - 1u) << 28) | tstart;
                // write the subtree's triangles in the order of their (submesh, primitive): the triangle phase of the
                // traversal resolves hits at equal distance inside a leaf by position
                uint32_t stack[16], src[PT_LEAF_MAX];
                unsigned long long ord[PT_LEAF_MAX];
                int sp = 0;
                uint32_t m = 0;
                stack[sp++] = c;
                while (sp) {
                    uint32_t id = stack[--sp];
                    if (is_leaf2(id)) {
                        if (m < PT_LEAF_MAX) {
                            const uint32_t t = vals[id - (uint32_t)(n - 1)];
                            const unsigned long long key = ((unsigned long long)tris_in[t].submesh << 32) | tris_in[t].prim;
                            uint32_t at = m++;
                            while (at > 0 && ord[at - 1] > key) { ord[at] = ord[at - 1]; src[at] = src[at - 1]; at--; }
                            ord[at] = key; src[at] = t;
                        }
                    } else {
                        uint2 c2 = children[id];
                        stack[sp++] = c2.y;
                        stack[sp++] = c2.x;
                    }
                }
                for (uint32_t w = 0; w < m; w++) tris_out[tstart + w] = tris_in[src[w]];
                atomicAdd(leaf_counter, 1u);
            } else {
                uint32_t slot = atomicAdd(node_counter, 1u);
                nd.child[k] = slot;
                uint32_t o = atomicAdd(out_count, 1u);
                out[o] = CollapseItem{c, slot, tstart, 0u};
            }
            tstart += cnt;
        }
        store_child_planes(nd, k, qlo, qhi);
    }
    nodes[it.slot] = nd;
}

// Refit, one level at a time from the deepest up: one thread per node recomputes the node's grid and its quantised
// child boxes from the children's current bounds (a leaf's from its triangles, an internal child's from nlo / nhi,
// written when the level below was done) and records its own bounds.  Child references do not change.
__global__ void k_refit_nodes(uint32_t first, uint32_t count, Node8* __restrict__ nodes, const float4* __restrict__ tlo,
                              const float4* __restrict__ thi, float4* nlo, float4* nhi, float pad)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= count) return;
    Node8 nd = nodes[first + g];
    float4 clo[8], chi[8];
    int nc = 0;
    float4 ulo = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.0f);
    float4 uhi = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), 0.0f);
    for (int k = 0; k < 8; k++) {
        uint32_t ch = nd.child[k];
        if (ch == PT_CHILD_EMPTY) break;               // children fill the slots from 0
        float4 lo, hi;
        if (ch & PT_CHILD_LEAF) {
            uint32_t t0 = ch & 0x0FFFFFFFu, cnt = ((ch >> 28) & 7u) + 1u;
            lo = tlo[t0]; hi = thi[t0];
            for (uint32_t j = 1; j < cnt; j++) {
                float4 l = tlo[t0 + j], h = thi[t0 + j];
                lo.x = fminf(lo.x, l.x); lo.y = fminf(lo.y, l.y); lo.z = fminf(lo.z, l.z);
                hi.x = fmaxf(hi.x, h.x); hi.y = fmaxf(hi.y, h.y); hi.z = fmaxf(hi.z, h.z);
            }
        } else {
            lo = nlo[ch]; hi = nhi[ch];
        }
        clo[nc] = lo; chi[nc] = hi;
        nc++;
        ulo.x = fminf(ulo.x, lo.x); ulo.y = fminf(ulo.y, lo.y); ulo.z = fminf(ulo.z, lo.z);
        uhi.x = fmaxf(uhi.x, hi.x); uhi.y = fmaxf(uhi.y, hi.y); uhi.z = fmaxf(uhi.z, hi.z);
    }
    NodeGrid gr = node_grid(clo, chi, nc, pad);
    nd.origin[0] = gr.lo[0]; nd.origin[1] = gr.lo[1]; nd.origin[2] = gr.lo[2];
    nd.exps = gr.e[0] | (gr.e[1] << 8) | (gr.e[2] << 16);
    for (int w = 0; w < 12; w++) nd.q[w] = 0u;
    for (int k = 0; k < 8; k++) {
        uint32_t qlo[3] = {255u, 255u, 255u}, qhi[3] = {0u, 0u, 0u};
        if (k < nc) quantise_child(gr, clo[k], chi[k], pad, qlo, qhi);
        store_child_planes(nd, k, qlo, qhi);
    }
    nodes[first + g] = nd;
    nlo[first + g] = ulo;
    nhi[first + g] = uhi;
}

// the shading record of every triangle, in leaf order (see ShadeTri)
__global__ void k_shade_tris(uint32_t n, const TriRec* __restrict__ tris, const uint32_t* __restrict__ submeshes,
                             const uint32_t* __restrict__ indices, const float* __restrict__ vertices, ShadeTri* __restrict__ out)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint32_t s = tris[g].submesh, prim = tris[g].prim;
    const uint32_t vo = submeshes[s * 4 + 1], io = submeshes[s * 4 + 2];
    ShadeTri r;
    for (int k = 0; k < 3; k++) {
        const float* v = vertices + (size_t)(indices[io + prim * 3 + k] + vo) * 12;
        r.uv[k][0] = v[3]; r.uv[k][1] = v[4];
        r.n[k][0] = v[5]; r.n[k][1] = v[6]; r.n[k][2] = v[7];
    }
    r.material = submeshes[s * 4 + 3];
    r.submesh = s;
    r.prim = prim;
    r.pad[0] = r.pad[1] = 0u;
    out[g] = r;
}

// surface-area-heuristic cost: sum over nodes of area(child box)/area(root) * (1 for a node, count for a leaf)
__global__ void k_sah(uint32_t num_nodes, const Node8* __restrict__ nodes, float root_area, float* cost)
{
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    float c = 0.0f;
    if (g < num_nodes) {
        const Node8& nd = nodes[g];
        float st[3];
        for (int a = 0; a < 3; a++) st[a] = __uint_as_float(((nd.exps >> (8 * a)) & 255u) << 23);
        for (int k = 0; k < 8; k++) {
            uint32_t ch = nd.child[k];
            if (ch == PT_CHILD_EMPTY) continue;
            float e[3];
            for (int a = 0; a < 3; a++) {
                uint32_t ql = (nd.q[a * 2 + (k >> 2)] >> ((k & 3) * 8)) & 255u;
                uint32_t qh = (nd.q[(3 + a) * 2 + (k >> 2)] >> ((k & 3) * 8)) & 255u;
                e[a] = (float)(qh - ql) * st[a];
            }
            float a = (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]) / root_area;
            if (ch & PT_CHILD_LEAF) c += a * (float)(((ch >> 28) & 7u) + 1u);
            else c += a;
        }
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0 && c != 0.0f) atomicAdd(cost, c);
}

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair()
    {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

// scene bounds over the triangles that can be hit (read back from the ordered-int minima / maxima) and the padding
// of every child box, which covers the rounding of the slab test and of the triangle test's t
static int scene_extent(pt_ctx* ctx, int* d_bounds, float* lo, float* hi, float& pad)
{
    hipStream_t st = ctx->stream;
    int hb[6];
    PT_HIP(ctx, hipMemcpyAsync(hb, d_bounds, sizeof(hb), hipMemcpyDeviceToHost, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    auto o2f = [](int i) { uint32_t u = (uint32_t)(i >= 0 ? i : i ^ 0x7fffffff); float f; memcpy(&f, &u, 4); return f; };
    for (int k = 0; k < 3; k++) { lo[k] = o2f(hb[k]); hi[k] = o2f(hb[3 + k]); }
    if (!(lo[0] <= hi[0])) {                      // no triangle has finite coordinates
        for (int k = 0; k < 3; k++) lo[k] = hi[k] = 0.0f;
        PT_HIP(ctx, hipMemsetAsync(d_bounds, 0, 6 * sizeof(int), st));
    }
    float mc = 0.0f;
    for (int k = 0; k < 3; k++) mc = fmaxf(mc, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
    // the quantised boxes and the slab test multiply extents by 1/direction: keep well inside the float range
    if (!(mc <= 1.0e18f)) return set_error(ctx, PT_ERR_INVALID, "scene coordinates beyond +-1e18 are not supported");
    float diag = sqrtf((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    pad = fmaxf(diag, mc) * (1.0f / 262144.0f);
    if (!(pad == pad) || pad > 3.0e38f) return set_error(ctx, PT_ERR_INVALID, "scene bounds are not finite");
    return PT_OK;
}

static float padded_area(const float* lo, const float* hi, float pad)
{
    float ex = hi[0] - lo[0] + 2 * pad, ey = hi[1] - lo[1] + 2 * pad, ez = hi[2] - lo[2] + 2 * pad;
    return ex * ey + ey * ez + ez * ex;
}

int build_accel(pt_ctx* ctx, uint32_t flags)
{
    PT_SYNC(ctx);
    hipStream_t st = ctx->stream;
    uint32_t n = ctx->num_tris;
    if (ctx->d_nodes) { (void)hipFree(ctx->d_nodes); ctx->d_nodes = nullptr; }
    if (ctx->d_tris) { (void)hipFree(ctx->d_tris); ctx->d_tris = nullptr; }
    if (ctx->d_shade_tris) { (void)hipFree(ctx->d_shade_tris); ctx->d_shade_tris = nullptr; }
    ctx->accel_built = false;
    ctx->num_nodes = 0;
    pt_accel_stats& stats = ctx->accel_stats;
    stats = pt_accel_stats{};
    stats.num_triangles = n;
    ctx->num_refs = 0;
    if (n == 0) { ctx->accel_built = true; return PT_OK; }
    if (n > 0x0FFFFFFFu) return set_error(ctx, PT_ERR_TOO_LARGE, "more than 2^28-1 triangles");

    EventPair ev;        // destroyed on every way out
    PT_HIP(ctx, hipEventCreate(&ev.a));
    PT_HIP(ctx, hipEventCreate(&ev.b));
    hipEvent_t e0 = ev.a, e1 = ev.b;
    PT_HIP(ctx, hipEventRecord(e0, st));

    DevBuf<TriRec> tris_tmp, tris_ref;
    DevBuf<float4> tlo, thi, rlo, rhi, b2lo, b2hi;
    DevBuf<int> bounds;
    DevBuf<unsigned long long> keys, keys_sorted;
    DevBuf<uint32_t> vals, vals_sorted, parent, flagsb, counters;
    DevBuf<uint2> children;
    DevBuf<CollapseItem> fr[2];
    DevBuf<CollapseDp> dp;
    std::vector<std::pair<uint32_t, uint32_t>> dp_rounds;     // (first node, count) of every clustering round: children before parents
    DevBuf<char> sort_tmp;
    DevBuf<float> cost;
    const int B = 256;
    PT_HIP(ctx, tris_tmp.alloc(n));
    PT_HIP(ctx, tlo.alloc(n));
    PT_HIP(ctx, thi.alloc(n));
    PT_HIP(ctx, bounds.alloc(6));
    k_init_bounds<<<1, 64, 0, st>>>(bounds.p);
    k_world_tris<false><<<(n + B - 1) / B, B, 0, st>>>(n, ctx->num_submeshes, ctx->d_sub_tri_offset, ctx->d_submeshes, ctx->d_xforms,
                                                      ctx->d_indices, ctx->d_vertices, tris_tmp.p, tlo.p, thi.p, bounds.p);
    float lo[3], hi[3], pad;
    {
        int rc = scene_extent(ctx, bounds.p, lo, hi, pad);
        if (rc) return rc;
    }
    // from here on n counts references: a split triangle is entered once per piece
    TriRec* src_tris = tris_tmp.p;
    float4 *src_lo = tlo.p, *src_hi = thi.p;
    if (!(flags & PT_BUILD_NO_SPLIT)) {
        DevBuf<uint32_t> cnt, off, cand, ncand;
        DevBuf<unsigned char> mode;
        DevBuf<char> scan_tmp;
        PT_HIP(ctx, cnt.alloc(n));
        PT_HIP(ctx, off.alloc(n));
        PT_HIP(ctx, cand.alloc(n));
        PT_HIP(ctx, ncand.alloc(1));
        PT_HIP(ctx, mode.alloc(n));
        PT_HIP(ctx, hipMemsetAsync(ncand.p, 0, sizeof(uint32_t), st));
        k_split_classify<<<(n + B - 1) / B, B, 0, st>>>(n, tris_tmp.p, tlo.p, thi.p, bounds.p, cnt.p, mode.p, cand.p, ncand.p);
        uint32_t nsplit = 0;
        PT_HIP(ctx, hipMemcpyAsync(&nsplit, ncand.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
        if (nsplit) {
            k_split_cells<false><<<nsplit, 64, 0, st>>>(nsplit, cand.p, tris_tmp.p, tlo.p, thi.p, cnt.p, mode.p, nullptr, nullptr, nullptr, nullptr);
            size_t scan_bytes = 0;
            PT_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, cnt.p, off.p, (int)n, st));
            PT_HIP(ctx, scan_tmp.alloc(scan_bytes));
            PT_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, scan_bytes, cnt.p, off.p, (int)n, st));
            uint32_t last[2];
            PT_HIP(ctx, hipMemcpyAsync(&last[0], off.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            PT_HIP(ctx, hipMemcpyAsync(&last[1], cnt.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            PT_HIP(ctx, hipStreamSynchronize(st));
            const uint64_t nref = (uint64_t)last[0] + last[1];
            if (nref > 0x0FFFFFFFull) return set_error(ctx, PT_ERR_TOO_LARGE, "more than 2^28-1 triangle references after splitting");
            if (nref > n) {
                PT_HIP(ctx, tris_ref.alloc(nref));
                PT_HIP(ctx, rlo.alloc(nref));
                PT_HIP(ctx, rhi.alloc(nref));
                k_split_copy_whole<<<(n + B - 1) / B, B, 0, st>>>(n, tris_tmp.p, tlo.p, thi.p, mode.p, off.p, tris_ref.p, rlo.p, rhi.p);
                k_split_cells<true><<<nsplit, 64, 0, st>>>(nsplit, cand.p, tris_tmp.p, tlo.p, thi.p, cnt.p, mode.p, off.p, tris_ref.p, rlo.p, rhi.p);
                src_tris = tris_ref.p; src_lo = rlo.p; src_hi = rhi.p;
                n = (uint32_t)nref;
            }
        }
        PT_HIP(ctx, hipStreamSynchronize(st));      // the buffers of this block go out of scope
    }
    ctx->num_refs = n;
    stats.num_references = n;
    PT_HIP(ctx, b2lo.alloc(2 * (size_t)n));
    PT_HIP(ctx, b2hi.alloc(2 * (size_t)n));
    PT_HIP(ctx, keys.alloc(n));
    PT_HIP(ctx, keys_sorted.alloc(n));
    PT_HIP(ctx, vals.alloc(n));
    PT_HIP(ctx, vals_sorted.alloc(n));
    PT_HIP(ctx, parent.alloc(2 * (size_t)n));
    PT_HIP(ctx, flagsb.alloc(n));
    PT_HIP(ctx, counters.alloc(8));
    PT_HIP(ctx, children.alloc(n));
    PT_HIP(ctx, fr[0].alloc(n));
    PT_HIP(ctx, fr[1].alloc(n));
    PT_HIP(ctx, cost.alloc(1));
    PT_HIP(ctx, hipMalloc((void**)&ctx->d_nodes, (size_t)n * sizeof(Node8)));
    PT_HIP(ctx, hipMalloc((void**)&ctx->d_tris, (size_t)n * sizeof(TriRec)));
    PT_HIP(ctx, hipMalloc((void**)&ctx->d_shade_tris, (size_t)n * sizeof(ShadeTri)));

    uint32_t G = (n + B - 1) / B;
    k_morton<<<G, B, 0, st>>>(n, src_lo, src_hi, bounds.p, keys.p, vals.p);
    size_t tmp_bytes = 0;
    PT_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys.p, keys_sorted.p, vals.p, vals_sorted.p, (int)n, 0, 63, st));
    PT_HIP(ctx, sort_tmp.alloc(tmp_bytes));
    PT_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(sort_tmp.p, tmp_bytes, keys.p, keys_sorted.p, vals.p, vals_sorted.p, (int)n, 0, 63, st));
    // Binary tree over the sorted references (node ids: internal 0..n-2, leaf j -> (n-1)+j): the Morton radix tree, improved
    // by treelet restructuring passes (7 % more rays per second than agglomerative clustering without them on the
    // architectural test scenes, 4 % on the large one, at a lower build time).  PT_BUILD_CLUSTER builds the clustered tree
    // instead; with PT_BUILD_LBVH as well both are built and the one whose surface-area-optimal collapse is cheaper
    // (c(root, 1) of k_collapse_dp) is kept.  PT_BUILD_NO_TREELET leaves the restructuring out.
    const bool greedy = (flags & PT_BUILD_GREEDY) != 0;
    const bool want_cluster = (flags & PT_BUILD_CLUSTER) != 0;
    const bool want_radix = !want_cluster || ((flags & PT_BUILD_LBVH) && !greedy && n > 1);
    const bool treelets = !(flags & PT_BUILD_NO_TREELET) && n > 1;
    uint32_t root_id = 0;
    DevBuf<uint2> children_r;                  // the radix tree when both are built
    DevBuf<float4> b2lo_r, b2hi_r;
    DevBuf<CollapseDp> dp_r;
    uint2* children_p = children.p;
    float4 *b2lo_p = b2lo.p, *b2hi_p = b2hi.p;
    CollapseDp* dp_p = nullptr;
    if (want_radix) {
        uint2* ch = children.p;
        float4 *lo2 = b2lo.p, *hi2 = b2hi.p;
        DevBuf<CollapseDp>* dpb = &dp;
        if (want_cluster) {
            PT_HIP(ctx, children_r.alloc(n));
            PT_HIP(ctx, b2lo_r.alloc(2 * (size_t)n));
            PT_HIP(ctx, b2hi_r.alloc(2 * (size_t)n));
            ch = children_r.p; lo2 = b2lo_r.p; hi2 = b2hi_r.p; dpb = &dp_r;
        }
        PT_HIP(ctx, hipMemsetAsync(flagsb.p, 0, (size_t)n * sizeof(uint32_t), st));
        PT_HIP(ctx, hipMemsetAsync(parent.p, 0xFF, 2 * (size_t)n * sizeof(uint32_t), st));
        if (n > 1) k_karras<<<G, B, 0, st>>>((int)n, keys_sorted.p, ch, parent.p);
        k_refit<<<G, B, 0, st>>>((int)n, vals_sorted.p, src_lo, src_hi, ch, parent.p, flagsb.p, lo2, hi2);
        for (int pass = 0; treelets && pass < PT_TREELET_PASSES; pass++) {
            PT_HIP(ctx, hipMemsetAsync(flagsb.p, 0, (size_t)n * sizeof(uint32_t), st));
            k_treelet_up<<<(n + 63) / 64, 64, 0, st>>>((int)n, ch, parent.p, flagsb.p, lo2, hi2);
        }
        if (n > 1 && !greedy) {
            PT_HIP(ctx, dpb->alloc(n));
            PT_HIP(ctx, hipMemsetAsync(flagsb.p, 0, (size_t)n * sizeof(uint32_t), st));
            k_collapse_dp_up<<<G, B, 0, st>>>((int)n, 0u, ch, parent.p, flagsb.p, lo2, hi2, dpb->p);
        }
        children_p = ch; b2lo_p = lo2; b2hi_p = hi2; dp_p = dpb->p;      // the radix tree's root is node 0
    }
    if (want_cluster) {
        DevBuf<uint32_t> cid[2], nearest, totals;
        DevBuf<unsigned long long> pflags, pscan;
        DevBuf<char> scan_tmp;
        PT_HIP(ctx, cid[0].alloc(n));
        PT_HIP(ctx, cid[1].alloc(n));
        PT_HIP(ctx, nearest.alloc(n));
        PT_HIP(ctx, totals.alloc(2));
        PT_HIP(ctx, pflags.alloc(n));
        PT_HIP(ctx, pscan.alloc(n));
        size_t scan_bytes = 0;
        PT_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, pflags.p, pscan.p, (int)n, st));
        PT_HIP(ctx, scan_tmp.alloc(scan_bytes));
        k_ploc_init<<<G, B, 0, st>>>((int)n, vals_sorted.p, src_lo, src_hi, cid[0].p, b2lo.p, b2hi.p);
        uint32_t nc = n, next_id = 0;
        int cur = 0, rounds = 0;
        while (nc > 1) {
            uint32_t g = (nc + PLOC_BLOCK - 1) / PLOC_BLOCK;
            k_ploc_nearest<<<g, PLOC_BLOCK, 0, st>>>((int)nc, cid[cur].p, b2lo.p, b2hi.p, nearest.p);
            k_ploc_flags<<<g, PLOC_BLOCK, 0, st>>>((int)nc, nearest.p, pflags.p);
            PT_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, scan_bytes, pflags.p, pscan.p, (int)nc, st));
            k_ploc_merge<<<g, PLOC_BLOCK, 0, st>>>((int)nc, next_id, nearest.p, pflags.p, pscan.p, cid[cur].p, cid[cur ^ 1].p,
                                                   children.p, b2lo.p, b2hi.p, totals.p);
            uint32_t ht[2];
            PT_HIP(ctx, hipMemcpyAsync(ht, totals.p, sizeof(ht), hipMemcpyDeviceToHost, st));
            PT_HIP(ctx, hipStreamSynchronize(st));
            if (ht[1] == 0 || ht[0] != nc - ht[1]) return set_error(ctx, PT_ERR_INVALID, "hierarchy build: clustering made no progress");
            nc = ht[0];
            dp_rounds.emplace_back(next_id, ht[1]);
            next_id += ht[1];
            cur ^= 1;
            if (++rounds > 4096) return set_error(ctx, PT_ERR_INVALID, "hierarchy build: clustering did not terminate");
        }
        const uint32_t cluster_root = n > 1 ? next_id - 1 : 0;   // the last node created is the root
        if (treelets) {
            k_parents<<<G, B, 0, st>>>((int)n, cluster_root, children.p, parent.p);
            for (int pass = 0; pass < PT_TREELET_PASSES; pass++) {
                PT_HIP(ctx, hipMemsetAsync(flagsb.p, 0, (size_t)n * sizeof(uint32_t), st));
                k_treelet_up<<<(n + 63) / 64, 64, 0, st>>>((int)n, children.p, parent.p, flagsb.p, b2lo.p, b2hi.p);
            }
        }
        if (n > 1 && !greedy) {
            PT_HIP(ctx, dp.alloc(n));
            if (treelets) {      // restructuring moved node numbers between depths: bottom-up by the parent pointers instead
                PT_HIP(ctx, hipMemsetAsync(flagsb.p, 0, (size_t)n * sizeof(uint32_t), st));
                k_collapse_dp_up<<<G, B, 0, st>>>((int)n, cluster_root, children.p, parent.p, flagsb.p, b2lo.p, b2hi.p, dp.p);
            } else {
                for (auto& r : dp_rounds)
                    k_collapse_dp<<<(r.second + B - 1) / B, B, 0, st>>>(r.first, r.second, (int)n, cluster_root, children.p, b2lo.p, b2hi.p, dp.p);
            }
        }
        bool keep_cluster = true;
        if (want_radix && n > 1 && !greedy) {
            float cost_c = 0.0f, cost_r = 0.0f;
            PT_HIP(ctx, hipMemcpyAsync(&cost_c, &dp.p[cluster_root].c[0], sizeof(float), hipMemcpyDeviceToHost, st));
            PT_HIP(ctx, hipMemcpyAsync(&cost_r, &dp_r.p[0].c[0], sizeof(float), hipMemcpyDeviceToHost, st));
            PT_HIP(ctx, hipStreamSynchronize(st));
            keep_cluster = !(cost_r < cost_c);
        }
        if (keep_cluster) {
            children_p = children.p; b2lo_p = b2lo.p; b2hi_p = b2hi.p; dp_p = greedy ? nullptr : dp.p;
            root_id = cluster_root;
        }
    }

    if (dp_p && n > 1 && getenv("PT_BUILD_STATS")) {      // the cost the collapse minimises, for comparisons with other builders
        float c = 0.0f;
        PT_HIP(ctx, hipMemcpyAsync(&c, &dp_p[root_id].c[0], sizeof(float), hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
        fprintf(stderr, "pathtracer_amd: hierarchy build: %u references, collapse cost c(root, 1) = %.3f\n", n, c);
    }
    // collapse: counters[0] = nodes allocated, [1] = leaves, [2],[3] = frontier sizes (ping-pong)
    uint32_t init[4] = {1u, 0u, 0u, 0u};
    PT_HIP(ctx, hipMemcpyAsync(counters.p, init, sizeof(init), hipMemcpyHostToDevice, st));
    CollapseItem root{root_id, 0u, 0u, 0u};   // n == 1: the single leaf has id n-1 = 0
    PT_HIP(ctx, hipMemcpyAsync(fr[0].p, &root, sizeof(root), hipMemcpyHostToDevice, st));
    uint32_t nitems = 1, depth = 0;
    int cur = 0;
    ctx->level_start.assign(1, 0u);        // nodes are numbered level by level: level d is [level_start[d], level_start[d + 1])
    while (nitems > 0) {
        depth++;
        ctx->level_start.push_back(ctx->level_start.back() + nitems);
        PT_HIP(ctx, hipMemsetAsync(counters.p + 2 + (cur ^ 1), 0, sizeof(uint32_t), st));
        k_collapse<<<(nitems + 63) / 64, 64, 0, st>>>(nitems, fr[cur].p, fr[cur ^ 1].p, counters.p + 2 + (cur ^ 1), counters.p, counters.p + 1,
                                                     (int)n, children_p, b2lo_p, b2hi_p, vals_sorted.p, src_tris, ctx->d_tris, ctx->d_nodes, pad, dp_p);
        PT_HIP(ctx, hipMemcpyAsync(&nitems, counters.p + 2 + (cur ^ 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PT_HIP(ctx, hipStreamSynchronize(st));
        cur ^= 1;
        if (depth > 4096) return set_error(ctx, PT_ERR_BVH_DEPTH, "hierarchy collapse did not terminate");
    }
    k_shade_tris<<<G, B, 0, st>>>(n, ctx->d_tris, ctx->d_submeshes, ctx->d_indices, ctx->d_vertices, ctx->d_shade_tris);
    uint32_t hc[2];
    PT_HIP(ctx, hipMemcpyAsync(hc, counters.p, sizeof(hc), hipMemcpyDeviceToHost, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    ctx->num_nodes = hc[0];
    stats.num_nodes = hc[0];
    stats.num_leaves = hc[1];
    stats.max_depth = depth;

    float root_area = padded_area(lo, hi, pad);
    PT_HIP(ctx, hipMemsetAsync(cost.p, 0, sizeof(float), st));
    if (root_area > 0.0f) k_sah<<<(ctx->num_nodes + B - 1) / B, B, 0, st>>>(ctx->num_nodes, ctx->d_nodes, root_area, cost.p);
    PT_HIP(ctx, hipMemcpyAsync(&stats.sah_cost, cost.p, sizeof(float), hipMemcpyDeviceToHost, st));
    PT_HIP(ctx, hipEventRecord(e1, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipEventElapsedTime(&stats.build_ms, e0, e1));

    // a ray's stack holds at most 7 siblings per level and one entry more; the spill area is sized for that
    if (7u * depth + 1u > (uint32_t)PT_STACK_MAX)
        return set_error(ctx, PT_ERR_BVH_DEPTH, "hierarchy has " + std::to_string(depth) + " levels; the traversal stack is limited to " +
                                                    std::to_string(PT_STACK_MAX) + " entries");
    ctx->stack_entries = depth ? ((7u * depth + 1u + 7u) & ~7u) : 8u;
    ctx->accel_built = true;
    return PT_OK;
}

// The hierarchy keeps its topology (which triangles share a leaf, which children a node has); triangles, child
// boxes and node grids follow the current transforms.  Exactly what a rebuild would write into the same topology.
int refit_accel(pt_ctx* ctx)
{
    PT_SYNC(ctx);
    if (!ctx->accel_built) return set_error(ctx, PT_ERR_INVALID, "refit: no hierarchy has been built");
    hipStream_t st = ctx->stream;
    const uint32_t n = ctx->num_refs;     // a reference to a part of a split triangle is bounded again from the part's barycentric bounds
    pt_accel_stats& stats = ctx->accel_stats;
    if (n == 0) return PT_OK;
    EventPair ev;        // destroyed on every way out
    PT_HIP(ctx, hipEventCreate(&ev.a));
    PT_HIP(ctx, hipEventCreate(&ev.b));
    hipEvent_t e0 = ev.a, e1 = ev.b;
    PT_HIP(ctx, hipEventRecord(e0, st));
    DevBuf<float4> tlo, thi, nlo, nhi;
    DevBuf<int> bounds;
    DevBuf<float> cost;
    PT_HIP(ctx, tlo.alloc(n));
    PT_HIP(ctx, thi.alloc(n));
    PT_HIP(ctx, nlo.alloc(ctx->num_nodes));
    PT_HIP(ctx, nhi.alloc(ctx->num_nodes));
    PT_HIP(ctx, bounds.alloc(6));
    PT_HIP(ctx, cost.alloc(1));
    const int B = 256;
    k_init_bounds<<<1, 64, 0, st>>>(bounds.p);
    k_world_tris<true><<<(n + B - 1) / B, B, 0, st>>>(n, ctx->num_submeshes, ctx->d_sub_tri_offset, ctx->d_submeshes, ctx->d_xforms,
                                                     ctx->d_indices, ctx->d_vertices, ctx->d_tris, tlo.p, thi.p, bounds.p);
    float lo[3], hi[3], pad;
    {
        int rc = scene_extent(ctx, bounds.p, lo, hi, pad);
        if (rc) { ctx->accel_built = false; return rc; }     // the triangles are already rewritten: the old hierarchy no longer fits them
    }
    for (size_t d = ctx->level_start.size() - 1; d-- > 0;) {
        uint32_t first = ctx->level_start[d], count = ctx->level_start[d + 1] - first;
        if (count) k_refit_nodes<<<(count + B - 1) / B, B, 0, st>>>(first, count, ctx->d_nodes, tlo.p, thi.p, nlo.p, nhi.p, pad);
    }
    float root_area = padded_area(lo, hi, pad);
    PT_HIP(ctx, hipMemsetAsync(cost.p, 0, sizeof(float), st));
    if (root_area > 0.0f) k_sah<<<(ctx->num_nodes + B - 1) / B, B, 0, st>>>(ctx->num_nodes, ctx->d_nodes, root_area, cost.p);
    PT_HIP(ctx, hipMemcpyAsync(&stats.sah_cost, cost.p, sizeof(float), hipMemcpyDeviceToHost, st));
    PT_HIP(ctx, hipEventRecord(e1, st));
    PT_HIP(ctx, hipStreamSynchronize(st));
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipEventElapsedTime(&stats.build_ms, e0, e1));
    return PT_OK;
}

} // namespace ptd
