// TEST INFRASTRUCTURE (CPU oracle) -- not part of the shipped library.
// PARITY UNPINNED for this file: the reference has no source for ray traversal;
// traceRayEXT (PathTracer/assets/shaders/RayTracing/RayGen.glsl:124) runs inside
// the Vulkan driver over a TLAS with one instance per submesh
// (PathTracer/src/RayTracingLayer.cpp:342-348).  This file states the semantics
// the oracle and the GPU kernels both implement:
//   * every submesh's triangles are moved to world space once with its 3x4
//     object-to-world transform;
//   * closest hit with tmin < t < tmax, opaque, no face culling;
//   * ray/triangle test: Woop/Benthin/Wald watertight test (edge functions in a
//     sheared ray space, double-precision fallback when an edge function is 0),
//     t = T / det by one IEEE division;
//   * ties: smaller t wins; at equal t the smaller (submesh, primitive) wins, so
//     the result does not depend on traversal order or on the BVH at all;
//   * hit attributes (u, v) are the weights of vertex 1 and vertex 2.
// A brute-force loop over all triangles defines the answer; the binned-SAH BVH
// below only makes the oracle fast and must agree with it (tests check that).
#pragma once
#include "glsl.h"
#include <algorithm>
#include <vector>

namespace orc {

struct WorldTri {
    vec3 v0, v1, v2;
    uint32_t submesh, prim;
};

struct Hit {
    float t, u, v;
    uint32_t tri;      // index into the world-triangle array, 0xFFFFFFFF = miss
};

struct RayPre {
    int kx, ky, kz;
    float Sx, Sy, Sz;
    vec3 o;
};

static inline RayPre ray_precompute(vec3 o, vec3 d)
{
    RayPre p;
    float ax = gabs(d.x), ay = gabs(d.y), az = gabs(d.z);
    int kz = 0;
    float m = ax;
    if (ay > m) { kz = 1; m = ay; }
    if (az > m) { kz = 2; }
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    if (d[kz] < 0.0f) { int t = kx; kx = ky; ky = t; }
    p.kx = kx; p.ky = ky; p.kz = kz;
    p.Sx = d[kx] / d[kz];
    p.Sy = d[ky] / d[kz];
    p.Sz = 1.0f / d[kz];
    p.o = o;
    return p;
}

// Returns true and fills (t,u,v) when the ray hits the triangle with t > tmin.
// The caller applies the upper bound and the tie rule.
static inline bool tri_test(const RayPre& r, const WorldTri& tr, float tmin, float& t, float& u, float& v)
{
    vec3 A = tr.v0 - r.o, B = tr.v1 - r.o, C = tr.v2 - r.o;
    float Akz = A[r.kz], Bkz = B[r.kz], Ckz = C[r.kz];
    float Ax = A[r.kx] - r.Sx * Akz, Ay = A[r.ky] - r.Sy * Akz;
    float Bx = B[r.kx] - r.Sx * Bkz, By = B[r.ky] - r.Sy * Bkz;
    float Cx = C[r.kx] - r.Sx * Ckz, Cy = C[r.ky] - r.Sy * Ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
        U = (float)(CxBy - CyBx);
        double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
        V = (float)(AxCy - AyCx);
        double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
        W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    float det = U + V + W;
    if (det == 0.0f) return false;
    float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
    float T = U * Az + V * Bz + W * Cz;
    float tt = T / det;
    if (!(tt > tmin)) return false;
    t = tt;
    u = V / det;
    v = W / det;
    return true;
}

static inline bool closer(float t, uint32_t tri_sub, uint32_t tri_prim, const Hit& best, const std::vector<WorldTri>& tris)
{
    if (t < best.t) return true;
    if (t > best.t || best.tri == 0xFFFFFFFFu) return false;
    const WorldTri& b = tris[best.tri];
    return tri_sub < b.submesh || (tri_sub == b.submesh && tri_prim < b.prim);
}

static inline Hit intersect_brute(const std::vector<WorldTri>& tris, vec3 o, vec3 d, float tmin, float tmax)
{
    RayPre r = ray_precompute(o, d);
    Hit best{tmax, 0.0f, 0.0f, 0xFFFFFFFFu};
    for (uint32_t i = 0; i < (uint32_t)tris.size(); i++) {
        float t, u, v;
        if (!tri_test(r, tris[i], tmin, t, u, v)) continue;
        if (!(t < tmax)) continue;
        if (closer(t, tris[i].submesh, tris[i].prim, best, tris)) best = Hit{t, u, v, i};
    }
    return best;
}

// ---- binned-SAH BVH2 (oracle speed only) ------------------------------------
struct Aabb {
    vec3 lo{INFINITY, INFINITY, INFINITY}, hi{-INFINITY, -INFINITY, -INFINITY};
    void grow(vec3 p) {
        lo = V3(gmin(lo.x, p.x), gmin(lo.y, p.y), gmin(lo.z, p.z));
        hi = V3(gmax(hi.x, p.x), gmax(hi.y, p.y), gmax(hi.z, p.z));
    }
    void grow(const Aabb& b) { grow(b.lo); grow(b.hi); }
    float area() const {
        vec3 e = hi - lo;
        return 2.0f * (e.x * e.y + e.y * e.z + e.z * e.x);
    }
};

struct BvhNode {
    Aabb box;
    uint32_t left, right;    // children (internal) ...
    uint32_t first, count;   // ... or triangle range in `order` (leaf when count > 0)
};

struct Bvh {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> order;
    float pad = 0.0f;
    uint64_t visited_nodes = 0, tested_tris = 0;   // statistics (not thread safe; per-thread copies are summed by the caller)

    void build(const std::vector<WorldTri>& tris)
    {
        uint32_t n = (uint32_t)tris.size();
        order.resize(n);
        std::vector<Aabb> tb(n);
        std::vector<vec3> cen(n);
        Aabb all;
        for (uint32_t i = 0; i < n; i++) {
            order[i] = i;
            tb[i].grow(tris[i].v0); tb[i].grow(tris[i].v1); tb[i].grow(tris[i].v2);
            cen[i] = (tb[i].lo + tb[i].hi) * 0.5f;
            all.grow(tb[i]);
        }
        vec3 e = all.hi - all.lo;
        float diag = sqrtf(dot(e, e));
        float mc = 0.0f;
        for (int k = 0; k < 3; k++) mc = gmax(mc, gmax(gabs((&all.lo.x)[k]), gabs((&all.hi.x)[k])));
        pad = gmax(diag, mc) * (1.0f / 262144.0f);
        nodes.clear();
        nodes.reserve(2 * n + 1);
        nodes.push_back(BvhNode{});
        if (n == 0) { nodes[0].count = 0; nodes[0].left = nodes[0].right = 0; nodes[0].first = 0; return; }
        build_rec(0, 0, n, tb, cen);
        for (auto& nd : nodes) {
            nd.box.lo = nd.box.lo - V3(pad);
            nd.box.hi = nd.box.hi + V3(pad);
        }
    }

    void build_rec(uint32_t ni, uint32_t first, uint32_t count, const std::vector<Aabb>& tb, const std::vector<vec3>& cen)
    {
        Aabb box, cb;
        for (uint32_t i = first; i < first + count; i++) { box.grow(tb[order[i]]); cb.grow(cen[order[i]]); }
        nodes[ni].box = box;
        nodes[ni].first = first;
        nodes[ni].count = count;
        nodes[ni].left = nodes[ni].right = 0;
        if (count <= 4) return;
        const int NB = 16;
        int bestAxis = -1, bestSplit = 0;
        float bestCost = INFINITY;
        for (int ax = 0; ax < 3; ax++) {
            float lo = (&cb.lo.x)[ax], hi = (&cb.hi.x)[ax];
            if (!(hi > lo)) continue;
            Aabb bb[NB]; uint32_t bc[NB] = {0};
            float scale = NB / (hi - lo);
            for (uint32_t i = first; i < first + count; i++) {
                int b = (int)(((&cen[order[i]].x)[ax] - lo) * scale);
                b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                bb[b].grow(tb[order[i]]); bc[b]++;
            }
            float la[NB], ra[NB]; uint32_t lc[NB], rc[NB];
            Aabb acc; uint32_t c = 0;
            for (int b = 0; b < NB - 1; b++) { if (bc[b]) acc.grow(bb[b]); c += bc[b]; la[b] = c ? acc.area() : 0.0f; lc[b] = c; }
            acc = Aabb(); c = 0;
            for (int b = NB - 1; b > 0; b--) { if (bc[b]) acc.grow(bb[b]); c += bc[b]; ra[b - 1] = c ? acc.area() : 0.0f; rc[b - 1] = c; }
            for (int b = 0; b < NB - 1; b++) {
                if (lc[b] == 0 || rc[b] == 0) continue;
                float cost = la[b] * lc[b] + ra[b] * rc[b];
                if (cost < bestCost) { bestCost = cost; bestAxis = ax; bestSplit = b; }
            }
        }
        uint32_t mid;
        if (bestAxis < 0) {
            mid = first + count / 2;   // all centroids equal: split the range in half
        } else {
            float lo = (&cb.lo.x)[bestAxis], hi = (&cb.hi.x)[bestAxis];
            float scale = NB / (hi - lo);
            auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
                int b = (int)(((&cen[t].x)[bestAxis] - lo) * scale);
                b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                return b <= bestSplit;
            });
            mid = (uint32_t)(it - order.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        uint32_t l = (uint32_t)nodes.size();
        nodes.push_back(BvhNode{});
        nodes.push_back(BvhNode{});
        nodes[ni].left = l;
        nodes[ni].right = l + 1;
        nodes[ni].count = 0;
        build_rec(l, first, mid - first, tb, cen);
        build_rec(l + 1, mid, first + count - mid, tb, cen);
    }

    static inline bool box_test(const Aabb& b, vec3 o, vec3 id, float tmin, float tbest, float& tnear)
    {
        float x0 = (b.lo.x - o.x) * id.x, x1 = (b.hi.x - o.x) * id.x;
        float y0 = (b.lo.y - o.y) * id.y, y1 = (b.hi.y - o.y) * id.y;
        float z0 = (b.lo.z - o.z) * id.z, z1 = (b.hi.z - o.z) * id.z;
        float tn = gmax(gmax(gmin(x0, x1), gmin(y0, y1)), gmin(z0, z1));
        float tf = gmin(gmin(gmax(x0, x1), gmax(y0, y1)), gmax(z0, z1)) * 1.0000004f;
        tnear = tn;
        return tn <= tf && tn <= tbest && tf >= tmin;
    }

    Hit intersect(const std::vector<WorldTri>& tris, vec3 o, vec3 d, float tmin, float tmax, uint64_t* nn = nullptr, uint64_t* nt = nullptr) const
    {
        Hit best{tmax, 0.0f, 0.0f, 0xFFFFFFFFu};
        if (tris.empty()) return best;
        RayPre r = ray_precompute(o, d);
        vec3 id;
        for (int k = 0; k < 3; k++) {
            float c = d[k];
            if (gabs(c) < 1e-20f) c = (f2u(c) >> 31) ? -1e-20f : 1e-20f;
            (&id.x)[k] = 1.0f / c;
        }
        uint32_t stack[128];
        int sp = 0;
        stack[sp++] = 0;
        uint64_t vn = 0, vt = 0;
        while (sp) {
            const BvhNode& nd = nodes[stack[--sp]];
            float tn;
            vn++;
            if (!box_test(nd.box, o, id, tmin, best.t, tn)) continue;
            if (nd.count) {
                for (uint32_t i = nd.first; i < nd.first + nd.count; i++) {
                    uint32_t ti = order[i];
                    float t, u, v;
                    vt++;
                    if (!tri_test(r, tris[ti], tmin, t, u, v)) continue;
                    if (!(t < tmax)) continue;
                    if (closer(t, tris[ti].submesh, tris[ti].prim, best, tris)) best = Hit{t, u, v, ti};
                }
            } else {
                float tl, tr2;
                bool hl = box_test(nodes[nd.left].box, o, id, tmin, best.t, tl);
                bool hr = box_test(nodes[nd.right].box, o, id, tmin, best.t, tr2);
                if (hl && hr) {
                    if (tl <= tr2) { stack[sp++] = nd.right; stack[sp++] = nd.left; }
                    else { stack[sp++] = nd.left; stack[sp++] = nd.right; }
                } else if (hl) stack[sp++] = nd.left;
                else if (hr) stack[sp++] = nd.right;
            }
        }
        if (nn) *nn += vn;
        if (nt) *nt += vt;
        return best;
    }
};

} // namespace orc
