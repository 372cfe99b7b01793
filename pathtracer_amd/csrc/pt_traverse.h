// Closest-hit traversal of the 4-wide hierarchy, one ray per lane.
//
// Replaces traceRayEXT (PathTracer/assets/shaders/RayTracing/RayGen.glsl:124), which the
// reference leaves to the Vulkan driver and RT hardware.  Semantics: closest hit with
// tmin < t < tmax, opaque, no culling; watertight Woop-style triangle test with a
// double-precision fallback for zero edge functions; at equal t the smaller
// (submesh, primitive) wins, so the answer does not depend on traversal order.
//
// Stack: the first PT_STACK_LDS entries of each lane live in LDS (one column per lane,
// conflict free), deeper entries spill to per-lane scratch.  An entry carries the
// child's entry distance so that a popped subtree that has fallen behind the current
// hit is dropped without fetching its node.
#pragma once
#include "pt_math.h"
#include "pt_types.h"

namespace ptd {

#define PT_STACK_LDS 16
#define PT_STACK_OVF 80    // host refuses hierarchies whose depth could need more than LDS+OVF entries

struct HitRec {
    float t, u, v;
    uint32_t tri;          // index into the TriRec array, 0xFFFFFFFF = miss
    uint32_t submesh, prim;
};

struct RayPre {
    int kx, ky, kz;
    float Sx, Sy, Sz;
};

PT_DEV RayPre ray_precompute(v3 d)
{
    RayPre p;
    float ax = gabs(d.x), ay = gabs(d.y), az = gabs(d.z);
    int kz = 0;
    float m = ax;
    if (ay > m) { kz = 1; m = ay; }
    if (az > m) { kz = 2; }
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    float dkz = comp(d, kz);
    if (dkz < 0.0f) { int t = kx; kx = ky; ky = t; }
    p.kx = kx; p.ky = ky; p.kz = kz;
    p.Sx = fdiv(comp(d, kx), dkz);
    p.Sy = fdiv(comp(d, ky), dkz);
    p.Sz = fdiv(1.0f, dkz);
    return p;
}

// true when the ray hits with t > tmin; fills t,u,v (u,v = weights of vertex 1 and 2)
PT_DEV bool tri_test(const RayPre& r, v3 o, v3 v0, v3 v1, v3 v2, float tmin, float& t, float& u, float& v)
{
    v3 A = v0 - o, B = v1 - o, C = v2 - o;
    float Akz = comp(A, r.kz), Bkz = comp(B, r.kz), Ckz = comp(C, r.kz);
    float Ax = comp(A, r.kx) - r.Sx * Akz, Ay = comp(A, r.ky) - r.Sy * Akz;
    float Bx = comp(B, r.kx) - r.Sx * Bkz, By = comp(B, r.ky) - r.Sy * Bkz;
    float Cx = comp(C, r.kx) - r.Sx * Ckz, Cy = comp(C, r.ky) - r.Sy * Ckz;
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
    float tt = fdiv(T, det);
    if (!(tt > tmin)) return false;
    t = tt;
    u = fdiv(V, det);
    v = fdiv(W, det);
    return true;
}

// lds_stack points at this lane's column: entry k is lds_stack[k * stride]
template <bool STATS>
PT_DEV HitRec traverse(const Node4* __restrict__ nodes, const TriRec* __restrict__ tris, uint32_t num_tris,
                       v3 o, v3 d, float tmin, float tmax, uint2* lds_stack, int stride,
                       uint32_t& stat_nodes, uint32_t& stat_tris)
{
    HitRec best;
    best.t = tmax; best.u = 0.0f; best.v = 0.0f;
    best.tri = 0xFFFFFFFFu; best.submesh = 0xFFFFFFFFu; best.prim = 0xFFFFFFFFu;
    if (num_tris == 0) return best;

    RayPre rp = ray_precompute(d);

    // slab test set-up: t = plane * idir + ood, near/far plane picked by the direction sign
    float idx, idy, idz;
    {
        float cx = d.x, cy = d.y, cz = d.z;
        if (gabs(cx) < 1e-20f) cx = (f2u(cx) >> 31) ? -1e-20f : 1e-20f;
        if (gabs(cy) < 1e-20f) cy = (f2u(cy) >> 31) ? -1e-20f : 1e-20f;
        if (gabs(cz) < 1e-20f) cz = (f2u(cz) >> 31) ? -1e-20f : 1e-20f;
        idx = fdiv(1.0f, cx); idy = fdiv(1.0f, cy); idz = fdiv(1.0f, cz);
    }
    float oix = o.x * idx, oiy = o.y * idy, oiz = o.z * idz;
    // rounding of o*idir moves the planes by at most |o*idir| * 2^-24; widen by 4x that
    float slack = fmaxf(fmaxf(gabs(oix), gabs(oiy)), gabs(oiz)) * 0x1p-22f;
    float onx = -oix - slack, ony = -oiy - slack, onz = -oiz - slack;   // near planes
    float ofx = -oix + slack, ofy = -oiy + slack, ofz = -oiz + slack;   // far planes
    // float4 index of the near / far plane arrays inside a node (lo = 0..2, hi = 3..5)
    const int nx = idx < 0.0f ? 3 : 0, ny = idy < 0.0f ? 4 : 1, nz = idz < 0.0f ? 5 : 2;
    const int fx = 3 - nx, fy = 5 - ny, fz = 7 - nz;

    uint2 ovf[PT_STACK_OVF];
    int sp = 0;
    uint32_t cur = 0;

    for (;;) {
        if (!(cur & PT_CHILD_LEAF)) {
            const float4* np = reinterpret_cast<const float4*>(nodes + cur);
            float4 bnx = np[nx], bny = np[ny], bnz = np[nz];
            float4 bfx = np[fx], bfy = np[fy], bfz = np[fz];
            uint4 ch = reinterpret_cast<const uint4*>(np)[6];
            if (STATS) stat_nodes++;

            float tn0 = fmaxf(fmaxf(fmaf(bnx.x, idx, onx), fmaf(bny.x, idy, ony)), fmaxf(fmaf(bnz.x, idz, onz), tmin));
            float tn1 = fmaxf(fmaxf(fmaf(bnx.y, idx, onx), fmaf(bny.y, idy, ony)), fmaxf(fmaf(bnz.y, idz, onz), tmin));
            float tn2 = fmaxf(fmaxf(fmaf(bnx.z, idx, onx), fmaf(bny.z, idy, ony)), fmaxf(fmaf(bnz.z, idz, onz), tmin));
            float tn3 = fmaxf(fmaxf(fmaf(bnx.w, idx, onx), fmaf(bny.w, idy, ony)), fmaxf(fmaf(bnz.w, idz, onz), tmin));
            float tf0 = fminf(fminf(fmaf(bfx.x, idx, ofx), fmaf(bfy.x, idy, ofy)), fmaf(bfz.x, idz, ofz)) * 1.0000005f;
            float tf1 = fminf(fminf(fmaf(bfx.y, idx, ofx), fmaf(bfy.y, idy, ofy)), fmaf(bfz.y, idz, ofz)) * 1.0000005f;
            float tf2 = fminf(fminf(fmaf(bfx.z, idx, ofx), fmaf(bfy.z, idy, ofy)), fmaf(bfz.z, idz, ofz)) * 1.0000005f;
            float tf3 = fminf(fminf(fmaf(bfx.w, idx, ofx), fmaf(bfy.w, idy, ofy)), fmaf(bfz.w, idz, ofz)) * 1.0000005f;
            const float INF = __builtin_inff();
            // a child is entered when its slab interval is non-empty and starts before the best hit
            float k0 = (tn0 <= tf0 && tn0 <= best.t && ch.x != PT_CHILD_EMPTY) ? tn0 : INF;
            float k1 = (tn1 <= tf1 && tn1 <= best.t && ch.y != PT_CHILD_EMPTY) ? tn1 : INF;
            float k2 = (tn2 <= tf2 && tn2 <= best.t && ch.z != PT_CHILD_EMPTY) ? tn2 : INF;
            float k3 = (tn3 <= tf3 && tn3 <= best.t && ch.w != PT_CHILD_EMPTY) ? tn3 : INF;
            uint32_t c0 = ch.x, c1 = ch.y, c2 = ch.z, c3 = ch.w;
#define PT_CSWAP(ka, ca, kb, cb) { if (kb < ka) { float tk = ka; ka = kb; kb = tk; uint32_t tc = ca; ca = cb; cb = tc; } }
            PT_CSWAP(k0, c0, k1, c1)
            PT_CSWAP(k2, c2, k3, c3)
            PT_CSWAP(k0, c0, k2, c2)
            PT_CSWAP(k1, c1, k3, c3)
            PT_CSWAP(k1, c1, k2, c2)
#undef PT_CSWAP
            // farthest first, so the nearest is popped (or followed) first
            if (k3 < INF) { uint2 e = make_uint2(c3, f2u(k3)); if (sp < PT_STACK_LDS) lds_stack[sp * stride] = e; else ovf[sp - PT_STACK_LDS] = e; sp++; }
            if (k2 < INF) { uint2 e = make_uint2(c2, f2u(k2)); if (sp < PT_STACK_LDS) lds_stack[sp * stride] = e; else ovf[sp - PT_STACK_LDS] = e; sp++; }
            if (k1 < INF) { uint2 e = make_uint2(c1, f2u(k1)); if (sp < PT_STACK_LDS) lds_stack[sp * stride] = e; else ovf[sp - PT_STACK_LDS] = e; sp++; }
            if (k0 < INF) { cur = c0; continue; }
        } else {
            uint32_t first = cur & 0x0FFFFFFFu;
            uint32_t cnt = ((cur >> 28) & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; i++) {
                const float4* tp = reinterpret_cast<const float4*>(tris + first + i);
                float4 a = tp[0], b = tp[1], c = tp[2];
                if (STATS) stat_tris++;
                float t, u, v;
                if (!tri_test(rp, o, V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), V3(c.x, c.y, c.z), tmin, t, u, v)) continue;
                if (!(t < tmax)) continue;
                uint32_t sm = f2u(a.w), pr = f2u(b.w);
                bool take = t < best.t ||
                            (t == best.t && best.tri != 0xFFFFFFFFu && (sm < best.submesh || (sm == best.submesh && pr < best.prim)));
                if (take) {
                    best.t = t; best.u = u; best.v = v;
                    best.tri = first + i; best.submesh = sm; best.prim = pr;
                }
            }
        }
        // pop, dropping subtrees that start behind the current best hit
        bool found = false;
        while (sp > 0) {
            sp--;
            uint2 e = (sp < PT_STACK_LDS) ? lds_stack[sp * stride] : ovf[sp - PT_STACK_LDS];
            if (u2f(e.y) <= best.t) { cur = e.x; found = true; break; }
        }
        if (!found) break;
    }
    return best;
}

} // namespace ptd
