// Closest-hit traversal of the 8-wide quantised hierarchy, one ray per lane, persistent waves.
//
// Replaces traceRayEXT (PathTracer/assets/shaders/RayTracing/RayGen.glsl:124), which the
// reference leaves to the Vulkan driver and RT hardware.  Semantics: closest hit with
// tmin < t < tmax, opaque, no culling; watertight Woop-style triangle test with a
// double-precision fallback for zero edge functions; at equal t the smaller
// (submesh, primitive) wins, so the answer does not depend on traversal order.
//
// Execution: a wave keeps 64 traversals in flight.  Rays differ widely in length, so a wave that
// waited for its longest ray would idle most lanes; instead, whenever PT_REFILL_MIN lanes are free
// the wave takes that many new rays from its queue with one atomic and carries on (dynamic ray
// fetch).  Node steps and triangle tests are separate phases.  A lane that reaches a leaf parks it
// in a pending slot and walks on through its stack; once PT_LEAF_MIN lanes have a pending leaf
// (or nobody has a node left) the wave runs the triangle phase, in which the pending triangles of
// all lanes are dealt out one per lane, whichever lane owns the ray: leaves hold 1..4 triangles
// and only some lanes have one, so testing them in the owning lane would run the triangle code
// mostly empty.  The ray travels to the testing lane, and the result back, through lane permutes.
//
// Tail: when the queue has run dry a wave soon walks a handful of long rays with most lanes idle, and the
// launch lasts until the last of them is done.  Such a wave instead suspends those rays (where the caller allows
// it): best hit so far, position and stack go to a record in global memory, and the next launch picks the ray up
// among its first rays, in a full wave, and carries on where it stopped.  The closest hit does not depend on when
// or by whom a ray is traced, so results are unchanged.
//
// Stack: the top PT_STACK_LDS entries of each lane live in an LDS ring (one column per lane,
// conflict free); deeper entries move in blocks to a per-wave spill area in global memory, laid
// out [entry][lane] so that a wave's spills coalesce.  The kernel itself uses no scratch memory:
// a private spill array would reserve tens of KB of scratch per wave and cap the number of
// resident waves.  An entry carries the child's entry distance so that a popped subtree that
// has fallen behind the current hit is dropped without fetching its node.
#pragma once
#include "pt_math.h"
#include "pt_types.h"

namespace ptd {

#define PT_STACK_LDS 16     // ring in LDS, power of two
#define PT_STACK_BLK 8      // entries moved per spill / restore
#define PT_STACK_MAX 1024   // host refuses hierarchies whose depth could need more entries than this (146 levels)
#ifndef PT_REFILL_MIN
#define PT_REFILL_MIN 20   // free lanes that trigger a refill
#endif
#ifndef PT_LEAF_MIN
#define PT_LEAF_MIN 16     // lanes with a pending leaf that trigger the triangle phase while others still walk nodes
#endif
#ifndef PT_CUT_LANES
#define PT_CUT_LANES 16    // a wave whose queue ran dry suspends its rays once no more than this many are in flight
#endif
#define PT_RAY_CUTTABLE 1u // fetch flags: the ray may be suspended; the ray resumes a suspended traversal
#define PT_RAY_RESUME 2u
#define PT_SUSPEND_STATE_WORDS 16   // a suspended traversal is 16 words of state followed by the stack (2 words per entry)
#define PT_CUR_NONE 0x7FFFFFFFu   // s.cur when the stack ran empty; node indices are below, leaf references above
// LDS scratch per wave, used by one phase at a time: in a node step the node's eight child references, [slot][lane]
// (a lane picks the reference of a sorted child with one LDS read instead of a tree of selects over eight registers);
// in the triangle phase the work list (one uint16 per triangle, 64 * PT_LEAF_MAX of them)
#define PT_SCRATCH_WORDS (8 * 64)
static_assert(64 * PT_LEAF_MAX * 2 + 64 * 8 <= PT_SCRATCH_WORDS * 4, "work list and result slots of the triangle phase share the scratch");

struct HitRec {
    float t, u, v;
    uint32_t tri;          // index into the TriRec array, 0xFFFFFFFF = miss
};

// what one wave did during a launch (statistics builds only); times are wall_clock64() ticks (10 ns)
struct WaveStats {
    uint32_t node_phases = 0, node_lanes = 0;      // node phases run, lanes active in them
    uint32_t leaf_phases = 0, leaf_lanes = 0;
    uint32_t refills = 0;
    uint32_t idle_lanes = 0;                       // lanes without a ray during node phases (the others not on a node wait with a leaf)
    uint32_t node_hist[4] = {0, 0, 0, 0};          // node phases with <= 8, 9..16, 17..32, 33..64 active lanes
    uint32_t longest_ray = 0;                      // loop iterations the longest-lived ray of this wave stayed in flight
    unsigned long long t_start = 0, t_exhausted = 0, t_end = 0;   // t_exhausted: the queue ran dry, only own rays left
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

// true when the ray hits with t > tmin; fills t and V, W, det: the barycentric weights of vertex 1 and 2
// are V / det and W / det, divided out only for the hit that is finally reported
PT_DEV bool tri_test(const RayPre& r, v3 o, v3 v0, v3 v1, v3 v2, float tmin, float& t, float& oV, float& oW, float& odet)
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
    oV = V; oW = W; odet = det;
    return true;
}

// per-lane traversal state
struct Trav {
    v3 o;
    float tmin, tmax;
    RayPre rp;
    float idx, idy, idz;           // 1 / direction (zero components replaced by +-1e-20)
    float onx, ony, onz;           // -(o * idir) - slack : near planes
    float ofx, ofy, ofz;           // -(o * idir) + slack : far planes
    bool negx, negy, negz;         // direction signs: pick which quantised plane of a child box is the near one
    HitRec best;
    float bV, bW, bdet;            // best.u = bV / bdet, best.v = bW / bdet, divided when the ray is finished
    uint32_t cur;                  // node index, leaf reference (bit 31), or PT_CUR_NONE
    float cur_key;                 // entry distance of cur
    uint32_t pend;                 // leaf reference waiting for the triangle phase, 0 = none
    int sp, base;                  // stack depth; entries below base are in the spill area
};

PT_DEV void trav_init(Trav& s, v3 o, v3 d, float tmin, float tmax)
{
    s.o = o;
    s.tmin = tmin;
    s.tmax = tmax;
    s.rp = ray_precompute(d);
    float cx = d.x, cy = d.y, cz = d.z;
    if (gabs(cx) < 1e-20f) cx = (f2u(cx) >> 31) ? -1e-20f : 1e-20f;
    if (gabs(cy) < 1e-20f) cy = (f2u(cy) >> 31) ? -1e-20f : 1e-20f;
    if (gabs(cz) < 1e-20f) cz = (f2u(cz) >> 31) ? -1e-20f : 1e-20f;
    s.idx = fdiv(1.0f, cx); s.idy = fdiv(1.0f, cy); s.idz = fdiv(1.0f, cz);
    float oix = o.x * s.idx, oiy = o.y * s.idy, oiz = o.z * s.idz;
    // rounding of o*idir moves an axis' planes by at most |o*idir| * 2^-24 of that axis; widen by 4x that, per axis:
    // a slack shared by the axes would open the other two slabs of a ray that runs almost parallel to a coordinate
    // plane (|idir| of one axis huge) so far that no box is culled, and such a ray walks a large part of the hierarchy
    float slx = gabs(oix) * 0x1p-22f, sly = gabs(oiy) * 0x1p-22f, slz = gabs(oiz) * 0x1p-22f;
    s.onx = -oix - slx; s.ony = -oiy - sly; s.onz = -oiz - slz;
    s.ofx = -oix + slx; s.ofy = -oiy + sly; s.ofz = -oiz + slz;
    s.negx = s.idx < 0.0f;
    s.negy = s.idy < 0.0f;
    s.negz = s.idz < 0.0f;
    s.best.t = tmax; s.best.u = 0.0f; s.best.v = 0.0f;
    s.best.tri = 0xFFFFFFFFu;
    s.bV = 0.0f; s.bW = 0.0f; s.bdet = 1.0f;
    s.cur = 0;
    s.cur_key = tmin;
    s.pend = 0;
    s.sp = 0;
    s.base = 0;
}

// The LDS part of the stack is a ring of PT_STACK_LDS entries holding the top of the logical
// stack: entries [s.base, s.sp) live in ring slot (k % PT_STACK_LDS), entries [0, s.base) in the
// global spill area.  Blocks of PT_STACK_BLK entries move between the two, so pushes and pops
// themselves are plain LDS accesses.

// makes room for `room` pushes (room <= PT_STACK_LDS - PT_STACK_BLK)
PT_DEV void stack_reserve(Trav& s, const uint2* lds_stack, int stride, uint2* ovf, int room)
{
    if (s.sp - s.base + room > PT_STACK_LDS) {
#pragma unroll
        for (int i = 0; i < PT_STACK_BLK; i++) ovf[(s.base + i) * 64] = lds_stack[((s.base + i) & (PT_STACK_LDS - 1)) * stride];
        s.base += PT_STACK_BLK;
    }
}

#define PT_PUSH(ref, key)                                                                   \
    {                                                                                       \
        lds_stack[(s.sp & (PT_STACK_LDS - 1)) * stride] = make_uint2(ref, f2u(key));        \
        s.sp++;                                                                             \
    }

// pops until an entry that can still hold a closer hit; false (and cur = PT_CUR_NONE) when the stack is empty
PT_DEV bool trav_pop(Trav& s, uint2* lds_stack, int stride, const uint2* ovf)
{
    while (s.sp > 0) {
        if (s.sp == s.base) {
            s.base -= PT_STACK_BLK;
#pragma unroll
            for (int i = 0; i < PT_STACK_BLK; i++) lds_stack[((s.base + i) & (PT_STACK_LDS - 1)) * stride] = ovf[(s.base + i) * 64];
        }
        s.sp--;
        uint2 e = lds_stack[(s.sp & (PT_STACK_LDS - 1)) * stride];
        if (u2f(e.y) <= s.best.t) { s.cur = e.x; s.cur_key = u2f(e.y); return true; }
    }
    s.cur = PT_CUR_NONE;
    return false;
}

PT_DEV float ubyte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 255u); }

#define PT_KEY_MISS 0x7f800000u

// one internal node: decodes the eight quantised child boxes, slab-tests them, follows the nearest
// hit child and stacks the others, farthest first.  A sort key is the entry distance with the
// child's slot number in its three lowest mantissa bits, so the eight candidates are ordered with a
// 19-comparator network of integer min/max.
PT_DEV bool node_step(Trav& s, const Node8* __restrict__ nodes, uint2* lds_stack, int stride, uint2* ovf, uint32_t* child_row)
{
    const uint4* np = reinterpret_cast<const uint4*>(nodes + s.cur);
    uint4 hdr = np[0], c03 = np[1], c47 = np[2], qa = np[3], qb = np[4], qc = np[5];
    // child_row points at this lane's column of the wave's scratch: reference of slot k at child_row[k * 64]
    child_row[0 * 64] = c03.x; child_row[1 * 64] = c03.y; child_row[2 * 64] = c03.z; child_row[3 * 64] = c03.w;
    child_row[4 * 64] = c47.x; child_row[5 * 64] = c47.y; child_row[6 * 64] = c47.z; child_row[7 * 64] = c47.w;
    // plane = origin + q * step  =>  t = q * (step * idir) + (origin * idir - o * idir -+ slack)
    float ox = u2f(hdr.x), oy = u2f(hdr.y), oz = u2f(hdr.z);
    float ax = u2f((hdr.w & 255u) << 23) * s.idx;
    float ay = u2f(((hdr.w >> 8) & 255u) << 23) * s.idy;
    float az = u2f(((hdr.w >> 16) & 255u) << 23) * s.idz;
    float bnx = fmaf(ox, s.idx, s.onx), bny = fmaf(oy, s.idy, s.ony), bnz = fmaf(oz, s.idz, s.onz);
    float bfx = fmaf(ox, s.idx, s.ofx), bfy = fmaf(oy, s.idy, s.ofy), bfz = fmaf(oz, s.idz, s.ofz);
    // q words: qa = lo.x[0..7] lo.y[0..7], qb = lo.z[0..7] hi.x[0..7], qc = hi.y[0..7] hi.z[0..7]
    uint32_t nxw[2] = {s.negx ? qb.z : qa.x, s.negx ? qb.w : qa.y}, fxw[2] = {s.negx ? qa.x : qb.z, s.negx ? qa.y : qb.w};
    uint32_t nyw[2] = {s.negy ? qc.x : qa.z, s.negy ? qc.y : qa.w}, fyw[2] = {s.negy ? qa.z : qc.x, s.negy ? qa.w : qc.y};
    uint32_t nzw[2] = {s.negz ? qc.z : qb.x, s.negz ? qc.w : qb.y}, fzw[2] = {s.negz ? qb.x : qc.z, s.negz ? qb.y : qc.w};
    const float bt = s.best.t, tmin = s.tmin;
    uint32_t key[8];
    // two children per packed fma
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 vax = {ax, ax}, vay = {ay, ay}, vaz = {az, az};
    const f2 vbnx = {bnx, bnx}, vbny = {bny, bny}, vbnz = {bnz, bnz}, vbfx = {bfx, bfx}, vbfy = {bfy, bfy}, vbfz = {bfz, bfz};
    const f2 widen = {1.0000005f, 1.0000005f};
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const int w = k >> 2, b = k & 3;
        f2 qnx = {ubyte_f(nxw[w], b), ubyte_f(nxw[w], b + 1)}, qny = {ubyte_f(nyw[w], b), ubyte_f(nyw[w], b + 1)}, qnz = {ubyte_f(nzw[w], b), ubyte_f(nzw[w], b + 1)};
        f2 qfx = {ubyte_f(fxw[w], b), ubyte_f(fxw[w], b + 1)}, qfy = {ubyte_f(fyw[w], b), ubyte_f(fyw[w], b + 1)}, qfz = {ubyte_f(fzw[w], b), ubyte_f(fzw[w], b + 1)};
        f2 tnx = __builtin_elementwise_fma(qnx, vax, vbnx), tny = __builtin_elementwise_fma(qny, vay, vbny), tnz = __builtin_elementwise_fma(qnz, vaz, vbnz);
        f2 tfx = __builtin_elementwise_fma(qfx, vax, vbfx), tfy = __builtin_elementwise_fma(qfy, vay, vbfy), tfz = __builtin_elementwise_fma(qfz, vaz, vbfz);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float tn = fmaxf(fmaxf(tnx[h], tny[h]), fmaxf(tnz[h], tmin));
            float tf = fminf(fminf(tfx[h], tfy[h]), tfz[h]) * widen[h];
            // entered when the slab interval is non-empty and starts before the best hit (empty children decode to tn > tf)
            bool hit = tn <= fminf(tf, bt);
            key[k + h] = hit ? ((f2u(tn) & 0x7ffffff8u) | (uint32_t)(k + h)) : (PT_KEY_MISS | (uint32_t)(k + h));
        }
    }
#define PT_CE(i, j) { uint32_t lo_ = min(key[i], key[j]); key[j] = max(key[i], key[j]); key[i] = lo_; }
    PT_CE(0, 1) PT_CE(2, 3) PT_CE(4, 5) PT_CE(6, 7)
    PT_CE(0, 2) PT_CE(1, 3) PT_CE(4, 6) PT_CE(5, 7)
    PT_CE(1, 2) PT_CE(5, 6)
    PT_CE(0, 4) PT_CE(1, 5) PT_CE(2, 6) PT_CE(3, 7)
    PT_CE(2, 4) PT_CE(3, 5)
    PT_CE(1, 2) PT_CE(3, 4) PT_CE(5, 6)
#undef PT_CE
    // farthest first, so the nearest is popped (or followed) first
    stack_reserve(s, lds_stack, stride, ovf, 7);
#pragma unroll
    for (int j = 7; j >= 1; j--) {
        if (key[j] < PT_KEY_MISS) PT_PUSH(child_row[(key[j] & 7u) * 64], u2f(key[j] & 0x7ffffff8u))
    }
    if (key[0] < PT_KEY_MISS) { s.cur = child_row[(key[0] & 7u) * 64]; s.cur_key = u2f(key[0] & 0x7ffffff8u); return true; }
    return trav_pop(s, lds_stack, stride, ovf);
}

// suspended traversal: [0] best.t [1] best.tri [2] bV [3] bW [4] bdet [5] cur [6] cur_key [7] pend [8] sp, then sp stack entries
PT_DEV void trav_save(const Trav& s, uint32_t* rec, const uint2* lds_stack, int stride, const uint2* ovf)
{
    rec[0] = f2u(s.best.t); rec[1] = s.best.tri; rec[2] = f2u(s.bV); rec[3] = f2u(s.bW); rec[4] = f2u(s.bdet);
    rec[5] = s.cur; rec[6] = f2u(s.cur_key); rec[7] = s.pend; rec[8] = (uint32_t)s.sp;
    uint2* st = reinterpret_cast<uint2*>(rec + PT_SUSPEND_STATE_WORDS);
    for (int k = 0; k < s.base; k++) st[k] = ovf[k * 64];
    for (int k = s.base; k < s.sp; k++) st[k] = lds_stack[(k & (PT_STACK_LDS - 1)) * stride];
}

// call after trav_init of the same ray
PT_DEV void trav_restore(Trav& s, const uint32_t* rec, uint2* lds_stack, int stride, uint2* ovf)
{
    s.best.t = u2f(rec[0]); s.best.tri = rec[1]; s.bV = u2f(rec[2]); s.bW = u2f(rec[3]); s.bdet = u2f(rec[4]);
    s.cur = rec[5]; s.cur_key = u2f(rec[6]); s.pend = rec[7]; s.sp = (int)rec[8];
    // the top of the stack goes to the LDS ring, whole blocks below it to the spill column
    s.base = s.sp > PT_STACK_LDS ? ((s.sp - PT_STACK_BLK - 1) & ~(PT_STACK_BLK - 1)) : 0;
    const uint2* st = reinterpret_cast<const uint2*>(rec + PT_SUSPEND_STATE_WORDS);
    for (int k = 0; k < s.base; k++) ovf[k * 64] = st[k];
    for (int k = s.base; k < s.sp; k++) lds_stack[(k & (PT_STACK_LDS - 1)) * stride] = st[k];
}

// lanes below this one among the set bits of mask
PT_DEV uint32_t lanes_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

PT_DEV float lane_read(float v, uint32_t src_lane) { return u2f((uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)f2u(v))); }
PT_DEV uint32_t lane_read(uint32_t v, uint32_t src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }

// at equal t the smaller (submesh, primitive) wins; both records are read only in that rare case
PT_DEV bool wins_tie(const TriRec* __restrict__ tris, uint32_t cand, uint32_t best)
{
    uint32_t cs = tris[cand].submesh, cp = tris[cand].prim, bs = tris[best].submesh, bp = tris[best].prim;
    return cs < bs || (cs == bs && cp < bp);
}

// Triangle phase, run by the whole wave.  Every lane with a pending leaf owns 1..PT_LEAF_MAX work items (its
// triangles); the items are numbered through the wave in lane order and item j is tested by lane j, with the
// owner's ray fetched by lane permutes.  Every hit is sent to its owner's LDS slot with a 64-bit minimum on (distance,
// item); the owner compares the winner with its closest hit so far under the usual rules and fetches its barycentric terms.  One call tests at most 64 items: lanes whose
// items do not fit keep their leaf pending for the next call (returns whether this lane's leaf was tested).
// `owner` is this wave's LDS work list (the wave's scratch).  Must be called in wave-uniform control flow.
template <bool STATS>
PT_DEV bool leaf_phase(Trav& s, bool has_pend, const TriRec* __restrict__ tris, uint16_t* owner, uint32_t lane, uint32_t& stat_tris)
{
    const uint32_t first = s.pend & 0x0FFFFFFFu;
    uint32_t cnt = has_pend ? ((s.pend >> 28) & 7u) + 1u : 0u;
    // exclusive prefix sum of cnt (0..PT_LEAF_MAX) over the lanes, from one ballot per bit
    unsigned long long b0 = __ballot((cnt & 1u) != 0), b1 = __ballot((cnt & 2u) != 0), b2 = __ballot((cnt & 4u) != 0);
    uint32_t off = lanes_below(b0) + 2u * lanes_below(b1) + 4u * lanes_below(b2);
    if (PT_LEAF_MAX >= 8u) {
        unsigned long long b3 = __ballot((cnt & 8u) != 0);
        off += 8u * lanes_below(b3);
    }
    const bool mine = has_pend && off + cnt <= 64u;       // the lanes taken are a prefix of the pending lanes
    if (!mine) cnt = 0u;
#pragma unroll
    for (uint32_t k = 0; k < PT_LEAF_MAX; k++)
        if (k < cnt) owner[off + k] = (uint16_t)(lane | (k << 6));
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)(off + cnt), 63 - __builtin_clzll(__ballot(mine) | 1ull));
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(owner + 64 * PT_LEAF_MAX);      // one per lane, after the work list
    if (cnt != 0u) slot[lane] = ~0ull;
    __syncthreads();
    const uint32_t kpack = (uint32_t)s.rp.kx | ((uint32_t)s.rp.ky << 2) | ((uint32_t)s.rp.kz << 4);
    // ---- test item `lane`
    const bool work = lane < total;
    const uint32_t ow = work ? owner[lane] : lane;
    const uint32_t src = ow & 63u, k = ow >> 6;
    const uint32_t w_first = lane_read(first, src);
    v3 wo = V3(lane_read(s.o.x, src), lane_read(s.o.y, src), lane_read(s.o.z, src));
    RayPre wr;
    wr.Sx = lane_read(s.rp.Sx, src); wr.Sy = lane_read(s.rp.Sy, src); wr.Sz = lane_read(s.rp.Sz, src);
    const uint32_t wk = lane_read(kpack, src);
    wr.kx = (int)(wk & 3u); wr.ky = (int)((wk >> 2) & 3u); wr.kz = (int)(wk >> 4);
    const float w_tmin = lane_read(s.tmin, src);
    float wt = __builtin_inff(), wV = 0.0f, wW = 0.0f, wdet = 1.0f;
    if (work) {
        const float4* tp = reinterpret_cast<const float4*>(tris + w_first + k);
        float4 a = tp[0], b = tp[1], c = tp[2];
        if (STATS) stat_tris++;
        float t, V, W, det;
        if (tri_test(wr, wo, V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), V3(c.x, c.y, c.z), w_tmin, t, V, W, det)) { wt = t; wV = V; wW = W; wdet = det; }
    }
    // ---- every hit goes to its owner's slot: the smallest (distance, item) wins.  Distances are positive, so their bit
    // patterns order like the numbers; the items of a leaf follow each other in the order of their (submesh, primitive)
    // (the builder writes leaves that way), so the item number settles hits at equal distance inside the leaf
    if (work && wt < __builtin_inff()) atomicMin(&slot[src], ((unsigned long long)f2u(wt) << 32) | lane);
    __syncthreads();
    uint32_t win = lane;
    bool won = false;
    if (cnt != 0u) {
        const unsigned long long got = slot[lane];
        const float tk = u2f((uint32_t)(got >> 32));
        const uint32_t jj = (uint32_t)got & 63u;
        if (got != ~0ull && tk < s.tmax) {
            bool take = tk < s.best.t;
            if (tk == s.best.t && s.best.tri != 0xFFFFFFFFu) take = wins_tie(tris, first + (jj - off), s.best.tri);
            if (take) { s.best.t = tk; s.best.tri = first + (jj - off); win = jj; won = true; }
        }
    }
    const float gV = lane_read(wV, win), gW = lane_read(wW, win), gdet = lane_read(wdet, win);
    if (won) { s.bV = gV; s.bW = gW; s.bdet = gdet; }
    __syncthreads();
    return mine;
}

// The barycentric weights of the reported hit are best.u = bV / bdet and best.v = bW / bdet.  The traversal does not
// divide: a finished ray hands (t, bV, bW, bdet, triangle) to emit(), and whoever consumes the hit divides, at full
// lane occupancy (in the traversal loop a few lanes finish in almost every iteration, so the two correctly rounded
// divisions would be issued for a handful of lanes each time).
PT_DEV void finish_hit(Trav& s)
{
    if (s.best.tri != 0xFFFFFFFFu) {
        s.best.u = fdiv(s.bV, s.bdet);
        s.best.v = fdiv(s.bW, s.bdet);
    }
}

// Persistent-wave driver.  fetch(i, o, d, tmin, tmax) loads ray i of the queue and returns PT_RAY_* flags;
// emit(i, traversal) stores its result (see finish_hit); suspend(i, record) hands ray i on with its traversal saved (see PT_CUT_LANES),
// resume_record(i) is where a ray flagged PT_RAY_RESUME finds the traversal it was suspended with.
// its result; *cursor hands out queue indices (must be 0 at launch), n = queue length.
// lds_stack points at this lane's column: entry k is lds_stack[k * stride]; ovf points at this lane's
// column of the wave's global spill area: entry k is ovf[k * 64]; scratch is the wave's PT_SCRATCH_WORDS words of LDS.
// iter_cap bounds the loop (PT_ITER_CAP_DEFAULT; never reached by a well-formed hierarchy, where a ray takes a few hundred
// steps at most): a wave that reaches it leaves, with the hit records of its unfinished and unfetched rays unwritten, and sets
// *fault, which the host turns into PT_ERR_TRAVERSAL at its next synchronisation.
#define PT_ITER_CAP_DEFAULT (1u << 24)
template <bool STATS, bool CUT, class Fetch, class Emit, class Suspend, class ResumeRecord>
PT_DEV void trace_wave(const Node8* __restrict__ nodes, const TriRec* __restrict__ tris, uint32_t num_tris, uint32_t n,
                       uint32_t* cursor, uint2* lds_stack, int stride, uint2* ovf, uint32_t* scratch, Fetch fetch, Emit emit, Suspend suspend, ResumeRecord resume_record,
                       uint32_t& stat_nodes, uint32_t& stat_tris, uint32_t iter_cap, uint32_t* fault, WaveStats* wave_stats = nullptr)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint16_t* owner = reinterpret_cast<uint16_t*>(scratch);
    uint32_t* child_row = scratch + lane;
    Trav s;
    bool has = false;
    bool exhausted = false;      // wave-uniform
    uint32_t my = 0;
    bool cuttable = false;
    s.cur = PT_CUR_NONE;
    s.pend = 0;
    unsigned long long t_start = 0, t_exh = 0;
    uint32_t ray_steps = 0, ray_steps_max = 0;
    if (STATS && wave_stats) t_start = wall_clock64();
    // the cap only guarantees that every wave leaves the loop whatever the data
    uint32_t iter = 0;
    for (; iter < iter_cap; iter++) {
        unsigned long long idle = __ballot(!has);
        uint32_t nidle = (uint32_t)__popcll(idle);
        if (!exhausted && nidle >= PT_REFILL_MIN) {
            if (STATS && wave_stats && lane == 0) wave_stats->refills++;
            uint32_t base = 0;
            if (lane == (uint32_t)(__ffsll((long long)idle) - 1)) base = atomicAdd(cursor, nidle);
            base = __shfl(base, __ffsll((long long)idle) - 1);
            if (!has) {
                uint32_t i = base + lanes_below(idle);
                if (i < n) {
                    v3 o, d;
                    float tmin, tmax;
                    const uint32_t flags = fetch(i, o, d, tmin, tmax);
                    cuttable = (flags & PT_RAY_CUTTABLE) != 0;
                    trav_init(s, o, d, tmin, tmax);
                    if (CUT && (flags & PT_RAY_RESUME)) trav_restore(s, resume_record(i), lds_stack, stride, ovf);
                    my = i;
                    ray_steps = 0;
                    if (num_tris == 0) emit(my, s);
                    else has = true;
                }
            }
            if (base + nidle >= n) {
                exhausted = true;
                if (STATS && wave_stats) t_exh = wall_clock64();
            }
        }
        unsigned long long hasm = __ballot(has);
        if (CUT && exhausted && hasm != 0 && (uint32_t)__popcll(hasm) <= PT_CUT_LANES) {
            // the tail of a launch: hand the last few rays back instead of walking them with a nearly empty wave
            if (has && cuttable) { suspend(my, s, lds_stack, stride, ovf); has = false; }
            hasm = __ballot(has);
        }
        if (hasm == 0) {
            if (exhausted) break;
            continue;
        }
        // node phase: every lane standing on an internal node takes one step
        bool at_node = has && s.cur < PT_CUR_NONE;
        if (STATS && wave_stats) {
            unsigned long long nm = __ballot(at_node), im = __ballot(!has);
            if (lane == 0 && nm) {
                uint32_t c = (uint32_t)__popcll(nm);
                wave_stats->node_phases++; wave_stats->node_lanes += c;
                wave_stats->idle_lanes += (uint32_t)__popcll(im);
                wave_stats->node_hist[c <= 8 ? 0 : c <= 16 ? 1 : c <= 32 ? 2 : 3]++;
            }
        }
        if (STATS && has) { ray_steps++; ray_steps_max = max(ray_steps_max, ray_steps); }
        if (at_node) {
            if (STATS) stat_nodes++;
            node_step(s, nodes, lds_stack, stride, ovf, child_row);
        }
        // a leaf goes to the pending slot and the lane walks on; with the slot taken the lane waits on the leaf
        if (has && s.cur > PT_CUR_NONE && s.pend == 0) {
            s.pend = s.cur;
            trav_pop(s, lds_stack, stride, ovf);
        }
        if (has && s.cur == PT_CUR_NONE && s.pend == 0) { emit(my, s); has = false; }
        // triangle phase: once enough lanes hold a pending leaf, or nobody is left on a node
        bool has_pend = has && s.pend != 0;
        unsigned long long pendm = __ballot(has_pend);
        unsigned long long nodem = __ballot(has && s.cur < PT_CUR_NONE);
        if (pendm != 0 && ((uint32_t)__popcll(pendm) >= PT_LEAF_MIN || nodem == 0)) {
            if (STATS && wave_stats && lane == 0) { wave_stats->leaf_phases++; wave_stats->leaf_lanes += (uint32_t)__popcll(pendm); }
            if (leaf_phase<STATS>(s, has_pend, tris, owner, lane, stat_tris)) {
                s.pend = 0;
                // the hit may have come closer than the subtree this lane went on to meanwhile
                if (s.cur != PT_CUR_NONE && s.cur_key > s.best.t) trav_pop(s, lds_stack, stride, ovf);
                if (s.cur > PT_CUR_NONE) {
                    s.pend = s.cur;
                    trav_pop(s, lds_stack, stride, ovf);
                }
                if (s.cur == PT_CUR_NONE && s.pend == 0) { emit(my, s); has = false; }
            }
        }
    }
    // the loop was cut short (wave-uniform): only the flag is raised.  Writing the hit records the wave still owes at this point
    // (its rays' closest hits so far, misses for the rest of the queue) kept ray state alive behind the loop and cost the loop
    // itself 1 % (2.7 % in the variant that can suspend rays); the records stay as they were, the shading stage takes a
    // triangle number beyond the table for a miss, and the host declares everything since its last synchronisation invalid
    if (iter >= iter_cap && lane == 0) __hip_atomic_fetch_or(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (STATS && wave_stats) {
        unsigned long long t_end = wall_clock64();
        wave_stats->t_start = t_start;
        wave_stats->t_exhausted = t_exh ? t_exh : t_end;
        wave_stats->t_end = t_end;
        for (int o = 32; o > 0; o >>= 1) ray_steps_max = max(ray_steps_max, (uint32_t)__shfl_xor((int)ray_steps_max, o));
        wave_stats->longest_ray = ray_steps_max;
    }
}

#undef PT_PUSH

} // namespace ptd
