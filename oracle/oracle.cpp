// TEST INFRASTRUCTURE (CPU oracle) -- not part of the shipped library.
// PARITY UNPINNED: see oracle/shaders.h.
//
// Scene tables, ClosestHit / Miss / RayGen restatement and the C entry points.
//   ClosestHit.glsl:50-226  closest_hit()      Miss.glsl:8-29   (Distance = -1)
//   RayGen.glsl:106-176     trace_path()       RayGen.glsl:312-373  render_pixel()
//   RayTracingLayer.h:17-33 CameraBuffer / SceneBuffer layouts
#include "oracle.h"
#include "intersect.h"
#include "shaders.h"
#include "passes.h"
#include <atomic>
#include <thread>

namespace orc {

struct Texture {
    uint32_t w, h, flags;
    std::vector<uint8_t> px;
};

struct Params {
    uint32_t spp = 5, max_bounces = 20, rr_depth = 0;
    float sky_scale = 10.0f, eps = 0.0003f, tmin = 0.00001f, tmax = 1e27f;
};

struct Ctx {
    std::vector<float> vertices;
    std::vector<uint32_t> indices, submeshes, counts;
    std::vector<float> xforms, materials;
    std::vector<Texture> textures;
    std::vector<float> env;
    uint32_t env_n = 0;
    std::vector<WorldTri> tris;
    Bvh bvh;
    uint32_t W = 0, H = 0;
    mat4 cam[5];
    Params prm;
    std::vector<float> image, accum;
    std::vector<uint32_t> primary;
    std::atomic<uint64_t> rays{0}, nodes{0}, tritests{0};
};

static vec3 xform_point(const float* m, vec3 p)
{
    return V3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3],
              m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}

static vec3 xform_dir(const float* m, vec3 n)   // mat3(gl_ObjectToWorldEXT) * n
{
    return V3(m[0] * n.x + m[1] * n.y + m[2] * n.z,
              m[4] * n.x + m[5] * n.y + m[6] * n.z,
              m[8] * n.x + m[9] * n.y + m[10] * n.z);
}

static void build_world_tris(Ctx& c)
{
    c.tris.clear();
    uint32_t ns = (uint32_t)c.counts.size();
    for (uint32_t s = 0; s < ns; s++) {
        uint32_t vo = c.submeshes[s * 4 + 1], io = c.submeshes[s * 4 + 2];
        const float* m = &c.xforms[s * 12];
        for (uint32_t p = 0; p < c.counts[s] / 3; p++) {
            WorldTri t;
            vec3* dst[3] = {&t.v0, &t.v1, &t.v2};
            for (int k = 0; k < 3; k++) {
                uint32_t vi = c.indices[io + p * 3 + k] + vo;
                const float* v = &c.vertices[(size_t)vi * 12];
                *dst[k] = xform_point(m, V3(v[0], v[1], v[2]));
            }
            t.submesh = s;
            t.prim = p;
            c.tris.push_back(t);
        }
    }
}

// sampler state of a texture (flags): wrap mode of u in bits 0-1 and of v in bits 2-3 (0 repeat, 1 clamp to
// edge, 2 mirrored repeat), bit 4 = nearest instead of bilinear.  The reference's sampler objects are created by
// its absent engine; glTF's sampler fields map onto these (parity unpinned).
static float wrap_coord(float u, uint32_t mode)
{
    if (mode == 0u) {
        float f = u - floorf(u);
        return f >= 0.0f ? f : 0.0f;
    }
    if (mode == 2u) {
        float h = u * 0.5f;
        float t = (h - floorf(h)) * 2.0f;
        float m = t <= 1.0f ? t : 2.0f - t;
        return m >= 0.0f ? m : 0.0f;
    }
    if (!(u >= -1.0f)) u = -1.0f;
    return u > 2.0f ? 2.0f : u;
}

static int wrap_index(int i, int n, uint32_t mode)
{
    if (mode == 0u) {
        if (i < 0) i += n;
        if (i >= n) i -= n;
        return i;
    }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// texture(sampler2D, uv): level 0, RGBA8 UNORM (no sRGB decode); bilinear + repeat unless the flags say otherwise
static vec4 sample_texture(const Texture& tx, float u, float v)
{
    const uint32_t ws = tx.flags & 3u, wt = (tx.flags >> 2) & 3u;
    const bool nearest = (tx.flags & 16u) != 0;
    float uu = wrap_coord(u, ws), vv = wrap_coord(v, wt);
    float x = uu * (float)tx.w, y = vv * (float)tx.h;
    if (!nearest) { x = x - 0.5f; y = y - 0.5f; }
    float x0f = floorf(x), y0f = floorf(y);
    float fx = nearest ? 0.0f : x - x0f, fy = nearest ? 0.0f : y - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;
    int x1 = nearest ? x0 : x0 + 1, y1 = nearest ? y0 : y0 + 1;
    x0 = wrap_index(x0, (int)tx.w, ws); x1 = wrap_index(x1, (int)tx.w, ws);
    y0 = wrap_index(y0, (int)tx.h, wt); y1 = wrap_index(y1, (int)tx.h, wt);
    auto texel = [&](int xx, int yy, int ch) { return (float)tx.px[((size_t)yy * tx.w + xx) * 4 + ch] * (1.0f / 255.0f); };
    float out[4];
    for (int ch = 0; ch < 4; ch++) {
        float a = mix(texel(x0, y0, ch), texel(x1, y0, ch), fx);
        float b = mix(texel(x0, y1, ch), texel(x1, y1, ch), fx);
        out[ch] = mix(a, b, fy);
    }
    return vec4{out[0], out[1], out[2], out[3]};
}

// texture(samplerCube, d).rgb: Vulkan face selection (z wins ties, then y), bilinear
// inside the face, clamp to edge, level 0.
static vec3 sample_env(const Ctx& c, vec3 d)
{
    float ax = gabs(d.x), ay = gabs(d.y), az = gabs(d.z);
    int face;
    float sc, tc, ma;
    if (az >= ax && az >= ay) {
        ma = az;
        if (d.z < 0.0f) { face = 5; sc = -d.x; tc = -d.y; } else { face = 4; sc = d.x; tc = -d.y; }
    } else if (ay >= ax) {
        ma = ay;
        if (d.y < 0.0f) { face = 3; sc = d.x; tc = -d.z; } else { face = 2; sc = d.x; tc = d.z; }
    } else {
        ma = ax;
        if (d.x < 0.0f) { face = 1; sc = d.z; tc = -d.y; } else { face = 0; sc = -d.z; tc = -d.y; }
    }
    float n = (float)c.env_n;
    float s = 0.5f * (sc / ma + 1.0f), t = 0.5f * (tc / ma + 1.0f);
    float x = s * n - 0.5f, y = t * n - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    if (!(x0f == x0f)) x0f = 0.0f;
    if (!(y0f == y0f)) y0f = 0.0f;
    int N = (int)c.env_n;
    auto cl = [&](float f) { return f < 0.0f ? 0 : (f > (float)(N - 1) ? N - 1 : (int)f); };
    int x0 = cl(x0f), x1 = cl(x0f + 1.0f), y0 = cl(y0f), y1 = cl(y0f + 1.0f);
    const float* base = &c.env[(size_t)face * N * N * 4];
    float out[3];
    for (int ch = 0; ch < 3; ch++) {
        float a = mix(base[((size_t)y0 * N + x0) * 4 + ch], base[((size_t)y0 * N + x1) * 4 + ch], fx);
        float b = mix(base[((size_t)y1 * N + x0) * 4 + ch], base[((size_t)y1 * N + x1) * 4 + ch], fx);
        out[ch] = mix(a, b, fy);
    }
    return V3(out[0], out[1], out[2]);
}

// ClosestHit.glsl:144-226
static Payload closest_hit(const Ctx& c, const Hit& h, vec3 rayDir)
{
    const WorldTri& wt = c.tris[h.tri];
    uint32_t inst = wt.submesh, prim = wt.prim;
    uint32_t vertexOffset = c.submeshes[inst * 4 + 1];
    uint32_t indexOffset = c.submeshes[inst * 4 + 2];
    uint32_t materialIndex = c.submeshes[inst * 4 + 3];
    const float* M = &c.materials[(size_t)materialIndex * 22];
    const float* o2w = &c.xforms[inst * 12];

    float bary[3] = {1.0f - h.u - h.v, h.u, h.v};
    vec2 uv{0.0f, 0.0f};
    vec3 nrm = V3(0.0f);
    for (int i = 0; i < 3; i++) {
        uint32_t idx = c.indices[prim * 3 + i + indexOffset] + vertexOffset;
        const float* v = &c.vertices[(size_t)idx * 12];
        uv.x += v[3] * bary[i];
        uv.y += v[4] * bary[i];
        nrm += V3(v[5], v[6], v[7]) * bary[i];
    }
    nrm = normalize(nrm);
    vec3 worldNormal = normalize(xform_dir(o2w, nrm));
    vec3 view = normalize(-rayDir);

    vec3 albedoTex = V3(1.0f);
    int32_t albedoMap = (int32_t)f2u(M[10]);
    int32_t mrMap = (int32_t)f2u(M[11]);
    if (albedoMap != -1) {
        vec4 t = sample_texture(c.textures[albedoMap], uv.x, uv.y);
        albedoTex = V3(t.x, t.y, t.z);
    }
    vec2 mrTex{1.0f, 1.0f};
    if (mrMap != -1) {
        vec4 t = sample_texture(c.textures[mrMap], uv.x, uv.y);
        mrTex = vec2{t.z, t.y};   // .bg
    }
    // the normal map is fetched but never applied: "if (false && ...)" ClosestHit.glsl:189

    Payload p;
    p.Distance = h.t;
    p.Albedo = V3(M[0], M[1], M[2]) * albedoTex;
    p.Roughness = M[4] * mrTex.y;
    p.Metallic = M[3] * mrTex.x;
    p.Emission = V3(M[5], M[6], M[7]) * M[8];
    p.WorldNormal = worldNormal;
    p.Anisotropic = M[13];
    p.Subsurface = M[14];
    p.SpecularTint = M[15];
    p.Sheen = M[16];
    p.SheenTint = M[17];
    p.Clearcoat = M[18];
    p.ClearcoatRoughness = M[19];
    p.SpecTrans = M[20];
    p.ior = M[21];
    float aspect = sqrtf(1.0f - p.Anisotropic * 0.9f);
    p.ax = gmax(0.001f, p.Roughness / aspect);
    p.ay = gmax(0.001f, p.Roughness * aspect);
    p.eta = dot(view, worldNormal) < 0.0f ? (1.0f / p.ior) : p.ior;
    return p;
}

struct ThreadStats { uint64_t rays = 0, nodes = 0, tris = 0; };

static Hit trace_ray(const Ctx& c, vec3 o, vec3 d, bool use_bvh, ThreadStats& st)
{
    st.rays++;
    if (use_bvh) return c.bvh.intersect(c.tris, o, d, c.prm.tmin, c.prm.tmax, &st.nodes, &st.tris);
    return intersect_brute(c.tris, o, d, c.prm.tmin, c.prm.tmax);
}

// RayGen.glsl:106-176
static vec3 trace_path(const Ctx& c, vec3 origin, vec3 direction, uint32_t& seed, bool use_bvh, ThreadStats& st, uint32_t* primary)
{
    vec3 radiance = V3(0.0f);
    vec3 throughput = V3(1.0f);
    for (uint32_t bounce = 0; bounce < c.prm.max_bounces; bounce++) {
        Hit h = trace_ray(c, origin, direction, use_bvh, st);
        if (primary && bounce == 0) {
            if (h.tri == 0xFFFFFFFFu) { primary[0] = primary[1] = 0xFFFFFFFFu; }
            else { primary[0] = c.tris[h.tri].submesh; primary[1] = c.tris[h.tri].prim; }
        }
        if (h.tri == 0xFFFFFFFFu) {   // Miss.glsl: Distance = -1
            vec3 sky = sample_env(c, direction) * c.prm.sky_scale;
            radiance += sky * throughput;
            break;
        }
        Payload p = closest_hit(c, h, direction);
        radiance += p.Emission * throughput;

        vec3 negD = -direction;
        vec3 ffNormal = dot(negD, p.WorldNormal) < 0.0f ? -p.WorldNormal : p.WorldNormal;
        vec3 L;
        float pdf;
        vec3 f = DisneySample(p, negD, ffNormal, L, pdf, seed);
        if (pdf > 0.0f) throughput *= f / pdf;
        else break;

        vec3 fhp = origin + direction * p.Distance;
        direction = L;
        origin = fhp + direction * c.prm.eps;

        if (c.prm.rr_depth > 0 && bounce >= c.prm.rr_depth) {   // RayGen.glsl:163-173 (off in the reference)
            float q = gmin(gmax(throughput.x, gmax(throughput.y, throughput.z)) + 0.001f, 0.95f);
            if (RandomValue(seed) > q) break;
            throughput /= q;
        }
    }
    return radiance;
}

// RayGen.glsl:312-373
static void render_pixel(Ctx& c, uint32_t x, uint32_t y, uint32_t frame, bool use_bvh, ThreadStats& st)
{
    uint32_t seed = x + y * c.W;
    seed *= frame;
    vec3 color = V3(0.0f);
    const mat4& invView = c.cam[3];
    const mat4& invProj = c.cam[4];
    uint32_t S = c.prm.spp;
    size_t pix = (size_t)y * c.W + x;
    for (uint32_t i = 0; i < S; i++) {
        vec2 pc{(float)x + 0.5f, (float)y + 0.5f};
        if (i > 0) {
            vec2 j = RandomPointInCircle(seed);
            pc.x += j.x;
            pc.y += j.y;
        }
        vec2 inUV{pc.x / (float)c.W, pc.y / (float)c.H};
        vec2 d{inUV.x * 2.0f - 1.0f, inUV.y * 2.0f - 1.0f};
        vec4 target = mul(invProj, vec4{d.x, d.y, 1.0f, 1.0f});
        vec3 tn = normalize(V3(target.x, target.y, target.z) / target.w);
        vec4 dir4 = mul(invView, vec4{tn.x, tn.y, tn.z, 0.0f});
        vec3 origin = V3(invView.c[3][0], invView.c[3][1], invView.c[3][2]);
        vec3 direction = normalize(V3(dir4.x, dir4.y, dir4.z));
        color += trace_path(c, origin, direction, seed, use_bvh, st, i == 0 ? &c.primary[pix * 2] : nullptr);
    }
    float numPaths = (float)S;
    float* acc = &c.accum[pix * 4];
    if (frame > 1) {
        color += V3(acc[0], acc[1], acc[2]);
        numPaths = acc[3] + (float)S;
        acc[0] = color.x; acc[1] = color.y; acc[2] = color.z; acc[3] = numPaths;
    } else {
        acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
    }
    color /= numPaths;
    float* img = &c.image[pix * 4];
    if (gisnan(color.x) || gisnan(color.y) || gisnan(color.z)) { img[0] = 1.0f; img[1] = 0.0f; img[2] = 0.0f; img[3] = 1.0f; }
    else { img[0] = color.x; img[1] = color.y; img[2] = color.z; img[3] = 1.0f; }
}

template <class F>
static void parallel_for(uint32_t n, int num_threads, F body)
{
    if (num_threads <= 1) { ThreadStats st; for (uint32_t i = 0; i < n; i++) body(i, st); return; }
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < num_threads; t++)
        th.emplace_back([&]() {
            ThreadStats st;
            for (;;) {
                uint32_t i = next.fetch_add(1);
                if (i >= n) break;
                body(i, st);
            }
        });
    for (auto& t : th) t.join();
}

static Payload payload_from_surf(const float* s)
{
    Payload p{};
    p.Distance = 1.0f;
    p.Albedo = V3(s[0], s[1], s[2]);
    p.Metallic = s[3]; p.Roughness = s[4]; p.Subsurface = s[5]; p.SpecularTint = s[6];
    p.Sheen = s[7]; p.SheenTint = s[8]; p.Clearcoat = s[9]; p.ClearcoatRoughness = s[10];
    p.SpecTrans = s[11]; p.ior = s[12]; p.ax = s[13]; p.ay = s[14]; p.eta = s[15]; p.Anisotropic = s[16];
    p.Emission = V3(0.0f);
    p.WorldNormal = V3(0.0f);
    return p;
}

} // namespace orc

using namespace orc;

extern "C" {

struct orc_ctx { Ctx c; };

orc_ctx* orc_create(const orc_scene_desc* s)
{
    orc_ctx* h = new orc_ctx();
    Ctx& c = h->c;
    c.vertices.assign(s->vertices, s->vertices + (size_t)s->num_vertices * 12);
    c.indices.assign(s->indices, s->indices + s->num_indices);
    c.submeshes.assign(s->submeshes, s->submeshes + (size_t)s->num_submeshes * 4);
    c.counts.assign(s->submesh_index_counts, s->submesh_index_counts + s->num_submeshes);
    c.xforms.assign(s->submesh_xforms, s->submesh_xforms + (size_t)s->num_submeshes * 12);
    c.materials.assign(s->materials, s->materials + (size_t)s->num_materials * 22);
    for (uint32_t i = 0; i < s->num_textures; i++) {
        Texture t;
        t.w = s->textures[i].width; t.h = s->textures[i].height; t.flags = s->textures[i].flags;
        t.px.assign(s->textures[i].rgba8, s->textures[i].rgba8 + (size_t)t.w * t.h * 4);
        c.textures.push_back(std::move(t));
    }
    if (s->env_rgba && s->env_face_size) {
        c.env_n = s->env_face_size;
        c.env.assign(s->env_rgba, s->env_rgba + (size_t)6 * c.env_n * c.env_n * 4);
    } else {   // no HDR supplied: constant white sky, one texel per face
        c.env_n = 1;
        c.env.assign(24, 1.0f);
    }
    build_world_tris(c);
    c.bvh.build(c.tris);
    for (int i = 0; i < 5; i++)
        for (int a = 0; a < 4; a++)
            for (int b = 0; b < 4; b++) c.cam[i].c[a][b] = (a == b) ? 1.0f : 0.0f;
    return h;
}

void orc_destroy(orc_ctx* h) { delete h; }

int orc_update_materials(orc_ctx* h, const float* m, uint32_t n)
{
    if (n * 22 != h->c.materials.size()) return 1;
    h->c.materials.assign(m, m + (size_t)n * 22);
    return 0;
}

int orc_resize(orc_ctx* h, uint32_t W, uint32_t Hh)
{
    Ctx& c = h->c;
    c.W = W; c.H = Hh;
    c.image.assign((size_t)W * Hh * 4, 0.0f);
    c.accum.assign((size_t)W * Hh * 4, 0.0f);
    c.primary.assign((size_t)W * Hh * 2, 0xFFFFFFFFu);
    return 0;
}

int orc_set_camera(orc_ctx* h, const float* m)
{
    std::memcpy(h->c.cam, m, sizeof(float) * 80);
    return 0;
}

int orc_set_params(orc_ctx* h, const orc_params* p)
{
    Params& q = h->c.prm;
    q.spp = p->spp_per_frame; q.max_bounces = p->max_bounces; q.rr_depth = p->rr_depth;
    q.sky_scale = p->sky_scale; q.eps = p->eps; q.tmin = p->tmin; q.tmax = p->tmax;
    return 0;
}

int orc_trace(orc_ctx* h, uint32_t frame, int num_threads, int use_bvh)
{
    Ctx& c = h->c;
    if (c.W == 0 || c.H == 0) return 1;
    std::atomic<uint64_t> rays{0}, nodes{0}, tris{0};
    // one work item = one image row
    if (num_threads <= 1) {
        ThreadStats st;
        for (uint32_t y = 0; y < c.H; y++)
            for (uint32_t x = 0; x < c.W; x++) render_pixel(c, x, y, frame, use_bvh != 0, st);
        rays += st.rays; nodes += st.nodes; tris += st.tris;
    } else {
        std::atomic<uint32_t> next{0};
        std::vector<std::thread> th;
        for (int t = 0; t < num_threads; t++)
            th.emplace_back([&]() {
                ThreadStats st;
                for (;;) {
                    uint32_t y = next.fetch_add(1);
                    if (y >= c.H) break;
                    for (uint32_t x = 0; x < c.W; x++) render_pixel(c, x, y, frame, use_bvh != 0, st);
                }
                rays += st.rays; nodes += st.nodes; tris += st.tris;
            });
        for (auto& t : th) t.join();
    }
    c.rays += rays.load(); c.nodes += nodes.load(); c.tritests += tris.load();
    return 0;
}

int orc_read_image(orc_ctx* h, float* out) { std::memcpy(out, h->c.image.data(), h->c.image.size() * 4); return 0; }
int orc_read_accum(orc_ctx* h, float* out) { std::memcpy(out, h->c.accum.data(), h->c.accum.size() * 4); return 0; }
int orc_write_accum(orc_ctx* h, const float* in) { std::memcpy(h->c.accum.data(), in, h->c.accum.size() * 4); return 0; }
int orc_read_primary_ids(orc_ctx* h, uint32_t* out) { std::memcpy(out, h->c.primary.data(), h->c.primary.size() * 4); return 0; }
uint64_t orc_ray_count(orc_ctx* h) { return h->c.rays.load(); }
void orc_traversal_stats(orc_ctx* h, uint64_t* n, uint64_t* t) { *n = h->c.nodes.load(); *t = h->c.tritests.load(); }
uint32_t orc_num_triangles(orc_ctx* h) { return (uint32_t)h->c.tris.size(); }

int orc_intersect(orc_ctx* h, const float* rays, uint32_t n, uint32_t* hits, int use_bvh, int num_threads)
{
    Ctx& c = h->c;
    parallel_for(n, num_threads, [&](uint32_t i, ThreadStats&) {
        const float* r = &rays[(size_t)i * 8];
        vec3 o = V3(r[0], r[1], r[2]), d = V3(r[4], r[5], r[6]);
        Hit hit = use_bvh ? c.bvh.intersect(c.tris, o, d, r[3], r[7]) : intersect_brute(c.tris, o, d, r[3], r[7]);
        uint32_t* out = &hits[(size_t)i * 5];
        if (hit.tri == 0xFFFFFFFFu) {
            out[0] = f2u(r[7]); out[1] = out[2] = 0; out[3] = out[4] = 0xFFFFFFFFu;
        } else {
            out[0] = f2u(hit.t); out[1] = f2u(hit.u); out[2] = f2u(hit.v);
            out[3] = c.tris[hit.tri].submesh; out[4] = c.tris[hit.tri].prim;
        }
    });
    return 0;
}

uint32_t orc_pcg_hash(uint32_t* seed) { return PCG_Hash(*seed); }
float orc_random_value(uint32_t* seed) { return RandomValue(*seed); }

void orc_det_math(int fn, const float* x, const float* y, float* out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = det_sin(x[i]); break;
        case 1: out[i] = det_cos(x[i]); break;
        case 2: out[i] = det_log(x[i]); break;
        case 3: out[i] = det_exp(x[i]); break;
        case 4: out[i] = det_pow(x[i], y[i]); break;
        case 5: out[i] = det_atan2(x[i], y[i]); break;
        case 6: out[i] = det_acos(x[i]); break;
        case 7: out[i] = det_tan(x[i]); break;
        default: out[i] = round_to_half(x[i]); break;
        }
    }
}

void orc_sample_texture(const orc_texture_desc* tex, const float* uv, uint32_t n, float* rgba)
{
    Texture t;
    t.w = tex->width; t.h = tex->height; t.flags = tex->flags;
    t.px.assign(tex->rgba8, tex->rgba8 + (size_t)tex->width * tex->height * 4);
    for (uint32_t i = 0; i < n; i++) {
        vec4 c = sample_texture(t, uv[2 * i], uv[2 * i + 1]);
        rgba[4 * i] = c.x; rgba[4 * i + 1] = c.y; rgba[4 * i + 2] = c.z; rgba[4 * i + 3] = c.w;
    }
}

void orc_postprocess(const float* rgba, uint32_t num_pixels, float exposure, uint8_t* rgba8)
{
    for (uint32_t i = 0; i < num_pixels; i++) PostProcessPixel(rgba + (size_t)i * 4, exposure, rgba8 + (size_t)i * 4);
}

void orc_equirect_to_cube(const float* rgba, uint32_t w, uint32_t h, uint32_t n, float* cube)
{
    for (uint32_t f = 0; f < 6; f++)
        for (uint32_t y = 0; y < n; y++)
            for (uint32_t x = 0; x < n; x++) EquirectToCubeTexel(rgba, w, h, x, y, f, n, cube + (((size_t)f * n + y) * n + x) * 4);
}

void orc_preetham_sky(float turbidity, float azimuth, float inclination, uint32_t n, float* cube)
{
    for (uint32_t f = 0; f < 6; f++)
        for (uint32_t y = 0; y < n; y++)
            for (uint32_t x = 0; x < n; x++) PreethamTexel(turbidity, azimuth, inclination, x, y, f, n, cube + (((size_t)f * n + y) * n + x) * 4);
}

void orc_disney_sample(const float* surf, const float* V, const float* N, uint32_t* seed, float* f, float* L, float* pdf)
{
    Payload p = payload_from_surf(surf);
    vec3 l;
    vec3 r = DisneySample(p, V3(V[0], V[1], V[2]), V3(N[0], N[1], N[2]), l, *pdf, *seed);
    f[0] = r.x; f[1] = r.y; f[2] = r.z;
    L[0] = l.x; L[1] = l.y; L[2] = l.z;
}

void orc_disney_eval(const float* surf, const float* V, const float* N, const float* L, float* f, float* pdf)
{
    Payload p = payload_from_surf(surf);
    vec3 r = DisneyEval(p, V3(V[0], V[1], V[2]), V3(N[0], N[1], N[2]), V3(L[0], L[1], L[2]), *pdf);
    f[0] = r.x; f[1] = r.y; f[2] = r.z;
}

// ClosestHit.glsl:144-226 for one given hit: out = 26 floats
// Distance, Albedo3, Roughness, Metallic, Emission3, WorldNormal3, Anisotropic, Subsurface, SpecularTint, Sheen,
// SheenTint, Clearcoat, ClearcoatRoughness, SpecTrans, ior, ax, ay, eta.  Returns -1 if the scene has no such triangle.
int orc_closest_hit(orc_ctx* h, uint32_t submesh, uint32_t prim, float u, float v, float t, const float* dir, float* out)
{
    const Ctx& c = h->c;
    for (size_t k = 0; k < c.tris.size(); k++) {
        if (c.tris[k].submesh != submesh || c.tris[k].prim != prim) continue;
        Hit hit;
        hit.t = t; hit.u = u; hit.v = v; hit.tri = (uint32_t)k;
        Payload p = closest_hit(c, hit, V3(dir[0], dir[1], dir[2]));
        const float vals[26] = {p.Distance, p.Albedo.x, p.Albedo.y, p.Albedo.z, p.Roughness, p.Metallic, p.Emission.x, p.Emission.y, p.Emission.z,
                                p.WorldNormal.x, p.WorldNormal.y, p.WorldNormal.z, p.Anisotropic, p.Subsurface, p.SpecularTint, p.Sheen, p.SheenTint,
                                p.Clearcoat, p.ClearcoatRoughness, p.SpecTrans, p.ior, p.ax, p.ay, p.eta};
        std::memcpy(out, vals, sizeof(vals));
        return 0;
    }
    return -1;
}

// TracePath (RayGen.glsl:106-176) of one given camera ray with the context's parameters; *seed advances as in the shader
void orc_trace_path(orc_ctx* h, const float* origin, const float* dir, uint32_t* seed, int use_bvh, float* radiance)
{
    ThreadStats st;
    vec3 r = trace_path(h->c, V3(origin[0], origin[1], origin[2]), V3(dir[0], dir[1], dir[2]), *seed, use_bvh != 0, st, nullptr);
    radiance[0] = r.x; radiance[1] = r.y; radiance[2] = r.z;
}

} // extern "C"
