// Flat C entry points over the C++ host layer (include/pathtracer_host.h).  Exceptions stop here.
#include "../../include/pathtracer_host.h"
#include "host.h"
#include <cstdio>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

using namespace ptamd;

struct pth_mesh {
    Ref<MeshSource> source;
    Ref<Mesh> mesh;
};

struct pth_layer {
    Ref<Device> device;
    std::unique_ptr<RayTracingLayer> layer;
    std::string error;
};

static void put_err(char* err, size_t n, const char* msg)
{
    if (err && n) std::snprintf(err, n, "%s", msg);
}

static CameraSpecification to_spec(const pth_camera_spec* s)
{
    CameraSpecification c;
    if (!s) return c;
    c.pitch = s->pitch; c.yaw = s->yaw;
    std::memcpy(c.position, s->position, sizeof(c.position));
    if (s->fov_y > 0) c.fovY = s->fov_y;
    if (s->near_clip > 0) c.nearClip = s->near_clip;
    if (s->far_clip > 0) c.farClip = s->far_clip;
    return c;
}

template <class F>
static int guarded(pth_layer* l, F f)
{
    if (!l) return PT_ERR_INVALID;
    try { f(); return PT_OK; }
    catch (const std::exception& e) { l->error = e.what(); return PT_ERR_INVALID; }
}

extern "C" {

pth_mesh* pth_mesh_load(const char* path, char* err, size_t errlen)
{
    try {
        pth_mesh* m = new pth_mesh();
        m->source = CreateRef<MeshSource>(std::string(path));
        m->mesh = CreateRef<Mesh>(m->source);
        return m;
    } catch (const std::exception& e) {
        put_err(err, errlen, e.what());
        return nullptr;
    }
}

void pth_mesh_free(pth_mesh* m) { delete m; }

void pth_mesh_counts(pth_mesh* m, uint32_t* nv, uint32_t* ni, uint32_t* ns, uint32_t* nm, uint32_t* nt)
{
    *nv = (uint32_t)m->source->GetVertices().size();
    *ni = (uint32_t)m->source->GetIndices().size();
    *ns = (uint32_t)m->source->GetSubMeshes().size();
    *nm = (uint32_t)m->source->GetMaterials().size();
    *nt = (uint32_t)m->source->GetTextures().size();
}

int pth_mesh_tables(pth_mesh* m, const float* layer, float* vertices, uint32_t* indices, uint32_t* submeshes,
                    uint32_t* counts, float* xforms, float* materials)
{
    const MeshSource& s = *m->source;
    std::memcpy(vertices, s.GetVertices().data(), s.GetVertices().size() * sizeof(Vertex));
    std::memcpy(indices, s.GetIndices().data(), s.GetIndices().size() * 4);
    Mat4 L = Mat4::Identity();
    if (layer) std::memcpy(L.m, layer, 64);
    for (size_t i = 0; i < s.GetSubMeshes().size(); i++) {
        const SubMesh& sm = s.GetSubMeshes()[i];
        submeshes[i * 4 + 0] = 0;
        submeshes[i * 4 + 1] = sm.VertexOffset;
        submeshes[i * 4 + 2] = sm.IndexOffset;
        submeshes[i * 4 + 3] = sm.MaterialIndex;
        counts[i] = sm.IndexCount;
        Mat4 w = L * sm.WorldTransform;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) xforms[i * 12 + r * 4 + c] = w.at(r, c);
    }
    const std::vector<TextureData>& tex = s.GetTextures();
    for (size_t i = 0; i < s.GetMaterials().size(); i++) {
        MaterialData d = s.GetMaterials()[i];
        auto fix = [&](int32_t& idx) { if (idx >= 0 && ((size_t)idx >= tex.size() || tex[(size_t)idx].RGBA8.empty())) idx = -1; };
        fix(d.AlbedoMapIndex); fix(d.MetallicRoughnessMapIndex); fix(d.NormalMapIndex);
        std::memcpy(materials + i * 22, &d, 88);
    }
    return PT_OK;
}

const char* pth_mesh_submesh_name(pth_mesh* m, uint32_t i) { return m->source->GetSubMeshes().at(i).Name.c_str(); }
const char* pth_mesh_texture_uri(pth_mesh* m, uint32_t i) { return m->source->GetTextures().at(i).Uri.c_str(); }
uint32_t pth_mesh_texture_sampler(pth_mesh* m, uint32_t i) { return i < m->source->GetTextures().size() ? m->source->GetTextures()[i].SamplerFlags : 0u; }

int pth_mesh_get_texture(pth_mesh* m, uint32_t i, uint32_t* w, uint32_t* h, uint8_t* rgba8)
{
    if (i >= m->source->GetTextures().size() || !w || !h) return PT_ERR_INVALID;
    const TextureData& t = m->source->GetTextures()[i];
    *w = t.Width; *h = t.Height;
    if (rgba8 && !t.RGBA8.empty()) std::memcpy(rgba8, t.RGBA8.data(), t.RGBA8.size());
    return PT_OK;
}

int pth_mesh_set_texture(pth_mesh* m, uint32_t i, uint32_t w, uint32_t h, const uint8_t* rgba8)
{
    if (i >= m->source->GetTextures().size() || !w || !h || !rgba8) return PT_ERR_INVALID;
    TextureData& t = m->source->GetTextures()[i];
    t.Width = w; t.Height = h;
    t.RGBA8.assign(rgba8, rgba8 + (size_t)w * h * 4);
    return PT_OK;
}

int pth_mesh_set_material(pth_mesh* m, uint32_t i, const float* words22)
{
    if (i >= m->source->GetMaterials().size() || !words22) return PT_ERR_INVALID;
    std::memcpy(&m->source->GetMaterials()[i], words22, 88);
    return PT_OK;
}

void pth_camera_matrices(const pth_camera_spec* spec, uint32_t width, uint32_t height, float* out80)
{
    Camera cam(to_spec(spec));
    if (spec && spec->use_target)
        cam.LookAt({spec->position[0], spec->position[1], spec->position[2]}, {spec->target[0], spec->target[1], spec->target[2]});
    cam.Resize(width, height);
    CameraBuffer b{cam.GetViewProjection(), cam.GetInverseViewProjection(), cam.GetView(), cam.GetInverseView(), cam.GetInverseProjection()};
    std::memcpy(out80, &b, sizeof(b));
}

pth_layer* pth_layer_create(int device, pth_mesh* mesh, const float* transform16, const pth_camera_spec* camera, char* err, size_t errlen)
{
    try {
        pth_layer* l = new pth_layer();
        l->device = CreateRef<Device>(device);
        Mat4 T = Mat4::Identity();
        if (transform16) std::memcpy(T.m, transform16, 64);
        l->layer.reset(new RayTracingLayer(l->device, mesh->mesh, T, to_spec(camera)));
        if (camera && camera->use_target)
            l->layer->GetCamera()->LookAt({camera->position[0], camera->position[1], camera->position[2]},
                                          {camera->target[0], camera->target[1], camera->target[2]});
        return l;
    } catch (const std::exception& e) {
        put_err(err, errlen, e.what());
        return nullptr;
    }
}

void pth_layer_destroy(pth_layer* l)
{
    if (!l) return;
    l->layer.reset();
    l->device.reset();
    delete l;
}

int pth_layer_set_viewport(pth_layer* l, uint32_t w, uint32_t h) { return guarded(l, [&] { l->layer->SetViewportSize(w, h); }); }
int pth_layer_set_params(pth_layer* l, const pt_params* p) { return guarded(l, [&] { l->layer->SetParams(*p); }); }
int pth_layer_set_accumulate(pth_layer* l, int on) { return guarded(l, [&] { l->layer->SetAccumulate(on != 0); }); }
int pth_layer_set_camera_position(pth_layer* l, const float* p) { return guarded(l, [&] { l->layer->GetCamera()->SetPosition({p[0], p[1], p[2]}); }); }
int pth_layer_materials_changed(pth_layer* l) { return guarded(l, [&] { l->layer->MaterialsChanged(); }); }
int pth_layer_on_update(pth_layer* l) { return guarded(l, [&] { l->layer->OnUpdate(); }); }
int pth_layer_on_render(pth_layer* l) { return guarded(l, [&] { l->layer->OnRender(); }); }
uint32_t pth_layer_frame_index(pth_layer* l) { return l->layer->GetFrameIndex(); }

int pth_layer_read_image(pth_layer* l, float* rgba)
{
    return guarded(l, [&] { const std::vector<float>& h = l->layer->GetImage()->Download(); std::memcpy(rgba, h.data(), h.size() * 4); });
}
int pth_layer_read_accum(pth_layer* l, float* rgba)
{
    return guarded(l, [&] { const std::vector<float>& h = l->layer->GetAccumulationImage()->Download(); std::memcpy(rgba, h.data(), h.size() * 4); });
}
int pth_layer_ray_count(pth_layer* l, uint64_t* rays) { return guarded(l, [&] { *rays = l->layer->GetRayCount(); }); }
const char* pth_layer_last_error(pth_layer* l) { return l ? l->error.c_str() : ""; }

int pth_layer_set_submesh_transform(pth_layer* l, uint32_t submesh, const float* column_major16)
{
    return guarded(l, [&] {
        ptamd::Mat4 m;
        std::memcpy(&m, column_major16, sizeof(float) * 16);
        l->layer->SetSubMeshTransform(submesh, m);
    });
}
int pth_layer_drag_submesh_transform(pth_layer* l, uint32_t submesh, const float* column_major16)
{
    return guarded(l, [&] {
        ptamd::Mat4 m;
        std::memcpy(&m, column_major16, sizeof(float) * 16);
        l->layer->SetSubMeshTransform(submesh, m, true);
    });
}
int pth_layer_set_radiance_map(pth_layer* l, const char* hdr_path, uint32_t face_size)
{
    return guarded(l, [&] { l->layer->SetRadianceMap(hdr_path, face_size); });
}
int pth_layer_set_skybox_settings(pth_layer* l, float turbidity, float azimuth, float inclination, int use_as_environment, uint32_t face_size)
{
    return guarded(l, [&] { l->layer->SetSkyboxSettings(turbidity, azimuth, inclination, use_as_environment != 0, face_size); });
}
int pth_layer_set_post_processing(pth_layer* l, int on, float exposure) { return guarded(l, [&] { l->layer->SetPostProcessing(on != 0, exposure); }); }
int pth_layer_read_postprocessed(pth_layer* l, uint8_t* rgba8)
{
    return guarded(l, [&] { std::vector<uint8_t> px = l->layer->ReadPostProcessed(); std::memcpy(rgba8, px.data(), px.size()); });
}

int pth_hdr_load(const char* path, uint32_t* width, uint32_t* height, float** rgba, char* err, size_t errlen)
{
    try {
        ptamd::HdrImage img = ptamd::LoadHdr(path);
        float* px = (float*)malloc(img.Pixels.size() * sizeof(float));
        if (!px) throw std::runtime_error("out of memory");
        std::memcpy(px, img.Pixels.data(), img.Pixels.size() * sizeof(float));
        *width = img.Width; *height = img.Height; *rgba = px;
        return 0;
    } catch (const std::exception& e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.what());
        return 1;
    }
}
void pth_pixels_free(float* rgba) { free(rgba); }

int pth_image_load(const char* path, uint32_t* width, uint32_t* height, uint8_t** rgba8, char* err, size_t errlen)
{
    try {
        ptamd::TextureData t = ptamd::LoadImageFile(path);
        uint8_t* px = (uint8_t*)malloc(t.RGBA8.size());
        if (!px) throw std::runtime_error("out of memory");
        std::memcpy(px, t.RGBA8.data(), t.RGBA8.size());
        *width = t.Width; *height = t.Height; *rgba8 = px;
        return 0;
    } catch (const std::exception& e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.what());
        return 1;
    }
}
void pth_image_free(uint8_t* rgba8) { free(rgba8); }
int pth_mesh_load_textures(pth_mesh* m, char* err, size_t errlen)
{
    try {
        return (int)m->source->LoadTextures();
    } catch (const std::exception& e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.what());
        return -1;
    }
}
int pth_ppm_write(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba8, char* err, size_t errlen)
{
    try {
        ptamd::WritePpm(path, width, height, rgba8);
        return 0;
    } catch (const std::exception& e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.what());
        return 1;
    }
}

} // extern "C"
