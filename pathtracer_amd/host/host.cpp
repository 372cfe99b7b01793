// Camera, images, acceleration structure and the headless RayTracingLayer of the C++ host layer
// (declared in host.h), all on top of the C ABI of the HIP backend.
#include "host.h"
#include <cmath>
#include <cstring>

namespace ptamd {

// ---- Mat4 -------------------------------------------------------------------------------------
Mat4 Mat4::Identity()
{
    Mat4 r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
}
Mat4 Mat4::Scale(float x, float y, float z)
{
    Mat4 r = Identity();
    r.m[0] = x; r.m[5] = y; r.m[10] = z;
    return r;
}
Mat4 Mat4::Translate(float x, float y, float z)
{
    Mat4 r = Identity();
    r.m[12] = x; r.m[13] = y; r.m[14] = z;
    return r;
}
Mat4 Mat4::Perspective(float fovYDegrees, float aspect, float zNear, float zFar)
{
    double f = 1.0 / std::tan(fovYDegrees * 3.14159265358979323846 / 180.0 * 0.5);
    Mat4 r{};
    r.at(0, 0) = (float)(f / aspect);
    r.at(1, 1) = (float)(-f);
    r.at(2, 2) = (float)(zFar / ((double)zNear - zFar));
    r.at(2, 3) = (float)(((double)zNear * zFar) / ((double)zNear - zFar));
    r.at(3, 2) = -1.0f;
    return r;
}
Mat4 Mat4::LookAt(const float eye[3], const float target[3], const float up[3])
{
    double f[3] = {(double)target[0] - eye[0], (double)target[1] - eye[1], (double)target[2] - eye[2]};
    double fl = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (double& v : f) v /= fl;
    double s[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
    double sl = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    for (double& v : s) v /= sl;
    double u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
    Mat4 r = Identity();
    for (int c = 0; c < 3; c++) {
        r.at(0, c) = (float)s[c];
        r.at(1, c) = (float)u[c];
        r.at(2, c) = (float)-f[c];
    }
    r.at(0, 3) = (float)-(s[0] * eye[0] + s[1] * eye[1] + s[2] * eye[2]);
    r.at(1, 3) = (float)-(u[0] * eye[0] + u[1] * eye[1] + u[2] * eye[2]);
    r.at(2, 3) = (float)(f[0] * eye[0] + f[1] * eye[1] + f[2] * eye[2]);
    return r;
}
Mat4 Mat4::operator*(const Mat4& b) const
{
    Mat4 r{};
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += (double)at(row, k) * b.at(k, col);
            r.at(row, col) = (float)s;
        }
    return r;
}
Mat4 Mat4::Inverse() const
{
    // Gauss-Jordan in double with partial pivoting
    double a[4][8];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) { a[r][c] = at(r, c); a[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int i = 0; i < 4; i++) {
        int p = i;
        for (int r = i + 1; r < 4; r++) if (std::fabs(a[r][i]) > std::fabs(a[p][i])) p = r;
        if (a[p][i] == 0.0) throw std::runtime_error("Mat4::Inverse: singular matrix");
        if (p != i) for (int c = 0; c < 8; c++) std::swap(a[p][c], a[i][c]);
        double inv = 1.0 / a[i][i];
        for (int c = 0; c < 8; c++) a[i][c] *= inv;
        for (int r = 0; r < 4; r++) {
            if (r == i) continue;
            double f = a[r][i];
            if (f != 0.0) for (int c = 0; c < 8; c++) a[r][c] -= f * a[i][c];
        }
    }
    Mat4 out{};
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out.at(r, c) = (float)a[r][4 + c];
    return out;
}

// ---- Camera -----------------------------------------------------------------------------------
Camera::Camera(const CameraSpecification& spec) : m_Spec(spec) { Recalculate(); }

void Camera::Recalculate()
{
    float target[3];
    if (m_UseLookAt) {
        std::memcpy(target, m_Target, sizeof(target));
    } else {
        double cp = std::cos(m_Spec.pitch), sp = std::sin(m_Spec.pitch), cy = std::cos(m_Spec.yaw), sy = std::sin(m_Spec.yaw);
        target[0] = (float)(m_Spec.position[0] + sy * cp);
        target[1] = (float)(m_Spec.position[1] - sp);
        target[2] = (float)(m_Spec.position[2] + cy * cp);
    }
    const float up[3] = {0.0f, 1.0f, 0.0f};
    m_View = Mat4::LookAt(m_Spec.position, target, up);
    m_Projection = Mat4::Perspective(m_Spec.fovY, (float)m_Width / (float)m_Height, m_Spec.nearClip, m_Spec.farClip);
    m_ViewProjection = m_Projection * m_View;
    m_InverseView = m_View.Inverse();
    m_InverseProjection = m_Projection.Inverse();
    m_InverseViewProjection = m_ViewProjection.Inverse();
}

bool Camera::Update()
{
    bool moved = m_Dirty;
    if (m_Dirty) Recalculate();
    m_Dirty = false;
    return moved;
}
void Camera::Resize(uint32_t width, uint32_t height)
{
    if (width == m_Width && height == m_Height) return;
    m_Width = width ? width : 1;
    m_Height = height ? height : 1;
    Recalculate();
}
void Camera::SetPosition(const std::array<float, 3>& p)
{
    std::memcpy(m_Spec.position, p.data(), sizeof(float) * 3);
    m_Dirty = true;
    Recalculate();
}
void Camera::SetPitchYaw(float pitch, float yaw)
{
    m_Spec.pitch = pitch;
    m_Spec.yaw = yaw;
    m_UseLookAt = false;
    m_Dirty = true;
    Recalculate();
}
void Camera::LookAt(const std::array<float, 3>& eye, const std::array<float, 3>& target)
{
    std::memcpy(m_Spec.position, eye.data(), sizeof(float) * 3);
    std::memcpy(m_Target, target.data(), sizeof(float) * 3);
    m_UseLookAt = true;
    m_Dirty = true;
    Recalculate();
}

// ---- Device -----------------------------------------------------------------------------------
Device::Device(int ordinal)
{
    int rc = pt_create(ordinal, &m_Ctx);
    if (rc) throw std::runtime_error(std::string("pt_create: ") + pt_last_error(nullptr));
}
Device::~Device() { pt_destroy(m_Ctx); }
void Device::Check(int rc, const char* what) const
{
    if (rc) throw std::runtime_error(std::string(what) + ": " + pt_last_error(m_Ctx));
}

// ---- Image ------------------------------------------------------------------------------------
Image::Image(Ref<Device> device, Kind kind, const ImageSpecification& spec) : m_Device(std::move(device)), m_Kind(kind), m_Spec(spec)
{
    if (spec.Format != ImageFormat::RGBA32F) throw std::runtime_error("Image: the ray-tracing images are RGBA32F (RayTracingLayer.cpp:101,111)");
}
void Image::Resize(uint32_t width, uint32_t height)
{
    m_Spec.Width = width;
    m_Spec.Height = height;
}
const std::vector<float>& Image::Download()
{
    m_Host.resize((size_t)m_Spec.Width * m_Spec.Height * 4);
    if (m_Kind == Kind::Final) m_Device->Check(pt_read_image(m_Device->Get(), m_Host.data()), "pt_read_image");
    else m_Device->Check(pt_read_accum(m_Device->Get(), m_Host.data()), "pt_read_accum");
    return m_Host;
}
void Image::Upload(const std::vector<float>& rgba)
{
    if (m_Kind != Kind::Accumulation) throw std::runtime_error("Image::Upload: only the accumulation image can be restored");
    if (rgba.size() != (size_t)m_Spec.Width * m_Spec.Height * 4) throw std::runtime_error("Image::Upload: size mismatch");
    m_Device->Check(pt_write_accum(m_Device->Get(), rgba.data()), "pt_write_accum");
}

// ---- AccelerationStructure ----------------------------------------------------------------------
static std::vector<MaterialData> upload_materials(const Mesh& mesh_const)
{
    Mesh& mesh = const_cast<Mesh&>(mesh_const);
    std::vector<MaterialData> mats = mesh.GetMaterialBuffers();
    const std::vector<TextureData>& tex = mesh.GetMeshSource()->GetTextures();
    auto fix = [&](int32_t& idx) {
        // a map whose image was not decoded cannot be bound: fall back to the factor alone
        if (idx >= 0 && ((size_t)idx >= tex.size() || tex[(size_t)idx].RGBA8.empty())) idx = -1;
    };
    for (MaterialData& m : mats) { fix(m.AlbedoMapIndex); fix(m.MetallicRoughnessMapIndex); fix(m.NormalMapIndex); }
    return mats;
}

AccelerationStructure::AccelerationStructure(Ref<Device> device, const AccelerationStructureSpecification& spec, uint32_t buildFlags)
    : m_Device(std::move(device)), m_Spec(spec)
{
    Mesh& mesh = *m_Spec.Mesh;
    const std::vector<SubMesh>& subs = mesh.GetSubMeshes();
    std::vector<uint32_t> counts;
    std::vector<float> xforms;
    for (const SubMesh& s : subs) {
        m_SubmeshData.push_back(SubmeshData{0u, s.VertexOffset, s.IndexOffset, s.MaterialIndex});
        counts.push_back(s.IndexCount);
        Mat4 w = m_Spec.Transform * s.WorldTransform;       // instance transform of the submesh's TLAS instance
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) xforms.push_back(w.at(r, c));
    }
    std::vector<MaterialData> mats = upload_materials(mesh);
    const std::vector<TextureData>& tex = mesh.GetMeshSource()->GetTextures();
    std::vector<pt_texture_desc> td(tex.size());
    static const uint8_t white[4] = {255, 255, 255, 255};
    for (size_t i = 0; i < tex.size(); i++) {
        bool have = !tex[i].RGBA8.empty();
        td[i].width = have ? tex[i].Width : 1;
        td[i].height = have ? tex[i].Height : 1;
        td[i].flags = tex[i].SamplerFlags;
        td[i].pad_ = 0;
        td[i].rgba8 = have ? tex[i].RGBA8.data() : white;
    }
    pt_scene_desc d{};
    d.vertices = reinterpret_cast<const float*>(mesh.GetVertexBuffer().data());
    d.indices = mesh.GetIndexBuffer().data();
    d.submeshes = reinterpret_cast<const uint32_t*>(m_SubmeshData.data());
    d.submesh_index_counts = counts.data();
    d.submesh_xforms = xforms.data();
    d.materials = reinterpret_cast<const float*>(mats.data());
    d.textures = td.data();
    d.env_rgba = nullptr;
    d.num_vertices = (uint32_t)mesh.GetVertexBuffer().size();
    d.num_indices = (uint32_t)mesh.GetIndexBuffer().size();
    d.num_submeshes = (uint32_t)subs.size();
    d.num_materials = (uint32_t)mats.size();
    d.num_textures = (uint32_t)td.size();
    d.env_face_size = 0;
    m_Device->Check(pt_scene_upload(m_Device->Get(), &d), "pt_scene_upload");
    m_Device->Check(pt_build_accel(m_Device->Get(), buildFlags, &m_Stats), "pt_build_accel");
}

void AccelerationStructure::UpdateMaterialData()
{
    std::vector<MaterialData> mats = upload_materials(*m_Spec.Mesh);
    m_Device->Check(pt_update_materials(m_Device->Get(), reinterpret_cast<const float*>(mats.data()), (uint32_t)mats.size()), "pt_update_materials");
}

void AccelerationStructure::UpdateTransforms(bool refit)
{
    std::vector<float> xforms;
    for (const SubMesh& s : m_Spec.Mesh->GetSubMeshes()) {
        Mat4 w = m_Spec.Transform * s.WorldTransform;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) xforms.push_back(w.at(r, c));
    }
    m_Device->Check(pt_update_transforms(m_Device->Get(), xforms.data(), (uint32_t)(xforms.size() / 12), refit ? PT_UPDATE_REFIT : PT_BUILD_DEFAULT, nullptr), "pt_update_transforms");
}

// ---- RayTracingLayer ------------------------------------------------------------------------------
RayTracingLayer::RayTracingLayer(Ref<Device> device, Ref<Mesh> mesh, const Mat4& transform, const CameraSpecification& cameraSpec)
    : m_Device(std::move(device)), m_Mesh(std::move(mesh)), m_Transform(transform)
{
    m_Camera = CreateRef<Camera>(cameraSpec);
    CreateAccelerationStructure();
    ImageSpecification spec;
    spec.DebugName = "RT-FinalImage";
    m_Image = CreateRef<Image>(m_Device, Image::Kind::Final, spec);
    spec.DebugName = "RT-AccumulationImage";
    m_AccumulationImage = CreateRef<Image>(m_Device, Image::Kind::Accumulation, spec);
    m_SceneBuffer.FrameIndex = 1;        // RayTracingLayer.cpp:120,336
}

void RayTracingLayer::CreateAccelerationStructure()
{
    AccelerationStructureSpecification spec;
    spec.Mesh = m_Mesh;
    spec.Transform = m_Transform;
    m_AccelerationStructure = CreateRef<AccelerationStructure>(m_Device, spec);
}

void RayTracingLayer::SetSubMeshTransform(uint32_t submesh, const Mat4& worldTransform, bool refit)
{
    m_Mesh->GetSubMeshes().at(submesh).WorldTransform = worldTransform;
    m_AccelerationStructure->UpdateTransforms(refit);
}

void RayTracingLayer::SetViewportSize(uint32_t width, uint32_t height)
{
    if (width == m_ViewportWidth && height == m_ViewportHeight && !m_Resized) return;
    m_ViewportWidth = width;
    m_ViewportHeight = height;
    m_Resized = true;
}

void RayTracingLayer::SetParams(const pt_params& p)
{
    m_Device->Check(pt_set_params(m_Device->Get(), &p), "pt_set_params");
    m_SceneBuffer.FrameIndex = 1;
}

void RayTracingLayer::MaterialsChanged()
{
    m_AccelerationStructure->UpdateMaterialData();
    m_SceneBuffer.FrameIndex = 1;        // RayTracingLayer.cpp:620
}

void RayTracingLayer::OnUpdate()
{
    bool moved = m_Camera->Update();
    if (!m_Accumulate || moved || m_UpdateSkyBox) m_SceneBuffer.FrameIndex = 1;   // RayTracingLayer.cpp:356-357
    if (m_UpdateSkyBox) {                                       // RayTracingLayer.cpp:366-380
        if (m_SkyAsEnvironment)
            m_Device->Check(pt_env_preetham(m_Device->Get(), m_SkyboxSettings[0], m_SkyboxSettings[1], m_SkyboxSettings[2], m_SkyFaceSize), "pt_env_preetham");
        m_UpdateSkyBox = false;
    }
}

void RayTracingLayer::SetSkyboxSettings(float turbidity, float azimuth, float inclination, bool useAsEnvironment, uint32_t faceSize)
{
    m_SkyboxSettings[0] = turbidity; m_SkyboxSettings[1] = azimuth; m_SkyboxSettings[2] = inclination;
    m_SkyAsEnvironment = useAsEnvironment;
    m_SkyFaceSize = faceSize;
    m_UpdateSkyBox = true;
}

void RayTracingLayer::OnRender()
{
    if (m_Resized) {                      // RayTracingLayer.cpp:390-397
        m_Device->Check(pt_resize(m_Device->Get(), m_ViewportWidth, m_ViewportHeight), "pt_resize");
        m_Image->Resize(m_ViewportWidth, m_ViewportHeight);
        m_AccumulationImage->Resize(m_ViewportWidth, m_ViewportHeight);
        m_SceneBuffer.FrameIndex = 1;
        m_Resized = false;
    }
    m_Camera->Resize(m_ViewportWidth, m_ViewportHeight);          // RayTracingLayer.cpp:403-411
    m_CameraBuffer.ViewProjection = m_Camera->GetViewProjection();
    m_CameraBuffer.InverseViewProjection = m_Camera->GetInverseViewProjection();
    m_CameraBuffer.View = m_Camera->GetView();
    m_CameraBuffer.InverseView = m_Camera->GetInverseView();
    m_CameraBuffer.InverseProjection = m_Camera->GetInverseProjection();
    m_Device->Check(pt_set_camera(m_Device->Get(), reinterpret_cast<const float*>(&m_CameraBuffer)), "pt_set_camera");
    RayTracingPass();
    PostProcessingPass();                 // RayTracingLayer.cpp:421
}

void RayTracingLayer::PostProcessingPass()
{
    if (m_DoPostProcessing) m_Device->Check(pt_postprocess(m_Device->Get(), m_Exposure), "pt_postprocess");
}

std::vector<uint8_t> RayTracingLayer::ReadPostProcessed()
{
    std::vector<uint8_t> out((size_t)m_ViewportWidth * m_ViewportHeight * 4);
    m_Device->Check(pt_read_postprocessed(m_Device->Get(), out.data()), "pt_read_postprocessed");
    return out;
}

void RayTracingLayer::SetRadianceMap(const std::string& hdrPath, uint32_t faceSize)
{
    HdrImage img = LoadHdr(hdrPath);
    m_Device->Check(pt_env_from_equirect(m_Device->Get(), img.Pixels.data(), img.Width, img.Height, faceSize), "pt_env_from_equirect");
    m_SceneBuffer.FrameIndex = 1;         // a new sky restarts the accumulation (RayTracingLayer.cpp:356-357)
}

void RayTracingLayer::RayTracingPass()
{
    m_Device->Check(pt_trace(m_Device->Get(), m_SceneBuffer.FrameIndex), "pt_trace");
    m_SceneBuffer.FrameIndex++;           // RayTracingLayer.cpp:283
}

uint64_t RayTracingLayer::GetRayCount() const
{
    uint64_t n = 0;
    m_Device->Check(pt_ray_count(m_Device->Get(), &n), "pt_ray_count");
    return n;
}

} // namespace ptamd
