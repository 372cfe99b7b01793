// C++ host layer above the C ABI (include/pathtracer_amd.h).
//
// It keeps the names and call shapes of the reference's scene-load, camera, image and
// acceleration-structure API as they appear at its call sites (the engine submodule that defines
// them, VulkanLibrary, is absent from the reference checkout, so the surface is the one
// PathTracer/src/RayTracingLayer.cpp uses):
//   MeshSource(path), Mesh(source)            RayTracingLayer.cpp:27-32
//   GetVertexBuffer / GetIndexBuffer          RayTracingLayer.cpp:229,235
//   GetSubMeshes()[i].{Name, MaterialIndex, WorldTransform}, GetMaterialBuffers()   :531-604
//   Camera(CameraSpecification{pitch,yaw}), Update, Resize, SetPosition, Get*       :38-42,354,403-409,632-635
//   Image(ImageSpecification{Width,Height,Format,Usage,DebugName}), Resize          :98-116,392-394
//   AccelerationStructure(spec{Mesh, Transform}), UpdateMaterialData               :342-348,621
//   RayTracingLayer::OnUpdate / OnRender / RayTracingPass frame sequencing         :201-284,350-425
// The Vulkan objects behind them are replaced by one pt_ctx of the HIP backend.
#pragma once
#include "../../include/pathtracer_amd.h"
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace ptamd {

template <class T> using Ref = std::shared_ptr<T>;
template <class T, class... A> Ref<T> CreateRef(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }

// column-major 4x4, the layout of glm::mat4
struct Mat4 {
    float m[16];
    static Mat4 Identity();
    static Mat4 Scale(float x, float y, float z);
    static Mat4 Translate(float x, float y, float z);
    static Mat4 Perspective(float fovYDegrees, float aspect, float zNear, float zFar);   // depth 0..1, Y flipped: image row 0 is the top
    static Mat4 LookAt(const float eye[3], const float target[3], const float up[3]);
    Mat4 operator*(const Mat4& b) const;
    Mat4 Inverse() const;
    float& at(int row, int col) { return m[col * 4 + row]; }
    float at(int row, int col) const { return m[col * 4 + row]; }
};

// 48-byte vertex the closest-hit stage reads (ClosestHit.glsl:17-24,56-81)
struct Vertex {
    float Position[3];
    float TextureCoords[2];
    float Normal[3];
    float Tangent[4];
};
static_assert(sizeof(Vertex) == 48, "vertex stride is fixed by ClosestHit.glsl:56");

// 88-byte material record (ClosestHit.glsl:26-48)
struct MaterialData {
    float AlbedoValue[3];
    float MetallicValue;
    float RoughnessValue;
    float EmissiveValue[3];
    float EmissiveStrength;
    uint32_t UseNormalMap;
    int32_t AlbedoMapIndex;
    int32_t MetallicRoughnessMapIndex;
    int32_t NormalMapIndex;
    float Anisotropic, Subsurface, SpecularTint, Sheen, SheenTint, Clearcoat, ClearcoatRoughness, SpecTrans, ior;
};
static_assert(sizeof(MaterialData) == 88, "material stride is fixed by ClosestHit.glsl:115");

struct SubMesh {
    std::string Name;
    uint32_t VertexOffset = 0, IndexOffset = 0, IndexCount = 0, VertexCount = 0;
    uint32_t MaterialIndex = 0;
    Mat4 WorldTransform = Mat4::Identity();
};

struct TextureData {
    std::string Uri;              // as written in the glTF file
    uint32_t Width = 0, Height = 0;
    std::vector<uint8_t> RGBA8;   // empty until a decoder fills it
    uint32_t SamplerFlags = 0;    // PT_TEX_* of pathtracer_amd.h, from the glTF sampler of the first texture that uses the image
};

// glTF 2.0 import: .gltf JSON with embedded (base64) or external buffers.
class MeshSource {
public:
    explicit MeshSource(const std::string& path);
    const std::vector<Vertex>& GetVertices() const { return m_Vertices; }
    const std::vector<uint32_t>& GetIndices() const { return m_Indices; }
    const std::vector<SubMesh>& GetSubMeshes() const { return m_SubMeshes; }
    std::vector<SubMesh>& GetSubMeshes() { return m_SubMeshes; }
    const std::vector<MaterialData>& GetMaterials() const { return m_Materials; }
    std::vector<MaterialData>& GetMaterials() { return m_Materials; }
    const std::vector<TextureData>& GetTextures() const { return m_Textures; }
    std::vector<TextureData>& GetTextures() { return m_Textures; }
    const std::string& GetPath() const { return m_Path; }
    size_t LoadTextures();        // decodes the external PNG / JPEG images next to the .gltf; returns how many

private:
    std::string m_Path;
    std::vector<Vertex> m_Vertices;
    std::vector<uint32_t> m_Indices;
    std::vector<SubMesh> m_SubMeshes;
    std::vector<MaterialData> m_Materials;
    std::vector<TextureData> m_Textures;
};

class Mesh {
public:
    explicit Mesh(Ref<MeshSource> source) : m_Source(std::move(source)) {}
    Ref<MeshSource> GetMeshSource() const { return m_Source; }
    const std::vector<Vertex>& GetVertexBuffer() const { return m_Source->GetVertices(); }
    const std::vector<uint32_t>& GetIndexBuffer() const { return m_Source->GetIndices(); }
    std::vector<SubMesh>& GetSubMeshes() { return m_Source->GetSubMeshes(); }
    std::vector<MaterialData>& GetMaterialBuffers() { return m_Source->GetMaterials(); }

private:
    Ref<MeshSource> m_Source;
};

struct CameraSpecification {
    float pitch = 0.0f, yaw = 0.0f;          // radians; yaw from +Z towards +X, positive pitch looks down
    float position[3] = {0.0f, 0.0f, 1.0f};
    float fovY = 45.0f, nearClip = 0.1f, farClip = 1000.0f;
};

class Camera {
public:
    explicit Camera(const CameraSpecification& spec = CameraSpecification());
    bool Update();                            // true when the pose changed since the last call
    void Resize(uint32_t width, uint32_t height);
    void SetPosition(const std::array<float, 3>& p);
    void SetPitchYaw(float pitch, float yaw);
    void LookAt(const std::array<float, 3>& eye, const std::array<float, 3>& target);
    const Mat4& GetViewProjection() const { return m_ViewProjection; }
    const Mat4& GetInverseViewProjection() const { return m_InverseViewProjection; }
    const Mat4& GetView() const { return m_View; }
    const Mat4& GetInverseView() const { return m_InverseView; }
    const Mat4& GetProjection() const { return m_Projection; }
    const Mat4& GetInverseProjection() const { return m_InverseProjection; }
    std::array<float, 3> GetPosition() const { return {m_Spec.position[0], m_Spec.position[1], m_Spec.position[2]}; }
    std::array<float, 2> GetPitchYaw() const { return {m_Spec.pitch, m_Spec.yaw}; }

private:
    void Recalculate();
    CameraSpecification m_Spec;
    uint32_t m_Width = 1, m_Height = 1;
    bool m_Dirty = true;
    bool m_UseLookAt = false;
    float m_Target[3] = {0, 0, 0};
    Mat4 m_View, m_Projection, m_ViewProjection, m_InverseView, m_InverseProjection, m_InverseViewProjection;
};

// CameraBuffer of RayTracingLayer.h:17-24: five mat4, 320 bytes
struct CameraBuffer {
    Mat4 ViewProjection, InverseViewProjection, View, InverseView, InverseProjection;
};
static_assert(sizeof(CameraBuffer) == 320, "CameraBuffer is five mat4");

// One GPU context shared by the objects below (the reference's Application::GetVulkanDevice()).
class Device {
public:
    explicit Device(int ordinal = 0);
    ~Device();
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    pt_ctx* Get() const { return m_Ctx; }
    void Check(int rc, const char* what) const;

private:
    pt_ctx* m_Ctx = nullptr;
};

enum class ImageFormat { RGBA32F, RGBA8 };
enum class ImageUsage { STORAGE_IMAGE_2D };
struct ImageSpecification {
    uint32_t Width = 1, Height = 1;
    ImageFormat Format = ImageFormat::RGBA32F;
    ImageUsage Usage = ImageUsage::STORAGE_IMAGE_2D;
    std::string DebugName;
};

// m_Image / m_AccumulationImage of the reference: a view on the context's images with a host copy on demand.
class Image {
public:
    enum class Kind { Final, Accumulation };
    Image(Ref<Device> device, Kind kind, const ImageSpecification& spec);
    void Resize(uint32_t width, uint32_t height);     // both images of a context share one size
    uint32_t GetWidth() const { return m_Spec.Width; }
    uint32_t GetHeight() const { return m_Spec.Height; }
    const std::vector<float>& Download();              // RGBA32F, row 0 first
    void Upload(const std::vector<float>& rgba);       // accumulation image only (checkpoint / resume)

private:
    Ref<Device> m_Device;
    Kind m_Kind;
    ImageSpecification m_Spec;
    std::vector<float> m_Host;
};

struct AccelerationStructureSpecification {
    Ref<ptamd::Mesh> Mesh;
    Mat4 Transform = Mat4::Identity();
};

struct SubmeshData { uint32_t BufferIndex, VertexOffset, IndexOffset, MaterialIndex; };   // ClosestHit.glsl:147-150

class AccelerationStructure {
public:
    AccelerationStructure(Ref<Device> device, const AccelerationStructureSpecification& spec, uint32_t buildFlags = PT_BUILD_DEFAULT);
    const AccelerationStructureSpecification& GetSpecification() const { return m_Spec; }
    const std::vector<SubmeshData>& GetSubmeshDataStorageBuffer() const { return m_SubmeshData; }
    const std::vector<MaterialData>& GetMaterialBuffer() const { return m_Spec.Mesh->GetMaterialBuffers(); }
    const std::vector<TextureData>& GetTextures() const { return m_Spec.Mesh->GetMeshSource()->GetTextures(); }
    const pt_accel_stats& GetStats() const { return m_Stats; }
    void UpdateMaterialData();
    void UpdateTransforms(bool refit = false);                                   // new instance transforms, hierarchy rebuilt                         // re-uploads the mesh's material buffers

private:
    Ref<Device> m_Device;
    AccelerationStructureSpecification m_Spec;
    std::vector<SubmeshData> m_SubmeshData;
    pt_accel_stats m_Stats{};
};

struct SceneBuffer {                                   // RayTracingLayer.h:26-33
    uint32_t FrameIndex = 1;
    float padding0 = 0, padding1 = 0, padding2 = 0;
    float AbsorptionFactor[3] = {1, 1, 1};
};

// image_decode.cpp: PNG (zlib) and baseline JPEG -> RGBA8
TextureData DecodeImage(const std::vector<uint8_t>& bytes);
TextureData LoadImageFile(const std::string& path);

// image_io.cpp
struct HdrImage {
    uint32_t Width = 0, Height = 0;
    std::vector<float> Pixels;     // RGBA32F, row 0 = top, alpha 1
};
HdrImage LoadHdr(const std::string& path);                                               // Radiance RGBE, flat or run-length encoded
void WritePpm(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rgba8);

// Headless counterpart of the reference's RayTracingLayer: same members, same frame sequencing.
class RayTracingLayer {
public:
    RayTracingLayer(Ref<Device> device, Ref<Mesh> mesh, const Mat4& transform, const CameraSpecification& cameraSpec = CameraSpecification());
    void SetViewportSize(uint32_t width, uint32_t height);    // the ImGui viewport panel's size
    void OnUpdate();                                           // RayTracingLayer.cpp:350-381
    void OnRender();                                           // RayTracingLayer.cpp:383-425
    void SetAccumulate(bool on) { m_Accumulate = on; }
    void SetParams(const pt_params& p);
    void MaterialsChanged();                                   // RayTracingLayer.cpp:618-622
    // the transform editor of RayTracingLayer.cpp:569-604 with "Automatically update AS" on: new WorldTransform of a
    // submesh, acceleration structure recreated; like the reference it leaves FrameIndex alone
    // refit: keep the hierarchy's topology (an object being dragged); otherwise the hierarchy is rebuilt, as the reference does
    void SetSubMeshTransform(uint32_t submesh, const Mat4& worldTransform, bool refit = false);
    void SetRadianceMap(const std::string& hdrPath, uint32_t faceSize = 1024);   // RayTracingLayer.cpp:50-54 (TextureCube of the engine)
    // the sky sliders of RayTracingLayer.cpp:526-529: new Preetham parameters (turbidity, azimuth, inclination) mark the sky
    // dirty; the next OnUpdate restarts the accumulation and runs the sky pass (RayTracingLayer.cpp:356-357,366-380).  The
    // reference renders that cube but never binds it to the ray-tracing set (:258 binds the .hdr radiance map), so by
    // default the pass is skipped here; with useAsEnvironment the Preetham cube becomes the environment the paths see.
    void SetSkyboxSettings(float turbidity, float azimuth, float inclination, bool useAsEnvironment, uint32_t faceSize = 2048);
    void SetPostProcessing(bool on, float exposure) { m_DoPostProcessing = on; m_Exposure = exposure; }   // RayTracingLayer.cpp:522-524
    std::vector<uint8_t> ReadPostProcessed();                  // m_PostProcessingImage, RGBA8
    Ref<Camera> GetCamera() const { return m_Camera; }
    Ref<Image> GetImage() const { return m_Image; }
    Ref<Image> GetAccumulationImage() const { return m_AccumulationImage; }
    Ref<AccelerationStructure> GetAccelerationStructure() const { return m_AccelerationStructure; }
    uint32_t GetFrameIndex() const { return m_SceneBuffer.FrameIndex; }
    uint64_t GetRayCount() const;

private:
    void RayTracingPass();                                     // RayTracingLayer.cpp:201-284
    void PostProcessingPass();                                 // RayTracingLayer.cpp:286-318
    void CreateAccelerationStructure();                        // RayTracingLayer.cpp:342-348
    Ref<Device> m_Device;
    Ref<Mesh> m_Mesh;
    Mat4 m_Transform;
    Ref<Camera> m_Camera;
    CameraBuffer m_CameraBuffer;
    Ref<AccelerationStructure> m_AccelerationStructure;
    Ref<Image> m_Image, m_AccumulationImage;
    bool m_Accumulate = true;
    SceneBuffer m_SceneBuffer;
    uint32_t m_ViewportWidth = 1, m_ViewportHeight = 1;
    bool m_Resized = true;
    float m_SkyboxSettings[3] = {3.14f, 0.0f, 0.0f};           // RayTracingLayer.h:88
    bool m_UpdateSkyBox = true;                                // RayTracingLayer.h:89
    bool m_SkyAsEnvironment = false;
    uint32_t m_SkyFaceSize = 2048;                             // RayTracingLayer.cpp:58-61
    bool m_DoPostProcessing = true;                            // RayTracingLayer.h:90
    float m_Exposure = 0.8f;                                   // RayTracingLayer.h:92
};

} // namespace ptamd
