/* Flat C view of the C++ host layer (pathtracer_amd/host/host.h) for foreign-function callers
 * (the Python tests use it through ctypes).  The host layer mirrors the reference's scene-load,
 * camera and layer API (PathTracer/src/RayTracingLayer.cpp:27-42,201-284,342-425); the GPU work
 * itself goes through include/pathtracer_amd.h. */
#ifndef PATHTRACER_HOST_H
#define PATHTRACER_HOST_H
#include <stddef.h>
#include <stdint.h>
#include "pathtracer_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_mesh pth_mesh;     /* MeshSource + Mesh */
typedef struct pth_layer pth_layer;   /* Device + RayTracingLayer */

typedef struct pth_camera_spec {      /* CameraSpecification (RayTracingLayer.cpp:38-42) */
    float pitch, yaw;
    float position[3];
    float fov_y, near_clip, far_clip;
    float target[3];
    uint32_t use_target;              /* 1: look from position to target instead of pitch/yaw */
} pth_camera_spec;

/* MeshSource(path): glTF 2.0 import.  Returns NULL and fills err on failure. */
pth_mesh* pth_mesh_load(const char* gltf_path, char* err, size_t errlen);
void pth_mesh_free(pth_mesh* m);
void pth_mesh_counts(pth_mesh* m, uint32_t* num_vertices, uint32_t* num_indices, uint32_t* num_submeshes,
                     uint32_t* num_materials, uint32_t* num_textures);
/* Copies the tables in pt_scene_desc layout.  layer_transform: column-major mat4 multiplied onto every
 * submesh's world transform (AccelerationStructureSpecification::Transform), or NULL. */
int pth_mesh_tables(pth_mesh* m, const float* layer_transform, float* vertices, uint32_t* indices,
                    uint32_t* submeshes, uint32_t* index_counts, float* xforms, float* materials);
const char* pth_mesh_submesh_name(pth_mesh* m, uint32_t i);
const char* pth_mesh_texture_uri(pth_mesh* m, uint32_t i);
/* sampler state of glTF image i as PT_TEX_* flags (from the sampler of the first glTF texture that uses the image) */
uint32_t pth_mesh_texture_sampler(pth_mesh* m, uint32_t i);
/* supplies decoded RGBA8 texels for glTF image i (this layer has no JPEG/PNG decoder) */
/* the decoded texels of image i (after pth_mesh_load_textures or pth_mesh_set_texture): size always; texels when rgba8 != NULL
 * (width*height*4 bytes; 0 x 0 = not decoded) */
int pth_mesh_get_texture(pth_mesh* m, uint32_t i, uint32_t* width, uint32_t* height, uint8_t* rgba8);
int pth_mesh_set_texture(pth_mesh* m, uint32_t i, uint32_t width, uint32_t height, const uint8_t* rgba8);
/* edits one 22-word material record (the ImGui material sliders, RayTracingLayer.cpp:536-564) */
int pth_mesh_set_material(pth_mesh* m, uint32_t i, const float* words22);

/* Camera: the five CameraBuffer matrices for a viewport size */
void pth_camera_matrices(const pth_camera_spec* spec, uint32_t width, uint32_t height, float* out80);

/* RayTracingLayer on device `device` */
pth_layer* pth_layer_create(int device, pth_mesh* mesh, const float* transform16, const pth_camera_spec* camera,
                            char* err, size_t errlen);
void pth_layer_destroy(pth_layer* l);
int pth_layer_set_viewport(pth_layer* l, uint32_t width, uint32_t height);
int pth_layer_set_params(pth_layer* l, const pt_params* p);
int pth_layer_set_accumulate(pth_layer* l, int on);
int pth_layer_set_camera_position(pth_layer* l, const float* xyz);
int pth_layer_materials_changed(pth_layer* l);
int pth_layer_on_update(pth_layer* l);
int pth_layer_on_render(pth_layer* l);
uint32_t pth_layer_frame_index(pth_layer* l);
int pth_layer_read_image(pth_layer* l, float* rgba);
int pth_layer_read_accum(pth_layer* l, float* rgba);
int pth_layer_ray_count(pth_layer* l, uint64_t* rays);
const char* pth_layer_last_error(pth_layer* l);
/* transform editor (RayTracingLayer.cpp:569-604, "Automatically update AS" on): new WorldTransform of a submesh as a
 * column-major mat4, acceleration structure recreated */
int pth_layer_set_submesh_transform(pth_layer* l, uint32_t submesh, const float* column_major16);
/* the same edit while the gizmo is being dragged: the hierarchy is refitted (topology kept, boxes recomputed), which
 * is several times faster; finish the drag with pth_layer_set_submesh_transform to get a rebuilt hierarchy */
int pth_layer_drag_submesh_transform(pth_layer* l, uint32_t submesh, const float* column_major16);
/* TextureCube{path = "...hdr"} (RayTracingLayer.cpp:50-54): load a Radiance image and make it the environment */
int pth_layer_set_radiance_map(pth_layer* l, const char* hdr_path, uint32_t face_size);
/* m_SkyboxSettings + m_UpdateSkyBox (RayTracingLayer.h:88-89, sliders at RayTracingLayer.cpp:526-529): new Preetham
 * parameters; the next on_update restarts the accumulation (RayTracingLayer.cpp:356-357) and, with use_as_environment,
 * makes the Preetham cube (face_size texels per face; the reference uses 2048) the environment of the ray-tracing pass */
int pth_layer_set_skybox_settings(pth_layer* l, float turbidity, float azimuth, float inclination, int use_as_environment, uint32_t face_size);
/* m_DoPostProcessing / m_Exposure (RayTracingLayer.h:90-92, RayTracingLayer.cpp:522-524); on by default, 0.8 */
int pth_layer_set_post_processing(pth_layer* l, int on, float exposure);
int pth_layer_read_postprocessed(pth_layer* l, uint8_t* rgba8);   /* m_PostProcessingImage after the last on_render */

/* Radiance .hdr (RGBE) reader: *rgba = malloc'ed width*height*4 floats (row 0 = top, alpha 1), free with pth_pixels_free */
int pth_hdr_load(const char* path, uint32_t* width, uint32_t* height, float** rgba, char* err, size_t errlen);
void pth_pixels_free(float* rgba);
/* PNG (non-interlaced) or baseline JPEG -> malloc'ed width*height*4 bytes RGBA8, free with pth_image_free */
int pth_image_load(const char* path, uint32_t* width, uint32_t* height, uint8_t** rgba8, char* err, size_t errlen);
void pth_image_free(uint8_t* rgba8);
/* decodes every glTF image of the mesh that is an external PNG / JPEG file next to the .gltf (what the engine's
 * MeshSource does on import); returns the number decoded, -1 on error.  Images left undecoded count as absent. */
int pth_mesh_load_textures(pth_mesh* m, char* err, size_t errlen);
/* binary PPM (P6) of an RGBA8 image, alpha dropped */
int pth_ppm_write(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba8, char* err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
