"""ctypes binding of the C++ host layer (libpt_host.so, include/pathtracer_host.h): glTF scene
load, camera and the headless RayTracingLayer.  Class and method names follow the reference's
call sites (PathTracer/src/RayTracingLayer.cpp:27-42, 350-425)."""
import ctypes as C
import os

import numpy as np

from .scene import Params, SceneData

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpt_host.so")

EXPORTS = [
    "pth_mesh_load", "pth_mesh_free", "pth_mesh_counts", "pth_mesh_tables", "pth_mesh_submesh_name", "pth_mesh_texture_uri", "pth_mesh_texture_sampler",
    "pth_mesh_get_texture", "pth_mesh_set_texture", "pth_mesh_set_material", "pth_camera_matrices", "pth_layer_create", "pth_layer_destroy",
    "pth_layer_set_viewport", "pth_layer_set_params", "pth_layer_set_accumulate", "pth_layer_set_camera_position",
    "pth_layer_materials_changed", "pth_layer_on_update", "pth_layer_on_render", "pth_layer_frame_index",
    "pth_layer_read_image", "pth_layer_read_accum", "pth_layer_ray_count", "pth_layer_last_error",
    "pth_layer_set_submesh_transform", "pth_layer_drag_submesh_transform", "pth_layer_set_radiance_map", "pth_layer_set_skybox_settings", "pth_layer_set_post_processing", "pth_layer_read_postprocessed",
    "pth_hdr_load", "pth_pixels_free", "pth_ppm_write", "pth_image_load", "pth_image_free", "pth_mesh_load_textures",
]


class CameraSpec(C.Structure):
    _fields_ = [("pitch", C.c_float), ("yaw", C.c_float), ("position", C.c_float * 3), ("fov_y", C.c_float),
                ("near_clip", C.c_float), ("far_clip", C.c_float), ("target", C.c_float * 3), ("use_target", C.c_uint32)]


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("host layer %s is missing: build it with `make -C pathtracer_amd/host`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, fp, up = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.pth_mesh_load.restype = vp
    L.pth_mesh_load.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.pth_mesh_free.argtypes = [vp]
    L.pth_mesh_free.restype = None
    L.pth_mesh_counts.argtypes = [vp] + [up] * 5
    L.pth_mesh_counts.restype = None
    L.pth_mesh_tables.argtypes = [vp, fp, fp, up, up, up, fp, fp]
    L.pth_mesh_submesh_name.argtypes = [vp, C.c_uint32]
    L.pth_mesh_submesh_name.restype = C.c_char_p
    L.pth_mesh_texture_uri.argtypes = [vp, C.c_uint32]
    L.pth_mesh_texture_uri.restype = C.c_char_p
    L.pth_mesh_texture_sampler.argtypes = [vp, C.c_uint32]
    L.pth_mesh_texture_sampler.restype = C.c_uint32
    L.pth_mesh_get_texture.argtypes = [vp, C.c_uint32, up, up, C.POINTER(C.c_uint8)]
    L.pth_mesh_set_texture.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)]
    L.pth_mesh_set_material.argtypes = [vp, C.c_uint32, fp]
    L.pth_camera_matrices.argtypes = [C.POINTER(CameraSpec), C.c_uint32, C.c_uint32, fp]
    L.pth_camera_matrices.restype = None
    L.pth_layer_create.restype = vp
    L.pth_layer_create.argtypes = [C.c_int, vp, fp, C.POINTER(CameraSpec), C.c_char_p, C.c_size_t]
    L.pth_layer_destroy.argtypes = [vp]
    L.pth_layer_destroy.restype = None
    L.pth_layer_set_viewport.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.pth_layer_set_params.argtypes = [vp, C.POINTER(Params)]
    L.pth_layer_set_accumulate.argtypes = [vp, C.c_int]
    L.pth_layer_set_camera_position.argtypes = [vp, fp]
    for n in ("pth_layer_materials_changed", "pth_layer_on_update", "pth_layer_on_render"):
        getattr(L, n).argtypes = [vp]
    L.pth_layer_frame_index.argtypes = [vp]
    L.pth_layer_frame_index.restype = C.c_uint32
    L.pth_layer_read_image.argtypes = [vp, fp]
    L.pth_layer_read_accum.argtypes = [vp, fp]
    L.pth_layer_ray_count.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.pth_layer_last_error.argtypes = [vp]
    L.pth_layer_set_submesh_transform.argtypes = [vp, C.c_uint32, fp]
    L.pth_layer_drag_submesh_transform.argtypes = [vp, C.c_uint32, fp]
    L.pth_layer_set_radiance_map.argtypes = [vp, C.c_char_p, C.c_uint32]
    L.pth_layer_set_skybox_settings.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_uint32]
    L.pth_layer_set_post_processing.argtypes = [vp, C.c_int, C.c_float]
    L.pth_layer_read_postprocessed.argtypes = [vp, C.POINTER(C.c_uint8)]
    L.pth_hdr_load.argtypes = [C.c_char_p, up, up, C.POINTER(fp), C.c_char_p, C.c_size_t]
    L.pth_pixels_free.argtypes = [fp]
    L.pth_pixels_free.restype = None
    L.pth_image_load.argtypes = [C.c_char_p, up, up, C.POINTER(C.POINTER(C.c_uint8)), C.c_char_p, C.c_size_t]
    L.pth_image_free.argtypes = [C.POINTER(C.c_uint8)]
    L.pth_image_free.restype = None
    L.pth_mesh_load_textures.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.pth_ppm_write.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), C.c_char_p, C.c_size_t]
    L.pth_layer_last_error.restype = C.c_char_p
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def camera_spec(position=(0, 0, 1), pitch=0.0, yaw=0.0, target=None, fov_y=45.0, near=0.1, far=1000.0):
    s = CameraSpec()
    s.pitch, s.yaw = pitch, yaw
    s.position[:] = position
    s.fov_y, s.near_clip, s.far_clip = fov_y, near, far
    if target is not None:
        s.target[:] = target
        s.use_target = 1
    return s


def camera_matrices(spec, width, height):
    out = np.empty(80, np.float32)
    load_library().pth_camera_matrices(C.byref(spec), width, height, _fp(out))
    return out


class MeshSource:
    """MeshSource(path): glTF 2.0 import by the C++ host layer."""

    def __init__(self, path):
        self.L = load_library()
        err = C.create_string_buffer(512)
        self.h = self.L.pth_mesh_load(path.encode(), err, len(err))
        if not self.h:
            raise RuntimeError("MeshSource(%s): %s" % (path, err.value.decode()))
        self.path = path

    def __del__(self):
        if getattr(self, "h", None):
            self.L.pth_mesh_free(self.h)
            self.h = None

    def counts(self):
        v = [C.c_uint32() for _ in range(5)]
        self.L.pth_mesh_counts(self.h, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def texture_uris(self):
        return [self.L.pth_mesh_texture_uri(self.h, i).decode() for i in range(self.counts()[4])]

    def texture_samplers(self):
        """Sampler state per glTF image as PT_TEX_* flags (wrap of u bits 0-1, of v bits 2-3, bit 4 nearest)."""
        return [int(self.L.pth_mesh_texture_sampler(self.h, i)) for i in range(self.counts()[4])]

    def set_texture(self, i, rgba8):
        t = np.ascontiguousarray(rgba8, np.uint8)
        assert self.L.pth_mesh_set_texture(self.h, i, t.shape[1], t.shape[0], t.ctypes.data_as(C.POINTER(C.c_uint8))) == 0

    def load_textures(self):
        """Decodes the external PNG / JPEG images next to the .gltf; returns how many."""
        err = C.create_string_buffer(512)
        n = self.L.pth_mesh_load_textures(self.h, err, len(err))
        if n < 0:
            raise RuntimeError(err.value.decode())
        return n

    def set_material(self, i, words):
        m = np.ascontiguousarray(words, np.float32)
        assert m.size == 22 and self.L.pth_mesh_set_material(self.h, i, _fp(m)) == 0

    def scene_data(self, layer_transform=None, with_textures=False):
        """The tables in pt_scene_desc layout; layer_transform is a 4x4 (row-major numpy) matrix.  with_textures: the
        decoded images too (load_textures() first); otherwise the materials' map indices are cleared."""
        nv, ni, ns, nm, nt = self.counts()
        vertices = np.empty((nv, 12), np.float32)
        indices = np.empty(ni, np.uint32)
        submeshes = np.empty((ns, 4), np.uint32)
        counts = np.empty(ns, np.uint32)
        xforms = np.empty((ns, 12), np.float32)
        materials = np.empty((nm, 22), np.float32)
        lt = None
        if layer_transform is not None:
            lt = np.ascontiguousarray(np.asarray(layer_transform, np.float32).T.reshape(-1))   # column-major
        up = C.POINTER(C.c_uint32)
        self.L.pth_mesh_tables(self.h, None if lt is None else _fp(lt), _fp(vertices), indices.ctypes.data_as(up),
                               submeshes.ctypes.data_as(up), counts.ctypes.data_as(up), _fp(xforms), _fp(materials))
        names = [self.L.pth_mesh_submesh_name(self.h, i).decode() for i in range(ns)]
        sc = SceneData(vertices, indices, submeshes, counts, xforms, materials, names=names)
        if with_textures:
            # the decoded images with their sampler state; materials keep their map indices only if every image is there
            tex = []
            for i in range(nt):
                w, h = C.c_uint32(), C.c_uint32()
                self.L.pth_mesh_get_texture(self.h, i, C.byref(w), C.byref(h), None)
                if w.value == 0 or h.value == 0:
                    raise RuntimeError("texture %d (%s) has not been decoded: call load_textures() first" % (i, self.texture_uris()[i]))
                t = np.empty((h.value, w.value, 4), np.uint8)
                self.L.pth_mesh_get_texture(self.h, i, C.byref(w), C.byref(h), t.ctypes.data_as(C.POINTER(C.c_uint8)))
                tex.append(t)
            sc.textures = tex
            sc.texture_flags = self.texture_samplers()
        else:
            sc.materials.view(np.int32)[:, 10:13] = -1
        return sc


class RayTracingLayer:
    """Headless counterpart of the reference's layer: OnUpdate / OnRender with its FrameIndex rules."""

    def __init__(self, mesh: MeshSource, transform=None, camera=None, device=0):
        self.L = load_library()
        self.mesh = mesh
        t = np.eye(4, dtype=np.float32) if transform is None else np.asarray(transform, np.float32)
        tc = np.ascontiguousarray(t.T.reshape(-1))
        err = C.create_string_buffer(512)
        cam = camera if camera is not None else camera_spec()
        self.h = self.L.pth_layer_create(device, mesh.h, _fp(tc), C.byref(cam), err, len(err))
        if not self.h:
            raise RuntimeError("RayTracingLayer: " + err.value.decode())
        self.width = self.height = 1

    def close(self):
        if getattr(self, "h", None):
            self.L.pth_layer_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc:
            raise RuntimeError(self.L.pth_layer_last_error(self.h).decode())

    def SetViewportSize(self, w, h):
        self._check(self.L.pth_layer_set_viewport(self.h, w, h))
        self.width, self.height = w, h

    def SetParams(self, p: Params):
        self._check(self.L.pth_layer_set_params(self.h, C.byref(p)))

    def SetAccumulate(self, on):
        self._check(self.L.pth_layer_set_accumulate(self.h, int(on)))

    def SetCameraPosition(self, xyz):
        p = np.asarray(xyz, np.float32)
        self._check(self.L.pth_layer_set_camera_position(self.h, _fp(p)))

    def MaterialsChanged(self):
        self._check(self.L.pth_layer_materials_changed(self.h))

    def OnUpdate(self):
        self._check(self.L.pth_layer_on_update(self.h))

    def OnRender(self):
        self._check(self.L.pth_layer_on_render(self.h))

    @property
    def FrameIndex(self):
        return self.L.pth_layer_frame_index(self.h)

    def ReadImage(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self.L.pth_layer_read_image(self.h, _fp(out)))
        return out

    def ReadAccumulation(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self.L.pth_layer_read_accum(self.h, _fp(out)))
        return out

    def RayCount(self):
        v = C.c_uint64()
        self._check(self.L.pth_layer_ray_count(self.h, C.byref(v)))
        return v.value

    def SetSubMeshTransform(self, submesh, matrix, refit=False):
        """matrix: 4x4 (row-major numpy) WorldTransform of the submesh; the acceleration structure is recreated
        (refit=True: refitted, for an object that is being dragged)."""
        m = np.ascontiguousarray(np.asarray(matrix, np.float32).T.reshape(-1))
        fn = self.L.pth_layer_drag_submesh_transform if refit else self.L.pth_layer_set_submesh_transform
        self._check(fn(self.h, submesh, _fp(m)))

    def SetRadianceMap(self, hdr_path, face_size=1024):
        self._check(self.L.pth_layer_set_radiance_map(self.h, os.fsencode(hdr_path), face_size))

    def SetSkyboxSettings(self, turbidity=3.14, azimuth=0.0, inclination=0.0, use_as_environment=False, face_size=2048):
        """New Preetham parameters: the next OnUpdate restarts the accumulation and (optionally) re-renders the sky cube."""
        self._check(self.L.pth_layer_set_skybox_settings(self.h, turbidity, azimuth, inclination, int(use_as_environment), face_size))

    def SetPostProcessing(self, on=True, exposure=0.8):
        self._check(self.L.pth_layer_set_post_processing(self.h, int(on), exposure))

    def ReadPostProcessed(self):
        out = np.empty((self.height, self.width, 4), np.uint8)
        self._check(self.L.pth_layer_read_postprocessed(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out


def load_hdr(path):
    """Radiance .hdr -> (H, W, 4) float32, row 0 = top, alpha 1 (C++ reader of the host layer)."""
    L = load_library()
    w, h, px = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_float)()
    err = C.create_string_buffer(512)
    if L.pth_hdr_load(os.fsencode(path), C.byref(w), C.byref(h), C.byref(px), err, len(err)):
        raise RuntimeError(err.value.decode())
    try:
        return np.ctypeslib.as_array(px, shape=(h.value, w.value, 4)).copy()
    finally:
        L.pth_pixels_free(px)


def load_image(path):
    """PNG or baseline JPEG -> (H, W, 4) uint8 (C++ decoders of the host layer)."""
    L = load_library()
    w, h, px = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
    err = C.create_string_buffer(512)
    if L.pth_image_load(os.fsencode(path), C.byref(w), C.byref(h), C.byref(px), err, len(err)):
        raise RuntimeError(err.value.decode())
    try:
        return np.ctypeslib.as_array(px, shape=(h.value, w.value, 4)).copy()
    finally:
        L.pth_image_free(px)


def write_ppm(path, rgba8):
    L = load_library()
    a = np.ascontiguousarray(rgba8, np.uint8)
    err = C.create_string_buffer(512)
    if L.pth_ppm_write(os.fsencode(path), a.shape[1], a.shape[0], a.ctypes.data_as(C.POINTER(C.c_uint8)), err, len(err)):
        raise RuntimeError(err.value.decode())
