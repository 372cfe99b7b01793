"""Scene tables in the layout the reference's shaders read (PathTracer/assets/shaders/RayTracing/
ClosestHit.glsl:11-15,56-81,113-150) plus what its engine keeps inside the TLAS (per-submesh index
count and object-to-world transform, PathTracer/src/RayTracingLayer.cpp:342-348).

`SceneData` is a plain container of numpy arrays; `as_desc()` turns it into the C struct that
`pt_scene_upload` (include/pathtracer_amd.h) takes.
"""
import ctypes as C

import numpy as np

MATERIAL_WORDS = 22
VERTEX_FLOATS = 12


class TextureDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("flags", C.c_uint32), ("pad_", C.c_uint32),
                ("rgba8", C.POINTER(C.c_uint8))]


class SceneDesc(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("indices", C.POINTER(C.c_uint32)),
                ("submeshes", C.POINTER(C.c_uint32)), ("submesh_index_counts", C.POINTER(C.c_uint32)),
                ("submesh_xforms", C.POINTER(C.c_float)), ("materials", C.POINTER(C.c_float)),
                ("textures", C.POINTER(TextureDesc)), ("env_rgba", C.POINTER(C.c_float)),
                ("num_vertices", C.c_uint32), ("num_indices", C.c_uint32), ("num_submeshes", C.c_uint32),
                ("num_materials", C.c_uint32), ("num_textures", C.c_uint32), ("env_face_size", C.c_uint32)]


class Params(C.Structure):
    """Constants of RayGen.glsl (:111 MAX_BOUNCES, :132 sky x10, :159 EPS, :319 SAMPLE_COUNT, :341-342 tmin/tmax)."""
    _fields_ = [("spp_per_frame", C.c_uint32), ("max_bounces", C.c_uint32), ("rr_depth", C.c_uint32),
                ("sky_scale", C.c_float), ("eps", C.c_float), ("tmin", C.c_float), ("tmax", C.c_float)]

    @classmethod
    def reference(cls, **over):
        p = cls(5, 20, 0, 10.0, 0.0003, 0.00001, 1e27)
        for k, v in over.items():
            setattr(p, k, v)
        return p


def make_material(albedo=(1, 1, 1), metallic=0.0, roughness=1.0, emissive=(0, 0, 0), emissive_strength=1.0,
                  albedo_map=-1, mr_map=-1, normal_map=-1, anisotropic=0.0, subsurface=0.0, specular_tint=0.0,
                  sheen=0.0, sheen_tint=0.0, clearcoat=0.0, clearcoat_roughness=0.0, spec_trans=0.0, ior=1.5):
    """One 22-word material record (ClosestHit.glsl:26-48).  Words 9..12 are integers stored as bits."""
    m = np.zeros(MATERIAL_WORDS, np.float32)
    m[0:3] = albedo
    m[3] = metallic
    m[4] = roughness
    m[5:8] = emissive
    m[8] = emissive_strength
    ints = m.view(np.int32)
    ints[9] = 0
    ints[10] = albedo_map
    ints[11] = mr_map
    ints[12] = normal_map
    m[13:22] = (anisotropic, subsurface, specular_tint, sheen, sheen_tint, clearcoat, clearcoat_roughness, spec_trans, ior)
    return m


class SceneData:
    def __init__(self, vertices, indices, submeshes, index_counts, xforms, materials, textures=(), env=None, names=None):
        self.vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1, VERTEX_FLOATS)
        self.indices = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        self.submeshes = np.ascontiguousarray(submeshes, np.uint32).reshape(-1, 4)
        self.index_counts = np.ascontiguousarray(index_counts, np.uint32).reshape(-1)
        self.xforms = np.ascontiguousarray(xforms, np.float32).reshape(-1, 12)
        self.materials = np.ascontiguousarray(materials, np.float32).reshape(-1, MATERIAL_WORDS)
        self.textures = [np.ascontiguousarray(t, np.uint8) for t in textures]   # each (h, w, 4)
        self.texture_flags = [0] * len(self.textures)   # sampler state per texture: PT_TEX_* of include/pathtracer_amd.h
        self.env = None if env is None else np.ascontiguousarray(env, np.float32)   # (6, n, n, 4)
        self.names = list(names) if names is not None else ["submesh%d" % i for i in range(len(self.submeshes))]
        assert len(self.submeshes) == len(self.index_counts) == len(self.xforms)

    @property
    def num_triangles(self):
        return int(self.index_counts.sum() // 3)

    def as_desc(self):
        """Returns (SceneDesc, keepalive).  The arrays of `self` must outlive the call that uses the struct."""
        fp = C.POINTER(C.c_float)
        up = C.POINTER(C.c_uint32)
        d = SceneDesc()
        d.vertices = self.vertices.ctypes.data_as(fp)
        d.indices = self.indices.ctypes.data_as(up)
        d.submeshes = self.submeshes.ctypes.data_as(up)
        d.submesh_index_counts = self.index_counts.ctypes.data_as(up)
        d.submesh_xforms = self.xforms.ctypes.data_as(fp)
        d.materials = self.materials.ctypes.data_as(fp)
        tex = (TextureDesc * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            tex[i].height, tex[i].width = t.shape[0], t.shape[1]
            tex[i].flags = int(self.texture_flags[i])
            tex[i].rgba8 = t.ctypes.data_as(C.POINTER(C.c_uint8))
        d.textures = tex
        d.num_textures = len(self.textures)
        if self.env is not None:
            d.env_rgba = self.env.ctypes.data_as(fp)
            d.env_face_size = self.env.shape[1]
        else:
            d.env_rgba = None
            d.env_face_size = 0
        d.num_vertices = len(self.vertices)
        d.num_indices = len(self.indices)
        d.num_submeshes = len(self.submeshes)
        d.num_materials = len(self.materials)
        return d, tex

    def save(self, path):
        """Everything `as_desc` hands over, textures with their sampler flags and the env cube included."""
        extra = {"texture_%d" % i: t for i, t in enumerate(self.textures)}
        if self.env is not None:
            extra["env"] = self.env
        np.savez_compressed(path, vertices=self.vertices, indices=self.indices, submeshes=self.submeshes,
                            index_counts=self.index_counts, xforms=self.xforms, materials=self.materials,
                            names=np.array(self.names), texture_flags=np.array(self.texture_flags, np.uint32), **extra)

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        flags = [int(f) for f in z["texture_flags"]] if "texture_flags" in z.files else []
        sc = cls(z["vertices"], z["indices"], z["submeshes"], z["index_counts"], z["xforms"], z["materials"],
                 textures=[z["texture_%d" % i] for i in range(len(flags))],
                 env=z["env"] if "env" in z.files else None, names=[str(s) for s in z["names"]])
        sc.texture_flags = flags
        return sc
