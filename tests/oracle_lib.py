"""ctypes wrapper of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")


def _has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    name = "liboracle.so" if _has_fma() else "liboracle_nofma.so"
    path = os.path.join(_ORACLE_DIR, name)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp, fp, up = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.orc_create.restype = vp
    L.orc_create.argtypes = [vp]
    L.orc_destroy.argtypes = [vp]
    L.orc_destroy.restype = None
    L.orc_update_materials.argtypes = [vp, fp, C.c_uint32]
    L.orc_resize.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.orc_set_camera.argtypes = [vp, fp]
    L.orc_set_params.argtypes = [vp, vp]
    L.orc_trace.argtypes = [vp, C.c_uint32, C.c_int, C.c_int]
    L.orc_read_image.argtypes = [vp, fp]
    L.orc_read_accum.argtypes = [vp, fp]
    L.orc_write_accum.argtypes = [vp, fp]
    L.orc_read_primary_ids.argtypes = [vp, up]
    L.orc_ray_count.argtypes = [vp]
    L.orc_ray_count.restype = C.c_uint64
    L.orc_traversal_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_traversal_stats.restype = None
    L.orc_num_triangles.argtypes = [vp]
    L.orc_num_triangles.restype = C.c_uint32
    L.orc_intersect.argtypes = [vp, fp, C.c_uint32, up, C.c_int, C.c_int]
    L.orc_pcg_hash.argtypes = [up]
    L.orc_pcg_hash.restype = C.c_uint32
    L.orc_random_value.argtypes = [up]
    L.orc_random_value.restype = C.c_float
    L.orc_det_math.argtypes = [C.c_int, fp, fp, fp, C.c_uint32]
    L.orc_det_math.restype = None
    L.orc_postprocess.argtypes = [fp, C.c_uint32, C.c_float, C.POINTER(C.c_uint8)]
    L.orc_postprocess.restype = None
    L.orc_sample_texture.argtypes = [vp, fp, C.c_uint32, fp]
    L.orc_sample_texture.restype = None
    L.orc_equirect_to_cube.argtypes = [fp, C.c_uint32, C.c_uint32, C.c_uint32, fp]
    L.orc_equirect_to_cube.restype = None
    L.orc_preetham_sky.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint32, fp]
    L.orc_preetham_sky.restype = None
    L.orc_disney_sample.argtypes = [fp, fp, fp, up, fp, fp, fp]
    L.orc_disney_sample.restype = None
    L.orc_disney_eval.argtypes = [fp, fp, fp, fp, fp, fp]
    L.orc_disney_eval.restype = None
    L.orc_closest_hit.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, fp, fp]
    L.orc_trace_path.argtypes = [vp, fp, fp, up, C.c_int, fp]
    L.orc_trace_path.restype = None
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class Oracle:
    """CPU restatement of the reference's path tracer over one scene."""

    def __init__(self, scene, threads=None):
        self.L = lib()
        self.scene = scene
        desc, keep = scene.as_desc()      # same struct layout as the product's pt_scene_desc
        self.h = self.L.orc_create(C.byref(desc))
        del keep
        self.threads = threads or min(os.cpu_count() or 1, 16)
        self.width = self.height = 0

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def resize(self, w, h):
        self.L.orc_resize(self.h, w, h)
        self.width, self.height = w, h

    def set_camera(self, m):
        m = np.ascontiguousarray(m, np.float32).reshape(-1)
        self.L.orc_set_camera(self.h, _fp(m))

    def set_params(self, params):
        self.L.orc_set_params(self.h, C.byref(params))

    def update_materials(self, m):
        m = np.ascontiguousarray(m, np.float32)
        assert self.L.orc_update_materials(self.h, _fp(m), m.size // 22) == 0

    def trace(self, frame, use_bvh=True):
        assert self.L.orc_trace(self.h, frame, self.threads, 1 if use_bvh else 0) == 0

    def read_image(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self.L.orc_read_image(self.h, _fp(out))
        return out

    def read_accum(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self.L.orc_read_accum(self.h, _fp(out))
        return out

    def write_accum(self, a):
        a = np.ascontiguousarray(a, np.float32)
        self.L.orc_write_accum(self.h, _fp(a))

    def read_primary_ids(self):
        out = np.empty((self.height, self.width, 2), np.uint32)
        self.L.orc_read_primary_ids(self.h, _up(out))
        return out

    def ray_count(self):
        return int(self.L.orc_ray_count(self.h))

    def traversal_stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.L.orc_traversal_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def num_triangles(self):
        return int(self.L.orc_num_triangles(self.h))

    def intersect(self, rays, use_bvh=True):
        r = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.empty((len(r), 5), np.uint32)
        self.L.orc_intersect(self.h, _fp(r), len(r), _up(hits), 1 if use_bvh else 0, self.threads)
        return hits


PAYLOAD_FIELDS = ["Distance", "Albedo.x", "Albedo.y", "Albedo.z", "Roughness", "Metallic", "Emission.x", "Emission.y", "Emission.z",
                  "WorldNormal.x", "WorldNormal.y", "WorldNormal.z", "Anisotropic", "Subsurface", "SpecularTint", "Sheen", "SheenTint",
                  "Clearcoat", "ClearcoatRoughness", "SpecTrans", "ior", "ax", "ay", "eta"]


def closest_hit(oracle, submesh, prim, u, v, t, direction):
    """Payload of ClosestHit.glsl for one hit, as a dict of the fields downstream code reads."""
    d = np.ascontiguousarray(direction, np.float32)
    out = np.zeros(26, np.float32)
    assert oracle.L.orc_closest_hit(oracle.h, submesh, prim, u, v, t, _fp(d), _fp(out)) == 0
    return dict(zip(PAYLOAD_FIELDS, out[:24].tolist()))


def trace_path(oracle, origin, direction, seed, use_bvh=False):
    """TracePath of one ray: (radiance (3,), advanced seed)."""
    o = np.ascontiguousarray(origin, np.float32)
    d = np.ascontiguousarray(direction, np.float32)
    s = C.c_uint32(int(seed))
    rad = np.zeros(3, np.float32)
    oracle.L.orc_trace_path(oracle.h, _fp(o), _fp(d), C.byref(s), 1 if use_bvh else 0, _fp(rad))
    return rad, s.value


def pcg_hash(seed):
    s = C.c_uint32(seed)
    v = lib().orc_pcg_hash(C.byref(s))
    return v, s.value


def random_value(seed):
    s = C.c_uint32(seed)
    v = lib().orc_random_value(C.byref(s))
    return v, s.value


def det_math(fn, x, y=None):
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    y = np.zeros_like(x) if y is None else np.ascontiguousarray(y, np.float32).reshape(-1)
    out = np.empty_like(x)
    lib().orc_det_math(fn, _fp(x), _fp(y), _fp(out), len(x))
    return out


def sample_texture(texels, uv, flags=0):
    """texture(sampler2D, uv) of one (h, w, 4) uint8 texture with sampler state `flags` at (n, 2) coordinates."""
    from pathtracer_amd.scene import TextureDesc
    t = np.ascontiguousarray(texels, np.uint8)
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    d = TextureDesc()
    d.height, d.width, d.flags = t.shape[0], t.shape[1], flags
    d.rgba8 = t.ctypes.data_as(C.POINTER(C.c_uint8))
    out = np.empty((len(uv), 4), np.float32)
    lib().orc_sample_texture(C.byref(d), _fp(uv), len(uv), _fp(out))
    return out


def postprocess(rgba, exposure=0.8):
    rgba = np.ascontiguousarray(rgba, np.float32)
    out = np.empty(rgba.shape[:-1] + (4,), np.uint8)
    lib().orc_postprocess(_fp(rgba), rgba.size // 4, exposure, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def equirect_to_cube(rgba, face_size):
    rgba = np.ascontiguousarray(rgba, np.float32)
    h, w = rgba.shape[:2]
    out = np.empty((6, face_size, face_size, 4), np.float32)
    lib().orc_equirect_to_cube(_fp(rgba), w, h, face_size, _fp(out))
    return out


def preetham_sky(turbidity, azimuth, inclination, face_size):
    out = np.empty((6, face_size, face_size, 4), np.float32)
    lib().orc_preetham_sky(turbidity, azimuth, inclination, face_size, _fp(out))
    return out


def disney_sample(surf, V, N, seeds):
    surf = np.ascontiguousarray(surf, np.float32).reshape(-1, 17)
    n = len(surf)
    V = np.ascontiguousarray(V, np.float32).reshape(n, 3)
    N = np.ascontiguousarray(N, np.float32).reshape(n, 3)
    seeds = np.array(seeds, np.uint32).reshape(n).copy()
    f = np.empty((n, 3), np.float32)
    L = np.empty((n, 3), np.float32)
    pdf = np.empty(n, np.float32)
    Lb = lib()
    for i in range(n):
        s = C.c_uint32(int(seeds[i]))
        p = C.c_float()
        Lb.orc_disney_sample(_fp(surf[i]), _fp(V[i]), _fp(N[i]), C.byref(s), _fp(f[i]), _fp(L[i]), C.byref(p))
        seeds[i] = s.value
        pdf[i] = p.value
    return f, L, pdf, seeds


def disney_eval(surf, V, N, Lin):
    surf = np.ascontiguousarray(surf, np.float32).reshape(-1, 17)
    n = len(surf)
    V = np.ascontiguousarray(V, np.float32).reshape(n, 3)
    N = np.ascontiguousarray(N, np.float32).reshape(n, 3)
    Lin = np.ascontiguousarray(Lin, np.float32).reshape(n, 3)
    f = np.empty((n, 3), np.float32)
    pdf = np.empty(n, np.float32)
    Lb = lib()
    for i in range(n):
        p = C.c_float()
        Lb.orc_disney_eval(_fp(surf[i]), _fp(V[i]), _fp(N[i]), _fp(Lin[i]), _fp(f[i]), C.byref(p))
        pdf[i] = p.value
    return f, pdf
