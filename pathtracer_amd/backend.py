"""ctypes binding of libpathtracer_amd.so (include/pathtracer_amd.h): the gfx950 backend.

There is no CPU fallback: if the shared library is missing, `load_library()` raises, and every call
that needs a GPU returns an error code that is turned into `BackendError`.
"""
import ctypes as C
import os

import numpy as np

from .scene import Params, SceneData, SceneDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpathtracer_amd.so")

EXPORTS = [
    "pt_create", "pt_destroy", "pt_last_error", "pt_last_warning", "pt_hardware_queues", "pt_default_params", "pt_scene_upload", "pt_build_accel",
    "pt_update_materials", "pt_update_transforms", "pt_resize", "pt_bind_images", "pt_set_camera", "pt_set_params", "pt_set_precision", "pt_trace", "pt_sync",
    "pt_read_image", "pt_read_accum", "pt_write_accum", "pt_resolve", "pt_read_primary_ids", "pt_ray_count",
    "pt_postprocess", "pt_read_postprocessed", "pt_device_postprocessed", "pt_env_upload", "pt_env_from_equirect", "pt_env_preetham", "pt_read_env",
    "pt_set_chains", "pt_set_batch", "pt_set_profiling", "pt_last_trace_timing", "pt_stream", "pt_wait_stream", "pt_reduce_accum", "pt_device_image", "pt_device_accum",
    "pt_debug_intersect", "pt_intersect_noalloc", "pt_debug_math", "pt_debug_disney_sample", "pt_debug_disney_eval", "pt_debug_read_bvh",
]


class BackendError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("pathtracer_amd error %d: %s" % (code, message))
        self.code = code


class AccelStats(C.Structure):
    _fields_ = [("num_triangles", C.c_uint32), ("num_nodes", C.c_uint32), ("max_depth", C.c_uint32),
                ("num_leaves", C.c_uint32), ("build_ms", C.c_float), ("sah_cost", C.c_float), ("num_references", C.c_uint32)]


class TraceTiming(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("raygen_ms", C.c_float), ("extend_ms", C.c_float), ("shade_ms", C.c_float),
                ("extend_busy_ms", C.c_float), ("chains", C.c_uint32),
                ("extend_launches", C.c_uint32), ("shade_launches", C.c_uint32), ("rays", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64)]


_lib = None


def load_library():
    """Loads the HIP backend.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PATHTRACER_AMD_LIB", LIB_PATH)     # override: A/B runs of differently tuned builds
    if not os.path.exists(path):
        raise ImportError("HIP backend %s is missing: build it with `make -C pathtracer_amd/csrc`" % path)
    lib = C.CDLL(path)
    vp, fp, up = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    lib.pt_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.pt_destroy.argtypes = [vp]
    lib.pt_destroy.restype = None
    lib.pt_last_error.argtypes = [vp]
    lib.pt_last_error.restype = C.c_char_p
    lib.pt_last_warning.argtypes = [vp]
    lib.pt_last_warning.restype = C.c_char_p
    lib.pt_hardware_queues.argtypes = [vp]
    lib.pt_hardware_queues.restype = C.c_uint32
    lib.pt_default_params.argtypes = [C.POINTER(Params)]
    lib.pt_default_params.restype = None
    lib.pt_scene_upload.argtypes = [vp, C.POINTER(SceneDesc)]
    lib.pt_build_accel.argtypes = [vp, C.c_uint32, C.POINTER(AccelStats)]
    lib.pt_update_materials.argtypes = [vp, fp, C.c_uint32]
    lib.pt_resize.argtypes = [vp, C.c_uint32, C.c_uint32]
    lib.pt_bind_images.argtypes = [vp, vp, vp]
    lib.pt_set_camera.argtypes = [vp, fp]
    lib.pt_set_params.argtypes = [vp, C.POINTER(Params)]
    lib.pt_set_precision.argtypes = [vp, C.c_int]
    lib.pt_trace.argtypes = [vp, C.c_uint32]
    lib.pt_sync.argtypes = [vp]
    lib.pt_read_image.argtypes = [vp, fp]
    lib.pt_read_accum.argtypes = [vp, fp]
    lib.pt_write_accum.argtypes = [vp, fp]
    lib.pt_resolve.argtypes = [vp]
    lib.pt_read_primary_ids.argtypes = [vp, up]
    lib.pt_ray_count.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.pt_update_transforms.argtypes = [vp, fp, C.c_uint32, C.c_uint32, C.POINTER(AccelStats)]
    lib.pt_postprocess.argtypes = [vp, C.c_float]
    lib.pt_read_postprocessed.argtypes = [vp, C.POINTER(C.c_uint8)]
    lib.pt_env_from_equirect.argtypes = [vp, fp, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.pt_env_upload.argtypes = [vp, fp, C.c_uint32]
    lib.pt_env_preetham.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_uint32]
    lib.pt_read_env.argtypes = [vp, fp, up]
    lib.pt_set_chains.argtypes = [vp, C.c_uint32]
    lib.pt_set_batch.argtypes = [vp, C.c_uint32]
    lib.pt_set_profiling.argtypes = [vp, C.c_int]
    lib.pt_last_trace_timing.argtypes = [vp, C.POINTER(TraceTiming)]
    for name in ("pt_stream", "pt_device_image", "pt_device_accum", "pt_device_postprocessed"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = vp
    lib.pt_wait_stream.argtypes = [vp, vp]
    lib.pt_reduce_accum.argtypes = [vp, vp, C.c_int]
    lib.pt_debug_intersect.argtypes = [vp, fp, C.c_uint32, up, C.POINTER(C.c_uint64)]
    lib.pt_intersect_noalloc.argtypes = [vp, vp, C.c_uint32, vp]
    lib.pt_debug_math.argtypes = [vp, C.c_int, fp, fp, fp, C.c_uint32]
    lib.pt_debug_disney_sample.argtypes = [vp, C.c_uint32, fp, fp, fp, up, fp, fp, fp]
    lib.pt_debug_disney_eval.argtypes = [vp, C.c_uint32, fp, fp, fp, fp, fp, fp]
    lib.pt_debug_read_bvh.argtypes = [vp, up, up]
    _lib = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class Backend:
    """One rendering context on one MI355X (pt_ctx)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.pt_create(device, C.byref(h))
        if rc:
            raise BackendError(rc, self.lib.pt_last_error(None).decode())
        self.h = h
        self.width = self.height = 0
        self.accel = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.pt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise BackendError(rc, self.lib.pt_last_error(self.h).decode())

    # ---- scene ---------------------------------------------------------------------------
    def scene_upload(self, scene: SceneData):
        desc, keep = scene.as_desc()
        self._check(self.lib.pt_scene_upload(self.h, C.byref(desc)))
        del keep

    def build_accel(self, flags=0):
        st = AccelStats()
        self._check(self.lib.pt_build_accel(self.h, flags, C.byref(st)))
        self.accel = st
        return st

    def update_materials(self, materials):
        m = np.ascontiguousarray(materials, np.float32).reshape(-1, 22)
        self._check(self.lib.pt_update_materials(self.h, _fp(m), len(m)))

    # ---- images / camera / parameters -----------------------------------------------------
    def resize(self, width, height):
        self._check(self.lib.pt_resize(self.h, width, height))
        self.width, self.height = width, height

    def bind_images(self, image_ptr, accum_ptr):
        self._check(self.lib.pt_bind_images(self.h, image_ptr, accum_ptr))

    def set_camera(self, matrices):
        m = np.ascontiguousarray(matrices, np.float32).reshape(-1)
        assert m.size == 80
        self._check(self.lib.pt_set_camera(self.h, _fp(m)))

    def set_params(self, params: Params):
        self._check(self.lib.pt_set_params(self.h, C.byref(params)))

    def set_precision(self, mode):
        """0 = bit-exact shading arithmetic (default), 1 = the hardware's approximate division / square root / sin / cos / log / exp."""
        self._check(self.lib.pt_set_precision(self.h, int(mode)))

    # ---- rendering ------------------------------------------------------------------------
    def trace(self, frame_index):
        self._check(self.lib.pt_trace(self.h, frame_index))

    def sync(self):
        self._check(self.lib.pt_sync(self.h))

    def resolve(self):
        self._check(self.lib.pt_resolve(self.h))

    def read_image(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self.lib.pt_read_image(self.h, _fp(out)))
        return out

    def read_accum(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self.lib.pt_read_accum(self.h, _fp(out)))
        return out

    def write_accum(self, rgba):
        a = np.ascontiguousarray(rgba, np.float32)
        assert a.size == self.width * self.height * 4
        self._check(self.lib.pt_write_accum(self.h, _fp(a)))

    def read_primary_ids(self):
        out = np.empty((self.height, self.width, 2), np.uint32)
        self._check(self.lib.pt_read_primary_ids(self.h, _up(out)))
        return out

    def ray_count(self):
        v = C.c_uint64()
        self._check(self.lib.pt_ray_count(self.h, C.byref(v)))
        return v.value

    def update_transforms(self, xforms, flags=0):
        """New per-submesh 3x4 object-to-world matrices (n, 12); rebuilds the hierarchy, or refits it with flags=0x100 (PT_UPDATE_REFIT)."""
        x = np.ascontiguousarray(xforms, np.float32)
        st = AccelStats()
        self._check(self.lib.pt_update_transforms(self.h, _fp(x), x.size // 12, flags, C.byref(st)))
        return st

    def postprocess(self, exposure=0.8):
        """Exposure, ACES fit, gamma 2.2 -> RGBA8 (H, W, 4) uint8 of the current image."""
        self._check(self.lib.pt_postprocess(self.h, float(exposure)))
        out = np.empty((self.height, self.width, 4), np.uint8)
        self._check(self.lib.pt_read_postprocessed(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def env_from_equirect(self, rgba, face_size):
        """rgba: (H, W, 4) float32 lat-long radiance image, row 0 = top."""
        rgba = np.ascontiguousarray(rgba, np.float32)
        h, w = rgba.shape[:2]
        self._check(self.lib.pt_env_from_equirect(self.h, _fp(rgba), w, h, int(face_size)))

    def env_upload(self, cube):
        """cube: (6, n, n, 4) float32 radiance, faces +X -X +Y -Y +Z -Z; replaces the environment of the uploaded scene."""
        cube = np.ascontiguousarray(cube, np.float32)
        assert cube.ndim == 4 and cube.shape[0] == 6 and cube.shape[1] == cube.shape[2] and cube.shape[3] == 4
        self._check(self.lib.pt_env_upload(self.h, _fp(cube), cube.shape[1]))

    def env_preetham(self, turbidity=3.14, azimuth=0.0, inclination=0.0, face_size=2048):
        self._check(self.lib.pt_env_preetham(self.h, float(turbidity), float(azimuth), float(inclination), int(face_size)))

    def read_env(self):
        n = C.c_uint32(0)
        self._check(self.lib.pt_read_env(self.h, None, C.byref(n)))
        out = np.empty((6, n.value, n.value, 4), np.float32)
        self._check(self.lib.pt_read_env(self.h, _fp(out), C.byref(n)))
        return out

    def set_chains(self, chains):
        """Number of independent pipelines (1, 2, 4 or 8) a frame is split into; images do not depend on it."""
        self._check(self.lib.pt_set_chains(self.h, int(chains)))

    def last_warning(self):
        return self.lib.pt_last_warning(self.h).decode()

    def hardware_queues(self):
        return int(self.lib.pt_hardware_queues(self.h))

    def set_batch(self, frames):
        """Consecutive trace() calls (1..8) that share their kernel launches; images do not depend on it."""
        self._check(self.lib.pt_set_batch(self.h, int(frames)))

    def set_profiling(self, mode):
        """0 off, 1 per-kernel HIP-event timing, 2 timing + traversal node/triangle counters."""
        self._check(self.lib.pt_set_profiling(self.h, int(mode)))

    def last_trace_timing(self):
        t = TraceTiming()
        self._check(self.lib.pt_last_trace_timing(self.h, C.byref(t)))
        return t

    def stream(self):
        return self.lib.pt_stream(self.h)

    def wait_stream(self, stream):
        """Later work of the context runs after what is enqueued so far on `stream` (hipStream_t as int; 0 = null stream)."""
        self._check(self.lib.pt_wait_stream(self.h, C.c_void_p(stream)))

    def reduce_accum(self, nccl_comm, root=0):
        """One in-place RCCL sum-reduce of the accumulation image to `root` (nccl_comm: ncclComm_t as int), on the context's stream."""
        self._check(self.lib.pt_reduce_accum(self.h, C.c_void_p(nccl_comm), int(root)))

    def device_image(self):
        return self.lib.pt_device_image(self.h)

    def device_accum(self):
        return self.lib.pt_device_accum(self.h)

    # ---- inspection hooks -------------------------------------------------------------------
    def debug_intersect(self, rays):
        r = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.empty((len(r), 5), np.uint32)
        counters = (C.c_uint64 * 2)()
        self._check(self.lib.pt_debug_intersect(self.h, _fp(r), len(r), _up(hits), counters))
        return hits, (counters[0], counters[1])

    def intersect_noalloc(self, device_rays_ptr, n, device_hits_ptr):
        """Closest hits of n rays in device memory into device memory, without any allocation (asynchronous: sync() before reading)."""
        self._check(self.lib.pt_intersect_noalloc(self.h, C.c_void_p(device_rays_ptr), n, C.c_void_p(device_hits_ptr)))

    def debug_math(self, fn, x, y=None):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        out = np.empty_like(x)
        yy = None if y is None else np.ascontiguousarray(y, np.float32).reshape(-1)
        self._check(self.lib.pt_debug_math(self.h, fn, _fp(x), None if yy is None else _fp(yy), _fp(out), len(x)))
        return out

    def debug_disney_sample(self, surf, V, N, seeds):
        surf = np.ascontiguousarray(surf, np.float32).reshape(-1, 17)
        n = len(surf)
        V = np.ascontiguousarray(V, np.float32).reshape(n, 3)
        N = np.ascontiguousarray(N, np.float32).reshape(n, 3)
        seeds = np.array(seeds, np.uint32).reshape(n).copy()
        f = np.empty((n, 3), np.float32)
        L = np.empty((n, 3), np.float32)
        pdf = np.empty(n, np.float32)
        self._check(self.lib.pt_debug_disney_sample(self.h, n, _fp(surf), _fp(V), _fp(N), _up(seeds), _fp(f), _fp(L), _fp(pdf)))
        return f, L, pdf, seeds

    def debug_disney_eval(self, surf, V, N, L):
        surf = np.ascontiguousarray(surf, np.float32).reshape(-1, 17)
        n = len(surf)
        V = np.ascontiguousarray(V, np.float32).reshape(n, 3)
        N = np.ascontiguousarray(N, np.float32).reshape(n, 3)
        L = np.ascontiguousarray(L, np.float32).reshape(n, 3)
        f = np.empty((n, 3), np.float32)
        pdf = np.empty(n, np.float32)
        self._check(self.lib.pt_debug_disney_eval(self.h, n, _fp(surf), _fp(V), _fp(N), _fp(L), _fp(f), _fp(pdf)))
        return f, pdf

    def debug_read_bvh(self):
        st = self.accel
        nodes = np.empty((st.num_nodes, 32), np.uint32)
        tris = np.empty((st.num_references, 12), np.uint32)
        self._check(self.lib.pt_debug_read_bvh(self.h, _up(nodes), _up(tris)))
        return nodes, tris
