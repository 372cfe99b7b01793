"""The C-ABI library loads and exports every symbol include/pathtracer_amd.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from pathtracer_amd import backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pathtracer_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ("pt_create", "pt_scene_upload", "pt_build_accel", "pt_trace", "pt_read_image", "pt_read_accum",
                 "pt_write_accum", "pt_read_primary_ids", "pt_ray_count", "pt_update_materials", "pt_resize", "pt_set_camera"):
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = backend.load_library()
    for name in declared_functions():
        assert hasattr(lib, name), "libpathtracer_amd.so does not export %s" % name
    assert sorted(backend.EXPORTS) == declared_functions()


def test_struct_sizes_match_the_header():
    from pathtracer_amd.scene import Params, SceneDesc, TextureDesc
    assert C.sizeof(Params) == 28
    assert C.sizeof(TextureDesc) == 24
    assert C.sizeof(SceneDesc) == 8 * 8 + 6 * 4
    assert C.sizeof(backend.AccelStats) == 28
    assert C.sizeof(backend.TraceTiming) == 56


def test_default_params_are_the_reference_constants():
    from pathtracer_amd.scene import Params
    lib = backend.load_library()
    p = Params()
    lib.pt_default_params(C.byref(p))
    # RayGen.glsl:111 MAX_BOUNCES, :319 SAMPLE_COUNT, :132 sky x10, :159 EPS, :341-342 tmin/tmax; RR compiled out
    assert (p.spp_per_frame, p.max_bounces, p.rr_depth) == (5, 20, 0)
    assert p.sky_scale == 10.0 and abs(p.eps - 0.0003) < 1e-9 and abs(p.tmin - 1e-5) < 1e-12 and p.tmax == C.c_float(1e27).value


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(backend.BackendError) as e:
        backend.Backend(0)
    assert e.value.code == 3
    assert "device" in str(e.value).lower()
