"""Per-ray cost of the shade kernel by material mix: the benchmark scene with its full-Disney materials against the same
geometry with plain diffuse + dielectric materials (every wave then runs one lobe set).  Prints rays, frame time and the
summed kernel times per frame."""
import os
import sys
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 16:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time

from pathtracer_amd import procedural
from pathtracer_amd.backend import Backend
from pathtracer_amd.scene import Params

W, H = 1920, 1080
for full in (True, False):
    sc = procedural.sponza_class(full_disney=full)
    b = Backend(0)
    b.scene_upload(sc)
    b.build_accel()
    b.resize(W, H)
    b.set_camera(procedural.sponza_camera(W, H))
    b.set_params(Params.reference(spp_per_frame=1, max_bounces=8))
    for f in range(1, 10):
        b.trace(f)
    b.sync()
    r0 = b.ray_count()
    t0 = time.perf_counter()
    for f in range(10, 74):
        b.trace(f)
    b.sync()
    dt = time.perf_counter() - t0
    rays = b.ray_count() - r0
    b.set_profiling(1)
    for f in range(74, 90):
        b.trace(f)
    b.sync()
    t = b.last_trace_timing()
    print("full_disney=%s: %.1f Mrays/s, %.3f ms/frame, %.2f M rays/frame; per frame extend_sum %.2f ms shade_sum %.2f ms; per M rays: extend %.3f shade %.3f"
          % (full, rays / dt / 1e6, dt / 64 * 1e3, rays / 64 / 1e6, t.extend_ms / 16, t.shade_ms / 16, t.extend_ms / 16 / (t.rays / 16 / 1e6), t.shade_ms / 16 / (t.rays / 16 / 1e6)))
    b.close()
