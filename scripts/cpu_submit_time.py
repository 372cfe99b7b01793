"""How long the host spends inside pt_trace (submission only) against the frame time."""
import os, sys, time
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd.backend import Backend
from pathtracer_amd.scene import Params
scene, cam, _ = bench.make_workload("sponza_class", 1920, 1080)
b = Backend(0)
b.scene_upload(scene); b.build_accel(); b.resize(1920, 1080); b.set_camera(cam)
b.set_params(Params.reference(spp_per_frame=1, max_bounces=8))
for f in range(1, 6):
    b.trace(f)
b.sync()
N = 64
t0 = time.perf_counter()
sub = 0.0
for f in range(6, 6 + N):
    a = time.perf_counter()
    b.trace(f)
    sub += time.perf_counter() - a
t1 = time.perf_counter()
b.sync()
t2 = time.perf_counter()
print("submit %.3f ms/frame, loop %.3f ms/frame, total with sync %.3f ms/frame" % (sub / N * 1e3, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
