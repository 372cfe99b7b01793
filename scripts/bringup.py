"""First GPU bring-up: compares the HIP backend with the CPU oracle on the Cornell fixture."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd.backend import Backend
from pathtracer_amd.scene import SceneData, Params
from pathtracer_amd import camera
from tests import oracle_lib

sc = SceneData.load(os.path.join(ROOT, "tests/golden/cornell_scene.npz"))
b = Backend(0)
print("created", flush=True)
# math
rng = np.random.default_rng(1)
x = rng.uniform(0, 2 * np.pi, 100000).astype(np.float32)
for fn, xs in [(0, x), (1, x), (2, np.exp(rng.uniform(-80, 80, 100000)).astype(np.float32)), (3, rng.uniform(-100, 88, 100000).astype(np.float32))]:
    g = b.debug_math(fn, xs); c = oracle_lib.det_math(fn, xs)
    print("math", fn, "bit-equal:", np.array_equal(g.view(np.uint32), c.view(np.uint32)), flush=True)
xa = rng.uniform(0, 1, 100000).astype(np.float32); ya = rng.uniform(0, 3, 100000).astype(np.float32)
print("pow bit-equal:", np.array_equal(b.debug_math(4, xa, ya).view(np.uint32), oracle_lib.det_math(4, xa, ya).view(np.uint32)), flush=True)

b.scene_upload(sc)
st = b.build_accel()
print("accel: tris %d nodes %d depth %d leaves %d build %.3f ms sah %.2f" % (st.num_triangles, st.num_nodes, st.max_depth, st.num_leaves, st.build_ms, st.sah_cost), flush=True)

# random rays
n = 200000
o = rng.uniform([-0.3, 0.0, -0.58], [0.25, 0.53, 0.6], (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, np.full((n, 1), 1e-5, np.float32), d, np.full((n, 1), 1e27, np.float32)], axis=1).astype(np.float32)
t = time.time(); hg, cnt = b.debug_intersect(rays); tg = time.time() - t
orc = oracle_lib.Oracle(sc)
t = time.time(); hc = orc.intersect(rays, use_bvh=True); tc = time.time() - t
print("intersect equal:", np.array_equal(hg, hc), "mismatch rows:", int((hg != hc).any(axis=1).sum()), "gpu %.3fs cpu %.3fs nodes/ray %.1f tris/ray %.1f" % (tg, tc, cnt[0] / n, cnt[1] / n), flush=True)

for (W, H, S, B, frames) in [(128, 128, 1, 4, 3), (96, 64, 5, 20, 2), (512, 512, 1, 4, 2)]:
    p = Params.reference(spp_per_frame=S, max_bounces=B)
    cam = camera.cornell_camera(W, H)
    b.resize(W, H); b.set_camera(cam); b.set_params(p)
    orc.resize(W, H); orc.set_camera(cam); orc.set_params(p)
    r0 = b.ray_count(); c0 = orc.ray_count()
    for f in range(1, frames + 1):
        b.trace(f); orc.trace(f)
    ig, ic = b.read_image(), orc.read_image()
    ag, ac = b.read_accum(), orc.read_accum()
    pg, pc = b.read_primary_ids(), orc.read_primary_ids()
    print("%dx%d S%d B%d: image bit-equal %s accum bit-equal %s ids equal %s rays gpu %d cpu %d; max abs diff %g, differing px %d" % (
        W, H, S, B, np.array_equal(ig.view(np.uint32), ic.view(np.uint32)), np.array_equal(ag.view(np.uint32), ac.view(np.uint32)),
        np.array_equal(pg, pc), b.ray_count() - r0, orc.ray_count() - c0, np.nanmax(np.abs(ig - ic)), int((ig != ic).any(axis=2).sum())), flush=True)

# timing
W, H = 1920, 1080
p = Params.reference(spp_per_frame=1, max_bounces=8)
b.resize(W, H); b.set_camera(camera.cornell_camera(W, H)); b.set_params(p)
b.trace(1); b.sync()
b.set_profiling(True)
r0 = b.ray_count(); t = time.time()
for f in range(2, 12):
    b.trace(f)
b.sync(); dt = time.time() - t
rays = b.ray_count() - r0
tt = b.last_trace_timing()
print("1080p 8 bounces: %.1f Mrays/s (%d rays, %.3f s); last frame total %.3f ms raygen %.3f extend %.3f (%d) shade %.3f (%d) rays %d" % (
    rays / dt / 1e6, rays, dt, tt.total_ms, tt.raygen_ms, tt.extend_ms, tt.extend_launches, tt.shade_ms, tt.shade_launches, tt.rays), flush=True)
