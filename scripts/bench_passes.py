"""Times the three streaming passes (post-process, equirect -> cube, Preetham sky) and the CPU oracle beside them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathtracer_amd.backend import Backend
from pathtracer_amd.scene import SceneData
from tests import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = Backend(0)
b.scene_upload(SceneData.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz")))
W, H = 1920, 1080
b.resize(W, H)
rng = np.random.default_rng(0)
img = np.exp(rng.uniform(-6, 3, (H, W, 4))).astype(np.float32)
b.write_accum(img); b.resolve(); b.sync()


def timed(fn, reps):
    fn(); b.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    b.sync()
    return (time.perf_counter() - t0) / reps


t = timed(lambda: b.lib.pt_postprocess(b.h, 0.8), 200)
print("postprocess 1920x1080: %.1f us, %.0f GB/s (20 B/pixel)" % (t * 1e6, W * H * 20 / t / 1e9))
t0 = time.perf_counter(); oracle_lib.postprocess(img, 0.8); tc = time.perf_counter() - t0
print("  CPU oracle, 1 thread: %.1f ms" % (tc * 1e3))
n = 2048
t = timed(lambda: b.env_preetham(3.14, 0.0, 0.0, n), 10)
print("preetham 6x%dx%d (incl. the cube's reallocation): %.1f us, %.0f GB/s (16 B/texel)" % (n, n, t * 1e6, 6 * n * n * 16 / t / 1e9))
t0 = time.perf_counter(); oracle_lib.preetham_sky(3.14, 0.0, 0.0, 256); tc = time.perf_counter() - t0
print("  CPU oracle, 1 thread, scaled from 256^2 faces: %.0f ms" % (tc * 1e3 * (n / 256) ** 2))
eq = np.exp(rng.uniform(-3, 3, (2048, 4096, 4))).astype(np.float32)
t = timed(lambda: b.env_from_equirect(eq, n), 5)
print("equirect 4096x2048 -> 6x%dx%d (incl. the 134 MB host-to-device copy): %.1f ms" % (n, n, t * 1e3))
t0 = time.perf_counter(); oracle_lib.equirect_to_cube(eq, 256); tc = time.perf_counter() - t0
print("  CPU oracle, 1 thread, scaled from 256^2 faces: %.0f ms" % (tc * 1e3 * (n / 256) ** 2))
b.close()
