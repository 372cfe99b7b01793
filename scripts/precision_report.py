"""Error distribution of the fast shading arithmetic (pt_set_precision 1) against the bit-exact default: BSDF evaluation,
sampled directions, one frame, a converged Cornell image.  The tolerances in tests/test_gpu_precision.py come from this."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd.backend import Backend  # noqa: E402
from pathtracer_amd.scene import Params, SceneData  # noqa: E402
from tests.conftest import random_frames, random_surfaces  # noqa: E402

b = Backend(0)
rng = np.random.default_rng(11)
n = 200_000
for kind in ["diffuse", "metal", "mixed", "full", "glass"]:
    surf = random_surfaces(rng, n, kind)
    V, N = random_frames(rng, n)
    _, L = random_frames(rng, n)
    seeds = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    b.set_precision(0)
    f0, p0 = b.debug_disney_eval(surf, V, N, L)
    sf0, sL0, sp0, _ = b.debug_disney_sample(surf, V, N, seeds)
    b.set_precision(1)
    f1, p1 = b.debug_disney_eval(surf, V, N, L)
    sf1, sL1, sp1, _ = b.debug_disney_sample(surf, V, N, seeds)
    ok = np.isfinite(f0).all(1) & np.isfinite(p0)
    nf = (~np.isfinite(f1[ok]).all(1) | ~np.isfinite(p1[ok])).sum()
    ef = np.abs(f1[ok] - f0[ok]) / (np.abs(f0[ok]) + 5e-4)
    ep = np.abs(p1[ok] - p0[ok]) / (np.abs(p0[ok]) + 5e-4)
    ef, ep = np.nan_to_num(ef, nan=9.0), np.nan_to_num(ep, nan=9.0)
    okl = np.isfinite(sL0).all(1) & np.isfinite(sL1).all(1)
    d = np.abs(sL1[okl] - sL0[okl]).max(1)
    flipped = d > 1e-3
    print("%-8s eval: finite %.4f (fast non-finite among them %d); f rel max %.3g p99.9 %.3g median %.3g; pdf rel max %.3g p99.9 %.3g median %.3g | "
          "sample: L max (unflipped) %.3g, flipped %d of %d, finiteness equal %s"
          % (kind, ok.mean(), nf, ef.max(), np.quantile(ef, 0.999), np.median(ef), ep.max(), np.quantile(ep, 0.999), np.median(ep[p0[ok] > 0]),
             d[~flipped].max(), flipped.sum(), okl.sum(), np.array_equal(np.isfinite(sL0).all(1), np.isfinite(sL1).all(1))))

golden = os.path.join(ROOT, "tests", "golden")
scene = SceneData.load(os.path.join(golden, "cornell_scene.npz"))
cam = np.load(os.path.join(golden, "cornell_cameras.npz"))["512x512"]
b.scene_upload(scene)
b.build_accel()
W = H = 128
imgs = {}
for mode in (0, 1):
    b.set_precision(mode)
    b.resize(W, H)
    b.set_camera(cam)
    b.set_params(Params.reference(spp_per_frame=1, max_bounces=8))
    t = time.time()
    r0 = b.ray_count()
    for f in range(1, 1026):
        b.trace(f)
    imgs[mode] = b.read_image()[..., :3].astype(np.float64)
    print("mode %d: 1025 frames in %.2f s, %d rays" % (mode, time.time() - t, b.ray_count() - r0))
a, c = imgs[1], imgs[0]
red = (a == (1.0, 0.0, 0.0)).all(-1) | (c == (1.0, 0.0, 0.0)).all(-1)
print("red pixels:", red.sum(), " relRMSE fast vs exact over 1024 spp: %.3g" % np.sqrt(((a[~red] - c[~red]) ** 2).sum() / (c[~red] ** 2).sum()))
b.close()
