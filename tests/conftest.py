import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cornell():
    from pathtracer_amd.scene import SceneData
    return SceneData.load(os.path.join(GOLDEN, "cornell_scene.npz"))


@pytest.fixture(scope="session")
def cameras():
    """Committed CameraBuffer matrices for the Cornell view, keyed "WxH"."""
    z = np.load(os.path.join(GOLDEN, "cornell_cameras.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def backend():
    """One context on cuda:0 for the whole GPU session (a box allows few processes on its card)."""
    from pathtracer_amd.backend import Backend
    b = Backend(0)
    yield b
    b.close()


def random_rays(rng, n, lo, hi, tmin=1e-5, tmax=1e27):
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, np.full((n, 1), tmin, np.float32), d, np.full((n, 1), tmax, np.float32)], axis=1).astype(np.float32)


def random_surfaces(rng, n, kind):
    """n x 17 BSDF inputs (albedo3 metallic roughness subsurface specularTint sheen sheenTint clearcoat
    clearcoatRoughness specTrans ior ax ay eta anisotropic) the way ClosestHit.glsl:198-226 derives them."""
    s = np.zeros((n, 17), np.float32)
    s[:, 0:3] = rng.uniform(0.05, 1, (n, 3))
    s[:, 4] = rng.uniform(0.02, 1, n)
    s[:, 12] = 1.5
    if kind == "metal":
        s[:, 3] = 1.0
    elif kind == "mixed":
        s[:, 3] = rng.uniform(0, 1, n)
    elif kind == "full":
        s[:, 3] = rng.uniform(0, 1, n)
        s[:, 5:12] = rng.uniform(0, 1, (n, 7))
        s[:, 12] = rng.uniform(1.1, 2.0, n)
        s[:, 16] = rng.uniform(0, 1, n)
    elif kind == "glass":
        s[:, 11] = 1.0
        s[:, 4] = rng.uniform(0.02, 0.4, n)
    aspect = np.sqrt(np.float32(1) - s[:, 16] * np.float32(0.9)).astype(np.float32)
    s[:, 13] = np.maximum(np.float32(0.001), s[:, 4] / aspect)
    s[:, 14] = np.maximum(np.float32(0.001), s[:, 4] * aspect)
    inside = rng.uniform(0, 1, n) < 0.3
    s[:, 15] = np.where(inside, np.float32(1) / s[:, 12], s[:, 12])
    return s


def random_frames(rng, n):
    N = rng.normal(size=(n, 3)).astype(np.float32)
    N /= np.linalg.norm(N, axis=1, keepdims=True)
    V = rng.normal(size=(n, 3)).astype(np.float32)
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    flip = (V * N).sum(1) < 0
    N[flip] *= -1
    return V, N
