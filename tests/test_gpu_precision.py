"""The stated-tolerance arithmetic of the shading stage (pt_set_precision(ctx, PT_PRECISION_FAST)) against the bit-exact
default and against the CPU oracle, on a real MI355X.

The reference's shaders run under GLSL's precision rules (division 2.5 ulp, inversesqrt 2 ulp, sin / cos 2^-11 absolute,
pow / exp / log a few ulp), so two conforming Vulkan drivers already differ in the last bits of every BSDF value, and a path
whose lobe choice (`r3 < cdf`) or next hit flips on such a difference continues elsewhere.  The tolerances stated here:

  * one BSDF evaluation (DisneyEval, Disney.glsl:252-359) over random materials, normals and directions: f and pdf relative to
    the bit-exact value (+ 5e-4 absolute): median <= 1e-6, 99.9 % of the evaluations <= 5e-4, every evaluation <= 5e-2 (the
    tail is ill-conditioned input: 1 - cos^2 near grazing angles, (F - F0) / (1 - F0), refraction next to total internal
    reflection amplify a last-bit difference; measured maxima 1e-2 for f, 2e-3 for pdf); sampled directions (DisneySample,
    Disney.glsl:150-250): 99.9 % within 1e-4 absolute, all within 2e-3 except samples whose lobe choice flips (fewer than
    1 in 10 000; none in 10^6 measured);
  * a frame: primary hit ids identical (camera rays and traversal stay bit-exact); ray count within 0.1 %;
  * per-sample radiance of one-bounce paths (camera ray, hit, emission + one sky or emitter look-up): within 1e-4 relative;
  * a converged image (1024 samples per pixel): relRMSE = sqrt(sum (a - b)^2 / sum b^2) <= 1e-3 against the CPU oracle.
"""
import numpy as np
import pytest

from pathtracer_amd.scene import Params
from tests import oracle_lib
from tests.conftest import random_frames, random_surfaces

pytestmark = pytest.mark.gpu

EXACT, FAST = 0, 1


@pytest.fixture()
def fast(backend):
    backend.set_precision(FAST)
    yield backend
    backend.set_precision(EXACT)


@pytest.mark.parametrize("kind", ["diffuse", "metal", "mixed", "full", "glass"])
def test_bsdf_evaluation_within_the_stated_tolerance(backend, kind):
    rng = np.random.default_rng(11)
    n = 100_000
    surf = random_surfaces(rng, n, kind)
    V, N = random_frames(rng, n)
    _, L = random_frames(rng, n)
    backend.set_precision(EXACT)
    f0, p0 = backend.debug_disney_eval(surf, V, N, L)
    backend.set_precision(FAST)
    try:
        f1, p1 = backend.debug_disney_eval(surf, V, N, L)
    finally:
        backend.set_precision(EXACT)
    ok = np.isfinite(f0).all(1) & np.isfinite(p0)
    assert ok.mean() > 0.99
    # where the bit-exact value is not finite (unguarded 1/0 in the shader) the fast one is not finite either
    assert (~np.isfinite(f1[~ok]).all(1) | ~np.isfinite(p1[~ok])).all()
    ef = np.nan_to_num(np.abs(f1[ok] - f0[ok]) / (np.abs(f0[ok]) + 5e-4), nan=9.0)
    ep = np.nan_to_num(np.abs(p1[ok] - p0[ok]) / (np.abs(p0[ok]) + 5e-4), nan=9.0)
    assert ef.max() <= 5e-2 and ep.max() <= 5e-2, (ef.max(), ep.max())
    assert np.quantile(ef, 0.999) <= 5e-4 and np.quantile(ep, 0.999) <= 5e-4
    assert np.median(ep[p0[ok] > 0]) <= 1e-6 and np.median(ef) <= 1e-6


@pytest.mark.parametrize("kind", ["diffuse", "mixed", "full", "glass"])
def test_sampled_directions_within_the_stated_tolerance(backend, kind):
    rng = np.random.default_rng(12)
    n = 100_000
    surf = random_surfaces(rng, n, kind)
    V, N = random_frames(rng, n)
    seeds = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    backend.set_precision(EXACT)
    f0, L0, p0, s0 = backend.debug_disney_sample(surf, V, N, seeds)
    backend.set_precision(FAST)
    try:
        f1, L1, p1, s1 = backend.debug_disney_sample(surf, V, N, seeds)
    finally:
        backend.set_precision(EXACT)
    assert np.array_equal(s0, s1)                       # three draws per sample in both modes
    ok = np.isfinite(L0).all(1) & np.isfinite(L1).all(1)
    d = np.abs(L1[ok] - L0[ok]).max(1)
    flipped = d > 1e-2
    assert flipped.mean() < 1e-4, flipped.mean()        # a lobe choice that flips on the last bit of a cdf
    assert d[~flipped].max() <= 2e-3 and np.quantile(d, 0.999) <= 1e-4, (d[~flipped].max(), np.quantile(d, 0.999))
    # total internal reflection gives normalize(0) = NaN in both modes
    assert np.array_equal(np.isfinite(L0).all(1), np.isfinite(L1).all(1))


def _render(backend, cam, W, H, frames, **params):
    backend.resize(W, H)
    backend.set_camera(cam)
    backend.set_params(Params.reference(**params))
    r0 = backend.ray_count()
    for f in frames:
        backend.trace(f)
    return backend.ray_count() - r0


def test_frame_primary_ids_and_one_bounce_radiance(backend, cornell, cameras):
    W, H = 256, 256
    backend.scene_upload(cornell)
    backend.build_accel()
    img = np.exp(np.random.default_rng(9).uniform(-2, 3, (64, 128, 4))).astype(np.float32)
    backend.env_from_equirect(img, 64)            # a sky that differs from texel to texel (and is finite everywhere)
    cam = cameras["512x512"]
    out = {}
    for mode in (EXACT, FAST):
        backend.set_precision(mode)
        try:
            rays1 = _render(backend, cam, W, H, (1, 2), spp_per_frame=1, max_bounces=1)
            acc1 = backend.read_accum()
            ids = backend.read_primary_ids()
            rays8 = _render(backend, cam, W, H, (1, 2, 3), spp_per_frame=2, max_bounces=8)
            out[mode] = (rays1, acc1, ids, rays8, backend.read_accum())
        finally:
            backend.set_precision(EXACT)
    (r1e, a1e, ide, r8e, a8e), (r1f, a1f, idf, r8f, a8f) = out[EXACT], out[FAST]
    assert np.array_equal(ide, idf) and r1e == r1f
    assert np.array_equal(a1e[..., 3], a1f[..., 3])
    # one-bounce paths: emission of the hit surface, or one sky look-up
    rel = np.abs(a1f[..., :3] - a1e[..., :3]) / (np.abs(a1e[..., :3]) + 1e-6)
    assert rel.max() <= 1e-4, rel.max()
    assert abs(r8f - r8e) <= 1e-3 * r8e
    # most pixels of a short accumulation agree closely; the rest hold a path that went elsewhere
    rel8 = np.abs(a8f[..., :3] - a8e[..., :3]).sum(-1) / (np.abs(a8e[..., :3]).sum(-1) + 1e-6)
    assert np.median(rel8) < 1e-5 and (rel8 > 1e-2).mean() < 0.01


def test_converged_image_within_relrmse_1e3_of_the_oracle(fast, cornell, cameras):
    """Cornell box 128x128, 1 sample per pixel and frame, 8 bounces, frames 1..1025 (1024 accumulated), same per-pixel seeds
    (RayGen.glsl:314-315) on both sides: GPU with the fast shading arithmetic against the CPU oracle."""
    W, H, frames = 128, 128, range(1, 1026)
    cam = cameras["512x512"]
    p = Params.reference(spp_per_frame=1, max_bounces=8)
    fast.scene_upload(cornell)
    fast.build_accel()
    fast.resize(W, H)
    fast.set_camera(cam)
    fast.set_params(p)
    o = oracle_lib.Oracle(cornell)
    o.resize(W, H)
    o.set_camera(cam)
    o.set_params(p)
    for f in frames:
        fast.trace(f)
        o.trace(f)
    a, b = fast.read_image()[..., :3].astype(np.float64), o.read_image()[..., :3].astype(np.float64)
    acc = fast.read_accum()
    assert (acc[..., 3] == 1024).all()
    red = (a == (1.0, 0.0, 0.0)).all(-1) | (b == (1.0, 0.0, 0.0)).all(-1)      # NaN pixels, painted red by both
    assert red.mean() < 0.01
    a, b = a[~red], b[~red]
    rel_rmse = np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum())
    assert rel_rmse <= 1e-3, rel_rmse
    o.close()
