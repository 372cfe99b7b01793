"""pt_intersect_noalloc (the entry point that allocates nothing: its traversal scratch is a fixed array inside
the library) against pt_debug_intersect, through the C ABI: same rays, hit records equal word for word, on two
calls in a row, the second with fewer rays, so that what the first call left in the shared scratch is seen to do
no harm."""
import numpy as np
import pytest
import torch  # before the backend library is loaded: both then share the HIP runtime that torch ships

from pathtracer_amd.backend import BackendError
from tests.conftest import random_rays

pytestmark = pytest.mark.gpu


def _query(backend, rays):
    d_rays = torch.from_numpy(rays).to("cuda:0")
    d_hits = torch.full((len(rays), 5), 0x5A5A5A5A, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    backend.intersect_noalloc(d_rays.data_ptr(), len(rays), d_hits.data_ptr())
    backend.sync()
    return d_hits.cpu().numpy().view(np.uint32)


def test_no_alloc_ray_query_equals_the_allocating_one(backend, cornell):
    backend.scene_upload(cornell)
    backend.build_accel()
    rng = np.random.default_rng(11)
    lo, hi = cornell_bounds(cornell)
    big = random_rays(rng, 300_000, lo, hi)
    small = random_rays(rng, 777, lo, hi)
    want_big, _ = backend.debug_intersect(big)
    want_small, _ = backend.debug_intersect(small)
    assert (want_big[:, 3] != 0xFFFFFFFF).sum() > 1000          # the rays do hit the box
    got_big = _query(backend, big)
    got_small = _query(backend, small)                          # straight after, smaller: shares the first call's scratch
    assert np.array_equal(got_big, want_big)
    assert np.array_equal(got_small, want_small)
    # interleaved with frames: the scratch is independent of the renderer's own buffers
    from pathtracer_amd import camera
    from pathtracer_amd.scene import Params
    backend.resize(96, 64)
    backend.set_camera(camera.cornell_camera(96, 64))
    backend.set_params(Params.reference(spp_per_frame=1, max_bounces=4))
    for f in (1, 2, 3):
        backend.trace(f)
    backend.sync()
    assert np.array_equal(_query(backend, small), want_small)


def test_no_alloc_ray_query_limits(backend, cornell):
    backend.scene_upload(cornell)
    backend.build_accel()
    buf = torch.zeros(64, dtype=torch.float32, device="cuda:0")
    with pytest.raises(BackendError) as e:
        backend.intersect_noalloc(buf.data_ptr(), (1 << 24) + 1, buf.data_ptr())
    assert e.value.code == 5 and "2^24" in str(e.value)
    with pytest.raises(BackendError):
        backend.intersect_noalloc(buf.data_ptr() + 4, 1, buf.data_ptr())      # misaligned rays
    backend.intersect_noalloc(0, 0, 0)                                        # nothing to do
    with pytest.raises(BackendError) as e:
        backend.intersect_noalloc(buf.data_ptr(), 2, buf.data_ptr() + 32)     # hit records inside the rays
    assert e.value.code == 1 and "overlaps" in str(e.value)


def test_no_alloc_ray_query_of_two_contexts_on_one_gpu(backend, cornell):
    """Every context on a GPU shares that GPU's scratch array: a query is refused while another context's query is still
    running, instead of returning hits computed with a fetch cursor the other query moves."""
    from pathtracer_amd.backend import Backend
    backend.scene_upload(cornell)
    backend.build_accel()
    lo, hi = cornell_bounds(cornell)
    rays = random_rays(np.random.default_rng(5), 1 << 22, lo, hi)
    want, _ = backend.debug_intersect(rays[:1000])
    d_rays = torch.from_numpy(rays).to("cuda:0")
    d_hits = torch.zeros((len(rays), 5), dtype=torch.int32, device="cuda:0")
    d_hits2 = torch.zeros((1000, 5), dtype=torch.int32, device="cuda:0")
    other = Backend(0)
    try:
        other.scene_upload(cornell)
        other.build_accel()
        torch.cuda.synchronize()
        backend.intersect_noalloc(d_rays.data_ptr(), len(rays), d_hits.data_ptr())       # milliseconds of work, asynchronous
        with pytest.raises(BackendError) as e:
            other.intersect_noalloc(d_rays.data_ptr(), 1000, d_hits2.data_ptr())
        assert e.value.code == 1 and "another context" in str(e.value)
        backend.sync()
        other.intersect_noalloc(d_rays.data_ptr(), 1000, d_hits2.data_ptr())             # now the array is free
        other.sync()
        assert np.array_equal(d_hits2.cpu().numpy().view(np.uint32), want)
        assert np.array_equal(d_hits[:1000].cpu().numpy().view(np.uint32), want)
    finally:
        other.close()


def cornell_bounds(sc):
    p = sc.vertices[:, :3]
    return p.min() * 0.1 - 0.2, p.max() * 0.1 + 0.2
