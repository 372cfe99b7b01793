"""The multi-GPU split on the HIP backend itself: G contexts (one per rank of a G-GPU run) each behind
pathtracer_amd.distributed.GpuRenderer, each rendering its stripe of frame indices into a torch-owned
accumulation tensor (pt_bind_images); the tensors are summed the way the RCCL reduce sums them, loaded into
rank 0's context and resolved (pt_wait_stream + pt_resolve).  On a one-GPU box all contexts share GPU 0, which
exercises every call of the N > 1 path except the collective's transport.

Reference for the property: a frame depends only on (pixel, FrameIndex) (RayGen.glsl:314-315) and frames add
into (sum rgb, n) (RayGen.glsl:351-359), so the merged image holds exactly the samples of one context's
frames 2 .. 1 + G*K; only the order of the fp32 additions differs.  Tolerance: sample counts exact; rgb
within 4 ulp-scale relative error (rtol 1e-6 per addition, G*K <= 8 additions); bit-exact when every rank
renders one frame and G = 2 (a + b = b + a).
"""
import numpy as np
import pytest

from pathtracer_amd import distributed as ptdist
from pathtracer_amd import procedural
from pathtracer_amd.backend import Backend
from pathtracer_amd.scene import Params

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _contexts(scene, cam, params, W, H, G):
    import torch
    dev = torch.device("cuda", 0)
    out = []
    for _ in range(G):
        b = Backend(0)
        b.scene_upload(scene)
        b.build_accel()
        r = ptdist.GpuRenderer(b, W, H, dev)
        b.set_camera(cam)
        b.set_params(params)
        out.append(r)
    return out


def _striped(renderers, K):
    """What render_striped + the reduce do, for every rank of the group in one process."""
    G = len(renderers)
    for rank, r in enumerate(renderers):
        r.zero_accum()
        for f in ptdist.stripe(rank, G, K):
            r.trace(f)                  # all ranks' frames are in flight together, as on G GPUs
    for r in renderers:
        r.sync()
    merged = renderers[0].accum_tensor()
    for r in renderers[1:]:
        merged += r.accum_tensor()      # rank order, on torch's stream: the sum the reduce delivers on rank 0
    renderers[0].load_accum(merged)
    renderers[0].resolve()
    return renderers[0].accum.cpu().numpy(), renderers[0].image.cpu().numpy()


def _sequential(r, frames):
    r.zero_accum()
    for f in frames:
        r.trace(f)
    r.sync()
    return r.accum.cpu().numpy().copy()


@pytest.mark.parametrize("W,H,G,K", [(1920, 1080, 2, 1), (1920, 1080, 2, 2), (1920, 1080, 4, 2), (3840, 2160, 2, 2), (3840, 2160, 4, 1)])
def test_striped_contexts_merge_to_the_sequential_render(W, H, G, K):
    scene = procedural.sponza_class()
    cam = procedural.sponza_camera(W, H)
    params = Params.reference(spp_per_frame=1, max_bounces=8)
    rs = _contexts(scene, cam, params, W, H, G)
    try:
        seq = _sequential(rs[0], range(2, 2 + G * K))
        acc, img = _striped(rs, K)
        assert (acc[..., 3] == G * K).all() and np.array_equal(acc[..., 3], seq[..., 3])     # sample counts add exactly
        if G == 2 and K == 1:
            assert np.array_equal(bits(acc), bits(seq))
        else:
            assert np.allclose(acc[..., :3], seq[..., :3], rtol=1e-5, atol=1e-6, equal_nan=True)
        # pt_resolve after the merge: image = sum / n, NaN -> red (RayGen.glsl:367-372)
        want = (acc[..., :3] / acc[..., 3:4]).astype(np.float32)
        nan = np.isnan(want).any(-1)
        assert np.array_equal(bits(img[..., :3][~nan]), bits(want[~nan]))
        assert (img[nan] == np.array([1, 0, 0, 1], np.float32)).all()
        assert (img[..., 3] == 1).all()
    finally:
        for r in rs:
            r.b.close()


def test_resolve_waits_for_work_on_the_callers_stream():
    """pt_wait_stream orders pt_resolve behind a long chain of torch kernels that write the bound accumulation
    tensor, without a host synchronisation in between."""
    import torch
    W, H = 1920, 1080
    scene = procedural.sponza_class(detail=0.05)
    r = _contexts(scene, procedural.sponza_camera(W, H), Params.reference(spp_per_frame=1, max_bounces=2), W, H, 1)[0]
    try:
        r.zero_accum()
        side = torch.cuda.Stream(device=r.device)
        with torch.cuda.stream(side):
            r.accum.fill_(1.0)
            for _ in range(200):                   # 200 dependent full-image kernels on the side stream
                r.accum[..., :3].mul_(1.01)        # a resolve that ran early would see a smaller product
            r.b.wait_stream(side.cuda_stream)
            r.b.resolve()
        img = r.b.read_image()
        want = r.accum.cpu().numpy()
        assert want[0, 0, 0] > 7.0
        assert np.array_equal(bits(img[..., :3]), bits(want[..., :3] / want[..., 3:4]))
    finally:
        r.b.close()
