"""The multi-GPU merge on real RCCL (torch.distributed backend "nccl" is RCCL on ROCm, and the C ABI's pt_reduce_accum).

On a one-GPU box RCCL runs with a communicator of one rank: initialisation, the reduce call into the memory the path tracer
renders into (pt_bind_images) and the stream ordering around it (pt_wait_stream, pt_resolve) all execute; only the xGMI
transport is missing.  With two or more GPUs visible, two ranks on two devices render stripes of frame indices and merge
them with one reduce, exactly as `bench.py --gpus N` does, and the result is compared with one context's sequential
frames.  Frames add (RayGen.glsl:351-359) and a frame depends only on (pixel, FrameIndex) (RayGen.glsl:314-315), so the
merged image holds the single-GPU sample set; sample counts are equal and colours differ by fp32 summation order only."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from pathtracer_amd.scene import Params, SceneData
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu

W, H, K = 256, 192, 3        # image size, frames per rank


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sequential(b, scene, cam, frames):
    b.scene_upload(scene)
    b.build_accel()
    b.resize(W, H)
    b.set_camera(cam)
    b.set_params(Params.reference(spp_per_frame=1, max_bounces=4))
    b.trace(1)
    for f in frames:
        b.trace(f)
    return b.read_accum(), b.read_image()


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _torch_rccl():
    """The RCCL that torch brought into the process, with its symbols made global so that pt_reduce_accum finds the same one."""
    import torch
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(path):
        path = "librccl.so.1"
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    return lib


def test_one_rank_reduce_through_torch_and_through_the_c_abi(backend, cornell, cameras):
    import torch
    import torch.distributed as dist

    from pathtracer_amd import distributed as ptdist

    cam = cameras["512x512"]
    frames = ptdist.stripe(0, 1, K)
    want_acc, want_img = _sequential(backend, cornell, cam, frames)

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        r = ptdist.GpuRenderer(backend, W, H, dev)
        r.zero_accum()
        for f in frames:
            r.trace(f)
        r.sync()
        dist.reduce(r.accum, dst=0, op=dist.ReduceOp.SUM)       # RCCL, on torch's stream, into the bound image
        r.resolve()                                             # pt_wait_stream on that stream, then pt_resolve
        assert np.array_equal(r.accum.cpu().numpy().view(np.uint32), want_acc.view(np.uint32))
        assert np.array_equal(r.image.cpu().numpy().view(np.uint32), want_img.view(np.uint32))

        # the same merge without torch in the data path: pt_reduce_accum with an ncclComm_t made by RCCL's C API
        rccl = _torch_rccl()
        uid = _UniqueId()
        assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
        comm = C.c_void_p()
        assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
        try:
            r.zero_accum()
            for f in frames:
                r.trace(f)
            backend.reduce_accum(comm.value, 0)                 # on the context's stream, after the frames
            backend.resolve()
            backend.sync()
            assert np.array_equal(r.accum.cpu().numpy().view(np.uint32), want_acc.view(np.uint32))
            assert np.array_equal(r.image.cpu().numpy().view(np.uint32), want_img.view(np.uint32))
        finally:
            rccl.ncclCommDestroy(comm)
    finally:
        backend.resize(W, H)        # back to images the context owns: the tensors go away with this test
        dist.destroy_process_group()


def _rank_main(rank, world, port, out_path):
    """One process per GPU, as `bench.py --gpus N` runs: stripes, one RCCL reduce, resolve on rank 0; then the same through pt_reduce_accum."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    from pathtracer_amd import distributed as ptdist
    from pathtracer_amd.backend import Backend

    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    scene = SceneData.load(os.path.join(GOLDEN, "cornell_scene.npz"))
    cam = np.load(os.path.join(GOLDEN, "cornell_cameras.npz"))["512x512"]
    b = Backend(rank)
    b.scene_upload(scene)
    b.build_accel()
    r = ptdist.GpuRenderer(b, W, H, dev)
    b.set_camera(cam)
    b.set_params(Params.reference(spp_per_frame=1, max_bounces=4))
    acc = ptdist.render_striped(r, K, rank, world)
    merged_torch = acc.cpu().numpy().copy()
    image_torch = r.image.cpu().numpy().copy()

    rccl = _torch_rccl()
    uid = _UniqueId()
    if rank == 0:
        assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    box = [bytes(bytearray(uid))]
    dist.broadcast_object_list(box, src=0)
    C.memmove(C.byref(uid), box[0], 128)
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
    r.zero_accum()
    for f in ptdist.stripe(rank, world, K):
        r.trace(f)
    b.reduce_accum(comm.value, 0)
    if rank == 0:
        b.resolve()
    b.sync()
    merged_cabi = r.accum.cpu().numpy().copy()
    image_cabi = r.image.cpu().numpy().copy()
    dist.barrier()
    rccl.ncclCommDestroy(comm)
    if rank == 0:
        b.bind_images(None, None)
        want_acc, want_img = _sequential(b, scene, cam, list(range(2, 2 + world * K)))
        np.savez(out_path, merged_torch=merged_torch, image_torch=image_torch, merged_cabi=merged_cabi, image_cabi=image_cabi,
                 want_acc=want_acc, want_img=want_img)
    b.close()
    dist.destroy_process_group()


def test_two_ranks_on_two_devices(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL refuses two ranks on one device")
    import torch.multiprocessing as mp
    out = str(tmp_path / "merged.npz")
    mp.spawn(_rank_main, args=(2, _free_port(), out), nprocs=2, join=True)
    z = np.load(out)
    for acc, img in ((z["merged_torch"], z["image_torch"]), (z["merged_cabi"], z["image_cabi"])):
        assert np.array_equal(acc[..., 3], z["want_acc"][..., 3]) and (acc[..., 3] == 2 * K).all()
        assert np.allclose(acc[..., :3], z["want_acc"][..., :3], rtol=2e-6, atol=1e-6)
        assert np.allclose(img, z["want_img"], rtol=2e-6, atol=1e-6)
