"""The multi-GPU split (frame-index stripes + one sum-reduce of the accumulation images) on CPU:
two gloo ranks, the CPU oracle standing in for the GPU backend behind the same renderer interface."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pathtracer_amd import distributed as ptdist  # noqa: E402
from pathtracer_amd.scene import Params, SceneData  # noqa: E402

W, H, K = 40, 24, 3


class OracleRenderer:
    """Same interface as pathtracer_amd.distributed.GpuRenderer, backed by the CPU oracle (test stand-in)."""

    def __init__(self, scene, cam, params):
        from tests import oracle_lib
        self.o = oracle_lib.Oracle(scene, threads=2)
        self.o.resize(W, H)
        self.o.set_camera(cam)
        self.o.set_params(params)

    def zero_accum(self):
        self.o.write_accum(np.zeros((H, W, 4), np.float32))

    def trace(self, frame):
        self.o.trace(frame)

    def sync(self):
        pass

    def accum_tensor(self):
        return torch.from_numpy(self.o.read_accum().copy())

    def load_accum(self, t):
        self.o.write_accum(t.numpy())

    def resolve(self):
        pass


def _load():
    golden = os.path.join(ROOT, "tests", "golden")
    scene = SceneData.load(os.path.join(golden, "cornell_scene.npz"))
    cam = np.load(os.path.join(golden, "cornell_cameras.npz"))["24x16"]   # any fixed matrices do
    return scene, cam, Params.reference(spp_per_frame=2, max_bounces=4)


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, cam, params = _load()
    r = OracleRenderer(scene, cam, params)
    acc = ptdist.render_striped(r, K, rank, world)
    if rank == 0:
        np.save(out_path, acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_stripes_cover_every_frame_once():
    for world in (1, 2, 4, 8):
        frames = sorted(f for r in range(world) for f in ptdist.stripe(r, world, 5))
        assert frames == list(range(2, 2 + 5 * world))


def test_two_rank_merge_equals_sequential_render(tmp_path):
    out = str(tmp_path / "merged.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    merged = np.load(out)
    scene, cam, params = _load()
    r = OracleRenderer(scene, cam, params)
    r.trace(1)                                   # the reference's frame 1: clears, contributes nothing
    for f in range(2, 2 + 2 * K):
        r.trace(f)
    seq = r.o.read_accum()
    assert np.array_equal(merged[..., 3], seq[..., 3])           # sample counts add exactly: 2 ranks * K frames * 2 spp
    assert (merged[..., 3] == 2 * K * 2).all()
    assert np.allclose(merged[..., :3], seq[..., :3], rtol=2e-6, atol=1e-6)   # fp32 sums in a different order


def test_single_rank_path_needs_no_process_group():
    scene, cam, params = _load()
    r = OracleRenderer(scene, cam, params)
    acc = ptdist.render_striped(r, 2, 0, 1)
    assert (acc.numpy()[..., 3] == 4).all()
