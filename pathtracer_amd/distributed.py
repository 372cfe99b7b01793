"""Multi-GPU progressive rendering: scene and hierarchy replicated per GPU, frame indices striped
over ranks, one sum-reduce of the accumulation images at the end.

A frame depends only on (pixel, FrameIndex) (RayGen.glsl:314-315) and frames combine by adding
(sum of radiance, sample count) per pixel (RayGen.glsl:351-359), so rank r of G renders
FrameIndex = 2 + r, 2 + r + G, ... into its own accumulation image; the G images are then summed
with one reduce (RCCL over xGMI on GPUs, gloo in the CPU tests).  The merged image holds exactly
the samples of a single-GPU run over frames 2 .. 1 + G*K; only the fp32 summation order differs.
Frame 1 of the reference clears the accumulation image and contributes nothing
(RayGen.glsl:361-365); here every rank starts from a zeroed image instead.
"""
import torch
import torch.distributed as dist


def stripe(rank, world_size, frames_per_rank, first_frame=2):
    """Frame indices rendered by `rank`."""
    return [first_frame + rank + i * world_size for i in range(frames_per_rank)]


def merge_accumulation(accum: torch.Tensor, dst=0, group=None):
    """Sums every rank's accumulation image (rgb sums and the sample count in .w) onto `dst`, in place.
    With RCCL the reduce is enqueued on torch's current stream order: device work that reads `accum`
    afterwards on another stream must wait for that stream (GpuRenderer.resolve does).  A gloo group
    (CPU tests, one-GPU rehearsals) reduces host memory, so a device tensor is staged through the host."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if accum.is_cuda and dist.get_backend(group) == "gloo":
            host = accum.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
            accum.copy_(host)
        else:
            dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum


def render_striped(renderer, frames_per_rank, rank, world_size, dst=0, group=None):
    """renderer: object with zero_accum(), trace(frame), sync() and accum_tensor() -> torch tensor that
    aliases (GPU) or copies (CPU stand-in) its accumulation image, and load_accum(tensor), resolve().
    Returns the merged accumulation tensor (meaningful on `dst`)."""
    renderer.zero_accum()
    for f in stripe(rank, world_size, frames_per_rank):
        renderer.trace(f)
    renderer.sync()
    acc = renderer.accum_tensor()
    merge_accumulation(acc, dst=dst, group=group)
    if rank == dst:
        renderer.load_accum(acc)
        renderer.resolve()
    return acc


class GpuRenderer:
    """Backend context whose accumulation / final images live in torch tensors, so that
    torch.distributed (backend "nccl" = RCCL) can reduce them in place."""

    def __init__(self, backend, width, height, device):
        self.b = backend
        self.device = torch.device(device)
        backend.resize(width, height)
        self.accum = torch.zeros((height, width, 4), dtype=torch.float32, device=self.device)
        self.image = torch.zeros((height, width, 4), dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        backend.bind_images(self.image.data_ptr(), self.accum.data_ptr())

    def zero_accum(self):
        self.b.sync()
        self.accum.zero_()
        torch.cuda.synchronize(self.device)

    def trace(self, frame):
        self.b.trace(frame)

    def sync(self):
        self.b.sync()

    def accum_tensor(self):
        return self.accum

    def load_accum(self, tensor):
        if tensor.data_ptr() != self.accum.data_ptr():
            self.accum.copy_(tensor)       # on torch's stream; resolve() waits for it

    def resolve(self):
        # the merge ran on torch's stream, the context resolves on its own: order the two on the device
        self.b.wait_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.b.resolve()
        self.b.sync()
