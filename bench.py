"""Throughput benchmark of the path-tracing hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pt_trace frame (1 sample per pixel, up to `bounces` ray segments per path) of the
workload named in config.workload.  Every rank holds the whole scene and hierarchy and renders its
own stripe of frame indices (weak scaling: K frames per rank); at N > 1 the accumulation images are
summed with one RCCL reduce inside the timed region.  value = ray segments traced by all ranks /
max-over-ranks wall time, inputs resident in HBM.

The headline workload is the closest stand-in this repository has for the reference's textured Crytek Sponza
(whose geometry file is absent from the reference checkout): `sponza_like`, an atrium with large wall / floor quads, thin
diagonal parts, 16 textures and split triangles, lit through the open roof by the reference's Preetham sky
(RayTracingLayer.h:88: turbidity 3.14, azimuth 0, inclination 0) as a 2048-texel-per-face cube (texels on and below the
horizon, where that shader has no values, mirrored from above), so that a miss pays a real environment lookup (RayGen.glsl:132).  At N = 1 the other workloads are timed briefly after the headline and
printed on the same JSON line as `other_workloads`: the uniformly tessellated Sponza-class atrium, the 3 M-triangle
Bistro-class street (hierarchy build time + rays per second), the headline scene at 3840x2160 and the Cornell box.

Also reported on the same JSON line:
  roofline     what binds the path is vector-instruction issue, not HBM: `bound` "valu", achieved = wave64 vector
               instructions issued per second by k_extend + k_shade (instructions per ray segment from the committed
               rocprofv3 --pmc passes over this same workload, profiles/counters_per_ray_<workload>.json, times this run's rays
               per second) against 1024 SIMDs x 2.4 GHz / 4 cycles.  The counters file names the hash of the kernel sources
               it was taken at; when the sources have changed since, `counters_stale` is true and the line falls back to
               the HBM figure.  The HBM side stays in roofline.hbm: algorithmic bytes of k_extend
               (ray 32 B read + hit 16 B write per ray, 128 B per hierarchy node visited, 48 B per triangle tested;
               counted by the kernel's own counting variant) over the time at least one launch was running, and the
               bytes that crossed the fabric by the counters, both against the 8 TB/s peak.  `traffic` = fabric bytes
               per k_extend launch.  A frame runs as `chains` pipelines side by side, so `concurrency` launches share
               the chip at a time: avg_launch_ms is the average duration of one launch (HIP events on the stream it runs
               on; what rocprofv3 reports).
  cpu_baseline the CPU oracle (a port of the reference shaders, not the reference itself) on the
               host cores, rank 0, bounded sample of the same workload; `lavapipe` says why the reference's own shaders
               could not be run on a CPU Vulkan driver here
  phases_ms    (N > 1) max over ranks of the timed region's parts: render, reduce, resolve
"""
import argparse
import ctypes.util
import glob
import hashlib
import json
import os
import shutil
import sys
import time

# one hardware queue per chain (16 chains); HIP reads this when its runtime starts (its default is 4).  More queues
# than chains cost throughput (32 queues: -30 %), so a larger setting from outside is left alone but not made here
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 16:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak
SIMDS, CLOCK_HZ = 1024, 2.4e9           # 256 CUs x 4 SIMDs, peak engine clock
VALU_PEAK_GINSTR = SIMDS * CLOCK_HZ / 4.0 / 1e9   # one wave64 vector instruction per SIMD every 4 cycles
NODE_BYTES, TRI_BYTES, RAY_BYTES, HIT_BYTES = 128, 48, 32, 16
SKY = (3.14, 0.0, 0.0, 2048)            # RayTracingLayer.h:88 (turbidity, azimuth, inclination), 2048 texels per face (RayTracingLayer.cpp:58-63)
METRIC = "Mrays/s (paths×bounces) per GPU + node, Sponza 1080p 8 bounces; % HBM roofline"


def host_threads():
    """(threads to use, cores the machine reports, cores this process may run on).  A GPU box hands a job a share of a
    larger machine: os.cpu_count() and the affinity mask show the whole machine, the job's share is what the box exports as
    OMP_NUM_THREADS; without that variable every core of the affinity mask is used."""
    nproc = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = nproc
    share = 0
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        share = 0
    return max(1, min(aff, share) if share > 0 else aff), nproc, aff


def lavapipe_probe():
    """The preferred CPU baseline is the reference's own GLSL on Mesa lavapipe.  That needs a Vulkan loader, a lavapipe ICD with
    the ray-tracing extensions, a GLSL compiler and a headless harness around the shaders; says which of them are missing."""
    missing = []
    if not ctypes.util.find_library("vulkan"):
        missing.append("no libvulkan loader")
    icds = []
    for d in ("/usr/share/vulkan/icd.d", "/etc/vulkan/icd.d", "/usr/local/share/vulkan/icd.d"):
        icds += glob.glob(os.path.join(d, "*lvp*.json"))
    if not icds:
        missing.append("no lavapipe ICD (lvp_icd*.json)")
    if not (shutil.which("glslangValidator") or shutil.which("glslc")):
        missing.append("no GLSL compiler (glslangValidator / glslc)")
    missing.append("no headless Vulkan harness for the reference shaders in this repository (it cannot be built or tested without the above)")
    return "unavailable: " + "; ".join(missing)


KERNEL_SOURCES = ["trace.hip", "pt_traverse.h", "pt_shade.inc", "pt_bsdf.inc", "pt_fast.h", "pt_math.h", "pt_vec.inc", "pt_types.h", "bvh_build.hip"]


def csrc_sha16():
    """Hash of the sources that decide the instruction streams of the two hot kernels and the shape of the hierarchy (not the
    host-side API and the image passes): the per-ray counters under profiles/ are properties of this code."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pathtracer_amd", "csrc")
    for name in KERNEL_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def gltf_workload(path, width, height, scale, cam_text):
    """A glTF file of the user's own (the reference selects its model by path, RayTracingLayer.cpp:27-32): imported by the
    C++ host layer with its PNG / JPEG textures; layer transform = uniform scale (the reference uses 0.1 for its Cornell
    box, RayTracingLayer.cpp:34).  Camera: "ex,ey,ez,tx,ty,tz[,fov_y]" (eye, target), default a view of the scene's box."""
    import numpy as np
    from pathtracer_amd import camera, host
    src = host.MeshSource(path)
    ntex = src.load_textures()
    sc = src.scene_data(np.diag([scale, scale, scale, 1.0]), with_textures=True)
    if cam_text:
        v = [float(x) for x in cam_text.split(",")]
        eye, target, fov = v[0:3], v[3:6], (v[6] if len(v) > 6 else 45.0)
    else:
        pts = []
        for s_, xf in zip(sc.submeshes, sc.xforms):
            n = int(s_[1])
            m = xf.reshape(3, 4)
            p = sc.vertices[n:n + 8192:64, 0:3] if len(sc.vertices) > n else sc.vertices[:1, 0:3]
            pts.append(p @ m[:, :3].T + m[:, 3])
        pts = np.concatenate(pts)
        lo, hi = pts.min(0), pts.max(0)
        c, ext = (lo + hi) / 2, float((hi - lo).max())
        eye, target, fov = (c + np.array([0.35, 0.25, 1.0]) * ext * 0.9).tolist(), c.tolist(), 45.0
    near = max(1e-4, 1e-4 * float(np.linalg.norm(np.subtract(eye, target))))
    cam = camera.camera_buffer(camera.look_at(eye, target), camera.perspective(fov, width / height, near, near * 1e5))
    return sc, cam, "%s (%d tris, %d submeshes, %d materials, %d textures), scale %g" % (
        os.path.basename(path), sc.num_triangles, len(sc.submeshes), len(sc.materials), ntex, scale)


def make_workload(name, width, height, detail=1.0):
    from pathtracer_amd import camera, procedural
    from pathtracer_amd.scene import SceneData
    if name == "sponza_class":
        sc = procedural.sponza_class(detail=detail)
        return sc, procedural.sponza_camera(width, height), \
            "procedural Sponza-class atrium (%d tris, %d submeshes, 25 full-Disney materials; Sponza.bin is absent from the reference)" % (
                sc.num_triangles, len(sc.submeshes))
    if name == "sponza_like":
        sc = procedural.sponza_class(detail=detail, layout="sponza", textures=True)
        return sc, procedural.sponza_camera(width, height), \
            "procedural Sponza-like atrium: large wall / floor / slab quads and long thin trims beside finely tessellated detail, %d tris, %d submeshes, " \
            "25 full-Disney materials, %d textures of 512x512 (Sponza.bin is absent from the reference)" % (sc.num_triangles, len(sc.submeshes), len(sc.textures))
    if name == "bistro_class":
        sc = procedural.bistro_class()
        return sc, procedural.bistro_camera(width, height), "procedural Bistro-class street (%d tris; the Bistro scene is not in the reference)" % sc.num_triangles
    if name == "cornell":
        sc = SceneData.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
        return sc, camera.cornell_camera(width, height), "CornellBox.gltf from the reference assets (63978 tris)"
    raise SystemExit("unknown workload " + name)


class Case:
    """One workload resident on one GPU: scene, hierarchy, environment, images, camera and parameters set."""

    def __init__(self, args, name, width, height, local_rank, device, gltf=None, precision=None):
        from pathtracer_amd import distributed as ptdist
        from pathtracer_amd.backend import Backend
        from pathtracer_amd.scene import Params
        self.name, self.width, self.height = name, width, height
        if gltf:
            self.scene, self.cam, self.text = gltf_workload(gltf, width, height, args.scale, args.camera)
        else:
            self.scene, self.cam, self.text = make_workload(name, width, height, args.detail)
        self.params = Params.reference(spp_per_frame=1, max_bounces=args.bounces)
        self.b = b = Backend(local_rank)
        b.scene_upload(self.scene)
        self.first_build_ms = b.build_accel(args.build_flags).build_ms     # includes the allocator's first touches
        self.accel = b.build_accel(args.build_flags)                       # the same build again: its steady-state time
        self.sky = args.sky
        if self.sky == "preetham":
            # the reference's sky pass on the GPU; its shader has no values on and below the horizon (NaN texels, which would turn
            # every pixel a path leaves that way NaN for good), so those texels take the sky's value mirrored at the horizon
            from pathtracer_amd import procedural
            b.env_preetham(*SKY)
            b.env_upload(procedural.sky_mirrored_below_the_horizon(b.read_env()))
            self.sky_text = "Preetham sky (turbidity %g, azimuth %g, inclination %g) as a %d-texel cube, mirrored below the horizon, x10" % SKY
        else:
            self.sky_text = "constant white sky x10"
        self.precision = args.precision if precision is None else precision
        if self.precision:
            b.set_precision(self.precision)
        self.r = ptdist.GpuRenderer(b, width, height, device)
        b.set_camera(self.cam)
        b.set_params(self.params)

    def set_batch(self, steps, batch=0):
        # frames that share their launches (pt_set_batch, at most 8): the timed frames in equal batches, so that the run does
        # not end on a short one (20 frames: 7 + 7 + 6 instead of 8 + 8 + 4)
        nbatches = (steps + 7) // 8
        self.b.set_batch(batch if batch > 0 else max(1, min(8, (steps + nbatches - 1) // nbatches)))

    def describe(self, bounces):
        return "%s, %dx%d, 1 spp/frame, %d bounces, no Russian roulette, %s%s; %s" % (
            self.name, self.width, self.height, bounces, self.sky_text,
            ", shading arithmetic within a stated tolerance (pt_set_precision 1)" if self.precision else "", self.text)

    def hierarchy(self):
        a = self.accel
        return {"nodes": a.num_nodes, "levels": a.max_depth, "build_ms": round(a.build_ms, 3), "first_build_ms": round(self.first_build_ms, 3),
                "sah_cost": round(a.sah_cost, 2), "references": a.num_references}

    def nan_pixels(self):
        """Pixels the last frame painted red because their colour was NaN (RayGen.glsl:369-370)."""
        im = self.r.image
        return int(((im[..., 0] == 1.0) & (im[..., 1] == 0.0) & (im[..., 2] == 0.0) & (im[..., 3] == 1.0)).sum().item())

    def close(self):
        self.b.close()


def load_counters(workload, args, width, height, precision):
    """Per-ray counters of the two hot kernels from the committed rocprofv3 --pmc passes over this workload, or (None, why)."""
    cpath = os.path.join(ROOT, "profiles", "counters_per_ray_%s.json" % workload)
    # per-ray figures carry over to other image sizes of the same scene and camera (the rays are distributed alike)
    if (args.bounces, args.detail, args.build_flags, args.sky, precision) != (8, 1.0, 0, "preetham", 0) or abs(width * 9 - height * 16) > 16:
        return None, None, "non-default parameters"
    try:
        ctr = json.load(open(cpath))
    except (OSError, ValueError):
        return None, None, "no committed counters for this workload"
    return ctr, cpath, None


def profile_case(case, frames, args, timed_rays_per_s):
    """Roofline of the dominant kernel: a timed pass (HIP events per launch), then a counting pass."""
    b = case.b
    nprof = min(16, len(frames))
    b.set_profiling(1)
    for f in frames[:nprof]:
        b.trace(f)
    b.sync()
    t = b.last_trace_timing()
    ext_ms, ext_busy_ms, ext_launches, chains = t.extend_ms, t.extend_busy_ms, t.extend_launches, t.chains
    shade_ms, raygen_ms, total_ms = t.shade_ms, t.raygen_ms, t.total_ms
    b.set_profiling(2)
    for f in frames[:nprof]:
        b.trace(f)
    b.sync()
    t = b.last_trace_timing()
    nodes, tris, count_rays = t.nodes_visited, t.tris_tested, t.rays
    b.set_profiling(0)
    alg_bytes = count_rays * (RAY_BYTES + HIT_BYTES) + nodes * NODE_BYTES + tris * TRI_BYTES
    alg_gbs = alg_bytes / (ext_busy_ms * 1e-3) / 1e9 if ext_busy_ms > 0 else 0.0
    rays_per_launch = count_rays / max(1, ext_launches)
    ctr, cpath, why = load_counters(case.name, args, case.width, case.height, case.precision)
    stale = None
    if ctr is not None:
        stale = ctr.get("csrc_sha16") != csrc_sha16()
    use = ctr if (ctr is not None and not stale) else {}
    ce, cs = use.get("k_extend", {}), use.get("k_shade", {})
    insts_per_ray = (ce.get("valu_insts_per_ray") or 0.0) + (cs.get("valu_insts_per_ray") or 0.0)
    busy_per_ray = (ce.get("valu_busy_cycles_per_ray") or 0.0) + (cs.get("valu_busy_cycles_per_ray") or 0.0)
    if insts_per_ray > 0:
        # what binds the path: vector-instruction issue.  One wave64 instruction holds a SIMD's issue port for 4 cycles
        # (2 for v_add / v_mul / v_fma / v_mov); peak = 1024 SIMDs x 2.4 GHz / 4
        achieved = insts_per_ray * timed_rays_per_s / 1e9
        roofline = {"kernel": "k_extend + k_shade", "bound": "valu", "achieved": round(achieved, 1), "peak": VALU_PEAK_GINSTR, "unit": "Ginstr/s",
                    "frac": round(achieved / VALU_PEAK_GINSTR, 4)}
    else:
        roofline = {"kernel": "k_extend", "bound": "hbm", "achieved": round(alg_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(alg_gbs / HBM_PEAK_GBS, 4),
                    "note": "%s: algorithmic bytes only (every node visit counted as memory traffic), which is not what binds the path; "
                            "the profiled workloads are bound by vector-instruction issue" % (why or "the committed counters were taken at other kernel sources")}
    fabric_per_ray = ce.get("fabric_bytes_per_ray")
    roofline.update({
        "counters_stale": stale,
        "traffic": int(fabric_per_ray * rays_per_launch) if fabric_per_ray else None,
        "avg_launch_ms": round(ext_ms / max(1, ext_launches), 4), "launches": ext_launches,
        "chains": chains, "concurrency": round(ext_ms / ext_busy_ms, 2) if ext_busy_ms > 0 else None,
        "alg_bytes_per_launch": int(alg_bytes / max(1, ext_launches)),
        "bytes_per_ray": round(alg_bytes / max(1, count_rays), 1),
        "nodes_per_ray": round(nodes / max(1, count_rays), 2), "tris_per_ray": round(tris / max(1, count_rays), 2),
        "profiled_frames": nprof, "frame_ms": {"wall": round(total_ms / nprof, 3), "raygen_sum": round(raygen_ms / nprof, 3),
                                               "extend_busy": round(ext_busy_ms / nprof, 3), "extend_sum": round(ext_ms / nprof, 3),
                                               "shade_sum": round(shade_ms / nprof, 3)},
        # the HBM side of k_extend: algorithmic bytes (every node visit and triangle test counted as memory traffic) over the
        # time at least one launch was running, and the bytes that crossed the fabric (counters, doubled FETCH_SIZE: an estimate
        # for per-lane gathers) at this run's ray rate
        "hbm": {"alg_bandwidth_gbs": round(alg_gbs, 1), "alg_frac": round(alg_gbs / HBM_PEAK_GBS, 4),
                "fabric_gbs": round(fabric_per_ray * timed_rays_per_s / 1e9, 1) if fabric_per_ray else None,
                "fabric_frac": round(fabric_per_ray * timed_rays_per_s / 1e9 / HBM_PEAK_GBS, 4) if fabric_per_ray else None,
                "l2_hit_rate": round(ce["l2_hit_rate"], 3) if "l2_hit_rate" in ce else None, "peak_gbs": HBM_PEAK_GBS},
        "valu": None if insts_per_ray <= 0 else {
            "extend_insts_per_ray": round(ce.get("valu_insts_per_ray", 0.0), 1), "shade_insts_per_ray": round(cs.get("valu_insts_per_ray", 0.0), 1),
            "insts_per_launch": int(ce.get("valu_insts_per_ray", 0.0) * rays_per_launch),
            "issue_slot_frac": round(insts_per_ray * timed_rays_per_s * 4.0 / (SIMDS * CLOCK_HZ), 4),
            "pipe_busy_frac": round(busy_per_ray * timed_rays_per_s / (SIMDS * CLOCK_HZ), 4) if busy_per_ray else None,
            "note": "wave64 vector instructions per ray segment x this run's rays per second x 4 cycles / (1024 SIMDs x 2.4 GHz); "
                    "pipe_busy_frac uses SQ_ACTIVE_INST_VALU instead of 4 cycles per instruction; the clock under this load is below 2.4 GHz; "
                    "v_add / v_mul / v_fma issue in 2 cycles, so the fraction is an upper estimate of how full the issue ports are",
            "profile_commit": ctr.get("commit"), "profile_csrc_sha16": ctr.get("csrc_sha16"), "profile": "profiles/" + os.path.basename(cpath)},
        # second kernel: sky or surface fetch + Disney sample / eval + next ray; algorithmic bytes per ray segment: queue and path
        # state 168 B, surface <= 308 B on a hit
        "k_shade": {"avg_launch_ms": round(shade_ms / max(1, ext_launches), 4), "launches": ext_launches,
                    "alg_bytes_per_ray": {"queue_and_state": 168, "surface_on_hit_max": 308},
                    "alg_bandwidth_gbs": round((168 + 308) * timed_rays_per_s / 1e9, 1),
                    "fabric_bytes_per_ray": round(cs["fabric_bytes_per_ray"], 1) if "fabric_bytes_per_ray" in cs else None,
                    "l2_hit_rate": round(cs["l2_hit_rate"], 3) if "l2_hit_rate" in cs else None,
                    "parked_share_of_wave_cycles": round(cs["parked_share_of_wave_cycles"], 3) if "parked_share_of_wave_cycles" in cs else None,
                    "valu_insts_per_ray": round(cs["valu_insts_per_ray"], 1) if "valu_insts_per_ray" in cs else None},
    })
    return roofline


def cpu_baseline(case, args, steps):
    """The CPU oracle on the host cores over a bounded sample of the same workload (rank 0, N = 1)."""
    from tests import oracle_lib
    threads, nproc, aff = host_threads()
    scene = case.scene
    if case.sky == "preetham":
        from pathtracer_amd import procedural
        scene.env = procedural.sky_mirrored_below_the_horizon(oracle_lib.preetham_sky(*SKY))      # the oracle's own sky pass (bit-equal to the GPU's, tests/test_gpu_parity.py)
    o = oracle_lib.Oracle(scene, threads=threads)
    o.resize(case.width, case.height)
    o.set_camera(case.cam)
    o.set_params(case.params)
    c0 = o.ray_count()
    tc = time.perf_counter()
    nframes = 0
    while True:
        o.trace(2 + nframes)
        nframes += 1
        if time.perf_counter() - tc >= args.cpu_seconds or nframes >= steps:
            break
    tc = time.perf_counter() - tc
    cpu = {"value": round((o.ray_count() - c0) / tc / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
           "nproc": nproc, "affinity": aff,
           "threads_note": "threads = the job's CPU share (OMP_NUM_THREADS as exported by the box, else the affinity mask); nproc = cores the machine reports",
           "sample": "%d full %dx%d frames of the same scene, sky, camera and parameters (%.1f s)" % (nframes, case.width, case.height, tc),
           "lavapipe": lavapipe_probe()}
    o.close()
    scene.env = None
    return cpu


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="sponza_like")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--sky", default="preetham", choices=["preetham", "white"], help="environment: the reference's Preetham sky as a 2048-texel cube, or a 1-texel white cube")
    ap.add_argument("--precision", type=int, default=0, help="0 = bit-exact shading arithmetic (default), 1 = stated-tolerance arithmetic (pt_set_precision)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--others", type=int, default=-1, help="frames per extra workload timed after the headline at N = 1 (0 = none; default 16)")
    ap.add_argument("--build-flags", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0, help="frames that share their launches (1..8; 0 = the timed frames in equal batches)")
    ap.add_argument("--detail", type=float, default=1.0, help="tessellation scale of the procedural scenes (1 = the benchmark scene)")
    ap.add_argument("--gltf", default=None, help="benchmark this glTF file (with its textures) instead of a built-in workload")
    ap.add_argument("--scale", type=float, default=1.0, help="layer transform of --gltf: uniform scale")
    ap.add_argument("--camera", default=None, help='camera of --gltf: "ex,ey,ez,tx,ty,tz[,fov_y]"')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from pathtracer_amd import distributed as ptdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) does not match WORLD_SIZE (%d)" % (args.gpus, world))
    # rehearsal on a one-GPU box: PT_BENCH_REHEARSE=1 puts every rank on GPU 0 and reduces through gloo (host-staged:
    # RCCL refuses two ranks on one device)
    rehearse = os.environ.get("PT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    reduce_backend = (dist.get_backend() if world > 1 else None)

    def barrier(b):
        b.sync()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def timed(case, steps, warmup):
        """W untimed frames, then exactly `steps` frames between barriers; returns (seconds, ray segments, phase seconds, frames)."""
        b, r = case.b, case.r
        case.set_batch(steps, args.batch)
        # warm-up: frame 1 clears the accumulation image, the rest touch every kernel and the reduce path
        b.trace(1)
        frames = ptdist.stripe(rank, world, warmup + steps)
        for f in frames[:warmup]:
            b.trace(f)
        b.sync()
        if world > 1:
            ptdist.merge_accumulation(r.accum.clone())
        r.zero_accum()
        rays0 = b.ray_count()
        barrier(b)
        t0 = time.perf_counter()
        for f in frames[warmup:]:
            b.trace(f)
        b.sync()
        t1 = time.perf_counter()
        t2 = t3 = t1
        if world > 1:
            ptdist.merge_accumulation(r.accum)
            torch.cuda.synchronize(device)
            t2 = time.perf_counter()
            if rank == 0:
                r.resolve()         # waits for the reduce on the device (pt_wait_stream), then pt_resolve
            t3 = time.perf_counter()
        barrier(b)
        dt = time.perf_counter() - t0
        return dt, b.ray_count() - rays0, (t1 - t0, t2 - t1, t3 - t2), frames[warmup:]

    headline = Case(args, "gltf" if args.gltf else args.workload, args.width, args.height, local_rank, device, gltf=args.gltf)
    dt, rays, phases, frames = timed(headline, args.steps, args.warmup)
    nan_pixels = headline.nan_pixels() if world == 1 else None
    stat = torch.tensor([dt, phases[0], phases[1], phases[2], float(rays)], dtype=torch.float64, device="cpu" if rehearse else device)
    if world > 1:
        tmax = stat[0:4].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rsum = stat[4:5].clone()
        dist.all_reduce(rsum, op=dist.ReduceOp.SUM)
        dt_max, rays_all = float(tmax[0].item()), float(rsum.item())
        phases_max = [float(x) for x in tmax[1:4].tolist()]
    else:
        dt_max, rays_all, phases_max = dt, float(rays), list(phases)

    out = None
    if rank == 0:
        timed_rays_per_s = rays_all / dt_max / max(1, world)          # one GPU's share of the timed region
        roofline = profile_case(headline, frames, args, timed_rays_per_s)
        cpu = cpu_baseline(headline, args, args.steps) if (args.cpu_seconds > 0 and world == 1) else None
        out = {
            "metric": METRIC,
            "value": round(rays_all / dt_max / 1e6, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": headline.describe(args.bounces),
                       "rays_per_step_per_gpu": int(rays / args.steps), "triangles": headline.accel.num_triangles, "nan_pixels": nan_pixels,
                       "hierarchy": headline.hierarchy(),
                       "shading_arithmetic": "stated tolerance (pt_set_precision 1)" if args.precision else "bit-exact against the CPU oracle (0 ulp)",
                       "parallelism": "frame-index stripes per GPU, scene replicated" + (", one reduce of the accumulation image (%s)" % reduce_backend if world > 1 else "")},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if world > 1:
            out["phases_ms"] = {"render": round(phases_max[0] * 1e3, 3), "reduce": round(phases_max[1] * 1e3, 3), "resolve": round(phases_max[2] * 1e3, 3),
                                "note": "max over ranks; render = the rank's K frames until its streams are idle, reduce = one sum-reduce of the accumulation "
                                        "image to rank 0 until the device is idle, resolve = pt_resolve on rank 0",
                                "ranks": world, "reduce_backend": reduce_backend,
                                "accum_bytes": headline.width * headline.height * 16}
    barrier(headline.b)
    headline.close()
    del headline

    # ---- the other workloads, timed briefly (N = 1): every BASELINE configuration gets a driver-timed figure ----
    nother = 16 if args.others < 0 else args.others
    if rank == 0 and world == 1 and nother > 0 and not args.gltf:
        cases = [("sponza_class", 1920, 1080, None), ("bistro_class", 1920, 1080, None), (args.workload, 3840, 2160, None), ("cornell", 1920, 1080, None),
                 (args.workload, args.width, args.height, 1 - args.precision)]      # the headline again with the other shading arithmetic
        others = []
        for name, w, h, prec in cases:
            if (name, w, h, prec) == (args.workload, args.width, args.height, None):
                continue
            c = Case(args, name, w, h, local_rank, device, precision=prec)
            d, n, _, fr = timed(c, nother, 4)
            nanpx = c.nan_pixels()
            rl = profile_case(c, fr[:8], args, n / d)
            others.append({"workload": c.describe(args.bounces), "value": round(n / d / 1e6, 2), "unit": "Mrays/s", "steps": nother, "warmup": 4,
                           "ms_per_step": round(d / nother * 1e3, 4), "rays_per_step": int(n / nother), "triangles": c.accel.num_triangles, "nan_pixels": nanpx,
                           "hierarchy": c.hierarchy(),
                           "roofline": {k: rl[k] for k in ("bound", "achieved", "peak", "unit", "frac", "counters_stale", "avg_launch_ms", "nodes_per_ray", "tris_per_ray", "bytes_per_ray", "hbm", "valu")}})
            c.close()
            del c
        out["other_workloads"] = others

    if world > 1:
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out, ensure_ascii=False), flush=True)


if __name__ == "__main__":
    main()
