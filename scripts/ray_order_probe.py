"""What ordering a round's rays by origin could buy the traversal of a large scene: bounce rays of the Bistro-class street
(hit points of camera rays, directions drawn on the hemisphere facing back along the incoming ray; then one and two bounces
further) are traced through pt_debug_intersect in four orders: as the wavefront queue holds them (image order: 8x8 pixel
tiles), sorted by the 8x8x8 cell of the scene box their origin lies in, sorted by a 30-bit Morton code of the origin plus
the direction octant, and shuffled.  Prints nodes and triangles per ray; the kernel times come from running this script under
`rocprofv3 --kernel-trace` (k_debug_intersect dispatches in the order printed), hit rates from a `--pmc TCC_HIT_sum TCC_MISS_sum` run.

usage: python scripts/ray_order_probe.py [width height]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 16:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"
from pathtracer_amd import procedural  # noqa: E402
from pathtracer_amd.backend import Backend  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
workload = sys.argv[3] if len(sys.argv) > 3 else "bistro_class"
rng = np.random.default_rng(7)
if workload == "bistro_class":
    sc, cam = procedural.bistro_class(), procedural.bistro_camera(W, H)
else:
    sc, cam = procedural.sponza_class(layout="sponza", textures=True), procedural.sponza_camera(W, H)
b = Backend(0)
b.scene_upload(sc)
st = b.build_accel()
print("%s: %d triangles, %d nodes" % (workload, st.num_triangles, st.num_nodes))

# camera rays in the order of the wavefront queue: 8x8 tiles, row-major inside a tile
inv_view = cam[48:64].reshape(4, 4).T.astype(np.float64)
inv_proj = cam[64:80].reshape(4, 4).T.astype(np.float64)
ty, tx = np.meshgrid(np.arange(0, H, 8), np.arange(0, W, 8), indexing="ij")
ly, lx = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
px = (tx.reshape(-1, 1) + lx.reshape(1, -1)).reshape(-1)
py = (ty.reshape(-1, 1) + ly.reshape(1, -1)).reshape(-1)
keep = (px < W) & (py < H)
px, py = px[keep], py[keep]
ndc = np.stack([(px + 0.5) / W * 2 - 1, (py + 0.5) / H * 2 - 1, np.ones_like(px, np.float64), np.ones_like(px, np.float64)], 1)
tgt = ndc @ inv_proj.T
tgt = tgt[:, :3] / tgt[:, 3:4]
tgt /= np.linalg.norm(tgt, axis=1, keepdims=True)
d = tgt @ inv_view[:3, :3].T
d /= np.linalg.norm(d, axis=1, keepdims=True)
o = np.broadcast_to(inv_view[:3, 3], d.shape).copy()


def pack(o, d):
    r = np.empty((len(o), 8), np.float32)
    r[:, 0:3] = o
    r[:, 3] = 1e-5
    r[:, 4:7] = d
    r[:, 7] = 1e27
    return r


def trace(o, d, label):
    hits, counters = b.debug_intersect(pack(o, d))
    n = len(o)
    print("  %-34s %8d rays  nodes/ray %6.2f  tris/ray %6.2f" % (label, n, counters[0] / n, counters[1] / n), flush=True)
    return hits


def bounce(o, d, hits):
    t = hits[:, 0].view(np.float32).astype(np.float64)
    hit = hits[:, 3] != 0xFFFFFFFF
    o2 = (o + d * t[:, None])[hit]
    din = d[hit]
    v = rng.normal(size=o2.shape)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    flip = (v * din).sum(1) > 0
    v[flip] = -v[flip]
    return o2 + v * 3e-4, v


lo = np.minimum.reduce([sc.vertices[:, 0:3].min(0)]).astype(np.float64)
hi = sc.vertices[:, 0:3].max(0).astype(np.float64)


def cell_key(o, bits):
    q = np.clip(((o - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    key = np.zeros(len(o), np.int64)
    for k in range(bits):          # Morton interleave
        for a in range(3):
            key |= ((q[:, a] >> k) & 1) << (3 * k + a)
    return key


hits = trace(o, d, "camera rays, image order")
for depth in (1, 2, 3):
    o, d = bounce(o, d, hits)
    print("bounce %d" % depth)
    hits = trace(o, d, "queue (image) order")
    order = np.argsort(cell_key(o, 3), kind="stable")
    trace(o[order], d[order], "sorted by 8x8x8 origin cell")
    octant = ((d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2))
    order = np.argsort(cell_key(o, 10) * 8 + octant, kind="stable")
    trace(o[order], d[order], "sorted by Morton(origin) + octant")
    order = rng.permutation(len(o))
    trace(o[order], d[order], "shuffled")
b.close()
