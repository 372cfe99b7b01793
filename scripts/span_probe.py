"""Unprofiled timeline of a short run: twenty-one frames (three batches of seven) after a synchronisation, every launch bracketed by
HIP events on its own stream (pt_set_profiling 1) and dumped through PT_SPAN_DUMP; prints per batch and chain when it started
and ended, and per 0.5 ms bin how many traversal / shading launches were running.
usage: PT_SPAN_DUMP=/tmp/spans.txt python scripts/span_probe.py [frames]"""
import os, sys, time
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 16:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"
path = os.environ.setdefault("PT_SPAN_DUMP", "/tmp/spans.txt")
if os.path.exists(path):
    os.remove(path)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd.backend import Backend
from pathtracer_amd.scene import Params
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 21
scene, cam, _ = bench.make_workload("sponza_like", 1920, 1080)
b = Backend(0)
b.scene_upload(scene); b.build_accel(); b.resize(1920, 1080); b.set_camera(cam)
b.set_params(Params.reference(spp_per_frame=1, max_bounces=8))
b.set_batch(7)
f = 1
for _ in range(8):
    b.trace(f); f += 1
b.sync()
b.set_profiling(1)
t0 = time.perf_counter()
for _ in range(frames):
    b.trace(f); f += 1
b.sync()
dt = time.perf_counter() - t0
t = b.last_trace_timing()
b.set_profiling(0)
print("%d frames in %.2f ms (window %.2f ms)" % (frames, dt * 1e3, t.total_ms))
batches, cur = [], []
for line in open(path):
    if line.startswith("#"):
        batches.append(cur); cur = []
    else:
        c, k, r, a, e = line.split()
        cur.append((int(c), k, int(r), float(a), float(e)))
for i, bt in enumerate(batches):
    st = {}; en = {}; busy = {}
    for c, k, r, a, e in bt:
        st[c] = min(st.get(c, 1e9), a); en[c] = max(en.get(c, 0), e); busy[c] = busy.get(c, 0) + e - a
    print("batch %d: chain starts %.2f..%.2f ms, ends %.2f..%.2f ms; per chain span %.2f ms" % (
        i, min(st.values()), max(st.values()), min(en.values()), max(en.values()), sum(en[c] - st[c] for c in st) / len(st)))
    print("   chain ends: " + " ".join("%d:%.1f" % (c, en[c]) for c in sorted(en)))
    for r in range(0, 10):
        ex = [(a, e) for c, k, rr, a, e in bt if k == 'e' and rr == r]
        sh = [(a, e) for c, k, rr, a, e in bt if k == 's' and rr == r]
        if ex:
            print("   round %d: traversal starts %.2f..%.2f, mean %.3f ms; shading mean %.3f ms" % (
                r, min(a for a, e in ex), max(a for a, e in ex), sum(e - a for a, e in ex) / len(ex), sum(e - a for a, e in sh) / len(sh)))
allsp = [x for bt in batches for x in bt]
end = max(e for *_, e in allsp)
nb = int(end / 0.5) + 1
ext = [0.0] * nb; shd = [0.0] * nb; oth = [0.0] * nb
for c, k, r, a, e in allsp:
    arr = ext if k == 'e' else shd if k == 's' else oth
    x = a
    while x < e:
        bi = int(x / 0.5)
        edge = (bi + 1) * 0.5
        seg = min(e, edge) - x
        arr[bi] += seg / 0.5
        x += seg
print("time ms: traversal launches running, shading, other")
for i in range(nb):
    print("%6.1f  %5.2f %5.2f %5.2f  %s" % (i * 0.5, ext[i], shd[i], oth[i], "#" * int(round(ext[i] + shd[i] + oth[i]))))
b.close()
