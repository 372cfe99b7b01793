"""Turns the rocprofv3 databases written by scripts/profile.sh into the summaries kept under profiles/.

usage: python scripts/summarize_profile.py gpurun_out/prof_r01 r01 [commit]
writes profiles/<tag>_kernel_stats.md, profiles/<tag>_pmc.json, profiles/extend_traffic.json
(the HBM bytes per k_extend launch that bench.py reports as roofline.traffic) and
profiles/counters_per_ray_<workload>.json (BENCH_ARGS="--workload ..." to profile.sh selects the workload): vector instructions, vector-pipe busy cycles, fabric bytes, L2 hit rate and
parked share of k_extend and k_shade per ray segment, which bench.py scales by the rays of its own run.
"""
import glob
import json
import os
import re
import sqlite3
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import csrc_sha16  # noqa: E402  (hash of the kernel sources the counters belong to)
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)


def short(name):
    m = re.search(r"(ptd::k_\w+(<[\w, ]+>)?|__amd_rocclr_\w+|rocprim\S*?::(\w+_kernel|trampoline_kernel))", name)
    return m.group(1) if m else name[:60]


db = sqlite3.connect(os.path.join(src, "trace", "trace_results.db"))
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
lines = ["# rocprofv3 --kernel-trace --stats, `python bench.py --steps 16 --warmup 2 --cpu-seconds 0 --others 0` (%s)" % tag, "",
         "| kernel | calls | total us | average us | % |", "|---|---|---|---|---|"]
agg = {}
for name, calls, total, avg, pct in rows:
    k = short(name)
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += calls
    a[1] += total
    a[2] += pct
for k, (calls, total, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append("| %s | %d | %.1f | %.2f | %.2f |" % (k, calls, total, total / calls, pct))
res = db.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, grid_x from kernels group by name").fetchall()
lines += ["", "| kernel | vgpr | agpr | sgpr | lds B | scratch B | workgroup | grid |", "|---|---|---|---|---|---|---|---|"]
for r in res:
    if "ptd::" in r[0]:
        lines.append("| %s | %s |" % (short(r[0]), " | ".join(str(x) for x in r[1:])))
# the bench line printed by the profiled run itself: its event-timed launch average next to rocprof's
try:
    bench = [json.loads(l) for l in open(os.path.join(src, "trace.log")) if l.startswith('{"metric"')][-1]
    rl = bench["roofline"]
    # the production traversal kernel: k_extend<false, false> in batches that do not suspend rays (four frames or more),
    # k_extend<false, true> in those that do; whichever ran more often in this profile
    prod_name = max(("ptd::k_extend<false, false>", "ptd::k_extend<false, true>"), key=lambda k: agg.get(k, (0, 0.0))[0])
    prod = agg.get(prod_name)
    # the same launches bench.py timed: the plain traversal launches just before the counting pass (k_extend<true, ...>) begins
    window_avg = float("nan")
    try:
        disp = db.execute("select S.display_name, K.start, K.end from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S "
                          "on S.id = K.kernel_id and S.guid = K.guid where S.display_name like '%k_extend<%' order by K.start").fetchall()
        first_counting = min((st for name, st, en in disp if "k_extend<true" in name), default=None)
        plain = [(st, en) for name, st, en in disp if "k_extend<false" in name and (first_counting is None or st < first_counting)]
        win = plain[-int(rl["launches"]):]
        if win:
            window_avg = sum(en - st for st, en in win) / len(win) * 1e-6
    except sqlite3.Error:
        pass
    lines += ["", "bench.py inside this run: value %.1f %s, roofline.avg_launch_ms %.4f (HIP events on the launch's stream, %d launches of the timing window), "
              "concurrency %.2f; rocprofv3's average duration of those same launches: %.4f ms (of the production traversal kernel over the whole run: %.4f ms)"
              % (bench["value"], bench["unit"], rl["avg_launch_ms"], rl["launches"], rl["concurrency"], window_avg, prod[1] / prod[0] * 1e-3 if prod else float("nan")),
              "", "Kernel tracing costs throughput here: the profiler's per-dispatch bookkeeping keeps fewer launches in flight at a time "
              "(concurrency above, against roughly twice as many without the profiler), so each launch shares the chip with fewer others and is shorter "
              "than in an unprofiled run, where bench.py reports a higher value and a longer avg_launch_ms.  Compare the two numbers of this "
              "run with each other."]
except (OSError, IndexError, KeyError, ValueError):
    pass
open(os.path.join(out_dir, tag + "_kernel_stats.md"), "w").write("\n".join(lines) + "\n")

pmc = {}
for path in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_results.db"))):
    d = sqlite3.connect(path)
    q = "select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection group by kernel_name, counter_name"
    for kname, cname, n, avg, total in d.execute(q):
        if "ptd::" not in kname:
            continue
        pmc.setdefault(short(kname), {})[cname] = {"dispatches": n, "avg": avg, "sum": total}
# the production traversal kernel: k_extend<false, false> in batches that do not suspend rays, k_extend<false, true> in those
# that do (and then <false, false> only for the last round); whichever has more dispatches in this profile
ext = max((pmc.get("ptd::k_extend<false, false>", {}), pmc.get("ptd::k_extend<false, true>", {})),
          key=lambda v: next(iter(v.values()))["dispatches"] if v else 0)
summary = {"source": "rocprofv3 --pmc, one pass per counter group, bench.py --steps 16 --warmup 2 --cpu-seconds 0 --others 0", "per_kernel": pmc}
if "FETCH_SIZE" in ext:
    raw_kb = ext["FETCH_SIZE"]["avg"]
    wr_kb = ext.get("WRITE_SIZE", {}).get("avg", 0.0)
    # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes, so wide (16 B/lane) reads are under-reported 2x
    traffic = {"kernel": "k_extend", "fetch_kb_raw_per_launch": raw_kb, "write_kb_per_launch": wr_kb,
               "hbm_bytes_per_launch": int((2.0 * raw_kb + wr_kb) * 1024),
               "note": "FETCH_SIZE doubled per the gfx950 correction for 16-byte-per-lane reads; WRITE_SIZE as reported; "
                       "Infinity-Cache hits are included in these fabric-side counters"}
    summary["extend_traffic"] = traffic
    if tag.count("_") == 0:      # the benchmark workload's own profile
        json.dump(traffic, open(os.path.join(out_dir, "extend_traffic.json"), "w"), indent=1)
if "TCC_HIT_sum" in ext and "TCC_MISS_sum" in ext:
    h, m = ext["TCC_HIT_sum"]["sum"], ext["TCC_MISS_sum"]["sum"]
    summary["extend_l2_hit_rate"] = h / (h + m) if h + m else None

# ---- per ray segment: counter sums of the production kernels over the rays those launches traced ----
# every pass runs the same bench.py command: frame 1, the warm-up, the timed frames and 16 event-timed frames run the plain
# traversal kernels (k_extend<false, *>), then 16 counted frames run k_extend<true, *>; k_shade serves all of them
try:
    bench = [json.loads(l) for l in open(os.path.join(src, "trace.log")) if l.startswith('{"metric"')][-1]
    rays_per_frame = bench["config"]["rays_per_step_per_gpu"]
    plain_frames = 1 + bench["warmup"] + bench["steps"] + bench["roofline"]["profiled_frames"]
    all_frames = plain_frames + bench["roofline"]["profiled_frames"]

    def per_ray(kernels, frames):
        out = {}
        tot = {}
        for k in kernels:
            for c, v in pmc.get(k, {}).items():
                tot[c] = tot.get(c, 0.0) + v["sum"]
        rays = float(rays_per_frame) * frames
        if "SQ_INSTS_VALU" in tot:
            out["valu_insts_per_ray"] = tot["SQ_INSTS_VALU"] / rays
        if "SQ_ACTIVE_INST_VALU" in tot:
            out["valu_busy_cycles_per_ray"] = 4.0 * tot["SQ_ACTIVE_INST_VALU"] / rays     # the counter ticks every four cycles
        if "FETCH_SIZE" in tot:
            out["fabric_bytes_per_ray"] = (2.0 * tot["FETCH_SIZE"] + tot.get("WRITE_SIZE", 0.0)) * 1024.0 / rays
        if "TCC_HIT_sum" in tot and tot["TCC_HIT_sum"] + tot.get("TCC_MISS_sum", 0.0) > 0:
            out["l2_hit_rate"] = tot["TCC_HIT_sum"] / (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"])
        # GRBM_GUI_ACTIVE comes summed over the 8 XCDs; the _sum counters over the 256 per-CU units.  Counter passes run the
        # kernels one at a time, so these are each launch alone on the chip
        cu_cycles = 256.0 * tot["GRBM_GUI_ACTIVE"] / 8.0 if tot.get("GRBM_GUI_ACTIVE") else 0.0
        if "TA_TA_BUSY_sum" in tot and cu_cycles:
            out["ta_busy_frac"] = tot["TA_TA_BUSY_sum"] / cu_cycles
        if "TCP_PENDING_STALL_CYCLES_sum" in tot and cu_cycles:
            out["l1_pending_stall_frac"] = tot["TCP_PENDING_STALL_CYCLES_sum"] / cu_cycles     # the in-order L1 waiting on a line in flight
        if "TA_FLAT_READ_WAVEFRONTS_sum" in tot:
            out["vmem_read_insts_per_ray"] = tot["TA_FLAT_READ_WAVEFRONTS_sum"] / rays
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in tot and tot["TCP_TOTAL_CACHE_ACCESSES_sum"] > 0:
            out["l1_accesses_per_ray"] = tot["TCP_TOTAL_CACHE_ACCESSES_sum"] / rays
            if "TCP_TCC_READ_REQ_sum" in tot:
                out["l1_read_miss_per_access"] = tot["TCP_TCC_READ_REQ_sum"] / tot["TCP_TOTAL_CACHE_ACCESSES_sum"]
        if "SQ_WAIT_ANY" in tot and tot.get("SQ_WAVE_CYCLES"):
            out["parked_share_of_wave_cycles"] = tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"]
        return out

    per = {"source": "rocprofv3 --pmc passes of scripts/profile.sh (FETCH_SIZE doubled per the gfx950 rule for 16-byte-per-lane reads: an "
                     "estimate for per-lane node gathers), summed over all launches of the kernel and divided by the ray segments they traced",
           "commit": sys.argv[3] if len(sys.argv) > 3 else None, "csrc_sha16": csrc_sha16(), "tag": tag,
           "workload": bench["config"]["workload"], "rays_per_frame": rays_per_frame,
           "k_extend": per_ray(["ptd::k_extend<false, true>", "ptd::k_extend<false, false>"], plain_frames),
           "k_shade": per_ray([k for k in pmc if "k_shade" in k], all_frames)}
    summary["per_ray"] = per
    wl = bench["config"]["workload"].split(",")[0]
    json.dump(per, open(os.path.join(out_dir, "counters_per_ray_%s.json" % wl), "w"), indent=1)
except (OSError, IndexError, KeyError, ValueError) as e:
    print("no per-ray summary:", e)
json.dump(summary, open(os.path.join(out_dir, tag + "_pmc.json"), "w"), indent=1)
print("\n".join(lines[:14]))
print(json.dumps({k: v for k, v in summary.items() if k != "per_kernel"}, indent=1))
for k, v in pmc.items():
    print(k, {c: round(x["avg"], 1) for c, x in v.items()})
