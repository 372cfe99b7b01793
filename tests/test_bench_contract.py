"""The JSON line of bench.py against the benchmark contract: the line committed under profiles/ from the last profiled run
(no GPU needed: the file is data), and, on a GPU, the line a short live run prints."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check_line(d, live=False):
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "Mrays/s"
    for key in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "sponza_like" in d["config"]["workload"] and "Preetham" in d["config"]["workload"]       # the textured scene under a real sky
    assert d["config"]["nan_pixels"] == 0
    # value = rays per step / time per step
    assert abs(d["value"] - d["config"]["rays_per_step_per_gpu"] * d["n_gpus"] / d["ms_per_step"] / 1e3) < 0.01 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "counters_stale"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
    if r["counters_stale"]:
        # the committed counters belong to other kernel sources: the line says so and falls back to the algorithmic HBM figure
        assert r["bound"] == "hbm" and r["valu"] is None
    else:
        assert r["bound"] == "valu"
        assert 0.0 < r["hbm"]["fabric_frac"] < r["hbm"]["alg_frac"] < 1.0       # cache hits are why the algorithmic figure is the larger
        assert r["valu"]["profile_commit"] and r["valu"]["profile_csrc_sha16"] and r["k_shade"]["valu_insts_per_ray"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and 1 <= c["cores"] <= c["nproc"] and c["unit"] == "Mrays/s" and c["sample"]
    assert c["lavapipe"].startswith("unavailable:") or c["lavapipe"].startswith("available")
    if not live:
        names = " ".join(o["workload"] for o in d["other_workloads"])
        for want in ("sponza_class", "bistro_class", "3840x2160", "cornell", "pt_set_precision 1"):
            assert want in names, want
        for o in d["other_workloads"]:
            assert o["value"] > 0 and o["hierarchy"]["build_ms"] > 0 and o["hierarchy"]["first_build_ms"] > 0 and o["nan_pixels"] == 0


def test_committed_bench_line_has_the_contract_fields():
    path = os.path.join(ROOT, "profiles", "r03_bench_line.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    check_line(d)
    assert d["roofline"]["counters_stale"] is False


def test_committed_counters_belong_to_the_kernel_sources_in_the_tree():
    sys.path.insert(0, ROOT)
    import bench
    for wl in ("sponza_like", "sponza_class", "bistro_class"):
        c = json.load(open(os.path.join(ROOT, "profiles", "counters_per_ray_%s.json" % wl)))
        assert c["csrc_sha16"] == bench.csrc_sha16(), "profiles/counters_per_ray_%s.json was taken at other kernel sources: run scripts/profile_and_summarize.sh" % wl


@pytest.mark.gpu
def test_live_bench_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--others", "0", "--cpu-seconds", "1"],
                         capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"metric"')][-1])
    check_line(d, live=True)
    assert d["steps"] == 3 and d["warmup"] == 1 and d["n_gpus"] == 1
