"""The JSON line of bench.py, as committed under profiles/ from the last profiled run: every field the benchmark contract
names is there and consistent (no GPU needed: the file is data)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    path = os.path.join(ROOT, "profiles", "r02_bench_line.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "Mrays/s"
    for key in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = rays per step / time per step
    assert abs(d["value"] - d["config"]["rays_per_step_per_gpu"] * d["n_gpus"] / d["ms_per_step"] / 1e3) < 0.01 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["bound"] == "valu" and 0.0 < r["frac"] < 1.0
    assert 0.0 < r["hbm"]["fabric_frac"] < r["hbm"]["alg_frac"] < 1.0       # cache hits are why the algorithmic figure is the larger
    assert r["valu"]["profile_commit"] and r["k_shade"]["valu_insts_per_ray"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "Mrays/s" and c["sample"]
