"""CPU check of the bench.py output contract on the committed bench line (profiles/): required keys, types, and the
roofline / cpu_baseline objects the harness reads."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_bench():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line under profiles/"
    return json.load(open(files[-1]))


def test_bench_line_has_the_contract_keys():
    j = latest_bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["unit"] == "instances/s" and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and abs(j["value"] - j["config"]["instances_per_gpu"] * j["n_gpus"] / (j["ms_per_step"] / 1e3)) < 1e-6 * j["value"]


def test_roofline_and_cpu_baseline_objects():
    j = latest_bench()
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["peak"] in (8000.0, 157.3)
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    g = j["roofline_gls"]
    assert g["delta_evals_per_s"] > 1e10 and g["launches"] == j["steps"] * -(-j["config"]["instances_per_gpu"] // j["config"]["resident_instances_per_gpu"])
