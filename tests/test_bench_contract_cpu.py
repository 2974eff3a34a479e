"""CPU check of the bench.py output contract on the newest committed bench line (profiles/): required keys, types, and the
roofline / cpu_baseline objects the harness reads; plus the host-side pieces of bench.py (instance blocks, best-known file)."""
import glob
import importlib.util
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_bench():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[2-9]*_bench.json")))
    assert files, "no committed round-2+ bench line under profiles/"
    return json.load(open(files[-1]))


def bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_line_has_the_contract_keys():
    j = latest_bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "mean_gap_pct", "gap_reference"):
        assert k in j, k
    assert j["unit"] == "instances/s" and j["higher_is_better"] is True and j["scaling"] in ("weak", "strong")
    assert j["vs_baseline"] is None and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and abs(j["value"] - j["config"]["total_instances"] / (j["ms_per_step"] / 1e3)) < 1e-6 * j["value"]
    # the gap is measured against a committed artifact that is independent of the timed run
    assert j["mean_gap_pct"] is not None and "bench_data/" in j["gap_reference"]
    assert os.path.isfile(os.path.join(ROOT, j["gap_reference"].split(":")[0]))
    assert j["watchdog_aborts"] == 0


def test_roofline_and_cpu_baseline_objects():
    j = latest_bench()
    r = j["roofline"]
    # the kernel that owns the timed step is the search kernel: LDS-bound by design
    assert r["kernel"] == "gls_kernel" and r["bound"] == "lds" and r["unit"] == "GB/s" and r["peak"] == 150000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["delta_evals_per_s"] > 1e10 and r["launches"] == j["steps"] * j["config"]["rounds_per_rank"][0]
    for k in j["kernels"].values():
        assert k["bound"] in ("hbm", "mfma") and k["peak"] in (8000.0, 157.3)
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert c["per_core_value"] > 0 and "all" in c["sample"]


def test_instance_blocks_and_best_known_file(tmp_path):
    b = bench_module()
    from gnngls_amd.synthetic import random_instances
    D = b.instance_range(7, 6, 1020, 1030)                       # straddles blocks 0 and 1
    assert D.shape == (10, 6, 6)
    blk0 = random_instances(np.random.default_rng(7), 1024, 6)[0]
    blk1 = random_instances(np.random.default_rng(1007), 1024, 6)[0]
    assert np.array_equal(D[:4], blk0[1020:]) and np.array_equal(D[4:], blk1[:6])
    assert b.instance_range(7, 6, 5, 5).shape == (0, 6, 6)
    path = str(tmp_path / "bk.npz")
    a0 = np.arange(1024, dtype=np.float64)
    a1 = np.full(1024, np.nan)
    a1[:100] = 5.0
    np.savez(path, n=6, seed=7, how=np.array("unit test"), block0=a0, block1=a1)
    bk, how = b.load_best_known(path, 6, 7, 1000, 1030)
    assert np.array_equal(bk[:24], a0[1000:]) and (bk[24:] == 5.0).all() and "unit test" in how
    assert b.load_best_known(path, 6, 7, 1000, 1200)[0] is None         # beyond the covered part of block 1
    assert b.load_best_known(path, 6, 8, 0, 10)[0] is None              # another seed
    assert b.load_best_known(str(tmp_path / "none.npz"), 6, 7, 0, 10)[0] is None


@pytest.mark.parametrize("traffic", ["traffic_r02.json", "traffic_r03.json"])
def test_committed_pmc_figures_follow_from_the_committed_counter_files(traffic):
    """The gls_kernel entry of profiles/traffic_r0*.json (what bench.py prints as roofline.pmc / roofline.traffic) is exactly
    what scripts/pmc_summary.py derives from the counter CSVs it names -- no hand-edited numbers; likewise the forward
    kernels' entries of round 3 and scripts/pmc_forward_summary.py."""
    import subprocess
    import sys
    root = ROOT
    table = json.load(open(os.path.join(root, "profiles", traffic)))
    entry = table["gls_kernel"]
    src = entry["source"].split("/*_counter_collection.csv")[0]
    assert os.path.isdir(os.path.join(root, src)), src
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "pmc_summary.py"), os.path.join(root, src)],
                         capture_output=True, text=True, check=True).stdout
    derived = json.loads(out[out.index("{"):out.rindex("}") + 1])
    for k, v in derived.items():
        if isinstance(v, float):
            assert abs(entry[k] - v) <= 1e-9 * max(1.0, abs(v)), k
        else:
            assert entry[k] == v, k
    if "mfma_busy_frac" in table.get("ffn_fused", {}):
        fsrc = table["ffn_fused"]["source"].split("/*_counter_collection.csv")[0]
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", "pmc_forward_summary.py"), os.path.join(root, fsrc)],
                             capture_output=True, text=True, check=True).stdout
        derived = json.loads(out)
        for name, kern in (("ffn_fused", "ffn_fused_kernel"), ("gemm_fc", "gemm_f32_kernel"), ("gat_aggregate", "gat_rows_kernel")):
            for k, v in derived[kern].items():
                assert abs(table[name][k] - v) <= 1e-9 * max(1.0, abs(v)), (name, k)
            assert 0.3 < table[name]["mfma_busy_frac"] < 1.0


def test_shard_and_round_plan_of_an_eight_rank_launch(monkeypatch):
    """The driver's 8-GPU launch (one process per GPU, WORLD_SIZE=8) cannot be rehearsed on a one-GPU lease: the shard /
    round arithmetic every rank derives for BASELINE configs[3] (TSP100, 10,000 instances over 8 GPUs) and configs[4]
    (TSP200 x 256 per GPU) is checked here, and the gather layout it implies."""
    monkeypatch.setenv("WORLD_SIZE", "8")
    b = bench_module()
    from gnngls_amd import parallel
    # configs[3]: 10,000 instances, strong scaling: 1250 per rank = 2 rounds of 625 at the TSP100 residency of 1024
    covered = []
    for rank in range(8):
        total, lo, hi, sizes = b.shard_plan(10000, 1024, 8, rank)
        assert total == 10000 and hi - lo == 1250 and sizes == [1250] * 8 and lo == 1250 * rank
        assert b.round_plan(hi - lo, 0, 1024) == (1024, 2, 625)
        covered += list(range(lo, hi))
        D = b.instance_range(2024, 5, lo, min(lo + 3, hi))                    # straddles the seeded 1024-blocks correctly
        assert D.shape == (3, 5, 5)
    assert covered == list(range(10000))
    # an uneven test set: the last rank is short, the sizes every rank computes agree with its own shard
    for rank in range(8):
        total, lo, hi, sizes = b.shard_plan(10001, 1024, 8, rank)
        assert sizes == parallel.shard_sizes(10001, 8) and sizes[rank] == hi - lo and sum(sizes) == 10001
    # configs[4] per GPU: weak scaling, 256 TSP200 instances per rank, one round at the TSP200 residency (256)
    for rank in range(8):
        total, lo, hi, sizes = b.shard_plan(0, 256, 8, rank)
        assert (total, lo, hi, sizes) == (2048, 1024 * rank, 1024 * rank + 256, None)
        assert b.round_plan(256, 0, 256) == (256, 1, 256)
    # headline, weak scaling: block r per rank, all 1024 resident
    assert b.shard_plan(0, 1024, 8, 7) == (8192, 7168, 8192, None) and b.round_plan(1024, 0, 1024) == (1024, 1, 1024)
    # a device that cannot keep the instance size resident (capacity query 0) still gets a plan
    assert b.round_plan(100, 0, 0) == (64, 2, 50) and b.round_plan(0, 0, 1024) == (1024, 0, 1024)
