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
    """Every fraction of the committed line is re-derived from the line's own numbers.  Round 6 froze the primary one: SURVEY
    8(d)'s executed-LDS figure (executed delta evaluations x their algorithmic LDS bytes / aggregate LDS rate), measured in the run;
    the counters, the critical path of the serial phase and the issue model of the descent scans are side records."""
    j = latest_bench()
    r = j["roofline"]
    assert r["kernel"] == "gls_kernel" and r["launches"] == j["steps"] * j["config"]["rounds_per_rank"][0]
    assert r["bound"] == "lds" and r["unit"] == "GB/s" and r["peak"] == 150000.0 and "SURVEY 8(d)" in r["frac_definition"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1 and r["frac_source"] == "measured in this run"
    assert 0 < r["executed_evals_per_s"] <= r["reference_equivalent_evals_per_s"]
    assert abs(r["prune_ratio"] - r["executed_evals_per_s"] / r["reference_equivalent_evals_per_s"]) < 1e-9
    assert abs(r["achieved"] - r["executed_evals_per_s"] * r["lds_bytes_per_eval"] / 1e9) < 1e-6 * r["achieved"]
    assert abs(r["reference_equivalent_frac"] - r["reference_equivalent_evals_per_s"] * r["lds_bytes_per_eval"] / 1e9 / 150000.0) < 1e-9
    # the counters were collected on the workload of the line itself; two busiest pipes, sorted, no label
    pmc = r["pmc"]
    assert r["pmc_matches_workload"] and pmc["workload"]["n"] == j["config"]["n"] and pmc["workload"]["guide"] == "model"
    busy = {k[:-len("_busy_frac")]: v for k, v in pmc.items() if k.endswith("_busy_frac")}
    order = sorted(busy, key=busy.get, reverse=True)
    assert [p["pipe"] for p in r["busiest_pipes"]] == order[:2] and r["wave_wait_frac"] == pmc["wave_wait_frac"]
    # critical path: the penalty step against its committed chain floor, the descent scans against their committed issue model,
    # and the two phases together cover the kernel's cycles
    cp = r["critical_path"]
    committed = json.load(open(os.path.join(ROOT, "profiles", "r05_isa", "critical_path.json")))["tsp%d" % j["config"]["n"]]
    assert cp["floor_cycles"] == committed["chain_floor_cycles_per_step"] and abs(cp["frac"] - cp["floor_cycles"] / cp["measured_cycles"]) < 1e-9
    assert 1.5 < cp["clock_ghz"] < 2.5 and 10 < cp["penalty_steps_per_outer_iteration"] < 25 and 0.2 < cp["share_of_kernel_cycles"] < 0.8
    d = cp["descent"]
    model = json.load(open(os.path.join(ROOT, "profiles", "r06_isa", "descent_model.json")))["tsp%d" % j["config"]["n"]]
    assert cp["coverage_of_kernel_cycles"] >= 0.9 and abs(cp["coverage_of_kernel_cycles"] - cp["share_of_kernel_cycles"] - d["share_of_kernel_cycles"]) < 1e-9
    for key, sc in d["scans"].items():
        assert sc["issue_model_cycles"] == model[key]["issue_model_cycles"] and abs(sc["measured_over_issue_model"] - sc["measured_cycles"] / sc["issue_model_cycles"]) < 1e-9
        assert 0.8 < sc["measured_over_issue_model"] < 2.5
    # one full relocate scan per descent, the others over the flagged rows only; as many 2-opt scans as relocate scans
    assert abs(d["scans"]["relocate_full"]["per_outer_iteration"] - 1.0) < 1e-9
    assert abs(d["scans"]["two_opt"]["per_outer_iteration"] - 1.0 - d["scans"]["relocate_flagged"]["per_outer_iteration"]) < 1e-6
    # gap-versus-budget of the same timed step: never rises, ends at the headline gap
    c = j["gap_vs_budget"]
    assert len(c) >= 3 and all(a["t_s"] < b_["t_s"] for a, b_ in zip(c, c[1:]))
    assert all(a["mean_gap_pct"] >= b_["mean_gap_pct"] - 1e-12 for a, b_ in zip(c, c[1:]))
    assert abs(c[-1]["mean_gap_pct"] - j["mean_gap_pct"]) < 1e-9 and abs(c[-1]["budget_t_s"] - j["config"]["time_limit_s"]) < 1e-6
    q = j["iso_quality"]
    assert q["budget"] == "per_batch" and q["rounds"] >= 2 and abs(q["instances_per_s"] - q["instances"] / q["wall_s"]) < 1e-9 * q["instances_per_s"]
    assert q["wall_s"] < 1.1 * q["time_limit_s"] + 1.0 and q["instances_per_s"] > 2 * j["value"]
    f = q["frontier"]
    assert len(f) == 3 and f[0]["instances_per_s"] == q["instances_per_s"] and all(0 < p["forward_share"] < 1.2 for p in f)
    assert f[0]["instances_per_s"] < f[1]["instances_per_s"] < f[2]["instances_per_s"]
    assert f[0]["mean_gap_pct"] <= f[1]["mean_gap_pct"] <= f[2]["mean_gap_pct"]
    # throughput at fixed quality: three targets, measured passes within 15 % of their gap target, faster for the looser target,
    # the CPU port's single-core rate at the same mean gap beside each
    t = q["at_fixed_quality"]
    assert [e["target_mean_gap_pct"] for e in t] == [0.1, 0.03, 0.01]
    assert all(abs(e["measured_mean_gap_pct"] - e["target_mean_gap_pct"]) < 0.15 * e["target_mean_gap_pct"] for e in t)
    assert t[0]["instances_per_s"] > t[1]["instances_per_s"] > t[2]["instances_per_s"] > q["instances_per_s"]
    assert all(e["cpu_port_instances_per_s_per_core"] > 0 and e["instances_per_s"] > 100 * e["cpu_port_instances_per_s_per_core"] for e in t)
    e = j["true_gap_exact_sample"]                            # gap against PROVEN optima (branch and bound, checker side)
    assert e["instances"] >= 64 and e["mean_gap_pct"] >= -1e-9 and "bench_data/exact_optima" in e["source"]
    assert j["true_gap_bracket"]["bound_above_known_tour_instances"] == 0
    w = j["cpu_baseline"]["whole_box_estimate"]
    assert abs(w["instances_per_s"] - j["cpu_baseline"]["per_core_value"] * w["physical_cores"]) < 1e-9 * w["instances_per_s"]
    assert abs(w["gpu_over_whole_box"] - j["value"] / w["instances_per_s"]) < 1e-9 * w["gpu_over_whole_box"]
    cc = j["cpu_baseline"]["gap_vs_budget"]
    assert all(a["mean_gap_pct"] >= b_["mean_gap_pct"] - 1e-12 for a, b_ in zip(cc, cc[1:]))
    for k in j["kernels"].values():
        assert k["bound"] in ("hbm", "mfma") and (k["peak"] in (8000.0, 157.3) or abs(k["peak"] - 16 * 157.3 / 6) < 1e-6)      # (bf16 MFMA peak / 6: the bf16x3 feed-forward block)
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and "_raw_traces" not in c
    assert c["per_core_value"] > 0 and "all" in c["sample"]


def test_search_progress_record_helpers():
    """best_at_times / gap_curve_sums / gap_curve: the gap-versus-budget record of bench.py (test.py:97-117: best_cost =
    cummin over the progress rows, read at fixed times)."""
    b = bench_module()
    imp_cost = np.array([[10., 9., 8., 8., 0.], [5., 5., 0., 0., 0.], [7., 6., 5., 4., 3.]])
    imp_time = np.array([[0.01, 0.5, 2.0, 9.8, 0.], [0.2, 9.9, 0., 0., 0.], [0.1, 0.2, 0.3, 0.4, 9.7]])
    imp_len = np.array([4, 2, 9])                                  # third instance: 8 improvements + terminal in a buffer of 5
    bt, trunc = b.best_at_times(imp_cost, imp_time, imp_len, np.array([50., 40., 30.]), [0.1, 0.3, 1.0, 3.0, np.inf])
    assert bt.tolist() == [[10., 10., 9., 8., 8.], [40., 5., 5., 5., 5.], [7., 5., 4., 4., 3.]]
    assert trunc.tolist() == [False, False, True]
    sums = b.gap_curve_sums(bt[:2], np.array([8., 5.]))
    assert sums[:, 2].tolist() == [2.0] * 5 and sums[:, 1].tolist() == [0., 1., 1., 2., 2.]
    assert abs(sums[0, 0] - (25.0 + 700.0)) < 1e-9
    c = b.gap_curve(sums, [0.1, 0.3, 1.0, 3.0, 9.85], 0.15)
    assert [p["mean_gap_pct"] for p in c] == [362.5, 12.5, 6.25, 0.0, 0.0] and abs(c[-1]["budget_t_s"] - 10.0) < 1e-12
    # empty buffers: nothing but the start tour is known
    bt0, _ = b.best_at_times(np.zeros((2, 0)), np.zeros((2, 0)), np.array([1, 1]), np.array([3., 4.]), [1.0])
    assert bt0.tolist() == [[3.], [4.]]
    cores, threads = b.physical_cores()
    assert 1 <= cores <= threads


def test_time_to_gap_reads_the_improvement_traces():
    """bench.py's fixed-quality frontier: the search time after which the MEAN gap first meets a target."""
    b = bench_module()
    # two instances with best-known 10: instance 0 reaches 10.1 at 0.5 s and 10 at 2 s; instance 1 reaches 10 at 0.1 s
    imp_cost = np.array([[12., 10.1, 10., 10.], [10., 10., 0., 0.]])
    imp_time = np.array([[0.01, 0.5, 2.0, 9.9], [0.1, 9.9, 0., 0.]])
    imp_len = np.array([4, 2])
    t = b.time_to_gap(imp_cost, imp_time, imp_len, np.array([20., 20.]), np.array([10., 10.]), [15.0, 0.6, 0.51, 0.0, -1.0], 10.0)
    assert abs(t[0] - 0.1) < 0.003          # mean gap 10 % from 0.1 s on: (20 % + 0 %) / 2
    assert abs(t[1] - 0.5) < 0.011 and abs(t[2] - 0.5) < 0.011          # 0.5 % from 0.5 s on
    assert abs(t[3] - 2.0) < 0.041 and t[4] is None


def test_search_roofline_object_is_self_consistent():
    """Round 6: the primary fraction is SURVEY 8(d)'s executed-LDS figure, measured in the run; counters and the critical path are
    side records and there is no derived `bound` label any more."""
    b = bench_module()
    traffic = {"valu_busy_frac": 0.6, "lds_busy_frac": 0.7, "clock_ghz": 2.0, "hbm_bytes_per_instance_second": 10.0, "wave_wait_frac": 0.65,
               "workload": {"n": 100, "instances": 1024, "guide": "model"}}
    r = b.search_roofline(100, 2, 4000.0, 4, 8e9, 0.25, 1024, traffic, {"n": 100, "instances": 1024, "guide": "model"})
    # 4 launches of 1 s over 2 steps -> 2 launches per step; the evaluation counts are one step's
    assert abs(r["reference_equivalent_evals_per_s"] - 4e9) < 1 and abs(r["executed_evals_per_s"] - 1e9) < 1 and r["prune_ratio"] == 0.25
    assert r["bound"] == "lds" and r["unit"] == "GB/s" and r["peak"] == 150000.0 and "SURVEY 8(d)" in r["frac_definition"]
    assert abs(r["achieved"] - 1e9 * r["lds_bytes_per_eval"] / 1e9) < 1e-9 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-15
    assert abs(r["reference_equivalent_frac"] - 4.0 * r["frac"]) < 1e-12 and r["critical_path"] is None
    assert [p["pipe"] for p in r["busiest_pipes"]] == ["lds", "valu"] and r["wave_wait_frac"] == 0.65      # sorted, not hard-coded; no label
    assert r["pmc_matches_workload"] and r["traffic"] == 10.0 * 1024 * 1.0
    cyc = {"kernel_cycles": np.array([2e9, 2e9]), "pert_cycles": np.array([8e8, 8e8]), "steps": np.array([1.6e5, 1.6e5]),
           "ticks": np.array([1e8, 1e8]), "outer_iters": np.array([9000., 9000.])}
    rc = b.search_roofline(100, 2, 4000.0, 4, 8e9, 0.25, 1024, traffic, {"n": 100, "instances": 1024, "guide": "model"}, cyc)
    cp = rc["critical_path"]
    assert rc["frac"] == r["frac"] and cp["measured_cycles"] == 5000.0 and abs(cp["clock_ghz"] - 2.0) < 1e-12      # a side record only
    assert abs(cp["frac"] - cp["floor_cycles"] / 5000.0) < 1e-12
    assert abs(cp["share_of_kernel_cycles"] - 0.4) < 1e-12 and abs(cp["penalty_steps_per_outer_iteration"] - 160000 / 9000) < 1e-9
    # counters of another workload: no pipes, no traffic derived from them; the fraction is this run's own and stays
    rm = b.search_roofline(200, 1, 1000.0, 1, 1e9, 0.03, 256, traffic, {"n": 200, "instances": 256, "guide": "model"})
    assert rm["frac"] is not None and rm["traffic"] is None and rm["busiest_pipes"] is None and rm["pmc"] is None and not rm["pmc_matches_workload"]
    r2 = b.search_roofline(50, 1, 1000.0, 1, 1e9, 1.0, 128, {}, {"n": 50, "instances": 128, "guide": "model"})
    assert r2["prune_ratio"] == 1.0 and abs(r2["frac"] - r2["reference_equivalent_frac"]) < 1e-15 and r2["traffic"] is None
    r3 = b.search_roofline(100, 1, 1000.0, 1, 1e9, None, 64, {}, {"n": 100, "instances": 64, "guide": "model"})   # pruning, but no counting instantiation
    assert r3["prune_ratio"] is None and r3["executed_evals_per_s"] is None and r3["frac"] is None and r3["achieved"] is None
    r4 = b.search_roofline(50, 1, 1000.0, 1, 1e9, None, 64, {}, {"n": 50, "instances": 64, "guide": "model"})     # below n = 80 nothing prunes
    assert r4["prune_ratio"] is None and abs(r4["frac"] - r4["reference_equivalent_frac"]) < 1e-15


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


@pytest.mark.parametrize("traffic", [t for t in ("traffic_r02.json", "traffic_r03.json", "traffic_r04.json")
                                     if os.path.isfile(os.path.join(ROOT, "profiles", t))])
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
        ffn = "ffn_fused_bf16x3_kernel" if "ffn_fused_bf16x3_kernel" in derived else "ffn_fused_kernel"
        for name, kern in (("ffn_fused", ffn), ("gemm_fc", "gemm_f32_kernel"), ("gat_aggregate", "gat_rows_kernel")):
            if kern not in derived:          # (the inference forward has no gemm launch since the first fc rides in the embedding pass)
                continue
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


def test_best_known_files_cover_every_rank_of_an_eight_gpu_launch():
    """The gap of an N-rank line needs best-known lengths for every rank's instances: weak scaling gives rank r the first
    `batch` instances of block r -- TSP100: blocks 0-9 complete (also configs[3]'s 10,000-instance test set), TSP200: the first
    256 instances of blocks 0-7 (configs[4] on 8 GPUs), TSP50: block 0."""
    b = bench_module()
    for r in range(8):
        bk, how = b.load_best_known(None, 100, 2024, r * 1024, r * 1024 + 1024)
        assert bk is not None and np.isfinite(bk).all() and (bk > 5).all(), (r, how)
        bk, how = b.load_best_known(None, 200, 2024, r * 1024, r * 1024 + 256)
        assert bk is not None and np.isfinite(bk).all() and (bk > 8).all(), (r, how)
    assert b.load_best_known(None, 100, 2024, 0, 10000)[0] is not None
    assert b.load_best_known(None, 200, 2024, 0, 257)[0] is None                 # beyond the covered part of a block: no silent gaps
    assert b.load_best_known(None, 50, 2024, 0, 1024)[0] is not None
