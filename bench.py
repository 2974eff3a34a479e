#!/usr/bin/env python
"""bench.py -- headline benchmark of the gnngls hot path on MI355X.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--total_instances 10000]

Metric (BASELINE.json): TSP instances/sec + mean optimality gap at a fixed 10 s search budget,
TSP100.  One *step* = one pass of the hot path over one batch of synthetic instances already
resident in HBM: scaled edge features -> edge-regret GNN forward -> regret_pred guide ->
nearest-neighbour initial tour -> guided local search with the remaining part of the 10 s budget
(the reference starts the budget before the forward pass, scripts/test.py:64) -> ONE gather of the
per-instance results.

Workloads
  default                 BASELINE.json configs[2]: TSP100, 1024 instances per GPU, weak scaling (rank r searches
                          instance block r); all 1024 are resident at once -> one 10 s round per step.
  --total_instances N     BASELINE.json configs[3] with N = 10000: a FIXED test set split into contiguous shards
                          (gnngls_amd.parallel.shard_range, the reference's loop test.py:59 cut into W pieces), strong
                          scaling; a shard larger than the device residency takes ceil(shard / residency) rounds of the
                          full budget each (per-rank rounds are reported).

Instances: block k (1024 instances) = synthetic.random_instances(default_rng(seed + 1000 k)); the test set of
--total_instances is the concatenation of blocks 0, 1, ...  The gap denominator is the committed best-known file
(bench_data/, scripts/make_best_known.py: min over >= 60 s GPU searches with both guides, and a long CPU-oracle run
on a sample) -- independent of the timed run; Held-Karp optima with --exact_gap (n <= 20).

Prints ONE JSON line on rank 0 (see the keys in main()).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 (matrix)
PEAK_MFMA_BF16_TFLOPS = 16 * 157.3   # ibid.: the fp32 MFMA rate is 1/16 of the BF16 MFMA rate ("~2.5 PF dense")
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak
PEAK_LDS_GBS = 150000.0          # MI355X_MICROARCH.md: aggregate ds_read_b64 rate, every CU streaming
BLOCK = 1024                     # instances per seeded block
N_SIMDS = 1024                   # MI355X: 256 CUs x 4 SIMDs
GAP_GRID_S = (0.1, 0.3, 1.0, 3.0)   # search seconds at which the gap-versus-budget record is read (+ the end of the budget)
IMP_CAP = 256                    # improvement-trace entries kept per instance (a 10 s TSP100 search improves its best a few dozen times)
ISO_ROUNDS = 10                  # device loads that share ONE time limit in the iso-quality pass
ISO_FRONTIER = (1.0, 3.0, 10.0)  # ... at the full limit, a third and a tenth of it: per load what 10 / 30 / 100 loads get in the full one
ISO_GAP_TARGETS = (0.1, 0.03, 0.01)   # mean gaps (percent) at which throughput is reported: instances/s at FIXED quality
COUNT_PASS_S = 2.0               # length of the untimed pass that measures executed / reference-equivalent evaluations


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    # --tsp_n: the spelling to use under torchrun (its own parser rejects `--n` as an ambiguous prefix of --nnodes ...)
    ap.add_argument("--n", "--tsp_n", dest="n", type=int, default=100, help="TSP nodes per instance")
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU per step (weak scaling)")
    ap.add_argument("--total_instances", type=int, default=0,
                    help="strong scaling: a fixed test set of this many instances sharded over the ranks (configs[3]: 10000)")
    ap.add_argument("--resident_instances", type=int, default=0,
                    help="instances searched concurrently per GPU (0 = the device capacity for this n)")
    ap.add_argument("--budget", choices=["per_instance", "per_batch"], default="per_instance",
                    help="per_instance (the reference's meaning of --time_limit, the headline); per_batch: the rounds of a shard "
                         "SHARE one time limit (throughput end of the same trade; the gap says what it costs)")
    ap.add_argument("--time_limit", type=float, default=10.0)
    ap.add_argument("--perturbation_moves", type=int, default=20)
    ap.add_argument("--guides", nargs="+", default=["regret_pred"])
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_cores", type=int, default=0, help="host cores of the CPU baseline (0 = all)")
    ap.add_argument("--best_known", default=None, help="best-known file (default: bench_data/best_known_tsp{n}_seed{seed}.npz)")
    ap.add_argument("--no_iso_quality", action="store_true",
                    help="skip the untimed iso-quality pass (ISO_ROUNDS device loads sharing ONE time limit; rank 0, N = 1 only)")
    ap.add_argument("--no_gap_bracket", action="store_true",
                    help="skip the Held-Karp 1-tree lower bounds (oracle/one_tree.c, host cores, after the timed region)")
    ap.add_argument("--exact_gap", action="store_true",
                    help="n <= 20: gap against the exact optimum (oracle/held_karp.c, host cores) instead of best-known")
    return ap.parse_args()


def instance_range(seed, n, lo, hi):
    """Instances lo..hi-1 of the seeded test set (block k = instances 1024k .. 1024k+1023)."""
    from gnngls_amd.synthetic import random_instances
    parts = []
    for k in range(lo // BLOCK, -(-hi // BLOCK) if hi > lo else lo // BLOCK):
        D, _ = random_instances(np.random.default_rng(seed + 1000 * k), BLOCK, n)
        parts.append(D[max(lo - k * BLOCK, 0):min(hi - k * BLOCK, BLOCK)])
    return np.concatenate(parts) if parts else np.zeros((0, n, n))


def load_best_known(path, n, seed, lo, hi):
    """-> (fp64 [hi-lo] best-known tour lengths, description) or (None, reason)."""
    path = path or os.path.join(ROOT, "bench_data", f"best_known_tsp{n}_seed{seed}.npz")
    if not os.path.isfile(path):
        return None, f"no best-known file ({os.path.relpath(path, ROOT)})"
    z = np.load(path, allow_pickle=False)
    if int(z["n"]) != n or int(z["seed"]) != seed:
        return None, "best-known file is for another instance set"
    out = np.full(hi - lo, np.nan)
    for k in range(lo // BLOCK, -(-hi // BLOCK)):
        key = f"block{k}"
        if key not in z.files:
            continue
        a, b = max(lo, k * BLOCK), min(hi, (k + 1) * BLOCK)
        out[a - lo:b - lo] = z[key][a - k * BLOCK:b - k * BLOCK]
    if np.isnan(out).any():
        return None, "best-known file does not cover these instance blocks"
    return out, f"{os.path.relpath(path, ROOT)}: {str(z['how'])}"


def exact_sample_gap(n, seed, best_cost, best_known):
    """Gap against PROVEN optima for the leading instances of block 0 that have one (bench_data/exact_optima_tsp{n}_seed{seed}.npz,
    scripts/make_exact_optima.py): test.py:104's gap with the denominator the reference uses (the optimum), on a sample."""
    path = os.path.join(ROOT, "bench_data", f"exact_optima_tsp{n}_seed{seed}.npz")
    if not os.path.isfile(path):
        return None
    z = np.load(path, allow_pickle=False)
    sel = z["proven"] & (z["index"] < len(best_cost))
    idx, opt = z["index"][sel], z["optimum"][sel]
    if len(idx) == 0:
        return None
    gap = (best_cost[idx] / opt - 1.0) * 100.0
    return {"instances": int(len(idx)), "mean_gap_pct": float(gap.mean()), "max_gap_pct": float(gap.max()),
            "at_optimum_pct": float((np.abs(gap) <= 1e-9).mean() * 100.0),
            "best_known_is_optimal_pct": float((np.abs(best_known[idx] / opt - 1.0) <= 1e-11).mean() * 100.0),
            "mean_gap_vs_best_known_same_instances_pct": float(((best_cost[idx] / best_known[idx] - 1.0) * 100.0).mean()),
            "source": f"{os.path.relpath(path, ROOT)}: {str(z['how'])}"}


def best_at_times(imp_cost, imp_time, imp_len, init_cost, grid):
    """Search-progress record -> best tour length known at each search time of `grid` (test.py:97-117: best_cost = cummin
    over the progress rows, dt = time since the start).  imp_cost / imp_time [B, cap]: the returned best after every
    improvement and its time in seconds since the search started, imp_len [B] entries written (the last valid one is the
    terminal entry = state at the end of the search; see include/gnngls_hip.h), init_cost [B] = the start tour, which is all
    that exists before the first entry.  -> (best [B, len(grid)], truncated [B] bool: improvements beyond the buffer were
    dropped, so values between the last kept improvement and the end of the search are upper bounds)."""
    imp_cost, imp_time = np.asarray(imp_cost, dtype=np.float64), np.asarray(imp_time, dtype=np.float64)
    B, cap = imp_cost.shape
    valid = np.arange(cap)[None, :] < np.minimum(np.asarray(imp_len), cap)[:, None]
    out = np.empty((B, len(grid)))
    for g, t in enumerate(grid):
        m = valid & (imp_time <= t)
        c = np.where(m, imp_cost, np.inf).min(axis=1) if cap else np.full(B, np.inf)
        out[:, g] = np.minimum(c, np.asarray(init_cost, dtype=np.float64))
    return out, np.asarray(imp_len) > cap


def gap_curve_sums(best_t, best_known):
    """-> [G, 3] (sum of gaps in percent, instances at the best-known length, instances): additive over ranks."""
    gap = (best_t / best_known[:, None] - 1.0) * 100.0
    return np.stack([gap.sum(axis=0), (np.abs(gap) <= 1e-9).sum(axis=0).astype(np.float64),
                     np.full(gap.shape[1], float(gap.shape[0]))], axis=1)


def gap_curve(sums, grid, pre_search_s):
    """[G, 3] sums -> the gap_vs_budget list of the bench line.  t_s = seconds of SEARCH; the reference's budget clock
    (test.py:64 starts it before the forward pass) reads t_s + pre_search_s at that moment."""
    return [{"t_s": float(t), "budget_t_s": float(t + pre_search_s), "mean_gap_pct": float(sg / max(cnt, 1.0)),
             "at_best_known_pct": float(100.0 * ab / max(cnt, 1.0))} for t, (sg, ab, cnt) in zip(grid, sums)]


def time_to_gap(imp_cost, imp_time, imp_len, init_cost, best_known, targets, t_max):
    """Search seconds after which the MEAN gap of these instances is first <= each target (percent), from their improvement
    traces (best tour length known at t = cummin over the progress rows, test.py:97-117), on a 2 % geometric grid; None where the
    record never gets there within t_max."""
    grid = np.geomspace(1e-3, max(t_max, 2e-3), int(np.ceil(np.log(max(t_max, 2e-3) / 1e-3) / np.log(1.02))) + 1)
    bt, _ = best_at_times(imp_cost, imp_time, imp_len, init_cost, grid)
    gaps = ((bt / np.asarray(best_known)[:, None] - 1.0) * 100.0).mean(axis=0)
    out = []
    for g in targets:
        ok = np.nonzero(gaps <= g)[0]
        out.append(float(grid[ok[0]]) if len(ok) else None)
    return out


def physical_cores():
    """(physical cores, hardware threads) of the host from /proc/cpuinfo (unique (physical id, core id) pairs)."""
    pairs, threads, phys = set(), 0, None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                threads += 1
            elif line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((phys, line.split(":")[1].strip()))
    except OSError:
        pass
    threads = threads or (os.cpu_count() or 1)
    return (len(pairs) or threads), threads


def load_critical_path(n):
    """Committed dependent-chain / issue model of one penalty step of the search kernel's serial perturbation phase for TSP<n>
    (profiles/r05_isa/critical_path.json, made by scripts/isa_critical_path.py from the disassembly of the shipped instantiation
    and the measured per-instruction constants of scripts/isa_probe/latency_probe.hip), with the round-6 issue model of the descent
    scans under "descent" if profiles/r06_isa/descent_model.json has this size; None if there is no model for this size."""
    try:
        cp = json.load(open(os.path.join(ROOT, "profiles", "r05_isa", "critical_path.json"))).get(f"tsp{n}")
    except (OSError, ValueError):
        cp = None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r06_isa", "descent_model.json"))).get(f"tsp{n}")
    except (OSError, ValueError):
        d = None
    if d is not None:
        cp = dict(cp or {}, descent=d)
    return cp


def critical_path(n, cyc):
    """The search kernel's roofline as a latency / issue bound: MEASURED shader cycles per penalty step of the serial perturbation
    phase (the counting pass of this run: s_memtime around the phase, penalty steps counted by the kernel) against the committed
    floors of that step -- `floor_cycles` = the dependent chain alone (every instruction on the longest chain at its measured
    latency, memory round trips at their idle-chip latency: what no schedule of this algorithm on one wavefront can beat) and
    `issue_model_cycles` = an ESTIMATE of this code's own time: its executed instruction stream at the per-instruction costs of
    the micro-probe (one wavefront alone) plus the memory round trips nothing covers; the product kernel measures ~0.9 of it
    (independent scalar and vector instructions overlap more than the probe's chains)."""
    if cyc is None or cyc["steps"].sum() <= 0:
        return None
    steps, pert, kern = cyc["steps"].sum(), cyc["pert_cycles"].sum(), cyc["kernel_cycles"].sum()
    iters = max(cyc["outer_iters"].sum(), 1.0)
    measured = pert / steps
    cp = load_critical_path(n)
    out = {"phase": "penalty step of the serial perturbation phase (algorithms.py:150-185), wavefront 0 of the instance's workgroup",
           "measured_cycles": float(measured), "floor_cycles": None, "frac": None, "issue_model_cycles": None, "measured_over_issue_model": None,
           "penalty_steps_per_outer_iteration": float(steps / iters), "cycles_per_outer_iteration": float(kern / iters),
           "share_of_kernel_cycles": float(pert / kern), "clock_ghz": float(kern / (cyc["ticks"].sum() * 1e-8) / 1e9),
           "measured_how": "untimed %g s pass of this workload on the counting instantiation: s_memtime around every perturbation "
                           "phase / penalty steps counted by the kernel, all instances" % COUNT_PASS_S}
    if cp and "chain_floor_cycles_per_step" in cp:
        out.update({"floor_cycles": cp["chain_floor_cycles_per_step"], "frac": float(cp["chain_floor_cycles_per_step"] / measured),
                    "issue_model_cycles": cp["issue_model_cycles_per_step"], "measured_over_issue_model": float(measured / cp["issue_model_cycles_per_step"]),
                    "floor_source": cp.get("source")})
    out["coverage_of_kernel_cycles"] = float(pert / kern)
    if "descent_cycles" in cyc:
        # round 6: the other phase of an outer iteration, the descent (algorithms.py:188 -> 111-132), from the same counting pass:
        # shader cycles of wavefront 0 in the all-to-all scans of each kind, in the workgroup arg-min (incl. waiting for the slowest
        # wavefront) and in move application + barrier, against the issue model of profiles/r06_isa/descent_model.json: executed
        # instructions of ONE wavefront's share of a scan x the cycles an instruction costs a wavefront that shares its SIMD with
        # three others (the model's constant, stated there)
        desc = cyc["descent_cycles"].sum()
        model = (cp or {}).get("descent", {})
        kinds = ("two_opt_a2a (pruned scan from n = 80)",
                 "relocate_a2a over every row: the lean scan (for 80 <= n <= 127 only the first relocate scan of a descent), from n = 128 the scan "
                 "pruned by neighbour lists", "relocate_a2a, flagged rows only (quiet rows, 80 <= n <= 127)")
        keys = ("two_opt", "relocate_full", "relocate_flagged")
        scans = {}
        for q, (name, key) in enumerate(zip(kinds, keys)):
            cnt, cy = cyc["scans"][q].sum(), cyc["scan_cycles"][q].sum()
            if cnt <= 0:
                continue
            m = model.get(key, {}).get("issue_model_cycles")
            scans[key] = {"what": name, "per_outer_iteration": float(cnt / iters), "measured_cycles": float(cy / cnt),
                          "instructions_executed": model.get(key, {}).get("instructions_executed"),
                          "issue_model_cycles": m, "measured_over_issue_model": float(cy / cnt / m) if m else None}
        nscans = sum(c.sum() for c in cyc["scans"])
        out["descent"] = {"cycles_per_outer_iteration": float(desc / iters), "share_of_kernel_cycles": float(desc / kern),
                          "scans": scans,
                          "argmin_and_wait_cycles_per_scan": float(cyc["argmin_cycles"].sum() / max(nscans, 1)),
                          "apply_and_barrier_cycles_per_move": float(cyc["apply_cycles"].sum() / max(cyc["moves"].sum(), 1)),
                          "moves_per_outer_iteration": float(cyc["moves"].sum() / iters),
                          "model_source": model.get("source"), "cycles_per_instruction_of_the_model": model.get("cycles_per_instruction")}
        out["coverage_of_kernel_cycles"] = float((pert + desc) / kern)
    return out


def search_roofline(n, steps, gls_ms, gls_launches, ref_evals, exec_ratio, resident, traffic, workload, cyc=None):
    """Roofline object of the search kernel (the kernel that owns the timed step).

    What binds gls_kernel is vector-instruction issue: the primary fraction is the measured busy
    fraction of the vector ALUs -- a counter ratio, <= 1 by construction -- from the committed PMC passes on this workload
    (profiles/traffic_r0*.json; `pmc.workload` says what they were collected on).  Measured in this run, per launch, from
    the HIP events around the timed launches and the kernel's own counters: the reference-equivalent evaluation rate (every
    move the reference's scans evaluate, SURVEY 8d -- what rounds 1-3 were compared on: a work rate, not a physical
    fraction), and the EXECUTED rate = that x exec_ratio, the executed / reference-equivalent evaluations of a short untimed
    pass of the same workload on the counting instantiation of the kernel (the pruned descent scans only evaluate the moves
    that can qualify; None = not available for this configuration).  `lds_executed` = algorithmic LDS bytes of the executed
    evaluations against the aggregate LDS rate: <= 1 by construction, every executed evaluation reads at least these bytes."""
    n2 = (n - 2) * (n - 3) / 2.0
    nr = float((n - 2) * (n - 2))
    lds_bytes_per_eval = (48.0 * n2 + 68.0 * nr) / (n2 + nr)                # 48 B / 68 B per evaluation
    avg_launch_s = gls_ms / max(gls_launches, 1) * 1e-3
    launches_per_step = max(int(gls_launches) // max(steps, 1), 1)
    per_launch = lambda evals: evals / launches_per_step / avg_launch_s if avg_launch_s > 0 else 0.0   # noqa: E731
    ref_rate = per_launch(ref_evals)
    exec_rate = ref_rate * exec_ratio if exec_ratio is not None else None
    matches = bool(traffic.get("workload")) and all(traffic["workload"].get(k) == v for k, v in workload.items())
    busy = {k[:-len("_busy_frac")]: traffic[k] for k in traffic if k.endswith("_busy_frac") and traffic[k] is not None}
    order = sorted(busy, key=busy.get, reverse=True)
    crit = critical_path(n, cyc)
    # PRIMARY fraction, frozen from round 6 on (the round-5 review: "a roofline statement that means the same thing two rounds in a
    # row"): SURVEY 8(d)'s K3 figure -- algorithmic LDS bytes of the EXECUTED delta evaluations (48 B per 2-opt, 68 B per relocate
    # evaluation) per second of the dominant kernel's launches, against the aggregate LDS rate of the device.  Measured in this
    # run: HIP events around the timed launches x the kernel's own evaluation counter x the executed / reference-equivalent ratio
    # of the counting pass.  The exact pruning of the descent scans LOWERS it while the search gets faster (fewer evaluations for
    # the same moves), so outer iterations per budget stay the figure of merit beside it.
    exec_ratio_is_one = n < 80                               # no descent scan prunes below n = 80 (gls_prune_supported): executed = reference
    lds_rate = exec_rate if exec_rate is not None else (ref_rate if exec_ratio_is_one else None)
    ach = lds_rate * lds_bytes_per_eval / 1e9 if lds_rate is not None else None
    frac = ach / PEAK_LDS_GBS if ach is not None else None
    out = {
        "kernel": "gls_kernel", "bound": "lds",
        "achieved": ach, "peak": PEAK_LDS_GBS, "unit": "GB/s", "frac": frac,
        "frac_definition": "SURVEY 8(d), K3: executed delta evaluations per second of the gls_kernel launches (timed steps, HIP events) x "
                           "their algorithmic LDS bytes (48 B per 2-opt, 68 B per relocate evaluation) / aggregate LDS rate "
                           "(256 CUs x 128 B/clk x 2.4 GHz).  Same definition every round from round 6 on; r01-r05 recomputed: "
                           "0.22 / 0.305 / 0.33 (reference-equivalent, nothing pruned yet) / 0.231 / 0.293",
        "frac_source": "measured in this run" if frac is not None else None,
        "traffic": traffic["hbm_bytes_per_instance_second"] * resident * avg_launch_s
        if matches and "hbm_bytes_per_instance_second" in traffic else None,
        "avg_launch_ms": avg_launch_s * 1e3, "launches": int(gls_launches), "resident_instances": resident,
        "executed_evals_per_s": exec_rate, "reference_equivalent_evals_per_s": ref_rate,
        "prune_ratio": exec_ratio,
        "lds_bytes_per_eval": lds_bytes_per_eval,
        "reference_equivalent_frac": ref_rate * lds_bytes_per_eval / 1e9 / PEAK_LDS_GBS,
        "delta_evals_per_s": ref_rate,
        # side records, none of them the fraction: the serial phase against its dependent-chain floor and the descent scans against
        # their issue model (critical_path), the committed counters of this workload (pmc) and its two busiest pipes -- no label
        "critical_path": crit,
        "pmc": {k: traffic[k] for k in ("valu_busy_frac", "lds_busy_frac", "lds_bank_conflict_frac", "wave_wait_frac",
                                        "valu_insts_per_s", "lds_insts_per_s", "hbm_gbs", "clock_ghz", "workload", "source")
                if k in traffic} if matches else None,
        "busiest_pipes": [{"pipe": k, "busy_frac": busy[k]} for k in order[:2]] if order and matches else None,
        "wave_wait_frac": traffic.get("wave_wait_frac") if matches else None,
        "pmc_matches_workload": matches,
        "note": "reference_equivalent_evals_per_s is measured on the timed launches (HIP events + the kernel's counter of what the "
                "reference evaluates); prune_ratio = executed / reference-equivalent evaluations of a %g s untimed pass of the same "
                "workload on the counting instantiation of the kernel (pruned 2-opt scan from n = 80, quiet rows of the relocate scan for "
                "80 <= n <= 127, pruned relocate scan from n = 128: exact, they evaluate only the moves that can qualify); "
                "executed_evals_per_s is their product.  reference_equivalent_frac = the same LDS figure for every move the reference "
                "evaluates (a work rate: 1 / prune_ratio above the executed one).  The forward kernels' MFMA / HBM rooflines are "
                "under `kernels`" % COUNT_PASS_S,
    }
    return out


def kernel_rooflines(prof, n, B_chunk, n_layers):
    """Per forward-kernel class: algorithmic work per launch / measured average launch duration (HIP events).
    Algorithmic figures per (instance, layer) follow SURVEY.md 8(d) / DESIGN.md."""
    N = n * (n - 1) // 2
    E = N * 2 * (n - 2)
    M = B_chunk * N
    out = {}

    def add(name, kinds, bound, work_per_launch, unit_peak, unit):
        ms = sum(prof[k][0] for k in kinds)
        launches = min(prof[k][1] for k in kinds)
        if launches == 0 or ms <= 0:
            return
        avg_s = ms / launches * 1e-3
        achieved = work_per_launch / avg_s / (1e12 if unit == "TFLOP/s" else 1e9)
        out[name] = {"bound": bound, "achieved": achieved, "peak": unit_peak, "unit": unit,
                     "frac": achieved / unit_peak, "traffic": None, "avg_launch_ms": avg_s * 1e3,
                     "launches": int(launches), "total_ms": ms}

    # The feed-forward block runs on the bf16 matrix pipe with three bf16 pieces per fp32 operand and six piece products per product
    # (model_kernels.hip, ffn_fused_bf16x3_kernel: fp32 accuracy, the 1e-5 parity bar unchanged): `achieved` stays in algorithmic, i.e.
    # fp32-equivalent FLOP -- 4 M 128 512 of the block plus 2 M 128 128 of the next layer's fc that all but the last launch carry -- and the
    # peak is the dense bf16 MFMA peak / 6.  GNNGLS_FFN_FP32=1 keeps the block on the fp32 pipe (peak 157.3).
    ffn_fp32 = os.environ.get("GNNGLS_FFN_FP32", "0") not in ("", "0")
    if ffn_fp32:
        add("ffn_fused", ["ffn_fused"], "mfma", 4.0 * M * 128 * 512, PEAK_MFMA_F32_TFLOPS, "TFLOP/s")
    else:
        fold = (n_layers - 1) / n_layers if n_layers > 0 else 0.0
        add("ffn_fused", ["ffn_fused"], "mfma", 4.0 * M * 128 * 512 + fold * 2.0 * M * 128 * 128, PEAK_MFMA_BF16_TFLOPS / 6.0, "TFLOP/s")
        if "ffn_fused" in out:
            out["ffn_fused"].update({"arithmetic": "bf16 MFMA, three pieces per fp32 operand, six products (fp32-equivalent FLOP; peak = 2516.8 / 6)",
                                     "frac_of_fp32_mfma_peak": out["ffn_fused"]["achieved"] / PEAK_MFMA_F32_TFLOPS,
                                     "carries_next_layers_fc": True})
    add("gemm_fc", ["gemm_fc"], "mfma", 2.0 * M * 128 * 128, PEAK_MFMA_F32_TFLOPS, "TFLOP/s")
    # K1 (attention + aggregation = gat_rows; the merge + skip + BN1 is fused into ffn_fused), SURVEY 8(d): N*1600 B and E*304 FLOP per
    # (instance, layer).  Arithmetic intensity = 0.19*deg FLOP/B: above the fp32 ridge (157.3 TF / 8 TB/s = 19.7
    # FLOP/B, i.e. n > ~54) the weighted sums on the f32 MFMA are the floor, below it HBM is.
    k1_bytes, k1_flops = 1600.0 * M, 304.0 * E * B_chunk
    if k1_flops / (PEAK_MFMA_F32_TFLOPS * 1e12) > k1_bytes / (PEAK_HBM_GBS * 1e9):
        add("gat_aggregate", ["gat_rows"], "mfma", k1_flops, PEAK_MFMA_F32_TFLOPS, "TFLOP/s")
    else:
        add("gat_aggregate", ["gat_rows"], "hbm", k1_bytes, PEAK_HBM_GBS, "GB/s")
    # K1' (round 6): the first GATConv in its rank-1 form (one input feature): no MFMA, no ft tile; its algorithmic traffic is what it
    # writes -- two compact partials of 24 floats (shift, sum of weights, sum of weights x feature per head) per line-graph node --
    # over the node's one feature.  It is bound by its vector instructions (one softmax weight per (destination, source, head)), so
    # the HBM fraction is small by construction.
    add("gat_rank1", ["gat_rows_rank1"], "hbm", (2 * 96.0 + 4.0) * M, PEAK_HBM_GBS, "GB/s")
    traffic = load_traffic()
    for name, v in out.items():
        if name in traffic and "hbm_bytes_per_row" in traffic[name]:
            v["traffic"] = traffic[name]["hbm_bytes_per_row"] * M
        if name in traffic and "mfma_busy_frac" in traffic[name]:       # committed SQ_VALU_MFMA_BUSY_CYCLES pass (profiles/r03_pmc_forward)
            v["pmc"] = {k: traffic[name][k] for k in ("mfma_busy_frac", "valu_busy_frac", "lds_busy_frac", "valu_insts_per_mfma",
                                                      "wave_wait_frac", "source") if k in traffic[name]}
    if "gat_aggregate" in out:
        t = out["gat_aggregate"]["avg_launch_ms"] * 1e-3
        out["gat_aggregate"]["algorithmic_gbs"] = k1_bytes / t / 1e9
        out["gat_aggregate"]["algorithmic_tflops"] = k1_flops / t / 1e12
    return out


def load_traffic():
    """HBM traffic from the committed PMC passes (profiles/traffic_r0*.json, newest first: FETCH_SIZE/WRITE_SIZE collected
    and corrected as MI355X_MICROARCH.md prescribes)."""
    merged = {}
    for name in ("traffic_r01.json", "traffic_r02.json", "traffic_r03.json", "traffic_r04.json", "traffic_r05.json", "traffic_r06.json"):
        try:
            merged.update(json.load(open(os.path.join(ROOT, "profiles", name))))
        except (OSError, ValueError):
            pass
    return merged


def available_cores():
    """CPUs this process may actually use: min(online CPUs, scheduler affinity, cgroup CPU quota).  On the GPU boxes the
    container sees 256 hardware threads but its cgroup grants 16 CPUs; more worker processes than that only time-slice."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(D, guides, init_tour, init_cost, best_known, time_limit, pm, cores):
    """The CPU oracle (oracle/gls_oracle.c, the parity-pinned restatement of the reference's
    guided_local_search) timed on ALL CPUs available to this container (available_cores()): one instance per core (the
    reference is single-threaded, test.py:59), same instances, same budget, in child processes.  Bounded sample: `cores`
    instances, one budget."""
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "sample.npz")
        np.savez(path, D=D, guides=guides, init_tour=init_tour, init_cost=init_cost)
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline_worker.py"), path,
                                   str(i), str(time_limit), str(pm)], stdout=subprocess.PIPE, cwd=ROOT)
                 for i in range(cores)]
        outs = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in procs]
        wall = time.time() - t0
    costs = np.array([o["best_cost"] for o in outs])
    gaps = (costs / best_known[:cores] - 1.0) * 100.0 if best_known is not None else None
    search = float(np.mean([o["search_s"] for o in outs]))
    curve = None
    if best_known is not None:
        cap = max(len(o["imp_cost"]) for o in outs)
        pad = lambda xs, fill: np.array([list(o[xs]) + [fill] * (cap - len(o[xs])) for o in outs])   # noqa: E731
        grid = list(GAP_GRID_S) + [time_limit]
        bt, trunc = best_at_times(pad("imp_cost", np.inf), pad("imp_time", np.inf), np.array([o["imp_len"] for o in outs]),
                                  init_cost[:cores], [t for t in grid[:-1]] + [np.inf])
        curve = [p for p in gap_curve(gap_curve_sums(bt, best_known[:cores]), grid, 0.0) if p["t_s"] <= time_limit]
    phys, threads = physical_cores()
    per_core = 1.0 / search
    raw = None
    if best_known is not None:
        raw = {"imp_cost": pad("imp_cost", np.inf), "imp_time": pad("imp_time", np.inf), "imp_len": np.array([o["imp_len"] for o in outs]),
               "init_cost": np.asarray(init_cost[:cores]), "best_known": np.asarray(best_known[:cores])}
    return {"_raw_traces": raw, "value": cores / wall, "unit": "instances/s", "cores": cores, "kind": "port",
            "per_core_value": per_core,
            # the same search-progress record as the GPU line's gap_vs_budget (the CPU leg is not charged the forward pass:
            # its budget clock is its search clock)
            "gap_vs_budget": curve,
            # NOT measured: what the whole host would deliver if every physical core ran one instance at the measured
            # per-core rate (the container's cgroup grants `cores` CPUs of the box's hardware threads) -- an upper bound for
            # the CPU (no shared-cache / memory / clock effects), so gpu_value / this is the conservative GPU-vs-box ratio
            "whole_box_estimate": {"physical_cores": phys, "hardware_threads": threads, "instances_per_s": per_core * phys,
                                   "assumption": "per_core_value x physical cores, linear scaling, one single-threaded search per "
                                                 "core (the reference is single-threaded, test.py:59); not measured"},
            "sample": f"{cores} TSP{D.shape[1]} instances of the same batch, one per core on all {available_cores()} CPUs "
                      f"available to this container (cgroup quota; the box has {os.cpu_count()} hardware threads), "
                      f"{time_limit:g} s search budget each (GNN forward not charged to the CPU), guides as on "
                      f"the GPU; value = all available cores incl. process start-up, per_core_value = 1 / mean search time",
            "mean_gap_pct": float(gaps.mean()) if gaps is not None else None,
            "outer_iters_per_instance": float(np.mean([o["outer_iters"] for o in outs])),
            "delta_evals_per_s": float(sum(o["evals"] for o in outs) / search),
            "delta_evals_per_s_per_core": float(np.mean([o["evals"] / o["search_s"] for o in outs])), "wall_s": wall,
            # context, NOT measured on this box: the reference's own Python on one core of the build container
            # (SURVEY.md section 6 / BASELINE.md section 2, TSP100): the C port above is ~1400x faster per core
            "reference_python_probe": {"outer_iters_per_instance_10s": 41, "delta_evals_per_s": 5.1e5,
                                       "instances_per_s_per_core": 0.1, "source": "BASELINE.md section 2"}}


def iso_quality_pass(args, n, chunk, model, scalers, pipeline, trace=None, cpu=None):
    """Untimed extra pass (rank 0, N = 1): ISO_ROUNDS device loads of `chunk` instances searched within ONE time limit
    (budget="per_batch": the rounds share it, each instance is searched for time_limit / rounds including its forward
    pass) -- the throughput end of the budget-versus-quality trade, with the gap there to judge it.  This is the number
    that moves with kernel speed: the headline's instances/s is residency / budget by construction."""
    covered = 0                                          # leading instances of the seeded set with a best-known length
    while load_best_known(args.best_known, n, args.seed, covered, covered + chunk)[0] is not None and covered < ISO_ROUNDS * chunk:
        covered += chunk
    rounds = covered // max(chunk, 1)
    if rounds < 2:
        return {"skipped": f"needs >= 2 device loads ({chunk} instances each) with best-known lengths; the file covers {covered}"}
    total = rounds * chunk
    Dh = instance_range(args.seed, n, 0, total)
    bk, _ = load_best_known(args.best_known, n, args.seed, 0, total)
    D = torch.from_numpy(Dh).cuda()

    def point(limit):
        torch.cuda.synchronize()
        t0 = time.time()
        r = pipeline.solve_batch(D, model, scalers, guides=args.guides, time_limit=limit,
                                 perturbation_moves=args.perturbation_moves, chunk=chunk, budget="per_batch")
        best = r.best_cost.cpu().numpy()
        wall = time.time() - t0
        gap = (best / bk - 1.0) * 100.0
        return {"instances": total, "rounds": rounds, "instances_per_round": chunk, "time_limit_s": limit,
                "per_load_budget_s": limit / rounds, "equivalent_loads_in_full_limit": rounds * args.time_limit / limit,
                "budget": "per_batch", "wall_s": wall, "instances_per_s": total / wall,
                "mean_gap_pct": float(gap.mean()), "max_gap_pct": float(gap.max()),
                "instances_at_reference_pct": float((np.abs(gap) <= 1e-9).mean() * 100.0),
                "outer_iters_per_instance": float(r.outer_iters.double().mean()),
                "forward_s": r.timing["forward_s"], "init_s": r.timing["init_s"], "search_s": r.timing["search_s"],
                # share of the wall the GNN forward passes take: what bounds the throughput end of the trade
                "forward_share": r.timing["forward_s"] / wall}

    # throughput-at-quality frontier: the same `rounds` device loads inside ONE limit of time_limit, / 3 and / 10 -- per load
    # what 10 / 30 / 100 loads would get inside the full limit (1 / 0.33 / 0.1 s at the headline)
    pts = [point(args.time_limit / f) for f in ISO_FRONTIER]
    out = dict(pts[0])
    out["frontier"] = pts
    # throughput at FIXED quality: the per-load budget at which the mean gap reaches each target, read from the improvement traces
    # of the last timed step (a short search is the prefix of a long one: same kernel, same trajectory), then MEASURED: the same
    # `rounds` device loads inside rounds x (pre-search + that search time); the CPU port's single-core rate at the same mean gaps
    # from its own record beside it (search only: the CPU leg is not charged the forward pass)
    if trace is not None:
        t_gap = time_to_gap(trace["imp_cost"], trace["imp_time"], trace["imp_len"], trace["init_cost"], trace["best_known"],
                            ISO_GAP_TARGETS, args.time_limit)
        targets = []
        for g, tg in zip(ISO_GAP_TARGETS, t_gap):
            e = {"target_mean_gap_pct": g, "search_s_from_trace": tg, "pre_search_s": trace["pre_search_s"]}
            if tg is not None:
                m = point(rounds * (trace["pre_search_s"] + tg))
                e.update({"predicted_instances_per_s": chunk / (trace["pre_search_s"] + tg), "measured": m,
                          "instances_per_s": m["instances_per_s"], "measured_mean_gap_pct": m["mean_gap_pct"]})
            if cpu is not None:
                tc = time_to_gap(cpu["imp_cost"], cpu["imp_time"], cpu["imp_len"], cpu["init_cost"], cpu["best_known"], [g], args.time_limit)[0]
                e["cpu_port_search_s"] = tc
                e["cpu_port_instances_per_s_per_core"] = (1.0 / tc) if tc else None
                e["cpu_port_sample_instances"] = int(len(cpu["init_cost"]))
            targets.append(e)
        out["at_fixed_quality"] = targets
    out["how"] = (f"{rounds} device loads of {chunk} instances (blocks 0.. of the seeded set) through solve_batch(budget='per_batch'): "
                  f"ONE limit for all of them, forward passes included; untimed passes after the timed steps at limits "
                  + ", ".join(f"{args.time_limit / f:g} s" for f in ISO_FRONTIER) + " (`frontier`; the top-level fields are the first)")
    return out


def shard_plan(total_instances, batch, world, rank):
    """-> (total, lo, hi, sizes): which instances of the seeded test set rank `rank` of `world` searches.
    total_instances > 0: strong scaling, a fixed test set in contiguous shards (gnngls_amd.parallel.shard_range), sizes =
    rows per rank for the gather; else weak scaling: rank r takes the first `batch` instances of block r, sizes = None."""
    from gnngls_amd import parallel
    if total_instances > 0:
        lo, hi = parallel.shard_range(total_instances, world, rank)
        return total_instances, lo, hi, parallel.shard_sizes(total_instances, world)
    if batch > BLOCK:
        raise SystemExit("--batch > 1024: use --total_instances for larger test sets")
    return world * batch, rank * BLOCK, rank * BLOCK + batch, None


def round_plan(B, resident_instances, capacity):
    """-> (chunk, rounds, chunk_eff): a shard of B instances is searched in `rounds` device loads of `chunk_eff` instances
    (equal rounds, each with the full budget); chunk = instances resident at once (the device capacity unless overridden)."""
    chunk = resident_instances or (capacity if capacity > 0 else 64)
    rounds = -(-B // chunk) if B > 0 else 0
    return chunk, rounds, (-(-B // rounds) if rounds else chunk)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if world > 1:
        # the host driver of the GPU boxes only supports dmabuf IPC: without this RCCL's buffer exchange between the ranks of a
        # node fails with `hipIpcGetMemHandle: invalid argument`.  Must be in the environment before the first HIP call (none
        # has been made: device_count() below does not initialise the runtime); never overrides the launcher's own setting
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    n_dev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % max(n_dev, 1))
    import torch.distributed as dist
    # backend "nccl" is RCCL on ROCm; GNNGLS_DIST_BACKEND=gloo lets the multi-rank path be exercised on a box
    # with fewer GPUs than ranks (the gather then goes through host memory)
    backend = os.environ.get("GNNGLS_DIST_BACKEND", "nccl")
    # GNNGLS_DIST_SINGLE=1: form the process group also for ONE rank, so that a one-GPU box runs every collective of the
    # N-rank path on RCCL (tests/test_bench_gpu.py); the driver's N=1 run does not set it and touches no collective
    grouped = world > 1 or os.environ.get("GNNGLS_DIST_SINGLE", "0") == "1"
    def die(stage, exc):
        # a multi-GPU launch that cannot form its group or run its first collective must fail LOUDLY (non-zero exit, who / where /
        # what), in this fresh process: never re-exec a process that has touched the GPU
        print(f"[bench] FATAL rank {rank}/{world} (local_rank {local_rank}, device cuda:{local_rank % max(n_dev, 1)} of {n_dev} visible, "
              f"backend {backend}, MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')}, "
              f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}): {stage} failed: {type(exc).__name__}: {exc}",
              file=sys.stderr, flush=True)
        os._exit(3)

    if grouped:
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
            else:
                dist.init_process_group(backend)
        except Exception as exc:  # noqa: BLE001
            die("init_process_group", exc)

    from gnngls_amd import _lib, ops, parallel, pipeline

    n = args.n
    strong = args.total_instances > 0
    total, lo, hi, sizes = shard_plan(args.total_instances, args.batch, world, rank)
    B = hi - lo
    D_host = instance_range(args.seed, n, lo, hi)
    D = torch.from_numpy(D_host).cuda()
    need_model = "regret_pred" in args.guides
    model = pipeline.synthetic_model(seed=1234) if need_model else None
    # the feature scaler belongs to the test set (preprocess_dataset.py:39-48), not to a shard: fitted on block 0
    scalers = pipeline.Scalers.fit_weights(torch.from_numpy(instance_range(args.seed, n, 0, min(BLOCK, max(total, 1)))).cuda()) \
        if need_model else None
    chunk, rounds, chunk_eff = round_plan(B, args.resident_instances, ops.gls_resident_capacity(n))
    n_layers = len(model.message_passing_layers) if need_model else 0

    def barrier():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    gathered = None

    def step():
        nonlocal gathered
        r = pipeline.solve_batch(D, model, scalers, guides=args.guides, time_limit=args.time_limit,
                                 perturbation_moves=args.perturbation_moves, chunk=chunk_eff, budget=args.budget,
                                 imp_cap=IMP_CAP)
        local = torch.stack([r.best_cost, r.init_cost, r.outer_iters.double(), r.evals.double(),
                             r.status.double()], dim=1).contiguous()           # [B, 5] fp64
        if grouped and backend != "nccl":
            g = parallel.gather_results(local.cpu(), sizes)
            gathered = g.cuda() if g is not None else None
        else:
            gathered = parallel.gather_results(local, sizes)                   # the one collective of the path
        return r

    try:
        if grouped:
            barrier()                                        # the first collective of the run: fail here, with a message, not in a step
        for _ in range(args.warmup):
            step()
        barrier()
    except Exception as exc:  # noqa: BLE001
        if grouped:
            die("first collective / warm-up step", exc)
        raise
    # count the collectives the timed steps issue (the path has ONE per step: the gather; the barriers bracket the region)
    collectives = {}
    originals = {}
    if grouped:
        for name in ("gather", "all_gather", "all_gather_into_tensor", "all_reduce", "broadcast", "reduce_scatter", "all_to_all",
                     "scatter", "reduce", "send", "recv"):
            if hasattr(dist, name):
                originals[name] = getattr(dist, name)

                def counted(*a, _f=originals[name], _n=name, **k):
                    collectives[_n] = collectives.get(_n, 0) + 1
                    return _f(*a, **k)
                setattr(dist, name, counted)
    _lib.profile_enable(True)
    t0 = time.time()
    last = None
    for _ in range(args.steps):
        last = step()
    for name, f in originals.items():
        setattr(dist, name, f)
    barrier()
    dt = time.time() - t0
    prof = _lib.profile_collect()
    _lib.profile_enable(False)

    stats_dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([dt], dtype=torch.float64, device=stats_dev)
    # per-rank search-kernel statistics of the timed region (for the roofline of the whole job), summed / maxed below
    gls_ms, gls_launches = prof["gls"]
    # gap-versus-budget record of this rank's instances (last timed step): best tour length known after t seconds of search,
    # from the improvement trace the search kernel wrote (zero extra device time), against the best-known lengths
    grid = list(GAP_GRID_S)
    search_end = np.inf
    if args.exact_gap:
        bk_local, curve_reason = None, "exact optima are computed on rank 0 after the run"
    else:
        bk_local, curve_reason = load_best_known(args.best_known, n, args.seed, lo, hi)
    curve_sums = np.zeros((len(grid) + 1, 3))
    truncated = 0
    pre_search_s = 0.0
    if B > 0:
        pre_search_s = float((last.launch_time - last.start_time).mean())
        if bk_local is not None:
            bt, trunc = best_at_times(last.imp_cost.cpu().numpy(), last.imp_time.cpu().numpy(), last.imp_len.cpu().numpy(),
                                      last.init_cost.cpu().numpy(), grid + [search_end])
            curve_sums = gap_curve_sums(bt, bk_local)
            truncated = int(trunc.sum())
    mine = torch.tensor([gls_ms, float(gls_launches), float(rounds), -1.0,
                         float(last.evals.sum()) if B > 0 else 0.0, float(truncated), pre_search_s * B,
                         float(torch.cuda.current_device())]
                        + curve_sums.reshape(-1).tolist(), dtype=torch.float64, device=stats_dev)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    if grouped:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_gather(per_rank, mine)                                        # outside the timed region: reporting only
    else:
        per_rank = [mine]
    dt = t.item()

    # executed-versus-reference evaluation ratio and cycle budget of the search kernel: one short UNTIMED pass of the same
    # workload on the counting instantiation of the kernel (counting costs 2-3 %, so the timed steps never run it); rank 0
    # only, AFTER the reporting collectives (the other ranks do not wait for it there; they meet again at the final barrier)
    exec_ratio, cyc = -1.0, None
    if rank == 0 and B > 0:
        rc = pipeline.solve_batch(D[:min(chunk_eff, B)], model, scalers, guides=args.guides, time_limit=min(COUNT_PASS_S, args.time_limit),
                                  perturbation_moves=args.perturbation_moves, count_executed=True)
        if int(rc.evals_executed.min()) >= 0:
            exec_ratio = float(rc.evals_executed.sum()) / max(float(rc.evals.sum()), 1.0)
            rec = [x.double().cpu().numpy() for x in rc.timing.get("cycle_records", [])]
            if len(rec) >= 4 and rec[0].sum() > 0:
                cyc = {"kernel_cycles": rec[0], "pert_cycles": rec[1], "steps": rec[2], "ticks": rec[3],
                       "outer_iters": rc.outer_iters.double().cpu().numpy()}
                if len(rec) >= 14:           # ABI v4: the cycle account of the descent (include/gnngls_hip.h, GNNGLS_EXEC_RECORDS)
                    cyc.update({"descent_cycles": rec[4], "scans": [rec[5], rec[7], rec[9]], "scan_cycles": [rec[6], rec[8], rec[10]],
                                "argmin_cycles": rec[11], "apply_cycles": rec[12], "moves": rec[13]})

    if rank == 0:
        g = gathered.cpu().numpy()                                             # [total, 5], all ranks' instances in order
        value = total * args.steps / dt
        if args.exact_gap:
            # the exact optimum replaces the Concorde labels of the reference's (LFS-stub) instance
            # files; checker only -- computed on the host after the timed region
            from oracle import held_karp
            bk = held_karp.optima(instance_range(args.seed, n, 0, total) if strong else
                                  np.concatenate([instance_range(args.seed, n, r * BLOCK, r * BLOCK + args.batch) for r in range(world)]))
            gap_reference = "exact optimum (Held-Karp DP on the host, oracle/held_karp.c)"
        elif strong:
            bk, gap_reference = load_best_known(args.best_known, n, args.seed, 0, total)
        else:
            parts = [load_best_known(args.best_known, n, args.seed, r * BLOCK, r * BLOCK + args.batch) for r in range(world)]
            gap_reference = parts[0][1]
            bk = np.concatenate([p[0] for p in parts]) if all(p[0] is not None for p in parts) else None
            if bk is None:
                gap_reference = next(p[1] for p in parts if p[0] is None)
        gap = (g[:, 0] / bk - 1.0) * 100.0 if bk is not None else None
        # True optimality gap (test.py:62,104 divides by Concorde's optimum, which the LFS-stub instance files do not hold):
        # bracketed by the best-known tour (upper bound of the optimum -> lower bound of the gap) and the Held-Karp 1-tree
        # bound (oracle/one_tree.c, checker side, host cores, outside the timed region; first <= 1024 instances of rank 0).
        bracket = None
        if gap is not None and not args.exact_gap and not args.no_gap_bracket and n > 20:
            from oracle import one_tree
            m = min(B, 1024)
            lb = one_tree.lower_bounds(D_host[:m], g[:m, 0], workers=available_cores())
            # a lower bound above a known tour length would be a checker bug: reported in the line, never a reason to discard the run
            bad_bound = int((lb > bk[:m] * (1 + 1e-9)).sum())
            bracket = {"bound_above_known_tour_instances": bad_bound,
                       "vs_best_known_pct": float(gap[:m].mean()), "vs_lower_bound_pct": float(((g[:m, 0] / lb - 1.0) * 100.0).mean()),
                       "instances": int(m), "best_known_above_lower_bound_pct": float(((bk[:m] / lb - 1.0) * 100.0).mean()),
                       "how": "mean over the instances of (best_cost / x - 1) * 100 with x = best-known tour length (>= optimum) "
                              "and x = Held-Karp 1-tree lower bound (<= optimum, subgradient ascent, oracle/one_tree.c)"}
        # ... and exactly, on the sample of block 0 whose optima are PROVEN (bench_data/exact_optima_*.npz: branch and bound on the
        # 1-tree bound, oracle/bnb_tsp.c, made in the build container -- data only): the reference's own definition of the gap
        exact = None
        if gap is not None and not args.exact_gap and lo == 0:
            exact = exact_sample_gap(n, args.seed, g[:B, 0], bk[:B])
        search_s = last.timing["search_s"]
        kern = kernel_rooflines(prof, n, min(chunk_eff, B), n_layers) if need_model else {}
        fwd_ms = sum(v["total_ms"] for v in kern.values())
        # dominant kernel of the timed step: the search kernel (by construction it runs for the whole budget).  One launch
        # per round; duration from the HIP events recorded around the launch on its stream.  Rank 0's launches, last step's
        # evaluation counts (reference-equivalent: evals_out; executed: the measurement hook).
        resident = min(chunk_eff, B)
        # counters collected on this very workload if there are any (profiles/traffic_r0*.json: `gls_kernel@tsp<n>x<B>`), else the
        # headline's -- search_roofline() then withholds the fractions (pmc_matches_workload false)
        traffic = load_traffic()
        traffic = traffic.get("gls_kernel@tsp%dx%d" % (n, resident), traffic.get("gls_kernel", {}))
        ratio = exec_ratio
        roof = search_roofline(n, args.steps, gls_ms, gls_launches, per_rank[0][4].item(), ratio if ratio >= 0 else None, resident, traffic,
                               {"n": n, "instances": resident, "guide": "model" if args.guides == ["regret_pred"] else "+".join(args.guides)}, cyc)
        roof["device_time_share"] = gls_ms / (gls_ms + fwd_ms) if gls_ms + fwd_ms > 0 else None
        # gap-versus-budget (test.py:97-117): all ranks' sums; the last point is the end of the budget
        sums = sum(p[8:].cpu().numpy().reshape(-1, 3) for p in per_rank)
        pre_search = sum(p[6].item() for p in per_rank) / max(total, 1)
        curve, curve_note = None, None
        if not args.exact_gap and sums[0, 2] == total:
            end_s = max(args.time_limit / (rounds if args.budget == "per_batch" and rounds else 1) - pre_search, 0.0)
            curve = [p for p in gap_curve(sums, list(GAP_GRID_S) + [end_s], pre_search) if p["t_s"] < end_s or p["t_s"] == end_s]
            if args.budget == "per_instance" and abs(curve[-1]["mean_gap_pct"] - float(gap.mean())) > 1e-9 * max(1.0, abs(float(gap.mean()))):
                # the end point IS the headline gap (same instances, same run); an incomplete improvement record (aborted instance)
                # breaks that: say so in the line instead of discarding the measurement
                curve_note = "end point of the record differs from mean_gap_pct (incomplete improvement trace of an instance)"
        if grouped:
            par = f"instance-sharded x{world}, one gather ({'RCCL' if backend == 'nccl' else backend}, world_size {dist.get_world_size()})"
        else:
            par = "instance-sharded x1, no process group (single rank: the gather is the identity)"
        out = {
            "metric": f"TSP instances/sec + mean opt-gap @{args.time_limit:g}s, TSP{n}", "value": value, "unit": "instances/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64 (search) / f32 (GNN)" if os.environ.get("GNNGLS_FFN_FP32", "0") not in ("", "0") else
                     "f64 (search) / f32 (GNN; its feed-forward block as three bf16 pieces per f32 operand on the bf16 MFMA, f32 accumulate, f32-equivalent results)",
            "data": "synthetic",
            "config": {"workload": (f"TSP{n}, fixed test set of {total} instances sharded over {world} GPU(s), GNN forward + "
                                    f"guided_local_search {args.time_limit:g} s budget "
                                    + ("per instance" if args.budget == "per_instance" else "per device-load sequence (rounds share it)")
                                    + (" (BASELINE.json configs[3])" if (n, total) == (100, 10000) else "")) if strong else
                                   (f"TSP{n}, batch of {args.batch} instances per GPU, GNN forward + guided_local_search "
                                    f"{args.time_limit:g} s budget" + (" (BASELINE.json configs[2])" if (n, args.batch) == (100, 1024) else "")),
                       "n": n, "instances_per_gpu": B, "total_instances": total, "resident_instances_per_gpu": chunk,
                       "rounds_per_rank": [int(p[2].item()) for p in per_rank],
                       "time_limit_s": args.time_limit, "budget": args.budget, "perturbation_moves": args.perturbation_moves,
                       "guides": args.guides, "parallelism": par,
                       "backend": (("RCCL (nccl)" if backend == "nccl" else backend) if grouped else None), "world_size": world,
                       "device_of_rank": [int(p[7].item()) for p in per_rank], "visible_devices": n_dev,
                       "collectives_per_step": {k: v / max(args.steps, 1) for k, v in collectives.items()} if grouped else {},
                       # strong scaling: instances a rank searches / (rounds x device residency) -- configs[3] on 8 GPUs is
                       # 1250 / (2 x 1024) = 0.61 by arithmetic ("10 s per instance" makes a partly filled round cost a full one)
                       "residency_utilisation": [float(sz) / (max(int(p[2].item()), 1) * max(chunk, 1)) for sz, p in
                                                 zip(sizes if sizes is not None else [B] * world, per_rank)]},
            "mean_gap_pct": float(gap.mean()) if gap is not None else None, "gap_reference": gap_reference,
            "true_gap_bracket_pct": [bracket["vs_best_known_pct"], bracket["vs_lower_bound_pct"]] if bracket else None,
            "true_gap_bracket": bracket, "true_gap_exact_sample": exact,
            "max_gap_pct": float(gap.max()) if gap is not None else None,
            "instances_at_reference_pct": float((np.abs(gap) <= 1e-9).mean() * 100.0) if gap is not None else None,
            "instances_below_reference": int((gap < -1e-9).sum()) if gap is not None else None,
            "mean_best_cost": float(g[:, 0].mean()), "mean_init_cost": float(g[:, 1].mean()),
            "outer_iters_per_instance": float(g[:, 2].mean()),
            "watchdog_aborts": int((g[:, 4] == ops.STATUS_WATCHDOG).sum()),
            "penalty_overflows": int((g[:, 4] == ops.STATUS_PENALTY_OVERFLOW).sum()),
            "forward_s_per_step": last.timing["forward_s"], "search_s_per_step": search_s,
            # search-progress record of the SAME timed step (zero extra device time): mean gap / share of instances at the
            # best-known length after t_s seconds of search (the budget clock of test.py:64 also counts the forward pass:
            # budget_t_s); the last point is the end of the budget = mean_gap_pct
            "gap_vs_budget": curve, "gap_vs_budget_note": curve_note if curve else (curve_reason if bk_local is None else "incomplete"),
            "pre_search_s": pre_search, "improvement_trace_truncated_instances": int(sum(p[5].item() for p in per_rank)),
            # the kernel that owns the timed step (98 % of device time)
            "roofline": roof,
            "gls_ms_per_rank": [p[0].item() for p in per_rank],
            "kernels": kern, "forward_kernels_ms_total": fwd_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            cores = max(1, min(args.cpu_cores or available_cores(), B))
            Ds = D[:cores].contiguous()
            Rs = pipeline.predict_regret(model, Ds, scalers) if need_model else None
            guides_host = torch.stack([Rs if gname == "regret_pred" else Ds for gname in args.guides]).cpu().numpy()
            init = ops.nearest_neighbor(Rs if need_model else Ds)                  # test.py:70-88
            init_cost = ops.tour_cost(init, Ds)
            out["cpu_baseline"] = cpu_baseline(D_host[:cores], guides_host, init.cpu().numpy(), init_cost.cpu().numpy(),
                                               bk, args.time_limit, args.perturbation_moves, cores)
            wb = out["cpu_baseline"]["whole_box_estimate"]
            wb["gpu_over_whole_box"] = value / wb["instances_per_s"] if wb["instances_per_s"] > 0 else None
        cpu_raw = out.get("cpu_baseline", {}).pop("_raw_traces", None) if "cpu_baseline" in out else None
        if world == 1 and not args.no_iso_quality and not strong and not args.exact_gap:
            trace = None
            if bk_local is not None and B > 0 and last.imp_cost is not None:
                m = min(resident, B)
                trace = {"imp_cost": last.imp_cost[:m].cpu().numpy(), "imp_time": last.imp_time[:m].cpu().numpy(),
                         "imp_len": last.imp_len[:m].cpu().numpy(), "init_cost": last.init_cost[:m].cpu().numpy(),
                         "best_known": bk_local[:m], "pre_search_s": pre_search}
            out["iso_quality"] = iso_quality_pass(args, n, resident, model, scalers, pipeline, trace, cpu_raw)
        print(json.dumps(out))
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
