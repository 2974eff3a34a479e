"""Diagnostic: the per-batch budget policy of pipeline.solve_batch.  R device-loads of TSP100 instances searched within ONE
time limit; prints throughput and the gap of the first 1024 instances against their converged tours (the 60 s run of
profiles/r01h_bench_weight_guide_60s.json uses the same generator and seed).
usage: python scripts/probe_budget.py R [time_limit]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnngls_amd import ops, pipeline  # noqa: E402
from gnngls_amd.synthetic import random_instances  # noqa: E402

R = int(sys.argv[1])
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
n, cap = 100, ops.gls_resident_capacity(100)
D_host, _ = random_instances(np.random.default_rng(2024), cap * R, n)
D = torch.from_numpy(D_host).cuda()
ref = pipeline.solve_batch(D[:cap].contiguous(), guides=("weight",), time_limit=limit)           # full budget each
torch.cuda.synchronize()
t0 = time.time()
res = pipeline.solve_batch(D, guides=("weight",), time_limit=limit, budget="per_batch")
torch.cuda.synchronize()
dt = time.time() - t0
gap = (res.best_cost[:cap] / ref.best_cost - 1.0) * 100.0
print(f"R={R}: {cap * R} instances in {dt:.2f} s = {cap * R / dt:.1f} instances/s; outer iterations per instance "
      f"{res.outer_iters.float().mean().item():.0f} (full budget: {ref.outer_iters.float().mean().item():.0f}); gap of the "
      f"first {cap} vs their full-budget tours: mean {gap.mean().item():.4f} % max {gap.max().item():.3f} % "
      f"identical {100.0 * (gap.abs() < 1e-9).float().mean().item():.1f} %")
