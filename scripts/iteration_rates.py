#!/usr/bin/env python
"""Outer-iteration rates of the search kernel on the BASELINE.json shapes (1 s of search, regret_pred guide of the synthetic
model) -> profiles/r06_iteration_rates.json, the rates tests/test_perf_floor_gpu.py asserts 90 % of (median of five runs).

    python scripts/iteration_rates.py profiles/r06_iteration_rates.json"""
import json
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnngls_amd import ops, pipeline  # noqa: E402
from gnngls_amd.synthetic import random_instances  # noqa: E402
out = {"how": "python scripts/iteration_rates.py on an MI355X: mean outer iterations per instance in 1 s of gls_run, regret_pred guide of "
              "pipeline.synthetic_model(seed=1234), instances random_instances(default_rng(0)); MEDIAN of 5 (all five under `runs`)", "shapes": {}}
for n, B in ((100, 1024), (200, 256), (50, 128), (20, 1000)):
    D = torch.from_numpy(random_instances(np.random.default_rng(0), B, n)[0]).cuda()
    R = pipeline.predict_regret(pipeline.synthetic_model(seed=1234), D, pipeline.Scalers.fit_weights(D))
    init = ops.nearest_neighbor(R); cost = ops.tour_cost(init, D); g = R[None].contiguous()
    runs = []
    for _ in range(5):
        torch.cuda.synchronize()
        r = ops.gls_run(D, g, init, cost, perturbation_moves=20, max_outer_iters=-1, time_limit_s=1.0)
        torch.cuda.synchronize()
        runs.append(float(r.outer_iters.double().mean()))
    best = float(np.median(runs))
    run = ops.gls_describe_run(n, B)
    out["shapes"][f"tsp{n}x{B}"] = {"n": n, "instances": B, "outer_iters_per_s": best, "runs": runs, "config": run, "resources": ops.gls_kernel_resources(n, B)}
    print(n, B, best, runs, run, out["shapes"][f"tsp{n}x{B}"]["resources"])
json.dump(out, open(os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "profiles/r06_iteration_rates.json"), "w"), indent=1)
