#!/usr/bin/env python
"""GLS with a FIXED number of outer iterations (same trajectory on every build: the search is bit-exact), for dynamic
instruction counts under rocprofv3 --pmc: the difference between two builds is the difference of their code paths.
    python scripts/probe_gls_fixed.py n B iters guide"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from gnngls_amd import ops  # noqa: E402
from gnngls_amd.synthetic import random_instances  # noqa: E402
n, B, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
guide_kind = sys.argv[4] if len(sys.argv) > 4 else "model"
D = torch.from_numpy(random_instances(np.random.default_rng(0), B, n)[0]).cuda()
init = ops.nearest_neighbor(D); cost = ops.tour_cost(init, D); g = D[None].contiguous()
if guide_kind == "model":
    from gnngls_amd import pipeline
    R = pipeline.predict_regret(pipeline.synthetic_model(seed=1234), D, pipeline.Scalers.fit_weights(D))
    g = R[None].contiguous(); init = ops.nearest_neighbor(R); cost = ops.tour_cost(init, D)
torch.cuda.synchronize()
t0 = time.time()
r = ops.gls_run(D, g, init, cost, perturbation_moves=20, max_outer_iters=K)
torch.cuda.synchronize()
print(f"n={n} B={B} K={K} guide={guide_kind} wall={time.time() - t0:.3f}s moves mean={r.trace_len.double().mean():.1f} "
      f"best={r.best_cost.mean():.6f} evals={r.evals.sum().item():.4e}")
