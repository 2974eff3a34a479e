#!/usr/bin/env python
"""Quick GLS throughput probe on the GPU box: outer iterations and delta evaluations per second."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from gnngls_amd import ops  # noqa: E402
from gnngls_amd.synthetic import random_instances  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
Bs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 512, 1024]
tl = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
bits = int(sys.argv[4]) if len(sys.argv) > 4 else 0              # penalty_bits: 0 auto, -2 compact store, 32 / 16 LDS counters
guide_kind = sys.argv[5] if len(sys.argv) > 5 else "weight"      # weight | noise (clamped fp32 noise) | model (regret_pred of the synthetic model, what bench.py uses)
threads = int(sys.argv[6]) if len(sys.argv) > 6 else 0           # workgroup size override (0 = default policy)
team = int(sys.argv[7]) if len(sys.argv) > 7 else -1             # perturbation phase: -1 policy, 0 wavefront 0, 1 all wavefronts
from gnngls_amd import _lib  # noqa: E402
_lib.check(_lib.load().gnngls_debug_set_gls_threads(threads))
_lib.check(_lib.load().gnngls_debug_set_gls_team(team))
import os  # noqa: E402
_lib.check(_lib.load().gnngls_debug_set_gls_prune(int(os.environ.get("PRUNE", "-1"))))
print("capacity", ops.gls_resident_capacity(n), ops.gls_describe_config(n, Bs[0], bits))
for B in Bs:
    D = torch.from_numpy(random_instances(np.random.default_rng(0), B, n)[0]).cuda()
    init = ops.nearest_neighbor(D)
    cost = ops.tour_cost(init, D)
    g = D[None].contiguous()
    if guide_kind == "noise":
        rng = np.random.default_rng(1)
        x = np.maximum(rng.normal(0.05, 0.1, size=(B, n, n)).astype(np.float32).astype(np.float64), 0)
        x = np.triu(x, 1)
        g = torch.from_numpy((x + x.transpose(0, 2, 1))[None]).cuda().contiguous()
        init = ops.nearest_neighbor(g[0])
        cost = ops.tour_cost(init, D)
    if guide_kind == "model":                                 # bench.py's guide: regret_pred of the synthetic model
        from gnngls_amd import pipeline
        R = pipeline.predict_regret(pipeline.synthetic_model(seed=1234), D, pipeline.Scalers.fit_weights(D))
        g = R[None].contiguous()
        init = ops.nearest_neighbor(R)
        cost = ops.tour_cost(init, D)
    torch.cuda.synchronize()
    t0 = time.time()
    r = ops.gls_run(D, g, init, cost, perturbation_moves=20, max_outer_iters=-1, time_limit_s=tl, penalty_bits=bits)
    torch.cuda.synchronize()
    dt = time.time() - t0
    it = r.outer_iters.double()
    print(f"n={n} B={B} bits={bits} thr={threads} team={team} guide={guide_kind} wall={dt:.2f}s outer_iters mean={it.mean():.0f} min={it.min():.0f} "
          f"evals/s={r.evals.sum().item() / dt:.3e} moves mean={r.trace_len.double().mean():.0f} "
          f"init={cost.mean():.4f} best={r.best_cost.mean():.4f} status={r.status.sum().item()}")
