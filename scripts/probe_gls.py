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
print("capacity", ops.gls_resident_capacity(n))
for B in Bs:
    D = torch.from_numpy(random_instances(np.random.default_rng(0), B, n)[0]).cuda()
    init = ops.nearest_neighbor(D)
    cost = ops.tour_cost(init, D)
    g = D[None].contiguous()
    torch.cuda.synchronize()
    t0 = time.time()
    r = ops.gls_run(D, g, init, cost, perturbation_moves=20, max_outer_iters=-1, time_limit_s=tl)
    torch.cuda.synchronize()
    dt = time.time() - t0
    it = r.outer_iters.double()
    print(f"n={n} B={B} wall={dt:.2f}s outer_iters mean={it.mean():.0f} min={it.min():.0f} "
          f"evals/s={r.evals.sum().item() / dt:.3e} moves mean={r.trace_len.double().mean():.0f} "
          f"init={cost.mean():.4f} best={r.best_cost.mean():.4f} status={r.status.sum().item()}")
