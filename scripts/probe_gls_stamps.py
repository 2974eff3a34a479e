#!/usr/bin/env python
"""Diagnostic (library built with GNNGLS_EXTRA_FLAGS=-DGLS_STAMPS): where the search kernel spends its cycles."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from gnngls_amd import ops
from gnngls_amd.synthetic import random_instances
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
D = torch.from_numpy(random_instances(np.random.default_rng(0), B, n)[0]).cuda()
init = ops.nearest_neighbor(D); cost = ops.tour_cost(init, D)
# trace_time buffer doubles as the stamp sink: needs trace_cap >= 16 floats (= 8 int64) per instance; trace_cost must exist
r = ops.gls_run(D, D[None].contiguous(), init, cost, perturbation_moves=20, max_outer_iters=-1, time_limit_s=1.0,
                trace_cap=16, want_trace_time=True)
torch.cuda.synchronize()
st = r.trace_time.view(torch.int64).double().mean(0).cpu().numpy()
names = ["utility argmax", "o2a scan (+pen, pos search)", "o2a reduce", "apply+reload", "phase tail", "descent (LS)", "steps"]
tot = st[:6].sum()
it = r.outer_iters.double().mean().item()
print(f"outer iters {it:.0f}, perturbation steps/iter {st[6] / it:.1f}")
for k in range(6):
    print(f"{names[k]:30s} {st[k] / tot * 100:5.1f}%   {st[k] / it:9.0f} cycles/outer-iter")
print(f"cycles per perturbation step (wave 0): {(st[0] + st[1] + st[2] + st[3]) / st[6]:.0f}")
