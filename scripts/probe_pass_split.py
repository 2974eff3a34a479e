#!/usr/bin/env python
"""Diagnostic used for profiles/r02_stamps_guided_pass_split.log: reads the stamp slots 7 / 12..15 of a library built with
-DGLS_STAMPS AND with clock reads placed inside scan_two_opt_o2a_guided (after the tour reads, after the loads are issued,
after `s_waitcnt vmcnt(0) lgkmcnt(0)`, after `consider`), accumulated into those slots.  That instrumentation is not kept in
the tree (it perturbs the loop it measures); the log describes it.  With the plain -DGLS_STAMPS build the slots hold the
per-wave descent figures instead (scripts/probe_gls_stamps.py)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from gnngls_amd import ops, _lib
from gnngls_amd.synthetic import random_instances
n, B = 100, 1024
D = torch.from_numpy(random_instances(np.random.default_rng(0), B, n)[0]).cuda()
init = ops.nearest_neighbor(D); cost = ops.tour_cost(init, D)
stamps = torch.zeros((B, 16), dtype=torch.int64, device="cuda")
_lib.check(_lib.load().gnngls_debug_set_stamp_buffer(_lib.ptr(stamps)))
r = ops.gls_run(D, D[None].contiguous(), init, cost, penalty_bits=0, perturbation_moves=20, max_outer_iters=-1, time_limit_s=1.0, trace_cap=0)
torch.cuda.synchronize()
st = stamps.double().mean(0).cpu().numpy()
p = max(st[7], 1)
print(f"2-opt guided passes per instance {st[7]:.0f}; cycles per pass: tour reads {st[12]/p:.0f}, index + issue {st[13]/p:.0f}, "
      f"wait for penalties/distances {st[14]/p:.0f}, arithmetic + consider {st[15]/p:.0f}; sum {(st[12]+st[13]+st[14]+st[15])/p:.0f}")
print(f"o2a scan total per step {st[1]/max(st[6],1):.0f} cycles; steps {st[6]:.0f}")
