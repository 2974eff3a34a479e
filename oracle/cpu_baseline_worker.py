#!/usr/bin/env python
"""oracle/cpu_baseline_worker.py -- TEST/BENCH INFRASTRUCTURE (bench.py's cpu_baseline leg only).

Runs the CPU oracle's guided_local_search (oracle/gls_oracle.c, wall-clock mode, reference semantics
algorithms.py:135-195) on ONE instance of a sample file and prints one JSON line.  Started as a child
process per host core by bench.py; never imported by the product path."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gls_oracle as go  # noqa: E402

path, idx, time_limit, pm = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
import time  # noqa: E402

z = np.load(path)
D, guides, init_tour, init_cost = z["D"][idx], z["guides"][:, idx], z["init_tour"][idx], float(z["init_cost"][idx])
go.lib()
t0 = time.time()
r = go.guided_local_search(D, guides, init_tour, init_cost,
                           perturbation_moves=pm, max_outer_iters=-1, time_limit_s=time_limit, trace_cap=1,
                           want_penalty=False)
print(json.dumps({"best_cost": r["best_cost"], "outer_iters": r["outer_iters"], "evals": r["evals"],
                  "moves": r["trace_len"], "search_s": time.time() - t0,
                  # search-progress record (the returned best at every improvement + the terminal entry, test.py:97-117)
                  "imp_cost": r["imp_cost"].tolist(), "imp_time": r["imp_time"].tolist(), "imp_len": r["imp_len"]}))
