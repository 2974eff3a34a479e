#!/usr/bin/env python
"""Continue bench_data/exact_optima_tsp{n}_seed{seed}.npz on its UNPROVEN instances with a larger node limit (checker side; CPU
only; build container): every finished instance is merged into the file at once, so the run can be stopped at any time.

    python scripts/extend_exact_optima.py --n 200 --max_nodes 450000 --workers 4 [--first 12]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(args):
    i, D, ub, max_nodes = args
    from oracle import bnb_tsp
    t0 = time.time()
    r = bnb_tsp.solve(D, ub, max_nodes)
    return i, r["value"], r["proven"], r["nodes"], time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--max_nodes", type=int, default=450000)
    ap.add_argument("--first", type=int, default=0, help="only the first K unproven instances (0 = all)")
    a = ap.parse_args()
    from gnngls_amd.synthetic import random_instances
    from oracle import bnb_tsp
    bnb_tsp.lib()
    path = os.path.join(ROOT, "bench_data", f"exact_optima_tsp{a.n}_seed{a.seed}.npz")
    cur = dict(np.load(path))
    D, _ = random_instances(np.random.default_rng(a.seed), 1024, a.n)
    todo = [int(i) for i in cur["index"][~cur["proven"]]]
    if a.first:
        todo = todo[:a.first]
    jobs = [(i, D[i], float(min(cur["best_known"][i], cur["optimum"][i])), a.max_nodes) for i in todo]
    print(f"{len(jobs)} unproven instances, node limit {a.max_nodes}: {todo}", flush=True)
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(a.workers) as pool:
        for i, val, proven, nodes, secs in pool.imap_unordered(one, jobs, chunksize=1):
            k = int(np.nonzero(cur["index"] == i)[0][0])
            cur["optimum"][k] = min(val, cur["optimum"][k]); cur["proven"][k] = bool(proven); cur["nodes"][k] = nodes; cur["seconds"][k] = secs
            cur["how"] = np.array(str(cur["how"]).split("; extended")[0] + f"; extended on the unproven instances with max_nodes={a.max_nodes} (scripts/extend_exact_optima.py)")
            np.savez(path, **cur)
            print(f"instance {i}: proven {bool(proven)}, value {val:.6f} (best known {cur['best_known'][k]:.6f}), {nodes} subproblems, {secs:.0f} s; "
                  f"proven so far {int(cur['proven'].sum())} of {len(cur['proven'])}", flush=True)


if __name__ == "__main__":
    main()
