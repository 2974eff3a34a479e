#!/usr/bin/env python
"""Exact optima for a sample of the seeded synthetic instances (checker side; CPU only; runs in the build container):

    python scripts/make_exact_optima.py --n 100 --count 96 --workers 7 --max_nodes 400000

For instances 0 .. count-1 of block 0 (gnngls_amd.synthetic.random_instances(default_rng(seed)), what bench.py searches):
branch and bound on the Held-Karp 1-tree bound (oracle/bnb_tsp.c) with the best-known tour length as the incumbent.  A
completed search PROVES that length optimal (or returns the shorter tour it found).  Writes
bench_data/exact_optima_tsp{n}_seed{seed}.npz: index, optimum, proven, nodes, seconds, best_known -- data only.  The reference's
gap divides by the Concorde optimum of its instance files (scripts/test.py:62,104), which are git-LFS stubs here."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(args):
    D, ub, max_nodes = args
    from oracle import bnb_tsp
    t0 = time.time()
    r = bnb_tsp.solve(D, ub, max_nodes)
    return r["value"], r["proven"], r["nodes"], time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--count", type=int, default=96)
    ap.add_argument("--workers", type=int, default=7)
    ap.add_argument("--max_nodes", type=int, default=400000)
    a = ap.parse_args()
    from gnngls_amd.synthetic import random_instances
    from oracle import bnb_tsp
    bnb_tsp.lib()
    D, _ = random_instances(np.random.default_rng(a.seed), 1024, a.n)
    bk = np.load(os.path.join(ROOT, "bench_data", f"best_known_tsp{a.n}_seed{a.seed}.npz"))["block0"]
    jobs = [(D[i], float(bk[i]), a.max_nodes) for i in range(a.count)]
    import multiprocessing as mp
    t0 = time.time()
    with mp.get_context("spawn").Pool(a.workers) as pool:
        res = pool.map(one, jobs, chunksize=1)
    val = np.array([r[0] for r in res]); proven = np.array([r[1] for r in res]); nodes = np.array([r[2] for r in res])
    secs = np.array([r[3] for r in res])
    out = os.path.join(ROOT, "bench_data", f"exact_optima_tsp{a.n}_seed{a.seed}.npz")
    np.savez(out, n=a.n, seed=a.seed, index=np.arange(a.count), optimum=val, proven=proven, nodes=nodes, seconds=secs,
             best_known=bk[:a.count],
             how=f"oracle/bnb_tsp.c (branch and bound on the Held-Karp 1-tree bound, incumbent = best-known length), "
                 f"max_nodes={a.max_nodes}; scripts/make_exact_optima.py")
    better = int((val < bk[:a.count] * (1 - 1e-12)).sum())
    print(f"{a.count} instances in {time.time() - t0:.0f} s wall: proven {int(proven.sum())}, best-known improved on {better}, "
          f"nodes median {int(np.median(nodes))} max {int(nodes.max())}, seconds median {np.median(secs):.1f} max {secs.max():.1f}")
    print("unproven:", np.nonzero(~proven)[0].tolist())


if __name__ == "__main__":
    main()
