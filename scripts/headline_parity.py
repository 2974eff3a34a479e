#!/usr/bin/env python
"""Parity of the headline run itself: the TSP100 x 1024 workload of bench.py (classical `weight` guide so that no GNN
rounding enters) searched for the full time limit on the GPU; for a sample of instances the CPU oracle then runs exactly
the number of outer iterations the GPU completed for that instance and must return the same best tour and cost bit for bit.

    python scripts/headline_parity.py [--time_limit 10] [--sample 16]
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def oracle_run(job):
    from oracle import gls_oracle as go
    D, G, init, cost, iters = job
    o = go.guided_local_search(D, G[None], init, cost, perturbation_moves=20, max_outer_iters=int(iters), trace_cap=1,
                               want_penalty=False)
    return o["best_tour"], o["best_cost"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time_limit", type=float, default=10.0)
    ap.add_argument("--sample", type=int, default=16)
    ap.add_argument("--n", "--tsp_n", dest="n", type=int, default=100)
    ap.add_argument("--batch", type=int, default=1024, help="TSP200: 256 (one 16-wave workgroup per CU, team form of the perturbation phase)")
    ap.add_argument("--guide", choices=["weight", "regret_pred"], default="weight",
                    help="regret_pred: the guide matrix is the GPU forward of the synthetic model; the oracle gets the same matrix")
    args = ap.parse_args()
    from gnngls_amd import ops
    from gnngls_amd.synthetic import random_instances
    from oracle import gls_oracle as go
    go.build()
    n, B = args.n, args.batch
    D_host, _ = random_instances(np.random.default_rng(2024), B, n)
    D = torch.from_numpy(D_host).cuda()
    G = D
    if args.guide == "regret_pred":
        from gnngls_amd import pipeline
        G = pipeline.predict_regret(pipeline.synthetic_model(), D, pipeline.Scalers.fit_weights(D))
    G_host = G.cpu().numpy()
    init = ops.nearest_neighbor(G)                                            # test.py:85
    cost = ops.tour_cost(init, D)
    r = ops.gls_run(D, G[None].contiguous(), init, cost, perturbation_moves=20, max_outer_iters=-1, time_limit_s=args.time_limit)
    torch.cuda.synchronize()
    iters = r.outer_iters.cpu().numpy()
    pick = np.random.default_rng(1).choice(B, size=args.sample, replace=False)
    init_h, cost_h = init.cpu().numpy(), cost.cpu().numpy()
    jobs = [(D_host[b], G_host[b], init_h[b], float(cost_h[b]), iters[b]) for b in pick]
    t0 = time.time()
    with mp.get_context("spawn").Pool(min(args.sample, os.cpu_count() or 1)) as pool:
        res = pool.map(oracle_run, jobs)
    bad = 0
    for b, (tour, c) in zip(pick, res):
        same = (r.best_tour[b].cpu().tolist() == tour
                and np.float64(r.best_cost[b].item()).view(np.uint64) == np.float64(c).view(np.uint64))
        bad += not same
    print(f"TSP{n} x {B} ({ops.gls_describe_config(n, B)}), guide {args.guide}, {args.time_limit:g} s on the GPU: {iters.mean():.0f} outer iterations per instance; {args.sample} sampled "
          f"instances re-run on the CPU oracle for their own iteration counts ({iters[pick].min()}..{iters[pick].max()}) in "
          f"{time.time() - t0:.0f} s: {args.sample - bad} identical best tours and costs, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
