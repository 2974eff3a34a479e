#!/usr/bin/env python
"""Measurement infrastructure (not product code): builds the best-known tour lengths that bench.py uses as the
denominator of the optimality gap -- the stand-in for the Concorde optimum the reference reads from its instance
files (gnngls/__init__.py:55-60, scripts/test.py:62,104; those files are git-LFS stubs here).

    python scripts/make_best_known.py --n 100 --blocks 10 --seconds 60 --out gpurun_out/best_known_tsp100_seed2024.npz

For every seeded instance block of bench.py (block k = synthetic.random_instances(default_rng(seed + 1000 k), 1024, n))
the value kept per instance is the MINIMUM tour length over independent long searches:
  * GPU guided local search, `seconds` per instance, guide 'regret_pred' (synthetic model) -- perturbation_moves 20
  * GPU guided local search, `seconds` per instance, guide 'weight' (classical GLS)       -- perturbation_moves 20
  * block 0 only: guide 'weight' with perturbation_moves 30 (the library default, algorithms.py:135) and the alternating
    guide list ['weight', 'regret_pred']
  * block 0, first `oracle_instances` instances: the CPU oracle (oracle/gls_oracle.c) for `seconds`, one instance per host
    core, guide 'weight'
GPU tour lengths are recomputed with tour_cost (left-to-right fp64 sum) from the returned tours, never taken from the
searches' own bookkeeping; the CPU-oracle leg contributes its returned best_cost (the reference's incremental sum,
algorithms.py:124,190 -- the same tour can differ from tour_cost by an ulp, which is why that leg "wins" ties at the
1e-15 level).  The file records how every block was made (`how`, `log`, `winner*`); bench.py only ever reads it.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BLOCK = 1024


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--count", type=int, default=BLOCK, help="instances per block to cover (the rest stays NaN)")
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--oracle_instances", type=int, default=-1, help="-1 = one per host core")
    ap.add_argument("--out", required=True)
    ap.add_argument("--start_block", type=int, default=0, help="first block to compute (with --merge: the earlier blocks of that file are kept)")
    ap.add_argument("--merge", default=None, help="existing best-known file of the same (n, seed): its blocks are kept, new blocks are "
                                                  "added, a block present in both keeps the elementwise minimum")
    args = ap.parse_args()

    from gnngls_amd import ops, pipeline
    from gnngls_amd.synthetic import random_instances
    n = args.n
    model = pipeline.synthetic_model(seed=1234)
    scalers = None
    out, log = {}, []
    t_all = time.time()
    if scalers is None:                                                                   # as bench.py: fitted on block 0
        scalers = pipeline.Scalers.fit_weights(torch.from_numpy(random_instances(np.random.default_rng(args.seed), BLOCK, n)[0]).cuda())
    for k in range(args.start_block, args.blocks):
        D_host, _ = random_instances(np.random.default_rng(args.seed + 1000 * k), BLOCK, n)
        D = torch.from_numpy(D_host[:args.count]).cuda()
        runs = [(("regret_pred",), 20), (("weight",), 20)]
        if k == 0:
            runs += [(("weight",), 30), (("weight", "regret_pred"), 20)]
        best = np.full(BLOCK, np.nan)
        winners = np.full(BLOCK, -1, dtype=np.int32)
        for ri, (guides, pm) in enumerate(runs):
            t0 = time.time()
            r = pipeline.solve_batch(D, model, scalers, guides=guides, time_limit=args.seconds, perturbation_moves=pm)
            c = ops.tour_cost(r.best_tour, D).cpu().numpy()
            assert (r.status.cpu().numpy() == 0).all()
            better = np.isnan(best[:args.count]) | (c < best[:args.count])
            best[:args.count] = np.where(better, c, best[:args.count])
            winners[:args.count][better] = ri
            log.append({"block": k, "guides": list(guides), "perturbation_moves": pm, "seconds": args.seconds,
                        "mean": float(c.mean()), "outer_iters": float(r.outer_iters.double().mean()), "wall_s": time.time() - t0,
                        "new_best": int(better.sum())})
            print(json.dumps(log[-1]), flush=True)
        if k == 0 and args.oracle_instances != 0:
            m = min(args.oracle_instances if args.oracle_instances > 0 else (os.cpu_count() or 1), args.count)   # NB: beyond the
            # container's CPU quota the workers only time-slice (the committed TSP100 file: 256 workers on 16 CPUs)
            init = ops.nearest_neighbor(D[:m].contiguous())
            init_cost = ops.tour_cost(init, D[:m].contiguous())
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "sample.npz")
                np.savez(path, D=D_host[:m], guides=D_host[None, :m], init_tour=init.cpu().numpy(), init_cost=init_cost.cpu().numpy())
                procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline_worker.py"), path, str(i),
                                           str(args.seconds), "20"], stdout=subprocess.PIPE, cwd=ROOT) for i in range(m)]
                res = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in procs]
            c = np.array([x["best_cost"] for x in res])
            better = c < best[:m]
            best[:m] = np.where(better, c, best[:m])
            winners[:m][better] = 100
            log.append({"block": 0, "cpu_oracle_instances": m, "seconds": args.seconds, "mean": float(c.mean()),
                        "outer_iters": float(np.mean([x["outer_iters"] for x in res])), "new_best": int(better.sum())})
            print(json.dumps(log[-1]), flush=True)
        out[f"block{k}"] = best
        out[f"winner{k}"] = winners
    how = (f"min over GPU guided_local_search runs of {args.seconds:g} s per instance with guides regret_pred and weight "
           f"(perturbation_moves 20; block 0 also weight/30, [weight,regret_pred]/20 and a {args.seconds:g} s CPU-oracle run on "
           f"one instance per host core), lengths recomputed by tour_cost; scripts/make_best_known.py")
    if args.merge:
        z = np.load(args.merge, allow_pickle=False)
        assert int(z["n"]) == n and int(z["seed"]) == args.seed, "--merge: another instance set"
        for key in z.files:
            if key.startswith("block"):
                out[key] = np.fmin(z[key], out[key]) if key in out else z[key]           # fmin: NaN = not covered
            elif key.startswith("winner") and key not in out:
                out[key] = z[key]
        log = json.loads(str(z["log"])) + log
        how = str(z["how"]) + f"; blocks {args.start_block}..{args.blocks - 1} (first {args.count} instances each) added later the same way"
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, n=n, seed=args.seed, how=np.array(how), log=np.array(json.dumps(log)), **out)
    print(f"wrote {args.out} ({args.blocks} blocks, {time.time() - t_all:.0f} s)")


if __name__ == "__main__":
    main()
