#!/usr/bin/env python
"""Extended randomised parity campaign of the search kernel against the CPU oracle (the generator of
tests/test_gls_fuzz_gpu.py with a different master seed, more cases and longer runs).  Every accepted move, the best
tour, its cost and the final penalties must match bit for bit.

    python scripts/fuzz_campaign.py [--cases 400] [--seed 1] [--max_k 30] [--min_n 4] [--max_n 130]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gnngls_amd import ops  # noqa: E402
from oracle import gls_oracle as go  # noqa: E402
from test_gls_fuzz_gpu import make_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max_k", type=int, default=30)
    ap.add_argument("--min_n", type=int, default=4)
    ap.add_argument("--max_n", type=int, default=130)
    args = ap.parse_args()
    master = np.random.default_rng(args.seed)
    bad, moves, teamed, t0 = [], 0, 0, time.time()
    for ci in range(args.cases):
        c = dict(n=int(master.integers(args.min_n, args.max_n + 1)), kind=str(master.choice(["euclid", "lattice", "noisy"])),
                 pm=int(master.choice([1, 5, 20, 30])), fi=bool(master.integers(0, 2)), K=int(master.integers(1, args.max_k + 1)),
                 bits=int(master.choice([0, 16, 32, -1, -2])), guides=int(master.integers(1, 3)), seed=int(master.integers(1 << 30)),
                 team=int(master.integers(0, 2)))                 # form of the perturbation phase (1 = team wherever it exists)
        rng = np.random.default_rng(c["seed"])
        n, B = c["n"], 3
        Ds, Gs = zip(*[make_case(rng, n, c["kind"]) for _ in range(B)])
        D = np.stack(Ds)
        guides = np.stack([np.stack(Gs), D][:c["guides"]])
        d = torch.from_numpy(D).cuda()
        gd = torch.from_numpy(np.ascontiguousarray(guides)).cuda()
        init = ops.nearest_neighbor(gd[0].contiguous())
        cost = ops.tour_cost(init, d)
        with ops.gls_team_mode(c["team"]):
            teamed += int(ops.gls_describe_config(n, B, c["bits"])["team"])
            r = ops.gls_run(d, gd, init, cost, perturbation_moves=c["pm"], first_improvement=c["fi"], max_outer_iters=c["K"],
                            trace_cap=1 << 15, want_penalty=True, penalty_bits=c["bits"])
        init_h, cost_h = init.cpu().numpy(), cost.cpu().numpy()
        for b in range(B):
            o = go.guided_local_search(D[b], guides[:, b], init_h[b], cost_h[b], perturbation_moves=c["pm"],
                                       first_improvement=c["fi"], max_outer_iters=c["K"])
            L = o["trace_len"]
            ok = (init_h[b].tolist() == go.nearest_neighbor(guides[0, b]) and int(r.status[b]) == 0
                  and int(r.trace_len[b]) == L and L <= (1 << 15)
                  and np.array_equal(r.trace_cost[b, :L].cpu().numpy().view(np.uint64), o["trace"].view(np.uint64))
                  and r.best_tour[b].cpu().tolist() == o["best_tour"]
                  and np.float64(r.best_cost[b].item()).view(np.uint64) == np.float64(o["best_cost"]).view(np.uint64)
                  and np.array_equal(r.penalty[b].cpu().numpy(), o["penalty"]))
            moves += L
            if not ok:
                bad.append((ci, b, c))
    print(f"{args.cases} cases x 3 instances ({teamed} cases on the team form of the perturbation phase), {moves} accepted moves "
          f"compared, {len(bad)} mismatches, {time.time() - t0:.0f} s")
    for item in bad[:10]:
        print("MISMATCH", item)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
