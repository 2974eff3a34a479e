"""Checker-side exact TSP solver (oracle/bnb_tsp.c: branch and bound on the Held-Karp 1-tree bound) -- the denominator of the
reference's gap (scripts/test.py:62,104) for the sample of bench_data/exact_optima_*.npz: pinned to the exact DP
(oracle/held_karp.c) where that reaches, on Euclidean and integer-tie matrices, from loose and from tight incumbents."""
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bnb_equals_held_karp_dp():
    from oracle import bnb_tsp, held_karp
    rng = np.random.default_rng(5)
    for n in (5, 8, 10, 12, 13, 14):
        for rep in range(5):
            if rep < 3:
                pos = rng.random((n, 2))
                D = np.sqrt(((pos[:, None] - pos[None]) ** 2).sum(-1))
            else:                                              # integer costs: many ties
                D = np.triu(rng.integers(1, 9, size=(n, n)).astype(np.float64), 1)
                D = D + D.T
            opt = held_karp.optima(D[None])[0]
            t = np.concatenate([[0], 1 + rng.permutation(n - 1)])
            ub = float(sum(D[t[i], t[(i + 1) % n]] for i in range(n)))          # a random tour: loose incumbent
            r = bnb_tsp.solve(D, ub)
            assert r["proven"] and abs(r["value"] - opt) <= 1e-9 * max(1.0, opt), (n, rep, opt, r)
            if r["tour"] is not None:                          # a tour the search found: a permutation of that length
                tt = r["tour"]
                assert sorted(tt) == list(range(n))
                assert abs(sum(D[tt[i], tt[(i + 1) % n]] for i in range(n)) - r["value"]) <= 1e-9 * max(1.0, opt)
            r2 = bnb_tsp.solve(D, opt)                         # the optimum as incumbent: certified, nothing shorter found
            assert r2["proven"] and r2["value"] == opt and r2["tour"] is None
    # node limit: an unproven answer says so and still returns a valid upper bound
    pos = rng.random((40, 2))
    D = np.sqrt(((pos[:, None] - pos[None]) ** 2).sum(-1))
    ub = float(sum(D[i, (i + 1) % 40] for i in range(40)))
    r = bnb_tsp.solve(D, ub, max_nodes=0)
    assert not r["proven"] and r["value"] <= ub


def test_committed_exact_optima_files():
    """bench_data/exact_optima_*.npz (made by scripts/make_exact_optima.py in the build container): data only; an optimum never
    exceeds the best-known length of the same instance, and the Held-Karp 1-tree bound never exceeds a proven optimum."""
    from oracle import one_tree
    from gnngls_amd.synthetic import random_instances
    files = glob.glob(os.path.join(ROOT, "bench_data", "exact_optima_tsp*_seed*.npz"))
    assert files
    for f in files:
        z = np.load(f, allow_pickle=False)
        n, seed = int(z["n"]), int(z["seed"])
        bk = np.load(os.path.join(ROOT, "bench_data", f"best_known_tsp{n}_seed{seed}.npz"))["block0"]
        idx = z["index"]
        assert (z["optimum"] <= bk[idx] * (1 + 1e-12)).all() and np.array_equal(z["best_known"], bk[idx])
        assert z["proven"].sum() >= 1
        D, _ = random_instances(np.random.default_rng(seed), 1024, n)
        for i in idx[z["proven"]][:3]:
            lb = one_tree.lower_bound(D[i], float(z["optimum"][list(idx).index(i)]), max_iters=300)
            assert lb <= z["optimum"][list(idx).index(i)] * (1 + 1e-9)
