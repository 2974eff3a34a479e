"""CPU checks of oracle/one_tree.c (Held-Karp 1-tree lower bound, the far side of the optimality-gap bracket bench.py
reports where the exact DP does not reach): the bound never exceeds the exact optimum, is tight on small instances, and
sits below every tour the search finds on larger ones."""
import numpy as np
import pytest

from oracle import gls_oracle as go
from oracle import held_karp as hk
from oracle import one_tree as ot


def euclid(rng, n):
    pos = rng.random((n, 2))
    return np.linalg.norm(pos[:, None] - pos[None], axis=2)


@pytest.mark.parametrize("n", [3, 4, 5, 8, 11, 14])
def test_bound_is_below_the_exact_optimum_and_close_to_it(n):
    rng = np.random.default_rng(100 + n)
    ratios = []
    for _ in range(6):
        D = euclid(rng, n)
        opt, _ = hk.optimum(D)
        for ub in (opt, 1.3 * opt, 10.0 * opt):                     # the upper bound only steers the step size
            lb = ot.lower_bound(D, ub)
            assert lb <= opt * (1 + 1e-9), (n, lb, opt)
            ratios.append(lb / opt)
    assert min(ratios) > 0.9 and np.mean(ratios) > 0.98              # Held-Karp bound: within a few per cent


def test_non_euclidean_and_lattice_matrices():
    rng = np.random.default_rng(5)
    for _ in range(5):
        D = rng.random((9, 9)); D = np.triu(D, 1); D = D + D.T        # symmetric, no triangle inequality
        opt, _ = hk.optimum(D)
        assert ot.lower_bound(D, opt) <= opt * (1 + 1e-9)
        pos = rng.integers(0, 4, size=(10, 2)).astype(float)
        L = np.abs(pos[:, None] - pos[None]).sum(-1) + 1.0
        np.fill_diagonal(L, 0.0)
        opt, _ = hk.optimum(L)
        assert ot.lower_bound(L, opt) <= opt * (1 + 1e-9)


def test_bound_is_below_search_results_at_bench_sizes_and_batches_agree():
    rng = np.random.default_rng(9)
    Ds = np.stack([euclid(rng, 60) for _ in range(4)])
    ubs = []
    for D in Ds:
        t = go.nearest_neighbor(D)
        o = go.guided_local_search(D, D[None], np.asarray(t, dtype=np.int32), go.tour_cost(t, D), perturbation_moves=20,
                                   max_outer_iters=60)
        ubs.append(o["best_cost"])
    lbs = ot.lower_bounds(Ds, ubs, workers=2)
    one = np.array([ot.lower_bound(D, u) for D, u in zip(Ds, ubs)])
    assert np.array_equal(lbs, one)
    gap = (np.array(ubs) / lbs - 1) * 100
    assert (gap >= -1e-9).all() and gap.mean() < 3.0, gap               # typically ~0.7 % at this size
    # more ascent steps never lower the reported bound
    assert ot.lower_bound(Ds[0], ubs[0], max_iters=3000) >= ot.lower_bound(Ds[0], ubs[0], max_iters=300) - 1e-12
