"""CPU checks of oracle/held_karp.c (the exact-optimum gap denominator for TSP<=20): brute force
agreement, tour validity, and consistency with the search oracle (no search result may beat the optimum)."""
import itertools

import numpy as np
import pytest

from oracle import gls_oracle as go
from oracle import held_karp as hk


def euclid(rng, n):
    pos = rng.random((n, 2))
    return np.linalg.norm(pos[:, None] - pos[None], axis=2)


@pytest.mark.parametrize("n", [2, 3, 4, 6, 8])
def test_matches_brute_force(n):
    rng = np.random.default_rng(n)
    for _ in range(3):
        D = euclid(rng, n)
        cost, tour = hk.optimum(D)
        assert tour[0] == 0 and tour[-1] == 0 and sorted(tour[:-1]) == list(range(n))
        assert cost == go.tour_cost(tour, D)                         # summed like gnngls.tour_cost
        brute = min(go.tour_cost([0, *p, 0], D) for p in itertools.permutations(range(1, n)))
        assert abs(cost - brute) <= 1e-12 * brute


def test_asymmetric_matrix_and_range():
    rng = np.random.default_rng(0)
    D = rng.random((7, 7))
    np.fill_diagonal(D, 0.0)
    cost, tour = hk.optimum(D)
    brute = min(go.tour_cost([0, *p, 0], D) for p in itertools.permutations(range(1, 7)))
    assert abs(cost - brute) <= 1e-12
    with pytest.raises(ValueError):
        hk.optimum(np.zeros((22, 22)))


def test_search_never_beats_the_optimum_and_finds_it_on_small_instances():
    rng = np.random.default_rng(5)
    Ds = np.stack([euclid(rng, 13) for _ in range(8)])
    opts = hk.optima(Ds, workers=2)
    for D, opt in zip(Ds, opts):
        t0 = go.nearest_neighbor(D, 0)
        r = go.guided_local_search(D, D[None], t0, go.tour_cost(t0, D), perturbation_moves=20, max_outer_iters=60)
        assert r["best_cost"] >= opt * (1 - 1e-12)
        assert r["best_cost"] <= opt * (1 + 1e-12)                  # guided local search closes TSP13 within 60 iterations
