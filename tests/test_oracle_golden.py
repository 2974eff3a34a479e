"""Pins oracle/gls_oracle.c (the CPU restatement) bit-for-bit against golden vectors captured from
the reference's own Python (oracle/gen_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import gls_oracle as go

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64)).view(np.uint64)


def assert_bits(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    both_nan = np.isnan(a) & np.isnan(b)
    assert np.array_equal(bits(a)[~both_nan], bits(b)[~both_nan])


def check_ops_case(g, prefix=""):
    tour, D = g[prefix + "tour"], g[prefix + "D"]
    n = len(tour) - 1
    assert_bits(go.two_opt_delta_all(tour, D), g[prefix + "two_opt_table"])
    assert_bits(go.relocate_delta_all(tour, D), g[prefix + "relocate_table"])
    for fi in (0, 1):
        for name in ("two_opt_a2a", "relocate_a2a"):
            d, t, _ = getattr(go, name)(tour, D, bool(fi))
            assert_bits(d, g[f"{prefix}{name}_fi{fi}_delta"])
            assert t == g[f"{prefix}{name}_fi{fi}_tour"].tolist()
        for name in ("two_opt_o2a", "relocate_o2a"):
            for i in range(1, n):
                d, t, _ = getattr(go, name)(tour, D, i, bool(fi))
                assert_bits(d, g[f"{prefix}{name}_fi{fi}_delta"][i - 1])
                assert t == g[f"{prefix}{name}_fi{fi}_tour"][i - 1].tolist()


@pytest.mark.parametrize("n", [5, 8, 20, 50, 100])
def test_operators(n):
    check_ops_case(np.load(os.path.join(GOLD, f"ops_n{n}.npz")))


def test_operators_ties_and_isclose():
    g = np.load(os.path.join(GOLD, "ops_ties.npz"))
    for c in range(int(g["n_cases"])):
        check_ops_case(g, prefix=f"c{c}_")


@pytest.mark.parametrize("n", [8, 20, 50, 100])
def test_local_search(n):
    g = np.load(os.path.join(GOLD, f"ls_n{n}.npz"))
    for fi in (0, 1):
        t, c, trace = go.local_search(g[f"fi{fi}_init_tour"], float(g[f"fi{fi}_init_cost"]), g["D"], bool(fi))
        assert t == g[f"fi{fi}_tour"].tolist()
        assert_bits(c, g[f"fi{fi}_cost"])
        assert_bits(trace, g[f"fi{fi}_trace"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "gls_c*.npz"))), ids=os.path.basename)
def test_guided_local_search(path):
    g = np.load(path)
    r = go.guided_local_search(g["D"], g["guides"], g["init_tour"], float(g["init_cost"]),
                               perturbation_moves=int(g["perturbation_moves"]),
                               first_improvement=bool(g["first_improvement"]),
                               max_outer_iters=int(g["K"]))
    assert r["outer_iters"] == int(g["K"])
    assert_bits(r["trace"], g["trace"])
    assert r["best_tour"] == g["best_tour"].tolist()
    assert_bits(r["best_cost"], g["best_cost"])
    assert np.array_equal(r["penalty"], g["penalty"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "gls_imp_c*.npz"))), ids=os.path.basename)
def test_guided_local_search_improvement_record(path):
    """The trajectory of the RETURNED best (algorithms.py:143,190-191), captured from the reference's own local_search
    return values, and the scripts/test.py rule for the start tour (greedy on 'regret_pred' whenever it is among the
    guides, test.py:70-88 -- also when 'weight' comes first)."""
    g = np.load(path)
    assert go.nearest_neighbor(g["nn_guide"]) == g["init_tour"].tolist()
    assert_bits(go.tour_cost(g["init_tour"], g["D"]), g["init_cost"])
    r = go.guided_local_search(g["D"], g["guides"], g["init_tour"], float(g["init_cost"]),
                               perturbation_moves=int(g["perturbation_moves"]), max_outer_iters=int(g["K"]))
    assert_bits(r["trace"], g["trace"])
    assert r["best_tour"] == g["best_tour"].tolist() and np.array_equal(r["penalty"], g["penalty"])
    assert r["imp_len"] == len(g["imp_cost"])
    assert_bits(r["imp_cost"], g["imp_cost"])
    assert np.array_equal(r["imp_iter"], g["imp_iter"])
    assert_bits(r["imp_cost"][-1], r["best_cost"])
    # a buffer that is too small still ends on the terminal entry
    small = go.guided_local_search(g["D"], g["guides"], g["init_tour"], float(g["init_cost"]),
                                   perturbation_moves=int(g["perturbation_moves"]), max_outer_iters=int(g["K"]), imp_cap=2)
    assert small["imp_len"] == len(g["imp_cost"]) and len(small["imp_cost"]) == 2
    assert_bits(small["imp_cost"], [g["imp_cost"][0], g["best_cost"]])
    assert small["imp_iter"].tolist() == [0, int(g["K"])]


def test_misc():
    g = np.load(os.path.join(GOLD, "misc.npz"))
    assert go.nearest_neighbor(g["nn_W_weight"]) == g["nn_tour_weight"].tolist()
    assert go.nearest_neighbor(g["nn_W_regret"]) == g["nn_tour_regret"].tolist()
    assert_bits(go.tour_cost(g["tc_tour"], g["nn_W_weight"]), g["tc_cost"])


def test_timed_improvement_trace_and_perturbation_statistics():
    """Round-4 additions of the oracle (bench.py's CPU leg, kernel-design diagnostics): the timed improvement trace is the plain
    one plus non-decreasing wall-clock stamps; the per-step statistics add up."""
    import ctypes
    from oracle import gls_oracle as go
    rng = np.random.default_rng(5)
    n = 30
    pos = rng.random((n, 2))
    D = np.sqrt(((pos[:, None] - pos[None]) ** 2).sum(-1))
    D = np.triu(D, 1)
    D = D + D.T
    init = go.nearest_neighbor(D)
    c0 = go.tour_cost(init, D)
    L = go.lib()
    i64 = ctypes.c_int64
    L.gls_oracle_perturbation_stats.argtypes = [ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.c_int]
    L.gls_oracle_perturbation_stats(None, None, None, 1)                       # reset
    r = go.guided_local_search(D, D[None], init, c0, perturbation_moves=20, max_outer_iters=200, trace_cap=1 << 16)
    k = len(r["imp_cost"])
    assert k == min(r["imp_len"], 4096) and len(r["imp_time"]) == k and len(r["imp_iter"]) == k
    assert (np.diff(r["imp_time"]) >= 0).all() and r["imp_time"][0] >= 0
    assert (np.diff(r["imp_cost"][:-1]) < 0).all() and r["imp_cost"][-1] == r["best_cost"] == r["imp_cost"][-2]
    assert r["imp_iter"][-1] == 200 and (np.diff(r["imp_iter"]) >= 0).all()
    h, m, st = (i64 * 5)(), (i64 * 4)(), i64()
    L.gls_oracle_perturbation_stats(h, m, ctypes.byref(st), 0)
    h, m = np.array(h[:]), np.array(m[:])
    assert h.sum() == st.value > 0 and (h >= 0).all()
    # every step with a first move at scan q accepted at least that move; moves counted per scan never exceed the steps
    assert (m <= st.value).all() and m.sum() >= st.value - h[4] and (m >= h[:4]).all()
    # the perturbation phase of 200 outer iterations accepted >= 20 moves each (algorithms.py:151: `while moves < perturbation_moves`)
    assert m.sum() >= 200 * 20
