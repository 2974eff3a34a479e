"""GPU tests of the search-progress record (algorithms.py:127-130,180-183 -> test.py:97-117):
the bounded improvement trace the device keeps at any run length, the explicit truncation of the per-move
trace, and the DataFrame scripts/test.py writes at headline length (TSP100, seconds of wall clock, ~1e5+ accepted
moves per instance)."""
import glob
import importlib.util
import itertools
import json
import os
import pickle
import time
import warnings

import networkx as nx
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64)).view(np.uint64)


def dev(x, dtype):
    return torch.as_tensor(np.ascontiguousarray(x)).to(dtype).cuda().contiguous()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "gls_imp_c*.npz"))), ids=os.path.basename)
@pytest.mark.parametrize("trace", [True, False])
def test_improvement_trace_golden(path, trace):
    """Fixtures captured from the reference's own local_search return values (oracle/gen_golden.py gen_progress):
    the device's improvement record is bit for bit the trajectory of the returned best, with and without the
    per-move trace (throughput path: per-move tour_cost deferred)."""
    from gnngls_amd import ops
    g = np.load(path)
    init = ops.nearest_neighbor(dev(g["nn_guide"][None], torch.float64))          # test.py:70-88 start-tour rule
    assert init[0].cpu().tolist() == g["init_tour"].tolist()
    d = dev(g["D"][None], torch.float64)
    cost = ops.tour_cost(init, d)
    assert np.array_equal(bits(cost.cpu().numpy()[0]), bits(g["init_cost"]))
    r = ops.gls_run(d, dev(g["guides"][:, None], torch.float64), init, cost,
                    perturbation_moves=int(g["perturbation_moves"]), max_outer_iters=int(g["K"]),
                    trace_cap=4096 if trace else 0, imp_cap=64, want_penalty=True)
    L = len(g["imp_cost"])
    assert int(r.status[0]) == 0 and int(r.imp_len[0]) == L
    assert np.array_equal(bits(r.imp_cost[0, :L].cpu().numpy()), bits(g["imp_cost"]))
    assert r.imp_iter[0, :L].cpu().tolist() == g["imp_iter"].tolist()
    t = r.imp_time[0, :L].cpu().numpy()
    assert (np.diff(t) >= 0).all() and t[0] > 0 and t[-1] < 30.0
    assert r.best_tour[0].cpu().tolist() == g["best_tour"].tolist()
    assert np.array_equal(bits(r.best_cost[0].item()), bits(g["best_cost"]))
    assert np.array_equal(r.penalty[0].cpu().numpy(), g["penalty"])
    if trace:
        assert not bool(r.trace_truncated[0])
        assert np.array_equal(bits(r.trace_cost[0, :len(g["trace"])].cpu().numpy()), bits(g["trace"]))
    # a buffer smaller than the record: the count is exact and the last slot is the terminal entry
    small = ops.gls_run(d, dev(g["guides"][:, None], torch.float64), init, cost,
                        perturbation_moves=int(g["perturbation_moves"]), max_outer_iters=int(g["K"]), imp_cap=2)
    assert int(small.imp_len[0]) == L
    assert np.array_equal(bits(small.imp_cost[0].cpu().numpy()), bits([g["imp_cost"][0], g["best_cost"]]))
    assert small.imp_iter[0].cpu().tolist() == [0, int(g["K"])]


def test_improvement_trace_vs_oracle_batch():
    """Seeded random batch, two guides, compared with the CPU oracle's record instance by instance."""
    from gnngls_amd import ops
    from gnngls_amd.synthetic import random_instances
    from oracle import gls_oracle as go
    n, B, K = 60, 12, 40
    rng = np.random.default_rng(123)
    D, _ = random_instances(rng, B, n)
    guide = np.maximum(rng.normal(0.05, 0.1, size=D.shape).astype(np.float32).astype(np.float64), 0)
    guide = np.triu(guide, 1) + np.triu(guide, 1).transpose(0, 2, 1)
    guides = np.stack([D, guide])
    d, gd = dev(D, torch.float64), dev(guides, torch.float64)
    init = ops.nearest_neighbor(gd[1].contiguous())
    cost = ops.tour_cost(init, d)
    r = ops.gls_run(d, gd, init, cost, perturbation_moves=20, max_outer_iters=K, imp_cap=128)
    for b in range(B):
        o = go.guided_local_search(D[b], guides[:, b], init[b].cpu().numpy(), cost[b].item(), perturbation_moves=20,
                                   max_outer_iters=K, trace_cap=1, want_penalty=False)
        L = o["imp_len"]
        assert int(r.imp_len[b]) == L
        assert np.array_equal(bits(r.imp_cost[b, :L].cpu().numpy()), bits(o["imp_cost"]))
        assert r.imp_iter[b, :L].cpu().tolist() == o["imp_iter"].tolist()


def test_headline_length_record_ends_on_returned_cost():
    """TSP100, 2 s of wall clock: ~1e5 accepted moves per instance, far beyond any per-move buffer a batch could hold.
    The per-move trace reports its truncation, the improvement record stays a few dozen entries, is strictly
    decreasing, and its terminal entry is the returned best cost bit for bit at the end of the budget."""
    from gnngls_amd import pipeline
    from gnngls_amd.synthetic import random_instances
    n, B, limit = 100, 48, 2.0
    D = torch.from_numpy(random_instances(np.random.default_rng(9), B, n)[0]).cuda()
    r = pipeline.solve_batch(D, guides=("weight",), time_limit=limit, perturbation_moves=20, trace_cap=1 << 10,
                             want_trace_time=True, imp_cap=256)
    moves = r.moves.cpu().numpy()
    assert (moves > 1 << 14).all(), moves.min()                      # the old fixed cap of scripts/test.py
    assert (r.status == 0).all()
    il = r.imp_len.cpu().numpy()
    assert (il >= 2).all() and (il <= 256).all()
    ic, it, ii = r.imp_cost.cpu().numpy(), r.imp_time.cpu().numpy(), r.imp_iter.cpu().numpy()
    best, iters = r.best_cost.cpu().numpy(), r.outer_iters.cpu().numpy()
    for b in range(B):
        L = il[b]
        assert (np.diff(ic[b, :L - 1]) < 0).all()                    # improvements are strict
        assert bits(ic[b, L - 1]) == bits(best[b]) and bits(ic[b, L - 2]) == bits(best[b])
        assert ii[b, L - 1] == iters[b] and (np.diff(ii[b, :L]) >= 0).all()
        assert (np.diff(it[b, :L]) >= 0).all() and limit * 0.9 < it[b, L - 1] < limit + 1.0
    # the per-move prefix is still exact where it exists: costs of the initial descent decrease monotonically
    tc = r.trace_cost.cpu().numpy()
    assert (tc[:, 0] < r.init_cost.cpu().numpy()).all()
    assert r.launch_time.shape == (B,) and (r.launch_time >= r.start_time).all()


def _write_dataset(root, n, count, seed):
    from gnngls_amd import datasets
    from sklearn.preprocessing import MinMaxScaler
    rng = np.random.default_rng(seed)
    data = root / f"tsp{n}"
    data.mkdir()
    scalers = {"features": MinMaxScaler(), "regret": MinMaxScaler()}
    names = []
    for k in range(count):
        pos = rng.random((n, 2))
        G = nx.Graph()
        for v, p in enumerate(pos):
            G.add_node(v, pos=p)
        for i, j in itertools.combinations(G.nodes, 2):
            G.add_edge(i, j, weight=np.linalg.norm(pos[j] - pos[i]), in_solution=False, regret=0.0)
        for v in range(n):
            G.edges[v, (v + 1) % n]["in_solution"] = True
        datasets.set_features(G)
        for key in scalers:
            scalers[key].partial_fit(np.vstack([G.edges[e][key] for e in G.edges]))
        pickle.dump(G, open(data / f"i{k}.pkl", "wb"))
        names.append(f"i{k}.pkl")
    (data / "test.txt").write_text("\n".join(names) + "\n")
    pickle.dump(scalers, open(data / "scalers.pkl", "wb"))
    return data, names


def test_cli_complete_per_move_record(tmp_path):
    """`--full_trace CAP` with CAP above the number of accepted moves: the rows are the reference's per-move record
    verbatim plus ONE terminal row (returned best, end of the search; no other improvement rows mixed in), and the
    `best_cost` column ends on the returned cost."""
    import argparse
    spec = importlib.util.spec_from_file_location("gnngls_cli_test2", os.path.join(ROOT, "scripts", "test.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    from gnngls_amd import datasets
    data, names = _write_dataset(tmp_path, 30, 4, seed=5)
    test_set = datasets.TSPDataset(data / "test.txt")
    args = argparse.Namespace(guides=["weight"], time_limit=0.05, perturbation_moves=20, full_trace=1 << 16)
    records, gaps = cli.solve_block(names, test_set, None, None, args, chunk=64)
    run_dir = tmp_path / "runs"
    cli.write_progress(records, run_dir)
    df = pickle.load(open(next(run_dir.glob("*.pkl")), "rb"))
    last = df.groupby("instance").tail(1).set_index("instance")
    for name, gap in zip(names, gaps):
        assert bits(last.loc[name, "gap"]) == bits(gap)
    costs = df[df["instance"] == names[0]]["cost"].dropna().to_numpy()
    assert len(costs) > 50 and (np.diff(costs) > 0).any()            # per-move rows: perturbation moves raise the cost


@pytest.mark.parametrize("full_trace", [0, 2000])
def test_cli_dataframe_at_headline_length(tmp_path, full_trace):
    """scripts/test.py's own record building at TSP100 with a 2 s budget (default record and `--full_trace CAP` with
    CAP far below the number of accepted moves): per instance, the last `best_cost` of the pickled DataFrame equals the
    returned cost bitwise -- i.e. the gap column ends on the gap the run reports."""
    import argparse
    spec = importlib.util.spec_from_file_location("gnngls_cli_test", os.path.join(ROOT, "scripts", "test.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    from gnngls_amd import datasets
    data, names = _write_dataset(tmp_path, 100, 6, seed=31)
    test_set = datasets.TSPDataset(data / "test.txt")
    args = argparse.Namespace(guides=["weight"], time_limit=2.0, perturbation_moves=20, full_trace=full_trace)
    t0 = time.time()
    records, gaps = cli.solve_block(names, test_set, None, None, args, chunk=64)
    assert time.time() - t0 < 10.0
    run_dir = tmp_path / "runs"
    cli.write_progress(records, run_dir)
    df = pickle.load(open(next(run_dir.glob("*.pkl")), "rb"))
    assert set(["instance", "time", "opt_cost", "cost", "best_cost", "gap", "dt"]) <= set(df.columns)
    last = df.groupby("instance").tail(1).set_index("instance")
    for name, gap in zip(names, gaps):
        assert bits(last.loc[name, "gap"]) == bits(gap)                      # (best_cost / opt_cost - 1) * 100, test.py:104
        assert 1.8 < last.loc[name, "dt"] < 4.0                              # the terminal row sits at the end of the budget
    per_instance = df.groupby("instance").size()
    assert (per_instance <= full_trace + 300).all()                         # bounded: never ~1e5 rows per instance
    assert (df["dt"] >= 0).all()
    if full_trace:
        assert (per_instance > full_trace).all()
        # the first CAP rows are the reference's per-move record: costs of the initial descent decrease
        first = df[df["instance"] == names[0]].dropna(subset=["cost"]).head(10)
        assert (np.diff(first["cost"].to_numpy()) < 0).all()


def test_mirror_progress_when_the_trace_overflows(monkeypatch):
    """gnngls_amd.algorithms.guided_local_search returns the reference's per-move search_progress; when a run accepts
    more moves than the device trace holds (forced here by shrinking the buffer) it warns and continues the list with
    the improvement record, so the last rows always carry the returned cost."""
    import gnngls_amd
    from gnngls_amd import algorithms as alg
    rng = np.random.default_rng(0)
    pos = rng.random((40, 2))
    D = np.linalg.norm(pos[:, None] - pos[None], axis=-1)
    G = nx.Graph()
    G.add_nodes_from(range(40))
    for i, j in itertools.combinations(range(40), 2):
        G.add_edge(i, j, weight=np.float64(D[i, j]))
    init = alg.nearest_neighbor(G, 0)
    cost = gnngls_amd.tour_cost(G, init)
    # exact when it fits
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _, c_full, prog_full = alg.guided_local_search(G, init, cost, time.time() + 100, perturbation_moves=20,
                                                       max_outer_iters=30)
    monkeypatch.setattr(alg, "TRACE_CAP", 50)
    monkeypatch.setattr(alg, "TRACE_PER_SECOND", 0)
    with pytest.warns(RuntimeWarning, match="truncated"):
        tour, c, prog = alg.guided_local_search(G, init, cost, time.time() + 100, perturbation_moves=20, max_outer_iters=30)
    assert bits(c) == bits(c_full) and len(prog_full) > 50
    assert [bits(r["cost"]) for r in prog[:50]] == [bits(r["cost"]) for r in prog_full[:50]]
    assert 50 < len(prog) < len(prog_full) and bits(prog[-1]["cost"]) == bits(c)
    assert all(b["time"] >= a["time"] for a, b in zip(prog, prog[1:]))


def test_watchdog_abort_is_reported():
    """A search the watchdog stops returns best-so-far with status 1; the mirrors and solve_batch say so."""
    import gnngls_amd
    from gnngls_amd import algorithms as alg
    from gnngls_amd import ops
    from gnngls_amd.synthetic import random_instances
    D_host, _ = random_instances(np.random.default_rng(1), 1, 60)
    G = nx.Graph()
    G.add_nodes_from(range(60))
    for i, j in itertools.combinations(range(60), 2):
        G.add_edge(i, j, weight=np.float64(D_host[0, i, j]))
    init = alg.nearest_neighbor(G, 0)
    with pytest.warns(alg.SearchAborted):
        tour, c, _ = alg.guided_local_search(G, init, gnngls_amd.tour_cost(G, init), time.time() + 100,
                                             max_outer_iters=10 ** 9, watchdog_s=0.3)
    assert gnngls_amd.is_valid_tour(G, tour)
    d = torch.from_numpy(D_host).cuda()
    it = ops.nearest_neighbor(d)
    r = ops.gls_run(d, d[None].contiguous(), it, ops.tour_cost(it, d), max_outer_iters=10 ** 9, watchdog_s=0.2)
    assert int(r.status[0]) == ops.STATUS_WATCHDOG
